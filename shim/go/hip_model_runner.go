// Package hip is the cgo shim that puts libnvllm_hip.so behind nano-vllm-go's ModelRunner interface
// (nanovllm/model_runner.go:9-16).  It has the surface of TensorModelRunner
// (nanovllm/tensor_model_runner.go:11-117): NewHipModelRunner, SetSamplingParams,
// SetSamplingParamsWithRepetition, Run, ClearCache, ClearAllCaches, Close.
//
// Drop this file into the reference tree as nanovllm/hip/hip_model_runner.go and build with
//
//	CGO_CFLAGS="-I$REPO/include" \
//	CGO_LDFLAGS="-L$REPO/nano-vllm-go_amd/lib -lnvllm_hip -Wl,-rpath,$REPO/nano-vllm-go_amd/lib" go build ./...
//
// STATUS: reviewed against include/nvllm.h, NOT built — neither this image nor the GPU box has a Go
// toolchain.  The same call sequence runs in C (examples/ask_greedy.c), in the library's own runner
// (nvl_runner_run, csrc/nvllm.hip) and through ctypes (nano-vllm-go_amd/runner.py), all tested on the GPU.
package hip

/*
#include <stdint.h>
#include <stdlib.h>
#include "nvllm.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math/rand"
	"runtime"
	"sync"
	"unsafe"

	"nano-vllm-go/nanovllm"
	"nano-vllm-go/purego/tensor"
)

// Options sizes the device side.  The zero value is NOT valid: MaxSeqs must be >= 1 (nvl_create refuses 0).
type Options struct {
	Device         int  // HIP device ordinal
	MaxSeqs        int  // KV slots; Config.MaxNumSeqs is the natural value (nanovllm/config.go:30)
	MaxBatchTokens int  // >= Config.MaxNumBatchedTokens; 0 = the model's MaxSeqLen
	Greedy         bool // return the device argmax (cmd/ask/main.go:389-402) instead of sampling
	DeviceSampling bool // run tensor.SampleWithHistory on the device (only token ids cross PCIe)
}

// HipModelRunner implements nanovllm.ModelRunner on one MI355X.
type HipModelRunner struct {
	h              *C.nvl_model
	vocab          int
	mu             sync.Mutex             // the handle is not re-entrant (one HIP stream)
	sampling       *tensor.SamplingParams // defaultSampling of the reference runner
	greedy         bool
	deviceSampling bool
}

var _ nanovllm.ModelRunner = (*HipModelRunner)(nil)

func b2i(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}

func attnEnum(a tensor.AttentionType) (C.int32_t, error) {
	switch a {
	case tensor.AttentionMHA:
		return C.NVL_ATTN_MHA, nil
	case tensor.AttentionMQA:
		return C.NVL_ATTN_MQA, nil
	case tensor.AttentionGQA:
		return C.NVL_ATTN_GQA, nil
	}
	return 0, fmt.Errorf("nvllm: unsupported attention type %q", a)
}

func normEnum(n tensor.NormType) C.int32_t {
	if n == tensor.NormRMS {
		return C.NVL_NORM_RMS
	}
	return C.NVL_NORM_LAYER
}

func posEnum(p tensor.PositionType) (C.int32_t, error) {
	switch p {
	case tensor.PositionLearned:
		return C.NVL_POS_LEARNED, nil
	case tensor.PositionRoPE:
		return C.NVL_POS_ROPE, nil
	case tensor.PositionNoPE:
		return C.NVL_POS_NONE, nil
	}
	return 0, fmt.Errorf("nvllm: unsupported position type %q", p)
}

func actEnum(a tensor.ActivationType) C.int32_t {
	if a == tensor.ActivationSwiGLU {
		return C.NVL_ACT_SWIGLU
	}
	return C.NVL_ACT_GELU
}

func blockEnum(b tensor.BlockStyle) C.int32_t {
	if b == tensor.BlockParallel {
		return C.NVL_BLOCK_PARALLEL
	}
	return C.NVL_BLOCK_SEQUENTIAL
}

func createErr() error { return fmt.Errorf("nvllm: %s", C.GoString(C.nvl_last_error(nil))) }

func (r *HipModelRunner) err() error {
	return fmt.Errorf("nvllm: %s", C.GoString(C.nvl_last_error(r.h)))
}

// NewHipModelRunner = NewTensorModelRunner (tensor_model_runner.go:21-33): the Go loader still reads
// config.json + safetensors (generic_loader.go:1016), then each tensor is handed to the device once.
// (A deployment that does not need the fp32 Go model uses NewHipModelRunnerFromDir below.)
func NewHipModelRunner(modelDir string, o Options) (*HipModelRunner, error) {
	model, err := tensor.LoadModelFromDirectory(modelDir)
	if err != nil {
		return nil, fmt.Errorf("failed to load model: %w", err)
	}
	c := model.Config
	at, err := attnEnum(c.AttentionType)
	if err != nil {
		return nil, err
	}
	pt, err := posEnum(c.PositionType)
	if err != nil {
		return nil, err
	}
	var cfg C.nvl_model_config
	cfg.vocab_size = C.int32_t(c.VocabSize)
	cfg.hidden = C.int32_t(c.Hidden)
	cfg.num_layers = C.int32_t(c.NumLayers)
	cfg.num_heads = C.int32_t(c.NumHeads)
	cfg.num_kv_heads = C.int32_t(c.NumKVHeads)
	cfg.head_dim = C.int32_t(c.HeadDim)
	cfg.ffn_dim = C.int32_t(c.FFNDim)
	cfg.max_seq_len = C.int32_t(c.MaxSeqLen)
	cfg.attention_type = at
	cfg.norm_type = normEnum(c.NormType)
	cfg.position_type = pt
	cfg.activation_type = actEnum(c.ActivationType)
	cfg.block_style = blockEnum(c.BlockStyle)
	cfg.rope_base = C.double(c.RoPEBase)
	cfg.norm_eps = C.float(c.NormEps)
	cfg.tied_embedding = b2i(c.TiedEmbedding)
	cfg.use_moe = b2i(c.UseMoE)
	cfg.num_experts = C.int32_t(c.NumExperts)
	cfg.num_experts_per_tok = C.int32_t(c.NumExpertsPerTok)
	cfg.embedding_multiplier = C.float(c.EmbeddingMultiplier)
	cfg.attention_multiplier = C.float(c.AttentionMultiplier)
	cfg.residual_multiplier = C.float(c.ResidualMultiplier)
	cfg.logits_scaling = C.float(c.LogitsScaling)
	// Granite-4 hybrid (config.go:100-115): which layers are Mamba2 blocks, and their geometry
	for li, kind := range c.HybridLayers {
		if (kind == "mamba" || kind == "mamba2") && li < 128 {
			cfg.mamba_layer_mask[li/64] |= C.uint64_t(1) << uint(li%64)
			cfg.mamba_expand = C.int32_t(c.Mamba2Expand)
			cfg.mamba_state_size = C.int32_t(c.Mamba2StateSize)
			cfg.mamba_num_heads = C.int32_t(c.Mamba2NumHeads)
			cfg.mamba_head_dim = C.int32_t(c.Mamba2HeadDim)
			cfg.mamba_n_groups = C.int32_t(c.Mamba2NGroups)
			cfg.mamba_conv_kernel = C.int32_t(c.Mamba2ConvKernel)
		}
	}

	r, err := create(&cfg, o)
	if err != nil {
		return nil, err
	}
	r.vocab = c.VocabSize
	fail := func(e error) (*HipModelRunner, error) { C.nvl_destroy(r.h); r.h = nil; return nil, e }

	// tensors are fp32 in the reference's post-load layout ([in,out] for projections) after LoadModel
	up := func(kind C.int, layer int, t *tensor.Tensor) error {
		if t == nil || len(t.Data) == 0 {
			return nil
		}
		rows, cols := int64(len(t.Data)), int64(1)
		switch len(t.Shape) {
		case 2:
			rows, cols = int64(t.Shape[0]), int64(t.Shape[1])
		case 3: // MoE experts [E, out, in] (generic_loader.go:578-586: loaded without transpose)
			rows, cols = int64(t.Shape[0]*t.Shape[1]), int64(t.Shape[2])
		}
		// the library copies out of the buffer before returning (synchronous upload): a Go pointer for
		// the duration of the call is within the cgo rules
		layout := C.int(C.NVL_LAYOUT_IN_OUT)
		if kind == C.NVL_T_MAMBA_IN_PROJ || kind == C.NVL_T_MAMBA_OUT_PROJ {
			layout = C.NVL_LAYOUT_OUT_IN // loadMamba2 keeps the PyTorch layout (generic_loader.go:461-512)
		}
		rc := C.nvl_upload_tensor(r.h, kind, C.int(layer), unsafe.Pointer(&t.Data[0]), C.NVL_DTYPE_F32,
			C.int64_t(rows), C.int64_t(cols), layout)
		runtime.KeepAlive(t)
		if rc != 0 {
			return r.err()
		}
		return nil
	}
	type item struct {
		kind  C.int
		layer int
		t     *tensor.Tensor
	}
	var items []item
	add := func(kind C.int, layer int, t *tensor.Tensor) { items = append(items, item{kind, layer, t}) }
	norm := func(kw, kb C.int, layer int, ln *tensor.LayerNormLayer) {
		if ln != nil {
			add(kw, layer, ln.Weight)
			add(kb, layer, ln.Bias)
		}
	}
	// model-level tensors (generic_model.go:4-19)
	add(C.NVL_T_TOK_EMB, 0, model.TokenEmbedding)
	add(C.NVL_T_POS_EMB, 0, model.PosEmbedding)
	if !c.TiedEmbedding {
		add(C.NVL_T_LM_HEAD, 0, model.LMHead)
	}
	norm(C.NVL_T_FINAL_NORM_W, C.NVL_T_FINAL_NORM_B, 0, model.LNFinal)
	for i, b := range model.Blocks { // generic_model.go:22-31
		ln1 := b.AttnLN
		if ln1 == nil {
			ln1 = b.InputLN
		}
		norm(C.NVL_T_ATTN_NORM_W, C.NVL_T_ATTN_NORM_B, i, ln1)
		norm(C.NVL_T_FFN_NORM_W, C.NVL_T_FFN_NORM_B, i, b.FFNLN)
		switch a := b.Attention.(type) {
		case *tensor.GroupedQueryAttention:
			add(C.NVL_T_WQ, i, a.QWeight)
			add(C.NVL_T_WK, i, a.KWeight)
			add(C.NVL_T_WV, i, a.VWeight)
			add(C.NVL_T_WO, i, a.OutWeight)
		case *tensor.MultiHeadAttention:
			add(C.NVL_T_WQ, i, a.QWeight)
			add(C.NVL_T_WK, i, a.KWeight)
			add(C.NVL_T_WV, i, a.VWeight)
			add(C.NVL_T_WO, i, a.OutWeight)
			add(C.NVL_T_BQ, i, a.QBias)
			add(C.NVL_T_BK, i, a.KBias)
			add(C.NVL_T_BV, i, a.VBias)
			add(C.NVL_T_BO, i, a.OutBias)
		case *tensor.MultiQueryAttention:
			add(C.NVL_T_WQ, i, a.QWeight)
			add(C.NVL_T_WKV, i, a.KVWeight)
			add(C.NVL_T_WO, i, a.OutWeight)
		case *tensor.Mamba2Layer: // mamba2.go:9-27; the state lives per sequence on the device (DESIGN.md §9 f-4)
			add(C.NVL_T_MAMBA_IN_PROJ, i, a.InProj)
			add(C.NVL_T_MAMBA_CONV_W, i, a.ConvWeight)
			add(C.NVL_T_MAMBA_CONV_B, i, a.ConvBias)
			add(C.NVL_T_MAMBA_A_LOG, i, a.ALog)
			add(C.NVL_T_MAMBA_D, i, a.D)
			add(C.NVL_T_MAMBA_DT_BIAS, i, a.DeltaBias)
			add(C.NVL_T_MAMBA_NORM, i, a.Norm)
			add(C.NVL_T_MAMBA_OUT_PROJ, i, a.OutProj)
		default:
			return fail(fmt.Errorf("nvllm: layer %d: attention layer type %T is not on the device path", i, b.Attention))
		}
		if b.MoE != nil {
			add(C.NVL_T_ROUTER, i, b.MoE.Router)
			add(C.NVL_T_MOE_IN, i, b.MoE.InputLinear)
			add(C.NVL_T_MOE_OUT, i, b.MoE.OutputLinear)
		} else if b.FFN != nil {
			add(C.NVL_T_W1, i, b.FFN.W1)
			add(C.NVL_T_B1, i, b.FFN.B1)
			add(C.NVL_T_W2, i, b.FFN.W2)
			add(C.NVL_T_B2, i, b.FFN.B2)
		}
	}
	for _, it := range items {
		if err := up(it.kind, it.layer, it.t); err != nil {
			return fail(err)
		}
	}
	if rc := C.nvl_finalize(r.h); rc != 0 {
		return fail(r.err())
	}
	return r, nil // the fp32 Go tensors can be dropped now: the weights live on the device
}

// NewHipModelRunnerFromDir skips the fp32 Go model entirely: config.json and the safetensors file(s) are
// read by the library (mmap -> device; generic_loader.go:184-265, 808-1163 restated in csrc/loader.h).
func NewHipModelRunnerFromDir(modelDir string, o Options) (*HipModelRunner, error) {
	cdir := C.CString(modelDir)
	defer C.free(unsafe.Pointer(cdir))
	ccfg := C.CString(modelDir + "/config.json")
	defer C.free(unsafe.Pointer(ccfg))
	var cfg C.nvl_model_config
	if rc := C.nvl_load_config_json(ccfg, &cfg); rc != 0 {
		return nil, createErr()
	}
	r, err := create(&cfg, o)
	if err != nil {
		return nil, err
	}
	r.vocab = int(cfg.vocab_size)
	if rc := C.nvl_load_safetensors(r.h, cdir); rc != 0 {
		e := r.err()
		C.nvl_destroy(r.h)
		return nil, e
	}
	if rc := C.nvl_finalize(r.h); rc != 0 {
		e := r.err()
		C.nvl_destroy(r.h)
		return nil, e
	}
	return r, nil
}

func create(cfg *C.nvl_model_config, o Options) (*HipModelRunner, error) {
	if o.MaxSeqs < 1 {
		return nil, errors.New("nvllm: Options.MaxSeqs must be >= 1 (use Config.MaxNumSeqs)")
	}
	var opts C.nvl_runtime_opts
	opts.device = C.int32_t(o.Device)
	opts.precision = C.NVL_PRECISION_BF16
	opts.max_seqs = C.int32_t(o.MaxSeqs)
	opts.max_batch_tokens = C.int32_t(o.MaxBatchTokens)
	opts.tp_rank = 0
	opts.tp_size = 1
	r := &HipModelRunner{sampling: tensor.DefaultSamplingParams(), greedy: o.Greedy, deviceSampling: o.DeviceSampling}
	if rc := C.nvl_create(cfg, &opts, &r.h); rc != 0 {
		return nil, createErr()
	}
	return r, nil
}

// cbuf is a block of C memory viewed as a Go slice; nothing the library sees is Go-managed memory.
func cInt32s(n int) []C.int32_t {
	if n == 0 {
		n = 1
	}
	return unsafe.Slice((*C.int32_t)(C.malloc(C.size_t(4*n))), n)
}

// Run replaces TensorModelRunner.Run (tensor_model_runner.go:55-97): ONE device call for the whole
// scheduler batch instead of a serial loop over sequences.
//
// Slot lifetime: the engine never tells a runner that a sequence finished (ClearCache has no caller,
// llm_engine.go:35-37), and the reference's cache map grows until Close.  The library has MaxSeqs KV
// slots; when all are taken nvl_runner_run evicts the least-recently-forwarded sequence that is not in
// the current batch.  A later decode of an evicted (or unknown, or stale) sequence is transparently
// re-prefilled from its full TokenIDs — the same recovery pre-emption relies on (scheduler.go:115-119).
func (r *HipModelRunner) Run(seqs []*nanovllm.Sequence, isPrefill bool) ([]int, error) {
	n := len(seqs)
	if n == 0 {
		return []int{}, nil
	}
	r.mu.Lock()
	defer r.mu.Unlock()
	if r.h == nil {
		return nil, errors.New("nvllm: runner is closed")
	}
	total := 0
	for _, s := range seqs {
		if len(s.TokenIDs) == 0 {
			return nil, fmt.Errorf("nvllm: sequence %d has no tokens", s.SeqID)
		}
		total += len(s.TokenIDs)
	}
	// flatten into C memory: no Go pointer is retained by the library after the call returns
	ids := unsafe.Slice((*C.int64_t)(C.malloc(C.size_t(8*n))), n)
	defer C.free(unsafe.Pointer(&ids[0]))
	ptrs := unsafe.Slice((**C.int32_t)(C.malloc(C.size_t(uintptr(n)*unsafe.Sizeof(uintptr(0))))), n)
	defer C.free(unsafe.Pointer(&ptrs[0]))
	lens := cInt32s(n)
	defer C.free(unsafe.Pointer(&lens[0]))
	out := cInt32s(n)
	defer C.free(unsafe.Pointer(&out[0]))
	toks := cInt32s(total) // every history back to back; ptrs[i] points into it
	defer C.free(unsafe.Pointer(&toks[0]))
	off := 0
	for i, s := range seqs {
		ids[i] = C.int64_t(s.SeqID)
		lens[i] = C.int32_t(len(s.TokenIDs))
		ptrs[i] = &toks[off]
		for _, t := range s.TokenIDs { // Sequence.TokenIDs is []int (sequence.go:18)
			toks[off] = C.int32_t(t)
			off++
		}
	}
	pre := C.int(0)
	if isPrefill {
		pre = 1
	}
	tokens := make([]int, n)

	if !r.greedy && r.deviceSampling {
		// tensor.SampleWithHistory on the device (nvl_runner_run_sampled): only ids come back.  The draws stay on
		// the host: one rand.Float32() per sequence, in sequence order — the calls sampleMultinomial would make
		// (sampling.go:205), so math/rand's global stream is what it would have been.
		us := unsafe.Slice((*C.float)(C.malloc(C.size_t(4*n))), n)
		defer C.free(unsafe.Pointer(&us[0]))
		for i := range us {
			us[i] = C.float(rand.Float32())
		}
		var sp C.nvl_sampling_params
		sp.temperature = C.float(r.sampling.Temperature)
		sp.top_p = C.float(r.sampling.TopP)
		sp.top_k = C.int32_t(r.sampling.TopK)
		sp.repetition_penalty = C.float(r.sampling.RepetitionPenalty)
		if rc := C.nvl_runner_run_sampled(r.h, C.int(n), &ids[0], &ptrs[0], &lens[0], pre, &sp, &us[0], &out[0]); rc != 0 {
			return nil, r.err()
		}
		for i := range tokens {
			tokens[i] = int(out[i])
		}
		return tokens, nil
	}

	var logits *C.float
	if !r.greedy {
		logits = (*C.float)(C.malloc(C.size_t(4 * n * r.vocab)))
		defer C.free(unsafe.Pointer(logits))
	}
	if rc := C.nvl_runner_run(r.h, C.int(n), &ids[0], &ptrs[0], &lens[0], pre, &out[0], logits); rc != 0 {
		return nil, r.err() // LLMEngine.Step wraps and returns it (llm_engine.go:66-68)
	}
	if r.greedy {
		for i := range tokens {
			tokens[i] = int(out[i])
		}
		return tokens, nil
	}
	// exactly what the reference does with the last-row logits (tensor_model_runner.go:89-93)
	all := unsafe.Slice((*float32)(unsafe.Pointer(logits)), n*r.vocab)
	for i := range tokens {
		row := make([]float32, r.vocab) // SampleWithHistory may modify its input (repetition penalty)
		copy(row, all[i*r.vocab:(i+1)*r.vocab])
		tokens[i] = tensor.SampleWithHistory(row, seqs[i].TokenIDs, r.sampling)
	}
	return tokens, nil
}

// SetSamplingParams: tensor_model_runner.go:35-43.
func (r *HipModelRunner) SetSamplingParams(temperature float32, topP float32, topK int) {
	r.mu.Lock()
	defer r.mu.Unlock()
	r.sampling = &tensor.SamplingParams{Temperature: temperature, TopP: topP, TopK: topK, RepetitionPenalty: 1.2}
}

// SetSamplingParamsWithRepetition: tensor_model_runner.go:45-53.
func (r *HipModelRunner) SetSamplingParamsWithRepetition(temperature float32, topP float32, topK int, repetitionPenalty float32) {
	r.mu.Lock()
	defer r.mu.Unlock()
	r.sampling = &tensor.SamplingParams{Temperature: temperature, TopP: topP, TopK: topK, RepetitionPenalty: repetitionPenalty}
}

// ClearCache: tensor_model_runner.go:100-104 (optional here: slots are reclaimed by LRU eviction).
func (r *HipModelRunner) ClearCache(seqID int64) {
	r.mu.Lock()
	defer r.mu.Unlock()
	if r.h != nil {
		C.nvl_seq_close(r.h, C.int64_t(seqID))
	}
}

// ClearAllCaches: tensor_model_runner.go:107-111.
func (r *HipModelRunner) ClearAllCaches() {
	r.mu.Lock()
	defer r.mu.Unlock()
	if r.h != nil {
		C.nvl_seq_close_all(r.h)
	}
}

// Close: tensor_model_runner.go:114-117.
func (r *HipModelRunner) Close() error {
	r.mu.Lock()
	defer r.mu.Unlock()
	if r.h != nil {
		C.nvl_destroy(r.h)
		r.h = nil
	}
	return nil
}
