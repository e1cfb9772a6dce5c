"""Host-side mirror of nanovllm.TensorModelRunner (nanovllm/tensor_model_runner.go:11-117) and of the
fields of nanovllm.Sequence a runner reads (nanovllm/sequence.go:15-28), over nvl_runner_run.

In a Go build this file is replaced by the cgo shim in INTEGRATION.md; here it lets the parity tests
read like tests of the reference's own runner."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L
from .model import HipTransformerModel


@dataclass
class Sequence:
    """The part of nanovllm.Sequence (sequence.go:15-28) a ModelRunner touches."""
    seq_id: int
    token_ids: list = field(default_factory=list)
    temperature: float = 1.0
    num_cached_tokens: int = 0                     # sequence.go:22, set by BlockManager.Allocate on a prefix-cache hit
    block_table: list = field(default_factory=list)   # sequence.go:23
    block_size: int = 256                          # sequence.go:51

    def append_token(self, tok: int):          # sequence.go AppendToken
        self.token_ids.append(int(tok))

    def __len__(self):
        return len(self.token_ids)


class HipModelRunner:
    """ModelRunner{Run, Close} (nanovllm/model_runner.go:9-16) with TensorModelRunner's extra methods.

    run() returns greedy token ids (cmd/ask/main.go:389-402), optionally with the last-row logits for host-side
    sampling.  run_sampled() is the reference's Run proper: tensor.SampleWithHistory(logits, seq.TokenIDs,
    defaultSampling) (tensor_model_runner.go:93) on the device, with the rand.Float32() draws supplied by the
    caller (one per sequence, in order) so the host keeps its own RNG stream."""

    def __init__(self, model: HipTransformerModel):
        self.model = model
        self.lib = model.lib
        self.default_sampling = dict(temperature=1.0, top_p=1.0, top_k=0, repetition_penalty=1.2)

    def set_sampling_params(self, temperature, top_p, top_k):                 # tensor_model_runner.go:35-43
        self.default_sampling = dict(temperature=temperature, top_p=top_p, top_k=top_k, repetition_penalty=1.2)

    def set_sampling_params_with_repetition(self, temperature, top_p, top_k, repetition_penalty):   # :45-53
        self.default_sampling = dict(temperature=temperature, top_p=top_p, top_k=top_k,
                                     repetition_penalty=repetition_penalty)

    def run(self, seqs, is_prefill: bool, return_logits: bool = False):        # tensor_model_runner.go:55-97
        n = len(seqs)
        if n == 0:
            return ([], None) if return_logits else []
        ids = np.asarray([s.seq_id for s in seqs], np.int64)
        arrs = [np.ascontiguousarray(s.token_ids, dtype=np.int32) for s in seqs]
        ptrs = (C.c_void_p * n)(*[a.ctypes.data_as(C.c_void_p).value for a in arrs])
        lens = np.asarray([a.size for a in arrs], np.int32)
        out = np.empty(n, np.int32)
        logits = np.empty((n, self.model.V), np.float32) if return_logits else None
        L.check(self.lib.nvl_runner_run(self.model.h, n, ids.ctypes.data_as(C.c_void_p), ptrs,
                                        lens.ctypes.data_as(C.c_void_p), int(bool(is_prefill)),
                                        out.ctypes.data_as(C.c_void_p),
                                        None if logits is None else logits.ctypes.data_as(C.c_void_p)),
                self.model.h)
        toks = [int(t) for t in out]
        return (toks, logits) if return_logits else toks

    def run_sampled(self, seqs, is_prefill: bool, uniforms):                  # tensor_model_runner.go:55-97 incl. :93
        n = len(seqs)
        if n == 0:
            return []
        ids = np.asarray([s.seq_id for s in seqs], np.int64)
        arrs = [np.ascontiguousarray(s.token_ids, dtype=np.int32) for s in seqs]
        ptrs = (C.c_void_p * n)(*[a.ctypes.data_as(C.c_void_p).value for a in arrs])
        lens = np.asarray([a.size for a in arrs], np.int32)
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        assert u.size == n
        out = np.empty(n, np.int32)
        sp = L.sampling_params(**self.default_sampling)
        L.check(self.lib.nvl_runner_run_sampled(self.model.h, n, ids.ctypes.data_as(C.c_void_p), ptrs,
                                                lens.ctypes.data_as(C.c_void_p), int(bool(is_prefill)), C.byref(sp),
                                                u.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)),
                self.model.h)
        return [int(t) for t in out]

    def clear_cache(self, seq_id: int):                                        # :100-104
        self.model.seq_close(seq_id)

    def clear_all_caches(self):                                                # :107-111
        self.model.seq_close_all()

    def close(self):                                                           # :114-117
        self.clear_all_caches()
        return None


class HipPagedModelRunner:
    """A ModelRunner that honours what the scheduler's BlockManager computed (block_manager.go:128-263) — the
    reference's runners ignore Sequence.BlockTable / NumCachedTokens.  The model must be created with
    kv_num_blocks = the block manager's pool size; the KV cache then IS that pool: prefix-cache hits skip their
    prefill work, finished sequences free memory the moment the block manager reuses their blocks."""

    def __init__(self, model: HipTransformerModel):
        self.model = model
        self.lib = model.lib

    def run(self, seqs, is_prefill: bool, return_logits: bool = False):
        n = len(seqs)
        if n == 0:
            return ([], None) if return_logits else []
        arrs = [np.ascontiguousarray(s.token_ids, dtype=np.int32) for s in seqs]
        tbls = [np.ascontiguousarray(s.block_table, dtype=np.int32) for s in seqs]
        ptrs = (C.c_void_p * n)(*[a.ctypes.data_as(C.c_void_p).value for a in arrs])
        tptrs = (C.c_void_p * n)(*[a.ctypes.data_as(C.c_void_p).value for a in tbls])
        lens = np.asarray([a.size for a in arrs], np.int32)
        tlens = np.asarray([a.size for a in tbls], np.int32)
        cached = np.asarray([s.num_cached_tokens for s in seqs], np.int32)
        out = np.empty(n, np.int32)
        logits = np.empty((n, self.model.V), np.float32) if return_logits else None
        L.check(self.lib.nvl_runner_run_paged(self.model.h, n, ptrs, lens.ctypes.data_as(C.c_void_p),
                                              cached.ctypes.data_as(C.c_void_p), tptrs, tlens.ctypes.data_as(C.c_void_p),
                                              int(bool(is_prefill)), out.ctypes.data_as(C.c_void_p),
                                              None if logits is None else logits.ctypes.data_as(C.c_void_p)),
                self.model.h)
        toks = [int(t) for t in out]
        return (toks, logits) if return_logits else toks

    def close(self):
        return None
