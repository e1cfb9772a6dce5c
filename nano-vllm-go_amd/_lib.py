"""ctypes binding of libnvllm_hip.so (include/nvllm.h).  There is no fallback: if the HIP library is
missing this module raises, and if no GPU is visible every compute call returns NVL_ERR_NO_DEVICE."""
from __future__ import annotations

import ctypes as C
import os
import re
import sys
from pathlib import Path

_PKG = Path(__file__).resolve().parent
# NVLLM_LIB: another build of the same library (the diagnostic build libnvllm_hip_diag.so of csrc/Makefile's `diag` target)
LIB_PATH = Path(os.environ["NVLLM_LIB"]).resolve() if os.environ.get("NVLLM_LIB") else _PKG / "lib" / "libnvllm_hip.so"
HEADER = _PKG.parent / "include" / "nvllm.h"

SLOTS = [
    "tok_emb", "pos_emb", "lm_head", "final_norm_w", "final_norm_b",
    "attn_norm_w", "attn_norm_b", "ffn_norm_w", "ffn_norm_b",
    "wq", "wk", "wv", "wkv", "wo", "bq", "bk", "bv", "bo",
    "w1", "b1", "w2", "b2", "router", "moe_in", "moe_out",
    "mamba_in_proj", "mamba_conv_w", "mamba_conv_b", "mamba_a_log", "mamba_d", "mamba_dt_bias", "mamba_norm", "mamba_out_proj",
]
SLOT_ID = {n: i for i, n in enumerate(SLOTS)}
ONE_D = {"final_norm_w", "final_norm_b", "attn_norm_w", "attn_norm_b", "ffn_norm_w", "ffn_norm_b",
         "bq", "bk", "bv", "bo", "b1", "b2",
         "mamba_conv_w", "mamba_conv_b", "mamba_a_log", "mamba_d", "mamba_dt_bias", "mamba_norm"}
OUT_IN_SLOTS = {"mamba_in_proj", "mamba_out_proj"}      # kept in the checkpoint's [out, in] layout by the reference's loader

ATTN = {"mha": 0, "mqa": 1, "gqa": 2}
NORM = {"layernorm": 0, "rmsnorm": 1}
POS = {"learned": 0, "rope": 1, "nope": 2}
ACT = {"gelu": 0, "swiglu": 1}
BLOCK = {"sequential": 0, "parallel": 1}
PRECISION = {"bf16": 0, "f32": 1}
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
LAYOUT_IN_OUT, LAYOUT_OUT_IN = 0, 1
FWD_ALL_LOGITS = 1

STATUS = {0: "NVL_OK", -1: "NVL_ERR_INVALID", -2: "NVL_ERR_POSITION", -3: "NVL_ERR_UNKNOWN_SEQ",
          -4: "NVL_ERR_NO_SLOT", -5: "NVL_ERR_OOM", -6: "NVL_ERR_HIP", -7: "NVL_ERR_STATE",
          -8: "NVL_ERR_NO_DEVICE"}


class NvlError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"nvllm: {STATUS.get(code, code)}: {msg}")
        self.code = code


class ModelConfigC(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden", C.c_int32), ("num_layers", C.c_int32),
        ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32), ("head_dim", C.c_int32),
        ("ffn_dim", C.c_int32), ("max_seq_len", C.c_int32),
        ("attention_type", C.c_int32), ("norm_type", C.c_int32), ("position_type", C.c_int32),
        ("activation_type", C.c_int32), ("block_style", C.c_int32),
        ("rope_base", C.c_double), ("norm_eps", C.c_float), ("tied_embedding", C.c_int32),
        ("use_moe", C.c_int32), ("num_experts", C.c_int32), ("num_experts_per_tok", C.c_int32),
        ("embedding_multiplier", C.c_float), ("attention_multiplier", C.c_float),
        ("residual_multiplier", C.c_float), ("logits_scaling", C.c_float),
        ("mamba_expand", C.c_int32), ("mamba_state_size", C.c_int32), ("mamba_num_heads", C.c_int32),
        ("mamba_head_dim", C.c_int32), ("mamba_n_groups", C.c_int32), ("mamba_conv_kernel", C.c_int32),
        ("mamba_layer_mask", C.c_uint64 * 2),
    ]


class RuntimeOptsC(C.Structure):
    _fields_ = [("device", C.c_int32), ("precision", C.c_int32), ("max_seqs", C.c_int32),
                ("max_batch_tokens", C.c_int32), ("tp_rank", C.c_int32), ("tp_size", C.c_int32),
                ("tp_force_single", C.c_int32), ("kv_num_blocks", C.c_int32), ("kv_block_size", C.c_int32),
                ("reserved", C.c_int32)]


class StatsC(C.Structure):
    _fields_ = [("forward_calls", C.c_uint64), ("prefill_tokens", C.c_uint64), ("decode_tokens", C.c_uint64),
                ("prefill_ms", C.c_double), ("decode_ms", C.c_double),
                ("gemm_ms", C.c_double), ("gemm_flops", C.c_double), ("gemm_launches", C.c_uint64),
                ("attn_ms", C.c_double), ("attn_flops", C.c_double), ("attn_launches", C.c_uint64),
                ("other_ms", C.c_double), ("other_launches", C.c_uint64), ("weight_bytes", C.c_double),
                ("evictions", C.c_uint64), ("graph_replays", C.c_uint64)]


class KernelStatC(C.Structure):         # nvl_kernel_stat
    _fields_ = [("site", C.c_int32), ("phase", C.c_int32), ("launches", C.c_uint64), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


class SamplingParamsC(C.Structure):     # nvl_sampling_params == tensor.SamplingParams (sampling.go:10-15)
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32),
                ("repetition_penalty", C.c_float)]


def sampling_params(temperature=1.0, top_p=1.0, top_k=0, repetition_penalty=1.2) -> SamplingParamsC:
    """DefaultSamplingParams (sampling.go:18-25) unless overridden."""
    return SamplingParamsC(float(temperature), float(top_p), int(top_k), float(repetition_penalty))


def declared_symbols() -> list[str]:
    """Every function include/nvllm.h declares (the drop-in surface)."""
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nvl_[a-z0-9_]+)\s*\(", text)))


_lib = None
vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64, this library links /opt/rocm's.  Whichever is
    # loaded first owns the device; loaded second, torch then finds "no ROCm-capable device" (seen on the GPU box when a test
    # touched the library before its first `import torch`).  If torch is installed, load it first — both then share its runtime,
    # which is the order every script here already used.  The library itself never needs torch.
    if "torch" not in sys.modules and not os.environ.get("NVLLM_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(str(LIB_PATH))
    L.nvl_create.argtypes = [C.POINTER(ModelConfigC), C.POINTER(RuntimeOptsC), C.POINTER(vp)]
    L.nvl_upload_tensor.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, i64, i64, C.c_int]
    L.nvl_upload_gpt2_qkv.argtypes = [vp, C.c_int, vp, vp]
    L.nvl_upload_falcon_qkv.argtypes = [vp, C.c_int, vp]
    L.nvl_load_safetensors.argtypes = [vp, C.c_char_p]
    L.nvl_load_config_json.argtypes = [C.c_char_p, C.POINTER(ModelConfigC)]
    L.nvl_finalize.argtypes = [vp]
    L.nvl_destroy.argtypes = [vp]
    L.nvl_destroy.restype = None
    for fn in ("nvl_seq_open", "nvl_seq_reset", "nvl_seq_close", "nvl_seq_len"):
        getattr(L, fn).argtypes = [vp, i64]
    L.nvl_seq_close_all.argtypes = [vp]
    L.nvl_forward.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_uint32, vp, vp]
    L.nvl_decode_greedy.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp]
    L.nvl_forward_paged.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_uint32, vp, vp]
    L.nvl_runner_run_paged.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp]
    L.nvl_decode_greedy_paged.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp]
    L.nvl_get_kv_paged.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.nvl_set_debug.argtypes = [vp, C.c_int]
    L.nvl_get_hidden.argtypes = [vp, C.c_int, vp, i64]
    L.nvl_get_kv.argtypes = [vp, i64, C.c_int, vp, vp]
    L.nvl_get_mamba_state.argtypes = [vp, i64, C.c_int, vp]
    L.nvl_get_weight.argtypes = [vp, C.c_int, C.c_int, vp, i64, i64]
    L.nvl_get_stamps.argtypes = [vp, vp, C.c_int, vp, i64]
    L.nvl_runner_run.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp]
    L.nvl_decode_sampled.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.POINTER(SamplingParamsC), vp, vp, vp, vp]
    L.nvl_sample.argtypes = [vp, C.c_int, C.POINTER(SamplingParamsC), vp, vp, vp, vp]
    L.nvl_runner_run_sampled.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(SamplingParamsC), vp, vp]
    L.nvl_op_sample.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.POINTER(SamplingParamsC), vp, vp, vp, vp, vp]
    L.nvl_set_profile.argtypes = [vp, C.c_int]
    L.nvl_get_stats.argtypes = [vp, C.POINTER(StatsC)]
    L.nvl_reset_stats.argtypes = [vp]
    L.nvl_get_kernel_stats.argtypes = [vp, C.POINTER(KernelStatC), C.c_int]
    L.nvl_kernel_site_name.argtypes = [C.c_int]
    L.nvl_kernel_site_name.restype = C.c_char_p
    L.nvl_last_error.argtypes = [vp]
    L.nvl_last_error.restype = C.c_char_p
    L.nvl_op_matmul.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int]
    L.nvl_op_layernorm.argtypes = [C.c_int, vp, vp, vp, f32, vp, C.c_int, C.c_int]
    L.nvl_op_softmax.argtypes = [C.c_int, vp, vp, C.c_int, C.c_int]
    L.nvl_op_gelu.argtypes = [C.c_int, vp, vp, i64]
    L.nvl_op_silu.argtypes = [C.c_int, vp, vp, i64]
    L.nvl_op_rope.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
    L.nvl_op_attention.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp]
    L.nvl_op_ffn.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.nvl_op_moe.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.nvl_op_argmax.argtypes = [C.c_int, vp, C.c_int, C.c_int, vp]
    L.nvl_bench_gemm.argtypes = [C.c_int] * 8 + [vp]
    L.nvl_set_tuning.argtypes = [C.c_int, C.c_int]
    L.nvl_tp_get_unique_id.argtypes = [vp, C.c_int]
    L.nvl_tp_init.argtypes = [vp, vp, C.c_int]
    L.nvl_tp_attach_local.argtypes = [vp, C.c_int]
    L.nvl_tp_p2p_export.argtypes = [vp, vp, C.c_int]
    L.nvl_tp_p2p_attach.argtypes = [vp, vp, C.c_int]
    L.nvl_tp_p2p_rearm.argtypes = [vp]
    L.nvl_sizeof.argtypes = [C.c_int]
    _lib = L
    return L


def check(rc: int, handle=None):
    if rc < 0:
        msg = lib().nvl_last_error(handle)
        raise NvlError(rc, msg.decode() if msg else "")
    return rc
