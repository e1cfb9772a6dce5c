"""purego/tensor's leaf functions on the device, one call each, through the nvl_op_* entry points
(host fp32 in, host fp32 out).  Names follow the reference (tensor.go / rope.go / transformer.go /
moe.go); used by the parity tests."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def mat_mul(a, b, precision="bf16", device=0):                 # MatMul tensor.go:62
    a, b = _f(a), _f(b)
    m, k = a.shape
    _, n = b.shape
    c = np.empty((m, n), np.float32)
    L.check(L.lib().nvl_op_matmul(device, L.PRECISION[precision], _p(a), _p(b), _p(c), m, k, n))
    return c


def layer_norm(x, weight, bias, eps, device=0):                # LayerNorm tensor.go:193
    x, w = _f(x), _f(weight)
    b = None if bias is None else _f(bias)
    y = np.empty_like(x)
    L.check(L.lib().nvl_op_layernorm(device, _p(x), _p(w), _p(b), eps, _p(y), x.shape[0], x.shape[1]))
    return y


def softmax(x, device=0):                                      # Softmax tensor.go:128
    x = _f(x)
    y = np.empty_like(x)
    L.check(L.lib().nvl_op_softmax(device, _p(x), _p(y), x.shape[0], x.shape[1]))
    return y


def gelu(x, device=0):                                         # GELU tensor.go:181
    x = _f(x)
    y = np.empty_like(x)
    L.check(L.lib().nvl_op_gelu(device, _p(x), _p(y), x.size))
    return y


def silu(x, device=0):                                         # SiLU mamba2.go:360
    x = _f(x)
    y = np.empty_like(x)
    L.check(L.lib().nvl_op_silu(device, _p(x), _p(y), x.size))
    return y


def apply_rope_single_tensor(t, start_pos, base, max_seq, device=0):   # rope.go:153
    t = _f(t).copy()
    heads, seq, hd = t.shape
    L.check(L.lib().nvl_op_rope(device, _p(t), heads, seq, hd, start_pos, base, max_seq))
    return t


def attention(q, k, v, scale=0.0, precision="bf16", device=0):  # attention.go:354-470 / mqa.go:184
    q, k, v = _f(q), _f(k), _f(v)
    nH, S, hd = q.shape
    nKV, T, _ = k.shape
    out = np.empty_like(q)
    L.check(L.lib().nvl_op_attention(device, L.PRECISION[precision], _p(q), _p(k), _p(v), nH, nKV, S, T, hd,
                                     scale, _p(out)))
    return out


def feed_forward(x, w1, b1, w2, b2, swiglu, precision="bf16", device=0):   # FeedForward.Forward transformer.go:40
    x, w1, w2 = _f(x), _f(w1), _f(w2)
    b1 = None if b1 is None else _f(b1)
    b2 = None if b2 is None else _f(b2)
    rows, hidden = x.shape
    y = np.empty((rows, hidden), np.float32)
    L.check(L.lib().nvl_op_ffn(device, L.PRECISION[precision], _p(x), _p(w1), _p(b1), _p(w2), _p(b2), rows, hidden,
                               w2.shape[0], int(swiglu), _p(y)))
    return y


def moe_forward(x, router, w_in, w_out, top_k, precision="bf16", device=0):   # MoELayer.Forward moe.go:43
    x, router, w_in, w_out = _f(x), _f(router), _f(w_in), _f(w_out)
    rows, hidden = x.shape
    y = np.empty((rows, hidden), np.float32)
    L.check(L.lib().nvl_op_moe(device, L.PRECISION[precision], _p(x), _p(router), _p(w_in), _p(w_out), rows, hidden,
                               router.shape[1], top_k, w_out.shape[2], _p(y)))
    return y


def argmax(x, device=0):                                       # argmax cmd/ask/main.go:389
    x = _f(x)
    if x.ndim == 1:
        x = x[None]
    out = np.empty(x.shape[0], np.int32)
    L.check(L.lib().nvl_op_argmax(device, _p(x), x.shape[0], x.shape[1], _p(out)))
    return out


def _histories(histories, rows):
    """-> (pointer array, int32 lengths, keep-alive list) for `rows` token histories (None = no history)."""
    arrs = [np.ascontiguousarray([] if h is None else h, dtype=np.int32) for h in (histories or [None] * rows)]
    ptrs = (C.c_void_p * rows)(*[a.ctypes.data_as(C.c_void_p).value if a.size else None for a in arrs])
    lens = np.asarray([a.size for a in arrs], np.int32)
    return ptrs, lens, arrs


def sample_with_history(logits, histories, uniforms, *, temperature=1.0, top_p=1.0, top_k=0, repetition_penalty=1.2,
                        return_probs=False, device=0):        # SampleWithHistory sampling.go:33, u = rand.Float32()
    lg = _f(logits)
    if lg.ndim == 1:
        lg = lg[None]
    rows, V = lg.shape
    ptrs, lens, keep = _histories(histories, rows)
    u = _f(uniforms).reshape(rows)
    out = np.empty(rows, np.int32)
    probs = np.empty((rows, V), np.float32) if return_probs else None
    sp = L.sampling_params(temperature, top_p, top_k, repetition_penalty)
    L.check(L.lib().nvl_op_sample(device, _p(lg), rows, V, C.byref(sp), ptrs, _p(lens), _p(u), _p(out), _p(probs)))
    return (out, probs) if return_probs else out
