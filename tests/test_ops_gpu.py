"""Op-level parity: each purego/tensor leaf function on the HIP path (through the C ABI's nvl_op_*
entry points) against the CPU oracle on the same seeded inputs.

Tolerances (stated, per precision):
  f32 mode : relative-to-max error <= 2e-5 (fp32 FMA vs the reference's unfused fp32 sums)
  bf16 mode: relative-to-max error <= 2e-2 (operands rounded to bf16, fp32 accumulate)
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

F32_TOL = 2e-5
BF16_TOL = 2e-2


def rng(seed):
    return np.random.default_rng(seed)


@pytest.mark.parametrize("m,k,n", [(1, 64, 128), (7, 128, 100), (130, 256, 257), (256, 2048, 384), (33, 192, 1003)])
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_matmul(gpu, oracle, m, k, n, precision):
    r = rng(m * 1000 + n)
    a = r.standard_normal((m, k), dtype=np.float32)
    b = r.standard_normal((k, n), dtype=np.float32) * 0.05
    want = oracle.matmul(a, b)
    got = gpu.ops.mat_mul(a, b, precision=precision)
    assert got.shape == want.shape
    assert rel_err(got, want) <= (F32_TOL if precision == "f32" else BF16_TOL)


@pytest.mark.parametrize("tile", [1, 2, 3, 5])
@pytest.mark.parametrize("m,k,n", [(300, 128, 200), (517, 256, 384), (256, 64, 128)])
def test_matmul_prefill_tile_kernels(gpu, oracle, tile, m, k, n):
    """All prefill tile instances (128x128x2st, 256x128x3st, 256x256x2st, 256x256 ping-pong) on ragged M/N edges."""
    r = rng(m + n)
    a = r.standard_normal((m, k), dtype=np.float32)
    b = r.standard_normal((k, n), dtype=np.float32) * 0.05
    old = gpu.lib().nvl_set_tuning(0, tile)
    try:
        got = gpu.ops.mat_mul(a, b, precision="bf16")
    finally:
        gpu.lib().nvl_set_tuning(0, old)
    assert rel_err(got, oracle.matmul(a, b)) <= BF16_TOL
    ai = np.zeros((m, k), np.float32)
    ai[np.arange(min(m, k)), np.arange(min(m, k))] = 1.0        # A = I block, asymmetric integer B: exact in bf16
    bi = (np.arange(k * n).reshape(k, n) % 251).astype(np.float32)
    old = gpu.lib().nvl_set_tuning(0, tile)
    try:
        assert np.array_equal(gpu.ops.mat_mul(ai, bi, precision="bf16"), oracle.matmul(ai, bi))
    finally:
        gpu.lib().nvl_set_tuning(0, old)


@pytest.mark.parametrize("interleave", [1, 0])
@pytest.mark.parametrize("m,k,n", [(65, 256, 128), (130, 2048, 2048), (200, 512, 3072), (512, 192, 1003), (96, 128, 200),
                                   (100, 256, 16384)])     # the wide-N form in groups
def test_matmul_row_groups_of_decode_form(gpu, oracle, interleave, m, k, n):
    """64 < M <= 512 with a small tile grid: ceil(M/64) groups of 64 activation rows of the decode kernels — one
    interleaved launch when the weight-block count is a multiple of 8 (n = 128, 2048, 3072), one launch per group
    otherwise (n = 1003, 200) or under key 13 = 0; ragged last groups of 1, 2, 8 and 32 rows.  The integer case
    is exact and asymmetric, so a group landing on the wrong output rows cannot pass."""
    r = rng(m + n)
    a = r.standard_normal((m, k), dtype=np.float32)
    b = r.standard_normal((k, n), dtype=np.float32) * 0.05
    ai = (np.arange(m * k).reshape(m, k) % 7).astype(np.float32)
    bi = (np.arange(k * n).reshape(k, n) % 5).astype(np.float32) - 2.0        # |sums| < 2^24: exact in fp32
    old = gpu.lib().nvl_set_tuning(13, interleave)
    try:
        got = gpu.ops.mat_mul(a, b, precision="bf16")
        goti = gpu.ops.mat_mul(ai, bi, precision="bf16")
    finally:
        gpu.lib().nvl_set_tuning(13, old)
    assert rel_err(got, oracle.matmul(a, b)) <= BF16_TOL
    assert np.array_equal(goti, oracle.matmul(ai, bi))


def test_matmul_bf16_exact_on_integers(gpu, oracle):
    """A=I-style check with ASYMMETRIC B (catches a transposed C write): small integers are exact in bf16."""
    k, n = 128, 192
    a = np.zeros((k, k), np.float32)
    a[np.arange(k), np.arange(k)] = 1.0
    b = (np.arange(k * n).reshape(k, n) % 251).astype(np.float32)
    got = gpu.ops.mat_mul(a, b, precision="bf16")
    assert np.array_equal(got, oracle.matmul(a, b))


@pytest.mark.parametrize("rows,hidden", [(1, 64), (5, 768), (9, 2048), (3, 4544)])
@pytest.mark.parametrize("rms", [True, False])
def test_layernorm(gpu, oracle, rows, hidden, rms):
    r = rng(hidden + rows)
    x = r.standard_normal((rows, hidden), dtype=np.float32) * 3 + 0.5
    w = 1 + 0.1 * r.standard_normal(hidden, dtype=np.float32)
    b = None if rms else 0.1 * r.standard_normal(hidden, dtype=np.float32)
    want = oracle.layernorm(x, w, b, 1e-5)
    got = gpu.ops.layer_norm(x, w, b, 1e-5)
    assert rel_err(got, want) <= 1e-5


@pytest.mark.parametrize("rows,cols", [(4, 32), (3, 1000), (1, 1)])
def test_softmax(gpu, oracle, rows, cols):
    x = rng(cols).standard_normal((rows, cols), dtype=np.float32) * 4
    assert rel_err(gpu.ops.softmax(x), oracle.softmax(x)) <= 1e-5


def test_gelu_silu(gpu, oracle):
    x = np.linspace(-12, 12, 4097, dtype=np.float32)
    assert rel_err(gpu.ops.gelu(x), oracle.gelu(x)) <= 1e-5
    assert rel_err(gpu.ops.silu(x), oracle.silu(x)) <= 1e-5


@pytest.mark.parametrize("base", [10000.0, 500000.0])
@pytest.mark.parametrize("start", [0, 17])
def test_rope(gpu, oracle, base, start):
    t = rng(3).standard_normal((3, 9, 64), dtype=np.float32)
    want = oracle.rope_apply(t, start, base, 64)
    got = gpu.ops.apply_rope_single_tensor(t, start, base, 64)
    assert rel_err(got, want) <= 1e-6


def test_rope_position_overflow_is_an_error_not_a_panic(gpu, oracle):
    t = np.zeros((1, 4, 64), np.float32)
    with pytest.raises(RuntimeError):
        oracle.rope_apply(t, 62, 10000.0, 64)          # reference panics (rope.go:176)
    with pytest.raises(gpu.NvlError) as e:
        gpu.ops.apply_rope_single_tensor(t, 62, 10000.0, 64)
    assert e.value.code == -2                           # NVL_ERR_POSITION


@pytest.mark.parametrize("nH,nKV,S,T", [(4, 2, 5, 5), (4, 2, 1, 70), (2, 2, 33, 100), (3, 1, 7, 64), (71, 1, 2, 130),
                                        (8, 2, 130, 130)])
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_attention(gpu, oracle, nH, nKV, S, T, precision):
    r = rng(nH * 100 + T)
    q = r.standard_normal((nH, S, 64), dtype=np.float32)
    k = r.standard_normal((nKV, T, 64), dtype=np.float32)
    v = r.standard_normal((nKV, T, 64), dtype=np.float32)
    want = oracle.gqa_core(q, k, v)
    got = gpu.ops.attention(q, k, v, precision=precision)
    assert rel_err(got, want) <= (5e-5 if precision == "f32" else BF16_TOL)


@pytest.mark.parametrize("nH,nKV,S,T", [(4, 2, 5, 5), (2, 2, 33, 100), (3, 1, 7, 64), (71, 1, 2, 130), (8, 2, 130, 130),
                                        (4, 2, 300, 1400), (8, 2, 257, 257)])
def test_attention_two_query_subtiles_per_wave(gpu, oracle, nH, nKV, S, T):
    """attn_prefill_bf16_kernel (256 query rows per workgroup, two 16-row sub-tiles per wave) on small shapes: ragged row
    counts, rows past the end of a sub-tile, waves without a valid row, diagonal and fully visible key tiles, more key
    tiles than the 3-slot LDS ring."""
    r = rng(nH * 10 + T)
    q = r.standard_normal((nH, S, 64), dtype=np.float32)
    k = r.standard_normal((nKV, T, 64), dtype=np.float32)
    v = r.standard_normal((nKV, T, 64), dtype=np.float32)
    old = gpu.lib().nvl_set_tuning(10, 2)
    try:
        got = gpu.ops.attention(q, k, v, precision="bf16")
    finally:
        gpu.lib().nvl_set_tuning(10, old)
    assert rel_err(got, oracle.gqa_core(q, k, v)) <= BF16_TOL


@pytest.mark.parametrize("nH,nKV,S,T,hd", [(8, 2, 1, 2100, 64),      # decode: > 1024 keys -> the split-T loop runs twice per wave
                                            (4, 4, 1, 1100, 64), (4, 1, 1, 1500, 128),
                                            (8, 1, 1, 300, 64), (12, 2, 1, 1300, 64), (16, 1, 1, 200, 128),   # 5..16 query heads per kv head
                                            (4, 2, 300, 1400, 64),   # chunked prefill against a long cache
                                            (4, 2, 200, 900, 128), (2, 2, 520, 520, 128)])   # hd 128 through the LDS ring
def test_attention_long_context(gpu, oracle, nH, nKV, S, T, hd):
    r = rng(T)
    q = r.standard_normal((nH, S, hd), dtype=np.float32)
    k = r.standard_normal((nKV, T, hd), dtype=np.float32)
    v = r.standard_normal((nKV, T, hd), dtype=np.float32)
    want = oracle.gqa_core(q, k, v)
    assert rel_err(gpu.ops.attention(q, k, v, precision="bf16"), want) <= BF16_TOL
    if S == 1:
        assert rel_err(gpu.ops.attention(q, k, v, precision="f32"), want) <= 5e-5


@pytest.mark.parametrize("nw", [2, 4, 8])
@pytest.mark.parametrize("nKV,T,hd", [(2, 70, 64), (2, 2100, 64), (1, 700, 128), (4, 1, 64)])
def test_decode_attention_wave_counts(gpu, oracle, nw, nKV, T, hd):
    """The decode kernel's waves-per-workgroup variants (chosen from the batch's longest context; key 15 forces one)
    give the same result for any context length: 2 waves walk 2100 keys in 9 rounds, 8 waves sit idle on 1 key."""
    r = rng(T + nw)
    q = r.standard_normal((4 * nKV, 1, hd), dtype=np.float32)
    k = r.standard_normal((nKV, T, hd), dtype=np.float32)
    v = r.standard_normal((nKV, T, hd), dtype=np.float32)
    old = gpu.lib().nvl_set_tuning(15, nw)
    try:
        got = gpu.ops.attention(q, k, v, precision="bf16")
    finally:
        gpu.lib().nvl_set_tuning(15, old)
    assert rel_err(got, oracle.gqa_core(q, k, v)) <= BF16_TOL


def test_attention_custom_scale_and_hd128(gpu, oracle):
    r = rng(5)
    q = r.standard_normal((4, 6, 128), dtype=np.float32)
    k = r.standard_normal((2, 40, 128), dtype=np.float32)
    v = r.standard_normal((2, 40, 128), dtype=np.float32)
    want = oracle.gqa_core(q, k, v, scale=0.015625)
    for precision, tol in (("f32", 5e-5), ("bf16", BF16_TOL)):
        assert rel_err(gpu.ops.attention(q, k, v, scale=0.015625, precision=precision), want) <= tol


def test_attention_online_softmax_rescale_branch(gpu, oracle):
    """Force the running max to jump at a later key tile (cdna guide rule 26): spike one key per tile."""
    r = rng(11)
    nH, S, T = 2, 4, 200
    q = r.standard_normal((nH, S, 64), dtype=np.float32)
    k = r.standard_normal((1, T, 64), dtype=np.float32) * 0.1
    v = r.standard_normal((1, T, 64), dtype=np.float32)
    k[0, 70] = q[0, -1] * 3.0      # tile 1
    k[0, 150] = q[0, -1] * 9.0     # tile 2, much larger
    want = oracle.gqa_core(q, k, v)
    assert rel_err(gpu.ops.attention(q, k, v, precision="bf16"), want) <= BF16_TOL
    assert rel_err(gpu.ops.attention(q, k, v, precision="f32"), want) <= 5e-5


@pytest.mark.parametrize("swiglu", [True, False])
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_ffn(gpu, oracle, swiglu, precision):
    r = rng(7)
    rows, H, F = 9, 128, 256
    x = r.standard_normal((rows, H), dtype=np.float32)
    w1 = r.standard_normal((H, 2 * F if swiglu else F), dtype=np.float32) * 0.1
    w2 = r.standard_normal((F, H), dtype=np.float32) * 0.1
    b1 = None if swiglu else r.standard_normal(F, dtype=np.float32) * 0.1
    b2 = None if swiglu else r.standard_normal(H, dtype=np.float32) * 0.1
    want = oracle.ffn(x, w1, b1, w2, b2, swiglu)
    got = gpu.ops.feed_forward(x, w1, b1, w2, b2, swiglu, precision=precision)
    assert rel_err(got, want) <= (5e-5 if precision == "f32" else BF16_TOL)


@pytest.mark.parametrize("rows,E,k", [(11, 8, 2), (700, 32, 4),      # 2800 pairs: several workgroups in the histogram / scatter
                                      (40, 64, 8)])                   # decode-sized: the single-launch plan, E = 64
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_moe(gpu, oracle, precision, rows, E, k):
    r = rng(9 + rows)
    H, I = 128, 64
    x = r.standard_normal((rows, H), dtype=np.float32)
    router = r.standard_normal((H, E), dtype=np.float32) * 0.3
    w_in = r.standard_normal((E, 2 * I, H), dtype=np.float32) * 0.1
    w_out = r.standard_normal((E, H, I), dtype=np.float32) * 0.1
    want = oracle.moe(x, router, w_in, w_out, k)
    got = gpu.ops.moe_forward(x, router, w_in, w_out, k, precision=precision)
    if precision == "bf16" and rows > 64:
        # the prefill variants: 256-row m-tiles (key 17) and the in-GEMM row gather (key 16 = 0) give the same rows
        for key, val in ((17, 256), (16, 0)):
            old = gpu.lib().nvl_set_tuning(key, val)
            try:
                alt = gpu.ops.moe_forward(x, router, w_in, w_out, k, precision=precision)
            finally:
                gpu.lib().nvl_set_tuning(key, old)
            assert np.array_equal(alt, got), (key, val)
    if precision == "f32":
        assert rel_err(got, want) <= 5e-5
        return
    # bf16 router logits can swap near-tied experts at the top-k boundary: check the rows whose k-th and (k+1)-th
    # logits are further apart than the rounding of the bf16 router GEMM (the 11-row fixture's rows all are)
    logits = np.sort(x.astype(np.float64) @ router.astype(np.float64), axis=1)[:, ::-1]
    clear = (logits[:, k - 1] - logits[:, k]) > 0.05
    assert clear.mean() > 0.5 and (rows > 11 or clear.all())
    assert rel_err(got[clear], want[clear]) <= 3e-2


def test_moe_prefill_pingpong_tiles(gpu, oracle):
    """Prefill MoE on shapes the grouped ping-pong GEMM takes (H % 256 == 0, 2I % 256 == 0, rows in expert order):
    256-row m-tiles that start on arbitrary rows of the expert-ordered buffer and end ragged.  Same k order as the
    lock-step 128-row tiles, so the two are bit-identical; against the oracle on the rows with a clear top-k."""
    r = rng(77)
    rows, E, k, H, I = 900, 8, 2, 256, 256
    x = r.standard_normal((rows, H), dtype=np.float32)
    router = r.standard_normal((H, E), dtype=np.float32) * 0.3
    w_in = r.standard_normal((E, 2 * I, H), dtype=np.float32) * 0.1
    w_out = r.standard_normal((E, H, I), dtype=np.float32) * 0.1
    got = gpu.ops.moe_forward(x, router, w_in, w_out, k, precision="bf16")
    old = gpu.lib().nvl_set_tuning(17, 128)
    try:
        alt = gpu.ops.moe_forward(x, router, w_in, w_out, k, precision="bf16")
    finally:
        gpu.lib().nvl_set_tuning(17, old)
    assert np.array_equal(alt, got)
    want = oracle.moe(x, router, w_in, w_out, k)
    logits = np.sort(x.astype(np.float64) @ router.astype(np.float64), axis=1)[:, ::-1]
    clear = (logits[:, k - 1] - logits[:, k]) > 0.1
    assert clear.mean() > 0.5
    assert rel_err(got[clear], want[clear]) <= 3e-2


def test_argmax_first_max_tie_rule(gpu, oracle):
    x = np.zeros((3, 50257), np.float32)
    x[0, [5, 40000]] = 2.0          # tie -> lowest index (cmd/ask/main.go:396 strict >)
    x[1, 50256] = 1.0
    x[2, :] = -1.0                  # all equal -> index 0
    got = gpu.ops.argmax(x)
    assert list(got) == [oracle.argmax(x[i]) for i in range(3)] == [5, 50256, 0]


@pytest.mark.parametrize("form", [0, 1, 8])       # automatic, narrow (one weight tile per wave), wide-N (4 tiles per wave)
@pytest.mark.parametrize("m,k,n", [(1, 256, 16400), (16, 192, 16384), (17, 512, 16448), (32, 2048, 16384),
                                   (33, 320, 20000), (64, 1024, 16640)])
def test_matmul_decode_forms(gpu, oracle, form, m, k, n):
    """The two decode GEMM forms (gemm_skinny_bf16_kernel / gemm_skinny_wide_bf16_kernel) on ragged M, K (k-steps that
    do not divide over the waves) and N (a last 64-row group that is partly padding)."""
    r = rng(m * 7 + n)
    a = r.standard_normal((m, k), dtype=np.float32)
    b = r.standard_normal((k, n), dtype=np.float32) * 0.05
    old = gpu.lib().nvl_set_tuning(2, form)
    try:
        got = gpu.ops.mat_mul(a, b, precision="bf16")
        ai = np.zeros((m, k), np.float32)
        ai[np.arange(min(m, k)), (np.arange(min(m, k)) * 5) % k] = 1.0      # rows pick distinct k: exact in bf16
        bi = ((np.arange(k * n, dtype=np.int64).reshape(k, n) * 7) % 251).astype(np.float32)
        goti = gpu.ops.mat_mul(ai, bi, precision="bf16")
    finally:
        gpu.lib().nvl_set_tuning(2, old)
    assert rel_err(got, oracle.matmul(a, b)) <= BF16_TOL
    assert np.array_equal(goti, oracle.matmul(ai, bi))


@pytest.mark.parametrize("rows", [20, 100, 128])     # 100, 128: two 64-row groups in one launch (the second ragged / full)
@pytest.mark.parametrize("swiglu", [True, False])
def test_ffn_wide_decode_form(gpu, oracle, swiglu, rows):
    """SwiGLU / GELU epilogues of the wide-N decode form (F large enough that its 64-row groups fill the chip)."""
    r = rng(11 + rows)
    H, F = 64, 8192 if swiglu else 16384
    x = r.standard_normal((rows, H), dtype=np.float32)
    w1 = r.standard_normal((H, 2 * F if swiglu else F), dtype=np.float32) * 0.1
    w2 = r.standard_normal((F, H), dtype=np.float32) * 0.02
    b1 = None if swiglu else r.standard_normal(F, dtype=np.float32) * 0.1
    b2 = None if swiglu else r.standard_normal(H, dtype=np.float32) * 0.1
    want = oracle.ffn(x, w1, b1, w2, b2, swiglu)
    assert rel_err(gpu.ops.feed_forward(x, w1, b1, w2, b2, swiglu, precision="bf16"), want) <= BF16_TOL
