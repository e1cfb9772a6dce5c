"""Parity AT the BASELINE workload (SURVEY.md §8 d / BASELINE.json configs[1]): Llama-3.2-1B at FULL depth and width
(16 layers, H 2048, 32/8 heads, F 8192, V 128256), 32 sequences x 512 prompt tokens + 128 greedy decode steps — the
point bench.py measures.

  (a) the CPU oracle runs one 24-token sequence through all 16 layers and the 128256-row LM head: the bf16 product path's
      logits are within the bound DESIGN.md §3's error model gives for 16 layers (2e-2 max, 5e-3 RMS: asserted), the fp32
      parity mode's within 1e-4;
  (b) the whole 32 x 512 batch: bf16 logits vs the fp32 parity mode on the device (which (a) and the small-model tests
      tie to the oracle) within the bf16 tolerance, for the prefill and along the decode;
  (c) the FLIP RATE: over the 32 x 128 greedy decisions, teacher-forced on the fp32 path's tokens, the fraction on which
      the bf16 path picks a different id.  A random-weight model has near-flat logits (its top-2 margins are tiny), so
      the raw rate is a property of the synthetic weights, printed and loosely bounded; the asserted property is that NO
      decision flips whose fp32 top-2 margin exceeds twice the bf16 tolerance stated for this point (the condition under
      which "greedy ids bit-exact" is claimed everywhere else).
"""
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL_BF16, TOL_F32 = 1.5e-2, 1e-4
# Bounds for THIS depth from DESIGN.md §3 (tests/test_depth_parity_gpu.py checks the model layer by layer, and the device
# against a CPU emulation of its rounding points, which puts the EXPECTED last-row error at 16 layers at 1.45e-2 RMS /
# 1.46e-2 max — round 2 measured 1.4e-2 .. 1.7e-2):
#   one row of V logits:                 max |error| / max |logit| <= 1.25 x the emulated 1.46e-2 (the max of 128256 errors
#                                        moves ~5 % run to run) = 1.83e-2 ... stated as 2e-2
#   the whole 32 x 129-row run:          stated 3e-2 since round 2 (measured 1.85e-2 .. 2.26e-2: the largest of N V errors sits
#                                        sqrt(ln(N V) / ln V) = 1.31 further out than one row's), kept
#   RMS(error) / max |logit|:            stated 5e-3 since round 2 (measured 3.5e-3), kept
L_BENCH, V_BENCH = 16, 128256
TOL_BF16_ROW_MAX = 2e-2
TOL_BF16_BATCH_MAX = 3e-2
TOL_BF16_BATCH_RMS = 5e-3


def rms_rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.abs(b).max() + 1e-30))


def test_bench_workload_parity_and_flip_rate(gpu, oracle, capsys):
    cfg = dict(gpu.synth.FULL_CONFIGS["llama-3.2-1b"])
    assert (cfg["num_layers"], cfg["vocab_size"], cfg["hidden"]) == (16, 128256, 2048)
    B, S, G = 32, 512, 128
    w = gpu.synth.make_weights(cfg, seed=42, scale=0.02)
    rng = np.random.default_rng(1234 + 1)                       # bench.py's prompt seed (rank 0)
    prompts = rng.integers(0, cfg["vocab_size"], (B, S)).astype(np.int32)
    ids = list(range(B))
    hb = gpu.HipTransformerModel(cfg, w, precision="bf16", max_seqs=B, max_batch_tokens=16384)
    hf = gpu.HipTransformerModel(cfg, w, precision="f32", max_seqs=B, max_batch_tokens=16384)

    # ---- (a) the oracle at full depth on one short sequence --------------------------------------------------
    om = oracle.OracleModel(cfg, w)
    oracle.set_threads(min(16, os.cpu_count() or 1))
    try:
        short = prompts[0, :24].tolist()
        kv = om.new_cache()
        want = om.forward_with_cache(short, kv, 0, last_only=True)[-1]
        tok = oracle.argmax(want)
        want2 = om.forward_with_cache([tok], kv, 24, last_only=True)[-1]
    finally:
        oracle.set_threads(1)
    for model, tol in ((hb, TOL_BF16_ROW_MAX), (hf, TOL_F32)):
        model.seq_reset(99)
        got, _ = model.forward_batch([99], [short], [0])
        assert rel_err(got[0], want) <= tol
        got2, _ = model.forward_batch([99], [[tok]], [24])
        assert rel_err(got2[0], want2) <= tol
        if model is hb:
            assert rms_rel(got[0], want) <= TOL_BF16_BATCH_RMS and rms_rel(got2[0], want2) <= TOL_BF16_BATCH_RMS
        model.seq_close(99)
    top2 = np.partition(want, -2)[-2:]
    if (top2[1] - top2[0]) / np.abs(want).max() > 2 * TOL_BF16:
        assert int(np.argmax(got[0])) == tok
    del om

    # ---- (b) the whole batch: prefill logits, bf16 vs the fp32 parity mode ---------------------------------------
    def prefill(model):
        out = np.empty((B, cfg["vocab_size"]), np.float32)
        for i in ids:
            model.seq_reset(i)
        for b0 in range(0, B, 16384 // S):                      # max_batch_tokens-sized forward calls, as bench.py
            sel = ids[b0:b0 + 16384 // S]
            lg, _ = model.forward_batch(sel, [prompts[i] for i in sel], [0] * len(sel))
            out[b0:b0 + len(sel)] = lg
        return out
    lb, lf = prefill(hb), prefill(hf)
    assert rel_err(lb, lf) <= TOL_BF16_BATCH_MAX and rms_rel(lb, lf) <= TOL_BF16_BATCH_RMS
    # ---- (c) 128 teacher-forced greedy steps: flip rate ----------------------------------------------------------
    flips = decisions = big_margin = big_flips = 0
    worst = worst_rms = 0.0
    for step in range(G + 1):
        am_f, am_b = lf.argmax(1), lb.argmax(1)
        srt = np.partition(lf, -2, axis=1)[:, -2:]
        margin = (srt[:, 1] - srt[:, 0]) / np.abs(lf).max(axis=1)
        diff = am_f != am_b
        flips += int(diff.sum()); decisions += B
        big = margin > 2 * TOL_BF16_BATCH_MAX
        big_margin += int(big.sum()); big_flips += int((diff & big).sum())
        worst = max(worst, rel_err(lb, lf))
        worst_rms = max(worst_rms, rms_rel(lb, lf))
        if step == G:
            break
        nxt = [[int(t)] for t in am_f]                            # both paths continue on the fp32 path's token
        lf, _ = hf.forward_batch(ids, nxt, [S + step] * B)
        lb, _ = hb.forward_batch(ids, nxt, [S + step] * B)
    rate = flips / decisions
    with capsys.disabled():
        print(f"\n[bench-point parity] Llama-3.2-1B 16 layers, 32 x (512 + 128): bf16 vs fp32-mode logits max rel err {worst:.2e} (RMS {worst_rms:.2e}); "
              f"greedy flip rate {flips}/{decisions} = {rate:.4f} (random weights, near-flat logits); "
              f"decisions with fp32 top-2 margin > 2 x tol: {big_margin}, flipped among them: {big_flips}")
    assert worst <= TOL_BF16_BATCH_MAX and worst_rms <= TOL_BF16_BATCH_RMS
    assert big_flips == 0
    assert rate <= 0.5          # loose: documents the number, catches a broken path (a wrong path flips ~ everything)
    hb.close()
    hf.close()
