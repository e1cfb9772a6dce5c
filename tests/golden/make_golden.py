#!/usr/bin/env python3
"""Generates tests/golden/*.npz — small input/output vectors for the forward path.

Two kinds, both DATA (no reference source text):
  falcon_split.npz  the fixture of the reference's own test purego/tensor/falcon_split_test.go:7-158
                    (its identifiable-pattern QKV matrix, regenerated from the formulas the test states,
                    and the values it asserts).  This is the ONLY vector the reference pins on the path.
  tiny_<family>.npz oracle outputs on seeded tiny models: last-row logits, 10 greedy token ids and a
                    per-layer checksum of the residual stream.  ORACLE-GENERATED: the reference is Go and
                    cannot be run in this image, so these pin the restatement (regression + GPU parity
                    without the oracle present), not the Go binary — "parity unpinned" in that sense.
Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import purego_oracle as O  # noqa: E402

pkg = importlib.import_module("nano-vllm-go_amd")
OUT = Path(__file__).resolve().parent


def falcon_fixture():
    nH, hd = 3, 4
    hidden = nH * hd
    qkv = np.zeros((hidden, (nH + 2) * hd), np.float32)
    for row in range(hidden):
        for h in range(nH):
            qkv[row, h * hd:(h + 1) * hd] = 10 * row + h        # falcon_split_test.go:26-30
        qkv[row, nH * hd:(nH + 1) * hd] = 100 * row              # :33-35
        qkv[row, (nH + 1) * hd:] = 1000 * row                    # :38-40
    # expectations asserted by the test (:66-98): Q[row][head h] == 10*row+h, K[row] == 100*row, V[row] == 1000*row
    q_want = np.array([[10 * r + h for h in range(nH)] for r in range(3)], np.float32)
    k_want = np.array([100 * r for r in range(3)], np.float32)
    v_want = np.array([1000 * r for r in range(3)], np.float32)
    # the 71x64 case (:100-158): only row 0 is filled
    row0 = np.zeros((71 + 2) * 64, np.float32)
    for h in range(71):
        row0[h * 64:(h + 1) * 64] = h
    row0[71 * 64:72 * 64] = 999.0
    row0[72 * 64:] = 888.0
    np.savez_compressed(OUT / "falcon_split.npz", qkv=qkv, q_want=q_want, k_want=k_want, v_want=v_want, row0_7b=row0)


def tiny(family):
    cfg = pkg.synth.tiny_config(family, tied_embedding=False)
    w = pkg.synth.make_weights(cfg, seed=11, scale=0.05, peaked_head=4.0)
    om = O.OracleModel(cfg, w)
    for seed in range(2000):   # a prompt whose oracle top-2 margins are all comfortable (see tests/test_model_gpu.py)
        prompt = np.random.default_rng(500 + seed).integers(0, cfg["vocab_size"], 12).tolist()
        toks, margins = om.greedy(prompt, 10, return_margins=True)
        if min(margins) > 0.03 and len(set(toks)) > 3:
            break
    else:
        raise SystemExit(f"{family}: no prompt with safe margins")
    logits, hidden = om.forward_with_cache(prompt, om.new_cache(), 0, want_hidden=True)
    np.savez_compressed(OUT / f"tiny_{family}.npz", cfg=json.dumps(cfg), seed=11, scale=0.05, peaked_head=4.0,
                        prompt=np.asarray(prompt, np.int32), greedy=np.asarray(toks, np.int32),
                        margins=np.asarray(margins, np.float32), last_logits=logits[-1].astype(np.float32),
                        hidden_abs_mean=np.abs(hidden).mean(axis=(1, 2)).astype(np.float32),
                        hidden_last_row=hidden[:, -1, :].astype(np.float32))
    print(family, "seed", seed, "min margin", min(margins), toks)


if __name__ == "__main__":
    falcon_fixture()
    for fam in ("llama", "gpt2", "falcon", "granite_moe"):
        tiny(fam)
