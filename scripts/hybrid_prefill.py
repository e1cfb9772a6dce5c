"""Hybrid (Mamba2 + attention) prefill rate with the chunked SSD scan against the sequential scan (nvl_set_tuning key 30),
on a Granite-4-like geometry (Mamba2 head_dim 64, state 128; 7 of 8 layers Mamba2).  usage: hybrid_prefill.py [batch] [prompt]"""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pkg = importlib.import_module("nano-vllm-go_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = pkg.synth.tiny_config("granite_hybrid", hidden=1024, ffn_dim=2048, num_heads=16, num_kv_heads=4, vocab_size=32000, num_layers=8,
                            max_seq_len=4096, mamba_expand=2, mamba_num_heads=32, mamba_head_dim=64, mamba_state_size=128, mamba_n_groups=1,
                            hybrid_layers=["mamba"] * 3 + ["attention"] + ["mamba"] * 4)
w = pkg.synth.make_weights(cfg, seed=1, scale=0.02)
hm = pkg.HipTransformerModel(cfg, w, precision="bf16", max_seqs=B, max_batch_tokens=B * S)
rng = np.random.default_rng(0)
prompts = [rng.integers(0, cfg["vocab_size"], S).tolist() for _ in range(B)]
ids = list(range(B))
NAMES = {1: "chunked SSD scan, chunks in parallel", 2: "chunked SSD scan, chunk after chunk ", 0: "sequential scan                     "}
for ssd in (1, 2, 0, 1, 2, 0):
    pkg.lib().nvl_set_tuning(30, ssd)
    best = 1e9
    for rep in range(4):
        for i in ids:
            hm.seq_reset(i)
        t0 = time.perf_counter()
        hm.forward_batch(ids, prompts, [0] * B, want_logits=False)
        best = min(best, time.perf_counter() - t0)
    print(f"hybrid prefill B={B} S={S}: {NAMES[ssd]}: {B * S / best:10.0f} tok/s ({best * 1e3:.2f} ms)")
hm.set_profile(True)
for ssd in (1, 2, 0):
    pkg.lib().nvl_set_tuning(30, ssd)
    hm.reset_stats()
    for i in ids:
        hm.seq_reset(i)
    hm.forward_batch(ids, prompts, [0] * B, want_logits=False)
    ks = {k["site"]: k for k in hm.kernel_stats() if k["phase"] == "prefill"}
    sc = ks.get("mamba_scan")
    print(f"  {NAMES[ssd].strip()}: mamba_scan {1e3 * sc['ms'] / sc['launches']:.1f} us per layer; " +
          ", ".join(f"{n} {1e3 * v['ms'] / v['launches']:.1f}" for n, v in ks.items() if n.startswith("mamba") and n != "mamba_scan"))
