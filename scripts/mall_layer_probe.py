"""Does a decode layer run faster when its weights were streamed a moment ago?  The microarchitecture guide's
launches-baseline (Llama-3.2-1B decode layer, batch 1, five captured launches: 30.6 us) replays ONE layer; a model
streams 16 different ones plus a 525 MB LM head between two visits of the same layer, so nothing of a layer's 121.6 MB
is left in the 256 MB memory-side cache.  This times the fused decode loop (hipGraph replay per step) of Llama-3.2-1B
shapes with L layers and a small vocabulary (LM head 0.5 MB): per-layer time = (step(L) - step(0 layers' worth)) / L.
usage (GPU box): mall_layer_probe.py [--batch 1] [--ctx 512]"""
import argparse
import importlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--ctx", type=int, default=512)
    ap.add_argument("--steps", type=int, default=256)
    a = ap.parse_args()
    import torch
    from bench import gen_weights_on_device
    pkg = importlib.import_module("nano-vllm-go_amd")
    B, S = a.batch, a.ctx
    res = {}
    for L in (1, 2, 4, 16):
        cfg = dict(pkg.synth.FULL_CONFIGS["llama-3.2-1b"])
        cfg["num_layers"] = L
        cfg["vocab_size"] = 128
        cfg["max_seq_len"] = S + a.steps + 64
        hm = pkg.HipTransformerModel(cfg, None, precision="bf16", max_seqs=B, max_batch_tokens=min(16384, B * S))
        gen_weights_on_device(pkg, cfg, hm, torch, torch.device("cuda", 0), keep_host=False)
        hm.finalize()
        rng = np.random.default_rng(1)
        ids = list(range(B))
        for i in ids:
            hm.seq_reset(i)
        _, am = hm.forward_batch(ids, [rng.integers(0, 128, S).tolist() for _ in ids], [0] * B, want_logits=False)
        hm.decode_greedy(ids, am, 8)                     # capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = hm.decode_greedy(ids, am, a.steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps * 1e6
        res[L] = dt
        print(f"B={B} ctx={S}: {L:2d} layers, vocab 128: {dt:8.1f} us per decode step", flush=True)
        hm.close()
    for L in (2, 4, 16):
        print(f"  per layer from {L} vs 1 layers: {(res[L] - res[1]) / (L - 1):6.1f} us   (weights per layer 121.6 MB; {L} layers = {L * 121.6:.0f} MB)")
    print(f"  one layer alone (incl. embed, LM head, argmax, seam): {res[1]:.1f} us")


if __name__ == "__main__":
    main()
