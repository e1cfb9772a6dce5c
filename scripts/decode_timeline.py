"""In-kernel timeline of ONE decode step (nvl_set_debug mode 4: csrc/common.h nvl_stamp).

Every decode-sized projection / attention launch records, per workgroup, the chip's 100 MHz clock at six points:
  0 entered   1 first loads issued   2 first data used   3 own stream done   4 workgroup's streams done   5 stores retired
Per launch this prints (microseconds): the gap from the previous launch's last store to this launch's first workgroup
entry (the kernel boundary as the shader sees it), the spread of workgroup entries (dispatch ramp), and the median /
maximum over workgroups of each phase.  usage (GPU box): decode_timeline.py [--batch 32] [--prompt 512] [--steps 3]"""
import argparse
import ctypes as C
import importlib
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
# the diagnostic build (make -C nano-vllm-go_amd/csrc diag): the product library has no stamp sites
os.environ.setdefault("NVLLM_LIB", str(ROOT / "nano-vllm-go_amd" / "lib" / "libnvllm_hip_diag.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--model", default="llama-3.2-1b")
    ap.add_argument("--tune", default="")
    a = ap.parse_args()
    import torch
    from bench import gen_weights_on_device
    pkg = importlib.import_module("nano-vllm-go_amd")
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("=")
        pkg.lib().nvl_set_tuning(int(k), int(v))
    cfg = dict(pkg.synth.FULL_CONFIGS[a.model])
    B, S = a.batch, a.prompt
    hm = pkg.HipTransformerModel(cfg, None, precision="bf16", max_seqs=B, max_batch_tokens=min(16384, B * S))
    gen_weights_on_device(pkg, cfg, hm, torch, torch.device("cuda", 0), keep_host=False)
    hm.finalize()
    rng = np.random.default_rng(1)
    ids = list(range(B))
    per = max(1, min(B, 16384 // S))
    am = np.zeros(B, np.int32)
    for i in ids:
        hm.seq_reset(i)
    for b0 in range(0, B, per):
        sel = ids[b0:b0 + per]
        _, r = hm.forward_batch(sel, [rng.integers(0, cfg["vocab_size"], S).tolist() for _ in sel], [0] * len(sel), want_logits=False)
        am[b0:b0 + len(sel)] = r
    pos = S
    for _ in range(4):                                   # warm: code objects, caches
        _, am = hm.forward_batch(ids, [[int(t)] for t in am], [pos] * B, want_logits=False); pos += 1
    L = pkg.lib()
    for step in range(a.steps):
        hm.set_debug(4)
        _, am = hm.forward_batch(ids, [[int(t)] for t in am], [pos] * B, want_logits=False); pos += 1
        recs = np.zeros((128, 3), np.int32)
        stamps = np.zeros(128 * 2048 * 8, np.uint64)
        n = L.nvl_get_stamps(hm.h, recs.ctypes.data_as(C.c_void_p), 128, stamps.ctypes.data_as(C.c_void_p), stamps.size)
        hm.set_debug(0)
        assert n > 0, n
        if step < a.steps - 1:
            continue
        off, prev_end, rows = 0, None, []
        tot = dict(gap=0.0, ramp=0.0, pro=0.0, first=0.0, stream=0.0, wait=0.0, epi=0.0, span=0.0)
        for i in range(n):
            site, phase, nwg = (int(x) for x in recs[i])
            st = stamps[off:off + nwg * 8].reshape(nwg, 8).astype(np.int64); off += nwg * 8
            live = st[:, 0] > 0                           # workgroups that ran (a launch may use fewer than the slab)
            st = st[live]
            if st.shape[0] == 0:
                continue
            t0 = st[:, 0].min()
            us = lambda v: float(v) / 100.0               # 100 MHz ticks -> us
            d = lambda a_, b_: (st[:, b_] - st[:, a_])[(st[:, b_] > 0) & (st[:, a_] > 0)]
            med = lambda v: us(np.median(v)) if v.size else float("nan")
            mx = lambda v: us(v.max()) if v.size else float("nan")
            end = st[:, 5].max() if (st[:, 5] > 0).any() else st[:, 4].max()
            row = dict(site=L.nvl_kernel_site_name(site).decode(), wgs=int(st.shape[0]), gap=us(t0 - prev_end) if prev_end else float("nan"),
                       ramp=us(st[:, 0].max() - t0), pro=med(d(0, 1)), first=med(d(1, 2)), stream=med(d(2, 3)), wait=med(d(3, 4)),
                       epi=med(d(4, 5)), epi_max=mx(d(4, 5)), span=us(end - t0))
            rows.append(row)
            prev_end = end
        print(f"== {a.model} B={B} ctx={pos}: {n} stamped launches of one decode step (us; median over workgroups unless noted)")
        print(f"{'site':10s} {'wgs':>5s} {'gap':>6s} {'ramp':>6s} {'prolog':>6s} {'1stdat':>6s} {'stream':>6s} {'wgwait':>6s} {'epilog':>6s} {'epimax':>6s} {'span':>7s}")
        agg = {}
        for r in rows:
            g = agg.setdefault(r["site"], dict(n=0, **{k: 0.0 for k in ("gap", "ramp", "pro", "first", "stream", "wait", "epi", "epi_max", "span")}))
            g["n"] += 1
            for k in ("gap", "ramp", "pro", "first", "stream", "wait", "epi", "epi_max", "span"):
                if r[k] == r[k]:
                    g[k] += r[k]
        for site, g in agg.items():
            k = g["n"]
            print(f"{site:10s} {'':>5s} {g['gap']/k:6.2f} {g['ramp']/k:6.2f} {g['pro']/k:6.2f} {g['first']/k:6.2f} {g['stream']/k:6.2f} {g['wait']/k:6.2f} {g['epi']/k:6.2f} {g['epi_max']/k:6.2f} {g['span']/k:7.2f}   (avg of {k})")
        print("first 12 launches:")
        for r in rows[:12]:
            print(f"{r['site']:10s} {r['wgs']:5d} {r['gap']:6.2f} {r['ramp']:6.2f} {r['pro']:6.2f} {r['first']:6.2f} {r['stream']:6.2f} {r['wait']:6.2f} {r['epi']:6.2f} {r['epi_max']:6.2f} {r['span']:7.2f}")
        total = sum(r["span"] for r in rows) + sum(r["gap"] for r in rows if r["gap"] == r["gap"])
        print(f"sum of spans {sum(r['span'] for r in rows):.1f} us + gaps {sum(r['gap'] for r in rows if r['gap'] == r['gap']):.1f} us = {total:.1f} us")
    hm.close()


if __name__ == "__main__":
    main()
