#!/bin/bash
# launch counts of the tensor-parallel decode path on ONE GPU (two rank processes, P2P all-reduce over IPC-mapped buffers)
# against the single-rank path; usage: r03_tp_rehearse.sh TAG
T=$1
set -e
python bench.py --model llama-3.2-1b --batch 8 --gen 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_tp1.json 2> gpurun_out/${T}_tp.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --tp --rehearse-on-one-gpu --model llama-3.2-1b --batch 8 --gen 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_tp2_rehearse.json 2>> gpurun_out/${T}_tp.err
python3 - <<PY
import json
for f in ("${T}_tp1", "${T}_tp2_rehearse"):
    d = json.loads([l for l in open("gpurun_out/" + f + ".json") if l.startswith("{")][-1])
    dec = [k for k in d["kernels"] if k["phase"] == "decode"]
    steps = 32          # gen x timed steps
    tot = sum(k["launches"] for k in dec)
    print(f, "decode launches per step: %.1f" % (tot / steps), {k["site"]: round(k["launches"] / steps, 1) for k in dec}, "graph replays", d.get("graph_replays"))
PY
