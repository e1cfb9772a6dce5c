#!/bin/bash
# Granite-3.0 MoE decode: bench lines (B=8, 32) + per-site rocprofv3 breakdown at B=8; usage: r03_granite.sh TAG [extra bench args]
T=$1; shift
export TMPDIR=/tmp
set -e
for B in 8 32; do
  python bench.py --model granite-3.0-1b-a400m --batch $B --no-cpu-baseline "$@" > gpurun_out/${T}_granite_b$B.json 2>> gpurun_out/${T}_granite.err
done
rm -rf /tmp/pg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o bench -- python3 bench.py --model granite-3.0-1b-a400m --batch 8 --gen 32 --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/${T}_granite_under_rocprof.json 2>> gpurun_out/${T}_granite.err
python scripts/prof_decode.py /tmp/pg > gpurun_out/${T}_granite_phase_breakdown_b8.txt
python3 - <<PY
import json
for B in (8, 32):
    d = json.load(open("gpurun_out/${T}_granite_b%d.json" % B))
    print("granite B=%d" % B, d["value"], d["prefill_tokens_per_s"], d["decode_tokens_per_s"], d["decode_roofline"]["frac"])
PY
grep -A 16 "== decode" gpurun_out/${T}_granite_phase_breakdown_b8.txt | cut -c1-150
