#!/bin/bash
# quick loop on the GPU box: bench (B=32, B=1) + per-phase kernel breakdown; usage: r03_quick.sh TAG [extra bench args]
set -e
T=$1; shift
export TMPDIR=/tmp
python bench.py --no-cpu-baseline "$@" > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --batch 1 --no-cpu-baseline "$@" > gpurun_out/${T}_bench_b1.json 2>> gpurun_out/${T}_bench.err
for B in 32 1; do
  rm -rf /tmp/prof_$B
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$B -o bench -- python3 bench.py --batch $B --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/${T}_bench_under_rocprof_b$B.json 2>> gpurun_out/${T}_bench.err
  python scripts/prof_decode.py /tmp/prof_$B > gpurun_out/${T}_phase_breakdown_b$B.txt
done
python3 - <<PY
import json
for f in ("${T}_bench", "${T}_bench_b1"):
    d = json.load(open("gpurun_out/" + f + ".json"))
    print(f, d["value"], d["prefill_tokens_per_s"], d["decode_tokens_per_s"])
PY
grep -E "attn|== " gpurun_out/${T}_phase_breakdown_b32.txt gpurun_out/${T}_phase_breakdown_b1.txt
