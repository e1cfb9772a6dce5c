#!/bin/bash
# rocprofv3 per-site breakdown of one config; usage: r03_one_profile.sh TAG MODEL BATCH [extra bench args]
T=$1; M=$2; B=$3; shift 3
export TMPDIR=/tmp
rm -rf /tmp/po
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po -o bench -- python3 bench.py --model $M --batch $B --gen 32 --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/${T}_${M}_under_rocprof.json 2>/dev/null || exit 1
python scripts/prof_decode.py /tmp/po > gpurun_out/${T}_${M}_phase_breakdown.txt
grep -A 14 "== decode" gpurun_out/${T}_${M}_phase_breakdown.txt | cut -c1-160
