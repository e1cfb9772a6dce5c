#!/bin/bash
# bench + per-phase kernel breakdown at B=32 (the bench point) and B=1; run on the GPU box from the repo root
set -e
T=${1:-r02b}
export TMPDIR=/tmp
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --batch 1 --no-cpu-baseline > gpurun_out/${T}_bench_b1.json 2>> gpurun_out/${T}_bench.err
python bench.py --decode stepwise --no-cpu-baseline > gpurun_out/${T}_bench_stepwise.json 2>> gpurun_out/${T}_bench.err
for B in 32 1; do
  rm -rf /tmp/prof_$B
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$B -o bench -- python3 bench.py --batch $B --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_under_rocprof_b$B.json 2>> gpurun_out/${T}_bench.err
  python scripts/prof_decode.py /tmp/prof_$B > gpurun_out/${T}_phase_breakdown_b$B.txt
  cp $(ls /tmp/prof_$B/*/*kernel_stats.csv /tmp/prof_$B/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats_b$B.csv
done
cat gpurun_out/${T}_bench.json
