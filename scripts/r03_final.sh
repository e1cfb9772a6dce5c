#!/bin/bash
# the round's committed measurements: PMC traffic (first: bench.py reads it), bench lines, kernel traces.
# usage (GPU box, repo root): r03_final.sh TAG      -> gpurun_out/TAG_*
set -e
T=${1:-r03}
export TMPDIR=/tmp
bash scripts/pmc_r02.sh gpurun_out/${T}_pmc_traffic.json > gpurun_out/${T}_pmc.log 2>&1
cp gpurun_out/${T}_pmc_traffic.json profiles/r03_pmc_traffic.json
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --batch 1 --no-cpu-baseline > gpurun_out/${T}_bench_b1.json 2>> gpurun_out/${T}_bench.err
python bench.py --decode stepwise --no-cpu-baseline > gpurun_out/${T}_bench_stepwise.json 2>> gpurun_out/${T}_bench.err
python bench.py --prompt 2048 --batch 8 --no-cpu-baseline > gpurun_out/${T}_bench_s2048_b8.json 2>> gpurun_out/${T}_bench.err
for B in 32 1; do
  rm -rf /tmp/prof_$B
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$B -o bench -- python3 bench.py --batch $B --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_under_rocprof_b$B.json 2>> gpurun_out/${T}_bench.err
  python scripts/prof_decode.py /tmp/prof_$B > gpurun_out/${T}_phase_breakdown_b$B.txt
  cp $(ls /tmp/prof_$B/*/*kernel_stats.csv /tmp/prof_$B/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats_b$B.csv
done
cat gpurun_out/${T}_bench.json
