#!/bin/bash
# rocprofv3 kernel traces of the other BASELINE configs (per-site breakdowns via scripts/prof_decode.py); usage: r02_other_profiles.sh TAG
T=${1:-r02}
export TMPDIR=/tmp
for cfg in "granite-3.0-1b-a400m 32" "llama-3-8b 16" "falcon-7b 16" "gpt2 128"; do set -- $cfg
  rm -rf /tmp/po
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po -o bench -- python3 bench.py --model $1 --batch $2 --gen 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_$1_under_rocprof.json 2>/dev/null || exit 1
  python scripts/prof_decode.py /tmp/po > gpurun_out/${T}_$1_phase_breakdown.txt
  echo "$1 done"
done
