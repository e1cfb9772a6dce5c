#!/bin/bash
# two --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) of the bench workload with a short decode phase
# -> profiles-ready JSON of HBM-side bytes per launch and site.  usage: pmc_r02.sh <out.json> [bench args]
set -e
OUT=$1; shift
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --gen 6 --no-cpu-baseline $*"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 bench.py $ARGS > /tmp/pmc_$c.log 2>&1
  echo "pass $c done"
done
python3 scripts/pmc_traffic_r02.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT "bench.py $ARGS"
