#!/bin/bash
# same-box per-kernel comparison (rocprofv3 kernel trace, B=32 decode): round-2 tree in ab_r02/ vs this tree
export TMPDIR=/tmp
for tree in ab_r02 .; do
  rm -rf /tmp/pk
  (cd $tree && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o bench -- python3 bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
  echo "== tree $tree"
  python scripts/prof_decode.py /tmp/pk | grep -A 8 "== decode" | cut -c1-140
done
