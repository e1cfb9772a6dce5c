#!/bin/bash
# same-box A/B of one tuning override: usage ab_tune.sh "<key=value>" <batch> [more batches]
TUNE=$1; shift
for B in "$@"; do for i in 1 2; do for t in "" "--tune $TUNE"; do
  python bench.py --no-cpu-baseline --batch $B $t 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$B [$t]: value %.0f decode %.0f' % (d['value'], d['decode_tokens_per_s']))"
done; done; done
