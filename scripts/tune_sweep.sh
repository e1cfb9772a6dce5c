#!/bin/bash
# decode tok/s for a list of nvl_set_tuning settings; usage: tune_sweep.sh "B list" "setting1" "setting2" ...
BL=$1; shift
for B in $BL; do for T in "$@"; do
  r=$(python bench.py --batch $B --prompt 512 --gen 128 --steps 2 --warmup 1 --no-cpu-baseline ${T:+--tune $T} 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['decode_tokens_per_s'], d['ms_per_step'])")
  echo "B=$B tune='$T' decode_tok/s ms_per_step: $r"
done; done
