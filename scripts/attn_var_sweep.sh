#!/bin/bash
# prefill attention site time at two shapes, one line per nvl_set_tuning setting; usage: attn_var_sweep.sh "" "k=v" ...
for V in "$@"; do
  for SH in "2048 8" "512 32"; do set -- $SH
    python bench.py --no-cpu-baseline --prompt $1 --batch $2 --gen 2 --steps 2 --warmup 1 ${V:+--tune $V} 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin)
a=[k for k in d['kernels'] if k['site']=='attention' and k['phase']=='prefill'][0]
print('tune=$V S=$1 B=$2 attn_us', a['avg_launch_us'], 'TF/s', a.get('achieved_tflops'), 'prefill tok/s', d['prefill_tokens_per_s'])"
  done
done
