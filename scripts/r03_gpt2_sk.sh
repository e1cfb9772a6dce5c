#!/bin/bash
# GPT-2 large decode batches with / without the split-K tile kernel for the residual projections (key 26); usage: r03_gpt2_sk.sh TAG
T=$1
for B in 128 256 512; do for t in "" "--tune 26=0"; do
  python bench.py --model gpt2 --batch $B --prompt 128 --gen 64 --steps 2 --no-cpu-baseline $t > /tmp/o.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('/tmp/o.json')); print('gpt2 B=$B $t decode tok/s %.0f' % d['decode_tokens_per_s'])" | tee -a gpurun_out/${T}_gpt2_sk.txt
done; done
