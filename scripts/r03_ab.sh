#!/bin/bash
# A/B helper on the GPU box: decode tok/s of the usual configs; usage: r03_ab.sh TAG
T=$1; shift
for cfg in "llama-3.2-1b 32" "llama-3.2-1b 1" "gpt2 128" "llama-3-8b 16" "falcon-7b 16" "granite-3.0-1b-a400m 8"; do set -- $cfg
  python bench.py --model $1 --batch $2 --no-cpu-baseline > gpurun_out/${T}_$1_b$2.json 2>/dev/null || { echo "$1 failed"; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/${T}_$1_b$2.json')); print('$T $1 B=$2: value %.0f prefill %.0f decode %.0f' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
done
