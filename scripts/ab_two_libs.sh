#!/bin/bash
# same-box A/B of two builds of the library: usage ab_two_libs.sh <other.so (path under nano-vllm-go_amd/lib)> [bench args]
O=$1; shift
L=nano-vllm-go_amd/lib
cp $L/libnvllm_hip.so /tmp/main.so
for i in 1 2; do
  for which in main other; do
    if [ $which = main ]; then cp /tmp/main.so $L/libnvllm_hip.so; else cp $L/$O $L/libnvllm_hip.so; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which: value %.0f prefill %.0f decode %.0f' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
  done
done
cp /tmp/main.so $L/libnvllm_hip.so
