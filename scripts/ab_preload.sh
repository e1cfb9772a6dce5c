#!/bin/bash
# A/B of the kernarg-preload build flag on ONE box: bench with the committed build, rebuild without the flag, bench again.
set -e
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', ' '.join(sys.argv[1:]), d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s'])" "$@"; }
TAG=preload; run; run --batch 1; run --model gpt2 --batch 128 --steps 2
make -C nano-vllm-go_amd/csrc -s -B CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable" > /dev/null 2>&1
TAG=nopreload; run; run --batch 1; run --model gpt2 --batch 128 --steps 2
make -C nano-vllm-go_amd/csrc -s -B > /dev/null 2>&1
TAG=preload2; run; run --model gpt2 --batch 128 --steps 2
