#!/bin/bash
cd scripts/micro && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o /tmp/launch_floor && cd ../..
echo "== default"; /tmp/launch_floor
echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 /tmp/launch_floor
for E in 0 1; do for B in 32 1; do
  echo "== bench B=$B HIP_FORCE_DEV_KERNARG=$E"
  HIP_FORCE_DEV_KERNARG=$E python bench.py --batch $B --no-cpu-baseline --steps 3 --warmup 1 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s'])"
done; done
