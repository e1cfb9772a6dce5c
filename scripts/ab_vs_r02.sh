#!/bin/bash
# same-box A/B of the headline bench: round-2 tree vs this tree.  Prepare the reference tree first (CPU container):
#   mkdir ab_r02 && git archive cbb0a1a | tar -x -C ab_r02 && make -C ab_r02/nano-vllm-go_amd/csrc -s   (any commit works; keep ab_r02/ untracked)
for i in 1 2; do
  (cd ab_r02 && python bench.py --no-cpu-baseline 2>/dev/null) | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r02 tree: value %.0f prefill %.0f decode %.0f' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
  python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('this tree: value %.0f prefill %.0f decode %.0f' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
done
