#!/bin/bash
# same-box A/B of the headline bench: round-2 tree (ab_r02/, git archive of the round-2 head, built in place) vs this tree
for i in 1 2; do
  (cd ab_r02 && python bench.py --no-cpu-baseline 2>/dev/null) | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r02 tree: value %.0f prefill %.0f decode %.0f' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
  python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('this tree: value %.0f prefill %.0f decode %.0f' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
done
