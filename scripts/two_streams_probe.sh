#!/bin/bash
# PROBE: do two independent half-batches sharing one GPU (two rank processes, B = 16 each, same device) decode faster in
# aggregate than one batch of 32?  (each process streams the weights itself: twice the HBM traffic, but one process's
# launch boundaries / ramps / tails overlap the other's streams)
python bench.py --no-cpu-baseline --batch 32 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('one process  B=32: value %.0f prefill %.0f decode %.0f' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
python bench.py --no-cpu-baseline --batch 16 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('one process  B=16: value %.0f prefill %.0f decode %.0f' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
python bench.py --gpus 2 --rehearse-on-one-gpu --no-cpu-baseline --batch 16 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('two processes B=16 each, one GPU: aggregate value %.0f prefill %.0f decode %.0f (per-rank figures x2 inside)' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
python bench.py --gpus 2 --rehearse-on-one-gpu --no-cpu-baseline --batch 32 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('two processes B=32 each, one GPU: aggregate value %.0f prefill %.0f decode %.0f' % (d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s']))"
