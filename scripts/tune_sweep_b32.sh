#!/bin/bash
# one-box sweep of the decode launch-shape knobs at the bench point (B=32); default first and last as the noise reference
run() { python bench.py --no-cpu-baseline $1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$1] value %.0f decode %.0f' % (d['value'], d['decode_tokens_per_s']))"; }
run ""
for t in 5=1024 5=4096 4=1 4=4 4=8 6=4 6=16 15=4 27=0 24=0; do run "--tune $t"; done
run ""
