# bench.py at "B S tune" points (decode-focused), one result line each
rm -f gpurun_out/b_chunk_sweep.log
for cfg in "128 128 20=0" "128 128 20=1" "128 128 20=2" "128 128 20=4" "256 128 20=0" "256 128 20=2"; do set -- $cfg; echo "B=$1 S=$2 $3" >> gpurun_out/b_chunk_sweep.log; timeout -k 10 200 python bench.py --no-cpu-baseline --prompt $2 --gen 128 --batch $1 --steps 2 --tune $3 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s'])
" >> gpurun_out/b_chunk_sweep.log || exit 1; done
