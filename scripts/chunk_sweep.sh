# bench.py at "B S tune" points (default: the large-decode-batch ladder), one result line each:
#   bash scripts/chunk_sweep.sh "128 128 13=1" "128 128 13=0"      -> gpurun_out/b_chunk_sweep.log
# prints: value tok/s, prefill tok/s, decode tok/s
rm -f gpurun_out/b_chunk_sweep.log
if [ $# -eq 0 ]; then set -- "64 128 13=1" "96 128 13=1" "128 128 13=1" "192 128 13=1" "256 128 13=1" "384 128 13=1" "512 128 13=1"; fi
for cfg in "$@"; do set -- $cfg; echo "B=$1 S=$2 $3" >> gpurun_out/b_chunk_sweep.log; timeout -k 10 200 python bench.py --no-cpu-baseline --prompt $2 --gen 128 --batch $1 --steps 2 --tune $3 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s'])
" >> gpurun_out/b_chunk_sweep.log || exit 1; done
