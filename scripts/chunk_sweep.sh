# decode throughput of large batches: bench.py at B x 128 prompt x 128 steps; optional tuning overrides per point
rm -f gpurun_out/b_chunk_sweep.log
for cfg in "64 13=1" "96 13=1" "128 13=1" "192 13=1" "256 13=1" "384 13=1" "512 13=1"; do set -- $cfg; echo "B=$1 $2" >> gpurun_out/b_chunk_sweep.log; timeout -k 10 200 python bench.py --no-cpu-baseline --prompt 128 --gen 128 --batch $1 --steps 2 --tune $2 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s'])
" >> gpurun_out/b_chunk_sweep.log || exit 1; done
