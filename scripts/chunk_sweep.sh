# bench.py at B x S under tuning overrides: "B S tune" (short generation: prefill-focused)
rm -f gpurun_out/b_chunk_sweep.log
for cfg in "1 128 18=64" "1 128 18=256" "1 256 18=64" "1 256 18=256" "1 512 18=64" "1 512 18=512" "4 128 18=64" "4 128 18=512"; do set -- $cfg; echo "B=$1 S=$2 $3" >> gpurun_out/b_chunk_sweep.log; timeout -k 10 200 python bench.py --no-cpu-baseline --prompt $2 --gen 8 --batch $1 --steps 5 --tune $3 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['prefill_tokens_per_s'], d['decode_tokens_per_s'])
" >> gpurun_out/b_chunk_sweep.log || exit 1; done
