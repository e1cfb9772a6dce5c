# SURVEY §8(d) grid on one GPU: prompt length S x batch B, 128 greedy decode steps; prints one row per point
out=${1:-gpurun_out/grid.txt}
echo "#    S     B   value tok/s  prefill tok/s  decode tok/s  ms/step(decode)  prefill GEMM TF/s" > $out
for S in 128 512 2048; do for B in 1 8 32 128; do
  timeout -k 10 280 python bench.py --no-cpu-baseline --prompt $S --gen 128 --batch $B --steps 3 --warmup 1 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); dec=d['decode_tokens_per_s']
        print('%6d %5d %12.1f %14.1f %13.1f %16.3f %18.1f' % ($S,$B,d['value'],d['prefill_tokens_per_s'],dec,1000.0*$B/dec,d['roofline']['achieved']))
" >> $out || exit 1
done; done
