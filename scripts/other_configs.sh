# the other BASELINE configs at their bench sizes (one line each), then a kernel trace of Granite MoE decode
out=gpurun_out/other_configs.txt; rm -f $out
for cfg in "gpt2 32" "granite-3.0-1b-a400m 8" "granite-3.0-1b-a400m 32" "falcon-7b 4" "llama-3-8b 8"; do set -- $cfg
  echo "== $1 batch $2" >> $out
  timeout -k 10 280 python bench.py --no-cpu-baseline --model $1 --batch $2 --steps 2 2>/dev/null >> $out || exit 1
done
