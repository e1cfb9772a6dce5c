# the other BASELINE configs at sizes that fill the chip (one bench line each); usage: other_configs.sh [out]
out=${1:-gpurun_out/other_configs.txt}; rm -f $out
for cfg in "gpt2 128" "granite-3.0-1b-a400m 8" "granite-3.0-1b-a400m 32" "falcon-7b 16" "llama-3-8b 16"; do set -- $cfg
  echo "== $1 batch $2" >> $out
  timeout -k 10 400 python bench.py --no-cpu-baseline --model $1 --batch $2 --steps 2 2>/dev/null >> $out || exit 1
done
