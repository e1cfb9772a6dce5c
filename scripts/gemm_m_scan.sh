#!/bin/bash
# ping-pong GEMM time vs rows at fixed N, K: separates a per-launch cost from the per-tile time.  usage: gemm_m_scan.sh
for shape in "2048 2048 1" "2048 2048 0" "16384 2048 2" "2048 8192 1"; do set -- $shape
  for M in 8192 16384 32768 65536 131072; do
    echo -n "N=$1 K=$2 epi=$3 M=$M: "; python scripts/gemm_one.py $M $1 $2 $3 5 20 | tail -1
  done
done
