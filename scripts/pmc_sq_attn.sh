#!/bin/bash
# SQ counters of the prefill attention kernel (two passes); usage: pmc_sq_attn.sh TAG "bench args"
T=$1; shift
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  rm -rf /tmp/sq_$i
  timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/sq_$i -- python3 bench.py --steps 1 --warmup 0 --gen 2 --no-cpu-baseline $* > /tmp/sq_$i.log 2>&1 || { tail -5 /tmp/sq_$i.log; exit 1; }
done
python3 - <<PY > gpurun_out/${T}_sq_attn.txt
import csv, glob, collections
for i in (1, 2):
    f = glob.glob("/tmp/sq_%d/**/*counter_collection.csv" % i, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_prefill" in n: acc["attn_prefill"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in sorted(d.items()): print(k, c, "n=%d" % len(v), "avg=%.0f" % (sum(v) / len(v)))
PY
cat gpurun_out/${T}_sq_attn.txt
