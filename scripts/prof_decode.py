"""Aggregate a rocprofv3 kernel trace of bench.py by kernel, separately for the decode phase
(everything after the last prefill attention launch)."""
import collections, csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_prefill = max(i for i, r in enumerate(rows) if "attn_bf16_kernel" in r["Kernel_Name"])
first_prefill = min(i for i, r in enumerate(rows) if "attn_bf16_kernel" in r["Kernel_Name"])
for name, part in (("prefill", rows[first_prefill - 3:last_prefill + 8]), ("decode", rows[last_prefill + 8:])):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in part:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = r["Kernel_Name"]
        k = k[:k.find("(")] if "(" in k else k
        agg[k[:90]][0] += 1
        agg[k[:90]][1] += d
    tot = sum(v[1] for v in agg.values())
    span = int(part[-1]["End_Timestamp"]) - int(part[0]["Start_Timestamp"])
    print(f"== {name}: launches {len(part)} busy {tot/1e3:.1f} us span {span/1e3:.1f} us")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{v[1]/1e3:10.1f} us  n={v[0]:5d}  avg={v[1]/v[0]/1e3:8.2f} us  {k}")
