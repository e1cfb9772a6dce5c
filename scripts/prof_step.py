"""Print the kernel timeline (duration, gap to previous kernel's end) of ONE late decode step of bench.py
from a rocprofv3 kernel trace, and per-kernel aggregates over all decode steps."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(k):
    k = k[:k.find("(")] if "(" in k else k
    return k.replace("void nvl::", "").replace("_ZN3nvl", "")[:60]
emb = [i for i, r in enumerate(rows) if "embed_kernel" in r["Kernel_Name"]]
a, b = emb[-3], emb[-2]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
print(f"one decode step: {len(step)} launches, span {(int(step[-1]['End_Timestamp'])-t0)/1e3:.1f} us")
prev_end = t0
busy = 0
for r in step[:14] + step[-6:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  +{(s-t0)/1e3:8.1f}us gap {(s-prev_end)/1e3:5.1f} dur {(e-s)/1e3:6.1f}  {short(r['Kernel_Name'])}")
    prev_end = e
agg = collections.defaultdict(lambda: [0, 0, 0])
prev_end = None
for r in rows[emb[len(emb)//2]:emb[-1]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    agg[k][0] += 1; agg[k][1] += e - s
    if prev_end is not None: agg[k][2] += max(0, s - prev_end)
    prev_end = e
nsteps = len(emb) - 1 - len(emb)//2
print(f"per decode step (avg over {nsteps} steps):")
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {v[1]/nsteps/1e3:8.1f} us busy  {v[2]/nsteps/1e3:7.1f} us gap-before  n/step={v[0]/nsteps:5.1f} avg={v[1]/v[0]/1e3:6.2f}us  {k}")
    tot += v[1] + v[2]
print(f"  total {tot/nsteps/1e3:.1f} us/step")
