"""Aggregate a rocprofv3 kernel trace of bench.py by launch SITE and PHASE.

The trace is cut into forward passes at every argmax / seam launch; a pass with the prefill attention kernel is a prefill
pass, every other pass a decode step (scripts/kernel_sites.py).  Per site: launches per pass, average kernel time, time
per pass.  usage: prof_decode.py <rocprofv3 output dir> [skip_passes]"""
import collections, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sites import label_passes

f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
passes = label_passes(rows)[skip:]
for name in ("prefill", "decode"):
    ps = [p for p in passes if p[0]["phase"] == name]
    if not ps:
        continue
    agg = collections.defaultdict(lambda: [0, 0])
    busy = span = 0
    for p in ps:
        for r in p:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            k = f"{r['site']:11s} {r['kname'][:84]}"
            agg[k][0] += 1
            agg[k][1] += d
            busy += d
        span += int(p[-1]["End_Timestamp"]) - int(p[0]["Start_Timestamp"])
    n = len(ps)
    print(f"== {name}: {n} passes, per pass: launches {sum(len(p) for p in ps)/n:.1f} busy {busy/n/1e3:.1f} us span {span/n/1e3:.1f} us")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"{v[1]/n/1e3:10.1f} us/pass  n/pass={v[0]/n:6.1f}  avg={v[1]/v[0]/1e3:8.2f} us  {k}")
