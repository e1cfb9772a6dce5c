"""Aggregate a rocprofv3 kernel trace of bench.py by kernel and by PHASE.

The trace is cut into forward passes at every argmax launch (the last kernels of a pass); a pass that contains the
prefill attention kernel (attn_prefill_bf16_kernel) is a prefill pass, every other pass is a decode step.  (Round 1 printed
"everything between the first and the last prefill attention" as prefill, which swallowed earlier steps' decode
kernels.)  usage: prof_decode.py <rocprofv3 output dir> [skip_passes]"""
import collections, csv, glob, sys

f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(k):
    k = k[:k.find("(")] if "(" in k else k
    return k.replace("void nvl::", "").replace("nvl::", "")[:96]


passes, cur = [], []
for r in rows:
    cur.append(r)
    n = r["Kernel_Name"]
    if "argmax_final_kernel" in n or "decode_seam_kernel" in n or "sample_row_kernel" in n:
        passes.append(cur)
        cur = []
if cur:
    passes.append(cur)
passes = [p for p in passes if any("gemm" in r["Kernel_Name"] for r in p)][skip:]
phases = {"prefill": [], "decode": []}
for p in passes:
    phases["prefill" if any("attn_prefill_bf16_kernel" in r["Kernel_Name"] for r in p) else "decode"].append(p)
for name, ps in phases.items():
    if not ps:
        continue
    agg = collections.defaultdict(lambda: [0, 0])
    busy = span = 0
    for p in ps:
        for r in p:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            agg[short(r["Kernel_Name"])][0] += 1
            agg[short(r["Kernel_Name"])][1] += d
            busy += d
        span += int(p[-1]["End_Timestamp"]) - int(p[0]["Start_Timestamp"])
    n = len(ps)
    print(f"== {name}: {n} passes, per pass: launches {sum(len(p) for p in ps)/n:.1f} busy {busy/n/1e3:.1f} us span {span/n/1e3:.1f} us")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"{v[1]/n/1e3:10.1f} us/pass  n/pass={v[0]/n:6.1f}  avg={v[1]/v[0]/1e3:8.2f} us  {k}")
