"""Summarise rocprofv3 --pmc passes of scripts/gemm_one.py (one projection shape, one kernel form): MFMA busy fraction,
effective clock (GRBM_GUI_ACTIVE / 8 / time), LDS bank conflicts.  usage: pmc_mfma.py <dir with pass_*/...csv> <out.json>"""
import collections, csv, glob, json, sys
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_bf16" not in k: continue
        cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, c in cnt.items():
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    t_ns = sum(dur[k]) / max(1, len(dur[k]))
    e = {"launches": len(dur[k]), "avg_ns_under_pmc": t_ns, "counters": avg}
    if "GRBM_GUI_ACTIVE" in avg and t_ns > 0:
        e["effective_clock_GHz"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / t_ns
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg:
        # MFMA busy cycles are summed over the 1024 SIMDs; GUI_ACTIVE over the 8 XCDs
        e["mfma_busy_fraction"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (avg["GRBM_GUI_ACTIVE"] / 8.0)
    out[k[:100]] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
