"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-launch HBM traffic of the
prefill GEMM class, corrected as MI355X_MICROARCH.md §HBM prescribes (units KiB; on gfx950 FETCH_SIZE counts
exactly half of a wide coalesced read stream -> doubled).  usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys
def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"]
                k = k[:k.find("(")] if "(" in k else k
                agg[k].append(float(r["Counter_Value"]))
    return agg
fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
tot_b = tot_n = 0
for k in sorted(fetch):
    if "gemm_bf16_kernel" not in k and "gemm_bf16_pp_kernel" not in k: continue
    f = fetch[k]; w = write.get(k, [0.0])
    per = (2.0 * sum(f) / len(f) + sum(w) / max(1, len(w))) * 1024.0
    out[k] = {"launches": len(f), "fetch_KiB_raw_avg": sum(f) / len(f), "write_KiB_avg": sum(w) / max(1, len(w)),
              "hbm_bytes_per_launch": per}
    tot_b += per * len(f); tot_n += len(f)
res = {"note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes; separate --pmc passes of bench.py --steps 1 --warmup 0",
       "prefill_gemm_class_avg_bytes_per_launch": tot_b / max(1, tot_n), "launches": tot_n, "kernels": out}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: (v if k != "kernels" else {kk: round(vv["hbm_bytes_per_launch"] / 1e6, 1) for kk, vv in v.items()}) for k, v in res.items()}, indent=1))
