"""Per-kernel HBM-side traffic (bytes past L2, Infinity-Cache hits included) from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE), corrected as MI355X_MICROARCH.md prescribes (KiB units; FETCH_SIZE doubled on gfx950).
usage: pmc_kernel_bytes.py <fetch_dir> <write_dir> <out.json> [substring ...]"""
import collections, csv, glob, json, sys
def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"]
                agg[k[:k.find("(")] if "(" in k else k].append(float(r["Counter_Value"]))
    return agg
fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
want = sys.argv[4:] or [""]
out = {}
for k in sorted(fetch):
    if not any(w in k for w in want): continue
    f = fetch[k]; w = write.get(k, [0.0])
    out[k] = {"launches": len(f), "fetch_bytes_avg": 2.0 * 1024.0 * sum(f) / len(f), "write_bytes_avg": 1024.0 * sum(w) / max(1, len(w))}
    out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_avg"] + out[k]["write_bytes_avg"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print("%8.1f MB/launch  n=%5d  %s" % (v["hbm_bytes_per_launch"] / 1e6, v["launches"], k[:110]))
