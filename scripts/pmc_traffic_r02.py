"""HBM-side traffic per launch SITE from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py, corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes (counter unit KiB; on gfx950 FETCH_SIZE reports half of a wide
coalesced read stream -> doubled; WRITE_SIZE exact for 16-B-per-lane stores).  Bytes past L2 (Infinity-Cache hits are
counted, not excluded).  Dispatches are labelled with the library's site names (scripts/kernel_sites.py).
usage: pmc_traffic_r02.py <fetch_dir> <write_dir> <out.json> <workload string>"""
import collections, csv, glob, hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sites import label_passes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_src_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nano-vllm-go_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".h", ".hip")):
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def load(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append(r)
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    label_passes(rows)
    agg = collections.defaultdict(list)
    names = collections.defaultdict(set)
    for r in rows:
        if "site" in r:
            agg[(r["phase"], r["site"])].append(float(r["Counter_Value"]))
            names[(r["phase"], r["site"])].add(r["kname"][:100])
    return agg, names


fetch, names = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
sites = {}
for key in sorted(fetch):
    f = fetch[key]; w = write.get(key, [0.0])
    fb, wb = 2.0 * 1024.0 * sum(f) / len(f), 1024.0 * sum(w) / max(1, len(w))
    sites[f"{key[0]}/{key[1]}"] = {"launches": len(f), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                                   "hbm_bytes_per_launch": fb + wb, "kernels": sorted(names[key])}
res = {"note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes; separate --pmc passes; bytes past L2 per launch, "
               "Infinity-Cache hits included",
       "workload": sys.argv[4] if len(sys.argv) > 4 else "", "kernel_src_sha16": kernel_src_sha16(), "sites": sites}
json.dump(res, open(sys.argv[3], "w"), indent=1)
for k, v in sites.items():
    print("%9.2f MB/launch (fetch %9.2f write %8.2f)  n=%5d  %-22s %s" % (v["hbm_bytes_per_launch"] / 1e6, v["fetch_bytes_per_launch"] / 1e6,
          v["write_bytes_per_launch"] / 1e6, v["launches"], k, v["kernels"][0][:70]))
