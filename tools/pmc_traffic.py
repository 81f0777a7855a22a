#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs).
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> blocks block_size data order > json
Counters are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.  Per kernel the
average over its working launches (within a factor ten of its largest; the chain kernels are launched once per
LDS size class and all but one exit at once)."""
import csv, sys, json, collections

def per_launch(path, counter):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Dispatch_Id"])] += float(r["Counter_Value"]) * 1024.0
    byk = collections.defaultdict(list)
    for (k, d), v in agg.items():
        byk[k].append(v)
    return byk

fetch = per_launch(sys.argv[1], "FETCH_SIZE")
write = per_launch(sys.argv[2], "WRITE_SIZE")
out = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate run, --pmc WRITE_SIZE) --output-format csv "
              "-- python bench.py --steps 1 --warmup 1 --no-cpu; counters are KB; FETCH_SIZE is doubled per "
              "MI355X_MICROARCH.md (gfx950 reports half of a streaming read); average per working launch "
              "(launches within a factor ten of the kernel's largest)",
       "workload": {"blocks": int(sys.argv[3]), "block_size": int(sys.argv[4]), "data": sys.argv[5], "order": int(sys.argv[6])},
       "kernels": {}}
def working(vals):          # the launches that did the work: within a factor ten of the largest
    m = max(vals) if vals else 0.0
    return [v for v in vals if v > 1e6 and v >= 0.1 * m]

for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"): continue
    f = working(fetch.get(k, []))
    w = working(write.get(k, []))
    if not f and not w: continue
    fr = sum(f) / len(f) if f else 0.0
    wr = sum(w) / len(w) if w else 0.0
    out["kernels"][k] = {"fetch_bytes_raw": int(fr), "fetch_bytes_corrected": int(2 * fr), "write_bytes": int(wr),
                         "traffic_bytes": int(2 * fr + wr), "working_launches": max(len(f), len(w))}
print(json.dumps(out, indent=1))
