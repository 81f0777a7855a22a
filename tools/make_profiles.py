#!/usr/bin/env python3
"""Turn gpurun_out/refresh/ (tools/refresh_profiles.sh, run on the GPU box) into the tracked summaries under
profiles/:  <tag>_final_bench.json, <tag>_final_kernel_stats.csv (rocprofv3's own --stats table),
<tag>_final_working_launches.csv (tools/summarize_trace.py), <tag>_pmc_traffic.json (HBM bytes per launch from the
FETCH_SIZE / WRITE_SIZE passes + the SQ counters of the chain kernels, keyed to the hash of the library that ran).
usage: make_profiles.py <tag, e.g. r02>"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1]


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {SRC}")
    return hits[0]


bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
with open(os.path.join(DST, f"{tag}_final_bench.json"), "w") as f:
    f.write(json.dumps(bench) + "\n")
shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(DST, f"{tag}_final_kernel_stats.csv"))
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_trace.py"), one("stats/**/*kernel_trace.csv")],
                     check=True, stdout=subprocess.PIPE, text=True).stdout
open(os.path.join(DST, f"{tag}_final_working_launches.csv"), "w").write(out)

cfg = bench["config"]
data = cfg["workload"].split(" B ")[1].split(" ")[0]
traffic = json.loads(subprocess.run(
    [sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), one("fetch/**/*counter_collection.csv"),
     one("write/**/*counter_collection.csv"), str(cfg["blocks_per_gpu"]), str(cfg["block_size"]), data, str(cfg["order"])],
    check=True, stdout=subprocess.PIPE, text=True).stdout)
traffic["library_sha256_16"] = bench["roofline"]["library_sha256_16"]

# SQ counters of the working launch of each chain kernel (the largest dispatch), summed over the passes
sq = collections.defaultdict(dict)
for d in ("sq1", "sq2"):
    hits = glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True)
    if not hits:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(hits[0])):
        k = r["Kernel_Name"]
        if "chain" not in k:
            continue
        agg[(k.split("(")[0].replace("void ", "").split("<")[0], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    byk = collections.defaultdict(list)
    for (k, _), v in agg.items():
        byk[k].append(v)
    for k, l in byk.items():
        best = max(l, key=lambda v: v.get("SQ_WAVE_CYCLES", 0) + v.get("SQ_BUSY_CYCLES", 0) + v.get("SQ_WAVES", 0))
        sq[k].update({n: int(x) for n, x in best.items()})
for k, v in sq.items():
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        # SQ_WAVE_CYCLES counts in units of 4 cycles per wave on this part; ratios are unit-free
        if "SQ_INSTS_VALU" in v:
            v["valu_issue_frac"] = round(v["SQ_INSTS_VALU"] / wc, 4)          # VALU instructions per wave quad-cycle
        if "SQ_WAIT_INST_ANY" in v:
            v["wait_frac"] = round(v["SQ_WAIT_INST_ANY"] / wc, 4)
    if v.get("SQ_INSTS_LDS"):
        if "SQ_LDS_BANK_CONFLICT" in v:
            v["lds_conflict_cycles_per_inst"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_INSTS_LDS"], 3)
traffic["sq"] = sq
traffic["sq_how"] = ("rocprofv3 --kernel-trace --pmc <SQ counters> (own runs, no other tracing) -- python bench.py --steps 1 "
                     "--warmup 1 --no-cpu --no-host; the working launch of each chain kernel")
with open(os.path.join(DST, f"{tag}_pmc_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
os.makedirs(os.path.join(DST, "pmc"), exist_ok=True)
shutil.copy(one("fetch/**/*counter_collection.csv"), os.path.join(DST, "pmc", f"{tag}_fetch_counter_collection.csv"))
shutil.copy(one("write/**/*counter_collection.csv"), os.path.join(DST, "pmc", f"{tag}_write_counter_collection.csv"))
if os.path.exists(os.path.join(SRC, "hetero.jsonl")):
    with open(os.path.join(DST, f"{tag}_hetero.jsonl"), "w") as f:
        for name in ("hetero.jsonl", "hetero_4g.jsonl"):
            if os.path.exists(os.path.join(SRC, name)):
                f.write(open(os.path.join(SRC, name)).read())
if os.path.exists(os.path.join(SRC, "hetero_sched_trace.txt")):
    with open(os.path.join(DST, f"{tag}_hetero_sched_trace.txt"), "w") as f:
        f.write("".join(l for l in open(os.path.join(SRC, "hetero_sched_trace.txt")) if "Warning" not in l and "d_base" not in l and "amdgpu.ids" not in l))
x8 = glob.glob(os.path.join(SRC, "x8stats", "**", "*kernel_stats.csv"), recursive=True)
if x8:
    shutil.copy(x8[0], os.path.join(DST, f"{tag}_4x8_kernel_stats.csv"))
for name, dst in (("batch_sweep.jsonl", f"{tag}_batch_sweep.jsonl"), ("shapes.jsonl", f"{tag}_shapes.jsonl"),
                  ("sq_small_summary.txt", f"{tag}_sq_one_block.txt"), ("rate_4x8.txt", f"{tag}_4x8_rates.txt"),
                  ("host_probe.txt", f"{tag}_host_probe.txt"), ("host_batch_rate.txt", f"{tag}_host_batch_rate.txt")):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, dst))
print(json.dumps({k: v for k, v in traffic["kernels"].items()}, indent=1))
print(json.dumps(sq, indent=1))
