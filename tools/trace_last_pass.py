"""Kernels of the LAST encode + decode pass in a rocprofv3 kernel-trace CSV (from the last k_enc_front on): start, end,
duration (ms, relative), hardware queue, workgroups, kernel.  Usage: trace_last_pass.py kernel_trace.csv [min_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1e5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_enc_front" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void ", "")
    if b - a >= min_ns and name.startswith("k"):
        wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
        print("%8.2f %8.2f %8.2f ms  q%-2s wgs %-6d lds %-6s %s" % (a / 1e6, b / 1e6, (b - a) / 1e6, r.get("Queue_Id"), wg, r.get("LDS_Block_Size"), name.split("(")[0]))
