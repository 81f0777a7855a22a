"""Every encode pass of a rocprofv3 kernel-trace CSV (a pass starts at a k_enc_front): the chain launches that took at
least min_ms, with start, duration, queue, workgroups and LDS.  Usage: trace_passes.py kernel_trace.csv [min_ms] [enc|dec]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 1e6
what = sys.argv[3] if len(sys.argv) > 3 else "enc"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
front = "k_enc_front" if what == "enc" else "k_dec_front"
chain = "k_enc_chain" if what == "enc" else "k_dec_chain"
starts = [i for i, r in enumerate(rows) if front in r["Kernel_Name"]]
for n, i in enumerate(starts):
    t0 = int(rows[i]["Start_Timestamp"])
    j = starts[n + 1] if n + 1 < len(starts) else len(rows)
    ks = [r for r in rows[i:j] if chain in r["Kernel_Name"]]
    if not ks: continue
    end = max(int(r["End_Timestamp"]) for r in ks)
    print("pass %d: chain kernels end %.2f ms after the front's start" % (n, (end - t0) / 1e6))
    for r in ks:
        a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        if b - a >= min_ns:
            wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
            print("   %8.2f +%7.2f ms  q%-2s wgs %-6d lds %-6s %s" % (a / 1e6, (b - a) / 1e6, r.get("Queue_Id"), wg, r.get("LDS_Block_Size"), r["Kernel_Name"].replace("void ", "").split("(")[0]))
