"""Timeline of the last pass in a rocprofv3 kernel trace CSV: start / end (us, relative) of every kernel longer than 20 us."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
rows = [r for r in rows if t_end - int(r["Start_Timestamp"]) < int(float(sys.argv[2]) * 1e6)] if len(sys.argv) > 2 else rows
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if b - a > 20000:
        print("%9.1f %9.1f  %8.1f us  q%s grid %s lds %s  %s" % (a / 1e3, b / 1e3, (b - a) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:70]))
