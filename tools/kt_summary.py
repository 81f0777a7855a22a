#!/usr/bin/env python3
"""Per-launch durations of the last pass in a rocprofv3 kernel trace: tools/kt_summary.py <kernel_trace.csv> [min_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
t0 = min(int(r['Start_Timestamp']) for r in rows)
out = []
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d >= thr and not r['Kernel_Name'].startswith('void at::'):
        out.append(((int(r['Start_Timestamp']) - t0) / 1e6, d, r.get('Queue_Id', '?'), r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('LDS_Block_Size', '?'), r['Kernel_Name'][:60]))
for o in out[-int(sys.argv[3]) if len(sys.argv) > 3 else -24:]:
    print("%10.2f %8.2f q%s grid %s lds %s %s" % o)
