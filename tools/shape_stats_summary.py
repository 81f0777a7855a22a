#!/usr/bin/env python3
"""gpurun_out/shape_stats/<shape>/ (tools/shape_stats.sh) -> profiles/<tag>_shape_<shape>_kernel_stats.csv: per kernel the
working launches (tools/summarize_trace.py's split) of one timed sweep.py run (1 warm step + 2 timed ones)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "shape_stats", "*/"))):
    name = os.path.basename(d.rstrip("/"))
    tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not tr:
        continue
    tr.sort(key=os.path.getmtime, reverse=True)          # gpurun merges into gpurun_out/: earlier runs' files stay
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_trace.py"), tr[0], "100"], check=True,
                         stdout=subprocess.PIPE, text=True).stdout
    open(os.path.join(ROOT, "profiles", f"{tag}_shape_{name}_kernel_stats.csv"), "w").write(out)
    print("==", name); print("\n".join(out.splitlines()[:9]))
