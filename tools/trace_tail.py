"""kernels of the last `ms` milliseconds of a rocprofv3 --kernel-trace run (csv), longer than `min_ms`: start, end, ms, queue, wg, lds, name
    python tools/trace_tail.py <dir> [ms] [min_ms]"""
import csv
import glob
import os
import sys

d, span, min_ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 30.0, float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")),
                     r.get("LDS_Block_Size", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), name))
rows.sort()
t_end = rows[-1][1]
t0 = t_end - int(span * 1e6)
print("%9s %9s %7s  %-5s %-5s %-7s %-9s %s" % ("start", "end", "ms", "queue", "wg", "lds", "grid", "kernel"))
for s, e, q, w, lds, g, name in rows:
    if e >= t0 and (e - s) / 1e6 >= min_ms:
        print("%9.3f %9.3f %7.3f  %-5s %-5s %-7s %-9s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, w, lds, g, name))
