"""Raw kernel order of the last steady-state steps of a rocprofv3 --kernel-trace run (csv): start, end, queue, grid, workgroup, name.
    python tools/step_trace.py <dir> [n_steps]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")),
                     r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), name))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "-", "-", "-", "-", "-", "COPY " + r.get("Direction", "?")[12:]))
rows.sort()
fe = [i for i, r in enumerate(rows) if r[7].startswith("k_front_end")]
lo = fe[-(n_steps + 1)]
t0 = rows[lo][0]
print("%9s %9s %7s  %-6s %-9s %-5s %-6s %-5s %s" % ("start", "end", "ms", "queue", "grid", "wg", "lds", "vgpr", "kernel"))
for s, e, q, g, w, lds, vg, name in rows[max(lo - 3, 0):fe[-1]]:
    print("%9.3f %9.3f %7.3f  %-6s %-9s %-5s %-6s %-5s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, g, w, lds, vg, name))
