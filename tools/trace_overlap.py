#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last steady step, the kernels in start order with start / end
relative to the step, so that what runs beside what (several HIP streams) can be seen.
usage: trace_overlap.py kernel_trace.csv [first_kernel_regex] [step index, default -3 = the third last]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = re.compile(sys.argv[2] if len(sys.argv) > 2 else "k_front_end")
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:44], r.get("Stream_Id", r.get("Queue_Id", ""))) for r in rows))
starts = [i for i, e in enumerate(ev) if first.search(e[2])]
# steps: group the first-kernel launches that lie close together
groups = []
for i in starts:
    if not groups or ev[i][0] - ev[groups[-1][-1]][0] > 3_000_000:
        groups.append([i])
    else:
        groups[-1].append(i)
if len(groups) < 3:
    print("too few steps", len(groups))
    sys.exit(0)
idx = int(sys.argv[3]) if len(sys.argv) > 3 else -3
a, b = groups[idx][0], groups[idx + 1][0]
t0 = ev[a][0]
print("step %.3f ms" % ((ev[b][0] - t0) / 1e6))
for s, e, nme, q in ev[a:b]:
    if (e - s) > 20000:
        print("%8.3f %8.3f  %7.3f ms  q=%s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, nme))
