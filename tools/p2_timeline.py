#!/usr/bin/env python3
"""Kernel timeline of the last ddn_p25p2_groups_batch call in a rocprofv3 --kernel-trace csv (argv[1]): duration, start offset, name."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in rows]
idx = max(i for i, n in enumerate(names) if "k_p2_rows" in n)
t0 = int(rows[idx]["Start_Timestamp"])
for r, n in list(zip(rows, names))[idx:]:
    print("%9.1f us  +%8.1f  %-28s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3, n[:28]))
