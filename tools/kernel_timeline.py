#!/usr/bin/env python3
"""Timeline of the LAST `count` kernel dispatches of a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous end.
usage: kernel_timeline.py <kernel_trace.csv> [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-count:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else st - prev_end
    name = r["Kernel_Name"]
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("chip::", "")
    name = name[:name.find("(")] if "(" in name else name
    print(f"{(st - t0) / 1e3:9.1f} us  dur {(en - st) / 1e3:7.1f}  gap {gap / 1e3:7.1f}  grid {r.get('Grid_Size', '?'):>8s} wg {r.get('Workgroup_Size', '?'):>5s}  {name[:100]}")
    prev_end = en
