#!/usr/bin/env python3
"""Prints the device timeline of the last jobs in a rocprofv3 --kernel-trace CSV (start / end relative to the first
listed dispatch, in microseconds), to see what overlaps what."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{name:28s} queue {r.get('Queue_Id', '?'):>3s}  {a:9.1f} -> {b:9.1f}  ({b - a:7.1f} us)")
