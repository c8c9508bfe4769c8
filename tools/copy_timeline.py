#!/usr/bin/env python3
"""Merges a rocprofv3 kernel trace and memory-copy trace into one device timeline (last N entries)."""
import csv
import sys

ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:30] or "kernel"))
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", "copy ")))
ev.sort()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ev = ev[-n:]
t0 = ev[0][0]
for a, b, name in ev:
    print(f"{name:32s} {(a - t0) / 1e3:10.1f} -> {(b - t0) / 1e3:10.1f}  ({(b - a) / 1e3:9.1f} us)")
