#!/usr/bin/env python3
"""Host API calls (rocprofv3 --hip-trace) and device activity (--kernel-trace, --memory-copy-trace) on one time axis:
when was each launch CALLED and when did it RUN.  Args: hip_api_trace.csv kernel_trace.csv [first_ms last_ms]."""
import csv
import sys

api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host  " + r["Function"]) for r in csv.DictReader(open(sys.argv[1]))
       if r["Function"] in ("hipLaunchKernel", "hipModuleLaunchKernel", "hipEventSynchronize", "hipMemcpyAsync", "hipStreamSynchronize",
                            "hipEventRecord", "hipMemsetAsync", "hipExtModuleLaunchKernel")]
ker = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU   " + (r["Kernel_Name"].split("(")[0].split("::")[-1][:26] or "kernel"))
       for r in csv.DictReader(open(sys.argv[2]))]
ev = sorted(api + ker)
t0 = ker[0][0]
lo = float(sys.argv[3]) if len(sys.argv) > 3 else -1e18
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e18
for a, b, name in ev:
    ms = (a - t0) / 1e6
    if lo <= ms <= hi:
        print(f"{ms:10.3f} ms  +{(b - a) / 1e3:10.1f} us  {name}")
