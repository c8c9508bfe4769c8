#!/bin/bash
# Runs on the GPU box (via gpurun): kernel + memory-copy timeline of bench.py's timed jobs from a rocprofv3 trace -- every
# dispatch / copy longer than MIN_US (default 50) with its queue and the device-wide idle time in front of it, for the last
# FRACTION (default 0.25) of the run.
# Usage: tools/bench_trace.sh <bench.py arguments>   -> gpurun_out/bench_trace/
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/bench_trace
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" "$@" --no-cpu-baseline --no-extras > "$OUT/run.log" 2>&1
python3 - "$OUT" "${MIN_US:-50}" "${FRACTION:-0.25}" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1][:44], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "").replace("MEMORY_COPY_", ""), ""))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
tail = [r for r in rows if r[0] - t0 > (1 - float(sys.argv[3])) * (t1 - t0)]
busy_until = tail[0][0]
min_ns = float(sys.argv[2]) * 1e3
small_n = small_t = 0
for s, e, n, q in tail:
    idle = max(0, s - busy_until)
    if e - s > min_ns or idle > min_ns:
        if small_n:
            print(f"{'':24s}     ({small_n} shorter operations, {small_t / 1e6:.3f} ms in all)")
            small_n = small_t = 0
        print(f"{(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f} ms  dur {(e - s) / 1e6:8.3f}  idle before {idle / 1e6:7.3f}  {n} {q}")
    else:
        small_n += 1
        small_t += e - s
    busy_until = max(busy_until, e)
PY
find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
