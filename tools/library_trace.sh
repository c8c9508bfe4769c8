#!/bin/bash
# Runs on the GPU box (via gpurun): kernel + memory-copy timeline of the library-scale job (tools/library_device.py) from a
# rocprofv3 trace -- every dispatch / copy longer than 50 us with its queue and the idle time in front of it.
# Usage: tools/library_trace.sh [episodes=1000]   -> gpurun_out/library_trace/
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
E=${1:-1000}
OUT=$REPO/gpurun_out/library_trace
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -- python3 "$REPO/tools/library_device.py" $E 3 3 > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), ""))
rows.sort()
t0 = rows[0][0]
tail = [r for r in rows if r[0] - t0 > 0.55 * (rows[-1][1] - t0)]   # the pipelined jobs at the end of the run
busy_until = tail[0][0]
for s, e, n, q in tail:
    if e - s > 50_000:
        print(f"{(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f} ms  dur {(e - s) / 1e6:8.3f}  idle before {max(0, s - busy_until) / 1e6:7.3f}  {n} {q}")
    busy_until = max(busy_until, e)
PY
find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
