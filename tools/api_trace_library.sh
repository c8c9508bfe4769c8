#!/bin/bash
# On the GPU box: HIP API + kernel trace of a library-scale bench run, then the longest HIP API calls (which call of a
# job's enqueue blocks, and for how long).  Usage: tools/api_trace_library.sh [episodes=400]
E=${1:-400}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/apitrace_$E
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
start=$(date +%s)
( while sleep 60; do echo "[api_trace] $(( $(date +%s) - start )) s: still running" >&2; done ) &
HB=$!
rocprofv3 --hip-trace --kernel-trace --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --episodes "$E" --minutes 45 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/run.log" 2>&1
kill $HB 2>/dev/null
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
calls = []
for f in glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        calls.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], int(r["Start_Timestamp"])))
calls.sort(reverse=True)
print("longest HIP API calls (ms):")
for d, fn, t0 in calls[:25]:
    print("  %9.3f  %s" % (d / 1e6, fn))
tot = collections.Counter()
for d, fn, _ in calls:
    tot[fn] += d
print("total per function (ms):")
for fn, d in tot.most_common(12):
    print("  %9.3f  %s" % (d / 1e6, fn))
PY
find "$OUT" -name "*.csv" -size +8M -delete; find "$OUT" -name "*.db" -delete
