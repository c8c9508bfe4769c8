#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the secondary measurements
# (search-only at configs[2] scale, resampler).  Usage: tools/profile_secondary.sh <tag>
set -u
TAG=${1:-run}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof2_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/search" -- python3 "$REPO/tools/bench_search_only.py" > "$OUT/search.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/resample" -- python3 "$REPO/tools/bench_resample.py" > "$OUT/resample.log" 2>&1
for d in search resample; do
  echo "## $d"; cat "$OUT/$d.log" | grep -v "^\[" | tail -8
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1)
  echo; echo '```'; head -8 "$f"; echo '```'
done > "$OUT/summary.md"
cat "$OUT/summary.md"
find "$OUT" -name "*_kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
