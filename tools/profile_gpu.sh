#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats, then separate PMC passes, of bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{stats,pmc_*}/
set -u
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras $*"   # bench.py defaults: 50 timed steps, 10 warm-up (+ 11 untimed breakdown steps)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
python3 "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/summary.md" 2>&1
cat "$OUT/summary.md"
# keep the merged-back payload small: drop the raw per-dispatch traces, keep stats + summaries
find "$OUT" -name "*_kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
