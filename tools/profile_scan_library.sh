#!/bin/bash
# Runs on the GPU box (via gpurun): the scan kernel where it dominates -- E episodes x 45 min of synthetic AUDIO generated
# in HBM (tools/library_device.py) -- under rocprofv3: kernel trace + stats, then separate PMC passes.
# Usage: tools/profile_scan_library.sh <tag> [episodes=1000]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-scanlib}; E=${2:-1000}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROG="$REPO/tools/library_device.py $E 3 3"
python3 $PROG > "$OUT/plain.json" 2> "$OUT/plain.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $PROG > "$OUT/stats.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_sq" -- python3 $PROG > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- python3 $PROG > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $PROG > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $PROG > "$OUT/pmc_write.log" 2>&1
python3 "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/summary.md" 2>&1
cat "$OUT/summary.md"
find "$OUT" -name "*_kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
