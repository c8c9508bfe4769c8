#!/bin/bash
# The run list behind a round's profiles/rNN_final_* files, in stages (one gpurun call each: a call is capped at 20 min).
# Usage (on the GPU box, from the repo root):  bash tools/round_final.sh <round tag, e.g. r06> <stage> [...]
#   suite    the GPU test-suite (pytest -m gpu)
#   bench    bench.py in the driver's form (--steps 20 --warmup 5) and in its default form
#   profile  tools/profile_gpu.sh <tag>_final: rocprofv3 kernel trace + the separate PMC passes of the driver's form
#   library  BASELINE.json configs[4] at full size on one GPU: bench line, kernel trace, the scan's counters at 79 800 pairs
#   fuzz     the fuzzers on the final tree (GPU against the oracle), each with the seed printed in its log
# Every stage writes under gpurun_out/<tag>/; what is to be judged is copied into profiles/ by hand afterwards.
set -u
TAG=${1:?round tag}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
( while sleep 50; do echo "[round_final] $(date +%T) still running"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
for stage in "$@"; do
  case $stage in
    suite)
      timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$OUT/gputest_final.log" ;;
    bench)
      timeout -k 10 400 python bench.py --steps 20 --warmup 5 > "$OUT/bench_final_steps20_warmup5.json" 2> "$OUT/bench_final.err"; echo "bench rc=$?"
      timeout -k 10 400 python bench.py > "$OUT/bench_final_default.json" 2>> "$OUT/bench_final.err"; echo "bench default rc=$?" ;;
    profile)
      bash tools/profile_gpu.sh "${TAG}_final" --steps 20 --warmup 5 2>&1 | tail -30 ;;
    library)
      timeout -k 10 500 python bench.py --episodes 2000 --minutes 45 --device-synth --steps 3 --warmup 2 > "$OUT/library_2000.json" 2> "$OUT/library_2000.err"; echo "library 2000 rc=$?"
      ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_${TAG}_library_2000" -- python3 "$ROOT/bench.py" --episodes 2000 --minutes 45 --device-synth --steps 3 --warmup 2 --no-cpu-baseline --no-extras > "$ROOT/$OUT/library_2000_traced.json" 2> "$ROOT/$OUT/library_2000_traced.err"; echo "traced rc=$?" )
      find "gpurun_out/prof_${TAG}_library_2000" -name "*_kernel_trace.csv" -size +2M -delete; find "gpurun_out/prof_${TAG}_library_2000" -name "*.db" -delete
      bash tools/scan_mfma_counters.sh 400 2>&1 | tail -4 | tee "$OUT/counters_final.log" ;;
    fuzz)
      NEEDLE_HIP_SCAN_MFMA=1 timeout -k 10 500 python tools/fuzz_search.py 400 161 2>&1 | tail -3 | tee "$OUT/fuzz_search_mfma.log"
      timeout -k 10 300 python tools/fuzz_search.py 200 162 2>&1 | tail -2 | tee "$OUT/fuzz_search.log"
      NEEDLE_HIP_SCAN_MFMA=1 timeout -k 10 400 python tools/fuzz_pipeline.py 200 163 2>&1 | tail -2 | tee "$OUT/fuzz_pipeline_mfma.log"
      timeout -k 10 400 python tools/fuzz_fingerprint.py 1000 164 2>&1 | tail -2 | tee "$OUT/fuzz_fingerprint.log"
      NEEDLE_HIP_DEVICE_EPILOGUE=1 timeout -k 10 300 python tools/fuzz_epilogue.py 400 165 2>&1 | tail -2 | tee "$OUT/fuzz_epilogue.log" ;;
    *) echo "unknown stage $stage"; exit 2 ;;
  esac
done
