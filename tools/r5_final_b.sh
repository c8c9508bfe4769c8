# round 5, final tree: BASELINE.json configs[4] at full size on one GPU (bench line, kernel trace), scan counters
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( while sleep 50; do echo "[r5_final_b] $(date +%T) still running"; done ) &
HB=$!
timeout -k 10 500 python bench.py --episodes 2000 --minutes 45 --device-synth --steps 3 --warmup 2 > gpurun_out/r5/library_2000.json 2> gpurun_out/r5/library_2000.err; echo "library 2000 rc=$?"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05_library_2000 -- python3 $GRAFT_REPO_ROOT/bench.py --episodes 2000 --minutes 45 --device-synth --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r5/library_2000_traced.json 2> $GRAFT_REPO_ROOT/gpurun_out/r5/library_2000_traced.err; echo "traced rc=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r05_library_2000 -name "*_kernel_trace.csv" -size +2M -delete; find gpurun_out/prof_r05_library_2000 -name "*.db" -delete
bash tools/scan_mfma_counters.sh 400 2>&1 | tail -4 | tee gpurun_out/r5/counters_final.log
kill $HB
