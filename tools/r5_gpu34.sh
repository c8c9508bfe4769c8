mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( while sleep 50; do echo "[r5_gpu34] $(date +%T) still running"; done ) &
HB=$!
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r5/gputest_fp4.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_fp4_steps20.json 2> gpurun_out/r5/bench_fp4.err; echo "bench rc=$?"
timeout -k 10 400 python bench.py --episodes 2000 --minutes 45 --device-synth --steps 3 --warmup 2 > gpurun_out/r5/library_2000_fp4.json 2> gpurun_out/r5/library_2000_fp4.err; echo "library 2000 rc=$?"
kill $HB
