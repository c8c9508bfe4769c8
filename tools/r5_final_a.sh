# round 5, final tree: GPU suite, then the driver's form of the bench under rocprofv3 (kernel trace + PMC passes)
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r5/gputest_final.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench_final_steps20_warmup5.json 2> gpurun_out/r5/bench_final.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py > gpurun_out/r5/bench_final_default.json 2>> gpurun_out/r5/bench_final.err; echo "bench default rc=$?"
bash tools/profile_gpu.sh r05_final --steps 20 --warmup 5 2>&1 | tail -30
