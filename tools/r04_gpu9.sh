#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
bash tools/profile_gpu.sh r04_final > gpurun_out/r04_profile_final.log 2>&1; tail -25 gpurun_out/r04_profile_final.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_final_bench_steps20_warmup5.json 2> gpurun_out/r04_final_bench.err || tail -20 gpurun_out/r04_final_bench.err
python bench.py > gpurun_out/r04_final_bench.json 2>> gpurun_out/r04_final_bench.err || tail -20 gpurun_out/r04_final_bench.err
python bench.py --episodes 2000 --minutes 45 --device-synth --steps 10 --warmup 2 > gpurun_out/r04_final_library_2000.json 2> gpurun_out/r04_final_library_2000.err || tail -20 gpurun_out/r04_final_library_2000.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_lib2000_trace -- python3 $GRAFT_REPO_ROOT/bench.py --episodes 2000 --minutes 45 --device-synth --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04_lib2000_trace.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r04_lib2000_trace -name "*kernel_stats.csv" | head -2
find $GRAFT_REPO_ROOT/gpurun_out/r04_lib2000_trace -name "*_kernel_trace.csv" -size +2M -delete; find $GRAFT_REPO_ROOT/gpurun_out/r04_lib2000_trace -name "*.db" -delete
