#!/bin/bash
# round 4, first GPU session: the new multi-rank library-scale test, the audit, bench entries
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_library_scale.py -x -q -s -k "ranks" > gpurun_out/r04_t_ranks.log 2>&1 || { tail -60 gpurun_out/r04_t_ranks.log; exit 1; }
echo "ranks ok"; tail -3 gpurun_out/r04_t_ranks.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err || { tail -30 gpurun_out/r04_bench_a.err; exit 1; }
echo "bench ok"
python bench.py --episodes 2000 --minutes 45 --device-synth --steps 10 --warmup 2 > gpurun_out/r04_lib2000.json 2> gpurun_out/r04_lib2000.err || { tail -30 gpurun_out/r04_lib2000.err; exit 1; }
echo "bench 2000 ok"
NEEDLE_HIP_COMM=host python bench.py --gpus 4 --episodes 2000 --minutes 45 --device-synth --steps 6 --warmup 2 --no-extras --no-cpu-baseline --launch-timeout 500 > gpurun_out/r04_lib2000_g4.json 2> gpurun_out/r04_lib2000_g4.err || { tail -30 gpurun_out/r04_lib2000_g4.err; exit 1; }
echo "bench 2000 x4 ranks ok"
