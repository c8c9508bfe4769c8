export TMPDIR=/tmp
mkdir -p gpurun_out/r5
( while sleep 50; do echo "[r5_gpu40] $(date +%T) still running"; done ) &
HB=$!
timeout -k 10 900 python -m pytest tests/test_gpu_library_scale.py tests/test_gpu_epilogue.py tests/test_gpu_multi.py -x -q 2>&1 | tail -4 | tee gpurun_out/r5/epi40.log
NEEDLE_HIP_DEVICE_EPILOGUE=1 timeout -k 10 600 python tools/fuzz_epilogue.py 600 82 2>&1 | tail -2 | tee gpurun_out/r5/fuzz_epi40.log
timeout -k 10 300 python bench.py --episodes 2000 --minutes 45 --device-synth --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/r5/library_2000_epi.json; python tools/brief.py < gpurun_out/r5/library_2000_epi.json | cut -c1-300
kill $HB
