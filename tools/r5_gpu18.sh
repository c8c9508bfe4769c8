mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r5/gputest18.log
