#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r04_full_gpu_suite.log 2>&1; rc=$?; tail -6 gpurun_out/r04_full_gpu_suite.log; [ $rc -eq 0 ] || exit 1
