#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 ./tools/issue_rates > gpurun_out/r04_issue_rates.log 2>&1; echo "issue rates rc=$?"
timeout -k 10 600 ./tools/stft32_lab 30 > gpurun_out/r04_stft32_lab.log 2>&1; echo "lab rc=$?"; cat gpurun_out/r04_stft32_lab.log
python -m pytest tests/test_gpu_library_scale.py -x -q -s -k "config3_at or streamed_from_pinned" > gpurun_out/r04_t_gaps.log 2>&1 || { tail -60 gpurun_out/r04_t_gaps.log; exit 1; }
echo "gap tests ok"; tail -3 gpurun_out/r04_t_gaps.log
