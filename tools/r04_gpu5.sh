#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "fast_kernels or sampled or hamming" > gpurun_out/r04_t_scan.log 2>&1 || { tail -40 gpurun_out/r04_t_scan.log; exit 1; }
tail -2 gpurun_out/r04_t_scan.log
python tools/scan_shape_sweep.py 1000 45 > gpurun_out/r04_scan_shape_sweep_1000x45.log 2>&1; echo "sweep rc=$?"; cat gpurun_out/r04_scan_shape_sweep_1000x45.log
