mkdir -p gpurun_out/r5
export TMPDIR=/tmp
set -o pipefail
timeout -k 10 120 tools/mfma_fp4_probe 2>&1 | tee gpurun_out/r5/fp4_probe.log &&
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "sampled-mfma or library or config" 2>&1 | tail -3 | tee gpurun_out/r5/parity30.log &&
timeout -k 10 600 python tools/scan_mfma_sweep.py 400 45 w8 2>&1 | grep variant | cut -c1-170 | tee gpurun_out/r5/sweep30.log &&
timeout -k 10 600 python tools/scan_mfma_sweep.py 280 24 w8 2>&1 | grep variant | cut -c1-170 | tee -a gpurun_out/r5/sweep30.log
