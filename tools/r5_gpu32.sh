mkdir -p gpurun_out/r5
export TMPDIR=/tmp
set -o pipefail
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "sampled-mfma or library or config" 2>&1 | tail -3 | tee gpurun_out/r5/parity32.log &&
NEEDLE_HIP_MFMA_NO_IMAGES=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "sampled-mfma" 2>&1 | tail -3 | tee -a gpurun_out/r5/parity32.log &&
NEEDLE_HIP_SCAN_MFMA=1 timeout -k 10 500 python tools/fuzz_search.py 150 70 2>&1 | tail -2 | tee gpurun_out/r5/fuzz_search_mfma32.log &&
timeout -k 10 600 python tools/scan_mfma_sweep.py 280 24 w8,m2lab1@w8,m2lab2@w8,m2lab4@w8 2>&1 | grep variant | cut -c1-120 | tee gpurun_out/r5/sweep32.log &&
timeout -k 10 600 python tools/scan_mfma_sweep.py 400 45 w8,m2lab1@w8,m2lab2@w8,m2lab4@w8 2>&1 | grep variant | cut -c1-120 | tee -a gpurun_out/r5/sweep32.log
