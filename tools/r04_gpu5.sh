#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "fast_kernels or sampled or hamming or pairs_too_long or window_of" > gpurun_out/r04_t_scan.log 2>&1 || { tail -40 gpurun_out/r04_t_scan.log; exit 1; }
tail -2 gpurun_out/r04_t_scan.log
timeout -k 10 400 python tools/fuzz_search.py 150 11 | tail -1
python tools/library_device.py 1000 3 2 45 | tail -1
python bench.py --steps 50 --warmup 10 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step']['hamming_runs'], d['roofline_search']['frac'], d['search_only']['scan_kernel_ms'], d['search_only']['roofline']['frac'])"
