#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_epilogue.py -x -q > gpurun_out/r04_t_epilogue.log 2>&1 || { tail -60 gpurun_out/r04_t_epilogue.log; exit 1; }
tail -2 gpurun_out/r04_t_epilogue.log
NEEDLE_HIP_DEVICE_EPILOGUE=1 python -m pytest tests/test_gpu_multi.py -x -q -k "job_api or host_transport or endings or hash_sharding" > gpurun_out/r04_t_multi_dev.log 2>&1 || { tail -60 gpurun_out/r04_t_multi_dev.log; exit 1; }
tail -2 gpurun_out/r04_t_multi_dev.log
python -m pytest tests/test_gpu_library_scale.py -x -q -s -k "ranks or config4_full" > gpurun_out/r04_t_lib_dev.log 2>&1 || { tail -60 gpurun_out/r04_t_lib_dev.log; exit 1; }
tail -3 gpurun_out/r04_t_lib_dev.log
NEEDLE_HIP_TRACE=1 python tools/library_device.py 2000 3 2 45 > gpurun_out/r04_lib2000_devepi.json 2> gpurun_out/r04_lib2000_devepi.err; tail -1 gpurun_out/r04_lib2000_devepi.json; grep -a "epilogue" gpurun_out/r04_lib2000_devepi.err | tail -5
NEEDLE_HIP_DEVICE_EPILOGUE=0 NEEDLE_HIP_TRACE=1 python tools/library_device.py 2000 3 2 45 > gpurun_out/r04_lib2000_hostepi.json 2> gpurun_out/r04_lib2000_hostepi.err; tail -1 gpurun_out/r04_lib2000_hostepi.json; grep -a "epilogue" gpurun_out/r04_lib2000_hostepi.err | tail -5
