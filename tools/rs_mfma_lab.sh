# matrix-core resampler (needle_amd/csrc/resample_mfma.h): parity first, then the 48 kHz lines of tools/bench_resample.py,
# then the ablations of the lab build (wrong results on purpose): 1 no global loads, 2 no LDS writes, 4 no MFMA loop,
# 8 no output stores, 16 the phases of block 0 by s_memtime, 32 (with 1) opaque values in place of the loads
# NEEDLE_HIP_RESAMPLE_REPEAT=2: the timed launch follows two launches on the resident input
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k resampler 2>&1 | tail -3
export NEEDLE_HIP_RESAMPLE_REPEAT=2
echo "---- DPP kernel (NEEDLE_HIP_RESAMPLE_QUAD=1)"
NEEDLE_HIP_RESAMPLE_QUAD=1 timeout -k 10 100 python tools/bench_resample.py 2>&1 | grep 48000
echo "---- matrix-core kernel"
timeout -k 10 100 python tools/bench_resample.py 2>&1
export NEEDLE_CAPI_LIB=needle_amd/lib/ab/rslab.so
for lab in ${LABS:-3 4 33 2 15}; do
  echo "---- LAB $lab"
  NEEDLE_HIP_RESAMPLE_LAB=$lab timeout -k 10 100 python tools/bench_resample.py 2>&1 | grep "48000 Hz x2\|ticks per tile\|block 0:" | tail -5
done
