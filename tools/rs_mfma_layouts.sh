# wave-role layouts of the matrix-core resampler (NEEDLE_HIP_RESAMPLE_LAYOUT=0: blocks first, staging waves last; default:
# by SIMD, see resample.hip): parity and timing
export NEEDLE_HIP_RESAMPLE_REPEAT=2
for layout in 1 0; do
  echo "---- layout $layout"
  NEEDLE_HIP_RESAMPLE_LAYOUT=$layout timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k resampler 2>&1 | tail -1
  NEEDLE_HIP_RESAMPLE_LAYOUT=$layout timeout -k 10 100 python tools/bench_resample.py 2>&1
done
