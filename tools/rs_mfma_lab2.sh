# ablations of the matrix-core resampler (wrong results on purpose; lab build): 1 no global loads, 2 no LDS writes, 4 no MFMA
# loop, 8 no output stores
export NEEDLE_CAPI_LIB=needle_amd/lib/ab/rslab.so
for lab in 0 1 2 3 4 7 8 12 15; do
  echo "---- LAB $lab"
  NEEDLE_HIP_RESAMPLE_LAB=$lab timeout -k 10 100 python tools/bench_resample.py 2>&1 | grep "48000 Hz x2"
done
