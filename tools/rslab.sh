# resampler tuning / ablations on the GPU box (48 kHz stereo line of tools/bench_resample.py)
# NEEDLE_HIP_RESAMPLE_SKEW: start skew unit (x 8 128 cycles); LAB: 1 no staging, 2 no FMA loop, 3 neither
for skew in 0 1 2 3; do echo "skew unit $skew"; NEEDLE_HIP_RESAMPLE_SKEW=$skew timeout -k 10 100 python tools/bench_resample.py 2>&1 | head -1; done
for lab in 1 2 3; do echo "LAB $lab"; NEEDLE_HIP_RESAMPLE_LAB=$lab timeout -k 10 100 python tools/bench_resample.py 2>&1 | head -1; done
