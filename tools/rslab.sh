# resampler tuning / ablations on the GPU box (48 kHz stereo line of tools/bench_resample.py)
# NEEDLE_HIP_RESAMPLE_SPLITS: workgroups per tile (rows cut along their outputs); NEEDLE_HIP_RESAMPLE_SKEW: start skew
# unit (x 8 128 cycles); NEEDLE_HIP_RESAMPLE_LAB: 1 no staging, 2 no FMA loop, 3 neither (wrong results)
for sp in 1 2 3 4 6; do echo "splits $sp"; NEEDLE_HIP_RESAMPLE_SPLITS=$sp timeout -k 10 100 python tools/bench_resample.py 2>&1 | head -1; done
# the wrong-result variants exist only in a lab build of the library
VARIANT_SRC=resample bash "$(dirname "$0")/build_variant.sh" rslab -DNEEDLE_HIP_LAB_BUILD > /dev/null
export NEEDLE_CAPI_LIB=needle_amd/lib/ab/rslab.so
for lab in 1 2 3; do echo "LAB $lab"; NEEDLE_HIP_RESAMPLE_LAB=$lab timeout -k 10 100 python tools/bench_resample.py 2>&1 | head -1; done
