# Is the box power-limited?  Samples the hwmon power / cap / live clocks of card 0 while a resampler variant runs in a
# loop (NEEDLE_HIP_RESAMPLE_REPEAT), and once when idle.  Lab build for the ablations (tools/rs_mfma_lab.sh).
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
show() {
  for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input temp1_input; do
    [ -r "$HW/$f" ] && printf "%s=%s " $f "$(cat $HW/$f)"
  done
  echo
}
echo "hwmon: $HW"; echo -n "idle: "; show
/opt/rocm/bin/rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -v "^=\|^$" | head -12
export NEEDLE_CAPI_LIB=needle_amd/lib/ab/rslab.so
for lab in 0 3 4 15; do
  echo "-- LAB $lab (0 = the product kernel)"
  NEEDLE_HIP_RESAMPLE_REPEAT=5000 NEEDLE_HIP_RESAMPLE_LAB=$lab timeout -k 10 100 python tools/bench_resample.py > /tmp/probe_$lab.log 2>&1 &
  BENCH=$!
  sleep 7
  for k in 1 2 3 4 5; do show; sleep 0.25; done
  wait $BENCH || true
  grep "48000 Hz x2" /tmp/probe_$lab.log
done
