#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline"
for cfg in "base::" "share35:1:0" "share48:1:49152" "share56:1:57344" "share64:1:65536" "noshare56:0:57344"; do
  IFS=: read name share lds <<< "$cfg"
  NEEDLE_HIP_STFT_SHARE=$share NEEDLE_HIP_STFT_LDS_BYTES=$lds $B > gpurun_out/r04_share_$name.json 2> gpurun_out/r04_share_$name.err || { echo "$name failed"; tail -5 gpurun_out/r04_share_$name.err; }
  python - "$name" gpurun_out/r04_share_$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('cold',{}) and d['cold']['ms_per_step'], d['kernel_ms_per_step'], d['detected'])
PY
done
LAB_ONLY=0 LAB_LDS_BYTES=57344 timeout -k 10 120 ./tools/stft32_lab 20 | tail -2
LAB_ONLY=0 LAB_LDS_BYTES=81920 timeout -k 10 120 ./tools/stft32_lab 20 | tail -2
