#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline"
for cfg in "base:0:0:0" "shareprio:1:0:1" "share:1:0:0"; do
  IFS=: read name share lds prio <<< "$cfg"
  NEEDLE_HIP_STFT_SHARE=$share NEEDLE_HIP_STFT_LDS_BYTES=$lds NEEDLE_HIP_LIBRARY_PRIORITY=$prio $B > gpurun_out/r04_share_$name.json 2> gpurun_out/r04_share_$name.err || { echo "$name failed"; tail -5 gpurun_out/r04_share_$name.err; }
  python - "$name" gpurun_out/r04_share_$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('cold',{}) and d['cold']['ms_per_step'], d['kernel_ms_per_step'], d['detected'])
PY
done
