#!/bin/bash
# Library-scale run of bench.py on ONE GPU (BASELINE.json configs[4] is 2000 x 45 min over 8 GPUs): E episodes of
# 45 min, analyze + all-pairs search, 3 timed jobs behind 3 warm-up jobs (third and fourth argument; the first jobs of a library grow the run-list buffers, which costs 150 ms of hipHostFree alone).  Needs ~60 MB of host memory per episode for the synthetic PCM.
E=${1:-1000}
cd "$(dirname "$0")/.."
free -g | sed -n 2p
start=$(date +%s)
# the synthesis of the episodes is silent for minutes: a heartbeat keeps a supervised run (gpurun) from looking hung
( while sleep 60; do echo "[library_scale] $(( $(date +%s) - start )) s: still running" >&2; done ) &
HEARTBEAT=$!
timeout ${2:-1500} python bench.py --episodes "$E" --minutes 45 --steps ${3:-3} --warmup ${4:-3} --no-cpu-baseline --no-extras
rc=$?
kill $HEARTBEAT 2>/dev/null
echo "exit $rc after $(( $(date +%s) - start )) s"
