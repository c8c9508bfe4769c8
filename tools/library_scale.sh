#!/bin/bash
# Library-scale run of bench.py on ONE GPU (BASELINE.json configs[4] is 2000 x 45 min over 8 GPUs): E episodes of
# 45 min, analyze + all-pairs search, 2 timed jobs (third argument: another number of them).  Needs ~60 MB of host memory per episode for the synthetic PCM.
E=${1:-1000}
cd "$(dirname "$0")/.."
free -g | sed -n 2p
start=$(date +%s)
# the synthesis of the episodes is silent for minutes: a heartbeat keeps a supervised run (gpurun) from looking hung
( while sleep 60; do echo "[library_scale] $(( $(date +%s) - start )) s: still running" >&2; done ) &
HEARTBEAT=$!
timeout ${2:-1500} python bench.py --episodes "$E" --minutes 45 --steps ${3:-2} --warmup 1 --no-cpu-baseline
rc=$?
kill $HEARTBEAT 2>/dev/null
echo "exit $rc after $(( $(date +%s) - start )) s"
