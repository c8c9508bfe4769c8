#!/bin/bash
# Wall-clock of the `needle` CLI, one process per command, over a WAV library made by tools/bench_files.py.
D=${1:-/tmp/needle_files/28x24min_11025_mono}
B=$(dirname "$0")/../needle_amd/bin/needle
TIMEFORMAT="%R s wall, %U user, %S sys"
for i in 1 2; do echo -n "analyze --force: "; { time $B analyze --force "$D" > /dev/null; } 2>&1; done
for i in 1 2; do echo -n "search --no-display: "; { time $B search --no-display "$D" > /dev/null; } 2>&1; done
echo -n "search (display): "; { time $B search "$D" > /tmp/needle_search.out; } 2>&1; tail -3 /tmp/needle_search.out
echo -n "info: "; { time $B info > /dev/null; } 2>&1
