#!/bin/bash
# Builds needle_amd/lib/ab/<name>.so: libneedle_capi.so with extra -D flags on fingerprint.hip (kernel A/B timing:
# NEEDLE_CAPI_LIB=needle_amd/lib/ab/<name>.so python bench.py ...).  Usage: tools/build_variant.sh <name> [-DFLAG ...]
set -e
NAME=$1; shift
cd "$(dirname "$0")/../needle_amd/csrc"
make -s
mkdir -p ../lib/ab ../../build/ab
/opt/rocm/bin/hipcc -w -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include --offload-arch=gfx950 "$@" -c fingerprint.hip -o ../../build/ab/fingerprint_$NAME.o
OBJS=$(ls ../../build/csrc/*.o | grep -v fingerprint.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/ab/$NAME.so $OBJS ../../build/ab/fingerprint_$NAME.o -Wl,-soname,libneedle_capi.so
echo built needle_amd/lib/ab/$NAME.so
