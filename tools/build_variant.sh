#!/bin/bash
# Builds needle_amd/lib/ab/<name>.so: libneedle_capi.so with extra -D flags on fingerprint.hip, or on the .hip file
# named by VARIANT_SRC (kernel A/B timing:
# NEEDLE_CAPI_LIB=needle_amd/lib/ab/<name>.so python bench.py ...).  Usage: tools/build_variant.sh <name> [-DFLAG ...]
set -e
NAME=$1; shift
cd "$(dirname "$0")/../needle_amd/csrc"
make -s
mkdir -p ../lib/ab ../../build/ab
SRC=${VARIANT_SRC:-fingerprint}
/opt/rocm/bin/hipcc -w -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include --offload-arch=gfx950 "$@" -c $SRC.hip -o ../../build/ab/${SRC}_$NAME.o
OBJS=$(ls ../../build/csrc/*.o | grep -v /$SRC.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/ab/$NAME.so $OBJS ../../build/ab/${SRC}_$NAME.o -Wl,-soname,libneedle_capi.so
echo built needle_amd/lib/ab/$NAME.so
