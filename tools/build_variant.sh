#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "EXTRA FLAGS": a second library dsd-neo_amd/libdsdneo_hip_NAME.so in which one source is
# compiled with extra flags (timing experiments: -DDDN_RX_CYCLES=1 ...), everything else taken from the normal build's objects.
# Use with DDN_LIB_PATH=dsd-neo_amd/libdsdneo_hip_NAME.so (bindings/ddn.py).
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; EXTRA=$3
B=$HERE/dsd-neo_amd/build; V=$HERE/dsd-neo_amd/build/variant_$NAME
mkdir -p "$V"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I"$HERE/include" $EXTRA \
    -c "$HERE/dsd-neo_amd/csrc/$FILE" -o "$V/$FILE.o"
OBJS=$(ls "$B"/*.o | grep -v "/$FILE.o\$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/dsd-neo_amd/libdsdneo_hip_$NAME.so" $OBJS "$V/$FILE.o" -lm -L/opt/rocm/lib -lhsa-runtime64
echo "built dsd-neo_amd/libdsdneo_hip_$NAME.so"
