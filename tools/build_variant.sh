#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "EXTRA FLAGS": a second library dsd-neo_amd/libdsdneo_hip_NAME.so in which one source is
# compiled with extra flags (timing experiments: -DDDN_RX_CYCLES=1 ...), everything else taken from the normal build's objects.
# tools/build_variant.sh exp: the whole library with -DDDN_EXPERIMENTS (the environment knobs of DESIGN / profiles/README.md live;
# the product build reads no environment at all) -> dsd-neo_amd/libdsdneo_hip_exp.so.
# Use with DDN_LIB_PATH=dsd-neo_amd/libdsdneo_hip_NAME.so (bindings/ddn.py).
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; EXTRA=$3
if [ "$NAME" = exp ] && [ -z "$FILE" ]; then
    make -s -C "$HERE/dsd-neo_amd" -j8 OBJDIR="$HERE/dsd-neo_amd/build/variant_exp" OUT="$HERE/dsd-neo_amd/libdsdneo_hip_exp.so" EXTRA=-DDDN_EXPERIMENTS
    echo "built dsd-neo_amd/libdsdneo_hip_exp.so"
    exit 0
fi
B=$HERE/dsd-neo_amd/build; V=$HERE/dsd-neo_amd/build/variant_$NAME
mkdir -p "$V"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I"$HERE/include" $EXTRA \
    -c "$HERE/dsd-neo_amd/csrc/$FILE" -o "$V/$FILE.o"
OBJS=$(ls "$B"/*.o | grep -v "/$FILE.o\$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/dsd-neo_amd/libdsdneo_hip_$NAME.so" $OBJS "$V/$FILE.o" -lm -L/opt/rocm/lib -lhsa-runtime64
echo "built dsd-neo_amd/libdsdneo_hip_$NAME.so"
