#!/bin/bash
cp dsd-neo_amd/libdsdneo_hip.so /tmp/lib_product.so
cp tools/scratch/lib_prof.so dsd-neo_amd/libdsdneo_hip.so
for d in "$@"; do for sp in "" 1; do echo "== extra dbg $d spread=$sp"; DDN_BENCH_SPREAD=$sp python tools/scratch/rx_cyc.py 16 $((8192 + d)) 2>&1 | grep "cpw" | grep -v "loader\|winprep"; done; done
cp /tmp/lib_product.so dsd-neo_amd/libdsdneo_hip.so
