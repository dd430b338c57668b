#!/bin/bash
# HERE (no GPU): build a profiling copy of the library (DDN_RX_CYCLES=1) as tools/scratch/lib_prof.so
set -e
R=/root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$R/include -DDDN_RX_CYCLES=1 -c $R/dsd-neo_amd/csrc/ddn_rx.hip -o /tmp/ddn_rx_prof.o
objs=$(ls $R/dsd-neo_amd/build/*.o | grep -v ddn_rx.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/scratch/lib_prof.so $objs /tmp/ddn_rx_prof.o -lm
