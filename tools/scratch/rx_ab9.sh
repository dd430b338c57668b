#!/bin/bash
for v in head new; do
  cp tools/scratch/lib_$v.so dsd-neo_amd/libdsdneo_hip.so
  for sp in "" 1; do
    echo "== lib=$v spread=$sp"
    DDN_BENCH_SPREAD=$sp python tools/bench_rx.py 4096 48000 8 16 32 2>/dev/null | grep "^{" | grep '"matched_filter": 1' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['cpw'], j['matched_filter'], round(j['ms'], 3), j['symbols'], j['syncs'])"
  done
done
