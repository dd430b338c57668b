#!/bin/bash
for v in head new; do
  cp tools/scratch/lib_$v.so dsd-neo_amd/libdsdneo_hip.so
  for d in 0; do
    echo "== lib=$v DDN_RX_DBG=$d"
    DDN_RX_DBG=$d python tools/bench_rx.py 4096 48000 8 16 32 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['cpw'], j['matched_filter'], round(j['ms'], 3), j['symbols'], j['syncs'])"
  done
done
