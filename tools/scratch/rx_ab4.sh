#!/bin/bash
for rep in 1 2; do
for d in 0 4096 3072; do
  echo "== DDN_RX_DBG=$d"
  DDN_RX_DBG=$d python tools/bench_rx.py 4096 48000 8 16 2>&1 | grep '"matched_filter": 1' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['cpw'], round(j['ms'], 3), j['symbols'], j['syncs'])"
done
done
