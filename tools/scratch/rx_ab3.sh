#!/bin/bash
# which part bounds the loop now: 256 = no window-suffix wave, 512 = no drain (slice + record stores), both are timing-only (wrong results)
for d in 0 256 512 768; do
  echo "== DDN_RX_DBG=$d"
  DDN_RX_DBG=$d python tools/bench_rx.py 4096 48000 8 16 2>&1 | grep '"matched_filter": 1' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['cpw'], round(j['ms'], 3), j['symbols'], j['syncs'])"
done
