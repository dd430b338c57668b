#!/bin/bash
python -m pytest tests/test_e2e_voice.py -x -q -m gpu 2>&1 | tail -2
for ns in 2 3; do for dbg in 0 32768; do
  echo "== streams=$ns rx_dbg=$dbg"
  DDN_RX_DBG=$dbg DDN_BENCH_STREAMS=$ns python bench.py --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'])"
done; done
