#!/bin/bash
for p in 0 1 0 1; do
  DDN_BENCH_PRIO=$p python bench.py --no-cpu-baseline --no-extras 2>/dev/null > /tmp/b.json
  python - "$p" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print("prio", sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["launch_ms"])
PY
done
