#!/bin/bash
# kernel stats of the cqpsk/p2 chains
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_c -o b -- python $R/tools/bench_chains.py cqpsk_p2 > /tmp/pr_c.log 2>&1
tail -1 /tmp/pr_c.log | cut -c250-800
f=$(find /tmp/pr_c -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %4s avg %9.3f ms"%(r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:60], r['Calls'], float(r['AverageNs'])/1e6))
PY
