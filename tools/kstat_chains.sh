#!/bin/bash
# tools/kstat_chains.sh: rocprofv3 kernel stats of tools/bench_chains.py cqpsk_p2 (the P25 CQPSK chain and the Phase 2 chain at 1365
# and 4096 channels): average per kernel, and every launch of the symbol-rate loop (its time at the two batch sizes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_c -o b -- python $R/tools/bench_chains.py cqpsk_p2 > /tmp/pr_c.log 2>&1
f=$(find /tmp/pr_c -name "*kernel_stats.csv" | head -1)
t=$(find /tmp/pr_c -name "*kernel_trace.csv" | head -1)
python3 - $f $t <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %4s avg %9.3f ms"%(r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:60], r['Calls'], float(r['AverageNs'])/1e6))
d=[(int(r['Start_Timestamp']), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',0))) for r in csv.DictReader(open(sys.argv[2])) if 'k_cq_rx' in r['Kernel_Name']]
d.sort()
print("k_cq_rx launches (grid size: ms):", ", ".join("%d: %.2f"%(g,ms) for _,ms,g in d))
PY
