#!/bin/bash
# SQ counters of k_fsk4_rx at the bench shape (run on the GPU box through gpurun)
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM" "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/prx4_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/prx4_$i -o p -- python $R/tools/bench_rx4.py > /tmp/prx4_$i.log 2>&1
  f=$(find /tmp/prx4_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
t = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_fsk4_rx" in r["Kernel_Name"]:
        t[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(t.items()):
    print(f"{k:24s} mean {sum(v)/len(v):16.0f} first6 {[int(x) for x in v[:6]]}")
PY
  else tail -3 /tmp/prx4_$i.log; fi
done
