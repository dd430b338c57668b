#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" ${PMC_MORE:+"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"}; do
  i=$((i+1)); rm -rf /tmp/prx_$i
  DDN_NO_TORCH=1 timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/prx_$i -o p -- python $R/tools/pmc_rx.py ${CPW:-16} ${FR:-864} > /tmp/prx_$i.log 2>&1
  f=$(find /tmp/prx_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
t = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_p25_rx" in r["Kernel_Name"]:
        t[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(t.items()):
    print(f"{k:24s} {sum(v)/len(v):16.0f} (launches {len(v)})")
PY
  else tail -3 /tmp/prx_$i.log; fi
done
