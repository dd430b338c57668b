#!/bin/bash
# Collect SQ counter groups for k_front_end_fused, one rocprofv3 pass per group (PMC passes must not be combined with
# tracing domains other than --kernel-trace).  usage: tools/pmc_groups.sh <out_prefix>
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p "$OUT"
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  DDN_NO_TORCH=1 timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$i -o p -- \
      python $R/tools/pmc_front_end.py 2 > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    head -1 "$f" > "$OUT/${1}_grp$i.csv"
    grep k_front_end_fused "$f" >> "$OUT/${1}_grp$i.csv"
  else
    echo "group $i failed"; tail -3 /tmp/pmc_$i.log
  fi
done
python3 - "$OUT" "$1" <<'PY'
import csv, glob, sys, collections
out, pre = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in sorted(glob.glob(f"{out}/{pre}_grp*.csv")):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
disp = {}
for k in tot:
    disp[k] = tot[k] / max(1, cnt[k])
for k in sorted(disp):
    print(f"{k:28s} {disp[k]:18.0f}  (rows {cnt[k]})")
PY
