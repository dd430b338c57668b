#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in C B; do
  cp $R/build_variants/$v.so $R/dsd-neo_amd/libdsdneo_hip.so
  for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM; do
    rm -rf /tmp/pa; MODES=handlers:8 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pa -o p -- python $R/tools/bench_rx_handlers.py > /dev/null 2>&1
    f=$(find /tmp/pa -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - $f $v $c <<'PY'
import csv,sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_p25_rxw" in r["Kernel_Name"]]
print(sys.argv[2], sys.argv[3], "launches", len(v), "mean %.4g"%(sum(v)/max(1,len(v))))
PY
  done
done
