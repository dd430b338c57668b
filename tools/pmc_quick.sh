#!/bin/bash
# tools/pmc_quick.sh TAG "COUNTER ..." -- CMD...: one rocprofv3 --pmc pass per counter of CMD, per-kernel means into
# gpurun_out/pmc_TAG_<COUNTER>.txt (a pass with --kernel-trace only, as gpurun requires)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CL=$2; shift 3
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for c in $CL; do
  rm -rf /tmp/pq_${TAG}_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pq_${TAG}_$c -o p -- "$@" > /tmp/pq_${TAG}_$c.log 2>&1
  f=$(find /tmp/pq_${TAG}_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$c" > $R/gpurun_out/pmc_${TAG}_$c.txt <<'PY'
import csv, sys, collections
t = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    t[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]].append(float(r["Counter_Value"]))
print("# counter", sys.argv[2], "per launch (mean over launches), unit as rocprofv3 reports it")
for k, v in sorted(t.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith("k_") or "k_" in k[:12]:
        print("%-62s launches %4d  mean %16.1f  total %18.1f" % (k, len(v), sum(v) / len(v), sum(v)))
PY
done
