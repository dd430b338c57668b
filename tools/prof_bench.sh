#!/bin/bash
# rocprofv3 kernel trace + HBM / instruction counters of the bench step (run on the GPU box through gpurun; counters in passes of their
# own, kernel trace only beside them); writes gpurun_out/prof_r03/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r03; mkdir -p $OUT
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rm -rf /tmp/pb_trace; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_trace -o b -- $CMD > $OUT/bench_under_trace.log 2>&1
f=$(find /tmp/pb_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r03_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVES; do
  rm -rf /tmp/pb_$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pb_$c -o p -- $CMD > $OUT/pmc_$c.log 2>&1
  f=$(find /tmp/pb_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" $c > $OUT/r03_pmc_$c.txt <<'PY'
import csv, sys, collections
t = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    t[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
print("# counter", sys.argv[2], "per launch (mean over launches), unit as rocprofv3 reports it")
for k, v in sorted(t.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s launches %4d  mean %16.1f  total %18.1f" % (k, len(v), sum(v) / len(v), sum(v)))
PY
  fi
done
tail -2 $OUT/bench_under_trace.log | cut -c1-300
head -12 $OUT/r03_bench_kernel_stats.csv
