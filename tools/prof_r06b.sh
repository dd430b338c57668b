#!/bin/bash
# Round 6, after the loop stopped staging the unfiltered row of channels in sync: kernel stats + HBM counters of the bench step.
# writes gpurun_out/prof_r06b/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r06b; mkdir -p $OUT
summ() { # counter csv -> per-kernel mean
python3 - "$1" "$2" <<'PY'
import csv, sys, collections
t = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    t[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]].append(float(r["Counter_Value"]))
print("# counter", sys.argv[2], "per launch (mean over launches), unit as rocprofv3 reports it")
for k, v in sorted(t.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith("k_") or "k_" in k[:12]:
        print("%-62s launches %4d  mean %16.1f  total %18.1f" % (k, len(v), sum(v) / len(v), sum(v)))
PY
}
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rm -rf /tmp/pr_b; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_b -o b -- $CMD > $OUT/bench_under_trace.log 2>&1
f=$(find /tmp/pr_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r06b_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pr_b_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pr_b_$c -o p -- $CMD > $OUT/bench_pmc_$c.log 2>&1
  f=$(find /tmp/pr_b_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && summ $f $c > $OUT/r06b_bench_pmc_$c.txt
done
head -6 $OUT/r06b_bench_kernel_stats.csv | cut -c1-150
grep rxw $OUT/r06b_bench_pmc_*.txt
# the CQPSK-side chains after the AGC / FLL kernel's rework (AGC on the helper wave, packed taps, product sums inside the chain)
rm -rf /tmp/pr_ch; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_ch -o b -- python $R/tools/bench_chains.py all > $OUT/chains_under_trace.log 2>&1
f=$(find /tmp/pr_ch -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r06b_chains_kernel_stats.csv
grep -h "^{" $OUT/chains_under_trace.log > $OUT/r06b_chains_bench.jsonl
cut -c1-700 $OUT/r06b_chains_bench.jsonl
