#!/bin/bash
# Round 5, last part (row-parallel hunting passes in both loops: k_p25_rxw 4.6 ms, NXDN48 loop 3.4 ms at four channels per wave):
# the bench step's kernel stats + HBM / instruction / instruction-cache counters, the mixed step's kernel stats and kernel order, the
# three loops side by side.  writes gpurun_out/prof_r05c/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r05c; mkdir -p $OUT
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
for tag in bench mixed; do
  if [ $tag = bench ]; then CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"; CL="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQC_ICACHE_MISSES SQC_ICACHE_REQ"; else CMD="python $R/tools/bench_mixed.py 4096 8"; CL="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVES"; fi
  rm -rf /tmp/pr_$tag; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$tag -o b -- $CMD > $OUT/${tag}_under_trace.log 2>&1
  f=$(find /tmp/pr_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r05_${tag}_kernel_stats.csv
  f=$(find /tmp/pr_$tag -name "*kernel_trace.csv" | head -1)
  if [ $tag = bench ]; then python3 $R/tools/step_trace.py /tmp/pr_$tag 2 > $OUT/r05_resident_step_kernel_order.txt 2>&1; else python3 $R/tools/trace_overlap.py $f > $OUT/r05_mixed_step_kernel_order.txt 2>&1; fi
  for c in $CL; do
    rm -rf /tmp/pr_${tag}_$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pr_${tag}_$c -o p -- $CMD > $OUT/${tag}_pmc_$c.log 2>&1
    f=$(find /tmp/pr_${tag}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && summ $f $c > $OUT/r05_${tag}_pmc_$c.txt
  done
done
cd $R
timeout 300 python tools/loops_side_by_side.py 2>&1 | grep -v amdgpu.ids > $OUT/r05_loops_side_by_side.txt
FSK4_CPW=4 timeout 300 python tools/loops_side_by_side.py 2>&1 | grep -v amdgpu.ids | sed 's/^/(fsk4 loops at 4 channels per wave, the mixed chain'"'"'s shape) /' >> $OUT/r05_loops_side_by_side.txt
for t in mixed voice ctrl; do
  for d in 0 4096; do echo -n "k_p25_rxw, traffic $t, DDN_RX_DBG=$d (4096 = hunting passes one owner at a time): "; TRAFFIC=$t DDN_RX_DBG=$d MODES=handlers:8 timeout 200 python tools/bench_rx_handlers.py 2>&1 | grep loop | head -1; done
done > $OUT/r05_hunting_pass_ab.txt
for c in 4 2; do for d in 0 65536; do echo "k_fsk4_rx at $c channels per wave, DDN_RX4_DBG=$d (65536 = hunting passes one owner at a time):"; FSK4_CPW=$c DDN_RX4_DBG=$d HANDLERS=1 timeout 200 python tools/bench_rx4.py 1365 2>&1 | grep loop; done; done >> $OUT/r05_hunting_pass_ab.txt
head -8 $OUT/r05_bench_kernel_stats.csv | cut -c1-150
cat $OUT/r05_hunting_pass_ab.txt
