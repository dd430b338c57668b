#!/bin/bash
# Round 5, second half (the receive loop at 4.7 ms): the bench step's kernel stats + the HBM / instruction counters again, and the
# loop's cycle tables (instrumented library dsd-neo_amd/libdsdneo_hip_cyc.so = tools/build_variant.sh cyc ddn_rx.hip -DDDN_RX_CYCLES=1).
# writes gpurun_out/prof_r05b/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r05b; mkdir -p $OUT
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
tag=bench
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rm -rf /tmp/pr_$tag; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$tag -o b -- $CMD > $OUT/${tag}_under_trace.log 2>&1
f=$(find /tmp/pr_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r05_${tag}_kernel_stats.csv
python3 $R/tools/step_trace.py /tmp/pr_$tag 2 > $OUT/r05_resident_step_kernel_order.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES; do
  rm -rf /tmp/pr_${tag}_$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pr_${tag}_$c -o p -- $CMD > $OUT/${tag}_pmc_$c.log 2>&1
  f=$(find /tmp/pr_${tag}_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && summ $f $c > $OUT/r05_${tag}_pmc_$c.txt
done
cd $R
for t in mixed voice ctrl; do
  echo "== traffic: $t" >> $OUT/r05_rxw_cycle_tables.txt
  TRAFFIC=$t DDN_LIB_PATH=dsd-neo_amd/libdsdneo_hip_cyc.so DDN_RX_DBG=8192 MODES=handlers:8 timeout 200 python tools/bench_rx_handlers.py 2>&1 | grep -v amdgpu.ids | tail -14 >> $OUT/r05_rxw_cycle_tables.txt
  TRAFFIC=$t MODES=handlers:8 timeout 200 python tools/bench_rx_handlers.py 2>&1 | tail -1 | sed 's/^/   (uninstrumented library) /' >> $OUT/r05_rxw_cycle_tables.txt
done
echo "== decisions on the handler wave (DDN_RX_DBG=65536, cycles per decision by kind; 196608: stage stamps)" >> $OUT/r05_rxw_cycle_tables.txt
DDN_RX_DBG=65536 MODES=handlers:8 timeout 200 python tools/bench_rx_handlers.py 2>&1 | tail -1 >> $OUT/r05_rxw_cycle_tables.txt
DDN_RX_DBG=196608 MODES=handlers:8 timeout 200 python tools/bench_rx_handlers.py 2>&1 | tail -1 >> $OUT/r05_rxw_cycle_tables.txt
DDN_LIB_PATH=dsd-neo_amd/libdsdneo_hip_cyc.so DDN_RX_DBG=1073807360 MODES=handlers:8 timeout 200 python tools/bench_rx_handlers.py 2>&1 | tail -1 >> $OUT/r05_rxw_cycle_tables.txt
timeout 300 python tools/loops_side_by_side.py 2>&1 | grep -v amdgpu.ids > $OUT/r05_loops_side_by_side.txt
FSK4_CPW=4 timeout 300 python tools/loops_side_by_side.py 2>&1 | grep -v amdgpu.ids | sed 's/^/(fsk4 loops at 4 channels per wave, the mixed chain'"'"'s shape) /' >> $OUT/r05_loops_side_by_side.txt
head -12 $OUT/r05_${tag}_kernel_stats.csv | cut -c1-150
cat $OUT/r05_rxw_cycle_tables.txt | head -60
