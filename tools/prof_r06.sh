#!/bin/bash
# Round 6: kernel stats + HBM / instruction counters of the bench step (rings [channel][slot], mixed chain with one front-end launch),
# the mixed step's kernel stats and kernel order, the chains round 5 built (P25-CQPSK, Phase 2, M17, YSF) at batch scale, the loop with
# the matched filter inside against the filter kernel, the front end's phase table on clean and noisy input.  writes gpurun_out/prof_r06/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r06; mkdir -p $OUT
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
for tag in bench mixed chains; do
  case $tag in
    bench) CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"; CL="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES";;
    mixed) CMD="python $R/tools/bench_mixed.py 4096 12"; CL="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU";;
    chains) CMD="python $R/tools/bench_chains.py all"; CL="";;
  esac
  rm -rf /tmp/pr_$tag; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$tag -o b -- $CMD > $OUT/${tag}_under_trace.log 2>&1
  f=$(find /tmp/pr_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r06_${tag}_kernel_stats.csv
  f=$(find /tmp/pr_$tag -name "*kernel_trace.csv" | head -1)
  if [ $tag = bench ]; then python3 $R/tools/trace_overlap.py $f k_front_end 12 > $OUT/r06_resident_step_kernel_order.txt 2>&1; fi
  if [ $tag = mixed ]; then python3 $R/tools/trace_overlap.py $f k_front_end 9 > $OUT/r06_mixed_step_kernel_order.txt 2>&1; fi
  for c in $CL; do
    rm -rf /tmp/pr_${tag}_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pr_${tag}_$c -o p -- $CMD > $OUT/${tag}_pmc_$c.log 2>&1
    f=$(find /tmp/pr_${tag}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && summ $f $c > $OUT/r06_${tag}_pmc_$c.txt
  done
done
cd $R
grep -h "^{" $OUT/chains_under_trace.log > $OUT/r06_chains_bench.jsonl
# the loop: the filter kernel + loop against the filter inside the loop (ddn_p25_rx_set_filter_in_loop), three traffic kinds
for t in mixed voice ctrl; do
  for f in 0 1; do echo -n "k_p25_rxw, traffic $t, filter_in_loop=$f: "; TRAFFIC=$t FIL=$f MODES=handlers:8 timeout 200 python tools/bench_rx_handlers.py 2>&1 | grep loop | head -1; done
done > $OUT/r06_filter_in_loop_ab.txt
timeout 200 python tools/bench_fe_segments.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_front_end_segments.txt
if [ -f dsd-neo_amd/libdsdneo_hip_exp.so ]; then
  for k in clean noise; do echo "== front end, 4096 x 48000, $k input: cycles per 256-sample tile and wave (0-7 filter waves, 8 = dc wave S1, 9 = peak wave S2)"; DDN_LIB_PATH=dsd-neo_amd/libdsdneo_hip_exp.so DDN_DBG=64 DDN_DBG_PRINT=1 timeout 200 python tools/fe_phase_times.py $k 2>&1 | grep "^wave" | tail -10; done > $OUT/r06_front_end_phase_table.txt
fi
head -6 $OUT/r06_bench_kernel_stats.csv | cut -c1-150
cat $OUT/r06_filter_in_loop_ab.txt
