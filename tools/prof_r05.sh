#!/bin/bash
# Round-5 rocprofv3 evidence (run on the GPU box through gpurun; counters in passes of their own beside --kernel-trace only):
#   1. the bench step: kernel stats + FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU / SQ_WAVES per kernel
#   2. the configs[3] mixed chain (tools/bench_mixed.py): kernel stats + the same counters (k_fsk4_rx DMR / NXDN48, k_mbe_* on AMBE)
#   3. tools/bench_stages.py: kernel stats of every other batched kernel (k_audio_s16, k_agf, FEC ...)
# writes gpurun_out/prof_r05/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r05; mkdir -p $OUT
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
run() { # tag, command...
  tag=$1; shift
  rm -rf /tmp/pr_$tag; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$tag -o b -- "$@" > $OUT/${tag}_under_trace.log 2>&1
  f=$(find /tmp/pr_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r05_${tag}_kernel_stats.csv
  for c in $COUNTERS; do
    rm -rf /tmp/pr_${tag}_$c; timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pr_${tag}_$c -o p -- "$@" > $OUT/${tag}_pmc_$c.log 2>&1
    f=$(find /tmp/pr_${tag}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && summ $f $c > $OUT/r05_${tag}_pmc_$c.txt
  done
}
# the bench step: HBM traffic (roofline.traffic of bench.py is read from these two files), instruction mix, and the front-end kernel's
# LDS / busy counters
COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
run bench python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras
COUNTERS="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVES"
run mixed python $R/tools/bench_mixed.py 4096 5
rm -rf /tmp/pr_stages; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stages -o b -- python $R/tools/bench_stages.py > $OUT/stages_under_trace.log 2>&1
f=$(find /tmp/pr_stages -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r05_stage_kernel_stats.csv
tail -3 $OUT/stages_under_trace.log | cut -c1-200 > /dev/null
ls -la $OUT | head -40
head -8 $OUT/r05_bench_kernel_stats.csv | cut -c1-160
