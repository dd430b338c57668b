#!/bin/bash
# HBM traffic of the secondary kernels: one rocprofv3 pass per counter (PMC + --kernel-trace only), torch-free driver.
# usage: tools/pmc_stages.sh   -> gpurun_out/pmc_stages.txt  (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, bytes per launch)
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pst_$c
  DDN_NO_TORCH=1 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pst_$c -o p -- \
      python $R/tools/pmc_stages.py > /tmp/pst_$c.log 2>&1
done
python3 - <<'PY' | tee $R/gpurun_out/pmc_stages.txt
import csv, glob, collections
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"/tmp/pst_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(c, "pass failed"); continue
    t = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        t[name.split("(")[0][:40]].append(float(r["Counter_Value"]))
    v[c] = {k: sum(x) / len(x) for k, x in t.items()}
ks = sorted(set(v.get("FETCH_SIZE", {})) | set(v.get("WRITE_SIZE", {})))
print(f"{'kernel':42s} {'FETCH_SIZE KB':>14s} {'WRITE_SIZE KB':>14s} {'HBM MB (2*F+W)':>15s}")
for k in ks:
    f, w = v.get("FETCH_SIZE", {}).get(k, 0.0), v.get("WRITE_SIZE", {}).get(k, 0.0)
    print(f"{k:42s} {f:14.0f} {w:14.0f} {(2 * f + w) / 1024:15.1f}")
PY
