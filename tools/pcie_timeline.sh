#!/bin/bash
# timeline of the PCIe-inclusive step: kernels + copies of three steady-state steps, for each host-output mode
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pcie_timeline; mkdir -p $OUT
for mode in ${@:-resident h2d compact}; do
  rm -rf /tmp/tl_$mode
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$mode -o t -- python $R/tools/pcie_timeline.py run $mode > $OUT/$mode.log 2>&1
  grep "ms per step" $OUT/$mode.log
  python $R/tools/pcie_timeline.py summarise /tmp/tl_$mode > $OUT/$mode.timeline.txt 2>&1
done
