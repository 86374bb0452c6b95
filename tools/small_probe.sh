#!/bin/bash
# On the GPU box: small-batch probe under the tuning knobs + a kernel trace of one configuration -> gpurun_out/
mkdir -p gpurun_out
TAG=${1:-x}
{
  for lw in 16 8 4; do MCQ_LOAD_WAVES=$lw timeout -k 10 200 python tools/small_probe.py 2>/dev/null || exit 1; echo; done
} > gpurun_out/small_probe_$TAG.txt
cat gpurun_out/small_probe_$TAG.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/small_$TAG -o p -- python3 $GRAFT_REPO_ROOT/tools/small_probe.py > /dev/null 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof/small_$TAG/*/p_kernel_stats.csv 2>/dev/null || head -8 $GRAFT_REPO_ROOT/gpurun_out/prof/small_$TAG/p_kernel_stats.csv
