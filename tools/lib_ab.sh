#!/bin/bash
# On the GPU box: the headline bench (kernel time, evals/s) for several builds of the library, same box, interleaved twice.
#   tools/lib_ab.sh gpurun_in/lib_base.so gpurun_in/lib_bitop3.so ...   ("default" = the in-tree build)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset MCQ_LIBRARY; else export MCQ_LIBRARY=$R/$lib; fi
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s kernel ms %.4f  evals/s %.4g  spot %s' % ('$lib', d['roofline']['kernel_ms'], d['value'], d.get('parity_spot_check')))"
  done
done
