#!/bin/bash
# On the GPU box: bulk-kernel time of bench.py's workload for each library given ($@ = paths relative to the repo root; "-" = in-tree)
R=$GRAFT_REPO_ROOT
for L in "$@"; do
  if [ "$L" = "-" ]; then unset MCQ_LIBRARY; else export MCQ_LIBRARY=$R/$L; fi
  python3 $R/bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L: kernel ms %.4f  evals/s %.4g' % (d['roofline']['kernel_ms'], d['value']))"
done
