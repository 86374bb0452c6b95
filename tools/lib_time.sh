#!/bin/bash
# On the GPU box: bulk-kernel time of one or more library builds (paths relative to the repo): tools/lib_time.sh lib1.so lib2.so ...
for L in "$@"; do
  MCQ_LIBRARY=$PWD/$L python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'kernel ms %.4f  evals/s %.4g' % (d['roofline']['kernel_ms'], d['value']))"
done
