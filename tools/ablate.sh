#!/bin/bash
# Diagnostic: build timing-only variants of libmcq_hip.so with parts of the iteration stubbed out.
# usage: tools/ablate.sh build   (here)      tools/ablate.sh run   (on the GPU box)
cd "$(dirname "$0")/.."
VARIANTS="full:-DMCQ_NOP QUADS:-DMCQ_ABLATE_QUADS EVAL:-DMCQ_ABLATE_EVAL HOLES:-DMCQ_ABLATE_HOLES RNG:-DMCQ_ABLATE_RNG EVAL_HOLES:-DMCQ_ABLATE_EVAL;-DMCQ_ABLATE_HOLES ALL:-DMCQ_ABLATE_EVAL;-DMCQ_ABLATE_HOLES;-DMCQ_ABLATE_RNG"
if [ "$1" = build ]; then
  mkdir -p gpurun_in/ablate
  for v in $VARIANTS; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr ';' ' ')
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden $flags \
       -o gpurun_in/ablate/libmcq_$name.so neuron_poker_amd/csrc/mcq_host.cpp neuron_poker_amd/csrc/mcq_tables.cpp neuron_poker_amd/csrc/mcq_kernels.hip 2>/dev/null || echo "build $name failed"
  done
  ls -la gpurun_in/ablate
else
  for v in $VARIANTS; do
    name=${v%%:*}
    MCQ_LIBRARY=$PWD/gpurun_in/ablate/libmcq_$name.so python - <<PY
import json, subprocess, sys, os
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "10", "--warmup", "2"] + "$2".split(), capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    print("%-12s kernel_ms %.3f" % ("$name", d["roofline"]["kernel_ms"]))
except Exception as e:
    print("$name failed", out.stderr[-500:])
PY
  done
fi
