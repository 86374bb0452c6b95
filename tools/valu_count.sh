#!/bin/bash
# On the GPU box: VALU instructions per wave-iteration and kernel time of the bulk kernel for the library given by
# $MCQ_LIBRARY (default: the in-tree build).  One PMC pass + one timed run.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/valu_$$; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $OUT -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "mcq_eval_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU"]
print("VALU instructions per wave-iteration: %.1f" % (sum(v) / len(v) / (4096 * 100000 / 64)))
PY
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel ms %.4f  evals/s %.4g' % (d['roofline']['kernel_ms'], d['value']))"
