#!/bin/bash
# on the GPU box: instruction counts per launch for each ablation variant
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof; cd /tmp; export TMPDIR=/tmp
for name in full EVAL HOLES RNG ALL; do
  MCQ_LIBRARY=$R/gpurun_in/ablate/libmcq_$name.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/prof/ab_$name -o ab -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $1 > $R/gpurun_out/prof/ab_$name.log 2>&1
  python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$R/gpurun_out/prof/ab_$name/ab_counter_collection.csv")))
agg = collections.defaultdict(list)
for r in rows:
    if "eval_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$name", {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
PY
done
