#!/bin/bash
# on the GPU box: instruction counters of a variant on configs[2]'s shape
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for b in "$@"; do
  OUT=$R/gpurun_out/prof/mt_$b; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT -o p -- $R/tools/mt_bench/$b 4096 50000 3 0 > $OUT/log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "parse_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
words = 4096 * 50000 * 12.8
print("$b", {k: "%.3g" % v for k, v in m.items()})
print("   per 64 words: VALU %.1f  SALU %.1f  LDS %.1f" % (m["SQ_INSTS_VALU"] / (words / 64), m["SQ_INSTS_SALU"] / (words / 64), m["SQ_INSTS_LDS"] / (words / 64)))
PY
done
