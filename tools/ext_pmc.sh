#!/bin/bash
# On the GPU box: VALU / LDS / wave-cycle counters of the extended-query kernel per launch of tools/ext_probe.py
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/ext_$$; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o p -- python3 $R/tools/ext_probe.py > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "mcq_eval_ext_kernel" in r["Kernel_Name"] or "mcq_eval_kernel<0, false>" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"][:60], r["Dispatch_Id"])].append((r["Counter_Name"], float(r["Counter_Value"])))
seen = []
for (k, d), v in by.items():
    dd = dict(v)
    line = "%s  VALU/wave-iter %.0f  LDS %.0f  VMEM_RD %.0f" % (k[28:60], dd["SQ_INSTS_VALU"] / (2048 * 20000 / 64), dd["SQ_INSTS_LDS"] / (2048 * 20000 / 64), dd.get("SQ_INSTS_VMEM_RD", 0) / (2048 * 20000 / 64))
    if line not in seen:
        seen.append(line); print(line)
PY
