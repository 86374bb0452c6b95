#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace stats + PMC passes of the bench workload -> gpurun_out/prof/<TAG>/
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o p -- $B --steps 10 --warmup 2 > $OUT/kt.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -o p -- $B --steps 2 --warmup 1 > $OUT/pmc1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -o p -- $B --steps 2 --warmup 1 > $OUT/pmc2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -o p -- $B --steps 2 --warmup 1 > $OUT/pmc3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc4 -o p -- $B --steps 2 --warmup 1 > $OUT/pmc4.log 2>&1 || exit 1
python3 - <<PY
import csv, collections, json
out = {}
for d in ["pmc1", "pmc2", "pmc3", "pmc4"]:
    rows = list(csv.DictReader(open("$OUT/%s/p_counter_collection.csv" % d)))
    agg = collections.defaultdict(list)
    for r in rows:
        if "eval_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(out, open("$OUT/pmc_eval_kernel.json", "w"), indent=1)
print(json.dumps({k: v["mean_per_launch"] for k, v in out.items()}, indent=1))
PY
head -4 $OUT/kt/p_kernel_stats.csv
