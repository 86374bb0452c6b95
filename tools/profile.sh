#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace stats + PMC passes of the bench workload -> gpurun_out/prof/<TAG>/, and
# gpurun_out/prof/<TAG>/summary.json: per-launch means of the evaluation kernel, the derived utilisation figures and
# the hash of the kernel sources they were measured on (bench.py only quotes counters whose hash matches its own).
# The program stands directly after `--` (no env / bash -c hops: rocprofv3's preloaded library initialises the GPU).
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o p -- $B --steps 10 --warmup 2 > $OUT/kt.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -o p -- $B --steps 2 --warmup 1 > $OUT/pmc1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -o p -- $B --steps 2 --warmup 1 > $OUT/pmc2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -o p -- $B --steps 2 --warmup 1 > $OUT/pmc3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc4 -o p -- $B --steps 2 --warmup 1 > $OUT/pmc4.log 2>&1 || exit 1
# the side configurations (configs[1]-[4], parity mode, multi-GPU entry, table driver) under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_extras -o p -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/kt_extras.log 2>&1 || exit 1
python3 - <<PY
import csv, collections, glob, json, sys
sys.path.insert(0, "$R")
from bench import kernel_source_hash
out = {}
for d in ["pmc1", "pmc2", "pmc3", "pmc4"]:
    f = glob.glob("$OUT/%s/**/p_counter_collection.csv" % d, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mcq_eval_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(out, open("$OUT/pmc_eval_kernel.json", "w"), indent=1)
m = {k: v["mean_per_launch"] for k, v in out.items()}
kt = glob.glob("$OUT/kt/**/p_kernel_stats.csv", recursive=True)[0]
row = [r for r in csv.DictReader(open(kt)) if "mcq_eval_kernel" in r["Name"]][0]
avg_ns = float(row["AverageNs"])
simds, xcds = 1024, 8
s = {"kernel_sources_sha256": kernel_source_hash(), "workload": {"states": 4096, "iters": 100000, "players": 6},
     "kernel": __import__("re").search(r"mcq_\w+<[^>]*>", row["Name"]).group(0), "kernel_avg_ms_rocprof": avg_ns / 1e6, "launches_timed": int(row["Calls"]),
     "valu_wave_instructions_per_launch": m.get("SQ_INSTS_VALU"),
     "valu_instructions_per_wave_iteration": m["SQ_INSTS_VALU"] / (4096 * 100000 / 64) if "SQ_INSTS_VALU" in m else None,
     # rocprof's derived VALUBusy: 4 cycles per active VALU instruction, per SIMD, over the GPU-active cycles of one XCD
     "valu_busy": 4 * m["SQ_ACTIVE_INST_VALU"] / simds / (m["GRBM_GUI_ACTIVE"] / xcds) if "GRBM_GUI_ACTIVE" in m else None,
     "lane_utilisation": m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]) if "SQ_ACTIVE_INST_VALU" in m else None,
     "lds_bank_conflict_over_lds_active": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if "SQ_LDS_IDX_ACTIVE" in m else None,
     "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
     # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact
     "hbm_bytes_per_launch": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in m and "WRITE_SIZE" in m else None}
if s["hbm_bytes_per_launch"]:
    s["hbm_GBps"] = s["hbm_bytes_per_launch"] / (avg_ns * 1e-9) / 1e9
if s["valu_wave_instructions_per_launch"]:
    lane_ops = s["valu_wave_instructions_per_launch"] * 64 * (s["lane_utilisation"] or 1.0)
    s["issued_lane_ops_per_s"] = lane_ops / (avg_ns * 1e-9)
    s["issued_lane_ops_frac_of_peak"] = s["issued_lane_ops_per_s"] / (256 * 4 * 2.4e9 * 32)
json.dump(s, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
PY
head -12 $(find $OUT/kt_extras -name p_kernel_stats.csv | head -1)
