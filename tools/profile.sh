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
# BASELINE configs[3] on this one GPU (bench.py --workload configs3): kernel trace + the instruction counters
C3="python3 $R/bench.py --workload configs3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3kt -o p -- $C3 --steps 5 --warmup 1 > $OUT/c3kt.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/c3pmc1 -o p -- $C3 --steps 2 --warmup 1 > $OUT/c3pmc1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/c3pmc4 -o p -- $C3 --steps 2 --warmup 1 > $OUT/c3pmc4.log 2>&1 || exit 1
# the side kernels (parity mode's stream walk, one-launch kernel, extended queries) on tools/side_kernels.py's workloads
S="python3 $R/tools/side_kernels.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/skt -o p -- $S > $OUT/skt.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/spmc1 -o p -- $S > $OUT/spmc1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/spmc2 -o p -- $S > $OUT/spmc2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/spmc3 -o p -- $S > $OUT/spmc3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/spmc4 -o p -- $S > $OUT/spmc4.log 2>&1 || exit 1
# the side configurations (configs[1]-[4], parity mode, multi-GPU entry, table driver) under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_extras -o p -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/kt_extras.log 2>&1 || exit 1
python3 - <<PY
import csv, collections, glob, json, re, sys
sys.path.insert(0, "$R")
from bench import kernel_source_hash
simds, xcds, peak = 1024, 8, 256 * 4 * 2.4e9 * 32

def counters(dirs, match):
    """mean per launch of every counter over the dispatches of the kernels whose name contains `match`"""
    out = {}
    for d in dirs:
        f = glob.glob("$OUT/%s/**/p_counter_collection.csv" % d, recursive=True)[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    return out

def summary(kt_dir, dirs, match, iterations):
    raw = counters(dirs, match)
    m = {k: v["mean_per_launch"] for k, v in raw.items()}
    kt = glob.glob("$OUT/%s/**/p_kernel_stats.csv" % kt_dir, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(kt)) if match in r["Name"]]
    if not rows:
        return None, raw
    row = rows[0]
    avg_ns = float(row["AverageNs"])
    s = {"kernel": re.search(r"mcq_\w+(<[^>]*>)?", row["Name"]).group(0), "kernel_avg_ms_rocprof": avg_ns / 1e6,
         "launches_timed": int(row["Calls"]), "iterations_per_launch": iterations,
         "valu_wave_instructions_per_launch": m.get("SQ_INSTS_VALU"),
         "valu_instructions_per_wave_iteration": m["SQ_INSTS_VALU"] / (iterations / 64) if "SQ_INSTS_VALU" in m and iterations else None,
         "lds_instructions_per_wave_iteration": m["SQ_INSTS_LDS"] / (iterations / 64) if "SQ_INSTS_LDS" in m and iterations else None,
         "vmem_rd_instructions_per_wave_iteration": m["SQ_INSTS_VMEM_RD"] / (iterations / 64) if "SQ_INSTS_VMEM_RD" in m and iterations else None,
         # rocprof's derived VALUBusy: 4 cycles per active VALU instruction, per SIMD, over the GPU-active cycles of one XCD
         "valu_busy": 4 * m["SQ_ACTIVE_INST_VALU"] / simds / (m["GRBM_GUI_ACTIVE"] / xcds) if "GRBM_GUI_ACTIVE" in m and "SQ_ACTIVE_INST_VALU" in m else None,
         "lane_utilisation": m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]) if m.get("SQ_ACTIVE_INST_VALU") else None,
         "lds_bank_conflict_over_lds_active": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else None,
         # LDS pipe: index-active cycles per CU over the GPU-active cycles of one XCD (256 CUs in 8 XCDs)
         "lds_pipe_busy": m["SQ_LDS_IDX_ACTIVE"] / 256 / (m["GRBM_GUI_ACTIVE"] / xcds) if "GRBM_GUI_ACTIVE" in m and "SQ_LDS_IDX_ACTIVE" in m else None,
         # shader clock during the kernel: GPU-active cycles of one XCD over the kernel's time
         "shader_clock_ghz": m["GRBM_GUI_ACTIVE"] / xcds / avg_ns if "GRBM_GUI_ACTIVE" in m else None,
         "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
         # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact
         "hbm_bytes_per_launch": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in m and "WRITE_SIZE" in m else None}
    if s["hbm_bytes_per_launch"]:
        s["hbm_GBps"] = s["hbm_bytes_per_launch"] / (avg_ns * 1e-9) / 1e9
    if s["valu_wave_instructions_per_launch"]:
        s["issued_lane_ops_per_s"] = s["valu_wave_instructions_per_launch"] * 64 / (avg_ns * 1e-9)
        s["issued_lane_ops_frac_of_peak"] = s["issued_lane_ops_per_s"] / peak
    return s, raw

s, raw = summary("kt", ["pmc1", "pmc2", "pmc3", "pmc4"], "mcq_eval_kernel", 4096 * 100000)
json.dump(raw, open("$OUT/pmc_eval_kernel.json", "w"), indent=1)
s = dict({"kernel_sources_sha256": kernel_source_hash(), "workload": {"states": 4096, "iters": 100000, "players": 6}}, **s)
c3, _ = summary("c3kt", ["c3pmc1", "c3pmc4"], "mcq_eval_kernel", 65536 * 20000)
if c3:
    s["configs3"] = c3
units = [json.loads(l) for l in open("$OUT/skt.log") if l.startswith("{")][-1]
s["side_kernels"] = {}
for name, u in units.items():
    ss, _ = summary("skt", ["spmc1", "spmc2", "spmc3", "spmc4"], name, u["iterations"])
    if ss:
        ss["units"] = u
        s["side_kernels"][name] = ss
json.dump(s, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
PY
head -12 $(find $OUT/kt_extras -name p_kernel_stats.csv | head -1)
head -14 $(find $OUT/skt -name p_kernel_stats.csv | head -1)
