#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X equity engine.

Metric (BASELINE.json): Monte-Carlo 7-card hand evaluations per second @100k iterations, 6-max preflop.
A "step" = one pass of the hot path (deal -> evaluate -> tally) over one batch of synthetic equity queries:
4 096 random preflop states PER GPU (the batch size and state generator of BASELINE configs[2]), 6 players,
no table cards, 100 000 iterations each = 2.4576e9 hand evaluations per GPU per step.  Queries and tallies
are resident in HBM (torch tensors, device-pointer C-ABI entry mcq_eval_batch_device); with N > 1 ranks each
rank evaluates its own shard and one RCCL all-reduce of the integer tally matrix closes the step (weak
scaling: per-GPU work is fixed).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      VALU integer issue roofline of the evaluation kernel (this path is integer/branch work, neither
                HBM- nor MFMA-bound: 120 B of HBM traffic per QUERY, none per iteration).  `achieved` = int32
                lane-ops the kernel ISSUES per launch (SQ_INSTS_VALU x 64 lanes, rocprofv3 PMC pass committed
                under profiles/ for exactly these kernel sources) / mean kernel time from HIP events recorded
                on the launch stream inside the timed region; `frac` = achieved / the 2-cycle wave64 issue
                peak, <= 1 by construction.  SURVEY 8(d)'s a-priori cost model is reported beside it as
                `model` (it over-counts this kernel's instructions 2.3x, so it is not a fraction of anything)
  cpu_baseline  the oracle (a C port of tools/montecarlo_python.py, bit-exact to it) timed on this host
  parity_spot_check  rows of the LAST timed launch compared with the oracle (outside the timed region); a
                mismatch makes the process exit non-zero
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

JSON_OUT = sys.stdout  # where the ONE JSON line goes (see main)
PEAK_VALU_TOPS = 256 * 4 * 2.4e9 * 32 / 1e12  # 256 CU x 4 SIMD-32 x 2.4 GHz full-rate int32 lane-ops
HBM_PEAK_GBPS = 8000.0


KERNEL_SOURCES = ["neuron_poker_amd/csrc/mcq_device.hpp", "neuron_poker_amd/csrc/mcq_kernels.hip", "neuron_poker_amd/csrc/mcq_mt.hpp",
                  "neuron_poker_amd/csrc/mcq_exact.hpp", "neuron_poker_amd/csrc/mcq_mt_ext.hpp", "neuron_poker_amd/csrc/mcq_mt_blocks.hpp",
                  # round 4: the kernels' text passes through tools/isa_resched.py on its way into the library
                  "neuron_poker_amd/csrc/mcq_mt_jump_table.inc", "neuron_poker_amd/csrc/Makefile", "tools/isa_resched.py",
                  "tools/isa_cadence.py"]


def kernel_source_hash():
    """sha256 over the kernel sources: ties a rocprof summary under profiles/ to the code it was measured on (the GPU
    box has no .git, so a commit id is not available there)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def profile_counters():
    """Counter-based figures of the kernels from the committed rocprofv3 PMC passes (profiles/current.json, written by
    tools/profile.sh).  -> (summary or None, fresh): fresh = they were measured on exactly these kernel sources."""
    try:
        with open(os.path.join(ROOT, "profiles", "current.json")) as f:
            s = json.load(f)
    except (OSError, ValueError):
        return None, False
    return s, s.get("kernel_sources_sha256") == kernel_source_hash()


def issue_roofline(kernel, kernel_ms, counters, fresh, model_ops_per_launch=None, model_basis=None):
    """VALU issue roofline of one kernel: achieved = issued int32 lane-ops per launch (SQ_INSTS_VALU wave instructions
    x 64 lanes, from the committed PMC pass of the same workload) / kernel time; peak = 2-cycle wave64 issue on every
    SIMD.  An instruction cannot issue faster than that, so frac <= 1 by construction."""
    r = {"bound": "valu", "kernel": kernel, "kernel_ms": kernel_ms, "peak": PEAK_VALU_TOPS, "unit": "Tops/s (int32 lane-ops)",
         "achieved": None, "frac": None,
         "frac_basis": "issued: SQ_INSTS_VALU x 64 lanes per launch (rocprofv3 --pmc, profiles/current.json) / kernel time "
                       "/ (256 CU x 4 SIMD x 2.4 GHz x 32 lanes)"}
    insts = counters.get("valu_wave_instructions_per_launch") if counters else None
    if insts and kernel_ms and kernel_ms == kernel_ms:
        r["achieved"] = insts * 64.0 / (kernel_ms * 1e-3) / 1e12
        r["frac"] = r["achieved"] / PEAK_VALU_TOPS
        r["counters_fresh"] = bool(fresh)  # False: the PMC pass was taken on EARLIER kernel sources (stale count)
        r["valu_busy"] = counters.get("valu_busy")
        r["counters"] = {k: counters.get(k) for k in (
            "kernel_avg_ms_rocprof", "valu_wave_instructions_per_launch", "valu_instructions_per_wave_iteration",
            "lane_utilisation", "lds_bank_conflict_over_lds_active", "lds_pipe_busy", "hbm_bytes_per_launch", "hbm_GBps",
            "shader_clock_ghz")
            if counters.get(k) is not None}
        r["counters"]["source"] = "rocprofv3 --pmc passes, profiles/current.json"
    if model_ops_per_launch:
        m = model_ops_per_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms else None
        r["model"] = {"basis": model_basis, "lane_ops_per_launch": model_ops_per_launch, "Tops": m,
                      "over_peak": m / PEAK_VALU_TOPS if m else None,
                      "note": "a-priori cost model, NOT a fraction of a hardware bound: the kernel needs fewer instructions "
                              "than the model counts, so this ratio can exceed 1"}
    return r


def practical_roofline(kernel_ms, wave_iterations_per_launch, counters):
    """The mix-weighted issue bound of the bulk kernel's straight-line iteration (tools/isa_hist.py ->
    profiles/current_isa_hist.json: every VALU instruction of the 6-max preflop loop priced at the cadence of its issue
    class as measured on the hardware, profiles/r07_issue_probe*.txt / r07_replay.txt) against the cycles a SIMD spends
    per wave-iteration in this run (kernel time x shader clock x 1024 SIMDs / wave-iterations, four waves per SIMD)."""
    try:
        with open(os.path.join(ROOT, "profiles", "current_isa_hist.json")) as f:
            h = json.load(f)
    except (OSError, ValueError):
        return None
    clock = (counters or {}).get("shader_clock_ghz") or 2.4
    measured = kernel_ms * 1e-3 * clock * 1e9 * 1024 / wave_iterations_per_launch if kernel_ms and wave_iterations_per_launch else None
    bound = h.get("cycles_per_wave_iteration_bound")
    return {"cycles_per_wave_iteration_bound": bound,
            "cycles_per_wave_iteration_as_scheduled": h.get("cycles_as_scheduled"),
            "cycles_per_wave_iteration_at_2_per_instruction": h.get("cycles_at_2_per_instruction"),
            "cycles_per_wave_iteration_measured": measured, "shader_clock_ghz": clock,
            "frac_of_practical": bound / measured if bound and measured else None,
            "valu_instructions": h.get("valu"), "by_class": {k: h.get("classes", {}).get(k) for k in ("fast", "slow", "mul64", "sel_vcc")},
            "fast_class_instructions_at_the_slow_cadence": h.get("poisoned_fast_instructions"),
            "separators_behind_slow_class": h.get("separators_behind_slow"),
            "slow_class_directly_behind_slow_class": h.get("slow_directly_behind_slow"),
            "class_costs_simd_cycles": h.get("costs_simd_cycles"),
            "histogram_fresh": h.get("kernel_sources_sha256") == kernel_source_hash(),
            "basis": "bound = sum over the loop's VALU instructions of the cost of their issue class (fast 2.6, slow 4.7 "
                     "SIMD-cycles per wave64 instruction: this loop's own instructions replayed class by class, "
                     "tools/ubench/replay_loop.py); as_scheduled = the same with the cadence rule applied to the compiler's "
                     "ORDER of the shipped text, i.e. behind tools/isa_resched.py (a fast-class instruction behind a slow-class one of its "
                     "VALU run issues at the slow cadence; a separator costs 0.8, a slow-class instruction directly behind another "
                     "one 1.0 more: profiles/r07_issue_probe3.txt); "
                     "measured = kernel time x clock x 1024 SIMDs / wave-iterations"}


def alg_ops_per_iteration(n_players, n_board):
    """SURVEY.md 8(d) canonical cost model: A(N, b) = 48 D + 30 + 82 N int32 lane-ops, D = 2(N-1) + (5-b)."""
    d = 2 * (n_players - 1) + (5 - n_board)
    return 48 * d + 30 + 82 * n_players


def make_states(n_states, rank):
    g = np.random.default_rng(4096 + rank)  # rank 0 == the generator of BASELINE configs[2]
    hole = np.array([g.choice(52, 2, replace=False) for _ in range(n_states)], np.uint8)
    board = np.full((n_states, 5), 255, np.uint8)
    return hole, board


def make_configs3_states():
    """BASELINE configs[3] / SURVEY 8d config 4: 65 536 states, default_rng(65536), flop (even i) / turn (odd i) tables,
    -> (hole [B,2], board [B,5] with 0xFF = empty)."""
    g3 = np.random.default_rng(65536)
    keys = g3.random((65536, 52)).argsort(axis=1)[:, :6].astype(np.uint8)   # 6 distinct cards per state
    b3 = np.full((65536, 5), 255, np.uint8)
    b3[:, :3] = keys[:, 2:5]
    b3[1::2, 3] = keys[1::2, 5]
    return keys[:, :2].copy(), b3


CONFIGS3 = {"states": 65536, "players": 6, "iters": 20000}
METRIC = "Monte Carlo hand evals/sec @100k iters, 6-max preflop"
METRIC_CONFIGS3 = ("Monte Carlo hand evals/sec, BASELINE configs[3]: 65 536 mixed flop/turn states x 6 players x 20k iters, "
                   "sharded over the GPUs, one tally all-reduce")


def cpu_baseline(n_players, runs, seconds=12.0, threads=None):
    """Time the oracle (kind 'port') on this host's cores on a bounded sample of the same workload."""
    from oracle import oracle as O
    cores = max(1, len(os.sched_getaffinity(0)))
    threads = cores if threads is None else threads   # ALL the host's cores; `cores` in the result says how many
    hole, board = make_states(threads, 0)
    probe = O.pack_queries(hole, board, n_players, 20000)
    t0 = time.perf_counter()
    O.run_batch(O.MODE_MT, probe, 1, 0, threads=threads)
    rate = threads * 20000 * n_players / max(time.perf_counter() - t0, 1e-3)  # evals/s estimate
    iters = runs
    n_states = int(min(4096, max(threads, threads * round(rate * seconds / (runs * n_players * threads)))))
    hole, board = make_states(n_states, 0)
    q = O.pack_queries(hole, board, n_players, iters)
    t0 = time.perf_counter()
    O.run_batch(O.MODE_MT, q, 1, 0, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": len(hole) * iters * n_players / dt, "unit": "hand-evals/s", "cores": threads,
            "host_cores_available": cores, "host_cores_total": os.cpu_count(), "kind": "port",
            "sample": "%d of the workload's preflop states x %d iterations x %d players, oracle MT19937 mode "
                      "(bit-exact to tools/montecarlo_python.py), %d threads, %.1f s" %
                      (len(hole), iters, n_players, threads, dt)}


def cpu_reference_cpp(n_players, seconds=4.0):
    """The reference's own C++ variant (tools/montecarlo_cpp, built into oracle/_ref by `make -C oracle ref` where
    /root/reference exists; the binary travels with the repo snapshot), one thread, timed on this host."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_mc")
    if not os.path.exists(exe):
        return None
    iters = 4000
    try:
        out = subprocess.run([exe, "equity", "AS", "KS", str(n_players), str(iters)], capture_output=True, text=True,
                             timeout=120, check=True).stdout.split()
        dt = float(out[1])
        iters = int(max(iters, min(400000, iters * seconds / max(dt, 1e-3))))
        out = subprocess.run([exe, "equity", "AS", "KS", str(n_players), str(iters)], capture_output=True, text=True,
                             timeout=300, check=True).stdout.split()
        dt = float(out[1])
    except (OSError, subprocess.SubprocessError, ValueError, IndexError):
        return None
    return {"value": iters * n_players / dt, "unit": "hand-evals/s", "cores": 1, "kind": "reference",
            "sample": "tools/montecarlo_cpp/Montecarlo.cpp montecarlo(AS KS, no table cards, %d players, %d "
                      "iterations), %.1f s; uniform dealing, so a timing reference only" % (n_players, iters, dt)}


def launch_ranks(n, argv):
    """Run this script as n ranks (one process per GPU) under torch.distributed.run in a child process; returns its
    exit code.  stdout / stderr are inherited, so rank 0's JSON line is this command's JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def native_multi(args):
    """--native-multi: the whole job in ONE process through the C ABI's multi-GPU entry (include/mcq.h): shards -> one
    ncclAllReduce of the tally matrix.  Queries and tallies resident in HBM (mcq_multi_eval_batch_device; torch only holds
    the buffers), so the figure carries no PCIe -- the same keys as the one-rank-per-GPU line."""
    import torch
    import neuron_poker_amd as npa
    n_dev = npa.load_library().mcq_device_count()
    devices = [0] * args.gpus if args.single_device else list(range(args.gpus))
    if not args.single_device and args.gpus > n_dev:
        raise SystemExit("--gpus %d but %d devices visible" % (args.gpus, n_dev))
    me = npa.MultiEngine(devices)
    k = len(devices)
    configs3 = args.workload == "configs3"
    N, runs = args.players, args.iters
    if configs3:
        hole, board = make_configs3_states()
        B = len(hole)
    else:
        B = args.states * k
        hole = np.concatenate([make_states(args.states, r)[0] for r in range(k)])
        board = np.full((B, 5), 255, np.uint8)
    raw = npa.pack_queries(hole, board, N, runs).view(np.uint8).reshape(B, 16)
    d_q = [torch.from_numpy(raw[B * s // k:B * (s + 1) // k].copy()).to("cuda:%d" % d) for s, d in enumerate(devices)]
    d_res = [torch.zeros((B, 13), dtype=torch.int64, device="cuda:%d" % d) for d in devices]
    qp, rp = [t.data_ptr() for t in d_q], [t.data_ptr() for t in d_res]
    seed = 20261004

    def fence():
        for d in set(devices):
            torch.cuda.synchronize(d)

    for i in range(args.warmup):
        me.eval_batch_device(qp, B, seed + i, rp, partition="queries")
    fence()
    kmax, ar = [], []
    t0 = time.perf_counter()
    for i in range(args.steps):
        me.eval_batch_device(qp, B, seed + args.warmup + i, rp, partition="queries")   # returns when every device is done
        t = me.last_times_ms
        kmax.append(t["kernel_max"])
        ar.append(t["all_reduce"])
    fence()
    elapsed = time.perf_counter() - t0
    t = d_res[0].cpu().numpy().view(np.uint64)
    assert (t[:, 0] == runs).all() and np.array_equal(t[:, 2] + t[:, 3], t[:, 4:].sum(1))
    assert all(torch.equal(d_res[0].cpu(), r.cpu()) for r in d_res[1:]), "the shards' matrices differ after the all-reduce"
    evals = float(B) * runs * N
    info = me.info
    kernel_ms = float(np.mean(kmax))
    pc, fresh = profile_counters()
    if configs3 or (pc and pc.get("workload") != {"states": args.states, "iters": runs, "players": N}):
        pc = None
    per_it = 0.5 * (alg_ops_per_iteration(N, 3) + alg_ops_per_iteration(N, 4)) if configs3 else alg_ops_per_iteration(N, 0)
    roof = issue_roofline("mcq_eval_kernel<0, false>", kernel_ms, pc, fresh,
                          model_ops_per_launch=float(B) / k * runs * per_it,
                          model_basis="SURVEY 8(d): %d lane-ops per iteration" % per_it)
    roof["traffic"] = pc.get("hbm_bytes_per_launch") if pc else None
    roof["note"] = "per shard: the slowest shard's evaluation kernel"
    import hashlib
    what = ("BASELINE configs[3]: 65 536 flop/turn states (default_rng(65536)) IN ALL" if configs3
            else "%d random preflop states per GPU" % args.states)
    print(json.dumps({
        "metric": METRIC_CONFIGS3 if configs3 else METRIC, "value": evals * args.steps / elapsed,
        "unit": "hand-evals/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if configs3 else "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s x %d players x %d iterations, ONE process, %d shards on "
                               "devices %s via mcq_multi_eval_batch_device (queries and tallies resident in HBM), partition "
                               "'%s', one ncclAllReduce of the [%d,13] uint64 tally matrix over %d device(s)" %
                               (what, N, runs, info["shards"], devices, info["last_partition"], B, info["devices"]),
                   "states_per_gpu": B // k, "states_total": B, "n_players": N, "iterations": runs,
                   "n_board": "3 / 4 alternating" if configs3 else 0, "hand_evals_per_step": evals},
        "tallies_sha256": hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest(),
        "roofline": roof,
        "collective": {"all_reduce_ms": float(np.mean(ar)), "bytes": B * 104, "ranks": info["devices"],
                       "rccl_version": info["rccl_version"]},
        "native_multi": {"kernel_max_ms": kernel_ms, "all_reduce_ms": float(np.mean(ar)),
                         "rccl_version": info["rccl_version"]}}), file=JSON_OUT, flush=True)
    me.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--states", type=int, default=4096, help="states per GPU")
    ap.add_argument("--iters", type=int, default=100000)
    ap.add_argument("--players", type=int, default=6)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real runs; gloo rehearses the N > 1 control "
                    "flow (tallies all-reduced through host memory)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--native-multi", action="store_true", help="ONE process drives all --gpus devices through the C ABI's "
                    "mcq_multi_* entry (shards + one ncclAllReduce from ncclCommInitAll) instead of one rank per GPU; "
                    "host buffers, so the figure includes PCIe")
    ap.add_argument("--workload", default="headline", choices=["headline", "configs3"],
                    help="headline: --states preflop states PER GPU x 6 players x 100k iterations (weak scaling). configs3: "
                         "BASELINE configs[3], 65 536 flop/turn states x 6 players x 20k iterations IN ALL, block-sharded over "
                         "the --gpus ranks (strong scaling), one all-reduce of the [65536,13] tally matrix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()
    if args.workload == "configs3":
        args.players, args.iters = CONFIGS3["players"], CONFIGS3["iters"]

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.native_multi or under_launcher or args.gpus == 1:
        # RCCL prints a version banner on STDOUT when a communicator is created; this process's stdout carries the
        # JSON line and nothing else, so everything else written to fd 1 from here on goes to stderr
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        global JSON_OUT
        JSON_OUT = os.fdopen(json_fd, "w")
    if args.native_multi:
        return native_multi(args)
    if args.gpus > 1 and not under_launcher:
        # `python bench.py --gpus N` as such: start the one-process-per-GPU job as a CHILD (torch.distributed.run ->
        # N ranks) before this process has touched the GPU or imported torch, and hand its output and exit code on.
        # (Never an exec: a process that has initialised the GPU must not be replaced.)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    import neuron_poker_amd as npa

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under the launcher the process group is formed even for one rank, so that a 1-GPU box runs the very RCCL
    # calls (init, all-reduce, barrier) the N-GPU job makes
    grouped = under_launcher
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    eng = npa.Engine(local_rank, kernel_times=True)
    N, runs = args.players, args.iters
    configs3 = args.workload == "configs3"
    if configs3:   # a fixed batch, block-distributed (queries [n*r/k, n*(r+1)/k) to rank r, ids = global index)
        total = CONFIGS3["states"]
        hole_all, board_all = make_configs3_states()
        first_qid, hi_q = total * rank // world, total * (rank + 1) // world
        hole, board = hole_all[first_qid:hi_q], board_all[first_qid:hi_q]
    else:          # --states per rank, every rank its own generator (weak scaling)
        total = world * args.states
        first_qid = rank * args.states
        hole, board = make_states(args.states, rank)
    B = len(hole)   # this rank's queries
    q = npa.pack_queries(hole, board, N, runs)
    d_q = torch.from_numpy(q.view(np.uint8).reshape(B, 16).copy()).to(dev)
    tallies = torch.zeros((total, 13), dtype=torch.int64, device=dev)  # mcq_result rows of the whole job
    mine = tallies[first_qid:first_qid + B]
    stream = torch.cuda.current_stream()
    seed = 20261004

    def step(i):
        if grouped:
            tallies.zero_()
        eng.eval_batch_device(d_q.data_ptr(), B, seed + i, mine.data_ptr(), first_query_id=first_qid,
                              stream=stream.cuda_stream)
        if grouped:
            if args.backend == "nccl":
                dist.all_reduce(tallies, op=dist.ReduceOp.SUM)  # the path's one collective: integer tallies over xGMI
            else:  # rehearsal through host memory
                host = tallies.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                tallies.copy_(host)

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if grouped:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # sanity on the last step's tallies (cheap; outside the timed region)
    t = tallies.cpu().numpy().view(np.uint64)
    assert (t[:, 0] == runs).all(), "runs column wrong"
    assert np.array_equal(t[:, 2] + t[:, 3], t[:, 4:].sum(1)), "sum(by_type) != win + tie"

    ar_ms = None
    if grouped and args.backend == "nccl":   # the collective alone (outside the timed region): same key as --native-multi
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dist.all_reduce(tallies, op=dist.ReduceOp.SUM)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 5
    kt = eng.kernel_times(min(args.steps, 64))  # HIP events on the launch stream, recorded inside the timed region
    kernel_ms = float(np.mean(kt)) if len(kt) else float("nan")
    evals_per_step = float(total) * runs * N
    value = evals_per_step * args.steps / elapsed

    if rank != 0:
        dist.destroy_process_group()
        return

    # counter-based figures: from the committed PMC passes of exactly this workload (kernel time is measured live)
    pc, fresh = profile_counters()
    if configs3:   # counters of this workload: profiles/current.json -> "configs3" (tools/profile.sh), for the whole batch on ONE GPU
        pc = (pc or {}).get("configs3")
        if pc and world != 1:
            pc = None
        model_ops = float(B) * runs * 0.5 * (alg_ops_per_iteration(N, 3) + alg_ops_per_iteration(N, 4))
        model_basis = "SURVEY 8(d): A(6,3) = %d / A(6,4) = %d lane-ops per iteration" % (alg_ops_per_iteration(N, 3),
                                                                                         alg_ops_per_iteration(N, 4))
    else:
        if pc and pc.get("workload") != {"states": B, "iters": runs, "players": N}:
            pc = None
        model_ops = float(B) * runs * alg_ops_per_iteration(N, 0)
        model_basis = "SURVEY 8(d): A(N,b) = 48 D + 30 + 82 N = %d lane-ops per iteration" % alg_ops_per_iteration(N, 0)
    roof = issue_roofline("mcq_eval_kernel<0, false>", kernel_ms, pc, fresh, model_ops_per_launch=model_ops,
                          model_basis=model_basis)
    roof["traffic"] = pc.get("hbm_bytes_per_launch") if pc else None
    if not configs3 and N == 6:   # the histogram is of the 6-max preflop iteration (5 opponents, 5 table cards to come)
        roof["practical"] = practical_roofline(kernel_ms, float(B) * runs / 64.0, pc)
    roof["hbm"] = {"algorithmic_bytes_per_launch": B * 120, "achieved_GBps": B * 120 / (kernel_ms * 1e-3) / 1e9,
                   "peak_GBps": HBM_PEAK_GBPS}
    ar_text = (", one %s all-reduce of the [%d,13] int64 tally matrix per step" %
               ("RCCL" if args.backend == "nccl" else args.backend, total)) if grouped else ""
    if configs3:
        workload = ("BASELINE configs[3]: 65 536 states (default_rng(65536)), flop / turn tables alternating, x %d players x %d "
                    "iterations IN ALL, block-sharded over %d rank(s) (queries [n*r/k, n*(r+1)/k) on rank r under their global "
                    "ids), production RNG (MCQ-CTR v5), queries and tallies resident in HBM%s" % (N, runs, world, ar_text))
    else:
        workload = ("%d random preflop states per GPU (default_rng(4096+rank), as BASELINE configs[2]) x %d players x %d "
                    "iterations, production RNG (Philox-keyed MWC64X, MCQ-CTR v5), queries and tallies resident in HBM%s" %
                    (B, N, runs, ar_text))
    out = {
        "metric": METRIC_CONFIGS3 if configs3 else METRIC, "value": value, "unit": "hand-evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong" if configs3 else "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": workload, "states_per_gpu": B, "states_total": total, "n_players": N, "iterations": runs,
                   "n_board": "3 / 4 alternating" if configs3 else 0, "hand_evals_per_step": evals_per_step},
        "roofline": roof,
    }
    if configs3:   # a checksum of the whole tally matrix: equal for every --gpus (the tallies do not depend on the partition)
        import hashlib
        out["tallies_sha256"] = hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()
    if grouped:
        out["collective"] = {"all_reduce_ms": ar_ms, "bytes": total * 104, "ranks": world,
                             "backend": "RCCL (torch.distributed nccl)" if args.backend == "nccl" else args.backend}
    side_counters = (profile_counters()[0] or {}).get("side_kernels", {})  # PMC passes of tools/side_kernels.py's workloads
    if world == 1 and not args.no_extras and not configs3:  # single-GPU side measurements; never delay the other ranks' teardown
        extras = {}

        def side_measurements():  # a failure here must never cost the headline line
            # BASELINE configs[1]: single query AhKh heads-up 100k iterations (latency-bound: 98 wave tasks)
            def call_time(q, reps):
                # call time from plain launches (the library's default) -- a timestamped launch adds ~6 us to a call
                # (tools/launch_floor.hip), which matters for the small configurations -- then one timestamped call
                # for the kernel time
                eng.set_kernel_timing(False)
                try:
                    for _ in range(3):
                        eng.eval_batch(q, seed=1)
                    t = time.perf_counter()
                    for i in range(reps):
                        eng.eval_batch(q, seed=i)
                    dt = (time.perf_counter() - t) / reps
                finally:
                    eng.set_kernel_timing(True)
                eng.eval_batch(q, seed=1)
                return dt

            # the headline workload handed over in HOST buffers (H2D of the records, D2H of the rows: PCIe-inclusive)
            dt = call_time(q, 5)
            extras["headline_from_host_buffers"] = {"call_ms_host_buffers": 1e3 * dt, "kernel_ms": eng.last_kernel_ms,
                                                    "hand_evals_per_s": float(B) * runs * N / dt}
            q1 = npa.pack_queries([[npa.card_id("AH"), npa.card_id("KH")]], [[255] * 5], 2, 100000)
            dt = call_time(q1, 20)
            extras["configs[1]_single_query_100k"] = {"call_ms_host_buffers": 1e3 * dt, "kernel_ms": eng.last_kernel_ms,
                                                     "hand_evals_per_s": 2e5 / dt}
            # BASELINE configs[2]: 4096 preflop states x 3 players x 50k iterations, host buffers (PCIe-inclusive)
            q3 = npa.pack_queries(hole[:4096], board[:4096], 3, 50000) if B >= 4096 else None
            if q3 is not None:
                eng.eval_batch(q3, seed=1)
                t1 = time.perf_counter()
                for i in range(5):
                    eng.eval_batch(q3, seed=i)
                dt = (time.perf_counter() - t1) / 5
                extras["configs[2]_4096x3x50k"] = {"call_ms_host_buffers": 1e3 * dt, "kernel_ms": eng.last_kernel_ms,
                                                   "hand_evals_per_s": 4096 * 3 * 50000 / dt}
                # the same batch in the PARITY mode (bit-exact replay of np.random.seed(s) per query): MT19937 walked on
                # the device, one wave per query, then the evaluation kernel on the draws it left in HBM
                eng.eval_batch(q3, seed=1, mode=npa.MODE_REPLAY_MT19937)
                t1 = time.perf_counter()
                for i in range(3):
                    eng.eval_batch(q3, seed=i, mode=npa.MODE_REPLAY_MT19937)
                dt = (time.perf_counter() - t1) / 3
                extras["configs[2]_replay"] = {"call_ms_host_buffers": 1e3 * dt, "kernel_ms_parse_plus_eval": eng.last_kernel_ms,
                                               "hand_evals_per_s": 4096 * 3 * 50000 / dt,
                                               "mt19937_words_per_s": 4096 * 50000 * 12.8 / dt}
                sc = side_counters.get("mcq_mt_parse_kernel")  # the stream walk: the dominant kernel of this mode
                if sc:
                    extras["configs[2]_replay"]["roofline"] = issue_roofline("mcq_mt_parse_kernel", sc.get("kernel_avg_ms_rocprof"),
                                                                             sc, fresh)
            # BASELINE configs[1] in its bit-exact form: ONE query of 100 000 runs replaying np.random.seed(0) -- latency
            q1r = npa.pack_queries([[npa.card_id("AH"), npa.card_id("KH")]], [[255] * 5], 2, 100000)
            r1 = eng.eval_batch(q1r, seed=0, mode=npa.MODE_REPLAY_MT19937).view(np.uint64).reshape(-1, 13)
            t1 = time.perf_counter()
            for i in range(5):
                eng.eval_batch(q1r, seed=i, mode=npa.MODE_REPLAY_MT19937)
            dt = (time.perf_counter() - t1) / 5
            extras["configs[1]_single_query_100k_replay"] = {
                "call_ms_host_buffers": 1e3 * dt, "kernel_ms_parse_plus_eval": eng.last_kernel_ms, "hand_evals_per_s": 2e5 / dt,
                "wins_seed0": int(r1[0, 2] + r1[0, 3]), "passes_seed0": int(r1[0, 1]),
                "equals_reference_known_answer": bool(int(r1[0, 2] + r1[0, 3]) == 65807 and int(r1[0, 1]) == 102091)}
            # ... and a 6-max one (few long queries are parsed with their 624-word state blocks side by side, mcq_mt_blocks.hpp)
            q6r = npa.pack_queries([[npa.card_id("AH"), npa.card_id("KH")]], [[255] * 5], 6, 100000)
            eng.eval_batch(q6r, seed=0, mode=npa.MODE_REPLAY_MT19937)
            t1 = time.perf_counter()
            for i in range(5):
                eng.eval_batch(q6r, seed=i, mode=npa.MODE_REPLAY_MT19937)
            dt = (time.perf_counter() - t1) / 5
            extras["single_query_6max_100k_replay"] = {"call_ms_host_buffers": 1e3 * dt, "kernel_ms_parse_plus_eval": eng.last_kernel_ms,
                                                       "hand_evals_per_s": 6e5 / dt}
            mtb = {k: side_counters[k].get("kernel_avg_ms_rocprof") for k in
                   ("mcq_mtb_generate_kernel", "mcq_mtb_scan_kernel", "mcq_mtb_parse_kernel") if k in side_counters}
            if mtb:   # the kernels of this call as rocprofv3 timed them (profiles/current.json); the first is one work-group
                extras["single_query_6max_100k_replay"]["kernels_ms_rocprof"] = dict(
                    mtb, counters_fresh=fresh, bound="the MT19937 recurrence: one dependent LDS round trip per 624-word state "
                                                     "block in ONE work-group (mcq_mtb_generate_kernel, ~0.19 us per block)")
            # run_montecarlo's other arguments: opponents restricted to the top quarter of the preflop classes
            # (the range the reference's own test uses, tests/test_montecarlo_python.py:215-222), 2048 states x 6 players x
            # 20k iterations through mcq_eval_batch_ext / mcq_eval_ext_kernel (candidate lists instead of the re-draw loop)
            with open(os.path.join(ROOT, "neuron_poker_amd", "preflop_classes.json")) as f:
                order = json.load(f)
            qx = npa.pack_queries(hole[:2048], board[:2048], 6, 20000) if B >= 2048 else None
            if qx is not None:
                ex = npa.pack_query_ext(2048, opp_range=npa.range_bits(order[-int(169 * 0.25):]))
                eng.eval_batch_ext(qx, ex, 1)
                t1 = time.perf_counter()
                for i in range(3):
                    eng.eval_batch_ext(qx, ex, i)
                dt = (time.perf_counter() - t1) / 3
                kms = eng.last_kernel_ms
                extras["ext_opponents_top25pct_2048x6x20k"] = {
                    "call_ms_host_buffers": 1e3 * dt, "kernel_ms": kms, "hand_evals_per_s": 2048 * 6 * 20000 / dt,
                    "roofline": issue_roofline("mcq_eval_ext_kernel<0>", kms, side_counters.get("mcq_eval_ext_kernel<0>"), fresh,
                                               model_ops_per_launch=2048.0 * 20000 * alg_ops_per_iteration(6, 0),
                                               model_basis="the plain path's 1242 lane-ops per iteration (SURVEY 8d)")}
                # the same range in the bit-exact mode (the reference's re-draw loop walked on numpy's MT19937 stream)
                qxs, exs = qx[:256], npa.pack_query_ext(256, opp_range=npa.range_bits(order[-int(169 * 0.25):]))
                eng.eval_batch_ext(qxs, exs, 1, mode=npa.MODE_REPLAY_MT19937)
                t1 = time.perf_counter()
                eng.eval_batch_ext(qxs, exs, 2, mode=npa.MODE_REPLAY_MT19937)
                dt = time.perf_counter() - t1
                extras["ext_opponents_top25pct_256x6x20k_replay"] = {"call_ms_host_buffers": 1e3 * dt,
                                                                     "hand_evals_per_s": 256 * 6 * 20000 / dt}
                # ... and as 16 384 queries of 2 500 iterations: the stream walk is serial per query (one wave each), so
                # its throughput is a matter of how many queries are in flight -- this is where it levels off
                g16 = np.random.default_rng(16384)
                c16 = g16.random((16384, 52)).argsort(axis=1)[:, :2].astype(np.uint8)
                q16 = npa.pack_queries(c16, np.full((16384, 5), 255, np.uint8), 6, 2500)
                e16 = npa.pack_query_ext(16384, opp_range=npa.range_bits(order[-int(169 * 0.25):]))
                eng.eval_batch_ext(q16[:256], e16[:256], 1, mode=npa.MODE_REPLAY_MT19937)
                t1 = time.perf_counter()
                eng.eval_batch_ext(q16, e16, 2, mode=npa.MODE_REPLAY_MT19937)
                dt = time.perf_counter() - t1
                extras["ext_opponents_top25pct_16384x6x2500_replay"] = {"call_ms_host_buffers": 1e3 * dt,
                                                                        "hand_evals_per_s": 16384 * 6 * 2500 / dt}
                # what a decision of the reference's agents asks for: ONE ranged query of 1000 iterations per call
                # (agent_*.py -> get_equity / run_montecarlo with opponent_range): the one-launch path of
                # mcq_eval_batch_ext, and the whole Python shim around it
                q1, e1 = npa.pack_queries(hole[:1], board[:1], 4, 1000), npa.pack_query_ext(1, opp_range=npa.range_bits(order[-42:]))
                eng.eval_batch_ext(q1, e1, 0)
                kms = eng.last_kernel_ms
                eng.set_kernel_timing(False)  # (timestamped launches cost a small call 6 us)
                for i in range(20):
                    eng.eval_batch_ext(q1, e1, i)
                t1 = time.perf_counter()
                for i in range(200):
                    eng.eval_batch_ext(q1, e1, i)
                dt = (time.perf_counter() - t1) / 200
                from neuron_poker_amd import montecarlo_hip as mh
                sim = mh.MonteCarlo(eng)
                for i in range(20):
                    sim.run_montecarlo([["AH", "KH"]], ["2C", "7D", "JS"], 4, 1, maxRuns=1000, timeout=0, ghost_cards="",
                                       opponent_range=0.25, seed=i)
                t1 = time.perf_counter()
                for i in range(200):
                    sim.run_montecarlo([["AH", "KH"]], ["2C", "7D", "JS"], 4, 1, maxRuns=1000, timeout=0, ghost_cards="",
                                       opponent_range=0.25, seed=i)
                dts = (time.perf_counter() - t1) / 200
                eng.set_kernel_timing(True)
                extras["ext_single_ranged_query_4x1000"] = {"call_us_host_buffers": 1e6 * dt, "kernel_ms": kms,
                                                            "run_montecarlo_call_us": 1e6 * dts,
                                                            "hand_evals_per_s": 4 * 1000 / dt}
            # BASELINE configs[3] (the 8-GPU config) on this one GPU: 65 536 states, flop / turn tables alternating, 6 players,
            # 20k iterations, host buffers; one rank of 8 would take an eighth of the queries
            h3, b3 = make_configs3_states()
            q4 = npa.pack_queries(h3, b3, 6, 20000)
            eng.eval_batch(q4, seed=1)  # warm-up at full size: the pinned staging buffers grow once
            t1 = time.perf_counter()
            eng.eval_batch(q4, seed=2)
            dt = time.perf_counter() - t1
            extras["configs[3]_65536x6x20k_on_one_gpu"] = {"call_ms_host_buffers": 1e3 * dt, "kernel_ms": eng.last_kernel_ms,
                                                           "hand_evals_per_s": 65536 * 6 * 20000 / dt}
            # the same batch through the single-process multi-GPU entry of the C ABI (mcq_multi_*: shards -> ONE
            # ncclAllReduce of the tally matrix): 1 shard, and the 8-way partition of configs[3] with all 8 shards on this
            # one device (what 8 GPUs run side by side, serialised here); the communicator has one rank on a 1-GPU box
            ref4 = eng.eval_batch(q4, seed=2)
            for shards in (1, 8):
                me = npa.MultiEngine([local_rank] * shards)
                me.eval_batch(q4, seed=1)
                t1 = time.perf_counter()
                r4 = me.eval_batch(q4, seed=2)
                dt = time.perf_counter() - t1
                tm = me.last_times_ms
                extras["configs[3]_native_multi_%d_shard%s_on_one_gpu" % (shards, "s" if shards > 1 else "")] = {
                    "call_ms_host_buffers": 1e3 * dt, "kernel_max_ms": tm["kernel_max"], "all_reduce_ms": tm["all_reduce"],
                    "hand_evals_per_s": 65536 * 6 * 20000 / dt, "rccl_version": me.info["rccl_version"],
                    "equals_single_context": bool(np.array_equal(r4, ref4))}
                me.close()
            # BASELINE configs[4], equity side only: one lock-step of 512 six-seat tables issues <= 2 x 512 queries of 1000
            # runs (gym_env/env.py:22,261-262) in ONE call; state mix as observed in reference episodes (SURVEY 8c F5:
            # table cards 0/3/4/5 = 59/19/11/10 %, players alive 2..6 = 41/28/17/9/6 %).  The table logic itself is
            # not part of this number; the whole loop is measured below.
            g = np.random.default_rng(512)
            nb = g.choice([0, 3, 4, 5], size=1024, p=[0.59, 0.19, 0.11, 0.11])
            npl = g.choice([2, 3, 4, 5, 6], size=1024, p=[0.41, 0.28, 0.17, 0.09, 0.05])
            hq, bq = [], []
            for i in range(1024):
                cards = g.choice(52, 2 + nb[i], replace=False)
                hq.append(cards[:2])
                bq.append(list(cards[2:]) + [255] * (5 - nb[i]))
            q5 = npa.pack_queries(hq, bq, npl, 1000)
            dt = call_time(q5, 50)
            extras["configs[4]_equity_side_only_1024x1000"] = {"call_ms_host_buffers": 1e3 * dt, "kernel_ms": eng.last_kernel_ms,
                                                               "hand_evals_per_s": float((npl * 1000).sum()) / dt,
                                                               "lock_steps_per_s": 1.0 / dt}
            extras["configs[4]_equity_side_only_1024x1000"]["roofline"] = issue_roofline(
                "mcq_eval_direct_kernel<0>", eng.last_kernel_ms, side_counters.get("mcq_eval_direct_kernel<0>"), fresh)
            # the two remaining kernels of the library, so that they have a number and a row in the kernel trace
            # (tools/profile.sh runs this command under rocprofv3): exact enumeration of AhKh heads-up
            # preflop under the reference's law -- C(50,5) completions x 990 candidate hands -- and of a three-player
            # flop; the showdown evaluator on 65 536 six-seat tables (7 cards per seat from one deck per table)
            qx = npa.pack_queries([[npa.card_id("AH"), npa.card_id("KH")]], [[255] * 5], 2, 1)
            eng.exact(qx)
            t1 = time.perf_counter()
            rx = eng.exact(qx)
            dt = time.perf_counter() - t1
            extras["exact_heads_up_preflop"] = {"call_ms": 1e3 * dt, "showdowns_per_s_nominal": 2118760 * 990 / dt,  # C(50,5) x 990, the order of magnitude
                                                "equity": float((int(rx["win"][0]) + int(rx["tie"][0])) / int(rx["runs"][0]))}
            qf = npa.pack_queries([[npa.card_id("AH"), npa.card_id("KH")]], [[0, 13, 30, 255, 255]], 3, 1)
            eng.exact(qf)
            t1 = time.perf_counter()
            eng.exact(qf)
            extras["exact_three_players_flop"] = {"call_ms": 1e3 * (time.perf_counter() - t1)}
            gs = np.random.default_rng(65536)
            decks = np.argsort(gs.random((65536, 52)), axis=1).astype(np.uint8)
            hands = np.concatenate([decks[:, 5:17].reshape(65536, 6, 2),
                                    np.repeat(decks[:, None, :5], 6, axis=1)], axis=2)
            eng.showdown(hands)
            t1 = time.perf_counter()
            eng.showdown(hands)
            dt = time.perf_counter() - t1
            extras["showdown_65536_tables_6_seats"] = {"call_ms_host_buffers": 1e3 * dt, "hand_evals_per_s": 65536 * 6 / dt}
            # BASELINE configs[4], the whole loop: 512 six-seat tables (seats as main.py:142-145 + two random seats) driven
            # by the native lock-step driver (mcq_tables_run): table rules on the host, ONE equity batch per lock-step
            seats = [("equity", .5, -.5), ("equity", .8, -.8), ("equity", .7, -.7), ("equity", .2, -.3), ("random",), ("random",)]
            eng.set_kernel_timing(False)   # the driver's contexts copy the engine's setting: plain launches
            try:
                tb = npa.Tables(eng, 512, seats, runs=1000, initial_stacks=100, small_blind=1, big_blind=2, seed=5)
                tb.run(50)
                s0 = tb.stats()
                t1 = time.perf_counter()
                tb.run(2000)
                dt = time.perf_counter() - t1
                s1 = tb.stats()
            finally:
                eng.set_kernel_timing(True)
            extras["configs[4]_native_driver_512_tables"] = {"lock_steps": 2000, "ms_per_lock_step": 1e3 * dt / 2000,
                                                             "env_steps_per_s": (s1["env_steps"] - s0["env_steps"]) / dt,
                                                             "equity_queries_per_s": (s1["queries"] - s0["queries"]) / dt}
            tb.close()

        try:
            side_measurements()
        except Exception as e:  # noqa: BLE001 -- recorded in the line, the headline stands
            extras["error"] = "%s: %s" % (type(e).__name__, e)
        out["other_configs"] = extras
    spot_ok = True
    if world == 1 and not args.no_cpu_baseline:
        try:
            # Parity spot check of the LAST TIMED launch: its rows are still in `tallies`; the oracle (MCQ-CTR mode:
            # the same specification as the production kernel, montecarlo_python.py:223-250 for what a row means)
            # recomputes 16 of them, spread over the batch, from the same (seed, query id).  Outside the timed region.
            from oracle import oracle as O
            O.lib()
            rows = np.unique(np.linspace(0, B - 1, 16).astype(np.int64))
            last_seed = seed + args.warmup + args.steps - 1
            raw = q.view(np.uint8).reshape(B, 16)
            equal, first_bad = True, None
            threads = max(1, min(len(rows), len(os.sched_getaffinity(0))))
            exp = np.zeros((len(rows), 13), np.uint64)
            import concurrent.futures as cf
            with cf.ThreadPoolExecutor(threads) as ex:  # one oracle call per row: the row's own query id keys its streams
                futs = {ex.submit(O.run_batch, O.MODE_CTR, raw[r:r + 1], last_seed, first_qid + int(r)): k
                        for k, r in enumerate(rows)}
                for f, k in futs.items():
                    exp[k] = f.result()[0]
            got = t[first_qid + rows]
            equal = bool(np.array_equal(got, exp))
            if not equal:
                first_bad = int(rows[np.nonzero((got != exp).any(1))[0][0]])
            out["parity_spot_check"] = {"rows": int(len(rows)), "equal": equal, "row_ids": [int(r) for r in rows],
                                        "seed": int(last_seed), "against": "oracle/mcq_oracle.c MODE_CTR (bit-exact integers: "
                                        "runs, passes, win, tie, by_type[9])", "first_mismatch": first_bad}
            spot_ok = equal
        except Exception as e:  # noqa: BLE001
            out["parity_spot_check"] = {"rows": 0, "equal": None, "error": "%s: %s" % (type(e).__name__, e)}
        try:
            out["cpu_baseline"] = cpu_baseline(N, runs)
            out["cpu_baseline_1thread"] = cpu_baseline(N, runs, seconds=5.0, threads=1)
            ref = cpu_reference_cpp(N)
            if ref:
                out["cpu_baseline_reference_cpp"] = ref
        except Exception as e:  # noqa: BLE001 -- the oracle is test infrastructure: its absence must not cost the line
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    print(json.dumps(out), file=JSON_OUT, flush=True)
    if grouped:
        dist.destroy_process_group()
    if not spot_ok:
        raise SystemExit("parity spot check FAILED: the timed launch's rows differ from the oracle's")


if __name__ == "__main__":
    main()
