#!/usr/bin/env python3
"""Small-batch latency probe (on the GPU box): call / kernel time of the reference's call pattern -- few 1000-run
queries per call -- through the host entry, the device entry and the table driver.  MCQ_LOAD_WAVES / MCQ_SPLIT_MAX
are read by the engine at creation."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuron_poker_amd as npa  # noqa: E402


def mix(n, seed=512):
    g = np.random.default_rng(seed)
    nb = g.choice([0, 3, 4, 5], size=n, p=[0.59, 0.19, 0.11, 0.11])
    npl = g.choice([2, 3, 4, 5, 6], size=n, p=[0.41, 0.28, 0.17, 0.09, 0.05])
    hq, bq = [], []
    for i in range(n):
        cards = g.choice(52, 2 + nb[i], replace=False)
        hq.append(cards[:2])
        bq.append(list(cards[2:]) + [255] * (5 - nb[i]))
    return npa.pack_queries(hq, bq, npl, 1000), int(npl.sum()) * 1000


def timeit(f, reps):
    for _ in range(5):
        f(0)
    ks = []
    t0 = time.perf_counter()
    for i in range(reps):
        ks.append(f(i))
    return (time.perf_counter() - t0) / reps, float(np.median(ks))


def main():
    eng = npa.Engine(0)
    tag = "LOAD_WAVES=%s SPLIT_MAX=%s" % (os.environ.get("MCQ_LOAD_WAVES", "-"), os.environ.get("MCQ_SPLIT_MAX", "-"))
    cases = [("1 x 1 run", npa.pack_queries([[50, 46]], [[255] * 5], 2, 1), 2),
             ("1 x 1000 runs", npa.pack_queries([[50, 46]], [[255] * 5], 2, 1000), 2000),
             ("1 x 100k runs", npa.pack_queries([[50, 46]], [[255] * 5], 2, 100000), 200000)]
    for n in (64, 512, 1024, 4096):
        q, ev = mix(n)
        cases.append(("%d x 1000 runs (mix)" % n, q, ev))
    for name, q, evals in cases:
        def host(i):
            eng.eval_batch(q, seed=i)
            return eng.last_kernel_ms
        eng.set_kernel_timing(True)      # kernel time from timestamped launches, call time from plain ones (the default)
        _, k = timeit(host, 200)
        eng.set_kernel_timing(False)
        dt, _ = timeit(host, 200)
        print("%-26s host entry: call %7.1f us  kernel %7.1f us  %.3g evals/s   [%s]" % (name, dt * 1e6, k * 1e3, evals / dt, tag))
    from neuron_poker_amd import montecarlo_hip as mh
    mh.seed(1)
    hole, board = {"AH", "KH"}, {"2C", "7D", "JS"}
    for _ in range(20):
        mh.get_equity(hole, board, 4, 1000)
    t0 = time.perf_counter()
    for _ in range(2000):
        mh.get_equity(hole, board, 4, 1000)
    print("montecarlo_hip.get_equity(set, set, 4, 1000) as gym_env/env.py:261 calls it: %.1f us per call" % ((time.perf_counter() - t0) / 2000 * 1e6))
    sim = mh.MonteCarlo()
    t0 = time.perf_counter()
    for _ in range(500):
        sim.run_montecarlo([["AH", "KH"]], ["2C", "7D", "JS"], 4, 1, maxRuns=1000, timeout=0, ghost_cards="")
    print("MonteCarlo().run_montecarlo(..., maxRuns=1000): %.1f us per call" % ((time.perf_counter() - t0) / 500 * 1e6))
    try:
        import torch
        dev = torch.device("cuda", 0)
        for n in (1, 512, 1024):
            q, ev = mix(n)
            d_q = torch.from_numpy(q.view(np.uint8).reshape(n, 16).copy()).to(dev)
            out = torch.zeros((n, 13), dtype=torch.int64, device=dev)
            s = torch.cuda.current_stream()

            def devf(i):
                eng.eval_batch_device(d_q.data_ptr(), n, i, out.data_ptr(), stream=s.cuda_stream)
                torch.cuda.synchronize()
                kt = eng.kernel_times(1)
                return float(kt[0]) if len(kt) else 0.0
            eng.set_kernel_timing(True)
            _, k = timeit(devf, 200)
            eng.set_kernel_timing(False)
            dt, _ = timeit(devf, 200)
            print("%-26s device entry: call+sync %7.1f us  kernel %7.1f us" % ("%d x 1000 runs (mix)" % n, dt * 1e6, k * 1e3))

            def devs(i):
                eng.eval_batch_device_small(d_q.data_ptr(), n, i, out.data_ptr(), stream=s.cuda_stream)
                torch.cuda.synchronize()
                kt = eng.kernel_times(1)
                return float(kt[0]) if len(kt) else 0.0
            eng.set_kernel_timing(True)
            _, k = timeit(devs, 200)
            eng.set_kernel_timing(False)
            dt, _ = timeit(devs, 200)
            print("%-26s device entry, one launch (small): call+sync %7.1f us  kernel %7.1f us" % ("%d x 1000 runs (mix)" % n, dt * 1e6, k * 1e3))
    except ImportError:
        pass
    seats = [("equity", .5, -.5), ("equity", .8, -.8), ("equity", .7, -.7), ("equity", .2, -.3), ("random",), ("random",)]
    for T in (512,):
        for overlap in (True, False):
            tb = npa.Tables(eng, T, seats, runs=1000, initial_stacks=100, small_blind=1, big_blind=2, seed=5, overlap=overlap)
            tb.run(100)
            s0 = tb.stats()
            t1 = time.perf_counter()
            tb.run(3000)
            dt = time.perf_counter() - t1
            s1 = tb.stats()
            print("%d tables %s: %.1f us per lock-step, %.3g env-steps/s" % (T, "two streams" if overlap else "one stream", dt / 3000 * 1e6,
                                                                           (s1["env_steps"] - s0["env_steps"]) / dt))
            tb.close()


if __name__ == "__main__":
    main()
