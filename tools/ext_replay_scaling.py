import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import neuron_poker_amd as npa
from neuron_poker_amd import _lib
with open(os.path.join(os.path.dirname(_lib.__file__), "preflop_classes.json")) as f:
    ORDER = json.load(f)
top = lambda fr: _lib.range_bits(ORDER[-int(169 * fr):])
eng = npa.Engine(0, kernel_times=True)
g = np.random.default_rng(7)
for nq, runs in ((256, 5000), (2048, 5000), (8192, 5000), (16384, 2500)):
    cards = np.array([g.permutation(52)[:8] for _ in range(nq)], np.uint8)
    q = npa.pack_queries(cards[:, :2], np.full((nq, 5), 255, np.uint8), 6, runs)
    ex = _lib.pack_query_ext(nq, opp_range=top(0.25))
    eng.eval_batch_ext(q[:64], ex[:64], 1, mode=npa.MODE_REPLAY_MT19937)
    t0 = time.perf_counter()
    eng.eval_batch_ext(q, ex, 1, mode=npa.MODE_REPLAY_MT19937)
    dt = time.perf_counter() - t0
    print("ext parity top25: %6d queries x 6 x %d: %.1f ms call, %.1f ms kernels -> %.3g evals/s" % (nq, runs, dt * 1e3, eng.last_kernel_ms, nq * 6 * runs / dt), flush=True)
