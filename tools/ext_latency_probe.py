import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neuron_poker_amd import montecarlo_hip as mh
mh.seed(1)
sim = mh.MonteCarlo()
for rng in (1, 0.25):
    for _ in range(30):
        sim.run_montecarlo([["AH", "KH"]], ["2C", "7D", "JS"], 4, 1, maxRuns=1000, timeout=0, ghost_cards="", opponent_range=rng)
    t0 = time.perf_counter()
    for _ in range(500):
        sim.run_montecarlo([["AH", "KH"]], ["2C", "7D", "JS"], 4, 1, maxRuns=1000, timeout=0, ghost_cards="", opponent_range=rng)
    print("run_montecarlo(maxRuns=1000, opponent_range=%s): %.1f us per call" % (rng, (time.perf_counter() - t0) / 500 * 1e6))
# the same through the C ABI's batch entry (no Python range handling): 1 and 8 ranged queries per call, and the general
# path (MCQ_EXT_SMALL=0 contexts are made by the environment: run the script twice)
import numpy as np
import neuron_poker_amd as npa
from neuron_poker_amd import _lib
eng = npa.Engine(0, kernel_times=False)
with open(os.path.join(os.path.dirname(_lib.__file__), "preflop_classes.json")) as f:
    import json
    order = json.load(f)
opp = _lib.range_bits(order[-42:])
for n, runs in ((1, 1000), (8, 1000), (1, 8192), (1, 100)):
    g = np.random.default_rng(3)
    cards = np.array([g.permutation(52)[:5] for _ in range(n)], np.uint8)
    q = npa.pack_queries(cards[:, :2], np.concatenate([cards[:, 2:5], np.full((n, 2), 255, np.uint8)], 1), 4, runs)
    e = npa.pack_query_ext(n, opp_range=opp)
    for _ in range(30):
        eng.eval_batch_ext(q, e, 1)
    t0 = time.perf_counter()
    for i in range(500):
        eng.eval_batch_ext(q, e, i)
    dt = (time.perf_counter() - t0) / 500
    eng.set_kernel_timing(True)
    eng.eval_batch_ext(q, e, 1)
    k = eng.last_kernel_ms
    eng.set_kernel_timing(False)
    print("eval_batch_ext(%d ranged queries x %d runs, 4 players, flop): %.1f us per call, kernel %.1f us (MCQ_EXT_SMALL=%s)"
          % (n, runs, dt * 1e6, 1e3 * k, os.environ.get("MCQ_EXT_SMALL", "1")))
