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
