import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import numpy as np
import neuron_poker_amd as npa
from small_probe import mix
eng = npa.Engine(0)
for n in (1, 1024):
    q, ev = mix(n)
    for i in range(8):
        t0 = time.perf_counter(); eng.eval_batch(q, seed=i); dt = time.perf_counter() - t0
        print("python call n=%d: %.1f us" % (n, dt * 1e6), file=sys.stderr)
