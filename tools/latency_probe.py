import sys; sys.path.insert(0, '/root/repo')
import numpy as np, time
import neuron_poker_amd as npa
eng = npa.Engine(0, kernel_times=True)
for runs in (1, 64, 1024, 16384, 100000, 1000000):
    q = npa.pack_queries([[50, 46]], [[255]*5], 2, runs)
    for _ in range(5): eng.eval_batch(q, seed=1)
    ks = []
    t0 = time.perf_counter()
    for i in range(50):
        eng.eval_batch(q, seed=i); ks.append(eng.last_kernel_ms)
    dt = (time.perf_counter() - t0) / 50
    print("runs %8d  call %.1f us  kernel %.1f us" % (runs, dt * 1e6, 1e3 * np.median(ks)))
