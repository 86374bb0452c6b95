#!/usr/bin/env python3
"""One 1000-run query per call by players / table cards (on the GPU box): kernel and call time.  MCQ_SPLIT_MAX is read
by the engine at creation."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuron_poker_amd as npa  # noqa: E402

eng = npa.Engine(0)
for board in ([255] * 5, [0, 13, 30, 255, 255]):
    for n in (2, 3, 4, 6, 9):
        q = npa.pack_queries([[50, 46]], [board], n, 1000)
        eng.set_kernel_timing(True)
        ks = []
        for i in range(300):
            eng.eval_batch(q, seed=i)
            ks.append(eng.last_kernel_ms)
        eng.set_kernel_timing(False)
        for i in range(50):
            eng.eval_batch(q, seed=i)
        t0 = time.perf_counter()
        for i in range(1000):
            eng.eval_batch(q, seed=i)
        dt = (time.perf_counter() - t0) / 1000
        print("split<=%s  %d players, %s: kernel %5.1f us  call %5.1f us" % (os.environ.get("MCQ_SPLIT_MAX", "4"), n, "preflop" if board[0] == 255 else "flop", 1e3 * float(np.median(ks)), 1e6 * dt))
