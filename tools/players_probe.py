#!/usr/bin/env python3
"""Bulk-kernel time by the number of players (on the GPU box): 4096 preflop states x N players x 20 000 iterations."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuron_poker_amd as npa  # noqa: E402

eng = npa.Engine(0, kernel_times=True)
g = np.random.default_rng(4096)
hole = np.array([g.choice(52, 2, replace=False) for _ in range(4096)], np.uint8)
board = np.full((4096, 5), 255, np.uint8)
for n in [int(a) for a in sys.argv[1:]] or range(2, 11):
    q = npa.pack_queries(hole, board, n, 20000)
    eng.eval_batch(q, seed=1)
    ks = []
    for i in range(5):
        eng.eval_batch(q, seed=i)
        ks.append(eng.last_kernel_ms)
    k = float(np.median(ks))
    print("%2d players: kernel %.3f ms  %.3g hand-evals/s" % (n, k, 4096 * 20000 * n / (k * 1e-3)))
