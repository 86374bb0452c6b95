#!/usr/bin/env python3
"""Call time of mcq_showdown (65 536 six-seat tables and smaller) on the GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neuron_poker_amd as npa  # noqa: E402

eng = npa.Engine(0)
gs = np.random.default_rng(65536)
decks = np.argsort(gs.random((65536, 52)), axis=1).astype(np.uint8)
hands = np.concatenate([decks[:, 5:17].reshape(65536, 6, 2), np.repeat(decks[:, None, :5], 6, axis=1)], axis=2)
for n in (1, 512, 8192, 65536):
    h = np.ascontiguousarray(hands[:n])
    for keys in (False, True):
        eng.showdown(h, want_keys=keys)
        reps = 50 if n < 65536 else 20
        t = time.perf_counter()
        for _ in range(reps):
            eng.showdown(h, want_keys=keys)
        dt = (time.perf_counter() - t) / reps
        print("%6d tables x 6 seats%s: %.1f us per call, %.3g hands/s" % (n, " + keys" if keys else "       ", 1e6 * dt, n * 6 / dt))
