#!/usr/bin/env python3
"""Parity mode (MT19937 walked on the device): call times of BASELINE configs[2] (4096 x 3 players x 50k), of a
6-max batch and of ONE 100 000-run query, each checked against the reference's known answers where they exist."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neuron_poker_amd as npa  # noqa: E402
from bench import make_states  # noqa: E402

eng = npa.Engine(0, kernel_times=True)
hole, board = make_states(4096, 0)


def timed(q, reps, label, evals):
    eng.eval_batch(q, seed=1, mode=npa.MODE_REPLAY_MT19937)
    t = time.perf_counter()
    for i in range(reps):
        eng.eval_batch(q, seed=i, mode=npa.MODE_REPLAY_MT19937)
    dt = (time.perf_counter() - t) / reps
    print("%-44s %8.3f ms per call  (kernels %.3f ms)  %.3g hand-evals/s" % (label, 1e3 * dt, eng.last_kernel_ms, evals / dt))


timed(npa.pack_queries(hole, board, 3, 50000), 3, "configs[2] 4096 x 3 players x 50k", 4096 * 3 * 50000)
timed(npa.pack_queries(hole, board, 6, 20000), 3, "4096 x 6 players x 20k", 4096 * 6 * 20000)
timed(npa.pack_queries(hole, board, 10, 10000), 3, "4096 x 10 players x 10k", 4096 * 10 * 10000)
q1 = npa.pack_queries([[npa.card_id("AH"), npa.card_id("KH")]], [[255] * 5], 2, 100000)
r = eng.eval_batch(q1, seed=0, mode=npa.MODE_REPLAY_MT19937)
print("AhKh heads-up 100k seed 0: wins %d passes %d (reference: 65807 / 102091)" % (int(r["win"][0] + r["tie"][0]), int(r["passes"][0])))
timed(q1, 5, "configs[1] ONE query x 2 players x 100k", 2e5)
# few long queries: the state blocks of a query side by side (mcq_mt_blocks.hpp); MCQ_MT_BLOCKS=0 in the environment
# gives the serial walk's times for the same lines
for npl, runs, nq in ((6, 100000, 1), (10, 100000, 1), (6, 100000, 8), (6, 20000, 64), (2, 1000000, 1)):
    timed(npa.pack_queries(hole[:nq], board[:nq], npl, runs), 3, "%d query x %d players x %d%s" %
          (nq, npl, runs, "" if os.environ.get("MCQ_MT_BLOCKS", "1") != "0" else " (serial walk)"), nq * npl * runs)
