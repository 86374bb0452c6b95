#!/usr/bin/env python3
"""Timing of mcq_exact_batch (exact enumeration) on the GPU box."""
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuron_poker_amd as npa  # noqa: E402

eng = npa.Engine(0)


def q(hole, board, n):
    return npa.pack_queries([hole], [board + [255] * (5 - len(board))], n, 1)


for name, qq in [("HU river", q([50, 46], [0, 21, 31, 38, 40], 2)), ("HU turn", q([50, 46], [0, 21, 31, 38], 2)),
                 ("HU flop", q([50, 46], [0, 21, 31], 2)), ("HU preflop", q([50, 46], [], 2)),
                 ("3-way river", q([50, 46], [0, 21, 31, 38, 40], 3)), ("3-way turn", q([50, 46], [0, 21, 31, 38], 3)),
                 ("3-way flop", q([50, 46], [0, 21, 31], 3)), ("3-way preflop", q([50, 46], [], 3))]:
    for law in ("reference", "uniform"):
        if name == "3-way preflop" and law == "uniform":
            continue
        eng.exact(q([50, 46], [0, 21, 31, 38, 40], 2), law)
        t0 = time.perf_counter()
        r = eng.exact(qq, law)[0]
        dt = time.perf_counter() - t0
        print("%-14s %-9s %9.3f ms  total weight %d  equity %.6f" % (name, law, dt * 1e3, int(r["runs"]),
                                                                     (int(r["win"]) + int(r["tie"])) / int(r["runs"])), flush=True)
