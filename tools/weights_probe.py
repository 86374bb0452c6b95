#!/usr/bin/env python3
"""Measure the cost of one wave task per (n_players, n_board) and the balance of a mixed batch (GPU box).
Feeds mcq_task_weight() in csrc/mcq_device.hpp (scheduling only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuron_poker_amd as npa  # noqa: E402

eng = npa.Engine(0, kernel_times=True)
g = np.random.default_rng(3)


def batch(n_players, n_board, B, runs):
    hole, board = [], []
    for _ in range(B):
        c = g.choice(52, 2 + n_board, replace=False)
        hole.append(c[:2])
        board.append(list(c[2:]) + [255] * (5 - n_board))
    return npa.pack_queries(hole, board, n_players, runs)


def kernel_ms(q, reps=5):
    eng.eval_batch(q, seed=1)
    ks = []
    for i in range(reps):
        eng.eval_batch(q, seed=2 + i)
        ks.append(eng.last_kernel_ms)
    return float(np.median(ks))


B, runs = 4096, 20480  # 20 tasks per query -> 81920 tasks
per_task = {}
print("n_players n_board  ms   ns/task(per wave-slot)  current weight  ratio")
for nb in (0, 3, 4, 5):
    for n in (1, 2, 3, 6, 10):
        ms = kernel_ms(batch(n, nb, B, runs))
        tasks = B * runs / 1024
        w = 45 * n + 65 * (n - 1) + 40 * (5 - nb) + 60
        per_task[(n, nb)] = ms / tasks
        print("%9d %7d %7.3f %10.2f %14d %8.4f" % (n, nb, ms, 1e6 * ms / tasks, w, 1e6 * ms / tasks / w))
# mixed batch: predicted = sum of homogeneous per-task times
mix = []
pred = 0.0
for (n, nb), t in per_task.items():
    mix.append(batch(n, nb, 512, runs))
    pred += 512 * runs / 1024 * t
q = np.concatenate(mix)
ms = kernel_ms(q)
print("mixed batch (grouped by type): %.3f ms, sum of homogeneous parts: %.3f ms, ratio %.3f" % (ms, pred, ms / pred))
perm = g.permutation(len(q))
ms2 = kernel_ms(q[perm])
print("mixed batch (shuffled):        %.3f ms, ratio %.3f" % (ms2, ms2 / pred))
