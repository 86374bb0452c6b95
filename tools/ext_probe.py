#!/usr/bin/env python3
"""Throughput of the extended-query kernel (ranges, known hands) beside the plain path, on the GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neuron_poker_amd as npa  # noqa: E402
from neuron_poker_amd import _lib  # noqa: E402

with open(os.path.join(ROOT, "neuron_poker_amd", "preflop_classes.json")) as f:
    ORDER = json.load(f)


def top(frac):
    return _lib.range_bits(ORDER[-int(169 * frac):])


def main():
    eng = npa.Engine(0, kernel_times=True)
    g = np.random.default_rng(7)
    B, N, runs = 2048, 6, 20000
    cards = np.array([g.permutation(52)[:8] for _ in range(B)], np.uint8)
    q = npa.pack_queries(cards[:, :2], np.full((B, 5), 255, np.uint8), N, runs)
    evals = B * N * runs

    def rate(f, reps=3):
        f()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        return evals * reps / (time.perf_counter() - t0)

    print("plain path                         %.3g hand-evals/s  (kernel %.2f ms)" % (rate(lambda: eng.eval_batch(q, 1)), eng.last_kernel_ms))
    for name, ext in [("ext, every class", _lib.pack_query_ext(B)),
                      ("ext, opponents top 50 %", _lib.pack_query_ext(B, opp_range=top(0.5))),
                      ("ext, opponents top 25 %", _lib.pack_query_ext(B, opp_range=top(0.25))),
                      ("ext, opponents top 10 %", _lib.pack_query_ext(B, opp_range=top(0.10))),
                      ("ext, hero range top 25 %", _lib.pack_query_ext(B, hero_range=top(0.25))),
                      ("ext, 3 known hands (1 range), top 25 %", None)]:
        if ext is None:
            ext = _lib.pack_query_ext(B, opp_range=top(0.25), known=[[0, 0], top(0.3), [0, 0]])
            ext["known"]["cards"][:, 0] = cards[:, 2:4]
            ext["known"]["cards"][:, 2] = cards[:, 4:6]
        r = rate(lambda: eng.eval_batch_ext(q, ext, 1))
        print("%-34s %.3g hand-evals/s  (kernel %.2f ms)" % (name, r, eng.last_kernel_ms))
    for nq in (256, 2048):   # the bit-exact mode: numpy's MT19937 stream walked on the device, one wave per query
        ex = _lib.pack_query_ext(nq, opp_range=top(0.25))
        r = rate(lambda: eng.eval_batch_ext(q[:nq], ex, 1, mode=npa.MODE_REPLAY_MT19937), 1)
        print("%-34s %.3g hand-evals/s (%d queries, stream walk + evaluation %.1f ms)" %
              ("ext, parity mode, top 25 %", r * nq / B, nq, eng.last_kernel_ms))


if __name__ == "__main__":
    main()
