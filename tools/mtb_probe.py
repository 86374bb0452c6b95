#!/usr/bin/env python3
"""Parity mode by state blocks (mcq_mt_blocks.hpp): a few calls of ONE 6-max 100 000-run query, for a kernel trace
(rocprofv3 --kernel-trace --stats -- python3 tools/mtb_probe.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neuron_poker_amd as npa  # noqa: E402

eng = npa.Engine(0)
npl = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
q = npa.pack_queries([[npa.card_id("AH"), npa.card_id("KH")]] * nq, [[255] * 5] * nq, npl, 100000)
for i in range(3):
    eng.eval_batch(q, seed=i, mode=npa.MODE_REPLAY_MT19937)
t = time.perf_counter()
for i in range(10):
    eng.eval_batch(q, seed=i, mode=npa.MODE_REPLAY_MT19937)
print("%d query x %d players x 100000, parity mode: %.3f ms per call" % (nq, npl, (time.perf_counter() - t) / 10 * 1e3))
