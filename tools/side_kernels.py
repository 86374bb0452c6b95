#!/usr/bin/env python3
"""The workloads behind profiles/current.json's `side_kernels` block (tools/profile.sh runs this under rocprofv3):
the kernels of the library that are not the bulk kernel, each on the batch bench.py's `other_configs` quotes.

  mcq_mt_parse_kernel          BASELINE configs[2] in its bit-exact form: 4096 states x 3 players x 50 000 runs, parity mode
  mcq_eval_kernel<1, false>    ... the evaluation kernel behind it (reads the draws the stream walk left in HBM)
  mcq_eval_direct_kernel<0>    one lock-step of configs[4]: 1024 queries x 1000 runs, state mix of reference episodes
  mcq_eval_ext_kernel<0>       2048 states x 6 players x 20 000 runs, opponents restricted to the top quarter of the classes
  mcq_eval_ext_kernel<1> / mcq_mt_parse_ext_kernel   256 of those in the bit-exact mode
  mcq_mtb_generate / scan / parse_kernel              ONE 6-max query of 100 000 runs in the bit-exact mode (state blocks side by side)

Prints one JSON object: per kernel the units one launch processes, so that instruction counts can be put per unit.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neuron_poker_amd as npa  # noqa: E402
from bench import make_states  # noqa: E402

REPS = int(os.environ.get("SIDE_REPS", "2"))


def main():
    eng = npa.Engine(0, kernel_times=True)
    hole, board = make_states(4096, 0)
    units = {}
    q3 = npa.pack_queries(hole, board, 3, 50000)
    for i in range(REPS):
        eng.eval_batch(q3, seed=i, mode=npa.MODE_REPLAY_MT19937)
    units["mcq_mt_parse_kernel"] = {"iterations": 4096 * 50000, "mt_words_approx": 4096 * 50000 * 12.8}
    units["mcq_eval_kernel<1, false>"] = {"iterations": 4096 * 50000}

    g = np.random.default_rng(512)
    nb = g.choice([0, 3, 4, 5], size=1024, p=[0.59, 0.19, 0.11, 0.11])
    npl = g.choice([2, 3, 4, 5, 6], size=1024, p=[0.41, 0.28, 0.17, 0.09, 0.05])
    hq, bq = [], []
    for i in range(1024):
        cards = g.choice(52, 2 + nb[i], replace=False)
        hq.append(cards[:2])
        bq.append(list(cards[2:]) + [255] * (5 - nb[i]))
    q5 = npa.pack_queries(hq, bq, npl, 1000)
    for i in range(REPS + 2):
        eng.eval_batch(q5, seed=i)
    units["mcq_eval_direct_kernel<0>"] = {"iterations": 1024 * 1000, "hand_evals": int((npl * 1000).sum())}

    with open(os.path.join(ROOT, "neuron_poker_amd", "preflop_classes.json")) as f:
        order = json.load(f)
    top25 = npa.range_bits(order[-int(169 * 0.25):])
    qx = npa.pack_queries(hole[:2048], board[:2048], 6, 20000)
    ex = npa.pack_query_ext(2048, opp_range=top25)
    for i in range(REPS):
        eng.eval_batch_ext(qx, ex, i)
    units["mcq_eval_ext_kernel<0>"] = {"iterations": 2048 * 20000}
    for i in range(REPS):
        eng.eval_batch_ext(qx[:256], npa.pack_query_ext(256, opp_range=top25), i, mode=npa.MODE_REPLAY_MT19937)
    units["mcq_eval_ext_kernel<1>"] = {"iterations": 256 * 20000}
    units["mcq_mt_parse_ext_kernel"] = {"iterations": 256 * 20000}
    # few long queries of the bit-exact mode: the state blocks of a query side by side (mcq_mt_blocks.hpp)
    q6 = npa.pack_queries([[npa.card_id("AH"), npa.card_id("KH")]], [[255] * 5], 6, 100000)
    for i in range(REPS + 1):
        eng.eval_batch(q6, seed=i, mode=npa.MODE_REPLAY_MT19937)
    for k in ("mcq_mtb_generate_kernel", "mcq_mtb_scan_kernel", "mcq_mtb_parse_kernel"):
        units[k] = {"iterations": 100000, "state_blocks_approx": 3900}
    print(json.dumps(units))


if __name__ == "__main__":
    main()
