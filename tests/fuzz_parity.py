#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: random batches through the C ABI against the oracle, bit for bit.

    python tests/fuzz_parity.py [--seconds 120] [--seed 1]

Every round draws a batch of random valid queries (1-10 players, 0/3/4/5 table cards, ragged run counts incl. 1,
task boundaries and a few long ones) and checks, against oracle/ (test infrastructure):
  * production mode (MCQ_MODE_PHILOX), reference and uniform dealing law, random first_query_id;
  * parity mode (MCQ_MODE_REPLAY_MT19937);
  * shares of an iteration split (mcq_eval_batch_part) add up to the whole;
  * every eighth round a few LONG queries in parity mode (20 000 .. 400 000 runs: the stream parsed by state blocks, segments by
    jump-ahead, one or several rounds of them -- csrc/mcq_mt_blocks.hpp);
  * every fourth round a batch of extended queries (ranges, hero range, ghost cards, second known hand) in both modes;
  * every 64th round exact enumerations (1-3 players, river / turn / flop) against tests/hostsim, both laws.
Prints one summary line; exits non-zero on the first mismatch.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import neuron_poker_amd as npa  # noqa: E402
from oracle import oracle as O  # noqa: E402


def batch(g):
    n = int(g.integers(1, 160))
    hole, board, npl, runs = [], [], [], []
    for _ in range(n):
        nb = int(g.choice([0, 3, 4, 5]))
        c = g.permutation(52)[:2 + nb]
        hole.append(c[:2])
        board.append(list(c[2:]) + [255] * (5 - nb))
        npl.append(int(g.integers(1, 11)))
        runs.append(int(g.choice([1, 2, 63, 64, 65, 1000, 1023, 1024, 1025, 2048, int(g.integers(1, 6000)),
                                  int(g.integers(1, 40000)) if g.random() < 0.05 else 17])))
    return npa.pack_queries(hole, board, npl, runs)


RANKS = "23456789TJQKA"
CLASSES = [a + a for a in RANKS] + [RANKS[i] + RANKS[j] + t for i in range(13) for j in range(i) for t in "SO"]  # 169


def ext_round(g, eng):
    """Extended queries (opponent range, ghost cards, up to four further known hands -- cards or ranges --, hero as
    a range), a batch per call, both modes."""
    from neuron_poker_amd import _lib
    n = int(g.integers(1, 40))
    q = np.zeros(n, npa.QUERY_DTYPE)
    e = np.zeros(n, npa.QUERY_EXT_DTYPE)
    spec = []

    def some_range(lo, hi):
        return sorted(g.choice(CLASSES, size=int(g.integers(lo, hi)), replace=False))

    for i in range(n):
        nb = int(g.choice([0, 3, 4, 5]))
        c = [int(x) for x in g.permutation(52)[:14 + nb]]
        hero_range = some_range(25, 90) if g.random() < 0.3 else None
        opp = some_range(70, 169) if g.random() < 0.6 else None
        ghost = c[2:4] if g.random() < 0.3 else None
        n_known = int(g.choice([0, 0, 1, 1, 2, 3, 4]))
        known = [some_range(30, 100) if g.random() < 0.35 else c[4 + 2 * k:6 + 2 * k] for k in range(n_known)]
        npl = int(g.integers(max(2, 1 + n_known), 8))
        runs = int(g.choice([1, 64, 65, 300, 1024, 1100]))
        q[i] = _lib.pack_queries([[0, 1] if hero_range else c[:2]], [c[14:] + [255] * (5 - nb)], npl, runs)[0]
        if hero_range:
            q["hole"][i] = 0
        e[i] = _lib.pack_query_ext(1, ghost=ghost, known=[h if isinstance(h[0], int) else _lib.range_bits(h) for h in known],
                                   hero_range=_lib.range_bits(hero_range) if hero_range else None,
                                   opp_range=_lib.range_bits(opp) if opp else None)[0]
        spec.append((hero_range if hero_range else c[:2], c[14:], npl, runs, known, ghost, opp))
    seed, first = int(g.integers(0, 2 ** 31)), int(g.integers(0, 2 ** 20))
    for mode, om in ((npa.MODE_PHILOX, O.MODE_CTR), (npa.MODE_REPLAY_MT19937, O.MODE_MT)):
        got = eng.eval_batch_ext(q, e, seed, first_query_id=first, mode=mode).view(np.uint64).reshape(-1, 13)
        for i, (hero, board, npl, runs, known, ghost, opp) in enumerate(spec):
            s_i = (seed + first + i) & 0xFFFFFFFF if om == O.MODE_MT else seed
            want = O.run_ex(om, hero, board, npl, runs, s_i, qid=first + i, known=known, ghost=ghost, opp_range=opp)["tallies"]
            assert np.array_equal(got[i], want), ("ext", mode, i, spec[i])
    return n, int(q["runs"].astype(np.int64).sum())


def long_round(g, eng):
    """A few long queries in parity mode: the walk by state blocks (scan in parts, stitch, parse), the generator in segments
    with jump-ahead when the estimate says so, more than one round of 4096 blocks now and then."""
    n = int(g.integers(1, 9))
    hole, board, npl, runs = [], [], [], []
    for _ in range(n):
        nb = int(g.choice([0, 0, 3, 4, 5]))
        c = g.permutation(52)[:2 + nb]
        hole.append(c[:2])
        board.append(list(c[2:]) + [255] * (5 - nb))
        npl.append(int(g.integers(1, 11)))
        runs.append(int(g.integers(20000, 400000 if g.random() < 0.15 else 120000)))
    q = npa.pack_queries(hole, board, npl, runs)
    s32, first = int(g.integers(0, 2 ** 32)), int(g.integers(0, 2 ** 40))
    got = eng.eval_batch(q, s32, first_query_id=first, mode=npa.MODE_REPLAY_MT19937).view(np.uint64).reshape(-1, 13)
    want = O.run_batch(O.MODE_MT, q.view(np.uint8).reshape(-1, 16), s32, first_qid=first, threads=16)
    assert np.array_equal(got, want), ("long replay", [int(x) for x in npl], runs, s32, first)
    return n, int(sum(runs))


def exact_round(g, eng):
    """Exact enumeration on the GPU against the host build of the same lane code (tests/hostsim), integers."""
    from tests import hostsim as H
    n = 6
    hole, board, npl = [], [], []
    for i in range(n):
        nb, players = [(5, 1), (5, 2), (4, 2), (3, 2), (5, 3), (4, 1)][i]
        c = g.permutation(52)[:2 + nb]
        hole.append(c[:2])
        board.append(list(c[2:]) + [255] * (5 - nb))
        npl.append(players)
    q = npa.pack_queries(hole, board, npl, 1)
    for law, uni in (("reference", False), ("uniform", True)):
        got = eng.exact(q, law).view(np.uint64).reshape(-1, 13)
        for i in range(n):
            assert np.array_equal(got[i], H.exact(q[i:i + 1].view(np.uint8).reshape(16), uni)), ("exact", law, i)
    return 2 * n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    g = np.random.default_rng(a.seed)
    eng = npa.Engine(0)
    t0 = time.time()
    rounds = queries = iters = ext_q = ext_i = exact_n = long_q = long_i = 0
    t_said = t0
    while time.time() - t0 < a.seconds:
        if time.time() - t_said > 60:   # a sign of life for long runs (the GPU pool kills a command that stays silent)
            t_said = time.time()
            print("... %d rounds, %.0f s" % (rounds, t_said - t0), flush=True)
        q = batch(g)
        raw = q.view(np.uint8).reshape(-1, 16)
        seed, first = int(g.integers(0, 2 ** 63)), int(g.integers(0, 2 ** 40))
        want = O.run_batch(O.MODE_CTR, raw, seed, first_qid=first, threads=16)
        got = eng.eval_batch(q, seed, first_query_id=first).view(np.uint64).reshape(-1, 13)
        assert np.array_equal(got, want), ("philox", rounds)
        n_parts = int(g.integers(2, 9))
        tot = sum(eng.eval_batch(q, seed, first_query_id=first, part=(p, n_parts)).view(np.uint64).reshape(-1, 13)
                  for p in range(n_parts))
        assert np.array_equal(tot, want), ("parts", rounds, n_parts)
        eng.set_dealing_law("uniform")
        got = eng.eval_batch(q, seed, first_query_id=first).view(np.uint64).reshape(-1, 13)
        eng.set_dealing_law("reference")
        assert np.array_equal(got, O.run_batch(O.MODE_CTR_UNIFORM, raw, seed, first_qid=first, threads=16)), ("uniform", rounds)
        s32 = int(g.integers(0, 2 ** 32))
        got = eng.eval_batch(q, s32, first_query_id=first, mode=npa.MODE_REPLAY_MT19937).view(np.uint64).reshape(-1, 13)
        assert np.array_equal(got, O.run_batch(O.MODE_MT, raw, s32, first_qid=first, threads=16)), ("replay", rounds)
        rounds += 1
        queries += 4 * len(q)
        iters += 4 * int(q["runs"].astype(np.int64).sum())
        if rounds % 4 == 0:
            nq, ni = ext_round(g, eng)
            ext_q += 2 * nq
            ext_i += 2 * ni
        if rounds % 8 == 0:
            nq, ni = long_round(g, eng)
            long_q += nq
            long_i += ni
        if rounds % 64 == 0:
            exact_n += exact_round(g, eng)
    print("fuzz parity: %d rounds, %d query evaluations, %d iterations in 4 configurations; %d extended-query "
          "evaluations, %d iterations in both modes -- all bit-exact against the oracle; %d exact enumerations equal "
          "to the host build of the lane code; %d long queries (%d iterations) in parity mode by state blocks (%.0f s, seed %d)"
          % (rounds, queries, iters, ext_q, ext_i, exact_n, long_q, long_i, time.time() - t0, a.seed))


if __name__ == "__main__":
    main()
