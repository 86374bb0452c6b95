"""neuron_poker_amd/table_driver.py against seeded episodes of the REFERENCE's own table
(tests/golden/env_traces.json, recorded by tests/golden/gen_env_traces.py from gym_env/env.py + gym_env/cycle.py
with agents/agent_consider_equity.py): every equity query, every action with its legal moves and stacks, the
per-hand stacks and the winner must be identical.  Equity comes from the parity mode coupled to numpy's global
stream (host-compiled lane code here; the GPU variant is in tests/test_gpu_parity.py), so the whole trajectory --
dealing included, which shares np.random with the equity calls (env.py:142,680,686) -- is reproduced."""
import json
import os

import numpy as np
import pytest

from neuron_poker_amd import table_driver as td
from oracle import oracle as O
from tests import hostsim as H

G = os.path.join(os.path.dirname(__file__), "golden")


def _equity_numpy_stream(hole, board, alive):
    b = list(board) + [255] * (5 - len(board))
    q = O.pack_queries([hole], [b], alive, 1000)[0]
    r = H.run_replay_numpy_stream(q)
    return float(int(r[2] + r[3]) / 1000)


def _showdown(hands):
    return O.best_hand(hands)[0]


def _play(ep):
    table = td.TableSim([td.equity_policy(c, b) for c, b in ep["policies"]], initial_stacks=ep["stacks"],
                        showdown=_showdown)
    table.log = []
    np.random.seed(ep["seed"])
    g = table.episode()
    try:
        q = next(g)
        while True:
            q = g.send(_equity_numpy_stream(*q))
    except StopIteration:
        pass
    return table


def test_reference_episodes_are_reproduced_event_by_event():
    with open(os.path.join(G, "env_traces.json")) as f:
        eps = json.load(f)
    assert len(eps) >= 8
    for ep in eps:
        t = _play(ep)
        ref = ep["events"]
        for i, (a, b) in enumerate(zip(t.log, ref)):
            assert a == b, (ep["seed"], i, a, b)
        assert len(t.log) == len(ref)
        assert t.winner_ix == ep["winner"]
        assert [float(s) for s in t.stacks] == ep["final_stacks"]
        assert [[float(x) for x in row] for row in t.funds_history] == [[float(x) for x in row] for row in ep["funds_history"]]
        assert [int(x) for x in np.random.randint(0, 2 ** 32, size=2, dtype=np.uint32)] == ep["np_next_words"]


def test_batch_lock_step_equals_sequential_tables():
    """TableBatch only interleaves the tables' generators: with a deterministic equity the outcome per table is
    what the table gives when played alone."""
    def eq_fn(hole, board, alive):
        return ((hole[0] * 7 + hole[1] * 13 + sum(board) + alive) % 97) / 97.0

    def mk(seed):
        g = np.random.default_rng(seed)
        pol = [td.equity_policy(.3, .5), td.equity_policy(.45, .6), td.equity_policy(.2, .75), td.random_policy(g)]
        return td.TableSim(pol, showdown=_showdown, randint=lambda n, g=g: int(g.integers(0, n)))

    solo = []
    for s in range(6):
        t = mk(s)
        gen = t.episode()
        try:
            q = next(gen)
            while True:
                q = gen.send(eq_fn(*q))
        except StopIteration:
            pass
        solo.append((t.winner_ix, [float(x) for x in t.stacks], t.queries, t.env_steps))
    batch = td.TableBatch([mk(s) for s in range(6)])
    while batch.running():
        batch.step(lambda hole, board, npl: [eq_fn([int(x) for x in h], [int(c) for c in b if c != 255], int(a))
                                             for h, b, a in zip(hole, board, npl)])
    got = [(t.winner_ix, [float(x) for x in t.stacks], t.queries, t.env_steps) for t in batch.tables]
    assert got == solo
    assert all(abs(sum(s) - 400) < 1e-9 for _, s, _, _ in got)  # chips are conserved
