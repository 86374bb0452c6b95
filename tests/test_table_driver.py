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
                        showdown=_showdown, calculate_equity=ep.get("calculate_equity", False))
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
    assert len(eps) >= 9 and any(ep.get("calculate_equity") for ep in eps)
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


# ---------------------------------------------------------------------------------------------------------
# The native driver (mcq_tables_* in include/mcq.h, csrc/mcq_tables.cpp) against the Python driver above, which
# the reference's own episodes pin: same per-table generator for dealing and random seats, same deterministic
# stand-in for the equity, so every table must issue the same query at every lock-step and end with the same
# stacks.  begin()/resume() need no GPU.
M32 = 0xFFFFFFFF


def _philox(c, k):
    c, k = list(c), list(k)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & M32, (p0 >> 32) ^ c[3] ^ k[1], p0 & M32]
        k = [(k[0] + 0x9E3779B9) & M32, (k[1] + 0xBB67AE85) & M32]
    return c


class _Xoshiro:
    def __init__(self, seed, table):
        self.s = _philox([table, 0, 0, 0x54424C31], [seed & M32, seed >> 32])

    def next(self):
        s = self.s
        rotl = lambda x, k: ((x << k) | (x >> (32 - k))) & M32  # noqa: E731
        result = (rotl((s[0] + s[3]) & M32, 7) + s[0]) & M32
        t = (s[1] << 9) & M32
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 11)
        return result

    def randint(self, n):
        return (self.next() * n) >> 32

    def integers(self, lo, hi):
        return lo + self.randint(hi - lo)


def _fake_equity(hole, board, alive):
    x = (int(hole[0]) * 131 + int(hole[1]) * 31 + sum(int(b) for b in board) * 7 + alive * 13 + len(board) * 3) % 101
    return x / 100.0


def _host_showdown(hands):
    keys = H.eval7(np.array(hands, np.uint8))
    return int(np.argmax(keys))          # first of equals


@pytest.mark.parametrize("seats,stacks,steps,calc", [
    ([("equity", .3, .5), ("equity", .45, .6), ("random",), ("equity", .2, .75), ("random",), ("equity", .5, .9)], 100, 1500, False),
    ([("equity", .5, -.5), ("equity", .8, -.8), ("equity", .7, -.7), ("equity", .2, -.3), ("random",), ("random",)], 100, 600, False),
    ([("equity", .35, .55), ("random",)], 60, 800, False),
    ([("random",)] * 3, 40, 800, False),
    ([("equity", .25, .65)] * 9 + [("random",)], 250.5, 800, False),
    # HoldemTable(calculate_equity=True): three more queries per observation (gym_env/env.py:248-256)
    ([("equity", .3, .5), ("equity", .45, .6), ("random",), ("equity", .2, .75)], 100, 1600, True),
])
def test_native_tables_follow_the_python_driver(seats, stacks, steps, calc):
    from neuron_poker_amd import _lib
    T, seed = 12, 0xC0FFEE12345
    nat = _lib.Tables(None, T, seats, runs=1000, initial_stacks=stacks, seed=seed, calculate_equity=calc)
    sims, gens, pend, episodes = [], [], [], [0] * T
    for t in range(T):
        rng = _Xoshiro(seed, t)
        pol = [td.equity_policy(s[1], s[2]) if s[0] == "equity" else td.random_policy(rng) for s in seats]
        sim = td.TableSim(pol, initial_stacks=stacks, showdown=_host_showdown, randint=rng.randint, calculate_equity=calc)
        sims.append(sim)
        gens.append(sim.episode())
        pend.append(next(gens[-1]))
    for step in range(steps):
        q = nat.begin()
        eq = np.zeros(T)
        for t in range(T):
            hole, board, alive = pend[t]
            assert list(q["hole"][t]) == hole, (step, t)
            assert q["n_board"][t] == len(board) and list(q["board"][t][:len(board)]) == board, (step, t)
            assert q["n_players"][t] == alive and q["runs"][t] == 1000, (step, t)
            eq[t] = _fake_equity(hole, board, alive)
        nat.resume(eq)
        for t in range(T):
            try:
                pend[t] = gens[t].send(eq[t])
            except StopIteration:
                episodes[t] += 1
                gens[t] = sims[t].episode()
                pend[t] = next(gens[t])
    tot_steps = 0
    for t in range(T):
        st = nat.state(t)
        assert st["episodes"] == episodes[t]
        assert list(st["stacks"]) == [float(s) for s in sims[t].stacks], t
        tot_steps += st["env_steps"]
    assert nat.stats()["queries"] == T * steps
    assert nat.stats()["env_steps"] == sum(s.env_steps for s in sims) == tot_steps
    assert tot_steps > 0 and sum(episodes) > 0


def test_native_tables_reject_bad_configuration():
    from neuron_poker_amd import _lib
    with pytest.raises(ValueError):
        _lib.Tables(None, 4, [("random",)])
    with pytest.raises(ValueError):
        _lib.Tables(None, 0, [("random",)] * 2)
    with pytest.raises(ValueError):
        _lib.Tables(None, 4, [("random",)] * 2, runs=0)


def test_native_tables_do_not_depend_on_the_thread_count():
    from neuron_poker_amd import _lib
    seats = [("equity", .3, .5), ("equity", .45, .6), ("random",), ("equity", .2, .75), ("random",), ("equity", .5, .9)]
    T = 3000
    a = _lib.Tables(None, T, seats, seed=9, threads=1)
    b = _lib.Tables(None, T, seats, seed=9, threads=5)
    for step in range(120):
        qa, qb = a.begin(), b.begin()
        assert np.array_equal(qa, qb), step
        assert np.array_equal(a.begin(), qa)          # asking again returns the same pending queries, uncounted
        eq = ((qa["hole"].astype(np.int64) * [131, 31]).sum(1) + qa["board"].astype(np.int64).sum(1) * 7
              + qa["n_players"] * 13) % 101 / 100.0
        a.resume(eq)
        b.resume(eq)
    assert a.stats() == b.stats() and a.stats()["queries"] == T * 120
    for t in range(0, T, 97):
        sa, sb = a.state(t), b.state(t)
        assert np.array_equal(sa.pop("stacks"), sb.pop("stacks")) and sa == sb


@pytest.mark.skipif(os.environ.get("MCQ_SAN_STUB") != "1", reason="needs the stand-in equity of tests/sanitize_cpu.sh")
@pytest.mark.parametrize("T,threads,overlap", [(200, 1, True), (200, 1, False), (4100, 3, True), (63, 0, True)])
def test_native_run_loop_with_stand_in_equity(T, threads, overlap):
    """mcq_tables_run itself (two halves on two threads, thread pool) without a GPU: tests/sanitize_cpu.sh links the
    driver against a stand-in mcq_eval_batch that returns _fake_equity; run(k) must leave the tables where k rounds
    of begin() -> _fake_equity -> resume() leave them."""
    from neuron_poker_amd import _lib

    class FakeEngine:
        _ctx = 1

    seats = [("equity", .3, .5), ("equity", .45, .6), ("random",), ("equity", .2, .75), ("random",), ("equity", .5, .9)]
    a = _lib.Tables(FakeEngine(), T, seats, seed=21, threads=threads, overlap=overlap)
    b = _lib.Tables(None, T, seats, seed=21, threads=1)
    for chunk in (37, 5, 1, 60):
        st = a.run(chunk)
        for _ in range(chunk):
            q = b.begin()
            eq = ((q["hole"].astype(np.int64) * [131, 31]).sum(1) + q["n_players"] * 13 + q["n_board"] * 3
                  + (q["board"].astype(np.int64) * (np.arange(5)[None, :] < q["n_board"][:, None])).sum(1) * 7) % 101 / 100.0
            b.resume(eq)
        assert st == b.stats()
    for t in range(0, T, 41):
        sa, sb = a.state(t), b.state(t)
        assert np.array_equal(sa.pop("stacks"), sb.pop("stacks")) and sa == sb
    assert np.array_equal(a.begin(), b.begin())


@pytest.mark.skipif(os.environ.get("MCQ_SAN_STUB") != "1", reason="needs the stand-in equity of tests/sanitize_cpu.sh")
@pytest.mark.parametrize("overlap", [True, False])
def test_native_run_loop_failing_half_way_marks_the_driver_failed(overlap, monkeypatch):
    """A device error in the middle of mcq_tables_run (the stand-in's 5th batch) comes back as an error code -- no
    exception crosses the C ABI, the helper thread is joined -- and when the two halves of the tables are out of step
    afterwards the driver refuses every later call instead of reusing query ids on desynchronised tables."""
    from neuron_poker_amd import _lib

    class FakeEngine:
        _ctx = 1

    seats = [("equity", .3, .5), ("random",), ("equity", .2, .75)]
    t = _lib.Tables(FakeEngine(), 200, seats, seed=3, threads=1, overlap=overlap)
    t.run(3)
    monkeypatch.setenv("MCQ_STUB_FAIL_AFTER", "5")
    with pytest.raises(_lib.McqError, match="stand-in device failure"):
        t.run(8)
    monkeypatch.delenv("MCQ_STUB_FAIL_AFTER")
    if overlap:   # the halves are out of step: refused for good
        with pytest.raises(ValueError, match="failed in an earlier call"):
            t.run(1)
        with pytest.raises(ValueError, match="failed in an earlier call"):
            t.resume(np.zeros(200))
    else:         # one batch per step: the failed step's queries are still pending, the call can be repeated
        t.run(1)
    t.close()
