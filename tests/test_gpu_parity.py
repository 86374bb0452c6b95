"""GPU parity tests: the HIP path, called through the C ABI, against the oracle and the golden fixtures.

Bar (integer work): bit-exact.  The only floating-point number on the path is wins / runs.
  * parity mode (MT19937 replay)  == the reference's own tallies under np.random.seed (tests/golden/tallies.json)
  * production mode (MCQ-CTR v5)  == the oracle's CTR mode, query by query
  * evaluator                     == the reference's _calc_score ordering on the golden hands / showdowns
  * full-size configs             -> size-independent properties (sum of types, shard invariance, exact
                                     expectation from exhaustive enumeration)
"""
import json
import os

import numpy as np
import pytest

import neuron_poker_amd as npa
from neuron_poker_amd import montecarlo_hip as mh
from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def eng():
    e = npa.Engine(0, kernel_times=True)
    yield e
    e.close()


def jload(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def mkq(hero, board, n, runs):
    b = [npa.card_id(c) for c in board] + [255] * (5 - len(board))
    return npa.pack_queries([[npa.card_id(hero[0]), npa.card_id(hero[1])]], [b], n, runs)


def u64(res):
    return res.view(np.uint64).reshape(-1, 13)


# ------------------------------------------------------------------------------------------ evaluator
def test_keys_reproduce_reference_categories_and_order(eng):
    z = np.load(os.path.join(G, "evaluator_hands.npz"))
    cards, cat, nr, ranks = z["cards"], z["category"], z["n_ranks"], z["card_ranks"]
    _, _, keys = eng.showdown(cards.reshape(-1, 1, 7), want_keys=True)
    keys = keys.reshape(-1)
    assert np.array_equal(npa.key_type(keys), cat)
    tup = [(int(cat[i]), tuple(int(x) for x in ranks[i, :nr[i]])) for i in range(len(cards))]
    order = np.argsort(keys, kind="stable")
    for a, b in zip(order[:-1], order[1:]):
        assert (tup[a] == tup[b]) if keys[a] == keys[b] else (tup[a] < tup[b]), (cards[a], cards[b])


def test_showdown_winners_match_reference(eng):
    z = np.load(os.path.join(G, "showdowns.npz"))
    hands, n, win, wt = z["hands"], z["n_players"], z["winner"], z["winner_type"]
    for p in range(2, 11):
        sel = np.nonzero(n == p)[0]
        if len(sel) == 0:
            continue
        w, t = eng.showdown(hands[sel, :p])
        assert np.array_equal(w, win[sel]) and np.array_equal(t, wt[sel]), p


def test_reference_evaluator_cases(eng):
    for c in jload("evaluator_cases.json"):
        ids = [[npa.card_id(x) for x in h] for h in c["hands"]]
        if any(len(set(h)) != 7 for h in ids):
            with pytest.raises(ValueError):  # duplicate cards (tests/test_evaluator.py:27,63): outside the domain
                eng.showdown(np.array([ids], np.uint8))
            continue
        w, t = eng.showdown(np.array([ids], np.uint8))
        assert w[0] == c["winner"] and npa.TYPES[t[0]] == c["winner_type"], c


# ------------------------------------------------------------------------------------------ parity mode
def test_replay_reproduces_reference_tallies_bit_exact(eng):
    rows = jload("tallies.json")
    for t in rows:
        r = eng.eval_batch(mkq(t["hero"], t["board"], t["n_players"], t["runs"]), seed=t["seed"],
                           mode=npa.MODE_REPLAY_MT19937)[0]
        assert int(r["runs"]) == t["runs"]
        assert int(r["win"] + r["tie"]) == t["wins"], t
        assert int(r["passes"]) == t["passes"], t
        assert [int(x) for x in r["by_type"]] == t["by_type"], t


def test_replay_batch_equals_oracle_mt_per_query(eng):
    g = np.random.default_rng(11)
    B = 96
    hole, board, npl, runs = [], [], [], []
    for i in range(B):
        nb = [0, 3, 4, 5][i % 4]
        cards = g.choice(52, 2 + nb, replace=False)
        hole.append(cards[:2])
        board.append(list(cards[2:]) + [255] * (5 - nb))
        npl.append(1 + i % 10)
        runs.append(int(g.integers(1, 3000)))
    q = npa.pack_queries(hole, board, npl, runs)
    got = u64(eng.eval_batch(q, seed=1234, first_query_id=7, mode=npa.MODE_REPLAY_MT19937))
    exp = O.run_batch(O.MODE_MT, q.view(np.uint8).reshape(-1, 16), 1234, 7, threads=4)
    assert np.array_equal(got, exp)


def test_replay_stream_walk_runs_on_the_device(eng, monkeypatch):
    """The parity mode's MT19937 walk is a HIP kernel (mcq_mt_parse_kernel: one wave per query, mcq_mt.hpp): more
    queries than resident waves, ragged run counts around the 64-iteration flushes and the 624-word regenerations,
    queries that draw nothing (one player, five table cards), seeds wrapping 2^32, and the draw buffer cut into
    several launches -- all equal to the oracle's literal walk (MT19937 + numpy randint + the reference's loops)."""
    g = np.random.default_rng(77)
    B = 9000
    hole = np.zeros((B, 2), np.uint8)
    board = np.full((B, 5), 255, np.uint8)
    npl = np.zeros(B, np.uint8)
    for i in range(B):
        nb = [0, 3, 4, 5][i % 4]
        c = g.permutation(52)[:2 + nb]
        hole[i] = c[:2]
        board[i, :nb] = c[2:]
        npl[i] = 1 + i % 10
    runs = g.choice([1, 2, 31, 63, 64, 65, 127, 128, 129, 200, 777], B)
    runs[::1000] = 30000
    q = npa.pack_queries(hole, board, npl, runs)
    first = 2 ** 32 - 4000          # (seed + query id) wraps around 2^32 inside the batch
    exp = O.run_batch(O.MODE_MT, q.view(np.uint8).reshape(-1, 16), 99, first, threads=8)
    got = u64(eng.eval_batch(q, seed=99, first_query_id=first, mode=npa.MODE_REPLAY_MT19937))
    assert np.array_equal(got, exp)
    assert eng.last_kernel_ms > 0
    monkeypatch.setenv("MCQ_REPLAY_DEVICE_BYTES", str(1 << 20))   # many launches: 1 MB of draws each
    e2 = npa.Engine(0)
    try:
        assert np.array_equal(u64(e2.eval_batch(q, seed=99, first_query_id=first, mode=npa.MODE_REPLAY_MT19937)), exp)
    finally:
        e2.close()


def test_replay_few_long_queries_by_state_blocks(eng, monkeypatch):
    """Few long queries in parity mode are parsed with their 624-word state blocks side by side (mcq_mt_blocks.hpp:
    generate / scan every entry state / stitch / parse each block from its true entry): every number of players (zone
    31 with eight and nine opponents), boards of every length, queries that draw nothing and a short one among them;
    equal to the oracle's literal walk, to the serial walk on the device (MCQ_MT_BLOCKS=0), and seeds wrap 2^32."""
    g = np.random.default_rng(78)
    B = 24
    hole = np.zeros((B, 2), np.uint8)
    board = np.full((B, 5), 255, np.uint8)
    npl = np.zeros(B, np.uint8)
    for i in range(B):
        nb = [0, 3, 4, 5][i % 4]
        c = g.permutation(52)[:2 + nb]
        hole[i] = c[:2]
        board[i, :nb] = c[2:]
        npl[i] = 1 + i % 10
    runs = g.choice([2000, 5000, 12000, 20000], B)
    runs[3] = 1
    runs[7] = 0
    c = g.permutation(52)[:7]   # a query that draws nothing: one player, the table complete
    hole[13], board[13], npl[13] = c[:2], c[2:], 1
    q = npa.pack_queries(hole, board, npl, runs)
    first = 2 ** 32 - 10
    exp = O.run_batch(O.MODE_MT, q.view(np.uint8).reshape(-1, 16), 7, first, threads=8)
    got = u64(eng.eval_batch(q, seed=7, first_query_id=first, mode=npa.MODE_REPLAY_MT19937))
    assert np.array_equal(got, exp)
    one = u64(eng.eval_batch(q[9:10], seed=7, first_query_id=first + 9, mode=npa.MODE_REPLAY_MT19937))   # ten players, alone
    assert np.array_equal(one, exp[9:10])
    monkeypatch.setenv("MCQ_MT_BLOCKS", "0")
    e2 = npa.Engine(0)
    try:
        assert np.array_equal(u64(e2.eval_batch(q, seed=7, first_query_id=first, mode=npa.MODE_REPLAY_MT19937)), exp)
    finally:
        e2.close()
    # too few blocks (the estimate short by three): the streams that run past them are noticed and the call falls
    # back to the serial walk; exactly enough blocks (the estimate without its margin mostly is): still right
    monkeypatch.setenv("MCQ_MT_BLOCKS", "1")
    for margin in ("-3", "0", "1"):
        monkeypatch.setenv("MCQ_MT_BLOCKS_MARGIN", margin)
        e3 = npa.Engine(0)
        try:
            assert np.array_equal(u64(e3.eval_batch(q, seed=7, first_query_id=first, mode=npa.MODE_REPLAY_MT19937)), exp), margin
        finally:
            e3.close()


def test_replay_state_blocks_in_segments_by_jump_ahead(eng, monkeypatch):
    """Round 4: a query's state blocks are generated in segments of 128 side by side, segment g's start state by
    jump-ahead (csrc/mcq_mt_jump_table.inc: a fixed XOR combination of the round's first 33 blocks).  One segment (no
    jump), a few, all 32 of a round, and more than one round (> 4096 blocks: the next round starts from the last block
    of the one before); equal to the oracle's literal walk and to the one-work-group generator (MCQ_MT_JUMP=0)."""
    g = np.random.default_rng(404)
    cases = [(2, 3000, 0), (6, 100000, 0), (10, 125000, 0), (3, 260000, 3), (2, 100000, 0), (9, 40000, 4)]
    hole = np.zeros((len(cases), 2), np.uint8)
    board = np.full((len(cases), 5), 255, np.uint8)
    for i, (_, _, nb) in enumerate(cases):
        c = g.permutation(52)[:2 + nb]
        hole[i] = c[:2]
        board[i, :nb] = c[2:]
    q = npa.pack_queries(hole, board, [c[0] for c in cases], [c[1] for c in cases])
    first = 2 ** 32 - 3
    exp = O.run_batch(O.MODE_MT, q.view(np.uint8).reshape(-1, 16), 11, first, threads=8)
    got = u64(eng.eval_batch(q, seed=11, first_query_id=first, mode=npa.MODE_REPLAY_MT19937))
    assert np.array_equal(got, exp)
    for i in (1, 2):   # alone: the grid has one query
        one = u64(eng.eval_batch(q[i:i + 1], seed=11, first_query_id=first + i, mode=npa.MODE_REPLAY_MT19937))
        assert np.array_equal(one, exp[i:i + 1])
    for jump in ("0", "2"):   # never / whenever a query has more than one segment (the default goes by a cost estimate)
        monkeypatch.setenv("MCQ_MT_JUMP", jump)
        e2 = npa.Engine(0)
        try:
            assert np.array_equal(u64(e2.eval_batch(q, seed=11, first_query_id=first, mode=npa.MODE_REPLAY_MT19937)), exp)
            if jump == "2":   # many queries of a few segments each: the estimate alone would not jump
                s = np.arange(40) % len(cases)
                qs = npa.pack_queries(hole[s], board[s], [cases[k][0] for k in s], np.full(40, 9000))
                ex2 = O.run_batch(O.MODE_MT, qs.view(np.uint8).reshape(-1, 16), 5, 77, threads=8)
                assert np.array_equal(u64(e2.eval_batch(qs, seed=5, first_query_id=77, mode=npa.MODE_REPLAY_MT19937)), ex2)
        finally:
            e2.close()


# ------------------------------------------------------------------------------------------ production mode
def test_philox_equals_oracle_ctr_bit_exact(eng):
    g = np.random.default_rng(5)
    B = 200
    hole, board, npl, runs = [], [], [], []
    for i in range(B):
        nb = [0, 3, 4, 5][i % 4]
        cards = g.choice(52, 2 + nb, replace=False)
        hole.append(cards[:2])
        board.append(list(cards[2:]) + [255] * (5 - nb))
        npl.append(1 + (i * 7) % 10)
        runs.append(int(g.choice([1, 15, 16, 17, 1000, 1023, 1024, 1025, 4097])))
    q = npa.pack_queries(hole, board, npl, runs)
    for seed, first in [(0, 0), (0xDEADBEEF12345678, 2 ** 33 + 5)]:
        got = u64(eng.eval_batch(q, seed=seed, first_query_id=first, mode=npa.MODE_PHILOX))
        exp = O.run_batch(O.MODE_CTR, q.view(np.uint8).reshape(-1, 16), seed, first, threads=4)
        assert np.array_equal(got, exp)


def test_config2_single_query_100k(eng):
    """BASELINE configs[1]: AhKh, empty board, 1 opponent, 100k iterations."""
    q = mkq(["AH", "KH"], [], 2, 100000)
    for seed, wins, passes in [(0, 65807, 102091), (1, 65985, 102097)]:  # reference, SURVEY.md 8c F3
        r = eng.eval_batch(q, seed=seed, mode=npa.MODE_REPLAY_MT19937)[0]
        assert (int(r["win"] + r["tie"]), int(r["passes"])) == (wins, passes)
    got = u64(eng.eval_batch(q, seed=3, mode=npa.MODE_PHILOX))
    exp = O.run_batch(O.MODE_CTR, q.view(np.uint8).reshape(-1, 16), 3, 0)
    assert np.array_equal(got, exp)
    # production front end vs the reference's seeded equities: both estimate the same expectation
    eq = (got[0, 2] + got[0, 3]) / 1e5
    assert abs(eq - 0.65807) < 6e-3 and abs(eq - 0.65985) < 6e-3  # 4 sigma of the difference at 100k


def test_shard_invariance_and_device_entry(eng):
    """Splitting a batch (other ranks / other calls) never changes a per-query tally."""
    g = np.random.default_rng(4096)
    B = 512
    hole = np.array([g.choice(52, 2, replace=False) for _ in range(B)], np.uint8)
    q = npa.pack_queries(hole, np.full((B, 5), 255, np.uint8), 3, 5000)
    whole = u64(eng.eval_batch(q, seed=1, first_query_id=0))
    parts = [u64(eng.eval_batch(q[a:a + 128], seed=1, first_query_id=a)) for a in range(0, B, 128)]
    assert np.array_equal(whole, np.concatenate(parts))
    torch = pytest.importorskip("torch")
    dq = torch.from_numpy(q.view(np.uint8).reshape(B, 16).copy()).cuda()
    dres = torch.empty((B, 13), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    eng.eval_batch_device(dq.data_ptr(), B, 1, dres.data_ptr(), first_query_id=0,
                          stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dres.cpu().numpy().view(np.uint64), whole)


def test_config3_full_size_properties(eng):
    """BASELINE configs[2]: 4096 random preflop states, 2 opponents, 50k iterations each."""
    g = np.random.default_rng(4096)
    hole = np.array([g.choice(52, 2, replace=False) for _ in range(4096)], np.uint8)
    q = npa.pack_queries(hole, np.full((4096, 5), 255, np.uint8), 3, 50000)
    t = u64(eng.eval_batch(q, seed=1))
    assert (t[:, 0] == 50000).all()
    assert np.array_equal(t[:, 2] + t[:, 3], t[:, 4:].sum(1))          # reference: sum(types) == equity
    assert (t[:, 1] == 2 * 50000).all()  # production mode never re-draws: one attempt per opponent per iteration
    eq = (t[:, 2] + t[:, 3]) / 50000.0
    # 64 of the queries checked bit-for-bit against the oracle
    idx = np.arange(0, 4096, 64)
    exp = np.stack([O.run_batch(O.MODE_CTR, q[i:i + 1].view(np.uint8).reshape(-1, 16), 1, int(i))[0] for i in idx])
    assert np.array_equal(t[idx], exp)
    # suit isomorphism: states with the same ranks / suitedness have the same expectation (6 sigma)
    cls = {}
    for i in range(4096):
        a, b = sorted((int(hole[i, 0]) >> 2, int(hole[i, 1]) >> 2))
        cls.setdefault((a, b, (hole[i, 0] & 3) == (hole[i, 1] & 3)), []).append(eq[i])
    # the reference's dealing is order-biased, so classes agree only to ~1 point; this guards gross errors
    assert max(max(v) - min(v) for v in cls.values()) < 0.05


def test_philox_converges_to_exact_expectation(eng):
    """The production RNG front end follows the REFERENCE'S dealing law: at 4e8 iterations the estimate sits
    within 5 sigma (~1.2e-4) of the exact expectation obtained by exhaustive enumeration of that law."""
    for hero, board, n in [(["TC", "TH"], ["4D", "QD", "KC", "2S"], 2), (["3S", "QH"], ["2C", "5H", "7C"], 2),
                           (["JD", "JS"], ["8C", "TC", "JC", "5H", "QC"], 3)]:
        w, t, _ = O.exact(hero, board, n)
        p = w + t
        B, runs = 100, 4000000
        q = np.repeat(mkq(hero, board, n, runs), B)
        r = u64(eng.eval_batch(q, seed=99))
        total = float(r[:, 0].sum())
        est = float((r[:, 2] + r[:, 3]).sum()) / total
        est_tie = float(r[:, 3].sum()) / total
        assert abs(est - p) < 5 * (p * (1 - p) / total) ** 0.5 + 1e-9, (hero, est, p)
        assert abs(est_tie - t) < 5 * (max(t, 1e-9) / total) ** 0.5 + 1e-9, (hero, est_tie, t)
        assert abs(est - p) <= 1e-4


def test_statistical_expectations_of_reference_tests(eng):
    rows = jload("stat_expectations.json")
    hole = [[npa.card_id(c) for c in r["hero"]] for r in rows]
    board = [[npa.card_id(c) for c in r["board"]] + [255] * (5 - len(r["board"])) for r in rows]
    eq, _ = mh.get_equity_batch(hole, board, [r["n_players"] for r in rows], 200000, seed=7, engine=eng)
    for r, e in zip(rows, eq):
        assert abs(100 * e - r["expected_pct"]) < r["tol_pct"], (r, e)


# ------------------------------------------------------------------------------------------ edges and errors
def test_single_player_zero_board_and_small_runs(eng):
    r = eng.eval_batch(mkq(["7H", "2C"], [], 1, 3000), seed=9, mode=npa.MODE_REPLAY_MT19937)[0]
    assert (int(r["win"]), int(r["tie"]), int(r["passes"])) == (3000, 0, 0)
    for runs in (1, 2, 63, 64, 65):
        q = mkq(["AS", "AC"], ["2C", "2D", "2H"], 10, runs)
        got = u64(eng.eval_batch(q, seed=5))
        assert np.array_equal(got, O.run_batch(O.MODE_CTR, q.view(np.uint8).reshape(-1, 16), 5, 0))
    assert len(eng.eval_batch(npa.pack_queries(np.zeros((0, 2)), np.zeros((0, 5)), 2, 10), seed=0)) == 0


def test_invalid_queries_raise_value_error(eng):
    good = mkq(["AH", "KH"], [], 2, 100)
    for bad in [mkq(["AH", "KH"], ["AH", "2C", "3C"], 2, 100),   # hero card on the table
                mkq(["AH", "KH"], ["2C", "2C", "3C"], 2, 100),   # duplicate table card
                mkq(["AH", "KH"], [], 11, 100), mkq(["AH", "KH"], [], 0, 100)]:
        with pytest.raises(ValueError):
            eng.eval_batch(np.concatenate([good, bad]), seed=0)
    q = good.copy()
    q["hole"][0, 0] = 52
    with pytest.raises(ValueError):
        eng.eval_batch(q, seed=0)
    with pytest.raises(ValueError):
        eng.eval_batch(good, seed=0, mode=7)


# ------------------------------------------------------------------------------------------ drop-in surface
def test_dropin_module_matches_reference_call_surface(eng):
    mh.configure(mode="replay")
    try:
        mh.seed(0)
        sim = mh.MonteCarlo(eng)
        eq, types = sim.run_montecarlo([["AH", "KH"]], [], 2, 1, maxRuns=10000, timeout=0, ghost_cards="",
                                       opponent_range=1)
        # reference after np.random.seed(0): 6629 wins, 10213 passes (tests/golden/tallies.json row 0)
        row = jload("tallies.json")[0]
        assert (eq, sim.runs, sim.passes) == (row["wins"] / row["runs"], row["runs"], row["passes"])
        assert dict(types) == {npa.TYPES[i]: c / row["runs"] for i, c in enumerate(row["by_type"]) if c}
        assert abs(sum(sim.winnerCardTypeList.values()) - sim.equity) < 1e-4  # tests/test_montecarlo_python.py:32
        mh.seed(1)
        assert mh.get_equity({"AH", "KH"}, set(), np.int64(2), 10000) == 0.6529
    finally:
        mh.configure(mode="philox")
    mh.seed(42)
    a = mh.get_equity({"AS", "KS"}, set(), 2, 100000)
    mh.seed(42)
    assert mh.get_equity({"KS", "AS"}, set(), 2, 100000) == a
    assert abs(100 * a - 66.0) < 1.0
    with pytest.raises(ValueError):
        mh.get_equity({"AS", "KS"}, {"AS", "2C", "3C"}, 2, 100)
    with pytest.raises(ValueError):
        mh.get_equity({"AS", "Kx"}, set(), 2, 100)
    with pytest.raises(ValueError):   # eleven known hands
        mh.MonteCarlo(eng).run_montecarlo([["AS", "KS"]] + [["2C", "2D"]] * 10, [], 10, 1, maxRuns=10, timeout=0, ghost_cards="")


def test_numpy_stream_coupling_on_gpu(eng):
    """replay mode drawing from and advancing numpy's GLOBAL state like the reference's calls."""
    mh.configure(mode="replay", couple_numpy=True)
    try:
        for s in jload("sequence.json"):
            np.random.seed(s["seed"])
            for c in s["calls"]:
                sim = mh.MonteCarlo(eng)
                eq, _ = sim.run_montecarlo([c["hero"]], c["board"], c["n_players"], 1, maxRuns=c["runs"], timeout=0,
                                           ghost_cards="")
                assert (round(eq * c["runs"]), sim.passes) == (c["wins"], c["passes"]), c
                assert [int(x) for x in sim.result["by_type"]] == c["by_type"]
                assert int(np.random.randint(0, 52)) == c["randint52_after"]
            assert [int(x) for x in np.random.randint(0, 2 ** 32, size=4, dtype=np.uint32)] == s["next_words"]
    finally:
        mh.configure(mode="philox", couple_numpy=False)


def test_uniform_dealing_law(eng):
    """opt-in unbiased law == oracle bit for bit, converges to the exact uniform expectation, which
    is the one the reference's C++ tests expect (Test.cpp:176-217: 40.2 / 51.8 / 67.7 % within 1 %)."""
    eng.set_dealing_law("uniform")
    try:
        g = np.random.default_rng(8)
        hole, board, npl = [], [], []
        for i in range(64):
            nb = [0, 3, 4, 5][i % 4]
            cards = g.choice(52, 2 + nb, replace=False)
            hole.append(cards[:2]); board.append(list(cards[2:]) + [255] * (5 - nb)); npl.append(1 + i % 10)
        q = npa.pack_queries(hole, board, npl, 2500)
        got = u64(eng.eval_batch(q, seed=4, first_query_id=9))
        exp = O.run_batch(O.MODE_CTR_UNIFORM, q.view(np.uint8).reshape(-1, 16), 4, 9, threads=4)
        assert np.array_equal(got, exp)
        hero, brd = ["3H", "3S"], ["8S", "4S", "QH", "8C", "4H"]
        w, t, _ = O.exact(hero, brd, 2, uniform=True)
        r = u64(eng.eval_batch(np.repeat(mkq(hero, brd, 2, 4000000), 50), seed=1))
        est = float((r[:, 2] + r[:, 3]).sum()) / float(r[:, 0].sum())
        assert abs(est - (w + t)) < 1e-4 and abs(100 * est - 40.2) < 1.0
        # Test.cpp montecarlo3/4: AS KS preflop, 3 and 2 players: 51.8 % and 67.7 % within 1 %
        eq, _ = mh.get_equity_batch([[51, 47], [51, 47]], [[255] * 5] * 2, [3, 2], 4000000, seed=2, engine=eng)
        assert abs(100 * eq[0] - 51.8) < 1.0 and abs(100 * eq[1] - 67.7) < 1.0
    finally:
        eng.set_dealing_law("reference")


def test_ranges_ghost_cards_known_hands(eng):
    """SURVEY 8f-2 through the drop-in surface: replay mode == the reference's seeded results
    (tests/golden/ext_tallies.json, incl. the inputs of tests/test_montecarlo_python.py:215-232), production
    mode == the oracle's CTR mode."""
    rows = jload("ext_tallies.json")
    with open(os.path.join(os.path.dirname(G), "..", "neuron_poker_amd", "preflop_classes.json")) as f:
        order = json.load(f)
    for t in rows:
        pl = [p if O._is_cards(p) else set(p) for p in t["players"]]
        rng = set(t["opponent_range"]) if isinstance(t["opponent_range"], list) else t["opponent_range"]
        sim = mh.MonteCarlo(eng)
        eq, _ = sim.run_montecarlo(pl, t["board"], t["n_players"], 1, maxRuns=t["runs"], timeout=0,
                                   ghost_cards=t["ghost"] or "", opponent_range=rng, mode="replay", seed=t["seed"])
        assert (round(eq * t["runs"]), sim.passes) == (t["wins"], t["passes"]), t
        assert [int(x) for x in sim.result["by_type"]] == t["by_type"], t
        # production mode against the oracle
        sim.run_montecarlo(pl, t["board"], t["n_players"], 1, maxRuns=1500, timeout=0, ghost_cards=t["ghost"] or "",
                           opponent_range=rng, mode="philox", seed=31)
        if isinstance(rng, set):
            opp = sorted(rng)
        else:
            take = int(169 * rng)
            opp = None if take == 0 or take >= 169 else order[-take:]
        exp = O.run_ex(O.MODE_CTR, t["players"][0], t["board"], t["n_players"], 1500, 31, known=t["players"][1:],
                       ghost=t["ghost"] or None, opp_range=opp)["tallies"]
        assert np.array_equal(np.frombuffer(sim.result.tobytes(), np.uint64), exp), t
    # tests/test_montecarlo_python.py:215-232: 12.8 % and 77.8 % within 3 points
    board = ["3D", "9H", "AS", "7S", "QH"]
    a = mh.MonteCarlo(eng)
    a.run_montecarlo([["KS", "KC"]], board, 3, 1, maxRuns=100000, timeout=0, ghost_cards="", opponent_range=0.25, seed=1)
    b = mh.MonteCarlo(eng)
    b.run_montecarlo([{"AKO", "AA"}], board, 3, 1, maxRuns=100000, timeout=0, ghost_cards="", opponent_range=0.25, seed=2)
    assert abs(100 * a.equity - 12.8) < 3 and abs(100 * b.equity - 77.8) < 3
    assert abs(sum(b.winnerCardTypeList.values()) - b.equity) < 1e-4
    # an extension record that restricts nothing gives the plain path's tallies (unrestricted opponents are dealt by
    # index exactly as there); with ghost cards / known hands the unrestricted opponents still cost one word each
    g = np.random.default_rng(8)
    cards = np.array([g.permutation(52)[:7] for _ in range(64)], np.uint8)
    q = npa.pack_queries(cards[:, :2], np.full((64, 5), 255, np.uint8), 1 + np.arange(64) % 10, 1 + 37 * np.arange(64))
    assert np.array_equal(u64(eng.eval_batch_ext(q, npa.pack_query_ext(64), 9, first_query_id=3)), u64(eng.eval_batch(q, 9, first_query_id=3)))
    # a range that cannot be dealt (both remaining aces are dead) raises instead of hanging
    with pytest.raises(ValueError):
        mh.MonteCarlo(eng).run_montecarlo([["AS", "AH"]], ["AD", "AC", "2C"], 2, 1, maxRuns=100, timeout=0,
                                          ghost_cards="", opponent_range={"AA"}, seed=3)


def test_extended_queries_one_launch_path(eng):
    """A few extended queries per call (what a decision of the reference's agents asks: one ranged query, agent_*.py ->
    get_equity) take ONE launch (mcq_eval_ext_small_kernel: lists laid out in the block's LDS, row into pinned
    memory); more than eight take the general path (prep, lists, evaluation kernels).  Same streams -- two iterations
    each for a query of at most 8192 iterations that draws from a list, MCQ-CTR v5x -- so the same tallies, and both
    equal the oracle's."""
    from neuron_poker_amd import _lib
    ranks = "23456789TJQKA"
    classes = [a + a for a in ranks] + [ranks[i] + ranks[j] + t for i in range(13) for j in range(i) for t in "SO"]  # 169
    g = np.random.default_rng(77)
    n = 14
    q = np.zeros(n, npa.QUERY_DTYPE)
    e = np.zeros(n, npa.QUERY_EXT_DTYPE)
    spec = []
    for i in range(n):
        nb = int(g.choice([0, 3, 4, 5]))
        c = [int(x) for x in g.permutation(52)[:14 + nb]]
        pick = lambda lo, hi: sorted(g.choice(classes, size=int(g.integers(lo, hi)), replace=False))
        hero_range = pick(25, 90) if i % 5 == 4 else None
        opp = pick(20, 160) if i % 7 != 6 else None
        ghost = c[2:4] if i % 3 == 0 else None
        known = [pick(30, 100) if (i + k) % 3 == 0 else c[4 + 2 * k:6 + 2 * k] for k in range(i % 4)]
        npl = int(g.integers(max(2, 1 + len(known)), 8))
        runs = [1, 127, 128, 129, 1000, 2047, 4500, 8192, 8193, 0, 1000, 300, 64, 6000][i]
        q[i] = _lib.pack_queries([[0, 1] if hero_range else c[:2]], [c[14:] + [255] * (5 - nb)], npl, runs)[0]
        if hero_range:
            q["hole"][i] = 0
        e[i] = _lib.pack_query_ext(1, ghost=ghost, known=[h if isinstance(h[0], int) else _lib.range_bits(h) for h in known],
                                   hero_range=_lib.range_bits(hero_range) if hero_range else None,
                                   opp_range=_lib.range_bits(opp) if opp else None)[0]
        spec.append((hero_range if hero_range else c[:2], c[14:], npl, runs, known, ghost, opp))
    whole = u64(eng.eval_batch_ext(q, e, 5, first_query_id=100))  # 14 queries: the general path
    for lo, hi in ((0, 8), (8, 14), (3, 4), (8, 9)):  # at most eight: one launch
        assert np.array_equal(u64(eng.eval_batch_ext(q[lo:hi], e[lo:hi], 5, first_query_id=100 + lo)), whole[lo:hi]), (lo, hi)
    for i, (hero, board, npl, runs, known, ghost, opp) in enumerate(spec):
        want = O.run_ex(O.MODE_CTR, hero, board, npl, runs, 5, qid=100 + i, known=known, ghost=ghost, opp_range=opp)["tallies"]
        assert np.array_equal(whole[i], want), (i, spec[i])
    # a range that cannot be dealt, and an invalid record, on the one-launch path
    with pytest.raises(ValueError):
        mh.MonteCarlo(eng).run_montecarlo([["AS", "AH"]], ["AD", "AC", "2C"], 2, 1, maxRuns=100, timeout=0,
                                          ghost_cards="", opponent_range={"AA"}, seed=3)
    bad = e[:1].copy()
    bad["ghost"] = q["hole"][0]  # the ghost cards are hero's
    with pytest.raises(ValueError):
        eng.eval_batch_ext(q[:1], bad, 1)


def test_table_driver_reproduces_reference_episodes_on_gpu(eng):
    """seeded episodes of the reference's own table (tests/golden/env_traces.json) replayed by
    neuron_poker_amd/table_driver.py with the GPU doing every equity query (parity mode on numpy's global
    stream) and every showdown."""
    from neuron_poker_amd import table_driver as td

    def showdown(hands):
        w, _ = eng.showdown(np.array([hands], np.uint8))
        return int(w[0])

    for ep in jload("env_traces.json")[:6]:
        table = td.TableSim([td.equity_policy(c, b) for c, b in ep["policies"]], initial_stacks=ep["stacks"],
                            showdown=showdown)
        table.log = []
        np.random.seed(ep["seed"])
        g = table.episode()
        try:
            hole, board, alive = next(g)
            while True:
                q = npa.pack_queries([hole], [list(board) + [255] * (5 - len(board))], alive, 1000)
                r = eng.eval_batch_numpy_stream(q)[0]
                hole, board, alive = g.send(float(int(r["win"] + r["tie"]) / 1000))
        except StopIteration:
            pass
        assert table.log == ep["events"], ep["seed"]
        assert table.winner_ix == ep["winner"] and [float(s) for s in table.stacks] == ep["final_stacks"]
        assert [int(x) for x in np.random.randint(0, 2 ** 32, size=2, dtype=np.uint32)] == ep["np_next_words"]


def test_config4_full_size_sharded_like_8_gpus(eng):
    """BASELINE configs[3]: 65 536 states, mixed flop / turn boards, 5 opponents, 20k iterations, sharded over
    8 GPUs with one all-reduce of the tallies.  One GPU is reachable here, so the 8 shards run one after the
    other with their global query ids (exactly what 8 ranks do); the zero-initialised tally matrix summed over
    the shards (the all-reduce) must equal the unsharded result bit for bit."""
    from neuron_poker_amd import sharding
    g = np.random.default_rng(65536)  # SURVEY 8d generator
    B = 65536
    hole = np.zeros((B, 2), np.uint8)
    board = np.full((B, 5), 255, np.uint8)
    for i in range(B):
        b = 3 if i % 2 == 0 else 4
        c = g.choice(52, 2 + b, replace=False)
        hole[i] = c[:2]
        board[i, :b] = c[2:]
    q = npa.pack_queries(hole, board, 6, 20000)
    whole = u64(eng.eval_batch(q, seed=4, first_query_id=0))
    total = np.zeros((B, 13), np.uint64)
    for r in range(8):
        lo, hi = sharding.shard_bounds(B, r, 8)
        part = np.zeros((B, 13), np.uint64)
        part[lo:hi] = u64(eng.eval_batch(q[lo:hi], seed=4, first_query_id=lo))
        total += part  # the all-reduce(SUM)
    assert np.array_equal(total, whole)
    # the same 8-way partition through the C ABI's multi-GPU entry: 8 shards (here all on this device), their tally
    # matrices joined by the entry's ONE ncclAllReduce on a communicator from ncclCommInitAll
    me = npa.MultiEngine([0] * 8)
    try:
        got = u64(me.eval_batch(q, seed=4, first_query_id=0))
        info = me.info
    finally:
        me.close()
    assert info["shards"] == 8 and info["devices"] == 1 and info["last_partition"] == "queries" and info["rccl_version"] > 0
    assert np.array_equal(got, whole)
    assert (whole[:, 0] == 20000).all() and (whole[:, 1] == 5 * 20000).all()
    assert np.array_equal(whole[:, 2] + whole[:, 3], whole[:, 4:].sum(1))
    idx = np.arange(17, B, 4099)  # a few rows bit for bit against the oracle
    exp = np.stack([O.run_batch(O.MODE_CTR, q[i:i + 1].view(np.uint8).reshape(-1, 16), 4, int(i))[0] for i in idx])
    assert np.array_equal(whole[idx], exp)


def test_gate_b_production_rng_vs_reference_stream_at_1e9_iterations(eng):
    """SURVEY 8d config 2, gate B: AhKh heads-up preflop (the headline single query) -- the production RNG front end
    against the reference's own MT19937 law, BOTH at 1e9 iterations: |delta equity| <= 1e-4 (about 4.7 sigma of the
    difference).  The MT side runs in parity mode (1000 seeds x 1e6 runs), which the other tests prove bit-exact
    to tools/montecarlo_python.py, so this compares against the reference's distribution itself; preflop trees are
    too large for the exact enumeration used in test_philox_converges_to_exact_expectation."""
    q = np.repeat(mkq(["AH", "KH"], [], 2, 1000000), 1000)
    mt = u64(eng.eval_batch(q, seed=1000, mode=npa.MODE_REPLAY_MT19937))
    ph = u64(eng.eval_batch(q, seed=77, mode=npa.MODE_PHILOX))
    n = float(mt[:, 0].sum())
    assert n == 1e9 and float(ph[:, 0].sum()) == 1e9
    e_mt = float((mt[:, 2] + mt[:, 3]).sum()) / n
    e_ph = float((ph[:, 2] + ph[:, 3]).sum()) / n
    assert abs(e_mt - e_ph) <= 1e-4, (e_mt, e_ph)
    assert abs(e_mt - 0.659) < 2e-3  # the biased law's value, not the uniform 0.680
    # per hand type as well: shares of hero's winning types agree to 1e-4
    assert np.all(np.abs(mt[:, 4:].sum(0) / n - ph[:, 4:].sum(0) / n) <= 1e-4)


def b_run(b, eng, seed, T, done_steps, more):
    calls = T * done_steps
    for _ in range(more):
        q = b.begin()
        r = eng.eval_batch(q, seed, first_query_id=calls)
        calls += T
        b.resume((r["win"] + r["tie"]).astype(np.float64) / r["runs"].astype(np.float64))
    return b.stats()


def test_native_table_driver_on_gpu_equals_stepwise_batches(eng):
    """mcq_tables_run (native lock-step driver, BASELINE configs[4]) == the same tables stepped from Python with
    begin() -> eval_batch(seed, query ids counting up) -> resume(): same stacks, counts and pending queries; and the
    equities it acts on are the oracle's for the same (seed, query id)."""
    from neuron_poker_amd import _lib
    seats = [("equity", .5, -.5), ("equity", .8, -.8), ("equity", .3, .5), ("equity", .2, .75), ("random",), ("random",)]
    T, steps, seed = 96, 300, 77
    a = _lib.Tables(eng, T, seats, runs=1000, seed=seed)
    b = _lib.Tables(None, T, seats, runs=1000, seed=seed)
    st = a.run(steps)
    calls = 0
    for s in range(steps):
        q = b.begin()
        r = eng.eval_batch(q, seed, first_query_id=calls)
        if s % 100 == 0:
            o = O.run_batch(O.MODE_CTR, q.view(np.uint8).reshape(-1, 16), seed, first_qid=calls, threads=8)
            assert np.array_equal(o, r.view(np.uint64).reshape(-1, 13))
        calls += T
        b.resume((r["win"] + r["tie"]).astype(np.float64) / r["runs"].astype(np.float64))
    assert st == b.stats() and st["queries"] == T * steps and st["env_steps"] > 0 and st["episodes"] > 0
    c = _lib.Tables(eng, T, seats, runs=1000, seed=seed, overlap=False)   # one batch per lock-step on one stream
    assert c.run(steps) == st
    for t in range(T):
        sa, sb, sc = a.state(t), b.state(t), c.state(t)
        assert np.array_equal(sa["stacks"], sc.pop("stacks")) and {k: v for k, v in sa.items() if k != "stacks"} == sc
        assert np.array_equal(sa.pop("stacks"), sb.pop("stacks")) and sa == sb
    assert np.array_equal(a.begin(), b.begin())
    assert a.run(7) == b_run(b, eng, seed, T, steps, 7)   # a second run continues the query ids


def test_small_batch_task_split_does_not_change_tallies():
    """Small batches cut each 1024-iteration task into 2^split sub-tasks (pick_geometry in csrc/mcq_host.cpp);
    every setting of the cut must give the oracle's tallies, in both modes, for ragged run counts."""
    rng = np.random.default_rng(99)
    runs = [1000, 1, 17, 5000, 1024, 1025, 100000, 63, 64, 65, 2048, 999]
    hole, board, npl = [], [], []
    for i, r in enumerate(runs):
        c = rng.permutation(52)[:7]
        nb = [0, 3, 4, 5][i % 4]
        hole.append(c[:2])
        board.append(list(c[2:2 + nb]) + [255] * (5 - nb))
        npl.append(2 + i % 9)
    q = npa.pack_queries(hole, board, npl, runs)
    raw = q.view(np.uint8).reshape(-1, 16)
    want = {m: O.run_batch(om, raw, 31337, first_qid=5, threads=8)
            for m, om in [(npa.MODE_PHILOX, O.MODE_CTR), (npa.MODE_REPLAY_MT19937, O.MODE_MT)]}
    # the cut is limited to 512 pieces per query: the short queries alone reach the finest cut, the 100k one 2^2
    short = np.array([r <= 2048 for r in runs])
    old = os.environ.get("MCQ_SPLIT_MAX")
    try:
        for s in range(5):
            os.environ["MCQ_SPLIT_MAX"] = str(s)
            e = npa.Engine(0)
            for m in want:
                got = e.eval_batch(q, 31337, first_query_id=5, mode=m).view(np.uint64).reshape(-1, 13)
                assert np.array_equal(got, want[m]), (s, m)
                for i in np.nonzero(short)[0]:          # one query alone: first_query_id selects its stream
                    got = e.eval_batch(q[i:i + 1], 31337, first_query_id=5 + int(i), mode=m)
                    assert np.array_equal(got.view(np.uint64).reshape(-1, 13), want[m][i:i + 1]), (s, m, i)
            e.close()
    finally:
        if old is None:
            os.environ.pop("MCQ_SPLIT_MAX", None)
        else:
            os.environ["MCQ_SPLIT_MAX"] = old


def test_one_launch_path_for_small_queries_equals_the_general_path(monkeypatch):
    """Host entries send batches of small queries (the reference's pattern: 1000 runs each) through ONE launch
    (mcq_eval_direct_kernel: records read from and rows stored to pinned host memory, a query per block slot, no
    atomics).  Whatever the number of waves per query -- it follows the batch size -- the tallies are those of the
    general path (prep + sliced kernel + atomics) and of the oracle; both dealing laws; ragged runs and players."""
    g = np.random.default_rng(4242)

    def batch(n):
        hole, board, npl = [], [], []
        for i in range(n):
            nb = [0, 3, 4, 5][i % 4]
            c = g.permutation(52)[:2 + nb]
            hole.append(c[:2])
            board.append(list(c[2:]) + [255] * (5 - nb))
            npl.append(1 + (i * 7) % 10)
        runs = g.choice([1, 15, 16, 17, 64, 999, 1000, 1024, 1025, 3000, 8192], n)
        return npa.pack_queries(hole, board, npl, runs)

    monkeypatch.setenv("MCQ_DIRECT_MAX_TASKS", "0")
    general = npa.Engine(0)
    monkeypatch.setenv("MCQ_DIRECT_MAX_TASKS", "8")
    direct = npa.Engine(0, kernel_times=True)
    try:
        # 8, 8, 8, 8 (8 queries = 64 of the 128 wave slots: their work travels in the kernel arguments), 8, 8, 8, 4, 1, 1 waves
        # per query; 45000 queries = 11 rounds per block (the work is staged eight rounds at a time)
        for n in (1, 2, 3, 8, 9, 100, 257, 1000, 5000, 45000):
            q = batch(n)
            want = u64(general.eval_batch(q, seed=77, first_query_id=9))
            got = u64(direct.eval_batch(q, seed=77, first_query_id=9))
            assert np.array_equal(got, want), n
            if n <= 257:
                assert np.array_equal(want, O.run_batch(O.MODE_CTR, q.view(np.uint8).reshape(-1, 16), 77, first_qid=9, threads=8))
            assert direct.last_kernel_ms > 0
        direct.set_kernel_timing(False)          # the default: plain launches, nothing to report
        q = batch(3)
        assert np.array_equal(u64(direct.eval_batch(q, seed=77, first_query_id=9)), u64(general.eval_batch(q, seed=77, first_query_id=9)))
        assert direct.last_kernel_ms == 0 and general.last_kernel_ms == 0
        q = batch(300)
        for e in (general, direct):
            e.set_dealing_law("uniform")
        assert np.array_equal(u64(direct.eval_batch(q, seed=5)), u64(general.eval_batch(q, seed=5)))
        for e in (general, direct):
            e.set_dealing_law("reference")
        q["runs"][7] = 9000     # nine tasks: the whole batch takes the general path again
        assert np.array_equal(u64(direct.eval_batch(q, seed=5)), u64(general.eval_batch(q, seed=5)))
    finally:
        general.close()
        direct.close()


# ---- exact enumeration on the GPU (mcq_exact_batch)
def _xq(hero, board, n):
    b = [O.card_id(c) for c in board]
    return npa.pack_queries([[O.card_id(c) for c in hero]], [b + [255] * (5 - len(b))], n, 1)


def test_rows_through_the_publish_kernel_equal_rows_through_the_copy(monkeypatch):
    """Host-buffer calls on the general path (queries of more than 8 tasks) with up to 8192 rows get their rows through
    mcq_publish_kernel + the completion flag; larger ones (and MCQ_PUBLISH_MAX_ROWS=0) through D2H copy + stream
    synchronisation.  Same rows, odd and even counts, and the HBM rows are zero again for the next call."""
    g = np.random.default_rng(8192)

    def batch(n):
        hole = np.array([g.choice(52, 2, replace=False) for _ in range(n)], np.uint8)
        return npa.pack_queries(hole, np.full((n, 5), 255, np.uint8), g.integers(2, 8, n), g.choice([9000, 12000, 20000], n))

    monkeypatch.setenv("MCQ_PUBLISH_MAX_ROWS", "0")
    by_copy = npa.Engine(0)
    monkeypatch.delenv("MCQ_PUBLISH_MAX_ROWS")
    by_flag = npa.Engine(0)
    try:
        for n in (1, 2, 7, 64, 501):
            q = batch(n)
            want = u64(by_copy.eval_batch(q, seed=3, first_query_id=n))
            assert np.array_equal(u64(by_flag.eval_batch(q, seed=3, first_query_id=n)), want), n
            assert np.array_equal(u64(by_flag.eval_batch(q, seed=3, first_query_id=n)), want), n   # rows were left zero
        q = batch(5)
        assert np.array_equal(u64(by_flag.eval_batch(q, seed=9)), O.run_batch(O.MODE_CTR, q.view(np.uint8).reshape(-1, 16), 9, threads=8))
    finally:
        by_copy.close()
        by_flag.close()


def test_exact_enumeration_gpu_equals_host_lane_code_and_oracle(eng):
    from tests import hostsim as H
    from tests.test_lane_arithmetic_host import EXACT_CASES
    cases = EXACT_CASES + [(["TC", "TH"], ["4D", "QD", "KC", "2S"], 3), (["9C", "8C"], ["7C", "6D", "2S"], 3)]
    q = np.concatenate([_xq(*c) for c in cases])
    for law, uniform in (("reference", False), ("uniform", True)):
        got = eng.exact(q, law).view(np.uint64).reshape(-1, 13)
        for i, (hero, board, n) in enumerate(cases):
            if not (n == 3 and len(board) == 3):       # 1081 completions x 990 x 990: too slow for the host walk
                want = H.exact(q[i:i + 1].view(np.uint8).reshape(16), uniform)
                assert np.array_equal(got[i], want), (cases[i], law)
            if len(board) == 5 or (n == 2 and len(board) >= 3):
                win, tie, _ = O.exact(hero, board, n, uniform)
                assert abs(int(got[i][2]) / int(got[i][0]) - win) < 1e-9 and abs(int(got[i][3]) / int(got[i][0]) - tie) < 1e-9
            assert int(got[i][4:].sum()) == int(got[i][2] + got[i][3])


def test_exact_enumeration_three_players_on_the_flop_against_the_oracles_own_monte_carlo(eng):
    """Three players on the flop are too many leaves for the oracle's literal tree walk (7e9), so there the exact
    enumeration is checked against the ORACLE's Monte-Carlo of the reference itself -- MT19937 + numpy randint + the
    reference's loops, the code pinned to the reference's fixtures -- at 1.9e8 iterations: sigma of the estimate
    3.5e-5, bound 2e-4 (> 5 sigma).  Not a comparison of the product with itself."""
    hero, board, n = ["9C", "8C"], ["7C", "6D", "2S"], 3
    q = _xq(hero, board, n)
    ex = eng.exact(q, "reference").view(np.uint64).reshape(-1, 13)[0]
    exact = (int(ex[2]) + int(ex[3])) / int(ex[0])
    B, runs = 64, 3_000_000
    qq = npa.pack_queries([[O.card_id(c) for c in hero]] * B, [[O.card_id(c) for c in board] + [255, 255]] * B, n, runs)
    t = O.run_batch(O.MODE_MT, qq.view(np.uint8).reshape(-1, 16), 20261004, 0, threads=min(64, os.cpu_count() or 8))
    mc = float((t[:, 2] + t[:, 3]).sum()) / float(t[:, 0].sum())
    assert int(t[:, 0].sum()) == B * runs
    assert abs(mc - exact) < 2e-4, (mc, exact)


def test_exact_enumeration_preflop_known_values_and_symmetry(eng):
    """Heads-up preflop = 2.1e9 showdowns.  Uniform law: the textbook all-in equities vs a random hand (ties
    counted half there; here ties are the hero's, so compare win + tie/2), and invariance under relabelling suits.
    Reference law: total weight in closed form."""
    from math import comb
    q = np.concatenate([_xq(["AH", "KH"], [], 2), _xq(["AS", "KS"], [], 2), _xq(["AS", "AD"], [], 2),
                        _xq(["7C", "2D"], [], 2)])
    u = eng.exact(q, "uniform").view(np.uint64).reshape(-1, 13).astype(np.int64)
    assert (u[:, 0] == comb(50, 5) * 990).all()
    assert np.array_equal(u[0], u[1])                                      # AhKh == AsKs
    half = (u[:, 2] + u[:, 3] / 2) / u[:, 0]
    assert abs(half[0] - 0.6704) < 3e-4 and abs(half[2] - 0.8520) < 3e-4 and abs(half[3] - 0.346) < 5e-4
    r = eng.exact(q, "reference").view(np.uint64).reshape(-1, 13).astype(np.int64)
    assert (r[:, 0] == 49 ** 2 * comb(47, 5)).all()
    assert not np.array_equal(r[0][2:4] * u[1][0], u[0][2:4] * r[0][0])    # the index bias is real ...
    assert abs((r[0][2] + r[0][3]) / r[0][0] - 0.659) < 2e-3               # ... AhKh 0.659 vs 0.680


def test_production_rng_converges_to_exact_enumeration(eng):
    """The strongest form of gate B: 1e9 iterations of the production RNG against the exact expectation of the
    reference's dealing law (not another Monte-Carlo run): |delta| <= 1e-4 (~6.7 sigma of the MC error alone)."""
    q = _xq(["AH", "KH"], [], 2)
    x = eng.exact(q, "reference")[0]
    exact = (int(x["win"]) + int(x["tie"])) / int(x["runs"])
    qq = np.repeat(q, 250)
    qq["runs"] = 4_000_000
    r = eng.eval_batch(qq, seed=20261004)
    mc = (int(r["win"].sum()) + int(r["tie"].sum())) / int(r["runs"].sum())
    assert abs(mc - exact) <= 1e-4, (mc, exact)
    # three players on the flop, uniform law
    q3 = _xq(["9C", "8C"], ["7C", "6D", "2S"], 3)
    x3 = eng.exact(q3, "uniform")[0]
    eng.set_dealing_law("uniform")
    try:
        q3r = np.repeat(q3, 64)
        q3r["runs"] = 2_000_000
        r3 = eng.eval_batch(q3r, seed=5)
    finally:
        eng.set_dealing_law("reference")
    assert abs((int(r3["win"].sum()) + int(r3["tie"].sum())) / int(r3["runs"].sum())
               - (int(x3["win"]) + int(x3["tie"])) / int(x3["runs"])) <= 3e-4


def test_exact_enumeration_rejects_what_it_cannot_do(eng):
    with pytest.raises(ValueError):
        eng.exact(_xq(["AH", "KH"], [], 4))
    with pytest.raises(ValueError):
        eng.exact(_xq(["AH", "KH"], ["AH", "2C", "3C"], 2))
    eq, row = mh.get_equity_exact({"AH", "KH"}, {"2C", "7D", "9S", "JH", "QC"}, 2)
    assert int(row["runs"]) == 44 ** 2 and 0 < eq < 1


def test_iteration_shares_add_up_to_the_unsplit_result(eng):
    """SURVEY 8e, few large queries: every device takes one contiguous share of each query's iterations
    (mcq_eval_batch_part).  Each share == the oracle's tallies of exactly those iterations; the shares of 2, 3 and
    8 devices add up -- runs included -- to the unsplit call, bit for bit."""
    rng = np.random.default_rng(8)
    runs = [100000, 1, 1024, 1025, 20000, 3073, 64, 5000]
    hole, board, npl = [], [], []
    for i in range(len(runs)):
        c = rng.permutation(52)[:7]
        nb = [0, 3, 4, 5][i % 4]
        hole.append(c[:2])
        board.append(list(c[2:2 + nb]) + [255] * (5 - nb))
        npl.append(2 + i % 5)
    q = npa.pack_queries(hole, board, npl, runs)
    raw = q.view(np.uint8).reshape(-1, 16)
    whole = eng.eval_batch(q, 99, first_query_id=7).view(np.uint64).reshape(-1, 13)
    assert np.array_equal(whole, O.run_batch(O.MODE_CTR, raw, 99, first_qid=7, threads=8))
    for n_parts in (2, 3, 8):
        total = np.zeros_like(whole)
        for p in range(n_parts):
            got = eng.eval_batch(q, 99, first_query_id=7, part=(p, n_parts)).view(np.uint64).reshape(-1, 13)
            if n_parts == 3:
                assert np.array_equal(got, O.run_batch_part(O.MODE_CTR, raw, 99, 7, p, n_parts)), p
            total += got
        assert np.array_equal(total, whole), n_parts
    with pytest.raises(ValueError):
        eng.eval_batch(q, 99, part=(2, 2))
    with pytest.raises(ValueError):
        eng.eval_batch(q, 99, mode=npa.MODE_REPLAY_MT19937, part=(0, 2))
    # one query of 2e8 iterations over "8 devices": the same tallies as in one piece
    big = npa.pack_queries([[50, 46]], [[255] * 5], 2, 200_000_000)
    one = eng.eval_batch(big, 3).view(np.uint64).reshape(-1, 13)
    eight = sum(eng.eval_batch(big, 3, part=(p, 8)).view(np.uint64).reshape(-1, 13) for p in range(8))
    assert np.array_equal(one, eight)


def test_c_example_runs_against_the_c_abi(tmp_path):
    """examples/equity.c: the ABI from plain C (no Python, no torch) -- Monte-Carlo, exact and the table driver."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, lib = str(tmp_path / "equity"), npa.library_path()
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "equity.c"),
                           "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib)])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert "100000 iterations" in lines[0] and abs(float(lines[0].split("equity ")[1].split()[0]) - 0.6598) < 0.006
    assert "0.659833" in lines[1]
    assert lines[2].startswith("512 tables, 1000 lock-steps") and "512000 equity queries" in lines[2]


def test_bench_two_rank_rehearsal_on_one_gpu():
    """bench.py's N > 1 control flow (one process per rank, tallies all-reduced, max-over-ranks timing, one JSON line
    from rank 0) with two ranks sharing this GPU and gloo carrying the collective; the real runs use RCCL."""
    import json as _json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    small = ["--steps", "2", "--warmup", "1", "--states", "256", "--iters", "3000", "--no-extras", "--no-cpu-baseline"]
    # (a) exactly as the scaling driver may invoke it: no launcher prefix; bench.py starts its own ranks
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--single-device"] + small, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(line) == 1, out.stdout
    d = _json.loads(line[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["hand_evals_per_step"] == 2 * 256 * 3000 * 6
    # (b) under the launcher, as the contract's N > 1 command line
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--single-device"] + small,
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len([x for x in out.stdout.splitlines() if x.startswith("{")]) == 1, out.stdout


def test_bench_rccl_branch_runs_on_this_gpu():
    """The RCCL code path of bench.py (process group "nccl" = RCCL, all-reduce of the tally matrix on the device,
    barrier, max-over-ranks timing) executed for real: one rank under the launcher -- all a one-GPU box can hold,
    RCCL refuses two ranks on one device -- makes exactly the calls an N-GPU job makes."""
    import json as _json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "1", "--steps", "2", "--warmup", "1", "--states", "256", "--iters", "3000",
                          "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(line) == 1, out.stdout
    d = _json.loads(line[0])
    assert d["n_gpus"] == 1 and "RCCL all-reduce" in d["config"]["workload"] and d["value"] > 0
    assert d["collective"]["ranks"] == 1 and d["collective"]["all_reduce_ms"] > 0 and "roofline" in d


def test_bench_native_multi_prints_the_same_keys():
    """`bench.py --native-multi` (ONE process, mcq_multi_eval_batch_device: shards -> one ncclAllReduce) rehearsed with two
    shards on this GPU: one JSON line with the keys of the one-rank-per-GPU line (roofline, collective)."""
    import json as _json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--native-multi", "--single-device", "--gpus", "2",
                          "--steps", "2", "--warmup", "1", "--states", "256", "--iters", "3000"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [x for x in out.stdout.splitlines() if x.startswith("{")]
    assert len(line) == 1, out.stdout
    d = _json.loads(line[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["collective"]["all_reduce_ms"] >= 0 and d["roofline"]["kernel_ms"] > 0 and d["native_multi"]["kernel_max_ms"] > 0


def test_multi_gpu_device_entry_equals_single_context(eng):
    """mcq_multi_eval_batch_device: queries and results resident in HBM, one pointer pair per shard (here all shards on
    the one GPU; on a multi-GPU lease also one shard per device): every shard's buffer ends up holding the complete
    matrix, equal to mcq_eval_batch on one context."""
    import torch
    g = np.random.default_rng(77)
    B = 600
    hole = np.array([g.permutation(52)[:2] for _ in range(B)], np.uint8)
    q = npa.pack_queries(hole, np.full((B, 5), 255, np.uint8), 1 + np.arange(B) % 6, g.choice([1, 500, 1024, 2100], B))
    want = u64(eng.eval_batch(q, seed=5, first_query_id=9))
    raw = q.view(np.uint8).reshape(B, 16)
    n_dev = npa.load_library().mcq_device_count()
    layouts = [[0], [0, 0], [0, 0, 0]] + ([list(range(n_dev))] if n_dev > 1 else [])
    for devices in layouts:
        k = len(devices)
        me = npa.MultiEngine(devices)
        try:
            for part in ("queries", "iterations"):
                qs, rs = [], []
                for s, d in enumerate(devices):
                    lo, hi = (B * s // k, B * (s + 1) // k) if part == "queries" else (0, B)
                    qs.append(torch.from_numpy(raw[lo:hi].copy()).to("cuda:%d" % d))
                    rs.append(torch.full((B, 13), -1, dtype=torch.int64, device="cuda:%d" % d))
                torch.cuda.synchronize()
                me.eval_batch_device([t.data_ptr() if t.numel() else 0 for t in qs], B, 5, [t.data_ptr() for t in rs],
                                     first_query_id=9, partition=part)
                for t in rs:
                    assert np.array_equal(t.cpu().numpy().view(np.uint64), want), (devices, part)
        finally:
            me.close()


def test_get_equity_batch_over_several_gpus(eng):
    """the batched call of the shim with n_gpus=: the library's multi-GPU entry behind it, the same integers
    as on one GPU.  (Needs two devices; on a one-GPU box the argument is only checked.)"""
    with pytest.raises(ValueError):
        mh.get_equity_batch([[51, 47]], [[255] * 5], 2, 100, seed=1, mode="replay", n_gpus=2)
    n_dev = npa.load_library().mcq_device_count()
    if n_dev < 2:
        pytest.skip("one device")
    g = np.random.default_rng(3)
    cards = np.array([g.permutation(52)[:7] for _ in range(600)], np.uint8)
    board = np.full((600, 5), 255, np.uint8)
    board[::2, :3] = cards[::2, 2:5]
    one = mh.get_equity_batch(cards[:, :2], board, 6, 3000, seed=9, engine=eng)[1]
    many = mh.get_equity_batch(cards[:, :2], board, 6, 3000, seed=9, n_gpus=n_dev)[1]
    assert np.array_equal(one, many)


def test_multi_gpu_entry_partitions_equal_single_context(eng):
    """mcq_multi_eval_batch (include/mcq.h): shards + one ncclAllReduce.  Whatever the partition -- blocks of
    queries, shares of every query's iterations, more shards than queries, one shard -- the tallies are those of
    mcq_eval_batch on one context, and of the oracle."""
    g = np.random.default_rng(2026)
    B = 41
    hole, board, npl = [], [], []
    for i in range(B):
        nb = [0, 3, 4, 5][i % 4]
        c = g.permutation(52)[:2 + nb]
        hole.append(c[:2])
        board.append(list(c[2:]) + [255] * (5 - nb))
        npl.append(1 + i % 10)
    runs = g.choice([1, 700, 1024, 3073, 20000], B)
    q = npa.pack_queries(hole, board, npl, runs)
    want = u64(eng.eval_batch(q, seed=11, first_query_id=500))
    assert np.array_equal(want, O.run_batch(O.MODE_CTR, q.view(np.uint8).reshape(-1, 16), 11, first_qid=500, threads=8))
    n_dev = npa.load_library().mcq_device_count()
    layouts = [[0] * shards for shards in (1, 2, 3, 8)]
    if n_dev > 1:   # a multi-GPU lease: every visible device once, and twice (the cross-device all-reduce for real)
        layouts += [list(range(n_dev)), list(range(n_dev)) * 2]
    for devices in layouts:
        shards = len(devices)
        me = npa.MultiEngine(devices)
        try:
            for part in ("queries", "iterations", "auto"):
                got = u64(me.eval_batch(q, seed=11, first_query_id=500, partition=part))
                assert np.array_equal(got, want), (devices, part)
            assert me.info["last_partition"] == "iterations"      # 41 < 256 * shards
            assert me.info["devices"] == len(set(devices))
            assert np.array_equal(u64(me.eval_batch(q[:2], seed=11, first_query_id=500, partition="queries")), want[:2])
            bad = q.copy()
            bad["hole"][7] = (3, 3)
            with pytest.raises(ValueError):
                me.eval_batch(bad, seed=1)
            assert len(me.eval_batch(q[:0], seed=1)) == 0
            t = me.last_times_ms
            assert t["call"] > 0 and t["all_reduce"] >= 0
            me.set_dealing_law("uniform")
            eng.set_dealing_law("uniform")
            try:
                assert np.array_equal(u64(me.eval_batch(q, seed=3)), u64(eng.eval_batch(q, seed=3)))
            finally:
                eng.set_dealing_law("reference")
        finally:
            me.close()


def test_device_entry_for_small_queries_is_one_launch_and_equals_the_host_entry(eng):
    """mcq_eval_batch_device_small: queries of at most 8192 iterations resident in HBM, ONE kernel launch (the one-launch
    kernel lays its own work out: 2^lg waves per query from the query count alone, validation on the device).  Tallies
    == the host entry's for every batch size regime (one query ... more than one round per block); an invalid query, one
    without iterations and one that is too long get their marked rows; two streams at once; replay inside a HIP graph."""
    import torch
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(99)
    for B in (1, 7, 300, 1024, 5000):
        hole, board, npl = [], [], []
        for i in range(B):
            nb = [0, 3, 4, 5][i % 4]
            c = g.permutation(52)[:2 + nb]
            hole.append(c[:2])
            board.append(list(c[2:]) + [255] * (5 - nb))
            npl.append(1 + i % 10)
        runs = g.choice([1, 63, 1000, 1024, 1025, 5000, 8192], B)
        q = npa.pack_queries(hole, board, npl, runs)
        want = u64(eng.eval_batch(q, seed=21, first_query_id=77))
        bad = q.copy()
        if B >= 7:
            bad["hole"][2] = (9, 9)        # invalid: runs = 0, passes = 2^64 - 1
            bad["runs"][3] = 0             # nothing to do: a row of zeros
            bad["runs"][4] = 8193          # too long for this entry: marked like an invalid query
            want = want.copy()
            want[2] = 0; want[2, 1] = 2 ** 64 - 1
            want[3] = 0
            want[4] = 0; want[4, 1] = 2 ** 64 - 1
        d_q = torch.from_numpy(bad.view(np.uint8).reshape(-1, 16).copy()).to(dev)
        out = torch.full((B, 13), -7, dtype=torch.int64, device=dev)
        eng.eval_batch_device_small(d_q.data_ptr(), B, 21, out.data_ptr(), first_query_id=77,
                                    stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want), B
    # two streams at once, then a graph replay
    qa = npa.pack_queries([g.permutation(52)[:2] for _ in range(700)], np.full((700, 5), 255, np.uint8), 6, 1000)
    qb = npa.pack_queries([g.permutation(52)[:2] for _ in range(90)], np.full((90, 5), 255, np.uint8), 2, 3000)
    wa, wb = u64(eng.eval_batch(qa, 1)), u64(eng.eval_batch(qb, 2))
    da = torch.from_numpy(qa.view(np.uint8).reshape(-1, 16).copy()).to(dev)
    db = torch.from_numpy(qb.view(np.uint8).reshape(-1, 16).copy()).to(dev)
    oa = torch.zeros((700, 13), dtype=torch.int64, device=dev)
    ob = torch.zeros((90, 13), dtype=torch.int64, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        eng.eval_batch_device_small(da.data_ptr(), 700, 1, oa.data_ptr(), stream=s1.cuda_stream)
        eng.eval_batch_device_small(db.data_ptr(), 90, 2, ob.data_ptr(), stream=s2.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(oa.cpu().numpy().view(np.uint64), wa) and np.array_equal(ob.cpu().numpy().view(np.uint64), wb)
    s = torch.cuda.Stream()
    eng.eval_batch_device_small(da.data_ptr(), 700, 1, oa.data_ptr(), stream=s.cuda_stream)   # the stream's slot exists now
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        eng.eval_batch_device_small(da.data_ptr(), 700, 1, oa.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        oa.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(oa.cpu().numpy().view(np.uint64), wa)


def test_device_entry_on_two_streams_at_once(eng):
    """Two asynchronous device-entry calls in flight on different streams of one context: each stream has its own
    scheduling scratch, so neither disturbs the other (and a host-entry call in between uses the context's own)."""
    import torch
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(5)
    qa = npa.pack_queries([g.permutation(52)[:2] for _ in range(900)], np.full((900, 5), 255, np.uint8), 6, 40000)
    qb = npa.pack_queries([g.permutation(52)[:2] for _ in range(300)], np.full((300, 5), 255, np.uint8), 3, 2000)
    want_a, want_b = u64(eng.eval_batch(qa, 7, first_query_id=1)), u64(eng.eval_batch(qb, 8, first_query_id=2))
    d_a = torch.from_numpy(qa.view(np.uint8).reshape(-1, 16).copy()).to(dev)
    d_b = torch.from_numpy(qb.view(np.uint8).reshape(-1, 16).copy()).to(dev)
    o_a = torch.zeros((900, 13), dtype=torch.int64, device=dev)
    o_b = torch.zeros((300, 13), dtype=torch.int64, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        o_a.zero_(); o_b.zero_()
        torch.cuda.synchronize()
        eng.eval_batch_device(d_a.data_ptr(), 900, 7, o_a.data_ptr(), first_query_id=1, stream=s1.cuda_stream)   # long
        eng.eval_batch_device(d_b.data_ptr(), 300, 8, o_b.data_ptr(), first_query_id=2, stream=s2.cuda_stream)   # short
        mid = u64(eng.eval_batch(qb, 8, first_query_id=2))                                                        # host entry
        torch.cuda.synchronize()
        assert np.array_equal(o_a.cpu().numpy().view(np.uint64), want_a)
        assert np.array_equal(o_b.cpu().numpy().view(np.uint64), want_b) and np.array_equal(mid, want_b)


def test_device_entry_refuses_to_grow_its_scratch_inside_a_capture(eng):
    """A first call on a stream that is being captured would have to allocate: it fails cleanly with ValueError
    (MCQ_EINVAL) instead of breaking the capture half way; after one call outside the capture it works."""
    import torch
    dev = torch.device("cuda", 0)
    e2 = npa.Engine(0)
    try:
        q = npa.pack_queries([[1, 2]] * 64, np.full((64, 5), 255, np.uint8), 2, 1000)
        d_q = torch.from_numpy(q.view(np.uint8).reshape(-1, 16).copy()).to(dev)
        out = torch.zeros((64, 13), dtype=torch.int64, device=dev)
        s = torch.cuda.Stream()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with pytest.raises(ValueError):
            with torch.cuda.graph(graph, stream=s):
                e2.eval_batch_device(d_q.data_ptr(), 64, 1, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        prev = torch.cuda.current_device()
        e2.eval_batch_device(d_q.data_ptr(), 64, 1, out.data_ptr(), stream=s.cuda_stream)
        assert torch.cuda.current_device() == prev
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), u64(e2.eval_batch(q, 1)))
    finally:
        e2.close()


def test_device_entry_inside_a_hip_graph(eng):
    """mcq_eval_batch_device launches only asynchronous work on the caller's stream, so it can be captured into a
    HIP graph (here through torch.cuda.CUDAGraph) and replayed: a launch-bound small batch then costs one graph
    launch.  The replay must give the host entry's tallies."""
    import torch
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(12)
    B = 384
    hole = np.array([g.permutation(52)[:2] for _ in range(B)], np.uint8)
    q = npa.pack_queries(hole, np.full((B, 5), 255, np.uint8), 4, 1500)
    want = eng.eval_batch(q, 424242, first_query_id=9).view(np.uint64).reshape(-1, 13)
    d_q = torch.from_numpy(q.view(np.uint8).reshape(B, 16).copy()).to(dev)
    out = torch.zeros((B, 13), dtype=torch.int64, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):       # warm-up on the side stream: sizes the engine's prefix buffer before capture
        eng.eval_batch_device(d_q.data_ptr(), B, 1, out.data_ptr(), first_query_id=0, stream=s.cuda_stream)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        eng.eval_batch_device(d_q.data_ptr(), B, 424242, out.data_ptr(), first_query_id=9,
                              stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want)


def test_device_entry_small_batches_are_cut_by_the_prep_kernel(eng):
    """Queries resident in HBM: the host never sees them, so the prep kernel chooses the small-batch cut
    (MCQ_SPLIT_FROM_PREP).  Tallies == the host entry == the oracle, for batch sizes around the 1024 limit."""
    import torch
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(77)
    for B, runs in ((1, 100000), (1, 1000), (300, 1000), (1024, 1500), (1025, 700), (37, 1), (5, 1000000)):
        hole = np.array([g.permutation(52)[:2] for _ in range(B)], np.uint8)
        q = npa.pack_queries(hole, np.full((B, 5), 255, np.uint8), 2 + (B % 5), runs)
        want = eng.eval_batch(q, 4711, first_query_id=3).view(np.uint64).reshape(-1, 13)
        if B * runs <= 2_000_000:
            assert np.array_equal(want, O.run_batch(O.MODE_CTR, q.view(np.uint8).reshape(-1, 16), 4711, first_qid=3, threads=8))
        d_q = torch.from_numpy(q.view(np.uint8).reshape(B, 16).copy()).to(dev)
        out = torch.zeros((B, 13), dtype=torch.int64, device=dev)
        eng.eval_batch_device(d_q.data_ptr(), B, 4711, out.data_ptr(), first_query_id=3)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want), (B, runs)


def test_torch_tensors_in_hbm_end_to_end(eng):
    """neuron_poker_amd.torch_ops: states as CUDA tensors -> equity as a CUDA tensor, nothing through the host; the
    numbers are those of the host entry."""
    import torch
    from neuron_poker_amd import torch_ops
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(31)
    B = 700
    hole = np.zeros((B, 2), np.uint8)
    board = np.full((B, 5), 255, np.uint8)
    for i in range(B):
        nb = [0, 3, 4, 5][i % 4]
        c = g.permutation(52)[:2 + nb]
        hole[i] = c[:2]
        slots = g.permutation(5)[:nb]          # empty slots anywhere: the packing moves the cards to the left
        board[i, slots] = c[2:]
    npl = g.integers(1, 11, B).astype(np.uint8)
    runs = g.choice([1, 64, 1000, 1500], B).astype(np.int32)
    eq_ref, t_ref = mh.get_equity_batch(hole, board, npl, runs, seed=99, first_query_id=5, engine=eng)
    eq, t = torch_ops.get_equity_batch_torch(torch.from_numpy(hole).to(dev), torch.from_numpy(board).to(dev),
                                             torch.from_numpy(npl).to(dev), torch.from_numpy(runs).to(dev),
                                             seed=99, first_query_id=5, engine=eng)
    assert eq.is_cuda and t.is_cuda
    assert np.array_equal(t.cpu().numpy().view(np.uint64), t_ref) and np.array_equal(eq.cpu().numpy(), eq_ref)
    with pytest.raises(ValueError):
        torch_ops.get_equity_batch_torch(torch.from_numpy(hole), torch.from_numpy(board), 2, 10)


# ------------------------------------------------------------------------------------------ round 4: threads, get_winner
def test_hand_evaluator_dropin_matches_reference(eng):
    """neuron_poker_amd.hand_evaluator_hip -- get_winner / eval_best_hand with the reference's card strings
    (tools/hand_evaluator.py:9-24; gym_env/env.py:587 is the caller) -- on the reference's own showdowns and cases."""
    from neuron_poker_amd import hand_evaluator_hip as he
    z = np.load(os.path.join(G, "showdowns.npz"))
    hands, n, win, wt = z["hands"], z["n_players"], z["winner"], z["winner_type"]
    for i in range(0, len(hands), 97):   # every 97th of the 24 000 reference showdowns, 2 to 10 hands, one call each
        p = int(n[i])
        rows = [[npa.card_str(c) for c in hands[i, k]] for k in range(p)]
        table = rows[0][2:]
        if all(r[2:] == table for r in rows):   # one table for all: the get_winner form
            assert he.get_winner([r[:2] for r in rows], table) == (int(win[i]), npa.TYPES[int(wt[i])])
        best, kind = he.eval_best_hand(rows)
        assert best == rows[int(win[i])] and kind == npa.TYPES[int(wt[i])]
    for c in jload("evaluator_cases.json"):
        ids = [[npa.card_id(x) for x in h] for h in c["hands"]]
        if any(len(set(h)) != 7 for h in ids):
            with pytest.raises(ValueError):   # a card named twice: outside the domain (documented difference)
                he.eval_best_hand(c["hands"])
            continue
        best, kind = he.eval_best_hand(c["hands"])
        assert best == c["hands"][c["winner"]] and kind == c["winner_type"], c
    # many tables in one launch
    sel = np.nonzero(n == 6)[0][:500]
    same_table = [i for i in sel if (hands[i, :6, 2:] == hands[i, 0, 2:]).all()]
    if same_table:
        w, t = he.get_winner_batch(hands[same_table, :6, :2], hands[same_table, 0, 2:])
        assert np.array_equal(w, win[same_table]) and np.array_equal(t, wt[same_table])
    with pytest.raises(ValueError):
        he.get_winner([["AH", "KH"], ["QD", "2C"]], ["2C", "3C", "4C", "5D", "9S"])   # 2C twice in one hand
    with pytest.raises(ValueError):
        he.get_winner([["AH", "1H"]], ["2C", "3C", "4C", "5D", "9S"])


def _equity_calls(seed, count):
    """`count` get_equity calls as HoldemTable makes them (gym_env/env.py:261-262), stream `seed` of this thread."""
    g = np.random.default_rng(seed)
    mh.seed(seed)
    out = []
    for _ in range(count):
        nb = int(g.choice([0, 3, 4, 5]))
        cards = [npa.card_str(c) for c in g.choice(52, 2 + nb, replace=False)]
        out.append(mh.get_equity(set(cards[:2]), set(cards[2:]), int(g.integers(2, 7)), 1000))
    return out


def test_threads_calling_get_equity_equal_the_serial_calls():
    """SURVEY 8b threading: a context allows one call in flight, ctypes drops the GIL during a call -- so every thread
    has its own default engine and its own (seed, counter) stream: four threads x 200 get_equity calls == the same calls
    made one after the other."""
    import threading
    got, errors = {}, []

    def work(t):
        try:
            got[t] = _equity_calls(1000 + t, 200)
        except Exception as e:   # noqa: BLE001 -- reported below
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        assert got[t] == _equity_calls(1000 + t, 200), t
    engines = set()
    box = []
    th = threading.Thread(target=lambda: box.append(npa.default_engine()))
    th.start(); th.join()
    engines.add(id(box[0])); engines.add(id(npa.default_engine()))
    assert len(engines) == 2   # per thread


def test_second_call_on_a_busy_context_is_turned_away(eng):
    """include/mcq.h: one call in flight per context; the library checks it (MCQ_EBUSY) instead of letting two callers
    race on the context's staging buffers."""
    import threading
    import time
    g = np.random.default_rng(5)
    cards = np.array([g.permutation(52)[:2] for _ in range(4096)], np.uint8)
    big = npa.pack_queries(cards, np.full((4096, 5), 255, np.uint8), 6, 100000)   # about 6 ms on the GPU
    small = mkq(["AH", "KH"], [], 2, 1000)
    want_small = u64(eng.eval_batch(small, seed=3))
    started, results, busy = threading.Event(), [], [0]

    def long_call():
        started.set()
        results.append(u64(eng.eval_batch(big, seed=1)))
    th = threading.Thread(target=long_call)
    th.start()
    started.wait()
    deadline = time.time() + 5
    while th.is_alive() and time.time() < deadline:
        try:
            assert np.array_equal(u64(eng.eval_batch(small, seed=3)), want_small)   # got in between two calls: fine
        except npa.McqBusyError as e:
            assert "context busy" in str(e)
            busy[0] += 1
    th.join()
    assert busy[0] > 0
    assert np.array_equal(results[0], u64(eng.eval_batch(big, seed=1)))   # the long call was not disturbed
    assert np.array_equal(u64(eng.eval_batch(small, seed=3)), want_small)


def test_get_equity_batch_shards_on_one_device_follow_the_dealing_law(eng):
    """The shim's multi-GPU branch on the one GPU there is (devices=[0, 0, 0]: three shards, one all-reduce rank): same
    integers as the single-context call -- under the uniform law too, which configure() hands on to the multi-GPU engines
    (round-3 advice: it did not)."""
    g = np.random.default_rng(12)
    cards = np.array([g.permutation(52)[:7] for _ in range(900)], np.uint8)
    board = np.full((900, 5), 255, np.uint8)
    board[::2, :3] = cards[::2, 2:5]
    try:
        for law in ("reference", "uniform", "reference"):
            mh.configure(dealing=law)
            eng.set_dealing_law(law)
            one = mh.get_equity_batch(cards[:, :2], board, 6, 3000, seed=9, engine=eng)[1]
            many = mh.get_equity_batch(cards[:, :2], board, 6, 3000, seed=9, devices=[0, 0, 0])[1]
            own = mh.get_equity_batch(cards[:, :2], board, 6, 3000, seed=9)[1]   # this thread's default engine
            assert np.array_equal(one, many) and np.array_equal(one, own), law
        ref = one
        mh.configure(dealing="uniform")
        assert not np.array_equal(ref, mh.get_equity_batch(cards[:, :2], board, 6, 3000, seed=9, devices=[0, 0, 0])[1])
    finally:
        mh.configure(dealing="reference")
        eng.set_dealing_law("reference")


def test_bench_configs3_workload_is_partition_invariant():
    """`bench.py --workload configs3` (BASELINE configs[3]: 65 536 flop/turn states x 6 x 20k IN ALL, block-sharded over the
    ranks, strong scaling, one all-reduce): one rank, two ranks on this GPU (gloo carrying the collective), and the
    one-process multi-GPU entry with three shards all print ONE line with the same checksum of the whole tally matrix."""
    import json as _json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--workload", "configs3", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    runs = {"one rank": ["--gpus", "1"],
            "two ranks": ["--gpus", "2", "--single-device", "--backend", "gloo"],
            "three shards": ["--gpus", "3", "--single-device", "--native-multi"]}
    sums = {}
    for name, extra in runs.items():
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra + common, capture_output=True, text=True,
                             timeout=600, cwd=root)
        assert out.returncode == 0, (name, out.stderr[-2000:])
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        assert len(line) == 1, (name, out.stdout)
        d = _json.loads(line[0])
        assert d["scaling"] == "strong" and "configs[3]" in d["metric"] and "configs[3]" in d["config"]["workload"], name
        assert d["config"]["hand_evals_per_step"] == 65536 * 6 * 20000 and d["config"]["states_total"] == 65536
        assert d["value"] > 0 and d["roofline"]["kernel_ms"] > 0, name
        if name != "one rank":
            assert d["collective"]["bytes"] == 65536 * 104, name
        sums[name] = d["tallies_sha256"]
    assert len(set(sums.values())) == 1, sums
