"""Pins the CPU oracle (oracle/mcq_oracle.c) to the reference.

Every expected value below comes from tests/golden/*, which tests/golden/gen_golden.py produced by running
/root/reference/tools/{hand_evaluator,montecarlo_python}.py itself under np.random.seed.  numpy (a third-party
dependency of the reference, importable here) is additionally used live to pin the MT19937 / randint layer.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def jload(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


# ------------------------------------------------------------------ RNG layer (numpy legacy RandomState)
@pytest.mark.parametrize("seed", [0, 1, 5489, 12345, 2 ** 31, 2 ** 32 - 1])
def test_mt19937_words_match_numpy(seed):
    rs = np.random.RandomState(seed)
    # legacy randint over the full 32-bit range hands out raw tempered words
    ref = rs.randint(0, 2 ** 32, size=2000, dtype=np.uint32)
    assert np.array_equal(O.mt_words(seed, 2000), ref)


def test_mt19937_first_word_default_seed():
    assert O.mt_words(5489, 1)[0] == 3499211612


@pytest.mark.parametrize("seed", [0, 7, 99991])
def test_masked_randint_matches_numpy(seed):
    g = np.random.default_rng(seed)
    bounds = g.integers(1, 53, size=5000).astype(np.uint32)
    bounds[:10] = [1, 2, 3, 4, 5, 31, 32, 33, 50, 52]
    np.random.seed(seed)
    ref = np.array([np.random.randint(0, int(b)) for b in bounds], np.uint32)
    got, words = O.np_randint(seed, bounds)
    assert np.array_equal(got, ref)
    # both generators must now sit at the same word: the next raw word agrees
    nxt = np.random.randint(0, 2 ** 32, dtype=np.uint32)
    assert O.mt_words(seed, words + 1)[-1] == nxt


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert [hex(x) for x in O.philox4x32_10([0] * 4, [0] * 2)] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c',
                                                                    '0x9b00dbd8']
    assert [hex(x) for x in O.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == ['0x408f276d', '0x41c83b0e',
                                                                                      '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in O.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                            [0xa4093822, 0x299f31d0])] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420',
                                                                           '0x24126ea1']


def test_mwc64x_reference_sequence():
    # independent restatement of MWC64X (D. B. Thomas: x' = lo, c' = hi of A * x + c, A = 4294883355, output x ^ c)
    # in Python ints, seeded with the Philox block of (seed, query id, stream) as MCQ-CTR v5 does
    o = [int(v) for v in O.philox4x32_10([5, 0, 3, 0x4D435131], [77, 1])]
    x, c = o[0], o[1] >> 1
    exp = []
    for _ in range(64):
        exp.append(x ^ c)
        t = 4294883355 * x + c
        x, c = t & 0xffffffff, t >> 32
    assert list(O.ctr_stream(77 + (1 << 32), 5, 3, 64)) == exp
    # Thomas' published recurrence in its OpenCL form (mul_hi / carry) gives the same step
    x, c = 12345, 678
    hi = (x * 4294883355) >> 32
    nx = (x * 4294883355 + c) & 0xffffffff
    nc = hi + (1 if nx < c else 0)
    t = 4294883355 * x + c
    assert (nx, nc) == (t & 0xffffffff, t >> 32)


# ------------------------------------------------------------------ evaluator
def test_evaluator_hands_fixture():
    z = np.load(os.path.join(G, "evaluator_hands.npz"))
    cards, cat, nr, ranks = z["cards"], z["category"], z["n_ranks"], z["card_ranks"]
    assert len(cards) == 50000
    for i in range(len(cards)):
        _, r, t = O.calc_score(cards[i])
        assert t == cat[i], (i, cards[i])
        assert list(r) == list(ranks[i, :nr[i]]), (i, cards[i], r)


def test_evaluator_reference_cases():
    cases = jload("evaluator_cases.json")
    assert sum(c["source"] == "tests/test_evaluator.py" for c in cases) == 14
    for c in cases:
        w, t, _ = O.best_hand(c["hands"])
        assert w == c["winner"], c
        assert O.TYPES[t] == c["winner_type"], c
        for hand, (score, ranks, typ) in zip(c["hands"], c["scores"]):
            s, r, ti = O.calc_score(hand)
            assert (list(s), list(r), O.TYPES[ti]) == (score, ranks, typ), (hand, s, r)


def test_showdowns_fixture():
    z = np.load(os.path.join(G, "showdowns.npz"))
    hands, n, win, wt = z["hands"], z["n_players"], z["winner"], z["winner_type"]
    for i in range(len(hands)):
        w, t, _ = O.best_hand(hands[i, :n[i]])
        assert (w, t) == (win[i], wt[i]), i


# ------------------------------------------------------------------ dealing + tallies (MT mode == the reference)
def test_deal_traces_fixture():
    z = np.load(os.path.join(G, "deal_traces.npz"))
    meta = json.loads(str(z["meta"]))
    for i, m in enumerate(meta):
        r = O.run(O.MODE_MT, m["hero"], m["board"], m["n_players"], m["runs"], m["seed"], keep=m["runs"])
        assert np.array_equal(r["trace"], z["hands_%d" % i]), m
        assert np.array_equal(r["words"], z["words_%d" % i]), m
        assert (r["wins"], r["passes"], r["mt_words"]) == (m["wins"], m["passes"], m["mt_words"]), m


def test_tallies_fixture():
    rows = jload("tallies.json")
    assert len(rows) >= 18
    for t in rows:
        r = O.run(O.MODE_MT, t["hero"], t["board"], t["n_players"], t["runs"], t["seed"])
        assert r["runs"] == t["runs"]
        assert r["wins"] == t["wins"], t
        assert r["passes"] == t["passes"], t
        assert r["by_type"] == t["by_type"], t
        assert r["mt_words"] == t["mt_words"], t
        assert r["win"] + r["tie"] == sum(r["by_type"])


def test_known_answers_survey_f3():
    # AH KH heads-up, no board
    for seed, runs, wins, passes in [(0, 10000, 6629, 10213), (1, 10000, 6529, 10225), (12345, 10000, 6645, 10209),
                                     (0, 100000, 65807, 102091), (1, 100000, 65985, 102097)]:
        r = O.run(O.MODE_MT, ["AH", "KH"], [], 2, runs, seed)
        assert (r["wins"], r["passes"]) == (wins, passes)


def test_statistical_expectations_of_reference_tests():
    # tests/test_montecarlo_python.py asserts |mean - expected| < 3 points
    for row in jload("stat_expectations.json"):
        r = O.run(O.MODE_MT, row["hero"], row["board"], row["n_players"], 30000, 4242)
        assert abs(100 * r["equity"] - row["expected_pct"]) < row["tol_pct"], (row, r["equity"])


# ------------------------------------------------------------------ CTR mode and the exact enumerator
def test_ctr_mode_is_partition_invariant_and_close_to_mt():
    a = O.run(O.MODE_CTR, ["AH", "KH"], [], 2, 40000, 9, qid=3)
    q = O.pack_queries([[O.card_id("AH"), O.card_id("KH")]], [[255] * 5], 2, 40000)
    b = O.run_batch(O.MODE_CTR, q, 9, first_qid=3)[0]
    assert np.array_equal(a["tallies"], b)
    m = O.run(O.MODE_MT, ["AH", "KH"], [], 2, 40000, 9)
    assert abs(a["equity"] - m["equity"]) < 0.012  # ~3.5 sigma of the difference at 40k runs


def test_exact_enumeration_agrees_with_both_front_ends():
    hero, board = ["TC", "TH"], ["4D", "QD", "KC", "2S"]
    w, t, leaves = O.exact(hero, board, 2)
    assert leaves == 45 * 45 * 43
    p = w + t
    for mode in (O.MODE_MT, O.MODE_CTR):
        r = O.run(mode, hero, board, 2, 200000, 5)
        assert abs(r["equity"] - p) < 4.5 * (p * (1 - p) / 200000) ** 0.5
        assert abs(r["tie"] / r["runs"] - t) < 4.5 * (max(t, 1e-4) / 200000) ** 0.5 + 1e-4


def test_uniform_law_exact_and_reference_cpp_expectations():
    """under the UNIFORM law the oracle converges to the exact uniform expectation, and that
    expectation is what the reference's own C++ tests expect (tools/montecarlo_cpp/Test.cpp:176-217: 40.2 % for
    3H 3S on 8S 4S QH 8C 4H heads-up, within 1 %)."""
    hero, board = ["3H", "3S"], ["8S", "4S", "QH", "8C", "4H"]
    w, t, leaves = O.exact(hero, board, 2, uniform=True)
    assert leaves == 45 * 44
    assert abs(100 * (w + t) - 40.2) < 1.0
    r = O.run(O.MODE_CTR_UNIFORM, hero, board, 2, 300000, 3)
    p = w + t
    assert abs(r["equity"] - p) < 4.5 * (p * (1 - p) / 300000) ** 0.5
    # the reference's (biased) law differs measurably on the same query
    wb, tb, _ = O.exact(hero, board, 2)
    assert abs((wb + tb) - p) > 1e-3


def test_reference_cpp_agrees_with_uniform_law():
    """oracle/_ref/ref_mc is the reference's own C++ variant compiled from /root/reference (make -C oracle ref);
    it shuffles uniformly, so it must estimate the exact UNIFORM expectation (and not the Python law's)."""
    import subprocess
    exe = os.path.join(os.path.dirname(O.__file__), "_ref", "ref_mc")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_mc not built (needs /root/reference)")
    hero, board = ["TC", "TH"], ["4D", "QD", "KC", "2S"]
    w, t, _ = O.exact(hero, board, 2, uniform=True)
    out = subprocess.run([exe, "equity", hero[0], hero[1], "2", "40000"] + board, capture_output=True, text=True,
                         check=True).stdout.split()
    eq = float(out[0])
    p = w + t
    assert abs(eq - p) < 5 * (p * (1 - p) / 40000) ** 0.5, (eq, p)


def _ext_range(row):
    """opponent_range of a fixture row -> class strings (None = every class), as run_montecarlo derives it
    (montecarlo_python.py:105-112: the LAST int(169 * r) classes of the equity-sorted list; 0 -> all)."""
    r = row["opponent_range"]
    if isinstance(r, list):
        return r
    with open(os.path.join(os.path.dirname(G), "..", "neuron_poker_amd", "preflop_classes.json")) as f:
        order = json.load(f)
    take = int(169 * r)
    return None if take == 0 or take >= 169 else order[-take:]


def test_ranges_ghost_cards_known_hands_fixture():
    """tests/golden/ext_tallies.json was recorded from seeded reference runs (incl. the inputs of
    tests/test_montecarlo_python.py:215-232)."""
    rows = jload("ext_tallies.json")
    assert len(rows) >= 22 and max(len(t["players"]) for t in rows) >= 4   # incl. several known hands, ranges among them
    for t in rows:
        pl = t["players"]
        r = O.run_ex(O.MODE_MT, pl[0], t["board"], t["n_players"], t["runs"], t["seed"],
                     known=pl[1:], ghost=t["ghost"] or None, opp_range=_ext_range(t))
        assert (r["wins"], r["passes"], r["by_type"]) == (t["wins"], t["passes"], t["by_type"]), t
        if t["passes"] < 20 * t["runs"]:  # the recorder counts MT words modulo 624 per iteration: only valid
            assert r["mt_words"] == t["mt_words"], t  # while an iteration consumes fewer than 624 words


def test_reference_range_tests_statistically():
    # tests/test_montecarlo_python.py:215-232: 12.8 % and 77.8 % within 3 points
    board = ["3D", "9H", "AS", "7S", "QH"]
    rows = jload("ext_tallies.json")
    rng = _ext_range(rows[0])
    for mode in (O.MODE_MT, O.MODE_CTR):
        a = O.run_ex(mode, ["KS", "KC"], board, 3, 30000, 5, opp_range=rng)
        b = O.run_ex(mode, ["AKO", "AA"], board, 3, 30000, 6, opp_range=rng)
        assert abs(100 * a["wins"] / a["runs"] - 12.8) < 3 and abs(100 * b["wins"] / b["runs"] - 77.8) < 3


def test_production_law_of_extended_queries_equals_the_reference_law():
    """The production sampler of extended queries (MCQ-CTR v5x: rejection from a fixed candidate list, no index
    arithmetic) must deal the reference's LAW: on cases small enough to see differences of a few 1e-3, its equity
    agrees with the literal MT19937 walk of the reference's loops within Monte-Carlo noise (both 400k iterations,
    sigma of the difference 1.1e-3; bound 4.5e-3)."""
    cases = [(["TC", "TD"], [["AA", "KK", "QQ", "AKS", "AKO"]], ["2C", "7D", "JC", "JD", "3S"], 3, None),
             (["AA", "AKS", "AKO", "AQO"], [["AS", "KS"]], ["2C", "7D", "JC"], 3, None),
             (["KS", "KC"], [], ["3D", "9H", "AS", "7S", "QH"], 3, _ext_range(jload("ext_tallies.json")[0])),
             (["AH", "KH"], [["QS", "QD"], ["7C", "7D"]], ["2S", "8D", "9C", "TD"], 4, ["AA", "KK", "AKS", "QJS", "T9S", "22"])]
    for hero, known, board, n, rng in cases:
        a = O.run_ex(O.MODE_MT, hero, board, n, 400000, 3, known=known, opp_range=rng)
        b = O.run_ex(O.MODE_CTR, hero, board, n, 400000, 4, known=known, opp_range=rng)
        assert abs(a["wins"] / a["runs"] - b["wins"] / b["runs"]) < 4.5e-3, (hero, known, a["wins"], b["wins"])
        assert abs(sum(a["by_type"][:3]) - sum(b["by_type"][:3])) / a["runs"] < 4.5e-3


def test_single_player_and_invalid():
    r = O.run(O.MODE_MT, ["7H", "2C"], [], 1, 3000, 9)
    assert (r["wins"], r["passes"], r["win"]) == (3000, 0, 3000)
    with pytest.raises(ValueError):
        O.run(O.MODE_MT, ["AH", "AH"], [], 2, 10, 0)
    with pytest.raises(ValueError):
        O.run(O.MODE_MT, ["AH", "KH"], ["AH", "2C", "3C"], 2, 10, 0)
    with pytest.raises(ValueError):
        O.run(O.MODE_MT, ["AH", "KH"], [], 11, 10, 0)
