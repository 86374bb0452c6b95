"""Host-compiled build of the product's lane arithmetic (mcq_device.hpp / mcq_replay.hpp) against the oracle
and the golden fixtures.  This exercises the exact source the gfx950 kernels are built from; the GPU parity
tests (tests/test_gpu_parity.py) repeat the comparisons through the C ABI on the device."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import hostsim as H

G = os.path.join(os.path.dirname(__file__), "golden")


def q16(hero, board, n, runs):
    b = [255] * 5
    for i, c in enumerate(board):
        b[i] = O.card_id(c)
    return O.pack_queries([[O.card_id(hero[0]), O.card_id(hero[1])]], [b], n, runs)[0]


def test_select_pop_matches_list_pop():
    g = np.random.default_rng(1)
    for _ in range(300):
        deck = sorted(g.choice(52, g.integers(2, 53), replace=False).tolist())
        m = sum(1 << c for c in deck)
        dlo, dhi = m & 0xffffffff, m >> 32
        for _ in range(min(len(deck) - 1, 25)):
            k = int(g.integers(0, len(deck)))
            pos, dlo, dhi = H.select_pop(dlo, dhi, k)
            assert pos == deck.pop(k)
            assert dlo | dhi << 32 == sum(1 << c for c in deck)


def test_key_category_matches_reference_fixture():
    z = np.load(os.path.join(G, "evaluator_hands.npz"))
    keys = H.eval7(z["cards"])
    assert np.array_equal(H.key_type(keys), z["category"])


def test_key_order_matches_reference_tuple_order():
    # the 32-bit key must order hands exactly like Python orders _calc_score tuples
    z = np.load(os.path.join(G, "evaluator_hands.npz"))
    cards, cat, nr, ranks = z["cards"], z["category"], z["n_ranks"], z["card_ranks"]
    keys = H.eval7(cards)
    tup = [(int(cat[i]), tuple(int(x) for x in ranks[i, :nr[i]])) for i in range(len(cards))]
    # category order == score order (hand_evaluator.py:83-115), so (category, card_ranks) is the reference order
    g = np.random.default_rng(2)
    order = np.argsort(keys, kind="stable")
    for a, b in zip(order[:-1], order[1:]):  # adjacent in key order -> strongest test of the total order
        if keys[a] == keys[b]:
            assert tup[a] == tup[b], (cards[a], cards[b])
        else:
            assert tup[a] < tup[b], (cards[a], cards[b], tup[a], tup[b])
    for _ in range(20000):
        a, b = g.integers(0, len(cards), 2)
        ka, kb = int(keys[a]), int(keys[b])
        assert (ka > kb) - (ka < kb) == (tup[a] > tup[b]) - (tup[a] < tup[b])


def test_showdowns_fixture_with_keys():
    z = np.load(os.path.join(G, "showdowns.npz"))
    hands, n, win, wt = z["hands"], z["n_players"], z["winner"], z["winner_type"]
    for i in range(len(hands)):
        k = H.eval7(hands[i, :n[i]])
        w = int(np.argmax(k))  # first of the maxima
        assert (w, H.key_type(k[w])) == (win[i], wt[i]), i


def test_reference_evaluator_cases_without_duplicates():
    with open(os.path.join(G, "evaluator_cases.json")) as f:
        cases = json.load(f)
    for c in cases:
        ids = [[O.card_id(x) for x in h] for h in c["hands"]]
        if any(len(set(h)) != 7 for h in ids):
            continue  # duplicate cards (tests/test_evaluator.py:27,63) are outside the kernel's domain
        k = H.eval7(ids)
        w = int(np.argmax(k))
        assert w == c["winner"] and O.TYPES[H.key_type(k[w])] == c["winner_type"], c


def test_mt_and_philox_restated_identically():
    assert np.array_equal(H.mt_words(12345, 3000), O.mt_words(12345, 3000))
    assert list(H.philox([1, 2, 3, 4], [5, 6])) == list(O.philox4x32_10([1, 2, 3, 4], [5, 6]))


def test_replay_mode_equals_reference_tallies():
    with open(os.path.join(G, "tallies.json")) as f:
        rows = json.load(f)
    for t in rows:
        if t["runs"] > 20000:
            continue
        r = H.run_replay(q16(t["hero"], t["board"], t["n_players"], t["runs"]), t["seed"] & 0xffffffff)
        assert int(r[0]) == t["runs"] and int(r[1]) == t["passes"]
        assert int(r[2] + r[3]) == t["wins"], t
        assert [int(x) for x in r[4:]] == t["by_type"], t


@pytest.mark.parametrize("hero,board,n,runs", [
    (["AH", "KH"], [], 2, 5000), (["AH", "KH"], [], 6, 3001), (["2C", "7D"], ["AS", "KS", "QS"], 6, 2000),
    (["TC", "TH"], ["4D", "QD", "KC", "2S"], 3, 4097), (["3H", "3S"], ["8S", "4S", "QH", "8C", "4H"], 10, 1000),
    (["7H", "2C"], [], 1, 777), (["AS", "AC"], [], 10, 1500), (["9D", "9C"], ["9H"] * 0, 4, 15)])
def test_ctr_mode_equals_oracle_ctr(hero, board, n, runs):
    q = q16(hero, board, n, runs)
    for seed, qid in [(0, 0), (0xDEADBEEFCAFE, 7), (2 ** 64 - 1, 2 ** 40 + 3)]:
        got = H.run_ctr(q, seed, qid)
        exp = O.run(O.MODE_CTR, hero, board, n, runs, seed, qid=qid)["tallies"]
        assert np.array_equal(got, exp), (seed, qid, got, exp)


def test_straight_line_iterations_equal_the_general_form():
    """mcq_iterations (what the bulk kernel runs): for 1-6 opponents and 5 / 2 / 1 table cards to come the loop body is a
    branch-free specialisation of mcq_iteration; every one of them (and the general fall-backs: river, 7-9 opponents,
    hero alone) gives the general form's tallies, and the oracle's."""
    g = np.random.default_rng(99)
    for n_players in range(1, 11):
        for nb in (0, 3, 4, 5):
            c = g.permutation(52)[:2 + nb]
            hero = [O.card_str(int(x)) for x in c[:2]]
            board = [O.card_str(int(x)) for x in c[2:]]
            q = q16(hero, board, n_players, 333)
            a = H.run_ctr(q, 11, n_players * 7 + nb)
            b = H.run_ctr(q, 11, n_players * 7 + nb, general=True)
            assert np.array_equal(a, b), (n_players, nb)
            if nb in (0, 4):
                exp = O.run(O.MODE_CTR, hero, board, n_players, 333, 11, qid=n_players * 7 + nb)["tallies"]
                assert np.array_equal(a, exp), (n_players, nb)


def test_numpy_stream_coupling_matches_reference_sequence():
    """consecutive calls share numpy's global state; after each call np.random is exactly where the
    reference leaves it (tests/golden/sequence.json was recorded from the reference)."""
    with open(os.path.join(G, "sequence.json")) as f:
        seqs = json.load(f)
    for s in seqs:
        np.random.seed(s["seed"])
        for c in s["calls"]:
            r = H.run_replay_numpy_stream(q16(c["hero"], c["board"], c["n_players"], c["runs"]))
            assert int(r[2] + r[3]) == c["wins"] and int(r[1]) == c["passes"], c
            assert [int(x) for x in r[4:]] == c["by_type"], c
            assert int(np.random.randint(0, 52)) == c["randint52_after"]
        assert [int(x) for x in np.random.randint(0, 2 ** 32, size=4, dtype=np.uint32)] == s["next_words"]


@pytest.mark.parametrize("hero,board,n,runs", [(["AH", "KH"], [], 6, 3001), (["TC", "TH"], ["4D", "QD", "KC", "2S"], 3, 4097),
                                               (["AS", "AC"], [], 10, 1500), (["7H", "2C"], ["2D", "2S", "AC", "KD", "QD"], 2, 999)])
def test_uniform_law_equals_oracle(hero, board, n, runs):
    """the opt-in unbiased dealing law, lane arithmetic == oracle."""
    q = q16(hero, board, n, runs)
    got = H.run_ctr(q, 77, 5, uniform=True)
    exp = O.run(O.MODE_CTR_UNIFORM, hero, board, n, runs, 77, qid=5)["tallies"]
    assert np.array_equal(got, exp)


def _ext_case(t):
    """fixture row of tests/golden/ext_tallies.json -> (query16, ext record, oracle kwargs)"""
    import neuron_poker_amd as npa
    pl = t["players"]
    r = t["opponent_range"]
    if isinstance(r, list):
        opp = r
    else:
        with open(os.path.join(os.path.dirname(G), "..", "neuron_poker_amd", "preflop_classes.json")) as f:
            order = json.load(f)
        take = int(169 * r)
        opp = None if take == 0 or take >= 169 else order[-take:]
    hero_cards = ["2C", "2D"] if t["hero_is_range"] else pl[0]
    q = q16(hero_cards, t["board"], t["n_players"], t["runs"])
    if t["hero_is_range"]:
        q = q.copy()
        q[0:2] = 0
    known = [[O.card_id(c) for c in h] if O._is_cards(h) else npa.range_bits(h) for h in pl[1:]]
    ext = npa.pack_query_ext(1, ghost=[O.card_id(c) for c in t["ghost"]] if t["ghost"] else None, known=known,
                             hero_range=npa.range_bits(pl[0]) if t["hero_is_range"] else None,
                             opp_range=npa.range_bits(opp) if opp is not None else None)
    kw = dict(known=pl[1:], ghost=t["ghost"] or None, opp_range=opp)
    return q, ext, kw


def test_extended_queries_replay_equals_reference_and_ctr_equals_oracle():
    """ranges, ghost cards, any number of known hands (cards or ranges) through the product's lane code:
    parity mode == the reference's own seeded runs, production mode == the oracle's implementation of MCQ-CTR v5x."""
    with open(os.path.join(G, "ext_tallies.json")) as f:
        rows = json.load(f)
    for t in rows:
        q, ext, kw = _ext_case(t)
        r = H.run_ext(True, q, ext, t["seed"])
        assert (int(r[2] + r[3]), int(r[1]), [int(x) for x in r[4:]]) == (t["wins"], t["passes"], t["by_type"]), t
        small = q.copy()
        small[12:16] = np.array([700], "<u4").view(np.uint8)
        got = H.run_ext(False, small, ext, 99, qid=4)
        exp = O.run_ex(O.MODE_CTR, t["players"][0], t["board"], t["n_players"], 700, 99, qid=4, **kw)["tallies"]
        assert np.array_equal(got, exp), t


# ---- exact enumeration (csrc/mcq_exact.hpp): the set-based weighting against the oracle's literal walk of the
# dealing tree (every accepted index pair, every table draw), both laws, 1-3 players
EXACT_CASES = [(["AH", "KH"], ["2C", "7D", "9S", "JH", "QC"], 2), (["AH", "KH"], ["2C", "7D", "9S", "JH"], 2),
               (["AS", "KS"], ["2C", "7D", "9S", "JH"], 2),        # hero holds the deck's top card
               (["2C", "2D"], ["AS", "KS", "QH", "3C"], 2),        # top cards on the table
               (["AS", "AD"], ["2C", "7D", "9S"], 2), (["7C", "2D"], ["AS", "KS", "AH"], 2),
               (["AH", "KH"], ["2C", "7D", "9S", "JH", "QC"], 3), (["2C", "2D"], ["AS", "KS", "QS", "JS", "9D"], 3),
               (["AS", "KD"], ["2C", "7D", "9S", "JH", "QC"], 1), (["AS", "KD"], ["2C", "7D", "9S", "JH"], 1),
               (["AS", "KD"], ["2C", "7D", "9S"], 1)]


def _xq(hero, board, n):
    b = [O.card_id(c) for c in board]
    return O.pack_queries([[O.card_id(c) for c in hero]], [b + [255] * (5 - len(b))], n, 1)[0]


@pytest.mark.parametrize("hero,board,n", EXACT_CASES)
@pytest.mark.parametrize("uniform", [False, True])
def test_exact_enumeration_equals_the_oracles_tree_walk(hero, board, n, uniform):
    g = H.exact(_xq(hero, board, n), uniform)
    win, tie, _ = O.exact(hero, board, n, uniform)
    tot = int(g[0])
    assert abs(int(g[2]) / tot - win) < 1e-9 and abs(int(g[3]) / tot - tie) < 1e-9   # the oracle sums doubles
    assert int(g[4:].sum()) == int(g[2] + g[3]) and g[1] == 0
    L, k = 50 - len(board), 5 - len(board)
    from math import comb
    if uniform:
        want = comb(L, k) * [1, 990, 990 * 903][n - 1]
    else:  # accepted index pairs per opponent x completions without the highest remaining card
        want = [1, (L - 1) ** 2, (L - 1) ** 2 * (L - 3) ** 2][n - 1] * (comb(L - 2 * (n - 1) - 1, k) if k else 1)
    assert tot == want


def test_exact_enumeration_three_players_on_the_turn():
    # 137 M leaves in the oracle's walk (most of a minute); the product's weighting visits 46 completions
    hero, board = ["TC", "TH"], ["4D", "QD", "KC", "2S"]
    for uniform in (False,):
        g = H.exact(_xq(hero, board, 3), uniform)
        win, tie, _ = O.exact(hero, board, 3, uniform)
        assert abs(int(g[2]) / int(g[0]) - win) < 1e-9 and abs(int(g[3]) / int(g[0]) - tie) < 1e-9


def test_wave_cooperative_mt19937_parse_equals_the_sequential_walk():
    """mcq_mt.hpp -- the device's parse of numpy's stream (64 words at a time, fixed-point iteration over the lanes'
    positions, re-drawn pairs rewound, 64-iteration flushes) compiled with the lanes as arrays: accepted draws and
    `passes` must equal the sequential walk of mcq_replay.hpp (itself pinned to the reference's deal traces) byte for
    byte; the state words must be numpy's."""
    assert H.mt_magic_ok()
    for s in (0, 1, 5489, 12345, 2 ** 32 - 1):
        w = H.mt_wave_words(s, 2600)                      # four regenerations
        assert np.array_equal(w, H.mt_words(s, 2600))
        rs = np.random.RandomState(s)
        assert np.array_equal(w[:50], rs.randint(0, 2 ** 32, 50, dtype=np.uint64).astype(np.uint32))
    g = np.random.default_rng(20261004)
    for t in range(400):
        nb = int(g.choice([0, 3, 4, 5]))
        npl = int(g.integers(1, 11))
        runs = int(g.choice([1, 2, 63, 64, 65, 127, 128, 129, 1000, 4097]))
        c = g.permutation(52)[:2 + nb]
        q = O.pack_queries([c[:2]], [list(c[2:]) + [255] * (5 - nb)], npl, runs)[0]
        seed = int(g.integers(0, 2 ** 32))
        d1, p1 = H.mt_parse(q, seed)
        d2, p2 = H.mt_parse(q, seed, reference=True)
        assert p1 == p2 and np.array_equal(d1, d2), (nb, npl, runs, seed)


def test_block_parallel_mt19937_walk_equals_the_sequential_walk():
    """mcq_mt_blocks.hpp -- the stream of ONE query parsed with its 624-word state blocks side by side (every block
    scanned from every entry position by a sequential automaton, the exits stitched, every block then parsed from its
    true entry with the wave-cooperative batch code, draws stored by the second index of a pair): draws and `passes`
    must equal the sequential walk byte for byte, for every number of players (zone 31 with eight and nine opponents),
    also with exactly as many blocks as the stream touches; too few blocks are reported, not mis-parsed; the host's
    estimate of the blocks suffices."""
    g = np.random.default_rng(20261005)
    for t in range(120):
        nb = int(g.choice([0, 3, 4, 5]))
        npl = int(g.integers(1, 11)) if t >= 20 else 1 + t % 10
        runs = int(g.choice([1, 40, 129, 700, 3000, 9000]))
        if npl == 1 and nb == 5:
            nb = 3   # (a query that draws nothing has no stream)
        c = g.permutation(52)[:2 + nb]
        q = O.pack_queries([c[:2]], [list(c[2:]) + [255] * (5 - nb)], npl, runs)[0]
        seed = int(g.integers(0, 2 ** 32))
        d2, p2 = H.mt_parse(q, seed, reference=True)
        d1, p1 = H.mt_parse_blocks(q, seed)
        assert p1 == p2 and np.array_equal(d1, d2), (nb, npl, runs, seed)
        # the words the stream consumes -> the blocks it touches: exactly those suffice, one fewer does not
        lo = H.mtb_blocks_needed(q)   # (the estimate sufficed above; it carries a margin of eight blocks)
        while lo > 1 and H.mt_parse_blocks(q, seed, lo - 1)[1] is not None:
            lo -= 1
        d3, p3 = H.mt_parse_blocks(q, seed, lo)
        assert p3 == p2 and np.array_equal(d3, d2), (nb, npl, runs, seed, lo)
        assert lo == 1 or H.mt_parse_blocks(q, seed, lo - 1)[1] is None


def test_wave_cooperative_walk_of_extended_queries_equals_the_sequential_walk():
    """mcq_mt_ext.hpp -- the device's walk of numpy's stream through the reference's loops over ranges, ghost cards
    and known hands (stage by stage: every attempt of a batch of 64 words tested at once, first success ends the stage,
    deck shifted in LDS) compiled with the lanes as arrays: accepted draws (in list.pop order) and `passes` must equal
    the literal sequential walk of mcq_replay.hpp (which the reference's own runs pin: tests/golden/ext_tallies.json)."""
    from neuron_poker_amd import _lib
    ranks = "23456789TJQKA"
    classes = [a + a for a in ranks] + [ranks[i] + ranks[j] + t for i in range(13) for j in range(i) for t in "SO"]
    g = np.random.default_rng(20261005)

    def some_range(lo, hi):
        return sorted(g.choice(classes, size=int(g.integers(lo, hi)), replace=False))

    for t in range(250):
        nb = int(g.choice([0, 3, 4, 5]))
        c = [int(x) for x in g.permutation(52)[:14 + nb]]
        hero_range = some_range(5, 90) if g.random() < 0.3 else None
        opp = some_range(10, 169) if g.random() < 0.6 else None
        ghost = c[2:4] if g.random() < 0.3 else None
        n_known = int(g.choice([0, 0, 1, 1, 2, 3, 4, 9]))
        known = [some_range(10, 100) if g.random() < 0.35 else c[4 + 2 * k:6 + 2 * k] for k in range(min(n_known, 5))]
        npl = int(g.integers(max(2, 1 + len(known)), 11))
        runs = int(g.choice([1, 2, 63, 64, 65, 129, 300]))
        q = _lib.pack_queries([[0, 1] if hero_range else c[:2]], [c[14:] + [255] * (5 - nb)], npl, runs)
        if hero_range:
            q["hole"][0] = 0
        e = _lib.pack_query_ext(1, ghost=ghost, known=[h if isinstance(h[0], int) else _lib.range_bits(h) for h in known],
                                hero_range=_lib.range_bits(hero_range) if hero_range else None,
                                opp_range=_lib.range_bits(opp) if opp else None)
        seed = int(g.integers(0, 2 ** 32))
        q16 = q.view(np.uint8).reshape(16)
        d1, p1 = H.mt_parse_ext(q16, e, seed)
        d2, p2 = H.mt_parse_ext(q16, e, seed, reference=True)
        assert p1 == p2 and np.array_equal(d1, d2), (nb, npl, runs, seed, hero_range, opp, known, ghost)


def test_one_launch_layout_invariants():
    """mcq_layout.hpp (host side of mcq_eval_direct_kernel): every query owns 2^lg consecutive, size-aligned waves of
    ONE block and round with cut numbers 0 .. 2^lg - 1 in order; no slot is used twice; waves follow cost (within a
    factor of two of the query's share, at most 16); blocks carry about the same number of waves; the slot count
    stays inside the bound the host reserves for it."""
    g = np.random.default_rng(3)
    for n, n_cu, max_lg in [(1, 256, 4), (2, 256, 4), (17, 256, 4), (257, 256, 4), (1000, 256, 4), (1024, 256, 2), (5000, 256, 4),
                            (70000, 256, 4), (300, 8, 4), (64, 256, 0)]:
        cost = g.integers(1, 40, n).astype(np.uint64) * g.choice([1, 1, 1, 8], n).astype(np.uint64) * 1000
        cost[g.random(n) < 0.05] = 0
        if cost.sum() == 0:
            cost[0] = 5
        grid, rounds, lg, qi, sub = H.direct_layout(cost, n_cu, max_lg)
        assert grid == min(n, n_cu) and len(qi) == rounds * grid * 16 <= n + 96 * n_cu + 64
        used = qi != 0xFFFFFFFF
        assert np.array_equal(np.bincount(qi[used], minlength=n), 1 << lg.astype(np.int64))      # 2^lg slots each
        first = {}
        for k in np.nonzero(used)[0]:
            first.setdefault(int(qi[k]), int(k))
        for i, k in first.items():
            w = 1 << int(lg[i])
            assert k % w == 0 and k // 16 == (k + w - 1) // 16                   # aligned, inside one block-round
            assert np.all(qi[k:k + w] == i) and np.array_equal(sub[k:k + w], np.arange(w))
        assert lg.max() <= max_lg
        share = cost.astype(np.float64) * 16 * n_cu / cost.sum()
        want = np.clip(np.floor(np.log2(np.maximum(share, 1))), 0, max_lg)
        assert np.array_equal(lg[cost > 0], want[cost > 0].astype(np.uint8))
        per_block = np.zeros(grid, np.int64)
        blocks = (np.arange(len(qi)) // 16) % grid
        np.add.at(per_block, blocks[used], 1)
        assert per_block.max() - per_block.min() <= 31


def test_one_launch_work_records_fit_their_slots():
    """mcq_direct_write_records (mcq_layout.hpp): what the library hands mcq_eval_direct_kernel -- per wave slot the query's
    record with log2(waves) and the cut number in its reserved bytes, and the query index -- written into buffers of
    EXACTLY the layout's size (under tests/sanitize_cpu.sh a write past them is an AddressSanitizer finding); a buffer
    one slot short is refused, nothing written."""
    g = np.random.default_rng(11)
    for n, n_cu, max_lg in [(1, 256, 3), (7, 256, 3), (100, 256, 3), (1024, 256, 3), (3000, 256, 2), (5, 8, 4)]:
        cost = g.integers(1, 40, n).astype(np.uint64) * 1000
        q = O.pack_queries(g.integers(0, 52, (n, 2)), np.full((n, 5), 255), g.integers(1, 11, n), g.integers(1, 8000, n))
        rec, qi = H.direct_records(cost, q, n_cu, max_lg)
        grid, rounds, lg, slot_qi, slot_sub = H.direct_layout(cost, n_cu, max_lg)
        assert len(qi) == rounds * grid * 16 and np.array_equal(qi, slot_qi)
        used = qi != 0xFFFFFFFF
        assert np.array_equal(rec[used][:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 13, 14, 15]],
                              q[qi[used]][:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 13, 14, 15]])
        assert np.array_equal(rec[used][:, 9], lg[qi[used]]) and np.array_equal(rec[used][:, 10], slot_sub[used])
        assert H.direct_records(cost, q, n_cu, max_lg, cap=len(qi) - 1) is None
