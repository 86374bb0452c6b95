#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Only runnable in the build container (needs /root/reference); the fixtures it writes are
plain data (card ids, integer tallies, category ids) and are committed, the reference's
source never is.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen_golden.py

What is captured (ids follow SURVEY.md section 8c):
  F1  evaluator_hands.npz   7-card hands -> (category, card_ranks) from tools/hand_evaluator.py:_calc_score
      evaluator_cases.json  the 14 cases of tests/test_evaluator.py + constructed quirk cases
      showdowns.npz         N-hand showdowns -> winner index from tools/hand_evaluator.py:eval_best_hand
  F2  deal_traces.npz       per (seed, hero, board, N): first K iterations' dealt cards, MT words/iteration
  F3  tallies.json          seeded run_montecarlo tallies (wins, passes, per-type wins, MT words)
  F4  stat_expectations.json the (hero, board, N, expected %) rows of tests/test_montecarlo_python.py
  F7  ext_tallies.json      seeded runs with opponent ranges, hero ranges, ghost cards, two known hands (8f-2);
      ../../neuron_poker_amd/preflop_classes.json  the 169 classes in the reference's sort order
  F6  sequence.json         consecutive calls on ONE seeded numpy stream, with a randint in between (8f-4)
"""
import json
import os
import sys
import time

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from tools import hand_evaluator as he  # noqa: E402
from tools import montecarlo_python as mp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
RANKS = "23456789TJQKA"
SUITS = "CDHS"
TYPES = ["HighCard", "Pair", "TwoPair", "ThreeOfAKind", "Straight", "Flush", "FullHouse", "FoufOfAKind",
         "StraightFlush"]
FAR = time.time() + 1e9


def cid(s):
    return 4 * RANKS.index(s[0]) + SUITS.index(s[1])


def cstr(c):
    return RANKS[c >> 2] + SUITS[c & 3]


def score_of(cards):
    score, ranks, typ = he._calc_score([cstr(c) for c in cards])
    return TYPES.index(typ), list(ranks), list(score)


# ----------------------------------------------------------------------------- F1
def sample_hands(g):
    """Uniform hands plus sub-deck hands so that rare categories / quirks are well covered."""
    out = []
    for _ in range(30000):
        out.append(g.choice(52, 7, replace=False))
    for _ in range(8000):  # two suits only -> many flushes / straight flushes
        s = g.choice(4, 2, replace=False)
        sub = np.array([4 * r + x for r in range(13) for x in s])
        out.append(g.choice(sub, 7, replace=False))
    for _ in range(8000):  # window of 6 consecutive ranks (ace-low wrap) -> straights, quads, full houses
        lo = g.integers(-1, 8)
        rr = [(lo + k) % 13 for k in range(6)]
        sub = np.array([4 * r + x for r in rr for x in range(4)])
        out.append(g.choice(sub, 7, replace=False))
    for _ in range(4000):  # one suit + 10 random other cards -> straight flushes with 6/7 suited cards
        s = g.integers(0, 4)
        suited = [4 * r + s for r in range(13)]
        others = [c for c in range(52) if c & 3 != s]
        sub = np.array(suited + list(g.choice(others, 6, replace=False)))
        out.append(g.choice(sub, 7, replace=False))
    return np.array(out, dtype=np.uint8)


def gen_evaluator(g):
    hands = sample_hands(g)
    m = len(hands)
    cat = np.zeros(m, np.uint8)
    nr = np.zeros(m, np.uint8)
    ranks = np.full((m, 8), -128, np.int8)
    for i, h in enumerate(hands):
        c, r, _ = score_of(h)
        cat[i] = c
        nr[i] = len(r)
        ranks[i, :len(r)] = r
    np.savez_compressed(os.path.join(HERE, "evaluator_hands.npz"), cards=hands, category=cat, n_ranks=nr,
                        card_ranks=ranks)
    print("evaluator_hands:", m, "categories", np.bincount(cat, minlength=9))


EVAL_CASES = [  # (hands, expected winner index) -- tests/test_evaluator.py:9-133
    ([['3H', '3S', '4H', '4S', '8S', '8C', 'QH'], ['KH', '6C', '4H', '4S', '8S', '8C', 'QH']], 1),
    ([['8H', '8D', 'QH', '7H', '9H', 'JH', 'TH'], ['KH', '6C', 'QH', '7H', '9H', 'JH', 'TH']], 1),
    ([['AS', 'KS', 'TS', '9S', '7S', '2H', '2H'], ['AS', 'KS', 'TS', '9S', '8S', '2H', '2H']], 1),
    ([['8S', 'TS', '8H', 'KS', '9S', 'TH', 'KH'], ['TD', '7S', '8H', 'KS', '9S', 'TH', 'KH']], 0),
    ([['2D', '2H', 'AS', 'AD', 'AH', '8S', '7H'], ['7C', '7S', '7H', 'AD', 'AS', '8S', '8H']], 0),
    ([['7C', '7S', '7H', 'AD', 'KS', '5S', '8H'], ['2D', '3H', 'AS', '4D', '5H', '8S', '7H']], 1),
    ([['7C', '7C', 'AC', 'AC', '8C', '8S', '7H'], ['2C', '3C', '4C', '5C', '6C', '8S', 'KH']], 1),
    ([['AC', 'JS', 'AS', '2D', '5H', '3S', '3H'], ['QD', 'JD', 'TS', '9D', '6H', '8S', 'KH'],
      ['2D', '3D', '4S', '5D', '6H', '8S', 'KH']], 1),
    ([['7C', '5S', '3S', 'JD', '8H', '2S', 'KH'], ['AD', '3D', '4S', '5D', '9H', '8S', 'KH']], 1),
    ([['2C', '2D', '4S', '4D', '4H', '8S', 'KH'], ['7C', '7S', '7D', '7H', '8H', '8S', 'JH']], 1),
    ([['7C', '5S', '3S', 'JD', '8H', '2S', 'KH'], ['AD', '3D', '3S', '5D', '9H', '8S', 'KH']], 1),
    ([['7H', '7S', '3S', 'JD', '8H', '2S', 'KH'], ['7D', '3D', '3S', '7C', '9H', '8S', 'KH']], 1),
    ([['AS', '8H', 'TS', 'JH', '3H', '2H', 'AH'], ['QD', 'QH', 'TS', 'JH', '3H', '2H', 'AH']], 1),
    ([['9S', '7H', 'KS', 'KH', 'AH', 'AS', 'AC'], ['8D', '2H', 'KS', 'KH', 'AH', 'AS', 'AC']], 0),
]

QUIRK_CASES = [  # constructed: SURVEY.md 8a-E Q1..Q4 (hole cards first, then the shared board)
    # Q1 quads: tie-break is the two highest distinct ranks of all seven cards
    [['KC', 'QD', '5C', '5D', '5H', '5S', '2C'], ['KD', 'JD', '5C', '5D', '5H', '5S', '2C']],
    [['9C', '9D', '9H', '9S', '5C', '5D', 'KH'], ['5H', '5S', '9H', '9S', '5C', '5D', 'KH']],
    [['AS', '9S', '7H', 'AH', '7C', '7S', '7D'], ['JD', '9H', '7H', 'AH', '7C', '7S', '7D']],
    # Q2 straight flush: all suited ranks count, -1 when the suit holds the ace
    [['AH', '2C', '2H', '3H', '4H', '5H', '9D'], ['6H', '2D', '2H', '3H', '4H', '5H', '9D']],
    [['KH', '2C', '4H', '5H', '6H', '7H', '8H'], ['9H', '2D', '4H', '5H', '6H', '7H', '8H']],
    [['2H', '2C', '3H', '4H', '5H', '6H', '7H'], ['8H', '2D', '3H', '4H', '5H', '6H', '7H']],
    [['AH', 'KH', '2H', '3H', '4H', '5H', 'QH'], ['6H', '7H', '2H', '3H', '4H', '5H', 'QH']],
    # Q3 flush + off-suit straight -> Flush
    [['9H', '8D', '7H', '6H', '5C', '2H', 'KH'], ['9D', '8C', '7H', '6H', '5C', '2H', 'KH']],
    # wheel vs six-high straight, ace-high straight, broadway with pair
    [['AC', '2D', '3H', '4S', '5C', '9D', 'KH'], ['6C', '2D', '3H', '4S', '5C', '9D', 'KH']],
    [['AC', 'KD', 'QH', 'JS', 'TC', 'TD', '2H'], ['9C', 'KD', 'QH', 'JS', 'TC', 'TD', '2H']],
    # three pair, two trips, trips + two pairs
    [['AC', 'AD', 'KC', 'KD', 'QC', 'QD', '2H'], ['AH', 'AS', 'KC', 'KD', 'QC', 'QD', 'JH']],
    [['AC', 'AD', 'AH', 'KC', 'KD', 'KH', '2S'], ['KS', 'QS', 'AH', 'KC', 'KD', 'KH', '2S']],
    [['AC', 'AD', 'AH', 'KC', 'KD', 'QH', 'QS'], ['KH', 'KS', 'AH', 'KC', 'KD', 'QH', 'QS']],
    # identical ranks, different suits -> exact tie, first hand wins
    [['AC', 'KD', '2H', '5S', '9C', 'JD', 'QH'], ['AD', 'KC', '2H', '5S', '9C', 'JD', 'QH']],
]


def gen_cases():
    cases = []
    for hands, exp in EVAL_CASES:
        best, typ = he.eval_best_hand(hands)
        assert hands.index(best) == exp
        cases.append({"source": "tests/test_evaluator.py", "hands": hands, "winner": exp, "winner_type": typ,
                      "scores": [list(map(lambda t: list(t) if isinstance(t, tuple) else t, he._calc_score(h)))
                                 for h in hands]})
    for hands in QUIRK_CASES:
        best, typ = he.eval_best_hand(hands)
        cases.append({"source": "constructed", "hands": hands, "winner": hands.index(best), "winner_type": typ,
                      "scores": [list(map(lambda t: list(t) if isinstance(t, tuple) else t, he._calc_score(h)))
                                 for h in hands]})
    with open(os.path.join(HERE, "evaluator_cases.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print("evaluator_cases:", len(cases))


def gen_showdowns(g):
    rows = []
    for k in range(24000):
        n = int(g.integers(2, 11))
        mode = k % 4
        if mode == 0:
            sub = np.arange(52)
        elif mode == 1:  # two suits + a few
            s = g.choice(4, 2, replace=False)
            sub = np.array([c for c in range(52) if (c & 3) in s])
            n = min(n, 10)
        elif mode == 2:  # 8 consecutive ranks
            lo = g.integers(-1, 6)
            rr = [(lo + j) % 13 for j in range(8)]
            sub = np.array([4 * r + x for r in rr for x in range(4)])
        else:  # one full suit + 14 others
            s = g.integers(0, 4)
            others = [c for c in range(52) if c & 3 != s]
            sub = np.array([4 * r + s for r in range(13)] + list(g.choice(others, 14, replace=False)))
        need = 5 + 2 * n
        if need > len(sub):
            n = (len(sub) - 5) // 2
            need = 5 + 2 * n
        cards = g.choice(sub, need, replace=False)
        board = list(cards[:5])
        hands = [[int(cards[5 + 2 * i]), int(cards[6 + 2 * i])] + [int(b) for b in board] for i in range(n)]
        shands = [[cstr(c) for c in h] for h in hands]
        best, typ = he.eval_best_hand(shands)
        rows.append((n, hands, shands.index(best), TYPES.index(typ)))
    S = len(rows)
    arr = np.full((S, 10, 7), 255, np.uint8)
    nn = np.zeros(S, np.uint8)
    win = np.zeros(S, np.uint8)
    wt = np.zeros(S, np.uint8)
    for i, (n, hands, w, t) in enumerate(rows):
        nn[i] = n
        arr[i, :n] = np.array(hands, np.uint8)
        win[i] = w
        wt[i] = t
    np.savez_compressed(os.path.join(HERE, "showdowns.npz"), hands=arr, n_players=nn, winner=win, winner_type=wt)
    print("showdowns:", S, "winner types", np.bincount(wt, minlength=9))


# ----------------------------------------------------------------------------- F2 / F3
class Recorder:
    """Wraps montecarlo_python.eval_best_hand to capture every iteration's dealt hands and MT position."""

    def __init__(self, keep):
        self.keep = keep
        self.hands = []
        self.words = []
        self.total_words = 0
        self.pos = None
        self.orig = mp.eval_best_hand

    def start(self):
        self.pos = np.random.get_state()[2]

    def __call__(self, hands):
        pos = np.random.get_state()[2]
        d = (pos - self.pos) % 624
        self.pos = pos
        self.total_words += d
        if len(self.hands) < self.keep:
            self.hands.append([[cid(c) for c in h] for h in hands])
            self.words.append(d)
        return self.orig(hands)


def run_ref(hero, board, n, runs, seed, keep=0):
    rec = Recorder(keep)
    mp.eval_best_hand = rec
    try:
        np.random.seed(seed)
        rec.start()
        sim = mp.MonteCarlo()
        sim.run_montecarlo([list(hero)], list(board), n, 1, maxRuns=runs, timeout=FAR, ghost_cards='',
                           opponent_range=1)
    finally:
        mp.eval_best_hand = rec.orig
    by_type = {t: 0 for t in TYPES}
    for k, v in sim.winnerCardTypeList.items():
        by_type[k] = int(round(v * sim.runs))
    wins = int(round(sim.equity * sim.runs))
    assert sum(by_type.values()) == wins
    return {"hero": list(hero), "board": list(board), "n_players": int(n), "runs": int(sim.runs), "seed": int(seed),
            "wins": wins, "passes": int(sim.passes), "by_type": [by_type[t] for t in TYPES],
            "mt_words": int(rec.total_words)}, rec


TALLY_GRID = [  # (hero, board, n_players, runs, seed)
    (['AH', 'KH'], [], 2, 10000, 0), (['AH', 'KH'], [], 2, 10000, 1), (['AH', 'KH'], [], 2, 10000, 12345),
    (['AH', 'KH'], [], 2, 100000, 0), (['AH', 'KH'], [], 2, 100000, 1),
    (['AH', 'KH'], [], 3, 20000, 0), (['AH', 'KH'], [], 6, 20000, 0),
    (['2C', '7D'], ['AS', 'KS', 'QS'], 6, 20000, 0),
    (['TC', 'TH'], ['4D', 'QD', 'KC', '2S'], 3, 20000, 5),
    (['3H', '3S'], ['8S', '4S', 'QH', '8C', '4H'], 2, 20000, 7),
    (['8S', 'TS'], [], 5, 20000, 3), (['5C', 'JS'], [], 4, 20000, 11),
    (['AS', 'AC'], [], 10, 10000, 2), (['2C', '2D'], ['2H', '2S', 'AS'], 10, 10000, 4),
    (['AS', 'KS'], ['QS', 'JS', 'TS', '2C'], 9, 5000, 6),
    (['7H', '2C'], [], 1, 3000, 9),
    (['JD', 'JS'], ['8C', 'TC', 'JC', '5H', 'QC'], 3, 20000, 4294967295),
    (['AD', 'AS'], ['AC', 'AH', 'KD'], 2, 5000, 2 ** 31),
    # round 2: the player counts the straight-line kernels specialise on (1-7 opponents) and the general fall-back (8),
    # at every stage of the table
    (['QC', 'QD'], [], 7, 8000, 21), (['9H', '8H'], ['7H', '6C', '2D'], 7, 8000, 22),
    (['KD', 'QD'], [], 8, 8000, 23), (['AC', '2C'], ['KC', '7C', '3S', '9D'], 8, 6000, 24),
    (['4S', '4D'], ['4C', 'JH', 'JD', '9S', '2H'], 7, 6000, 25), (['TD', '9D'], ['8D', '7D', 'AS', 'KS'], 4, 8000, 26),
    (['6C', '5C'], ['AH', 'KH', 'QH', 'JH', 'TH'], 6, 6000, 27), (['AS', 'QS'], ['2D', '7C', 'TH'], 5, 8000, 28),
    (['3C', '3D'], [], 9, 6000, 29),
]

TRACE_GRID = [  # (hero, board, n_players, seed) -- first 1000 iterations kept
    (['AH', 'KH'], [], 2, 0), (['AH', 'KH'], [], 3, 1), (['2C', '7D'], ['AS', 'KS', 'QS'], 6, 0),
    (['TC', 'TH'], ['4D', 'QD', 'KC', '2S'], 6, 5), (['3H', '3S'], ['8S', '4S', 'QH', '8C', '4H'], 4, 7),
    (['AS', 'AC'], [], 10, 2), (['2C', '3C'], ['AS', 'AH', 'AD', 'AC'], 10, 3),
]


def gen_tallies():
    out = []
    for hero, board, n, runs, seed in TALLY_GRID:
        t0 = time.time()
        row, _ = run_ref(hero, board, n, runs, seed)
        out.append(row)
        print("tally", hero, board, n, runs, seed, "->", row["wins"], row["passes"], row["mt_words"],
              "%.1fs" % (time.time() - t0))
    with open(os.path.join(HERE, "tallies.json"), "w") as f:
        json.dump(out, f, indent=0)


def gen_traces():
    K = 1000
    d = {}
    meta = []
    for i, (hero, board, n, seed) in enumerate(TRACE_GRID):
        row, rec = run_ref(hero, board, n, K, seed, keep=K)
        hands = np.array(rec.hands, np.uint8)  # [K, n, 7]
        d["hands_%d" % i] = hands
        d["words_%d" % i] = np.array(rec.words, np.uint16)
        meta.append({"hero": hero, "board": board, "n_players": n, "seed": seed, "runs": K, "wins": row["wins"],
                     "passes": row["passes"], "mt_words": row["mt_words"]})
    d["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "deal_traces.npz"), **d)
    print("deal_traces:", len(meta))


# ----------------------------------------------------------------------------- F4
STAT_ROWS = [  # tests/test_montecarlo_python.py:44-212 (hero, board, players, expected %); tolerance there: +-3
    (['3H', '3S'], ['8S', '4S', 'QH', '8C', '4H'], 2, 40.2), (['8H', '8D'], ['QH', '7H', '9H', 'JH', 'TH'], 2, 95.6),
    (['AS', 'KS'], [], 3, 49.9 + 1.9), (['AS', 'KS'], [], 2, 66.1 + 1.6),
    (['8S', 'TS'], ['8H', 'KS', '9S', 'TH', 'KH'], 2, 71.5 + 5.9), (['8S', 'TS'], ['2S', '3S', '4S', 'KS', 'AS'], 2, 87),
    (['8S', '2S'], ['5S', '3S', '4S', 'KS', 'AS'], 2, 100), (['8S', 'TS'], [], 5, 22.6 + 2.9),
    (['2C', 'QS'], [], 2, 49.6), (['7H', '7S'], ['7C', '8C', '8S', 'AC', 'AH'], 2, 83),
    (['3S', 'QH'], ['2C', '5H', '7C'], 2, 30.9 + 2.2), (['5C', 'JS'], [], 4, 23),
    (['TC', 'TH'], ['4D', 'QD', 'KC'], 2, 66.7 + 0.38), (['JH', 'QS'], ['5C', 'JD', 'AS', 'KS', 'QD'], 2, 77),
    (['2H', '8S'], ['AC', 'AD', 'AS', 'KS', 'KD'], 2, 95), (['KD', 'KS'], ['4D', '6S', '9C', '9S', 'TC'], 2, 88),
    (['5H', 'KD'], ['KH', 'JS', '2C', 'QS'], 2, 75.6 + 3.6), (['JD', 'JS'], ['8C', 'TC', 'JC', '5H', 'QC'], 3, 26.1),
    (['TD', '7D'], ['8D', 'QD', '7C', '5D', '6D'], 2, 87),
]


def gen_stats():
    rows = [{"hero": h, "board": b, "n_players": n, "expected_pct": e, "tol_pct": 3.0} for h, b, n, e in STAT_ROWS]
    with open(os.path.join(HERE, "stat_expectations.json"), "w") as f:
        json.dump(rows, f, indent=0)
    print("stat_expectations:", len(rows))


# ----------------------------------------------------------------------------- F6
SEQUENCE = [  # consecutive calls sharing numpy's global state after ONE np.random.seed, like gym_env/env.py does
    (['AH', 'KH'], [], 2, 3000), (['2C', '7D'], ['AS', 'KS', 'QS'], 6, 2000), (['TC', 'TH'], ['4D', 'QD', 'KC', '2S'], 3, 1500),
    (['3H', '3S'], ['8S', '4S', 'QH', '8C', '4H'], 4, 1000), (['AS', 'AC'], [], 10, 500),
]


def gen_sequence():
    out = []
    for seed in (0, 20261004):
        np.random.seed(seed)
        calls = []
        for hero, board, n, runs in SEQUENCE:
            sim = mp.MonteCarlo()
            sim.run_montecarlo([list(hero)], list(board), n, 1, maxRuns=runs, timeout=FAR, ghost_cards='',
                               opponent_range=1)
            by_type = {t: 0 for t in TYPES}
            for k, v in sim.winnerCardTypeList.items():
                by_type[k] = int(round(v * sim.runs))
            between = int(np.random.randint(0, 52))  # the env deals with the same stream between equity calls
            calls.append({"hero": hero, "board": board, "n_players": n, "runs": runs,
                          "wins": int(round(sim.equity * sim.runs)), "passes": int(sim.passes),
                          "by_type": [by_type[t] for t in TYPES], "randint52_after": between})
        tail = [int(x) for x in np.random.randint(0, 2 ** 32, size=4, dtype=np.uint32)]
        out.append({"seed": seed, "calls": calls, "next_words": tail})
    with open(os.path.join(HERE, "sequence.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("sequence:", len(out))


# ----------------------------------------------------------------------------- F7
EXT_GRID = [  # (player_card_list, board, n_players, runs, seed, ghost_cards, opponent_range)
    ([['KS', 'KC']], ['3D', '9H', 'AS', '7S', 'QH'], 3, 8000, 0, '', 0.25),           # tests/test_montecarlo_python.py:215-222
    ([{'AKO', 'AA'}], ['3D', '9H', 'AS', '7S', 'QH'], 3, 8000, 1, '', 0.25),          # :225-232 hero given as a range
    ([['AH', 'KH']], [], 2, 6000, 2, '', 0.1), ([['AH', 'KH']], [], 4, 4000, 3, '', 0.5),
    ([['AH', 'KH']], [], 3, 4000, 4, ['AS', 'AD'], 1), ([['7C', '2D']], ['AS', 'KS', 'QS'], 6, 3000, 5, ['AH', 'KH'], 0.3),
    ([['AH', 'KH'], ['QS', 'QD']], [], 4, 4000, 6, '', 1), ([['AH', 'KH'], ['QS', 'QD']], ['2C', '3C', '4C'], 5, 3000, 7, '', 0.4),
    ([{'22', '33', '44', 'TJS', '9TS'}], [], 2, 4000, 8, '', 1), ([{'AA', 'KK', 'QQ', 'AKS'}], ['2H', '7D', 'JC', 'JD'], 6, 3000, 9, ['2C', '2D'], 0.6),
    ([['9S', '9H']], [], 3, 4000, 10, '', {'AA', 'KK', 'AKS', 'AKO', 'QQ', 'JJ', 'TT'}),
    ([{'AKS', 'KQS', '67S', '78S'}, ['QS', 'QD']], ['2C', '3C', '4C'], 4, 3000, 11, '', 0.8),
    ([['AH', 'KH']], [], 10, 1500, 12, '', 0.9), ([{'AA', 'KK', '72O', '27O', 'AKO'}], [], 10, 1500, 13, '', 0.35),
    ([['AH', 'KH']], [], 2, 3000, 14, '', 0.001),                                      # int(169 * r) == 0 -> every class
    # any number of known hands, each two cards or a set of classes (montecarlo_python.py:133-163)
    ([['AH', 'KH'], ['QS', 'QD'], ['7C', '7D']], [], 5, 3000, 15, '', 1),
    ([['AH', 'KH'], ['QS', 'QD'], ['7C', '7D'], ['2S', '3S']], ['AS', 'KD', '5C'], 4, 3000, 16, '', 1),   # no random opponent
    ([['TC', 'TD'], {'AA', 'KK', 'QQ', 'AKS', 'AKO'}], [], 3, 3000, 17, '', 0.5),       # a RANGE for the second known hand
    ([{'AKS', 'AKO', 'AQS'}, {'22', '33', '44', '55', '66'}, ['JS', 'JH']], ['2C', '9D', 'KC'], 5, 2500, 18, '', 0.7),
    # a list hand BEHIND a range hand that may already have drawn one of its cards (the try/except at :154-161)
    ([{'AA', 'AKS', 'AKO', 'AQO'}, ['AS', 'KS']], [], 3, 3000, 19, '', 1),
    ([['9C', '9D'], ['AH', 'AD'], {'KK', 'QQ', 'JJ', 'TT', 'AKS'}, ['5H', '6H']], ['7H', '8H', '2C', 'KD'], 6, 2500, 20,
     ['AS', 'AC'], 0.4),
    ([{'72O', '27O', '83O', '38O'}, {'AA'}, {'KK'}], [], 4, 2000, 21, '', {'QQ', 'JJ', 'AKS', 'AQS', 'KQS'}),
]


def gen_ext():
    out = []
    for pcl, board, n, runs, seed, ghost, rng in EXT_GRID:
        np.random.seed(seed)
        rec = Recorder(0)
        mp.eval_best_hand = rec
        try:
            rec.start()
            sim = mp.MonteCarlo()
            pl = [set(x) if isinstance(x, set) else list(x) for x in pcl]
            sim.run_montecarlo(pl, list(board), n, 1, maxRuns=runs, timeout=FAR, ghost_cards=ghost, opponent_range=rng)
        finally:
            mp.eval_best_hand = rec.orig
        by_type = {t: 0 for t in TYPES}
        for k, v in sim.winnerCardTypeList.items():
            by_type[k] = int(round(v * sim.runs))
        out.append({"players": [sorted(x) if isinstance(x, set) else x for x in pcl],
                    "hero_is_range": isinstance(pcl[0], set), "board": board, "n_players": n, "runs": runs,
                    "seed": seed, "ghost": list(ghost) if ghost else [], "opponent_range": sorted(rng) if isinstance(rng, set) else rng,
                    "wins": int(round(sim.equity * sim.runs)), "passes": int(sim.passes),
                    "by_type": [by_type[t] for t in TYPES], "mt_words": int(rec.total_words)})
        print("ext", out[-1]["players"], board, n, rng, "->", out[-1]["wins"], out[-1]["passes"])
    with open(os.path.join(HERE, "ext_tallies.json"), "w") as f:
        json.dump(out, f, indent=0)
    # the 169 preflop classes in the order the reference sorts them (ascending equity; ties keep dict order):
    # run_montecarlo(opponent_range=r) keeps the LAST int(169 * r) of them (montecarlo_python.py:105-112)
    sim = mp.MonteCarlo()
    sim.get_opponent_allowed_cards_list(1)
    import operator
    order = [k for k, _ in sorted(sim.preflop_equities.items(), key=operator.itemgetter(1))]
    assert len(order) == 169 and len(set(order)) == 169
    for r in (0.25, 0.5, 0.9, 1):
        assert set(order[-int(169 * r):]) == sim.get_opponent_allowed_cards_list(r)
    with open(os.path.join(HERE, "..", "..", "neuron_poker_amd", "preflop_classes.json"), "w") as f:
        json.dump(order, f)
    print("ext_tallies:", len(out), "preflop classes:", len(order))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ext":
        gen_ext()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tallies":
        gen_tallies()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sequence":
        gen_sequence()
        sys.exit(0)
    g = np.random.default_rng(20261004)
    gen_stats()
    gen_cases()
    gen_evaluator(g)
    gen_showdowns(g)
    gen_traces()
    gen_tallies()
    gen_sequence()
    gen_ext()
