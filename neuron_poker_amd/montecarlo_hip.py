"""Drop-in for tools/montecarlo_python.py of neuron_poker, computed on an MI355X.

Same call surface, same meaning of the arguments, same attributes afterwards
(reference: tools/montecarlo_python.py:191-252 and :401-406):

    get_equity(player_cards, table_cards, players, runs) -> float
    MonteCarlo().run_montecarlo(original_player_card_list, original_table_card_list, player_amount, ui,
                                maxRuns, timeout, ghost_cards, opponent_range=1) -> (equity, winTypesDict)
        then .equity .winnerCardTypeList .winTypesDict .runs .passes

so `HoldemTable.get_equity = montecarlo_hip.get_equity` (gym_env/env.py:75-81) is the whole integration.
New and additive: get_equity_batch() evaluates many states in one launch.

Deliberate differences:
  * exactly `runs` iterations are executed; the reference's 1 s wall-clock cut-off (montecarlo_python.py:235,
    :405) is not reproduced, `timeout` and `ui` are accepted and ignored;
  * the library never touches numpy's global random state.  seed(s) makes results reproducible; in
    mode 'replay' a call after seed(s) returns exactly what the reference returns after np.random.seed(s);
  * a hero card that is also on the table is rejected with ValueError (the reference silently swallows it,
    montecarlo_python.py:154-161);
  * all of run_montecarlo's arguments are supported: opponent ranges (a fraction of the 169 preflop classes or an
    explicit set), ghost cards, and any number of known hands, each two cards or a set of classes
    (tools/montecarlo_python.py:36-112, 133-181, 206-208; bit-exact in mode 'replay').  A range that cannot be
    dealt from the remaining cards raises ValueError where the reference would loop forever.
"""
import json
import os
import struct
import threading
from collections import Counter

import numpy as np

from . import _lib
from .cards import TYPES, card_id

__all__ = ["get_equity", "get_equity_batch", "get_equity_exact", "MonteCarlo", "seed", "configure"]

_state = {"couple_numpy": False,
          "mode": _lib.MODE_REPLAY_MT19937 if os.environ.get("MCQ_MODE", "philox").lower() == "replay"
          else _lib.MODE_PHILOX}
_lock = threading.Lock()


class _Stream(threading.local):
    """(seed, query counter) of the calling THREAD: seed(s) in a thread makes that thread's calls reproducible whatever
    other threads do meanwhile (every thread also has its own engine, _lib.default_engine()).  A thread that never
    called seed() starts from the operating system's entropy."""

    def __init__(self):
        self.seed = int.from_bytes(os.urandom(8), "little")
        self.counter = 0


_stream = _Stream()
_MODES = {"philox": _lib.MODE_PHILOX, "replay": _lib.MODE_REPLAY_MT19937,
          _lib.MODE_PHILOX: _lib.MODE_PHILOX, _lib.MODE_REPLAY_MT19937: _lib.MODE_REPLAY_MT19937}


def seed(s):
    """Counterpart of np.random.seed(s) for this module: the calling thread's next call uses stream `s`."""
    _stream.seed = int(s) & (2 ** 64 - 1)
    _stream.counter = 0


def configure(mode=None, couple_numpy=None, dealing=None):
    """mode: 'philox' (default, production) or 'replay' (bit-exact MT19937 replay of the reference).
    dealing: 'reference' (default: the Python reference's law incl. its index bias) or 'uniform' (unbiased, what
    the reference's Cython/C++ variants deal; production mode only) -- applied to every thread's default engine and
    to the multi-GPU engines of get_equity_batch(n_gpus=...); call it while no equity call is running.
    couple_numpy=True (replay mode only): draw from numpy's GLOBAL random state and advance it exactly as the
    reference does, so that code sharing np.random with the equity call (gym_env/env.py:142,680,686 deals with
    it) follows the reference's trajectory after np.random.seed(s)."""
    if mode is not None:
        if mode not in _MODES:
            raise ValueError("mode must be 'philox' or 'replay'")
        _state["mode"] = _MODES[mode]
    if couple_numpy is not None:
        _state["couple_numpy"] = bool(couple_numpy)
    if dealing is not None:
        _lib.set_default_dealing_law(dealing)
        with _lock:
            multis = list(_MULTI.values())
        for me in multis:
            with me.lock:
                me.set_dealing_law(dealing)


_CLASS_ORDER = None
_TOP_BITS = {}   # take -> bit set of the last `take` classes of the equity order


def _class_order():
    """The 169 preflop classes in the order the reference sorts them by equity (ascending); generated from the
    reference by tests/golden/gen_golden.py into preflop_classes.json."""
    global _CLASS_ORDER
    if _CLASS_ORDER is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "preflop_classes.json")) as f:
            _CLASS_ORDER = json.load(f)
    return _CLASS_ORDER


def _opponent_range_bits(opponent_range):
    """run_montecarlo's opponent_range -> 169-bit set or None for "every class" (montecarlo_python.py:194-199 and
    :105-112: a number keeps the LAST int(169 * r) classes of the equity-sorted list -- and all of them when that
    is 0, as list[-0:] does --, anything else is used as the set of allowed classes)."""
    if type(opponent_range) in (float, int):
        take = int(169 * opponent_range)
        if take <= 0 or take >= 169:
            return None
        bits = _TOP_BITS.get(take)
        if bits is None:
            bits = _TOP_BITS[take] = _lib.range_bits(_class_order()[-take:])
        return bits
    return _lib.range_bits(opponent_range)


_EXT_CACHE = {}   # (opponent_range, ghost_cards, hero range, further known hands) -> mcq_query_ext record (1-element array)


def _ext_record(hero, hero_is_range, known_hands, ghost_cards, opponent_range, opp_bits):
    """The extension record of a run_montecarlo call.  An agent asks with the same ranges decision after decision (what
    changes is its cards and the table), and building the record -- class strings to bit sets, a numpy structured
    array -- costs more than the GPU call (60 of 86 us): the records of the last few distinct settings are kept."""
    def freeze(h):
        return frozenset(h) if isinstance(h, (set, frozenset)) else tuple(h)
    try:
        key = (opponent_range if type(opponent_range) in (float, int) else frozenset(opponent_range),
               ghost_cards if isinstance(ghost_cards, str) or ghost_cards is None else tuple(ghost_cards),
               frozenset(hero) if hero_is_range else None, tuple(freeze(h) for h in known_hands))
        ext = _EXT_CACHE.get(key)
    except TypeError:  # something unhashable inside: build it afresh
        key, ext = None, None
    if ext is None:
        ghost = None
        if ghost_cards != '' and ghost_cards is not None:
            ghost = [card_id(ghost_cards[0]), card_id(ghost_cards[1])]
        known = [_lib.range_bits(h) if isinstance(h, (set, frozenset)) else [card_id(c) for c in h] for h in known_hands]
        ext = _lib.pack_query_ext(1, ghost=ghost, known=known,
                                  hero_range=_lib.range_bits(hero) if hero_is_range else None, opp_range=opp_bits)
        if key is not None:
            if len(_EXT_CACHE) >= 64:
                _EXT_CACHE.clear()
            _EXT_CACHE[key] = ext
    return ext


def _take_ids(n):
    first = _stream.counter
    _stream.counter = first + n
    return _stream.seed, first


def _query(player_cards, table_cards, players, runs):
    hole = [card_id(c) for c in player_cards]
    board = [card_id(c) for c in table_cards]
    if len(hole) != 2:
        raise ValueError("player_cards must hold exactly two cards")
    if len(board) > 5:
        raise ValueError("table_cards holds more than five cards")
    players = int(players)
    runs = int(runs)
    if runs < 1:
        raise ValueError("runs must be >= 1")
    return _lib.pack_query_one(hole, board, players, runs)


class MonteCarlo(object):
    """Mirror of tools/montecarlo_python.py:22 MonteCarlo for the path gym_env/env.py uses."""

    def __init__(self, engine=None):
        self._engine = engine
        self.equity = None
        self.winnerCardTypeList = Counter()
        self.winTypesDict = self.winnerCardTypeList.items()
        self.runs = 0
        self.passes = 0
        self.result = None

    def run_montecarlo(self, original_player_card_list, original_table_card_list, player_amount, ui, maxRuns,
                       timeout, ghost_cards, opponent_range=1, *, mode=None, seed=None):
        eng = self._engine or _lib.default_engine()
        m = _state["mode"] if mode is None else _MODES[mode]
        players = list(original_player_card_list)
        if not 1 <= len(players) <= 1 + _lib.MAX_KNOWN:
            raise ValueError("between one and ten known hands")
        hero = players[0]
        hero_is_range = isinstance(hero, (set, frozenset))
        opp_bits = _opponent_range_bits(opponent_range)
        plain = not hero_is_range and len(players) == 1 and opp_bits is None and (ghost_cards == '' or ghost_cards is None)
        q = _query(["2C", "2D"] if hero_is_range else list(hero), list(original_table_card_list), player_amount, maxRuns)
        if plain:
            if m == _lib.MODE_REPLAY_MT19937 and _state["couple_numpy"] and seed is None:
                res = eng.eval_batch_numpy_stream(q)[0]
            else:
                s, first = _take_ids(1) if seed is None else (int(seed), 0)
                res = eng.eval_batch(q, s, first_query_id=first, mode=m)[0]
        else:
            if hero_is_range:
                q["hole"] = 0
            ext = _ext_record(hero, hero_is_range, players[1:], ghost_cards, opponent_range, opp_bits)
            s, first = _take_ids(1) if seed is None else (int(seed), 0)
            # (straight to the C ABI: Engine.eval_batch_ext's conversions of arrays that are already right cost 8 us)
            out = np.zeros(1, _lib.RESULT_DTYPE)
            rc = eng._lib.mcq_eval_batch_ext(eng._ctx, q.ctypes.data, ext.ctypes.data, 1, s & 0xFFFFFFFFFFFFFFFF,
                                             first & 0xFFFFFFFFFFFFFFFF, m, out.ctypes.data)
            if rc:
                _lib._raise(rc)
            res = out[0]
        runs = int(res["runs"])
        wins = int(res["win"]) + int(res["tie"])
        self.result = res
        self.equity = wins / runs                                   # montecarlo_python.py:243
        self.winnerCardTypeList = Counter({TYPES[t]: int(c) / runs   # :244-246
                                           for t, c in enumerate(res["by_type"]) if c})
        self.winTypesDict = self.winnerCardTypeList.items()          # :248
        self.runs = runs                                             # :249
        self.passes = int(res["passes"])                             # :250
        return self.equity, self.winTypesDict


_CARD_ID = {r + s: 4 * i + j for i, r in enumerate("23456789TJQKA") for j, s in enumerate("CDHS")}
_fast = threading.local()   # per thread: (engine, ctypes query buffer, ctypes result buffer)


def get_equity(player_cards, table_cards, players, runs):
    """Get equity from a Monte-Carlo run -- tools/montecarlo_python.py:401-406, on the GPU.

    This is the call gym_env/env.py:261-262 makes at every step, so it does not go through MonteCarlo() and numpy: the
    16-byte record is packed into a reusable ctypes buffer and mcq_eval_batch is called directly (what
    run_montecarlo([list(player_cards)], list(table_cards), players, 1, maxRuns=runs, ...) computes, same streams)."""
    if _state["mode"] != _lib.MODE_PHILOX or _state["couple_numpy"]:
        simulation = MonteCarlo()
        simulation.run_montecarlo([list(player_cards)], list(table_cards), players, 1, maxRuns=runs, timeout=0,
                                  ghost_cards='', opponent_range=1)
        return simulation.equity
    try:
        hole = [_CARD_ID[c] for c in player_cards]
        board = [_CARD_ID[c] for c in table_cards]
    except (KeyError, TypeError):
        raise ValueError("a card is not in the deck: %r %r" % (player_cards, table_cards)) from None
    nb, runs, players = len(board), int(runs), int(players)
    if len(hole) != 2 or nb > 5:
        raise ValueError("player_cards must hold exactly two cards, table_cards at most five")
    if runs < 1:
        raise ValueError("runs must be >= 1")
    st = getattr(_fast, "st", None)
    eng = _lib.default_engine()
    if st is None or st[0] is not eng:
        import ctypes
        st = _fast.st = (eng, ctypes.create_string_buffer(16), (ctypes.c_uint64 * 13)())
    try:
        struct.pack_into("<2B5BBB3xI", st[1], 0, hole[0], hole[1], *(board + [0] * (5 - nb)), nb, players, runs)
    except struct.error as e:
        raise ValueError("n_players or runs out of range: %s" % e) from None
    s, first = _take_ids(1)
    rc = eng._lib.mcq_eval_batch(eng._ctx, st[1], 1, s, first, _lib.MODE_PHILOX, st[2])
    if rc:
        _lib._raise(rc)
    out = st[2]
    return (out[2] + out[3]) / out[0]


_MULTI = {}   # tuple of device ordinals -> MultiEngine (made at first use, kept; its callers take turns on its lock)


def get_equity_batch(hole, board, n_players, runs, seed=None, first_query_id=0, mode=None, engine=None, n_gpus=None,
                     devices=None):
    """Many states in one launch.

    hole [B,2] u8 card ids, board [B,5] u8 (0xFF = empty), n_players scalar or [B], runs scalar or [B].
    -> (equity[B] float64, tallies[B,13] uint64) with tally columns runs, passes, win, tie, by_type[9].
    Query i gets query id first_query_id + i: splitting a batch keeps every per-query tally identical.
    n_gpus > 1: the batch is sharded over the first n_gpus devices of the node -- counted from
    $MCQ_DEVICE / $LOCAL_RANK's device as the single-GPU path does -- by the library's multi-GPU entry (mcq_multi_*: one
    all-reduce of the integer tallies; production mode only); the tallies are the same integers as on one GPU, under the
    dealing law configure(dealing=...) has set.  devices=[...] names the shards' devices explicitly instead (one
    ordinal per shard, repeats allowed: several shards on one GPU).
    STATUS of n_gpus > 1: untested on more than one distinct device (no multi-GPU node was reachable; the partitions,
    the same-device add and a one-rank RCCL communicator are tested on one GPU with devices=[0, 0, ...]).
    """
    q = _lib.pack_queries(hole, board, n_players, runs)
    m = _state["mode"] if mode is None else _MODES[mode]
    if seed is None:
        s, base = _take_ids(len(q))
        first_query_id = base + first_query_id
    else:
        s = int(seed)
    if devices is None and n_gpus is not None and int(n_gpus) > 1:
        n_dev = _lib.load_library().mcq_device_count()
        if int(n_gpus) > n_dev:
            raise ValueError("n_gpus = %d, but %d HIP device(s) are visible" % (int(n_gpus), n_dev))
        dev0 = int(os.environ.get("MCQ_DEVICE", os.environ.get("LOCAL_RANK", "0"))) % max(n_dev, 1)
        devices = [(dev0 + k) % n_dev for k in range(int(n_gpus))]
    if devices is not None:
        if m != _lib.MODE_PHILOX or engine is not None:
            raise ValueError("n_gpus > 1 / devices: production mode on the library's own contexts only")
        key = tuple(int(d) for d in devices)
        with _lock:
            me = _MULTI.get(key)
            if me is None:
                me = _MULTI[key] = _lib.MultiEngine(list(key))
                if _lib.default_dealing_law() != "reference":
                    me.set_dealing_law(_lib.default_dealing_law())
        with me.lock:   # mcq_multi keeps per-call state: one call at a time per object
            res = me.eval_batch(q, s, first_query_id=first_query_id)
    else:
        eng = engine or _lib.default_engine()
        res = eng.eval_batch(q, s, first_query_id=first_query_id, mode=m)
    tallies = res.view(np.uint64).reshape(len(q), 13)
    runs_f = np.maximum(tallies[:, 0], 1).astype(np.float64)
    equity = (tallies[:, 2] + tallies[:, 3]).astype(np.float64) / runs_f
    return equity, tallies


def get_equity_exact(player_cards, table_cards, players, dealing="reference", engine=None):
    """The number get_equity() converges to, by exhaustive enumeration on the GPU (1 to 3 players).

    dealing='reference': the exact expectation of tools/montecarlo_python.py's dealing (index bias included);
    'uniform': every remaining card equally likely (what montecarlo_cython.pyx / Montecarlo.cpp intend and
    tools/montecarlo_cpp/Test.cpp:176-217 checks within 1 %).  -> (equity, result row of integer weights)."""
    q = _query(list(player_cards), list(table_cards), players, 1)
    res = (engine or _lib.default_engine()).exact(q, dealing)[0]
    return (int(res["win"]) + int(res["tie"])) / int(res["runs"]), res
