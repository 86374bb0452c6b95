"""Drop-in for tools/hand_evaluator.py of neuron_poker, computed on an MI355X (the showdown leaf of the hot path).

Same call surface as the reference (tools/hand_evaluator.py:9-24), same card strings, same results:

    get_winner(player_hands, table_cards) -> (best_hand_ix, winner_card_type)      # gym_env/env.py:587
    eval_best_hand(hands) -> (best_hand, winner_card_type)                          # tools/montecarlo_python.py:222

`player_hands` = list of two-card lists, `table_cards` = list of five cards, `hands` = list of seven-card lists, cards
as two-character strings ('AH', 'TC'); the hand type is the reference's string ('HighCard' ... 'FoufOfAKind' [sic],
'StraightFlush', tools/hand_evaluator.py:92-115); of equal hands the FIRST wins (stable sort, :23).  So
`from neuron_poker_amd.hand_evaluator_hip import get_winner` in place of `from tools.hand_evaluator import get_winner`
(gym_env/env.py:13) is the whole integration.  New and additive: get_winner_batch() ranks many tables in one launch.

All of it goes through mcq_showdown (include/mcq.h) -- the evaluator the Monte-Carlo kernels use (mcq_eval_key,
csrc/mcq_device.hpp): no CPU path here either.

Deliberate difference: a hand must be seven DISTINCT cards of the deck; the reference also ranks hands that name a card
twice (its own tests/test_evaluator.py:27,63 do) -- ValueError here, as for a card that is not in the deck.
"""
import numpy as np

from . import _lib
from .cards import TYPES, card_id

__all__ = ["get_winner", "eval_best_hand", "get_winner_batch"]


def _ids(cards):
    return [card_id(c) for c in cards]


def eval_best_hand(hands, engine=None):
    """Evaluate the best hand -- tools/hand_evaluator.py:20-24: -> (the best of `hands`, its hand type string)."""
    hands = list(hands)
    if not hands:
        raise IndexError("list index out of range")   # what sorted([])[0] raises in the reference
    if not 1 <= len(hands) <= 10:
        raise ValueError("between one and ten hands")
    ids = [_ids(h) for h in hands]
    if any(len(h) != 7 for h in ids):
        raise ValueError("a hand is seven cards (two hole cards and five table cards)")
    win, wtype = (engine or _lib.default_engine()).showdown(np.array([ids], np.uint8))
    return hands[int(win[0])], TYPES[int(wtype[0])]


def get_winner(player_hands, table_cards, engine=None):
    """Determine the winning hand of multiple players -- tools/hand_evaluator.py:9-17:
    -> (index of the best hand, first of equals; its hand type string)."""
    table = list(table_cards)
    with_table = [list(h) + table for h in player_hands]
    best, wtype = eval_best_hand(with_table, engine)
    return with_table.index(best), wtype   # (:16: the first hand EQUAL to the best one, as the reference's .index)


def get_winner_batch(hole, board, engine=None):
    """Many showdowns in one launch: hole [T, P, 2] card ids, board [T, 5] card ids -> (winner[T], type index[T])
    (TYPES[type index] is the reference's string)."""
    hole = np.asarray(hole, np.uint8)
    board = np.asarray(board, np.uint8)
    if hole.ndim != 3 or hole.shape[2] != 2 or board.shape != (hole.shape[0], 5):
        raise ValueError("hole [T, P, 2], board [T, 5]")
    T, P = hole.shape[0], hole.shape[1]
    hands = np.concatenate([hole, np.broadcast_to(board[:, None, :], (T, P, 5))], axis=2)
    return (engine or _lib.default_engine()).showdown(hands)
