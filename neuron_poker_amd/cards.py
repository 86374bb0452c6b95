"""Card notation of the reference: two characters, rank in '23456789TJQKA' then suit in 'CDHS'
(tools/hand_evaluator.py:5-6).  Card id = 4 * rank + suit, i.e. the position in the reference's ordered deck
(tools/montecarlo_python.py:114-119)."""
RANKS = "23456789TJQKA"
SUITS = "CDHS"
# hand types in by_type order; the spelling 'FoufOfAKind' is the reference's (tools/hand_evaluator.py:95)
TYPES = ["HighCard", "Pair", "TwoPair", "ThreeOfAKind", "Straight", "Flush", "FullHouse", "FoufOfAKind",
         "StraightFlush"]


def card_id(card):
    """'AH' -> 50.  Raises ValueError for anything that is not in the reference's deck, as list.index does
    there (tools/montecarlo_python.py:128)."""
    if not isinstance(card, str) or len(card) != 2:
        raise ValueError("%r is not in deck" % (card,))
    r, s = RANKS.find(card[0]), SUITS.find(card[1])
    if r < 0 or s < 0:
        raise ValueError("%r is not in deck" % (card,))
    return 4 * r + s


def card_str(cid):
    cid = int(cid)
    if not 0 <= cid < 52:
        raise ValueError("card id out of range: %r" % (cid,))
    return RANKS[cid >> 2] + SUITS[cid & 3]


def key_type(keys):
    """by_type index of the 32-bit ranking keys returned by Engine.showdown (MCQ_KEY_TYPE in include/mcq.h)."""
    import numpy as np
    code = np.asarray(keys, dtype=np.uint32) >> 28
    return (code - (code >= 6)).astype(np.uint32)
