"""Lock-step driver for many No-Limit Hold'em tables whose seats act on the GPU equity (BASELINE configs[4]).

The reference plays ONE table (gym_env/env.py HoldemTable + gym_env/cycle.py PlayerCycle) and spends 99 % of its
time in get_equity (gym_env/env.py:261-262), one 1000-run query before every agent action and one after it.
This module restates the table rules needed to generate that query stream -- blinds, betting rounds and their
stop rules, raise caps, all-in bookkeeping, side-pot payout, dealing from numpy's global stream -- as a
generator per table that YIELDS its equity queries, so that a batch of tables advances in lock-step and all
their queries of one step go to the GPU in ONE mcq_eval_batch call.

It is a restatement in this repo's own structure (flat arrays, one generator), not a copy; where the reference
has quirks that shape the query stream they are kept and cited:
  * the round pot is added to the community pot twice at the end of a round (env.py:547-568: _close_round adds
    the player pots, _clean_up_pots adds current_round_pot) -- it only feeds bet sizing and legal moves;
  * payouts use the per-seat contribution cap (env.py:595-606);
  * every observation issues an equity query, also the one after the hand that ended the episode
    (env.py:224-262), with whatever table cards / alive count the table was left with.
Pinned by tests/golden/env_traces.json (seeded episodes of the reference itself: every query, action, stack).

Policies: callable(seat, legal_moves: set[int], equity: float) -> action int (gym_env/enums.py Action values).
equity_policy(min_call, min_bet) is agents/agent_consider_equity.py:25-56.
"""
import numpy as np

FOLD, CHECK, CALL, RAISE_3BB, RAISE_HALF_POT, RAISE_POT, RAISE_2POT, ALL_IN, SMALL_BLIND, BIG_BLIND = range(10)
PREFLOP, FLOP, TURN, RIVER, END_HIDDEN, SHOWDOWN = range(6)


def equity_policy(min_call_equity, min_bet_equity):
    """agents/agent_consider_equity.py:25-56"""
    def act(seat, legal, equity):
        if equity > min_bet_equity + 0.2 and ALL_IN in legal:
            return ALL_IN
        if equity > min_bet_equity + 0.1 and RAISE_2POT in legal:
            return RAISE_2POT
        if equity > min_bet_equity and RAISE_POT in legal:
            return RAISE_POT
        if equity > min_bet_equity - 0.1 and RAISE_HALF_POT in legal:
            return RAISE_HALF_POT
        if equity > min_call_equity and CALL in legal:
            return CALL
        if CHECK in legal:
            return CHECK
        return FOLD
    return act


def random_policy(rng):
    """agents/agent_random.py:21-29 with this module's own generator (the reference uses Python's `random`)."""
    allowed = {FOLD, CHECK, CALL, RAISE_POT, RAISE_HALF_POT, RAISE_2POT}

    def act(seat, legal, equity):
        moves = sorted(allowed & set(legal))
        return moves[int(rng.integers(0, len(moves)))]
    return act


class _Cycle:
    """gym_env/cycle.py:10-167 -- who acts next and when a betting round stops."""

    def __init__(self, n, max_raises=2):
        self.n = n
        self.start_idx = 0
        self.max_steps_total = None
        self.last_raiser_step = None
        self.max_steps_after_raiser = (max_raises - 1) * n - 1   # env.py:160
        self.max_steps_after_big_blind = n                       # env.py:162
        self.last_raiser = None
        self.step_counter = 0
        self.steps_for_blind_betting = 2
        self.idx = 0
        self.dealer_idx = -1                                     # env.py:161
        self.alive = [True] * n
        self.max_raises = max_raises
        self.checkers = 0
        self.max_remaining_steps_without_raising = n
        self.new_hand_reset()

    def new_hand_reset(self):                                    # cycle.py:40-46
        self.idx = self.start_idx
        self.can_move = [True] * self.n
        self.out_of_cash = [False] * self.n
        self.folder = [False] * self.n
        self.step_counter = 0

    def new_street_reset(self):                                  # cycle.py:48-56
        self.step_counter = 0
        self.idx = self.dealer_idx
        self.last_raiser_step = self.n
        self.checkers = 0
        self.max_remaining_steps_without_raising = len(self.alive) - 1
        self.last_raiser = None

    def update_alive(self):                                      # cycle.py:155-158
        self.alive = [a or b for a, b in zip(self.can_move, self.out_of_cash)]

    def n_alive(self):
        return sum(self.alive)

    def next_player(self):                                       # cycle.py:58-101; None = the round is over
        if sum(1 for a, b in zip(self.can_move, self.out_of_cash) if a or b) < 2:
            return None
        self.idx += 1
        self.step_counter += 1
        self.idx %= self.n
        if self.max_steps_total and self.step_counter > self.max_steps_total:
            return None
        if self.last_raiser:
            if self.step_counter > self.last_raiser + self.max_remaining_steps_without_raising:
                return None
            if self.max_steps_after_raiser and self.step_counter > self.max_steps_after_raiser + self.last_raiser:
                return None
        elif self.max_steps_after_raiser and \
                self.step_counter > self.max_steps_after_big_blind + self.steps_for_blind_betting:
            return None
        if self.checkers == sum(self.alive):
            return None
        while not self.can_move[self.idx]:
            self.idx += 1
            self.step_counter += 1
            self.idx %= self.n
            if self.max_steps_total and self.step_counter >= self.max_steps_total:
                return None
        self.update_alive()
        return self.idx

    def next_dealer(self):                                       # cycle.py:103-114
        self.dealer_idx = (self.dealer_idx + 1) % self.n
        while not self.can_move[self.dealer_idx]:
            self.dealer_idx = (self.dealer_idx + 1) % self.n
        return self.dealer_idx

    def mark_out_of_cash_but_contributed(self):                  # cycle.py:141-144
        self.out_of_cash[self.idx] = True
        assert self.can_move[self.idx], "Already deactivated"
        self.can_move[self.idx] = False

    def mark_bb(self):                                           # cycle.py:146-149
        self.last_raiser_step = self.step_counter + self.n
        self.max_steps_total = self.step_counter + self.n * self.max_raises + 2

    def potential_winners(self):                                 # cycle.py:160-167
        return [(a or b) and not f for a, b, f in zip(self.can_move, self.out_of_cash, self.folder)]


class TableSim:
    """One table.  episode() is a generator: it yields (hole[2], table_cards[...], players_alive) card-id queries
    and expects the equity (float) to be sent back; it returns when the episode is over (env.py:138-168 reset)."""

    def __init__(self, policies, initial_stacks=100, small_blind=1, big_blind=2, max_raises=2, showdown=None,
                 randint=None, calculate_equity=False):
        self.policies = list(policies)
        self.calculate_equity = bool(calculate_equity)   # HoldemTable(calculate_equity=True): env.py:248-256
        self.n = len(self.policies)
        self.initial_stacks = initial_stacks
        self.small_blind, self.big_blind, self.max_raises = small_blind, big_blind, max_raises
        # showdown(hands[[7 ids]...]) -> index of the best hand (first of equals); default: the engine's evaluator
        self.showdown = showdown
        self.randint = randint or (lambda n: int(np.random.randint(0, n)))   # env.py:680,686: numpy's global stream
        self.log = None            # optional list: events like tests/golden/env_traces.json
        self.env_steps = 0
        self.queries = 0

    # ---- episode ------------------------------------------------------------------------------------------
    def episode(self):
        n = self.n
        self.done = False
        self.funds_history = []
        self.stacks = [self.initial_stacks] * n
        self.cards = [[] for _ in range(n)]
        self.table_cards = []
        self.stage = PREFLOP
        self.winner_ix = None
        self.current = None
        self.community_pot = 0
        self.current_round_pot = 9                               # env.py:115 (sic)
        self.player_pots = [0] * n
        self.player_max_win = [0] * n
        self.min_call = 0
        self.num_raises = [[0] * 4 for _ in range(n)]
        self.legal = []
        self.cycle = _Cycle(n, self.max_raises)
        self._start_new_hand()
        yield from self._observe()                               # env.py:164
        while not self.done:                                     # env.py:178-188 (every seat is an autoplay agent)
            equity = yield from self._observe()
            seat = self.current
            action = self.policies[seat](seat, set(self.legal), equity)
            if self.log is not None:
                self.log.append(["a", seat, int(action), sorted(self.legal), [float(s) for s in self.stacks],
                                 int(self.stage)])
            if action not in self.legal:
                continue                                         # env.py:183-184 illegal move: nothing happens
            self.env_steps += 1
            self._process_decision(action)                       # env.py:202-211 _execute_step
            self._next_player()
            if self.stage in (END_HIDDEN, SHOWDOWN):
                self._end_hand()
                self._start_new_hand()
            yield from self._observe()

    def _observe(self):                                          # env.py:224-281 _get_environment
        if not self.done:
            self._legal_moves()
        if self.current is None:                                 # game over
            self.current = self.winner_ix
        self.queries += 1
        hole = list(self.cards[self.current])
        alive = self.cycle.n_alive()
        if self.calculate_equity:                                # env.py:248-256: three more calls with the same
            for _ in range(3):                                   # arguments; their numbers only go into the observation
                self.queries += 1
                extra = yield (hole, list(self.table_cards), alive)
                if self.log is not None:
                    self.log.append(["q", sorted(hole), sorted(self.table_cards), alive, 1000, extra])
        equity = yield (hole, list(self.table_cards), alive)
        if self.log is not None:
            self.log.append(["q", sorted(hole), sorted(self.table_cards), alive, 1000, equity])
        self._legal_moves()
        return equity

    # ---- hands ----------------------------------------------------------------------------------------------
    def _start_new_hand(self):                                   # env.py:400-440
        self.funds_history.append(list(self.stacks))
        self.num_raises = [[0] * 4 for _ in range(self.n)]
        if self._check_game_over():
            return
        self.table_cards = []
        self.deck = list(range(52))                              # ascending ids == env.py:667-671 order
        self.stage = PREFLOP
        self.community_pot = 0
        self.current_round_pot = 0
        self.player_pots = [0] * self.n
        self.player_max_win = [0] * self.n
        self.cards = [[] for _ in range(self.n)]
        self.cycle.next_dealer()
        for seat in range(self.n):                               # env.py:673-682
            if self.stacks[seat] <= 0:
                continue
            for _ in range(2):
                self.cards[seat].append(self.deck.pop(self.randint(len(self.deck))))
        self._initiate_round()

    def _check_game_over(self):                                  # env.py:447-468
        self.cycle.new_hand_reset()
        alive = 0
        for seat in range(self.n):
            if self.stacks[seat] > 0:
                alive += 1
            else:
                self.cycle.can_move[seat] = False
        if alive < 2 or self.stacks[0] == 0:
            self.done = True
            return True
        return False

    def _initiate_round(self):                                   # env.py:490-528
        self.min_call = 0
        self.cycle.new_street_reset()
        if self.stage != PREFLOP and self.n == 2:
            self.cycle.idx += 1
        if self.stage == PREFLOP:
            self.cycle.max_steps_total = self.n * self.max_raises + 2
            self._next_player()
            self._process_decision(SMALL_BLIND)
            self._next_player()
            self._process_decision(BIG_BLIND)
            self._next_player()
        elif self.stage in (FLOP, TURN, RIVER):
            self.cycle.max_steps_total = self.n * self.max_raises
            self._next_player()

    def _next_player(self):                                      # env.py:611-628
        self.current = self.cycle.next_player()
        if self.current is None:
            if self.cycle.n_alive() < 2:
                self.stage = END_HIDDEN
            else:
                self._end_round()
                self._initiate_round()

    def _end_round(self):                                        # env.py:541-568
        self.community_pot += sum(self.player_pots)              # _close_round
        self.player_pots = [0] * self.n
        if self.stage == PREFLOP:
            self.stage = FLOP
            self._deal_table(3)
        elif self.stage == FLOP:
            self.stage = TURN
            self._deal_table(1)
        elif self.stage == TURN:
            self.stage = RIVER
            self._deal_table(1)
        elif self.stage == RIVER:
            self.stage = SHOWDOWN
        self._clean_up_pots()

    def _clean_up_pots(self):                                    # env.py:564-568
        self.community_pot += self.current_round_pot
        self.current_round_pot = 0
        self.player_pots = [0] * self.n

    def _deal_table(self, k):                                    # env.py:684-688
        for _ in range(k):
            self.table_cards.append(self.deck.pop(self.randint(len(self.deck))))

    def _end_hand(self):                                         # env.py:570-606
        self._clean_up_pots()
        pw = self.cycle.potential_winners()
        idx = [i for i in range(self.n) if pw[i]]
        if len(idx) == 1:
            w = idx[0]
        else:
            assert self.stage == SHOWDOWN
            hands = [list(self.cards[i]) + list(self.table_cards) for i in idx]
            w = idx[self.showdown(hands)]
        cap = self.player_max_win[w]
        total = sum(min(cap, x) for x in self.player_max_win)
        remains = [max(0, x - cap) for x in self.player_max_win]
        self.stacks[w] += total
        self.winner_ix = w
        if total < sum(self.player_max_win):
            for i in range(self.n):
                self.stacks[i] += remains[i]

    # ---- one decision ------------------------------------------------------------------------------------------
    def _process_decision(self, action):                         # env.py:308-393
        seat = self.current
        cyc = self.cycle
        if action == FOLD:
            assert cyc.can_move[cyc.idx], "Already deactivated"
            cyc.can_move[cyc.idx] = False
            cyc.folder[cyc.idx] = True
        else:
            pot = self.community_pot + self.current_round_pot
            if action == CALL:
                contribution = min(self.min_call - self.player_pots[seat], self.stacks[seat])
            elif action == CHECK:
                contribution = 0
                cyc.checkers += 1
            elif action == RAISE_3BB:
                contribution = 3 * self.big_blind - self.player_pots[seat]
            elif action == RAISE_HALF_POT:
                contribution = pot / 2
            elif action == RAISE_POT:
                contribution = pot
            elif action == RAISE_2POT:
                contribution = pot * 2
            elif action == ALL_IN:
                contribution = self.stacks[seat]
            elif action == SMALL_BLIND:
                contribution = min(self.small_blind, self.stacks[seat])
            elif action == BIG_BLIND:
                contribution = min(self.big_blind, self.stacks[seat])
                cyc.mark_bb()
            else:
                raise RuntimeError("Illegal action.")
            if action in (RAISE_3BB, RAISE_HALF_POT, RAISE_POT, RAISE_2POT, ALL_IN):
                self.num_raises[seat][self.stage] += 1
            if contribution > self.min_call and action not in (SMALL_BLIND, BIG_BLIND):
                cyc.last_raiser = cyc.step_counter               # mark_raiser
            self.stacks[seat] -= contribution
            self.player_pots[seat] += contribution
            self.current_round_pot += contribution
            if self.stacks[seat] == 0 and contribution > 0:
                cyc.mark_out_of_cash_but_contributed()
            self.min_call = max(self.min_call, contribution)
            self.player_max_win[seat] += contribution
        cyc.update_alive()

    def _legal_moves(self):                                      # env.py:630-659
        self.legal = []
        if self.stage == SHOWDOWN:
            return
        seat = self.current
        if self.player_pots[seat] == max(self.player_pots):
            self.legal.append(CHECK)
        else:
            self.legal.append(CALL)
            self.legal.append(FOLD)
        if self.num_raises[seat][min(self.stage, 3)] < self.max_raises:
            pot = self.community_pot + self.current_round_pot
            stack = self.stacks[seat]
            if stack >= 3 * self.big_blind - self.player_pots[seat]:
                self.legal.append(RAISE_3BB)
            if stack >= pot / 2 >= self.min_call:
                self.legal.append(RAISE_HALF_POT)
            if stack >= pot >= self.min_call:
                self.legal.append(RAISE_POT)
            if stack >= pot * 2 >= self.min_call:
                self.legal.append(RAISE_2POT)
            if stack > 0:
                self.legal.append(ALL_IN)


class TableBatch:
    """T tables advanced in lock-step: per step every running table contributes one equity query and all of
    them are evaluated in ONE call of `evaluate(hole[B,2], board[B,5], n_players[B]) -> equity[B]`."""

    def __init__(self, tables):
        self.tables = list(tables)
        self.gens = [t.episode() for t in self.tables]
        self.pending = [None] * len(self.tables)
        self.lock_steps = 0
        for i, g in enumerate(self.gens):
            self.pending[i] = next(g, None)

    def running(self):
        return sum(p is not None for p in self.pending)

    def step(self, evaluate):
        idx = [i for i, p in enumerate(self.pending) if p is not None]
        if not idx:
            return 0
        hole = np.zeros((len(idx), 2), np.uint8)
        board = np.full((len(idx), 5), 255, np.uint8)
        npl = np.zeros(len(idx), np.uint8)
        for k, i in enumerate(idx):
            h, b, a = self.pending[i]
            hole[k] = h
            board[k, :len(b)] = b
            npl[k] = a
        eq = evaluate(hole, board, npl)
        for k, i in enumerate(idx):
            try:
                self.pending[i] = self.gens[i].send(float(eq[k]))
            except StopIteration:
                self.pending[i] = None
        self.lock_steps += 1
        return len(idx)
