#!/usr/bin/env python3
"""Record seeded episodes of the REFERENCE's own table (gym_env/env.py HoldemTable + gym_env/cycle.py) with the
equity agents of agents/agent_consider_equity.py, for pinning neuron_poker_amd/table_driver.py.

Only runnable in the build container (needs /root/reference).  gymnasium and pyglet are not installed; the two
modules are stubbed in sys.modules exactly as far as gym_env/env.py touches them at import time (base class,
space constructors) -- no reference code is altered or copied.  What is written is data:

  tests/golden/env_traces.json   per episode: seed, seat policies, every equity query (hole, table cards, players
                                 alive, equity returned), every agent action (seat, action, legal moves, stacks),
                                 per-hand stacks (funds history), winner

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen_env_traces.py
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))

# ---- stubs for the two missing imports of gym_env/env.py
gym = types.ModuleType("gymnasium")


class _Env:  # gymnasium.Env: HoldemTable only calls super().__init__()
    def __init__(self, *a, **k):
        pass


class _Space:
    def __init__(self, *a, **k):
        pass


gym.Env = _Env
spaces = types.ModuleType("gymnasium.spaces")
spaces.Discrete = _Space
spaces.Box = _Space
gym.spaces = spaces
envs = types.ModuleType("gymnasium.envs")
reg = types.ModuleType("gymnasium.envs.registration")
reg.register = lambda *a, **k: None
envs.registration = reg
gym.envs = envs
sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs,
                    "gymnasium.envs.registration": reg, "pyglet": types.ModuleType("pyglet")})

from gym_env.env import HoldemTable  # noqa: E402
from gym_env.enums import Action  # noqa: E402
from agents.agent_consider_equity import Player as EquityPlayer  # noqa: E402
from tools import montecarlo_python as mp  # noqa: E402

RANKS, SUITS = "23456789TJQKA", "CDHS"
POLICIES = [(.5, -.5), (.8, -.8), (.7, -.7), (.2, -.3),  # main.py:142-145
            (.6, -.6), (.3, -.1)]                          # two more deterministic seats instead of the random ones


def cid(s):
    return 4 * RANKS.index(s[0]) + SUITS.index(s[1])


def run_episode(seed, policies, stacks=100, calculate_equity=False):
    env = HoldemTable(initial_stacks=stacks, funds_plot=False, calculate_equity=calculate_equity)
    events = []
    real = mp.get_equity

    def get_equity(hole, table, players, runs):
        # sets have no order; the result does not depend on it (the cards are located by value in the deck)
        eq = real(hole, table, players, runs)
        events.append(["q", sorted(cid(c) for c in hole), sorted(cid(c) for c in table), int(players), int(runs), eq])
        return eq

    for i, (c, b) in enumerate(policies):
        agent = EquityPlayer(name="eq%d" % i, min_call_equity=c, min_bet_equity=b)
        orig = agent.action

        def action(action_space, observation, info, funds_history, _orig=orig, _env=env):
            a = _orig(action_space, observation, info, funds_history)
            events.append(["a", int(_env.current_player.seat), int(Action(a).value),
                           sorted(int(x.value) for x in action_space), [float(p.stack) for p in _env.players],
                           int(_env.stage.value)])
            return a
        agent.action = action
        env.add_player(agent)
    env.get_equity = get_equity
    env.reset(seed=seed)
    assert env.done
    fh = env.funds_history.reset_index(drop=True).values.tolist()
    return {"seed": seed, "policies": policies, "stacks": stacks, "calculate_equity": bool(calculate_equity),
            "events": events, "winner": int(env.winner_ix),
            "final_stacks": [float(p.stack) for p in env.players], "funds_history": fh,
            "np_next_words": [int(x) for x in np.random.randint(0, 2 ** 32, size=2, dtype=np.uint32)]}


# main.py's thresholds make every seat go all-in at once; these play real streets (calls, checks, folds, raises)
MIXED = [(.3, .5), (.45, .6), (.2, .75), (.5, .9), (.35, .55), (.25, .65)]

if __name__ == "__main__":
    out = []
    for seed, pol in [(1, POLICIES), (7, MIXED), (2026, MIXED[:3]), (5, MIXED[:2]), (11, MIXED[1:5]), (3, MIXED),
                      (42, MIXED[::-1]), (8, [MIXED[0], MIXED[3], POLICIES[0], MIXED[2], MIXED[5]])]:
        ep = run_episode(seed, pol)
        nq = sum(e[0] == "q" for e in ep["events"])
        na = sum(e[0] == "a" for e in ep["events"])
        print("seed", seed, "seats", len(pol), "queries", nq, "actions", na, "hands", len(ep["funds_history"]),
              "winner", ep["winner"])
        out.append(ep)
    # calculate_equity=True: three more equity calls per observation (gym_env/env.py:248-256)
    ep = run_episode(13, MIXED[:4], calculate_equity=True)
    print("seed 13 calculate_equity queries", sum(e[0] == "q" for e in ep["events"]), "actions",
          sum(e[0] == "a" for e in ep["events"]), "winner", ep["winner"])
    out.append(ep)
    with open(os.path.join(HERE, "env_traces.json"), "w") as f:
        json.dump(out, f)
    print("env_traces:", len(out))
