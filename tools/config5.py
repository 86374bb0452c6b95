#!/usr/bin/env python3
"""BASELINE configs[4]: 6-player self-play on many tables, every seat acting on the GPU equity every step.

    python tools/config5.py [--tables 512] [--lock-steps 10000] [--runs 1000]

512 tables x 6 seats, 100-chip stacks, blinds 1/2; seats 0-3 play agents/agent_consider_equity.py with the
(min_call, min_bet) pairs of main.py:142-145, seats 4-5 play agents/agent_random.py's move set.  The tables
(neuron_poker_amd/table_driver.py, pinned to the reference's own table by tests/test_table_driver.py) advance in
lock-step; one lock-step = one equity query (1000 runs, gym_env/env.py:22,261-262) per running table, all in ONE
mcq_eval_batch call.  Tables whose episode ended start a new one.  Prints one JSON line: env steps (= agent
actions executed) per second, lock-steps per second and the share of wall time spent in the equity call
(reference: 99.3 % of its time).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=512)
    ap.add_argument("--lock-steps", type=int, default=10000)
    ap.add_argument("--runs", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--driver", choices=["native", "python"], default="native",
                    help="native: mcq_tables_* (csrc/mcq_tables.cpp); python: neuron_poker_amd/table_driver.py")
    args = ap.parse_args()

    import neuron_poker_amd as npa
    from neuron_poker_amd import table_driver as td

    eng = npa.Engine(int(os.environ.get("LOCAL_RANK", "0")), kernel_times=args.driver == "python")  # plain launches for the native driver
    rng = np.random.default_rng(args.seed)
    pairs = [(.5, -.5), (.8, -.8), (.7, -.7), (.2, -.3)]  # main.py:142-145

    if args.driver == "native":
        from neuron_poker_amd import _lib
        seats = [("equity", c, b) for c, b in pairs] + [("random",), ("random",)]
        tb = _lib.Tables(eng, args.tables, seats, runs=args.runs, seed=args.seed)
        tb.run(20)  # warm-up (first launch, buffers)
        s0 = tb.stats()
        t0 = time.perf_counter()
        done = 0
        while done < args.lock_steps:
            k = min(500, args.lock_steps - done)
            tb.run(k)
            done += k
        wall = time.perf_counter() - t0
        s1 = tb.stats()
        d = {k: s1[k] - s0[k] for k in s1}
        print(json.dumps({"workload": "configs[4]: %d tables x 6 seats, %d lock-steps, %d runs per query" %
                                      (args.tables, args.lock_steps, args.runs), "driver": "native (mcq_tables_run)",
                          "env_steps": d["env_steps"], "episodes_finished": d["episodes"],
                          "equity_queries": d["queries"], "wall_s": wall, "env_steps_per_s": d["env_steps"] / wall,
                          "lock_steps_per_s": args.lock_steps / wall,
                          "equity_queries_per_s": d["queries"] / wall,
                          "ms_per_lock_step": 1e3 * wall / args.lock_steps}))
        return

    def showdown(hands):
        w, _ = eng.showdown(np.array([hands], np.uint8))
        return int(w[0])

    def new_table():
        pol = [td.equity_policy(c, b) for c, b in pairs] + [td.random_policy(rng), td.random_policy(rng)]
        return td.TableSim(pol, initial_stacks=100, small_blind=1, big_blind=2, showdown=showdown,
                           randint=lambda n: int(rng.integers(0, n)))

    t_eq = 0.0
    calls = [0]

    def evaluate(hole, board, npl):
        nonlocal t_eq
        t0 = time.perf_counter()
        q = npa.pack_queries(hole, board, npl, args.runs)
        res = eng.eval_batch(q, seed=args.seed, first_query_id=calls[0])
        calls[0] += len(q)
        t_eq += time.perf_counter() - t0
        return (res["win"] + res["tie"]) / res["runs"]

    tables = [new_table() for _ in range(args.tables)]
    batch = td.TableBatch(tables)
    env_steps = episodes = queries = 0
    t0 = time.perf_counter()
    for _ in range(args.lock_steps):
        queries += batch.step(evaluate)
        for i, p in enumerate(batch.pending):  # a finished table starts its next episode
            if p is None:
                env_steps += batch.tables[i].env_steps
                episodes += 1
                batch.tables[i] = new_table()
                batch.gens[i] = batch.tables[i].episode()
                batch.pending[i] = next(batch.gens[i])
    wall = time.perf_counter() - t0
    env_steps += sum(t.env_steps for t in batch.tables)
    print(json.dumps({"workload": "configs[4]: %d tables x 6 seats, %d lock-steps, %d runs per query" %
                                  (args.tables, args.lock_steps, args.runs), "driver": "python (table_driver.py)",
                      "env_steps": env_steps, "episodes_finished": episodes, "equity_queries": queries,
                      "wall_s": wall, "env_steps_per_s": env_steps / wall, "lock_steps_per_s": args.lock_steps / wall,
                      "equity_call_share": t_eq / wall, "equity_ms_per_lock_step": 1e3 * t_eq / args.lock_steps,
                      "hand_evals_in_equity_calls_per_s": None}))


if __name__ == "__main__":
    main()
