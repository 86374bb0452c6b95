"""Multi-GPU sharding of an equity batch: one process per GPU, ONE integer all-reduce of the tally matrix (RCCL over
xGMI when the backend is "nccl").  Two partitions: many queries are block-distributed over the
ranks; few large queries are cut along their ITERATIONS, every rank taking one contiguous share of each query.

The path shards without any exchange during the computation: every (query, iteration) is independent and the
RNG streams are keyed by (seed, query id, stream), so a rank only needs its block of queries and the id of its
first query.  The all-reduce of a zero-initialised [n, 13] int64 matrix in which each rank has filled its own
rows is exact and order independent (integer sums) and leaves the complete result on every rank, which is what
the caller of a batched get_equity needs.  torch.distributed is plumbing here: process group,
device buffers, the collective.
"""
import numpy as np


def shard_bounds(n, rank, world):
    """Block distribution of n queries: rank r owns [n*r//world, n*(r+1)//world)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    return (n * rank) // world, (n * (rank + 1)) // world


MIN_QUERIES_PER_RANK = 256   # block-distribute the queries when n >= world * 256


def eval_batch_sharded(queries, seed, evaluate, first_query_id=0, group=None, device=None, split=None):
    """Evaluate `queries` (QUERY_DTYPE array, identical on every rank) across the ranks of the default process
    group and return the full [n, 13] uint64 tally matrix on every rank.

    evaluate(q_slice, seed, first_query_id, part=None) -> [m, 13] uint64 tallies of that slice (Engine.eval_batch on
    a GPU rank); part=(p, n) asks for share p of n of every query's iterations.  Query i always runs under query
    id first_query_id + i and iteration j under the same stream, whichever rank computes it, so the result is
    bit-identical to a single-rank call.
    split: 'queries' | 'iterations' | None (= iterations when there are fewer than 256 queries per rank).
    """
    import torch
    import torch.distributed as dist

    n = len(queries)
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    if split is None:
        split = "queries" if n >= world * MIN_QUERIES_PER_RANK else "iterations"
    if split not in ("queries", "iterations"):
        raise ValueError("split must be 'queries' or 'iterations'")
    tallies = torch.zeros((n, 13), dtype=torch.int64, device=device)
    if split == "iterations" and world > 1:
        if n:
            part = np.ascontiguousarray(evaluate(queries, seed, first_query_id, part=(rank, world))).view(np.uint64)
            tallies += torch.from_numpy(part.reshape(n, 13).view(np.int64)).to(tallies.device)
        dist.all_reduce(tallies, op=dist.ReduceOp.SUM, group=group)
        return tallies.cpu().numpy().view(np.uint64)
    lo, hi = shard_bounds(n, rank, world)
    if hi > lo:
        part = np.ascontiguousarray(evaluate(queries[lo:hi], seed, first_query_id + lo)).view(np.uint64)
        tallies[lo:hi] = torch.from_numpy(part.reshape(hi - lo, 13).view(np.int64)).to(tallies.device)
    if world > 1:
        dist.all_reduce(tallies, op=dist.ReduceOp.SUM, group=group)
    return tallies.cpu().numpy().view(np.uint64)
