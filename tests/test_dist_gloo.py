"""The N > 1 path on CPU: world_size-2 gloo process group, queries block-sharded or iterations split, one integer
all-reduce.
The per-rank evaluator here is the ORACLE (tests may use it as the checker's stand-in; the product's GPU
evaluator is exercised with the same sharding in tests/test_gpu_parity.py::test_shard_invariance...)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from neuron_poker_amd import sharding  # noqa: E402
import neuron_poker_amd as npa  # noqa: E402
from neuron_poker_amd._lib import pack_queries  # noqa: E402
from oracle import oracle as O  # noqa: E402


def _batch():
    g = np.random.default_rng(65536)  # BASELINE configs[3] generator, scaled down
    B = 37
    hole, board = [], []
    for i in range(B):
        b = 3 if i % 2 == 0 else 4
        cards = g.choice(52, 2 + b, replace=False)
        hole.append(cards[:2])
        board.append(list(cards[2:]) + [255] * (5 - b))
    runs = [700, 5000, 1024, 3073][:] * 10
    return pack_queries(hole, board, 6, runs[:B])


def _oracle_eval(mode):
    def f(q, seed, first, part=None):
        raw = np.ascontiguousarray(q).view(np.uint8).reshape(-1, 16)
        if part is not None:
            return O.run_batch_part(mode, raw, seed, first, part[0], part[1])
        return O.run_batch(mode, raw, seed, first)
    return f


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q = _batch()
        for mode in (O.MODE_CTR, O.MODE_MT):   # the parity mode cannot split a query's iterations
            t = sharding.eval_batch_sharded(q, 77, _oracle_eval(mode), first_query_id=1000, split="queries")
            np.save(os.path.join(out_dir, "t_%d_%d.npy" % (mode, rank)), t)
        t = sharding.eval_batch_sharded(q, 77, _oracle_eval(O.MODE_CTR), first_query_id=1000)   # 37 < 2 * 256
        np.save(os.path.join(out_dir, "t_iter_%d.npy" % rank), t)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 4096, 65536):
        for w in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)


def test_two_rank_gloo_allreduce_equals_single_rank(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    q = _batch()
    for mode in (O.MODE_CTR, O.MODE_MT):
        exp = _oracle_eval(mode)(q, 77, 1000)
        for r in range(world):
            got = np.load(os.path.join(str(tmp_path), "t_%d_%d.npy" % (mode, r)))
            assert np.array_equal(got, exp), (mode, r)
    exp = _oracle_eval(O.MODE_CTR)(q, 77, 1000)
    for r in range(world):   # iterations split over the ranks: same tallies, runs included
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "t_iter_%d.npy" % r)), exp), r


def test_iteration_shares_tile_a_query():
    q = _batch()[:8]
    raw = np.ascontiguousarray(q).view(np.uint8).reshape(-1, 16)
    whole = O.run_batch(O.MODE_CTR, raw, 5, 10)
    for n_parts in (1, 2, 3, 8):
        parts = [O.run_batch_part(O.MODE_CTR, raw, 5, 10, p, n_parts) for p in range(n_parts)]
        assert np.array_equal(sum(parts), whole), n_parts


def test_single_process_without_group():
    q = _batch()[:5]
    t = sharding.eval_batch_sharded(q, 3, _oracle_eval(O.MODE_CTR))
    assert np.array_equal(t, _oracle_eval(O.MODE_CTR)(q, 3, 0))


def test_launcher_hands_a_failing_rank_through_and_prints_no_json_line():
    """`python bench.py --gpus 2` starts its ranks as a child torch.distributed.run.  Where the ranks cannot run (no GPU
    in this container: the engine refuses to start without a HIP device -- there is no CPU fallback) the command must
    fail with the child's exit code and must not print a JSON result line."""
    import subprocess
    import sys
    if npa.load_library().mcq_device_count() > 0:
        pytest.skip("a GPU is present: the ranks would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--states", "8", "--iters", "100", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")], r.stdout[-500:]
