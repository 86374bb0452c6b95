"""Equity of states that already live in HBM as torch tensors -- no host round trip.

For pipelines that keep observations on the GPU (batched self-play, a policy network consuming the equity): the
queries are packed by torch ops on the current stream, evaluated by mcq_eval_batch_device on that same stream and
the result stays a device tensor, so nothing synchronises with the host.  torch is plumbing here (device memory,
streams); the evaluation is the library's HIP kernel.  Small batches (<= 1024 states) are cut into sub-tasks by
the prep kernel on the device.

    equity, tallies = get_equity_batch_torch(hole, board, n_players, runs, seed=1)

hole [B,2] uint8 card ids, board [B,5] uint8 (255 = empty slot, any position), n_players int or [B] uint8,
runs int or [B] int32 -- all on the same CUDA (HIP) device.  equity: float64 [B]; tallies: int64 [B,13] with
columns runs, passes, win, tie, by_type[9] (include/mcq.h mcq_result).  Query i runs under id first_query_id + i:
the numbers equal Engine.eval_batch / get_equity_batch on the same inputs, bit for bit.
"""
import torch

from . import _lib

_engines = {}


def _engine(index):
    if index not in _engines:
        _engines[index] = _lib.Engine(index)
    return _engines[index]


def pack_queries_torch(hole, board, n_players, runs):
    """mcq_query records ([B,16] uint8, on the inputs' device) from tensors; the table cards are left-packed."""
    dev = hole.device
    B = hole.shape[0]
    hole = hole.to(torch.uint8).reshape(B, 2)
    board = board.to(torch.uint8).reshape(B, 5)
    present = board != 255
    order = torch.argsort((~present).to(torch.uint8), dim=1, stable=True)
    packed = torch.gather(board, 1, order)
    nb = present.sum(1).to(torch.uint8)
    packed = torch.where(torch.arange(5, device=dev)[None, :] < nb[:, None], packed, torch.zeros_like(packed))
    q = torch.zeros((B, 16), dtype=torch.uint8, device=dev)
    q[:, 0:2] = hole
    q[:, 2:7] = packed
    q[:, 7] = nb
    q[:, 8] = torch.as_tensor(n_players, device=dev).to(torch.uint8).expand(B)
    r = torch.as_tensor(runs, device=dev).to(torch.int64).expand(B)
    for k in range(4):                                   # little-endian u32
        q[:, 12 + k] = ((r >> (8 * k)) & 255).to(torch.uint8)
    return q


def get_equity_batch_torch(hole, board, n_players, runs, seed=0, first_query_id=0, engine=None):
    if not hole.is_cuda:
        raise ValueError("the tensors must live on the GPU (use get_equity_batch for host arrays)")
    eng = engine or _engine(hole.device.index if hole.device.index is not None else torch.cuda.current_device())
    q = pack_queries_torch(hole, board, n_players, runs)
    out = torch.empty((q.shape[0], 13), dtype=torch.int64, device=q.device)
    eng.eval_batch_device(q.data_ptr(), q.shape[0], seed, out.data_ptr(), first_query_id=first_query_id,
                          stream=torch.cuda.current_stream(q.device).cuda_stream)
    q.record_stream(torch.cuda.current_stream(q.device))
    equity = (out[:, 2] + out[:, 3]).to(torch.float64) / out[:, 0].clamp(min=1).to(torch.float64)
    return equity, out
