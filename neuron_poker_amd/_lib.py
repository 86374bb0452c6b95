"""ctypes binding of libmcq_hip.so (C ABI: include/mcq.h).  No torch, no cffi.

The library is built in-tree by `python -m neuron_poker_amd.build` (or __graft_entry__.build()).  If it is
missing, or no HIP device can be opened, this module raises -- it never falls back to a CPU implementation.
"""
import ctypes as C
import importlib.util
import os
import struct
import sys
import threading

import numpy as np

MODE_PHILOX = 0
MODE_REPLAY_MT19937 = 1
MCQ_EINVAL, MCQ_EDEVICE, MCQ_ENOMEM, MCQ_EBUSY = -1, -2, -3, -4

QUERY_DTYPE = np.dtype([("hole", "u1", (2,)), ("board", "u1", (5,)), ("n_board", "u1"), ("n_players", "u1"),
                        ("reserved", "u1", (3,)), ("runs", "<u4")])
RESULT_DTYPE = np.dtype([("runs", "<u8"), ("passes", "<u8"), ("win", "<u8"), ("tie", "<u8"),
                         ("by_type", "<u8", (9,))])
KNOWN_HAND_DTYPE = np.dtype([("cards", "u1", (2,)), ("is_range", "u1"), ("reserved", "u1"), ("range", "<u4", (6,))])
MAX_KNOWN = 9
QUERY_EXT_DTYPE = np.dtype([("ghost", "u1", (2,)), ("hero_is_range", "u1"), ("n_known", "u1"), ("opp_range", "<u4", (6,)),
                            ("hero_range", "<u4", (6,)), ("known", KNOWN_HAND_DTYPE, (MAX_KNOWN,))])
TABLES_CONFIG_DTYPE = np.dtype([("n_tables", "<u4"), ("n_seats", "<u4"), ("runs", "<u4"), ("max_raises", "<u4"),
                                ("initial_stacks", "<f8"), ("small_blind", "<f8"), ("big_blind", "<f8"),
                                ("seed", "<u8"), ("seat_kind", "u1", (10,)), ("reserved", "u1", (6,)),
                                ("min_call_equity", "<f8", (10,)), ("min_bet_equity", "<f8", (10,))])
assert TABLES_CONFIG_DTYPE.itemsize == 224
assert QUERY_DTYPE.itemsize == 16 and RESULT_DTYPE.itemsize == 104 and QUERY_EXT_DTYPE.itemsize == 304
ALL_CLASSES = np.array([0xFFFFFFFF] * 5 + [0x1FF], np.uint32)  # 169 bits

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_lock = threading.Lock()


class McqError(RuntimeError):
    """HIP / allocation failure reported by libmcq_hip.so (MCQ_EDEVICE, MCQ_ENOMEM)."""


class McqBusyError(McqError):
    """MCQ_EBUSY: a second call on a context (Engine / MultiEngine) while one is running on it -- a context allows ONE
    call in flight; use one Engine per thread (default_engine() does)."""


def library_path():
    return os.environ.get("MCQ_LIBRARY", os.path.join(_HERE, "libmcq_hip.so"))


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP/HSA runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, the same as /opt/rocm's); if libmcq_hip.so bound the system copy and torch were imported
    afterwards, the second runtime would find no GPU.  So when a torch wheel with a bundled runtime is
    installed, that copy is mapped first and libmcq_hip.so resolves to it by SONAME.  MCQ_HIP_RUNTIME=system
    opts out (processes that never import torch)."""
    if os.environ.get("MCQ_HIP_RUNTIME", "auto") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """Load libmcq_hip.so and declare the prototypes of include/mcq.h.  Loud failure if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise ImportError("%s not found: build it with `python -m neuron_poker_amd.build` (hipcc, gfx950). "
                              "neuron_poker_amd has no CPU fallback." % path)
        _share_hip_runtime_with_torch()
        L = C.CDLL(path)
        vp, u64, sz = C.c_void_p, C.c_uint64, C.c_size_t
        L.mcq_device_count.argtypes = []
        L.mcq_device_count.restype = C.c_int
        L.mcq_create.argtypes = [C.c_int, C.c_int]
        L.mcq_create.restype = vp
        L.mcq_destroy.argtypes = [vp]
        L.mcq_destroy.restype = None
        L.mcq_eval_batch.argtypes = [vp, vp, sz, u64, u64, C.c_int, vp]
        L.mcq_eval_batch.restype = C.c_int
        L.mcq_eval_batch_part.argtypes = [vp, vp, sz, u64, u64, C.c_uint32, C.c_uint32, vp]
        L.mcq_eval_batch_part.restype = C.c_int
        L.mcq_eval_one.argtypes = [vp, vp, u64, C.c_int, vp]
        L.mcq_eval_one.restype = C.c_int
        L.mcq_eval_batch_ext.argtypes = [vp, vp, vp, sz, u64, u64, C.c_int, vp]
        L.mcq_eval_batch_ext.restype = C.c_int
        L.mcq_eval_batch_numpy_stream.argtypes = [vp, vp, sz, vp, vp, vp]
        L.mcq_eval_batch_numpy_stream.restype = C.c_int
        L.mcq_eval_batch_device.argtypes = [vp, vp, sz, u64, u64, vp, vp]
        L.mcq_eval_batch_device.restype = C.c_int
        L.mcq_eval_batch_device_small.argtypes = [vp, vp, sz, u64, u64, vp, vp]
        L.mcq_eval_batch_device_small.restype = C.c_int
        L.mcq_showdown.argtypes = [vp, vp, sz, C.c_int, vp, vp, vp]
        L.mcq_showdown.restype = C.c_int
        L.mcq_exact_batch.argtypes = [vp, vp, sz, C.c_int, vp]
        L.mcq_exact_batch.restype = C.c_int
        L.mcq_set_dealing_law.argtypes = [vp, C.c_int]
        L.mcq_set_dealing_law.restype = C.c_int
        L.mcq_set_kernel_timing.argtypes = [vp, C.c_int]
        L.mcq_set_kernel_timing.restype = C.c_int
        L.mcq_kernel_times.argtypes = [vp, vp, C.c_int]
        L.mcq_kernel_times.restype = C.c_int
        L.mcq_last_kernel_ms.argtypes = [vp]
        L.mcq_last_kernel_ms.restype = C.c_float
        L.mcq_tables_create.argtypes = [vp, vp]
        L.mcq_tables_create.restype = vp
        L.mcq_tables_destroy.argtypes = [vp]
        L.mcq_tables_destroy.restype = None
        L.mcq_tables_begin.argtypes = [vp, vp]
        L.mcq_tables_begin.restype = sz
        L.mcq_tables_resume.argtypes = [vp, vp]
        L.mcq_tables_resume.restype = C.c_int
        L.mcq_tables_run.argtypes = [vp, C.c_uint32, vp]
        L.mcq_tables_run.restype = C.c_int
        L.mcq_tables_stats.argtypes = [vp, vp]
        L.mcq_tables_stats.restype = None
        L.mcq_tables_state.argtypes = [vp, C.c_uint32, vp, vp]
        L.mcq_tables_state.restype = C.c_int
        L.mcq_multi_create.argtypes = [vp, C.c_int, C.c_int]
        L.mcq_multi_create.restype = vp
        L.mcq_multi_destroy.argtypes = [vp]
        L.mcq_multi_destroy.restype = None
        L.mcq_multi_eval_batch.argtypes = [vp, vp, sz, u64, u64, C.c_int, vp]
        L.mcq_multi_eval_batch.restype = C.c_int
        L.mcq_multi_eval_batch_device.argtypes = [vp, vp, sz, u64, u64, C.c_int, vp]
        L.mcq_multi_eval_batch_device.restype = C.c_int
        L.mcq_multi_set_dealing_law.argtypes = [vp, C.c_int]
        L.mcq_multi_set_dealing_law.restype = C.c_int
        L.mcq_multi_info.argtypes = [vp, vp]
        L.mcq_multi_info.restype = C.c_int
        L.mcq_multi_times.argtypes = [vp, vp]
        L.mcq_multi_times.restype = C.c_int
        L.mcq_last_error.argtypes = []
        L.mcq_last_error.restype = C.c_char_p
        L.mcq_version.argtypes = [C.POINTER(C.c_int)] * 3
        L.mcq_version.restype = None
        _lib = L
        return _lib


def _raise(rc):
    msg = (load_library().mcq_last_error() or b"").decode("utf-8", "replace")
    if rc == MCQ_EINVAL:
        raise ValueError(msg)
    if rc == MCQ_EBUSY:
        raise McqBusyError(msg)
    raise McqError("libmcq_hip error %d: %s" % (rc, msg))


def pack_queries(hole, board, n_players, runs):
    """Build mcq_query records.  hole [B,2] card ids; board [B,5] card ids, 0xFF = absent (any position: the
    known cards are left-packed here); n_players, runs scalar or [B]."""
    hole = np.asarray(hole, dtype=np.uint8).reshape(-1, 2)
    B = len(hole)
    board = np.asarray(board, dtype=np.uint8).reshape(B, 5)
    q = np.zeros(B, QUERY_DTYPE)
    q["hole"] = hole
    present = board != 255
    order = np.argsort(~present, axis=1, kind="stable")
    packed = np.take_along_axis(board, order, axis=1)
    nb = present.sum(1).astype(np.uint8)
    packed[np.arange(5)[None, :] >= nb[:, None]] = 0
    q["board"] = packed
    q["n_board"] = nb
    npl = np.broadcast_to(np.asarray(n_players), (B,))
    if (npl < 0).any() or (npl > 255).any():
        raise ValueError("n_players out of range")
    q["n_players"] = npl.astype(np.uint8)
    r = np.broadcast_to(np.asarray(runs), (B,))
    if (r < 0).any() or (r > 0xffffffff).any():
        raise ValueError("runs out of range")
    q["runs"] = r.astype(np.uint32)
    return q


def pack_query_one(hole, board, n_players, runs):
    """One mcq_query record without the numpy machinery of pack_queries (the per-call path of get_equity):
    hole = two card ids, board = up to five card ids (left-packed as given)."""
    nb = len(board)
    if len(hole) != 2 or nb > 5:
        raise ValueError("two hole cards and at most five table cards")
    try:
        raw = struct.pack("<2B5BBB3xI", hole[0], hole[1], *(list(board) + [0] * (5 - nb)), nb, n_players, runs)
    except struct.error as e:
        raise ValueError("card id, n_players or runs out of range: %s" % e)
    return np.frombuffer(bytearray(raw), QUERY_DTYPE)  # writable


def class_bit(name):
    """Bit of a preflop class string ('AKS', 'KAO', 'TT', ...) in a 169-bit range set; None if the string can never
    equal what get_two_short_notation produces (tools/montecarlo_python.py:24-34)."""
    from .cards import RANKS
    if not isinstance(name, str) or len(name) < 2:
        return None
    r1, r2 = RANKS.find(name[0]), RANKS.find(name[1])
    if r1 < 0 or r2 < 0:
        return None
    if r1 == r2:
        return 14 * r1 if len(name) == 2 else None
    if len(name) != 3 or name[2] not in "SO":
        return None
    lo, hi = min(r1, r2), max(r1, r2)
    return 13 * lo + hi if name[2] == "S" else 13 * hi + lo


def range_bits(classes):
    """169-bit set (6 little-endian uint32 words) of a collection of preflop class strings."""
    w = np.zeros(6, np.uint32)
    for c in classes:
        b = class_bit(c)
        if b is not None:
            w[b >> 5] |= np.uint32(1 << (b & 31))
    return w


def pack_query_ext(n, ghost=None, known2=None, hero_range=None, opp_range=None, known=None):
    """n mcq_query_ext records with the same settings.  ghost = two card ids or None; hero_range / opp_range = 6-word
    sets (range_bits) or None (hero given as cards / every class); known = the further known hands in the order of
    original_player_card_list, each two card ids or a 6-word set (known2 = one hand of two cards, kept for brevity)."""
    e = np.zeros(n, QUERY_EXT_DTYPE)
    e["ghost"] = 255 if ghost is None else np.asarray(ghost, np.uint8)
    e["hero_is_range"] = 0 if hero_range is None else 1
    e["hero_range"] = 0 if hero_range is None else np.asarray(hero_range, np.uint32)
    e["opp_range"] = ALL_CLASSES if opp_range is None else np.asarray(opp_range, np.uint32)
    hands = ([known2] if known2 is not None else []) + list(known or [])
    if len(hands) > MAX_KNOWN:
        raise ValueError("at most %d known hands besides the hero" % MAX_KNOWN)
    e["n_known"] = len(hands)
    for k, h in enumerate(hands):
        h = np.asarray(h)
        if h.size == 2:
            e["known"]["cards"][:, k] = h.astype(np.uint8)
        elif h.size == 6:
            e["known"]["is_range"][:, k] = 1
            e["known"]["range"][:, k] = h.astype(np.uint32)
        else:
            raise ValueError("a known hand is two card ids or a 6-word range set")
    return e


class Engine:
    """One mcq_ctx: an equity engine bound to one GPU.  Not re-entrant: one call in flight per engine -- a second thread
    calling into the same Engine meanwhile gets McqBusyError (the library checks).  One Engine per thread."""

    def __init__(self, device=0, kernel_times=False):
        self._lib = load_library()
        self._ctx = self._lib.mcq_create(int(device), 0)
        if not self._ctx:
            msg = (self._lib.mcq_last_error() or b"").decode("utf-8", "replace")
            raise McqError("mcq_create(device=%d) failed: %s" % (device, msg))
        self.device = int(device)
        if kernel_times:
            self.set_kernel_timing(True)

    def set_kernel_timing(self, on):
        """Timestamp every evaluation-kernel launch (kernel_times / last_kernel_ms); off by default: it costs a small
        query about 6 us of its call time."""
        rc = self._lib.mcq_set_kernel_timing(self._ctx, 1 if on else 0)
        if rc:
            _raise(rc)

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.mcq_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_dealing_law(self, law):
        """'reference' (default: the Python reference's law incl. its index bias) or 'uniform' (unbiased)."""
        code = {"reference": 0, "uniform": 1, 0: 0, 1: 1}.get(law)
        if code is None:
            raise ValueError("law must be 'reference' or 'uniform'")
        rc = self._lib.mcq_set_dealing_law(self._ctx, code)
        if rc:
            _raise(rc)

    def eval_batch(self, queries, seed, first_query_id=0, mode=MODE_PHILOX, part=None):
        """queries: array of QUERY_DTYPE (host).  -> array of RESULT_DTYPE.
        part=(p, n): only share p of n of every query's iterations (mcq_eval_batch_part; the rows of all shares
        add up to the unsplit result)."""
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE).reshape(-1)
        out = np.empty(len(q), RESULT_DTYPE)   # every row is written by the library on success; on failure we raise
        if part is not None:
            if mode != MODE_PHILOX:
                raise ValueError("only the production mode can split the iterations of a query")
            rc = self._lib.mcq_eval_batch_part(self._ctx, q.ctypes.data, len(q), int(seed) & (2 ** 64 - 1),
                                               int(first_query_id) & (2 ** 64 - 1), int(part[0]), int(part[1]),
                                               out.ctypes.data)
        else:
            rc = self._lib.mcq_eval_batch(self._ctx, q.ctypes.data, len(q), int(seed) & (2 ** 64 - 1),
                                          int(first_query_id) & (2 ** 64 - 1), int(mode), out.ctypes.data)
        if rc:
            _raise(rc)
        return out

    def exact(self, queries, law="reference"):
        """Exact enumeration (1..3 players; `runs` of the queries is ignored).  -> RESULT_DTYPE rows of integer
        weights: runs = total weight, equity = (win + tie) / runs exactly."""
        code = {"reference": 0, "uniform": 1, 0: 0, 1: 1}.get(law)
        if code is None:
            raise ValueError("law must be 'reference' or 'uniform'")
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE).reshape(-1)
        out = np.zeros(len(q), RESULT_DTYPE)
        rc = self._lib.mcq_exact_batch(self._ctx, q.ctypes.data, len(q), code, out.ctypes.data)
        if rc:
            _raise(rc)
        return out

    def eval_batch_ext(self, queries, ext, seed, first_query_id=0, mode=MODE_PHILOX):
        """queries with one QUERY_EXT_DTYPE record each (ranges, hero range, ghost cards, second known hand)."""
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE).reshape(-1)
        e = np.ascontiguousarray(ext, dtype=QUERY_EXT_DTYPE).reshape(-1)
        if len(e) != len(q):
            raise ValueError("one mcq_query_ext per query")
        out = np.zeros(len(q), RESULT_DTYPE)
        rc = self._lib.mcq_eval_batch_ext(self._ctx, q.ctypes.data, e.ctypes.data, len(q), int(seed) & (2 ** 64 - 1),
                                          int(first_query_id) & (2 ** 64 - 1), int(mode), out.ctypes.data)
        if rc:
            _raise(rc)
        return out

    def eval_batch_numpy_stream(self, queries):
        """Parity mode on numpy's GLOBAL random state: the queries consume np.random's MT19937 stream in order,
        exactly as consecutive reference calls do, and np.random is left where those calls would leave it."""
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE).reshape(-1)
        out = np.zeros(len(q), RESULT_DTYPE)
        st = np.random.get_state()
        key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        pos = C.c_uint32(int(st[2]))
        rc = self._lib.mcq_eval_batch_numpy_stream(self._ctx, q.ctypes.data, len(q), key.ctypes.data, C.byref(pos),
                                                   out.ctypes.data)
        if rc:
            _raise(rc)
        np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
        return out

    def eval_batch_device(self, d_queries, n, seed, d_results, first_query_id=0, stream=None):
        """Device-resident entry: d_queries / d_results are raw device pointers (ints), stream a hipStream_t
        handle (int) or None.  Asynchronous."""
        rc = self._lib.mcq_eval_batch_device(self._ctx, int(d_queries), int(n), int(seed) & (2 ** 64 - 1),
                                             int(first_query_id) & (2 ** 64 - 1), int(d_results),
                                             int(stream) if stream else None)
        if rc:
            _raise(rc)

    def eval_batch_device_small(self, d_queries, n, seed, d_results, first_query_id=0, stream=None):
        """eval_batch_device for queries of at most 8192 iterations each: ONE kernel launch, no pricing kernel, no
        atomics (a longer query gets runs = 0, passes = 2**64 - 1 like an invalid one)."""
        rc = self._lib.mcq_eval_batch_device_small(self._ctx, int(d_queries), int(n), int(seed) & (2 ** 64 - 1),
                                                   int(first_query_id) & (2 ** 64 - 1), int(d_results),
                                                   int(stream) if stream else None)
        if rc:
            _raise(rc)

    def showdown(self, hands, want_keys=False):
        """hands [T, P, 7] card ids -> (winner[T], winner_type[T][, keys[T, P]])."""
        h = np.ascontiguousarray(hands, dtype=np.uint8)
        if h.ndim != 3 or h.shape[2] != 7:
            raise ValueError("hands must have shape [tables, players, 7]")
        T, P = h.shape[0], h.shape[1]
        win = np.zeros(T, np.uint8)
        wt = np.zeros(T, np.uint8)
        keys = np.zeros((T, P), np.uint32) if want_keys else None
        rc = self._lib.mcq_showdown(self._ctx, h.ctypes.data, T, P, win.ctypes.data, wt.ctypes.data,
                                    keys.ctypes.data if want_keys else None)
        if rc:
            _raise(rc)
        return (win, wt, keys) if want_keys else (win, wt)

    def kernel_times(self, max_n=64):
        """Durations (ms, HIP events on the launch stream) of the most recent evaluation-kernel launches."""
        ms = np.zeros(max(int(max_n), 1), np.float32)
        n = self._lib.mcq_kernel_times(self._ctx, ms.ctypes.data, int(max_n))
        if n < 0:
            _raise(n)
        return ms[:n]

    @property
    def last_kernel_ms(self):
        return float(self._lib.mcq_last_kernel_ms(self._ctx))


PARTITION_AUTO, PARTITION_QUERIES, PARTITION_ITERATIONS = 0, 1, 2
_PARTITIONS = {None: 0, "auto": 0, "queries": 1, "iterations": 2, 0: 0, 1: 1, 2: 2}


class MultiEngine:
    """mcq_multi: one process driving several GPUs of a node (include/mcq.h).  The batch is partitioned over
    shards, ONE RCCL all-reduce of the integer tally matrix joins them; results are bit-identical to Engine.eval_batch.

    devices: one HIP device ordinal per shard (a device may repeat); None = every visible device once."""

    def __init__(self, devices=None):
        self._lib = load_library()
        if devices is None:
            n = self._lib.mcq_device_count()
            if n <= 0:
                raise McqError("no HIP device visible: " + (self._lib.mcq_last_error() or b"").decode("utf-8", "replace"))
            devices = list(range(n))
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        self._m = self._lib.mcq_multi_create(devs.ctypes.data, len(devs), 0)
        if not self._m:
            msg = (self._lib.mcq_last_error() or b"").decode("utf-8", "replace")
            raise McqError("mcq_multi_create(%s) failed: %s" % ([int(d) for d in devs], msg))
        self.devices = [int(d) for d in devs]
        self.lock = threading.Lock()   # the shim's shared MultiEngines serialise their callers on it

    def close(self):
        if getattr(self, "_m", None):
            self._lib.mcq_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_dealing_law(self, law):
        code = {"reference": 0, "uniform": 1, 0: 0, 1: 1}.get(law)
        if code is None:
            raise ValueError("law must be 'reference' or 'uniform'")
        rc = self._lib.mcq_multi_set_dealing_law(self._m, code)
        if rc:
            _raise(rc)

    def eval_batch(self, queries, seed, first_query_id=0, partition=None):
        """queries: array of QUERY_DTYPE (host) -> array of RESULT_DTYPE; partition 'auto' | 'queries' | 'iterations'."""
        if partition not in _PARTITIONS:
            raise ValueError("partition must be 'auto', 'queries' or 'iterations'")
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE).reshape(-1)
        out = np.zeros(len(q), RESULT_DTYPE)
        rc = self._lib.mcq_multi_eval_batch(self._m, q.ctypes.data, len(q), int(seed) & (2 ** 64 - 1),
                                            int(first_query_id) & (2 ** 64 - 1), _PARTITIONS[partition], out.ctypes.data)
        if rc:
            _raise(rc)
        return out

    def eval_batch_device(self, d_queries, n, seed, d_results, first_query_id=0, partition="auto"):
        """Queries and results resident in HBM: d_queries / d_results = one raw device pointer (int) per shard, on that
        shard's device -- the shard's block of the n queries (partition "queries") or all of them ("iterations"), and
        room for the complete [n] result matrix, which every shard's buffer holds afterwards.  Blocks until done."""
        k = len(self.devices)
        if len(d_queries) != k or len(d_results) != k:
            raise ValueError("one device pointer pair per shard")
        qp = (C.c_void_p * k)(*[int(p) if p else None for p in d_queries])
        rp = (C.c_void_p * k)(*[int(p) if p else None for p in d_results])
        rc = self._lib.mcq_multi_eval_batch_device(self._m, qp, int(n), int(seed) & (2 ** 64 - 1),
                                                   int(first_query_id) & (2 ** 64 - 1), _PARTITIONS[partition], rp)
        if rc:
            _raise(rc)

    @property
    def info(self):
        v = np.zeros(4, np.int32)
        rc = self._lib.mcq_multi_info(self._m, v.ctypes.data)
        if rc:
            _raise(rc)
        return {"shards": int(v[0]), "devices": int(v[1]), "rccl_version": int(v[2]),
                "last_partition": {0: "auto", 1: "queries", 2: "iterations"}[int(v[3])]}

    @property
    def last_times_ms(self):
        v = np.zeros(3, np.float32)
        rc = self._lib.mcq_multi_times(self._m, v.ctypes.data)
        if rc:
            _raise(rc)
        return {"kernel_max": float(v[0]), "all_reduce": float(v[1]), "call": float(v[2])}


class Tables:
    """mcq_tables: T Hold'em tables advanced in lock-step by the native driver (include/mcq.h, BASELINE configs[4]).

    seats: one entry per seat -- ("equity", min_call_equity, min_bet_equity) or ("random",).
    engine=None gives a driver without GPU: only begin()/resume() work (used to pin the rules on the CPU).
    The tables are independent (own generator each): results do not depend on `threads`."""

    def __init__(self, engine, n_tables, seats, runs=1000, initial_stacks=100, small_blind=1, big_blind=2,
                 max_raises=2, seed=0, threads=0, overlap=True, calculate_equity=False):
        self._lib = load_library()
        self._engine = engine
        cfg = np.zeros(1, TABLES_CONFIG_DTYPE)
        cfg["n_tables"], cfg["n_seats"], cfg["runs"], cfg["max_raises"] = n_tables, len(seats), runs, max_raises
        cfg["initial_stacks"], cfg["small_blind"], cfg["big_blind"] = initial_stacks, small_blind, big_blind
        cfg["seed"] = int(seed) & (2 ** 64 - 1)
        cfg["reserved"][0, 0] = threads          # host threads stepping the tables; 0 = automatic
        cfg["reserved"][0, 1] = 0 if overlap else 1   # run(): groups of tables on streams of their own (same results either way)
        cfg["reserved"][0, 2] = 1 if calculate_equity else 0   # three more queries per observation (env.py:248-256)
        if len(seats) > 10:
            raise ValueError("at most 10 seats")
        for i, s in enumerate(seats):
            if s[0] == "equity":
                cfg["min_call_equity"][0, i], cfg["min_bet_equity"][0, i] = s[1], s[2]
            elif s[0] == "random":
                cfg["seat_kind"][0, i] = 1
            else:
                raise ValueError("seat kind must be 'equity' or 'random'")
        self.n_tables, self.n_seats = int(n_tables), len(seats)
        self._t = self._lib.mcq_tables_create(engine._ctx if engine is not None else None, cfg.ctypes.data)
        if not self._t:
            raise ValueError((self._lib.mcq_last_error() or b"").decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "_t", None):
            self._lib.mcq_tables_destroy(self._t)
            self._t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin(self):
        """-> the pending query of every table (QUERY_DTYPE[n_tables])."""
        q = np.zeros(self.n_tables, QUERY_DTYPE)
        self._lib.mcq_tables_begin(self._t, q.ctypes.data)
        return q

    def resume(self, equity):
        e = np.ascontiguousarray(equity, np.float64)
        if e.shape != (self.n_tables,):
            raise ValueError("one equity per table")
        rc = self._lib.mcq_tables_resume(self._t, e.ctypes.data)
        if rc:
            _raise(rc)

    def run(self, lock_steps):
        """lock_steps rounds of (queries of all tables -> ONE GPU batch -> every table acts).  -> stats()."""
        st = np.zeros(3, np.uint64)
        rc = self._lib.mcq_tables_run(self._t, int(lock_steps), st.ctypes.data)
        if rc:
            _raise(rc)
        return {"env_steps": int(st[0]), "queries": int(st[1]), "episodes": int(st[2])}

    def stats(self):
        st = np.zeros(3, np.uint64)
        self._lib.mcq_tables_stats(self._t, st.ctypes.data)
        return {"env_steps": int(st[0]), "queries": int(st[1]), "episodes": int(st[2])}

    def state(self, table):
        stacks = np.zeros(self.n_seats, np.float64)
        info = np.zeros(8, np.int32)
        rc = self._lib.mcq_tables_state(self._t, int(table), stacks.ctypes.data, info.ctypes.data)
        if rc:
            _raise(rc)
        keys = ["stage", "current", "winner", "episodes", "env_steps", "queries", "legal", "phase"]
        d = dict(zip(keys, (int(x) for x in info)))
        d["stacks"] = stacks
        return d


_tls = threading.local()      # .engine: the calling thread's default engine
_default_law = "reference"    # dealing law of the default engines (montecarlo_hip.configure(dealing=...))
_default_engines = []         # weak references to every live default engine, for set_default_dealing_law


def default_engine():
    """The calling THREAD's engine on device $MCQ_DEVICE (else $LOCAL_RANK, else 0), created at the thread's first use.
    A context allows one call in flight (include/mcq.h), and ctypes drops the GIL for the duration of a call, so threads
    must not share one: every thread gets its own (own stream, own staging buffers), freed when the thread ends."""
    eng = getattr(_tls, "engine", None)
    if eng is None:
        dev = int(os.environ.get("MCQ_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = load_library().mcq_device_count()
        if n <= 0:
            raise McqError("no HIP device visible: neuron_poker_amd needs an AMD GPU (no CPU fallback). " +
                           (load_library().mcq_last_error() or b"").decode("utf-8", "replace"))
        eng = Engine(dev % n)
        import weakref
        with _lock:
            if _default_law != "reference":
                eng.set_dealing_law(_default_law)
            _default_engines[:] = [r for r in _default_engines if r() is not None]
            _default_engines.append(weakref.ref(eng))
        _tls.engine = eng
    return eng


def default_dealing_law():
    return _default_law


def set_default_dealing_law(law):
    """Dealing law of every thread's default engine, present and future.  Call it while no equity call is running on
    them (an engine in the middle of a call answers McqBusyError)."""
    global _default_law
    if law not in ("reference", "uniform"):
        raise ValueError("law must be 'reference' or 'uniform'")
    with _lock:
        _default_law = law
        for r in _default_engines:
            e = r()
            if e is not None and getattr(e, "_ctx", None):
                e.set_dealing_law(law)
