"""Test-only host build of the product's lane arithmetic (see hostsim.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MCQ_HOSTSIM_SO", os.path.join(_HERE, "libmcq_hostsim.so"))   # override: tests/sanitize_cpu.sh
_SRCS = [os.path.join(_HERE, "hostsim.cpp"),
         os.path.join(_HERE, "..", "..", "neuron_poker_amd", "csrc", "mcq_device.hpp"),
         os.path.join(_HERE, "..", "..", "neuron_poker_amd", "csrc", "mcq_replay.hpp"),
         os.path.join(_HERE, "..", "..", "neuron_poker_amd", "csrc", "mcq_exact.hpp"),
         os.path.join(_HERE, "..", "..", "neuron_poker_amd", "csrc", "mcq_mt.hpp"),
         os.path.join(_HERE, "..", "..", "neuron_poker_amd", "csrc", "mcq_mt_ext.hpp"),
         os.path.join(_HERE, "..", "..", "neuron_poker_amd", "csrc", "mcq_mt_blocks.hpp"),
         os.path.join(_HERE, "..", "..", "neuron_poker_amd", "csrc", "mcq_layout.hpp"),
         os.path.join(_HERE, "..", "..", "include", "mcq.h")]
_lib = None


def lib():
    global _lib
    if _lib is None:
        if "MCQ_HOSTSIM_SO" not in os.environ and (not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRCS)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-Wno-maybe-uninitialized", "-shared", "-fPIC", "-o", _SO, _SRCS[0]])
        L = C.CDLL(_SO)
        L.hs_select_pop.restype = C.c_uint32
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def eval7(cards):
    cards = np.ascontiguousarray(cards, np.uint8).reshape(-1, 7)
    keys = np.zeros(len(cards), np.uint32)
    lib().hs_eval7(_p(cards, C.c_uint8), C.c_size_t(len(cards)), _p(keys, C.c_uint32))
    return keys


def key_type(keys):
    """by_type index of internal ranking keys (bits 28.. hold a code with a gap at 5)."""
    code = np.asarray(keys, np.uint32) >> 28
    return (code - (code >= 6)).astype(np.uint32)


def run_ctr(query16, seed, qid, uniform=False, general=False):
    """general=True: every iteration through the general form of mcq_iteration (no straight-line specialisation)."""
    q = np.ascontiguousarray(query16, np.uint8)
    out = np.zeros(13, np.uint64)
    f = lib().hs_run_ctr_uniform if uniform else lib().hs_run_ctr_general if general else lib().hs_run_ctr
    rc = f(_p(q, C.c_uint8), C.c_uint64(seed), C.c_uint64(qid), _p(out, C.c_uint64))
    if rc:
        raise ValueError(rc)
    return out


def run_replay(query16, seed32):
    q = np.ascontiguousarray(query16, np.uint8)
    out = np.zeros(13, np.uint64)
    rc = lib().hs_run_replay(_p(q, C.c_uint8), C.c_uint32(seed32), _p(out, C.c_uint64))
    if rc:
        raise ValueError(rc)
    return out


def select_pop(dlo, dhi, k):
    a, b = C.c_uint32(dlo), C.c_uint32(dhi)
    pos = lib().hs_select_pop(C.byref(a), C.byref(b), C.c_uint32(k))
    return pos, a.value, b.value


def mt_words(seed, n):
    out = np.zeros(n, np.uint32)
    lib().hs_mt_words(C.c_uint32(seed), C.c_uint32(n), _p(out, C.c_uint32))
    return out


def philox(ctr, key):
    c, k, o = np.array(ctr, np.uint32), np.array(key, np.uint32), np.zeros(4, np.uint32)
    lib().hs_philox(_p(c, C.c_uint32), _p(k, C.c_uint32), _p(o, C.c_uint32))
    return o


def run_replay_numpy_stream(query16):
    """Consumes np.random's global MT19937 stream like a reference call and leaves it advanced."""
    q = np.ascontiguousarray(query16, np.uint8)
    out = np.zeros(13, np.uint64)
    st = np.random.get_state()
    key = np.ascontiguousarray(st[1], np.uint32).copy()
    pos = C.c_uint32(int(st[2]))
    rc = lib().hs_run_replay_stream(_p(q, C.c_uint8), _p(key, C.c_uint32), C.byref(pos), _p(out, C.c_uint64))
    if rc:
        raise ValueError(rc)
    np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return out


def run_ext(replay, query16, ext64, seed, qid=0):
    q = np.ascontiguousarray(query16, np.uint8)
    e = np.ascontiguousarray(ext64).view(np.uint8)
    out = np.zeros(13, np.uint64)
    lib().hs_run_ext.restype = C.c_int
    rc = lib().hs_run_ext(C.c_int(1 if replay else 0), _p(q, C.c_uint8), _p(e, C.c_uint8), C.c_uint64(seed),
                          C.c_uint64(qid), _p(out, C.c_uint64))
    if rc:
        raise ValueError(rc)
    return out


def exact(query16, uniform=False):
    """Exact enumeration by the product's lane code (mcq_exact.hpp) -> 13 weights like a result row."""
    q = np.ascontiguousarray(query16, np.uint8)
    out = np.zeros(13, np.uint64)
    rc = lib().hs_exact(_p(q, C.c_uint8), C.c_int(1 if uniform else 0), _p(out, C.c_uint64))
    if rc:
        raise ValueError(rc)
    return out


def mt_parse(query16, seed32, reference=False):
    """Accepted draws [D, stride] (r | 0x80 bytes) and passes of one query's MT19937 stream: the device's
    wave-cooperative parse (mcq_mt.hpp, lanes as arrays) or, reference=True, the sequential host walk."""
    q = np.ascontiguousarray(query16, np.uint8)
    runs = int(q[12:16].view(np.uint32)[0])
    D = 2 * (int(q[8]) - 1) + 5 - int(q[7])
    stride = (runs + 63) & ~63
    draws = np.zeros((max(D, 1), max(stride, 64)), np.uint8)
    f = lib().hs_mt_parse_reference if reference else lib().hs_mt_parse
    f.restype = C.c_uint64
    passes = f(_p(q, C.c_uint8), C.c_uint32(seed32), _p(draws, C.c_uint8), C.c_uint64(max(stride, 64)))
    return draws[:D, :runs], int(passes)


def mt_parse_blocks(query16, seed32, n_blocks=0):
    """The same stream parsed block by block (mcq_mt_blocks.hpp) -> (draws, passes); passes = None when the stream does
    not end within n_blocks state blocks (0: the host's estimate)."""
    q = np.ascontiguousarray(query16, np.uint8)
    runs = int(q[12:16].view(np.uint32)[0])
    D = 2 * (int(q[8]) - 1) + 5 - int(q[7])
    stride = max((runs + 63) & ~63, 64)
    draws = np.zeros((max(D, 1), stride), np.uint8)
    f = lib().hs_mt_parse_blocks
    f.restype = C.c_uint64
    passes = int(f(_p(q, C.c_uint8), C.c_uint32(seed32), _p(draws, C.c_uint8), C.c_uint64(stride), C.c_uint32(n_blocks)))
    return draws[:D, :runs], (None if passes == 2 ** 64 - 1 else passes)


def mtb_blocks_needed(query16):
    q = np.ascontiguousarray(query16, np.uint8)
    lib().hs_mtb_blocks_needed.restype = C.c_uint32
    return int(lib().hs_mtb_blocks_needed(_p(q, C.c_uint8)))


def mt_parse_ext(query16, ext, seed32, reference=False):
    """The same for an extended query (mcq_mt_ext.hpp against mcq_replay_parse_ext) -> (draws [rows, runs], passes);
    passes = None when a range cannot be dealt."""
    q = np.ascontiguousarray(query16, np.uint8)
    e = np.ascontiguousarray(ext).view(np.uint8)
    runs = int(q[12:16].view(np.uint32)[0])
    lib().hs_ext_draws_per_iteration.restype = C.c_uint32
    D = int(lib().hs_ext_draws_per_iteration(_p(q, C.c_uint8), _p(e, C.c_uint8)))
    stride = max((runs + 63) & ~63, 64)
    draws = np.zeros((max(D, 1), stride), np.uint8)
    f = lib().hs_mt_parse_ext_reference if reference else lib().hs_mt_parse_ext
    f.restype = C.c_uint64
    passes = int(f(_p(q, C.c_uint8), _p(e, C.c_uint8), C.c_uint32(seed32), _p(draws, C.c_uint8), C.c_uint64(stride)))
    return draws[:D, :runs], (None if passes == 2 ** 64 - 1 else passes)


def mt_wave_words(seed32, n):
    out = np.zeros(n, np.uint32)
    lib().hs_mt_regenerate_words(C.c_uint32(seed32), C.c_uint32(n), _p(out, C.c_uint32))
    return out


def mt_magic_ok():
    lib().hs_mt_magic_ok.restype = C.c_uint32
    return bool(lib().hs_mt_magic_ok())


def direct_layout(cost, n_cu=256, max_lg=4):
    """The host's wave layout for the one-launch path (mcq_layout.hpp) -> (grid, rounds, lg[n], slot_qi, slot_sub)."""
    cost = np.ascontiguousarray(cost, np.uint64)
    n = len(cost)
    cap = n * 16 + 80 * n_cu * 16 + 64
    lg = np.zeros(max(n, 1), np.uint8)
    qi = np.zeros(cap, np.uint32)
    sub = np.zeros(cap, np.uint8)
    grid = C.c_uint32(0)
    lib().hs_direct_layout.restype = C.c_uint32
    rounds = lib().hs_direct_layout(_p(cost, C.c_uint64), C.c_size_t(n), C.c_uint32(n_cu), C.c_uint32(max_lg), C.byref(grid),
                                    _p(lg, C.c_uint8), _p(qi, C.c_uint32), _p(sub, C.c_uint8), C.c_size_t(cap))
    assert rounds != 0xFFFFFFFF
    slots = rounds * grid.value * 16
    return grid.value, rounds, lg[:n], qi[:slots], sub[:slots]


def direct_records(cost, queries16, n_cu=256, max_lg=4, cap=None):
    """The per-wave work records the library writes for mcq_eval_direct_kernel (mcq_layout.hpp): (rec [slots, 16] u8,
    qi [slots] u32), or None when `cap` slots do not hold the layout (exact-size buffers: a write past them is an
    AddressSanitizer finding under tests/sanitize_cpu.sh)."""
    cost = np.ascontiguousarray(cost, np.uint64)
    q = np.ascontiguousarray(queries16, np.uint8).reshape(-1, 16)
    n = len(cost)
    if cap is None:
        grid, rounds, _, _, _ = direct_layout(cost, n_cu, max_lg)
        cap = rounds * grid * 16
    rec = np.zeros((max(cap, 1), 16), np.uint8)
    qi = np.zeros(max(cap, 1), np.uint32)
    lib().hs_direct_records.restype = C.c_size_t
    slots = lib().hs_direct_records(_p(cost, C.c_uint64), _p(q, C.c_uint8), C.c_size_t(n), C.c_uint32(n_cu), C.c_uint32(max_lg),
                                    _p(rec, C.c_uint8), _p(qi, C.c_uint32), C.c_size_t(cap))
    return (rec[:slots], qi[:slots]) if slots else None
