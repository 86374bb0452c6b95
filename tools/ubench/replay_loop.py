#!/usr/bin/env python3
"""Replays the straight-line iteration of the bulk kernel as a stand-alone gfx950 kernel -- same instructions, same
registers, same order, four waves per SIMD, one 1024-thread block per CU -- so that ISSUE behaviour can be studied (and
re-orderings tried) without the memory side and without rebuilding the library:

    python tools/ubench/replay_loop.py --out /tmp/replay [--nopp 5 --ndeal 5] [--variant as_compiled --variant sep ...]
    -> /tmp/replay/<variant>.s / .co  (+ replay_host: `replay_host <variant>.co [iters]` prints cycles per wave-iteration)

What is changed with respect to the loop in libmcq_hip.so: branches and EXEC manipulation are dropped (all lanes run
`iters` trips), the rarely executed quads lookups (behind s_cbranch_execz) are dropped with them, the twelve global
loads per trip read a small valid buffer at a fixed offset (they hit L1 as the real ones mostly do), register contents
are arbitrary (LDS reads go to arbitrary, mostly out-of-range, addresses: they return zeros, conflicts are random).
Results mean nothing; the instruction stream is what the SIMD sees in production.

Variants (functions below): as_compiled, sep (s_nop 0 behind every slow-class VALU instruction followed by a VALU
instruction: tools/isa_cadence.py's rule), sep_all (behind every slow-class instruction), no_lds (LDS reads dropped),
no_vmem, only_valu (neither), ... -- see VARIANTS.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import isa_cadence  # noqa: E402
import isa_hist  # noqa: E402
import isa_resched  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"


def loop_body(lib, kernel, nopp, ndeal):
    with tempfile.TemporaryDirectory() as wd:
        co = isa_hist.code_object(lib, wd)
        _, ins = isa_hist.kernel_instructions(isa_hist.disassemble(co), kernel)
    j, i = isa_hist.pick_loop(ins, nopp, ndeal)
    body = ins[j:i + 1]
    drop = set()
    for a, b in isa_hist.guarded_ranges(ins, j, i):
        drop.update(range(a - j, b - j))
    out = []
    for k, (_, op, args) in enumerate(body):
        if k in drop:
            continue
        if op.startswith("s_cbranch") or op == "s_branch":
            continue
        if re.match(r"exec\b", args) or op.startswith("s_and_saveexec") or op.startswith("s_or_saveexec"):
            continue
        args = re.sub(r"\s*//.*$", "", args).strip()
        if op.startswith("global_load_dword"):
            m = re.match(r"(v\d+|v\[\d+:\d+\]),\s*v\d+,\s*s\[\d+:\d+\](.*)$", args)
            if not m:
                raise SystemExit("unexpected global load: %s %s" % (op, args))
            args = "%s, v127, s[98:99]" % m.group(1)
        out.append((op, args))
    return out


def used_regs(body):
    v, s = set(), set()
    for _, args in body:
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", args):
            v.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bv(\d+)\b", args):
            v.add(int(m.group(1)))
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]", args):
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bs(\d+)\b", args):
            s.add(int(m.group(1)))
    return v, s


def is_slow(op, args):
    return isa_cadence.is_slow(op, args)


def v_as_compiled(body):
    return list(body)


def v_sep(body):
    out = []
    for k, (op, args) in enumerate(body):
        out.append((op, args))
        if is_slow(op, args) and k + 1 < len(body) and body[k + 1][0].startswith("v_"):
            out.append(("s_nop", "0"))
    return out


def sep_with(sep_op, sep_args, only_before_fast=False):
    def f(body):
        out = []
        for k, (op, args) in enumerate(body):
            out.append((op, args))
            if is_slow(op, args) and k + 1 < len(body) and body[k + 1][0].startswith("v_"):
                if only_before_fast and is_slow(*body[k + 1]):
                    continue
                out.append((sep_op, sep_args))
        return out
    return f


def compose(*fs):
    def f(body):
        for g in fs:
            body = g(body)
        return body
    return f


def v_sep_all(body):
    out = []
    for op, args in body:
        out.append((op, args))
        if is_slow(op, args):
            out.append(("s_nop", "0"))
    return out


def v_no_lds(body):
    return [(o, a) for o, a in body if not o.startswith("ds_")]


def v_no_vmem(body):
    return [(o, a) for o, a in body if not o.startswith("global_")]


def v_only_valu(body):
    return [(o, a) for o, a in body if o.startswith("v_")]


def v_only_valu_sep(body):
    return v_sep(v_only_valu(body))


def v_valu_salu(body):
    return [(o, a) for o, a in body if o.startswith("v_") or (o.startswith("s_") and not o.startswith("s_waitcnt"))]


def v_no_wait(body):
    return [(o, a) for o, a in body if not o.startswith("s_waitcnt")]


def v_fast_only(body):
    """only the fast-class VALU instructions (what they cost alone)"""
    return [(o, a) for o, a in body if o.startswith("v_") and not is_slow(o, a)]


def v_slow_only(body):
    return [(o, a) for o, a in body if o.startswith("v_") and is_slow(o, a)]


VARIANTS = {"as_compiled": v_as_compiled, "sep": v_sep, "sep_all": v_sep_all, "no_lds": v_no_lds, "no_vmem": v_no_vmem,
            "only_valu": v_only_valu, "only_valu_sep": v_only_valu_sep, "valu_salu": v_valu_salu, "no_wait": v_no_wait,
            "fast_only": v_fast_only, "slow_only": v_slow_only}
NOWAIT = ("s_waitcnt", "vmcnt(63) expcnt(7) lgkmcnt(15)")
for _n, _sep in (("w", NOWAIT), ("p", ("s_setprio", "0")), ("m", ("s_mov_b32", "s97, s97")), ("n", ("s_nop", "0"))):
    VARIANTS["valu_sep_" + _n] = compose(v_only_valu, sep_with(*_sep))
    VARIANTS["valu_sepf_" + _n] = compose(v_only_valu, sep_with(*_sep, only_before_fast=True))
    VARIANTS["nolds_sep_" + _n] = compose(v_no_lds, sep_with(*_sep))
    VARIANTS["nolds_sepf_" + _n] = compose(v_no_lds, sep_with(*_sep, only_before_fast=True))



def resched(**kw):
    def f(body):
        new, st = isa_resched.transform(body, **kw)
        print("    %s" % st, file=sys.stderr)
        return new
    return f


# round 4: three-operand logic as v_bitop3_b32 (fast class); the VALU runs between two non-VALU instructions re-ordered,
# fast class first (tools/isa_resched.py); "_s": a separator where a fast one still follows a slow one
for _b, _base in (("nolds", v_no_lds), ("full", v_as_compiled)):
    VARIANTS[_b + "_bitop3"] = compose(_base, resched(bitop3=True, reorder=False))
    VARIANTS[_b + "_resched"] = compose(_base, resched(bitop3=False, reorder=True))
    VARIANTS[_b + "_resched_s"] = compose(_base, resched(bitop3=False, reorder=True, sep=True))
    VARIANTS[_b + "_bitop3_resched"] = compose(_base, resched(bitop3=True, reorder=True))
    VARIANTS[_b + "_bitop3_resched_s"] = compose(_base, resched(bitop3=True, reorder=True, sep=True))
    VARIANTS[_b + "_bitop3_resched_sall"] = compose(_base, resched(bitop3=True, reorder=True, sep="all"))
    VARIANTS[_b + "_bitop3_sink_resched_s"] = compose(_base, resched(bitop3=True, reorder=True, sep=True, sink=True))
    VARIANTS[_b + "_bitop3_sink_resched"] = compose(_base, resched(bitop3=True, reorder=True, sep=False, sink=True))
    VARIANTS[_b + "_bitop3_split_resched_s"] = compose(_base, resched(bitop3=True, reorder=True, sep=True, split=True))
    VARIANTS[_b + "_bitop3_split"] = compose(_base, resched(bitop3=True, reorder=False, split=True))
    VARIANTS[_b + "_bitop3_sepf"] = compose(_base, resched(bitop3=True, reorder=False), sep_with("s_nop", "0", only_before_fast=True))



def v_no_sgpr_src(body):
    """every single-SGPR source operand of a VALU instruction becomes a VGPR (v127): what do the constant-bus reads cost?"""
    out = []
    for o, a in body:
        if o.startswith("v_") and not o.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            parts = [p.strip() for p in a.split(",")]
            n_dst = 2 if o.startswith("v_mad_u64") else 1
            for k in range(n_dst, len(parts)):
                if re.fullmatch(r"s\d+", parts[k]):
                    parts[k] = "v127"
            a = ", ".join(parts)
        out.append((o, a))
    return out


def renumber(seed):
    """the same stream with its single VGPRs renamed by a random permutation (registers that appear in a tuple v[a:b] keep
    their numbers): does the ASSIGNMENT of registers (banks, operand ports) matter for the issue rate?"""
    def f(body):
        import random
        fixed, used = {0, 127}, set()
        for _, a in body:
            for m in re.finditer(r"\bv\[(\d+):(\d+)\]", a):
                fixed.update(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r"\bv(\d+)\b", a):
                used.add(int(m.group(1)))
        free = sorted(used - fixed)
        perm = free[:]
        random.Random(seed).shuffle(perm)
        mp = dict(zip(free, perm))
        out = []
        for o, a in body:
            out.append((o, re.sub(r"\bv(\d+)\b", lambda m: "v%d" % mp.get(int(m.group(1)), int(m.group(1))), a)))
        return out
    return f


def canon(*keep):
    """every VALU instruction replaced by a plain representative of its issue class (fast: v_add_u32 d, d, v127; slow:
    v_bcnt_u32_b32 d, d, v127; d = its own destination register) -- except the opcodes starting with one of `keep`: does the
    stream then cost the sum of its classes, and which opcodes make it cost more?"""
    def f(body):
        out = []
        for o, a in body:
            if o.startswith("v_") and not o.startswith(keep if keep else ("\0",)):
                m = re.match(r"\s*(v\d+)\b", a)
                d = m.group(1) if m else "v126"
                out.append(("v_bcnt_u32_b32", "%s, %s, v127" % (d, d)) if is_slow(o, a) else ("v_add_u32_e32", "%s, %s, v127" % (d, d)))
            else:
                out.append((o, a))
        return out
    return f


def v_sepall_strict(body):
    """an s_nop 0 behind EVERY slow-class instruction that is not already followed by a non-VALU instruction"""
    out = []
    for k, (o, a) in enumerate(body):
        out.append((o, a))
        if o.startswith("v_") and is_slow(o, a) and k + 1 < len(body) and body[k + 1][0].startswith("v_"):
            out.append(("s_nop", "0"))
    return out


def v_sorted_classes(body):
    """(synthetic) all fast-class instructions first, then the slow ones, each followed by s_nop 0"""
    fast = [(o, a) for o, a in body if o.startswith("v_") and not is_slow(o, a)]
    slow = [(o, a) for o, a in body if o.startswith("v_") and is_slow(o, a)]
    out = list(fast)
    for x in slow:
        out.append(x)
        out.append(("s_nop", "0"))
    return out


def blocks_of(nf, ns):
    """(synthetic) the same instructions dealt into blocks of nf fast then ns slow ones, each slow one followed by s_nop 0"""
    def f(body):
        fast = [(o, a) for o, a in body if o.startswith("v_") and not is_slow(o, a)]
        slow = [(o, a) for o, a in body if o.startswith("v_") and is_slow(o, a)]
        out = []
        while fast or slow:
            out.extend(fast[:nf]); del fast[:nf]
            for x in slow[:ns]:
                out.append(x); out.append(("s_nop", "0"))
            del slow[:ns]
        return out
    return f


def v_rotate_regs(body):
    """(synthetic, behind canon()) the canonical instructions on eight registers in turn: no dependence left but one in eight"""
    out, k = [], 0
    for o, a in body:
        if o in ("v_add_u32_e32", "v_bcnt_u32_b32"):
            out.append((o, "v%d, v%d, v127" % (110 + k % 8, 110 + k % 8)))
            k += 1
        elif o == "s_nop":
            out.append((o, "0"))
        else:
            out.append((o, a))
    return out


def chunk_of(k, n):
    """(synthetic) the k-th of n equal pieces of the stream, repeated to the stream's length: which PART of the order costs
    more than the sum of its classes?"""
    def f(body):
        m = (len(body) + n - 1) // n
        piece = body[k * m:(k + 1) * m]
        out = []
        while len(out) < len(body):
            out.extend(piece)
        return out
    return f


def window_alt(w):
    """(synthetic) inside every window of w VALU instructions: fast and slow ones dealt alternately (f S n f S n ..., the
    leftover class at the end), each slow one followed by s_nop 0 -- how LOCAL may a re-ordering be and still reach the sum
    of the classes?"""
    def f(body):
        valu = [(o, a) for o, a in body if o.startswith("v_")]
        out = []
        for i in range(0, len(valu), w):
            win = valu[i:i + w]
            fast = [x for x in win if not is_slow(*x)]
            slow = [x for x in win if is_slow(*x)]
            while fast or slow:
                if fast:
                    out.append(fast.pop(0))
                if slow:
                    out.append(slow.pop(0))
                    out.append(("s_nop", "0"))
        return out
    return f


def shuffled(seed, period=0):
    """(synthetic) the VALU instructions in a random order (period > 0: a random order of the first `period` of them,
    repeated), each slow one followed by s_nop 0"""
    def f(body):
        import random
        valu = [(o, a) for o, a in body if o.startswith("v_")]
        if period:
            piece = valu[:period]
            random.Random(seed).shuffle(piece)
            valu = (piece * (len(valu) // period + 1))[:len(valu)]
        else:
            random.Random(seed).shuffle(valu)
        out = []
        for x in valu:
            out.append(x)
            if is_slow(*x):
                out.append(("s_nop", "0"))
        return out
    return f


def chunk_sorted(n):
    """(synthetic) the stream cut into n pieces, inside each piece the fast ones first, then the slow ones with separators"""
    def f(body):
        valu = [(o, a) for o, a in body if o.startswith("v_")]
        m = (len(valu) + n - 1) // n
        out = []
        for i in range(0, len(valu), m):
            out.extend(v_sorted_classes(valu[i:i + m]))
        return out
    return f


def barrier_every(k):
    """an s_barrier behind every k-th VALU instruction (at a point where a non-VALU instruction already stands, if one is
    near): do the waves of a SIMD pair their fast-class instructions better when they are kept in step?"""
    def f(body):
        out, n = [], 0
        for o, a in body:
            out.append((o, a))
            if o.startswith("v_"):
                n += 1
                if n % k == 0:
                    out.append(("s_barrier", ""))
        return out
    return f


def drop_ops(*prefixes):
    """the stream without the instructions whose opcode starts with one of `prefixes` (what that opcode class costs in situ)"""
    def f(body):
        return [(o, a) for o, a in body if not o.startswith(prefixes)]
    return f


for _n, _p in (("mad64", ("v_mad_u64_u32",)), ("cmp", ("v_cmp_",)), ("max", ("v_max",)), ("perm", ("v_perm_b32",)),
               ("sad", ("v_sad_u8",)), ("add3", ("v_add3_u32",)), ("bcnt", ("v_bcnt_",)), ("lshladd", ("v_lshl_add_u32",)),
               ("sel64", ("v_cndmask_b32_e64",)), ("bitop3", ("v_bitop3_b32",)), ("andor", ("v_and_b32", "v_or_b32")),
               ("sub", ("v_sub_u32",)), ("snop", ("s_nop",)), ("salu", ("s_and_b64", "s_or_b64", "s_mov_b64", "s_andn2_b64"))):
    VARIANTS["nolds_drop_" + _n] = compose(v_no_lds, drop_ops(*_p))

for _k in range(6):
    VARIANTS["nolds_renum%d" % _k] = compose(v_no_lds, renumber(_k))
VARIANTS["nolds_canon"] = compose(v_no_lds, canon())
for _n, _p in (("mad64", ("v_mad_u64",)), ("cmp", ("v_cmp",)), ("sel", ("v_cndmask",)), ("vop3", ("v_bitop3", "v_add3", "v_lshl_add", "v_lshl_or", "v_max3", "v_xad", "v_bfe", "v_perm", "v_sad", "v_bcnt")),
               ("vop2", ("v_and_b32", "v_or_b32", "v_sub_u32", "v_add_u32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_mov_b32", "v_max_u32"))):
    VARIANTS["nolds_canon_keep_" + _n] = compose(v_no_lds, canon(*_p))
VARIANTS["canon_valu_only"] = compose(v_only_valu, canon())
VARIANTS["canon_valu_nop"] = compose(lambda b: [(o, a) for o, a in b if o.startswith("v_") or o == "s_nop"], canon())
VARIANTS["canon_sepall"] = compose(lambda b: [(o, a) for o, a in b if o.startswith("v_") or o == "s_nop"], canon(), v_sepall_strict)
VARIANTS["canon_sepall_rot"] = compose(lambda b: [(o, a) for o, a in b if o.startswith("v_") or o == "s_nop"], canon(), v_sepall_strict, v_rotate_regs)
VARIANTS["canon_blocks_1_1_rot"] = compose(v_only_valu, canon(), blocks_of(1, 1), v_rotate_regs)
for _k in range(12):
    VARIANTS["canon_sepall_chunk%02d" % _k] = compose(lambda b: [(o, a) for o, a in b if o.startswith("v_") or o == "s_nop"], canon(), v_sepall_strict, v_rotate_regs, chunk_of(_k, 12))
_vn = lambda b: [(o, a) for o, a in b if o.startswith("v_") or o == "s_nop"]
for _n, _p in (("mad64", ("v_mad_u64",)), ("cmp", ("v_cmp",)), ("sel", ("v_cndmask",)),
               ("vop3slow", ("v_add3", "v_lshl_add", "v_lshl_or", "v_max3", "v_xad", "v_bfe", "v_perm", "v_sad", "v_bcnt")),
               ("bitop3", ("v_bitop3",)), ("max", ("v_max_u32",)), ("lshl", ("v_lshlrev",)),
               ("vop2fast", ("v_and_b32", "v_or_b32", "v_sub_u32", "v_add_u32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32")),
               ("all", ("v_",))):
    VARIANTS["canon_sepall_keep_" + _n] = compose(_vn, canon(*_p), v_sepall_strict)
for _w in (4, 8, 16, 32, 64):
    VARIANTS["canon_winalt%d" % _w] = compose(v_only_valu, canon(), window_alt(_w))
VARIANTS["canon_shuffle0"] = compose(v_only_valu, canon(), shuffled(0))
VARIANTS["canon_shuffle1"] = compose(v_only_valu, canon(), shuffled(1))
for _pd in (8, 16, 32, 64, 128, 256):
    VARIANTS["canon_shuffle_p%d" % _pd] = compose(v_only_valu, canon(), shuffled(3, _pd))
for _n in (4, 12, 32):
    VARIANTS["canon_chunksorted%d" % _n] = compose(v_only_valu, canon(), chunk_sorted(_n))
for _k in (16, 32, 64, 128):
    VARIANTS["canon_sepall_bar%d" % _k] = compose(_vn, canon(), v_sepall_strict, barrier_every(_k))
    VARIANTS["canon_chunksorted12_bar%d" % _k] = compose(v_only_valu, canon(), chunk_sorted(12), barrier_every(_k))
    VARIANTS["nolds_bar%d" % _k] = compose(v_no_lds, barrier_every(_k))
VARIANTS["canon_sorted"] = compose(v_only_valu, canon(), v_sorted_classes)
VARIANTS["canon_blocks_3_2"] = compose(v_only_valu, canon(), blocks_of(3, 2))
VARIANTS["canon_blocks_6_4"] = compose(v_only_valu, canon(), blocks_of(6, 4))
VARIANTS["canon_blocks_1_1"] = compose(v_only_valu, canon(), blocks_of(1, 1))
VARIANTS["nolds_nosgpr"] = compose(v_no_lds, v_no_sgpr_src)
VARIANTS["nolds_nosgpr_nosel"] = compose(v_no_lds, v_no_sgpr_src, drop_ops("v_cndmask_b32_e64"))

TEMPLATE = """\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
\t.text
\t.globl\treplay
\t.p2align\t8
\t.type\treplay,@function
replay:
\ts_load_dwordx2 s[98:99], s[0:1], 0x0
\ts_load_dword s100, s[0:1], 0x8
\tv_and_b32_e32 v127, 63, v0
\tv_lshlrev_b32_e32 v127, 2, v127
{init}
\ts_waitcnt lgkmcnt(0)
{prio}
.Lloop:
{body}
\ts_sub_u32 s100, s100, 1
\ts_cmp_lg_u32 s100, 0
\ts_cbranch_scc1 .Lloop
\ts_waitcnt vmcnt(0) lgkmcnt(0)
\ts_endpgm
.Lend:
\t.size\treplay, .Lend-replay
\t.rodata
\t.p2align\t6
\t.amdhsa_kernel replay
\t\t.amdhsa_group_segment_fixed_size {lds}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size 16
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr 128
\t\t.amdhsa_next_free_sgpr 102
\t\t.amdhsa_accum_offset 128
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
\t.text
\t.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     0
    .args:
      - .address_space:  global
        .offset:         0
        .size:           8
        .value_kind:     global_buffer
      - .offset:         8
        .size:           4
        .value_kind:     by_value
    .group_segment_fixed_size: {lds}
    .kernarg_segment_align: 8
    .kernarg_segment_size: 16
    .max_flat_workgroup_size: 1024
    .name:           replay
    .private_segment_fixed_size: 0
    .sgpr_count:     108
    .sgpr_spill_count: 0
    .symbol:         replay.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     128
    .vgpr_spill_count: 0
    .wavefront_size: 64
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
\t.end_amdgpu_metadata
"""


PRIO_STAGGER = """\tv_readfirstlane_b32 s96, v0
\ts_lshr_b32 s96, s96, 8
\ts_cmp_eq_u32 s96, 0
\ts_cbranch_scc1 .Lp0
\ts_cmp_eq_u32 s96, 1
\ts_cbranch_scc1 .Lp1
\ts_cmp_eq_u32 s96, 2
\ts_cbranch_scc1 .Lp2
\ts_setprio 3
\ts_branch .Lloop
.Lp0:
\ts_setprio 0
\ts_branch .Lloop
.Lp1:
\ts_setprio 1
\ts_branch .Lloop
.Lp2:
\ts_setprio 2"""


SLEEP_STAGGER = """\tv_readfirstlane_b32 s96, v0
\ts_lshr_b32 s96, s96, 8
\ts_cmp_eq_u32 s96, 0
\ts_cbranch_scc1 .Lloop
\ts_cmp_eq_u32 s96, 1
\ts_cbranch_scc1 .Lq1
\ts_cmp_eq_u32 s96, 2
\ts_cbranch_scc1 .Lq2
\ts_sleep {q3}
\ts_branch .Lloop
.Lq1:
\ts_sleep {q1}
\ts_branch .Lloop
.Lq2:
\ts_sleep {q2}"""


def emit(body, vregs, sregs, lds, prio=""):
    init = []
    for r in sorted(vregs):
        if r in (0, 127):
            continue
        init.append("\tv_mul_u32_u24_e32 v%d, %d, v0" % (r, 2654435 + 977 * r))
    for r in sorted(sregs):
        if r in (0, 1) or r >= 98:
            continue
        init.append("\ts_mov_b32 s%d, 0x%x" % (r, (0x9E3779B9 * (r + 3)) & 0xFFFF))
    text = "\n".join("\t%s %s" % (o, a) for o, a in body)
    return TEMPLATE.format(init="\n".join(init), body=text, lds=lds, prio=prio)


HOST = r'''// replay_host.cpp -- loads a code object made by tools/ubench/replay_loop.py and times its kernel `replay`:
//   replay_host <file.co> [iters]  ->  SIMD-cycles per wave-iteration at four waves per SIMD (2.4 GHz assumed)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    hipModule_t mod;
    hipFunction_t fn;
    CHK(hipModuleLoad(&mod, argv[1]));
    CHK(hipModuleGetFunction(&fn, mod, "replay"));
    void *buf;
    CHK(hipMalloc(&buf, 1 << 20));
    CHK(hipMemset(buf, 0, 1 << 20));
    struct { void *p; int iters; int pad; } args = {buf, 20, 0};
    size_t sz = sizeof args;
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipModuleLaunchKernel(fn, p.multiProcessorCount, 1, 1, 1024, 1, 1, 0, 0, nullptr, cfg));
    CHK(hipDeviceSynchronize());
    args.iters = iters;
    CHK(hipEventRecord(e0));
    CHK(hipModuleLaunchKernel(fn, p.multiProcessorCount, 1, 1, 1024, 1, 1, 0, 0, nullptr, cfg));
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s %8.3f ms  %8.1f SIMD-cycles per wave-iteration\n", argv[1], ms, ms * 1e-3 * 2.4e9 / (4.0 * iters));
    return 0;
}
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(isa_hist.ROOT, "neuron_poker_amd", "libmcq_hip.so"))
    ap.add_argument("--kernel", default="mcq_eval_kernelILi0ELb0")
    ap.add_argument("--nopp", type=int, default=5)
    ap.add_argument("--ndeal", type=int, default=5)
    ap.add_argument("--out", required=True)
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--lds", type=int, default=115712)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    body = loop_body(a.lib, a.kernel, a.nopp, a.ndeal)
    vregs, sregs = used_regs([(o, a) for o, a in body if not o.startswith("global_load")])
    if max(vregs) >= 127 or max(sregs) >= 98:
        raise SystemExit("the loop uses v%d / s%d: the harness's own registers collide" % (max(vregs), max(sregs)))
    for name in (a.variant or list(VARIANTS)):
        b = VARIANTS[re.sub(r"(_prio|_stag\d+)$", "", name)](body)
        n_valu = sum(1 for o, _ in b if o.startswith("v_"))
        s_path = os.path.join(a.out, name + ".s")
        with open(s_path, "w") as f:
            pre = PRIO_STAGGER if name.endswith("_prio") else ""
            m_st = re.search(r"_stag(\d+)$", name)
            if m_st:
                q = int(m_st.group(1))
                pre = SLEEP_STAGGER.format(q1=q, q2=2 * q, q3=3 * q)
            f.write(emit(b, vregs, sregs, a.lds, pre))
        obj = os.path.join(a.out, name + ".o")
        subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_path, "-o", obj])
        subprocess.check_call([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o",
                               os.path.join(a.out, name + ".co"), obj])
        os.remove(obj)
        print("%-16s %4d instructions, %4d VALU" % (name, len(b), n_valu))
    host = os.path.join(a.out, "replay_host.cpp")
    with open(host, "w") as f:
        f.write(HOST)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-o", os.path.join(a.out, "replay_host"), host])


if __name__ == "__main__":
    main()
