#!/usr/bin/env python3
"""Post-pass over the gfx950 assembly of the kernels: keep a wave's fast-class VALU instructions at their 2-cycle issue
cadence by ending the 4-cycle cadence that every slow-class instruction starts.

Measured on MI355X (tools/ubench/gen_issue_probe*.py, profiles/r07_issue_probe*.txt): a wave64 VALU instruction of the
simple class (v_add/sub/and/or/xor/not/mov/lshr/ashr, v_bitop3, v_cndmask_b32_e64, v_addc_co) issues in 2.1-2.4
SIMD-cycles, everything else (shifts left, compares, min/max, bit counts, v_perm, every other three-operand form,
multiplies, SDWA/DPP) in 4.1-4.3 -- and once a wave has issued one instruction of the second kind, ALL its following
VALU instructions go at the 4-cycle cadence until a NON-VALU instruction (s_nop, s_waitcnt, any SALU instruction, a
branch) comes by: `add, bcnt, add, bcnt ...` costs 4.3 cycles per instruction, `add, bcnt, s_nop 0, add, bcnt, s_nop 0`
3.3 = (2.3 + 4.3) / 2.  One separator behind a RUN of slow instructions is not enough (fSSn 3.9 where fSnSn reaches 3.4):
it must follow each of them.  The compiler knows nothing of this; its schedule leaves half of the bulk kernel's
instructions (the fast class) at the slow cadence.

So: behind every slow-class VALU instruction whose successor is another VALU instruction this pass puts an `s_nop 0`
(one SALU issue slot of the wave; free for the SIMD while other waves have vector work).  Only inside the functions
named by --only (substring match on the symbol; default: every kernel), only between instructions of one basic block
(labels, directives and comments end a pair), never inside an inline-asm region that declares itself with ';APP'.
Wait states only ever get longer by an s_nop, so every hazard the compiler has padded for stays padded.

    python tools/isa_cadence.py in.s out.s [--only mcq_eval_kernel --only mcq_eval_ext_kernel] [--stats]
"""
import argparse
import re
import sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32", "v_addc_co_u32", "v_nop", "v_add_f32", "v_mul_f32", "v_sub_f32"}
INS = re.compile(r"^\t([a-z_][a-z0-9_]*)\b(.*)$")


def base_of(op):
    for suf in ("_e64", "_e32"):
        if op.endswith(suf):
            return op[: -len(suf)]
    return op


def is_valu(op):
    return op.startswith("v_")


def is_slow(op, args):
    if not is_valu(op):
        return False
    if op.endswith("_sdwa") or op.endswith("_dpp"):
        return True
    b = base_of(op)
    if b == "v_cndmask_b32":
        return not op.endswith("_e64")  # the VOP3 form with its mask in an SGPR pair (or VCC) is of the fast class
    return b not in FAST


def process(lines, only, sep):
    out, stats = [], {}
    func, active, in_app = None, False, False
    prev_slow = False  # the last emitted line is a slow-class VALU instruction of this basic block
    for line in lines:
        s = line.rstrip("\n")
        m = INS.match(s)
        if s.startswith("\t.type") and "@function" in s:
            func = s.split()[1].split(",")[0]
            active = (not only) or any(o in func for o in only)
        if ";APP" in s:
            in_app = True
        if ";NO_APP" in s:
            in_app = False
        if m and not s.lstrip().startswith(("." , ";")):
            op, args = m.group(1), m.group(2)
            if prev_slow and is_valu(op) and active and not in_app:
                out.append("\t%s\n" % sep)
                stats[func] = stats.get(func, 0) + 1
            prev_slow = active and not in_app and is_slow(op, args)
        else:
            stripped = s.strip()
            if stripped and not stripped.startswith(";"):
                prev_slow = False  # a label or a directive: another block may enter here
        out.append(line)
    return out, stats


LLVM = "/opt/rocm/lib/llvm/bin"
BRANCH = re.compile(r"^\t(s_branch|s_cbranch_[a-z0-9]+)\s+(\S+)")
INVERT = {"s_cbranch_scc0": "s_cbranch_scc1", "s_cbranch_scc1": "s_cbranch_scc0", "s_cbranch_vccz": "s_cbranch_vccnz",
          "s_cbranch_vccnz": "s_cbranch_vccz", "s_cbranch_execz": "s_cbranch_execnz", "s_cbranch_execnz": "s_cbranch_execz"}


def assemble(lines, dst_s, obj, max_rounds=12):
    """Assemble; a branch the separators have pushed out of the 16-bit range (the bulk kernel is 128 KB of straight-line
    specialisations to begin with) is sent over an ISLAND: an unconditional `s_branch target` placed about half way,
    directly behind an existing unconditional branch (where nothing falls through), and the original branch aims at
    the island.  Repeats until the assembler is content."""
    import subprocess
    n_islands = 0
    for _ in range(max_rounds):
        with open(dst_s, "w") as f:
            f.writelines(lines)
        r = subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", dst_s,
                            "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode == 0:
            return n_islands
        bad = sorted({int(m.group(1)) for m in re.finditer(r":(\d+):\d+: error: branch size exceeds simm16", r.stderr)})
        if not bad:
            raise SystemExit("assembler failed:\n" + r.stderr[-4000:])
        label_line = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(\.L[A-Za-z0-9_$.]+):", l)
            if m:
                label_line[m.group(1)] = i
        inserts = []  # (line index to insert BEFORE, text)
        for ln in bad:
            i = ln - 1
            m = BRANCH.match(lines[i])
            if not m or m.group(2) not in label_line:
                raise SystemExit("cannot relax line %d: %s" % (ln, lines[i]))
            op, target = m.group(1), m.group(2)
            j = label_line[target]
            mid = (i + j) // 2
            # nearest line behind an unconditional branch / end of program, searching outwards from the middle
            spot = None
            for d in range(0, abs(j - i) // 2 - 8):
                for k in (mid + d, mid - d):
                    if min(i, j) + 4 < k < max(i, j) - 4 and re.match(r"^\t(s_branch|s_endpgm|s_setpc_b64)\b", lines[k]):
                        spot = k + 1
                        break
                if spot:
                    break
            if spot is None:
                raise SystemExit("no place for a branch island between lines %d and %d" % (i + 1, j + 1))
            n_islands += 1
            isl = ".Lmcq_island_%d" % n_islands
            lines[i] = lines[i].replace(target, isl, 1)
            inserts.append((spot, "%s:\n\ts_branch %s\n" % (isl, target)))
        for spot, text in sorted(inserts, reverse=True):
            lines.insert(spot, text)
    raise SystemExit("branches still out of range after %d rounds" % max_rounds)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--only", action="append", default=[])
    ap.add_argument("--sep", default="s_nop 0")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--obj", help="assemble the result into this object file (branch islands where the range demands)")
    a = ap.parse_args()
    with open(a.src) as f:
        lines = f.readlines()
    out, stats = process(lines, a.only, a.sep)
    if a.obj:
        n = assemble(out, a.dst, a.obj)
        if a.stats:
            print("%6d branch islands" % n, file=sys.stderr)
    with open(a.dst, "w") as f:
        f.writelines(out)
    if a.stats:
        for k, v in sorted(stats.items(), key=lambda kv: -kv[1]):
            print("%6d separators  %s" % (v, k), file=sys.stderr)
        print("%6d separators in all" % sum(stats.values()), file=sys.stderr)


if __name__ == "__main__":
    main()
