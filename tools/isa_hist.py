#!/usr/bin/env python3
"""Opcode histogram of a hot loop of libmcq_hip.so's gfx950 code object, priced with the issue costs measured on the
hardware (tools/ubench -> profiles/r01_ubench*.txt, profiles/r06_vcc_probe.txt, profiles/r07_issue_probe.txt).

    python tools/isa_hist.py [--lib neuron_poker_amd/libmcq_hip.so] [--kernel mcq_eval_kernelILi0ELb0] \
                             [--nopp 5 --ndeal 5] [--json profiles/<tag>_isa_hist.json] [--dump loop.s]

What it does: unbundles the gfx950 code object (clang-offload-bundler), disassembles it (llvm-objdump -d), takes the
kernel whose mangled name contains --kernel, finds its loops (backward branches) and picks the straight-line iteration
of `nopp` opponents and `ndeal` table cards to come: the innermost loop WITHOUT any other branch inside whose signature
fits -- 2 * nopp + ndeal card reads (ds_read_b128) and 3 * nopp + ndeal + ceil(ndeal / 2) 64-bit multiply-adds
(one per random word, one per bounded draw; mcq_device.hpp: McqCtrDrawsT, McqMwc64x).  Every instruction of that loop
is put into an issue class:

    fast     VALU of the 2-cycle class (profiles/r07_issue_probe.txt, priced among plain adds): v_add/sub/subrev_u32,
             v_and/or/xor/not_b32, v_mov_b32, v_lshrrev/ashrrev (also with a literal or an SGPR operand), v_bitop3_b32,
             v_cndmask_b32_e64 (mask in an SGPR pair), v_addc_co_u32
    slow     every other VALU instruction (4-cycle class): shifts left, compares, the VOP2 / SDWA selects, bit counts,
             min/max, every other three-operand form (v_and_or, v_or3, v_add3, v_lshl_add, v_lshl_or, v_bfi, v_bfe, v_perm,
             v_sad_u8, v_xad ...), multiplies, SDWA / DPP forms
    mul64    v_mad_u64_u32 (the full-width multiplier; priced with the slow class)
    sel_vcc  VOP2 / SDWA v_cndmask_b32 taking its mask from VCC (slow class; 16-23 cycles when VCC was not written by a
             vector compare just before -- profiles/r06_vcc_probe.txt -- counted as `stale_vcc_selects`)
    salu, lds, vmem, wait, branch, other: not VALU (counted, priced 0 on the VALU pipe)

THE CADENCE RULE (profiles/r07_issue_probe*.txt): the class of an instruction is not its price.  Once a wave has issued
a slow-class instruction, every VALU instruction it issues afterwards goes at the 4-cycle cadence -- fast-class ones too --
until a NON-VALU instruction of that wave (SALU, s_nop, s_waitcnt, LDS, VMEM, branch) comes by.  So the tool prices the loop
three ways, in SIMD-cycles per wave-iteration at four waves per SIMD:

    cycles_if_every_class_kept_its_cadence   sum(count x class cost): what a schedule that never lets a fast-class
                                             instruction follow a slow-class one inside a VALU run would need -- the
                                             PRACTICAL BOUND of this instruction mix (`cycles_per_wave_iteration_bound`)
    cycles_as_scheduled                      the same walk with the cadence rule applied to the instruction ORDER the
                                             compiler chose: a fast-class instruction behind a slow-class one of its run
                                             is priced slow (`poisoned_fast_instructions` says how many)
    cycles_at_2_per_instruction              2 x VALU count: the guide's peak, which no mix with a slow-class
                                             instruction in it can reach

Class costs: measured on THIS loop's own instructions replayed as a stand-alone kernel (tools/ubench/replay_loop.py,
profiles/r07_replay.txt): its 261 fast-class instructions alone 2.61 cycles each, its 245 slow-class ones alone 4.67, the
whole VALU stream in the compiler's order 4.28 per instruction (= the production kernel's 4.2-4.3).  With --roles the VALU
instructions are split by role by DIFFERENCE against diagnostic builds with one stage stubbed out (-DMCQ_ABLATE_RNG /
_HOLES / _EVAL, mcq_device.hpp).

bench.py reads the JSON (profiles/current_isa_hist.json) and reports roofline.practical from it.
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

# measured issue costs, SIMD-cycles per wave64 instruction at 4 waves per SIMD (profiles/r01_ubench.txt,
# r06_vcc_probe.txt; refined per context by profiles/r07_issue_probe.txt when that file's table is passed in)
COST = {"fast": 2.61, "slow": 4.67, "mul64": 4.67, "sel_vcc": 4.67, "fast_at_slow_cadence": 4.2, "separator": 0.8, "slow_behind_slow": 1.0}

FAST_OPS = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32",
            "v_lshrrev_b32", "v_ashrrev_i32", "v_add_f32", "v_mul_f32", "v_sub_f32", "v_bitop3_b32", "v_addc_co_u32",
            "v_nop"}


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw).stdout


def code_object(lib, workdir):
    """The gfx950 code object inside a host library (llvm-objdump --offloading writes the bundle's members beside its
    input, so the input is a copy in the work directory) -- or `lib` itself when it already is a device object."""
    with open(lib, "rb") as f:
        head = f.read(20)
    if head[18:20] == b"\xe0\x00":  # e_machine EM_AMDGPU
        return lib
    import shutil
    cp = os.path.join(workdir, "lib.so")
    shutil.copy(lib, cp)
    run([os.path.join(LLVM, "llvm-objdump"), "--offloading", cp], cwd=workdir)
    for f in sorted(os.listdir(workdir)):
        if f.startswith("lib.so.") and "gfx950" in f:
            return os.path.join(workdir, f)
    raise SystemExit("no gfx950 code object in " + lib)


def disassemble(co):
    return run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co])


INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")


def kernel_instructions(asm, needle):
    """[(address, opcode, operands)] of the first function whose symbol contains `needle`."""
    ins, inside, name = [], False, None
    for line in asm.split("\n"):
        if line.endswith(">:"):
            if inside:
                break
            if needle in line:
                inside, name = True, line.split("<", 1)[1][:-2]
            continue
        if inside:
            m = INS.match(line)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if not ins:
        raise SystemExit("no kernel matching %r in the code object" % needle)
    return name, ins


def branch_target(addr, operands):
    v = int(operands.split()[0])
    if v >= 32768:
        v -= 65536
    return addr + 4 + 4 * v


def loops_of(ins):
    """Loops of the kernel: (first index, index of the backward branch) for every backward branch whose body holds no
    other BACKWARD branch and no branch that leaves it except at its foot (forward skips inside the body -- a rare case
    guarded by s_cbranch_execz -- are fine).  A loop closed by two branches to the same head (a conditional one and,
    a few scalar instructions later, an unconditional one) is reported once, with the later foot."""
    at = {a: i for i, (a, _, _) in enumerate(ins)}
    by_head = {}
    for i, (a, op, args) in enumerate(ins):
        if not (op.startswith("s_cbranch") or op == "s_branch"):
            continue
        t = branch_target(a, args)
        if t > a or t not in at:
            continue
        j = at[t]
        ok = True
        for k in range(j, i):
            ak, ok_op, ok_args = ins[k]
            if ok_op in ("s_setpc_b64", "s_endpgm"):
                ok = False
            if ok_op.startswith("s_cbranch") or ok_op == "s_branch":
                tk = branch_target(ak, ok_args)
                if tk == t:
                    continue  # the loop's other closing branch
                if tk <= ak:
                    ok = False  # an inner loop
        if ok:
            by_head[j] = max(i, by_head.get(j, -1))
    return sorted(by_head.items())


def guarded_ranges(ins, j, i):
    """Index ranges inside the loop body that a forward s_cbranch_execz skips (executed only when some lane needs
    them -- the quads lookup of a hand) or that lie behind the loop's exit test."""
    at = {a: k for k, (a, _, _) in enumerate(ins)}
    out = []
    for k in range(j, i):
        ak, op, args = ins[k]
        if op == "s_cbranch_execz":
            t = branch_target(ak, args)
            if t > ak and t in at and at[t] <= i:
                out.append((k + 1, at[t]))
    return out


def classify(op, args):
    base = op
    for suf in ("_e64", "_e32", "_sdwa", "_dpp"):
        if base.endswith(suf):
            base = base[: -len(suf)]
    if op.startswith("v_"):
        if base == "v_mad_u64_u32":
            return "mul64"
        if base == "v_cndmask_b32":
            return "fast" if op.endswith("_e64") else "sel_vcc"
        if op.endswith("_sdwa") or op.endswith("_dpp"):
            return "slow"
        return "fast" if base in FAST_OPS else "slow"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def histogram(body, vcc_window, guarded=()):
    """guarded: index ranges of `body` that are executed only when some lane needs them (counted, priced 0)."""
    classes, ops = {}, {}
    bound = scheduled = 0.0
    last_vcc_writer, since, stale = None, 10 ** 9, 0
    slow_cadence = False  # a slow-class instruction has been issued since the wave's last non-VALU instruction
    poisoned = 0
    prev_class = None
    separators = slow_pairs = 0
    rare = set()
    for a, b in guarded:
        rare.update(range(a, b))
    for k, (_, op, args) in enumerate(body):
        c = classify(op, args)
        key = op
        if c == "sel_vcc" and not (last_vcc_writer == "v_cmp" and since <= vcc_window):
            stale += 1
            key = op + " (VCC not fresh from a v_cmp)"
        classes[c] = classes.get(c, 0) + 1
        ops.setdefault(c, {})
        ops[c][key] = ops[c].get(key, 0) + 1
        if k not in rare:
            # round 4 (profiles/r07_issue_probe3.txt): a separator is not free (`fSn` 6.64 cycles against `fnSn` 8.40: 0.4-1.1
            # each), and a slow-class instruction directly behind another one costs 0.9-1.3 more than alone (`fSSn` 11.6
            # against `fSnSn` 10.2 + a separator)
            if op == "s_nop" and prev_class in ("slow", "mul64", "sel_vcc"):
                separators += 1
                scheduled += COST["separator"]
            if c in ("slow", "mul64", "sel_vcc") and prev_class in ("slow", "mul64", "sel_vcc"):
                slow_pairs += 1
                scheduled += COST["slow_behind_slow"]
            prev_class = c
            if c == "fast":
                bound += COST["fast"]
                if slow_cadence:
                    poisoned += 1
                    scheduled += COST["fast_at_slow_cadence"]
                else:
                    scheduled += COST["fast"]
            elif c in ("slow", "mul64", "sel_vcc"):
                bound += COST[c]
                scheduled += COST[c]
                slow_cadence = True
            else:
                slow_cadence = False  # any non-VALU instruction of the wave ends the slow cadence
        dst = args.split(",")[0].strip() if args else ""
        if op.startswith("v_"):
            since += 1
            if op.startswith("v_cmp") and (dst == "vcc" or not op.endswith("_e64")):
                last_vcc_writer, since = "v_cmp", 0
            elif re.search(r"^\S+,\s*vcc\b", args) and not op.startswith("v_cndmask"):  # carry-out into vcc
                last_vcc_writer, since = "valu", 0
        elif op.startswith("s_") and re.match(r"vcc(_lo|_hi)?\b", dst):
            last_vcc_writer, since = "salu", 0
    valu = sum(classes.get(k, 0) for k in ("fast", "slow", "mul64", "sel_vcc"))
    return {"classes": classes, "opcodes": ops, "valu": valu, "stale_vcc_selects": stale,
            "cycles_per_wave_iteration_bound": round(bound, 1), "cycles_as_scheduled": round(scheduled, 1),
            "poisoned_fast_instructions": poisoned, "separators_behind_slow": separators, "slow_directly_behind_slow": slow_pairs,
            "cycles_at_2_per_instruction": 2 * valu,
            "fast_share": round(classes.get("fast", 0) / valu, 4) if valu else 0.0}


def signature(body):
    return (sum(1 for _, o, _ in body if o == "ds_read_b128"), sum(1 for _, o, _ in body if o.startswith("v_mad_u64_u32")))


def pick_loop(ins, nopp, ndeal, rng_stub=False):
    """The straight-line iteration of `nopp` opponents and `ndeal` table cards: ndeal whole card records (ds_read_b128;
    an opponent's card is read as b32 + b64: the evaluator needs three of its four words) and one 64-bit multiply-add
    per random word (nopp + ceil(ndeal / 2)) and per bounded draw (2 * nopp + ndeal)."""
    words = nopp + (ndeal + 1) // 2
    want = (ndeal, (0 if rng_stub else words) + 2 * nopp + ndeal)
    found = [(j, i) for j, i in loops_of(ins) if signature(ins[j:i + 1]) == want]
    if not found:
        raise SystemExit("no straight-line loop with %d card records and %d multiply-adds (nopp %d, ndeal %d)"
                         % (want[0], want[1], nopp, ndeal))
    if len(found) > 1:
        print("note: %d loops fit the signature, the first is taken" % len(found), file=sys.stderr)
    return found[0]


def build_variant(define, workdir):
    """A diagnostic build of mcq_kernels.hip with one stage stubbed out (device code only) -> code object path."""
    src = os.path.join(ROOT, "neuron_poker_amd", "csrc", "mcq_kernels.hip")
    out = os.path.join(workdir, "var_%s.co" % define)
    run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c", "-D" + define,
         "-Wno-unused-function", "-o", out, src])
    return out


def kernel_sources_sha256():
    """the same hash bench.py ties profiles/ to (bench.kernel_source_hash)"""
    sys.path.insert(0, ROOT)
    import bench
    return bench.kernel_source_hash()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "neuron_poker_amd", "libmcq_hip.so"))
    ap.add_argument("--kernel", default="mcq_eval_kernelILi0ELb0")
    ap.add_argument("--nopp", type=int, default=5)
    ap.add_argument("--ndeal", type=int, default=5)
    ap.add_argument("--vcc-window", type=int, default=2)
    ap.add_argument("--json")
    ap.add_argument("--dump", help="write the loop's disassembly here")
    ap.add_argument("--roles", action="store_true", help="split by role through the MCQ_ABLATE_* builds (three compiles)")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as wd:
        co = code_object(a.lib, wd)
        name, ins = kernel_instructions(disassemble(co), a.kernel)
        j, i = pick_loop(ins, a.nopp, a.ndeal)
        body = ins[j:i + 1]
        h = histogram(body, a.vcc_window, [(x - j, y - j) for x, y in guarded_ranges(ins, j, i)])
        if a.dump:
            with open(a.dump, "w") as f:
                for ad, op, args in body:
                    f.write("%06x  %-8s %s %s\n" % (ad, classify(op, args), op, args))
        roles = None
        if a.roles:
            roles = {}
            full_valu = h["valu"]
            rest = full_valu
            for define, role in (("MCQ_ABLATE_RNG", "rng_and_bounded_draws"), ("MCQ_ABLATE_HOLES", "hole_scan"),
                                 ("MCQ_ABLATE_EVAL", "evaluator")):
                vco = build_variant(define, wd)
                _, vins = kernel_instructions(disassemble(vco), a.kernel)
                # the stubbed build keeps the card reads; the RNG stub drops the random words' multiply-adds
                best = None
                for vj, vi in loops_of(vins):
                    vb = vins[vj:vi + 1]
                    if sum(1 for _, o, _ in vb if o == "ds_read_b128") == 2 * a.nopp + a.ndeal:
                        vh = histogram(vb, a.vcc_window)
                        if best is None or abs(vh["valu"] - full_valu) < abs(best["valu"] - full_valu):
                            best = vh
                if best is None:
                    roles[role] = None
                    continue
                roles[role] = {"valu": full_valu - best["valu"],
                               "by_class": {k: h["classes"].get(k, 0) - best["classes"].get(k, 0)
                                            for k in ("fast", "slow", "mul64", "sel_vcc")}}
                rest -= full_valu - best["valu"]
            roles["rest_cards_board_compare_tally_loop"] = {"valu": rest}
    out = {"kernel": name, "loop": {"nopp": a.nopp, "ndeal": a.ndeal, "first": "%x" % body[0][0], "last": "%x" % body[-1][0],
                                    "instructions": len(body)},
           "costs_simd_cycles": COST, "kernel_sources_sha256": kernel_sources_sha256(), **h}
    if roles is not None:
        out["roles"] = roles
    print(json.dumps({k: v for k, v in out.items() if k != "opcodes"}, indent=1))
    for c in ("fast", "slow", "mul64", "sel_vcc", "lds", "vmem", "salu", "wait"):
        if c in h["opcodes"]:
            print("%-8s %s" % (c, ", ".join("%s x%d" % kv for kv in sorted(h["opcodes"][c].items(), key=lambda kv: -kv[1]))))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
