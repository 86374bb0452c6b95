#!/usr/bin/env python3
"""Generates tools/ubench/issue_probe2.hip -- second round of the issue-cadence probes (see gen_issue_probe.py).

issue_probe found: a wave that executes ONE instruction of the 4-cycle class (v_bcnt, shifts left, compares, min/max,
every three-operand form except v_bitop3, multiplies, SDWA/DPP ...) issues ALL its VALU instructions at the 4-cycle
cadence for a long while afterwards -- the plain adds around it too --, a wave that never does runs at 2.3 cycles beside
such waves on the same SIMD, and 1023 adds + one v_bcnt directly in front of the loop's backward branch cost nothing.
So: what ends the slow cadence?

  A  `gap` adds, one v_bcnt, candidate C, N adds ... loop of ~2048 instructions, the v_bcnt far from the loop's own
     branch.  C = nothing | a taken s_branch to the next instruction | s_nop 15 | s_sleep 1 | s_waitcnt 0 | v_nop |
     s_setprio | s_cbranch not taken | a second taken branch ...
  B  the same with the v_bcnt inside a region that is always branched over (fetched perhaps, never executed)
  C  N sweep with the v_bcnt mid-loop (how many adds does one v_bcnt slow down?)
  D  LDS / global loads among adds: do they disturb the fast cadence?
"""
import os

from gen_issue_probe import HEAD, kernel

HERE = os.path.dirname(os.path.abspath(__file__))


def adds(n, start=0):
    return "\n".join("v_add_u32 %%%d, %%%d, %%8" % ((start + r) % 8, (start + r) % 8) for r in range(n))


def main():
    out = [HEAD]
    table = []
    uid = [0]

    def add(label, body, n_instr, clob=None):
        name = "k_%d" % uid[0]
        uid[0] += 1
        out.append(kernel(name, body, 1, clob or []))
        table.append((label, name, n_instr))

    total = 2048
    # C: N sweep, the slow op mid-run
    for n in (31, 63, 127, 255, 511, 1023, 2047):
        groups = total // (n + 1)
        body = "\n".join(adds(n // 2) + "\nv_bcnt_u32_b32 %0, %0, %8\n" + adds(n - n // 2, 3) for _ in range(groups))
        add("sweep: per %d adds one v_bcnt (mid-run, far from the loop branch)" % n, body, groups * (n + 1))
    add("sweep: 2048 adds, no slow instruction", adds(total), total)
    # A: what ends the slow cadence?  pattern per 256 instructions: 127 adds, bcnt, C, 128 adds
    cands = [
        ("nothing", ""),
        ("taken s_branch to the next instruction", "s_branch 0"),
        ("two taken s_branch", "s_branch 0\ns_branch 0"),
        ("s_nop 15", "s_nop 15"),
        ("4 x s_nop 15", "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15"),
        ("s_sleep 1", "s_sleep 1"),
        ("s_waitcnt vmcnt(0) lgkmcnt(0)", "s_waitcnt vmcnt(0) lgkmcnt(0)"),
        ("v_nop", "v_nop"),
        ("s_setprio 0", "s_setprio 0"),
        ("s_cbranch_scc0 not taken (s_cmp_eq_u32 s9, s9 first)", "s_cmp_eq_u32 %9, %9\ns_cbranch_scc0 0"),
        ("s_cbranch_scc1 taken to the next instruction", "s_cmp_eq_u32 %9, %9\ns_cbranch_scc1 0"),
        ("s_barrier", "s_barrier"),
        ("s_mov_b64 exec, exec", "s_mov_b64 exec, exec"),
        ("s_getpc_b64 + s_setpc_b64 (jump to the next instruction)", "s_getpc_b64 s[20:21]\ns_add_u32 s20, s20, 12\ns_addc_u32 s21, s21, 0\ns_setpc_b64 s[20:21]"),
    ]
    for label, c in cands:
        groups = total // 256
        one = adds(127) + "\nv_bcnt_u32_b32 %0, %0, %8\n" + (c + "\n" if c else "") + adds(128, 7)
        body = "\n".join(one for _ in range(groups))
        add("reset? 127 adds, v_bcnt, [%s], 128 adds" % label, body, groups * 256, ["s20", "s21", "scc"])
    # the same with the candidate in front of the v_bcnt (control: a branch alone is harmless?)
    one = adds(127) + "\ns_branch 0\n" + adds(129, 7)
    add("control: 127 adds, taken s_branch, 129 adds (no slow instruction)", "\n".join(one for _ in range(8)), 8 * 256)
    # B: the v_bcnt is branched over (never executed)
    one = adds(127) + "\ns_cmp_eq_u32 %9, %9\ns_cbranch_scc1 2\nv_bcnt_u32_b32 %0, %0, %8\n" + adds(128, 7)
    add("fetched, never executed: 127 adds, [branch over a v_bcnt], 128 adds", "\n".join(one for _ in range(8)), 8 * 255,
        ["scc"])
    # exec = 0 around the slow instruction
    one = adds(127) + "\ns_mov_b64 s[20:21], exec\ns_mov_b64 exec, 0\nv_bcnt_u32_b32 %0, %0, %8\ns_mov_b64 exec, s[20:21]\n" + adds(128, 7)
    add("executed with EXEC = 0: 127 adds, v_bcnt under an empty mask, 128 adds", "\n".join(one for _ in range(8)), 8 * 256,
        ["s20", "s21"])
    # bursts: k slow instructions together per 256
    for k in (4, 16, 64):
        one = adds(128 - k // 2) + "\n" + "\n".join("v_bcnt_u32_b32 %%%d, %%%d, %%8" % (r % 8, r % 8) for r in range(k)) + "\n" + adds(128 - k + k // 2, 5)
        add("burst: %d v_bcnt together per 256 instructions" % k, "\n".join(one for _ in range(8)), 8 * 256)
        one2 = one.replace(adds(128 - k + k // 2, 5), "s_branch 0\n" + adds(128 - k + k // 2, 5))
        add("burst: %d v_bcnt together per 256, then a taken s_branch" % k, "\n".join(one2 for _ in range(8)), 8 * 256)
    # D: memory instructions among adds (LDS address 0 of the block's allocation; results land in v40 / v41 unread)
    one = adds(7) + "\nds_read_b32 v40, v42"
    add("mix7: ds_read_b32", "v_mov_b32 v42, 0\n" + "\n".join(one for _ in range(256)), 256 * 8, ["v40", "v42"])
    one = adds(7) + "\nds_read_b128 v[44:47], v42"
    add("mix7: ds_read_b128", "v_mov_b32 v42, 0\n" + "\n".join(one for _ in range(256)), 256 * 8, ["v42", "v44", "v45", "v46", "v47"])
    out.append("struct B { const char *name; void (*k)(uint32_t *, int); int per_iter; };\n")
    out.append("static const B bs[] = {\n" + "".join('    {"%s", %s, %d},\n' % t for t in table) + "};\n")
    out.append(r'''
int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount, iters = 200;
    const double ghz = 2.4;
    uint32_t *out;
    CHK(hipMalloc(&out, (size_t)n_cu * 8 * 1024 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int W : {4, 2}) {
        printf("---- %d waves per SIMD: SIMD-cycles per VALU instruction (2.4 GHz assumed)\n", W);
        for (const B &b : bs) {
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 64, 0, out, 5);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 64, 0, out, iters);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-86s %6.2f\n", b.name, ms * 1e-3 * ghz * 1e9 / ((double)iters * b.per_iter * W));
        }
    }
    return 0;
}
''')
    with open(os.path.join(HERE, "issue_probe2.hip"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()
