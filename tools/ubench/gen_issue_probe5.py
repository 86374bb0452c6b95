#!/usr/bin/env python3
"""Generates tools/ubench/issue_probe5.hip -- fifth round: does the VGPR BANK of a fast-class instruction's sources decide
its issue cost?  (issue_probe4: v_bitop3_b32 and v_cndmask_b32_e64 cost 2.1 or 4.0 cycles depending on WHICH registers the
compiler happened to pick.)  Physical registers are named explicitly here: destination v10, sources chosen by their
number mod 4 (the register file has four banks on gfx9).  Independent instructions (the destination is not a source)."""
import os

from gen_issue_probe import HEAD

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [
    # (label, instruction text with physical registers)
    ("v_and_b32 (VOP2)  src banks 0,1  dst 2", "v_and_b32 v10, v12, v13"),
    ("v_and_b32 (VOP2)  src banks 0,0  dst 2", "v_and_b32 v10, v12, v16"),
    ("v_and_b32 (VOP2)  src banks 0,1  dst 0", "v_and_b32 v20, v12, v13"),
    ("v_and_b32 (VOP2)  src banks 0,0  dst 0", "v_and_b32 v20, v12, v16"),
    ("v_and_b32 (VOP2)  same register twice", "v_and_b32 v10, v12, v12"),
    ("v_and_b32 (VOP2)  src0 inline constant", "v_and_b32 v10, 7, v13"),
    ("v_and_b32 (VOP2)  src0 sgpr", "v_and_b32 v10, s20, v13"),
    ("v_and_b32 (VOP2)  src0 literal", "v_and_b32 v10, 0x12345, v13"),
    ("v_add_u32  src banks 0,1", "v_add_u32 v10, v12, v13"),
    ("v_add_u32  src banks 0,0", "v_add_u32 v10, v12, v16"),
    ("v_sub_u32  src banks 0,0", "v_sub_u32 v10, v12, v16"),
    ("v_lshrrev_b32  shift in a register, banks 0,0", "v_lshrrev_b32 v10, v12, v16"),
    ("v_lshrrev_b32  shift constant", "v_lshrrev_b32 v10, 3, v16"),
    ("v_mov_b32", "v_mov_b32 v10, v12"),
    ("v_bitop3_b32  src banks 0,1,2", "v_bitop3_b32 v11, v12, v13, v14 bitop3:0x96"),
    ("v_bitop3_b32  src banks 0,0,1", "v_bitop3_b32 v11, v12, v16, v13 bitop3:0x96"),
    ("v_bitop3_b32  src banks 0,1,1", "v_bitop3_b32 v11, v12, v13, v17 bitop3:0x96"),
    ("v_bitop3_b32  src banks 0,1,0", "v_bitop3_b32 v11, v12, v13, v16 bitop3:0x96"),
    ("v_bitop3_b32  src banks 0,0,0", "v_bitop3_b32 v11, v12, v16, v20 bitop3:0x96"),
    ("v_bitop3_b32  two sources the same register", "v_bitop3_b32 v11, v12, v13, v13 bitop3:0x96"),
    ("v_bitop3_b32  one source an inline constant", "v_bitop3_b32 v11, v12, 1, v14 bitop3:0x96"),
    ("v_bitop3_b32  one source an sgpr", "v_bitop3_b32 v11, v12, s20, v14 bitop3:0x96"),
    ("v_cndmask_b32_e64  src banks 0,1  mask sgpr pair", "v_cndmask_b32_e64 v10, v12, v13, s[22:23]"),
    ("v_cndmask_b32_e64  src banks 0,0  mask sgpr pair", "v_cndmask_b32_e64 v10, v12, v16, s[22:23]"),
    ("v_cndmask_b32_e64  src0 inline constant", "v_cndmask_b32_e64 v10, 0, v13, s[22:23]"),
    ("v_cndmask_b32_e64  both inline constants", "v_cndmask_b32_e64 v10, 0, 1, s[22:23]"),
    ("v_cndmask_b32_e64  mask vcc", "v_cndmask_b32_e64 v10, v12, v13, vcc"),
    ("v_or3_b32  src banks 0,1,2", "v_or3_b32 v11, v12, v13, v14"),
    ("v_add3_u32  src banks 0,1,2", "v_add3_u32 v11, v12, v13, v14"),
    ("v_bcnt_u32_b32  src banks 0,1", "v_bcnt_u32_b32 v10, v12, v13"),
]


def main():
    out = [HEAD]
    table = []
    for k, (label, ins) in enumerate(CASES):
        body = "\\n".join([ins] * 64)
        out.append('__global__ void k_%d(uint32_t *out, int iters) {\n    INIT\n    for (int i = 0; i < iters; i++)\n'
                   '        asm volatile(".rept 32\\n%s\\n.endr\\n" : OUTS : "v"(s), "s"(i) : "v10", "v11", "v12", "v13", "v14", "v15", '
                   '"v16", "v17", "v20", "s20", "s22", "s23", "vcc");\n    FIN\n}\n' % (k, body))
        table.append((label, "k_%d" % k, 32 * 64))
    out.append("struct B { const char *name; void (*k)(uint32_t *, int); int nv; };\n")
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
    printf("SIMD-cycles per VALU instruction (2.4 GHz assumed) by waves per SIMD:                2      4\n");
    for (const B &b : bs) {
        printf("%-56s", b.name);
        for (int W : {2, 4}) {
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 64, 0, out, 5);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 64, 0, out, iters);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %6.2f", ms * 1e-3 * ghz * 1e9 / ((double)iters * b.nv * W));
        }
        printf("\n");
    }
    return 0;
}
''')
    with open(os.path.join(HERE, "issue_probe5.hip"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()
