#!/usr/bin/env python3
"""Generates tools/ubench/issue_probe.hip: what does ONE instruction of kind X cost the SIMD when it sits among plain
`v_add_u32`s -- the context the equity kernels' instructions actually live in?

tools/ubench/ubench.hip measured every opcode back to back (128 of a kind): two classes, 2.3 and 4.3 SIMD-cycles per
wave64 instruction at four waves per SIMD, and found that runs of the fast class with even one v_bcnt per 64 cost 4 cycles
per instruction throughout.  tools/ubench/vcc_probe.hip then saw four adds + one v_cndmask_b32_e64 run at 2.1.  So "slow"
is not one class.  This probe prices each opcode of the hot loops IN CONTEXT:

  mix7:X    groups of seven independent v_add_u32 and one X       -> cycles per group; minus 7 x the add's own cost =
            what X adds (its marginal cost); an X that drags its neighbours into the 4-cycle cadence shows up as ~ +18
  mix3:X    three adds + one X (the density of the kernels)
  runs      N adds then one v_bcnt (N = 7 ... 1023): how long does the slow cadence last?
  per-wave  blocks of 16 waves, four per SIMD; the first `k` waves of every SIMD run adds only, the others v_bcnt only;
            every wave times itself (s_memtime): is the cadence a property of the wave or of the SIMD?

Output: one line per probe, SIMD-cycles per group and per instruction, at 4 waves per SIMD (and 1, 2, 8 for a few).
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# X: (label, asm with {x} = the X chain's register, {s} = a VGPR operand, {t} = a scratch VGPR pair v[40:41]); clobbers
XS = [
    ("v_add_u32 (reference: all adds)", "v_add_u32 {x}, {x}, {s}", []),
    ("v_xor_b32", "v_xor_b32 {x}, {x}, {s}", []),
    ("v_lshrrev_b32 7", "v_lshrrev_b32 {x}, 7, {x}", []),
    ("v_and_b32 literal", "v_and_b32 {x}, 0x1010101, {x}", []),
    ("v_add_u32 literal", "v_add_u32 {x}, 0x80808100, {x}", []),
    ("v_add_u32 sgpr", "v_add_u32 {x}, s9, {x}", []),
    ("v_and_b32 sgpr", "v_and_b32 {x}, s9, {x}", []),
    ("v_or_b32 sgpr", "v_or_b32 {x}, s9, {x}", []),
    ("v_mov_b32 sgpr", "v_mov_b32 {x}, s9", []),
    ("v_mov_b32 vgpr", "v_mov_b32 {x}, {s}", []),
    ("v_lshlrev_b32 13", "v_lshlrev_b32 {x}, 13, {x}", []),
    ("v_lshlrev_b32 1", "v_lshlrev_b32 {x}, 1, {x}", []),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {x}, {x}, {s}", []),
    ("v_max_u32", "v_max_u32 {x}, {x}, {s}", []),
    ("v_max3_u32", "v_max3_u32 {x}, {x}, {s}, {s}", []),
    ("v_perm_b32", "v_perm_b32 {x}, 0, {x}, 0", []),
    ("v_sad_u8", "v_sad_u8 {x}, {x}, 0, {s}", []),
    ("v_add3_u32", "v_add3_u32 {x}, {x}, {s}, {s}", []),
    ("v_add3_u32 sgpr", "v_add3_u32 {x}, {x}, {s}, s9", []),
    ("v_and_or_b32", "v_and_or_b32 {x}, {x}, {s}, {s}", []),
    ("v_and_or_b32 sgpr", "v_and_or_b32 {x}, {x}, s9, {s}", []),
    ("v_or3_b32", "v_or3_b32 {x}, {x}, {s}, {s}", []),
    ("v_bfi_b32 sgpr", "v_bfi_b32 {x}, s9, {x}, {s}", []),
    ("v_bfe_u32", "v_bfe_u32 {x}, {x}, 3, 16", []),
    ("v_bitop3_b32", "v_bitop3_b32 {x}, {x}, {s}, {s} bitop3:0xe0", []),
    ("v_lshl_add_u32 4", "v_lshl_add_u32 {x}, {x}, 4, {s}", []),
    ("v_lshl_or_b32 13", "v_lshl_or_b32 {x}, {x}, 13, {s}", []),
    ("v_mul_u32_u24", "v_mul_u32_u24 {x}, 6, {x}", []),
    ("v_mad_u64_u32 (sgpr multiplier)", "v_mad_u64_u32 v[40:41], s[20:21], {x}, s9, v[40:41]", ["v40", "v41", "s20", "s21"]),
    ("v_mad_u64_u32 (vgpr)", "v_mad_u64_u32 v[40:41], s[20:21], {x}, {s}, v[40:41]", ["v40", "v41", "s20", "s21"]),
    ("v_mul_hi_u32", "v_mul_hi_u32 {x}, {x}, {s}", []),
    ("v_mul_lo_u32", "v_mul_lo_u32 {x}, {x}, {s}", []),
    ("v_lshlrev_b64", "v_lshlrev_b64 v[40:41], {x}, v[40:41]", ["v40", "v41"]),
    ("v_lshl_add_u64", "v_lshl_add_u64 v[40:41], v[40:41], 0, v[42:43]", ["v40", "v41", "v42", "v43"]),
    ("v_cmp_eq_u32 -> vcc", "v_cmp_eq_u32 vcc, {x}, {s}", ["vcc"]),
    ("v_cmp_ne_u32_e64 -> sgpr pair", "v_cmp_ne_u32_e64 s[20:21], {x}, {s}", ["s20", "s21"]),
    ("v_cndmask_b32_e64 sgpr mask", "v_cndmask_b32_e64 {x}, {x}, {s}, s[22:23]", []),
    ("v_cmp vcc + v_cndmask_b32 vcc (2 instr)", "v_cmp_lt_u32 vcc, {x}, {s}\nv_cndmask_b32 {x}, {x}, {s}, vcc", ["vcc"]),
    ("v_cmp_e64 + v_cndmask_e64 (2 instr)", "v_cmp_lt_u32_e64 s[20:21], {x}, {s}\nv_cndmask_b32_e64 {x}, {x}, {s}, s[20:21]", ["s20", "s21"]),
    ("v_cmp vcc + v_addc_co_u32 (2 instr)", "v_cmp_ge_u32 vcc, {x}, {s}\nv_addc_co_u32 {x}, vcc, 0, {x}, vcc", ["vcc"]),
    ("v_addc_co_u32", "v_addc_co_u32 {x}, vcc, 0, {x}, vcc", ["vcc"]),
    ("v_or_b32_sdwa", "v_or_b32_sdwa {x}, {x}, {s} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", []),
    ("v_add_u32_dpp", "v_add_u32_dpp {x}, {x}, {x} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", []),
    ("v_min_u32", "v_min_u32 {x}, {x}, {s}", []),
    ("v_ffbh_u32", "v_ffbh_u32 {x}, {x}", []),
    ("v_pk_add_u16", "v_pk_add_u16 {x}, {x}, {s}", []),
    ("v_sub_u32 + v_lshrrev (2 fast)", "v_sub_u32 {x}, {x}, {s}\nv_lshrrev_b32 {x}, 7, {x}", []),
    ("s_nop 0", "s_nop 0", []),
    ("s_nop 1", "s_nop 1", []),
    ("s_and_b64", "s_and_b64 s[20:21], s[22:23], s[24:25]", ["s20", "s21"]),
    ("s_waitcnt lgkmcnt(0)", "s_waitcnt lgkmcnt(0)", []),
]

HEAD = r'''// GENERATED by tools/ubench/gen_issue_probe.py -- do not edit.  hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define INIT                                                                                                       \
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 77, a4 = a0 * 5, a5 = ~a0, a6 = a0 << 3,  \
             a7 = a0 + blockIdx.x;                                                                                 \
    uint32_t s = (blockIdx.x * 2654435761u) | 1u;
#define OUTS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define FIN out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
'''


def kernel(name, group, reps, clobbers):
    cl = ", ".join('"%s"' % c for c in (clobbers or ["memory"]))
    body = "\\n".join(group.split("\n"))
    return ('__global__ void %s(uint32_t *out, int iters) {\n    INIT\n    for (int i = 0; i < iters; i++)\n'
            '        asm volatile(".rept %d\\n%s\\n.endr\\n" : OUTS : "v"(s), "s"(i) : %s);\n    FIN\n}\n'
            % (name, reps, body, cl))


def main():
    out = [HEAD]
    table = []  # (label, kernel name, instructions per loop trip, groups per loop trip, n_adds per group)
    for k, (label, x, clob) in enumerate(XS):
        for n_add in (7, 3):
            adds = "\n".join("v_add_u32 %%%d, %%%d, %%8" % (r, r) for r in range(n_add))
            xi = x.format(x="%7", s="%8")
            group = adds + "\n" + xi
            reps = 16 if n_add == 7 else 32
            name = "k_mix%d_%d" % (n_add, k)
            out.append(kernel(name, group, reps, clob))
            n_x = len(xi.split("\n"))
            table.append(("mix%d: %s" % (n_add, label), name, reps * (n_add + n_x), reps, n_add))
    # runs of N adds then one bcnt
    for n in (7, 15, 31, 63, 127, 255, 511, 1023):
        adds = "\n".join("v_add_u32 %%%d, %%%d, %%8" % (r % 8, r % 8) for r in range(n))
        group = adds + "\nv_bcnt_u32_b32 %0, %0, %8"
        reps = max(1, 1024 // (n + 1))
        name = "k_run_%d" % n
        out.append(kernel(name, group, reps, []))
        table.append(("run: %d adds then 1 v_bcnt" % n, name, reps * (n + 1), reps, n))
    # the hole-scan group as compiled (mcq_hole_reg): sub, lshr, and-lit, sad_u8, add3 -- and a variant made of fast ops
    out.append(kernel("k_hole_now", "v_sub_u32 %6, %8, %0\nv_lshrrev_b32 %6, 7, %6\nv_and_b32 %6, 0x1010101, %6\n"
                                    "v_sad_u8 %1, %6, 0, %1\nv_add3_u32 %0, %0, %6, %7", 25, []))
    table.append(("hole scan as compiled: sub, lshr, and-lit, sad_u8, add3", "k_hole_now", 125, 25, 3))
    out.append(kernel("k_count_now", "v_sub_u32 %6, %8, %0\nv_and_b32 %6, 0x80808080, %6\nv_bcnt_u32_b32 %1, %6, %1", 40, []))
    table.append(("hole count as compiled: sub, and-lit, bcnt", "k_count_now", 120, 40, 2))
    out.append('''
// ---- is the slow cadence a property of the wave or of the SIMD?  Blocks of 1024 threads = 16 waves, four per SIMD
// (wave w on SIMD w % 4).  Waves with (w / 4) < n_fast run adds only, the others run `slow_kind` only; every wave
// times its own loop.  t[block][wave] = shader cycles.
template <int SLOW_KIND>
__global__ __launch_bounds__(1024) void k_per_wave(uint64_t *t, uint32_t *out, int iters, int n_fast) {
    INIT
    const int w = threadIdx.x >> 6;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if ((w >> 2) < n_fast) {
        for (int i = 0; i < iters; i++)
            asm volatile(".rept 16\\nv_add_u32 %0, %0, %8\\nv_add_u32 %1, %1, %8\\nv_add_u32 %2, %2, %8\\nv_add_u32 %3, %3, %8\\n"
                         "v_add_u32 %4, %4, %8\\nv_add_u32 %5, %5, %8\\nv_add_u32 %6, %6, %8\\nv_add_u32 %7, %7, %8\\n.endr\\n"
                         : OUTS : "v"(s), "s"(i) : "memory");
    } else if (SLOW_KIND == 0) {
        for (int i = 0; i < iters; i++)
            asm volatile(".rept 16\\nv_bcnt_u32_b32 %0, %0, %8\\nv_bcnt_u32_b32 %1, %1, %8\\nv_bcnt_u32_b32 %2, %2, %8\\nv_bcnt_u32_b32 %3, %3, %8\\n"
                         "v_bcnt_u32_b32 %4, %4, %8\\nv_bcnt_u32_b32 %5, %5, %8\\nv_bcnt_u32_b32 %6, %6, %8\\nv_bcnt_u32_b32 %7, %7, %8\\n.endr\\n"
                         : OUTS : "v"(s), "s"(i) : "memory");
    } else {
        for (int i = 0; i < iters; i++)
            asm volatile(".rept 16\\nv_mad_u64_u32 v[40:41], s[20:21], %0, %8, v[40:41]\\nv_mad_u64_u32 v[42:43], s[20:21], %1, %8, v[42:43]\\n"
                         "v_mad_u64_u32 v[44:45], s[20:21], %2, %8, v[44:45]\\nv_mad_u64_u32 v[46:47], s[20:21], %3, %8, v[46:47]\\n"
                         "v_mad_u64_u32 v[40:41], s[20:21], %4, %8, v[40:41]\\nv_mad_u64_u32 v[42:43], s[20:21], %5, %8, v[42:43]\\n"
                         "v_mad_u64_u32 v[44:45], s[20:21], %6, %8, v[44:45]\\nv_mad_u64_u32 v[46:47], s[20:21], %7, %8, v[46:47]\\n.endr\\n"
                         : OUTS : "v"(s), "s"(i) : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s20", "s21");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 16 + w] = t1 - t0;
    FIN
}
''')
    out.append("struct B { const char *name; void (*k)(uint32_t *, int); int per_iter, groups, n_add; };\n")
    out.append("static const B bs[] = {\n" + "".join('    {"%s", %s, %d, %d, %d},\n' % t for t in table) + "};\n")
    out.append(r'''
int main(int argc, char **argv) {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount, iters = 1000;
    const double ghz = 2.4; /* the shader clock under load, as tools/ubench measures it with s_memtime */
    uint32_t *out;
    CHK(hipMalloc(&out, (size_t)n_cu * 8 * 1024 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto time_of = [&](void (*k)(uint32_t *, int), int W) {
        hipLaunchKernelGGL(k, dim3(n_cu * W), dim3(256), 0, 0, out, 10);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(n_cu * W), dim3(256), 0, 0, out, iters);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms;
    };
    for (int W : {4}) {
        printf("---- %d waves per SIMD: SIMD-cycles per group | per instruction | marginal cost of X (group - n_add x add)\n", W);
        double add_cost = 0;
        for (const B &b : bs) {
            const double ms = time_of(b.k, W);
            const double per_group = ms * 1e-3 * ghz * 1e9 / ((double)iters * b.groups * W);
            const double per_instr = ms * 1e-3 * ghz * 1e9 / ((double)iters * b.per_iter * W);
            if (&b == &bs[0]) add_cost = per_instr;
            printf("%-58s %7.2f | %5.2f | %+6.2f\n", b.name, per_group, per_instr, per_group - b.n_add * add_cost);
        }
    }
    for (int W : {1, 2, 8}) {
        printf("---- %d waves per SIMD (first eight probes)\n", W);
        for (int i = 0; i < 8 && i < (int)(sizeof bs / sizeof bs[0]); i++) {
            const B &b = bs[i];
            const double ms = time_of(b.k, W);
            printf("%-58s %7.2f | %5.2f\n", b.name, ms * 1e-3 * ghz * 1e9 / ((double)iters * b.groups * W),
                   ms * 1e-3 * ghz * 1e9 / ((double)iters * b.per_iter * W));
        }
    }
    uint64_t *d_t;
    std::vector<uint64_t> h_t((size_t)n_cu * 16);
    CHK(hipMalloc(&d_t, h_t.size() * 8));
    for (int kind = 0; kind < 2; kind++)
        for (int n_fast = 0; n_fast <= 4; n_fast++) {
            for (int rep = 0; rep < 2; rep++) {
                if (kind == 0) hipLaunchKernelGGL(k_per_wave<0>, dim3(n_cu), dim3(1024), 0, 0, d_t, out, rep ? 1000 : 10, n_fast);
                else hipLaunchKernelGGL(k_per_wave<1>, dim3(n_cu), dim3(1024), 0, 0, d_t, out, rep ? 1000 : 10, n_fast);
                CHK(hipDeviceSynchronize());
            }
            CHK(hipMemcpy(h_t.data(), d_t, h_t.size() * 8, hipMemcpyDeviceToHost));
            double fast = 0, slow = 0;
            int nf = 0, ns = 0;
            for (int b = 0; b < n_cu; b++)
                for (int w = 0; w < 16; w++) {
                    if ((w >> 2) < n_fast) fast += (double)h_t[b * 16 + w], nf++;
                    else slow += (double)h_t[b * 16 + w], ns++;
                }
            printf("per-wave, slow kind %s, %d of 4 waves per SIMD run adds: a fast wave takes %.2f cycles per instruction of its "
                   "own, a slow wave %.2f  (SIMD total: %.2f instr per cycle)\n", kind ? "v_mad_u64_u32" : "v_bcnt", n_fast,
                   nf ? fast / nf / (1000.0 * 128) : 0.0, ns ? slow / ns / (1000.0 * 128) : 0.0,
                   (nf ? n_fast * (1000.0 * 128) / (fast / nf) : 0.0) + (ns ? (4 - n_fast) * (1000.0 * 128) / (slow / ns) : 0.0));
        }
    return 0;
}
''')
    with open(os.path.join(HERE, "issue_probe.hip"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()
