// ubench.hip -- instruction issue-cost microbenchmarks for gfx950 (design input for the equity kernels).
// Each kernel issues 8 independent chains x 16 repeats = 128 instructions of one kind per loop iteration.
// Reported: SIMD cycles per wave-instruction at W waves per SIMD (time * clock / (iters * 128 * W)).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define R8(M) M("%0") M("%1") M("%2") M("%3") M("%4") M("%5") M("%6") M("%7")

#define KERNEL32(name, M, ...)                                                                      \
    __global__ void name(uint32_t *out, int iters) {                                                  \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 77, a4 = a0 * 5, a5 = ~a0, a6 = a0 << 3, \
                 a7 = a0 + blockIdx.x;                                                                \
        uint32_t s = (blockIdx.x * 2654435761u) | 1u;                                                 \
        for (int i = 0; i < iters; i++) {                                                             \
            asm volatile(".rept 16\n" R8(M) ".endr\n"                                                 \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(s), "s"(i) : __VA_ARGS__);                                                    \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }

#define KERNEL64(name, M, ...)                                                                      \
    __global__ void name(uint32_t *out, int iters) {                                                  \
        uint64_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 77, a4 = a0 * 5, a5 = ~a0, a6 = a0 << 3, \
                 a7 = a0 + blockIdx.x;                                                                \
        uint32_t s = (blockIdx.x * 2654435761u) | 1u;                                                 \
        for (int i = 0; i < iters; i++) {                                                             \
            asm volatile(".rept 16\n" R8(M) ".endr\n"                                                 \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(s), "s"(i) : __VA_ARGS__);                                                    \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7); \
    }

#define I_ADD(x) "v_add_u32 " x ", " x ", %8\n"
#define I_AND(x) "v_and_b32 " x ", " x ", %8\n"
#define I_XOR(x) "v_xor_b32 " x ", " x ", %8\n"
#define I_BCNT(x) "v_bcnt_u32_b32 " x ", " x ", %8\n"
#define I_MULHI(x) "v_mul_hi_u32 " x ", " x ", %8\n"
#define I_MULLO(x) "v_mul_lo_u32 " x ", " x ", %8\n"
#define I_MUL24(x) "v_mul_u32_u24 " x ", " x ", %8\n"
#define I_MULHI24(x) "v_mul_hi_u32_u24 " x ", " x ", %8\n"
#define I_FFBH(x) "v_ffbh_u32 " x ", " x "\n"
#define I_BFE(x) "v_bfe_u32 " x ", " x ", %8, 5\n"
#define I_ALIGNBIT(x) "v_alignbit_b32 " x ", " x ", " x ", 7\n"
#define I_BITOP3(x) "v_bitop3_b32 " x ", " x ", %8, %8 bitop3:0x96\n"
#define I_MIN(x) "v_min_u32 " x ", " x ", %8\n"
#define I_PERM(x) "v_perm_b32 " x ", " x ", %8, %8\n"
#define I_ADD3(x) "v_add3_u32 " x ", " x ", %8, %8\n"
#define I_LSHLOR(x) "v_lshl_or_b32 " x ", " x ", 3, %8\n"
#define I_ANDOR(x) "v_and_or_b32 " x ", " x ", %8, %8\n"
#define I_LSHLADD(x) "v_lshl_add_u32 " x ", " x ", 1, %8\n"
#define I_CMPCND(x) "v_cmp_lt_u32 vcc, " x ", %8\nv_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_CMPE64(x) "v_cmp_lt_u32_e64 s[20:21], " x ", %8\nv_cndmask_b32_e64 " x ", " x ", %8, s[20:21]\n"
#define I_CNDVCC(x) "v_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_SUBB(x) "v_subrev_u32 " x ", %8, " x "\n"
#define I_PKMIN(x) "v_pk_min_u16 " x ", " x ", %8\n"
#define I_PKADD(x) "v_pk_add_u16 " x ", " x ", %8\n"
#define I_PKSUB(x) "v_pk_sub_u16 " x ", " x ", %8\n"
#define I_MADU24(x) "v_mad_u32_u24 " x ", " x ", %8, %8\n"
#define I_READLANE(x) "v_readlane_b32 s20, " x ", 3\nv_writelane_b32 " x ", s20, 5\n"
#define I_DPP(x) "v_add_u32_dpp " x ", " x ", " x " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_SDWA(x) "v_or_b32_sdwa " x ", " x ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define I_MOV(x) "v_mov_b32 " x ", %8\n"
#define I_SALU(x) "s_add_u32 s20, s20, %9\ns_and_b32 s21, s21, %9\n"
#define I_MIX_VS(x) "v_add_u32 " x ", " x ", %8\ns_add_u32 s20, s20, %9\n"

#define I_LSHL64(x) "v_lshlrev_b64 " x ", 3, " x "\n"
#define I_LSHL64V(x) "v_lshlrev_b64 " x ", %8, " x "\n"
#define I_LSHR64V(x) "v_lshrrev_b64 " x ", %8, " x "\n"
#define I_ADD64(x) "v_lshl_add_u64 " x ", " x ", 0, " x "\n"
#define I_MAD64(x) "v_mad_u64_u32 " x ", vcc, %8, %8, " x "\n"


#define I_OR(x) "v_or_b32 " x ", " x ", %8\n"
#define I_SUB(x) "v_sub_u32 " x ", " x ", %8\n"
#define I_SUBREV(x) "v_subrev_u32 " x ", " x ", %8\n"
#define I_LSHL(x) "v_lshlrev_b32 " x ", 3, " x "\n"
#define I_LSHLV(x) "v_lshlrev_b32 " x ", %8, " x "\n"
#define I_LSHR(x) "v_lshrrev_b32 " x ", 3, " x "\n"
#define I_LSHRV(x) "v_lshrrev_b32 " x ", %8, " x "\n"
#define I_ASHR(x) "v_ashrrev_i32 " x ", 3, " x "\n"
#define I_MAX(x) "v_max_u32 " x ", " x ", %8\n"
#define I_MAXI(x) "v_max_i32 " x ", " x ", %8\n"
#define I_NOT(x) "v_not_b32 " x ", " x "\n"
#define I_XNOR(x) "v_xnor_b32 " x ", " x ", %8\n"
#define I_CMPONLY(x) "v_cmp_lt_u32 vcc, " x ", %8\n"
#define I_CMPONLY64(x) "v_cmp_lt_u32_e64 s[20:21], " x ", %8\n"
#define I_CNDS(x) "v_cndmask_b32_e64 " x ", " x ", %8, s[22:23]\n"
#define I_ADDCO(x) "v_add_co_u32 " x ", vcc, " x ", %8\n"
#define I_ADDC(x) "v_addc_co_u32 " x ", vcc, " x ", %8, vcc\n"
#define I_BFM(x) "v_bfm_b32 " x ", " x ", %8\n"
#define I_MBCNT(x) "v_mbcnt_lo_u32_b32 " x ", " x ", %8\n"
#define I_ADDF(x) "v_add_f32 " x ", " x ", %8\n"
#define I_MAXF(x) "v_max_f32 " x ", " x ", %8\n"
#define I_FMA(x) "v_fma_f32 " x ", " x ", %8, %8\n"
#define I_MULF(x) "v_mul_f32 " x ", " x ", %8\n"
#define I_ANDI(x) "v_and_b32 " x ", 0xffff, " x "\n"
#define I_ADDI(x) "v_add_u32 " x ", 0x33333333, " x "\n"
#define I_ADDS(x) "v_add_u32 " x ", s9, " x "\n"
#define I_MOVREL(x) "v_mov_b32 " x ", " x "\n"
#define I_CVT(x) "v_cvt_f32_u32 " x ", " x "\n"
#define I_SAD(x) "v_sad_u32 " x ", " x ", %8, %8\n"
#define I_MED3(x) "v_med3_u32 " x ", " x ", %8, %8\n"
#define I_MAX3(x) "v_max3_u32 " x ", " x ", %8, %8\n"
#define I_XAD(x) "v_xad_u32 " x ", " x ", %8, %8\n"
#define I_OR3(x) "v_or3_b32 " x ", " x ", %8, %8\n"
#define I_ADDLSHL(x) "v_add_lshl_u32 " x ", " x ", %8, 2\n"
#define I_BFI(x) "v_bfi_b32 " x ", " x ", %8, %8\n"
KERNEL32(k_add, I_ADD, "memory")
KERNEL32(k_and, I_AND, "memory")
KERNEL32(k_xor, I_XOR, "memory")
KERNEL32(k_bcnt, I_BCNT, "memory")
KERNEL32(k_mulhi, I_MULHI, "memory")
KERNEL32(k_mullo, I_MULLO, "memory")
KERNEL32(k_mul24, I_MUL24, "memory")
KERNEL32(k_mulhi24, I_MULHI24, "memory")
KERNEL32(k_ffbh, I_FFBH, "memory")
KERNEL32(k_bfe, I_BFE, "memory")
KERNEL32(k_alignbit, I_ALIGNBIT, "memory")
KERNEL32(k_bitop3, I_BITOP3, "memory")
KERNEL32(k_min, I_MIN, "memory")
KERNEL32(k_perm, I_PERM, "memory")
KERNEL32(k_add3, I_ADD3, "memory")
KERNEL32(k_lshlor, I_LSHLOR, "memory")
KERNEL32(k_andor, I_ANDOR, "memory")
KERNEL32(k_lshladd, I_LSHLADD, "memory")
KERNEL32(k_cmpcnd, I_CMPCND, "vcc")
KERNEL32(k_cmpe64, I_CMPE64, "s20", "s21")
KERNEL32(k_cndvcc, I_CNDVCC, "memory")
KERNEL32(k_pkmin, I_PKMIN, "memory")
KERNEL32(k_pkadd, I_PKADD, "memory")
KERNEL32(k_pksub, I_PKSUB, "memory")
KERNEL32(k_madu24, I_MADU24, "memory")
KERNEL32(k_readlane, I_READLANE, "s20")
KERNEL32(k_dpp, I_DPP, "memory")
KERNEL32(k_sdwa, I_SDWA, "memory")
KERNEL32(k_mov, I_MOV, "memory")
KERNEL32(k_salu, I_SALU, "s20", "s21")
KERNEL32(k_mix_vs, I_MIX_VS, "s20")
KERNEL32(k_or, I_OR, "memory")
KERNEL32(k_sub, I_SUB, "memory")
KERNEL32(k_subrev, I_SUBREV, "memory")
KERNEL32(k_lshl, I_LSHL, "memory")
KERNEL32(k_lshlv, I_LSHLV, "memory")
KERNEL32(k_lshr, I_LSHR, "memory")
KERNEL32(k_lshrv, I_LSHRV, "memory")
KERNEL32(k_ashr, I_ASHR, "memory")
KERNEL32(k_max, I_MAX, "memory")
KERNEL32(k_maxi, I_MAXI, "memory")
KERNEL32(k_not, I_NOT, "memory")
KERNEL32(k_xnor, I_XNOR, "memory")
KERNEL32(k_cmponly, I_CMPONLY, "vcc")
KERNEL32(k_cmponly64, I_CMPONLY64, "s20","s21")
KERNEL32(k_cnds, I_CNDS, "memory")
KERNEL32(k_addco, I_ADDCO, "vcc")
KERNEL32(k_addc, I_ADDC, "vcc")
KERNEL32(k_bfm, I_BFM, "memory")
KERNEL32(k_mbcnt, I_MBCNT, "memory")
KERNEL32(k_addf, I_ADDF, "memory")
KERNEL32(k_maxf, I_MAXF, "memory")
KERNEL32(k_fma, I_FMA, "memory")
KERNEL32(k_mulf, I_MULF, "memory")
KERNEL32(k_andi, I_ANDI, "memory")
KERNEL32(k_addi, I_ADDI, "memory")
KERNEL32(k_adds, I_ADDS, "memory")
KERNEL32(k_cvt, I_CVT, "memory")
KERNEL32(k_sad, I_SAD, "memory")
KERNEL32(k_med3, I_MED3, "memory")
KERNEL32(k_max3, I_MAX3, "memory")
KERNEL32(k_xad, I_XAD, "memory")
KERNEL32(k_or3, I_OR3, "memory")
KERNEL32(k_addlshl, I_ADDLSHL, "memory")
KERNEL32(k_bfi, I_BFI, "memory")

// ---- realistic dependent groups (per group of 6: the hole-register update)
#define I_HOLEGRP(x) "v_sub_u32 v40, %8, " x "\nv_and_b32 v40, 0x80808080, v40\nv_bcnt_u32_b32 v41, v40, v41\nv_lshrrev_b32 v40, 7, v40\nv_xor_b32 v40, 0x1010101, v40\nv_sub_u32 " x ", " x ", v40\n"
KERNEL32(k_holegrp, I_HOLEGRP, "v40", "v41")
// dependent chain of fast ops on ONE register (no ILP inside the wave)
#define I_DEPCHAIN(x) "v_add_u32 %0, %0, %8\nv_xor_b32 %0, %0, %8\n"
KERNEL32(k_depchain, I_DEPCHAIN, "memory")
// fast op with literal
#define I_XORLIT(x) "v_xor_b32 " x ", 0x1010101, " x "\n"
KERNEL32(k_xorlit, I_XORLIT, "memory")
// alternating fast / slow
#define I_ALT(x) "v_add_u32 " x ", " x ", %8\nv_bcnt_u32_b32 " x ", " x ", %8\n"
KERNEL32(k_alt, I_ALT, "memory")
// fast op followed by s_waitcnt lgkmcnt(0) (no LDS outstanding)
#define I_WAITC(x) "v_add_u32 " x ", " x ", %8\ns_waitcnt lgkmcnt(0)\n"
KERNEL32(k_waitc, I_WAITC, "memory")
// s_nop interleave
#define I_NOP(x) "v_add_u32 " x ", " x ", %8\ns_nop 0\n"
KERNEL32(k_nop, I_NOP, "memory")

// ---- pairing-rule probes: G8 = one asm group touching %0..%7 (8 instructions unless noted)
#define KGROUP(name, BODY, ...)                                                                       \
    __global__ void name(uint32_t *out, int iters) {                                                  \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 77, a4 = a0 * 5, a5 = ~a0, a6 = a0 << 3, \
                 a7 = a0 + blockIdx.x;                                                                \
        uint32_t s = (blockIdx.x * 2654435761u) | 1u;                                                 \
        for (int i = 0; i < iters; i++) {                                                             \
            asm volatile(".rept 16\n" BODY ".endr\n"                                                \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(s), "s"(i) : __VA_ARGS__);                                             \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;           \
    }
KGROUP(g_mixfast, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_and_b32 %2,%2,%8\nv_or_b32 %3,%3,%8\nv_sub_u32 %4,%4,%8\nv_lshrrev_b32 %5,1,%5\nv_not_b32 %6,%6\nv_add_u32 %7,%7,%8\n", "memory")
KGROUP(g_deppairs, "v_add_u32 %0,%0,%8\nv_xor_b32 %0,%0,%8\nv_add_u32 %1,%1,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %2,%2,%8\nv_add_u32 %3,%3,%8\nv_xor_b32 %3,%3,%8\n", "memory")
KGROUP(g_altfs_indep, "v_add_u32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_bcnt_u32_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\n", "memory")
KGROUP(g_ffss, "v_add_u32 %0,%0,%8\nv_add_u32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_add_u32 %5,%5,%8\nv_bcnt_u32_b32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\n", "memory")
KGROUP(g_fffs, "v_add_u32 %0,%0,%8\nv_add_u32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_add_u32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\n", "memory")
KGROUP(g_hole_nolit, "v_sub_u32 %6,%8,%0\nv_and_b32 %6,%7,%6\nv_bcnt_u32_b32 %1,%6,%1\nv_lshrrev_b32 %6,7,%6\nv_xor_b32 %6,%5,%6\nv_sub_u32 %0,%0,%6\n", "memory")
KGROUP(g_hole_ilp2, "v_sub_u32 %6,%8,%0\nv_sub_u32 %7,%8,%2\nv_and_b32 %6,%4,%6\nv_and_b32 %7,%4,%7\nv_bcnt_u32_b32 %1,%6,%1\nv_bcnt_u32_b32 %3,%7,%3\nv_lshrrev_b32 %6,7,%6\nv_lshrrev_b32 %7,7,%7\nv_xor_b32 %6,%5,%6\nv_xor_b32 %7,%5,%7\nv_sub_u32 %0,%0,%6\nv_sub_u32 %2,%2,%7\n", "memory")
KGROUP(g_slow_dep, "v_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_bcnt_u32_b32 %3,%3,%8\n", "memory")
KGROUP(g_fast_samedst, "v_add_u32 %0,%1,%8\nv_add_u32 %0,%2,%8\nv_add_u32 %0,%3,%8\nv_add_u32 %0,%4,%8\nv_add_u32 %0,%5,%8\nv_add_u32 %0,%6,%8\nv_add_u32 %0,%7,%8\nv_add_u32 %0,%1,%8\n", "memory")
KGROUP(g_run_7_1, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_bcnt_u32_b32 %0,%0,%8\n", "memory")
KGROUP(g_run_15_1, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_bcnt_u32_b32 %0,%0,%8\n", "memory")
KGROUP(g_run_31_1, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_bcnt_u32_b32 %0,%0,%8\n", "memory")
KGROUP(g_run_63_1, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_bcnt_u32_b32 %0,%0,%8\n", "memory")
KGROUP(g_run_16_16, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_bcnt_u32_b32 %4,%4,%8\nv_bcnt_u32_b32 %5,%5,%8\nv_bcnt_u32_b32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_bcnt_u32_b32 %4,%4,%8\nv_bcnt_u32_b32 %5,%5,%8\nv_bcnt_u32_b32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\n", "memory")
KGROUP(g_run_32_32, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_bcnt_u32_b32 %4,%4,%8\nv_bcnt_u32_b32 %5,%5,%8\nv_bcnt_u32_b32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_bcnt_u32_b32 %4,%4,%8\nv_bcnt_u32_b32 %5,%5,%8\nv_bcnt_u32_b32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_bcnt_u32_b32 %4,%4,%8\nv_bcnt_u32_b32 %5,%5,%8\nv_bcnt_u32_b32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_bcnt_u32_b32 %4,%4,%8\nv_bcnt_u32_b32 %5,%5,%8\nv_bcnt_u32_b32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\n", "memory")
KGROUP(g_run_8_8, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\nv_bcnt_u32_b32 %4,%4,%8\nv_bcnt_u32_b32 %5,%5,%8\nv_bcnt_u32_b32 %6,%6,%8\nv_bcnt_u32_b32 %7,%7,%8\n", "memory")
KGROUP(g_run_4_4, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\n", "memory")
KGROUP(g_run_12_4, "v_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_add_u32 %4,%4,%8\nv_xor_b32 %5,%5,%8\nv_add_u32 %6,%6,%8\nv_xor_b32 %7,%7,%8\nv_add_u32 %0,%0,%8\nv_xor_b32 %1,%1,%8\nv_add_u32 %2,%2,%8\nv_xor_b32 %3,%3,%8\nv_bcnt_u32_b32 %0,%0,%8\nv_bcnt_u32_b32 %1,%1,%8\nv_bcnt_u32_b32 %2,%2,%8\nv_bcnt_u32_b32 %3,%3,%8\n", "memory")
KERNEL64(k_lshl64, I_LSHL64, "memory")
KERNEL64(k_lshl64v, I_LSHL64V, "memory")
KERNEL64(k_lshr64v, I_LSHR64V, "memory")
KERNEL64(k_add64, I_ADD64, "memory")
KERNEL64(k_mad64, I_MAD64, "vcc")

// LDS lookups: random index into a 256-dword table (like sel8), dependent chain of 8 independent lookups
__global__ void k_lds_rand(uint32_t *out, int iters) {
    __shared__ uint32_t tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) tab[i] = (i * 167u + 13u) & 255u;
    __syncthreads();
    uint32_t a[8];
    for (int k = 0; k < 8; k++) a[k] = (threadIdx.x * 31u + k * 17u) & 255u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = tab[a[k]];
        }
    }
    uint32_t x = 0;
    for (int k = 0; k < 8; k++) x ^= a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void k_lds_b64_rand(uint32_t *out, int iters) {
    __shared__ uint2 tab[64];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) tab[i] = make_uint2((i * 37u + 5u) & 63u, i);
    __syncthreads();
    uint32_t a[8];
    for (int k = 0; k < 8; k++) a[k] = (threadIdx.x * 31u + k * 17u) & 63u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) { uint2 v = tab[a[k]]; a[k] = v.x; acc += v.y; }
        }
    }
    uint32_t x = acc;
    for (int k = 0; k < 8; k++) x ^= a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void k_clock(uint64_t *out) {
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t a = threadIdx.x;
    for (int i = 0; i < 2000000; i++) asm volatile("v_add_u32 %0, %0, %0\n" : "+v"(a));
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = a; }
}

typedef void (*kfn)(uint32_t *, int);
struct Bench { const char *name; kfn f; int per_iter; };

int main(int argc, char **argv) {
    int iters = 2000;
    uint32_t *d_out;
    CHK(hipMalloc(&d_out, 256 * 8 * 4 * 64 * 4 * 4));
    uint64_t *d_clk, h_clk[3];
    CHK(hipMalloc(&d_clk, 24));
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d_clk);
    CHK(hipMemcpy(h_clk, d_clk, 24, hipMemcpyDeviceToHost));
    double ghz = (double)h_clk[0] / (double)h_clk[1] * 0.1;
    printf("shader clock (memtime/memrealtime*100MHz), one wave: %.3f GHz\n", ghz);
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    printf("CUs %d, clockRate %d kHz\n", n_cu, prop.clockRate);
    std::vector<Bench> bs = {
        {"v_add_u32", k_add, 128}, {"v_or_b32", k_or, 128}, {"v_sub_u32", k_sub, 128}, {"v_subrev_u32", k_subrev, 128}, {"v_lshlrev_b32 imm", k_lshl, 128}, {"v_lshlrev_b32 vgpr", k_lshlv, 128}, {"v_lshrrev_b32 imm", k_lshr, 128}, {"v_lshrrev_b32 vgpr", k_lshrv, 128}, {"v_ashrrev_i32", k_ashr, 128}, {"v_max_u32", k_max, 128}, {"v_max_i32", k_maxi, 128}, {"v_not_b32", k_not, 128}, {"v_xnor_b32", k_xnor, 128}, {"v_cmp_lt_u32 -> vcc only", k_cmponly, 128}, {"v_cmp_lt_u32_e64 -> sgpr only", k_cmponly64, 128}, {"v_cndmask_b32_e64 (sgpr mask, no cmp)", k_cnds, 128}, {"v_add_co_u32", k_addco, 128}, {"v_addc_co_u32", k_addc, 128}, {"v_bfm_b32", k_bfm, 128}, {"v_mbcnt_lo_u32_b32", k_mbcnt, 128}, {"v_add_f32", k_addf, 128}, {"v_max_f32", k_maxf, 128}, {"v_fma_f32", k_fma, 128}, {"v_mul_f32", k_mulf, 128}, {"v_and_b32 literal", k_andi, 128}, {"v_add_u32 literal", k_addi, 128}, {"v_add_u32 sgpr operand", k_adds, 128}, {"v_cvt_f32_u32", k_cvt, 128}, {"v_sad_u32", k_sad, 128}, {"v_med3_u32", k_med3, 128}, {"v_max3_u32", k_max3, 128}, {"v_xad_u32", k_xad, 128}, {"v_or3_b32", k_or3, 128}, {"v_add_lshl_u32", k_addlshl, 128}, {"v_bfi_b32", k_bfi, 128},  {"v_and_b32", k_and, 128}, {"v_xor_b32", k_xor, 128}, {"v_mov_b32", k_mov, 128},
        {"v_bcnt_u32_b32", k_bcnt, 128}, {"v_mul_hi_u32", k_mulhi, 128}, {"v_mul_lo_u32", k_mullo, 128},
        {"v_mul_u32_u24", k_mul24, 128}, {"v_mul_hi_u32_u24", k_mulhi24, 128}, {"v_mad_u32_u24", k_madu24, 128},
        {"v_ffbh_u32", k_ffbh, 128}, {"v_bfe_u32", k_bfe, 128}, {"v_alignbit_b32", k_alignbit, 128},
        {"v_bitop3_b32", k_bitop3, 128}, {"v_min_u32", k_min, 128}, {"v_perm_b32", k_perm, 128},
        {"v_add3_u32", k_add3, 128}, {"v_lshl_or_b32", k_lshlor, 128}, {"v_and_or_b32", k_andor, 128},
        {"v_lshl_add_u32", k_lshladd, 128}, {"v_cmp(vcc)+v_cndmask pair", k_cmpcnd, 256},
        {"v_cmp_e64(sgpr)+v_cndmask_e64 pair", k_cmpe64, 256}, {"v_cndmask_b32 vcc", k_cndvcc, 128},
        {"v_pk_min_u16", k_pkmin, 128}, {"v_pk_add_u16", k_pkadd, 128}, {"v_pk_sub_u16", k_pksub, 128},
        {"v_readlane+v_writelane pair", k_readlane, 256}, {"v_add_u32_dpp", k_dpp, 128}, {"v_or_b32_sdwa", k_sdwa, 128},
        {"s_add+s_and pair (SALU only)", k_salu, 256}, {"v_add + s_add interleaved", k_mix_vs, 256},
        {"hole group (6 instr: sub,and-lit,bcnt,lshr,xor-lit,sub)", k_holegrp, 768}, {"dependent chain add,xor on one reg", k_depchain, 256}, {"v_xor_b32 literal", k_xorlit, 128}, {"alternating v_add / v_bcnt", k_alt, 256}, {"v_add + s_waitcnt lgkmcnt(0)", k_waitc, 128}, {"v_add + s_nop 0", k_nop, 128}, {"8 different fast ops, independent", g_mixfast, 128}, {"dependent pairs add->xor (4 pairs)", g_deppairs, 128}, {"alt fast/slow, all independent", g_altfs_indep, 128}, {"f f s s independent", g_ffss, 128}, {"f f f s independent", g_fffs, 128}, {"hole group, registers not literals (6)", g_hole_nolit, 96}, {"two interleaved hole groups (12)", g_hole_ilp2, 192}, {"slow dependent pairs (bcnt)", g_slow_dep, 128}, {"fast ops same dst", g_fast_samedst, 128}, {"run: 7 fast then 1 slow", g_run_7_1, 128}, {"run: 15 fast then 1 slow", g_run_15_1, 256}, {"run: 31 fast then 1 slow", g_run_31_1, 512}, {"run: 63 fast then 1 slow", g_run_63_1, 1024}, {"run: 16 fast then 16 slow", g_run_16_16, 512}, {"run: 32 fast then 32 slow", g_run_32_32, 1024}, {"run: 8 fast then 8 slow", g_run_8_8, 256}, {"run: 4 fast then 4 slow", g_run_4_4, 128}, {"run: 12 fast then 4 slow", g_run_12_4, 256}, {"v_lshlrev_b64 imm", k_lshl64, 128}, {"v_lshlrev_b64 vgpr", k_lshl64v, 128}, {"v_lshrrev_b64 vgpr", k_lshr64v, 128},
        {"v_lshl_add_u64", k_add64, 128}, {"v_mad_u64_u32", k_mad64, 128},
        {"ds_read_b32 random 256-dword table (dependent x8)", k_lds_rand, 128},
        {"ds_read_b64 random 64-entry table (dependent x8)", k_lds_b64_rand, 128},
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int W : {4}) {
        printf("---- %d wave(s) per SIMD (grid %d blocks x 256 threads)\n", W, n_cu * W);
        for (auto &b : bs) {
            int it = iters;
            hipLaunchKernelGGL(b.f, dim3(n_cu * W), dim3(256), 0, 0, d_out, 10);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(b.f, dim3(n_cu * W), dim3(256), 0, 0, d_out, it);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            double cyc = ms * 1e-3 * ghz * 1e9 / ((double)it * b.per_iter * W);
            printf("%-52s %7.3f ms  %6.2f SIMD-cycles per wave-instruction\n", b.name, ms, cyc);
        }
    }
    return 0;
}
