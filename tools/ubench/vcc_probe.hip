// vcc_probe.hip -- what does a v_cndmask cost by where its mask comes from?  (tools/ubench measured 22.7 SIMD-cycles for a
// VOP2 v_cndmask_b32 that reads a VCC nobody wrote recently, against 4.4 for the same select with an SGPR-pair mask.)
//   hipcc --offload-arch=gfx950 -O3 -o vcc_probe tools/ubench/vcc_probe.hip && ./vcc_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define R8(M) M("%0") M("%1") M("%2") M("%3") M("%4") M("%5") M("%6") M("%7")
#define KERN(name, PRE, M, ...)                                                                         \
    __global__ void name(uint32_t *out, int iters) {                                                    \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 77, a4 = a0 * 5, a5 = ~a0, a6 = a0 << 3, \
                 a7 = a0 + blockIdx.x;                                                                  \
        uint32_t s = (blockIdx.x * 2654435761u) | 1u;                                                   \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(PRE ".rept 16\n" R8(M) ".endr\n"                                               \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(s), "s"(i) : __VA_ARGS__);                                               \
        }                                                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;             \
    }
#define I_CND_VCC(x) "v_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_CND_E64_VCC(x) "v_cndmask_b32_e64 " x ", " x ", %8, vcc\n"
#define I_CND_SGPR(x) "v_cndmask_b32_e64 " x ", " x ", %8, s[20:21]\n"
#define I_SAND_CND(x) "s_and_b64 vcc, s[20:21], s[22:23]\nv_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_SAND_CND64(x) "s_and_b64 s[24:25], s[20:21], s[22:23]\nv_cndmask_b32_e64 " x ", " x ", %8, s[24:25]\n"
#define I_VCMP_CND(x) "v_cmp_lt_u32 vcc, " x ", %8\nv_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_ADDC(x) "v_addc_co_u32 " x ", vcc, " x ", %8, vcc\n"
KERN(k_vcc_stale, "", I_CND_VCC, "memory")
KERN(k_vcc_vcmp_once, "v_cmp_lt_u32 vcc, %0, %8\n", I_CND_VCC, "vcc")
KERN(k_vcc_smov_once, "s_mov_b64 vcc, 0x5555\n", I_CND_VCC, "vcc")
KERN(k_e64_vcc, "v_cmp_lt_u32 vcc, %0, %8\n", I_CND_E64_VCC, "vcc")
KERN(k_sgpr, "s_mov_b64 s[20:21], 0x5555\n", I_CND_SGPR, "s20", "s21")
KERN(k_sand_vcc, "s_mov_b64 s[20:21], 0x5555\ns_mov_b64 s[22:23], 0x3333\n", I_SAND_CND, "vcc", "s20", "s21", "s22", "s23")
KERN(k_sand_sgpr, "s_mov_b64 s[20:21], 0x5555\ns_mov_b64 s[22:23], 0x3333\n", I_SAND_CND64, "s20", "s21", "s22", "s23", "s24", "s25")
KERN(k_vcmp_cnd, "", I_VCMP_CND, "vcc")
KERN(k_addc, "v_cmp_lt_u32 vcc, %0, %8\n", I_ADDC, "vcc")
#define I_VCMP_NOP_CND(x) "v_cmp_lt_u32 vcc, " x ", %8\ns_nop 1\nv_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_VCMP_VALU_CND(x) "v_cmp_lt_u32 vcc, " x ", %8\nv_add_u32 v40, v40, %8\nv_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_VCMP_2VALU_CND(x) "v_cmp_lt_u32 vcc, " x ", %8\nv_add_u32 v40, v40, %8\nv_add_u32 v41, v41, %8\nv_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_VCMP64_CND64(x) "v_cmp_lt_u32_e64 s[20:21], " x ", %8\nv_add_u32 v40, v40, %8\nv_cndmask_b32_e64 " x ", " x ", %8, s[20:21]\n"
#define I_VCMP_VALU_CND64VCC(x) "v_cmp_lt_u32 vcc, " x ", %8\nv_add_u32 v40, v40, %8\nv_cndmask_b32_e64 " x ", " x ", %8, vcc\n"
#define I_CND_SDWA(x) "v_cndmask_b32_sdwa " x ", " x ", %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
#define I_SMOV_CND_SDWA(x) "s_mov_b64 vcc, s[20:21]\nv_cndmask_b32_sdwa " x ", " x ", %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
#define I_VCMP_CND_SDWA(x) "v_cmp_lt_u32 vcc, " x ", %8\nv_add_u32 v40, v40, %8\nv_cndmask_b32_sdwa " x ", " x ", %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
#define I_SMOV_VALU8_CND(x) "s_mov_b64 vcc, s[20:21]\nv_add_u32 v40, v40, %8\nv_add_u32 v41, v41, %8\nv_add_u32 v42, v42, %8\nv_add_u32 v43, v43, %8\nv_cndmask_b32 " x ", " x ", %8, vcc\n"
KERN(k_cnd_sdwa_stale, "v_cmp_lt_u32 vcc, %0, %8\n", I_CND_SDWA, "vcc")
KERN(k_smov_cnd_sdwa, "s_mov_b64 s[20:21], 0x5555\n", I_SMOV_CND_SDWA, "vcc", "s20", "s21")
KERN(k_vcmp_cnd_sdwa, "", I_VCMP_CND_SDWA, "vcc", "v40")
KERN(k_smov_valu4_cnd, "s_mov_b64 s[20:21], 0x5555\n", I_SMOV_VALU8_CND, "vcc", "s20", "s21", "v40", "v41", "v42", "v43")
#define I_VALU5(x) "v_add_u32 v40, v40, %8\nv_add_u32 v41, v41, %8\nv_add_u32 v42, v42, %8\nv_add_u32 v43, v43, %8\nv_add_u32 " x ", " x ", %8\n"
#define I_SMOV_VALU5(x) "s_mov_b64 s[22:23], s[20:21]\nv_add_u32 v40, v40, %8\nv_add_u32 v41, v41, %8\nv_add_u32 v42, v42, %8\nv_add_u32 v43, v43, %8\nv_add_u32 " x ", " x ", %8\n"
#define I_VALU4_CND64(x) "v_add_u32 v40, v40, %8\nv_add_u32 v41, v41, %8\nv_add_u32 v42, v42, %8\nv_add_u32 v43, v43, %8\nv_cndmask_b32_e64 " x ", " x ", %8, s[20:21]\n"
#define I_VALU5_SAMEREG(x) "v_add_u32 " x ", " x ", %8\nv_add_u32 " x ", " x ", %8\nv_add_u32 " x ", " x ", %8\nv_add_u32 " x ", " x ", %8\nv_add_u32 " x ", " x ", %8\n"
#define I_ADD1(x) "v_add_u32 " x ", " x ", %8\n"
#define I_ADD2(x) "v_add_u32 " x ", " x ", %8\nv_add_u32 v40, v40, %8\n"
#define I_XOR1(x) "v_xor_b32 " x ", " x ", %8\n"
#define I_MAD641(x) "v_mad_u64_u32 v[40:41], s[20:21], " x ", %8, v[40:41]\n"
#define I_BITOP(x) "v_and_or_b32 " x ", " x ", %8, %8\n"
KERN(k_add1, "", I_ADD1, "memory")
KERN(k_add2, "", I_ADD2, "v40")
KERN(k_xor1, "", I_XOR1, "memory")
KERN(k_mad641, "", I_MAD641, "v40", "v41", "s20", "s21")
KERN(k_andor1, "", I_BITOP, "memory")
KERN(k_valu5, "", I_VALU5, "v40", "v41", "v42", "v43")
KERN(k_smov_valu5, "s_mov_b64 s[20:21], 0x5555\n", I_SMOV_VALU5, "s20", "s21", "s22", "s23", "v40", "v41", "v42", "v43")
KERN(k_valu4_cnd64, "s_mov_b64 s[20:21], 0x5555\n", I_VALU4_CND64, "s20", "s21", "v40", "v41", "v42", "v43")
KERN(k_valu5_same, "", I_VALU5_SAMEREG, "memory")
KERN(k_vcmp_nop_cnd, "", I_VCMP_NOP_CND, "vcc")
KERN(k_vcmp_valu_cnd, "", I_VCMP_VALU_CND, "vcc", "v40")
KERN(k_vcmp_2valu_cnd, "", I_VCMP_2VALU_CND, "vcc", "v40", "v41")
KERN(k_vcmp64_cnd64, "", I_VCMP64_CND64, "s20", "s21", "v40")
KERN(k_vcmp_valu_cnd64vcc, "", I_VCMP_VALU_CND64VCC, "vcc", "v40")
struct B { const char *name; void (*k)(uint32_t *, int); int per_iter; };
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount, iters = 2000;
    uint32_t *out;
    hipMalloc(&out, (size_t)n_cu * 8 * 256 * 4);
    B bs[] = {{"v_cndmask_b32 vcc (VCC written by nobody in the loop)", k_vcc_stale, 128},
              {"v_cmp once, then 128 x v_cndmask_b32 vcc", k_vcc_vcmp_once, 128},
              {"s_mov vcc once, then 128 x v_cndmask_b32 vcc", k_vcc_smov_once, 128},
              {"v_cmp once, then 128 x v_cndmask_b32_e64 .., vcc", k_e64_vcc, 128},
              {"128 x v_cndmask_b32_e64 .., s[20:21]", k_sgpr, 128},
              {"128 x (s_and_b64 vcc + v_cndmask_b32 vcc)", k_sand_vcc, 256},
              {"128 x (s_and_b64 s[24:25] + v_cndmask_b32_e64 s[24:25])", k_sand_sgpr, 256},
              {"128 x (v_cmp vcc + v_cndmask_b32 vcc)", k_vcmp_cnd, 256},
              {"v_cmp once, then 128 x v_addc_co_u32 vcc", k_addc, 128},
              {"v_cmp once, then 128 x v_cndmask_b32_sdwa vcc", k_cnd_sdwa_stale, 128},
              {"128 x (s_mov vcc, v_cndmask_b32_sdwa vcc)", k_smov_cnd_sdwa, 256},
              {"128 x (v_cmp vcc, v_add, v_cndmask_b32_sdwa vcc)", k_vcmp_cnd_sdwa, 384},
              {"128 x (s_mov vcc, 4 x v_add, v_cndmask_b32 vcc)", k_smov_valu4_cnd, 768},
              {"128 x v_add_u32 x, x, v", k_add1, 128},
              {"128 x (v_add x; v_add v40)", k_add2, 256},
              {"128 x v_xor_b32", k_xor1, 128},
              {"128 x v_mad_u64_u32 (one accumulator pair)", k_mad641, 128},
              {"128 x v_and_or_b32 (VOP3)", k_andor1, 128},
              {"128 x (5 x v_add: v40..v43, x)", k_valu5, 640},
              {"128 x (s_mov, 5 x v_add: v40..v43, x)", k_smov_valu5, 768},
              {"128 x (4 x v_add v40..v43, v_cndmask_b32_e64 x, s[20:21])", k_valu4_cnd64, 640},
              {"128 x (5 x v_add x, x) dependent", k_valu5_same, 640},
              {"128 x (v_cmp vcc, s_nop 1, v_cndmask_b32 vcc)  [per VALU instr]", k_vcmp_nop_cnd, 256},
              {"128 x (v_cmp vcc, v_add, v_cndmask_b32 vcc)", k_vcmp_valu_cnd, 384},
              {"128 x (v_cmp vcc, v_add, v_add, v_cndmask_b32 vcc)", k_vcmp_2valu_cnd, 512},
              {"128 x (v_cmp_e64 s[20:21], v_add, v_cndmask_b32_e64 s[20:21])", k_vcmp64_cnd64, 384},
              {"128 x (v_cmp vcc, v_add, v_cndmask_b32_e64 .., vcc)", k_vcmp_valu_cnd64vcc, 384}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int W : {1, 4}) {
        printf("---- %d wave(s) per SIMD\n", W);
        for (auto &b : bs) {
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(b.k, dim3(n_cu * W), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-62s %8.3f ms  %6.2f SIMD-cycles per wave-instruction\n", b.name, ms,
                   ms * 1e-3 * 2.4e9 / ((double)iters * b.per_iter * W));
        }
    }
    return 0;
}
