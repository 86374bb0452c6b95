// mt_bench.hip -- tuning harness for the parity mode's stream walk (csrc/mcq_mt.hpp): the kernel body of
// mcq_mt_parse_kernel alone, on synthetic queries, timed with HIP events; query 0's draws and passes are checked against
// the sequential host walk (csrc/mcq_replay.hpp).  Build here (cross-compiles), run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DVARIANT...] -o tools/mt_bench/mt_bench tools/mt_bench/mt_bench.hip
//   tools/mt_bench/mt_bench <queries> <runs> <players> <board cards>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#ifdef MT_STAMPS /* where a lone wave's batch spends its cycles: s_memtime deltas between the stamps of mcq_mt_batch */
__device__ unsigned long long g_stamps[8], g_last, g_nb;
#define MCQ_MT_STAMP(k)                                                                     \
    do {                                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                          \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                   \
            if (k == 0) g_nb++; else g_stamps[k] += now_ - g_last;                          \
            if (k == 0 && g_last) g_stamps[0] += now_ - g_last; /* loop, regeneration, flush */ \
            g_last = now_;                                                                  \
        }                                                                                   \
    } while (0)
#endif
#include "../../neuron_poker_amd/csrc/mcq_mt.hpp"
#include "../../neuron_poker_amd/csrc/mcq_replay.hpp"

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int kBlock = 256;
__global__ __launch_bounds__(kBlock) void parse_kernel(uint32_t n, uint32_t seed32, uint32_t L0, uint32_t n_opp, uint32_t n_deal,
                                                       uint32_t runs, uint8_t *draws, uint64_t per_query, unsigned long long *passes,
                                                       uint32_t *counter) {
    __shared__ __attribute__((aligned(16))) McqMtWave ws[kBlock / 64];
    McqMtWave &w = ws[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(counter, 1u);
        const uint32_t qi = __builtin_amdgcn_readfirstlane(t);
        if (qi >= n) break;
        MCQ_WAVE_SYNC();
        mcq_mt_seed(w, seed32 + qi);
        MCQ_WAVE_SYNC();
        McqMtState st = {MCQ_MT_N, 0u, 0u, 0u, 0ull};
        mcq_mt_parse_query(w, st, L0, n_opp, n_deal, runs, draws + (uint64_t)qi * per_query, ((uint64_t)runs + 63u) & ~63ull);
        passes[qi] = st.passes;
    }
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 4096, runs = argc > 2 ? atoi(argv[2]) : 50000;
    const uint32_t npl = argc > 3 ? atoi(argv[3]) : 3, nb = argc > 4 ? atoi(argv[4]) : 0;
    const uint32_t n_opp = npl - 1, n_deal = 5 - nb, D = 2 * n_opp + n_deal, L0 = 50 - nb;
    const uint64_t stride = ((uint64_t)runs + 63u) & ~63ull, per_query = stride * D;
    uint8_t *d_draws;
    unsigned long long *d_passes;
    uint32_t *d_counter;
    CHECK(hipMalloc(&d_draws, per_query * n + 64));
    CHECK(hipMalloc(&d_passes, sizeof(unsigned long long) * n));
    CHECK(hipMalloc(&d_counter, 4));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    uint32_t blocks = (n + kBlock / 64 - 1) / (kBlock / 64);
    if (blocks > 8u * prop.multiProcessorCount) blocks = 8u * prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipMemset(d_counter, 0, 4));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(parse_kernel, dim3(blocks), dim3(kBlock), 0, 0, n, 1000u, L0, n_opp, n_deal, runs, d_draws, per_query, d_passes,
                           d_counter);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    // query 0 against the sequential walk
    std::vector<uint8_t> got(per_query), ref(per_query + 64);
    unsigned long long p0;
    CHECK(hipMemcpy(got.data(), d_draws, per_query, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&p0, d_passes, 8, hipMemcpyDeviceToHost));
    mcq_query q;
    memset(&q, 0, sizeof q);
    q.n_board = (uint8_t)nb;
    q.n_players = (uint8_t)npl;
    q.runs = runs;
    const uint64_t pr = mcq_replay_parse(q, 1000u, ref.data(), stride);
    bool ok = pr == p0;
    for (uint32_t d = 0; d < D && ok; d++) ok = memcmp(got.data() + d * stride, ref.data() + d * stride, runs) == 0;
    const double words = (double)n * runs * (D * 64.0 / 50.0); /* rough: 78 % of the words are accepted */
    printf("%u queries x %u runs x %u players, %u board: %.3f ms  (~%.3g words/s)  query 0 %s\n", n, runs, npl, nb, best,
           words / (best * 1e-3), ok ? "== sequential walk" : "MISMATCH");
#ifdef MT_STAMPS
    unsigned long long st[8], nbatch;
    CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st));
    CHECK(hipMemcpyFromSymbol(&nbatch, HIP_SYMBOL(g_nb), sizeof nbatch));
    const char *names[6] = {"between batches (loop, regeneration, flush)", "words -> E, first guess", "rounds", "slot, ring write", "partner read, masks", "bookkeeping"};
    for (int k = 0; k < 6; k++) printf("  %-46s %7.1f memtime ticks per batch\n", names[k], (double)st[k] / (double)nbatch);
#endif
    return ok ? 0 : 1;
}
