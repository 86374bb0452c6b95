// mcq_kernels.hip -- gfx950 kernels of the equity engine.
//
// Work decomposition: a WAVE TASK is 1024 iterations of one query; the 64 lanes of the
// wave each run 16 of them (production mode: lane = one RNG stream of 16 consecutive iterations; parity
// mode: iterations interleaved so that the draw bytes of a wave are contiguous).  All lanes of a wave work
// on the same query, so deck length, loop bounds and the query record are wave-uniform (SGPRs, scalar
// branches); only the rare r1 == r2 re-draw diverges.  A persistent grid of waves strides over the task list.
//
// Memory: 16 B in / 104 B out per QUERY; per iteration nothing touches HBM in production mode (parity mode
// reads <= 23 draw bytes).  LDS holds the three lookup tables and the opponents' hole cards of every lane.
#include <hip/hip_runtime.h>

#include "mcq_device.hpp"
#include "mcq_internal.hpp"

namespace {

constexpr int kMaxBlock = 1024; /* 16 waves = 4 per SIMD; one block per CU: tables 97 KB + 16 base decks 16 KB of the 160 KB LDS */

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct LdsTables { /* per block */
    uint32_t tf[8192];
    uint32_t tops[8192];
    uint32_t sd[8192];
    uint32_t sel8[256];
    uint32_t inv[64];
};
static_assert(sizeof(LdsTables) == sizeof(McqTables), "table image is copied word by word");

__device__ __forceinline__ void load_tables(LdsTables &dst, const McqTables *__restrict__ g) {
    const uint4 *src = reinterpret_cast<const uint4 *>(g);
    uint4 *d = reinterpret_cast<uint4 *>(&dst);
    for (uint32_t i = threadIdx.x; i < sizeof(LdsTables) / 16; i += blockDim.x) d[i] = src[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------- prep
// One block.  Validates every query, zeroes its result row and builds the exclusive prefix of wave-task
// counts (prefix[n] = total).  Invalid queries get no tasks, runs = 0 and passes = UINT64_MAX.
__global__ __launch_bounds__(1024) void mcq_prep_kernel(const mcq_query *__restrict__ q, uint32_t n,
                                                        mcq_result *__restrict__ res, uint32_t *__restrict__ prefix) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + tid, tasks = 0;
        if (i < n) {
            const uint4 raw = reinterpret_cast<const uint4 *>(q)[i];
            const McqQueryWords qq = {raw.x, raw.y, raw.z, raw.w};
            bool ok = mcq_query_valid(qq);
            tasks = ok ? (qq.runs() + MCQ_TASK_ITERS - 1) / MCQ_TASK_ITERS : 0u;
            uint64_t *r = reinterpret_cast<uint64_t *>(res + i);
            r[0] = ok ? qq.runs() : 0ull;
            r[1] = ok ? 0ull : ~0ull;
#pragma unroll
            for (int k = 2; k < 13; k++) r[k] = 0;
        }
        part[tid] = tasks;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) { /* Hillis-Steele inclusive scan */
            uint32_t v = tid >= off ? part[tid - off] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (i < n) prefix[i] = carry + part[tid] - tasks;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) prefix[n] = carry;
}

// ---------------------------------------------------------------------------------------------- eval
template <int MODE>
__global__ __launch_bounds__(kMaxBlock) void mcq_eval_kernel(const mcq_query *__restrict__ queries, uint32_t n,
                                                                const uint32_t *__restrict__ prefix,
                                                                mcq_result *__restrict__ res, uint64_t seed,
                                                                uint64_t first_qid, const McqTables *__restrict__ g_tab,
                                                                const uint8_t *__restrict__ draws,
                                                                const uint64_t *__restrict__ draw_off) {
    __shared__ __attribute__((aligned(16))) LdsTables tab;
    __shared__ McqCard base_tab[kMaxBlock]; /* per wave: the query's ordered remaining deck, 64 entries x 16 B */
    load_tables(tab, g_tab);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * waves_per_block + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * waves_per_block;
    const uint32_t total = prefix[n];
    McqCard *base = base_tab + (threadIdx.x & ~63u);

    for (uint32_t t = wave; t < total; t += n_waves) {
        /* query of task t: last q with prefix[q] <= t (wave-uniform binary search, scalar loads) */
        uint32_t lo = 0, hi = n;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (prefix[mid] <= t) lo = mid; else hi = mid;
        }
        const uint32_t qi = __builtin_amdgcn_readfirstlane(lo);
        const uint32_t task = t - prefix[qi];
        const uint4 raw = reinterpret_cast<const uint4 *>(queries)[qi];
        const McqQueryWords q = {raw.x, raw.y, raw.z, raw.w};
        McqQueryCtx qc;
        mcq_query_ctx(q, qc);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); /* the previous task's lookups are done (same wave) */
        base[lane] = mcq_base_entry(qc, lane, tab.sel8);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();

        McqLaneAcc acc = {0, 0, 0};
        if (MODE == MCQ_MODE_PHILOX) {
            const uint32_t stream = task * MCQ_WAVE + lane;
            const uint64_t it0 = (uint64_t)stream * MCQ_STREAM_ITERS;
            if (it0 < qc.runs) {
                McqCtrDraws dr;
                dr.w = 0;
                dr.rng.seed(seed, first_qid + qi, stream);
                const uint32_t cnt = (uint32_t)min((uint64_t)MCQ_STREAM_ITERS, (uint64_t)qc.runs - it0);
                for (uint32_t j = 0; j < cnt; j++)
                    mcq_iteration(qc, dr, base, tab.tf, tab.tops, tab.sd, tab.inv, acc);
                acc.passes = cnt * qc.n_opp; /* MCQ-CTR v2: one attempt per opponent, never re-drawn */
            }
        } else {
            const uint64_t stride = (qc.runs + 63u) & ~63ull;
            const uint8_t *dbase = draws + draw_off[qi];
            for (uint32_t j = 0; j < MCQ_STREAM_ITERS; j++) {
                const uint64_t it = (uint64_t)task * MCQ_TASK_ITERS + j * MCQ_WAVE + lane;
                if (it < qc.runs) {
                    McqReplayDraws dr = {dbase + it, stride};
                    mcq_iteration(qc, dr, base, tab.tf, tab.tops, tab.sd, tab.inv, acc);
                }
            }
            acc.passes = 0; /* counted by the host while parsing the MT19937 stream */
        }

        /* tallies: lanes -> wave (shuffles) -> one 64-bit atomic per counter per task */
        uint32_t wins = 0;
        uint64_t mine = 0;
#pragma unroll
        for (uint32_t code = 0; code < MCQ_N_CODES; code++) {
            if (code == 5) continue;
            uint32_t v = wave_sum((uint32_t)(acc.types >> (6 * code)) & 63u);
            wins += v;
            if (lane == 3u + mcq_code_to_type(code)) mine = v;
        }
        const uint32_t ties = wave_sum(acc.tie);
        const uint32_t passes = wave_sum(acc.passes);
        if (lane == 0) mine = passes;
        if (lane == 1) mine = wins - ties;
        if (lane == 2) mine = ties;
        if (lane < 12 && mine != 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(res + qi) + 1 + lane, (unsigned long long)mine);
    }
}

// ---------------------------------------------------------------------------------------------- showdown
__global__ __launch_bounds__(256) void mcq_showdown_kernel(const uint8_t *__restrict__ hands, uint32_t n_tables,
                                                           uint32_t n_players, const McqTables *__restrict__ g_tab,
                                                           uint8_t *__restrict__ winner, uint8_t *__restrict__ wtype,
                                                           uint32_t *__restrict__ keys) {
    __shared__ __attribute__((aligned(16))) LdsTables tab;
    load_tables(tab, g_tab);
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_tables; t += gridDim.x * blockDim.x) {
        uint32_t best = 0, w = 0;
        for (uint32_t p = 0; p < n_players; p++) {
            const uint8_t *h = hands + ((size_t)t * n_players + p) * 7;
            McqBoard b;
            b.clear();
#pragma unroll
            for (int k = 2; k < 7; k++) b.add(mcq_card(h[k] < 52 ? h[k] : 0));
            McqHole hole;
            hole.set(mcq_card(h[0] < 52 ? h[0] : 0), mcq_card(h[1] < 52 ? h[1] : 0));
            McqFlushSel fs;
            fs.from_board(b);
            const uint32_t key = mcq_eval_key(b, fs, hole, tab.tf, tab.tops, tab.sd);
            if (keys) keys[(size_t)t * n_players + p] = key;
            if (key > best) { best = key; w = p; } /* strict: the first of equal hands stays (hand_evaluator.py:23) */
        }
        winner[t] = (uint8_t)w;
        wtype[t] = (uint8_t)mcq_key_type(best);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------- launchers
hipError_t mcq_launch_prep(const mcq_query *d_q, uint32_t n, mcq_result *d_res, uint32_t *d_prefix, hipStream_t s) {
    hipLaunchKernelGGL(mcq_prep_kernel, dim3(1), dim3(1024), 0, s, d_q, n, d_res, d_prefix);
    return hipGetLastError();
}

hipError_t mcq_launch_eval(int mode, const mcq_query *d_q, uint32_t n, const uint32_t *d_prefix, mcq_result *d_res,
                           uint64_t seed, uint64_t first_qid, const McqTables *d_luts, const uint8_t *d_draws,
                           const uint64_t *d_draw_off, uint32_t grid, uint32_t block, hipStream_t s) {
    if (mode == MCQ_MODE_PHILOX)
        hipLaunchKernelGGL(mcq_eval_kernel<MCQ_MODE_PHILOX>, dim3(grid), dim3(block), 0, s, d_q, n, d_prefix, d_res,
                           seed, first_qid, d_luts, d_draws, d_draw_off);
    else
        hipLaunchKernelGGL(mcq_eval_kernel<MCQ_MODE_REPLAY_MT19937>, dim3(grid), dim3(block), 0, s, d_q, n, d_prefix,
                           d_res, seed, first_qid, d_luts, d_draws, d_draw_off);
    return hipGetLastError();
}

hipError_t mcq_launch_showdown(const uint8_t *d_hands, uint32_t n_tables, uint32_t n_players, const McqTables *d_luts,
                               uint8_t *d_winner, uint8_t *d_wtype, uint32_t *d_keys, hipStream_t s) {
    uint32_t grid = (n_tables + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(mcq_showdown_kernel, dim3(grid), dim3(256), 0, s, d_hands, n_tables, n_players, d_luts,
                       d_winner, d_wtype, d_keys);
    return hipGetLastError();
}

hipError_t mcq_eval_occupancy(int mode, int block, int *blocks_per_cu) {
    if (mode == MCQ_MODE_PHILOX)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, mcq_eval_kernel<MCQ_MODE_PHILOX>, block, 0);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, mcq_eval_kernel<MCQ_MODE_REPLAY_MT19937>, block, 0);
}
