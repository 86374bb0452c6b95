// mcq_kernels.hip -- gfx950 kernels of the equity engine.
//
// Work decomposition: a WAVE TASK is 1024 iterations of one query; the 64 lanes of the
// wave each run 16 of them (production mode: lane = one RNG stream of 16 consecutive iterations; parity
// mode: iterations interleaved so that the draw bytes of a wave are contiguous).  All lanes of a wave work
// on the same query, so deck length, loop bounds and the query record are wave-uniform (SGPRs, scalar
// branches) and no lane diverges.  The task list, weighted by an instruction estimate per query, is cut into
// one contiguous slice per resident wave: a wave keeps its tallies in registers while the query stays the
// same and issues one set of atomics per (wave, query) -- HBM traffic stays at the algorithmic 120 B/query
// plus a few KB of table image per block.
//
// Memory: 16 B in / 104 B out per QUERY; per iteration nothing touches HBM in production mode (parity mode
// reads <= 23 draw bytes).  LDS holds the lookup tables (97 KB per block; the flush table is read from global memory) and one 64-entry base deck per wave.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "mcq_device.hpp"
#include "mcq_exact.hpp"
#include "mcq_internal.hpp"
#include "mcq_mt.hpp"
#include "mcq_mt_ext.hpp"
#include "mcq_mt_blocks.hpp"

namespace {

constexpr int kMaxBlock = 1024; /* 16 waves = 4 per SIMD; one block per CU: tables 97 KB + 16 base decks 16 KB of the 160 KB LDS */
constexpr int kExtBlock = 1024; /* extended queries: 40 KB of dealt card ids beside the tables */

// A launch with or without the pair of timestamp events (mcq_set_kernel_timing): the timestamped form costs a
// one-launch query about 6 us of its call time on this pool (tools/launch_floor.hip), so it is only used on request.
#define MCQ_LAUNCH_TIMED(kern, grid, block, ...)                                                  \
    do {                                                                                          \
        if (t0 || t1) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, s, t0, t1, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, s, __VA_ARGS__);                \
    } while (0)

// Sum over the 64 lanes of a whole wave (EXEC full), wave-uniform result: four DPP adds bring every 16-lane row to
// its row sum (quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8 -- plain VALU rate), four v_readlane add the rows.
// The shuffle form below goes through the LDS crossbar once per step (ds_bpermute): ~1 us for a row of twelve sums.
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct LdsTablesEval { /* per block: what the kernels keep in LDS; tf stays in global memory (see McqTables) */
    uint32_t tops[8192];
    uint32_t sd[16384];
    uint32_t sel8[256];
};
static_assert(__builtin_offsetof(LdsTablesEval, sel8) == MCQ_TF_BYTE_OFFSET, "first 96 KB of McqTables");

__device__ __forceinline__ void load_tables(LdsTablesEval &dst, const McqTables *__restrict__ g) {
    const uint4 *src = reinterpret_cast<const uint4 *>(g);
    uint4 *d = reinterpret_cast<uint4 *>(&dst);
    constexpr uint32_t kVec = MCQ_TF_BYTE_OFFSET / 16; /* tops, sd | kc */
    uint32_t i = threadIdx.x;
    for (; i + 7u * blockDim.x < kVec; i += 8u * blockDim.x) { /* eight 16-byte loads in flight per lane */
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = src[i + k * blockDim.x];
#pragma unroll
        for (int k = 0; k < 8; k++) d[i + k * blockDim.x] = v[k];
    }
    for (; i < kVec; i += blockDim.x) d[i] = src[i];
    for (uint32_t j = threadIdx.x; j < 256u; j += blockDim.x) dst.sel8[j] = g->sel8[j];
    __syncthreads();
}

// Exclusive prefix sum of one 64-bit value per thread over a 1024-thread block (two barriers): wave-level scan by
// shuffles, the 16 wave totals through LDS.  Returns the sum of the values of all lower threads; *total = block sum.
__device__ __forceinline__ uint64_t block_exclusive_scan_1024(uint64_t v, uint64_t *wave_tot /* LDS, 16 entries */,
                                                               uint64_t *total) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up((unsigned long long)inc, off, 64);
        if (lane >= (uint32_t)off) inc += o;
    }
    __syncthreads(); /* wave_tot may still be read from the previous call */
    if (lane == 63u) wave_tot[wv] = inc;
    __syncthreads();
    uint64_t before = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        const uint64_t t = wave_tot[k];
        before += k < wv ? t : 0ull;
        all += t;
    }
    *total = all;
    return before + inc - v;
}

// ---------------------------------------------------------------------------------------------- prep
// One block.  Validates every query, zeroes its result row and builds the exclusive prefix of the queries'
// scheduling cost (tasks x weight; prefix[n] = total).  Invalid queries cost nothing and get runs = 0,
// passes = UINT64_MAX.
__global__ __launch_bounds__(1024) void mcq_prep_kernel(const mcq_query *__restrict__ q, uint32_t n,
                                                        mcq_result *__restrict__ res, uint64_t *__restrict__ prefix,
                                                        uint32_t part_idx, uint32_t n_parts, uint32_t n_cu,
                                                        uint32_t split_max) {
    __shared__ uint64_t wave_tot[16];
    uint64_t carry = 0; /* every thread keeps the running total */
    __shared__ unsigned long long sum_tasks;
    __shared__ uint32_t max_tasks;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        sum_tasks = 0;
        max_tasks = 0;
    }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + tid;
        uint64_t cost = 0;
        uint32_t my_tasks = 0;
        if (i < n) {
            const uint4 raw = reinterpret_cast<const uint4 *>(q)[i];
            const McqQueryWords qq = {raw.x, raw.y, raw.z, raw.w};
            bool ok = mcq_query_valid(qq);
            /* this launch's share of the query: tasks [t_lo, t_hi) (everything unless the iterations are split
             * over devices, mcq_eval_batch_part) */
            const McqPart pt = mcq_part(mcq_task_count(qq), qq.runs(), part_idx, n_parts);
            cost = ok ? (uint64_t)(pt.t_hi - pt.t_lo) * mcq_task_weight(qq) : 0ull;
            my_tasks = ok ? pt.t_hi - pt.t_lo : 0u;
            uint64_t *r = reinterpret_cast<uint64_t *>(res + i);
            r[0] = ok ? pt.runs : 0ull;
            r[1] = ok || part_idx != 0u ? 0ull : ~0ull; /* the marker of an invalid query: from ONE share only, so that the
                                                          * sum of the shares (same-device add, all-reduce) still shows it */
#pragma unroll
            for (int k = 2; k < 13; k++) r[k] = 0;
        }
        if (n <= 1024u) { /* a possible small batch: what mcq_pick_split needs (below); one atomic per wave */
            const uint32_t ws = wave_sum(my_tasks);
            uint32_t wm = my_tasks;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const uint32_t o = __shfl_xor(wm, off, 64);
                wm = o > wm ? o : wm;
            }
            if ((tid & 63u) == 0u && ws) {
                atomicAdd(&sum_tasks, (unsigned long long)ws);
                atomicMax(&max_tasks, wm);
            }
        }
        uint64_t chunk_total;
        const uint64_t before = block_exclusive_scan_1024(cost, wave_tot, &chunk_total);
        if (i < n) prefix[i] = carry + before;
        carry += chunk_total;
    }
    __syncthreads(); /* sum_tasks / max_tasks are complete */
    if (tid == 0) {
        prefix[n] = carry;
        /* the cut for MCQ_SPLIT_FROM_PREP launches (queries resident in HBM, at most 1024 of them) */
        prefix[n + 1] = n <= 1024u && sum_tasks ? mcq_pick_split(sum_tasks, max_tasks, n_cu, split_max) : 0u;
        prefix[n + 2] = 0; /* work counter of mcq_mt_parse_kernel */
    }
}

// ---------------------------------------------------------------------------------------------- parity mode: MT19937
// TWO waves per query walk np.random.seed(seed32 + query index)'s stream (mcq_mt.hpp) and leave the accepted draws in
// the draw-major global buffer the evaluation kernel reads, `passes` in the query's result row:
//   wave 0, the PRODUCER, owns the 624-word state: it regenerates a block and tempers it into a buffer of
//           (y & 63) | 0x80 bytes -- independent work at the full issue rate, a block ahead of
//   wave 1, the PARSER, whose chain of dependent steps per batch of 64 words (mcq_mt_batch) is what bounds the walk: it
//           only reads bytes.
// One block = one pair = 128 threads: the block barrier is the pair's hand-over, once per 624 words (two buffers: the
// producer fills block b + 1 while block b is parsed).  The parser tells the producer at which barrier to stop.
// Queries are handed out through an atomic counter (their lengths differ).  7.7 KB of LDS per pair: 16 pairs per CU.
constexpr int kMtBlock = 128;
struct McqMtPairWave { /* what mcq_mt_batch touches: the position tables and the ring as in McqMtWave, the words as bytes */
    uint32_t ptab[MCQ_MT_POSITIONS];
    uint8_t ring[(MCQ_MT_MAX_DRAWS + 1u) * MCQ_MT_ROW];
    uint8_t yb[2][MCQ_MT_N + 64u]; /* + 64: a batch reads 64 bytes from its position */
    volatile uint32_t stop_at;     /* the producer leaves behind its barrier number stop_at (0 = not yet known) */
    uint32_t qi;
};
struct McqMtProducer {
    uint32_t mt[MCQ_MT_N + 64u];
};
/* the buffer being parsed: st.src, its byte offset (the first hand-over makes it buffer 0: the state starts at buffer 1) */
constexpr uint32_t kMtYbBytes = MCQ_MT_N + 64u;
__device__ __forceinline__ uint32_t mcq_mt_word_yb(const McqMtPairWave &w, const McqMtState &st, uint32_t i) {
    return (&w.yb[0][0])[st.src + i];
}
__device__ __forceinline__ void mcq_mt_emit_lane(McqMtPairWave &, bool, uint32_t, uint32_t, uint32_t, uint32_t) {}
__device__ __forceinline__ constexpr bool mcq_mt_padded(const McqMtPairWave &) { return true; } /* yb[.][624 .. 688) = 0xFF */
__device__ __forceinline__ void mcq_mt_next_block(McqMtPairWave &, McqMtState &st) {
    __syncthreads(); /* the producer has filled the other buffer; it may now overwrite the one just parsed */
    st.blocks += 1u; /* (= barriers the parser has passed) */
    st.src = kMtYbBytes - st.src;
}

__global__ __launch_bounds__(kMtBlock) void mcq_mt_parse_kernel(const mcq_query *__restrict__ queries, uint32_t n,
                                                                uint32_t seed32, uint8_t *__restrict__ draws,
                                                                const uint64_t *__restrict__ draw_off,
                                                                mcq_result *__restrict__ res, uint32_t *__restrict__ counter) {
    __shared__ __attribute__((aligned(16))) McqMtPairWave w;
    __shared__ __attribute__((aligned(16))) McqMtProducer prod;
    const uint32_t lane = threadIdx.x & 63u;
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0u;
    w.yb[threadIdx.x >> 6][MCQ_MT_N + lane] = 0xFFu; /* behind both buffers' words: accepted nowhere (mcq_mt_padded) */
    for (;;) {
        if (threadIdx.x == 0) {
            w.qi = atomicAdd(counter, 1u);
            w.stop_at = 0u;
        }
        __syncthreads();
        const uint32_t qi = __builtin_amdgcn_readfirstlane(w.qi);
        if (qi >= n) break;
        const uint4 raw = reinterpret_cast<const uint4 *>(queries)[qi];
        const McqQueryWords q = {(uint32_t)__builtin_amdgcn_readfirstlane(raw.x), (uint32_t)__builtin_amdgcn_readfirstlane(raw.y),
                                 (uint32_t)__builtin_amdgcn_readfirstlane(raw.z), (uint32_t)__builtin_amdgcn_readfirstlane(raw.w)};
        const uint32_t n_opp = q.n_players() - 1u, n_deal = 5u - q.n_board(), runs = q.runs();
        /* block-uniform: invalid queries and queries that draw nothing keep passes = 0 */
        if (mcq_query_valid(q) && 2u * n_opp + n_deal != 0u && runs != 0u) {
            if (producer) {
                mcq_mt_seed(prod, seed32 + qi);
                MCQ_WAVE_SYNC();
                for (uint32_t b = 0;; b++) {
                    mcq_mt_regenerate(prod);
                    uint8_t *dst = w.yb[b & 1u];
#pragma unroll
                    for (uint32_t k = 0; k < MCQ_MT_N + 63u; k += 64u) /* ten steps of independent lanes */
                        if (k + lane < MCQ_MT_N) dst[k + lane] = (uint8_t)((mcq_mt_temper(prod.mt[k + lane]) & 63u) | 0x80u);
                    __syncthreads(); /* barrier number b + 1: block b is there */
                    if (w.stop_at == b + 1u) break;
                }
            } else {
                McqMtState st = {MCQ_MT_N, 0u, 0u, 0u, 0ull, 0u, kMtYbBytes};
                mcq_mt_parse_query(w, st, 50u - q.n_board(), n_opp, n_deal, runs, draws + draw_off[qi],
                                   ((uint64_t)runs + 63u) & ~63ull);
                /* every lane stores the same word: a store under `lane == 0` here would be a divergent branch in front of
                 * the barrier */
                reinterpret_cast<unsigned long long *>(res + qi)[1] = st.passes;
                if (lane == 0) w.stop_at = st.blocks + 1u; /* the producer is filling one more block: it leaves behind the next barrier */
                __syncthreads();
            }
        }
        __syncthreads(); /* both waves are done with this query's LDS */
    }
}

// The same for extended queries (mcq_mt_ext.hpp: the reference's loops over ranges, ghost cards and known hands walked
// stage by stage).  6.5 KB of LDS per wave.  A query whose range cannot be dealt gets passes = UINT64_MAX.
constexpr int kMtExtBlock = 256;
__global__ __launch_bounds__(kMtExtBlock) void mcq_mt_parse_ext_kernel(const mcq_query *__restrict__ queries,
                                                                    const mcq_query_ext *__restrict__ ext, uint32_t n, uint32_t seed32,
                                                                    uint8_t *__restrict__ draws, const uint64_t *__restrict__ draw_off,
                                                                    mcq_result *__restrict__ res, uint32_t *__restrict__ counter) {
    __shared__ __attribute__((aligned(16))) McqMtExtWave ws[kMtExtBlock / 64];
    McqMtExtWave &w = ws[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(counter, 1u);
        const uint32_t qi = __builtin_amdgcn_readfirstlane(t);
        if (qi >= n) break;
        const uint4 raw = reinterpret_cast<const uint4 *>(queries)[qi];
        const McqQueryWords q = {(uint32_t)__builtin_amdgcn_readfirstlane(raw.x), (uint32_t)__builtin_amdgcn_readfirstlane(raw.y),
                                 (uint32_t)__builtin_amdgcn_readfirstlane(raw.z), (uint32_t)__builtin_amdgcn_readfirstlane(raw.w)};
        const uint32_t *ew = reinterpret_cast<const uint32_t *>(ext + qi);
        const McqExtRec er = {ew};
        if (mcq_query_ext_valid(q, er) && q.runs() != 0u) { /* wave-uniform */
            MCQ_WAVE_SYNC(); /* the previous query's reads of this wave's LDS are done */
            mcq_mt_seed(w, seed32 + qi);
            MCQ_WAVE_SYNC();
            McqMtExtState st = {MCQ_MT_N, 0u, 0u, 0u, 0u, 0u, 0ull, false};
            const bool ok = mcq_mt_parse_query_ext(w, st, q, ew, draws + draw_off[qi], ((uint64_t)q.runs() + 63u) & ~63ull);
            /* (every lane stores the same word: no divergent branch in front of the loop's back edge) */
            reinterpret_cast<unsigned long long *>(res + qi)[1] = ok ? st.passes : ~0ull;
        }
    }
}

// ---------------------------------------------------------------------------------------------- few long queries
// mcq_mt_blocks.hpp: the stream of a query parsed with its state blocks side by side -- four kernels behind one another
// on the stream.  blk_off[q] .. blk_off[q + 1]: the query's blocks in the shared arrays (none: the query draws nothing
// or is invalid); gridDim.y = queries where the grid is two-dimensional.
//
// 1. generate.  The MT19937 recurrence is the one serial thing left: x[k] of the next state needs x[k], x[k + 1] and
// x[k + 397] of this one, or -- from k = 227 on -- x[k - 227] of the NEXT: three sweeps of at most 227 words, each
// waiting for the one before.  But the word a sweep waits for is the one the SAME thread has just made (k - 227 is
// thread t's word of the sweep before), so thread t makes x[t], x[227 + t] and x[454 + t] out of registers, from seven
// words of the finished state that it reads at once -- one LDS round trip and one barrier per block instead of three
// (1.73 -> 0.6 ms for the 3 900 blocks of a 6-max 100 000-run query).  (x[623] needs the new x[0]: its thread makes that
// one a second time.)  Two state buffers; three more waves copy the state the first four are reading to HBM as it is --
// 2 496 B per block; the waves that read it temper it (mcq_mtb_load_block): off the chain, and side by side.
//
// Round 4: a query's blocks in SEGMENTS of MCQ_MTB_SEG, one work-group each, side by side.  The generator is linear over
// GF(2): the state J words ahead is a fixed XOR combination of 19 937 + 623 consecutive words, the same for every seed
// (mcq_mt_jump_table.inc: t^J mod the characteristic polynomial, made and checked against numpy by
// tools/mt_jump_table.py).  Per round of MCQ_MTB_MAX_SEG segments, from base block `base` (state number `base`: the seed
// state, or the last block of the round before): (a) this kernel, one work-group per query, makes the first 33 blocks --
// the words every jump of the round combines; (b) mcq_mtb_jump_kernel: segment g's start state, kMtbJumpSplit work-groups
// each XORing their share of the polynomial's terms over all 624 words (partial sums to HBM); (c) this kernel again, a
// work-group per (segment, query): start state = the seed state / the block in front / the XOR of the partial sums.
#define MCQ_MTB_TABLE_ATTR __device__ const
#include "mcq_mt_jump_table.inc"
constexpr uint32_t kMtbJumpSplit = 8, kMtbJumpWords = MCQ_MT_N / kMtbJumpSplit; /* 78 words of the polynomial per work-group */
constexpr uint32_t kMtbJumpSpan = MCQ_MT_N / 32u + 14u; /* blocks whose words a round's jumps read: 19 937 + 623 words = 33 */
static_assert(kMtbJumpSplit * kMtbJumpWords == MCQ_MT_N && kMtbJumpWords % 2u == 0u, "the polynomial's words split evenly, in pairs");
static_assert(kMtbJumpSpan * MCQ_MT_N >= 32u * MCQ_MT_N + MCQ_MT_N - 1u && kMtbJumpSpan == 33u && kMtbJumpSpan <= MCQ_MTB_SEG, "words 0 .. 20 590");
constexpr int kMtbGenBlock = 512;
/* part: [query][segment - 1][kMtbJumpSplit][624] partial sums of the jumps (mcq_mtb_jump_kernel).  limit: blocks a
 * work-group makes at most (kMtbJumpSpan for step (a), MCQ_MTB_SEG for step (c)) */
__global__ __launch_bounds__(kMtbGenBlock) void mcq_mtb_generate_kernel(const uint32_t *__restrict__ blk_off, uint32_t seed32,
                                                                       uint32_t *__restrict__ raw, const uint32_t *__restrict__ part,
                                                                       uint32_t base, uint32_t limit, uint32_t for_jumps) {
    __shared__ __attribute__((aligned(16))) uint32_t mt[2][MCQ_MT_N + 8u];
    const uint32_t qi = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, first = blk_off[qi], nb_q = blk_off[qi + 1u] - first;
    const uint32_t start = base + g * MCQ_MTB_SEG;
    if (start >= nb_q) return; /* (block-uniform) */
    if (for_jumps && nb_q - base <= MCQ_MTB_SEG) return; /* step (a) of a query with one segment: no jump reads it */
    const uint32_t nb = nb_q - start < limit ? nb_q - start : limit;
    if (start == 0u) {
        if (tid == 0) { /* np.random.seed: init_genrand, a serial recurrence */
            uint32_t x = seed32 + qi;
            mt[0][0] = x;
            for (uint32_t i = 1; i < MCQ_MT_N; i++) {
                x = 1812433253u * (x ^ (x >> 30)) + i;
                mt[0][i] = x;
            }
        }
    } else if (g == 0u) { /* a later round: the block in front */
        const uint32_t *src = raw + (uint64_t)(first + start - 1u) * MCQ_MT_N;
        for (uint32_t i = tid; i < MCQ_MT_N; i += kMtbGenBlock) mt[0][i] = src[i];
    } else { /* the jump's partial sums */
        const uint32_t *src = part + ((uint64_t)qi * (MCQ_MTB_MAX_SEG - 1u) + (g - 1u)) * kMtbJumpSplit * MCQ_MT_N;
        for (uint32_t i = tid; i < MCQ_MT_N; i += kMtbGenBlock) {
            uint32_t x = 0;
#pragma unroll
            for (uint32_t w = 0; w < kMtbJumpSplit; w++) x ^= src[w * MCQ_MT_N + i];
            mt[0][i] = x;
        }
    }
    __syncthreads();
    uint32_t *dst = raw + (uint64_t)(first + start) * MCQ_MT_N;
    /* two loops with the same barriers: the waves that make the state, the waves that copy it out */
    if (tid < 256u) { /* (wave-uniform) */
        /* every read up front, none behind a condition on the thread (indices clamped instead): ONE round trip to LDS per
         * block.  Threads 227 .. 255 repeat thread 226's work and store nothing. */
        const uint32_t t = tid < 226u ? tid : 226u, tc = t < 169u ? t : 169u;
        const bool w_ab = tid < 227u, w_c = tid < 170u, is_last = tid == 169u;
        for (uint32_t b = 0; b <= nb; b++) { /* state b -> state b + 1 */
            const uint32_t *S = mt[b & 1u];
            uint32_t *N = mt[(b + 1u) & 1u];
            if (b < nb) {
                const uint32_t a0 = S[t], a1 = S[t + 1u], f = S[t + MCQ_MT_M], b0 = S[227u + t], b1 = S[228u + t];
                const uint32_t c0 = S[454u + tc], c1o = S[455u + tc] /* (tc = 169: the padding behind the state) */;
                /* (opaque, or the compiler moves these three reads into thread 169's branch, behind the wait for the others) */
                uint32_t z0 = S[0], z1 = S[1], zf = S[MCQ_MT_M];
                asm volatile("" : "+v"(z0), "+v"(z1), "+v"(zf));
                const uint32_t c1 = is_last ? zf ^ mcq_mt_twist(z0, z1) /* the new x[0] */ : c1o;
                const uint32_t nA = f ^ mcq_mt_twist(a0, a1);
                const uint32_t nB = nA ^ mcq_mt_twist(b0, b1);
                if (w_ab) {
                    N[t] = nA;
                    N[227u + t] = nB;
                }
                if (w_c) N[454u + t] = nB ^ mcq_mt_twist(c0, c1);
            }
            __syncthreads(); /* state b + 1 is complete; state b has been read for the last time */
        }
    } else {
        const uint32_t t = tid - 256u;
        for (uint32_t b = 0; b <= nb; b++) { /* the segment's block b - 1 = state b */
            if (b >= 1u && t < MCQ_MT_N / 4u)
                reinterpret_cast<uint4 *>(dst + (uint64_t)(b - 1u) * MCQ_MT_N)[t] = *reinterpret_cast<const uint4 *>(mt[b & 1u] + 4u * t);
            __syncthreads();
        }
    }
}

// 1b. jump: work-group (segment g >= 1, share w) of a query: out[j] = XOR over the set coefficients i of its 78 words of
// the polynomial of y[i + j], j < 624 -- y = the round's first 33 blocks.  The floor is LDS bandwidth (6.2 M word reads per
// jump), so the reads are 16 bytes wide: a lane owns j = 4 l .. 4 l + 3 (+ 256, + 512) and reads y[i + 4 l ..] with ONE
// ds_read_b128 -- aligned whatever i is, because the share's window of y lies in LDS four times, shifted by 0 .. 3 words.
// The set coefficients are laid out once per work-group as a list of byte offsets (copy + aligned position) in LDS, so a
// term costs no scalar bit scan: a broadcast read serves eight terms, each term one address add, three wide reads and
// twelve XORs.  Sixteen waves take the list's terms in turn and add their sums up through LDS.
constexpr int kMtbJumpBlock = 1024;
constexpr uint32_t kMtbJumpTerms = kMtbJumpWords * 32u;           /* coefficients of a share: 2496 */
constexpr uint32_t kMtbJumpCopy = kMtbJumpTerms + 768u;           /* words of one copy of the window: terms + the lanes' offsets */
__global__ __launch_bounds__(kMtbJumpBlock) void mcq_mtb_jump_kernel(const uint32_t *__restrict__ blk_off,
                                                                    const uint32_t *__restrict__ raw, uint32_t *__restrict__ part,
                                                                    uint32_t base) {
    __shared__ __attribute__((aligned(16))) uint32_t s_y[4][kMtbJumpCopy]; /* s_y[c][k] = y[i_lo + k + c]; later: the waves' sums */
    __shared__ __attribute__((aligned(16))) uint16_t s_list[kMtbJumpTerms + 8u * (kMtbJumpBlock / 64)];
    __shared__ uint32_t s_cnt[kMtbJumpWords + 1u], s_pop[kMtbJumpWords];
    static_assert(sizeof(s_y) >= (kMtbJumpBlock / 64) * 768u * 4u, "the sums of sixteen waves fit where the window was");
    static_assert(4u * kMtbJumpCopy * 4u < 65536u, "byte offsets as 16-bit words");
    const uint32_t qi = blockIdx.y, g = blockIdx.x / kMtbJumpSplit + 1u, w = blockIdx.x % kMtbJumpSplit, tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t first = blk_off[qi], nb_q = blk_off[qi + 1u] - first;
    if (base + g * MCQ_MTB_SEG >= nb_q) return; /* (block-uniform) */
    const uint32_t i_lo = w * kMtbJumpTerms;
    const uint32_t *src = raw + (uint64_t)(first + base) * MCQ_MT_N;
    constexpr uint32_t kSpanWords = kMtbJumpSpan * MCQ_MT_N;
    for (uint32_t k = tid; k < kMtbJumpCopy + 3u; k += kMtbJumpBlock) {
        const uint32_t v = i_lo + k < kSpanWords ? src[i_lo + k] : 0u;
#pragma unroll
        for (uint32_t c = 0; c < 4u; c++)
            if (k >= c && k - c < kMtbJumpCopy) s_y[c][k - c] = v;
    }
    const uint32_t *gw = kMtJump[g - 1u] + w * kMtbJumpWords;
    if (tid < kMtbJumpWords) s_pop[tid] = (uint32_t)__builtin_popcount(gw[tid]);
    __syncthreads();
    if (tid <= kMtbJumpWords) { /* terms in front of word tid: independent reads, no chain */
        uint32_t x = 0;
#pragma unroll
        for (uint32_t k = 0; k < kMtbJumpWords; k++) x += k < tid ? s_pop[k] : 0u;
        s_cnt[tid] = x;
    }
    __syncthreads();
    const uint32_t n_terms = s_cnt[kMtbJumpWords];
    for (uint32_t t = tid; t < kMtbJumpTerms; t += kMtbJumpBlock) {
        const uint32_t word = gw[t >> 5], bit = t & 31u;
        if ((word >> bit) & 1u) {
            const uint32_t c = t & 3u;
            s_list[s_cnt[t >> 5] + (uint32_t)__builtin_popcount(word & ((1u << bit) - 1u))] = (uint16_t)((c * kMtbJumpCopy + (t - c)) * 4u);
        }
    }
    /* padding: a wave reads its terms eight at a time; the terms behind the last point at ... a term twice = no term */
    if (tid < 8u * (kMtbJumpBlock / 64)) s_list[n_terms + tid] = 0xFFFFu;
    __syncthreads();
    uint4 acc[3] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    const char *ybase = reinterpret_cast<const char *>(&s_y[0][0]) + 16u * lane;
    /* wave wv takes the terms 8 (16 k + wv) .. + 8 */
    for (uint32_t t0 = 8u * wv; t0 < n_terms; t0 += 8u * (kMtbJumpBlock / 64)) {
        const uint4 pkv = *reinterpret_cast<const uint4 *>(s_list + t0); /* (a broadcast read; into scalar registers) */
        const uint32_t pk[4] = {(uint32_t)__builtin_amdgcn_readfirstlane(pkv.x), (uint32_t)__builtin_amdgcn_readfirstlane(pkv.y),
                                (uint32_t)__builtin_amdgcn_readfirstlane(pkv.z), (uint32_t)__builtin_amdgcn_readfirstlane(pkv.w)};
        const uint32_t off[8] = {pk[0] & 0xFFFFu, pk[0] >> 16, pk[1] & 0xFFFFu, pk[1] >> 16, pk[2] & 0xFFFFu, pk[2] >> 16, pk[3] & 0xFFFFu, pk[3] >> 16};
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            if (off[u] == 0xFFFFu) break; /* (wave-uniform: behind the list's end) */
            const char *p = ybase + off[u];
#pragma unroll
            for (uint32_t h = 0; h < 3u; h++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(p + 1024u * h);
                acc[h].x ^= v.x;
                acc[h].y ^= v.y;
                acc[h].z ^= v.z;
                acc[h].w ^= v.w;
            }
        }
    }
    __syncthreads(); /* the window has been read for the last time */
    uint32_t *sums = &s_y[0][0]; /* [wave][768] */
#pragma unroll
    for (uint32_t h = 0; h < 3u; h++) *reinterpret_cast<uint4 *>(sums + wv * 768u + 256u * h + 4u * lane) = acc[h];
    __syncthreads();
    if (tid < MCQ_MT_N) {
        uint32_t x = 0;
#pragma unroll
        for (uint32_t k = 0; k < kMtbJumpBlock / 64; k++) x ^= sums[k * 768u + tid];
        part[(((uint64_t)qi * (MCQ_MTB_MAX_SEG - 1u) + (g - 1u)) * kMtbJumpSplit + w) * MCQ_MT_N + tid] = x;
    }
}

// 2. scan: one wave per (query, block); lane = entry state (mcq_mtb_automaton), the block's bytes through LDS.
constexpr int kMtbBlock = 256;
constexpr int kMtbParseBlock = 1024; /* parse: sixteen blocks per work-group (70 KB of LDS, two work-groups per CU), their attempts in one atomic */
/* a block's state words -> its bytes (y & 63) | 0x80 in LDS (dst: MCQ_MT_N + 64 bytes, the last 64 padding) */
__device__ __forceinline__ void mcq_mtb_load_block(const uint32_t *__restrict__ src, uint8_t *dst, uint32_t lane) {
    uint32_t y[10];
#pragma unroll
    for (uint32_t k = 0; k < 10u; k++) y[k] = src[64u * k + lane < MCQ_MT_N ? 64u * k + lane : MCQ_MT_N - 1u];
#pragma unroll
    for (uint32_t k = 0; k < 10u; k++)
        if (64u * k + lane < MCQ_MT_N) dst[64u * k + lane] = (uint8_t)((mcq_mt_temper(y[k]) & 63u) | 0x80u);
    dst[MCQ_MT_N + lane] = 0xFFu; /* (mcq_mt_padded) */
    MCQ_WAVE_SYNC();
}
__device__ __forceinline__ McqMtbPlan mcq_mtb_plan_of(const mcq_query *__restrict__ queries, uint32_t qi) {
    const uint4 raw = reinterpret_cast<const uint4 *>(queries)[qi];
    const McqQueryWords q = {(uint32_t)__builtin_amdgcn_readfirstlane(raw.x), (uint32_t)__builtin_amdgcn_readfirstlane(raw.y),
                             (uint32_t)__builtin_amdgcn_readfirstlane(raw.z), (uint32_t)__builtin_amdgcn_readfirstlane(raw.w)};
    return mcq_mtb_plan(50u - q.n_board(), q.n_players() - 1u, 5u - q.n_board(), q.runs());
}
__global__ __launch_bounds__(kMtbBlock) void mcq_mtb_scan_kernel(const mcq_query *__restrict__ queries,
                                                                const uint32_t *__restrict__ blk_off,
                                                                const uint32_t *__restrict__ raw, uint32_t *__restrict__ exits) {
    /* A block has n_st = D + n_opp <= 32 entry states, a wave has 64 lanes, and the automaton is a chain: a wave that is
     * alone or almost alone on its SIMD issues one vector instruction per 5-8 cycles whatever their dependence.  So a wave
     * takes ONE block and cuts it into U = 2 .. 8 PARTS (as many as 64 / n_st allows: eight for a heads-up query, three
     * six-max, two with nine or ten players), lane = (part, entry state): every part is walked from every entry state side
     * by side, and the block's exit words are composed from the parts' as a group's are from its blocks'
     * (mcq_mtb_compose_step) -- twice the waves of the version before, a chain of 624 / U words instead of 624.
     * (Round 4.  Measured on the 3 900 blocks of a 6-max 100 000-run query: one block per wave, 32 lanes idle, 98 us; two
     * blocks per wave 75 us; a branch-free step with both candidate table words fetched ahead 78 us; two or four blocks per
     * wave-half word by word in turn 106-125 us; two halves of one block 58 us; this: see profiles/r08_mtb_kernels.txt.) */
    __shared__ __attribute__((aligned(16))) uint8_t s_yb[kMtbBlock / 64][MCQ_MT_N + 64u];
    __shared__ uint32_t s_pos[kMtbBlock / 64][MCQ_MTB_POS];
    __shared__ uint32_t s_ex[kMtbBlock / 64][8][MCQ_MTB_LANES];
    static_assert(MCQ_MT_N % 8u == 0u && MCQ_MT_N % 6u == 0u, "parts of 78, 104, 156, 208, 312 words");
    const uint32_t qi = blockIdx.y, lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t first = blk_off[qi], nb = blk_off[qi + 1u] - first, b = blockIdx.x * (kMtbBlock / 64) + wv;
    if (b >= nb) return; /* (wave-uniform; no block barrier below) */
    const McqMtbPlan pl = mcq_mtb_plan_of(queries, qi);
    const uint32_t n_st = pl.D + (pl.two_opp >> 1), U = mcq_mtb_parts(pl), W = MCQ_MT_N / U; /* (wave-uniform) U * n_st <= 64 */
    if (lane < MCQ_MTB_POS) s_pos[wv][lane] = mcq_mtb_pos_word(pl, lane < pl.D ? lane : 0u);
    mcq_mtb_load_block(raw + (uint64_t)(first + b) * MCQ_MT_N, s_yb[wv], lane); /* (ends with a wave barrier) */
    const uint32_t u = (lane * ((65536u + n_st - 1u) / n_st)) >> 16, e = lane - u * n_st; /* lane / n_st, lane % n_st (lane < 64) */
    if (u < U) s_ex[wv][u][e] = mcq_mtb_automaton_n(s_yb[wv] + u * W, W, s_pos[wv], pl, e);
    MCQ_WAVE_SYNC();
    if (lane < MCQ_MTB_LANES)
        exits[(uint64_t)(first + b) * MCQ_MTB_LANES + lane] = mcq_mtb_exit_from_parts(&s_ex[wv][0][0], U, pl, lane);
}

// 3. stitch, in two levels (mcq_mt_blocks.hpp).  grp_off[q] .. grp_off[q + 1]: the query's groups of MCQ_MTB_GROUP blocks.
// 3a. compose: one wave per (query, group); the group's exit words through LDS, lane = entry state of the group.
__global__ __launch_bounds__(kMtbBlock) void mcq_mtb_compose_kernel(const mcq_query *__restrict__ queries,
                                                                   const uint32_t *__restrict__ blk_off,
                                                                   const uint32_t *__restrict__ grp_off,
                                                                   const uint32_t *__restrict__ exits,
                                                                   uint32_t *__restrict__ gword, uint32_t *__restrict__ gits) {
    __shared__ __attribute__((aligned(16))) uint32_t s_ex[kMtbBlock / 64][MCQ_MTB_GROUP * MCQ_MTB_LANES];
    const uint32_t qi = blockIdx.y, lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t first = blk_off[qi], nb = blk_off[qi + 1u] - first, g = blockIdx.x * (kMtbBlock / 64) + wv;
    if (g * MCQ_MTB_GROUP >= nb) return; /* (wave-uniform; no block barrier below) */
    const McqMtbPlan pl = mcq_mtb_plan_of(queries, qi);
    const uint32_t b0 = g * MCQ_MTB_GROUP, cnt = nb - b0 < MCQ_MTB_GROUP ? nb - b0 : MCQ_MTB_GROUP;
    const uint4 *src = reinterpret_cast<const uint4 *>(exits + (uint64_t)(first + b0) * MCQ_MTB_LANES);
    for (uint32_t k = lane; k < cnt * (MCQ_MTB_LANES / 4u); k += 64u) reinterpret_cast<uint4 *>(s_ex[wv])[k] = src[k];
    MCQ_WAVE_SYNC();
    McqMtbWalk s = mcq_mtb_walk_from(pl, lane);
    if (lane < pl.D + (pl.two_opp >> 1))
        for (uint32_t k = 0; k < cnt; k++) mcq_mtb_compose_step(s_ex[wv] + k * MCQ_MTB_LANES, pl, s);
    if (lane < MCQ_MTB_LANES) {
        gword[(uint64_t)(grp_off[qi] + g) * MCQ_MTB_LANES + lane] = mcq_mtb_walk_word(s);
        gits[(uint64_t)(grp_off[qi] + g) * MCQ_MTB_LANES + lane] = s.its;
    }
}

// 3b. one wave per query follows the GROUPS' exits, 128 groups' words in LDS at a time; every lane carries the same state
// (the LDS reads are broadcasts).  gentry[group] = the walk in front of the group.  ovf[q] = 1: the stream has not ended
// within the query's blocks -- its row gets passes = UINT64_MAX and the host falls back to the serial walk.
constexpr uint32_t kMtbChunk = 128;
__global__ __launch_bounds__(64) void mcq_mtb_stitch_kernel(const mcq_query *__restrict__ queries,
                                                            const uint32_t *__restrict__ blk_off,
                                                            const uint32_t *__restrict__ grp_off,
                                                            const uint32_t *__restrict__ gword, const uint32_t *__restrict__ gits,
                                                            McqMtbEntry *__restrict__ gentry, uint32_t *__restrict__ ovf,
                                                            mcq_result *__restrict__ res) {
    __shared__ __attribute__((aligned(16))) uint32_t s_w[kMtbChunk * MCQ_MTB_LANES], s_i[kMtbChunk * MCQ_MTB_LANES];
    __shared__ McqMtbEntry s_en[kMtbChunk];
    const uint32_t qi = blockIdx.x, lane = threadIdx.x;
    const uint32_t gfirst = grp_off[qi], ng = grp_off[qi + 1u] - gfirst;
    if (blk_off[qi + 1u] == blk_off[qi]) {
        if (lane == 0) ovf[qi] = 0u;
        return;
    }
    const McqMtbPlan pl = mcq_mtb_plan_of(queries, qi);
    uint32_t d = 0, pend = 0, it = 0;
    for (uint32_t g0 = 0; g0 < ng; g0 += kMtbChunk) {
        const uint32_t cnt = ng - g0 < kMtbChunk ? ng - g0 : kMtbChunk;
        const uint4 *sw = reinterpret_cast<const uint4 *>(gword + (uint64_t)(gfirst + g0) * MCQ_MTB_LANES);
        const uint4 *si = reinterpret_cast<const uint4 *>(gits + (uint64_t)(gfirst + g0) * MCQ_MTB_LANES);
        for (uint32_t k = lane; k < cnt * (MCQ_MTB_LANES / 4u); k += 64u) {
            reinterpret_cast<uint4 *>(s_w)[k] = sw[k];
            reinterpret_cast<uint4 *>(s_i)[k] = si[k];
        }
        MCQ_WAVE_SYNC();
        for (uint32_t k = 0; k < cnt; k++) {
            /* (every lane stores the same words: no branch on the lane number inside the chain) */
            s_en[k].it0 = it;
            s_en[k].dp = d | (pend << 8) | (it < pl.runs ? 0x80000000u : 0u);
            mcq_mtb_stitch_group(s_w + k * MCQ_MTB_LANES, s_i + k * MCQ_MTB_LANES, pl, d, pend, it);
        }
        MCQ_WAVE_SYNC();
        for (uint32_t k = lane; k < cnt; k += 64u) gentry[gfirst + g0 + k] = s_en[k];
        MCQ_WAVE_SYNC();
    }
    if (lane == 0) {
        ovf[qi] = it < pl.runs ? 1u : 0u;
        if (it < pl.runs) reinterpret_cast<unsigned long long *>(res + qi)[1] = ~0ull;
    }
}

// 3c. expand: one wave per (query, group) follows the group's blocks from the group's entry and notes theirs.
__global__ __launch_bounds__(kMtbBlock) void mcq_mtb_expand_kernel(const mcq_query *__restrict__ queries,
                                                                  const uint32_t *__restrict__ blk_off,
                                                                  const uint32_t *__restrict__ grp_off,
                                                                  const uint32_t *__restrict__ exits,
                                                                  const McqMtbEntry *__restrict__ gentry,
                                                                  McqMtbEntry *__restrict__ entries) {
    __shared__ __attribute__((aligned(16))) uint32_t s_ex[kMtbBlock / 64][MCQ_MTB_GROUP * MCQ_MTB_LANES];
    __shared__ McqMtbEntry s_en[kMtbBlock / 64][MCQ_MTB_GROUP];
    const uint32_t qi = blockIdx.y, lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t first = blk_off[qi], nb = blk_off[qi + 1u] - first, g = blockIdx.x * (kMtbBlock / 64) + wv;
    if (g * MCQ_MTB_GROUP >= nb) return; /* (wave-uniform; no block barrier below) */
    const McqMtbPlan pl = mcq_mtb_plan_of(queries, qi);
    const uint32_t b0 = g * MCQ_MTB_GROUP, cnt = nb - b0 < MCQ_MTB_GROUP ? nb - b0 : MCQ_MTB_GROUP;
    const uint4 *src = reinterpret_cast<const uint4 *>(exits + (uint64_t)(first + b0) * MCQ_MTB_LANES);
    for (uint32_t k = lane; k < cnt * (MCQ_MTB_LANES / 4u); k += 64u) reinterpret_cast<uint4 *>(s_ex[wv])[k] = src[k];
    const McqMtbEntry ge = gentry[grp_off[qi] + g];
    uint32_t d = ge.dp & 0xFFu, pend = (ge.dp >> 8) & 0x7Fu, it = ge.it0;
    MCQ_WAVE_SYNC();
    for (uint32_t k = 0; k < cnt; k++) {
        s_en[wv][k].it0 = it;
        s_en[wv][k].dp = d | (pend << 8) | (it < pl.runs ? 0x80000000u : 0u);
        mcq_mtb_stitch_step(s_ex[wv] + k * MCQ_MTB_LANES, pl, d, pend, it);
    }
    MCQ_WAVE_SYNC();
    if (lane < cnt) entries[first + b0 + lane] = s_en[wv][lane];
}

// 4. parse: one wave per (query, block) from the block's true entry (mcq_mtb_parse_block: the batch code of the serial
// walk), draws straight to the draw buffer, the attempts added to the row's `passes`.
__global__ __launch_bounds__(kMtbParseBlock) void mcq_mtb_parse_kernel(const mcq_query *__restrict__ queries,
                                                                 const uint32_t *__restrict__ blk_off,
                                                                 const uint32_t *__restrict__ state,
                                                                 const McqMtbEntry *__restrict__ entries,
                                                                 const uint32_t *__restrict__ ovf, uint8_t *__restrict__ draws,
                                                                 const uint64_t *__restrict__ draw_off, mcq_result *__restrict__ res) {
    __shared__ __attribute__((aligned(16))) McqMtBlockWave ws[kMtbParseBlock / 64];
    __shared__ unsigned long long s_passes[kMtbParseBlock / 64];
    const uint32_t qi = blockIdx.y, lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t first = blk_off[qi], nb = blk_off[qi + 1u] - first, b = blockIdx.x * (kMtbParseBlock / 64) + wv;
    if (blockIdx.x * (kMtbParseBlock / 64) >= nb) return; /* (block-uniform) */
    const bool mine = b < nb;                                         /* (wave-uniform) */
    if (ovf[qi] != 0u) { /* (block-uniform) */
        /* the stream ran past the query's blocks: the host repeats the call with the serial walk; the evaluation kernel
         * behind this one still reads the query's draws -- give it valid ones (index 0), not whatever the buffer held */
        if (!mine) return;
        const uint4 raw = reinterpret_cast<const uint4 *>(queries)[qi];
        const McqQueryWords q = {raw.x, raw.y, raw.z, raw.w};
        const uint64_t bytes = (((uint64_t)q.runs() + 63u) & ~63ull) * (2u * (q.n_players() - 1u) + 5u - q.n_board());
        const uint64_t per = ((bytes + nb - 1u) / nb + 15u) & ~15ull, lo = (uint64_t)b * per, hi = lo + per < bytes ? lo + per : bytes;
        uint4 *dst = reinterpret_cast<uint4 *>(draws + draw_off[qi]);
        for (uint64_t k = lo / 16u + lane; k < hi / 16u; k += 64u) dst[k] = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
        return;
    }
    uint64_t passes = 0;
    if (mine) {
        /* (into scalar registers by hand: the compiler knows that a load at a uniform address is uniform, drops a
         * readfirstlane of it -- and then cannot pin the walk's state in SGPRs where mcq_mt_batch asks for that) */
        const McqMtbEntry en0 = entries[first + b];
        McqMtbEntry en;
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(en.it0) : "v"(en0.it0));
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(en.dp) : "v"(en0.dp));
        if (en.dp >> 31) { /* (else: the stream ended in an earlier block) */
            McqMtBlockWave &w = ws[wv];
            const uint4 raw = reinterpret_cast<const uint4 *>(queries)[qi];
            const McqQueryWords q = {(uint32_t)__builtin_amdgcn_readfirstlane(raw.x), (uint32_t)__builtin_amdgcn_readfirstlane(raw.y),
                                     (uint32_t)__builtin_amdgcn_readfirstlane(raw.z), (uint32_t)__builtin_amdgcn_readfirstlane(raw.w)};
            const uint32_t L0 = 50u - q.n_board(), n_opp = q.n_players() - 1u, n_deal = 5u - q.n_board(), runs = q.runs();
            mcq_mtb_load_block(state + (uint64_t)(first + b) * MCQ_MT_N, w.yb, lane);
            if (lane == 0) {
                w.draws = draws + draw_off[qi];
                w.stride = ((uint64_t)runs + 63u) & ~63ull;
                w.two_opp = 2u * n_opp;
            }
            mcq_mt_fill_ptab(w, L0, n_opp, 2u * n_opp + n_deal); /* (ends with a wave barrier) */
            passes = mcq_mtb_parse_block(w, L0, n_opp, n_deal, runs, en);
        }
    }
    /* the attempts of the work-group's blocks in ONE atomic: thousands of them on one row's word serialise (14 ns each:
     * 3 900 blocks of a 6-max 100 000-run query 54 us, the whole parse kernel's time) */
    if (lane == 0) s_passes[wv] = passes;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < kMtbParseBlock / 64; k++) sum += s_passes[k];
        if (sum) atomicAdd(reinterpret_cast<unsigned long long *>(res + qi) + 1, sum);
    }
}

// ---------------------------------------------------------------------------------------------- multi-GPU helper
// dst[i] += src[i]: adds the tally matrix of a second shard on the same device (mcq_multi.cpp) -- HBM-bound
// streaming, two 16-byte loads and one store per lane.
__global__ __launch_bounds__(256) void mcq_add_u64_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src,
                                                          uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n2 = n >> 1;
    ulonglong2 *d2 = reinterpret_cast<ulonglong2 *>(dst);
    const ulonglong2 *s2 = reinterpret_cast<const ulonglong2 *>(src);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        ulonglong2 a = d2[i];
        const ulonglong2 b = s2[i];
        a.x += b.x;
        a.y += b.y;
        d2[i] = a;
    }
    if ((n & 1ull) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] += src[n - 1];
}

// ---------------------------------------------------------------------------------------------- publish
// Behind the evaluation kernel of a host-buffer call with few rows: moves the finished rows from HBM into pinned
// host memory, leaves the HBM rows zero for the next call and raises a flag the host is polling -- instead of a
// D2H copy, a stream synchronisation (24 us on this pool against 7.5 us for a flag, tools/launch_floor.hip) and a
// memset.  16-byte words; the last block to finish (device counter, reset for the next launch) writes the flag.
__global__ __launch_bounds__(256) void mcq_publish_kernel(ulonglong2 *__restrict__ d_rows, ulonglong2 *__restrict__ h_rows,
                                                          uint64_t n_words16, uint32_t *__restrict__ done,
                                                          volatile uint32_t *done_flag, uint32_t ticket) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const ulonglong2 zero = {0ull, 0ull};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words16; i += stride) {
        h_rows[i] = d_rows[i];
        d_rows[i] = zero;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system(); /* this block's rows have reached the host's memory */
        bool last = true;
        if (gridDim.x > 1u) {
            last = atomicAdd(done, 1u) + 1u == gridDim.x;
            if (last) *done = 0;
        }
        if (last) *done_flag = ticket;
    }
}

// ---------------------------------------------------------------------------------------------- eval
struct WaveTally { /* per-lane running sums of the current (wave, query) pair */
    uint32_t code[MCQ_N_CODES], tie, passes;
    bool dirty;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int c = 0; c < MCQ_N_CODES; c++) code[c] = 0;
        tie = passes = 0;
        dirty = false;
    }
    __device__ __forceinline__ void add(const McqLaneAcc &a) {
#pragma unroll
        for (int c = 0; c < MCQ_N_CODES; c++)
            if (c != 5) code[c] += (uint32_t)(a.types >> (6 * c)) & 63u;
        tie += a.tie;
        passes += a.passes;
        dirty = true;
    }
    /* lanes -> wave: lane k < 12 returns word k + 1 of the result row (passes, win, tie, by_type[9]).  For the
     * one-launch path: the whole wave is active and a lane has added at most MCQ_DIRECT_TASKS_LIMIT tasks of 16
     * iterations, so a counter's wave sum is below 2^16 and two counters share a reduction. */
    __device__ __forceinline__ unsigned long long row_words(uint32_t lane) {
        static_assert(64u * 16u * MCQ_DIRECT_TASKS_LIMIT < 65536u, "two counters per 32-bit reduction");
        uint32_t sums[MCQ_N_CODES + 1]; /* the codes (5 unused), then the ties */
#pragma unroll
        for (uint32_t c = 0; c < MCQ_N_CODES; c += 2) {
            const uint32_t hi = c + 1 < MCQ_N_CODES ? code[c + 1] : 0u;
            const uint32_t v = wave_sum_dpp(code[c] | (hi << 16));
            sums[c] = v & 0xFFFFu;
            sums[c + 1] = v >> 16;
        }
        const uint32_t ties = wave_sum_dpp(tie);
        const uint32_t pass = wave_sum_dpp(passes);
        uint32_t wins = 0;
        unsigned long long mine = 0;
#pragma unroll
        for (uint32_t c = 0; c < MCQ_N_CODES; c++) {
            if (c == 5) continue;
            wins += sums[c];
            if (lane == 3u + mcq_code_to_type(c)) mine = sums[c];
        }
        if (lane == 0) mine = pass;
        if (lane == 1) mine = wins - ties;
        if (lane == 2) mine = ties;
        return mine;
    }
    /* lanes -> wave (shuffles): lane l < 12 holds word 1 + l of the row (passes, win, tie, by_type[9]) */
    __device__ __forceinline__ uint64_t row_value(uint32_t lane) {
        uint32_t wins = 0;
        uint64_t mine = 0;
#pragma unroll
        for (uint32_t c = 0; c < MCQ_N_CODES; c++) {
            if (c == 5) continue;
            const uint32_t v = wave_sum(code[c]);
            wins += v;
            if (lane == 3u + mcq_code_to_type(c)) mine = v;
        }
        const uint32_t ties = wave_sum(tie);
        const uint32_t pass = wave_sum(passes);
        if (lane == 0) mine = pass;
        if (lane == 1) mine = wins - ties;
        if (lane == 2) mine = ties;
        return mine;
    }
    /* ... -> one 64-bit atomic per counter */
    __device__ __forceinline__ void flush(mcq_result *row, uint32_t lane) {
        if (!dirty) return;
        const uint64_t mine = row_value(lane);
        if (lane < 12 && mine != 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(row) + 1 + lane, (unsigned long long)mine);
        clear();
    }
};

template <int MODE, bool SPLIT>
__global__ __launch_bounds__(kMaxBlock) void mcq_eval_kernel(const mcq_query *__restrict__ queries, uint32_t n,
                                                             const uint64_t *__restrict__ prefix,
                                                             mcq_result *__restrict__ res, uint64_t seed,
                                                             uint64_t first_qid, const McqTables *__restrict__ g_tab,
                                                             const uint8_t *__restrict__ draws,
                                                             const uint64_t *__restrict__ draw_off, uint32_t split_arg,
                                                             uint32_t part, uint32_t n_parts, uint32_t work_wpb) {
    /* SPLIT = false: the bulk path, compiled without the cut */
    const uint32_t split = !SPLIT ? 0u
                           : split_arg == MCQ_SPLIT_FROM_PREP ? (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)prefix[n + 1])
                                                              : split_arg;
    /* split (0..4): small batches cut every 1024-iteration task into 2^split sub-tasks of 16 >> split iterations per
     * lane so that more waves share the work; the iterations and their random numbers stay the same (a sub-task
     * skips ahead in its lane's stream), so the tallies do not depend on it. */
    __shared__ __attribute__((aligned(16))) LdsTablesEval tab;
    __shared__ McqCard base_tab[kMaxBlock]; /* per wave: the query's ordered remaining deck, 64 entries x 16 B */
    /* SPLIT (small batches): the rows the work-group's waves END on are added up in LDS when they are one query's -- a
     * single long query is hundreds of waves, and twelve atomics per wave on ONE row serialise (a 100 000-run query: 392
     * waves, 18 us where the arithmetic takes 6) -- and the last wave sends the sum: s_key = that query (claimed by the
     * first wave to finish), s_done = waves that have finished */
    __shared__ unsigned long long s_row[12];
    __shared__ uint32_t s_key, s_done;
    if (SPLIT) {
        if (threadIdx.x < 12u) s_row[threadIdx.x] = 0ull;
        if (threadIdx.x == 12u) {
            s_key = 0xFFFFFFFFu;
            s_done = 0u;
        }
    }
    load_tables(tab, g_tab); /* (ends with a block barrier) */

    const uint32_t lane = threadIdx.x & 63u;
    /* small batches launch more waves than take work: the extra ones only help to bring the 97 KB table image in */
    const uint32_t waves_per_block = work_wpb ? work_wpb : blockDim.x >> 6;
    if ((threadIdx.x >> 6) >= waves_per_block) return;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * waves_per_block + (threadIdx.x >> 6));
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_block;
    const uint32_t chunk = MCQ_STREAM_ITERS >> split, sub_mask = (1u << split) - 1u;
    const uint64_t total = prefix[n] << split; /* cost axis in sub-task units */
    /* this wave's slice of the cost axis; a task belongs to the slice its start position falls into */
    const uint64_t lo = total * wave / n_waves, hi = total * (wave + 1ull) / n_waves;
    McqCard *base = base_tab + (threadIdx.x & ~63u);

    uint32_t a = 0, b = n; /* last query with prefix <= lo: it has a positive cost because lo < total */
    while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if ((prefix[mid] << split) <= lo) a = mid; else b = mid;
    }
    uint32_t qi = __builtin_amdgcn_readfirstlane(a);
    uint32_t task = 0, task0 = 0, n_tasks = 0, weight = 1; /* tasks [task0, n_tasks) of the query are this launch's */
    uint64_t pfx = 0;
    McqQueryCtx qc;
    WaveTally tally;
    tally.clear();
    bool fresh = true; /* query record qi not loaded yet */
    for (; lo < hi;) { /* (a wave without a slice goes straight to the end: the work-group counts it there) */
        if (fresh) {
            if (qi >= n) break;
            const uint4 raw = reinterpret_cast<const uint4 *>(queries)[qi];
            const McqQueryWords q = {raw.x, raw.y, raw.z, raw.w};
            pfx = prefix[qi] << split;
            const bool ok = (prefix[qi + 1] << split) > pfx; /* zero cost: invalid or runs == 0 */
            const McqPart pt = mcq_part(mcq_task_count(q), q.runs(), part, n_parts);
            task0 = pt.t_lo << split;
            n_tasks = ok ? pt.t_hi << split : 0u;
            weight = mcq_task_weight(q);
            task = task0;
            if (pfx < lo) task = task0 + (uint32_t)((lo - pfx + weight - 1) / weight); /* only for the first query */
            fresh = false;
            if (ok) {
                mcq_query_ctx(q, qc);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); /* earlier lookups are done (same wave) */
                base[lane] = mcq_base_entry(qc, lane, tab.sel8);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (task >= n_tasks) {
            tally.flush(res + qi, lane);
            qi++;
            fresh = true;
            continue;
        }
        if (pfx + (uint64_t)(task - task0) * weight >= hi) break;

        McqLaneAcc acc = {0, 0, 0};
        if (MODE != MCQ_MODE_REPLAY_MT19937) {
            const uint32_t stream = (task >> split) * MCQ_WAVE + lane, sub = task & sub_mask;
            const uint64_t it0 = (uint64_t)stream * MCQ_STREAM_ITERS + sub * chunk;
            if (it0 < qc.runs) {
                McqCtrDrawsT<MODE == MCQ_INTERNAL_MODE_UNIFORM> dr;
                dr.start(seed, first_qid + qi, stream);
                /* words per iteration: one per opponent, one per two table cards */
                for (uint32_t k = sub * chunk * (qc.n_opp + ((qc.n_deal + 1u) >> 1)); k != 0; k--) dr.rng.next();
                const uint32_t cnt = (uint32_t)min((uint64_t)chunk, (uint64_t)qc.runs - it0);
                mcq_iterations<true>(qc, dr, base - 128, g_tab->tf, tab.tops, tab.sd, acc, cnt); /* both dealing laws */
                acc.passes = cnt * qc.n_opp; /* MCQ-CTR v5: one attempt per opponent, never re-drawn */
            }
        } else {
            /* a lane takes FOUR consecutive iterations at a time (one 32-bit load per draw row, McqReplayDraws4); a task
             * is four such groups per lane, a sub-task (split <= 2: the host never cuts this mode finer) whole groups */
            const uint64_t stride = (qc.runs + 63u) & ~63ull;
            const uint8_t *dbase = draws + draw_off[qi];
            const uint32_t groups = chunk >> 2;
            for (uint32_t g = (task & sub_mask) * groups, ge = g + groups; g < ge; g++) {
                const uint64_t it4 = (uint64_t)(task >> split) * MCQ_TASK_ITERS + (g * MCQ_WAVE + lane) * 4u;
                if (it4 < qc.runs) {
                    McqReplayDraws4 dr;
                    dr.load(dbase + it4, stride, qc.n_opp, qc.n_deal);
                    const uint32_t cnt4 = (uint32_t)min((uint64_t)4u, (uint64_t)qc.runs - it4);
                    for (uint32_t k = 0; k < cnt4; k++) {
                        dr.sh = 8u * k;
                        mcq_iteration(qc, dr, base - 128, g_tab->tf, tab.tops, tab.sd, acc);
                    }
                }
            }
            acc.passes = 0; /* `passes` comes from the stream walk: mcq_mt_parse_kernel writes it into the row (the host walk
                             * of mcq_eval_batch_numpy_stream patches it in afterwards) */
        }
        tally.add(acc);
        task++;
    }
    if (!SPLIT) {
        if (qi < n) tally.flush(res + qi, lane);
        return;
    }
    const bool have = qi < n && tally.dirty; /* (wave-uniform) */
    if (have) {
        const uint64_t mine = tally.row_value(lane);
        uint32_t key = 0;
        if (lane == 0) key = atomicCAS(&s_key, 0xFFFFFFFFu, qi);
        key = __builtin_amdgcn_readfirstlane(key);
        if (key == 0xFFFFFFFFu || key == qi) {
            if (lane < 12u && mine != 0) atomicAdd(&s_row[lane], (unsigned long long)mine);
        } else if (lane < 12u && mine != 0) { /* another query's tail: straight to its row */
            atomicAdd(reinterpret_cast<unsigned long long *>(res + qi) + 1 + lane, (unsigned long long)mine);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); /* the sums are in LDS before this wave counts as finished */
    uint32_t done = 0;
    if (lane == 0) done = atomicAdd(&s_done, 1u);
    done = __builtin_amdgcn_readfirstlane(done);
    if (done + 1u == waves_per_block) { /* the last wave of the work-group */
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const uint32_t key = *(volatile uint32_t *)&s_key;
        if (key != 0xFFFFFFFFu && lane < 12u) {
            const unsigned long long v = *(volatile unsigned long long *)&s_row[lane];
            if (v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(res + key) + 1 + lane, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------- eval, small queries
// The reference's own call pattern is thousands of SMALL queries (1000 runs each, gym_env/env.py:22,261-262).  For
// those the cost of a call is not the arithmetic but the launches and copies around it, so this kernel needs
// nothing else: it takes the query records from its own arguments (a launch of up to 128 waves) or straight from the
// library's pinned, device-visible staging memory, every query is owned by ONE block -- 2^split <= 8 waves take the
// 2^split cuts of each of its 1024-iteration tasks -- the waves' sums meet in LDS and one wave stores the finished
// 104-byte row straight into pinned host memory.  No prep kernel, no atomics in HBM, no zeroing, no copy kernels.
// Iterations, random numbers and hence tallies are those of mcq_eval_kernel (same streams, same cut arithmetic).
// The host lays the work out (eval_host_philox in mcq_host.cpp): every query gets a power-of-two number of waves in
// proportion to its cost, so that all waves carry about the same work, and the waves of a query sit side by side in
// one block.  Wave `v` of block `b` in round `r` finds its work at index (r * gridDim.x + b) * 16 + v:
//   work_qi[]   the query's index (its RNG stream key and result row), or MCQ_DIRECT_IDLE
//   work_rec[]  a copy of the 16-byte query record whose reserved bytes carry log2(waves of the query) and this
//               wave's cut number
// -- both read by a few lanes per block, eight rounds at a time, into LDS while the table image travels global -> LDS
// (every wave fetching its own record over PCIe costs more than the arithmetic; from the kernel arguments a wave does
// read its first record itself, by scalar loads, and sets its generator up before the image has landed).  done[0]
// (device memory, zero before the launch, reset by the last block) counts finished blocks of a grid of several; the
// last one raises done_flag (pinned host memory) to `ticket` after a system-scope fence, which lets the host pick the
// rows up without waiting for the end-of-grid handshake.
#define MCQ_DIRECT_STAGE_ROUNDS 8u
#ifdef MCQ_DIRECT_STAMPS /* diagnostic build (tools/direct_stamps.py): 100 MHz timestamps of block 0's waves */
__device__ unsigned long long mcq_direct_stamps[16][16];
#define MCQ_STAMP(k)                                                                                           \
    do {                                                                                                       \
        if (blockIdx.x == 0 && (threadIdx.x & 63u) == 0) mcq_direct_stamps[threadIdx.x >> 6][k] = wall_clock64(); \
    } while (0)
extern "C" __attribute__((visibility("default"))) int mcq_debug_read_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcq_direct_stamps), sizeof(mcq_direct_stamps));
}
#else
#define MCQ_STAMP(k)
#endif
// Queries resident in HBM (mcq_eval_batch_device_small; use_karg == 2): nobody has laid the work out -- slot `at` IS the
// wave number: every query gets 2^lg waves (the host picks lg from the query COUNT alone), slot at -> query at >> lg, cut
// at & (2^lg - 1); work_rec points at the caller's mcq_query array.  The queries are validated here: one that is invalid
// or has more than MCQ_DIRECT_DEV_TASKS tasks gets runs = 0, passes = UINT64_MAX and no work, one without iterations a
// row of zeros (the slot of its first wave writes it).
#define MCQ_DIRECT_DEV_TASKS 8u
__device__ __forceinline__ void mcq_direct_fetch_dev(const uint4 *__restrict__ queries, size_t at, uint32_t n, uint32_t lg,
                                                     mcq_result *__restrict__ res, uint32_t &qi, uint4 &rec) {
    const uint32_t q = (uint32_t)(at >> lg), sub = (uint32_t)at & ((1u << lg) - 1u);
    qi = MCQ_DIRECT_IDLE;
    if (q >= n) return;
    rec = queries[q];
    const McqQueryWords w = {rec.x, rec.y, rec.z, rec.w};
    const bool ok = mcq_query_valid(w) && mcq_task_count(w) <= MCQ_DIRECT_DEV_TASKS;
    if (ok && w.runs() != 0u) {
        qi = q;
        rec.z |= (lg << 8) | (sub << 16); /* reserved[0], reserved[1]: as the host's layout writes them */
    } else if (sub == 0u) {
        unsigned long long *r = reinterpret_cast<unsigned long long *>(res + q);
        r[0] = 0ull;
        r[1] = ok ? 0ull : ~0ull;
#pragma unroll
        for (int k = 2; k < 13; k++) r[k] = 0ull;
    }
}

template <int MODE>
__global__ __launch_bounds__(kMaxBlock) void mcq_eval_direct_kernel(const uint4 *__restrict__ work_rec,
                                                                    const uint32_t *__restrict__ work_qi, uint32_t rounds,
                                                                    uint32_t merge, mcq_result *__restrict__ res, uint64_t seed,
                                                                    uint64_t first_qid, const McqTables *__restrict__ g_tab,
                                                                    uint32_t *__restrict__ done, volatile uint32_t *done_flag,
                                                                    uint32_t ticket, uint32_t use_karg, McqDirectKarg karg) {
    constexpr uint32_t kWaves = kMaxBlock / 64, kStage = MCQ_DIRECT_STAGE_ROUNDS * kWaves;
    __shared__ __attribute__((aligned(16))) LdsTablesEval tab;
    __shared__ McqCard base_tab[kMaxBlock];
    __shared__ unsigned long long partial[2][kWaves][12]; /* [round parity][wave][passes, win, tie, by_type[9]] */
    __shared__ uint4 s_rec[kStage];
    __shared__ uint32_t s_qi[kStage];

    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    McqCard *base = base_tab + (threadIdx.x & ~63u);
    typedef McqCtrDrawsT<MODE == MCQ_INTERNAL_MODE_UNIFORM> Draws;
    /* The start of a one-launch query is a chain of latencies, so they overlap: the first rounds' work is asked for,
     * the table image is sent on its way global -> LDS without passing through registers (global_load_lds_dwordx4:
     * every wave instruction moves 1 KB), and while it travels every wave prepares the generator of its first task
     * (Philox, and for a query cut over several waves the walk to this wave's place in the lane streams).  One
     * barrier then publishes the image and the staged work. */
    MCQ_STAMP(0);
    /* a small launch (its work came with the kernel arguments): this wave's round-0 record by scalar loads from its
     * own copy -- the staged one needs the barrier.  (Not for work in pinned host memory: a small read per wave
     * across PCIe costs more than it hides.) */
    uint32_t qi0 = MCQ_DIRECT_IDLE;
    uint4 raw0 = {0u, 0u, 0u, 0u};
    if (use_karg == 1u) {
        const size_t at0 = (size_t)blockIdx.x * kWaves + __builtin_amdgcn_readfirstlane(wib);
        qi0 = karg.qi[at0];
        raw0 = make_uint4(karg.rec[at0][0], karg.rec[at0][1], karg.rec[at0][2], karg.rec[at0][3]);
    }
    uint4 pre_rec = {0u, 0u, 0u, 0u};
    uint32_t pre_qi = MCQ_DIRECT_IDLE;
    if (threadIdx.x < kStage && threadIdx.x / kWaves < rounds) {
        const size_t at = ((size_t)(threadIdx.x / kWaves) * gridDim.x + blockIdx.x) * kWaves + threadIdx.x % kWaves;
        if (use_karg == 1u) {
            pre_qi = karg.qi[at];
            pre_rec = make_uint4(karg.rec[at][0], karg.rec[at][1], karg.rec[at][2], karg.rec[at][3]);
        } else if (use_karg == 2u) {
            mcq_direct_fetch_dev(work_rec, at, karg.qi[0], karg.qi[1], res, pre_qi, pre_rec);
        } else {
            pre_qi = work_qi[at];
            pre_rec = work_rec[at];
        }
    }
    {
        typedef const __attribute__((address_space(1))) void *GlobalPtr;
        typedef __attribute__((address_space(3))) void *LdsPtr;
        constexpr uint32_t kChunks = (MCQ_TF_BYTE_OFFSET + 1024u) / 1024u; /* tops, sd | kc, sel8: 1 KB per wave instruction */
        static_assert(sizeof(LdsTablesEval) == kChunks * 1024u, "table image in LDS: whole 1 KB pieces");
        const char *src = reinterpret_cast<const char *>(g_tab);
        char *dst = reinterpret_cast<char *>(&tab);
        const char *sel = reinterpret_cast<const char *>(g_tab->sel8); /* the last piece: not behind sd in the global image */
        /* the same number of pieces for every wave (the last one, sel8, is sent by all of them: same bytes, same
         * place), so that the count of loads in flight is a constant the compiler can wait against */
        constexpr uint32_t kPer = (kChunks + kWaves - 1u) / kWaves;
        static_assert((kPer - 1u) * kWaves == kChunks - 1u, "pieces per wave");
        const uint32_t wu = __builtin_amdgcn_readfirstlane(wib);
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            const uint32_t c = k + 1u < kPer ? wu + k * kWaves : kChunks - 1u;
            __builtin_amdgcn_global_load_lds((GlobalPtr)((k + 1u < kPer ? src + c * 1024u : sel) + lane * 16u),
                                             (LdsPtr)(dst + c * 1024u), 16, 0, 0);
        }
    }
    /* round 0, task 0 of this wave: the lanes that have iterations to run there */
    Draws dr0;
    {
        const uint32_t qi = __builtin_amdgcn_readfirstlane(qi0);
        if (qi != MCQ_DIRECT_IDLE) {
            const uint32_t w2 = __builtin_amdgcn_readfirstlane(raw0.z);
            const uint32_t split = (w2 >> 8) & 0xFFu, sub = (w2 >> 16) & 0xFFu;
            const uint32_t chunk = MCQ_STREAM_ITERS >> split;
            const McqQueryWords q = {(uint32_t)__builtin_amdgcn_readfirstlane(raw0.x), (uint32_t)__builtin_amdgcn_readfirstlane(raw0.y),
                                     w2 & 0xFFu, (uint32_t)__builtin_amdgcn_readfirstlane(raw0.w)};
            McqQueryCtx qc;
            mcq_query_ctx(q, qc);
            if ((uint64_t)lane * MCQ_STREAM_ITERS + sub * chunk < qc.runs) {
                dr0.start(seed, first_qid + qi, lane);
                for (uint32_t k = sub * chunk * (qc.n_opp + ((qc.n_deal + 1u) >> 1)); k != 0; k--) dr0.rng.next();
            }
        }
    }
    MCQ_STAMP(1);
    for (uint32_t g0 = 0; g0 < rounds; g0 += MCQ_DIRECT_STAGE_ROUNDS) {
        if (g0) __syncthreads(); /* the previous rounds' work has been read by every wave */
        else __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): this wave's pieces of the table image have landed */
        if (threadIdx.x < kStage) {
            if (g0) {
                const uint32_t r = g0 + threadIdx.x / kWaves;
                pre_qi = MCQ_DIRECT_IDLE;
                if (r < rounds) {
                    const size_t at = ((size_t)r * gridDim.x + blockIdx.x) * kWaves + threadIdx.x % kWaves;
                    if (use_karg == 2u) {
                        mcq_direct_fetch_dev(work_rec, at, karg.qi[0], karg.qi[1], res, pre_qi, pre_rec);
                    } else {
                        pre_qi = work_qi[at];
                        pre_rec = work_rec[at];
                    }
                }
            }
            s_qi[threadIdx.x] = pre_qi;
            s_rec[threadIdx.x] = pre_rec;
        }
        __syncthreads();
        MCQ_STAMP(2);
        const uint32_t g1 = g0 + MCQ_DIRECT_STAGE_ROUNDS < rounds ? g0 + MCQ_DIRECT_STAGE_ROUNDS : rounds;
        for (uint32_t round = g0; round < g1; round++) {
            const uint32_t at = (round - g0) * kWaves + wib;
            const uint32_t qi = __builtin_amdgcn_readfirstlane(s_qi[at]);
            const bool work = qi != MCQ_DIRECT_IDLE;
            const uint4 raw = s_rec[at];
            const uint32_t w2 = __builtin_amdgcn_readfirstlane(raw.z);
            const uint32_t split = (w2 >> 8) & 0xFFu, sub = (w2 >> 16) & 0xFFu, wpq = 1u << split;
            const uint32_t chunk = MCQ_STREAM_ITERS >> split;
            unsigned long long mine = 0; /* lane k < 12: word k + 1 of the row */
            uint32_t runs = 0;
            if (work) {
                const McqQueryWords q = {(uint32_t)__builtin_amdgcn_readfirstlane(raw.x), (uint32_t)__builtin_amdgcn_readfirstlane(raw.y),
                                         w2 & 0xFFu, (uint32_t)__builtin_amdgcn_readfirstlane(raw.w)};
                McqQueryCtx qc;
                mcq_query_ctx(q, qc);
                runs = qc.runs;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); /* the previous query's lookups are done */
                base[lane] = mcq_base_entry(qc, lane, tab.sel8);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                MCQ_STAMP(3);
                WaveTally tally;
                tally.clear();
                const uint32_t tasks = mcq_task_count(q);
                for (uint32_t task = 0; task < tasks; task++) {
                    McqLaneAcc acc = {0, 0, 0};
                    const uint32_t stream = task * MCQ_WAVE + lane;
                    const uint64_t it0 = (uint64_t)stream * MCQ_STREAM_ITERS + sub * chunk;
                    if (it0 < qc.runs) {
                        Draws dr;
                        if (use_karg == 1u && round == 0u && task == 0u) {
                            dr = dr0; /* prepared while the table image was on its way */
                        } else {
                            dr.start(seed, first_qid + qi, stream);
                            for (uint32_t k = sub * chunk * (qc.n_opp + ((qc.n_deal + 1u) >> 1)); k != 0; k--) dr.rng.next();
                        }
                        MCQ_STAMP(5);
                        const uint32_t cnt = (uint32_t)min((uint64_t)chunk, (uint64_t)qc.runs - it0);
                        for (uint32_t j = 0; j < cnt; j++) mcq_iteration(qc, dr, base - 128, g_tab->tf, tab.tops, tab.sd, acc);
                        acc.passes = cnt * qc.n_opp;
                        MCQ_STAMP(6);
                    }
                    tally.add(acc);
                    MCQ_STAMP(7);
                }
                mine = tally.row_words(lane);
            }
            /* Row word w = min(lane, 12): lanes 13..63 repeat lane 12's store (same address, same value) so that no
             * branch on the lane number stands in front of the loop's back edge -- the wave stays whole for the
             * cross-lane steps of the next round. */
            const uint32_t w = lane < 12u ? lane : 12u;
            if (!merge) { /* no query of this launch has more than one wave: a wave's sums are the row */
                const unsigned long long up = __shfl(mine, (int)((w + 63u) & 63u), 64); /* every lane takes part */
                if (work) reinterpret_cast<unsigned long long *>(res + qi)[w] = w == 0u ? (unsigned long long)runs : up;
                continue;
            }
            if (lane < 12u) partial[round & 1u][wib][lane] = mine;
            MCQ_STAMP(8);
            __syncthreads(); /* every wave of the block, every round; two buffers: a wave may run one round ahead */
            if (work && sub == 0u) {
                unsigned long long v = runs;
                if (w > 0u) {
                    v = 0;
                    for (uint32_t k = 0; k < wpq; k++) v += partial[round & 1u][wib + k][w - 1u];
                }
                reinterpret_cast<unsigned long long *>(res + qi)[w] = v;
            }
            MCQ_STAMP(9);
        }
    }
    /* completion: rows first (ONE system-scope release per block, behind the barrier that orders the other waves'
     * stores before it -- a fence in every wave would write the L2 back 16 times over), then the count; the block
     * that completes the count tells the host */
    __syncthreads();
    MCQ_STAMP(10);
    if (threadIdx.x == 0) {
        __threadfence_system(); /* this block's rows have reached the host's memory */
        MCQ_STAMP(11);
        bool last = true;
        if (gridDim.x > 1u) { /* each block counts itself in behind its own release; the one that completes the
                               * count has therefore seen every block's rows leave */
            last = atomicAdd(done, 1u) + 1u == gridDim.x;
            if (last) *done = 0; /* ready for the next launch on this stream */
        }
        if (last) *done_flag = ticket;
        MCQ_STAMP(12);
    }
}

// ---------------------------------------------------------------------------------------------- extended queries
// SURVEY 8f-2 (ranges, ghost cards, any number of known hands, each two cards or a range): same slicing and tallying
// as above, the mask-based iteration of mcq_iteration_ext.  Production mode draws from candidate lists that
// mcq_ext_lists_kernel lays out once per query (mcq_device.hpp).  A range that could not be dealt zeroes the row's
// `runs`.
__global__ __launch_bounds__(1024) void mcq_prep_ext_kernel(const mcq_query *__restrict__ q,
                                                            const mcq_query_ext *__restrict__ ext, uint32_t n, int mode,
                                                            mcq_result *__restrict__ res, uint64_t *__restrict__ prefix) {
    __shared__ uint64_t wave_tot[16];
    uint64_t carry = 0;
    const uint32_t tid = threadIdx.x;
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + tid;
        uint64_t cost = 0;
        if (i < n) {
            const uint4 raw = reinterpret_cast<const uint4 *>(q)[i];
            const McqQueryWords qq = {raw.x, raw.y, raw.z, raw.w};
            const McqExtRec er = {reinterpret_cast<const uint32_t *>(ext + i)};
            bool ok = mcq_query_ext_valid(qq, er);
            const uint32_t s_iters = ok && mode == MCQ_MODE_PHILOX ? mcq_ext_stream_iters(qq, er) : MCQ_STREAM_ITERS;
            cost = ok ? (uint64_t)mcq_ext_task_count(qq, s_iters) * mcq_ext_task_weight(qq, s_iters) : 0ull;
            uint64_t *r = reinterpret_cast<uint64_t *>(res + i);
            r[0] = ok ? qq.runs() : 0ull;
            r[1] = ok ? 0ull : ~0ull;
#pragma unroll
            for (int k = 2; k < 13; k++) r[k] = 0;
        }
        uint64_t chunk_total;
        const uint64_t before = block_exclusive_scan_1024(cost, wave_tot, &chunk_total);
        if (i < n) prefix[i] = carry + before;
        carry += chunk_total;
    }
    if (tid == 0) {
        prefix[n] = carry;
        prefix[n + 1] = 0;
        prefix[n + 2] = 0; /* work counter of mcq_mt_parse_ext_kernel */
    }
}

// The candidate lists of the production mode: one block per query; list li of the query = all ordered pairs (a, b) of
// cards of U_li whose class its range allows, in the order of 52 * a + b (the specification's order: the oracle builds
// the same list).  Every thread looks at eleven consecutive candidates; a block scan of the counts puts the survivors
// in order.  cnts[query * lists_stride + li] = the list's length (0 for a query that is invalid).
constexpr int kListBlock = 256;
__global__ __launch_bounds__(kListBlock) void mcq_ext_lists_kernel(const mcq_query *__restrict__ q,
                                                                   const mcq_query_ext *__restrict__ ext, uint32_t lists_stride,
                                                                   uint16_t *__restrict__ lists, uint32_t *__restrict__ cnts) {
    __shared__ uint32_t wave_tot[kListBlock / 64];
    const uint32_t qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint4 raw = reinterpret_cast<const uint4 *>(q)[qi];
    const McqQueryWords qq = {raw.x, raw.y, raw.z, raw.w};
    const McqExtRec er = {reinterpret_cast<const uint32_t *>(ext + qi)};
    const bool valid = mcq_query_ext_valid(qq, er);
    const uint32_t n_lists = valid ? mcq_ext_n_lists(qq, er) : 0u;
    for (uint32_t li = 0; li < lists_stride; li++) {
        if (li >= n_lists) { /* block-uniform */
            if (tid == 0) cnts[(size_t)qi * lists_stride + li] = 0;
            continue;
        }
        uint64_t U;
        uint32_t set_off;
        mcq_ext_list_plan(qq, er, li, U, set_off);
        constexpr uint32_t kPer = (2704u + kListBlock - 1) / kListBlock; /* 11 */
        uint32_t mask = 0;
        for (uint32_t k = 0; k < kPer; k++)
            if (mcq_ext_candidate(U, er.w + set_off, tid * kPer + k)) mask |= 1u << k;
        const uint32_t mine = (uint32_t)__popc(mask);
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += o;
        }
        __syncthreads(); /* wave_tot of the previous list has been read */
        if (lane == 63u) wave_tot[wv] = inc;
        __syncthreads();
        uint32_t at = inc - mine, total = 0;
        for (uint32_t k = 0; k < kListBlock / 64; k++) {
            at += k < wv ? wave_tot[k] : 0u;
            total += wave_tot[k];
        }
        uint16_t *dst = lists + ((size_t)qi * lists_stride + li) * MCQ_EXT_LIST_STRIDE;
        for (uint32_t k = 0; k < kPer; k++)
            if ((mask >> k) & 1u) {
                const uint32_t c = tid * kPer + k, a = c / 52u;
                dst[at++] = (uint16_t)(a | ((c - 52u * a) << 8));
            }
        if (tid == 0) cnts[(size_t)qi * lists_stride + li] = total;
    }
}

template <int MODE>
__global__ __launch_bounds__(kExtBlock) void mcq_eval_ext_kernel(const mcq_query *__restrict__ queries,
                                                                 const mcq_query_ext *__restrict__ ext, uint32_t n,
                                                                 const uint64_t *__restrict__ prefix,
                                                                 mcq_result *__restrict__ res, uint64_t seed,
                                                                 uint64_t first_qid, const McqTables *__restrict__ g_tab,
                                                                 const uint8_t *__restrict__ draws,
                                                                 const uint64_t *__restrict__ draw_off,
                                                                 const uint16_t *__restrict__ lists,
                                                                 const uint32_t *__restrict__ cnts, uint32_t lists_stride) {
    __shared__ __attribute__((aligned(16))) LdsTablesEval tab;
    __shared__ McqCard cards[64];
    __shared__ McqExtWaveCtx wave_ctx[kExtBlock / 64];
    __shared__ uint16_t ids[(MCQ_MAX_OPP + 1) * kExtBlock];
    /* The candidate lists of the production mode are read once per trial at a random index: from HBM that is a
     * dependent ~700-cycle load per trial, several per opponent, and it -- not the arithmetic -- bounded the kernel
     * (34 000 cycles per wave-iteration at the top quarter of the classes).  So a block first brings the lists of ITS
     * queries -- a block's waves cover a contiguous piece of the cost axis, hence consecutive queries -- into LDS when
     * they fit (36 KB beside the tables; 2 B per ordered card pair), and the trials read them there. */
    constexpr uint32_t kStageEntries = 18u * 1024u, kStageLists = 96u;
    __shared__ __attribute__((aligned(16))) uint16_t s_lists[kStageEntries];
    __shared__ uint32_t s_list_off[kStageLists + 1];
    __shared__ uint32_t s_stage[3]; /* first query, number of queries, staged? */
    if (threadIdx.x < 64) cards[threadIdx.x] = mcq_card(threadIdx.x < 52 ? threadIdx.x : 0u);
    load_tables(tab, g_tab);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * waves_per_block + (threadIdx.x >> 6));
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_block;
    const uint64_t total = prefix[n];
    const uint64_t lo = total * wave / n_waves, hi = total * (wave + 1ull) / n_waves;
    if (MODE == MCQ_MODE_PHILOX) {
        if (threadIdx.x == 0) { /* the block's queries: those whose cost interval meets the block's piece of the axis */
            const uint64_t blo = total * ((uint64_t)blockIdx.x * waves_per_block) / n_waves;
            const uint64_t bhi = total * ((uint64_t)(blockIdx.x + 1u) * waves_per_block) / n_waves;
            uint32_t qa = 0, qb = 0, nq = 0, staged = 0;
            if (blo < bhi) {
                uint32_t a = 0, b = n;
                while (b - a > 1) {
                    const uint32_t mid = (a + b) >> 1;
                    if (prefix[mid] <= blo) a = mid; else b = mid;
                }
                qa = a;
                a = qa, b = n;
                while (b - a > 1) { /* last query that starts before the block's end */
                    const uint32_t mid = (a + b) >> 1;
                    if (prefix[mid] < bhi) a = mid; else b = mid;
                }
                qb = a;
                nq = qb - qa + 1u;
                if ((uint64_t)nq * lists_stride <= kStageLists) {
                    uint32_t at = 0;
                    for (uint32_t k = 0; k < nq * lists_stride; k++) {
                        s_list_off[k] = at;
                        at += (cnts[(size_t)qa * lists_stride + k] + 1u) & ~1u; /* whole 32-bit words */
                        if (at > kStageEntries) break;
                    }
                    s_list_off[nq * lists_stride] = at;
                    staged = at <= kStageEntries ? 1u : 0u;
                }
            }
            s_stage[0] = qa;
            s_stage[1] = nq;
            s_stage[2] = staged;
        }
        __syncthreads();
        if (s_stage[2]) {
            const uint32_t qa = s_stage[0], n_l = s_stage[1] * lists_stride;
            for (uint32_t k = 0; k < n_l; k++) { /* (block-uniform loop; 1024 threads copy 4 KB in one step) */
                const uint32_t off = s_list_off[k], words = (s_list_off[k + 1] - off) >> 1;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(lists + ((size_t)qa * lists_stride + k) * MCQ_EXT_LIST_STRIDE);
                uint32_t *dst = reinterpret_cast<uint32_t *>(s_lists + off);
                for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
            }
        }
        __syncthreads();
    }
    if (lo >= hi) return;
    McqExtWaveCtx &wc = wave_ctx[threadIdx.x >> 6];
    uint16_t *my_ids = ids + threadIdx.x;

    uint32_t a = 0, b = n;
    while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (prefix[mid] <= lo) a = mid; else b = mid;
    }
    uint32_t qi = __builtin_amdgcn_readfirstlane(a);
    uint32_t task = 0, n_tasks = 0, weight = 1, s_iters = MCQ_STREAM_ITERS;
    uint64_t pfx = 0;
    McqExtCtx qc;
    WaveTally tally;
    tally.clear();
    bool failed = false, fresh = true, list_in_lds = false;
    for (;;) {
        if (fresh) {
            if (qi >= n) break;
            const uint4 raw = reinterpret_cast<const uint4 *>(queries)[qi];
            const McqQueryWords q = {raw.x, raw.y, raw.z, raw.w};
            pfx = prefix[qi];
            bool ok = prefix[qi + 1] > pfx;
            const McqExtRec er = {reinterpret_cast<const uint32_t *>(ext + qi)};
            s_iters = ok && MODE == MCQ_MODE_PHILOX ? mcq_ext_stream_iters(q, er) : MCQ_STREAM_ITERS;
            weight = mcq_ext_task_weight(q, s_iters);
            task = 0;
            if (pfx < lo) task = (uint32_t)((lo - pfx + weight - 1) / weight);
            fresh = false;
            if (ok) {
                mcq_ext_ctx(q, er, qc);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (lane < 10u) wc.hand[lane] = lane < qc.n_hands ? mcq_ext_hand(q, er, lane) : 0u;
                if (MODE == MCQ_MODE_PHILOX) {
                    const uint32_t n_lists = mcq_ext_n_lists(q, er);
                    list_in_lds = __builtin_amdgcn_readfirstlane(s_stage[2] && qi - s_stage[0] < s_stage[1] ? 1 : 0) != 0;
                    uint32_t c = 1;
                    if (lane < MCQ_EXT_MAX_LISTS) {
                        c = lane < n_lists ? cnts[(size_t)qi * lists_stride + lane] : 1u;
                        wc.cnt[lane] = c;
                        const uint16_t *lp = lists + ((size_t)qi * lists_stride + (lane < lists_stride ? lane : 0u)) * MCQ_EXT_LIST_STRIDE;
                        if (s_stage[2] && qi - s_stage[0] < s_stage[1] && lane < lists_stride)
                            lp = s_lists + s_list_off[(qi - s_stage[0]) * lists_stride + lane];
                        wc.list[lane] = lp;
                    }
                    if (__any(c == 0u)) { /* a range no pair of cards can satisfy: nothing to deal from */
                        failed = true;
                        ok = false;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            n_tasks = ok ? mcq_ext_task_count(q, s_iters) : 0u;
        }
        if (task >= n_tasks) {
            if (__any(failed)) {
                if (lane == 0) atomicExch(reinterpret_cast<unsigned long long *>(res + qi), 0ull); /* runs := 0 */
                failed = false;
            }
            tally.flush(res + qi, lane);
            qi++;
            fresh = true;
            continue;
        }
        if (pfx + (uint64_t)task * weight >= hi) break;

        McqLaneAcc acc = {0, 0, 0};
        if (MODE == MCQ_MODE_PHILOX) {
            const uint32_t stream = task * MCQ_WAVE + lane;
            const uint64_t it0 = (uint64_t)stream * s_iters;
            if (it0 < qc.runs) {
                McqExtCtrDraws dr;
                dr.start(seed, first_qid + qi, stream);
                const uint32_t cnt = (uint32_t)min((uint64_t)s_iters, (uint64_t)qc.runs - it0);
                if (qc.fast && list_in_lds) /* (wave-uniform) */
                    for (uint32_t j = 0; j < cnt && !failed; j++)
                        failed = !mcq_iteration_ext_fast<McqExtCtrDraws, true, true>(qc, wc, dr, cards, tab.sel8, g_tab->tf, tab.tops, tab.sd, acc);
                else if (qc.fast)
                    for (uint32_t j = 0; j < cnt && !failed; j++)
                        failed = !mcq_iteration_ext_fast(qc, wc, dr, cards, tab.sel8, g_tab->tf, tab.tops, tab.sd, acc);
                else if (list_in_lds)
                    for (uint32_t j = 0; j < cnt && !failed; j++)
                        failed = !mcq_iteration_ext<McqExtCtrDraws, true>(qc, wc, dr, cards, tab.sel8, my_ids, kExtBlock, g_tab->tf, tab.tops, tab.sd, acc);
                else
                    for (uint32_t j = 0; j < cnt && !failed; j++)
                        failed = !mcq_iteration_ext(qc, wc, dr, cards, tab.sel8, my_ids, kExtBlock, g_tab->tf, tab.tops, tab.sd, acc);
            }
        } else {
            const uint64_t stride = (qc.runs + 63u) & ~63ull;
            const uint8_t *dbase = draws + draw_off[qi];
            for (uint32_t j = 0; j < MCQ_STREAM_ITERS; j++) {
                const uint64_t it = (uint64_t)task * MCQ_TASK_ITERS + j * MCQ_WAVE + lane;
                if (it < qc.runs) {
                    McqExtReplayDraws dr = {dbase + it, stride};
                    mcq_iteration_ext(qc, wc, dr, cards, tab.sel8, my_ids, kExtBlock, g_tab->tf, tab.tops, tab.sd, acc);
                }
            }
            acc.passes = 0;
        }
        tally.add(acc);
        task++;
    }
    if (qi < n) {
        if (__any(failed) && lane == 0) atomicExch(reinterpret_cast<unsigned long long *>(res + qi), 0ull);
        tally.flush(res + qi, lane);
    }
}

// A FEW extended queries in ONE launch (production mode; what a decision of the reference's agents asks for: ONE ranged
// query of a thousand iterations, agent_*.py -> get_equity -> run_montecarlo).  The general path above is six stream
// operations (two copies, prep, lists, evaluation, copy back: 140 us for such a query); here the queries travel in the
// kernel arguments, a block takes ONE query, or one part of one -- a wave task per wave and as few working waves per
// block as the launch's 32 blocks allow (4, 8 or 16: waves that share a SIMD take turns at its vector unit, one wave
// per SIMD runs two iterations in 5 us, four in 9; all sixteen waves help with the tables and the lists): validates it, lays its candidate lists
// out in its own LDS (what mcq_ext_lists_kernel does into HBM), runs its wave tasks, adds the waves' rows up in LDS and
// stores the row into the host's pinned result buffer, one row per BLOCK (the host adds the parts of a query); the last
// block to finish raises the completion flag (as mcq_eval_direct_kernel does).  The host sends queries here whose lists fit (at most MCQ_EXT_SMALL_LISTS) and
// that have at most MCQ_EXT_SMALL_TASKS wave tasks.  Same streams, same draws, same tallies as the general path.
__global__ __launch_bounds__(kExtBlock) void mcq_eval_ext_small_kernel(McqExtSmallKarg karg, mcq_result *__restrict__ res,
                                                                       uint64_t seed, uint64_t first_qid,
                                                                       const McqTables *__restrict__ g_tab,
                                                                       uint32_t *__restrict__ done, volatile uint32_t *done_flag,
                                                                       uint32_t ticket) {
    constexpr uint32_t kWaves = kExtBlock / 64;
    constexpr uint32_t kStageEntries = MCQ_EXT_SMALL_LISTS * MCQ_EXT_LIST_STRIDE;
    __shared__ __attribute__((aligned(16))) LdsTablesEval tab;
    __shared__ McqCard cards[64];
    __shared__ McqExtWaveCtx wave_ctx; /* one query per block: one context */
    __shared__ uint16_t ids[(MCQ_MAX_OPP + 1) * kExtBlock];
    __shared__ __attribute__((aligned(16))) uint16_t s_lists[kStageEntries];
    __shared__ uint32_t s_rec[4 + MCQ_EXT_WORDS];
    __shared__ uint32_t wave_tot[kWaves];
    __shared__ unsigned long long partial[kWaves][12];
    __shared__ uint32_t s_failed;
    const uint32_t blk = karg.blk[blockIdx.x], qi = blk & 0xFFu, part = (blk >> 8) & 0xFFu, parts = (blk >> 16) & 0xFFu, wpb = blk >> 24;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid < 64) cards[tid] = mcq_card(tid < 52 ? tid : 0u);
    if (tid >= 4u && tid < 4u + MCQ_EXT_WORDS) s_rec[tid] = karg.ext[qi][tid - 4u];
    if (tid == 0) s_failed = 0;
    load_tables(tab, g_tab); /* (ends with a barrier) */
    /* (the query words by scalar loads from the kernel arguments: the iteration wants n_players in an SGPR) */
    const McqQueryWords q = {karg.q[qi][0], karg.q[qi][1], karg.q[qi][2], karg.q[qi][3]};
    const McqExtRec er = {s_rec + 4};
    const bool valid = __builtin_amdgcn_readfirstlane(mcq_query_ext_valid(q, er) ? 1 : 0) != 0;
    const uint32_t n_lists = __builtin_amdgcn_readfirstlane(valid ? mcq_ext_n_lists(q, er) : 0u);
    bool ok = valid && n_lists <= MCQ_EXT_SMALL_LISTS;
    /* the candidate lists, in the order of 52 * a + b: every thread looks at three consecutive candidates, a block scan
     * of the counts puts the survivors in order */
    uint32_t list_at = 0;
    for (uint32_t li = 0; li < n_lists && ok; li++) { /* (block-uniform) */
        uint64_t U;
        uint32_t set_off;
        mcq_ext_list_plan(q, er, li, U, set_off);
        constexpr uint32_t kPer = (2704u + kExtBlock - 1) / kExtBlock; /* 3 */
        uint32_t mask = 0;
        for (uint32_t k = 0; k < kPer; k++)
            if (mcq_ext_candidate(U, er.w + set_off, tid * kPer + k)) mask |= 1u << k;
        const uint32_t mine = (uint32_t)__popc(mask);
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += o;
        }
        __syncthreads(); /* wave_tot of the previous list has been read */
        if (lane == 63u) wave_tot[wv] = inc;
        __syncthreads();
        uint32_t at = inc - mine, total = 0;
        for (uint32_t k = 0; k < kWaves; k++) {
            at += k < wv ? wave_tot[k] : 0u;
            total += wave_tot[k];
        }
        total = __builtin_amdgcn_readfirstlane(total);
        uint16_t *dst = s_lists + list_at;
        for (uint32_t k = 0; k < kPer; k++)
            if ((mask >> k) & 1u) {
                const uint32_t c = tid * kPer + k, a = c / 52u;
                dst[at++] = (uint16_t)(a | ((c - 52u * a) << 8));
            }
        if (tid == 0) {
            wave_ctx.cnt[li] = total;
            wave_ctx.list[li] = s_lists + list_at;
        }
        if (total == 0u) ok = false; /* a range no pair of cards can satisfy: nothing to deal from */
        list_at += total;
    }
    McqExtCtx qc;
    mcq_ext_ctx(q, er, qc);
    /* (what came out of the record in LDS is wave-uniform: say so) */
    qc.deck_lo = __builtin_amdgcn_readfirstlane(qc.deck_lo);
    qc.deck_hi = __builtin_amdgcn_readfirstlane(qc.deck_hi);
    qc.fdeck_lo = __builtin_amdgcn_readfirstlane(qc.fdeck_lo);
    qc.fdeck_hi = __builtin_amdgcn_readfirstlane(qc.fdeck_hi);
    qc.n_hands = __builtin_amdgcn_readfirstlane(qc.n_hands);
    qc.opp_all = __builtin_amdgcn_readfirstlane(qc.opp_all ? 1 : 0) != 0;
    qc.fast = __builtin_amdgcn_readfirstlane(qc.fast ? 1 : 0) != 0;
    if (tid < 10u) wave_ctx.hand[tid] = tid < qc.n_hands ? mcq_ext_hand(q, er, tid) : 0u;
    if (tid >= 64u && tid < 64u + MCQ_EXT_MAX_LISTS && tid - 64u >= n_lists) { /* never read; defined all the same */
        wave_ctx.cnt[tid - 64u] = 1u;
        wave_ctx.list[tid - 64u] = s_lists;
    }
    __syncthreads();
    const uint32_t s_iters = __builtin_amdgcn_readfirstlane(valid ? mcq_ext_stream_iters(q, er) : MCQ_STREAM_ITERS);
    const uint32_t n_tasks = __builtin_amdgcn_readfirstlane(ok ? mcq_ext_task_count(q, s_iters) : 0u);
    WaveTally tally;
    tally.clear();
    bool failed = false;
    for (uint32_t task = part * wpb + __builtin_amdgcn_readfirstlane(wv); wv < wpb && task < n_tasks; task += parts * wpb) {
        McqLaneAcc acc = {0, 0, 0};
        const uint32_t stream = task * MCQ_WAVE + lane;
        const uint64_t it0 = (uint64_t)stream * s_iters;
        if (it0 < qc.runs) {
            McqExtCtrDraws dr;
            dr.start(seed, first_qid + qi, stream);
            const uint32_t cnt = (uint32_t)min((uint64_t)s_iters, (uint64_t)qc.runs - it0);
            if (qc.fast) /* (block-uniform) */
                for (uint32_t j = 0; j < cnt && !failed; j++)
                    failed = !mcq_iteration_ext_fast<McqExtCtrDraws, false, true>(qc, wave_ctx, dr, cards, tab.sel8, g_tab->tf, tab.tops, tab.sd, acc);
            else
                for (uint32_t j = 0; j < cnt && !failed; j++)
                    failed = !mcq_iteration_ext<McqExtCtrDraws, true>(qc, wave_ctx, dr, cards, tab.sel8, ids + tid, kExtBlock, g_tab->tf, tab.tops, tab.sd, acc);
        }
        tally.add(acc);
    }
    if (__any(failed) && lane == 0) atomicOr(&s_failed, 1u);
    const unsigned long long mine = tally.row_words(lane);
    if (lane < 12u) partial[wv][lane] = mine;
    __syncthreads();
    if (tid < 13u) { /* word tid of the row */
        unsigned long long v = 0;
        if (tid > 0u)
            for (uint32_t k = 0; k < kWaves; k++) v += partial[k][tid - 1u];
        const bool bad = !ok || s_failed != 0u;
        if (tid == 0u) v = bad ? 0ull : (unsigned long long)qc.runs;
        else if (bad) v = tid == 1u && !valid ? ~0ull : 0ull; /* invalid: passes = UINT64_MAX, as mcq_prep_ext_kernel marks it */
        reinterpret_cast<unsigned long long *>(res + blockIdx.x)[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        __threadfence_system(); /* this block's row has reached the host's memory */
        bool last = true;
        if (gridDim.x > 1u) {
            last = atomicAdd(done, 1u) + 1u == gridDim.x;
            if (last) *done = 0; /* ready for the next launch on this stream */
        }
        if (last) *done_flag = ticket;
    }
}

// ---------------------------------------------------------------------------------------------- showdown
// hand_evaluator.get_winner (tools/hand_evaluator.py:9-24) for many tables: the same ranking key as the Monte-Carlo path.
// HBM/PCIe-bound byte work (7 bytes in per hand, 2 bytes out per table), so the layout is what matters: a block takes a
// TILE of 256 tables, whose hands are one contiguous run of 256 * n_players * 7 bytes -- staged into LDS with 16-byte
// loads straight from the caller-visible buffer (pinned host memory or HBM), evaluated one table per thread with the
// lookup tables read through the vector L1 / L2 (2 M lookups: not worth 97 KB of LDS per block), results leave as
// 16-byte stores.  A hand that does not hold seven distinct ids < 52 marks `bad` (the host reports MCQ_EINVAL) --
// validation on the device, as mcq_prep_kernel does for queries.  With ticket != 0 the last block to finish raises the
// host's completion flag behind a system-scope release, as mcq_publish_kernel does.
constexpr int kShowBlock = 256;
__global__ __launch_bounds__(kShowBlock) void mcq_showdown_kernel(const uint8_t *__restrict__ hands, uint32_t n_tables,
                                                                  uint32_t n_players, const McqTables *__restrict__ g_tab,
                                                                  uint8_t *__restrict__ winner, uint8_t *__restrict__ wtype,
                                                                  uint32_t *__restrict__ keys, uint32_t *__restrict__ bad,
                                                                  uint32_t *__restrict__ done, volatile uint32_t *done_flag,
                                                                  uint32_t ticket) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[kShowBlock * 7 * 10];
    __shared__ __attribute__((aligned(16))) uint32_t tile_keys[kShowBlock * 10];
    __shared__ __attribute__((aligned(16))) uint8_t tile_win[kShowBlock], tile_type[kShowBlock];
    const uint32_t per_table = 7u * n_players, n_tiles = (n_tables + kShowBlock - 1u) / kShowBlock;
    for (uint32_t t0 = blockIdx.x; t0 < n_tiles; t0 += gridDim.x) {
        const uint32_t first = t0 * kShowBlock, cnt = min((uint32_t)kShowBlock, n_tables - first);
        const uint32_t bytes = cnt * per_table, vecs = (bytes + 15u) / 16u; /* the buffers are padded to 16 bytes */
        const uint4 *src = reinterpret_cast<const uint4 *>(hands + (size_t)first * per_table); /* 256 * 7 * P: a multiple of 16 */
        __syncthreads(); /* the previous tile has been read */
        for (uint32_t i = threadIdx.x; i < vecs; i += kShowBlock) reinterpret_cast<uint4 *>(tile)[i] = src[i];
        __syncthreads();
        if (threadIdx.x < cnt) {
            const uint8_t *h = tile + threadIdx.x * per_table;
            uint32_t best = 0, w = 0;
            bool ok = true;
            for (uint32_t p = 0; p < n_players; p++, h += 7) {
                uint64_t seen = 0;
                uint32_t c[7];
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    c[k] = h[k];
                    ok = ok && c[k] < 52u && !((seen >> (c[k] & 63u)) & 1ull);
                    seen |= 1ull << (c[k] & 63u);
                    c[k] = c[k] < 52u ? c[k] : 0u;
                }
                McqBoard b;
                b.clear();
#pragma unroll
                for (int k = 2; k < 7; k++) b.add(mcq_card(c[k]));
                McqHole hole;
                hole.set(mcq_card(c[0]), mcq_card(c[1]));
                McqFlushSel fs;
                fs.from_board(b);
                const uint32_t key = mcq_eval_key(b, fs, hole, g_tab->tf, g_tab->tops, g_tab->sd);
                tile_keys[threadIdx.x * n_players + p] = key;
                if (key > best) { best = key; w = p; } /* strict: the first of equal hands stays (hand_evaluator.py:23) */
            }
            tile_win[threadIdx.x] = (uint8_t)w;
            tile_type[threadIdx.x] = (uint8_t)mcq_key_type(best);
            if (!ok) *reinterpret_cast<volatile uint32_t *>(bad) = 1u; /* (a plain store: every writer writes the same word) */
        }
        __syncthreads();
        /* results: whole 16-byte words (the output buffers are padded to the tile) */
        if (threadIdx.x < kShowBlock / 16) {
            reinterpret_cast<uint4 *>(winner + first)[threadIdx.x] = reinterpret_cast<const uint4 *>(tile_win)[threadIdx.x];
            reinterpret_cast<uint4 *>(wtype + first)[threadIdx.x] = reinterpret_cast<const uint4 *>(tile_type)[threadIdx.x];
        }
        if (keys) {
            const uint32_t kv = (cnt * n_players + 3u) / 4u;
            uint4 *dst = reinterpret_cast<uint4 *>(keys + (size_t)first * n_players); /* 256 * P keys: a multiple of four */
            for (uint32_t i = threadIdx.x; i < kv; i += kShowBlock) dst[i] = reinterpret_cast<const uint4 *>(tile_keys)[i];
        }
    }
    if (ticket == 0u) return; /* not the last launch of the call */
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system(); /* this block's results (and `bad`) have reached the host's memory */
        bool last = true;
        if (gridDim.x > 1u) {
            last = atomicAdd(done, 1u) + 1u == gridDim.x;
            if (last) *done = 0;
        }
        if (last) *done_flag = ticket;
    }
}

// ---------------------------------------------------------------------------------------------- exact enumeration
// one query per launch, see mcq_exact.hpp.  Work unit = (table completion, slice of first
// opponent hands); a wave takes units wave, wave + n_waves, ...  Per-lane 64-bit sums by role: lane 0 total
// weight (-> runs), lane 2 strict wins, lane 3 ties, lane 4 + t hero's winning hand type t; one atomic each
// at the end.
template <bool TWO_OPP>
__global__ __launch_bounds__(TWO_OPP ? 384 : 1024) void mcq_exact_kernel(const McqExactJob *__restrict__ jobs, int law,
                                                                        mcq_result *__restrict__ rows,
                                                                        const McqTables *__restrict__ g_tab) {
    /* blockIdx.y = the job (one query); its first `grid` blocks work, the others leave at once */
    const McqExactJob job = jobs[blockIdx.y];
    if (blockIdx.x >= job.grid) return;
    const uint4 raw = make_uint4(job.rec[0], job.rec[1], job.rec[2], job.rec[3]);
    const uint32_t n_boards = job.n_boards, slices = job.slices;
    mcq_result *row = rows + job.row;
    constexpr uint32_t kWaves = TWO_OPP ? 6u : 16u; /* what fits beside the 97 KB of tables */
    __shared__ __attribute__((aligned(16))) LdsTablesEval tab; /* tf from global memory, as in the evaluation kernels */
    __shared__ uint16_t pair_xy[MCQ_EXACT_PAIRS + 2];
    __shared__ McqCard rem_card_all[kWaves][64];
    __shared__ uint32_t rem_pos_all[kWaves][64];
    __shared__ uint32_t keys_all[TWO_OPP ? kWaves : 1u][TWO_OPP ? MCQ_EXACT_PAIRS + 2 : 1u];
    __shared__ uint16_t rec_all[TWO_OPP ? kWaves : 1u][TWO_OPP ? MCQ_EXACT_PAIRS + 2 : 1u];
    load_tables(tab, g_tab);
    for (uint32_t i = threadIdx.x; i < MCQ_EXACT_PAIRS; i += blockDim.x) {
        uint32_t x, y;
        mcq_exact_pair_xy(i, x, y);
        pair_xy[i] = (uint16_t)(x | (y << 8));
    }
    __syncthreads();

    const McqQueryWords q = {raw.x, raw.y, raw.z, raw.w};
    McqExactQuery e;
    if (!mcq_exact_query(q, law, e)) return; /* the host has validated the query */
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves + wib);
    const uint32_t n_waves = job.grid * kWaves;
    McqCard *rem_card = rem_card_all[wib];
    uint32_t *rem_pos = rem_pos_all[wib];
    uint32_t *keys = TWO_OPP ? keys_all[wib] : nullptr;
    uint16_t *rec = TWO_OPP ? rec_all[wib] : nullptr;

    unsigned long long sum = 0;
    const uint64_t n_units = (uint64_t)n_boards * slices;
    for (uint64_t unit = wave; unit < n_units; unit += n_waves) {
        const uint32_t board = (uint32_t)(unit / slices), slice = (uint32_t)(unit % slices);
        uint32_t pos[5];
        mcq_exact_unrank(board, e.L, e.k, pos);
        McqExactBoard bd;
        mcq_exact_board(e, pos, tab.sel8, g_tab->tf, tab.tops, tab.sd, bd);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); /* the previous unit's reads are done (same wave) */
        if (lane < MCQ_EXACT_REM) {
            const uint32_t rp = mcq_exact_rem_pos(pos, lane);
            rem_pos[lane] = rp;
            rem_card[lane] = mcq_card(mcq_exact_card_at(e, rp, tab.sel8));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        McqExactAcc acc = {0, 0, 0};
        mcq_exact_pass_a(e, bd, lane, pair_xy, rem_card, rem_pos, g_tab->tf, tab.tops, tab.sd, keys, rec, acc);
        if (TWO_OPP) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            mcq_exact_pass_b(e, bd, lane, MCQ_EXACT_PAIRS * slice / slices, MCQ_EXACT_PAIRS * (slice + 1u) / slices,
                             pair_xy, keys, rec, acc);
        }
        const uint32_t win = wave_sum(acc.win), tie = wave_sum(acc.tie), tot = wave_sum(acc.tot);
        const uint32_t type = mcq_key_type(bd.hero_key);
        if (lane == 0u) sum += tot;
        if (lane == 2u) sum += win;
        if (lane == 3u) sum += tie;
        if (lane == 4u + type) sum += win + tie;
    }
    if (lane < 13u && sum != 0ull) atomicAdd(reinterpret_cast<unsigned long long *>(row) + lane, sum);
}

}  // namespace

// ---------------------------------------------------------------------------------------------- launchers
hipError_t mcq_launch_prep(const mcq_query *d_q, uint32_t n, mcq_result *d_res, uint64_t *d_prefix, uint32_t part,
                           uint32_t n_parts, uint32_t n_cu, uint32_t split_max, hipStream_t s) {
    if (n_parts == 0 || part >= n_parts) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mcq_prep_kernel, dim3(1), dim3(1024), 0, s, d_q, n, d_res, d_prefix, part, n_parts, n_cu, split_max);
    return hipGetLastError();
}

/* t0 / t1 (either may be null): events that take the kernel's own begin / end timestamps (hipExtLaunchKernel: read
 * from the dispatch packet's completion signal -- no marker packets in the queue around a short kernel) */
hipError_t mcq_launch_eval(int mode, const mcq_query *d_q, uint32_t n, const uint64_t *d_prefix, mcq_result *d_res,
                           uint64_t seed, uint64_t first_qid, const McqTables *d_luts, const uint8_t *d_draws,
                           const uint64_t *d_draw_off, uint32_t grid, uint32_t block, uint32_t split, uint32_t part,
                           uint32_t n_parts, hipStream_t s, hipEvent_t t0, hipEvent_t t1, uint32_t work_wpb) {
    if ((split > 4 && split != MCQ_SPLIT_FROM_PREP) || n_parts == 0 || part >= n_parts) return hipErrorInvalidValue;
    if (mode == MCQ_MODE_REPLAY_MT19937 && split > 2) return hipErrorInvalidValue; /* its lanes take four iterations at a time */
#define MCQ_LAUNCH_EVAL(M)                                                                                          \
    do {                                                                                                            \
        if (split)                                                                                                  \
            MCQ_LAUNCH_TIMED((mcq_eval_kernel<M, true>), grid, block, d_q, n,     \
                                  d_prefix, d_res, seed, first_qid, d_luts, d_draws, d_draw_off, split, part, n_parts, work_wpb); \
        else                                                                                                        \
            MCQ_LAUNCH_TIMED((mcq_eval_kernel<M, false>), grid, block, d_q, n,    \
                                  d_prefix, d_res, seed, first_qid, d_luts, d_draws, d_draw_off, 0u, part, n_parts, work_wpb); \
    } while (0)
    if (mode == MCQ_MODE_PHILOX) MCQ_LAUNCH_EVAL(MCQ_MODE_PHILOX);
    else if (mode == MCQ_INTERNAL_MODE_UNIFORM) MCQ_LAUNCH_EVAL(MCQ_INTERNAL_MODE_UNIFORM);
    else MCQ_LAUNCH_EVAL(MCQ_MODE_REPLAY_MT19937);
#undef MCQ_LAUNCH_EVAL
    return hipGetLastError();
}

hipError_t mcq_launch_eval_direct(int mode, const void *work_rec, const uint32_t *work_qi, uint32_t rounds, uint32_t merge,
                                  mcq_result *res, uint64_t seed, uint64_t first_qid, const McqTables *d_luts, uint32_t grid,
                                  uint32_t *d_done, uint32_t *done_flag, uint32_t ticket, hipStream_t s, hipEvent_t t0,
                                  hipEvent_t t1, const McqDirectKarg *karg, uint32_t dev_n, uint32_t dev_lg) {
    if (grid == 0 || rounds == 0) return hipErrorInvalidValue;
    if (karg && (uint64_t)grid * rounds * (kMaxBlock / 64) > MCQ_DIRECT_KARG_SLOTS) return hipErrorInvalidValue;
    if (karg && rounds > MCQ_DIRECT_STAGE_ROUNDS) return hipErrorInvalidValue; /* the kernel reads them in its first stage only */
    if (dev_n && (karg || work_qi || dev_lg > 4u)) return hipErrorInvalidValue;
    const uint4 *rec = static_cast<const uint4 *>(work_rec);
    static const McqDirectKarg none = {};
    McqDirectKarg dev = {}; /* queries in HBM: their number and the cut travel in the argument block */
    dev.qi[0] = dev_n;
    dev.qi[1] = dev_lg;
    const uint32_t use = dev_n ? 2u : karg ? 1u : 0u;
    const McqDirectKarg &ka = dev_n ? dev : karg ? *karg : none;
    if (mode == MCQ_INTERNAL_MODE_UNIFORM)
        MCQ_LAUNCH_TIMED((mcq_eval_direct_kernel<MCQ_INTERNAL_MODE_UNIFORM>), grid, kMaxBlock, rec,
                              work_qi, rounds, merge, res, seed, first_qid, d_luts, d_done, done_flag, ticket, use, ka);
    else
        MCQ_LAUNCH_TIMED((mcq_eval_direct_kernel<MCQ_MODE_PHILOX>), grid, kMaxBlock, rec, work_qi,
                              rounds, merge, res, seed, first_qid, d_luts, d_done, done_flag, ticket, use, ka);
    return hipGetLastError();
}

hipError_t mcq_launch_mt_parse(const mcq_query *d_q, uint32_t n, uint32_t seed32, uint8_t *d_draws, const uint64_t *d_draw_off,
                               mcq_result *d_res, uint32_t *d_counter, uint32_t n_cu, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t blocks = n; /* one pair of waves per query */
    if (blocks > 16u * n_cu) blocks = 16u * n_cu; /* what a CU holds at once: 16 work-groups */
    hipLaunchKernelGGL(mcq_mt_parse_kernel, dim3(blocks), dim3(kMtBlock), 0, s, d_q, n, seed32, d_draws, d_draw_off, d_res,
                       d_counter);
    return hipGetLastError();
}

uint64_t mcq_mtb_part_words(void) { return (uint64_t)(MCQ_MTB_MAX_SEG - 1u) * kMtbJumpSplit * MCQ_MT_N; }
hipError_t mcq_launch_mt_blocks(const mcq_query *d_q, uint32_t n, uint32_t seed32, const uint32_t *d_blk_off,
                                const uint32_t *d_grp_off, uint32_t max_blocks, uint32_t *d_raw, uint32_t *d_exits, void *d_entries,
                                uint32_t *d_gword, uint32_t *d_gits, void *d_gentry, uint32_t *d_ovf, uint8_t *d_draws,
                                const uint64_t *d_draw_off, mcq_result *d_res, uint32_t *d_part, hipStream_t s) {
    if (n == 0 || max_blocks == 0) return hipSuccess;
    constexpr uint32_t kWaves = kMtbBlock / 64;
    const uint32_t max_groups = (max_blocks + MCQ_MTB_GROUP - 1u) / MCQ_MTB_GROUP;
    const dim3 per_block((max_blocks + kWaves - 1) / kWaves, n), per_group((max_groups + kWaves - 1) / kWaves, n);
    if (!d_part) { /* no jumps (MCQ_MT_JUMP=0): one work-group per query makes all its blocks, one behind the other */
        hipLaunchKernelGGL(mcq_mtb_generate_kernel, dim3(1, n), dim3(kMtbGenBlock), 0, s, d_blk_off, seed32, d_raw, d_part, 0u,
                           max_blocks, 0u);
    } else {
        constexpr uint32_t kRound = MCQ_MTB_SEG * MCQ_MTB_MAX_SEG;
        for (uint32_t base = 0; base < max_blocks; base += kRound) {
            const uint32_t left = max_blocks - base, segs = left >= kRound ? MCQ_MTB_MAX_SEG : (left + MCQ_MTB_SEG - 1u) / MCQ_MTB_SEG;
            if (segs > 1u) {
                hipLaunchKernelGGL(mcq_mtb_generate_kernel, dim3(1, n), dim3(kMtbGenBlock), 0, s, d_blk_off, seed32, d_raw, d_part,
                                   base, kMtbJumpSpan, 1u);
                hipLaunchKernelGGL(mcq_mtb_jump_kernel, dim3((segs - 1u) * kMtbJumpSplit, n), dim3(kMtbJumpBlock), 0, s, d_blk_off,
                                   d_raw, d_part, base);
            }
            hipLaunchKernelGGL(mcq_mtb_generate_kernel, dim3(segs, n), dim3(kMtbGenBlock), 0, s, d_blk_off, seed32, d_raw, d_part,
                               base, (uint32_t)MCQ_MTB_SEG, 0u);
        }
    }
    hipLaunchKernelGGL(mcq_mtb_scan_kernel, per_block, dim3(kMtbBlock), 0, s, d_q, d_blk_off, d_raw, d_exits);
    hipLaunchKernelGGL(mcq_mtb_compose_kernel, per_group, dim3(kMtbBlock), 0, s, d_q, d_blk_off, d_grp_off, d_exits, d_gword, d_gits);
    hipLaunchKernelGGL(mcq_mtb_stitch_kernel, dim3(n), dim3(64), 0, s, d_q, d_blk_off, d_grp_off, d_gword, d_gits,
                       reinterpret_cast<McqMtbEntry *>(d_gentry), d_ovf, d_res);
    hipLaunchKernelGGL(mcq_mtb_expand_kernel, per_group, dim3(kMtbBlock), 0, s, d_q, d_blk_off, d_grp_off, d_exits,
                       reinterpret_cast<const McqMtbEntry *>(d_gentry), reinterpret_cast<McqMtbEntry *>(d_entries));
    const dim3 per_parse_block((max_blocks + kMtbParseBlock / 64 - 1) / (kMtbParseBlock / 64), n);
    hipLaunchKernelGGL(mcq_mtb_parse_kernel, per_parse_block, dim3(kMtbParseBlock), 0, s, d_q, d_blk_off, d_raw,
                       reinterpret_cast<const McqMtbEntry *>(d_entries), d_ovf, d_draws, d_draw_off, d_res);
    return hipGetLastError();
}

hipError_t mcq_launch_mt_parse_ext(const mcq_query *d_q, const mcq_query_ext *d_ext, uint32_t n, uint32_t seed32, uint8_t *d_draws,
                                   const uint64_t *d_draw_off, mcq_result *d_res, uint32_t *d_counter, uint32_t n_cu, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t blocks = (n + kMtExtBlock / 64 - 1) / (kMtExtBlock / 64);
    if (blocks > 5u * n_cu) blocks = 5u * n_cu; /* what fits a CU at once: 27 KB of LDS per block */
    hipLaunchKernelGGL(mcq_mt_parse_ext_kernel, dim3(blocks), dim3(kMtExtBlock), 0, s, d_q, d_ext, n, seed32, d_draws, d_draw_off,
                       d_res, d_counter);
    return hipGetLastError();
}

hipError_t mcq_launch_publish(mcq_result *d_rows, mcq_result *h_rows_dev, uint64_t n_rows, uint32_t *d_done,
                              uint32_t *done_flag, uint32_t ticket, hipStream_t s) {
    static_assert(sizeof(mcq_result) % 16 == 8, "13 x 8 bytes");
    if (n_rows == 0 || (n_rows & 1ull)) return hipErrorInvalidValue; /* whole 16-byte words: the caller rounds the rows up */
    const uint64_t words = n_rows * sizeof(mcq_result) / 16u;
    uint64_t blocks = (words + 1023u) / 1024u; /* four words per thread */
    if (blocks > 64u) blocks = 64u;
    hipLaunchKernelGGL(mcq_publish_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, reinterpret_cast<ulonglong2 *>(d_rows),
                       reinterpret_cast<ulonglong2 *>(h_rows_dev), words, d_done, done_flag, ticket);
    return hipGetLastError();
}

hipError_t mcq_launch_add_u64(uint64_t *d_dst, const uint64_t *d_src, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n / 2 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mcq_add_u64_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, d_dst, d_src, n);
    return hipGetLastError();
}

hipError_t mcq_launch_prep_ext(const mcq_query *d_q, const mcq_query_ext *d_ext, uint32_t n, int mode, mcq_result *d_res,
                               uint64_t *d_prefix, hipStream_t s) {
    hipLaunchKernelGGL(mcq_prep_ext_kernel, dim3(1), dim3(1024), 0, s, d_q, d_ext, n, mode, d_res, d_prefix);
    return hipGetLastError();
}

hipError_t mcq_launch_ext_lists(const mcq_query *d_q, const mcq_query_ext *d_ext, uint32_t n, uint32_t lists_stride,
                                uint16_t *d_lists, uint32_t *d_cnts, hipStream_t s) {
    if (n == 0 || lists_stride == 0) return hipSuccess;
    hipLaunchKernelGGL(mcq_ext_lists_kernel, dim3(n), dim3(kListBlock), 0, s, d_q, d_ext, lists_stride, d_lists, d_cnts);
    return hipGetLastError();
}

hipError_t mcq_launch_eval_ext(int mode, const mcq_query *d_q, const mcq_query_ext *d_ext, uint32_t n,
                               const uint64_t *d_prefix, mcq_result *d_res, uint64_t seed, uint64_t first_qid,
                               const McqTables *d_luts, const uint8_t *d_draws, const uint64_t *d_draw_off,
                               const uint16_t *d_lists, const uint32_t *d_cnts, uint32_t lists_stride, uint32_t grid,
                               uint32_t block, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    if (mode == MCQ_MODE_PHILOX)
        MCQ_LAUNCH_TIMED(mcq_eval_ext_kernel<MCQ_MODE_PHILOX>, grid, block, d_q, d_ext, n,
                              d_prefix, d_res, seed, first_qid, d_luts, d_draws, d_draw_off, d_lists, d_cnts, lists_stride);
    else
        MCQ_LAUNCH_TIMED(mcq_eval_ext_kernel<MCQ_MODE_REPLAY_MT19937>, grid, block, d_q, d_ext,
                              n, d_prefix, d_res, seed, first_qid, d_luts, d_draws, d_draw_off, d_lists, d_cnts, lists_stride);
    return hipGetLastError();
}

hipError_t mcq_launch_eval_ext_small(const McqExtSmallKarg *karg, uint32_t n, /* blocks */ mcq_result *h_res_dev, uint64_t seed,
                                     uint64_t first_qid, const McqTables *d_luts, uint32_t *d_done, uint32_t *done_flag,
                                     uint32_t ticket, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    if (n == 0 || n > MCQ_EXT_SMALL_BLOCKS) return hipErrorInvalidValue;
    MCQ_LAUNCH_TIMED(mcq_eval_ext_small_kernel, n, kExtBlock, *karg, h_res_dev, seed, first_qid, d_luts, d_done, done_flag, ticket);
    return hipGetLastError();
}

hipError_t mcq_launch_showdown(const uint8_t *hands, uint32_t n_tables, uint32_t n_players, const McqTables *d_luts,
                               uint8_t *winner, uint8_t *wtype, uint32_t *keys, uint32_t *bad, uint32_t *d_done,
                               uint32_t *done_flag, uint32_t ticket, uint32_t n_cu, hipStream_t s) {
    if (n_tables == 0 || n_players < 1 || n_players > 10) return hipErrorInvalidValue;
    uint32_t grid = (n_tables + kShowBlock - 1) / kShowBlock;
    if (grid > 5u * n_cu) grid = 5u * n_cu; /* 30 KB of LDS per block: five blocks per CU */
    hipLaunchKernelGGL(mcq_showdown_kernel, dim3(grid), dim3(kShowBlock), 0, s, hands, n_tables, n_players, d_luts, winner,
                       wtype, keys, bad, d_done, done_flag, ticket);
    return hipGetLastError();
}

void mcq_exact_plan(const mcq_query *q, uint32_t row, uint32_t n_cu, McqExactJob *job) {
    __builtin_memcpy(job->rec, q, 16);
    const uint32_t L = 50u - q->n_board, k = 5u - q->n_board;
    job->n_boards = mcq_exact_binom(L, k);
    job->row = row;
    if (q->n_players == 3) {
        /* few completions (turn, river): cut the first-opponent loop so that every wave has work */
        uint32_t slices = 1;
        while (slices < 64u && (uint64_t)job->n_boards * slices < 6ull * n_cu * 4ull) slices *= 2u;
        const uint64_t units = (uint64_t)job->n_boards * slices;
        job->slices = slices;
        job->grid = (uint32_t)((units + 5u) / 6u < n_cu ? (units + 5u) / 6u : n_cu);
    } else {
        job->slices = 1u;
        job->grid = (job->n_boards + 15u) / 16u < n_cu ? (job->n_boards + 15u) / 16u : n_cu;
    }
}

hipError_t mcq_launch_exact(const McqExactJob *d_jobs, uint32_t n_jobs, uint32_t max_grid, bool two_opp, int law,
                            mcq_result *d_rows, const McqTables *d_luts, hipStream_t s) {
    if (n_jobs == 0) return hipSuccess;
    if (n_jobs > 65535u || max_grid == 0) return hipErrorInvalidValue;
    if (two_opp)
        hipLaunchKernelGGL(mcq_exact_kernel<true>, dim3(max_grid, n_jobs), dim3(384), 0, s, d_jobs, law, d_rows, d_luts);
    else
        hipLaunchKernelGGL(mcq_exact_kernel<false>, dim3(max_grid, n_jobs), dim3(1024), 0, s, d_jobs, law, d_rows, d_luts);
    return hipGetLastError();
}

hipError_t mcq_eval_occupancy(int mode, int block, int *blocks_per_cu) {
    if (mode == MCQ_MODE_PHILOX)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, mcq_eval_kernel<MCQ_MODE_PHILOX, false>, block, 0);
    if (mode == MCQ_INTERNAL_MODE_UNIFORM)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, mcq_eval_kernel<MCQ_INTERNAL_MODE_UNIFORM, false>,
                                                            block, 0);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, mcq_eval_kernel<MCQ_MODE_REPLAY_MT19937, false>, block, 0);
}
