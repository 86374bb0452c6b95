// mcq_device.hpp -- per-lane arithmetic of the equity kernels: RNG front ends, search-free dealing from the
// ordered deck, the branch-free 7-card ranking key, and one Monte-Carlo iteration.
//
// Everything here is written against a few tiny primitives (popcount, count-leading-zeros, 32x32 high
// multiply, byte splat) so that the very same source is compiled for gfx950 by hipcc (the product) and, by
// tests/hostsim only, for the host compiler to unit-test the lane arithmetic where no GPU exists.
//
// What it reproduces (reference paths relative to /root/reference):
//   tools/montecarlo_python.py:121-189  dealing order and index semantics (see mcq_iteration)
//   tools/hand_evaluator.py:27-119      _calc_score ordering incl. its quirks (see mcq_eval_key)
//   tools/hand_evaluator.py:20-24       ties go to the first hand = hero
//
// Cost model, measured on gfx950 with tools/ubench (profiles/r01_ubench*.txt): in any instruction stream that
// contains even one non-trivial VALU op per 64 (shift-left, compare, select, popcount, multiply, any 3-operand
// form ...) EVERY wave64 VALU instruction issues in ~4.0-4.3 cycles; the 2.3-cycle rate exists only for pure
// add/and/or/xor/lshr streams.  A random LDS lookup costs ~8.5 cycles of the separate LDS pipe.  So the kernel
// is tuned for INSTRUCTION COUNT: fused 3-operand forms (and_or, or3, bitop3, lshl_or, add3, max3, sad_u8),
// selects where they are shorter than mask arithmetic, and LDS tables (indexed by byte offset, masks kept in
// the pre-shifted "x4 domain") wherever a lookup replaces more than one instruction.
#pragma once
#include <stdint.h>

#include "../../include/mcq.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MCQ_HD __host__ __device__ __forceinline__
#define MCQ_HDM __host__ __device__ __forceinline__ /* member functions */
#else
#define MCQ_HD static inline
#define MCQ_HDM inline
#endif

#define MCQ_STREAM_ITERS 16u /* iterations per RNG stream (MCQ-CTR v5) */
#define MCQ_WAVE 64u
#define MCQ_TASK_ITERS (MCQ_STREAM_ITERS * MCQ_WAVE) /* iterations per wave task */
#define MCQ_MAX_OPP 9

// ------------------------------------------------------------------------------------------ primitives
MCQ_HD uint32_t mcq_popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}
MCQ_HD uint32_t mcq_clz(uint32_t x) { /* x != 0 for a meaningful result */
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clz((int)x);
#else
    return x ? (uint32_t)__builtin_clz(x) : 32u;
#endif
}
MCQ_HD uint32_t mcq_mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
// mulhi32(a, b) + 128: draw indices travel as r | 0x80 (r < 64), the form the byte-SWAR hole scan wants, so the
// bias comes for free with the multiply (one v_mad_u64_u32)
MCQ_HD uint32_t mcq_mulhi_p128(uint32_t a, uint32_t b, uint64_t bias /* 128 << 32, see mcq_p128_bias */) {
    return (uint32_t)(((uint64_t)a * b + bias) >> 32);
}
// Both halves of the same product: the high word (biased as above) is returned, the low word -- the fraction that
// feeds the next draw -- goes to `lo`.  The product is made opaque as ONE 64-bit value: otherwise the compiler
// computes the low word a second time (v_mul_lo_u32 beside the v_mad_u64_u32; 32-bit multiplies issue at a quarter
// of the plain VALU rate).
MCQ_HD uint32_t mcq_mulhilo_p128(uint32_t a, uint32_t b, uint64_t bias, uint32_t &lo) {
    uint64_t p = (uint64_t)a * b + bias;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(p));
#endif
    lo = (uint32_t)p;
    uint32_t hi = (uint32_t)(p >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(hi)); /* a 32-bit value from here on: else comparing two of them becomes a 64-bit compare + moves */
#endif
    return hi;
}
// The bias as a value the optimiser cannot see through (device: pinned in a VGPR pair): left to itself the
// compiler re-creates the constant with a v_mov_b64 in front of every multiply (fifteen per 6-max iteration).
MCQ_HD uint64_t mcq_p128_bias() {
    uint64_t x = 128ull << 32;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}
MCQ_HD bool mcq_any(bool pred) { /* true if the predicate holds in any active lane of the wave */
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(pred) != 0;
#else
    return pred;
#endif
}
MCQ_HD uint32_t mcq_rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
// Hide a value from the optimiser (device: pins it in a VGPR): keeps the compiler from re-associating a sum into
// separately shifted parts, so that a table address stays ONE shift-add.
MCQ_HD uint32_t mcq_opaque(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}
// The same for a wave-uniform value (device: pins it in an SGPR).  A condition computed from it is evaluated
// where it is used (one scalar compare) instead of being carried across blocks as a lane mask, which the
// compiler rebuilds with VALU instructions.
MCQ_HD uint32_t mcq_opaque_uniform(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(x));
#endif
    return x;
}
MCQ_HD uint32_t mcq_bfe(uint32_t x, uint32_t off, uint32_t width) { /* (x >> off) & ((1 << width) - 1), off + width <= 32 */
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(x, off, width);
#else
    return (x >> off) & (width >= 32 ? 0xFFFFFFFFu : ((1u << width) - 1u));
#endif
}
MCQ_HD uint32_t mcq_sad_u8(uint32_t bytes, uint32_t acc) { /* acc + sum of the four bytes */
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sad_u8(bytes, 0u, acc);
#else
    return acc + (bytes & 0xFFu) + ((bytes >> 8) & 0xFFu) + ((bytes >> 16) & 0xFFu) + (bytes >> 24);
#endif
}
MCQ_HD uint32_t mcq_splat_byte(uint32_t x) { /* x < 256 -> x in all four bytes */
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, x, 0u); /* v_perm_b32: every selector byte 0 = byte 0 of x */
#else
    return x * 0x01010101u;
#endif
}

// ------------------------------------------------------------------------------------------ ranking keys
// Internal key = code << 28 | field, unsigned order == Python's order of _calc_score's (score, card_ranks).
// Codes leave a gap at 5; by_type index = code - (code >= 6).  Rank masks live in the "x4 domain" (bit r+2 =
// rank r) so that they are byte offsets into 32-bit LDS tables without a (slow) left shift; a field is
// [13-bit mask << 15][13-bit mask << 2].
#define MCQ_KEY_SHIFT 28
enum { MCQ_C_HIGH = 0, MCQ_C_PAIR = 1, MCQ_C_TWOPAIR = 2, MCQ_C_TRIPS = 3, MCQ_C_STRAIGHT = 4, MCQ_C_FLUSH = 6,
       MCQ_C_FULL = 7, MCQ_C_QUADS = 8, MCQ_C_SF = 9, MCQ_N_CODES = 10 };
MCQ_HD uint32_t mcq_code_to_type(uint32_t code) { return code - (code >= 6u ? 1u : 0u); } /* by_type index */
MCQ_HD uint32_t mcq_key_type(uint32_t key) { return mcq_code_to_type(key >> MCQ_KEY_SHIFT); }

// bit i of the result set <=> ranks i-1 .. i+3 all present (rank -1 = ace playing low)
MCQ_HD uint32_t mcq_straight_runs(uint32_t m) {
    uint32_t m2 = (m << 1) | (m >> 12);
    uint32_t r1 = m2 & (m2 >> 1);
    uint32_t r2 = r1 & (r1 >> 2);
    return r2 & (m2 >> 4);
}

// ------------------------------------------------------------------------------------------ lookup tables (LDS)
// sel8: only for laying out the per-query base deck (once per wave task).
// All mask-indexed tables are addressed with x4-domain masks (m4 = m << 2):
//   tops[m] (u32, byte offset m4):  top two set bits of m (x4 domain, 0 if fewer than two) | top bit << 16
//   sd[m]   (u32, byte offset m4):  (straight ? 0x80 | top position 1..10 : 0) << 23, i.e. the complete
//           Straight key (0 without a straight)
//   sd[8192 + m] ("kc", byte offset 32768 + m4, folded into the read instruction's offset field): the low part of
//           the HighCard / Pair / ThreeOfAKind key when m is the mask of the ranks held exactly once: m without
//           its two lowest set bits (the kickers that count) | the type code << 28, which the number of kickers
//           gives away for seven cards: 7 -> HighCard, 5 -> Pair, 4 -> ThreeOfAKind; any other count belongs to a
//           higher type whose own candidate key wins (code 0 here)
//   tf[m]   (u32, byte offset m4):  complete key of the SUIT mask m: StraightFlush (all ranks of the suit
//           plus the -1 slot when it holds the ace, hand_evaluator.py:71-80,93), Flush (top five, :98-100), or 0
//           when popcount(m) < 5
struct McqTables { /* order matters on the device: the first 64 KB are reachable through the 16-bit offset field of
                     the LDS read, i.e. the lookups whose index comes straight out of a logic instruction (tops,
                     sd); the index of kc is formed by an xor that takes the table base along.  The evaluation
                     kernels keep tops, sd | kc (the first 96 KB) and sel8 in LDS and read tf from GLOBAL memory
                     through the vector L1: the LDS pipe is their co-limiter and the flush lookup, whose index is
                     the sparsest, is the one that costs least there (measured) */
    uint32_t tops[8192];
    uint32_t sd[16384]; /* [0, 8192) sd, [8192, 16384) kc */
    uint32_t tf[16384]; /* [0, 8192) tf, [8192, 16384) a second copy of tops for the lookup that goes through
                           global memory with tf (tops[ge3], see mcq_eval_key) */
    uint32_t sel8[256];
};
#define MCQ_KC_BYTE_OFFSET 32768u /* kc relative to sd */
#define MCQ_TF_BYTE_OFFSET 98304u /* tf relative to tops = size of the part the evaluation kernels keep in LDS */
static_assert(__builtin_offsetof(McqTables, tops) == 0 && __builtin_offsetof(McqTables, tf) == MCQ_TF_BYTE_OFFSET,
              "McqTables layout");

static inline void mcq_fill_tables(McqTables *t) {
    for (uint32_t v = 0; v < 256; v++) {
        uint32_t e = 0, j = 0;
        for (uint32_t b = 0; b < 8; b++)
            if (v >> b & 1) e |= b << (3 * j++);
        t->sel8[v] = e;
    }
    for (uint32_t m = 0; m < 8192; m++) {
        uint32_t runs = mcq_straight_runs(m);
        uint32_t st = runs ? (0x80u | (32u - (uint32_t)__builtin_clz(runs))) : 0;
        uint32_t n = (uint32_t)__builtin_popcount(m);
        uint32_t d2 = m & (m - 1);
        d2 &= d2 - 1; /* m == 0 stays 0 */
        t->sd[m] = st << 23;
        t->sd[8192 + m] = (d2 << 2) | ((n == 5 ? (uint32_t)MCQ_C_PAIR : n == 4 ? (uint32_t)MCQ_C_TRIPS : 0u) << MCQ_KEY_SHIFT);
        uint32_t hi1 = m ? 0x80000000u >> __builtin_clz(m) : 0, hi2 = 0;
        if (n >= 2) hi2 = hi1 | (0x80000000u >> __builtin_clz(m ^ hi1));
        t->tops[m] = (hi2 << 2) | (hi1 << 18);
        t->tf[8192 + m] = t->tops[m];
        uint32_t key = 0;
        if (n >= 5) {
            if (runs) {
                key = ((uint32_t)MCQ_C_SF << MCQ_KEY_SHIFT) | (m << 1) | (m >> 12);
            } else {
                uint32_t f = m;
                for (uint32_t k = n; k > 5; k--) f &= f - 1;
                key = ((uint32_t)MCQ_C_FLUSH << MCQ_KEY_SHIFT) | f;
            }
        }
        t->tf[m] = key;
    }
}

// ------------------------------------------------------------------------------------------ RNG: MCQ-CTR v5
MCQ_HD void mcq_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t l0, l1; /* both halves of a product from ONE v_mad_u64_u32 (not v_mul_hi_u32 + v_mul_lo_u32) */
        const uint32_t h0 = mcq_mulhilo_p128(0xD2511F53u, c0, 0ull, l0);
        const uint32_t h1 = mcq_mulhilo_p128(0xCD9E8D57u, c2, 0ull, l1);
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct McqXoshiro { /* xoshiro128++ (Blackman & Vigna): the table driver's per-table generator (mcq_tables.cpp) */
    uint32_t s0, s1, s2, s3;
    MCQ_HDM void seed(uint64_t seed, uint64_t qid, uint32_t stream) {
        uint32_t o[4];
        mcq_philox4x32_10((uint32_t)qid, (uint32_t)(qid >> 32), stream, 0x4D435131u, (uint32_t)seed,
                          (uint32_t)(seed >> 32), o);
        s0 = o[0]; s1 = o[1]; s2 = o[2]; s3 = o[3];
        if ((s0 | s1 | s2 | s3) == 0) s0 = 1;
    }
    MCQ_HDM uint32_t next() {
#ifdef MCQ_ABLATE_RNG /* diagnostic timing build: wrong results */
        s0 += 0x9E3779B9u;
        return s0;
#endif
        uint32_t result = mcq_rotl(s0 + s3, 7) + s0;
        uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t;
        s3 = mcq_rotl(s3, 11);
        return result;
    }
};

// The production mode's per-stream generator (MCQ-CTR v5): MWC64X -- David B. Thomas' multiply-with-carry generator
// for GPUs: (x, c) -> (lo, hi) of A * x + c, A = 4294883355, output x ^ c, period ~2^63, passes TestU01 BigCrush.
// ONE v_mad_u64_u32, a move of the carry into the addend pair and the output xor: three instructions per word against
// seven of v4's jsf32 (measured on the headline workload: 548 -> 516 VALU instructions per wave-iteration, 6.01 ->
// 5.73 ms).  Its two state words come from the Philox block of (seed, query id, stream).
struct McqMwc64x {
    uint32_t x, c;
    MCQ_HDM void seed(uint64_t seed, uint64_t qid, uint32_t stream) {
        uint32_t o[4];
        mcq_philox4x32_10((uint32_t)qid, (uint32_t)(qid >> 32), stream, 0x4D435131u, (uint32_t)seed,
                          (uint32_t)(seed >> 32), o);
        x = o[0];
        c = o[1] >> 1; /* carry < A; (2^32 - 1, A - 1), the other fixed point, cannot be seeded */
        if ((x | c) == 0) x = 0xf1ea5eedu; /* (0, 0) is a fixed point */
    }
    MCQ_HDM uint32_t next() {
#ifdef MCQ_ABLATE_RNG /* diagnostic timing build: wrong results */
        x += 0x9E3779B9u;
        return x;
#endif
        const uint32_t r = x ^ c;
        const uint64_t t = (uint64_t)4294883355u * x + c;
        x = (uint32_t)t;
        c = (uint32_t)(t >> 32);
        return r;
    }
};
typedef McqMwc64x McqStreamRng;

// Draw policy of the production mode, "MCQ-CTR v5": the reference's dealing law without its re-draw loop.
//   Opponent pair on a deck of length L from ONE word u, d = L - 1:  a = mulhi32(u, d), c = mulhi32(u * d mod 2^32, d)
//   -- (a, c) is uniform on [0, d)^2 up to d^2 / 2^32 -- and (r1, r2) = (a, c) if a != c else (d, a).  That is a
//   bijection from [0, d)^2 onto the pairs the reference accepts (r1 in [0,L), r2 in [0,L-1), r1 != r2;
//   montecarlo_python.py:167-176), so every accepted pair is as likely as after the reference's rejection loop,
//   with no divergent loop on the GPU.  No attempt is ever rejected, hence `passes` = one per opponent per
//   iteration (added by the caller).
//   Table cards, two per word: even draw: u = next(), idx = mulhi32(u, n), w = u * n; odd draw: idx = mulhi32(w, n)
//   (n = deck length - 1: never the last card, montecarlo_python.py:188).  Bias of either <= 2401 / 2^32.
// UNIFORM = true is the opt-in unbiased law (what montecarlo_cython.pyx:188 and
// Montecarlo.cpp:296-312 intend): a = mulhi32(u, L), c = mulhi32(u * L, d), (r1, r2) = (a, c) -- every ordered pair of
// distinct cards -- and table draws over all n = deck length cards.
// All draws are returned as r | 0x80 (see mcq_mulhi_p128).
template <bool UNIFORM>
struct McqCtrDrawsT {
    static constexpr uint32_t kTableShort = UNIFORM ? 0u : 1u; /* table draw range = deck length - kTableShort */
    McqStreamRng rng;
    uint32_t w;
    uint64_t bias; /* mcq_p128_bias(), set once per stream (start) */
    MCQ_HDM void start(uint64_t seed, uint64_t qid, uint32_t stream) {
        w = 0;
        bias = mcq_p128_bias();
        rng.seed(seed, qid, stream);
    }
    template <int P>
    MCQ_HDM void pair(uint32_t L, uint32_t &r1, uint32_t &r2) {
        const uint32_t dd = L - 1u, m1 = UNIFORM ? L : dd;
        const uint32_t u = rng.next();
        uint32_t frac; /* u * m1 mod 2^32 */
        const uint32_t a = mcq_mulhilo_p128(u, m1, bias, frac); /* (opaque: else a == c becomes a 64-bit compare + moves) */
        const uint32_t c = mcq_mulhi_p128(frac, dd, bias);
        r1 = (!UNIFORM && a == c) ? dd + 128u : a;
        r2 = c;
    }
    template <int K>
    MCQ_HDM uint32_t table(uint32_t n) {
        if ((K & 1) == 0) {
            const uint32_t u = rng.next();
            return mcq_mulhilo_p128(u, n, bias, w); /* w = u * n mod 2^32 */
        }
        return mcq_mulhi_p128(w, n, bias);
    }
};
typedef McqCtrDrawsT<false> McqCtrDraws;
typedef McqCtrDrawsT<true> McqCtrDrawsUniform;

// Draw policy of the parity mode: the stream walk (mcq_mt.hpp on the device, mcq_replay.hpp on the host) has already
// turned the MT19937 stream into the accepted draw values (one byte each, r | 0x80, draw-major:
// draws[d * stride + iteration]); rejected pairs only count in `passes`, which the walk supplies.
struct McqReplayDraws { /* one iteration, byte by byte */
    static constexpr uint32_t kTableShort = 1u;
    const uint8_t *p; /* &draws[iteration] */
    uint64_t stride;
    template <int P>
    MCQ_HDM void pair(uint32_t, uint32_t &r1, uint32_t &r2) {
        r1 = p[0];
        r2 = p[stride];
        p += 2 * stride;
    }
    template <int K>
    MCQ_HDM uint32_t table(uint32_t) {
        uint32_t v = p[0];
        p += stride;
        return v;
    }
};
// The same for FOUR consecutive iterations of a lane: one 32-bit load per draw row brings the four iterations' bytes
// (a wave reads 256 contiguous bytes per row instead of 64: a quarter of the vector-memory instructions and no
// 64-bit address arithmetic per byte); iteration `k` of the four takes byte k of every word.
MCQ_HD uint32_t mcq_ld32_bytes(const uint8_t *p) { /* device: p is 4-aligned (rows start at multiples of 64) */
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const uint32_t *>(p);
#else
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
#endif
}
struct McqReplayDraws4 {
    static constexpr uint32_t kTableShort = 1u;
    uint32_t ow[2 * MCQ_MAX_OPP], tw[5];
    uint32_t sh; /* 8 * k */
    MCQ_HDM void load(const uint8_t *first /* &draws[4-aligned iteration] */, uint64_t stride, uint32_t n_opp, uint32_t n_deal) {
#pragma unroll
        for (int P = 0; P < MCQ_MAX_OPP; P++)
            if ((uint32_t)P < n_opp) {
                ow[2 * P] = mcq_ld32_bytes(first + (uint64_t)(2 * P) * stride);
                ow[2 * P + 1] = mcq_ld32_bytes(first + (uint64_t)(2 * P + 1) * stride);
            }
        const uint8_t *t = first + (uint64_t)(2u * n_opp) * stride;
#pragma unroll
        for (int K = 0; K < 5; K++)
            if ((uint32_t)K < n_deal) tw[K] = mcq_ld32_bytes(t + (uint64_t)K * stride);
    }
    template <int P>
    MCQ_HDM void pair(uint32_t, uint32_t &r1, uint32_t &r2) {
        r1 = mcq_bfe(ow[2 * P], sh, 8);
        r2 = mcq_bfe(ow[2 * P + 1], sh, 8);
    }
    template <int K>
    MCQ_HDM uint32_t table(uint32_t) { return mcq_bfe(tw[K], sh, 8); }
};

// ------------------------------------------------------------------------------------------ base deck
// k-th set bit of the 52-bit deck mask (card-id order = the reference's list order), cleared afterwards.
// Only used to lay out the per-query base deck once per wave task; the per-draw work is search-free (below).
MCQ_HD uint32_t mcq_select_pop(uint32_t &dlo, uint32_t &dhi, uint32_t k, const uint32_t *sel8) {
    uint32_t c = mcq_popc(dlo);
    bool up = k >= c;
    uint32_t w = up ? dhi : dlo;
    k = up ? k - c : k;
    uint32_t base = up ? 32u : 0u;
    c = mcq_popc(w & 0xFFFFu);
    bool u = k >= c;
    w = u ? w >> 16 : w;
    k = u ? k - c : k;
    base += u ? 16u : 0u;
    c = mcq_popc(w & 0xFFu);
    u = k >= c;
    w = u ? w >> 8 : w;
    k = u ? k - c : k;
    base += u ? 8u : 0u;
    uint32_t e = sel8[w & 0xFFu];
    uint32_t pos = base + ((e >> (3 * (k & 7u))) & 7u);
    uint32_t bit = 1u << (pos & 31u);
    dlo &= up ? 0xFFFFFFFFu : ~bit;
    dhi &= up ? ~bit : 0xFFFFFFFFu;
    return pos;
}

// One card as the evaluator wants it (16 bytes: one ds_read_b128 per dealt card).
struct __attribute__((aligned(16))) McqCard {
    uint32_t rb;  /* 4 << rank (x4 domain) */
    uint32_t cnt; /* 1 << 4*suit: packed per-suit counters */
    uint32_t los; /* suit-major bit, pre-shifted left by 2 (tf[] byte offset): clubs bits 2..14, diamonds 18..30 */
    uint32_t his; /* hearts bits 2..14, spades bits 18..30 */
};

MCQ_HD McqCard mcq_card(uint32_t c) { /* c < 52 */
    const uint32_t rank = c >> 2, suit = c & 3u;
    McqCard e;
    e.rb = 4u << rank;
    e.cnt = 1u << (4u * suit);
    const uint32_t bit = 4u << (rank + 16u * (suit & 1u));
    e.los = suit < 2 ? bit : 0u;
    e.his = suit < 2 ? 0u : bit;
    return e;
}

// ------------------------------------------------------------------------------------------ evaluator
// A hand is (table cards) + (two hole cards).  The table part is accumulated once per iteration:
//   any/ge2/ge3/eq4: ranks present at least once / twice / three times / four times; cnt: packed suit counters;
//   los/his: suit-major masks (pre-shifted).  Adding a card with rank bit r: eq4 |= ge3 & r; ge3 |= ge2 & r; ...
struct McqBoard {
    uint32_t any, ge2, ge3, eq4, cnt, los, his;
    MCQ_HDM void clear() { any = ge2 = ge3 = eq4 = cnt = los = his = 0; }
    MCQ_HDM void add(const McqCard &c) {
        eq4 |= ge3 & c.rb;
        ge3 |= ge2 & c.rb;
        ge2 |= any & c.rb;
        any |= c.rb;
        cnt += c.cnt;
        los |= c.los;
        his |= c.his;
    }
};

struct McqHole { /* two hole cards: B = r1 | r2, P = r1 & r2 (pocket pair) */
    uint32_t B, P, los, his;
    MCQ_HDM void set(const McqCard &a, const McqCard &b) {
        B = a.rb | b.rb;
        P = a.rb & b.rb;
        los = a.los | b.los;
        his = a.his | b.his;
    }
};

// Only a suit with at least three table cards can make a flush, and five table cards hold at most one such
// suit.  use_hi / sh locate that suit's 16-bit field in the (los, his) pairs, bfl4 is the table's mask in it.
// Without such a suit the fields of clubs are taken: table + hole then hold fewer than five bits and tf[] = 0.
struct McqFlushSel {
    bool use_hi;
    uint32_t sh, bfl4;
    MCQ_HDM void from_board(const McqBoard &b) {
        const uint32_t f = (b.cnt + 0x5555u) & 0x8888u; /* bit 4s+3 <=> suit s has >= 3 table cards */
        use_hi = (f & 0x8800u) != 0;                     /* hearts or spades */
        sh = (f & 0x8080u) != 0 ? 16u : 0u;              /* diamonds or spades: upper half-word */
        bfl4 = mcq_bfe(use_hi ? b.his : b.los, sh, 16);
    }
};

// Table lookups by BYTE offset (x4-domain masks; never shifted left here).
MCQ_HD uint32_t mcq_ld_u32(const uint32_t *t, uint32_t byte_off) {
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(t) + byte_off);
}

// Key of table + hole.  Branch-free: every family of hand types yields a candidate key that is 0 when the
// family does not apply or carries a lower code than the true type; the key is their maximum.
//   F1  HighCard / Pair / ThreeOfAKind: (pair or trips rank) then the kickers = all other ranks minus the two
//       lowest (hand_evaluator.py:104-106, 110-115)
//   F2  TwoPair (two best pairs, kicker = best of the rest incl. a third pair, :39-42, 107-109) and
//       FullHouse (trips, best remaining pair or second trips, :36-38)
//   straight (top rank decides, wheel lowest, :52-58), flush / straight flush (table tf), and
//   FoufOfAKind = the two highest distinct ranks of all seven cards (:43-46).
// All masks below are x4-domain.
MCQ_HD uint32_t mcq_eval_key(const McqBoard &b, const McqFlushSel &fs, const McqHole &h, const uint32_t *tf,
                             const uint32_t *tops, const uint32_t *sd) {
#ifdef MCQ_ABLATE_EVAL /* diagnostic timing build: wrong results */
    return (b.any ^ h.B ^ (h.los >> 3)) | (1u << MCQ_KEY_SHIFT);
#endif
    const uint32_t any = b.any | h.B;
    const uint32_t ge2 = b.ge2 | (b.any & h.B) | h.P;
    const uint32_t ge3 = b.ge3 | (b.ge2 & h.B) | (b.any & h.P);
    const uint32_t eq4 = b.eq4 | (b.ge3 & h.B) | (b.ge2 & h.P);

    /* lookups first: their latency overlaps the arithmetic below */
    const uint32_t e_ge2 = mcq_ld_u32(tops, ge2);
    /* tops[ge3] from the copy of tops behind tf: in the kernels tf is the GLOBAL image, and the vector-memory
     * path takes this second sparse lookup off the LDS pipe as well (ge3 is zero for 19 hands in 20: one cache
     * line); measured 7.17 -> 7.04 ms, a third lookup there loses */
    const uint32_t e_ge3 = mcq_ld_u32(tf + 8192, ge3);
    const uint32_t d_any = mcq_ld_u32(sd, any);
    const uint32_t d_kick = mcq_ld_u32(sd, (any ^ ge2) + MCQ_KC_BYTE_OFFSET); /* kickers | type code of family F1 */
    const uint32_t key_f = mcq_ld_u32(tf, fs.bfl4 | mcq_bfe(fs.use_hi ? h.his : h.los, fs.sh, 16)); /* tf: LDS or global */

    const uint32_t key_s = d_any;
    const uint32_t key1 = (ge2 << 13) | d_kick;

    const bool fh = ge3 != 0;
    const uint32_t H = fh ? (e_ge3 >> 16) : (e_ge2 & 0xFFFFu);
    const uint32_t R = (fh ? ge2 : any) ^ H;
    const uint32_t kick2 = mcq_ld_u32(tops, R) >> 16;
    uint32_t key2 = (H << 13) | kick2 |
                    (fh ? (uint32_t)MCQ_C_FULL << MCQ_KEY_SHIFT : (uint32_t)MCQ_C_TWOPAIR << MCQ_KEY_SHIFT);
    key2 = (H != 0 && R != 0) ? key2 : 0u;

#ifdef MCQ_ABLATE_QUADS /* diagnostic timing build: wrong results */
    (void)eq4;
    return (key1 > key2 ? key1 : key2) > (key_s > key_f ? key_s : key_f) ? (key1 > key2 ? key1 : key2) : (key_s > key_f ? key_s : key_f);
#endif
    /* quads (0.17 % of hands), without a branch: a vote of the wave in front of this lookup cost as many instructions as
     * the lookup itself and cut the straight-line iteration into seven blocks */
    const uint32_t key4 = eq4 != 0 ? ((mcq_ld_u32(tops, any) & 0xFFFFu) | ((uint32_t)MCQ_C_QUADS << MCQ_KEY_SHIFT)) : 0u;

    uint32_t k = key1 > key2 ? key1 : key2;
    k = k > key_s ? k : key_s;
    uint32_t k2 = key_f > key4 ? key_f : key4;
    return k > k2 ? k : k2;
}

// ------------------------------------------------------------------------------------------ one query, one lane
struct McqQueryCtx { /* wave-uniform */
    uint32_t deck_lo, deck_hi; /* remaining deck after removing the known table cards and hero (l.126-161) */
    uint32_t L0;               /* its length: 50 - n_board */
    uint32_t n_opp;            /* n_players - 1 */
    uint32_t n_deal;           /* 5 - n_board table cards still to come */
    uint32_t runs;
    McqHole hero;
    McqBoard board; /* the known table cards */
};

// The 16-byte query record as four little-endian words (kept in SGPRs by the kernels): bytes 0-1 hole,
// 2-6 board, 7 n_board, 8 n_players, 9-11 reserved, 12-15 runs.  Register-only access: no byte arrays that
// would be indexed at run time (those go to scratch).
struct McqQueryWords {
    uint32_t w0, w1, w2, w3;
    MCQ_HDM uint32_t card(uint32_t k) const { /* k = 0,1: hole; 2..6: board */
        uint64_t v = ((uint64_t)w1 << 32) | w0;
        return (uint32_t)(v >> (8u * k)) & 0xFFu;
    }
    MCQ_HDM uint32_t n_board() const { return w1 >> 24; }
    MCQ_HDM uint32_t n_players() const { return w2 & 0xFFu; }
    MCQ_HDM uint32_t reserved() const { return w2 >> 8; }
    MCQ_HDM uint32_t runs() const { return w3; }
};

MCQ_HD bool mcq_query_valid(const McqQueryWords &q) {
    if (q.n_board() > 5 || q.n_players() < 1 || q.n_players() > 10 || q.reserved() != 0) return false;
    uint64_t seen = 0;
    bool ok = true;
    for (uint32_t i = 0; i < 2u + q.n_board(); i++) {
        uint32_t c = q.card(i);
        ok = ok && c < 52 && !((seen >> (c & 63u)) & 1);
        seen |= 1ull << (c & 63u);
    }
    return ok;
}

MCQ_HD void mcq_query_ctx(const McqQueryWords &q, McqQueryCtx &c) {
    uint64_t deck = (1ull << 52) - 1;
    c.board.clear();
    for (uint32_t i = 0; i < q.n_board(); i++) {
        const uint32_t cd = q.card(2u + i);
        deck &= ~(1ull << cd);
        c.board.add(mcq_card(cd));
    }
    const uint32_t h0 = q.card(0), h1 = q.card(1);
    deck &= ~(1ull << h0);
    deck &= ~(1ull << h1);
    c.hero.set(mcq_card(h0), mcq_card(h1));
    c.deck_lo = (uint32_t)deck;
    c.deck_hi = (uint32_t)(deck >> 32);
    c.L0 = 50u - q.n_board();
    c.n_opp = q.n_players() - 1u;
    c.n_deal = 5u - q.n_board();
    c.runs = q.runs();
}

// Scheduling weight of one wave task of a query, in units of 0.01 ns of measured kernel time per task
// (tools/weights_probe.py on MI355X: fit within 6 % over n_players 1..10, n_board 0/3/4/5).  Used only to cut the
// task list into equally expensive contiguous slices, never for results.
MCQ_HD uint32_t mcq_task_weight(const McqQueryWords &q) {
    const uint32_t n = q.n_players(), deal = 5u - q.n_board();
    return 160u + 200u * n + 10u * n * n + deal * (162u - 9u * n);
}
MCQ_HD uint32_t mcq_task_count(const McqQueryWords &q) { /* no overflow for runs up to 2^32 - 1 */
    return q.runs() / MCQ_TASK_ITERS + (q.runs() % MCQ_TASK_ITERS != 0u ? 1u : 0u);
}

// Share `part` of `n_parts` of a query: tasks [t_lo, t_hi) of its `tasks` tasks and the iterations they hold.
// The shares of all parts tile the query, so their tallies add up to the unsplit result (mcq_eval_batch_part).
struct McqPart {
    uint32_t t_lo, t_hi, runs;
};
MCQ_HD McqPart mcq_part(uint32_t tasks, uint32_t runs, uint32_t part, uint32_t n_parts) {
    McqPart p;
    p.t_lo = (uint32_t)((uint64_t)tasks * part / n_parts);
    p.t_hi = (uint32_t)((uint64_t)tasks * (part + 1u) / n_parts);
    const uint64_t a = (uint64_t)p.t_lo * MCQ_TASK_ITERS, b = (uint64_t)p.t_hi * MCQ_TASK_ITERS;
    p.runs = (uint32_t)((b < runs ? b : runs) - (a < runs ? a : runs));
    return p;
}

// Small batches: how finely the 1024-iteration tasks are cut (2^split sub-tasks each).  A lone wave per SIMD is bound
// by the latency of its dependent LDS lookups, so tasks are cut while that leaves at most two waves per SIMD -- and
// at most 512 waves on one query: every (wave, query) pair ends in twelve atomics on the query's result row, and
// atomics on one address serialise (100k runs: 31 us uncut, 18 us in 392 pieces, 29 us in 1568).  The tallies do not
// depend on the cut.  Decided on the host when it sees the queries, by the prep kernel when they live in HBM.
#define MCQ_SPLIT_FROM_PREP 0xFFFFFFFFu /* evaluation kernel argument: read the cut the prep kernel chose */
MCQ_HD uint32_t mcq_pick_split(uint64_t total_tasks, uint64_t max_tasks, uint32_t n_cu, uint32_t split_max) {
    const uint64_t want = 8ull * n_cu;
    uint32_t s = 0;
    while (s < split_max && (total_tasks << (s + 1u)) <= want && (max_tasks << (s + 1u)) <= 512u) s++;
    return s;
}

static inline McqQueryWords mcq_query_words(const mcq_query &q) { /* host side */
    McqQueryWords w;
    __builtin_memcpy(&w, &q, 16);
    return w;
}

// entry l of the base deck = the l-th card of the ordered remaining deck (device: lane l computes entry l)
MCQ_HD McqCard mcq_base_entry(const McqQueryCtx &qc, uint32_t l, const uint32_t *sel8) {
    uint32_t dlo = qc.deck_lo, dhi = qc.deck_hi;
    uint32_t c = mcq_select_pop(dlo, dhi, l, sel8);
    return mcq_card(c < 52u ? c : 0u); /* lanes beyond the deck length write an entry nobody reads */
}

struct McqLaneAcc {
    uint64_t types; /* MCQ_N_CODES fields of 6 bits: hero's winning hand codes (<= 16 per lane per task) */
    uint32_t tie;
    uint32_t passes;
};

// ------------------------------------------------------------------------------------------ dealing without search
// list.pop(r) on the ordered remaining deck, reformulated so that no k-th-set-bit search is needed:
// the base deck (query's deck minus table cards and hero, wave-uniform) is a table base[0..L0) in LDS; every card
// already dealt in this iteration is a HOLE, remembered by its coordinate t in the CURRENT (shrunk) list, i.e.
// the number of not-yet-dealt cards below it.  Then for a draw r on the current list
//     base position s = r + #{holes with t <= r};   afterwards every hole with t > r moves down by one and the
//     new hole has t = r.
// Holes are independent of each other (no ordering to maintain), so four of them are handled per register with
// byte-SWAR arithmetic: with rb = 0x80808080 | r * 0x01010101, bit 7 of each byte of (rb - h) says t <= r.
// Unused byte slots hold the sentinel 0x7F and are decremented along (23 draws at most: always > 49 >= r).
// This is an exact restatement of montecarlo_python.py:178-179,188 (checked against list.pop in the tests).
#define MCQ_HOLE_SENTINEL 0x7F7F7F7Fu

MCQ_HD void mcq_hole_reg(uint32_t rb, uint32_t &h, uint32_t &k) {
#ifdef MCQ_ABLATE_HOLES /* diagnostic timing build: wrong results */
    k += (rb ^ h) & 1u;
    return;
#endif
    const uint32_t f = ((rb - h) >> 7) & 0x01010101u; /* byte i = 1 iff t_i <= r */
    k = mcq_sad_u8(f, k);                             /* k += number of such holes */
    h = h + f - 0x01010101u;                          /* every other hole moves down by one */
}

MCQ_HD uint32_t mcq_bfi(uint32_t mask, uint32_t a, uint32_t b) { /* (mask & a) | (~mask & b): one v_bfi_b32 */
    return (mask & a) | (~mask & b);
}

// Store t = r in byte SLOT.  rb7 = r (optionally with bit 7 set) in every byte, as the scan needs it anyway, so
// the insertion is a single bit-field insert of the low seven bits (the slot's bit 7 is always clear).
template <int SLOT>
MCQ_HD void mcq_hole_put(uint32_t &h, uint32_t rb7) {
    constexpr uint32_t sh = 8u * (SLOT & 3);
    h = mcq_bfi(0x7Fu << sh, rb7, h);
}

// Draws arrive as rp = r | 0x80 (r < 64).  Returned: base position + 128 (the caller's table pointer is biased).
// opponent draw number J (0-based; J holes precede it, all in H[0 .. (J+3)/4))
template <int J>
MCQ_HD uint32_t mcq_draw_opp(uint32_t rp, uint32_t (&H)[5]) {
    uint32_t k = rp;
    if (J > 0) {
        const uint32_t rb = mcq_splat_byte(rp);
#pragma unroll
        for (int i = 0; i < (J + 3) / 4; i++) mcq_hole_reg(rb, H[i], k);
        mcq_hole_put<J>(H[J / 4], rb);
    } else {
        mcq_hole_put<0>(H[0], rp);
    }
    return mcq_opaque(k); /* materialise k so that the table address is one shift-add */
}

// Count only: k += number of holes of the register with t <= r (the register is not updated).
MCQ_HD void mcq_hole_count(uint32_t rb, uint32_t h, uint32_t &k) {
#ifdef MCQ_ABLATE_HOLES /* diagnostic timing build: wrong results */
    k += (rb ^ h) & 1u;
    return;
#endif
    k = mcq_opaque(mcq_popc((rb - h) & 0x80808080u) + k); /* v_sub, v_and, v_bcnt with its add (opaque: else the compiler
                                                            counts into zero and sums afterwards: one more add3) */
}

// table draw number K (0..4).  Two levels, because the table is dealt after ALL opponents (l.215-217): the K earlier
// table holes live in their own register hb, in coordinates of the current list, and are counted and moved as in
// mcq_hole_reg; that gives the draw's position p in the list as it was when the opponents had been dealt -- and in
// THAT list the opponents' holes (NREGS registers) have fixed coordinates from here on: they are only counted
// against p (3 instructions per register instead of 5) and never moved again.
// NREGS is a template parameter: with a run-time bound the compiler scans all five registers and discards the
// unused ones with selects (7 instructions per register instead of 0).
template <int K, int NREGS>
MCQ_HD uint32_t mcq_draw_table(uint32_t rp, const uint32_t (&H)[5], uint32_t &hb) {
    uint32_t p = rp; /* r | 0x80 */
    if (K > 0) {
        const uint32_t rb = mcq_splat_byte(rp);
        mcq_hole_reg(rb, hb, p);
        if (K < 4) mcq_hole_put<K>(hb, rb); /* the hole of a fifth table card is never looked at */
    } else {
        mcq_hole_put<0>(hb, rp);
    }
    uint32_t k = p; /* position among the cards the opponents left, still | 0x80 */
    if (NREGS > 0) {
        const uint32_t pb = mcq_splat_byte(p);
#pragma unroll
        for (int i = 0; i < NREGS; i++) mcq_hole_count(pb, H[i], k);
    }
    return mcq_opaque(k);
}

// the missing table cards (montecarlo_python.py:185-189) after opponents whose holes fill NREGS registers
template <int NREGS, class Draws, int NDEAL = -1>
MCQ_HD void mcq_deal_table(const McqQueryCtx &qc, Draws &dr, const McqCard *base128, const uint32_t (&H)[5], uint32_t L,
                           McqBoard &b) {
    uint32_t hb = MCQ_HOLE_SENTINEL;
    /* scalar compares, no lane masks kept in SGPR pairs; NDEAL >= 0: known at compile time, no branches at all */
    const uint32_t n_deal = NDEAL >= 0 ? (uint32_t)NDEAL : mcq_opaque_uniform(qc.n_deal);
    /* A card joins the table one draw late: its LDS read leaves at the end of its block and is waited for behind the
     * next draw's arithmetic (the blocks are separate basic blocks, the compiler cannot move the wait itself). */
    McqCard pend = {0u, 0u, 0u, 0u};
#define MCQ_TABLE(K)                                                                                            \
    if (K < n_deal) {                                                                                           \
        const uint32_t at = mcq_draw_table<K, NREGS>(dr.template table<K>(L - Draws::kTableShort), H, hb); /* l.188 */ \
        if (K > 0) b.add(pend);                                                                                 \
        pend = base128[at];                                                                                     \
        L -= 1;                                                                                                 \
    }
    MCQ_TABLE(0) MCQ_TABLE(1) MCQ_TABLE(2) MCQ_TABLE(3) MCQ_TABLE(4)
#undef MCQ_TABLE
    if (n_deal > 0u) b.add(pend);
}

// One Monte-Carlo iteration of one lane.  The opponents' hole cards stay in registers (statically indexed:
// everything below is unrolled over the opponent number) until the table is complete: the reference deals
// ALL opponents before any table card (montecarlo_python.py:215-217).
// NOPP / NDEAL >= 0: the number of opponents / of table cards to come is known at compile time (it must equal
// qc.n_opp / qc.n_deal): the iteration is ONE basic block, which lets the compiler send lookups early and wait late
// across hands and draws -- the wave-uniform branches of the general form are scheduling barriers.
template <class Draws, int NOPP = -1, int NDEAL = -1>
// base128 = (base deck table) - 128 entries: draw indices carry a bias of 128 (r | 0x80), folded into the pointer.
MCQ_HD void mcq_iteration(const McqQueryCtx &qc, Draws &dr, const McqCard *base128, const uint32_t *tf,
                          const uint32_t *tops, const uint32_t *sd, McqLaneAcc &acc) {
    uint32_t H[5] = {MCQ_HOLE_SENTINEL, MCQ_HOLE_SENTINEL, MCQ_HOLE_SENTINEL, MCQ_HOLE_SENTINEL, MCQ_HOLE_SENTINEL};
    uint32_t L = qc.L0;
    McqHole opp[MCQ_MAX_OPP];
    const uint32_t n_opp_d = NOPP >= 0 ? (uint32_t)NOPP : mcq_opaque_uniform(qc.n_opp);
#define MCQ_OPP(P)                                                                                             \
    if (P < n_opp_d) {                                                                                         \
        uint32_t r1, r2;                                                                                       \
        dr.template pair<P>(L, r1, r2); /* r1 in [0,L-1], r2 in [0,L-2], r1 != r2 (l.167-176), both | 0x80 */  \
        const McqCard c1 = base128[mcq_draw_opp<2 * P>(r1, H)];     /* deck.pop(r1) (l.178) */                 \
        const McqCard c2 = base128[mcq_draw_opp<2 * P + 1>(r2, H)]; /* deck.pop(r2), shrunk list (l.179) */    \
        opp[P].set(c1, c2);                                                                                    \
        L -= 2;                                                                                                \
    }
    MCQ_OPP(0) MCQ_OPP(1) MCQ_OPP(2) MCQ_OPP(3) MCQ_OPP(4) MCQ_OPP(5) MCQ_OPP(6) MCQ_OPP(7) MCQ_OPP(8)
#undef MCQ_OPP
    McqBoard b = qc.board;
    if (NOPP >= 0) {
        mcq_deal_table<(NOPP >= 0 ? (2 * NOPP + 3) / 4 : 0), Draws, NDEAL>(qc, dr, base128, H, L, b);
    } else switch (mcq_opaque_uniform((2u * qc.n_opp + 3u) / 4u)) { /* registers holding the opponents' holes: wave-uniform */
        case 0: mcq_deal_table<0>(qc, dr, base128, H, L, b); break;
        case 1: mcq_deal_table<1>(qc, dr, base128, H, L, b); break;
        case 2: mcq_deal_table<2>(qc, dr, base128, H, L, b); break;
        case 3: mcq_deal_table<3>(qc, dr, base128, H, L, b); break;
        case 4: mcq_deal_table<4>(qc, dr, base128, H, L, b); break;
        default: mcq_deal_table<5>(qc, dr, base128, H, L, b); break;
    }
    McqFlushSel fs;
    fs.from_board(b);
    const uint32_t hk = mcq_eval_key(b, fs, qc.hero, tf, tops, sd);
    uint32_t best = 0;
    const uint32_t n_opp_e = NOPP >= 0 ? (uint32_t)NOPP : mcq_opaque_uniform(qc.n_opp); /* a fresh scalar compare per block, see mcq_opaque_uniform */
#define MCQ_EVAL(P)                                                        \
    if (P < n_opp_e) {                                                     \
        const uint32_t k = mcq_eval_key(b, fs, opp[P], tf, tops, sd);     \
        best = k > best ? k : best;                                        \
    }
    MCQ_EVAL(0) MCQ_EVAL(1) MCQ_EVAL(2) MCQ_EVAL(3) MCQ_EVAL(4) MCQ_EVAL(5) MCQ_EVAL(6) MCQ_EVAL(7) MCQ_EVAL(8)
#undef MCQ_EVAL
    uint64_t won = hk >= best ? 1u : 0u; /* ties go to hero (hand_evaluator.py:23) */
    acc.types += won << (6u * (hk >> MCQ_KEY_SHIFT));
    acc.tie += hk == best ? 1u : 0u;
}

// `cnt` iterations of one lane.  STRAIGHT: by the (wave-uniform) number of opponents, and before the flop also by the
// number of table cards, the loop body is a specialisation of mcq_iteration without branches -- one basic block, in
// which the compiler sends lookups early and waits late across hands and draws (6-max before the flop: 6.37 -> 6.00 ms;
// the general form's wave-uniform branches are scheduling barriers).  Same arithmetic, same results.
template <bool STRAIGHT, class Draws>
MCQ_HD void mcq_iterations(const McqQueryCtx &qc, Draws &dr, const McqCard *base128, const uint32_t *tf,
                                               const uint32_t *tops, const uint32_t *sd, McqLaneAcc &acc, uint32_t cnt) {
    if (STRAIGHT) {
#define MCQ_STRAIGHT(N)                                                                                   \
    case N:                                                                                               \
        if (qc.n_deal == 5u)                                                                              \
            for (uint32_t j = 0; j < cnt; j++) mcq_iteration<Draws, N, 5>(qc, dr, base128, tf, tops, sd, acc); \
        else if (qc.n_deal == 2u)                                                                         \
            for (uint32_t j = 0; j < cnt; j++) mcq_iteration<Draws, N, 2>(qc, dr, base128, tf, tops, sd, acc); \
        else if (qc.n_deal == 1u)                                                                         \
            for (uint32_t j = 0; j < cnt; j++) mcq_iteration<Draws, N, 1>(qc, dr, base128, tf, tops, sd, acc); \
        else                                                                                              \
            for (uint32_t j = 0; j < cnt; j++) mcq_iteration<Draws, N, -1>(qc, dr, base128, tf, tops, sd, acc); \
        return;
        switch (qc.n_opp) {
            MCQ_STRAIGHT(1) MCQ_STRAIGHT(2) MCQ_STRAIGHT(3) MCQ_STRAIGHT(4) MCQ_STRAIGHT(5) MCQ_STRAIGHT(6) MCQ_STRAIGHT(7)
            case 8: /* (only the form with the table cards counted at run time: the others spill registers) */
                for (uint32_t j = 0; j < cnt; j++) mcq_iteration<Draws, 8, -1>(qc, dr, base128, tf, tops, sd, acc);
                return;
            case 9:
                for (uint32_t j = 0; j < cnt; j++) mcq_iteration<Draws, 9, -1>(qc, dr, base128, tf, tops, sd, acc);
                return;
            default: break; /* hero alone: the general form */
        }
#undef MCQ_STRAIGHT
    }
    for (uint32_t j = 0; j < cnt; j++) mcq_iteration(qc, dr, base128, tf, tops, sd, acc);
}

// ================================================================================================ extended queries
// SURVEY 8f-2, all of run_montecarlo's arguments: opponent ranges (montecarlo_python.py:165-181 with :36-112), ghost
// cards (:206-208), any number of known hands in the order of original_player_card_list, each two cards or a SET of
// preflop classes (:133-163).  The deck of an iteration is a 52-bit mask per lane (card-id order = the reference's
// list order); cards come from a 52-entry table, the dealt hands' card ids wait in LDS.
//
// A range is a 169-bit set; the bit of two cards is how get_two_short_notation (:24-34) names them:
// suited -> 13*min+max, off-suit -> 13*max+min, pair -> 14*rank.
//
// Production mode ("MCQ-CTR v5x") deals the reference's LAW without its index arithmetic and without its re-draw
// loop over all L(L-1) index pairs.  The reference accepts, equally often, every ordered index pair (r1, r2),
// r1 in [0,L), r2 in [0,L-1), r1 != r2, whose classes are allowed (:167-176); as cards: every ordered pair (A, B) of
// distinct cards of the current deck with B not the deck's highest card.  Per range there is a fixed CANDIDATE LIST
//     P' = [(a, b) for a in 0..51 for b in 0..51 if a != b and a, b in U and class(a, b) allowed]
// over U = the cards that can still be in the deck at that point whatever was drawn before (52 minus ghost, table and
// the LIST hands dealt earlier; all list hands for the opponents), laid out once per query by mcq_ext_lists_kernel.
// A trial: one word u, (A, B) = P'[mulhi32(u, |P'|)], accepted iff A and B are still in the deck and B is not its
// highest card -- uniform on exactly the reference's accepted set.  A known hand leaves by value (:146-161); an
// opponent is dealt A and, as deck.pop(r1); deck.pop(r2) deal (:178-179), B if B lies below A, else the card that
// FOLLOWS B in the deck (the reference's quirk: the range test looks at the unpopped list).  With the top quarter of
// the classes a trial of the reference's loop succeeds one time in ~25, a trial here three times in four.
// Opponents to whom every class is allowed are dealt by index exactly as the plain path deals them (MCQ-CTR v5, one
// word per pair), so an extension record that restricts nothing gives the plain path's tallies bit for bit.
#define MCQ_EXT_WORDS 76u         /* sizeof(mcq_query_ext) / 4 */
#define MCQ_EXT_MAX_LISTS 11u     /* ten known hands as ranges + the opponents */
#define MCQ_EXT_LIST_STRIDE 2704u /* entries reserved per candidate list (52 * 52 >= 52 * 51) */
#define MCQ_EXT_MAX_TRIALS 65536u /* bound of a re-draw loop: a range that cannot be dealt must not hang */

struct McqExtRec { /* the 304-byte mcq_query_ext record as words (any address space) */
    const uint32_t *w;
    MCQ_HDM uint32_t ghost(uint32_t i) const { return (w[0] >> (8u * i)) & 0xFFu; } /* 0xFF = none */
    MCQ_HDM uint32_t hero_is_range() const { return (w[0] >> 16) & 0xFFu; }
    MCQ_HDM uint32_t n_known() const { return w[0] >> 24; }
    MCQ_HDM uint32_t known_head(uint32_t k) const { return w[13u + 7u * k]; } /* cards[2], is_range, reserved */
    /* word offsets of the 6-word sets */
    MCQ_HDM uint32_t opp_set() const { return 1u; }
    MCQ_HDM uint32_t hero_set() const { return 7u; }
    MCQ_HDM uint32_t known_set(uint32_t k) const { return 14u + 7u * k; }
};

MCQ_HD uint32_t mcq_class_index(uint32_t a, uint32_t b) {
    const uint32_t ra = a >> 2, rb = b >> 2, lo = ra < rb ? ra : rb, hi = ra < rb ? rb : ra;
    if (ra == rb) return 14u * ra;
    return (a & 3u) == (b & 3u) ? 13u * lo + hi : 13u * hi + lo;
}
MCQ_HD bool mcq_in_range(const uint32_t *bits, uint32_t a, uint32_t b) {
    const uint32_t i = mcq_class_index(a, b);
    return (bits[i >> 5] >> (i & 31u)) & 1u;
}

/* hand h of the query (0 = hero, 1.. = known[h - 1]): cards a | b << 8 | is_range << 16 */
MCQ_HD uint32_t mcq_ext_hand(const McqQueryWords &q, const McqExtRec &e, uint32_t h) {
    if (h == 0) return e.hero_is_range() ? 0x10000u : (q.card(0) | (q.card(1) << 8));
    const uint32_t hd = e.known_head(h - 1u);
    return (hd >> 16) & 0xFFu ? 0x10000u : (hd & 0xFFFFu);
}
MCQ_HD uint32_t mcq_ext_hand_set(const McqExtRec &e, uint32_t h) { return h == 0 ? e.hero_set() : e.known_set(h - 1u); }

/* opponents unrestricted (every one of the 169 classes allowed)? */
MCQ_HD bool mcq_ext_opp_all(const McqExtRec &e) {
    uint32_t all = e.w[e.opp_set() + 5u] | ~0x1FFu;
    for (uint32_t i = 0; i < 5; i++) all &= e.w[e.opp_set() + i];
    return all == 0xFFFFFFFFu;
}

/* candidate lists of a query: one per known hand given as a range, in hand order, then one for the opponents
 * unless every class is allowed to them (they are then dealt by index, exactly as the plain path deals) */
MCQ_HD uint32_t mcq_ext_n_lists(const McqQueryWords &q, const McqExtRec &e) {
    uint32_t n = q.n_players() > 1u + e.n_known() && !mcq_ext_opp_all(e) ? 1u : 0u;
    for (uint32_t h = 0; h <= e.n_known(); h++) n += mcq_ext_hand(q, e, h) >> 16;
    return n;
}

/* Streams of the production mode (MCQ-CTR v5x): a query that draws from candidate lists and has at most
 * MCQ_EXT_SHORT_RUNS iterations cuts them into streams of MCQ_EXT_SHORT_STREAM iterations instead of MCQ_STREAM_ITERS --
 * a trial loop consumes a number of words nobody knows beforehand, so a stream cannot be entered half way as the plain
 * path's can, and a 1000-run query would otherwise be 63 lanes of ONE wave, sixteen iterations each, one behind the
 * other (34 us); so it is 500 lanes of eight waves, two iterations each.  Long queries keep the long streams (one
 * Philox block per sixteen iterations), and so does a query without lists: it deals exactly as the plain path. */
#define MCQ_EXT_SHORT_STREAM 2u
#define MCQ_EXT_SHORT_RUNS 8192u
MCQ_HD uint32_t mcq_ext_stream_iters(const McqQueryWords &q, const McqExtRec &e) {
    return q.runs() <= MCQ_EXT_SHORT_RUNS && mcq_ext_n_lists(q, e) != 0u ? MCQ_EXT_SHORT_STREAM : MCQ_STREAM_ITERS;
}
/* wave tasks (64 streams) of a query whose streams hold s_iters iterations, and what one costs (the unit of the cost
 * axis the kernels cut: three per weight and two iterations of a stream) */
MCQ_HD uint32_t mcq_ext_task_count(const McqQueryWords &q, uint32_t s_iters) {
    const uint32_t per = s_iters * MCQ_WAVE;
    return q.runs() / per + (q.runs() % per != 0u ? 1u : 0u);
}
MCQ_HD uint32_t mcq_ext_task_weight(const McqQueryWords &q, uint32_t s_iters) { return 3u * mcq_task_weight(q) * (s_iters >> 1); }

MCQ_HD uint64_t mcq_ext_base_deck(const McqQueryWords &q, const McqExtRec &e) { /* 52 cards minus ghost and table */
    uint64_t deck = (1ull << 52) - 1;
    for (uint32_t i = 0; i < q.n_board(); i++) deck &= ~(1ull << q.card(2u + i));
    if (e.ghost(0) != 0xFFu) deck &= ~((1ull << e.ghost(0)) | (1ull << e.ghost(1)));
    return deck;
}

/* list `li` of the query: the cards U its candidates are made of and the word offset of its class set */
MCQ_HD void mcq_ext_list_plan(const McqQueryWords &q, const McqExtRec &e, uint32_t li, uint64_t &U, uint32_t &set_off) {
    U = mcq_ext_base_deck(q, e);
    set_off = e.opp_set();
    uint32_t seen = 0;
    for (uint32_t h = 0; h <= e.n_known(); h++) {
        const uint32_t hd = mcq_ext_hand(q, e, h);
        if (hd >> 16) {
            if (seen == li) { set_off = mcq_ext_hand_set(e, h); return; }
            seen++;
        } else {
            U &= ~((1ull << (hd & 0xFFu)) | (1ull << ((hd >> 8) & 0xFFu)));
        }
    }
}

/* is candidate c = 52 * a + b on a list? */
MCQ_HD bool mcq_ext_candidate(uint64_t U, const uint32_t *set, uint32_t c) {
    const uint32_t a = c / 52u, b = c - 52u * a;
    return c < 2704u && a != b && ((U >> a) & 1u) && ((U >> b) & 1u) && mcq_in_range(set, a, b);
}

// every card named twice is an error; a range that is used must not be empty
MCQ_HD bool mcq_query_ext_valid(const McqQueryWords &q, const McqExtRec &e) {
    if (q.n_board() > 5 || q.n_players() < 1 || q.n_players() > 10 || q.reserved() != 0) return false;
    if (e.hero_is_range() > 1 || e.n_known() > MCQ_MAX_KNOWN || q.n_players() < 1u + e.n_known()) return false;
    uint64_t seen = 0;
    bool ok = true;
    for (uint32_t i = 0; i < q.n_board(); i++) {
        const uint32_t c = q.card(2u + i);
        ok = ok && c < 52 && !((seen >> (c & 63u)) & 1);
        seen |= 1ull << (c & 63u);
    }
    if (e.ghost(0) != 0xFFu || e.ghost(1) != 0xFFu)
        for (uint32_t i = 0; i < 2; i++) {
            const uint32_t c = e.ghost(i);
            ok = ok && c < 52 && !((seen >> (c & 63u)) & 1);
            seen |= 1ull << (c & 63u);
        }
    for (uint32_t h = 0; h <= e.n_known(); h++) {
        if (h > 0) {
            const uint32_t hd = e.known_head(h - 1u);
            if (((hd >> 16) & 0xFFu) > 1u || (hd >> 24) != 0u) return false;
        }
        const uint32_t hd = mcq_ext_hand(q, e, h);
        if (hd >> 16) {
            const uint32_t off = mcq_ext_hand_set(e, h);
            uint32_t any = 0;
            for (uint32_t i = 0; i < 6; i++) any |= e.w[off + i];
            ok = ok && any != 0;
        } else {
            for (uint32_t i = 0; i < 2; i++) {
                const uint32_t c = (hd >> (8u * i)) & 0xFFu;
                ok = ok && c < 52 && !((seen >> (c & 63u)) & 1);
                seen |= 1ull << (c & 63u);
            }
        }
    }
    if (q.n_players() > 1u + e.n_known()) {
        uint32_t any = 0;
        for (uint32_t i = 0; i < 6; i++) any |= e.w[e.opp_set() + i];
        ok = ok && any != 0;
    }
    return ok;
}

struct McqExtCtx { /* wave-uniform */
    uint32_t deck_lo, deck_hi;            /* 52 cards minus ghost and table */
    uint32_t n_players, n_hands, n_deal, runs; /* n_hands = 1 + n_known */
    bool opp_all;                         /* every class allowed to the opponents: dealt by index as in the plain path */
    McqBoard board;
    /* the common ranged query -- hero two cards, no further known hand, every opponent drawn from ONE candidate list --
     * takes mcq_iteration_ext_fast: */
    bool fast;
    uint32_t fdeck_lo, fdeck_hi;          /* the deck without hero's cards too */
    McqHole hero;
};
struct McqExtWaveCtx { /* per wave, in LDS: what the iteration indexes at run time */
    uint32_t hand[10];                  /* mcq_ext_hand of every known hand */
    uint32_t cnt[MCQ_EXT_MAX_LISTS];    /* sizes of the candidate lists */
    uint32_t pad_[3];
    const uint16_t *list[MCQ_EXT_MAX_LISTS]; /* where each list lies: HBM, or the block's LDS when its queries' lists fit */
};

MCQ_HD void mcq_ext_ctx(const McqQueryWords &q, const McqExtRec &e, McqExtCtx &c) {
    const uint64_t deck = mcq_ext_base_deck(q, e);
    c.board.clear();
    for (uint32_t i = 0; i < q.n_board(); i++) c.board.add(mcq_card(q.card(2u + i)));
    c.deck_lo = (uint32_t)deck;
    c.deck_hi = (uint32_t)(deck >> 32);
    c.n_players = q.n_players();
    c.n_hands = 1u + e.n_known();
    c.n_deal = 5u - q.n_board();
    c.runs = q.runs();
    c.opp_all = mcq_ext_opp_all(e);
    c.fast = !e.hero_is_range() && e.n_known() == 0u && !c.opp_all && q.n_players() >= 2u;
    const uint32_t h0 = q.card(0), h1 = q.card(1);
    const uint64_t fdeck = deck & ~(((uint64_t)1 << (h0 & 63u)) | ((uint64_t)1 << (h1 & 63u)));
    c.fdeck_lo = (uint32_t)fdeck;
    c.fdeck_hi = (uint32_t)(fdeck >> 32);
    c.hero.set(mcq_card(h0 < 52u ? h0 : 0u), mcq_card(h1 < 52u ? h1 : 0u));
}

/* deck mask helpers (52 bits in two words) */
MCQ_HD bool mcq_deck_has(uint32_t lo, uint32_t hi, uint32_t c) { return (((c & 32u) ? hi : lo) >> (c & 31u)) & 1u; }
MCQ_HD void mcq_deck_take(uint32_t &lo, uint32_t &hi, uint32_t c) {
    const uint32_t bit = 1u << (c & 31u);
    lo &= (c & 32u) ? 0xFFFFFFFFu : ~bit;
    hi &= (c & 32u) ? ~bit : 0xFFFFFFFFu;
}
MCQ_HD uint32_t mcq_deck_top(uint32_t lo, uint32_t hi) { /* highest card, deck not empty */
    return hi ? 63u - mcq_clz(hi) : 31u - mcq_clz(lo | 1u);
}
MCQ_HD uint32_t mcq_deck_next(uint32_t lo, uint32_t hi, uint32_t c) { /* the card that follows c; c is not the top */
    const uint64_t above = ((((uint64_t)hi << 32) | lo) >> c) >> 1; /* cards above c, bit 0 = c + 1 */
    const uint32_t alo = (uint32_t)above, ahi = (uint32_t)(above >> 32);
    const uint32_t low = alo ? alo & (0u - alo) : ahi & (0u - ahi);
    return c + 1u + (alo ? 31u - mcq_clz(low | 1u) : 63u - mcq_clz(low | 1u));
}

// Draw policies (the extended path is not unrolled).
struct McqExtCtrDraws {
    static constexpr bool kReplay = false;
    McqStreamRng rng;
    uint32_t w;
    MCQ_HDM void start(uint64_t seed, uint64_t qid, uint32_t stream) {
        w = 0;
        rng.seed(seed, qid, stream);
    }
    MCQ_HDM uint32_t pick(uint32_t n) { return mcq_mulhi(rng.next(), n); } /* a candidate of a list of n */
    MCQ_HDM void pair(uint32_t &, uint32_t &) {}
    MCQ_HDM void index_pair(uint32_t L, uint32_t &r1, uint32_t &r2) { /* MCQ-CTR v5 as the plain path: McqCtrDrawsT::pair */
        const uint32_t dd = L - 1u;
        const uint32_t u = rng.next();
        uint32_t frac;
        const uint32_t a = mcq_mulhilo_p128(u, dd, 0ull, frac);
        const uint32_t c = mcq_mulhi(frac, dd);
        r1 = a == c ? dd : a;
        r2 = c;
    }
    MCQ_HDM uint32_t table(uint32_t k, uint32_t n) {
        if ((k & 1u) == 0) {
            const uint32_t u = rng.next();
            return mcq_mulhilo_p128(u, n, 0ull, w);
        }
        return mcq_mulhi(w, n);
    }
};
struct McqExtReplayDraws { /* accepted draws from the host, all in list.pop order */
    static constexpr bool kReplay = true;
    const uint8_t *p;
    uint64_t stride;
    MCQ_HDM uint32_t pick(uint32_t) { return 0; }
    MCQ_HDM void index_pair(uint32_t, uint32_t &, uint32_t &) {}
    MCQ_HDM void pair(uint32_t &r1, uint32_t &r2) {
        r1 = p[0] & 0x7Fu; /* the host stores every draw as r | 0x80 */
        r2 = p[stride] & 0x7Fu;
        p += 2 * stride;
    }
    MCQ_HDM uint32_t table(uint32_t, uint32_t) {
        uint32_t v = p[0] & 0x7Fu;
        p += stride;
        return v;
    }
};

// One iteration of an extended query.  ids: this lane's slot array (stride `ids_stride` words) for the dealt
// hands; cards: the 52-entry card table; wc.list: the query's candidate lists,
// wc: the known hands and the list sizes.  Returns false when a range could not be dealt within
// MCQ_EXT_MAX_TRIALS attempts.
template <class Draws, bool LDS_LIST = false /* the candidate lists lie in LDS: see mcq_iteration_ext_fast */>
MCQ_HD bool mcq_iteration_ext(const McqExtCtx &qc, const McqExtWaveCtx &wc, Draws &dr, const McqCard *cards,
                              const uint32_t *sel8, uint16_t *ids, uint32_t ids_stride,
                              const uint32_t *tf, const uint32_t *tops, const uint32_t *sd, McqLaneAcc &acc) {
    uint32_t dlo = qc.deck_lo, dhi = qc.deck_hi;
    bool dealt = true;
    uint32_t li = 0; /* candidate list of the next known hand given as a range; the opponents' comes after them */
    for (uint32_t h = 0; h < qc.n_players; h++) {
        const bool known = h < qc.n_hands;
        const uint32_t hd = known ? wc.hand[h] : 0x10000u;
        uint32_t c1, c2;
        if (!(hd >> 16)) { /* two cards: they leave by value, if they are still there (l.150-161) */
            c1 = hd & 0xFFu;
            c2 = (hd >> 8) & 0xFFu;
        } else if (Draws::kReplay) {
            uint32_t r1, r2;
            dr.pair(r1, r2);
            c1 = mcq_select_pop(dlo, dhi, r1, sel8);
            c2 = mcq_select_pop(dlo, dhi, r2, sel8);
        } else if (!known && qc.opp_all) { /* no range to respect: one word, never re-drawn, popped in turn (l.178-179) */
            uint32_t r1, r2;
            acc.passes++;
            dr.index_pair(mcq_popc(dlo) + mcq_popc(dhi), r1, r2);
            c1 = mcq_select_pop(dlo, dhi, r1, sel8);
            c2 = mcq_select_pop(dlo, dhi, r2, sel8);
        } else {
            const uint32_t n = wc.cnt[li];
            const uint16_t *list = wc.list[li];
#if defined(__HIP_DEVICE_COMPILE__)
            typedef __attribute__((address_space(3))) const uint16_t *LdsList;
            const LdsList list_l = (LdsList)(uintptr_t)list;
#define MCQ_XLIST(i) (LDS_LIST ? (uint32_t)list_l[i] : (uint32_t)list[i])
#else
#define MCQ_XLIST(i) ((uint32_t)list[i])
#endif
            const uint32_t top = mcq_deck_top(dlo, dhi);
            bool ok = false;
            c1 = c2 = 0;
            uint32_t trial = 0;
            for (; trial < MCQ_EXT_MAX_TRIALS && !ok; trial++) {
                const uint32_t e = MCQ_XLIST(dr.pick(n));
                c1 = e & 0xFFu;
                c2 = e >> 8;
                ok = mcq_deck_has(dlo, dhi, c1) && mcq_deck_has(dlo, dhi, c2) && c2 != top;
            }
#undef MCQ_XLIST
            acc.passes += trial;
            dealt = dealt && ok;
            if (!known && ok && c2 > c1) c2 = mcq_deck_next(dlo, dhi, c2); /* deck.pop(r2) after deck.pop(r1), l.178-179 */
        }
        if (known && (hd >> 16)) li++;
        mcq_deck_take(dlo, dhi, c1 < 52u ? c1 : 0u);
        mcq_deck_take(dlo, dhi, c2 < 52u ? c2 : 0u);
        ids[h * ids_stride] = (uint16_t)(c1 | (c2 << 8));
    }
    McqBoard b = qc.board;
    for (uint32_t k = 0; k < qc.n_deal; k++) {
        const uint32_t L = mcq_popc(dlo) + mcq_popc(dhi);
        const uint32_t c = mcq_select_pop(dlo, dhi, dr.table(k, L - 1u), sel8);
        b.add(cards[c < 52u ? c : 0u]);
    }
    McqFlushSel fs;
    fs.from_board(b);
    uint32_t hk = 0, best = 0;
    for (uint32_t h = 0; h < qc.n_players; h++) {
        const uint32_t v = ids[h * ids_stride];
        McqHole hh;
        hh.set(cards[v & 0xFFu], cards[(v >> 8) & 0xFFu]);
        const uint32_t k = mcq_eval_key(b, fs, hh, tf, tops, sd);
        if (h == 0) hk = k;
        else best = k > best ? k : best;
    }
    uint64_t won = hk >= best ? 1u : 0u;
    acc.types += won << (6u * (hk >> MCQ_KEY_SHIFT));
    acc.tie += hk == best ? 1u : 0u;
    return dealt;
}

// The common ranged query in straight code: hero two cards, no further known hand, all n_players - 1 opponents drawn
// from ONE candidate list (wc.list[0]).  The same draws in the same order as mcq_iteration_ext -- one word per trial,
// then the table cards two per word -- hence the same tallies bit for bit; what it saves is the general form's
// bookkeeping: the deck is one 64-bit mask tested with two shifts per trial, the opponents' hands stay in registers
// (unrolled over the opponent number, as in mcq_iteration) instead of travelling through LDS as card ids, the deck's
// length is known without counting (every lane deals two cards per opponent), hero's hand is part of the query context.
// 6-max at the top quarter of the classes: ~1650 -> ~900 VALU instructions per wave-iteration.
// (PIN: keep the opponent count in a scalar register of its own, see mcq_opaque_uniform; the one-launch kernel, whose
// query context comes out of LDS, cannot: the backend refuses the copy.)
// (LDS_LIST: the caller knows the candidate list lies in LDS -- staged by the block -- so a trial reads it with a
// 32-bit LDS address instead of through a generic pointer: a flat load waits for both memory counters and takes
// several times as long, in a loop where every trial is one dependent chain word -> index -> entry -> test.)
template <class Draws, bool PIN = true, bool LDS_LIST = false>
MCQ_HD bool mcq_iteration_ext_fast(const McqExtCtx &qc, const McqExtWaveCtx &wc, Draws &dr, const McqCard *cards,
                                   const uint32_t *sel8, const uint32_t *tf, const uint32_t *tops, const uint32_t *sd,
                                   McqLaneAcc &acc) {
    uint64_t deck = ((uint64_t)qc.fdeck_hi << 32) | qc.fdeck_lo;
    const uint32_t n = wc.cnt[0];
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const uint16_t *LdsList;
    const uint16_t *list_g = wc.list[0];
    const LdsList list_l = (LdsList)(uintptr_t)list_g; /* (the low half of a generic LDS address is the LDS address) */
#define MCQ_XLIST(i) (LDS_LIST ? (uint32_t)list_l[i] : (uint32_t)list_g[i])
#else
    const uint16_t *list_g = wc.list[0];
#define MCQ_XLIST(i) ((uint32_t)list_g[i])
#endif
    const uint32_t n_opp = PIN ? mcq_opaque_uniform(qc.n_players - 1u) : qc.n_players - 1u;
    McqHole opp[MCQ_MAX_OPP];
    bool dealt = true;
#define MCQ_XOPP(P)                                                                                                 \
    if (P < n_opp) {                                                                                                \
        const uint32_t dhi = (uint32_t)(deck >> 32), dlo = (uint32_t)deck;                                          \
        const uint32_t top = mcq_deck_top(dlo, dhi);                                                                \
        uint32_t c1 = 0, c2 = 0, trial = 0;                                                                         \
        bool ok = false;                                                                                            \
        for (; trial < MCQ_EXT_MAX_TRIALS && !ok; trial++) {                                                        \
            const uint32_t e = MCQ_XLIST(dr.pick(n));                                                               \
            c1 = e & 0xFFu;                                                                                         \
            c2 = e >> 8;                                                                                            \
            ok = (((deck >> c1) & (deck >> c2)) & 1u) != 0u && c2 != top; /* both still there, B not the highest */ \
        }                                                                                                           \
        acc.passes += trial; /* (the lane's own count: lanes leave the loop at different trials) */                 \
        dealt = dealt && ok;                                                                                        \
        if (ok && c2 > c1) c2 = mcq_deck_next(dlo, dhi, c2); /* deck.pop(r2) after deck.pop(r1), l.178-179 */       \
        /* (c1, c2 are entries of the list -- card ids below 52 -- or, with an empty trial budget, zero: no clamp) */ \
        deck &= ~(((uint64_t)1 << c1) | ((uint64_t)1 << c2));                                                       \
        opp[P].set(cards[c1], cards[c2]);                                                                           \
    }
    MCQ_XOPP(0) MCQ_XOPP(1) MCQ_XOPP(2) MCQ_XOPP(3) MCQ_XOPP(4) MCQ_XOPP(5) MCQ_XOPP(6) MCQ_XOPP(7) MCQ_XOPP(8)
#undef MCQ_XOPP
#undef MCQ_XLIST
    uint32_t dlo = (uint32_t)deck, dhi = (uint32_t)(deck >> 32);
    uint32_t L = mcq_popc(qc.fdeck_lo) + mcq_popc(qc.fdeck_hi) - 2u * n_opp; /* wave-uniform: no counting per lane */
    McqBoard b = qc.board;
    for (uint32_t k = 0; k < qc.n_deal; k++, L--) {
        const uint32_t c = mcq_select_pop(dlo, dhi, dr.table(k, L - 1u), sel8);
        b.add(cards[c < 52u ? c : 0u]);
    }
    McqFlushSel fs;
    fs.from_board(b);
    const uint32_t hk = mcq_eval_key(b, fs, qc.hero, tf, tops, sd);
    uint32_t best = 0;
    const uint32_t n_opp_e = PIN ? mcq_opaque_uniform(qc.n_players - 1u) : qc.n_players - 1u;
#define MCQ_XEVAL(P)                                                  \
    if (P < n_opp_e) {                                                \
        const uint32_t k = mcq_eval_key(b, fs, opp[P], tf, tops, sd); \
        best = k > best ? k : best;                                   \
    }
    MCQ_XEVAL(0) MCQ_XEVAL(1) MCQ_XEVAL(2) MCQ_XEVAL(3) MCQ_XEVAL(4) MCQ_XEVAL(5) MCQ_XEVAL(6) MCQ_XEVAL(7) MCQ_XEVAL(8)
#undef MCQ_XEVAL
    uint64_t won = hk >= best ? 1u : 0u;
    acc.types += won << (6u * (hk >> MCQ_KEY_SHIFT));
    acc.tie += hk == best ? 1u : 0u;
    return dealt;
}
