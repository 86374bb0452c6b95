// mcq_device.hpp -- per-lane arithmetic of the equity kernels: RNG front ends, k-th-card selection on the
// 52-bit deck mask, the branch-light 7-card ranking key, and one Monte-Carlo iteration.
//
// Everything here is written against three tiny primitives (popcount, count-leading-zeros, 32x32 high
// multiply) so that the very same source is compiled for gfx950 by hipcc (the product) and, by
// tests/hostsim only, for the host compiler to unit-test the lane arithmetic where no GPU exists.
//
// What it reproduces (reference paths relative to /root/reference):
//   tools/montecarlo_python.py:121-189  dealing order and index semantics (see mcq_iteration)
//   tools/hand_evaluator.py:27-119      _calc_score ordering incl. its quirks (see mcq_eval7)
//   tools/hand_evaluator.py:20-24       ties go to the first hand = hero
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

#define MCQ_STREAM_ITERS 16u /* iterations per RNG stream (MCQ-CTR v1) */
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
MCQ_HD uint32_t mcq_clz(uint32_t x) { /* x != 0 */
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clz((int)x);
#else
    return (uint32_t)__builtin_clz(x);
#endif
}
MCQ_HD uint32_t mcq_mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
MCQ_HD uint32_t mcq_rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
MCQ_HD uint32_t mcq_topbit(uint32_t m) { return 0x80000000u >> mcq_clz(m); } /* m != 0 */
MCQ_HD uint32_t mcq_droplow(uint32_t m) { return m & (m - 1); }

// ------------------------------------------------------------------------------------------ lookup tables
// Staged in LDS by the kernels.  sel8[v]: positions of the set bits of byte v, 3 bits each (j-th set bit at
// bits 3j..3j+2).  suit[c]: the bit of card c in the suit-major hand layout (lo = clubs | diamonds << 16,
// hi = hearts | spades << 16, bit r of each half-word = rank r).
struct McqLuts {
    uint32_t sel8[256];
    uint32_t suit_lo[64];
    uint32_t suit_hi[64];
};

static inline void mcq_fill_luts(McqLuts *t) {
    for (uint32_t v = 0; v < 256; v++) {
        uint32_t e = 0, j = 0;
        for (uint32_t b = 0; b < 8; b++)
            if (v >> b & 1) e |= b << (3 * j++);
        t->sel8[v] = e;
    }
    for (uint32_t c = 0; c < 64; c++) {
        uint32_t pos = ((c & 3) << 4) | (c >> 2);
        t->suit_lo[c] = (c < 52 && pos < 32) ? 1u << pos : 0;
        t->suit_hi[c] = (c < 52 && pos >= 32) ? 1u << (pos - 32) : 0;
    }
}

// ------------------------------------------------------------------------------------------ RNG: MCQ-CTR v1
MCQ_HD void mcq_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t h0 = mcq_mulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = mcq_mulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct McqXoshiro { /* xoshiro128++ (Blackman & Vigna) */
    uint32_t s0, s1, s2, s3;
    MCQ_HDM void seed(uint64_t seed, uint64_t qid, uint32_t stream) {
        uint32_t o[4];
        mcq_philox4x32_10((uint32_t)qid, (uint32_t)(qid >> 32), stream, 0x4D435131u, (uint32_t)seed,
                          (uint32_t)(seed >> 32), o);
        s0 = o[0]; s1 = o[1]; s2 = o[2]; s3 = o[3];
        if ((s0 | s1 | s2 | s3) == 0) s0 = 1;
    }
    MCQ_HDM uint32_t next() {
        uint32_t result = mcq_rotl(s0 + s3, 7) + s0;
        uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t;
        s3 = mcq_rotl(s3, 11);
        return result;
    }
};

// Draw policy of the production mode: same procedure as montecarlo_python.py:165-176 / :188 with
// mulhi32(word, n) in place of numpy's masked rejection.
struct McqCtrDraws {
    McqXoshiro rng;
    MCQ_HDM void pair(uint32_t L, uint32_t &r1, uint32_t &r2, uint32_t &passes) {
        do {
            passes++;
            r1 = mcq_mulhi(rng.next(), L);
            r2 = mcq_mulhi(rng.next(), L - 1);
        } while (r1 == r2);
    }
    MCQ_HDM uint32_t single(uint32_t n) { return mcq_mulhi(rng.next(), n); }
};

// Draw policy of the parity mode: the host has already turned the MT19937 stream into the accepted draw
// values (one byte each, draw-major: draws[d * stride + iteration]); rejected pairs only count in `passes`,
// which the host supplies.
struct McqReplayDraws {
    const uint8_t *p; /* &draws[iteration] */
    uint64_t stride;
    MCQ_HDM void pair(uint32_t, uint32_t &r1, uint32_t &r2, uint32_t &) {
        r1 = p[0];
        r2 = p[stride];
        p += 2 * stride;
    }
    MCQ_HDM uint32_t single(uint32_t) {
        uint32_t v = p[0];
        p += stride;
        return v;
    }
};

// ------------------------------------------------------------------------------------------ deck
// The deck is a 52-bit mask in card-id order (= the reference's list order); list.pop(k) of the ordered
// remaining deck is "find the k-th set bit, clear it".  Returns the card id.
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
    uint32_t pos = base + ((e >> (3 * k)) & 7u);
    uint32_t bit = 1u << (pos & 31u);
    dlo &= up ? 0xFFFFFFFFu : ~bit;
    dhi &= up ? ~bit : 0xFFFFFFFFu;
    return pos;
}

// ------------------------------------------------------------------------------------------ evaluator
// bit i of the result set <=> ranks i-1 .. i+3 all present (rank -1 = ace playing low)
MCQ_HD uint32_t mcq_straight_runs(uint32_t m) {
    uint32_t m2 = (m << 1) | (m >> 12);
    uint32_t r1 = m2 & (m2 >> 1);
    uint32_t r2 = r1 & (r1 >> 2);
    return r2 & (m2 >> 4);
}

// 32-bit key whose unsigned order equals Python's order of _calc_score's (score, card_ranks) tuples
// (hand_evaluator.py:27-119) for 7 distinct cards.  key >> 28 = by_type index.  Every card_ranks vector that
// is a strictly descending sequence is encoded as a rank bit mask (lexicographic order of such sequences,
// including the shorter-is-smaller rule, equals integer order of the masks); (count-major) prefixes such as
// the pair or trips rank sit above the kicker mask.
//   lo = clubs | diamonds << 16, hi = hearts | spades << 16; bit r = rank r (0 = deuce .. 12 = ace).
MCQ_HD uint32_t mcq_eval7(uint32_t lo, uint32_t hi) {
    uint32_t X = lo ^ hi, A = lo & hi; /* half-adders of (clubs,hearts) and (diamonds,spades), both halves at once */
    uint32_t x0 = X & 0xFFFFu, x1 = X >> 16, a0 = A & 0xFFFFu, a1 = A >> 16;
    uint32_t any = (lo | hi);
    any = (any | (any >> 16)) & 0xFFFFu;
    uint32_t ge2 = a0 | a1 | (x0 & x1);
    uint32_t ge3 = (a0 & (x1 | a1)) | (a1 & x0);
    uint32_t eq4 = a0 & a1;

    uint32_t s0 = lo & 0xFFFFu, s1 = lo >> 16, s2 = hi & 0xFFFFu, s3 = hi >> 16;
    uint32_t fl = mcq_popc(s0) >= 5 ? s0 : mcq_popc(s1) >= 5 ? s1 : mcq_popc(s2) >= 5 ? s2 : mcq_popc(s3) >= 5 ? s3 : 0u;

    uint32_t n2 = mcq_popc(ge2);
    uint32_t key;
    if (fl != 0 && mcq_straight_runs(fl) != 0) {
        /* StraightFlush: ALL ranks of the suit, plus the -1 slot when it holds the ace (l.71-80, l.93) */
        key = (8u << 28) | (fl << 1) | (fl >> 12);
    } else if (eq4 != 0) {
        /* FoufOfAKind: the two highest distinct ranks of all seven cards (l.43-46) */
        uint32_t t = mcq_topbit(any);
        key = (7u << 28) | t | mcq_topbit(any ^ t);
    } else if (ge3 != 0 && n2 >= 2) {
        /* FullHouse: (trips, best remaining pair or second trips) (l.36-38) */
        uint32_t t = mcq_topbit(ge3);
        key = (6u << 28) | (t << 13) | mcq_topbit(ge2 ^ t);
    } else if (fl != 0) {
        /* Flush: top five ranks of the suit (l.98-100) */
        uint32_t n = mcq_popc(fl), f = fl;
        f = n > 5 ? mcq_droplow(f) : f;
        f = n > 6 ? mcq_droplow(f) : f;
        key = (5u << 28) | f;
    } else {
        uint32_t runs = mcq_straight_runs(any);
        if (runs != 0) {
            /* Straight: decided by its top rank (l.52-58); wheel = lowest */
            key = (4u << 28) | (32u - mcq_clz(runs));
        } else if (ge3 != 0) {
            /* ThreeOfAKind: trips, two kickers (l.104-106) */
            uint32_t s = mcq_droplow(mcq_droplow(any ^ ge3));
            key = (3u << 28) | (ge3 << 13) | s;
        } else if (n2 >= 2) {
            /* TwoPair: two best pairs, kicker = best of everything else incl. a third pair (l.39-42, l.107-109) */
            uint32_t P = n2 == 3 ? mcq_droplow(ge2) : ge2;
            key = (2u << 28) | (P << 13) | mcq_topbit(any ^ P);
        } else if (ge2 != 0) {
            /* Pair: pair, three kickers (l.110-112) */
            uint32_t s = mcq_droplow(mcq_droplow(any ^ ge2));
            key = (1u << 28) | (ge2 << 13) | s;
        } else {
            /* HighCard: top five (l.113-115) */
            key = mcq_droplow(mcq_droplow(any));
        }
    }
    return key;
}

// ------------------------------------------------------------------------------------------ one query, one lane
struct McqQueryCtx { /* wave-uniform */
    uint32_t deck_lo, deck_hi; /* remaining deck after removing the known table cards and hero (l.126-161) */
    uint32_t L0;               /* its length: 50 - n_board */
    uint32_t n_opp;            /* n_players - 1 */
    uint32_t n_deal;           /* 5 - n_board table cards still to come */
    uint32_t hero_lo, hero_hi; /* hero's two cards, suit-major */
    uint32_t board_lo, board_hi;
    uint32_t runs;
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

MCQ_HD void mcq_query_ctx(const McqQueryWords &q, const uint32_t *suit_lo, const uint32_t *suit_hi, McqQueryCtx &c) {
    uint64_t deck = (1ull << 52) - 1;
    c.board_lo = c.board_hi = c.hero_lo = c.hero_hi = 0;
    for (uint32_t i = 0; i < 2u + q.n_board(); i++) {
        uint32_t cd = q.card(i);
        deck &= ~(1ull << cd);
        if (i < 2) {
            c.hero_lo |= suit_lo[cd];
            c.hero_hi |= suit_hi[cd];
        } else {
            c.board_lo |= suit_lo[cd];
            c.board_hi |= suit_hi[cd];
        }
    }
    c.deck_lo = (uint32_t)deck;
    c.deck_hi = (uint32_t)(deck >> 32);
    c.L0 = 50u - q.n_board();
    c.n_opp = q.n_players() - 1u;
    c.n_deal = 5u - q.n_board();
    c.runs = q.runs();
}

static inline McqQueryWords mcq_query_words(const mcq_query &q) { /* host side */
    McqQueryWords w;
    __builtin_memcpy(&w, &q, 16);
    return w;
}

struct McqLaneAcc {
    uint64_t types; /* 9 fields of 6 bits: hero's winning hand types (<= 16 per lane per task) */
    uint32_t tie;
    uint32_t passes;
};

// One Monte-Carlo iteration of one lane.  Store keeps each opponent's two cards (suit-major) until the table
// is complete: the reference deals ALL opponents before any table card (montecarlo_python.py:215-217).
template <class Draws, class Store>
MCQ_HD void mcq_iteration(const McqQueryCtx &qc, Draws &dr, const uint32_t *sel8, const uint32_t *suit_lo,
                          const uint32_t *suit_hi, Store &st, McqLaneAcc &acc) {
    uint32_t dlo = qc.deck_lo, dhi = qc.deck_hi, L = qc.L0;
    for (uint32_t p = 0; p < qc.n_opp; p++) {
        uint32_t r1, r2;
        dr.pair(L, r1, r2, acc.passes);                   /* r1 in [0,L-1], r2 in [0,L-2], r1 != r2 (l.167-176) */
        uint32_t c1 = mcq_select_pop(dlo, dhi, r1, sel8); /* deck.pop(r1) (l.178) */
        uint32_t c2 = mcq_select_pop(dlo, dhi, r2, sel8); /* deck.pop(r2) on the shrunk list (l.179) */
        st.put(p, suit_lo[c1] | suit_lo[c2], suit_hi[c1] | suit_hi[c2]);
        L -= 2;
    }
    uint32_t blo = qc.board_lo, bhi = qc.board_hi;
    for (uint32_t k = 0; k < qc.n_deal; k++) {
        uint32_t c = mcq_select_pop(dlo, dhi, dr.single(L - 1), sel8); /* randint(0, len-1): never the last card (l.188) */
        blo |= suit_lo[c];
        bhi |= suit_hi[c];
        L -= 1;
    }
    uint32_t hk = mcq_eval7(qc.hero_lo | blo, qc.hero_hi | bhi);
    uint32_t best = 0;
    for (uint32_t p = 0; p < qc.n_opp; p++) {
        uint32_t lo, hi;
        st.get(p, lo, hi);
        uint32_t k = mcq_eval7(lo | blo, hi | bhi);
        best = k > best ? k : best;
    }
    uint64_t won = hk >= best ? 1u : 0u; /* ties go to hero (hand_evaluator.py:23) */
    acc.types += won << (6u * (hk >> 28));
    acc.tie += hk == best ? 1u : 0u;
}
