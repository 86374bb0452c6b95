// mcq_replay.hpp -- host half of MCQ_MODE_REPLAY_MT19937.
//
// The reference draws every random index from numpy's legacy global RandomState: one serial MT19937 stream
// whose consumption is data dependent (masked rejection inside randint, re-draw when r1 == r2).  To make the
// GPU tallies bit-exact against tools/montecarlo_python.py under np.random.seed(s), the host walks that stream
// once per query and hands the kernel the ACCEPTED draw values (one byte each, draw-major); the deal ->
// evaluate -> tally work stays on the GPU.  Rejected pairs only show up in `passes`, which is counted here.
//
//   numpy 1.26.4 legacy seeding      init_genrand(s)                 -> McqMt19937::seed
//   numpy legacy randint(0, n)       masked rejection, 1 word/trial  -> mcq_np_randint
//   montecarlo_python.py:165-176     opponent pair loop              -> mcq_replay_parse
//   montecarlo_python.py:185-189     table draws randint(0, len-1)   -> mcq_replay_parse
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/mcq.h"

struct McqMt19937 {
    uint32_t mt[624];
    uint32_t pos;

    void seed(uint32_t s) {
        mt[0] = s;
        for (uint32_t i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i;
        pos = 624;
    }
    static uint32_t twist(uint32_t u, uint32_t v) {
        uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
        return (y >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu);
    }
    void regenerate() {
        uint32_t k = 0;
        for (; k < 624 - 397; k++) mt[k] = mt[k + 397] ^ twist(mt[k], mt[k + 1]);
        for (; k < 623; k++) mt[k] = mt[k + 397 - 624] ^ twist(mt[k], mt[k + 1]);
        mt[623] = mt[396] ^ twist(mt[623], mt[0]);
        pos = 0;
    }
    uint32_t next() {
        if (pos >= 624) regenerate();
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        return y ^ (y >> 18);
    }
};

static inline uint32_t mcq_np_randint(McqMt19937 &g, uint32_t n) { /* value in [0, n-1] */
    uint32_t rng = n - 1;
    if (rng == 0) return 0;
    uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(rng);
    uint32_t v;
    do v = g.next() & mask;
    while (v > rng);
    return v;
}

static inline uint32_t mcq_draws_per_iteration(const mcq_query &q) {
    return 2u * (q.n_players - 1u) + (5u - q.n_board);
}

// Fills draws[d * stride + it] for it < q.runs, d < mcq_draws_per_iteration(q), continuing the stream of `g`
// (so consecutive queries can share one generator, as the reference's calls share numpy's global state);
// returns `passes`.
static inline uint64_t mcq_replay_parse_stream(const mcq_query &q, McqMt19937 &g, uint8_t *draws, size_t stride) {
    const uint32_t n_opp = q.n_players - 1u, n_deal = 5u - q.n_board;
    uint64_t passes = 0;
    for (uint32_t it = 0; it < q.runs; it++) {
        uint32_t L = 50u - q.n_board;
        uint8_t *p = draws + it;
        for (uint32_t o = 0; o < n_opp; o++) {
            uint32_t r1, r2;
            do {
                passes++;
                r1 = mcq_np_randint(g, L);
                r2 = mcq_np_randint(g, L - 1);
            } while (r1 == r2);
            p[0] = (uint8_t)(r1 | 0x80u); /* every draw travels as r | 0x80 */
            p[stride] = (uint8_t)(r2 | 0x80u);
            p += 2 * stride;
            L -= 2;
        }
        for (uint32_t k = 0; k < n_deal; k++) {
            p[0] = (uint8_t)(mcq_np_randint(g, L - 1) | 0x80u);
            p += stride;
            L -= 1;
        }
    }
    return passes;
}

// One query replaying np.random.seed(seed32).
static inline uint64_t mcq_replay_parse(const mcq_query &q, uint32_t seed32, uint8_t *draws, size_t stride) {
    McqMt19937 g;
    g.seed(seed32);
    return mcq_replay_parse_stream(q, g, draws, stride);
}

// ------------------------------------------------------------------------------------------------ extended queries
// Ranges / ghost cards / any number of known hands, each two cards or a range: the range tests depend on
// the cards, so the host walks the reference's loop literally on an explicit deck list (montecarlo_python.py:121-189)
// and emits the ACCEPTED draws, all converted to list.pop order: for a known hand given as a range the two looked-at
// cards deck[r1], deck[r2] (both on the unpopped list, l.142-148) become pops r1 and r2 - (r2 > r1).
struct McqExtDeck {
    uint8_t c[52];
    int n;
    uint8_t pop(int i) {
        uint8_t v = c[i];
        for (int k = i; k + 1 < n; k++) c[k] = c[k + 1];
        n--;
        return v;
    }
    void remove(uint8_t card) {
        for (int i = 0; i < n; i++)
            if (c[i] == card) { pop(i); return; }
    }
};

static inline bool mcq_ext_in_range(const uint32_t *bits, uint8_t a, uint8_t b) {
    int ra = a >> 2, rb = b >> 2, lo = ra < rb ? ra : rb, hi = ra < rb ? rb : ra;
    int i = ra == rb ? 14 * ra : ((a & 3) == (b & 3) ? 13 * lo + hi : 13 * hi + lo);
    return (bits[i >> 5] >> (i & 31)) & 1u;
}

/* draws per iteration: two per hand that is not given as cards (range hands and random opponents), then the table */
static inline uint32_t mcq_ext_draws_per_iteration(const mcq_query &q, const mcq_query_ext &e) {
    uint32_t lists = e.hero_is_range ? 0u : 1u;
    for (uint32_t k = 0; k < e.n_known; k++) lists += e.known[k].is_range ? 0u : 1u;
    return 2u * (q.n_players - lists) + (5u - q.n_board);
}

// returns passes, or UINT64_MAX when a range could not be dealt within max_trials attempts
static inline uint64_t mcq_replay_parse_ext(const mcq_query &q, const mcq_query_ext &e, McqMt19937 &g, uint8_t *draws,
                                            size_t stride, uint32_t max_trials) {
    McqExtDeck original;
    original.n = 52;
    for (int i = 0; i < 52; i++) original.c[i] = (uint8_t)i;
    if (e.ghost[0] != 0xFF) { original.remove(e.ghost[0]); original.remove(e.ghost[1]); }
    const uint32_t n_hands = 1u + e.n_known, n_deal = 5u - q.n_board;
    uint64_t passes = 0;
    for (uint32_t it = 0; it < q.runs; it++) {
        McqExtDeck d = original;
        uint8_t *p = draws + it;
        for (uint32_t i = 0; i < q.n_board; i++) d.remove(q.board[i]);
        for (uint32_t h = 0; h < q.n_players; h++) {
            const bool known = h < n_hands;
            const bool is_range = known ? (h == 0 ? e.hero_is_range != 0 : e.known[h - 1].is_range != 0) : true;
            if (!is_range) { /* l.150-161: by value, if it is still there */
                const uint8_t *cd = h == 0 ? q.hole : e.known[h - 1].cards;
                d.remove(cd[0]);
                d.remove(cd[1]);
                continue;
            }
            const uint32_t *set = !known ? e.opp_range : (h == 0 ? e.hero_range : e.known[h - 1].range);
            uint32_t r1, r2, trial = 0;
            for (;;) { /* l.136-148 / l.165-176: both indices on the UNPOPPED list */
                if (trial++ >= max_trials) return ~0ull;
                passes++;
                r1 = mcq_np_randint(g, (uint32_t)d.n);
                r2 = mcq_np_randint(g, (uint32_t)d.n - 1);
                if (r1 != r2 && mcq_ext_in_range(set, d.c[r1], d.c[r2])) break;
            }
            p[0] = (uint8_t)(r1 | 0x80u);
            if (known) { /* the two cards looked at leave by value: as pops, r1 and then r2 - (r2 > r1) */
                p[stride] = (uint8_t)((r2 - (r2 > r1 ? 1u : 0u)) | 0x80u);
                const uint8_t a = d.c[r1], b = d.c[r2];
                d.remove(a);
                d.remove(b);
            } else { /* deck.pop(r1); deck.pop(r2) on the shrunk list (l.178-179) */
                p[stride] = (uint8_t)(r2 | 0x80u);
                d.pop((int)r1);
                d.pop((int)r2);
            }
            p += 2 * stride;
        }
        for (uint32_t k = 0; k < n_deal; k++) {
            const uint32_t idx = mcq_np_randint(g, (uint32_t)d.n - 1);
            p[0] = (uint8_t)(idx | 0x80u);
            p += stride;
            d.pop((int)idx);
        }
    }
    return passes;
}
