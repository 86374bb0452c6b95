// mcq_mt.hpp -- MCQ_MODE_REPLAY_MT19937 on the device: one WAVE walks numpy's legacy random stream of one query.
//
// What is reproduced (reference paths relative to /root/reference; numpy 1.26.4 legacy RandomState is third party):
//   np.random.seed(s)                 init_genrand(s), MT19937                           -> mcq_mt_seed / mcq_mt_regenerate
//   np.random.randint(0, n)           rng = n - 1, mask = 2^k - 1 >= rng, one tempered 32-bit word per trial,
//                                     `do v = word & mask while v > rng`                  -> the accept test below
//   montecarlo_python.py:165-176      per opponent: repeat {passes += 1; r1 = randint(0, L); r2 = randint(0, L-1)}
//                                     until r1 != r2                                      -> the re-draw rewind below
//   montecarlo_python.py:185-189      per missing table card: randint(0, L-1)
//
// The stream is serial and its consumption data dependent (rejected words, re-drawn pairs), but the bound of a draw
// depends only on the draw's POSITION in the iteration, and whether a word is accepted hardly depends on that
// position (neighbouring bounds differ by one).  So a wave parses 64 words at a time by fixed-point iteration:
//   guess every lane's position p  ->  accept bit of every word under its guessed bound  ->  positions again as
//   p0 + (accepted words before the lane)  ->  repeat until nothing moves.
// Every round makes at least one more lane final (lane 0 always is), so the result is the sequential parse; four or
// five rounds are typical.  A pair that has to be drawn again (r1 == r2, one in fifty) ends the batch at its second
// word: the position rewinds by two, the words behind go back into the stream.  Accepted draws go, as r | 0x80 bytes, to a per-wave ring in LDS laid out like the
// draw-major global buffer and are flushed 64 iterations at a time with full-row stores; the evaluation kernel
// (mcq_eval_kernel<MCQ_MODE_REPLAY_MT19937>) consumes that buffer exactly as it consumed the host's.
//
// One source for two builds: hipcc compiles the MCQ_FOR_LANES regions for ONE lane (the thread), cross-lane steps
// are wave intrinsics; tests/hostsim compiles the same text with every per-lane variable as a 64-entry array and
// the regions as loops, which is how the parse is checked byte for byte against the sequential host walk
// (mcq_replay.hpp) where no GPU exists.
#pragma once
#include <stdint.h>

#include "mcq_device.hpp"

#define MCQ_MT_N 624u
#define MCQ_MT_M 397u
#define MCQ_MT_RING 128u   /* iterations the ring holds: < 64 pending + at most 64 from one batch of words */
#define MCQ_MT_MAX_DRAWS 23u /* 2 * 9 opponents + 5 table cards */

#if defined(__HIP_DEVICE_COMPILE__)
#define MCQ_PL(T, name) T name                /* a per-lane variable */
#define MCQ_L(name) name                      /* ... and its value in the current lane */
#define MCQ_AT(name, idx) mcq_mt_shfl(name, idx) /* ... and in lane idx (written in an EARLIER region) */
#define MCQ_AT_UNIFORM(name, idx) ((uint32_t)__builtin_amdgcn_readlane((int)(name), (int)(idx))) /* idx wave-uniform */
#define MCQ_FOR_LANES(l) for (uint32_t l __attribute__((unused)) = mcq_mt_lane(), once_ = 1; once_; once_ = 0)
#define MCQ_BALLOT(name) __ballot(name)
/* number of set bits of the wave mask m below this lane */
#define MCQ_COUNT_BELOW(m, l) __builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u))
#define MCQ_WAVE_SYNC()                                                  \
    do {                                                                 \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");           \
        __builtin_amdgcn_wave_barrier();                                 \
    } while (0)
__device__ __forceinline__ uint32_t mcq_mt_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mcq_mt_shfl(uint32_t v, uint32_t idx) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)v);
}
#else
#define MCQ_PL(T, name) T name[64]
#define MCQ_L(name) name[l_]
#define MCQ_AT(name, idx) name[(idx) & 63u]
#define MCQ_AT_UNIFORM(name, idx) name[(idx) & 63u]
#define MCQ_FOR_LANES(l) for (uint32_t l_ = 0, l __attribute__((unused)) = 0; l_ < 64u; l_++, l = l_)
#define MCQ_BALLOT(name) mcq_mt_host_ballot(name)
#define MCQ_COUNT_BELOW(m, l) mcq_mt_popc64((m) & (((uint64_t)1 << (l)) - 1u))
#define MCQ_WAVE_SYNC() ((void)0)
template <class T>
static inline uint64_t mcq_mt_host_ballot(const T (&a)[64]) {
    uint64_t m = 0;
    for (uint32_t i = 0; i < 64; i++) {
        const uint64_t bit = a[i] ? 1u : 0u; /* (in one expression, g++ 11 -fsanitize=shift,bounds yields all ones) */
        m |= bit << i;
    }
    return m;
}
#endif

MCQ_HD uint32_t mcq_mt_popc64(uint64_t x) { return mcq_popc((uint32_t)x) + mcq_popc((uint32_t)(x >> 32)); }
MCQ_HD uint32_t mcq_mt_top64(uint64_t x) { /* index of the highest set bit, x != 0 */
    const uint32_t hi = (uint32_t)(x >> 32);
    return hi ? 63u - mcq_clz(hi) : 31u - mcq_clz((uint32_t)x);
}
MCQ_HD uint32_t mcq_mt_low64(uint64_t x) { /* index of the lowest set bit, x != 0 */
    return mcq_mt_top64(x & (0 - x));
}

MCQ_HD uint32_t mcq_mt_twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu);
}
MCQ_HD uint32_t mcq_mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    return y ^ (y >> 18);
}

// Draw number d (0-based within an iteration) of a query with deck length L0 = 50 - n_board, n_opp opponents:
// bound n of its randint(0, n), packed as rng (= n - 1) | is_r2 << 8 | mask << 16.  rng >= 26 for every legal query
// (ten players on a flop leave 27 cards for the last table draw), so randint's "rng == 0 consumes nothing" case
// cannot occur and every attempt is exactly one word.
MCQ_HD uint32_t mcq_mt_draw_entry(uint32_t L0, uint32_t n_opp, uint32_t d) {
    uint32_t n, r2 = 0;
    if (d < 2u * n_opp) {
        n = L0 - 2u * (d >> 1) - (d & 1u); /* r1: randint(0, L), r2: randint(0, L - 1) (l.169-170) */
        r2 = d & 1u;
    } else {
        n = L0 - 2u * n_opp - (d - 2u * n_opp) - 1u; /* randint(0, len(deck) - 1) (l.188) */
    }
    const uint32_t rng = n - 1u;
    const uint32_t mask = 0xFFFFFFFFu >> mcq_clz(rng | 1u);
    return rng | (r2 << 8) | (mask << 16);
}

// floor(p / D) for p < 128, 1 <= D <= 23, as (p * magic) >> 16 (checked exhaustively in the tests)
MCQ_HD uint32_t mcq_mt_magic(uint32_t D) { return 65536u / D + 1u; }

// Per-wave working set (device: LDS).
struct McqMtWave {
    uint32_t mt[MCQ_MT_N];
    uint32_t entry[24];                             /* mcq_mt_draw_entry per draw */
    uint8_t ring[MCQ_MT_MAX_DRAWS * MCQ_MT_RING];   /* ring[d * 128 + (iteration & 127)] = r | 0x80 */
    uint8_t pad_[16];
};

// np.random.seed(s): init_genrand.  A serial recurrence: every lane computes it (wave-uniform, scalar ALU on the
// device) and lane 0 stores.
template <class W>
MCQ_HD void mcq_mt_seed(W &w, uint32_t s) {
    uint32_t x = s;
    w.mt[0] = x;
    for (uint32_t i = 1; i < MCQ_MT_N; i++) {
        x = 1812433253u * (x ^ (x >> 30)) + i;
        w.mt[i] = x;
    }
}

// The next 624 state words.  new[k] needs old[k], old[k+1] and [k+397]: old for k < 227, else new[k-227] -- three
// sweeps of at most 227 independent elements each, 64 lanes at a time.  Within a 64-element step every lane
// reads before any lane writes (a wave executes in lockstep), and a step only reads elements of later steps.
template <class W>
MCQ_HD void mcq_mt_regenerate(W &w) {
    for (uint32_t base = 0; base < MCQ_MT_N; base += 64u) {
        /* steps must not straddle the sweep boundaries 227 and 454 (a lane would read a word an earlier lane of
         * the same step has yet to write), nor 623 */
        uint32_t lim = base < 227u ? 227u : base < 454u ? 454u : 623u;
        if (lim > base + 64u) lim = base + 64u;
        MCQ_PL(uint32_t, nv);
        MCQ_FOR_LANES(l) {
            const uint32_t k = base + l;
            MCQ_L(nv) = 0;
            if (k < lim) {
                const uint32_t far = k < 227u ? k + MCQ_MT_M : k - 227u;
                MCQ_L(nv) = w.mt[far] ^ mcq_mt_twist(w.mt[k], w.mt[k + 1u]);
            }
        }
        MCQ_WAVE_SYNC();
        MCQ_FOR_LANES(l) {
            const uint32_t k = base + l;
            if (k < lim) w.mt[k] = MCQ_L(nv);
        }
        MCQ_WAVE_SYNC();
        if (lim < base + 64u && lim < 623u) base = lim - 64u; /* next step starts at the sweep boundary */
    }
    MCQ_FOR_LANES(l) {
        if (l == 0) w.mt[623] = w.mt[396] ^ mcq_mt_twist(w.mt[623], w.mt[0]);
    }
    MCQ_WAVE_SYNC();
}

struct McqMtState { /* wave-uniform */
    uint32_t pos;     /* next unread state word, 624 = regenerate first */
    uint32_t it_done; /* complete iterations parsed */
    uint32_t d0;      /* draws of the current iteration already accepted */
    uint32_t v_last;  /* value of the latest accepted word (the r1 an r2 is compared with) */
    uint32_t flushed; /* iterations already written to the global buffer */
    uint64_t passes;
};

// 64 iterations [st.flushed, st.flushed + 64) (or the last `count` < 64) from the ring to draws[d * stride + it]:
// per draw one row of 64 bytes, 16 lanes x 4 bytes, four rows per step.
template <class W>
MCQ_HD void mcq_mt_flush(W &w, McqMtState &st, uint32_t D, uint32_t count, uint8_t *draws, uint64_t stride) {
    const uint32_t first = st.flushed;
    for (uint32_t d4 = 0; d4 < D; d4 += 4u) {
        MCQ_FOR_LANES(l) {
            const uint32_t d = d4 + (l >> 4), c4 = (l & 15u) * 4u;
            if (d < D && c4 < count) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(&w.ring[d * MCQ_MT_RING + ((first + c4) & (MCQ_MT_RING - 1u))]);
                *reinterpret_cast<uint32_t *>(draws + (uint64_t)d * stride + first + c4) = v;
            }
        }
    }
    MCQ_WAVE_SYNC();
    st.flushed += count;
}

// Parse the whole stream of one query: `runs` iterations of D = 2 * n_opp + n_deal draws (D >= 1).  The wave's MT
// state must be seeded (st.pos = 624).  draws: this query's block of the global buffer, stride = runs rounded up to
// 64 (rows may be written up to the stride).  Returns passes in st.passes.
template <class W>
MCQ_HD void mcq_mt_parse_query(W &w, McqMtState &st, uint32_t L0, uint32_t n_opp, uint32_t n_deal, uint32_t runs,
                               uint8_t *draws, uint64_t stride) {
    const uint32_t D = 2u * n_opp + n_deal, magic = mcq_mt_magic(D);
    MCQ_FOR_LANES(l) {
        if (l < D) w.entry[l] = mcq_mt_draw_entry(L0, n_opp, l);
    }
    MCQ_WAVE_SYNC();
    while (st.it_done < runs) {
        if (st.pos >= MCQ_MT_N) {
            mcq_mt_regenerate(w);
            st.pos = 0;
        }
        const uint32_t cnt = MCQ_MT_N - st.pos < 64u ? MCQ_MT_N - st.pos : 64u;
        const uint32_t left = runs - st.it_done; /* >= 1: iterations still to parse */
#ifdef MCQ_MT_STATS
        g_batches++;
#endif
        MCQ_PL(uint32_t, y);
        MCQ_PL(uint32_t, p);
        MCQ_FOR_LANES(l) {
            MCQ_L(y) = l < cnt ? mcq_mt_temper(w.mt[st.pos + l]) : 0u;
            MCQ_L(p) = st.d0 + ((l * 44u) >> 6); /* first guess: about two words in three are accepted */
        }
        MCQ_PL(uint32_t, v);
        MCQ_PL(uint32_t, dd);
        MCQ_PL(uint32_t, itr);
        MCQ_PL(bool, acc);
        MCQ_PL(bool, isr2);
        uint64_t M = 0;
        for (;;) { /* positions of the words as if no pair were ever drawn again (that is settled afterwards) */
            MCQ_FOR_LANES(l) {
                const uint32_t q = (MCQ_L(p) * magic) >> 16;
                const uint32_t d = MCQ_L(p) - q * D;
                const uint32_t e = w.entry[d];
                MCQ_L(itr) = q;
                MCQ_L(dd) = d;
                MCQ_L(v) = MCQ_L(y) & (e >> 16);
                MCQ_L(isr2) = (e & 0x100u) != 0u;
                MCQ_L(acc) = l < cnt && MCQ_L(v) <= (e & 0xFFu) && q < left; /* words past the last iteration stay unread */
            }
            M = MCQ_BALLOT(acc);
            MCQ_PL(bool, moved);
            MCQ_FOR_LANES(l) {
                const uint32_t pn = st.d0 + MCQ_COUNT_BELOW(M, l);
                MCQ_L(moved) = pn != MCQ_L(p);
                MCQ_L(p) = pn;
            }
#ifdef MCQ_MT_STATS
            g_rounds++;
#endif
            if (MCQ_BALLOT(moved) == 0) break;
        }
        /* r1 == r2: the pair is drawn again (l.171-176) -- about one pair in fifty.  The parse is right up to and
         * including the first such word; the batch ends there (the words behind it go back), the position rewinds by
         * the pair, and the two draws of the pair are dropped: their slots are written by the pair drawn again. */
        MCQ_PL(uint32_t, prev_ix);
        MCQ_FOR_LANES(l) {
            const uint64_t below = M & (((uint64_t)1 << l) - 1u);
            MCQ_L(prev_ix) = below ? mcq_mt_top64(below) : 64u;
        }
        MCQ_PL(bool, red);
        MCQ_FOR_LANES(l) {
            const uint32_t sv = MCQ_AT(v, MCQ_L(prev_ix)); /* unconditional: every lane takes part in the exchange */
            const uint32_t vp = MCQ_L(prev_ix) < 64u ? sv : st.v_last;
            MCQ_L(red) = MCQ_L(acc) && MCQ_L(isr2) && MCQ_L(v) == vp;
        }
        const uint64_t R = MCQ_BALLOT(red);
        uint32_t used = cnt, rewind = 0, drop_a = 64u, drop_b = 64u; /* words of the batch that count; lanes dropped */
        if (R) {
            drop_b = mcq_mt_low64(R);
            drop_a = MCQ_AT_UNIFORM(prev_ix, drop_b);
            used = drop_b + 1u;
            rewind = 2u;
            M &= ((uint64_t)2 << drop_b) - 1u;
        }
        MCQ_FOR_LANES(l) {
            if (MCQ_L(acc) && l < used && l != drop_a && l != drop_b)
                w.ring[MCQ_L(dd) * MCQ_MT_RING + ((st.it_done + MCQ_L(itr)) & (MCQ_MT_RING - 1u))] = (uint8_t)(MCQ_L(v) | 0x80u);
        }
        MCQ_PL(bool, pass);
        MCQ_FOR_LANES(l) { MCQ_L(pass) = MCQ_L(acc) && MCQ_L(isr2) && l < used; } /* one accepted r2 per attempt (l.168) */
        st.passes += mcq_mt_popc64(MCQ_BALLOT(pass));
        const uint32_t p_end = st.d0 + mcq_mt_popc64(M) - rewind;
        const uint32_t it_add = (p_end * magic) >> 16;
        st.d0 = p_end - it_add * D;
        st.it_done += it_add;
        if (M) st.v_last = MCQ_AT_UNIFORM(v, mcq_mt_top64(M));
        /* words consumed: up to the re-drawn pair, else everything up to the last accepted word and the rejected words
         * behind it unless the stream of this query ends there (the last iteration is complete) */
        st.pos += R ? used : (st.it_done < runs ? cnt : (M ? mcq_mt_top64(M) + 1u : 0u));
        MCQ_WAVE_SYNC();
        while (st.it_done - st.flushed >= 64u) mcq_mt_flush(w, st, D, 64u, draws, stride);
    }
    if (st.it_done > st.flushed) mcq_mt_flush(w, st, D, st.it_done - st.flushed, draws, stride);
}
