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
// Every round makes at least one more lane final (lane 0 always is), so the result is the sequential parse; three
// rounds are typical (the first guess takes every word at the query's middle depth).  A pair that has to be drawn
// again (r1 == r2, one in fifty) ends the batch at its second word: the position rewinds by two, the words behind go
// back into the stream.  Accepted draws go, as r | 0x80 bytes, to a per-wave ring in LDS laid out like the draw-major
// global buffer -- where an accepted r2 finds its r1 one row below -- and are flushed 64 iterations at a time with
// full-row stores; the evaluation kernel (mcq_eval_kernel<MCQ_MODE_REPLAY_MT19937>) consumes that buffer, four
// iterations per 32-bit load (McqReplayDraws4).
//
// One source for two builds: hipcc compiles the MCQ_FOR_LANES regions for ONE lane (the thread), cross-lane steps
// are wave intrinsics; tests/hostsim compiles the same text with every per-lane variable as a 64-entry array and
// the regions as loops, which is how the parse is checked byte for byte against the sequential host walk
// (mcq_replay.hpp) where no GPU exists.
#pragma once
#include <stdint.h>

#include "mcq_device.hpp"

#ifndef MCQ_MT_STAMP
#define MCQ_MT_STAMP(k) /* tuning builds (tools/mt_bench) take a cycle-counter stamp here */
#endif
#define MCQ_MT_N 624u
#define MCQ_MT_M 397u
#define MCQ_MT_RING 128u   /* iterations the ring holds: < 64 pending + at most 64 from one batch of words */
#define MCQ_MT_ROW 132u    /* bytes from one ring row to the next: 128 + 4, so that the rows of ONE iteration -- what the
                            * lanes of a batch write and read side by side -- fall into different LDS banks (at 128 they
                            * all met in one bank: 44 % of the LDS cycles of the walk were bank conflicts) */
#define MCQ_MT_MAX_DRAWS 23u /* 2 * 9 opponents + 5 table cards */
#define MCQ_MT_POSITIONS 128u /* positions a batch can form: d0 <= 22, + 64 words, rounded up */
#define MCQ_MT_FLUSH(w, st, D, n, draws, stride) mcq_mt_flush(w, st, D, n, draws, stride)

#if defined(__HIP_DEVICE_COMPILE__)
#define MCQ_PL(T, name) T name                /* a per-lane variable */
#define MCQ_L(name) name                      /* ... and its value in the current lane */
#define MCQ_AT(name, idx) mcq_mt_shfl(name, idx) /* ... and in lane idx (written in an EARLIER region) */
#define MCQ_AT_UNIFORM(name, idx) ((uint32_t)__builtin_amdgcn_readlane((int)(name), (int)(idx))) /* idx wave-uniform */
#define MCQ_FOR_LANES(l) for (uint32_t l __attribute__((unused)) = mcq_mt_lane(), once_ = 1; once_; once_ = 0)
#define MCQ_BALLOT(name) __ballot(name)
#define MCQ_BALLOT_K(name, k) __ballot(name[k])
#define MCQ_BALLOT_OF(l, expr) ({ const uint32_t l __attribute__((unused)) = mcq_mt_lane(); __ballot(expr); }) /* ballot of an expression of the lane's values */
#define MCQ_LANE_OF(m, l) __builtin_amdgcn_inverse_ballot_w64(m) /* is this lane's bit set in the wave mask m?  (a condition on
                                                                  * it makes m the EXEC mask: no vector instruction at all) */
/* number of set bits of the wave mask m below this lane */
#define MCQ_COUNT_BELOW(m, l) __builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u))
#define MCQ_WAVE_SYNC()                                                  \
    do {                                                                 \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");           \
        __builtin_amdgcn_wave_barrier();                                 \
    } while (0)
/* positions travel as the LDS byte address of their ptab word (device) / as four times the index (host): a round asks
 * for that word with no further address arithmetic */
typedef __attribute__((address_space(3))) const uint8_t *McqLdsU8;
typedef __attribute__((address_space(3))) const uint32_t *McqLdsU32;
#define MCQ_POS_BASE(w) ((uint32_t)(uintptr_t)(McqLdsU32)((w).ptab))
#define MCQ_PTAB_AT(w, pa4) (*(McqLdsU32)(uintptr_t)(pa4))
#define MCQ_POS4_FROM(m, l, base4) ((base4) + 4u * __builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u)))
/* bit `lane` of a wave mask as a per-lane condition: v_cmp of mbcnt difference would cost more than the mask AND the
 * compiler makes of this */
__device__ __forceinline__ bool mcq_mt_lane_of(uint64_t m) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return ((m >> lane) & 1u) != 0u;
}
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
#define MCQ_BALLOT_K(name, k) mcq_mt_host_ballot_k(name, k)
#define MCQ_BALLOT_OF(l, expr) ({ uint64_t m_ = 0; for (uint32_t l_ = 0, l __attribute__((unused)) = 0; l_ < 64u; l_++, l = l_) m_ |= (uint64_t)((expr) ? 1u : 0u) << l_; m_; })
#define MCQ_LANE_OF(m, l) ((((m) >> (l)) & 1u) != 0u)
#define MCQ_COUNT_BELOW(m, l) mcq_mt_popc64((m) & (((uint64_t)1 << (l)) - 1u))
#define MCQ_WAVE_SYNC() ((void)0)
#define MCQ_POS_BASE(w) 0u
#define MCQ_PTAB_AT(w, pa4) ((w).ptab[(pa4) >> 2])
#define MCQ_POS4_FROM(m, l, base4) ((base4) + 4u * MCQ_COUNT_BELOW(m, l))
template <class T>
static inline uint64_t mcq_mt_host_ballot(const T (&a)[64]) {
    uint64_t m = 0;
    for (uint32_t i = 0; i < 64; i++) {
        const uint64_t bit = a[i] ? 1u : 0u; /* (in one expression, g++ 11 -fsanitize=shift,bounds yields all ones) */
        m |= bit << i;
    }
    return m;
}
template <class T, size_t N>
static inline uint64_t mcq_mt_host_ballot_k(const T (&a)[64][N], uint32_t k) {
    uint64_t m = 0;
    for (uint32_t i = 0; i < 64; i++) {
        const uint64_t bit = a[i][k] ? 1u : 0u;
        m |= bit << i;
    }
    return m;
}
#endif

/* (device: the arguments are wave-uniform masks -- one scalar instruction each) */
MCQ_HD uint32_t mcq_mt_popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
MCQ_HD uint32_t mcq_mt_top64(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); } /* index of the highest set bit, x != 0 */
MCQ_HD uint32_t mcq_mt_low64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }        /* index of the lowest set bit, x != 0 */

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
// the bound n of its randint(0, n) is L0 - e - 1 + 1 with the EFFECTIVE DEPTH e = d + [d >= 2 n_opp]
//   r1: randint(0, L), r2: randint(0, L - 1) (l.169-170): n = L0 - d;  table: randint(0, len(deck) - 1) (l.188): n = L0 - d - 1
// so rng = n - 1 = L0 - 1 - e, strictly falling with d.  rng >= 26 for every legal query (ten players on a flop leave
// 27 cards for the last table draw), so randint's "rng == 0 consumes nothing" case cannot occur and every attempt is
// exactly one word; the mask is 63 while rng >= 32 (e <= L0 - 33) and 31 behind that ("zone 31": only with eight or
// nine opponents).  A word with low bits y63 = y & 63 is therefore accepted at depth e of zone 63 iff e <= L0 - 1 - y63.
MCQ_HD uint32_t mcq_mt_depth(uint32_t n_opp, uint32_t d) { return d + (d >= 2u * n_opp ? 1u : 0u); }

// floor(p / D) for p < MCQ_MT_POSITIONS, 1 <= D <= 23, as (p * magic) >> 16 (checked exhaustively in the tests)
MCQ_HD uint32_t mcq_mt_magic(uint32_t D) { return 65536u / D + 1u; }

// Per-wave working set (device: LDS).
#define MCQ_MT_ZONE31 0x80u
struct McqMtWave {
    uint32_t mt[MCQ_MT_N + 64u]; /* + 64: a batch reads 64 words from its position, the ones past the block unused */
    /* by position p = (draws of the current iteration already accepted) + (accepted words of the batch before the lane),
     * p < MCQ_MT_POSITIONS, ONE word per position (a round asks for it with one LDS read -- the LDS pipe, 70 % busy, is
     * what bounds the walk -- and when the positions have settled, what the write-out needs has arrived with it):
     *   bits 0-7    the effective depth e of draw d = p mod D, | MCQ_MT_ZONE31 in zone 31
     *   bits 8-15   the iteration offset p / D
     *   bits 16-27  d * MCQ_MT_ROW: the ring offset of the row BELOW the draw's own (its partner's row)
     *   bit 31      set for an r2 */
    uint32_t ptab[MCQ_MT_POSITIONS];
    /* ring[(d + 1) * MCQ_MT_ROW + (iteration & 127)] = r | 0x80; row 0 is spare: the partner of draw d -- the r1 an r2 is
     * compared with -- sits one row below it, and the address must exist for d = 0 */
    uint8_t ring[(MCQ_MT_MAX_DRAWS + 1u) * MCQ_MT_ROW];
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

// The next 624 state words.  new[k] needs old[k], old[k + 1] and old[k + 397] -- or, from k = 227 on, new[k - 227]: three
// sweeps of at most 227 words, each waiting for the one before.  But the word a sweep waits for is the one the SAME
// lane made in the sweep before (k - 227 is its word there), so a lane makes the chains new[t], new[227 + t], new[454 + t]
// of four t = 64 s + lane out of registers, from words of the OLD state that it reads all at once -- indices clamped,
// no read behind a condition on the lane (the compiler would put the wait in front of it): one round trip to LDS per
// block instead of twelve, 288 instead of 375 instructions per block in the producer wave.  (new[623] needs new[0]:
// every lane makes that one too.)  All reads precede all writes: in place.
template <class W>
MCQ_HD void mcq_mt_regenerate(W &w) {
    MCQ_PL(uint32_t, nA)[4];
    MCQ_PL(uint32_t, nB)[4];
    MCQ_PL(uint32_t, nC)[4];
    MCQ_FOR_LANES(l) {
        const uint32_t z = w.mt[MCQ_MT_M] ^ mcq_mt_twist(w.mt[0], w.mt[1]); /* the new word 0 */
#pragma unroll
        for (uint32_t s = 0; s < 4u; s++) {
            const uint32_t t0 = 64u * s + l, t = t0 < 226u ? t0 : 226u, tc = t < 169u ? t : 169u;
            const uint32_t a0 = w.mt[t], a1 = w.mt[t + 1u], f = w.mt[t + MCQ_MT_M], b0 = w.mt[227u + t], b1 = w.mt[228u + t];
            const uint32_t c0 = w.mt[454u + tc], c1o = w.mt[455u + tc]; /* (tc = 169: the padding behind the state) */
            const uint32_t c1 = t == 169u ? z : c1o;
            MCQ_L(nA)[s] = f ^ mcq_mt_twist(a0, a1);
            MCQ_L(nB)[s] = MCQ_L(nA)[s] ^ mcq_mt_twist(b0, b1);
            MCQ_L(nC)[s] = MCQ_L(nB)[s] ^ mcq_mt_twist(c0, c1);
        }
    }
    MCQ_WAVE_SYNC();
    MCQ_FOR_LANES(l) {
#pragma unroll
        for (uint32_t s = 0; s < 4u; s++) {
            const uint32_t t = 64u * s + l;
            if (t < 227u) {
                w.mt[t] = MCQ_L(nA)[s];
                w.mt[227u + t] = MCQ_L(nB)[s];
            }
            if (t < 170u) w.mt[454u + t] = MCQ_L(nC)[s];
        }
    }
    MCQ_WAVE_SYNC();
}

// Where a batch takes its words from.  One wave per query (this struct; tests/hostsim, the extended walk): the wave
// regenerates its own state block and tempers the words it parses.  Two waves per query (McqMtPairWave in
// mcq_kernels.hip): a PRODUCER wave regenerates and tempers a block ahead into a double buffer of (y & 63) | 0x80
// bytes, the parsing wave only reads bytes.
struct McqMtState;
MCQ_HD uint32_t mcq_mt_word_yb(const McqMtWave &w, const McqMtState &, uint32_t i) { return (mcq_mt_temper(w.mt[i]) & 63u) | 0x80u; }
MCQ_HD void mcq_mt_next_block(McqMtWave &w, McqMtState &) { mcq_mt_regenerate(w); }
MCQ_HD void mcq_mt_emit_lane(McqMtWave &, bool, uint32_t, uint32_t, uint32_t, uint32_t) {}
/* a wave-uniform count the batch wants computed where it stands (mcq_opaque_uniform: pinned in a scalar register);
 * waves whose state comes out of memory overload this with the identity -- the backend cannot pin those */
template <class W>
MCQ_HD uint32_t mcq_mt_pin(const W &, uint32_t x) { return mcq_opaque_uniform(x); }
/* does the wave's word source answer 0xFF behind the block's last word?  (E = k_e - 0xFF is negative: such a word is
 * accepted nowhere in zone 63, and the batch need not mask the lanes behind the block) */
template <class W>
MCQ_HD constexpr bool mcq_mt_padded(const W &) { return false; }

struct McqMtState { /* wave-uniform */
    uint32_t pos;     /* next unread state word, 624 = regenerate first */
    uint32_t it_done; /* complete iterations parsed */
    uint32_t d0;      /* draws of the current iteration already accepted */
    uint32_t flushed; /* iterations already written to the global buffer */
    uint64_t passes;
    uint32_t blocks;  /* state blocks taken over from a producer wave (two waves per query): which buffer is being parsed
                       * is a matter of registers, not of a word in LDS that every batch would have to wait for */
    uint32_t src;     /* ... and that buffer's byte offset */
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
                const uint32_t v = *reinterpret_cast<const uint32_t *>(&w.ring[(d + 1u) * MCQ_MT_ROW + ((first + c4) & (MCQ_MT_RING - 1u))]);
                *reinterpret_cast<uint32_t *>(draws + (uint64_t)d * stride + first + c4) = v;
            }
        }
    }
    MCQ_WAVE_SYNC();
    st.flushed += count;
}

// One batch of up to 64 words.  How a batch is parsed:
//   1. the lanes' words are tempered; E = L0 - 1 - (y & mask) is the deepest position the word is accepted at;
//   2. positions: first guess from the words accepted at the query's MIDDLE depth, then rounds of
//      (depth of my position, from ptab) -> (accept bits) -> (position = d0 + accepted words before me) until nothing
//      moves -- lane 0 is always final and every round makes at least one more lane final, so the fixed point is the
//      sequential parse as if no pair were ever drawn again;
//   3. every accepted word goes to the ring; an accepted r2 then looks one row down for its r1 (written by this batch or
//      an earlier one): r1 == r2 means the pair is drawn again (l.171-176, one pair in fifty).  The parse is right up to
//      and including the FIRST such word; the batch ends there (the words behind it go back into the stream; what they
//      wrote to the ring lies beyond every final position and is overwritten when those positions are parsed again),
//      the position rewinds by the pair, whose two slots the pair drawn again will fill.
// TWO_ZONE: the query reaches zone 31 (wave-uniform); TAIL: the batch may reach the query's last iteration -- words past
// it stay unread.
// Measured on gfx950 (tools/mt_bench): with one wave per query and 4 096 queries a CU holds 16 waves, and this loop is
// bound by INSTRUCTION ISSUE -- of the scalar unit (one instruction per cycle per CU, shared by the 16 waves) as much
// as of the vector units -- so it is written for instruction count: positions carry their LDS address (no address
// arithmetic in a round), masks stay in scalar registers from the compare that made them, the bookkeeping between two
// batches is a dozen scalar instructions.  (Two words per lane, or taking a re-drawn pair out of the batch in place
// instead of ending the batch, both cost more instructions than they save: 12.5 / 12.8 ms against 9.5 ms on BASELINE
// configs[2].)
struct McqMtPlan { /* wave-uniform constants of a query */
    uint32_t D, magic, runs;
    uint32_t k_e;   /* L0 - 1 + 0x80: E = k_e - ((y & mask) | 0x80) */
    uint32_t e_mid; /* depth of the middle draw: the first guess */
};

template <bool TWO_ZONE, bool TAIL, class W>
MCQ_HD void mcq_mt_batch(W &w, McqMtState &st, const McqMtPlan &pl) {
    const uint32_t rem = MCQ_MT_N - st.pos; /* state words left: lanes from there on hold no word */
    const uint32_t pos0 = MCQ_POS_BASE(w) + 4u * st.d0;
    const uint32_t p_lim = pos0 + 4u * ((pl.runs - st.it_done) * pl.D - st.d0); /* TAIL: positions from here on lie past the last iteration */
#ifdef MCQ_MT_STATS
    g_batches++;
#endif
    MCQ_MT_STAMP(0);
    MCQ_PL(uint32_t, yb);  /* (y & 63) | 0x80: the byte an accepted word of zone 63 leaves */
    MCQ_PL(int32_t, E63);
    MCQ_PL(int32_t, E31);
    MCQ_PL(uint32_t, pa);  /* the position as the byte address of its ptab word: pos0 + 4 * (accepted words before the lane) */
    MCQ_PL(uint32_t, t);   /* that word */
    MCQ_FOR_LANES(l) {
        MCQ_L(yb) = mcq_mt_word_yb(w, st, st.pos + l); /* (padded: lanes behind the block read words nobody uses) */
        if (!TWO_ZONE && mcq_mt_padded(w)) MCQ_L(E63) = (int32_t)(pl.k_e - MCQ_L(yb)); /* (0xFF behind the block: negative) */
        else MCQ_L(E63) = l < rem ? (int32_t)(pl.k_e - MCQ_L(yb)) : -1; /* never accepted */
        if (TWO_ZONE) MCQ_L(E31) = l < rem ? (int32_t)(pl.k_e - (MCQ_L(yb) & 0x9Fu)) : -1; /* (y & 31) | 0x80 */
    }
    uint64_t M = MCQ_BALLOT_OF(l, MCQ_L(E63) >= (int32_t)pl.e_mid);
    MCQ_MT_STAMP(1);
    MCQ_FOR_LANES(l) { MCQ_L(pa) = MCQ_POS4_FROM(M, l, pos0); }
    /* accept bits of the lanes at positions P; TWO_ZONE: bit 7 of the depth byte picks the zone */
#define MCQ_MT_ACCEPT_T(P)                                                                                        \
    ((TWO_ZONE ? (int32_t)(MCQ_L(t) & 63u) <= ((MCQ_L(t) & MCQ_MT_ZONE31) ? MCQ_L(E31) : MCQ_L(E63))              \
               : (int32_t)(MCQ_L(t) & 0xFFu) <= MCQ_L(E63)) &&                                                     \
     (!TAIL || MCQ_L(P) < p_lim))
#define MCQ_MT_ACCEPT(P) (MCQ_L(t) = MCQ_PTAB_AT(w, MCQ_L(P)), MCQ_MT_ACCEPT_T(P))
    /* settled when a round answers with the accept bits it was asked with (a scalar compare at the loop's foot) */
    uint64_t M1 = MCQ_BALLOT_OF(l, MCQ_MT_ACCEPT(pa));
#ifdef MCQ_MT_STATS
    g_rounds++;
#endif
    while (M1 != M) {
        M = M1;
        MCQ_FOR_LANES(l) { MCQ_L(pa) = MCQ_POS4_FROM(M, l, pos0); }
        M1 = MCQ_BALLOT_OF(l, MCQ_MT_ACCEPT(pa));
#ifdef MCQ_MT_STATS
        g_rounds++;
#endif
    }
    MCQ_MT_STAMP(2);
    /* write-out: every lane computes its slot (t is the word of the final position), the accepted ones store; then
     * every lane asks for its partner row, and while that answer travels the wave does the bookkeeping of the usual
     * case -- no pair drawn again */
    MCQ_PL(uint32_t, at); /* ring address of the word's partner row = its own row - MCQ_MT_ROW */
    MCQ_PL(uint32_t, v);
    MCQ_PL(uint32_t, pv);
    MCQ_FOR_LANES(l) {
        MCQ_L(at) = ((MCQ_L(t) >> 16) & 0xFFFu) + ((st.it_done + ((MCQ_L(t) >> 8) & 0xFFu)) & (MCQ_MT_RING - 1u));
        MCQ_L(v) = TWO_ZONE && (MCQ_L(t) & MCQ_MT_ZONE31) ? (MCQ_L(yb) & 0x9Fu) : MCQ_L(yb);
        /* the accepted lanes (M is the final round's answer; t and pa are that round's) -- lanes behind a re-drawn pair
         * too: see above */
        if (MCQ_LANE_OF(M, l)) w.ring[MCQ_L(at) + MCQ_MT_ROW] = (uint8_t)MCQ_L(v);
    }
    MCQ_WAVE_SYNC();
    MCQ_MT_STAMP(3);
    MCQ_FOR_LANES(l) { MCQ_L(pv) = w.ring[MCQ_L(at)]; }
    const uint64_t R2 = MCQ_BALLOT_OF(l, (int32_t)MCQ_L(t) < 0) & M;
    uint32_t used = rem < 64u ? rem : 64u;
    uint32_t n_r2 = mcq_mt_pin(w, mcq_mt_popc64(R2)), p_end = mcq_mt_pin(w, st.d0 + mcq_mt_popc64(M)); /* (here, not behind the wait) */
    const uint64_t R = MCQ_BALLOT_OF(l, MCQ_L(pv) == MCQ_L(v)) & R2;
    if (R) { /* one batch in five */
        const uint32_t j = mcq_mt_low64(R);
        used = j + 1u;
        M &= ((uint64_t)2 << j) - 1u;
        n_r2 = mcq_mt_popc64(R2 & M);
        p_end = st.d0 + mcq_mt_popc64(M) - 2u;
    }
    MCQ_MT_STAMP(4);
    /* walks that parse a query's blocks side by side (mcq_mt_blocks.hpp) send the FINAL draws of the batch straight to
     * the draw buffer: a pair by its r2 (with the r1 it has just read), a table card by itself; a pair that is drawn
     * again sends nothing.  (One wave per query: nothing to do, the ring is flushed in rows.) */
    {
        const uint64_t Mf = R ? M & ~((uint64_t)1 << mcq_mt_low64(R)) : M;
        MCQ_FOR_LANES(l) { mcq_mt_emit_lane(w, MCQ_LANE_OF(Mf, l), MCQ_L(t), MCQ_L(v), MCQ_L(pv), st.it_done); }
    }
    st.passes += n_r2; /* one accepted r2 per attempt (l.168) */
    const uint32_t it_add = (p_end * pl.magic) >> 16;
    st.d0 = p_end - it_add * pl.D;
    st.it_done += it_add;
    /* words consumed: up to the re-drawn pair, else all of them -- unless the stream of this query ends here (the last
     * iteration is complete): then up to the last accepted word */
    if (TAIL && !R && st.it_done >= pl.runs) used = M ? mcq_mt_top64(M) + 1u : 0u;
    st.pos += used;
    MCQ_WAVE_SYNC();
    MCQ_MT_STAMP(5);
#undef MCQ_MT_ACCEPT
#undef MCQ_MT_ACCEPT_T
}

// Parse the whole stream of one query: `runs` iterations of D = 2 * n_opp + n_deal draws (D >= 1).  The wave's MT
// state must be seeded (st.pos = 624).  draws: this query's block of the global buffer, stride = runs rounded up to
// 64 (rows may be written up to the stride).  Returns passes in st.passes.
template <bool TWO_ZONE, class W>
MCQ_HD void mcq_mt_parse_loop(W &w, McqMtState &st, const McqMtPlan &pl, uint8_t *draws, uint64_t stride) {
    /* a batch advances by at most (22 + 64) / D iterations: before it_tail no batch can reach the last one */
    const uint32_t reach = ((22u + 64u) * pl.magic >> 16) + 1u, it_tail = pl.runs > reach ? pl.runs - reach : 0u;
#define MCQ_MT_STEP(TAIL_)                                                                       \
    do {                                                                                         \
        if (st.pos >= MCQ_MT_N) {                                                                \
            mcq_mt_next_block(w, st);                                                            \
            st.pos = 0;                                                                          \
        }                                                                                        \
        mcq_mt_batch<TWO_ZONE, TAIL_>(w, st, pl);                                                \
        while (st.it_done - st.flushed >= 64u) MCQ_MT_FLUSH(w, st, pl.D, 64u, draws, stride);    \
    } while (0)
    while (st.it_done < it_tail) MCQ_MT_STEP(false);
    while (st.it_done < pl.runs) MCQ_MT_STEP(true);
#undef MCQ_MT_STEP
    if (st.it_done > st.flushed) mcq_mt_flush(w, st, pl.D, st.it_done - st.flushed, draws, stride);
}

template <class W>
MCQ_HD void mcq_mt_parse_query(W &w, McqMtState &st, uint32_t L0, uint32_t n_opp, uint32_t n_deal, uint32_t runs,
                               uint8_t *draws, uint64_t stride) {
    const uint32_t D = 2u * n_opp + n_deal, magic = mcq_mt_magic(D);
    const uint32_t z_max = L0 - 33u; /* deepest position of zone 63 */
    const bool two_zone = mcq_mt_depth(n_opp, D - 1u) > z_max;
    const McqMtPlan pl = {D, magic, runs, L0 - 1u + 0x80u, mcq_mt_depth(n_opp, D >> 1)};
    MCQ_FOR_LANES(l) {
        for (uint32_t pp = l; pp < MCQ_MT_POSITIONS; pp += 64u) {
            const uint32_t q = (pp * magic) >> 16, d = pp - q * D, e = mcq_mt_depth(n_opp, d);
            w.ptab[pp] = (e | (e > z_max ? MCQ_MT_ZONE31 : 0u)) | (q << 8) | ((d * MCQ_MT_ROW) << 16) |
                         ((d < 2u * n_opp && (d & 1u)) ? 0x80000000u : 0u);
        }
    }
    MCQ_WAVE_SYNC();
    if (two_zone) mcq_mt_parse_loop<true>(w, st, pl, draws, stride);
    else mcq_mt_parse_loop<false>(w, st, pl, draws, stride);
}
