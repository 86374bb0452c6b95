// mcq_mt_blocks.hpp -- MCQ_MODE_REPLAY_MT19937 for FEW LONG queries: the stream of one query parsed block by block, the
// 624-word state blocks side by side.
//
// The walk of mcq_mt.hpp is one serial chain per query: a single 100 000-run query (BASELINE configs[1] in its bit-exact
// form -- and what a user who checks results against the reference calls: ONE query, tools/montecarlo_python.py:191) is
// 3 500 state blocks parsed one behind the other by one wave, 6.5 ms.  What makes the stream serial is small, though:
// at a block boundary the walk is in one of at most D = 2 n_opp + n_deal positions of an iteration (plus, when it waits
// for the second index of a pair, the value of the first).  So:
//   1. generate: a work-group per query writes the tempered bytes (y & 63) | 0x80 of all its state blocks to HBM -- the
//      recurrence itself is serial, three dependent sweeps of 227 words per block (mcq_mtb_generate_kernel);
//   2. scan: one wave per (query, block); lane d < D walks the block's 624 words as a plain sequential automaton from
//      ENTRY position d and reports where it leaves the block: exit position, pending first index, iterations completed.
//      An entry that waits for a second index does not know the first one: lane d assumes the pair is NOT drawn again
//      (montecarlo_python.py:171-176) by the first accepted word, lane D + d / 2 assumes it IS, and both report the value
//      of that word (mcq_mtb_automaton);
//   3. stitch: one wave per query follows the exits from block to block (mcq_mtb_stitch_step): every block's true entry
//      (position, pending first index, iterations done), and the block the stream ends in;
//   4. parse: one wave per (query, block) parses its block from its true entry with mcq_mt_batch -- the same code as the
//      serial walk -- and sends the final draws straight to the draw buffer (mcq_mt_emit_lane): a pair by its second
//      index, which has read the first from the ring (of the previous block: put there from the entry).
// Same bytes as the serial walk (tests: host build against the literal sequential walk of mcq_replay.hpp; GPU against the
// oracle and the reference's known answers).  The host estimates how many blocks a query needs (the expected number of
// words + a margin of several standard deviations); a stream that has not ended within them marks the query and the
// call falls back to the serial walk.
#pragma once
#include <stdint.h>

#include "mcq_mt.hpp"

#define MCQ_MTB_LANES 32u /* entry states of a block: D <= 23 positions + n_opp <= 9 "drawn again" variants */

struct McqMtbPlan { /* wave-uniform constants of a query */
    uint32_t D, two_opp, L0, z_max, runs;
};
MCQ_HD McqMtbPlan mcq_mtb_plan(uint32_t L0, uint32_t n_opp, uint32_t n_deal, uint32_t runs) {
    const McqMtbPlan p = {2u * n_opp + n_deal, 2u * n_opp, L0, L0 - 33u, runs};
    return p;
}

/* exit word of an entry state: where the walk stands behind the block's last word */
#define MCQ_MTB_D(x) ((x) & 31u)               /* position in the iteration */
#define MCQ_MTB_PEND(x) (((x) >> 5) & 63u)     /* value of the accepted first index the position waits on ... */
#define MCQ_MTB_PEND_OK(x) (((x) >> 11) & 1u)  /* ... if it was drawn in this block (else: the entry's) */
#define MCQ_MTB_FIRST(x) (((x) >> 12) & 63u)   /* value of the first accepted word ... */
#define MCQ_MTB_SEEN(x) (((x) >> 18) & 1u)     /* ... if any word was accepted */
#define MCQ_MTB_ITS(x) ((x) >> 19)             /* iterations completed inside the block */

MCQ_HD bool mcq_mtb_is_r2(const McqMtbPlan &pl, uint32_t d) { return d < pl.two_opp && (d & 1u) != 0u; }

// The sequential walk of one block (yb: its 624 bytes (y & 63) | 0x80) from entry state `lane`:
//   lane < D: position lane, the first accepted word does not complete a pair that is drawn again;
//   D <= lane < D + n_opp: position 2 (lane - D) + 1 (a second index), the first accepted word DOES (it equals the
//   pending first index), so the walk steps back to the first index.
// Every lane of a wave runs this on the same words (device: the bytes sit in LDS, one broadcast read per word), so it is
// written for instruction count: what a position needs comes out of ONE table word (ptab, MCQ_MTB_POS entries, filled by
// mcq_mtb_fill_ptab; device: LDS) --
//   bits 0-5 the mask of randint's loop (63 / 31), bits 8-13 the largest accepted value, bits 16-20 the position behind
//   an accepted word (0 behind the last one), bit 24 "that completes an iteration", bit 25 first index, bit 26 second index
#define MCQ_MTB_POS 24u
MCQ_HD uint32_t mcq_mtb_pos_word(const McqMtbPlan &pl, uint32_t d) {
    const uint32_t e = d + (d >= pl.two_opp ? 1u : 0u), rng = pl.L0 - 1u - e, last = d + 1u == pl.D ? 1u : 0u;
    return (e > pl.z_max ? 31u : 63u) | (rng << 8) | ((last ? 0u : d + 1u) << 16) | (last << 24) |
           ((d < pl.two_opp && !(d & 1u)) ? 1u << 25 : 0u) | (mcq_mtb_is_r2(pl, d) ? 1u << 26 : 0u);
}
MCQ_HD uint32_t mcq_mtb_automaton(const uint8_t *yb, const uint32_t *ptab, const McqMtbPlan &pl, uint32_t lane) {
    if (lane >= pl.D + (pl.two_opp >> 1)) return 0u;
    /* `same` = the pending first index, or 64 ("no value") while it is unknown: a lane that entered on a second index
     * without knowing the first takes the pair as final -- its twin lane D + d / 2 takes it as drawn again (`again`: any
     * value matches, once), the stitch picks the one that is right */
    uint32_t again = lane >= pl.D ? 1u : 0u;
    uint32_t d = again ? 2u * (lane - pl.D) + 1u : lane;
    uint32_t pend = 64u, first = 64u, its = 0;
    for (uint32_t i = 0; i < MCQ_MT_N; i++) {
        const uint32_t t = ptab[d], v = yb[i] & t & 63u;
        if (v > ((t >> 8) & 63u)) continue; /* rejected by randint's mask loop */
        first = first == 64u ? v : first;
        const bool r2 = (t >> 26) & 1u;
        const bool back = r2 && ((again | (v == pend ? 1u : 0u)) != 0u); /* equal to the first = drawn again (l.171-176) */
        again = r2 ? 0u : again;
        pend = (t >> 25) & 1u ? v : pend;
        its += back ? 0u : (t >> 24) & 1u;
        d = back ? d - 1u : (t >> 16) & 31u;
    }
    const uint32_t pend_ok = pend != 64u ? 1u : 0u, seen = first != 64u ? 1u : 0u;
    return d | ((pend & 63u) << 5) | (pend_ok << 11) | ((first & 63u) << 12) | (seen << 18) | (its << 19);
}

struct McqMtbEntry { /* the walk in front of a block's first word */
    uint32_t it0;  /* iterations completed */
    uint32_t dp;   /* position | pending first index << 8 | 0x80000000: the stream has not ended (the block is parsed) */
};

// One step of the stitch: the block's exit words (exits[0 .. MCQ_MTB_LANES)), the walk's state in front of it -> behind it.
MCQ_HD void mcq_mtb_stitch_step(const uint32_t *exits, const McqMtbPlan &pl, uint32_t &d, uint32_t &pend, uint32_t &it) {
    uint32_t x = exits[d];
    if (mcq_mtb_is_r2(pl, d) && MCQ_MTB_SEEN(x) && MCQ_MTB_FIRST(x) == pend) x = exits[pl.D + (d >> 1)];
    if (MCQ_MTB_PEND_OK(x)) pend = MCQ_MTB_PEND(x);
    d = MCQ_MTB_D(x);
    it += MCQ_MTB_ITS(x);
}

// The stitch in two levels (one step per block behind the other would be 3 900 dependent LDS reads for a 6-max
// 100 000-run query: 0.5 ms): groups of MCQ_MTB_GROUP blocks are COMPOSED first, side by side -- lane = entry state of the
// group, walking the group's blocks by their exit words --, which gives every group an exit word of the same form (and
// its iterations in a word of their own); one wave then follows the GROUPS' exits, and every group, now knowing its
// entry, follows its own blocks once more to note their entries.
#define MCQ_MTB_GROUP 32u
struct McqMtbWalk { /* a walk from an entry state across blocks, the first index of the entry's pair unknown */
    uint32_t d, pend, pend_ok, first, seen, its;
    bool again; /* entry variant "the first accepted word is a second index equal to its first": until a word is accepted */
};
MCQ_HD McqMtbWalk mcq_mtb_walk_from(const McqMtbPlan &pl, uint32_t lane) { /* lane as in mcq_mtb_automaton */
    const bool again = lane >= pl.D;
    const McqMtbWalk s = {again ? 2u * (lane - pl.D) + 1u : lane, 0u, 0u, 0u, 0u, 0u, again};
    return s;
}
MCQ_HD void mcq_mtb_compose_step(const uint32_t *exits, const McqMtbPlan &pl, McqMtbWalk &s) {
    uint32_t x;
    if (!s.seen) { /* still in the entry state: the lane's own variant */
        x = exits[s.again ? pl.D + (s.d >> 1) : s.d];
    } else { /* a second index here has seen its first accepted inside the group */
        x = exits[s.d];
        if (mcq_mtb_is_r2(pl, s.d) && MCQ_MTB_SEEN(x) && MCQ_MTB_FIRST(x) == s.pend) x = exits[pl.D + (s.d >> 1)];
    }
    if (!s.seen && MCQ_MTB_SEEN(x)) s.first = MCQ_MTB_FIRST(x);
    s.seen |= MCQ_MTB_SEEN(x);
    if (MCQ_MTB_PEND_OK(x)) {
        s.pend = MCQ_MTB_PEND(x);
        s.pend_ok = 1u;
    }
    s.d = MCQ_MTB_D(x);
    s.its += MCQ_MTB_ITS(x);
}
MCQ_HD uint32_t mcq_mtb_walk_word(const McqMtbWalk &s) { /* the exit word of a group (its iterations travel beside it) */
    return s.d | (s.pend << 5) | (s.pend_ok << 11) | (s.first << 12) | (s.seen << 18);
}
/* one step of the walk over GROUPS: as mcq_mtb_stitch_step, the iterations from their own array */
MCQ_HD void mcq_mtb_stitch_group(const uint32_t *words, const uint32_t *its, const McqMtbPlan &pl, uint32_t &d, uint32_t &pend,
                                 uint32_t &it) {
    uint32_t at = d;
    if (mcq_mtb_is_r2(pl, d) && MCQ_MTB_SEEN(words[d]) && MCQ_MTB_FIRST(words[d]) == pend) at = pl.D + (d >> 1);
    const uint32_t x = words[at];
    if (MCQ_MTB_PEND_OK(x)) pend = MCQ_MTB_PEND(x);
    d = MCQ_MTB_D(x);
    it += its[at];
}

// State words a query consumes, as the host estimates them: per draw 1 / P(accept) words, per pair 1 / (1 - 1 / L)
// attempts; + a margin (half a percent and eight blocks: the count's standard deviation is about two words per
// iteration, 632 words at 100 000 iterations).  (a host function: double arithmetic)
static inline uint32_t mcq_mtb_blocks_needed(uint32_t L0, uint32_t n_opp, uint32_t n_deal, uint32_t runs, int margin = 8) {
    double per_it = 0.0;
    const uint32_t D = 2u * n_opp + n_deal;
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t e = mcq_mt_depth(n_opp, d), rng = L0 - 1u - e;
        double w = (rng >= 32u ? 64.0 : 32.0) / (double)(rng + 1u);
        if (d < 2u * n_opp) w /= 1.0 - 1.0 / (double)(L0 - (d & ~1u)); /* the pair at deck length L = L0 - d (d even) */
        per_it += w;
    }
    const double words = per_it * (double)runs * 1.005;
    const long blocks = (long)(words / (double)MCQ_MT_N) + margin; /* (a negative margin: tests of the fall-back) */
    return blocks < 1 ? 1u : (uint32_t)blocks;
}

// The parsing wave of step 4: what mcq_mt_batch touches (positions, ring, the block's bytes) and where the draws go.
struct McqMtBlockWave {
    uint32_t ptab[MCQ_MT_POSITIONS];
    uint8_t ring[(MCQ_MT_MAX_DRAWS + 1u) * MCQ_MT_ROW];
    uint8_t yb[MCQ_MT_N + 64u];
    uint8_t *draws;  /* this query's block of the draw buffer */
    uint64_t stride;
    uint32_t two_opp;
};
MCQ_HD uint32_t mcq_mt_word_yb(const McqMtBlockWave &w, const McqMtState &, uint32_t i) { return w.yb[i]; }
MCQ_HD void mcq_mt_next_block(McqMtBlockWave &, McqMtState &) {} /* never: a wave parses one block */
MCQ_HD uint32_t mcq_mt_pin(const McqMtBlockWave &, uint32_t x) { return x; }
MCQ_HD constexpr bool mcq_mt_padded(const McqMtBlockWave &) { return true; } /* yb[624 .. 688) = 0xFF */
MCQ_HD void mcq_mt_emit_lane(McqMtBlockWave &w, bool final_draw, uint32_t t, uint32_t v, uint32_t pv, uint32_t it_done) {
    if (!final_draw) return;
    const uint32_t d = (((t >> 16) & 0xFFFu) * 497u) >> 16; /* / MCQ_MT_ROW, exact below 24 rows (checked in the tests) */
    const uint64_t it = (uint64_t)it_done + ((t >> 8) & 0xFFu);
    if ((int32_t)t < 0) { /* a second index: the pair */
        w.draws[(uint64_t)(d - 1u) * w.stride + it] = (uint8_t)pv;
        w.draws[(uint64_t)d * w.stride + it] = (uint8_t)v;
    } else if (d >= w.two_opp) {
        w.draws[(uint64_t)d * w.stride + it] = (uint8_t)v;
    }
}

// Step 4 for one block: parse its words from the entry, draws straight to the buffer.  Returns the accepted second
// indices (= attempts, `passes`).  The wave's ptab must hold the query's position table (mcq_mt_fill_ptab).
template <class W>
MCQ_HD void mcq_mt_fill_ptab(W &w, uint32_t L0, uint32_t n_opp, uint32_t D) {
    const uint32_t magic = mcq_mt_magic(D), z_max = L0 - 33u;
    MCQ_FOR_LANES(l) {
        for (uint32_t pp = l; pp < MCQ_MT_POSITIONS; pp += 64u) {
            const uint32_t q = (pp * magic) >> 16, d = pp - q * D, e = mcq_mt_depth(n_opp, d);
            w.ptab[pp] = (e | (e > z_max ? MCQ_MT_ZONE31 : 0u)) | (q << 8) | ((d * MCQ_MT_ROW) << 16) |
                         ((d < 2u * n_opp && (d & 1u)) ? 0x80000000u : 0u);
        }
    }
    MCQ_WAVE_SYNC();
}

template <class W>
MCQ_HD uint64_t mcq_mtb_parse_block(W &w, uint32_t L0, uint32_t n_opp, uint32_t n_deal, uint32_t runs, const McqMtbEntry &en) {
    const uint32_t D = 2u * n_opp + n_deal;
    const McqMtPlan pl = {D, mcq_mt_magic(D), runs, L0 - 1u + 0x80u, mcq_mt_depth(n_opp, D >> 1)};
    const bool two_zone = mcq_mt_depth(n_opp, D - 1u) > L0 - 33u;
    McqMtState st = {0u, en.it0, en.dp & 0xFFu, 0u, 0ull};
    if (mcq_mtb_is_r2(mcq_mtb_plan(L0, n_opp, n_deal, runs), st.d0)) { /* the first index this entry waits on: where its second index looks for it */
        MCQ_FOR_LANES(l) {
            if (l == 0u) w.ring[st.d0 * MCQ_MT_ROW + (st.it_done & (MCQ_MT_RING - 1u))] = (uint8_t)(((en.dp >> 8) & 0x7Fu) | 0x80u);
        }
        MCQ_WAVE_SYNC();
    }
    while (st.pos < MCQ_MT_N && st.it_done < runs) {
        if (two_zone) mcq_mt_batch<true, true>(w, st, pl);
        else mcq_mt_batch<false, true>(w, st, pl);
    }
    return st.passes;
}
