// mcq_mt_ext.hpp -- MCQ_MODE_REPLAY_MT19937 for EXTENDED queries on the device: one wave walks numpy's stream of one
// query through the reference's loops over ranges, ghost cards and known hands.
//
// What is reproduced (reference paths relative to /root/reference):
//   montecarlo_python.py:206-208  ghost cards leave the deck                               -> deck0
//   montecarlo_python.py:126-128  the known table cards leave the deck                     -> deck0
//   montecarlo_python.py:133-163  known hands in the order of original_player_card_list: two cards leave by value
//                                 "if still there"; a hand given as a SET of classes is drawn by
//                                 repeat { passes += 1; r1 = randint(0, L); r2 = randint(0, L - 1) }
//                                 until r1 != r2 and class(deck[r1], deck[r2]) in the set -- both indices on the
//                                 UNPOPPED list -- and the two cards leave by value          -> stage_pair(known)
//   montecarlo_python.py:165-181  every further opponent: the same loop with the opponents' set, then deck.pop(r1);
//                                 deck.pop(r2) -- the second pop on the SHRUNK list          -> stage_pair(opponent)
//   montecarlo_python.py:185-189  every missing table card: deck.pop(randint(0, len(deck) - 1)) -> stage_single
//   numpy legacy randint(0, n)    rng = n - 1, mask = 2^k - 1 >= rng, one tempered word per trial, v = word & mask
//                                 accepted iff v <= rng (rng >= 20 here: never the rng == 0 case)
//
// Unlike the plain path (mcq_mt.hpp) the acceptance of a PAIR depends on the cards, so the walk is a sequence of
// stages -- one per drawn hand, one per table card -- and a stage scans the words of the current batch of 64 for its
// first success:
//   * the lanes' words alternate r1, r2, r1, ... with the accepted words (bounds L and L - 1 stay the same while the
//     stage fails): the phase of a lane = (phase at the batch start + accepted words before it) & 1, settled by the
//     same fixed-point rounds as the plain path (a word is accepted under one bound and rejected under the other one
//     time in sixty-four: one or two rounds);
//   * every r2 lane fetches its r1 (the accepted word before it: values compacted by rank through LDS), looks both
//     cards up in the wave's deck (64 bytes of LDS) and tests the class bit: ALL attempts of the batch are tested at
//     once, the first success ends the stage, the words behind it belong to the next stage;
//   * the deck loses the two cards by a wave-wide shift, the accepted indices go -- in list.pop order, r | 0x80 --
//     into the lanes' row registers and at the end of the iteration into the ring that mcq_mt_flush drains.
// With the top quarter of the classes a stage needs ~25 attempts = ~65 words: about one batch step per hand.  The
// stream is serial per query, so a batch of Q queries keeps Q waves busy, no more (256 queries: one wave per CU).
//
// One source for two builds, as mcq_mt.hpp: tests/hostsim compiles it with the lanes as arrays and checks it byte for
// byte against the literal sequential walk (mcq_replay.hpp, mcq_replay_parse_ext).
#pragma once
#include <stdint.h>

#include "mcq_device.hpp"
#include "mcq_mt.hpp"

#define MCQ_MTX_MAX_DRAWS 25u /* two per drawn hand (ten at most), five table cards */
#define MCQ_MTX_MAX_TRIALS 1000000u /* attempts of one stage before the query counts as "cannot be dealt" */

struct McqMtExtWave {
    uint32_t mt[MCQ_MT_N + 64u];
    uint32_t ext[MCQ_EXT_WORDS]; /* the query's mcq_query_ext: the class sets are read from here */
    uint32_t hand[12];           /* hand h: mcq_ext_hand(h) | word offset of its class set << 24 */
    uint8_t deck[80], deck0[80]; /* the ordered remaining deck (card ids) now / at the start of an iteration */
    uint8_t vals[80];            /* vals[1 + rank] = value of the accepted word of that rank; vals[0] = the pending r1 */
    uint8_t ring[(MCQ_MTX_MAX_DRAWS + 1u) * MCQ_MT_ROW]; /* rows as in McqMtWave: draw d in row d + 1 */
    uint64_t pm[52];             /* round 4: pm[a] bit b = class(a, b) is in the set at word offset pm_set (the opponents' set, or ... */
    uint32_t pm_set;             /* ... the first drawn hand's): the class test of a stage is one lookup and a shift */
};

struct McqMtExtState { /* wave-uniform */
    uint32_t pos, start, cnt; /* the batch: state words [pos, pos + cnt), lanes from `start` on are unread */
    uint32_t it_done, flushed;
    uint32_t n;               /* cards in the deck */
    uint64_t passes;
    bool failed;
};

// Parse the whole stream of one extended query.  The wave's MT state must be seeded.  Returns false when a range could
// not be dealt within MCQ_MTX_MAX_TRIALS attempts (st.failed).  draws / stride as mcq_mt_parse_query; rows in the order
// of mcq_ext_draws_per_iteration (two per hand that is not given as cards, then the table).
template <class W>
MCQ_HD bool mcq_mt_parse_query_ext(W &w, McqMtExtState &st, const McqQueryWords &q, const uint32_t *ext_words /* global or host */,
                                   uint8_t *draws, uint64_t stride) {
    const uint32_t n_players = q.n_players(), n_deal = 5u - q.n_board(), runs = q.runs();
    /* stage the record, the hands and the deck of an iteration's start */
    MCQ_FOR_LANES(l) {
        for (uint32_t k = l; k < MCQ_EXT_WORDS; k += 64u) w.ext[k] = ext_words[k];
    }
    MCQ_WAVE_SYNC();
    const McqExtRec er = {w.ext};
    const uint32_t n_hands = 1u + er.n_known();
    const bool opp_all = mcq_ext_opp_all(er);
    uint32_t n0 = 0, rows = 0;
    uint32_t h_first = 0;  /* the hands before the first DRAWN hand are cards that always leave the fresh deck: not walked */
    uint32_t h_need = 0;   /* the deck's CONTENT is looked at up to this hand (a set to test, cards to remove by value);
                            * behind it only its length counts: no more shifting */
    {
        uint64_t base = mcq_ext_base_deck(q, er);
        while (h_first < n_hands && !(mcq_ext_hand(q, er, h_first) >> 16)) {
            const uint32_t hd = mcq_ext_hand(q, er, h_first);
            base &= ~(((uint64_t)1 << (hd & 0xFFu)) | ((uint64_t)1 << ((hd >> 8) & 0xFFu)));
            h_first++;
        }
        for (uint32_t h = h_first; h < n_players; h++)
            if (h < n_hands || !opp_all) h_need = h;
        n0 = mcq_mt_popc64(base);
        MCQ_FOR_LANES(l) {
            if (l < 12u) {
                uint32_t hd = 0x10000u | (er.opp_set() << 24); /* an opponent: drawn under the opponents' set */
                if (l < n_hands) hd = mcq_ext_hand(q, er, l) | (mcq_ext_hand_set(er, l) << 24);
                w.hand[l] = hd;
            }
            /* card number l of the deck = the l-th set bit of the mask (once per query) */
            uint32_t c = 0, k = l;
            for (uint32_t b = 0; b < 52u; b++) {
                if ((base >> b) & 1u) {
                    if (k == 0u) { c = b; break; }
                    k--;
                }
            }
            w.deck0[l] = (uint8_t)c;
        }
        for (uint32_t h = 0; h < n_players; h++) {
            const bool is_list = h < n_hands && !(mcq_ext_hand(q, er, h) >> 16);
            rows += is_list ? 0u : 2u;
        }
        rows += n_deal;
    }
    MCQ_WAVE_SYNC();
    /* the partner masks of the set most stages test: the opponents' if they are restricted, else the first drawn hand's */
    {
        uint32_t set_off = 0xFFFFFFFFu;
        if (n_players > n_hands && !opp_all) set_off = er.opp_set();
        else
            for (uint32_t h = n_hands; h-- > h_first;)
                if (mcq_ext_hand(q, er, h) >> 16) set_off = mcq_ext_hand_set(er, h);
        MCQ_FOR_LANES(l) {
            if (l == 0u) w.pm_set = set_off;
            if (l < 52u && set_off != 0xFFFFFFFFu) {
                uint64_t m = 0;
                for (uint32_t b = 0; b < 52u; b++)
                    if (b != l && mcq_in_range(w.ext + set_off, l, b)) m |= (uint64_t)1 << b;
                w.pm[l] = m;
            }
        }
    }
    MCQ_WAVE_SYNC();
    const uint32_t pm_set = w.pm_set;

    MCQ_PL(uint32_t, y63);
    MCQ_PL(uint32_t, y31);
    MCQ_PL(uint32_t, out); /* lane d: draw d of the current iteration */
    MCQ_FOR_LANES(l) { MCQ_L(out) = 0x80u; MCQ_L(y63) = 0; MCQ_L(y31) = 0; }

    /* the next batch of words: everything left of the current one has been read */
    auto load = [&]() {
        st.pos += st.cnt;
        if (st.pos >= MCQ_MT_N) {
            mcq_mt_regenerate(w);
            st.pos = 0;
        }
        st.cnt = MCQ_MT_N - st.pos < 64u ? MCQ_MT_N - st.pos : 64u;
        st.start = 0;
        MCQ_FOR_LANES(l) {
            const uint32_t y = mcq_mt_temper(w.mt[st.pos + l]);
            MCQ_L(y63) = y & 63u;
            MCQ_L(y31) = y & 31u;
        }
    };
    /* deck.pop(i) */
    auto pop = [&](uint32_t i) {
        MCQ_PL(uint32_t, tmp);
        MCQ_FOR_LANES(l) { MCQ_L(tmp) = w.deck[l + (l >= i ? 1u : 0u)]; }
        MCQ_WAVE_SYNC();
        MCQ_FOR_LANES(l) { w.deck[l] = (uint8_t)MCQ_L(tmp); }
        MCQ_WAVE_SYNC();
        st.n--;
    };
    /* deck.pop(i1); deck.pop(i2) in one shift (i2 counts on the list pop(i1) left) */
    auto pop2 = [&](uint32_t i1, uint32_t i2) {
        MCQ_PL(uint32_t, tmp);
        MCQ_FOR_LANES(l) {
            const uint32_t m = l + (l >= i2 ? 1u : 0u);
            MCQ_L(tmp) = w.deck[m + (m >= i1 ? 1u : 0u)];
        }
        MCQ_WAVE_SYNC();
        MCQ_FOR_LANES(l) { w.deck[l] = (uint8_t)MCQ_L(tmp); }
        MCQ_WAVE_SYNC();
    };
    auto remove = [&](uint32_t card) { /* by value, if it is still there (l.150-161) */
        const uint64_t B = MCQ_BALLOT_OF(l, l < st.n && w.deck[l] == card);
        if (B) pop(mcq_mt_low64(B));
    };
    auto emit = [&](uint32_t row, uint32_t r) {
        MCQ_FOR_LANES(l) { MCQ_L(out) = l == row ? (r | 0x80u) : MCQ_L(out); }
    };
    /* one table card: the first accepted word under randint(0, bound) */
    auto stage_single = [&](uint32_t bound) -> uint32_t {
        const uint32_t rng = bound - 1u;
        for (;;) {
            if (st.start >= st.cnt) load();
            MCQ_PL(uint32_t, v);
            MCQ_FOR_LANES(l) { MCQ_L(v) = rng >= 32u ? MCQ_L(y63) : MCQ_L(y31); }
            const uint64_t A = MCQ_BALLOT_OF(l, l >= st.start && l < st.cnt && MCQ_L(v) <= rng);
            if (A) {
                const uint32_t j = mcq_mt_low64(A);
                st.start = j + 1u;
                return MCQ_AT_UNIFORM(v, j);
            }
            st.start = st.cnt;
        }
    };
    /* one drawn hand on a deck of n cards: the first pair r1 in [0, n), r2 in [0, n - 1), r1 != r2, whose cards' class is
     * in the set at word offset set_off (restricted = false: every class, no lookup).  false: not within the trial bound */
    auto stage_pair = [&](uint32_t set_off, bool restricted, uint32_t &r1, uint32_t &r2) -> bool {
        const uint32_t rng1 = st.n - 1u, rng2 = st.n - 2u;
        uint32_t phase0 = 0, pending = 0, trials = 0;
        for (;;) {
            if (st.start >= st.cnt) load();
            MCQ_PL(uint32_t, v1);
            MCQ_PL(uint32_t, v2);
            MCQ_PL(uint32_t, v);
            MCQ_PL(uint32_t, g);
            MCQ_FOR_LANES(l) {
                MCQ_L(v1) = rng1 >= 32u ? MCQ_L(y63) : MCQ_L(y31);
                MCQ_L(v2) = rng2 >= 32u ? MCQ_L(y63) : MCQ_L(y31);
            }
            const uint64_t In = MCQ_BALLOT_OF(l, l >= st.start && l < st.cnt);
            const uint64_t A1 = MCQ_BALLOT_OF(l, MCQ_L(v1) <= rng1) & In, A2 = MCQ_BALLOT_OF(l, MCQ_L(v2) <= rng2) & In;
            /* phases: settled when a round answers with the accept bits it was asked with (tested at the loop's foot: a
             * scalar compare and branch per round) */
            uint64_t M = A1, Odd, M1;
            MCQ_FOR_LANES(l) { MCQ_L(g) = phase0 + MCQ_COUNT_BELOW(M, l); }
            Odd = MCQ_BALLOT_OF(l, (MCQ_L(g) & 1u) != 0u);
            M1 = (A1 & ~Odd) | (A2 & Odd);
            while (M1 != M) {
                M = M1;
                MCQ_FOR_LANES(l) { MCQ_L(g) = phase0 + MCQ_COUNT_BELOW(M, l); }
                Odd = MCQ_BALLOT_OF(l, (MCQ_L(g) & 1u) != 0u);
                M1 = (A1 & ~Odd) | (A2 & Odd);
            }
            const uint64_t R2 = M & Odd; /* the accepted second indices: one attempt each (l.168) */
            MCQ_FOR_LANES(l) {
                MCQ_L(v) = (MCQ_L(g) & 1u) ? MCQ_L(v2) : MCQ_L(v1);
                if (l == 0u && phase0) w.vals[0] = (uint8_t)pending; /* rank -1: the r1 an earlier batch left behind */
                if (MCQ_LANE_OF(M, l)) w.vals[1u + MCQ_L(g) - phase0] = (uint8_t)MCQ_L(v);
            }
            MCQ_WAVE_SYNC();
            MCQ_PL(uint32_t, p1);
            MCQ_PL(bool, ok);
            MCQ_FOR_LANES(l) {
                /* my r1: the accepted word of the rank before mine (slot rank = 1 + (g - phase0) - 1) */
                MCQ_L(p1) = w.vals[(MCQ_L(g) - phase0) & 63u];
                MCQ_L(ok) = MCQ_L(p1) != MCQ_L(v);
                if (restricted) {
                    const uint32_t a = w.deck[MCQ_L(p1) & 63u], b = w.deck[MCQ_L(v) & 63u]; /* both on the unpopped list */
                    if (set_off == pm_set) { /* (wave-uniform) a < 52: deck entries behind the deck's end are 0 */
                        MCQ_L(ok) = MCQ_L(ok) && ((w.pm[a < 52u ? a : 0u] >> (b & 63u)) & 1u) != 0u;
                    } else {
                        const uint32_t i = mcq_class_index(a, b);
                        MCQ_L(ok) = MCQ_L(ok) && ((w.ext[set_off + (i >> 5)] >> (i & 31u)) & 1u) != 0u;
                    }
                }
            }
            MCQ_WAVE_SYNC(); /* vals are read: the next step may write them */
            const uint64_t S = MCQ_BALLOT_OF(l, MCQ_L(ok)) & R2;
            if (S) {
                const uint32_t j = mcq_mt_low64(S);
                st.passes += mcq_mt_popc64(R2 & (((uint64_t)2 << j) - 1u));
                r1 = MCQ_AT_UNIFORM(p1, j);
                r2 = MCQ_AT_UNIFORM(v, j);
                st.start = j + 1u;
                return true;
            }
            const uint32_t tried = mcq_mt_popc64(R2);
            st.passes += tried;
            trials += tried;
            if (trials >= MCQ_MTX_MAX_TRIALS) return false;
            if ((phase0 + mcq_mt_popc64(M)) & 1u) { /* the batch ends on an r1: it waits for its r2 */
                pending = MCQ_AT_UNIFORM(v, mcq_mt_top64(M));
                phase0 = 1;
            } else {
                phase0 = 0;
            }
            st.start = st.cnt;
        }
    };

    McqMtState fl = {0u, 0u, 0u, 0u, 0ull}; /* for mcq_mt_flush: it_done / flushed */
    while (st.it_done < runs) {
        MCQ_FOR_LANES(l) { w.deck[l] = w.deck0[l]; }
        MCQ_WAVE_SYNC();
        st.n = n0;
        uint32_t row = 0;
        for (uint32_t h = h_first; h < n_players; h++) {
            const uint32_t hd = w.hand[h < n_hands ? h : 11u];
            if (!((hd >> 16) & 0xFFu)) { /* two cards: they leave by value */
                remove(hd & 0xFFu);
                remove((hd >> 8) & 0xFFu);
                continue;
            }
            const bool known = h < n_hands;
            uint32_t r1 = 0, r2 = 0;
            if (!stage_pair(hd >> 24, known || !opp_all, r1, r2)) {
                st.failed = true;
                return false;
            }
            /* as pops: a known hand's two cards are looked up on the unpopped list and leave by value = pop(r1), then
             * pop(r2 - (r2 > r1)); an opponent is dealt deck.pop(r1); deck.pop(r2) (l.178-179) */
            const uint32_t second = known ? r2 - (r2 > r1 ? 1u : 0u) : r2;
            emit(row, r1);
            emit(row + 1u, second);
            row += 2u;
            if (h < h_need) pop2(r1, second);
            st.n -= 2u;
        }
        for (uint32_t k = 0; k < n_deal; k++) { /* (only the deck's length counts here) */
            const uint32_t idx = stage_single(st.n - 1u); /* randint(0, len(deck) - 1): never the last card (l.188) */
            emit(row++, idx);
            st.n--;
        }
        MCQ_FOR_LANES(l) {
            if (l < rows) w.ring[(l + 1u) * MCQ_MT_ROW + (st.it_done & (MCQ_MT_RING - 1u))] = (uint8_t)MCQ_L(out);
        }
        MCQ_WAVE_SYNC();
        st.it_done++;
        fl.it_done = st.it_done;
        if (st.it_done - fl.flushed >= 64u) mcq_mt_flush(w, fl, rows, 64u, draws, stride);
    }
    if (st.it_done > fl.flushed) mcq_mt_flush(w, fl, rows, st.it_done - fl.flushed, draws, stride);
    st.flushed = fl.flushed;
    return true;
}
