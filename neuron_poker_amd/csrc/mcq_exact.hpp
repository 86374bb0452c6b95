// mcq_exact.hpp -- exact equity by exhaustive enumeration, lane code shared by the kernel
// (mcq_exact.hip) and the host build of the tests (tests/hostsim).
//
// What is enumerated.  After the hero and the known table cards have left the ordered deck, R (L = 50 - b cards,
// ascending card id = the reference's list order) remains; the opponents are dealt from it one after the
// other, then the 5 - b missing table cards (tools/montecarlo_python.py:121-189).  Both dealing laws make
//   P(outcome) = weight / total weight   with small integer weights, so the result is exact integers:
//   * MCQ_LAW_UNIFORM: every ordered tuple of disjoint hands and every table completion is equally likely
//     (weight 1).
//   * MCQ_LAW_REFERENCE (the Python reference's index arithmetic, l.167-179 and l.188):
//       - an opponent's accepted index pairs (r1 in [0,L), r2 in [0,L-1), r1 != r2; card B taken from the list
//         that has already lost card A) reach the unordered hand {x < y} twice -- (A=x,B=y) and (A=y,B=x) -- except
//         when x and y are neighbours in the current deck: then (A=x,B=y) would need r2 == r1, which is re-drawn.
//         Weight 2, or 1 for neighbours; "current deck" = R for the first opponent, R minus his hand for the second.
//       - table cards are drawn with randint(0, len - 1): never the last = highest remaining card.  That card
//         stays the highest as long as it stays, so the completion is uniform over the remaining deck minus its
//         highest card.  A completion T is therefore possible iff some card of R above max(T) is not in an
//         opponent's hand.
// The work is organised table-completion-major: one completion per wave at a time; the 45 cards left are the
// same count for every b, so there are always C(45, 2) = 990 candidate opponent hands, 16 per lane.  Two
// opponents: the 990 keys are kept (LDS) and all ordered pairs of disjoint hands are visited.
#ifndef MCQ_EXACT_HPP
#define MCQ_EXACT_HPP

#include "mcq_device.hpp"

#define MCQ_EXACT_REM 45u    /* cards neither known nor in the completion */
#define MCQ_EXACT_PAIRS 990u /* C(45, 2) */

// pair index i <-> (x < y), i = y (y - 1) / 2 + x, x, y positions in the 45-card list
MCQ_HD void mcq_exact_pair_xy(uint32_t i, uint32_t &x, uint32_t &y) {
    uint32_t yy = 1;
    while ((yy + 1u) * yy / 2u <= i) yy++;
    y = yy;
    x = i - yy * (yy - 1u) / 2u;
}

MCQ_HD uint32_t mcq_exact_binom(uint32_t n, uint32_t k) { /* n <= 50, k <= 5: fits 32 bits at every step */
    if (k > n) return 0;
    uint32_t r = 1;
    for (uint32_t i = 1; i <= k; i++) r = r * (n - k + i) / i;
    return r;
}

// idx in [0, C(L, k)) -> k ascending positions in [0, L) (combinatorial number system)
// Unused entries (i >= k) are set to 255.  Loops have fixed bounds so that pos[] stays in registers.
MCQ_HD void mcq_exact_unrank(uint32_t idx, uint32_t L, uint32_t k, uint32_t pos[5]) {
    uint32_t c = L;
#pragma unroll
    for (uint32_t i = 5; i >= 1; i--) {
        if (i > k) {
            pos[i - 1] = 255u;
            continue;
        }
        c--; /* candidates strictly below the previous choice */
        uint32_t b = mcq_exact_binom(c, i);
        while (b > idx) { /* C(c-1, i) = C(c, i) (c - i) / c */
            b = b * (c - i) / c;
            c--;
        }
        pos[i - 1] = c;
        idx -= b;
    }
}

struct McqExactQuery { /* wave-uniform */
    uint32_t deck_lo, deck_hi; /* R as a card-id mask */
    uint32_t L, k, n_opp;
    bool ref_law;
    McqBoard known;
    McqHole hero;
};

MCQ_HD bool mcq_exact_query(const McqQueryWords &q, int law, McqExactQuery &e) {
    if (!mcq_query_valid(q) || q.n_players() > 3u) return false;
    McqQueryCtx qc;
    mcq_query_ctx(q, qc);
    e.deck_lo = qc.deck_lo;
    e.deck_hi = qc.deck_hi;
    e.L = qc.L0;
    e.k = qc.n_deal;
    e.n_opp = qc.n_opp;
    e.ref_law = law == MCQ_LAW_REFERENCE;
    e.known = qc.board;
    e.hero = qc.hero;
    return true;
}

// position p in R (0 = lowest card) -> card id
MCQ_HD uint32_t mcq_exact_card_at(const McqExactQuery &e, uint32_t p, const uint32_t *sel8) {
    uint32_t lo = e.deck_lo, hi = e.deck_hi;
    return mcq_select_pop(lo, hi, p, sel8);
}

// One table completion: pos[0..k) ascending R-positions of the new table cards.
struct McqExactBoard { /* wave-uniform */
    McqBoard b;
    McqFlushSel fs;
    uint32_t hero_key;
    uint32_t top; /* R-position of the highest new card (k > 0) */
    uint32_t u;   /* cards of R above it */
};

MCQ_HD void mcq_exact_board(const McqExactQuery &e, const uint32_t pos[5], const uint32_t *sel8, const uint32_t *tf,
                            const uint32_t *tops, const uint32_t *sd, McqExactBoard &o) {
    o.b = e.known;
    o.top = 0;
#pragma unroll
    for (uint32_t i = 0; i < 5; i++)
        if (i < e.k) {
            o.b.add(mcq_card(mcq_exact_card_at(e, pos[i], sel8)));
            o.top = pos[i]; /* ascending: the last one is the highest */
        }
    o.fs.from_board(o.b);
    o.hero_key = mcq_eval_key(o.b, o.fs, e.hero, tf, tops, sd);
    o.u = e.k ? e.L - 1u - o.top : 64u; /* nothing to draw: nothing is excluded */
}

// l-th (l < 45) card left after the completion: its R-position
MCQ_HD uint32_t mcq_exact_rem_pos(const uint32_t pos[5], uint32_t l) {
    uint32_t p = l;
#pragma unroll
    for (uint32_t i = 0; i < 5; i++) p += p >= pos[i] ? 1u : 0u; /* unused entries are 255 */
    return p;
}

// packed per-hand record of the two-opponent pass: R-positions of its cards and how many lie above `top`
MCQ_HD uint32_t mcq_exact_pack(uint32_t pa, uint32_t pb, uint32_t top, bool any_new) {
    const uint32_t above = any_new ? (pa > top ? 1u : 0u) + (pb > top ? 1u : 0u) : 0u;
    return pa | (pb << 6) | (above << 12);
}
MCQ_HD uint32_t mcq_exact_pa(uint32_t r) { return r & 63u; }
MCQ_HD uint32_t mcq_exact_pb(uint32_t r) { return (r >> 6) & 63u; }
MCQ_HD uint32_t mcq_exact_above(uint32_t r) { return r >> 12; }

// weight of the first opponent's hand (positions pa < pb in R)
MCQ_HD uint32_t mcq_exact_w1(bool ref_law, uint32_t pa, uint32_t pb) {
    if (!ref_law) return 1u;
    return pb == pa + 1u ? 1u : 2u;
}

// weight of the second hand r2 once the first hand r1 is gone: neighbours in R minus the first hand <=> every card
// between them belongs to the first hand
MCQ_HD uint32_t mcq_exact_w2(bool ref_law, uint32_t r1, uint32_t r2) {
    if (!ref_law) return 1u;
    const uint32_t pa = mcq_exact_pa(r2), pb = mcq_exact_pb(r2), qa = mcq_exact_pa(r1), qb = mcq_exact_pb(r1);
    const uint32_t between = (qa > pa && qa < pb ? 1u : 0u) + (qb > pa && qb < pb ? 1u : 0u);
    return pb - pa - 1u == between ? 1u : 2u;
}

struct McqExactAcc { /* one table completion, one lane: weights of strict wins, ties, everything */
    uint32_t win, tie, tot;
};

// Pass A, lane `lane` of 64: keys of the candidate hands lane, lane + 64, ... (pair_xy[i] = x | y << 8, positions
// in the 45-card list rem_card / rem_pos).  Zero or one opponent: the outcome is tallied here.  Two opponents
// (keys != nullptr): key and packed positions are stored for pass B.
MCQ_HD void mcq_exact_pass_a(const McqExactQuery &e, const McqExactBoard &bd, uint32_t lane, const uint16_t *pair_xy,
                             const McqCard *rem_card, const uint32_t *rem_pos, const uint32_t *tf, const uint32_t *tops,
                             const uint32_t *sd, uint32_t *keys, uint16_t *rec, McqExactAcc &acc) {
    if (e.n_opp == 0u) { /* hero alone: he wins whatever comes (run_montecarlo returns 1.0) */
        if (lane == 0u && (!e.ref_law || bd.u > 0u)) {
            acc.win += 1u;
            acc.tot += 1u;
        }
        return;
    }
    for (uint32_t i = lane; i < MCQ_EXACT_PAIRS; i += 64u) {
        const uint32_t xy = pair_xy[i], x = xy & 0xFFu, y = xy >> 8;
        McqHole h;
        h.set(rem_card[x], rem_card[y]);
        const uint32_t key = mcq_eval_key(bd.b, bd.fs, h, tf, tops, sd);
        const uint32_t r = mcq_exact_pack(rem_pos[x], rem_pos[y], bd.top, e.k != 0u);
        if (keys) {
            keys[i] = key;
            rec[i] = (uint16_t)r;
        } else {
            const bool ok = !e.ref_law || mcq_exact_above(r) < bd.u;
            const uint32_t w = ok ? mcq_exact_w1(e.ref_law, mcq_exact_pa(r), mcq_exact_pb(r)) : 0u;
            acc.win += key < bd.hero_key ? w : 0u;
            acc.tie += key == bd.hero_key ? w : 0u;
            acc.tot += w;
        }
    }
}

// Pass B (two opponents), lane `lane`: first hands p1 in [p1_lo, p1_hi), second hands lane, lane + 64, ...
MCQ_HD void mcq_exact_pass_b(const McqExactQuery &e, const McqExactBoard &bd, uint32_t lane, uint32_t p1_lo,
                             uint32_t p1_hi, const uint16_t *pair_xy, const uint32_t *keys, const uint16_t *rec,
                             McqExactAcc &acc) {
    for (uint32_t p1 = p1_lo; p1 < p1_hi; p1++) {
        const uint32_t xy1 = pair_xy[p1], x1 = xy1 & 0xFFu, y1 = xy1 >> 8, k1 = keys[p1], r1 = rec[p1];
        const uint32_t w1 = mcq_exact_w1(e.ref_law, mcq_exact_pa(r1), mcq_exact_pb(r1));
        for (uint32_t p2 = lane; p2 < MCQ_EXACT_PAIRS; p2 += 64u) {
            const uint32_t xy2 = pair_xy[p2], x2 = xy2 & 0xFFu, y2 = xy2 >> 8, r2 = rec[p2];
            const bool shared = x2 == x1 || x2 == y1 || y2 == x1 || y2 == y1;
            const bool ok = !shared && (!e.ref_law || mcq_exact_above(r1) + mcq_exact_above(r2) < bd.u);
            const uint32_t w = ok ? w1 * mcq_exact_w2(e.ref_law, r1, r2) : 0u;
            const uint32_t k2 = keys[p2], km = k1 > k2 ? k1 : k2;
            acc.win += km < bd.hero_key ? w : 0u;
            acc.tie += km == bd.hero_key ? w : 0u;
            acc.tot += w;
        }
    }
}

#endif /* MCQ_EXACT_HPP */
