// mcq_layout.hpp -- the host's layout of a batch of small queries for mcq_eval_direct_kernel (pure host code, no
// HIP: tests/hostsim checks its invariants where no GPU exists).
//
// Every query gets a power-of-two number of waves in proportion to its cost, about `want_waves` in all, so that all
// waves carry about the same work whatever the mix of players and table cards; the waves of a query sit side by side
// in one block.  The queries, sorted by descending wave count, are dealt to the blocks in turn: every block carries
// about the same number of waves, and inside a block the groups come in descending order, hence aligned to their
// size and never across a round's 16 waves.  Wave `v` of block `b` in round `r` works on slot (r * grid + b) * 16 + v.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string.h>

#include <vector>

#include "../../include/mcq.h"

#define MCQ_LAYOUT_IDLE 0xFFFFFFFFu
#define MCQ_LAYOUT_WAVES 16u /* waves of a block */

struct McqDirectLayout {
    uint32_t grid = 0, rounds = 0;
    bool merge = false;            /* some query has more than one wave */
    uint64_t waves = 0;
    size_t slots = 0;              /* rounds * grid * 16 */
    std::vector<uint8_t> lg;       /* per query: log2 of its wave count */
    std::vector<uint32_t> slot0;   /* per query: its first slot (its waves take slot0 .. slot0 + 2^lg - 1) */
    std::vector<uint32_t> sorted, fill; /* scratch */
};

/* cost[i]: scheduling cost of query i (0 = nothing to do); max_lg: finest cut allowed (waves per query <= 2^max_lg <= 16) */
static inline void mcq_direct_layout(const uint64_t *cost, size_t n, uint32_t n_cu, uint32_t max_lg, McqDirectLayout &L) {
    const uint64_t want_waves = (uint64_t)MCQ_LAYOUT_WAVES * n_cu;
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) total += cost[i];
    if (max_lg > 4u) max_lg = 4u;
    L.lg.resize(n);
    L.waves = 0;
    size_t count[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        uint32_t l = 0;
        const uint64_t mine = cost[i] * want_waves; /* 2^l <= the query's share of the waves */
        while (l < max_lg && (2ull << l) * total <= mine) l++;
        if (cost[i] == 0) l = 0;
        L.lg[i] = (uint8_t)l;
        count[l]++;
        L.waves += 1ull << l;
    }
    L.merge = count[0] != n;
    L.grid = (uint32_t)(n < (size_t)n_cu ? n : (size_t)n_cu);
    L.slot0.resize(n);
    if (L.grid == 0) { L.rounds = 0; L.slots = 0; return; }
    L.sorted.resize(n);
    {
        size_t at[5], pos = 0;
        for (int l = 4; l >= 0; l--) { at[l] = pos; pos += count[l]; }
        for (size_t i = 0; i < n; i++) L.sorted[at[L.lg[i]]++] = (uint32_t)i; /* counting sort, wide queries first */
    }
    /* dealt to the blocks in turn; a block's position counts waves, slot = (position / 16 * grid + block) * 16 + position % 16 */
    L.fill.assign(L.grid, 0u);
    uint32_t most = 0, b = 0;
    for (size_t k = 0; k < n; k++) {
        const uint32_t i = L.sorted[k], at = L.fill[b], w = 1u << L.lg[i];
        L.slot0[i] = ((at / MCQ_LAYOUT_WAVES) * L.grid + b) * MCQ_LAYOUT_WAVES + at % MCQ_LAYOUT_WAVES;
        L.fill[b] = at + w;
        most = at + w > most ? at + w : most;
        if (++b == L.grid) b = 0;
    }
    L.rounds = (most + MCQ_LAYOUT_WAVES - 1u) / MCQ_LAYOUT_WAVES;
    L.slots = (size_t)L.rounds * L.grid * MCQ_LAYOUT_WAVES;
}

/* The slot tables of a layout: slot_qi[s] = query index or MCQ_LAYOUT_IDLE, slot_sub[s] = the wave's cut number within
 * its query (both L.slots long).  The library writes its work records straight from slot0 / lg; this form is what the
 * kernel sees and what tests/hostsim checks. */
static inline void mcq_direct_layout_slots(const McqDirectLayout &L, uint32_t *slot_qi, uint8_t *slot_sub) {
    for (size_t s = 0; s < L.slots; s++) { slot_qi[s] = MCQ_LAYOUT_IDLE; slot_sub[s] = 0; }
    for (size_t i = 0; i < L.lg.size(); i++)
        for (uint32_t sub = 0; sub < (1u << L.lg[i]); sub++) {
            slot_qi[L.slot0[i] + sub] = (uint32_t)i;
            slot_sub[L.slot0[i] + sub] = (uint8_t)sub;
        }
}

/* The work of a launch as mcq_eval_direct_kernel reads it: per wave slot a copy of the query's 16-byte record whose
 * reserved bytes carry log2(waves of the query) and the wave's cut number, and the query's index (MCQ_LAYOUT_IDLE for a
 * slot without work).  rec / qi hold `cap` slots: false (nothing written) when the layout needs more.  Written as two
 * 64-bit words per record: a byte patched into a struct that is then copied whole stalls on its own store. */
static inline bool mcq_direct_write_records(const McqDirectLayout &L, const mcq_query *q, size_t n, void *rec, uint32_t *qi,
                                            size_t cap) {
    static_assert(offsetof(mcq_query, reserved) == 9 && sizeof(mcq_query) == 16, "record words");
    if (L.slots > cap || L.lg.size() != n || L.slot0.size() != n) return false;
    memset(qi, 0xFF, L.slots * sizeof(uint32_t)); /* MCQ_LAYOUT_IDLE */
    unsigned char *rec_bytes = static_cast<unsigned char *>(rec);
    for (size_t i = 0; i < n; i++) {
        uint64_t w[2];
        memcpy(w, &q[i], 16);
        const uint32_t l = L.lg[i], at = L.slot0[i];
        if ((size_t)at + (1u << l) > L.slots) return false;
        w[1] |= (uint64_t)l << 8; /* reserved[0] */
        for (uint32_t sub = 0; sub < (1u << l); sub++) {
            const uint64_t o[2] = {w[0], w[1] | ((uint64_t)sub << 16)}; /* reserved[1] */
            memcpy(rec_bytes + 16u * (size_t)(at + sub), o, 16);
            qi[at + sub] = (uint32_t)i;
        }
    }
    return true;
}
