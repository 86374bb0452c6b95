// hostsim.cpp -- TEST HARNESS ONLY (built and loaded by tests/, never by the product).
//
// Compiles the product's per-lane arithmetic (neuron_poker_amd/csrc/mcq_device.hpp, mcq_replay.hpp) for the
// HOST compiler and walks the kernels' task/lane decomposition sequentially, so that the lane code can be
// checked against the oracle and the golden fixtures in a container that has no GPU.  It is not a CPU
// backend: libmcq_hip.so contains none of this and fails loudly without a GPU.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../neuron_poker_amd/csrc/mcq_device.hpp"
#include "../../neuron_poker_amd/csrc/mcq_exact.hpp"
#include "../../neuron_poker_amd/csrc/mcq_layout.hpp"
#include "../../neuron_poker_amd/csrc/mcq_mt.hpp"
#include "../../neuron_poker_amd/csrc/mcq_mt_ext.hpp"
#include "../../neuron_poker_amd/csrc/mcq_mt_blocks.hpp"
#include "../../neuron_poker_amd/csrc/mcq_replay.hpp"

namespace {
McqTables g_tab;
bool g_init = false;
const McqTables &luts() {
    if (!g_init) { mcq_fill_tables(&g_tab); g_init = true; }
    return g_tab;
}
// the iteration takes the base-deck pointer biased by -128 entries: keep 128 unused entries in front
void make_base(const McqQueryCtx &qc, const McqTables &t, McqCard *store) {
    for (uint32_t l = 0; l < 64; l++) store[128 + l] = mcq_base_entry(qc, l, t.sel8);
}
void fold(const McqLaneAcc &a, mcq_result *r) {
    uint64_t wins = 0;
    for (uint32_t c = 0; c < MCQ_N_CODES; c++) {
        if (c == 5) continue;
        uint64_t v = (a.types >> (6 * c)) & 63;
        r->by_type[mcq_code_to_type(c)] += v;
        wins += v;
    }
    r->tie += a.tie;
    r->win += wins - a.tie;
    r->passes += a.passes;
}
}  // namespace

extern "C" {

int hs_query_valid(const mcq_query *q) { return mcq_query_valid(mcq_query_words(*q)) ? 1 : 0; }

// keys of 7-card hands: cards 0,1 = hole, 2..6 = table (the split does not matter for the key)
void hs_eval7(const uint8_t *cards, size_t n, uint32_t *keys) {
    const McqTables &t = luts();
    for (size_t i = 0; i < n; i++) {
        McqBoard b;
        b.clear();
        for (int k = 2; k < 7; k++) b.add(mcq_card(cards[7 * i + k]));
        McqHole h;
        h.set(mcq_card(cards[7 * i]), mcq_card(cards[7 * i + 1]));
        McqFlushSel fs;
        fs.from_board(b);
        keys[i] = mcq_eval_key(b, fs, h, t.tf, t.tops, t.sd);
    }
}
uint32_t hs_key_type(uint32_t key) { return mcq_key_type(key); }

uint32_t hs_select_pop(uint32_t *dlo, uint32_t *dhi, uint32_t k) { return mcq_select_pop(*dlo, *dhi, k, luts().sel8); }

void hs_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
    mcq_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

// production mode, lane/stream decomposition exactly as the kernel: lane <-> stream of 16 iterations
}  // extern "C"

template <class Draws, bool STRAIGHT = true>
static int run_ctr_t(const mcq_query *q, uint64_t seed, uint64_t qid, mcq_result *out) {
    if (!mcq_query_valid(mcq_query_words(*q))) return MCQ_EINVAL;
    const McqTables &t = luts();
    McqQueryCtx qc;
    mcq_query_ctx(mcq_query_words(*q), qc);
    static thread_local McqCard base[192];
    make_base(qc, t, base);
    memset(out, 0, sizeof(*out));
    out->runs = q->runs;
    uint32_t n_streams = (q->runs + MCQ_STREAM_ITERS - 1) / MCQ_STREAM_ITERS;
    for (uint32_t s = 0; s < n_streams; s++) {
        Draws dr;
        dr.start(seed, qid, s);
        McqLaneAcc acc = {0, 0, 0};
        const uint64_t left = (uint64_t)q->runs - (uint64_t)s * MCQ_STREAM_ITERS;
        const uint32_t cnt = left < MCQ_STREAM_ITERS ? (uint32_t)left : MCQ_STREAM_ITERS;
        if (STRAIGHT) mcq_iterations<true>(qc, dr, base, t.tf, t.tops, t.sd, acc, cnt); /* as the bulk kernel runs them */
        else for (uint32_t j = 0; j < cnt; j++) mcq_iteration(qc, dr, base, t.tf, t.tops, t.sd, acc);
        acc.passes += cnt * qc.n_opp; /* MCQ-CTR v5: one attempt per opponent, never re-drawn */
        fold(acc, out);
    }
    return MCQ_OK;
}

extern "C" {

int hs_run_ctr(const mcq_query *q, uint64_t seed, uint64_t qid, mcq_result *out) {
    return run_ctr_t<McqCtrDraws>(q, seed, qid, out);
}
int hs_run_ctr_uniform(const mcq_query *q, uint64_t seed, uint64_t qid, mcq_result *out) {
    return run_ctr_t<McqCtrDrawsUniform>(q, seed, qid, out);
}
int hs_run_ctr_general(const mcq_query *q, uint64_t seed, uint64_t qid, mcq_result *out) { /* without the straight-line forms */
    return run_ctr_t<McqCtrDraws, false>(q, seed, qid, out);
}

// parity mode: host parse of the MT19937 stream + the same lane arithmetic
int hs_run_replay(const mcq_query *q, uint32_t seed32, mcq_result *out) {
    if (!mcq_query_valid(mcq_query_words(*q))) return MCQ_EINVAL;
    const McqTables &t = luts();
    McqQueryCtx qc;
    mcq_query_ctx(mcq_query_words(*q), qc);
    static thread_local McqCard base[192];
    make_base(qc, t, base);
    memset(out, 0, sizeof(*out));
    out->runs = q->runs;
    size_t stride = q->runs ? q->runs : 1;
    std::vector<uint8_t> draws((size_t)mcq_draws_per_iteration(*q) * stride + 4);
    out->passes = mcq_replay_parse(*q, seed32, draws.data(), stride);
    /* as the kernel reads them: four iterations per 32-bit load of every draw row (McqReplayDraws4) */
    for (uint32_t it4 = 0; it4 < q->runs; it4 += 4) {
        McqReplayDraws4 dr;
        dr.load(draws.data() + it4, stride, qc.n_opp, qc.n_deal);
        for (uint32_t k = 0; k < 4 && it4 + k < q->runs; k++) {
            dr.sh = 8u * k;
            McqLaneAcc acc = {0, 0, 0};
            mcq_iteration(qc, dr, base, t.tf, t.tops, t.sd, acc);
            acc.passes = 0;
            fold(acc, out);
        }
    }
    return MCQ_OK;
}

// parity mode continuing an MT19937 stream given as numpy state (key[624], pos); state updated in place
int hs_run_replay_stream(const mcq_query *q, uint32_t *key, uint32_t *pos, mcq_result *out) {
    if (!mcq_query_valid(mcq_query_words(*q))) return MCQ_EINVAL;
    const McqTables &t = luts();
    McqQueryCtx qc;
    mcq_query_ctx(mcq_query_words(*q), qc);
    static thread_local McqCard base[192];
    make_base(qc, t, base);
    memset(out, 0, sizeof(*out));
    out->runs = q->runs;
    size_t stride = q->runs ? q->runs : 1;
    std::vector<uint8_t> draws((size_t)mcq_draws_per_iteration(*q) * stride + 1);
    McqMt19937 g;
    memcpy(g.mt, key, sizeof g.mt);
    g.pos = *pos;
    out->passes = mcq_replay_parse_stream(*q, g, draws.data(), stride);
    memcpy(key, g.mt, sizeof g.mt);
    *pos = g.pos;
    for (uint32_t it = 0; it < q->runs; it++) {
        McqReplayDraws dr = {draws.data() + it, stride};
        McqLaneAcc acc = {0, 0, 0};
        mcq_iteration(qc, dr, base, t.tf, t.tops, t.sd, acc);
        acc.passes = 0;
        fold(acc, out);
    }
    return MCQ_OK;
}

void hs_mt_words(uint32_t seed, uint32_t n, uint32_t *out) {
    McqMt19937 g;
    g.seed(seed);
    for (uint32_t i = 0; i < n; i++) out[i] = g.next();
}
}

// ---- extended queries (ranges, hero range, ghost cards, second known hand)

extern "C" int hs_run_ext(int replay, const mcq_query *q, const mcq_query_ext *e, uint64_t seed, uint64_t qid,
                          mcq_result *out) {
    const McqExtRec er = {reinterpret_cast<const uint32_t *>(e)};
    const McqQueryWords qw = mcq_query_words(*q);
    if (!mcq_query_ext_valid(qw, er)) return MCQ_EINVAL;
    const McqTables &t = luts();
    McqExtCtx qc;
    mcq_ext_ctx(qw, er, qc);
    McqExtWaveCtx wc;
    memset(&wc, 0, sizeof wc);
    for (uint32_t h = 0; h < qc.n_hands; h++) wc.hand[h] = mcq_ext_hand(qw, er, h);
    /* the candidate lists, as mcq_ext_lists_kernel lays them out */
    const uint32_t n_lists = mcq_ext_n_lists(qw, er);
    std::vector<uint16_t> lists((size_t)(n_lists ? n_lists : 1) * MCQ_EXT_LIST_STRIDE);
    for (uint32_t li = 0; li < n_lists; li++) {
        uint64_t U;
        uint32_t set_off, cnt = 0;
        mcq_ext_list_plan(qw, er, li, U, set_off);
        for (uint32_t c = 0; c < 2704u; c++)
            if (mcq_ext_candidate(U, er.w + set_off, c)) lists[(size_t)li * MCQ_EXT_LIST_STRIDE + cnt++] = (uint16_t)((c / 52u) | ((c % 52u) << 8));
        wc.cnt[li] = cnt;
        wc.list[li] = lists.data() + (size_t)li * MCQ_EXT_LIST_STRIDE;
        if (cnt == 0 && !replay) return MCQ_EINVAL;
    }
    McqCard cards[64];
    for (uint32_t c = 0; c < 64; c++) cards[c] = mcq_card(c < 52 ? c : 0);
    memset(out, 0, sizeof(*out));
    out->runs = q->runs;
    uint16_t ids[MCQ_MAX_OPP + 1];
    if (replay) {
        size_t stride = q->runs ? q->runs : 1;
        std::vector<uint8_t> draws((size_t)mcq_ext_draws_per_iteration(*q, *e) * stride + 1);
        McqMt19937 g;
        g.seed((uint32_t)seed);
        uint64_t passes = mcq_replay_parse_ext(*q, *e, g, draws.data(), stride, 1000000u);
        if (passes == ~0ull) return MCQ_EINVAL;
        out->passes = passes;
        for (uint32_t it = 0; it < q->runs; it++) {
            McqExtReplayDraws dr = {draws.data() + it, stride};
            McqLaneAcc acc = {0, 0, 0};
            mcq_iteration_ext(qc, wc, dr, cards, t.sel8, ids, 1, t.tf, t.tops, t.sd, acc);
            acc.passes = 0;
            fold(acc, out);
        }
        return MCQ_OK;
    }
    const uint32_t s_iters = mcq_ext_stream_iters(qw, er);
    uint32_t n_streams = (q->runs + s_iters - 1) / s_iters;
    for (uint32_t s = 0; s < n_streams; s++) {
        McqExtCtrDraws dr;
        dr.start(seed, qid, s);
        McqLaneAcc acc = {0, 0, 0};
        for (uint32_t j = 0; j < s_iters; j++) {
            if ((uint64_t)s * s_iters + j >= q->runs) break;
            if (!(qc.fast ? mcq_iteration_ext_fast(qc, wc, dr, cards, t.sel8, t.tf, t.tops, t.sd, acc)   /* as the kernel picks */
                          : mcq_iteration_ext(qc, wc, dr, cards, t.sel8, ids, 1, t.tf, t.tops, t.sd, acc)))
                return MCQ_EINVAL;
        }
        fold(acc, out);
    }
    return MCQ_OK;
}


// exact enumeration (mcq_exact.hpp) walked like mcq_exact_kernel: completion by completion, pass A then pass B,
// the 64 lanes one after the other
extern "C" int hs_exact(const mcq_query *q, int law, uint64_t *out13) {
    const McqTables &t = luts();
    McqExactQuery e;
    if (!mcq_exact_query(mcq_query_words(*q), law, e)) return -1;
    std::vector<uint16_t> pair_xy(MCQ_EXACT_PAIRS), rec(MCQ_EXACT_PAIRS);
    std::vector<uint32_t> keys(MCQ_EXACT_PAIRS);
    for (uint32_t i = 0; i < MCQ_EXACT_PAIRS; i++) {
        uint32_t x, y;
        mcq_exact_pair_xy(i, x, y);
        pair_xy[i] = (uint16_t)(x | (y << 8));
    }
    for (int k = 0; k < 13; k++) out13[k] = 0;
    const uint32_t n_boards = mcq_exact_binom(e.L, e.k);
    const bool two = e.n_opp == 2;
    for (uint32_t board = 0; board < n_boards; board++) {
        uint32_t pos[5];
        mcq_exact_unrank(board, e.L, e.k, pos);
        McqExactBoard bd;
        mcq_exact_board(e, pos, t.sel8, t.tf, t.tops, t.sd, bd);
        McqCard rem_card[64];
        uint32_t rem_pos[64];
        for (uint32_t l = 0; l < MCQ_EXACT_REM; l++) {
            rem_pos[l] = mcq_exact_rem_pos(pos, l);
            rem_card[l] = mcq_card(mcq_exact_card_at(e, rem_pos[l], t.sel8));
        }
        uint64_t win = 0, tie = 0, tot = 0;
        McqExactAcc acc[64];
        for (uint32_t lane = 0; lane < 64; lane++) {
            acc[lane] = {0, 0, 0};
            mcq_exact_pass_a(e, bd, lane, pair_xy.data(), rem_card, rem_pos, t.tf, t.tops, t.sd,
                             two ? keys.data() : nullptr, rec.data(), acc[lane]);
        }
        if (two)
            for (uint32_t lane = 0; lane < 64; lane++)
                mcq_exact_pass_b(e, bd, lane, 0, MCQ_EXACT_PAIRS, pair_xy.data(), keys.data(), rec.data(), acc[lane]);
        for (uint32_t lane = 0; lane < 64; lane++) { win += acc[lane].win; tie += acc[lane].tie; tot += acc[lane].tot; }
        out13[0] += tot;
        out13[2] += win;
        out13[3] += tie;
        out13[4 + mcq_key_type(bd.hero_key)] += win + tie;
    }
    return 0;
}

// ---- the wave-cooperative MT19937 parse of the device (mcq_mt.hpp), compiled with every per-lane variable as a
// 64-entry array: draws / passes must equal the sequential host walk (mcq_replay_parse) byte for byte
extern "C" uint64_t hs_mt_parse(const mcq_query *q, uint32_t seed32, uint8_t *draws, uint64_t stride) {
    static thread_local McqMtWave w;
    const uint32_t n_opp = q->n_players - 1u, n_deal = 5u - q->n_board;
    if (2u * n_opp + n_deal == 0u || q->runs == 0u) return 0;
    mcq_mt_seed(w, seed32);
    McqMtState st = {MCQ_MT_N, 0, 0, 0, 0};
    mcq_mt_parse_query(w, st, 50u - q->n_board, n_opp, n_deal, q->runs, draws, stride);
    return st.passes;
}
// ... the same stream parsed block by block (mcq_mt_blocks.hpp: generate, scan every entry state, stitch, parse each block
// from its true entry) as the four kernels of the device do it; n_blocks = 0: as many as the host would estimate.
// UINT64_MAX: the stream did not end within the blocks (the device then falls back to the serial walk)
extern "C" uint32_t hs_mtb_blocks_needed(const mcq_query *q) {
    return mcq_mtb_blocks_needed(50u - q->n_board, q->n_players - 1u, 5u - q->n_board, q->runs);
}
extern "C" uint64_t hs_mt_parse_blocks(const mcq_query *q, uint32_t seed32, uint8_t *draws, uint64_t stride, uint32_t n_blocks) {
    static thread_local McqMtWave gen;
    static thread_local McqMtBlockWave w;
    const uint32_t n_opp = q->n_players - 1u, n_deal = 5u - q->n_board, L0 = 50u - q->n_board, D = 2u * n_opp + n_deal;
    if (D == 0u || q->runs == 0u) return 0;
    if (!n_blocks) n_blocks = hs_mtb_blocks_needed(q);
    /* 1. generate */
    std::vector<uint8_t> yb((size_t)n_blocks * MCQ_MT_N);
    mcq_mt_seed(gen, seed32);
    for (uint32_t b = 0; b < n_blocks; b++) {
        mcq_mt_regenerate(gen);
        for (uint32_t k = 0; k < MCQ_MT_N; k++) yb[(size_t)b * MCQ_MT_N + k] = (uint8_t)mcq_mt_word_yb(gen, McqMtState(), k);
    }
    /* 2. scan */
    const McqMtbPlan pl = mcq_mtb_plan(L0, n_opp, n_deal, q->runs);
    std::vector<uint32_t> exits((size_t)n_blocks * MCQ_MTB_LANES);
    uint32_t pos_tab[MCQ_MTB_POS];
    for (uint32_t k = 0; k < MCQ_MTB_POS; k++) pos_tab[k] = mcq_mtb_pos_word(pl, k < D ? k : 0u);
    /* as the kernel does it (a block in U parts, each walked from every entry state, the block's exit words composed from
     * the parts') -- and checked against the walk of the whole block */
    const uint32_t U = mcq_mtb_parts(pl), W = MCQ_MT_N / U, n_st = pl.D + (pl.two_opp >> 1);
    if (U * n_st > 64u || U * W != MCQ_MT_N) return ~0ull - 2u;
    for (uint32_t b = 0; b < n_blocks; b++) {
        uint32_t part[8][MCQ_MTB_LANES];
        for (uint32_t u = 0; u < U; u++)
            for (uint32_t l = 0; l < MCQ_MTB_LANES; l++)
                part[u][l] = l < n_st ? mcq_mtb_automaton_n(&yb[(size_t)b * MCQ_MT_N + u * W], W, pos_tab, pl, l) : 0u;
        for (uint32_t l = 0; l < MCQ_MTB_LANES; l++) {
            const uint32_t x = mcq_mtb_exit_from_parts(&part[0][0], U, pl, l);
            if (x != mcq_mtb_automaton(&yb[(size_t)b * MCQ_MT_N], pos_tab, pl, l)) return ~0ull - 2u; /* parts != whole: a bug */
            exits[(size_t)b * MCQ_MTB_LANES + l] = x;
        }
    }
    /* 3. stitch: compose the groups, follow the groups, note every block's entry from its group's */
    const uint32_t n_groups = (n_blocks + MCQ_MTB_GROUP - 1u) / MCQ_MTB_GROUP;
    std::vector<uint32_t> gword((size_t)n_groups * MCQ_MTB_LANES), gits((size_t)n_groups * MCQ_MTB_LANES);
    for (uint32_t g = 0; g < n_groups; g++)
        for (uint32_t l = 0; l < MCQ_MTB_LANES; l++) {
            McqMtbWalk s = mcq_mtb_walk_from(pl, l);
            if (l < pl.D + (pl.two_opp >> 1))
                for (uint32_t b = g * MCQ_MTB_GROUP; b < n_blocks && b < (g + 1u) * MCQ_MTB_GROUP; b++)
                    mcq_mtb_compose_step(&exits[(size_t)b * MCQ_MTB_LANES], pl, s);
            gword[(size_t)g * MCQ_MTB_LANES + l] = mcq_mtb_walk_word(s);
            gits[(size_t)g * MCQ_MTB_LANES + l] = s.its;
        }
    std::vector<McqMtbEntry> entry(n_blocks);
    uint32_t d = 0, pend = 0, it = 0;
    for (uint32_t g = 0; g < n_groups; g++) {
        uint32_t bd = d, bp = pend, bi = it; /* the group's entry: its blocks follow from there */
        for (uint32_t b = g * MCQ_MTB_GROUP; b < n_blocks && b < (g + 1u) * MCQ_MTB_GROUP; b++) {
            entry[b].it0 = bi;
            entry[b].dp = bd | (bp << 8) | (bi < q->runs ? 0x80000000u : 0u);
            mcq_mtb_stitch_step(&exits[(size_t)b * MCQ_MTB_LANES], pl, bd, bp, bi);
        }
        mcq_mtb_stitch_group(&gword[(size_t)g * MCQ_MTB_LANES], &gits[(size_t)g * MCQ_MTB_LANES], pl, d, pend, it);
        if (bd != d || bi != it || (mcq_mtb_is_r2(pl, d) && bp != pend)) return ~0ull - 1u; /* the two levels disagree: a bug */
    }
    if (it < q->runs) return ~0ull;
    /* 4. parse */
    uint64_t passes = 0;
    w.draws = draws;
    w.stride = stride;
    w.two_opp = 2u * n_opp;
    mcq_mt_fill_ptab(w, L0, n_opp, D);
    for (uint32_t b = 0; b < n_blocks; b++) {
        if (!(entry[b].dp >> 31)) continue;
        memcpy(w.yb, &yb[(size_t)b * MCQ_MT_N], MCQ_MT_N);
        memset(w.yb + MCQ_MT_N, 0xFF, 64);
        passes += mcq_mtb_parse_block(w, L0, n_opp, n_deal, q->runs, entry[b]);
    }
    return passes;
}
// ... and the same for extended queries (mcq_mt_ext.hpp) against mcq_replay_parse_ext; UINT64_MAX: cannot be dealt
extern "C" uint64_t hs_mt_parse_ext(const mcq_query *q, const mcq_query_ext *e, uint32_t seed32, uint8_t *draws, uint64_t stride) {
    static thread_local McqMtExtWave w;
    if (q->runs == 0u) return 0;
    mcq_mt_seed(w, seed32);
    McqMtExtState st = {MCQ_MT_N, 0, 0, 0, 0, 0, 0, false};
    if (!mcq_mt_parse_query_ext(w, st, mcq_query_words(*q), reinterpret_cast<const uint32_t *>(e), draws, stride)) return ~0ull;
    return st.passes;
}
extern "C" uint64_t hs_mt_parse_ext_reference(const mcq_query *q, const mcq_query_ext *e, uint32_t seed32, uint8_t *draws, uint64_t stride) {
    McqMt19937 g;
    g.seed(seed32);
    return mcq_replay_parse_ext(*q, *e, g, draws, stride, 1000000u);
}
extern "C" uint32_t hs_ext_draws_per_iteration(const mcq_query *q, const mcq_query_ext *e) { return mcq_ext_draws_per_iteration(*q, *e); }
extern "C" uint64_t hs_mt_parse_reference(const mcq_query *q, uint32_t seed32, uint8_t *draws, uint64_t stride) {
    return mcq_replay_parse(*q, seed32, draws, stride);
}
extern "C" void hs_mt_regenerate_words(uint32_t seed32, uint32_t n, uint32_t *out) { /* tempered words via the wave code */
    static thread_local McqMtWave w;
    mcq_mt_seed(w, seed32);
    uint32_t pos = MCQ_MT_N;
    for (uint32_t i = 0; i < n; i++) {
        if (pos >= MCQ_MT_N) { mcq_mt_regenerate(w); pos = 0; }
        out[i] = mcq_mt_temper(w.mt[pos++]);
    }
}
extern "C" uint32_t hs_mt_magic_ok(void) { /* (p * magic) >> 16 == p / D for every p the parse can form */
    for (uint32_t D = 1; D <= MCQ_MT_MAX_DRAWS; D++)
        for (uint32_t p = 0; p < MCQ_MT_POSITIONS; p++)
            if (((p * mcq_mt_magic(D)) >> 16) != p / D) return 0;
    return 1;
}

// ---- the host's wave layout of the one-launch path (mcq_layout.hpp): slot_qi / slot_sub / lg out, returns rounds
extern "C" uint32_t hs_direct_layout(const uint64_t *cost, size_t n, uint32_t n_cu, uint32_t max_lg, uint32_t *grid,
                                     uint8_t *lg, uint32_t *slot_qi, uint8_t *slot_sub, size_t slot_cap) {
    McqDirectLayout L;
    mcq_direct_layout(cost, n, n_cu, max_lg, L);
    *grid = L.grid;
    if (L.slots > slot_cap) return 0xFFFFFFFFu;
    memcpy(lg, L.lg.data(), n);
    mcq_direct_layout_slots(L, slot_qi, slot_sub);
    return L.rounds;
}
// ... and the records the library writes from it (mcq_direct_write_records): rec = slots x 16 bytes, qi = slots words;
// returns the slot count, 0 when `cap` slots do not hold the layout
extern "C" size_t hs_direct_records(const uint64_t *cost, const mcq_query *q, size_t n, uint32_t n_cu, uint32_t max_lg, uint8_t *rec,
                                    uint32_t *qi, size_t cap) {
    McqDirectLayout L;
    mcq_direct_layout(cost, n, n_cu, max_lg, L);
    if (!mcq_direct_write_records(L, q, n, rec, qi, cap)) return 0;
    return L.slots;
}
