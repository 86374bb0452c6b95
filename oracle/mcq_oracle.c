/*
 * mcq_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's Monte-Carlo equity path.
 *
 * This file is the parity oracle for the HIP path in neuron_poker_amd/csrc.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker / the reported CPU baseline.  Nothing under neuron_poker_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every reference-facing function below against fixtures that
 * were produced by running the reference itself (tests/golden/gen_golden.py imports
 * /root/reference/tools/{montecarlo_python,hand_evaluator}.py under np.random.seed): evaluator scores,
 * showdown winners, per-iteration dealt cards, MT19937 word counts and final tallies.
 *
 * It is deliberately a LITERAL restatement (lists with pops, sort-by-count tuples, tuple comparison);
 * the product computes the same results with bit masks and packed keys, so agreement is meaningful.
 *
 * Reference lines followed (paths relative to /root/reference):
 *   tools/hand_evaluator.py:27-119    _calc_score           -> calc_score()
 *   tools/hand_evaluator.py:20-24     eval_best_hand        -> best_hand()
 *   tools/montecarlo_python.py:114-119 create_card_deck     -> deck_init()
 *   tools/montecarlo_python.py:121-183 distribute_cards_to_players -> deal_iteration(), mcqo_run_ex()
 *   tools/montecarlo_python.py:185-189 distribute_cards_to_table   -> deal_iteration(), mcqo_run_ex()
 *   tools/montecarlo_python.py:191-252 run_montecarlo       -> mcqo_run(), mcqo_run_ex()
 *   tools/montecarlo_python.py:24-112  preflop classes / ranges    -> class_index(), in_range()
 *   numpy 1.26.4 (uv.lock:325-326; third party, not under /root/reference):
 *     numpy/random/src/mt19937/mt19937.c  init_genrand / genrand  -> mt_seed()/mt_next()
 *     numpy/random/src/distributions/distributions.c  buffered_bounded_masked_uint32 (legacy
 *     RandomState.randint, use_masked=True, one 32-bit word per trial)             -> np_randint()
 *
 * Card id c = 4*rank + suit, rank = "23456789TJQKA".index, suit = "CDHS".index (hand_evaluator.py:5-6).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ MT19937 */
typedef struct {
    uint32_t mt[624];
    int idx;
    uint64_t words; /* tempered words handed out so far */
} mt_t;

static void mt_seed(mt_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
    s->words = 0;
}

static void mt_refill(mt_t *s) {
    uint32_t *mt = s->mt;
    for (int k = 0; k < 624; k++) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->idx = 0;
}

static uint32_t mt_next(mt_t *s) {
    if (s->idx >= 624) mt_refill(s);
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    s->words++;
    return y;
}

/* np.random.randint(0, n) of the legacy RandomState: value in [0, n-1] */
static uint32_t np_randint(mt_t *s, uint32_t n) {
    uint32_t rng = n - 1;
    if (rng == 0) return 0; /* no word consumed */
    uint32_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
        v = mt_next(s) & mask;
    } while (v > rng);
    return v;
}

/* ------------------------------------------------------- counter-based front end (product's CTR mode) */
/* Spec (DESIGN.md "MCQ-CTR v5"): iterations are grouped in streams of 16; stream s of query id q under
 * seed k starts MWC64X (state x, c) from Philox4x32-10(counter = {q_lo, q_hi, s, 'MCQ1'}, key = {k_lo, k_hi}):
 * x = word 0, c = word 1 >> 1.
 * Opponent pair on a deck of length L, ONE word u, d = L-1: a = mulhi32(u, d), c = mulhi32(u * d mod 2^32, d);
 * (r1, r2) = (a, c) if a != c else (d, a) -- a bijection onto the pairs the reference accepts
 * (r1 in [0,L), r2 in [0,L-1), r1 != r2), so they are equally likely exactly as after its re-draw loop
 * (montecarlo_python.py:167-176); no re-draw happens, so `passes` counts one attempt per opponent.
 * Table cards, two per word: even draw K: u = next(), idx = mulhi32(u, n), w = u * n (mod 2^32);
 * odd draw: idx = mulhi32(w, n)   (n = current deck length - 1, montecarlo_python.py:188). */
#define STREAM_ITERS 16u

static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* MWC64X (David B. Thomas' multiply-with-carry generator: (x, c) -> (lo, hi) of A * x + c, output x ^ c) */
typedef struct { uint32_t x, c; } js_t;

static uint32_t js_next(js_t *s) {
    const uint32_t r = s->x ^ s->c;
    const uint64_t t = (uint64_t)4294883355u * s->x + s->c;
    s->x = (uint32_t)t;
    s->c = (uint32_t)(t >> 32);
    return r;
}

static void js_seed(js_t *s, uint64_t seed, uint64_t qid, uint32_t stream) {
    uint32_t ctr[4] = {(uint32_t)qid, (uint32_t)(qid >> 32), stream, 0x4D435131u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    philox4x32_10(ctr, key, o);
    s->x = o[0];
    s->c = o[1] >> 1;                              /* carry < A */
    if ((s->x | s->c) == 0) s->x = 0xf1ea5eedu;    /* (0, 0) is a fixed point */
}

static uint32_t js_draw(js_t *x, uint32_t n) { return (uint32_t)(((uint64_t)js_next(x) * n) >> 32); }

/* one RNG handle for the dealing code */
typedef struct {
    int kind; /* 0 = MT19937 + numpy randint, 1 = MWC64X + mulhi */
    mt_t *mt;
    js_t *xo;
} rng_t;

static uint32_t draw(rng_t *r, uint32_t n) { return r->kind == 0 ? np_randint(r->mt, n) : js_draw(r->xo, n); }
/* kind: 0 = MT19937 + numpy randint (the reference), 1 = MCQ-CTR v5 with the reference's dealing law,
 * 2 = MCQ-CTR v5 with the UNIFORM law (what montecarlo_cython.pyx:188 and Montecarlo.cpp:296-312 intend) */

/* ------------------------------------------------------------------------------------------ evaluator */
enum { T_HIGH, T_PAIR, T_TWOPAIR, T_TRIPS, T_STRAIGHT, T_FLUSH, T_FULL, T_QUADS, T_SF };

typedef struct {
    int nscore, score[8];
    int nranks, ranks[9];
    int type;
} score_t;

/* sort (cnt, rank) pairs descending -- sorted(...)[::-1], hand_evaluator.py:30 */
static void sort_pairs_desc(int *cnt, int *rank, int n) {
    for (int i = 1; i < n; i++) {
        int c = cnt[i], r = rank[i], j = i - 1;
        while (j >= 0 && (cnt[j] < c || (cnt[j] == c && rank[j] < r))) {
            cnt[j + 1] = cnt[j];
            rank[j + 1] = rank[j];
            j--;
        }
        cnt[j + 1] = c;
        rank[j + 1] = r;
    }
}

static void sort_desc(int *a, int n) {
    for (int i = 1; i < n; i++) {
        int v = a[i], j = i - 1;
        while (j >= 0 && a[j] < v) { a[j + 1] = a[j]; j--; }
        a[j + 1] = v;
    }
}

static int tuple_eq(const int *a, int na, const int *b, int nb) {
    if (na != nb) return 0;
    for (int i = 0; i < na; i++) if (a[i] != b[i]) return 0;
    return 1;
}

static int tuple_cmp(const int *a, int na, const int *b, int nb) { /* python tuple ordering */
    int n = na < nb ? na : nb;
    for (int i = 0; i < n; i++) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return na == nb ? 0 : (na < nb ? -1 : 1);
}

static int has(const int *a, int n, int v) {
    for (int i = 0; i < n; i++) if (a[i] == v) return 1;
    return 0;
}

/* rank histogram of `cards` -> distinct (cnt, rank) sorted descending */
static int rcounts(const uint8_t *cards, int n, int *cnt, int *rank) {
    int hist[13] = {0}, m = 0;
    for (int i = 0; i < n; i++) hist[cards[i] >> 2]++;
    for (int r = 0; r < 13; r++) if (hist[r]) { cnt[m] = hist[r]; rank[m] = r; m++; }
    sort_pairs_desc(cnt, rank, m);
    return m;
}

static void calc_score(const uint8_t *hand, int ncards, score_t *out) {
    int score[16], cr[16];
    int n = rcounts(hand, ncards, score, cr); /* l.29-30 */
    int nscore = n, ncr = n;

    static const int S22111[] = {2, 2, 1, 1, 1}, S211111[] = {2, 1, 1, 1, 1, 1};
    int pot_trips = score[0] == 3;                     /* l.32 */
    int pot_twopair = tuple_eq(score, nscore, S22111, 5);  /* l.33 */
    int pot_pair = tuple_eq(score, nscore, S211111, 6);    /* l.34 */

    if (nscore >= 2 && score[0] == 3 && (score[1] == 2 || score[1] == 3)) { /* l.36-38 */
        ncr = 2;
        score[0] = 3; score[1] = 2; nscore = 2;
    } else if (nscore >= 4 && score[0] == 2 && score[1] == 2 && score[2] == 2 && score[3] == 1) { /* l.39-42 */
        int kicker = cr[2] > cr[3] ? cr[2] : cr[3];
        cr[2] = kicker; ncr = 3;
        score[0] = 2; score[1] = 2; score[2] = 1; nscore = 3;
    } else if (score[0] == 4) { /* l.43-46 */
        sort_desc(cr, ncr);
        ncr = 2;
        score[0] = 4; nscore = 1;
    } else if (nscore >= 5) { /* l.47-83 */
        int straight = 0, flush;
        if (has(cr, ncr, 12)) cr[ncr++] = -1; /* l.49-50 */
        int sorted[16], ns = ncr;
        memcpy(sorted, cr, sizeof(int) * ncr);
        sort_desc(sorted, ns); /* l.51 */
        for (int i = 0; i < ns - 4; i++) { /* l.52-58 */
            straight = sorted[i] - sorted[i + 4] == 4;
            if (straight) {
                for (int k = 0; k < 5; k++) cr[k] = sorted[i + k];
                ncr = 5;
                break;
            }
        }
        int sc[4] = {0};
        for (int i = 0; i < ncards; i++) sc[hand[i] & 3]++; /* l.61-62 */
        flush = 0;
        for (int s = 0; s < 4; s++) if (sc[s] >= 5) flush = 1;
        if (flush) {
            int fs = 0;
            for (fs = 0; fs < 4; fs++) if (sc[fs] >= 5) break; /* l.64-66: first suit in "CDHS" order */
            uint8_t fh[8];
            int nf = 0;
            for (int i = 0; i < ncards; i++) if ((hand[i] & 3) == fs) fh[nf++] = hand[i]; /* l.68 */
            int fcnt[16];
            ncr = rcounts(fh, nf, fcnt, cr); /* l.69-70 */
            memcpy(score, fcnt, sizeof(int) * ncr);
            nscore = ncr;
            sort_desc(cr, ncr); /* l.71-72 */
            if (has(cr, ncr, 12) && !has(cr, ncr, -1)) cr[ncr++] = -1; /* l.75-76 */
            for (int i = 0; i < ncr - 4; i++) { /* l.77-80 */
                straight = cr[i] - cr[i + 4] == 4;
                if (straight) break;
            }
        }
        /* l.83 */
        if (!flush && !straight) { score[0] = 1; nscore = 1; }
        else if (!flush && straight) { score[0] = 3; score[1] = 1; score[2] = 2; nscore = 3; }
        else if (flush && !straight) { score[0] = 3; score[1] = 1; score[2] = 3; nscore = 3; }
        else { score[0] = 5; nscore = 1; }
    }

    if (nscore == 1 && score[0] == 1 && pot_trips) { score[0] = 3; score[1] = 1; nscore = 2; }        /* l.85-86 */
    else if (nscore == 1 && score[0] == 1 && pot_twopair) { score[0] = 2; score[1] = 2; score[2] = 1; nscore = 3; }
    else if (nscore == 1 && score[0] == 1 && pot_pair) { score[0] = 2; score[1] = 1; score[2] = 1; nscore = 3; }

    int type;
    if (score[0] == 5) type = T_SF; /* l.92-117 */
    else if (score[0] == 4) type = T_QUADS;
    else if (nscore >= 2 && score[0] == 3 && score[1] == 2) type = T_FULL;
    else if (nscore >= 3 && score[0] == 3 && score[1] == 1 && score[2] == 3) { type = T_FLUSH; if (ncr > 5) ncr = 5; }
    else if (nscore >= 3 && score[0] == 3 && score[1] == 1 && score[2] == 2) { type = T_STRAIGHT; if (ncr > 5) ncr = 5; }
    else if (nscore >= 2 && score[0] == 3 && score[1] == 1) { type = T_TRIPS; if (ncr > 3) ncr = 3; }
    else if (nscore >= 2 && score[0] == 2 && score[1] == 2) { type = T_TWOPAIR; if (ncr > 3) ncr = 3; }
    else if (score[0] == 2) { type = T_PAIR; if (ncr > 4) ncr = 4; }
    else if (score[0] == 1) { type = T_HIGH; if (ncr > 5) ncr = 5; }
    else type = -1; /* 'Card Type error!' */

    out->nscore = nscore;
    memcpy(out->score, score, sizeof(int) * (nscore > 8 ? 8 : nscore));
    out->nranks = ncr > 9 ? 9 : ncr;
    memcpy(out->ranks, cr, sizeof(int) * out->nranks);
    out->type = type;
}

static int score_cmp(const score_t *a, const score_t *b) {
    int c = tuple_cmp(a->score, a->nscore, b->score, b->nscore);
    if (c) return c;
    return tuple_cmp(a->ranks, a->nranks, b->ranks, b->nranks);
}

/* eval_best_hand: stable descending sort, first element -> lowest index among the maxima. *tie = 1 when
 * another hand compares equal to the winner (extension: the reference does not report it). */
static int best_hand(const uint8_t *hands, int n, int *type, int *tie) {
    score_t best, s;
    int w = 0, t = 0;
    calc_score(hands, 7, &best);
    for (int i = 1; i < n; i++) {
        calc_score(hands + 7 * i, 7, &s);
        int c = score_cmp(&s, &best);
        if (c > 0) { best = s; w = i; t = 0; }
        else if (c == 0) t = 1;
    }
    if (type) *type = best.type;
    if (tie) *tie = t;
    return w;
}

/* ------------------------------------------------------------------------------------------ dealing */
typedef struct { uint8_t c[52]; int n; } deck_t;

static void deck_init(deck_t *d) { /* '2C','2D','2H','2S','3C',... == ascending id */
    for (int i = 0; i < 52; i++) d->c[i] = (uint8_t)i;
    d->n = 52;
}

static uint8_t deck_pop(deck_t *d, int i) {
    uint8_t v = d->c[i];
    memmove(d->c + i, d->c + i + 1, (size_t)(d->n - i - 1));
    d->n--;
    return v;
}

static int deck_remove(deck_t *d, uint8_t card) {
    for (int i = 0; i < d->n; i++) if (d->c[i] == card) { deck_pop(d, i); return 1; }
    return 0;
}

/* one iteration: fills hands[n_players][7]; returns passes added */
static uint32_t deal_iteration(rng_t *rng, const uint8_t hero[2], const uint8_t *board, int nb, int n_players,
                               uint8_t *hands) {
    deck_t d;
    uint32_t passes = 0;
    uint8_t table[5];
    uint8_t hole[10][2];
    deck_init(&d);
    for (int i = 0; i < nb; i++) { deck_remove(&d, board[i]); table[i] = board[i]; } /* l.126-128 */
    hole[0][0] = hero[0]; hole[0][1] = hero[1];
    deck_remove(&d, hero[0]); /* l.154-161 */
    deck_remove(&d, hero[1]);
    for (int p = 1; p < n_players && rng->kind == 2; p++) { /* MCQ-CTR v5, UNIFORM law */
        uint32_t dd = (uint32_t)d.n - 1, u = js_next(rng->xo);
        uint32_t r1 = (uint32_t)(((uint64_t)u * (dd + 1)) >> 32);                   /* in [0, L) */
        uint32_t r2 = (uint32_t)(((uint64_t)(uint32_t)(u * (dd + 1)) * dd) >> 32);  /* in [0, L-1): every ordered pair */
        passes++;
        hole[p][0] = deck_pop(&d, (int)r1);
        hole[p][1] = deck_pop(&d, (int)r2);
    }
    for (int p = 1; p < n_players && rng->kind == 1; p++) { /* MCQ-CTR v5 */
        uint32_t dd = (uint32_t)d.n - 1, u = js_next(rng->xo);
        uint32_t a = (uint32_t)(((uint64_t)u * dd) >> 32), c = (uint32_t)(((uint64_t)(uint32_t)(u * dd) * dd) >> 32);
        uint32_t r1 = a != c ? a : dd, r2 = a != c ? c : a;
        passes++;
        hole[p][0] = deck_pop(&d, (int)r1);
        hole[p][1] = deck_pop(&d, (int)r2);
    }
    for (int p = 1; p < n_players && rng->kind == 0; p++) { /* l.165-181 */
        uint32_t r1, r2;
        for (;;) {
            passes++;
            r1 = draw(rng, (uint32_t)d.n);
            r2 = draw(rng, (uint32_t)d.n - 1);
            if (r1 != r2) break; /* l.172; the range test l.175 is always true at opponent_range = 1 */
        }
        hole[p][0] = deck_pop(&d, (int)r1);
        hole[p][1] = deck_pop(&d, (int)r2);
    }
    if (rng->kind == 0) {
        for (int k = nb; k < 5; k++) table[k] = deck_pop(&d, (int)draw(rng, (uint32_t)d.n - 1)); /* l.186-188 */
    } else {
        uint32_t w = 0;
        for (int k = nb; k < 5; k++) {
            uint32_t n = (uint32_t)d.n - (rng->kind == 2 ? 0u : 1u); /* uniform law: any remaining card */
            if (((k - nb) & 1) == 0) {
                uint32_t u = js_next(rng->xo);
                table[k] = deck_pop(&d, (int)(((uint64_t)u * n) >> 32));
                w = u * n;
            } else {
                table[k] = deck_pop(&d, (int)(((uint64_t)w * n) >> 32));
            }
        }
    }
    for (int p = 0; p < n_players; p++) { /* l.218-220 */
        hands[7 * p] = hole[p][0];
        hands[7 * p + 1] = hole[p][1];
        memcpy(hands + 7 * p + 2, table, 5);
    }
    return passes;
}

/* out[13] = runs, passes, win (strict), tie, by_type[9]; the reference's wins = win + tie = sum(by_type) */
static void tally(uint64_t *out, const uint8_t *hands, int n_players) {
    int type, tie;
    int w = best_hand(hands, n_players, &type, &tie);
    if (w == 0) {
        if (tie) out[3]++; else out[2]++;
        out[4 + type]++;
    }
}

static int valid_query(const uint8_t hero[2], const uint8_t *board, int nb, int n_players) {
    uint64_t seen = 0;
    if (nb < 0 || nb > 5 || n_players < 1 || n_players > 10) return 0;
    for (int i = 0; i < 2 + nb; i++) {
        uint8_t c = i < 2 ? hero[i] : board[i - 2];
        if (c >= 52 || (seen >> c) & 1) return 0;
        seen |= 1ull << c;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------ exported */
int mcqo_version(void) { return 1; }

void mcqo_mt_words(uint32_t seed, uint32_t n, uint32_t *out) {
    mt_t s;
    mt_seed(&s, seed);
    for (uint32_t i = 0; i < n; i++) out[i] = mt_next(&s);
}

/* draws np.random.randint(0, bounds[i]) for i < n after np.random.seed(seed); returns words consumed */
uint64_t mcqo_np_randint(uint32_t seed, uint32_t n, const uint32_t *bounds, uint32_t *out) {
    mt_t s;
    mt_seed(&s, seed);
    for (uint32_t i = 0; i < n; i++) out[i] = np_randint(&s, bounds[i]);
    return s.words;
}

void mcqo_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }

void mcqo_ctr_stream(uint64_t seed, uint64_t qid, uint32_t stream, uint32_t n, uint32_t *out) {
    js_t x;
    js_seed(&x, seed, qid, stream);
    for (uint32_t i = 0; i < n; i++) out[i] = js_next(&x);
}

/* out[0]=nscore, out[1..3]=score (padded 0), out[4]=nranks, out[5..13]=ranks (padded -128), out[14]=type */
void mcqo_calc_score(const uint8_t *cards, int ncards, int32_t *out) {
    score_t s;
    calc_score(cards, ncards, &s);
    out[0] = s.nscore;
    for (int i = 0; i < 3; i++) out[1 + i] = i < s.nscore ? s.score[i] : 0;
    out[4] = s.nranks;
    for (int i = 0; i < 9; i++) out[5 + i] = i < s.nranks ? s.ranks[i] : -128;
    out[14] = s.type;
}

/* -1/0/+1: python comparison of _calc_score(a) vs _calc_score(b) */
int mcqo_compare(const uint8_t *a, const uint8_t *b) {
    score_t x, y;
    calc_score(a, 7, &x);
    calc_score(b, 7, &y);
    return score_cmp(&x, &y);
}

int mcqo_best_hand(const uint8_t *hands, int n, int *type, int *tie) { return best_hand(hands, n, type, tie); }

/* mode 0: np.random.seed((uint32)seed) then the reference loop; mode 1: MCQ-CTR v5 with query id qid.
 * trace (optional): first `keep` iterations' hands [keep][n_players][7]; words (optional, mode 0): MT words
 * per kept iteration; total_words (optional). Returns 0, or -1 on invalid input. */
int mcqo_run(int mode, const uint8_t *hero, const uint8_t *board, int nb, int n_players, uint32_t runs,
             uint64_t seed, uint64_t qid, uint64_t *out, uint8_t *trace, uint32_t keep, uint16_t *words,
             uint64_t *total_words) {
    if (!valid_query(hero, board, nb, n_players)) return -1;
    mt_t mt;
    js_t xo;
    rng_t rng = {mode, &mt, &xo};
    uint8_t hands[70];
    memset(out, 0, 13 * sizeof(uint64_t));
    mt.words = 0;
    if (mode == 0) mt_seed(&mt, (uint32_t)seed);
    for (uint32_t it = 0; it < runs; it++) {
        if (mode >= 1 && it % STREAM_ITERS == 0) js_seed(&xo, seed, qid, it / STREAM_ITERS);
        uint64_t w0 = mt.words;
        out[1] += deal_iteration(&rng, hero, board, nb, n_players, hands);
        out[0]++;
        tally(out, hands, n_players);
        if (it < keep) {
            if (trace) memcpy(trace + (size_t)it * 7 * n_players, hands, (size_t)7 * n_players);
            if (words) words[it] = (uint16_t)(mt.words - w0);
        }
    }
    if (total_words) *total_words = mode == 0 ? mt.words : 0;
    return 0;
}

/* Iterations [it_begin, it_end) of a CTR-mode query (mode 1 or 2): what one share of an iteration-split batch
 * (include/mcq.h mcq_eval_batch_part) must return.  The stream of it_begin is walked from its start. */
int mcqo_run_range(int mode, const uint8_t *hero, const uint8_t *board, int nb, int n_players, uint64_t seed,
                   uint64_t qid, uint32_t it_begin, uint32_t it_end, uint64_t *out) {
    if (mode < 1 || !valid_query(hero, board, nb, n_players)) return -1;
    mt_t mt;
    js_t xo;
    rng_t rng = {mode, &mt, &xo};
    uint8_t hands[70];
    uint64_t scratch[13];
    memset(out, 0, 13 * sizeof(uint64_t));
    for (uint32_t it = it_begin - it_begin % STREAM_ITERS; it < it_end; it++) {
        if (it % STREAM_ITERS == 0) js_seed(&xo, seed, qid, it / STREAM_ITERS);
        uint64_t passes = deal_iteration(&rng, hero, board, nb, n_players, hands);
        uint64_t *dst = it >= it_begin ? out : scratch;
        dst[1] += passes;
        dst[0]++;
        tally(dst, hands, n_players);
    }
    return 0;
}

/* batch: query i = {hole[2], board[5], n_board, n_players, runs(u32 LE)} (16 bytes, the C-ABI's mcq_query);
 * MT mode seeds query i with (uint32)(seed + first_qid + i); CTR mode uses query id first_qid + i.
 * out[n][13].  threads >= 1 (pthreads, queries interleaved).  Used as the timed CPU baseline. */
typedef struct {
    int mode, tid, nthreads;
    const uint8_t *q;
    size_t n;
    uint64_t seed, first_qid;
    uint64_t *out;
    int err;
} job_t;

static void *batch_worker(void *p) {
    job_t *j = (job_t *)p;
    for (size_t i = (size_t)j->tid; i < j->n; i += (size_t)j->nthreads) {
        const uint8_t *q = j->q + 16 * i;
        uint32_t runs;
        memcpy(&runs, q + 12, 4);
        uint64_t qid = j->first_qid + i;
        uint64_t seed = j->mode == 0 ? (uint32_t)(j->seed + qid) : j->seed;
        if (mcqo_run(j->mode, q, q + 2, q[7], q[8], runs, seed, qid, j->out + 13 * i, 0, 0, 0, 0)) j->err = 1;
    }
    return 0;
}

int mcqo_run_batch(int mode, const uint8_t *queries, size_t n, uint64_t seed, uint64_t first_qid, uint64_t *out,
                   int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    job_t jobs[256];
    int err = 0;
    for (int t = 0; t < threads; t++) {
        jobs[t] = (job_t){mode, t, threads, queries, n, seed, first_qid, out, 0};
        if (t > 0) pthread_create(&th[t], 0, batch_worker, &jobs[t]);
    }
    batch_worker(&jobs[0]);
    for (int t = 1; t < threads; t++) pthread_join(th[t], 0);
    for (int t = 0; t < threads; t++) err |= jobs[t].err;
    return err ? -1 : 0;
}

/* ------------------------------------------------------------------------- ranges, ghost cards, known hands
 * montecarlo_python.py:36-112 (preflop classes), :133-163 (known hands, hero given as a set of
 * classes), :165-181 (opponents restricted to a range), :206-208 (ghost cards).
 * A range is a 169-bit set; the bit of two cards is how get_two_short_notation (:24-34) names them:
 * suited -> 13*min+max, off-suit -> 13*max+min, pair -> 14*rank (both spellings of a class map to one bit). */
static int class_index(uint8_t a, uint8_t b) {
    int ra = a >> 2, rb = b >> 2, lo = ra < rb ? ra : rb, hi = ra < rb ? rb : ra;
    if (ra == rb) return 14 * ra;
    return (a & 3) == (b & 3) ? 13 * lo + hi : 13 * hi + lo;
}

static int in_range(const uint32_t *bits, uint8_t a, uint8_t b) {
    int i = class_index(a, b);
    return !bits || ((bits[i >> 5] >> (i & 31)) & 1u);
}

/* Extended queries in full (montecarlo_python.py:121-189): `n_known` known hands in the order of
 * original_player_card_list (known[0] = hero), each either two cards (known_cards[h] < 52) or a set of preflop
 * classes (known_cards[h][0] == 0xFF, known_ranges[h] = 6 words); ghost: two cards taken out of the deck or NULL;
 * opp_range: 6 words or NULL (= every class).
 *
 * mode 0 (MT19937): the reference's loops, literally -- a range hand looks at deck[r1], deck[r2] on the UNPOPPED list
 * and leaves by value (l.136-161; a card a LIST hand names that an earlier range hand has already drawn is simply not
 * there any more: the reference's try/except), a random opponent is tested on the unpopped list and then popped in turn
 * (l.165-181).
 *
 * mode 1 (production, "MCQ-CTR v5x"): the same LAW without index arithmetic.  The reference accepts every ordered
 * index pair (r1, r2), r1 in [0,L), r2 in [0,L-1), r1 != r2, whose classes are allowed, equally often; as cards that
 * is every ordered pair (A, B) of distinct cards of the current deck with B not the deck's highest card and
 * class(A, B) allowed.  Per draw (range hand h, or the opponents) there is a fixed candidate list
 *     P' = [(a, b) for a in 0..51 for b in 0..51 if a != b and a, b in U and class(a, b) allowed]   (this order)
 * over U = 52 cards minus ghost, table and the cards of the LIST hands before h (all list hands for the opponents);
 * a trial takes one word u, (A, B) = P'[mulhi32(u, len(P'))], and is accepted iff A and B are still in the deck and
 * B is not its highest card; passes counts trials.  A range hand leaves by value; an opponent is dealt A and, as
 * deck.pop(r1); deck.pop(r2) deal, B if B lies below A, else the card that follows B in the deck.  Table cards as
 * in the plain mode (two per word, never the highest card).  Opponents to whom every class is allowed are dealt by
 * index exactly as the plain mode deals them (one word per pair, never re-drawn), so a query that restricts nothing
 * gives the plain mode's tallies.  65536 failed trials in a row = the range cannot be dealt: -2.  Streams: sixteen
 * iterations each as in the plain mode, but EX_SHORT_STREAM for a query of at most EX_SHORT_RUNS iterations that draws
 * from at least one candidate list (see below). */
#define EX_MAX_TRIALS 65536u
#define EX_SHORT_STREAM 2u
#define EX_SHORT_RUNS 8192u
int mcqo_run_ex2(int mode, int n_known, const uint8_t *known_cards, const uint32_t *known_ranges, const uint8_t *ghost,
                 const uint8_t *board, int nb, int n_players, uint32_t runs, uint64_t seed, uint64_t qid,
                 const uint32_t *opp_range, uint64_t *out, uint64_t *total_words) {
    uint64_t seen = 0;
    if (mode < 0 || mode > 1 || nb < 0 || nb > 5 || n_known < 1 || n_known > 10 || n_players < n_known || n_players > 10) return -1;
#define EX_SEE(c) do { if ((c) >= 52 || (seen >> (c)) & 1) return -1; seen |= 1ull << (c); } while (0)
    for (int i = 0; i < nb; i++) EX_SEE(board[i]);
    if (ghost) { EX_SEE(ghost[0]); EX_SEE(ghost[1]); }
    for (int h = 0; h < n_known; h++)
        if (known_cards[2 * h] != 0xFF) { EX_SEE(known_cards[2 * h]); EX_SEE(known_cards[2 * h + 1]); }
#undef EX_SEE
    mt_t mt;
    js_t xo;
    rng_t rng = {mode, &mt, &xo};
    memset(out, 0, 13 * sizeof(uint64_t));
    mt.words = 0;
    if (mode == 0) mt_seed(&mt, (uint32_t)seed);
    deck_t original;
    deck_init(&original);
    if (ghost) { deck_remove(&original, ghost[0]); deck_remove(&original, ghost[1]); } /* l.206-208 */
    /* opponents to whom every class is allowed are dealt by index, as the plain mode deals them */
    int opp_all = opp_range == 0;
    if (opp_range) {
        uint32_t all = opp_range[5] | ~0x1FFu;
        for (int i = 0; i < 5; i++) all &= opp_range[i];
        opp_all = all == 0xFFFFFFFFu;
    }
    /* production mode: the candidate lists (index n_known = the opponents') */
    static __thread uint16_t plist[11][2704];
    uint32_t pcount[11];
    /* streams: STREAM_ITERS iterations each, but EX_SHORT_STREAM for a query of at most EX_SHORT_RUNS iterations that
     * draws from at least one candidate list (a trial loop takes an unknown number of words, so such a stream cannot be
     * entered half way: short streams are what lets a small query spread over many lanes) */
    uint32_t stream_iters = STREAM_ITERS;
    if (mode == 1) {
        uint64_t u = (1ull << 52) - 1;
        if (ghost) u &= ~((1ull << ghost[0]) | (1ull << ghost[1]));
        for (int i = 0; i < nb; i++) u &= ~(1ull << board[i]);
        for (int h = 0; h <= n_known; h++) {
            const uint32_t *set = h < n_known ? known_ranges + 6 * h : opp_range;
            const int wanted = h < n_known ? known_cards[2 * h] == 0xFF : (n_players > n_known && !opp_all);
            pcount[h] = 0;
            if (wanted)
                for (int a = 0; a < 52; a++)
                    for (int b = 0; b < 52; b++)
                        if (a != b && (u >> a) & 1 && (u >> b) & 1 && in_range(set, (uint8_t)a, (uint8_t)b))
                            plist[h][pcount[h]++] = (uint16_t)(a | (b << 8));
            if (wanted && pcount[h] == 0) return -2;
            if (wanted && runs <= EX_SHORT_RUNS) stream_iters = EX_SHORT_STREAM;
            if (h < n_known && known_cards[2 * h] != 0xFF)
                u &= ~((1ull << known_cards[2 * h]) | (1ull << known_cards[2 * h + 1]));
        }
    }
    for (uint32_t it = 0; it < runs; it++) {
        if (mode == 1 && it % stream_iters == 0) js_seed(&xo, seed, qid, it / stream_iters);
        deck_t d = original;
        uint8_t table[5], hole[10][2], hands[70];
        int w = 0; /* table draws of this iteration (CTR: two per word) */
        uint32_t word = 0;
        for (int i = 0; i < nb; i++) { deck_remove(&d, board[i]); table[i] = board[i]; } /* l.126-128 */
        for (int p = 0; p < n_players; p++) {
            const int is_known = p < n_known, is_list = is_known && known_cards[2 * p] != 0xFF;
            if (is_list) { /* l.150-161 */
                hole[p][0] = known_cards[2 * p]; hole[p][1] = known_cards[2 * p + 1];
                deck_remove(&d, hole[p][0]);
                deck_remove(&d, hole[p][1]);
                continue;
            }
            const uint32_t *set = is_known ? known_ranges + 6 * p : opp_range;
            uint8_t A, B;
            if (mode == 0) {
                uint32_t r1, r2;
                for (uint32_t trial = 0;; trial++) { /* l.136-148 / l.165-176 */
                    if (trial >= EX_MAX_TRIALS * 16u) return -2;
                    out[1]++;
                    r1 = np_randint(&mt, (uint32_t)d.n);
                    r2 = np_randint(&mt, (uint32_t)d.n - 1);
                    if (r1 != r2 && in_range(set, d.c[r1], d.c[r2])) break;
                }
                if (is_known) { /* the two cards looked at are the hand; they leave by value */
                    A = d.c[r1]; B = d.c[r2];
                    deck_remove(&d, A);
                    deck_remove(&d, B);
                } else { /* deck.pop(r1); deck.pop(r2) on the shrunk list (l.178-179) */
                    A = deck_pop(&d, (int)r1);
                    B = deck_pop(&d, (int)r2);
                }
            } else if (!is_known && opp_all) { /* MCQ-CTR v5, one word, never re-drawn */
                uint32_t dd = (uint32_t)d.n - 1, u = js_next(&xo);
                uint32_t a = (uint32_t)(((uint64_t)u * dd) >> 32), c = (uint32_t)(((uint64_t)(uint32_t)(u * dd) * dd) >> 32);
                out[1]++;
                A = deck_pop(&d, (int)(a != c ? a : dd));
                B = deck_pop(&d, (int)c);
            } else {
                const int li = is_known ? p : n_known;
                for (uint32_t trial = 0;; trial++) {
                    if (trial >= EX_MAX_TRIALS) return -2;
                    out[1]++;
                    const uint32_t k = (uint32_t)(((uint64_t)js_next(&xo) * pcount[li]) >> 32);
                    A = (uint8_t)(plist[li][k] & 0xFF);
                    B = (uint8_t)(plist[li][k] >> 8);
                    int ia = -1, ib = -1;
                    for (int i = 0; i < d.n; i++) { if (d.c[i] == A) ia = i; if (d.c[i] == B) ib = i; }
                    if (ia >= 0 && ib >= 0 && ib != d.n - 1) {
                        if (!is_known && ib > ia) B = d.c[ib + 1]; /* deck.pop(r2) after deck.pop(r1) */
                        break;
                    }
                }
                deck_remove(&d, A);
                deck_remove(&d, B);
            }
            hole[p][0] = A; hole[p][1] = B;
        }
        for (int k = nb; k < 5; k++, w++) { /* l.186-188 */
            uint32_t n = (uint32_t)d.n - 1, idx;
            if (mode == 0) idx = np_randint(&mt, n);
            else if ((w & 1) == 0) { uint32_t u = js_next(&xo); idx = (uint32_t)(((uint64_t)u * n) >> 32); word = u * n; }
            else idx = (uint32_t)(((uint64_t)word * n) >> 32);
            table[k] = deck_pop(&d, (int)idx);
        }
        for (int q = 0; q < n_players; q++) {
            hands[7 * q] = hole[q][0];
            hands[7 * q + 1] = hole[q][1];
            memcpy(hands + 7 * q + 2, table, 5);
        }
        out[0]++;
        tally(out, hands, n_players);
    }
    if (total_words) *total_words = mode == 0 ? mt.words : 0;
    (void)rng;
    return 0;
}

/* the two-hand form of round 1 (hero + at most one further known hand) on top of the general one */
int mcqo_run_ex(int mode, const uint8_t *hero, const uint32_t *hero_range, const uint8_t *known2, const uint8_t *ghost,
                const uint8_t *board, int nb, int n_players, uint32_t runs, uint64_t seed, uint64_t qid,
                const uint32_t *opp_range, uint64_t *out, uint64_t *total_words) {
    uint8_t cards[4] = {0xFF, 0xFF, 0xFF, 0xFF};
    uint32_t ranges[12];
    memset(ranges, 0, sizeof ranges);
    if ((hero == 0) == (hero_range == 0)) return -1;
    if (hero) { cards[0] = hero[0]; cards[1] = hero[1]; }
    else memcpy(ranges, hero_range, 24);
    if (known2) { cards[2] = known2[0]; cards[3] = known2[1]; }
    return mcqo_run_ex2(mode, known2 ? 2 : 1, cards, ranges, ghost, board, nb, n_players, runs, seed, qid, opp_range, out,
                        total_words);
}

/* ------------------------------------------------------------------------- exact expectation (small cases)
 * Exact probabilities of {hero wins strictly, hero wins a tie} under the REFERENCE'S dealing law (uniform = 0; the
 * accepted (r1, r2) pairs are equally likely; each table draw is uniform on [0, L-2]).  Enumerates the
 * whole tree, so only for n_players <= 3 and enough known board cards; uniform = 1 enumerates the unbiased
 * law instead (every ordered pair, every remaining table card).  out[0] = P(win strictly),
 * out[1] = P(tie, hero credited), out[2] = number of leaves.  Returns -1 if the tree is too large. */
typedef struct {
    const uint8_t *hero, *board;
    int nb, n_players, uniform;
    double win, tie, leaves;
} enum_t;

static void enum_table(enum_t *e, deck_t *d, uint8_t hole[][2], uint8_t *table, int k, double w) {
    if (k == 5) {
        uint8_t hands[70];
        int type, tie;
        for (int p = 0; p < e->n_players; p++) {
            hands[7 * p] = hole[p][0];
            hands[7 * p + 1] = hole[p][1];
            memcpy(hands + 7 * p + 2, table, 5);
        }
        if (best_hand(hands, e->n_players, &type, &tie) == 0) { if (tie) e->tie += w; else e->win += w; }
        e->leaves += 1;
        return;
    }
    int choices = d->n - (e->uniform ? 0 : 1);
    for (int i = 0; i < choices; i++) {
        deck_t d2 = *d;
        table[k] = deck_pop(&d2, i);
        enum_table(e, &d2, hole, table, k + 1, w / choices);
    }
}

static void enum_players(enum_t *e, deck_t *d, uint8_t hole[][2], uint8_t *table, int p, double w) {
    if (p == e->n_players) { enum_table(e, d, hole, table, e->nb, w); return; }
    int L = d->n;
    double pairs = e->uniform ? (double)L * (L - 1) : (double)(L - 1) * (L - 1); /* accepted ordered (r1, r2) */
    for (int r1 = 0; r1 < L; r1++)
        for (int r2 = 0; r2 < L - 1; r2++) {
            if (r1 == r2 && !e->uniform) continue;
            deck_t d2 = *d;
            hole[p][0] = deck_pop(&d2, r1);
            hole[p][1] = deck_pop(&d2, r2);
            enum_players(e, &d2, hole, table, p + 1, w / pairs);
        }
}

int mcqo_exact(const uint8_t *hero, const uint8_t *board, int nb, int n_players, int uniform, double *out) {
    if (!valid_query(hero, board, nb, n_players)) return -1;
    double leaves = 1;
    int L = 50 - nb;
    for (int p = 1; p < n_players; p++) { leaves *= (double)(L - 1) * (uniform ? L : L - 1); L -= 2; }
    for (int k = nb; k < 5; k++) { leaves *= uniform ? L : L - 1; L--; }
    if (leaves > 3e8) return -1;
    enum_t e = {hero, board, nb, n_players, uniform, 0, 0, 0};
    deck_t d;
    uint8_t hole[10][2], table[5];
    deck_init(&d);
    for (int i = 0; i < nb; i++) { deck_remove(&d, board[i]); table[i] = board[i]; }
    hole[0][0] = hero[0]; hole[0][1] = hero[1];
    deck_remove(&d, hero[0]);
    deck_remove(&d, hero[1]);
    enum_players(&e, &d, hole, table, 1, 1.0);
    out[0] = e.win; out[1] = e.tie; out[2] = e.leaves;
    return 0;
}
