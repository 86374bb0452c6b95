// mcq_tables.cpp -- native lock-step driver for many Hold'em tables on the GPU equity (BASELINE configs[4]).
//
// The same table rules as neuron_poker_amd/table_driver.py (which is pinned, event by event, to seeded episodes
// of the reference's own gym_env/env.py + gym_env/cycle.py), as an explicit state machine per table instead of a
// Python generator: every table always has ONE pending equity query; mcq_tables_begin() collects them,
// mcq_tables_resume() feeds the equities back and advances every table to its next query (finished episodes
// restart at once).  mcq_tables_run() = begin -> mcq_eval_batch (ONE call per lock-step) -> resume.
// tests/test_table_driver.py drives begin/resume with a deterministic equity function and checks that every
// table follows the Python driver step by step (same per-table xoshiro128++ stream for dealing and for the
// random seats).
//
// Reference lines (paths relative to /root/reference): gym_env/env.py:138-688, gym_env/cycle.py:10-167,
// agents/agent_consider_equity.py:25-56, agents/agent_random.py:21-29; showdowns use this library's own
// evaluator (mcq_eval_key compiled for the host) = tools/hand_evaluator.py:9-24.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mcq_device.hpp"

namespace {

enum { FOLD, CHECK, CALL, RAISE_3BB, RAISE_HALF_POT, RAISE_POT, RAISE_2POT, ALL_IN, SMALL_BLIND, BIG_BLIND };
enum { PREFLOP, FLOP, TURN, RIVER, END_HIDDEN, SHOWDOWN };
enum { PH_FIRST, PH_A, PH_B };
constexpr int kMaxSeats = 10;

const McqTables &host_tables() {
    static McqTables *t = [] {
        McqTables *p = new McqTables();
        mcq_fill_tables(p);
        return p;
    }();
    return *t;
}

uint32_t hand_key(const uint8_t *hole, const uint8_t *table5) {
    const McqTables &t = host_tables();
    McqBoard b;
    b.clear();
    for (int k = 0; k < 5; k++) b.add(mcq_card(table5[k]));
    McqHole h;
    h.set(mcq_card(hole[0]), mcq_card(hole[1]));
    McqFlushSel fs;
    fs.from_board(b);
    return mcq_eval_key(b, fs, h, t.tf, t.tops, t.sd);
}

struct Table {
    // configuration (shared values copied for locality)
    int n;
    double initial_stacks, small_blind, big_blind;
    int max_raises;
    int extra_queries = 0; /* HoldemTable(calculate_equity=True): three more equity calls per observation (env.py:248-256) */
    int extra_left = -1;   /* of the observation in hand; -1 = no observation begun */
    const uint8_t *seat_kind;
    const double *min_call_eq, *min_bet_eq;
    McqXoshiro rng;

    // cycle (gym_env/cycle.py)
    int idx, dealer_idx, step_counter, max_steps_total, max_steps_after_raiser, max_steps_after_big_blind;
    int last_raiser, checkers, max_remaining_steps_without_raising;
    bool can_move[kMaxSeats], out_of_cash[kMaxSeats], folder[kMaxSeats], alive[kMaxSeats];

    // table (gym_env/env.py)
    double stacks[kMaxSeats], player_pots[kMaxSeats], player_max_win[kMaxSeats];
    double community_pot, current_round_pot, min_call;
    int num_raises[kMaxSeats][4];
    uint8_t cards[kMaxSeats][2];
    int n_cards[kMaxSeats];
    uint8_t table_cards[5];
    int n_table;
    uint8_t deck[52];
    int n_deck;
    int stage, current, winner_ix;
    bool done;
    uint32_t legal; /* bit mask over actions */
    int phase;
    bool issued; /* the pending query was handed out by begin() */
    // statistics
    uint64_t env_steps, queries, episodes;
    uint64_t ep_env_steps, ep_queries; /* of the running episode */

    uint32_t draw(uint32_t bound) { return mcq_mulhi(rng.next(), bound); }

    int n_alive() const {
        int s = 0;
        for (int i = 0; i < n; i++) s += alive[i];
        return s;
    }
    void update_alive() {
        for (int i = 0; i < n; i++) alive[i] = can_move[i] || out_of_cash[i];
    }

    // ---- cycle.py
    void new_hand_reset() {
        idx = 0;
        for (int i = 0; i < n; i++) { can_move[i] = true; out_of_cash[i] = false; folder[i] = false; }
        step_counter = 0;
    }
    void new_street_reset() {
        step_counter = 0;
        idx = dealer_idx;
        checkers = 0;
        max_remaining_steps_without_raising = n - 1;
        last_raiser = 0;
    }
    int next_player() { /* -1 = the round is over (cycle.py:58-101) */
        int movers = 0;
        for (int i = 0; i < n; i++) movers += can_move[i] || out_of_cash[i];
        if (movers < 2) return -1;
        idx += 1;
        step_counter += 1;
        idx %= n;
        if (max_steps_total && step_counter > max_steps_total) return -1;
        if (last_raiser) {
            if (step_counter > last_raiser + max_remaining_steps_without_raising) return -1;
            if (max_steps_after_raiser && step_counter > max_steps_after_raiser + last_raiser) return -1;
        } else if (max_steps_after_raiser && step_counter > max_steps_after_big_blind + 2) {
            return -1;
        }
        if (checkers == n_alive()) return -1;
        while (!can_move[idx]) {
            idx += 1;
            step_counter += 1;
            idx %= n;
            if (max_steps_total && step_counter >= max_steps_total) return -1;
        }
        update_alive();
        return idx;
    }
    void next_dealer() {
        dealer_idx = ((dealer_idx + 1) % n + n) % n;
        while (!can_move[dealer_idx]) dealer_idx = (dealer_idx + 1) % n;
    }

    // ---- env.py
    uint8_t deck_pop(int i) {
        uint8_t v = deck[i];
        memmove(deck + i, deck + i + 1, (size_t)(n_deck - i - 1));
        n_deck--;
        return v;
    }
    void reset_episode() { /* env.py:138-168 */
        done = false;
        for (int i = 0; i < n; i++) {
            stacks[i] = initial_stacks;
            n_cards[i] = 0;
            player_pots[i] = player_max_win[i] = 0;
            alive[i] = true;
            for (int s = 0; s < 4; s++) num_raises[i][s] = 0;
        }
        n_table = 0;
        stage = PREFLOP;
        winner_ix = -1;
        current = -1;
        community_pot = 0;
        current_round_pot = 9;
        min_call = 0;
        legal = 0;
        max_steps_total = 0;
        max_steps_after_raiser = (max_raises - 1) * n - 1;
        max_steps_after_big_blind = n;
        last_raiser = 0;
        step_counter = 0;
        idx = 0;
        dealer_idx = -1;
        checkers = 0;
        max_remaining_steps_without_raising = n;
        new_hand_reset();
        ep_env_steps = ep_queries = 0;
        start_new_hand();
        phase = PH_FIRST;
    }
    bool check_game_over() {
        new_hand_reset();
        int live = 0;
        for (int s = 0; s < n; s++) {
            if (stacks[s] > 0) live++;
            else can_move[s] = false;
        }
        if (live < 2 || stacks[0] == 0) { done = true; return true; }
        return false;
    }
    void start_new_hand() {
        for (int i = 0; i < n; i++)
            for (int s = 0; s < 4; s++) num_raises[i][s] = 0;
        if (check_game_over()) return;
        n_table = 0;
        for (int i = 0; i < 52; i++) deck[i] = (uint8_t)i;
        n_deck = 52;
        stage = PREFLOP;
        community_pot = 0;
        current_round_pot = 0;
        for (int i = 0; i < n; i++) { player_pots[i] = player_max_win[i] = 0; n_cards[i] = 0; }
        next_dealer();
        for (int s = 0; s < n; s++) {
            if (stacks[s] <= 0) continue;
            for (int k = 0; k < 2; k++) cards[s][n_cards[s]++] = deck_pop((int)draw((uint32_t)n_deck));
        }
        initiate_round();
    }
    void initiate_round() {
        min_call = 0;
        new_street_reset();
        if (stage != PREFLOP && n == 2) idx += 1;
        if (stage == PREFLOP) {
            max_steps_total = n * max_raises + 2;
            go_next_player();
            process_decision(SMALL_BLIND);
            go_next_player();
            process_decision(BIG_BLIND);
            go_next_player();
        } else if (stage == FLOP || stage == TURN || stage == RIVER) {
            max_steps_total = n * max_raises;
            go_next_player();
        }
    }
    void go_next_player() {
        current = next_player();
        if (current < 0) {
            if (n_alive() < 2) stage = END_HIDDEN;
            else {
                end_round();
                initiate_round();
            }
        }
    }
    void clean_up_pots() {
        community_pot += current_round_pot;
        current_round_pot = 0;
        for (int i = 0; i < n; i++) player_pots[i] = 0;
    }
    void deal_table(int k) {
        for (int i = 0; i < k; i++) table_cards[n_table++] = deck_pop((int)draw((uint32_t)n_deck));
    }
    void end_round() {
        double s = 0;
        for (int i = 0; i < n; i++) { s += player_pots[i]; player_pots[i] = 0; }
        community_pot += s;
        if (stage == PREFLOP) { stage = FLOP; deal_table(3); }
        else if (stage == FLOP) { stage = TURN; deal_table(1); }
        else if (stage == TURN) { stage = RIVER; deal_table(1); }
        else if (stage == RIVER) stage = SHOWDOWN;
        clean_up_pots();
    }
    void end_hand() {
        clean_up_pots();
        int idxs[kMaxSeats], m = 0;
        for (int i = 0; i < n; i++)
            if ((can_move[i] || out_of_cash[i]) && !folder[i]) idxs[m++] = i;
        int w = m ? idxs[0] : 0;
        if (m > 1) {
            uint32_t best = 0;
            for (int k = 0; k < m; k++) {
                uint32_t key = hand_key(cards[idxs[k]], table_cards);
                if (key > best) { best = key; w = idxs[k]; } /* first of equals (hand_evaluator.py:23) */
            }
        }
        const double cap = player_max_win[w];
        double total = 0, sum = 0;
        for (int i = 0; i < n; i++) { total += player_max_win[i] < cap ? player_max_win[i] : cap; sum += player_max_win[i]; }
        stacks[w] += total;
        winner_ix = w;
        if (total < sum)
            for (int i = 0; i < n; i++) stacks[i] += player_max_win[i] > cap ? player_max_win[i] - cap : 0.0;
    }
    void process_decision(int action) {
        const int seat = current;
        if (action == FOLD) {
            can_move[idx] = false;
            folder[idx] = true;
        } else {
            const double pot = community_pot + current_round_pot;
            double contribution = 0;
            switch (action) {
                case CALL: contribution = min_call - player_pots[seat] < stacks[seat] ? min_call - player_pots[seat] : stacks[seat]; break;
                case CHECK: contribution = 0; checkers += 1; break;
                case RAISE_3BB: contribution = 3 * big_blind - player_pots[seat]; break;
                case RAISE_HALF_POT: contribution = pot / 2; break;
                case RAISE_POT: contribution = pot; break;
                case RAISE_2POT: contribution = pot * 2; break;
                case ALL_IN: contribution = stacks[seat]; break;
                case SMALL_BLIND: contribution = small_blind < stacks[seat] ? small_blind : stacks[seat]; break;
                case BIG_BLIND:
                    contribution = big_blind < stacks[seat] ? big_blind : stacks[seat];
                    max_steps_total = step_counter + n * max_raises + 2; /* mark_bb */
                    break;
            }
            if (action >= RAISE_3BB && action <= ALL_IN) num_raises[seat][stage] += 1;
            if (contribution > min_call && action != SMALL_BLIND && action != BIG_BLIND) last_raiser = step_counter;
            stacks[seat] -= contribution;
            player_pots[seat] += contribution;
            current_round_pot += contribution;
            if (stacks[seat] == 0 && contribution > 0) { /* mark_out_of_cash_but_contributed */
                out_of_cash[idx] = true;
                can_move[idx] = false;
            }
            if (contribution > min_call) min_call = contribution;
            player_max_win[seat] += contribution;
        }
        update_alive();
    }
    void legal_moves() {
        legal = 0;
        if (stage == SHOWDOWN) return;
        const int seat = current;
        double mx = 0;
        for (int i = 0; i < n; i++) mx = player_pots[i] > mx ? player_pots[i] : mx;
        if (player_pots[seat] == mx) legal |= 1u << CHECK;
        else legal |= (1u << CALL) | (1u << FOLD);
        if (num_raises[seat][stage < 3 ? stage : 3] < max_raises) {
            const double pot = community_pot + current_round_pot, stack = stacks[seat];
            if (stack >= 3 * big_blind - player_pots[seat]) legal |= 1u << RAISE_3BB;
            if (stack >= pot / 2 && pot / 2 >= min_call) legal |= 1u << RAISE_HALF_POT;
            if (stack >= pot && pot >= min_call) legal |= 1u << RAISE_POT;
            if (stack >= pot * 2 && pot * 2 >= min_call) legal |= 1u << RAISE_2POT;
            if (stack > 0) legal |= 1u << ALL_IN;
        }
    }

    // ---- observation = one equity query (env.py:224-281)
    void observe(mcq_query &q, uint32_t runs) {
        if (extra_left < 0) { /* a new observation */
            if (!done) legal_moves();
            if (current < 0) current = winner_ix;
            extra_left = extra_queries;
        }
        if (!issued) { /* asking again for the same pending query does not count twice */
            queries++;
            ep_queries++;
            issued = true;
        }
        memset(&q, 0, sizeof q);
        q.hole[0] = cards[current][0];
        q.hole[1] = cards[current][1];
        for (int i = 0; i < n_table; i++) q.board[i] = table_cards[i];
        q.n_board = (uint8_t)n_table;
        q.n_players = (uint8_t)n_alive();
        q.runs = runs;
    }
    int policy(double equity) {
        const int seat = current;
        auto has = [&](int a) { return (legal >> a) & 1u; };
        if (seat_kind[seat] == 0) { /* agents/agent_consider_equity.py:25-56 */
            const double call = min_call_eq[seat], bet = min_bet_eq[seat];
            if (equity > bet + 0.2 && has(ALL_IN)) return ALL_IN;
            if (equity > bet + 0.1 && has(RAISE_2POT)) return RAISE_2POT;
            if (equity > bet && has(RAISE_POT)) return RAISE_POT;
            if (equity > bet - 0.1 && has(RAISE_HALF_POT)) return RAISE_HALF_POT;
            if (equity > call && has(CALL)) return CALL;
            if (has(CHECK)) return CHECK;
            return FOLD;
        }
        /* agents/agent_random.py:21-29: uniform over its move set intersected with the legal moves */
        const uint32_t moves = legal & ((1u << FOLD) | (1u << CHECK) | (1u << CALL) | (1u << RAISE_POT) |
                                        (1u << RAISE_HALF_POT) | (1u << RAISE_2POT));
        uint32_t k = draw(mcq_popc(moves));
        for (int a = 0; a < 10; a++)
            if ((moves >> a) & 1u) {
                if (k == 0) return a;
                k--;
            }
        return FOLD;
    }
    // advance after the pending query was answered; afterwards the next query is pending (or the episode ended)
    void resume(double equity) {
        issued = false;
        if (extra_left > 0) { /* one of the extra calls of calculate_equity: its number only goes into the observation */
            extra_left--;
            return;
        }
        extra_left = -1;
        legal_moves(); /* second half of _get_environment (env.py:272) */
        if (phase == PH_FIRST || phase == PH_B) {
            if (done) { finish_episode(); return; }
            phase = PH_A;
            return;
        }
        /* PH_A: the agent acts on this equity */
        const int action = policy(equity);
        if (!((legal >> action) & 1u)) return; /* illegal move: observe again (env.py:183-184) */
        env_steps++;
        ep_env_steps++;
        process_decision(action);
        go_next_player();
        if (stage == END_HIDDEN || stage == SHOWDOWN) {
            end_hand();
            start_new_hand();
        }
        phase = PH_B;
    }
    void finish_episode() {
        episodes++;
        reset_episode();
    }
};

// Static-partition parallel-for over the tables for large table counts (the tables are independent and each
// has its own generator, so the result does not depend on the thread count).  Workers sleep on a condition
// variable between jobs; a job is a [begin, end) range function.
class Pool {
  public:
    explicit Pool(unsigned n_workers) {
        for (unsigned w = 0; w < n_workers; w++) th_.emplace_back([this, w] { loop(w); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    unsigned parts() const { return (unsigned)th_.size() + 1u; }
    void run(size_t n, const std::function<void(size_t, size_t)> &fn) {
        const unsigned p = parts();
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_ = n;
            left_.store((int)th_.size(), std::memory_order_relaxed);
            gen_++;
        }
        cv_.notify_all();
        fn(0, n / p); /* the caller takes part 0 */
        while (left_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }

  private:
    void loop(unsigned w) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t, size_t)> *fn;
            size_t n;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
                n = n_;
            }
            const unsigned p = parts();
            (*fn)(n * (w + 1) / p, n * (w + 2) / p);
            left_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(size_t, size_t)> *fn_ = nullptr;
    size_t n_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
    std::atomic<int> left_{0};
};

}  // namespace

struct mcq_tables {
    mcq_ctx *ctx;
    mcq_tables_config cfg;
    double min_call[kMaxSeats], min_bet[kMaxSeats];
    uint8_t kind[kMaxSeats];
    std::vector<Table> tables;
    std::vector<mcq_query> q;
    std::vector<mcq_result> r;
    std::vector<double> eq;
    uint64_t calls;
    bool failed = false; /* a lock-step died half way: the tables are out of step with the query ids, nothing more may run */
    Pool *pool = nullptr;
    /* further streams + buffers (and thread pools) of the multi-stream schedule: group g of the tables is stepped by
     * helper thread g while the other groups' batches are on the GPU (mcq_tables_run); [0] unused */
    static constexpr size_t kGroups = 8;
    Pool *pool_g[kGroups] = {};
    mcq_ctx *ctx_g[kGroups] = {};
    ~mcq_tables() {
        delete pool;
        for (size_t g = 1; g < kGroups; g++) {
            delete pool_g[g];
            if (ctx_g[g]) mcq_destroy(ctx_g[g]);
        }
    }
    template <class F>
    void for_tables(F &&f) { /* f(begin, end) */
        if (pool) pool->run(tables.size(), std::function<void(size_t, size_t)>(f));
        else f((size_t)0, tables.size());
    }
};

extern "C" {

int mcq_tables_set_error(const char *msg); /* mcq_host.cpp */
mcq_ctx *mcq_ctx_clone(const mcq_ctx *c);   /* mcq_host.cpp */

/* lock-steps of the tables [a, b) on context ctx: query ids are those of the whole-batch schedule (step * n + table),
 * so the tallies do not depend on how the tables are divided */
static int run_range(mcq_tables *t, mcq_ctx *ctx, Pool *pool, size_t a, size_t b, uint32_t lock_steps, std::string *err) {
    const size_t n = t->tables.size();
    const uint32_t runs = t->cfg.runs;
    for (uint32_t s = 0; s < lock_steps; s++) {
        int rc = mcq_eval_batch(ctx, t->q.data() + a, b - a, t->cfg.seed, t->calls + (uint64_t)s * n + a, MCQ_MODE_PHILOX,
                                t->r.data() + a);
        if (rc) {
            if (err) *err = mcq_last_error();
            return rc;
        }
        const bool more = s + 1 < lock_steps;
        auto step = [&](size_t x, size_t y) { /* answer, act, and issue the next query in one pass */
            for (size_t i = a + x; i < a + y; i++) {
                Table &tb = t->tables[i];
                tb.resume((double)(t->r[i].win + t->r[i].tie) / (double)t->r[i].runs);
                if (more) tb.observe(t->q[i], runs);
            }
        };
        if (pool) pool->run(b - a, std::function<void(size_t, size_t)>(step));
        else step(0, b - a);
    }
    return MCQ_OK;
}

mcq_tables *mcq_tables_create(mcq_ctx *ctx, const mcq_tables_config *cfg) {
    if (!cfg || cfg->n_tables == 0 || cfg->n_seats < 2 || cfg->n_seats > kMaxSeats || cfg->runs == 0 ||
        cfg->max_raises == 0) {
        mcq_tables_set_error("mcq_tables_create: bad configuration (2..10 seats, runs > 0)");
        return nullptr;
    }
    mcq_tables *t = new (std::nothrow) mcq_tables();
    if (!t) { mcq_tables_set_error("mcq_tables_create: out of memory"); return nullptr; }
    try {
        t->ctx = ctx;
        t->cfg = *cfg;
        t->calls = 0;
        for (uint32_t s = 0; s < cfg->n_seats; s++) {
            t->min_call[s] = cfg->min_call_equity[s];
            t->min_bet[s] = cfg->min_bet_equity[s];
            t->kind[s] = cfg->seat_kind[s];
        }
        t->tables.resize(cfg->n_tables);
        t->q.resize(cfg->n_tables);
        t->r.resize(cfg->n_tables);
        t->eq.resize(cfg->n_tables);
        for (uint32_t i = 0; i < cfg->n_tables; i++) {
            Table &tb = t->tables[i];
            memset(&tb, 0, sizeof tb);
            tb.n = (int)cfg->n_seats;
            tb.initial_stacks = cfg->initial_stacks;
            tb.small_blind = cfg->small_blind;
            tb.big_blind = cfg->big_blind;
            tb.max_raises = (int)cfg->max_raises;
            tb.extra_queries = cfg->reserved[2] ? 3 : 0;
            tb.extra_left = -1;
            tb.seat_kind = t->kind;
            tb.min_call_eq = t->min_call;
            tb.min_bet_eq = t->min_bet;
            uint32_t o[4];
            mcq_philox4x32_10(i, 0, 0, 0x54424C31u /* 'TBL1' */, (uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32), o);
            tb.rng.s0 = o[0]; tb.rng.s1 = o[1]; tb.rng.s2 = o[2]; tb.rng.s3 = o[3];
            if ((o[0] | o[1] | o[2] | o[3]) == 0) tb.rng.s0 = 1;
            tb.reset_episode();
        }
        /* threads: cfg.reserved[0], else $MCQ_TABLES_THREADS, else one per 1024 tables up to the core count (<= 16) */
        unsigned nt = cfg->reserved[0];
        if (nt == 0)
            if (const char *e = getenv("MCQ_TABLES_THREADS")) nt = (unsigned)atoi(e);
        if (nt == 0) {
            unsigned hw = std::thread::hardware_concurrency();
            if (hw == 0) hw = 4;
            if (hw > 16) hw = 16;
            nt = cfg->n_tables / 1024u;
            if (nt > hw) nt = hw;
        }
        if (nt > cfg->n_tables) nt = cfg->n_tables;
        if (nt > 1) t->pool = new Pool(nt - 1);
    } catch (...) {
        delete t;
        mcq_tables_set_error("mcq_tables_create: out of memory");
        return nullptr;
    }
    return t;
}

void mcq_tables_destroy(mcq_tables *t) { delete t; }

/* No C++ exception may cross the C ABI (std::system_error from thread creation, std::bad_alloc from std::function or
 * the pools): the bodies below run inside try blocks, as the entries of mcq_host.cpp do. */
size_t mcq_tables_begin(mcq_tables *t, mcq_query *q) {
    if (!t || !q) return 0;
    if (t->failed) { mcq_tables_set_error("mcq_tables_begin: the driver failed in an earlier call; create a new one"); return 0; }
    try {
        const uint32_t runs = t->cfg.runs;
        t->for_tables([&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) t->tables[i].observe(q[i], runs);
        });
    } catch (const std::exception &ex) {
        t->failed = true;
        mcq_tables_set_error((std::string("mcq_tables_begin: ") + ex.what()).c_str());
        return 0;
    } catch (...) {
        t->failed = true;
        mcq_tables_set_error("mcq_tables_begin: unexpected exception");
        return 0;
    }
    return t->tables.size();
}

int mcq_tables_resume(mcq_tables *t, const double *equity) {
    if (!t || !equity) return mcq_tables_set_error("mcq_tables_resume: null argument");
    if (t->failed) return mcq_tables_set_error("mcq_tables_resume: the driver failed in an earlier call; create a new one");
    try {
        t->for_tables([&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) t->tables[i].resume(equity[i]);
        });
    } catch (const std::exception &ex) {
        t->failed = true;
        return mcq_tables_set_error((std::string("mcq_tables_resume: ") + ex.what()).c_str());
    } catch (...) {
        t->failed = true;
        return mcq_tables_set_error("mcq_tables_resume: unexpected exception");
    }
    return MCQ_OK;
}

static int tables_run_impl(mcq_tables *t, uint32_t lock_steps, uint64_t *stats);

int mcq_tables_run(mcq_tables *t, uint32_t lock_steps, uint64_t *stats) {
    if (!t || !t->ctx) return mcq_tables_set_error("mcq_tables_run: needs a context");
    if (t->failed) return mcq_tables_set_error("mcq_tables_run: the driver failed in an earlier call; create a new one");
    try {
        return tables_run_impl(t, lock_steps, stats);
    } catch (const std::exception &ex) {
        t->failed = true;
        return mcq_tables_set_error((std::string("mcq_tables_run: ") + ex.what()).c_str());
    } catch (...) {
        t->failed = true;
        return mcq_tables_set_error("mcq_tables_run: unexpected exception");
    }
}

static int tables_run_impl(mcq_tables *t, uint32_t lock_steps, uint64_t *stats) {
    constexpr size_t kMaxGroups = mcq_tables::kGroups;
    const size_t n = t->tables.size();
    const uint32_t runs = t->cfg.runs;
    if (lock_steps && mcq_tables_begin(t, t->q.data()) == 0) return MCQ_EINVAL;
    /* Groups of tables on streams of their own (three; two below 192 tables; $MCQ_TABLES_GROUPS = 1..8).  While one group's
     * batch is on the GPU the other groups' tables are stepped on the host (each by its own thread, and its own thread
     * pool when the table count is large); per-query ids, hence all results, are as in one batch per step. */
    size_t groups = n >= 192 ? 3 : 2; /* measured (tools/groups_probe.sh, profiles/r03y_table_driver_groups.txt): 512 tables
                                         71 / 54 / 47 / 45 / 45 / 57 us per lock-step with 1 / 2 / 3 / 4 / 6 / 8 groups,
                                         4096 tables 233 / 151 / 111 / 131 / 157 / 192 us */
    if (const char *e = getenv("MCQ_TABLES_GROUPS")) groups = (size_t)atoi(e);
    if (groups > kMaxGroups) groups = kMaxGroups;
    if (n >= 64 && lock_steps >= 4 && t->cfg.reserved[1] == 0 && groups >= 2) {
        bool ready = true;
        for (size_t g = 1; g < groups; g++) {
            if (!t->ctx_g[g]) t->ctx_g[g] = mcq_ctx_clone(t->ctx);
            if (t->ctx_g[g] && t->pool && !t->pool_g[g]) t->pool_g[g] = new (std::nothrow) Pool(t->pool->parts() - 1u);
            ready = ready && t->ctx_g[g] && (!t->pool || t->pool_g[g]);
        }
        if (ready) {
            std::string err[kMaxGroups];
            int rc[kMaxGroups] = {};
            std::vector<std::thread> helpers;
            auto join_all = [&] { for (auto &h : helpers) if (h.joinable()) h.join(); };
            try {
                for (size_t g = 1; g < groups; g++)
                    helpers.emplace_back([&, g] {
                        try {
                            rc[g] = run_range(t, t->ctx_g[g], t->pool_g[g], n * g / groups, n * (g + 1) / groups, lock_steps, &err[g]);
                        } catch (const std::exception &ex) {
                            rc[g] = MCQ_EDEVICE;
                            err[g] = std::string("mcq_tables_run (helper thread): ") + ex.what();
                        } catch (...) {
                            rc[g] = MCQ_EDEVICE;
                            err[g] = "mcq_tables_run (helper thread): unexpected exception";
                        }
                    });
                rc[0] = run_range(t, t->ctx, t->pool, 0, n / groups, lock_steps, nullptr);
            } catch (...) {
                join_all(); /* never leave a helper running (or joinable: std::terminate) behind an exception */
                throw;
            }
            join_all();
            /* a failed group leaves its tables at the step it reached while the others went on: the groups are out of
             * step with each other and with the query ids -- the driver is marked failed and refuses further calls */
            for (size_t g = 0; g < groups; g++)
                if (rc[g]) t->failed = true;
            if (rc[0]) return rc[0];
            for (size_t g = 1; g < groups; g++)
                if (rc[g]) { mcq_tables_set_error(err[g].c_str()); return rc[g]; }
            t->calls += (uint64_t)lock_steps * n;
            if (stats) mcq_tables_stats(t, stats);
            return MCQ_OK;
        }
    }
    for (uint32_t s = 0; s < lock_steps; s++) {
        int rc = mcq_eval_batch(t->ctx, t->q.data(), n, t->cfg.seed, t->calls, MCQ_MODE_PHILOX, t->r.data());
        if (rc) return rc; /* the queries stay pending: the tables were not advanced, the call may be repeated */
        t->calls += n;
        const bool more = s + 1 < lock_steps;
        t->for_tables([&](size_t a, size_t b) { /* answer, act, and issue the next query in one pass */
            for (size_t i = a; i < b; i++) {
                Table &tb = t->tables[i];
                tb.resume((double)(t->r[i].win + t->r[i].tie) / (double)t->r[i].runs);
                if (more) tb.observe(t->q[i], runs);
            }
        });
    }
    if (stats) mcq_tables_stats(t, stats);
    return MCQ_OK;
}

void mcq_tables_stats(const mcq_tables *t, uint64_t *stats) {
    uint64_t env = 0, qs = 0, ep = 0;
    for (const Table &tb : t->tables) { env += tb.env_steps; qs += tb.queries; ep += tb.episodes; }
    stats[0] = env;
    stats[1] = qs;
    stats[2] = ep;
}

int mcq_tables_state(const mcq_tables *t, uint32_t table, double *stacks, int32_t *info) {
    if (!t || table >= t->tables.size()) return mcq_tables_set_error("mcq_tables_state: bad table index");
    const Table &tb = t->tables[table];
    for (int i = 0; i < tb.n; i++) stacks[i] = tb.stacks[i];
    info[0] = tb.stage;
    info[1] = tb.current;
    info[2] = tb.winner_ix;
    info[3] = (int32_t)tb.episodes;
    info[4] = (int32_t)tb.env_steps;
    info[5] = (int32_t)tb.queries;
    info[6] = (int32_t)tb.legal;
    info[7] = tb.phase;
    return MCQ_OK;
}

}  // extern "C"
