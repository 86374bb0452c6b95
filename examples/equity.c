/* The C ABI of include/mcq.h from plain C99: Monte-Carlo and exact equity of one hand, and a batch of tables.
 *
 *   gcc -std=c99 -Iinclude examples/equity.c -o equity neuron_poker_amd/libmcq_hip.so -Wl,-rpath,$PWD/neuron_poker_amd
 *   ./equity            # needs an AMD GPU; prints the error of mcq_last_error() otherwise
 *
 * Cards are ids 4 * rank + suit, rank = index in "23456789TJQKA", suit = index in "CDHS" (tools/hand_evaluator.py:5-6).
 */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "mcq.h"

/* the record layouts the Python binding (neuron_poker_amd/_lib.py) and the kernels rely on */
typedef char query_is_16_bytes[sizeof(mcq_query) == 16 ? 1 : -1];
typedef char result_is_104_bytes[sizeof(mcq_result) == 104 ? 1 : -1];
typedef char ext_is_304_bytes[sizeof(mcq_query_ext) == 304 && sizeof(mcq_known_hand) == 28 ? 1 : -1];
typedef char config_is_224_bytes[sizeof(mcq_tables_config) == 224 ? 1 : -1];
typedef char runs_at_12[offsetof(mcq_query, runs) == 12 ? 1 : -1];

static int card(const char *s) {
    const char *ranks = "23456789TJQKA", *suits = "CDHS";
    return 4 * (int)(strchr(ranks, s[0]) - ranks) + (int)(strchr(suits, s[1]) - suits);
}

int main(int argc, char **argv) {
    mcq_ctx *ctx;
    mcq_query q;
    mcq_result r;
    int rc;
    if (argc > 1 && strcmp(argv[1], "--layout") == 0) { /* used by tests/test_abi.py: no GPU needed */
        printf("%d %d %d %d %d %d %d\n", (int)sizeof(mcq_query), (int)sizeof(mcq_result), (int)sizeof(mcq_query_ext),
               (int)sizeof(mcq_tables_config), (int)offsetof(mcq_tables_config, seed),
               (int)offsetof(mcq_tables_config, seat_kind), (int)offsetof(mcq_tables_config, min_call_equity));
        return 0;
    }
    ctx = mcq_create(0, 0);
    if (!ctx) {
        fprintf(stderr, "mcq_create: %s\n", mcq_last_error());
        return 1;
    }
    memset(&q, 0, sizeof q);
    q.hole[0] = (uint8_t)card("AH");
    q.hole[1] = (uint8_t)card("KH");
    q.n_board = 0;
    q.n_players = 2;
    q.runs = 100000;
    rc = mcq_eval_batch(ctx, &q, 1, /*seed*/ 1, /*first_query_id*/ 0, MCQ_MODE_PHILOX, &r);
    if (rc) {
        fprintf(stderr, "mcq_eval_batch: %d %s\n", rc, mcq_last_error());
        return 1;
    }
    printf("AhKh heads-up, %llu iterations: equity %.4f (kernel %.1f us)\n", (unsigned long long)r.runs,
           (double)(r.win + r.tie) / (double)r.runs, 1e3 * mcq_last_kernel_ms(ctx));
    rc = mcq_exact_batch(ctx, &q, 1, MCQ_LAW_REFERENCE, &r);
    if (rc) {
        fprintf(stderr, "mcq_exact_batch: %d %s\n", rc, mcq_last_error());
        return 1;
    }
    printf("exact (the reference's dealing law): %.6f = %llu / %llu\n", (double)(r.win + r.tie) / (double)r.runs,
           (unsigned long long)(r.win + r.tie), (unsigned long long)r.runs);
    {
        mcq_tables_config cfg;
        mcq_tables *t;
        uint64_t st[3];
        const double call[4] = {.5, .8, .7, .2}, bet[4] = {-.5, -.8, -.7, -.3}; /* main.py:142-145 */
        int i;
        memset(&cfg, 0, sizeof cfg);
        cfg.n_tables = 512;
        cfg.n_seats = 6;
        cfg.runs = 1000;
        cfg.max_raises = 2;
        cfg.initial_stacks = 100;
        cfg.small_blind = 1;
        cfg.big_blind = 2;
        cfg.seed = 7;
        for (i = 0; i < 4; i++) {
            cfg.min_call_equity[i] = call[i];
            cfg.min_bet_equity[i] = bet[i];
        }
        cfg.seat_kind[4] = cfg.seat_kind[5] = 1; /* two random seats */
        t = mcq_tables_create(ctx, &cfg);
        if (!t || mcq_tables_run(t, 1000, st)) {
            fprintf(stderr, "mcq_tables: %s\n", mcq_last_error());
            return 1;
        }
        printf("512 tables, 1000 lock-steps: %llu agent actions, %llu equity queries, %llu episodes\n",
               (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2]);
        mcq_tables_destroy(t);
    }
    mcq_destroy(ctx);
    return 0;
}
