/*
 * mcq.h -- C ABI of libmcq_hip.so: the MI355X (gfx950) Monte-Carlo poker-equity engine.
 *
 * This is the drop-in boundary for ONE path of jaronlong52/neuron_poker: the equity query
 *   tools/montecarlo_python.py:401-406  get_equity(player_cards, table_cards, players, runs) -> float
 *   tools/montecarlo_python.py:191-252  MonteCarlo.run_montecarlo(...) -> (equity, winTypesDict)
 * which gym_env/env.py:75-81 selects once and calls at gym_env/env.py:249-262.  The reference has no C ABI
 * of its own (its native variant is a pybind11 module, tools/montecarlo_cpp/pymontecarlo.cpp:21-23, with the
 * same four-argument call); the functions below are what a Python binding for this path binds instead --
 * neuron_poker_amd/montecarlo_hip.py does so through ctypes, and INTEGRATION.md shows the stub.
 *
 * Plain C: pointers and sizes only, no C++ or torch types.  All functions return MCQ_OK (0) or a negative
 * MCQ_E* code and never throw or abort; mcq_last_error() returns a thread-local description of the last
 * failure.  Card id c = 4*rank + suit with rank = index in "23456789TJQKA", suit = index in "CDHS"
 * (tools/hand_evaluator.py:5-6; the deck order of tools/montecarlo_python.py:114-119 is ascending c).
 */
#ifndef MCQ_H
#define MCQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__) || defined(__clang__)
#define MCQ_API __attribute__((visibility("default")))
#else
#define MCQ_API
#endif

#define MCQ_VERSION_MAJOR 0
#define MCQ_VERSION_MINOR 3
#define MCQ_VERSION_PATCH 0

/* error codes */
#define MCQ_OK 0
#define MCQ_EINVAL (-1)  /* bad argument: card out of range, duplicate card, n_players not in [1,10], ... */
#define MCQ_EDEVICE (-2) /* HIP runtime failure; mcq_last_error() carries the HIP error string */
#define MCQ_ENOMEM (-3)  /* host or device allocation failed */
#define MCQ_EBUSY (-4)   /* another call is running on this context (or mcq_multi object): one call in flight per context */

/* random-number front ends (the deal -> evaluate -> tally body is shared) */
#define MCQ_MODE_PHILOX 0         /* production: counter-based streams, Philox4x32-10 keyed MWC64X (MCQ-CTR v5) */
#define MCQ_MODE_REPLAY_MT19937 1 /* parity: query i replays np.random.seed((seed + first_query_id + i) mod 2^32)
                                     exactly as tools/montecarlo_python.py consumes it -> bit-exact tallies */

/* dealing law of MCQ_MODE_PHILOX (mcq_set_dealing_law); the parity mode always follows the reference */
#define MCQ_LAW_REFERENCE 0 /* default: tools/montecarlo_python.py's law incl. its index bias (:170, :188) */
#define MCQ_LAW_UNIFORM 1   /* opt-in: every remaining card equally likely -- what tools/montecarlo_cython.pyx:188
                               and tools/montecarlo_cpp/Montecarlo.cpp:296-312 deal */

/* One equity query (16 bytes, no pointers).  Mirrors the arguments of get_equity
 * (tools/montecarlo_python.py:401): hero's two cards, 0/3/4/5 known table cards, players, runs. */
typedef struct mcq_query {
    uint8_t hole[2];     /* hero's cards */
    uint8_t board[5];    /* known table cards, first n_board entries used */
    uint8_t n_board;     /* 0..5 */
    uint8_t n_players;   /* hero + opponents, 1..10 (gym_env/env.py:249 passes sum(alive)) */
    uint8_t reserved[3]; /* must be 0 */
    uint32_t runs;       /* iterations (maxRuns); exactly this many are executed, no wall-clock cut-off */
} mcq_query;

/* Per-query tallies (104 bytes, unsigned integers only).
 * Reference's wins (montecarlo_python.py:223-229, ties go to hero: hand_evaluator.py:23) = win + tie
 * = sum(by_type); equity = (win + tie) / runs (montecarlo_python.py:243).
 * by_type order = HighCard, Pair, TwoPair, ThreeOfAKind, Straight, Flush, FullHouse, FoufOfAKind [sic],
 * StraightFlush (hand_evaluator.py:92-115): hero's hand type in the iterations he wins.
 * passes = opponent-deal attempts (montecarlo_python.py:168). */
typedef struct mcq_result {
    uint64_t runs;
    uint64_t passes;
    uint64_t win; /* hero strictly best */
    uint64_t tie; /* hero best together with at least one opponent (credited to hero by the reference) */
    uint64_t by_type[9];
} mcq_result;

/* Optional extension of a query (304 bytes) for the rest of run_montecarlo's arguments:
 * ghost_cards (tools/montecarlo_python.py:206-208), any number of further known hands (collusion players, :133-163),
 * the hero or any known hand given as a SET of preflop classes instead of two cards (:136-148), and opponents
 * restricted to a range (:165-181 with :36-112).  A range is a 169-bit set of classes; the bit of a class is how
 * get_two_short_notation (:24-34) names two cards: suited -> 13*min+max, off-suit -> 13*max+min, pair -> 14*rank
 * (rank = index in "23456789TJQKA").  "Every class" = all 169 bits set.
 * The known hands are dealt in the order of original_player_card_list: hero, known[0], known[1], ...; a hand
 * given as a range is drawn from the deck as it is at that point (:136-148) -- it may take a card that a LATER hand
 * names, which then simply is not in the deck any more (the reference's try/except, :154-161). */
#define MCQ_MAX_KNOWN 9
typedef struct mcq_known_hand {
    uint8_t cards[2];   /* the hand, when is_range == 0 */
    uint8_t is_range;   /* 1: drawn from `range` every iteration, cards ignored */
    uint8_t reserved;   /* must be 0 */
    uint32_t range[6];
} mcq_known_hand;       /* 28 bytes */

typedef struct mcq_query_ext {
    uint8_t ghost[2];      /* two cards taken out of the deck, 0xFF 0xFF = none */
    uint8_t hero_is_range; /* 1: mcq_query.hole is ignored, hero's hand is drawn from hero_range every iteration */
    uint8_t n_known;       /* further known hands after the hero, 0..MCQ_MAX_KNOWN; they count in n_players */
    uint32_t opp_range[6];
    uint32_t hero_range[6];
    mcq_known_hand known[MCQ_MAX_KNOWN];
} mcq_query_ext;           /* 304 bytes */

typedef struct mcq_ctx mcq_ctx;

/* Number of HIP devices visible, or a negative MCQ_E* code. */
MCQ_API int mcq_device_count(void);

/* Create an engine bound to HIP device `device` (one context per GPU / per process rank).  Owns its stream,
 * device and pinned staging buffers and lookup tables until mcq_destroy.  Not re-entrant: one call in flight
 * per context -- create one per thread (they are cheap: a stream, a table image, staging buffers that grow on demand).
 * The library checks it: an entry point called while another call is running on the same context returns MCQ_EBUSY
 * and touches nothing.  Returns NULL on failure (see mcq_last_error). flags must be 0. */
MCQ_API mcq_ctx *mcq_create(int device, int flags);
MCQ_API void mcq_destroy(mcq_ctx *ctx);

/* Evaluate n queries held in HOST memory; blocks until out[0..n) is written.  The caller owns q and out.
 * Query i uses query id first_query_id + i, so a batch split into shards (other ranks, other calls) with
 * the matching first_query_id gives bit-identical per-query tallies.  All queries are validated first;
 * on MCQ_EINVAL nothing is launched and out is untouched.
 * MCQ_MODE_PHILOX: batches of small queries (at most 8192 iterations each -- the reference asks for 1000,
 * gym_env/env.py:22) cost ONE kernel launch: the kernel takes the records from its arguments (up to eight queries) or
 * from pinned host memory and stores the finished rows to pinned host memory, no copy, prep or zeroing launches around
 * it, and the call returns when a flag the kernel raises there is seen (about 19 us for one 1000-run query); larger
 * queries are priced on the host and sliced over the whole GPU.  MCQ_MODE_REPLAY_MT19937: numpy's MT19937 stream of every
 * query is walked on the GPU -- a pair of waves per query for batches; for a call of at most 64 long queries the 624-word
 * state blocks of a query are generated in segments side by side (start states by jump-ahead) and parsed side by side
 * (one 100 000-run query: 0.16-0.33 ms; environment switches MCQ_MT_BLOCKS, MCQ_MT_JUMP: INTEGRATION.md). */
MCQ_API int mcq_eval_batch(mcq_ctx *ctx, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id, int mode,
                   mcq_result *out);

/* n == 1 convenience: exactly get_equity's arguments after card-string conversion. */
MCQ_API int mcq_eval_one(mcq_ctx *ctx, const mcq_query *q, uint64_t seed, int mode, mcq_result *out);

/* mcq_eval_batch with one mcq_query_ext per query (host buffers).  A range that cannot be dealt from the cards
 * left (the reference would loop forever) gives MCQ_EINVAL after a bounded number of attempts.
 * MCQ_MODE_PHILOX deals the reference's law without its re-draw loops: per range, a list of the ordered card pairs
 * the range allows is laid out once per query, a trial picks one of them with one random word and is accepted iff
 * both cards are still in the deck (and the second is not the deck's highest card, which the reference's index
 * range excludes) -- `passes` counts these trials, not the reference's.  Streams (MCQ-CTR v5x): as mcq_eval_batch,
 * sixteen iterations each, but two for a query that draws from a list and has at most 8192 iterations.
 * Up to eight queries of at most 8192 iterations (six candidate lists) per call take ONE kernel launch: the call
 * pattern of the reference's agents, one ranged query per decision -- 25 us per call. */
MCQ_API int mcq_eval_batch_ext(mcq_ctx *ctx, const mcq_query *q, const mcq_query_ext *ext, size_t n, uint64_t seed,
                       uint64_t first_query_id, int mode, mcq_result *out);

/* Parity mode coupled to numpy's GLOBAL random state, as consecutive reference calls are: the n
 * queries consume ONE MT19937 stream in order.  mt_key[624] / *mt_pos are numpy's state words and position
 * (np.random.get_state()[1], [2]); on return they hold the state after the last query, so
 * np.random.set_state(...) leaves numpy exactly where the reference's own calls would have left it. */
MCQ_API int mcq_eval_batch_numpy_stream(mcq_ctx *ctx, const mcq_query *q, size_t n, uint32_t *mt_key, uint32_t *mt_pos,
                                mcq_result *out);

/* Same computation with queries and results RESIDENT IN HBM: d_queries -> mcq_query[n], d_results ->
 * mcq_result[n] (overwritten), both device pointers on this context's device; hip_stream is the hipStream_t
 * to launch on (NULL = HIP's null stream, as everywhere in HIP).  Asynchronous: returns after enqueueing,
 * no host synchronisation once the context's scratch is large enough (first call / larger n allocate).
 * MCQ_MODE_PHILOX only.  Invalid queries cannot be rejected up front here: their result has runs = 0 and
 * passes = UINT64_MAX. */
MCQ_API int mcq_eval_batch_device(mcq_ctx *ctx, const void *d_queries, size_t n, uint64_t seed, uint64_t first_query_id,
                          void *d_results, void *hip_stream);

/* mcq_eval_batch_device for SMALL queries -- at most 8192 iterations each, the reference's own call pattern (1000 runs,
 * gym_env/env.py:22) -- in ONE kernel launch: no pricing kernel in front, no atomics; every query is owned by a few
 * waves of one block and its finished row is stored once.  Same arguments, same tallies (the RNG streams are keyed by
 * query id and iteration, not by the schedule).  A query with more than 8192 iterations is NOT evaluated here: like an
 * invalid one it gets runs = 0, passes = UINT64_MAX.  Asynchronous on hip_stream; can be captured into a HIP graph after
 * one ordinary call on the context. */
MCQ_API int mcq_eval_batch_device_small(mcq_ctx *ctx, const void *d_queries, size_t n, uint64_t seed, uint64_t first_query_id,
                                        void *d_results, void *hip_stream);

/* Showdown with the same device evaluator (tools/hand_evaluator.py:9-24 get_winner / eval_best_hand):
 * hands = n_tables x n_players x 7 card ids (host); winner[t] = index of the best hand (first of equals),
 * winner_type[t] = its by_type index, keys (optional, n_tables x n_players) = the 32-bit ranking keys: a
 * greater key is a stronger hand, equal keys tie; MCQ_KEY_TYPE(key) is the hand's by_type index. */
#define MCQ_KEY_TYPE(key) (((key) >> 28) - (((key) >> 28) >= 6u ? 1u : 0u))
MCQ_API int mcq_showdown(mcq_ctx *ctx, const uint8_t *hands, size_t n_tables, int n_players, uint8_t *winner,
                 uint8_t *winner_type, uint32_t *keys);

/* One share of a batch whose ITERATIONS are split over several devices (few large queries).  Every query
 * is cut at task boundaries (1024 iterations) into n_parts contiguous ranges; this call evaluates range `part` of
 * every query under the same streams (seed, first_query_id + i) an unsplit call uses, so the rows of all parts add
 * up -- runs included -- to exactly what mcq_eval_batch returns.  MCQ_MODE_PHILOX only. */
MCQ_API int mcq_eval_batch_part(mcq_ctx *ctx, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id,
                                uint32_t part, uint32_t n_parts, mcq_result *out);

/* Exact equity by exhaustive enumeration (what tools/montecarlo_cpp/Test.cpp:176-217 approximates with
 * 1 % bands): every opponent hand and every completion of the table, weighted as the dealing law `law` (MCQ_LAW_*)
 * deals them -- for MCQ_LAW_REFERENCE the exact distribution of tools/montecarlo_python.py:121-189, index bias
 * included.  n_players 1..3; q[i].runs is ignored.  out[i]: runs = total weight, win / tie / by_type = weight of
 * the outcomes (integers; equity = (win + tie) / runs exactly), passes = 0.  Preflop heads-up = 2.1e9 hand
 * evaluations, three players preflop = 1.2e12 pair comparisons. */
MCQ_API int mcq_exact_batch(mcq_ctx *ctx, const mcq_query *q, size_t n, int law, mcq_result *out);

/* Select the dealing law used by MCQ_MODE_PHILOX on this context (MCQ_LAW_*). */
MCQ_API int mcq_set_dealing_law(mcq_ctx *ctx, int law);

/* Kernel timing (off by default: a timestamped launch costs a small query about 6 us of its call time).  When on,
 * every evaluation-kernel launch carries a pair of HIP events that take the kernel's own begin and end timestamps
 * on the stream it is launched on (a ring of the 64 most recent launches; in parity mode the pair spans the stream
 * walk and the evaluation kernel; launches recorded into a stream capture are not timed).  mcq_kernel_times writes
 * the durations in milliseconds of the latest min(max_n, 64, launches so far) launches, oldest first, and returns how
 * many; it waits for the newest of them to have finished.  mcq_last_kernel_ms: the most recent host-entry call's
 * total (all chunks), else the latest launch; 0 while timing is off.  The contexts of an mcq_multi always time. */
MCQ_API int mcq_set_kernel_timing(mcq_ctx *ctx, int on);
MCQ_API int mcq_kernel_times(mcq_ctx *ctx, float *ms, int max_n);
MCQ_API float mcq_last_kernel_ms(mcq_ctx *ctx);

/* ---- Lock-step table driver (BASELINE configs[4]; replaces the loop gym_env/env.py:170-200 + 224-262 runs per
 * table: observe -> get_equity -> agent -> step).  T tables of n_seats seats play No-Limit Hold'em with the
 * reference's table rules (gym_env/env.py, gym_env/cycle.py); every table always has exactly one pending equity
 * query, finished episodes restart at once.  mcq_tables_begin writes the n_tables pending queries (table i ->
 * q[i]) and returns n_tables; mcq_tables_resume takes their equities ((win + tie) / runs) and advances every
 * table to its next query.  mcq_tables_run does `lock_steps` rounds of begin -> ONE mcq_eval_batch
 * (MCQ_MODE_PHILOX, seed cfg.seed, query ids counting up across calls) -> resume, and needs a context; it
 * runs the tables in two or three groups on streams (and host threads) of their own so that the host steps one
 * group while the other groups' batches are on the GPU -- every query keeps the id it has in the one-batch schedule,
 * so the results are the same.
 * begin/resume alone need no GPU (ctx may be NULL): that is how the CPU tests pin the rules.
 * seat_kind: 0 = equity agent (agents/agent_consider_equity.py:25-56 with min_call_equity / min_bet_equity of
 * the seat), 1 = random agent (agents/agent_random.py:21-29, drawing from the table's own generator).
 * Dealing and random seats draw from one xoshiro128++ per table, seeded by Philox4x32-10(counter = {table, 0, 0,
 * 'TBL1'}, key = seed); a bounded draw is mulhi32(next(), bound). */
typedef struct mcq_tables mcq_tables;
typedef struct mcq_tables_config {
    uint32_t n_tables, n_seats;     /* 2..10 seats */
    uint32_t runs;                  /* iterations per equity query (the reference uses 1000, env.py:261) */
    uint32_t max_raises;            /* per seat and street (env.py:92: 2) */
    double initial_stacks, small_blind, big_blind;
    uint64_t seed;
    uint8_t seat_kind[10];
    uint8_t reserved[6];            /* [0]: host threads stepping the tables (0 = automatic); [1]: 1 = do not split
                                     * the tables into groups on streams of their own (mcq_tables_run); [2]: 1 = three more
                                     * equity queries per observation, answers unused, as HoldemTable(calculate_equity=
                                     * True) issues them (gym_env/env.py:248-256); rest 0 */
    double min_call_equity[10], min_bet_equity[10];
} mcq_tables_config;

MCQ_API mcq_tables *mcq_tables_create(mcq_ctx *ctx, const mcq_tables_config *cfg); /* NULL + mcq_last_error */
MCQ_API void mcq_tables_destroy(mcq_tables *t);
MCQ_API size_t mcq_tables_begin(mcq_tables *t, mcq_query *q);
MCQ_API int mcq_tables_resume(mcq_tables *t, const double *equity);
MCQ_API int mcq_tables_run(mcq_tables *t, uint32_t lock_steps, uint64_t stats[3]);
/* stats: agent actions executed (env steps), equity queries issued, episodes finished -- totals since create */
MCQ_API void mcq_tables_stats(const mcq_tables *t, uint64_t stats[3]);
/* one table: stacks[n_seats]; info[8] = stage, seat to act, last winner, episodes, env steps, queries,
 * legal-move bit mask (bit = gym_env/enums.py Action value), driver phase */
MCQ_API int mcq_tables_state(const mcq_tables *t, uint32_t table, double *stacks, int32_t info[8]);

/* ---- One node, several GPUs, ONE process (the reference itself is single-device, so this has no
 * counterpart there -- it is what a batched caller of get_equity binds when a node has more than one GPU).
 * The batch is partitioned over n_shards SHARDS, shard s on HIP device devices[s] (devices = NULL: shard s on
 * device s); a device may appear more than once (an 8-way partition rehearsed on one GPU, or several streams per
 * GPU).  Each shard owns an engine context, a host worker thread and a stream.  Partition of a call:
 *   MCQ_PARTITION_QUERIES     shard s evaluates queries [n*s/k, n*(s+1)/k) under their own query ids
 *   MCQ_PARTITION_ITERATIONS  every shard evaluates share s of k of EVERY query's iterations (few large queries)
 *   MCQ_PARTITION_AUTO        queries when n >= 256 * n_shards, else iterations
 * Every shard fills its part of a zero-initialised [n, 13] uint64 tally matrix on its device; shards that share a
 * device are added there, and ONE ncclAllReduce(ncclUint64, ncclSum) over the distinct devices (communicators
 * from ncclCommInitAll, owned by the mcq_multi object; RCCL over xGMI) leaves the complete matrix on every device;
 * out[0..n) is copied from the first.  The tallies are bit-identical to mcq_eval_batch on one context with the same
 * (seed, first_query_id), whatever the partition.  MCQ_MODE_PHILOX.  RCCL is bound (dlopen) when the first
 * mcq_multi is created; a process that never creates one never maps it.
 * STATUS: with more than one DISTINCT device this entry has NOT been exercised on hardware yet (no multi-GPU node was
 * reachable while it was built): the partitions, the same-device add and a one-rank RCCL communicator are tested on
 * one GPU, where all shards share the device; ncclCommInitAll over several devices, the grouped all-reduce and the
 * per-device stream waits run for the first time on a multi-GPU node. */
#define MCQ_PARTITION_AUTO 0
#define MCQ_PARTITION_QUERIES 1
#define MCQ_PARTITION_ITERATIONS 2
typedef struct mcq_multi mcq_multi;
MCQ_API mcq_multi *mcq_multi_create(const int *devices, int n_shards, int flags); /* NULL + mcq_last_error; flags 0 */
MCQ_API void mcq_multi_destroy(mcq_multi *m);
MCQ_API int mcq_multi_eval_batch(mcq_multi *m, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id,
                                 int partition, mcq_result *out);
/* The same with queries and results RESIDENT IN HBM (no PCIe in the call): one device pointer pair per shard, on that
 * shard's device.  d_queries[s]: MCQ_PARTITION_QUERIES -> the shard's block, queries [n*s/k, n*(s+1)/k) of the batch
 * (k shards); MCQ_PARTITION_ITERATIONS -> all n queries.  d_results[s] -> mcq_result[n], overwritten: after the
 * all-reduce EVERY shard's buffer holds the complete matrix.  Blocks until all devices have finished.  The host never
 * sees the queries, so they are validated on the device: an invalid query's row has runs = 0 and passes = UINT64_MAX
 * (under either partition: only one share of a query writes the marker, so the sum over the shares keeps it). */
MCQ_API int mcq_multi_eval_batch_device(mcq_multi *m, const void *const *d_queries, size_t n, uint64_t seed,
                                        uint64_t first_query_id, int partition, void *const *d_results);
MCQ_API int mcq_multi_set_dealing_law(mcq_multi *m, int law);
/* info: shards, distinct devices (= ranks of the all-reduce), RCCL version code, partition of the last call */
MCQ_API int mcq_multi_info(const mcq_multi *m, int info[4]);
/* last call, milliseconds: slowest shard's evaluation kernel, the all-reduce (device 0's stream), whole call (wall) */
MCQ_API int mcq_multi_times(const mcq_multi *m, float ms[3]);

MCQ_API const char *mcq_last_error(void);
MCQ_API void mcq_version(int *major, int *minor, int *patch);

#ifdef __cplusplus
}
#endif
#endif /* MCQ_H */
