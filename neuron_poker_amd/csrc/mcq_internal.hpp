// mcq_internal.hpp -- launch functions shared between mcq_kernels.hip and mcq_host.cpp (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcq.h"

struct McqTables;

#define MCQ_INTERNAL_MODE_UNIFORM 2 /* MCQ_MODE_PHILOX with the context's dealing law set to MCQ_LAW_UNIFORM */

hipError_t mcq_launch_prep(const mcq_query *d_q, uint32_t n, mcq_result *d_res, uint64_t *d_prefix, uint32_t part,
                           uint32_t n_parts, uint32_t n_cu, uint32_t split_max, hipStream_t s);
hipError_t mcq_launch_eval(int mode, const mcq_query *d_q, uint32_t n, const uint64_t *d_prefix, mcq_result *d_res,
                           uint64_t seed, uint64_t first_qid, const McqTables *d_luts, const uint8_t *d_draws,
                           const uint64_t *d_draw_off, uint32_t grid, uint32_t block, uint32_t split, uint32_t part,
                           uint32_t n_parts, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr, uint32_t work_wpb = 0);
/* exact enumeration (n_players <= 3), any number of queries per launch: one job per query (blockIdx.y), all of the same
 * kind -- two opponents or fewer; every job adds into its zeroed row d_rows[job.row] */
struct McqExactJob {
    uint32_t rec[4];                        /* the 16-byte query record */
    uint32_t n_boards, slices, row, grid;   /* table completions, cuts of the first-opponent loop, result row, blocks that work */
};
void mcq_exact_plan(const mcq_query *q, uint32_t row, uint32_t n_cu, McqExactJob *job);
hipError_t mcq_launch_exact(const McqExactJob *d_jobs, uint32_t n_jobs, uint32_t max_grid, bool two_opp, int law,
                            mcq_result *d_rows, const McqTables *d_luts, hipStream_t s);
/* hands / winner / wtype / keys: device-visible memory (pinned host memory or HBM), 16-byte aligned and padded to whole
 * tiles of 256 tables; *bad is set when a hand is not seven distinct ids < 52; ticket != 0: the last block raises
 * *done_flag behind a system-scope release (d_done: a zeroed device word) */
hipError_t mcq_launch_showdown(const uint8_t *hands, uint32_t n_tables, uint32_t n_players, const McqTables *d_luts,
                               uint8_t *winner, uint8_t *wtype, uint32_t *keys, uint32_t *bad, uint32_t *d_done,
                               uint32_t *done_flag, uint32_t ticket, uint32_t n_cu, hipStream_t s);
/* a few extended queries in one launch (mcq_eval_ext_small_kernel): the queries and their extension records travel in
 * the kernel arguments */
#define MCQ_EXT_SMALL_Q 8u      /* queries per launch (8 x 320 B of kernel arguments) */
#define MCQ_EXT_SMALL_LISTS 6u  /* candidate lists per query that fit the block's LDS (6 x 2704 x 2 B) */
#define MCQ_EXT_SMALL_TASKS 64u /* wave tasks per query */
#define MCQ_EXT_SMALL_BLOCKS 32u /* blocks per launch: a query is cut into parts of 4, 8 or 16 wave tasks, whatever fits */
struct McqExtSmallKarg {
    uint32_t q[MCQ_EXT_SMALL_Q][4];
    uint32_t ext[MCQ_EXT_SMALL_Q][76];
    uint32_t blk[MCQ_EXT_SMALL_BLOCKS]; /* block b: query | part << 8 | parts << 16 | waves that take tasks << 24; its row goes to row b of the result buffer */
};
hipError_t mcq_launch_eval_ext_small(const McqExtSmallKarg *karg, uint32_t n_blocks, mcq_result *h_res_dev, uint64_t seed,
                                     uint64_t first_qid, const McqTables *d_luts, uint32_t *d_done, uint32_t *done_flag,
                                     uint32_t ticket, hipStream_t s, hipEvent_t t0, hipEvent_t t1);
hipError_t mcq_launch_prep_ext(const mcq_query *d_q, const mcq_query_ext *d_ext, uint32_t n, int mode, mcq_result *d_res,
                               uint64_t *d_prefix, hipStream_t s);
/* production mode of the extended queries: lays out the candidate lists (lists_stride per query, MCQ_EXT_LIST_STRIDE
 * uint16 entries each) and their lengths */
hipError_t mcq_launch_ext_lists(const mcq_query *d_q, const mcq_query_ext *d_ext, uint32_t n, uint32_t lists_stride,
                                uint16_t *d_lists, uint32_t *d_cnts, hipStream_t s);
hipError_t mcq_launch_eval_ext(int mode, const mcq_query *d_q, const mcq_query_ext *d_ext, uint32_t n,
                               const uint64_t *d_prefix, mcq_result *d_res, uint64_t seed, uint64_t first_qid,
                               const McqTables *d_luts, const uint8_t *d_draws, const uint64_t *d_draw_off,
                               const uint16_t *d_lists, const uint32_t *d_cnts, uint32_t lists_stride, uint32_t grid,
                               uint32_t block, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
/* host-buffer calls with few rows: d_rows[0..n_rows) -> pinned host memory (device address h_rows_dev), d_rows zeroed,
 * then *done_flag = ticket; n_rows even (buffers hold the odd row's neighbour), d_done a zeroed device word */
hipError_t mcq_launch_publish(mcq_result *d_rows, mcq_result *h_rows_dev, uint64_t n_rows, uint32_t *d_done,
                              uint32_t *done_flag, uint32_t ticket, hipStream_t s);
/* dst[i] += src[i], i < n (tally matrices of two shards on one device, both 16-byte aligned) */
hipError_t mcq_launch_add_u64(uint64_t *d_dst, const uint64_t *d_src, uint64_t n, hipStream_t s);
/* parity mode: one wave per query parses np.random.seed(seed32 + i)'s MT19937 stream into d_draws (+ passes into the
 * result rows); d_counter must be zero (the prep kernel leaves one behind the cost prefix) */
hipError_t mcq_launch_mt_parse(const mcq_query *d_q, uint32_t n, uint32_t seed32, uint8_t *d_draws, const uint64_t *d_draw_off,
                               mcq_result *d_res, uint32_t *d_counter, uint32_t n_cu, hipStream_t s);
/* ... for few long queries (mcq_mt_blocks.hpp): the state blocks of a query side by side.  d_blk_off[q] .. d_blk_off[q + 1]:
 * the query's blocks in d_raw (624 state words each), d_exits (MCQ_MTB_LANES words each), d_entries (8 B each); max_blocks = the most
 * blocks of one query; d_grp_off likewise for the query's groups of MCQ_MTB_GROUP blocks in d_gword / d_gits (MCQ_MTB_LANES
 * words each) and d_gentry (8 B each); d_ovf[n].  A query whose stream does not end within its blocks gets passes = UINT64_MAX.  The
 * rows' passes must be zero (the prep kernel). */
hipError_t mcq_launch_mt_blocks(const mcq_query *d_q, uint32_t n, uint32_t seed32, const uint32_t *d_blk_off,
                                const uint32_t *d_grp_off, uint32_t max_blocks, uint32_t *d_raw, uint32_t *d_exits, void *d_entries,
                                uint32_t *d_gword, uint32_t *d_gits, void *d_gentry, uint32_t *d_ovf, uint8_t *d_draws,
                                const uint64_t *d_draw_off, mcq_result *d_res,
                                uint32_t *d_part /* n x mcq_mtb_part_words() words: the segments' start states by jump-ahead;
                                                    null: one work-group per query makes all its blocks */,
                                hipStream_t s);
uint64_t mcq_mtb_part_words(void); /* per query */
/* ... and for extended queries (mcq_mt_ext.hpp); d_counter zero (mcq_prep_ext_kernel leaves one behind its prefix); a
 * query whose range cannot be dealt gets passes = UINT64_MAX */
hipError_t mcq_launch_mt_parse_ext(const mcq_query *d_q, const mcq_query_ext *d_ext, uint32_t n, uint32_t seed32, uint8_t *d_draws,
                                   const uint64_t *d_draw_off, mcq_result *d_res, uint32_t *d_counter, uint32_t n_cu, hipStream_t s);
/* small queries, one launch and nothing else: work_rec / work_qi (the work laid out wave by wave, see the kernel), res
 * and done_flag may be pinned host memory (device-visible); d_done: a zeroed device word */
#define MCQ_DIRECT_IDLE 0xFFFFFFFFu
#define MCQ_DIRECT_KARG_SLOTS 128u /* one-launch path: the work of a launch this small travels in the kernel arguments */
struct McqDirectKarg {
    uint32_t rec[MCQ_DIRECT_KARG_SLOTS][4]; /* per wave slot: the query record (16 B) */
    uint32_t qi[MCQ_DIRECT_KARG_SLOTS];     /* per wave slot: the query index or MCQ_DIRECT_IDLE */
};
#define MCQ_DIRECT_TASKS_LIMIT 32u /* most tasks of a query on the one-launch path: its row sums add two counters per word */
hipError_t mcq_launch_eval_direct(int mode, const void *work_rec, const uint32_t *work_qi, uint32_t rounds, uint32_t merge,
                                  mcq_result *res, uint64_t seed, uint64_t first_qid, const McqTables *d_luts, uint32_t grid,
                                  uint32_t *d_done, uint32_t *done_flag, uint32_t ticket, hipStream_t s, hipEvent_t t0,
                                  hipEvent_t t1, const McqDirectKarg *karg /* or null: read work_rec / work_qi */,
                                  uint32_t dev_n = 0 /* != 0: work_rec is the caller's mcq_query[dev_n] in HBM, work_qi null: */,
                                  uint32_t dev_lg = 0 /* every query 2^dev_lg waves; validated on the device */);
