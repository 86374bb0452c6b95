// mcq_host.cpp -- the C ABI of include/mcq.h on top of the gfx950 kernels.
//
// Host responsibilities only: argument validation, staging of the 16-byte query / 104-byte result records,
// the MT19937 walk of the parity mode (mcq_replay.hpp) and kernel launches.  There is no CPU evaluation path:
// without a HIP device mcq_create fails and every other entry point needs a context.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mcq_ctx.hpp"
#include "mcq_device.hpp"
#include "mcq_internal.hpp"
#include "mcq_mt_blocks.hpp"
#include "mcq_replay.hpp"

hipError_t mcq_eval_occupancy(int mode, int block, int *blocks_per_cu); /* mcq_kernels.hip */

namespace {

thread_local std::string g_err;

constexpr size_t kReplayChunkBytes = 256u << 20;  /* draw bytes staged per launch by the host walk (numpy-stream entry) */
constexpr uint32_t kBlock = 1024;                /* threads per block of the evaluation kernels */

}  // namespace

int mcq_fail(int code, const char *what, const char *detail) {
    g_err = what;
    if (detail) {
        g_err += ": ";
        g_err += detail;
    }
    return code;
}

namespace {

/* grid/block for a launch whose total task count is known (host entry) or unknown (0) */
void pick_geometry(const mcq_ctx *c, int mode, uint64_t total_tasks, uint32_t *grid, uint32_t *block,
                   uint32_t *split = nullptr, uint64_t max_tasks = 0, uint32_t *work_wpb = nullptr) {
    const uint32_t full = (uint32_t)c->n_cu * (uint32_t)c->occ[mode];
    if (split) *split = 0;
    if (work_wpb) *work_wpb = 0;
    if (total_tasks == 0) { *block = kBlock; *grid = full; return; }
    if (split) { /* small batches are cut finer, see mcq_pick_split */
        *split = mcq_pick_split(total_tasks, max_tasks ? max_tasks : total_tasks, (uint32_t)c->n_cu, c->split_max);
        total_tasks <<= *split;
    }
    /* one block per CU (the LDS tables allow no more); few tasks are spread over all CUs with fewer waves per
     * block: 98 tasks of a single 100k query -> 98 one-wave blocks, 1024 tasks -> 256 four-wave blocks */
    uint64_t wpb = (total_tasks + (uint64_t)c->n_cu - 1) / (uint64_t)c->n_cu;
    if (wpb < 1) wpb = 1;
    if (wpb > kBlock / 64) wpb = kBlock / 64;
    const uint64_t blocks = (total_tasks + wpb - 1) / wpb;
    *grid = (uint32_t)(blocks < (uint64_t)c->n_cu ? blocks : (wpb == kBlock / 64 ? full : (uint64_t)c->n_cu));
    /* the block brings 97 KB of tables into LDS before anything else happens: more waves than take work shorten
     * that (six 16-byte loads per lane with 16 waves, one round trip, instead of 24 with four) */
    uint64_t launch = wpb < c->load_waves ? c->load_waves : wpb;
    if (work_wpb && launch != wpb) *work_wpb = (uint32_t)wpb;
    else launch = wpb < 4 ? 4 : wpb, wpb = launch;
    *block = (uint32_t)(64 * launch);
}

/* is stream s being captured into a graph? (a failing query counts as "no") */
bool stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

/* The scheduling scratch of stream s, at least `bytes` large.  A new stream takes a free slot, else the least
 * recently used one (after making s wait for that slot's last reader).  Growing is not possible while s is being
 * captured (hipMalloc / hipFree inside a capture would break it): the call fails cleanly instead. */
int scratch_for(mcq_ctx *c, hipStream_t s, size_t bytes, bool capturing, mcq_ctx::Scratch **out) {
    mcq_ctx::Scratch *slot = nullptr;
    for (auto &sc : c->scratch)
        if (sc.used && sc.stream == s) { slot = &sc; break; }
    if (!slot) {
        for (int i = 1; i < mcq_ctx::kScratch && !slot; i++)
            if (!c->scratch[i].used) slot = &c->scratch[i];
        if (!slot) {
            for (int i = 1; i < mcq_ctx::kScratch; i++) {
                mcq_ctx::Scratch &sc = c->scratch[i];
                if (sc.pinned) continue;
                if (!slot || sc.last_use < slot->last_use) slot = &sc;
            }
            if (!slot) return mcq_fail(MCQ_EINVAL, "mcq: every scheduling scratch slot is pinned by a captured graph");
            if (capturing) {
                if (hipEventQuery(slot->done) != hipSuccess) {
                    (void)hipGetLastError();
                    return mcq_fail(MCQ_EINVAL, "mcq: no free scheduling scratch for a stream that is being captured");
                }
            } else {
                HIP_TRY(hipStreamWaitEvent(s, slot->done, 0));
            }
        }
        slot->stream = s;
        slot->used = true;
    }
    if (bytes > slot->prefix.cap) {
        if (capturing)
            return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_device: the context's scratch would have to grow inside a stream "
                                        "capture; issue one call of at least this size on this stream before capturing");
        if (slot->done_recorded) HIP_TRY(hipEventSynchronize(slot->done)); /* its last reader may still be running */
        HIP_TRY(slot->prefix.reserve(bytes));
    }
    slot->last_use = ++c->scratch_clock;
    if (capturing) slot->pinned = true; /* the graph keeps reading this buffer whenever it is replayed */
    *out = slot;
    return MCQ_OK;
}

}  // namespace

int mcq_run_slice(mcq_ctx *c, int mode, const mcq_query *d_q, uint32_t n, mcq_result *d_res, uint64_t seed,
                  uint64_t first_qid, uint64_t total_tasks, const uint8_t *d_draws, const uint64_t *d_off, hipStream_t s,
                  bool timed, uint64_t max_tasks, uint32_t part, uint32_t n_parts, const uint32_t *mt_seed32,
                  const uint64_t *d_prefix_ready, const McqMtbLaunch *mtb) {
    if (mode == MCQ_MODE_PHILOX && c->law == MCQ_LAW_UNIFORM) mode = MCQ_INTERNAL_MODE_UNIFORM;
    const bool capturing = stream_capturing(s);
    if (capturing || !c->timing) timed = false; /* events recorded inside a capture cannot be read back */
    mcq_ctx::Scratch *sc = nullptr;
    const uint64_t *d_prefix = d_prefix_ready;
    if (!d_prefix) {
        int rc = scratch_for(c, s, ((size_t)n + 3) * sizeof(uint64_t), capturing, &sc);
        if (rc) return rc;
        d_prefix = (const uint64_t *)sc->prefix.p;
        HIP_TRY(mcq_launch_prep(d_q, n, d_res, (uint64_t *)sc->prefix.p, part, n_parts, (uint32_t)c->n_cu, c->split_max, s));
    }
    uint32_t grid, block, split, work_wpb;
    pick_geometry(c, mode, total_tasks, &grid, &block, &split, max_tasks, &work_wpb);
    if (mode == MCQ_MODE_REPLAY_MT19937 && split > 2) split = 2; /* the parity kernel's lanes take four iterations at a time */
    /* queries in HBM (the host has not seen them): up to 1024 of them may be a small batch -- the prep kernel
     * decides the cut and the evaluation kernel reads it; more queries are at least as many tasks: never cut */
    if (total_tasks == 0 && n <= 1024u) split = MCQ_SPLIT_FROM_PREP;
    /* timing: the events take the evaluation kernel's own begin / end timestamps (no marker packets around a short
     * kernel); in parity mode the timed region starts in front of the stream walk */
    const int slot = (int)(c->n_timed % mcq_ctx::kRing);
    hipEvent_t t0 = timed ? c->ev0[slot] : nullptr, t1 = timed ? c->ev1[slot] : nullptr;
    if (mt_seed32) {
        if (timed) HIP_TRY(hipEventRecord(t0, s));
        t0 = nullptr;
        if (mtb)
            HIP_TRY(mcq_launch_mt_blocks(d_q, n, *mt_seed32, mtb->d_blk_off, mtb->d_grp_off, mtb->max_blocks, mtb->d_raw, mtb->d_exits,
                                         mtb->d_entries, mtb->d_gword, mtb->d_gits, mtb->d_gentry, mtb->d_ovf,
                                         const_cast<uint8_t *>(d_draws), d_off, d_res, mtb->d_part, s));
        else
            HIP_TRY(mcq_launch_mt_parse(d_q, n, *mt_seed32, const_cast<uint8_t *>(d_draws), d_off, d_res,
                                        const_cast<uint32_t *>(reinterpret_cast<const uint32_t *>(d_prefix + n + 2)),
                                        (uint32_t)c->n_cu, s));
    }
    HIP_TRY(mcq_launch_eval(mode, d_q, n, d_prefix, d_res, seed, first_qid, c->d_luts, d_draws, d_off, grid, block, split,
                            part, n_parts, s, t0, t1, work_wpb));
    if (timed) c->n_timed++;
    /* a caller's stream may run on while another stream's call takes over the scratch: mark its last reader.  The
     * context's own stream (slot 0) is synchronised by every host entry before it returns. */
    if (sc && !capturing && sc != &c->scratch[0]) {
        HIP_TRY(hipEventRecord(sc->done, s));
        sc->done_recorded = true;
    }
    return MCQ_OK;
}

uint32_t mcq_tasks_of(const mcq_query &q) { return q.runs / MCQ_TASK_ITERS + (q.runs % MCQ_TASK_ITERS != 0u ? 1u : 0u); }

int mcq_validate_queries(const mcq_query *q, size_t n) {
    for (size_t i = 0; i < n; i++)
        if (!mcq_query_valid(mcq_query_words(q[i]))) {
            char buf[160];
            snprintf(buf, sizeof buf,
                     "query %zu invalid (cards must be distinct ids < 52, n_board <= 5, 1 <= n_players <= 10)", i);
            return mcq_fail(MCQ_EINVAL, buf);
        }
    return MCQ_OK;
}

namespace {

inline uint32_t tasks_of(const mcq_query &q) { return mcq_tasks_of(q); }
inline int validate(const mcq_query *q, size_t n) { return mcq_validate_queries(q, n); }
}  // namespace

static int kernel_times_impl(mcq_ctx *c, float *ms, int max_n);
/* the completion flag of the calls that hand their rows over in pinned memory (defined with the small-batch paths below) */
static int flag_ready(mcq_ctx *c);
static uint32_t next_ticket(mcq_ctx *c);
static int wait_ticket(mcq_ctx *c, uint32_t ticket, bool *by_flag, double est_us = 0.0);

namespace {

/* parity mode, independent streams: query i replays np.random.seed((seed + first_qid + i) mod 2^32).  The stream walk
 * runs on the DEVICE (mcq_mt_parse_kernel, one wave per query) and fills the draw buffer the evaluation kernel
 * reads; the host only lays the buffer out.  Chunks of queries whose draws fit c->replay_device_bytes. */
/* Few long queries (at most kMtbQueries in a chunk, at least kMtbMinBlocks state blocks among them) are parsed with
 * their state blocks side by side (mcq_mt_blocks.hpp): one 100 000-run query 6.5 -> 1.x ms.  The blocks a query needs are
 * an estimate with a margin; a query whose stream runs past them comes back with passes = UINT64_MAX and the call is
 * repeated with the serial walk. */
constexpr size_t kMtbQueries = 64;
constexpr uint64_t kMtbMinBlocks = 64, kMtbMaxBlocks = 1u << 18; /* (2.7 KB of scratch per block) */
int replay_batch_device(mcq_ctx *c, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id, mcq_result *out,
                        bool allow_blocks = true) {
    HIP_TRY(c->h_off.reserve(n * sizeof(uint64_t)));
    HIP_TRY(c->d_off.reserve(n * sizeof(uint64_t)));
    uint64_t *off = (uint64_t *)c->h_off.p;
    struct Chunk { size_t a, b; uint64_t bytes, tasks, max_tasks; };
    std::vector<Chunk> chunks;
    size_t a = 0;
    uint64_t largest = 0;
    while (a < n) {
        Chunk ch = {a, a, 0, 0, 0};
        while (ch.b < n && ch.b - ch.a < 0x7fffffffu) {
            const uint64_t stride = ((uint64_t)q[ch.b].runs + 63u) & ~63ull;
            const uint64_t need = stride * mcq_draws_per_iteration(q[ch.b]);
            if (ch.b > ch.a && ch.bytes + need > c->replay_device_bytes) break;
            off[ch.b] = ch.bytes;
            ch.bytes += need;
            const uint64_t t = tasks_of(q[ch.b]);
            ch.tasks += t;
            if (t > ch.max_tasks) ch.max_tasks = t;
            ch.b++;
        }
        if (ch.bytes > largest) largest = ch.bytes;
        chunks.push_back(ch);
        a = ch.b;
    }
    HIP_TRY(c->d_draws.reserve(largest + 64));
    HIP_TRY(hipMemcpyAsync(c->d_off.p, off, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    const uint64_t timed0 = c->n_timed;
    bool by_blocks = false;
    for (const Chunk &ch : chunks) { /* stream order keeps a chunk's parse behind the previous chunk's evaluation */
        const uint32_t seed32 = (uint32_t)(seed + first_query_id + ch.a);
        const size_t m = ch.b - ch.a;
        McqMtbLaunch mtb = {};
        if (allow_blocks && c->mt_blocks && m <= kMtbQueries && chunks.size() == 1) {
            HIP_TRY(c->h_misc.reserve(2 * (m + 1) * sizeof(uint32_t)));
            uint32_t *blk = (uint32_t *)c->h_misc.p, *grp = blk + m + 1; /* block / group offsets of the queries */
            uint64_t total = 0, groups = 0, jumps = 0;
            for (size_t i = 0; i < m; i++) {
                const mcq_query &qq = q[ch.a + i];
                blk[i] = (uint32_t)total;
                grp[i] = (uint32_t)groups;
                const uint32_t n_opp = qq.n_players - 1u, n_deal = 5u - qq.n_board;
                if (qq.n_players >= 1 && qq.n_players <= 10 && qq.n_board <= 5 && 2u * n_opp + n_deal != 0u && qq.runs != 0u) {
                    const uint32_t nb = mcq_mtb_blocks_needed(50u - qq.n_board, n_opp, n_deal, qq.runs, c->mt_blocks_margin);
                    total += nb;
                    groups += (nb + MCQ_MTB_GROUP - 1u) / MCQ_MTB_GROUP;
                    if (nb > mtb.max_blocks) mtb.max_blocks = nb;
                    jumps += (nb - 1u) / MCQ_MTB_SEG;
                }
            }
            blk[m] = (uint32_t)total;
            grp[m] = (uint32_t)groups;
            const auto pad16 = [](uint64_t x) { return (x + 15u) & ~15ull; };
            /* segments side by side pay when the jumps (6.2 M word-XORs each: eight work-groups for 15 us, i.e. 0.4 us of
             * the whole GPU) and the three launches per round of 4096 blocks cost less than the longest query's blocks one
             * behind the other (0.19 us each) -- measured: ONE 6-max 100 000-run query 0.99 -> 0.29 ms; 64 queries of 780
             * blocks, all jumping: 1.18 -> 1.34 ms with the first jump kernel (49 us per 30 jumps) */
            const uint64_t rounds = (mtb.max_blocks + MCQ_MTB_SEG * MCQ_MTB_MAX_SEG - 1u) / (MCQ_MTB_SEG * MCQ_MTB_MAX_SEG);
            const bool use_jump = c->mt_jump && mtb.max_blocks > MCQ_MTB_SEG &&
                                  (c->mt_jump_always || 45.0 * (double)rounds + 0.4 * (double)jumps < 0.19 * (double)mtb.max_blocks);
            const uint64_t o_ovf = pad16(2 * (m + 1) * 4), o_ent = o_ovf + pad16(m * 4),
                           o_gen = o_ent + pad16(total * sizeof(McqMtbEntry)), o_gw = o_gen + pad16(groups * sizeof(McqMtbEntry)),
                           o_gi = o_gw + groups * MCQ_MTB_LANES * 4u, o_ex = o_gi + groups * MCQ_MTB_LANES * 4u,
                           o_raw = o_ex + total * MCQ_MTB_LANES * 4u, o_part = pad16(o_raw + total * MCQ_MT_N * 4u),
                           /* (the segments' start states by jump-ahead: only a query of more than one segment has any) */
                           part_bytes = use_jump ? m * mcq_mtb_part_words() * 4u : 0u,
                           bytes = o_part + part_bytes + 64u;
            bool room = total >= kMtbMinBlocks && total <= kMtbMaxBlocks;
            if (room && c->d_mt.reserve(bytes) != hipSuccess) { /* no room for the blocks' scratch (2.7 KB per block): the
                                                                  * serial walk, which needs none, serves the call */
                (void)hipGetLastError();
                room = false;
            }
            if (room) {
                char *base = (char *)c->d_mt.p;
                HIP_TRY(hipMemcpyAsync(base, blk, 2 * (m + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
                mtb.d_blk_off = (const uint32_t *)base;
                mtb.d_grp_off = mtb.d_blk_off + m + 1;
                mtb.d_ovf = (uint32_t *)(base + o_ovf);
                mtb.d_entries = base + o_ent;
                mtb.d_gentry = base + o_gen;
                mtb.d_gword = (uint32_t *)(base + o_gw);
                mtb.d_gits = (uint32_t *)(base + o_gi);
                mtb.d_exits = (uint32_t *)(base + o_ex);
                mtb.d_raw = (uint32_t *)(base + o_raw);
                mtb.d_part = use_jump ? (uint32_t *)(base + o_part) : nullptr;
                by_blocks = true;
            }
        }
        int rc = mcq_run_slice(c, MCQ_MODE_REPLAY_MT19937, (const mcq_query *)c->d_q.p + ch.a, (uint32_t)m,
                               (mcq_result *)c->d_res.p + ch.a, seed, first_query_id + ch.a, ch.tasks ? ch.tasks : 1,
                               (const uint8_t *)c->d_draws.p, (const uint64_t *)c->d_off.p + ch.a, c->stream, true,
                               ch.max_tasks, 0, 1, &seed32, nullptr, mtb.d_blk_off ? &mtb : nullptr);
        if (rc) return rc;
    }
    if (by_blocks && n <= c->publish_max_rows) {
        /* few long queries are a call of a quarter of a millisecond: the rows come back through pinned memory behind the
         * completion flag, as the small production calls' do (no copy, no stream synchronisation: 24 us against 7.5) */
        int rc = flag_ready(c);
        if (rc) return rc;
        const uint32_t ticket = next_ticket(c);
        HIP_TRY(mcq_launch_publish((mcq_result *)c->d_res.p, (mcq_result *)c->h_res.dev, n + (n & 1u), (uint32_t *)c->d_done.p,
                                   (uint32_t *)c->h_flag.dev, ticket, c->stream));
        rc = wait_ticket(c, ticket, nullptr, 0.0);
        if (rc) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(c->h_res.p, c->d_res.p, n * sizeof(mcq_result), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (by_blocks) { /* a stream that ran past its estimated blocks (never seen; the margin is eight blocks): the serial walk */
        const mcq_result *hr = (const mcq_result *)c->h_res.p;
        for (size_t i = 0; i < n; i++)
            if (hr[i].passes == ~0ull) return replay_batch_device(c, q, n, seed, first_query_id, out, false);
    }
    memcpy(out, c->h_res.p, n * sizeof(mcq_result));
    float total = 0.f;
    const int launched = (int)(c->n_timed - timed0);
    if (launched > 0 && launched <= mcq_ctx::kRing) {
        float ms[mcq_ctx::kRing];
        if (kernel_times_impl(c, ms, launched) == launched)
            for (int i = 0; i < launched; i++) total += ms[i];
    }
    c->last_ms = total;
    return MCQ_OK;
}

/* parity mode coupled to ONE MT19937 stream that all queries continue in order, as consecutive reference calls
 * share numpy's global state (mcq_eval_batch_numpy_stream): a serial walk by construction, done on the host
 * (mcq_replay.hpp); chunks of queries whose draw bytes fit the staging budget. */
int replay_batch(mcq_ctx *c, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id, McqMt19937 *stream,
                 mcq_result *out) {
    if (!stream) return replay_batch_device(c, q, n, seed, first_query_id, out);
    std::vector<uint64_t> passes(n, 0);
    float replay_ms = 0.f;
    size_t a = 0;
    while (a < n) {
        size_t b = a;
        uint64_t bytes = 0, tasks = 0, max_tasks = 0;
        HIP_TRY(c->h_off.reserve((n - a < 65536 ? n - a : 65536) * sizeof(uint64_t)));
        uint64_t *off = (uint64_t *)c->h_off.p;
        while (b < n && b - a < 65536) {
            uint64_t stride = ((uint64_t)q[b].runs + 63u) & ~63ull;
            uint64_t need = stride * mcq_draws_per_iteration(q[b]);
            if (b > a && bytes + need > kReplayChunkBytes) break;
            off[b - a] = bytes;
            bytes += need;
            tasks += tasks_of(q[b]);
            if (tasks_of(q[b]) > max_tasks) max_tasks = tasks_of(q[b]);
            b++;
        }
        const size_t m = b - a;
        HIP_TRY(c->h_draws.reserve(bytes + 64));
        HIP_TRY(c->d_draws.reserve(bytes + 64));
        HIP_TRY(c->d_off.reserve(m * sizeof(uint64_t)));
        uint8_t *hd = (uint8_t *)c->h_draws.p;
        for (size_t i = 0; i < m; i++) {
            const mcq_query &qq = q[a + i];
            uint64_t stride = ((uint64_t)qq.runs + 63u) & ~63ull;
            passes[a + i] = mcq_replay_parse_stream(qq, *stream, hd + off[i], stride);
        }
        HIP_TRY(hipMemcpyAsync(c->d_draws.p, hd, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_off.p, off, m * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        int rc = mcq_run_slice(c, MCQ_MODE_REPLAY_MT19937, (const mcq_query *)c->d_q.p + a, (uint32_t)m,
                               (mcq_result *)c->d_res.p + a, seed, first_query_id + a, tasks, (const uint8_t *)c->d_draws.p,
                               (const uint64_t *)c->d_off.p, c->stream, true, max_tasks);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream)); /* staging buffers are reused by the next chunk */
        float ms = 0.f;
        if (c->timing && kernel_times_impl(c, &ms, 1) == 1) replay_ms += ms;
        a = b;
    }
    HIP_TRY(hipMemcpyAsync(c->h_res.p, c->d_res.p, n * sizeof(mcq_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_res.p, n * sizeof(mcq_result));
    for (size_t i = 0; i < n; i++) out[i].passes = passes[i];
    c->last_ms = replay_ms;
    return MCQ_OK;
}

/* validation + query upload shared by the host entry points (the caller has selected the context's device) */
int stage_queries(mcq_ctx *c, const mcq_query *q, size_t n, mcq_result *out, const char *who) {
    if (!c) return mcq_fail(MCQ_EINVAL, who, "null context");
    if (!q || !out) return mcq_fail(MCQ_EINVAL, who, "null buffer");
    if (n > 0x7fffffffu) return mcq_fail(MCQ_EINVAL, who, "n too large");
    int rc = validate(q, n);
    if (rc) return rc;
    uint64_t total_tasks = 0;
    for (size_t i = 0; i < n; i++) total_tasks += tasks_of(q[i]);
    if (total_tasks > 0xfffffff0ull) return mcq_fail(MCQ_EINVAL, who, "too many iterations in one call");
    HIP_TRY(c->h_q.reserve(n * sizeof(mcq_query)));
    HIP_TRY(c->h_res.reserve((n + (n & 1u)) * sizeof(mcq_result))); /* (whole 16-byte words: mcq_publish_kernel) */
    HIP_TRY(c->d_q.reserve(n * sizeof(mcq_query)));
    HIP_TRY(c->d_res.reserve((n + (n & 1u)) * sizeof(mcq_result)));
    c->res_clean = 0; /* these entries zero their rows in the prep kernel and leave them filled */
    memcpy(c->h_q.p, q, n * sizeof(mcq_query));
    HIP_TRY(hipMemcpyAsync(c->d_q.p, c->h_q.p, n * sizeof(mcq_query), hipMemcpyHostToDevice, c->stream));
    c->last_ms = 0.f;
    return MCQ_OK;
}

}  // namespace

extern "C" {

const char *mcq_last_error(void) { return g_err.c_str(); }

int mcq_tables_set_error(const char *msg) { return mcq_fail(MCQ_EINVAL, msg); } /* for mcq_tables.cpp; not exported */

/* a second context like c (same device, dealing law and tuning) with its own stream and buffers; not exported */
mcq_ctx *mcq_ctx_clone(const mcq_ctx *c) {
    mcq_ctx *d = mcq_create(c->device, 0);
    if (d) {
        d->law = c->law;
        d->split_max = c->split_max;
        d->load_waves = c->load_waves;
        d->direct_max_tasks = c->direct_max_tasks;
        d->direct_poll = c->direct_poll;
        d->direct_sleep = c->direct_sleep;
        d->direct_uniform_min = c->direct_uniform_min;
        d->ext_small = c->ext_small;
        d->mt_blocks = c->mt_blocks;
        d->mt_blocks_margin = c->mt_blocks_margin;
        d->mt_jump = c->mt_jump;
        d->mt_jump_always = c->mt_jump_always;
        d->publish_max_rows = c->publish_max_rows;
        d->timing = c->timing;
        d->replay_device_bytes = c->replay_device_bytes;
    }
    return d;
}

void mcq_version(int *major, int *minor, int *patch) {
    if (major) *major = MCQ_VERSION_MAJOR;
    if (minor) *minor = MCQ_VERSION_MINOR;
    if (patch) *patch = MCQ_VERSION_PATCH;
}

int mcq_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return mcq_fail(MCQ_EDEVICE, "hipGetDeviceCount", hipGetErrorString(e));
    return n;
}

void mcq_destroy(mcq_ctx *c) {
    if (!c) return;
    McqDeviceScope dev_(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf *db[] = {&c->d_q, &c->d_res, &c->d_draws, &c->d_off, &c->d_ext,
                    &c->d_mt, &c->d_lists, &c->d_cnts};
    for (auto &sc : c->scratch) {
        sc.prefix.release();
        if (sc.done) (void)hipEventDestroy(sc.done);
    }
    for (DevBuf *b : db) b->release();
    PinBuf *pb[] = {&c->h_q, &c->h_res, &c->h_draws, &c->h_off, &c->h_misc, &c->h_flag};
    c->d_done.release();
    c->d_done_dev.release();
    for (PinBuf *b : pb) b->release();
    if (c->d_luts) (void)hipFree(c->d_luts);
    for (int i = 0; i < mcq_ctx::kRing; i++) {
        if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]);
        if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

mcq_ctx *mcq_create(int device, int flags) {
    if (flags != 0) { mcq_fail(MCQ_EINVAL, "mcq_create: flags must be 0"); return nullptr; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        mcq_fail(MCQ_EDEVICE, "mcq_create: no HIP device available (this library has no CPU path)",
             e != hipSuccess ? hipGetErrorString(e) : nullptr);
        return nullptr;
    }
    if (device < 0 || device >= n) { mcq_fail(MCQ_EINVAL, "mcq_create: device ordinal out of range"); return nullptr; }
    mcq_ctx *c = new (std::nothrow) mcq_ctx();
    if (!c) { mcq_fail(MCQ_ENOMEM, "mcq_create: out of host memory"); return nullptr; }
    c->device = device;
    hipDeviceProp_t prop;
    McqTables *tabs = new (std::nothrow) McqTables();
    if (!tabs) { mcq_fail(MCQ_ENOMEM, "mcq_create: out of host memory"); delete c; return nullptr; }
    mcq_fill_tables(tabs);
#define CREATE_TRY(expr)                                                      \
    do {                                                                      \
        hipError_t e2_ = (expr);                                              \
        if (e2_ != hipSuccess) {                                              \
            mcq_fail(MCQ_EDEVICE, #expr, hipGetErrorString(e2_));                 \
            mcq_destroy(c);                                                   \
            delete tabs;                                                      \
            if (prev_dev >= 0) (void)hipSetDevice(prev_dev);                  \
            return nullptr;                                                   \
        }                                                                     \
    } while (0)
    int prev_dev = -1; /* the caller's current device is left as it was */
    if (hipGetDevice(&prev_dev) != hipSuccess) prev_dev = -1;
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *e = getenv("MCQ_SPLIT_MAX")) { /* tuning knob, see pick_geometry */
        int v = atoi(e);
        c->split_max = (uint32_t)(v < 0 ? 0 : (v > 4 ? 4 : v));
    }
    if (const char *e = getenv("MCQ_DIRECT_MAX_TASKS")) { /* tuning knob, see eval_host_philox */
        const int v = atoi(e);
        c->direct_max_tasks = (uint32_t)(v < 0 ? 0 : v > (int)MCQ_DIRECT_TASKS_LIMIT ? (int)MCQ_DIRECT_TASKS_LIMIT : v);
    }
    if (const char *e = getenv("MCQ_DIRECT_POLL")) c->direct_poll = atoi(e) != 0;
    if (const char *e = getenv("MCQ_DIRECT_SLEEP")) c->direct_sleep = atoi(e) != 0; /* see wait_ticket */
    if (const char *e = getenv("MCQ_EXT_SMALL")) c->ext_small = atoi(e) != 0; /* see mcq_eval_batch_ext */
    if (const char *e = getenv("MCQ_MT_BLOCKS")) c->mt_blocks = atoi(e) != 0;  /* see replay_batch_device */
    if (const char *e = getenv("MCQ_MT_BLOCKS_MARGIN")) c->mt_blocks_margin = atoi(e);
    if (const char *e = getenv("MCQ_MT_JUMP")) {
        c->mt_jump = atoi(e) != 0;
        c->mt_jump_always = atoi(e) == 2; /* (tests: every query of more than one segment, whatever the estimate says) */
    }
    if (const char *e = getenv("MCQ_DIRECT_UNIFORM_MIN")) { /* tuning knob, see eval_host_philox */
        const long v = atol(e);
        c->direct_uniform_min = (size_t)(v < 0 ? 0 : v);
    }
    if (const char *e = getenv("MCQ_PUBLISH_MAX_ROWS")) { /* tuning knob, see eval_host_philox */
        const long v = atol(e);
        c->publish_max_rows = (size_t)(v < 0 ? 0 : v);
    }
    if (const char *e = getenv("MCQ_LOAD_WAVES")) { /* tuning knob, see pick_geometry */
        const int v = atoi(e);
        c->load_waves = (uint32_t)(v < 1 ? 1 : (v > 16 ? 16 : v));
    }
    if (const char *e = getenv("MCQ_REPLAY_DEVICE_BYTES")) { /* chunking of the parity mode's draw buffer (tests) */
        const long long v = atoll(e);
        if (v > 0) c->replay_device_bytes = (uint64_t)v;
    }
    CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto &sc : c->scratch) CREATE_TRY(hipEventCreateWithFlags(&sc.done, hipEventDisableTiming));
    c->scratch[0].stream = c->stream; /* slot 0: the context's own stream (host entries) */
    c->scratch[0].used = true;
    for (int i = 0; i < mcq_ctx::kRing; i++) {
        CREATE_TRY(hipEventCreate(&c->ev0[i]));
        CREATE_TRY(hipEventCreate(&c->ev1[i]));
    }
    CREATE_TRY(hipMalloc((void **)&c->d_luts, sizeof(McqTables)));
    CREATE_TRY(hipMemcpy(c->d_luts, tabs, sizeof(McqTables), hipMemcpyHostToDevice));
    for (int mode = 0; mode < 3; mode++) {
        int occ = 0;
        CREATE_TRY(mcq_eval_occupancy(mode, (int)kBlock, &occ));
        c->occ[mode] = occ < 1 ? 1 : (occ > 8 ? 8 : occ);
    }
#undef CREATE_TRY
    delete tabs;
    if (prev_dev >= 0 && prev_dev != device) (void)hipSetDevice(prev_dev);
    return c;
}

int mcq_kernel_times(mcq_ctx *c, float *ms, int max_n) {
    if (!c || !ms || max_n < 0) return mcq_fail(MCQ_EINVAL, "mcq_kernel_times: bad argument");
    MCQ_ENTER(c, "mcq_kernel_times");
    return kernel_times_impl(c, ms, max_n);
}

}  // extern "C"

static int kernel_times_impl(mcq_ctx *c, float *ms, int max_n) {
    uint64_t have = c->n_timed < (uint64_t)mcq_ctx::kRing ? c->n_timed : (uint64_t)mcq_ctx::kRing;
    int n = (int)(have < (uint64_t)max_n ? have : (uint64_t)max_n);
    McqDeviceScope dev_(c->device);
    HIP_TRY(dev_.err);
    if (n > 0) /* the one-launch path returns when the rows are out, a moment before the kernel has retired */
        HIP_TRY(hipEventSynchronize(c->ev1[(int)((c->n_timed - 1u) % mcq_ctx::kRing)]));
    for (int i = 0; i < n; i++) {
        int slot = (int)((c->n_timed - (uint64_t)n + (uint64_t)i) % mcq_ctx::kRing);
        HIP_TRY(hipEventElapsedTime(&ms[i], c->ev0[slot], c->ev1[slot]));
    }
    return n;
}

extern "C" {

int mcq_set_kernel_timing(mcq_ctx *c, int on) {
    if (!c) return mcq_fail(MCQ_EINVAL, "mcq_set_kernel_timing: null context");
    MCQ_ENTER(c, "mcq_set_kernel_timing");
    c->timing = on != 0;
    return MCQ_OK;
}

int mcq_set_dealing_law(mcq_ctx *c, int law) {
    if (!c || (law != MCQ_LAW_REFERENCE && law != MCQ_LAW_UNIFORM)) return mcq_fail(MCQ_EINVAL, "mcq_set_dealing_law: bad argument");
    MCQ_ENTER(c, "mcq_set_dealing_law");
    c->law = law;
    return MCQ_OK;
}

float mcq_last_kernel_ms(mcq_ctx *c) {
    float ms = 0.f;
    if (!c) return 0.f;
    McqBusyScope busy_(&c->busy);
    if (!busy_.ok) return 0.f; /* a call is running on the context: no time to report yet */
    if (c->last_ms > 0.f) return c->last_ms;
    if (!c->timing) return 0.f;
    return kernel_times_impl(c, &ms, 1) == 1 ? ms : 0.f;
}

int mcq_eval_batch_device(mcq_ctx *c, const void *d_queries, size_t n, uint64_t seed, uint64_t first_query_id,
                          void *d_results, void *hip_stream) {
    ABI_GUARD_BEGIN
    if (!c) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_device: null context");
    if (n == 0) return MCQ_OK;
    if (!d_queries || !d_results) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_device: null buffer");
    if (n > 0x7fffffffu) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_device: n too large");
    MCQ_ENTER(c, "mcq_eval_batch_device");
    McqDeviceScope dev_(c->device);
    HIP_TRY(dev_.err);
    hipStream_t s = (hipStream_t)hip_stream; /* NULL = the HIP null stream */
    /* the prefix buffer must already be large enough when the call is being captured into a graph */
    c->last_ms = 0.f;
    return mcq_run_slice(c, MCQ_MODE_PHILOX, (const mcq_query *)d_queries, (uint32_t)n, (mcq_result *)d_results, seed,
                     first_query_id, 0, nullptr, nullptr, s, true);
    ABI_GUARD_END("mcq_eval_batch_device")
}

/* Small queries resident in HBM -- the reference's call pattern (gym_env/env.py:22,261-262: 1000 runs per query) for a
 * caller whose states already live on the GPU: ONE kernel launch and nothing else (mcq_eval_direct_kernel in its device
 * mode): no prep kernel, no cost prefix, no atomics -- every query is owned by 2^lg waves of one block, lg chosen from
 * the query count alone, validation happens in the kernel. */
int mcq_eval_batch_device_small(mcq_ctx *c, const void *d_queries, size_t n, uint64_t seed, uint64_t first_query_id,
                                void *d_results, void *hip_stream) {
    ABI_GUARD_BEGIN
    if (!c) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_device_small: null context");
    if (n == 0) return MCQ_OK;
    if (!d_queries || !d_results) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_device_small: null buffer");
    if (n >= (1u << 24)) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_device_small: n too large");
    MCQ_ENTER(c, "mcq_eval_batch_device_small");
    McqDeviceScope dev_(c->device);
    HIP_TRY(dev_.err);
    hipStream_t s = (hipStream_t)hip_stream;
    const bool capturing = stream_capturing(s);
    /* the block counter of this stream (calls on different streams may overlap): a line of d_done_dev per scratch slot */
    if (!c->d_done_dev.p) {
        if (capturing)
            return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_device_small: issue one call on this context before capturing");
        HIP_TRY(c->d_done_dev.reserve(64 * mcq_ctx::kScratch));
        HIP_TRY(hipMemset(c->d_done_dev.p, 0, 64 * mcq_ctx::kScratch));
    }
    mcq_ctx::Scratch *sc = nullptr;
    int rc = scratch_for(c, s, 0, capturing, &sc);
    if (rc) return rc;
    uint32_t *d_done = reinterpret_cast<uint32_t *>(static_cast<char *>(c->d_done_dev.p) + 64 * (sc - c->scratch));
    /* about 16 waves per CU in all, at most eight per query (a sixteenth wave steps over fifteen iterations' words) */
    uint32_t lg = 0;
    while (lg < 3u && ((uint64_t)n << (lg + 1u)) <= 16ull * (uint64_t)c->n_cu) lg++;
    if (c->split_max < lg) lg = c->split_max;
    const uint64_t waves = (uint64_t)n << lg, blocks = (waves + 15u) / 16u;
    const uint32_t grid = (uint32_t)(blocks < (uint64_t)c->n_cu ? blocks : (uint64_t)c->n_cu);
    const uint32_t rounds = (uint32_t)((blocks + grid - 1u) / grid);
    const uint32_t mode = c->law == MCQ_LAW_UNIFORM ? MCQ_INTERNAL_MODE_UNIFORM : MCQ_MODE_PHILOX;
    const bool timed = c->timing && !capturing;
    const int slot = (int)(c->n_timed % mcq_ctx::kRing);
    c->last_ms = 0.f;
    HIP_TRY(mcq_launch_eval_direct((int)mode, d_queries, nullptr, rounds, lg ? 1u : 0u, (mcq_result *)d_results, seed, first_query_id,
                                   c->d_luts, grid, d_done, d_done + 8 /* nobody polls: the stream orders the call */, 1u, s,
                                   timed ? c->ev0[slot] : nullptr, timed ? c->ev1[slot] : nullptr, nullptr, (uint32_t)n, lg));
    if (timed) c->n_timed++;
    if (!capturing && sc != &c->scratch[0]) {
        HIP_TRY(hipEventRecord(sc->done, s));
        sc->done_recorded = true;
    }
    return MCQ_OK;
    ABI_GUARD_END("mcq_eval_batch_device_small")
}

/* The completion flag of the kernels that hand their rows over in pinned memory (mcq_eval_direct_kernel,
 * mcq_publish_kernel): a word the last block sets to the call's ticket, and the device counter that finds that block. */
static int flag_ready(mcq_ctx *c) {
    if (!c->h_flag.p) {
        HIP_TRY(c->h_flag.reserve(64));
        memset(c->h_flag.p, 0, 64);
        HIP_TRY(c->d_done.reserve(64));
        HIP_TRY(hipMemsetAsync(c->d_done.p, 0, 64, c->stream));
    }
    return MCQ_OK;
}
static uint32_t next_ticket(mcq_ctx *c) {
    if (++c->direct_ticket == 0u) c->direct_ticket = 1u; /* 0 is the flag's resting value */
    return c->direct_ticket;
}
/* Picking the rows up at the flag saves the end-of-kernel handshake of a stream synchronisation (24 us against 7.5 us
 * from launch to flag for an empty kernel, tools/launch_floor.hip).  Polling burns a core, so it is kept for SHORT
 * work: a call whose kernels are expected to run for est_us microseconds (scheduling cost x the measured time per cost
 * unit) first sleeps through most of that and only then polls; a kernel that has not answered a few milliseconds
 * after it was expected is left to hipStreamSynchronize, which reports what went wrong. */
static int wait_ticket(mcq_ctx *c, uint32_t ticket, bool *by_flag, double est_us) {
    const volatile uint32_t *flag = static_cast<const volatile uint32_t *>(c->h_flag.p);
    bool seen = false;
    if (c->direct_poll) {
        /* (half of the estimate: the constant behind est_us was measured on a whole MI355X under one dealing law; an
         * overshoot would add to the call what the flag was built to save.  MCQ_DIRECT_SLEEP=0: never sleep) */
        if (c->direct_sleep && est_us > 300.0) std::this_thread::sleep_for(std::chrono::microseconds((long long)(0.5 * est_us)));
        const auto t1 = std::chrono::steady_clock::now();
        for (uint32_t spin = 0;; spin++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ticket) { seen = true; break; }
            if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t1 > std::chrono::milliseconds(5)) break;
            __builtin_ia32_pause();
        }
    }
    if (!seen) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        /* late, or the flag never came: whatever the reason, the next launch starts from a clean block counter */
        if (c->direct_poll && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != ticket)
            HIP_TRY(hipMemsetAsync(c->d_done.p, 0, 64, c->stream));
    }
    if (by_flag) *by_flag = seen;
    return MCQ_OK;
}
/* expected kernel time of work of scheduling cost `cost` (mcq_task_weight units, about 0.0066 ns each on a whole MI355X) */
static double cost_to_us(const mcq_ctx *c, uint64_t cost) { return (double)cost * 6.6e-6 * 256.0 / (double)(c->n_cu > 0 ? c->n_cu : 256); }

/* Production mode from host buffers.  The host has the queries in its hands, so it prices them itself: the cost
 * prefix travels with the queries in ONE copy, the result rows are zero already (every call leaves them so), and
 * the only kernel of the call is the evaluation kernel -- no prep launch on the path of the reference's own call
 * pattern, thousands of 1000-run queries (gym_env/env.py:22,261-262).  `runs` (and nothing else) of a row is the
 * host's to fill in. */
static int eval_host_philox(mcq_ctx *c, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id, uint32_t part,
                            uint32_t n_parts, mcq_result *out, const char *who) {
    if (!q || !out) return mcq_fail(MCQ_EINVAL, who, "null buffer");
    if (n > 0x7fffffffu) return mcq_fail(MCQ_EINVAL, who, "n too large");
    int rc = validate(q, n);
    if (rc) return rc;
    const size_t q_bytes = n * sizeof(mcq_query), p_bytes = (n + 3) * sizeof(uint64_t), r_bytes = n * sizeof(mcq_result);
    const size_t r_pad = (n + (n & 1u)) * sizeof(mcq_result); /* whole 16-byte words (mcq_publish_kernel) */
    /* behind the records: the cost prefix, then (one-launch path) the wave layout: at most n + 2 * 16 * n_cu waves, dealt
     * to the blocks in whole rounds */
    const size_t a_off = (q_bytes + p_bytes + 15u) & ~(size_t)15u, a_cap = n + 96u * (size_t)c->n_cu + 64u; /* waves */
    HIP_TRY(c->h_q.reserve(a_off + a_cap * (sizeof(mcq_query) + sizeof(uint32_t))));
    HIP_TRY(c->d_q.reserve(q_bytes + p_bytes));
    HIP_TRY(c->h_res.reserve(r_pad));
    if (r_pad > c->d_res.cap) c->res_clean = 0;
    HIP_TRY(c->d_res.reserve(r_pad));
    memcpy(c->h_q.p, q, q_bytes);
    uint64_t *prefix = reinterpret_cast<uint64_t *>(static_cast<char *>(c->h_q.p) + q_bytes);
    uint64_t total_tasks = 0, max_tasks = 0, cost = 0, unsplit = 0;
    for (size_t i = 0; i < n; i++) {
        const McqQueryWords w = mcq_query_words(q[i]);
        const McqPart pt = mcq_part(mcq_task_count(w), w.runs(), part, n_parts);
        const uint64_t t = pt.t_hi - pt.t_lo;
        prefix[i] = cost;
        cost += t * mcq_task_weight(w);
        total_tasks += t;
        unsplit += tasks_of(q[i]);
        if (t > max_tasks) max_tasks = t;
    }
    if (unsplit > 0xfffffff0ull) return mcq_fail(MCQ_EINVAL, who, "too many iterations in one call");
    if (n_parts == 1 && max_tasks <= c->direct_max_tasks && total_tasks > 0 && n < (1u << 24)) {
        /* Small queries -- the reference's call pattern: ONE launch (mcq_eval_direct_kernel).  The kernel reads the
         * records from this pinned buffer and stores finished rows into the pinned result buffer.  Layout of the work:
         * about 16 waves per CU in all, every query a power-of-two number of them in proportion to its cost (so all
         * waves carry about the same work), the waves of a query side by side in one block. */
        const uint32_t mode = c->law == MCQ_LAW_UNIFORM ? MCQ_INTERNAL_MODE_UNIFORM : MCQ_MODE_PHILOX;
        if (n >= c->direct_uniform_min && max_tasks <= 8u) {
            /* Many small queries: the kernel lays its own work out (its device mode, as mcq_eval_batch_device_small:
             * 2^lg waves per query from the query count alone) and reads the caller's records -- copied into this pinned
             * buffer, nothing else -- across PCIe: the host neither prices nor places a thousand queries (12 us), and a
             * uniform cut turned out no slower on the GPU than the cost-proportional one. */
            uint32_t lg = 0;
            while (lg < 3u && ((uint64_t)n << (lg + 1u)) <= 16ull * (uint64_t)c->n_cu) lg++;
            if (c->split_max < lg) lg = c->split_max;
            const uint64_t blocks = (((uint64_t)n << lg) + 15u) / 16u;
            const uint32_t grid = (uint32_t)(blocks < (uint64_t)c->n_cu ? blocks : (uint64_t)c->n_cu);
            const uint32_t rounds = (uint32_t)((blocks + grid - 1u) / grid);
            rc = flag_ready(c);
            if (rc) return rc;
            const uint32_t ticket = next_ticket(c);
            const int slot = (int)(c->n_timed % mcq_ctx::kRing);
            c->last_ms = 0.f;
            HIP_TRY(mcq_launch_eval_direct((int)mode, c->h_q.dev, nullptr, rounds, lg ? 1u : 0u, (mcq_result *)c->h_res.dev, seed,
                                           first_query_id, c->d_luts, grid, (uint32_t *)c->d_done.p, (uint32_t *)c->h_flag.dev,
                                           ticket, c->stream, c->timing ? c->ev0[slot] : nullptr,
                                           c->timing ? c->ev1[slot] : nullptr, nullptr, (uint32_t)n, lg));
            if (c->timing) c->n_timed++;
            rc = wait_ticket(c, ticket, nullptr);
            if (rc) return rc;
            memcpy(out, c->h_res.p, r_bytes);
            return MCQ_OK;
        }
        std::vector<uint64_t> &qcost = c->direct_cost;
        qcost.resize(n);
        for (size_t i = 0; i < n; i++) qcost[i] = (i + 1 < n ? prefix[i + 1] : cost) - prefix[i];
        McqDirectLayout &lay = c->direct_layout;
        /* at most eight waves per query here: a sixteenth wave steps over fifteen iterations' words to run one
         * (tools/single_probe.py: one 1000-run query 1 us slower with sixteen, whatever the players and the table) */
        mcq_direct_layout(qcost.data(), n, (uint32_t)c->n_cu, c->split_max < 3u ? c->split_max : 3u, lay);
        const uint32_t grid = lay.grid, rounds = lay.rounds;
        const size_t a_words = lay.slots;
        if (a_words > a_cap) return mcq_fail(MCQ_EDEVICE, who, "internal: wave layout larger than its bound");
        /* behind the caller's records and the prefix: one record copy per wave (its reserved bytes carry log2 of the
         * query's wave count and the wave's cut number), then one query index per wave */
        mcq_query *work_rec = reinterpret_cast<mcq_query *>(static_cast<char *>(c->h_q.p) + a_off);
        uint32_t *work_qi = reinterpret_cast<uint32_t *>(work_rec + a_words);
        /* a launch of few waves carries its work in the kernel arguments instead (no read across PCIe by the kernel) */
        const bool by_karg = a_words <= MCQ_DIRECT_KARG_SLOTS && rounds <= 8u;
        McqDirectKarg &karg = c->direct_karg;
        if (by_karg) {
            work_rec = reinterpret_cast<mcq_query *>(karg.rec);
            work_qi = karg.qi;
        }
        if (!mcq_direct_write_records(lay, q, n, work_rec, work_qi, by_karg ? (size_t)MCQ_DIRECT_KARG_SLOTS : a_cap))
            return mcq_fail(MCQ_EDEVICE, who, "internal: wave layout does not fit its slots");
        rc = flag_ready(c);
        if (rc) return rc;
        const uint32_t ticket = next_ticket(c);
        const int slot = (int)(c->n_timed % mcq_ctx::kRing);
        const bool timed = c->timing;
        c->last_ms = 0.f;
        static const bool trace = getenv("MCQ_TRACE") != nullptr; /* phase times of this path on stderr (tuning) */
        const auto t0 = std::chrono::steady_clock::now();
        const char *dev_rec = static_cast<const char *>(c->h_q.dev) + a_off;
        HIP_TRY(mcq_launch_eval_direct((int)mode, dev_rec,
                                       reinterpret_cast<const uint32_t *>(dev_rec + a_words * sizeof(mcq_query)), rounds,
                                       lay.merge ? 1u : 0u, (mcq_result *)c->h_res.dev, seed, first_query_id, c->d_luts, grid,
                                       (uint32_t *)c->d_done.p, (uint32_t *)c->h_flag.dev, ticket, c->stream,
                                       timed ? c->ev0[slot] : nullptr, timed ? c->ev1[slot] : nullptr,
                                       by_karg ? &karg : nullptr));
        if (timed) c->n_timed++;
        const auto t1 = std::chrono::steady_clock::now();
        bool seen = false; /* the last block raises the flag once every row is out */
        rc = wait_ticket(c, ticket, &seen);
        if (rc) return rc;
        const auto t2 = std::chrono::steady_clock::now();
        memcpy(out, c->h_res.p, r_bytes);
        if (trace) {
            const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
                return std::chrono::duration<double, std::micro>(b - a).count();
            };
            float kms = 0.f;
            if (timed) (void)kernel_times_impl(c, &kms, 1);
            fprintf(stderr, "mcq direct n=%zu waves=%llu grid=%u rounds=%u: launch %.1f us, wait %.1f us (%s), copy out %.1f us, "
                    "kernel %.1f us\n", n, (unsigned long long)lay.waves, grid, rounds, us(t0, t1), us(t1, t2),
                    seen ? "flag" : "stream sync", us(t2, std::chrono::steady_clock::now()), 1e3 * kms);
        }
        return MCQ_OK;
    }
    prefix[n] = cost;
    prefix[n + 1] = 0;
    prefix[n + 2] = 0;
    HIP_TRY(hipMemcpyAsync(c->d_q.p, c->h_q.p, q_bytes + p_bytes, hipMemcpyHostToDevice, c->stream));
    if (c->res_clean < r_pad) HIP_TRY(hipMemsetAsync(c->d_res.p, 0, r_pad, c->stream));
    c->res_clean = 0; /* dirty until this call has put its rows back to zero */
    c->last_ms = 0.f;
    if (total_tasks) {
        rc = mcq_run_slice(c, MCQ_MODE_PHILOX, (const mcq_query *)c->d_q.p, (uint32_t)n, (mcq_result *)c->d_res.p, seed,
                           first_query_id, total_tasks, nullptr, nullptr, c->stream, true, max_tasks, part, n_parts, nullptr,
                           reinterpret_cast<const uint64_t *>(static_cast<const char *>(c->d_q.p) + q_bytes));
        if (rc) return rc;
    }
    if (n <= c->publish_max_rows) {
        /* few rows: a small kernel behind the evaluation moves them into pinned memory, zeroes them in HBM for the next
         * call and raises the flag (no D2H copy, no stream synchronisation, no memset) */
        rc = flag_ready(c);
        if (rc) return rc;
        const uint32_t ticket = next_ticket(c);
        HIP_TRY(mcq_launch_publish((mcq_result *)c->d_res.p, (mcq_result *)c->h_res.dev, n + (n & 1u), (uint32_t *)c->d_done.p,
                                   (uint32_t *)c->h_flag.dev, ticket, c->stream));
        rc = wait_ticket(c, ticket, nullptr, cost_to_us(c, cost));
        if (rc) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(c->h_res.p, c->d_res.p, r_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemsetAsync(c->d_res.p, 0, r_pad, c->stream)); /* not waited for: the next call finds its rows zero */
    }
    c->res_clean = r_pad;
    if (!total_tasks || !c->timing || kernel_times_impl(c, &c->last_ms, 1) != 1) c->last_ms = 0.f;
    memcpy(out, c->h_res.p, r_bytes);
    for (size_t i = 0; i < n; i++) out[i].runs = mcq_part(tasks_of(q[i]), q[i].runs, part, n_parts).runs;
    return MCQ_OK;
}

static int eval_batch_impl(mcq_ctx *c, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id, int mode,
                           uint32_t part, uint32_t n_parts, mcq_result *out, const char *who) {
    if (mode != MCQ_MODE_PHILOX && mode != MCQ_MODE_REPLAY_MT19937) return mcq_fail(MCQ_EINVAL, who, "bad mode");
    if (n_parts == 0 || part >= n_parts) return mcq_fail(MCQ_EINVAL, who, "part must be < n_parts");
    if (n_parts > 1 && mode != MCQ_MODE_PHILOX)
        return mcq_fail(MCQ_EINVAL, who, "only MCQ_MODE_PHILOX can split the iterations of a query");
    if (n == 0) return MCQ_OK;
    if (!c) return mcq_fail(MCQ_EINVAL, who, "null context");
    MCQ_ENTER(c, who);
    McqDeviceScope dev_(c->device);
    HIP_TRY(dev_.err);
    if (mode == MCQ_MODE_PHILOX) return eval_host_philox(c, q, n, seed, first_query_id, part, n_parts, out, who);
    int rc = stage_queries(c, q, n, out, who);
    if (rc) return rc;
    return replay_batch(c, q, n, seed, first_query_id, nullptr, out);
}

int mcq_eval_batch(mcq_ctx *c, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id, int mode,
                   mcq_result *out) {
    ABI_GUARD_BEGIN
    return eval_batch_impl(c, q, n, seed, first_query_id, mode, 0, 1, out, "mcq_eval_batch");
    ABI_GUARD_END("mcq_eval_batch")
}

int mcq_eval_batch_part(mcq_ctx *c, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id,
                        uint32_t part, uint32_t n_parts, mcq_result *out) {
    ABI_GUARD_BEGIN
    return eval_batch_impl(c, q, n, seed, first_query_id, MCQ_MODE_PHILOX, part, n_parts, out, "mcq_eval_batch_part");
    ABI_GUARD_END("mcq_eval_batch_part")
}

int mcq_eval_one(mcq_ctx *c, const mcq_query *q, uint64_t seed, int mode, mcq_result *out) {
    return mcq_eval_batch(c, q, 1, seed, 0, mode, out);
}

int mcq_eval_batch_ext(mcq_ctx *c, const mcq_query *q, const mcq_query_ext *ext, size_t n, uint64_t seed,
                       uint64_t first_query_id, int mode, mcq_result *out) {
    ABI_GUARD_BEGIN
    if (mode != MCQ_MODE_PHILOX && mode != MCQ_MODE_REPLAY_MT19937) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_ext: bad mode");
    if (n == 0) return MCQ_OK;
    if (!c) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_ext: null context");
    if (!q || !ext || !out) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_ext: null buffer");
    if (n > 0x7fffffffu) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_ext: n too large");
    uint64_t total_tasks = 0;
    uint32_t lists_stride = 1; /* candidate lists per query of the production mode: the batch's maximum */
    uint32_t most_tasks = 0;
    for (size_t i = 0; i < n; i++) {
        const McqExtRec er = {reinterpret_cast<const uint32_t *>(&ext[i])};
        const McqQueryWords qw = mcq_query_words(q[i]);
        if (!mcq_query_ext_valid(qw, er)) {
            char buf[220];
            snprintf(buf, sizeof buf, "extended query %zu invalid (distinct card ids < 52 among hole/table/ghost/known hands, "
                     "at most 9 further known hands, n_players >= known hands, used ranges not empty)", i);
            return mcq_fail(MCQ_EINVAL, buf);
        }
        const uint32_t nl = mcq_ext_n_lists(qw, er);
        if (nl > lists_stride) lists_stride = nl;
        const uint32_t t = mcq_ext_task_count(qw, mode == MCQ_MODE_PHILOX ? mcq_ext_stream_iters(qw, er) : MCQ_STREAM_ITERS);
        total_tasks += t;
        if (t > most_tasks) most_tasks = t;
    }
    if (total_tasks > 0xfffffff0ull) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_ext: too many iterations in one call");
    MCQ_ENTER(c, "mcq_eval_batch_ext");
    McqDeviceScope dev_(c->device);
    HIP_TRY(dev_.err);
    HIP_TRY(c->h_res.reserve(n * sizeof(mcq_result)));
    if (mode == MCQ_MODE_PHILOX && c->ext_small && n <= MCQ_EXT_SMALL_Q && lists_stride <= MCQ_EXT_SMALL_LISTS &&
        most_tasks <= MCQ_EXT_SMALL_TASKS && !stream_capturing(c->stream)) {
        /* what a decision of the reference's agents asks for -- one ranged query of a thousand iterations -- in ONE
         * launch: the records in the kernel arguments, a block per query (lists laid out in its LDS), the rows into
         * pinned memory, the flag: 140 -> ~25 us per call */
        static_assert(sizeof(mcq_query_ext) == sizeof(((McqExtSmallKarg *)0)->ext[0]), "extension record in the kernel arguments");
        McqExtSmallKarg karg;
        uint32_t n_blocks = 0, first_block[MCQ_EXT_SMALL_Q + 1], tasks[MCQ_EXT_SMALL_Q];
        for (size_t i = 0; i < n; i++) {
            memcpy(karg.q[i], &q[i], sizeof(mcq_query));
            memcpy(karg.ext[i], &ext[i], sizeof(mcq_query_ext));
            const McqExtRec er = {reinterpret_cast<const uint32_t *>(&ext[i])};
            const McqQueryWords qw = mcq_query_words(q[i]);
            tasks[i] = mcq_ext_task_count(qw, mcq_ext_stream_iters(qw, er));
        }
        uint32_t wpb = 4; /* working waves per block: the fewest with which the launch's blocks suffice */
        for (; wpb < 16u; wpb <<= 1) {
            uint32_t need = 0;
            for (size_t i = 0; i < n; i++) need += tasks[i] > wpb ? (tasks[i] + wpb - 1u) / wpb : 1u;
            if (need <= MCQ_EXT_SMALL_BLOCKS) break;
        }
        for (size_t i = 0; i < n; i++) {
            const uint32_t parts = tasks[i] > wpb ? (tasks[i] + wpb - 1u) / wpb : 1u;
            first_block[i] = n_blocks;
            for (uint32_t p = 0; p < parts; p++) karg.blk[n_blocks++] = (uint32_t)i | (p << 8) | (parts << 16) | (wpb << 24);
        }
        first_block[n] = n_blocks;
        HIP_TRY(c->h_res.reserve(MCQ_EXT_SMALL_BLOCKS * sizeof(mcq_result)));
        int rc = flag_ready(c);
        if (rc) return rc;
        const uint32_t ticket = next_ticket(c);
        const int slot = (int)(c->n_timed % mcq_ctx::kRing);
        const bool timed = c->timing;
        c->last_ms = 0.f;
        HIP_TRY(mcq_launch_eval_ext_small(&karg, n_blocks, (mcq_result *)c->h_res.dev, seed, first_query_id, c->d_luts,
                                          (uint32_t *)c->d_done.p, (uint32_t *)c->h_flag.dev, ticket, c->stream,
                                          timed ? c->ev0[slot] : nullptr, timed ? c->ev1[slot] : nullptr));
        if (timed) c->n_timed++;
        rc = wait_ticket(c, ticket, nullptr);
        if (rc) return rc;
        if (timed && kernel_times_impl(c, &c->last_ms, 1) != 1) c->last_ms = 0.f;
        const uint64_t *hr = (const uint64_t *)c->h_res.p; /* one 13-word row per block */
        static_assert(sizeof(mcq_result) == 13 * sizeof(uint64_t), "result row");
        for (size_t i = 0; i < n; i++) {
            uint64_t row[13] = {q[i].runs};
            for (uint32_t b = first_block[i]; b < first_block[i + 1]; b++) {
                if (hr[13u * b] != q[i].runs)
                    return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_ext: a range cannot be dealt from the remaining cards");
                for (int k = 1; k < 13; k++) row[k] += hr[13u * b + k];
            }
            memcpy(&out[i], row, sizeof row);
        }
        return MCQ_OK;
    }
    HIP_TRY(c->h_q.reserve(n * (sizeof(mcq_query) + sizeof(mcq_query_ext))));
    HIP_TRY(c->d_q.reserve(n * sizeof(mcq_query)));
    HIP_TRY(c->d_ext.reserve(n * sizeof(mcq_query_ext)));
    HIP_TRY(c->d_res.reserve(n * sizeof(mcq_result)));
    c->res_clean = 0;
    HIP_TRY(c->scratch[0].prefix.reserve((n + 3) * sizeof(uint64_t)));
    uint8_t *hq = (uint8_t *)c->h_q.p;
    memcpy(hq, q, n * sizeof(mcq_query));
    memcpy(hq + n * sizeof(mcq_query), ext, n * sizeof(mcq_query_ext));
    HIP_TRY(hipMemcpyAsync(c->d_q.p, hq, n * sizeof(mcq_query), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_ext.p, hq + n * sizeof(mcq_query), n * sizeof(mcq_query_ext), hipMemcpyHostToDevice,
                           c->stream));
    c->last_ms = 0.f;
    uint32_t grid, block;
    pick_geometry(c, mode, total_tasks, &grid, &block);
    if (mode == MCQ_MODE_REPLAY_MT19937) {
        /* the draw buffer the stream walk fills and the evaluation kernel reads (one chunk: the extended path is a
         * feature path, not a bulk path); the walk itself runs on the device: mcq_mt_parse_ext_kernel, one wave per query */
        HIP_TRY(c->h_off.reserve(n * sizeof(uint64_t)));
        uint64_t *off = (uint64_t *)c->h_off.p, bytes = 0;
        for (size_t i = 0; i < n; i++) {
            off[i] = bytes;
            bytes += (((uint64_t)q[i].runs + 63u) & ~63ull) * mcq_ext_draws_per_iteration(q[i], ext[i]);
        }
        if (bytes > (1ull << 32)) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_ext: replay batch too large (split it)");
        HIP_TRY(c->d_draws.reserve(bytes + 64));
        HIP_TRY(c->d_off.reserve(n * sizeof(uint64_t)));
        HIP_TRY(hipMemcpyAsync(c->d_off.p, off, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(mcq_launch_prep_ext((const mcq_query *)c->d_q.p, (const mcq_query_ext *)c->d_ext.p, (uint32_t)n, mode,
                                (mcq_result *)c->d_res.p, (uint64_t *)c->scratch[0].prefix.p, c->stream));
    const int slot = (int)(c->n_timed % mcq_ctx::kRing);
    hipEvent_t t0 = c->timing ? c->ev0[slot] : nullptr; /* parity mode: the timed region starts in front of the stream walk */
    if (mode == MCQ_MODE_REPLAY_MT19937) {
        if (t0) HIP_TRY(hipEventRecord(t0, c->stream));
        t0 = nullptr;
        HIP_TRY(mcq_launch_mt_parse_ext((const mcq_query *)c->d_q.p, (const mcq_query_ext *)c->d_ext.p, (uint32_t)n,
                                        (uint32_t)(seed + first_query_id), (uint8_t *)c->d_draws.p, (const uint64_t *)c->d_off.p,
                                        (mcq_result *)c->d_res.p,
                                        reinterpret_cast<uint32_t *>((uint64_t *)c->scratch[0].prefix.p + n + 2), (uint32_t)c->n_cu,
                                        c->stream));
    }
    if (mode == MCQ_MODE_PHILOX) { /* the candidate lists of the ranges, once per query */
        HIP_TRY(c->d_lists.reserve(n * (size_t)lists_stride * MCQ_EXT_LIST_STRIDE * sizeof(uint16_t)));
        HIP_TRY(c->d_cnts.reserve(n * (size_t)lists_stride * sizeof(uint32_t)));
        HIP_TRY(mcq_launch_ext_lists((const mcq_query *)c->d_q.p, (const mcq_query_ext *)c->d_ext.p, (uint32_t)n, lists_stride,
                                     (uint16_t *)c->d_lists.p, (uint32_t *)c->d_cnts.p, c->stream));
    }
    HIP_TRY(mcq_launch_eval_ext(mode, (const mcq_query *)c->d_q.p, (const mcq_query_ext *)c->d_ext.p, (uint32_t)n,
                                (const uint64_t *)c->scratch[0].prefix.p, (mcq_result *)c->d_res.p, seed, first_query_id, c->d_luts,
                                (const uint8_t *)c->d_draws.p, (const uint64_t *)c->d_off.p, (const uint16_t *)c->d_lists.p,
                                (const uint32_t *)c->d_cnts.p, lists_stride, grid, block, c->stream,
                                t0, c->timing ? c->ev1[slot] : nullptr));
    if (c->timing) c->n_timed++;
    HIP_TRY(hipMemcpyAsync(c->h_res.p, c->d_res.p, n * sizeof(mcq_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->timing || kernel_times_impl(c, &c->last_ms, 1) != 1) c->last_ms = 0.f;
    const mcq_result *hr = (const mcq_result *)c->h_res.p;
    for (size_t i = 0; i < n; i++)
        if (hr[i].runs != q[i].runs || hr[i].passes == ~0ull) /* (parity mode: the stream walk marks such a query) */
            return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_ext: a range cannot be dealt from the remaining cards");
    memcpy(out, hr, n * sizeof(mcq_result));
    return MCQ_OK;
    ABI_GUARD_END("mcq_eval_batch_ext")
}

int mcq_eval_batch_numpy_stream(mcq_ctx *c, const mcq_query *q, size_t n, uint32_t *mt_key, uint32_t *mt_pos,
                                mcq_result *out) {
    ABI_GUARD_BEGIN
    if (n == 0) return MCQ_OK;
    if (!mt_key || !mt_pos || *mt_pos > 624) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_numpy_stream: bad MT19937 state");
    if (!c) return mcq_fail(MCQ_EINVAL, "mcq_eval_batch_numpy_stream: null context");
    MCQ_ENTER(c, "mcq_eval_batch_numpy_stream");
    McqDeviceScope dev_(c->device);
    HIP_TRY(dev_.err);
    int rc = stage_queries(c, q, n, out, "mcq_eval_batch_numpy_stream");
    if (rc) return rc;
    McqMt19937 g;
    memcpy(g.mt, mt_key, sizeof g.mt);
    g.pos = *mt_pos;
    rc = replay_batch(c, q, n, 0, 0, &g, out);
    if (rc) return rc;
    memcpy(mt_key, g.mt, sizeof g.mt);
    *mt_pos = g.pos;
    return MCQ_OK;
    ABI_GUARD_END("mcq_eval_batch_numpy_stream")
}

int mcq_showdown(mcq_ctx *c, const uint8_t *hands, size_t n_tables, int n_players, uint8_t *winner,
                 uint8_t *winner_type, uint32_t *keys) {
    ABI_GUARD_BEGIN
    if (!c) return mcq_fail(MCQ_EINVAL, "mcq_showdown: null context");
    if (n_tables == 0) return MCQ_OK;
    if (!hands || !winner || !winner_type) return mcq_fail(MCQ_EINVAL, "mcq_showdown: null buffer");
    if (n_players < 1 || n_players > 10) return mcq_fail(MCQ_EINVAL, "mcq_showdown: n_players must be in [1,10]");
    if (n_tables > 0x7fffffffu / 70u) return mcq_fail(MCQ_EINVAL, "mcq_showdown: n_tables too large");
    /* The host only moves bytes: the hands go through pinned memory in pieces, each piece's kernel launched as soon as
     * the piece is there (the kernel reads it in place across PCIe with 16-byte loads while the host copies the next
     * piece), validation happens on the device, winners / types / keys come back in pinned memory and the last kernel
     * raises the completion flag -- no blocking copies, no stream synchronisation. */
    const size_t per_table = 7u * (size_t)n_players, nh = n_tables * (size_t)n_players;
    const size_t pad_tables = (n_tables + 255u) & ~(size_t)255u;
    const size_t in_bytes = pad_tables * per_table, key_bytes = keys ? pad_tables * n_players * sizeof(uint32_t) : 0;
    MCQ_ENTER(c, "mcq_showdown");
    McqDeviceScope dev_(c->device);
    HIP_TRY(dev_.err);
    HIP_TRY(c->h_misc.reserve(in_bytes + 2 * pad_tables + key_bytes + 64));
    int rc = flag_ready(c);
    if (rc) return rc;
    uint8_t *hp = (uint8_t *)c->h_misc.p, *dp = (uint8_t *)c->h_misc.dev;
    const size_t off_w = in_bytes, off_t = off_w + pad_tables, off_k = off_t + pad_tables; /* all multiples of 16 */
    volatile uint32_t *bad = static_cast<volatile uint32_t *>(c->h_flag.p) + 1;
    *bad = 0;
    size_t piece = (256u << 10) / per_table & ~(size_t)255u; /* tables per piece: about 256 KB, whole tiles */
    if (piece < 256u) piece = 256u;
    const uint32_t ticket = next_ticket(c);
    for (size_t a = 0; a < n_tables; a += piece) {
        const size_t b = a + piece < n_tables ? a + piece : n_tables;
        memcpy(hp + a * per_table, hands + a * per_table, (b - a) * per_table);
        HIP_TRY(mcq_launch_showdown(dp + a * per_table, (uint32_t)(b - a), (uint32_t)n_players, c->d_luts, dp + off_w + a,
                                    dp + off_t + a, keys ? reinterpret_cast<uint32_t *>(dp + off_k) + a * n_players : nullptr,
                                    static_cast<uint32_t *>(c->h_flag.dev) + 1, (uint32_t *)c->d_done.p,
                                    (uint32_t *)c->h_flag.dev, b == n_tables ? ticket : 0u, (uint32_t)c->n_cu, c->stream));
    }
    rc = wait_ticket(c, ticket, nullptr);
    if (rc) return rc;
    if (__atomic_load_n(bad, __ATOMIC_ACQUIRE) != 0u)
        return mcq_fail(MCQ_EINVAL, "mcq_showdown: a hand needs 7 distinct card ids < 52");
    memcpy(winner, hp + off_w, n_tables);
    memcpy(winner_type, hp + off_t, n_tables);
    if (keys) memcpy(keys, hp + off_k, nh * sizeof(uint32_t));
    return MCQ_OK;
    ABI_GUARD_END("mcq_showdown")
}

int mcq_exact_batch(mcq_ctx *c, const mcq_query *q, size_t n, int law, mcq_result *out) {
    ABI_GUARD_BEGIN
    if (!c) return mcq_fail(MCQ_EINVAL, "mcq_exact_batch: null context");
    if (n == 0) return MCQ_OK;
    if (!q || !out) return mcq_fail(MCQ_EINVAL, "mcq_exact_batch: null buffer");
    if (law != MCQ_LAW_REFERENCE && law != MCQ_LAW_UNIFORM) return mcq_fail(MCQ_EINVAL, "mcq_exact_batch: bad law");
    if (n > 0x7fffffffu) return mcq_fail(MCQ_EINVAL, "mcq_exact_batch: n too large");
    int rc = validate(q, n);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++)
        if (q[i].n_players > 3)
            return mcq_fail(MCQ_EINVAL, "mcq_exact_batch: exact enumeration covers 1 to 3 players");
    MCQ_ENTER(c, "mcq_exact_batch");
    McqDeviceScope dev_(c->device);
    HIP_TRY(dev_.err);
    HIP_TRY(c->d_res.reserve(n * sizeof(mcq_result)));
    c->res_clean = 0;
    HIP_TRY(c->h_res.reserve(n * sizeof(mcq_result)));
    HIP_TRY(hipMemsetAsync(c->d_res.p, 0, n * sizeof(mcq_result), c->stream));
    /* one launch per kind (two opponents / fewer) and 65 535 queries: the jobs travel in pinned memory */
    HIP_TRY(c->h_misc.reserve(n * sizeof(McqExactJob)));
    McqExactJob *jobs = static_cast<McqExactJob *>(c->h_misc.p);
    const McqExactJob *d_jobs = static_cast<const McqExactJob *>(c->h_misc.dev);
    size_t n_two = 0;
    for (size_t i = 0; i < n; i++) /* the two-opponent queries first, the others behind them */
        if (q[i].n_players == 3) mcq_exact_plan(&q[i], (uint32_t)i, (uint32_t)c->n_cu, &jobs[n_two++]);
    size_t at = n_two;
    for (size_t i = 0; i < n; i++)
        if (q[i].n_players != 3) mcq_exact_plan(&q[i], (uint32_t)i, (uint32_t)c->n_cu, &jobs[at++]);
    for (size_t a = 0; a < n;) {
        const bool two = a < n_two;
        const size_t end = two ? n_two : n, b = a + 65535u < end ? a + 65535u : end;
        uint32_t max_grid = 0;
        for (size_t i = a; i < b; i++) max_grid = jobs[i].grid > max_grid ? jobs[i].grid : max_grid;
        HIP_TRY(mcq_launch_exact(d_jobs + a, (uint32_t)(b - a), max_grid, two, law, (mcq_result *)c->d_res.p, c->d_luts, c->stream));
        a = b;
    }
    HIP_TRY(hipMemcpyAsync(c->h_res.p, c->d_res.p, n * sizeof(mcq_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_res.p, n * sizeof(mcq_result));
    return MCQ_OK;
    ABI_GUARD_END("mcq_exact_batch")
}

}  // extern "C"
