// mcq_ctx.hpp -- the engine context and the host-side helpers shared by the C-ABI translation units
// (mcq_host.cpp, mcq_multi.cpp).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/mcq.h"
#include "mcq_busy.hpp"
#include "mcq_internal.hpp"
#include "mcq_layout.hpp"

struct McqTables;

/* thread-local error text behind mcq_last_error(); returns `code` */
int mcq_fail(int code, const char *what, const char *detail = nullptr);

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return mcq_fail(e_ == hipErrorOutOfMemory ? MCQ_ENOMEM : MCQ_EDEVICE, #expr, hipGetErrorString(e_)); \
    } while (0)

/* No C++ exception may cross the C ABI (std::bad_alloc from the staging vectors, std::system_error from thread
 * creation): every entry point body runs inside this guard. */
#define ABI_GUARD_BEGIN try {
#define ABI_GUARD_END(who)                                                                      \
    }                                                                                           \
    catch (const std::bad_alloc &) { return mcq_fail(MCQ_ENOMEM, who, "out of host memory"); } \
    catch (const std::exception &ex) { return mcq_fail(MCQ_EDEVICE, who, ex.what()); }         \
    catch (...) { return mcq_fail(MCQ_EDEVICE, who, "unexpected exception"); }

/* Public entry points that take a context come in through this (after their null check): a second caller while a call
 * is running on the context is turned away. */
#define MCQ_ENTER(c, who)                      \
    McqBusyScope busy_(&(c)->busy);            \
    if (!busy_.ok) return mcq_fail(MCQ_EBUSY, who, MCQ_BUSY_MESSAGE)

/* Entry points select the context's device and leave the caller's current device as they found it (a host
 * application -- or torch -- keeps its own notion of "current device"). */
struct McqDeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit McqDeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
        else prev = -1; /* nothing to restore */
    }
    ~McqDeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    McqDeviceScope(const McqDeviceScope &) = delete;
    McqDeviceScope &operator=(const McqDeviceScope &) = delete;
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct PinBuf {
    void *p = nullptr;
    void *dev = nullptr; /* the same memory as the device sees it (pinned host memory is device-visible) */
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        dev = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        e = hipHostGetDevicePointer(&dev, p, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(p);
            p = nullptr;
            return e;
        }
        cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct mcq_ctx {
    McqBusyFlag busy; /* one call in flight per context: see MCQ_ENTER */
    int device = 0;
    int n_cu = 0;
    int occ[3] = {1, 1, 1}; /* resident kBlock-thread blocks per CU of the eval kernels (by internal mode) */
    int law = MCQ_LAW_REFERENCE;
    uint64_t replay_device_bytes = 4ull << 30; /* parity mode: draw bytes per launch (MCQ_REPLAY_DEVICE_BYTES) */
    uint32_t direct_max_tasks = 8; /* queries of at most this many 1024-iteration tasks take the one-launch path (MCQ_DIRECT_MAX_TASKS, 0 = never) */
    uint32_t load_waves = 16; /* waves per block that load the table image when fewer take work (MCQ_LOAD_WAVES) */
    uint32_t split_max = 4; /* finest cut of a task for small batches: 16 >> split_max iterations per lane */
    hipStream_t stream = nullptr;
    static constexpr int kRing = 64; /* event pairs around the most recent evaluation-kernel launches */
    hipEvent_t ev0[kRing] = {}, ev1[kRing] = {};
    uint64_t n_timed = 0;
    float last_ms = 0.f;
    McqTables *d_luts = nullptr;
    /* Scheduling scratch (cost prefix + the small-batch cut) of the evaluation launches, ONE PER STREAM the context
     * has launched on: the prep kernel of a later call must not overwrite what an evaluation kernel still running
     * on another stream reads.  Calls on one stream are ordered by the stream.  Slot 0 belongs to the context's own
     * stream (host entries). */
    static constexpr int kScratch = 16;
    struct Scratch {
        hipStream_t stream = nullptr;
        bool used = false, pinned = false, done_recorded = false;
        uint64_t last_use = 0;
        hipEvent_t done = nullptr; /* recorded after the latest evaluation kernel that reads this scratch */
        DevBuf prefix;
    } scratch[kScratch];
    uint64_t scratch_clock = 0;
    size_t res_clean = 0; /* leading bytes of d_res known to be zero (host entries leave their rows zeroed again) */
    DevBuf d_q, d_res, d_draws, d_off, d_ext, d_mt, d_lists, d_cnts;
    PinBuf h_q, h_res, h_draws, h_off, h_misc, h_flag;
    DevBuf d_done;                /* block counter of the one-launch path */
    DevBuf d_done_dev;            /* ... of mcq_eval_batch_device_small: one 64-byte line per scratch slot (stream), zeroed once */
    uint32_t direct_ticket = 0;   /* value the kernel raises the flag in h_flag to */
    McqDirectKarg direct_karg; /* one-launch path: the work of a small launch, passed by value */
    size_t publish_max_rows = 8192; /* host-buffer calls of at most this many rows get them through mcq_publish_kernel + flag (MCQ_PUBLISH_MAX_ROWS, 0 = never) */
    bool timing = false; /* mcq_set_kernel_timing: launches carry timestamp events */
    int mt_blocks_margin = 8; /* state blocks beyond the estimate (MCQ_MT_BLOCKS_MARGIN; negative: the tests' way to the fall-back) */
    bool mt_jump_always = false;
    bool mt_jump = true;   /* ... their blocks generated in segments side by side, start states by jump-ahead (MCQ_MT_JUMP=0: one work-group per query) */
    bool mt_blocks = true; /* parity mode, few long queries: the state blocks of a query parsed side by side (MCQ_MT_BLOCKS=0: always the serial walk) */
    bool ext_small = true; /* a few extended queries of the production mode in one launch (MCQ_EXT_SMALL=0: always the general path) */
    size_t direct_uniform_min = 128; /* one-launch path: from this many queries on the kernel lays its own work out (MCQ_DIRECT_UNIFORM_MIN; measured: 512 queries 39 -> 33 us per call, 1024: 48 -> 42, 4096: 137 -> 113) */
    bool direct_poll = true;      /* pick the rows up at the flag instead of synchronising the stream (MCQ_DIRECT_POLL) */
    bool direct_sleep = true;     /* sleep through half of a long kernel's expected time before polling (MCQ_DIRECT_SLEEP) */
    McqDirectLayout direct_layout;      /* the one-launch path's layout of the current call */
    std::vector<uint64_t> direct_cost;
};

/* shared between the translation units (defined in mcq_host.cpp) */
uint32_t mcq_tasks_of(const mcq_query &q);
int mcq_validate_queries(const mcq_query *q, size_t n);
/* prep + evaluation launch of n device-resident queries on stream s (asynchronous).  total_tasks = 0: unknown
 * (queries never seen by the host).  mt_seed32 (parity mode): the draws are not in d_draws yet -- the device parses
 * np.random.seed(*mt_seed32 + i)'s stream of query i into it first (mcq_mt.hpp).  d_prefix_ready: the cost prefix
 * has been computed by the host and the result rows are zero already -- no prep kernel (host entries). */
/* parity mode by state blocks (mcq_mt_blocks.hpp): where the chunk's arrays lie (device pointers) */
struct McqMtbLaunch {
    const uint32_t *d_blk_off, *d_grp_off;
    uint32_t max_blocks;
    uint32_t *d_raw;
    uint32_t *d_exits;
    void *d_entries;
    uint32_t *d_gword, *d_gits;
    void *d_gentry;
    uint32_t *d_ovf;
    uint32_t *d_part; /* the jumps' partial sums (mcq_mtb_part_words() words per query), or null: no jumps */
};
int mcq_run_slice(mcq_ctx *c, int mode, const mcq_query *d_q, uint32_t n, mcq_result *d_res, uint64_t seed,
                  uint64_t first_qid, uint64_t total_tasks, const uint8_t *d_draws, const uint64_t *d_off, hipStream_t s,
                  bool timed, uint64_t max_tasks = 0, uint32_t part = 0, uint32_t n_parts = 1,
                  const uint32_t *mt_seed32 = nullptr, const uint64_t *d_prefix_ready = nullptr,
                  const struct McqMtbLaunch *mtb = nullptr);
