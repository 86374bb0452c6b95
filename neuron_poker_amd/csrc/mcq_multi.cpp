// mcq_multi.cpp -- one process, several GPUs of one node: the batch is partitioned over SHARDS (one
// engine context, one host worker thread and one stream each), every shard fills its part of a zero-initialised
// [n, 13] uint64 tally matrix resident on its device, and ONE ncclAllReduce(ncclUint64, ncclSum) over xGMI leaves
// the complete matrix on every device.  The communicators come from ncclCommInitAll and belong to the mcq_multi
// object.  There is no exchange during the computation: every (query, iteration) is independent and the RNG
// streams are keyed by (seed, query id, stream), so the tallies are bit-identical to a single-context call
// whatever the partition.
//
// Several shards may sit on ONE device (rehearsal of an 8-way partition on a single-GPU box, or oversubscription):
// their matrices are added on the device before the all-reduce, which then runs over the distinct devices only.
//
// RCCL is bound at run time (dlopen) when the first mcq_multi is created: single-GPU users never map the 570 MB
// library, and a process that already holds an RCCL (PyTorch bundles one next to its HIP runtime) reuses that copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mcq_ctx.hpp"
#include "mcq_device.hpp"
#include "mcq_internal.hpp"
#include "mcq_worker.hpp"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string path, error;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;

bool rccl_bind(Rccl &r, void *h, const char *path) {
#define MCQ_SYM(name)                                                      \
    r.name = reinterpret_cast<decltype(r.name)>(dlsym(h, "nccl" #name));   \
    if (!r.name) {                                                         \
        r.error = std::string("symbol nccl" #name " missing in ") + path;  \
        return false;                                                      \
    }
    MCQ_SYM(GetVersion) MCQ_SYM(CommInitAll) MCQ_SYM(CommDestroy) MCQ_SYM(AllReduce) MCQ_SYM(GroupStart)
    MCQ_SYM(GroupEnd) MCQ_SYM(GetErrorString)
#undef MCQ_SYM
    r.handle = h;
    r.path = path;
    return true;
}

/* Bind RCCL once per process.  Order: $MCQ_RCCL_LIBRARY; an RCCL the process has already mapped; the copy that
 * sits beside the HIP runtime in use (the one built against it); the loader's search path. */
const Rccl *rccl_get(std::string *error /* out: why not, copied while the lock is held */) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    struct OnExit {
        std::string *e;
        ~OnExit() { if (e && !g_rccl.handle) *e = g_rccl.error; }
    } on_exit_{error};
    std::vector<std::string> cand;
    if (const char *e = getenv("MCQ_RCCL_LIBRARY")) cand.push_back(e);
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        if (void *h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) {
            if (rccl_bind(g_rccl, h, name)) return &g_rccl;
            dlclose(h);
        }
    }
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            cand.push_back(dir + "librccl.so.1");
            cand.push_back(dir + "librccl.so");
        }
    }
    cand.push_back("librccl.so.1");
    cand.push_back("librccl.so");
    std::string tried;
    for (const std::string &p : cand) {
        void *h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            tried += p + " (" + (dlerror() ? "not loadable" : "?") + "); ";
            continue;
        }
        if (rccl_bind(g_rccl, h, p.c_str())) return &g_rccl;
        dlclose(h);
    }
    g_rccl.error = "RCCL not found: " + tried;
    return nullptr;
}

struct Shard {
    mcq_ctx *ctx = nullptr;
    int device = 0;
    int group = 0;       /* index of this shard's device among the distinct devices */
    bool primary = false; /* first shard on its device: holds the device's sum and takes part in the all-reduce */
    DevBuf tally;        /* [n, 13] uint64 on the shard's device */
    mcq_result *matrix = nullptr; /* the matrix of the call in flight: tally.p or the caller's buffer on this device */
    hipEvent_t launched = nullptr;
    McqWorker worker;
};

struct Call { /* arguments of the call in flight, read by the workers */
    struct mcq_multi *m;
    const mcq_query *q; /* host buffers (mcq_multi_eval_batch) ... */
    size_t n;
    uint64_t seed, first_qid;
    int partition;
    const void *const *d_q = nullptr; /* ... or one device pointer pair per shard (mcq_multi_eval_batch_device) */
    void *const *d_res = nullptr;
};

}  // namespace

struct mcq_multi {
    McqBusyFlag busy; /* one call in flight per object (the call state below is per call) */
    std::vector<Shard> shards;
    std::vector<int> devices;       /* distinct devices, in order of first appearance */
    std::vector<int> primary_shard; /* per distinct device */
    std::vector<ncclComm_t> comms;  /* per distinct device (ncclCommInitAll) */
    const Rccl *rccl = nullptr;
    int rccl_version = 0;
    PinBuf h_out;
    hipEvent_t ar0 = nullptr, ar1 = nullptr; /* around the all-reduce on device group 0 */
    float last_ms[3] = {0.f, 0.f, 0.f};      /* slowest shard's kernel, all-reduce, whole call (wall) */
    int last_partition = 0;
    Call call;
};

namespace {

#define NCCL_TRY(m, expr)                                                                        \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) return mcq_fail(MCQ_EDEVICE, #expr, (m)->rccl->GetErrorString(r_)); \
    } while (0)

size_t shard_lo(size_t n, size_t s, size_t k) { return n * s / k; } /* block distribution, as sharding.shard_bounds */

/* what one shard enqueues for the call in flight: zero its matrix, upload its queries, prep + evaluation kernels */
int shard_job(void *arg, int s) {
    Call &cl = *static_cast<Call *>(arg);
    mcq_multi *m = cl.m;
    Shard &sh = m->shards[(size_t)s];
    mcq_ctx *c = sh.ctx;
    const size_t k = m->shards.size(), n = cl.n;
    McqDeviceScope dev(sh.device);
    HIP_TRY(dev.err);
    /* the shard's matrix: its own buffer, or the caller's on this shard's device */
    mcq_result *tally = cl.d_res ? static_cast<mcq_result *>(cl.d_res[s]) : nullptr;
    if (!tally) {
        HIP_TRY(sh.tally.reserve(n * sizeof(mcq_result)));
        tally = (mcq_result *)sh.tally.p;
    }
    sh.matrix = tally;
    HIP_TRY(hipMemsetAsync(tally, 0, n * sizeof(mcq_result), c->stream));
    size_t lo = 0, hi = n;
    uint32_t part = 0, n_parts = 1;
    if (cl.partition == MCQ_PARTITION_QUERIES) {
        lo = shard_lo(n, (size_t)s, k);
        hi = shard_lo(n, (size_t)s + 1, k);
    } else {
        part = (uint32_t)s;
        n_parts = (uint32_t)k;
    }
    const size_t cnt = hi - lo;
    c->last_ms = 0.f;
    if (cnt && cl.d_q) { /* queries resident in this shard's HBM: priced by the prep kernel (the host has not seen them) */
        int rc = mcq_run_slice(c, MCQ_MODE_PHILOX, static_cast<const mcq_query *>(cl.d_q[s]), (uint32_t)cnt, tally + lo, cl.seed,
                               cl.first_qid + lo, 0, nullptr, nullptr, c->stream, true, 0, part, n_parts);
        if (rc) return rc;
    } else if (cnt) {
        uint64_t total_tasks = 0, max_tasks = 0;
        for (size_t i = lo; i < hi; i++) {
            const McqPart pt = mcq_part(mcq_tasks_of(cl.q[i]), cl.q[i].runs, part, n_parts);
            const uint64_t t = pt.t_hi - pt.t_lo;
            total_tasks += t;
            if (t > max_tasks) max_tasks = t;
        }
        if (total_tasks == 0) total_tasks = 1; /* 0 means "unknown" to the launcher */
        HIP_TRY(c->h_q.reserve(cnt * sizeof(mcq_query)));
        HIP_TRY(c->d_q.reserve(cnt * sizeof(mcq_query)));
        memcpy(c->h_q.p, cl.q + lo, cnt * sizeof(mcq_query));
        HIP_TRY(hipMemcpyAsync(c->d_q.p, c->h_q.p, cnt * sizeof(mcq_query), hipMemcpyHostToDevice, c->stream));
        int rc = mcq_run_slice(c, MCQ_MODE_PHILOX, (const mcq_query *)c->d_q.p, (uint32_t)cnt, tally + lo,
                               cl.seed, cl.first_qid + lo, total_tasks, nullptr, nullptr, c->stream, true, max_tasks, part,
                               n_parts);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(sh.launched, c->stream));
    return MCQ_OK;
}

int multi_eval(mcq_multi *m, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_qid, int partition,
               mcq_result *out, const void *const *d_q = nullptr, void *const *d_res = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    const size_t k = m->shards.size();
    if (q) {
        int rc = mcq_validate_queries(q, n);
        if (rc) return rc;
        uint64_t total_tasks = 0;
        for (size_t i = 0; i < n; i++) total_tasks += mcq_tasks_of(q[i]);
        if (total_tasks > 0xfffffff0ull) return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch: too many iterations in one call");
    }
    if (partition == MCQ_PARTITION_AUTO) /* block-distribute the queries when there are >= 256 per shard */
        partition = n >= k * 256u ? MCQ_PARTITION_QUERIES : MCQ_PARTITION_ITERATIONS;
    m->last_partition = partition;
    m->call = Call{m, q, n, seed, first_qid, partition, d_q, d_res};

    /* 1. every shard enqueues its work from its own thread (shard 0: this thread) */
    for (size_t s = 1; s < k; s++) m->shards[s].worker.submit(shard_job, &m->call);
    int rc0 = shard_job(&m->call, 0);
    std::string err0 = rc0 ? mcq_last_error() : "";
    for (size_t s = 1; s < k; s++) {
        const int r = m->shards[s].worker.wait();
        if (r && !rc0) {
            rc0 = r;
            err0 = m->shards[s].worker.err;
        }
    }
    if (rc0) { /* drain what was enqueued, then report the first failure */
        for (Shard &sh : m->shards) {
            McqDeviceScope dev(sh.device);
            (void)hipStreamSynchronize(sh.ctx->stream);
        }
        return mcq_fail(rc0, err0.c_str());
    }

    /* 2. shards sharing a device: add their matrices into the device's primary shard */
    const uint64_t words = (uint64_t)n * 13u;
    for (Shard &sh : m->shards) {
        if (sh.primary) continue;
        Shard &pr = m->shards[(size_t)m->primary_shard[(size_t)sh.group]];
        McqDeviceScope dev(pr.device);
        HIP_TRY(dev.err);
        HIP_TRY(hipStreamWaitEvent(pr.ctx->stream, sh.launched, 0));
        HIP_TRY(mcq_launch_add_u64((uint64_t *)pr.matrix, (const uint64_t *)sh.matrix, words, pr.ctx->stream));
    }

    /* 3. the path's one collective: integer sum of the tally matrices over the distinct devices */
    {
        Shard &p0 = m->shards[(size_t)m->primary_shard[0]];
        McqDeviceScope dev(p0.device);
        HIP_TRY(dev.err);
        HIP_TRY(hipEventRecord(m->ar0, p0.ctx->stream));
    }
    NCCL_TRY(m, m->rccl->GroupStart());
    for (size_t g = 0; g < m->devices.size(); g++) {
        Shard &pr = m->shards[(size_t)m->primary_shard[g]];
        ncclResult_t r = m->rccl->AllReduce(pr.matrix, pr.matrix, (size_t)words, ncclUint64, ncclSum, m->comms[g],
                                            pr.ctx->stream);
        if (r != ncclSuccess) {
            (void)m->rccl->GroupEnd();
            return mcq_fail(MCQ_EDEVICE, "ncclAllReduce", m->rccl->GetErrorString(r));
        }
    }
    NCCL_TRY(m, m->rccl->GroupEnd());

    /* 4. results to the host from device group 0 (every device holds the full matrix) */
    {
        Shard &p0 = m->shards[(size_t)m->primary_shard[0]];
        McqDeviceScope dev(p0.device);
        HIP_TRY(dev.err);
        HIP_TRY(hipEventRecord(m->ar1, p0.ctx->stream));
        if (out) {
            HIP_TRY(m->h_out.reserve(n * sizeof(mcq_result)));
            HIP_TRY(hipMemcpyAsync(m->h_out.p, p0.matrix, n * sizeof(mcq_result), hipMemcpyDeviceToHost, p0.ctx->stream));
        }
    }
    if (d_res) /* device entry: every shard's buffer gets the complete matrix -- the primary's own copy on a shared device */
        for (Shard &sh : m->shards) {
            if (sh.primary) continue;
            Shard &pr = m->shards[(size_t)m->primary_shard[(size_t)sh.group]];
            McqDeviceScope dev(pr.device);
            HIP_TRY(dev.err);
            HIP_TRY(hipMemcpyAsync(sh.matrix, pr.matrix, n * sizeof(mcq_result), hipMemcpyDeviceToDevice, pr.ctx->stream));
        }
    for (size_t g = 0; g < m->devices.size(); g++) {
        Shard &pr = m->shards[(size_t)m->primary_shard[g]];
        McqDeviceScope dev(pr.device);
        HIP_TRY(dev.err);
        HIP_TRY(hipStreamSynchronize(pr.ctx->stream));
    }
    if (out) memcpy(out, m->h_out.p, n * sizeof(mcq_result));

    float kmax = 0.f;
    for (Shard &sh : m->shards) {
        float ms = 0.f;
        if (mcq_kernel_times(sh.ctx, &ms, 1) == 1 && ms > kmax) kmax = ms;
    }
    m->last_ms[0] = kmax;
    {
        Shard &p0 = m->shards[(size_t)m->primary_shard[0]];
        McqDeviceScope dev(p0.device);
        if (hipEventElapsedTime(&m->last_ms[1], m->ar0, m->ar1) != hipSuccess) m->last_ms[1] = 0.f;
    }
    m->last_ms[2] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MCQ_OK;
}

}  // namespace

extern "C" {

void mcq_multi_destroy(mcq_multi *m) {
    if (!m) return;
    for (Shard &sh : m->shards) sh.worker.join();
    if (m->rccl)
        for (ncclComm_t cm : m->comms)
            if (cm) (void)m->rccl->CommDestroy(cm);
    for (Shard &sh : m->shards) {
        McqDeviceScope dev(sh.device);
        if (sh.ctx && sh.ctx->stream) (void)hipStreamSynchronize(sh.ctx->stream);
        sh.tally.release();
        if (sh.launched) (void)hipEventDestroy(sh.launched);
        mcq_destroy(sh.ctx);
    }
    if (!m->shards.empty()) {
        McqDeviceScope dev(m->shards[0].device);
        m->h_out.release();
        if (m->ar0) (void)hipEventDestroy(m->ar0);
        if (m->ar1) (void)hipEventDestroy(m->ar1);
    }
    delete m;
}

mcq_multi *mcq_multi_create(const int *devices, int n_shards, int flags) {
    try {
        if (flags != 0) { mcq_fail(MCQ_EINVAL, "mcq_multi_create: flags must be 0"); return nullptr; }
        if (n_shards < 1 || n_shards > 64) { mcq_fail(MCQ_EINVAL, "mcq_multi_create: n_shards must be in [1, 64]"); return nullptr; }
        int n_dev = 0;
        hipError_t e = hipGetDeviceCount(&n_dev);
        if (e != hipSuccess || n_dev <= 0) {
            mcq_fail(MCQ_EDEVICE, "mcq_multi_create: no HIP device available (this library has no CPU path)",
                     e != hipSuccess ? hipGetErrorString(e) : nullptr);
            return nullptr;
        }
        /* whatever fails from here on -- an error return or an exception (std::system_error from a worker thread,
         * bad_alloc) -- the guard takes the half-built object apart: contexts, streams, events, communicators, workers */
        struct Guard {
            mcq_multi *m;
            ~Guard() { if (m) mcq_multi_destroy(m); }
        } guard{new mcq_multi()};
        mcq_multi *m = guard.m;
        m->shards = std::vector<Shard>((size_t)n_shards);
        for (int s = 0; s < n_shards; s++) {
            const int d = devices ? devices[s] : s;
            if (d < 0 || d >= n_dev) {
                mcq_fail(MCQ_EINVAL, "mcq_multi_create: device ordinal out of range (devices = NULL means shard s on device s)");
                return nullptr;
            }
            Shard &sh = m->shards[(size_t)s];
            sh.device = d;
            size_t g = 0;
            while (g < m->devices.size() && m->devices[g] != d) g++;
            if (g == m->devices.size()) {
                m->devices.push_back(d);
                m->primary_shard.push_back(s);
                sh.primary = true;
            }
            sh.group = (int)g;
        }
        for (int s = 0; s < n_shards; s++) {
            Shard &sh = m->shards[(size_t)s];
            sh.ctx = mcq_create(sh.device, 0);
            if (!sh.ctx) return nullptr;
            sh.ctx->timing = true; /* mcq_multi_times reports the shards' kernel times; these are bulk launches */
            McqDeviceScope dev(sh.device);
            hipError_t e2 = hipEventCreateWithFlags(&sh.launched, hipEventDisableTiming);
            if (e2 == hipSuccess && s == m->primary_shard[0]) {
                e2 = hipEventCreate(&m->ar0);
                if (e2 == hipSuccess) e2 = hipEventCreate(&m->ar1);
            }
            if (e2 != hipSuccess) {
                mcq_fail(MCQ_EDEVICE, "mcq_multi_create: hipEventCreate", hipGetErrorString(e2));
                return nullptr;
            }
        }
        std::string rccl_error;
        m->rccl = rccl_get(&rccl_error);
        if (!m->rccl) {
            mcq_fail(MCQ_EDEVICE, "mcq_multi_create", rccl_error.c_str());
            return nullptr;
        }
        (void)m->rccl->GetVersion(&m->rccl_version);
        m->comms.assign(m->devices.size(), nullptr);
        ncclResult_t r = m->rccl->CommInitAll(m->comms.data(), (int)m->devices.size(), m->devices.data());
        if (r != ncclSuccess) {
            mcq_fail(MCQ_EDEVICE, "ncclCommInitAll", m->rccl->GetErrorString(r));
            return nullptr;
        }
        for (int s = 1; s < n_shards; s++) m->shards[(size_t)s].worker.start(s);
        guard.m = nullptr;
        return m;
    } catch (const std::exception &ex) {
        mcq_fail(MCQ_ENOMEM, "mcq_multi_create", ex.what());
        return nullptr;
    } catch (...) {
        mcq_fail(MCQ_EDEVICE, "mcq_multi_create: unexpected exception");
        return nullptr;
    }
}

int mcq_multi_eval_batch(mcq_multi *m, const mcq_query *q, size_t n, uint64_t seed, uint64_t first_query_id, int partition,
                         mcq_result *out) {
    ABI_GUARD_BEGIN
    if (!m) return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch: null object");
    if (partition != MCQ_PARTITION_AUTO && partition != MCQ_PARTITION_QUERIES && partition != MCQ_PARTITION_ITERATIONS)
        return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch: bad partition");
    if (n == 0) return MCQ_OK;
    if (!q || !out) return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch: null buffer");
    if (n > 0x7fffffffu) return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch: n too large");
    MCQ_ENTER(m, "mcq_multi_eval_batch");
    return multi_eval(m, q, n, seed, first_query_id, partition, out);
    ABI_GUARD_END("mcq_multi_eval_batch")
}

int mcq_multi_eval_batch_device(mcq_multi *m, const void *const *d_queries, size_t n, uint64_t seed, uint64_t first_query_id,
                                int partition, void *const *d_results) {
    ABI_GUARD_BEGIN
    if (!m) return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch_device: null object");
    if (partition != MCQ_PARTITION_AUTO && partition != MCQ_PARTITION_QUERIES && partition != MCQ_PARTITION_ITERATIONS)
        return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch_device: bad partition");
    if (n == 0) return MCQ_OK;
    if (!d_queries || !d_results) return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch_device: null buffer");
    if (n > 0x7fffffffu) return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch_device: n too large");
    const size_t k = m->shards.size();
    const int eff = partition != MCQ_PARTITION_AUTO ? partition : (n >= k * 256u ? MCQ_PARTITION_QUERIES : MCQ_PARTITION_ITERATIONS);
    for (size_t s = 0; s < k; s++) {
        const bool has_work = eff != MCQ_PARTITION_QUERIES || shard_lo(n, s + 1, k) > shard_lo(n, s, k);
        if (!d_results[s] || (has_work && !d_queries[s]))
            return mcq_fail(MCQ_EINVAL, "mcq_multi_eval_batch_device: null device pointer of a shard");
    }
    MCQ_ENTER(m, "mcq_multi_eval_batch_device");
    return multi_eval(m, nullptr, n, seed, first_query_id, partition, nullptr, d_queries, d_results);
    ABI_GUARD_END("mcq_multi_eval_batch_device")
}

int mcq_multi_info(const mcq_multi *m, int info[4]) {
    if (!m || !info) return mcq_fail(MCQ_EINVAL, "mcq_multi_info: null argument");
    info[0] = (int)m->shards.size();
    info[1] = (int)m->devices.size();
    info[2] = m->rccl_version;
    info[3] = m->last_partition;
    return MCQ_OK;
}

int mcq_multi_times(const mcq_multi *m, float ms[3]) {
    if (!m || !ms) return mcq_fail(MCQ_EINVAL, "mcq_multi_times: null argument");
    ms[0] = m->last_ms[0];
    ms[1] = m->last_ms[1];
    ms[2] = m->last_ms[2];
    return MCQ_OK;
}

int mcq_multi_set_dealing_law(mcq_multi *m, int law) {
    if (!m) return mcq_fail(MCQ_EINVAL, "mcq_multi_set_dealing_law: null object");
    MCQ_ENTER(m, "mcq_multi_set_dealing_law");
    for (Shard &sh : m->shards) {
        const int rc = mcq_set_dealing_law(sh.ctx, law);
        if (rc) return rc;
    }
    return MCQ_OK;
}

}  // extern "C"
