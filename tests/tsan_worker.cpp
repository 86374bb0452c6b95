// tsan_worker.cpp -- TEST ONLY (tests/sanitize_cpu.sh): the shard worker of mcq_multi.cpp (csrc/mcq_worker.hpp) driven
// the way mcq_multi_eval_batch drives it -- submit to every worker, run shard 0 on the calling thread, wait for all,
// many calls in a row, then join -- under ThreadSanitizer, with a stand-in job that touches shared call state.
#include <stdio.h>

#include <atomic>
#include <string>
#include <vector>

#include "../neuron_poker_amd/csrc/mcq_worker.hpp"

static thread_local std::string g_err;
int mcq_fail(int code, const char *what, const char *detail) {
    g_err = what;
    if (detail) g_err += detail;
    return code;
}
extern "C" const char *mcq_last_error(void) { return g_err.c_str(); }

struct Call {
    std::vector<long> out;
    long seed;
};
static int job(void *arg, int s) {
    Call &c = *static_cast<Call *>(arg);
    long v = c.seed;
    for (int i = 0; i < 1000; i++) v = v * 31 + s;
    c.out[(size_t)s] = v;
    if (c.seed % 17 == 3 && s == 2) return mcq_fail(MCQ_EDEVICE, "stand-in failure", nullptr);
    return 0;
}

int main() {
    const int k = 8;
    std::vector<McqWorker> w((size_t)k);
    for (int s = 1; s < k; s++) w[(size_t)s].start(s);
    Call c;
    c.out.assign((size_t)k, 0);
    long failures = 0;
    for (long call = 0; call < 2000; call++) {
        c.seed = call;
        for (int s = 1; s < k; s++) w[(size_t)s].submit(job, &c);
        int rc = job(&c, 0);
        for (int s = 1; s < k; s++) {
            const int r = w[(size_t)s].wait();
            if (r && !rc) rc = r;
        }
        if (rc) {
            failures++;
            if (w[2].err != "stand-in failure") { printf("error text lost: '%s'\n", w[2].err.c_str()); return 1; }
        }
        for (int s = 0; s < k; s++) {
            long v = call;
            for (int i = 0; i < 1000; i++) v = v * 31 + s;
            if (c.out[(size_t)s] != v) { printf("shard %d of call %ld wrong\n", s, call); return 1; }
        }
    }
    for (int s = 1; s < k; s++) w[(size_t)s].join();
    printf("worker stress ok: 2000 calls x %d shards, %ld reported failures\n", k, failures);
    return failures == 118 ? 0 : 1; /* seeds 3, 20, 37, ... */
}
