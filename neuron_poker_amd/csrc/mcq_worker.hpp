// mcq_worker.hpp -- the per-shard worker thread of mcq_multi.cpp (pure host code, no HIP: tests/sanitize_cpu.sh
// drives submit / wait / join under ThreadSanitizer with a stand-in job).
#pragma once
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/mcq.h"

int mcq_fail(int code, const char *what, const char *detail); /* mcq_host.cpp: thread-local error text */

/* one worker thread per shard: runs the closure handed to it, then reports back */
struct McqWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, stop = false, done = true;
    int (*fn)(void *, int) = nullptr;
    void *arg = nullptr;
    int index = 0, rc = 0;
    std::string err;

    void start(int idx) {
        index = idx;
        th = std::thread([this] { loop(); });
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return has_job || stop; });
            if (stop) return;
            has_job = false;
            lk.unlock();
            int r;
            try {
                r = fn(arg, index);
            } catch (...) {
                r = mcq_fail(MCQ_EDEVICE, "mcq_multi: exception in a shard worker", nullptr);
            }
            const char *e = r ? mcq_last_error() : "";
            lk.lock();
            rc = r;
            err = e;
            done = true;
            cv.notify_all();
        }
    }
    void submit(int (*f)(void *, int), void *a) {
        std::lock_guard<std::mutex> lk(mu);
        fn = f;
        arg = a;
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
        return rc;
    }
    void join() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

