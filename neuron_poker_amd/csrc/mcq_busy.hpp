// mcq_busy.hpp -- "one call in flight per context" (include/mcq.h; SURVEY 8b, threading) made checkable: every public
// entry point that takes a context enters through McqBusyScope; a second entrant -- another thread using the same
// context while a call is running on it -- gets MCQ_EINVAL "context busy" instead of racing on the context's pinned
// staging buffers and completion flag.  Plain C++ (no HIP), so that the CPU test suite can hammer it under
// ThreadSanitizer (tests/tsan_busy.cpp).
#pragma once
#include <atomic>

struct McqBusyFlag {
    std::atomic<int> in_flight{0};
};

struct McqBusyScope {
    McqBusyFlag *f;
    bool ok;
    explicit McqBusyScope(McqBusyFlag *flag) : f(flag), ok(true) {
        if (f) {
            int expected = 0;
            ok = f->in_flight.compare_exchange_strong(expected, 1, std::memory_order_acquire, std::memory_order_relaxed);
            if (!ok) f = nullptr; /* not ours to clear */
        }
    }
    ~McqBusyScope() {
        if (f) f->in_flight.store(0, std::memory_order_release);
    }
    McqBusyScope(const McqBusyScope &) = delete;
    McqBusyScope &operator=(const McqBusyScope &) = delete;
};

#define MCQ_BUSY_MESSAGE "context busy: one call in flight per context (create one context per thread)"
