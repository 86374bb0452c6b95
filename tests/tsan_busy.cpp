// tsan_busy.cpp -- TEST ONLY (tests/test_abi.py, tests/sanitize_cpu.sh): the in-flight guard of the C ABI's entry points
// (csrc/mcq_busy.hpp, MCQ_ENTER in csrc/mcq_ctx.hpp) hammered by eight threads: whoever gets in owns the context -- a
// plain (non-atomic) counter stands in for its staging buffers, so ThreadSanitizer sees any second entrant -- and
// everybody else must be turned away.
#include <stdio.h>

#include <thread>
#include <vector>

#include "../neuron_poker_amd/csrc/mcq_busy.hpp"

int main() {
    McqBusyFlag flag;
    long owned = 0; /* touched only inside the scope */
    std::atomic<long> in(0), refused(0);
    std::vector<std::thread> ts;
    for (int t = 0; t < 8; t++)
        ts.emplace_back([&] {
            for (int i = 0; i < 200000; i++) {
                McqBusyScope s(&flag);
                if (s.ok) {
                    owned++;
                    in++;
                } else {
                    refused++;
                }
            }
        });
    for (auto &t : ts) t.join();
    McqBusyScope last(&flag); /* free again afterwards */
    if (!last.ok || owned != in.load() || in.load() + refused.load() != 8L * 200000L || in.load() == 0) {
        printf("busy guard broken: owned %ld, entered %ld, refused %ld\n", owned, in.load(), refused.load());
        return 1;
    }
    printf("busy guard ok: %ld entered, %ld turned away\n", in.load(), refused.load());
    return 0;
}
