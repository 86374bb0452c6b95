// Floors of a one-launch query on this box: what an EMPTY kernel costs between its two timestamps and from the
// host's point of view, and what each ingredient of mcq_eval_direct_kernel adds (work records read from pinned host
// memory or from the kernel arguments, a 97 KB table image staged into LDS, a result row stored to pinned host
// memory behind a system-scope fence, the completion flag).  Build: hipcc --offload-arch=gfx950 -O3 -o
// tools/launch_floor tools/launch_floor.hip ; run tools/launch_floor on the GPU box (the binary travels with the snapshot).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

struct Args320 {
    uint32_t w[80];
};

enum { F_HOSTREAD = 1, F_KERNARG = 2, F_TABLES = 4, F_ROW = 8, F_FLAG = 16, F_ATOMIC = 32 };

__global__ __launch_bounds__(1024) void probe(uint32_t flags, const uint32_t *host_work, Args320 karg, const uint4 *g_tab,
                                              unsigned long long *host_row, uint32_t *done, volatile uint32_t *flag,
                                              uint32_t ticket, uint32_t *sink) {
    __shared__ uint4 tab[97 * 64]; /* 97 KB */
    __shared__ uint32_t s_w[80];
    uint32_t acc = 0;
    if ((flags & F_HOSTREAD) && threadIdx.x < 80) s_w[threadIdx.x] = host_work[threadIdx.x];
    if ((flags & F_KERNARG) && threadIdx.x < 80) s_w[threadIdx.x] = karg.w[threadIdx.x];
    if (flags & F_TABLES)
        for (uint32_t i = threadIdx.x; i < 97 * 64; i += 1024) tab[i] = g_tab[i];
    __syncthreads();
    if (flags & (F_HOSTREAD | F_KERNARG)) acc += s_w[threadIdx.x % 80];
    if (flags & F_TABLES) acc += tab[(threadIdx.x * 7 + acc) % (97 * 64)].x;
    if ((flags & F_ROW) && threadIdx.x < 13) host_row[threadIdx.x] = acc + threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (flags & F_ROW) __threadfence_system();
        if (flags & F_ATOMIC) {
            const uint32_t prev = atomicAdd(done, 1u);
            if (prev + 1u == gridDim.x) *done = 0;
        }
        if (flags & F_FLAG) {
            __threadfence_system();
            *flag = ticket;
        }
    }
    if (acc == 0xFFFFFFFFu) *sink = acc; /* keeps the loads */
}

// The same flag-only kernel with N KB of code that is never executed behind a run-time-false branch: does the size of a
// kernel's code cost launch latency?
template <int KB>
__global__ __launch_bounds__(1024) void probe_fat(uint32_t never, volatile uint32_t *flag, uint32_t ticket, uint32_t *sink) {
    if (never) { /* ~16 bytes of code per step */
        uint32_t x = threadIdx.x + never;
#pragma unroll
        for (int i = 0; i < KB * 64; i++) x = x * 1664525u + (uint32_t)i * 2654435761u + (x >> 7);
        *sink = x;
    }
    if (threadIdx.x == 0) {
        __threadfence_system();
        *flag = ticket;
    }
}

// The same flag-only kernel with N bytes of by-value arguments (only one word of them read): what does the size of
// the kernel-argument block cost a launch?
template <int WORDS>
struct ArgsN {
    uint32_t w[WORDS];
};
template <int WORDS>
__global__ __launch_bounds__(1024) void probe_args(ArgsN<WORDS> a, volatile uint32_t *flag, uint32_t ticket, uint32_t *sink) {
    if (a.w[WORDS - 1] == 0xFFFFFFFFu) *sink = a.w[0];
    if (threadIdx.x == 0) {
        __threadfence_system();
        *flag = ticket;
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    CHECK(hipSetDevice(0));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0));
    CHECK(hipEventCreate(&t1));
    uint32_t *h_work, *h_flag;
    unsigned long long *h_row;
    CHECK(hipHostMalloc((void **)&h_work, 4096, hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&h_row, 4096, hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&h_flag, 64, hipHostMallocMapped));
    memset(h_work, 1, 4096);
    uint4 *d_tab;
    uint32_t *d_done, *d_sink;
    CHECK(hipMalloc((void **)&d_tab, 97 * 1024));
    CHECK(hipMemset(d_tab, 0, 97 * 1024));
    CHECK(hipMalloc((void **)&d_done, 64));
    CHECK(hipMemset(d_done, 0, 64));
    CHECK(hipMalloc((void **)&d_sink, 64));
    Args320 ka;
    memset(&ka, 2, sizeof ka);
    struct Case {
        const char *name;
        uint32_t flags;
    } cases[] = {
        {"empty kernel", 0},
        {"+ completion flag to pinned host memory", F_FLAG},
        {"+ 320 B of work read from pinned host memory", F_FLAG | F_HOSTREAD},
        {"+ 320 B of work in the kernel arguments", F_FLAG | F_KERNARG},
        {"+ 97 KB table image -> LDS", F_FLAG | F_TABLES},
        {"+ 104 B row to pinned host memory, system fence", F_FLAG | F_ROW},
        {"+ device atomic", F_FLAG | F_ATOMIC},
        {"all (work from host memory)", F_FLAG | F_HOSTREAD | F_TABLES | F_ROW | F_ATOMIC},
        {"all (work in kernel arguments)", F_FLAG | F_KERNARG | F_TABLES | F_ROW | F_ATOMIC},
    };
    const int reps = 2000;
    uint32_t ticket = 0;
    printf("%-52s %10s %10s %10s %10s\n", "one block of 1024 threads, medians of 2000", "kernel us", "launch us", "to flag us",
           "to sync us");
    for (const Case &c : cases) {
        std::vector<double> k, l, f, y;
        for (int mode = 0; mode < 2; mode++) { /* 0: wait by polling the flag (if the kernel writes it); 1: hipStreamSynchronize */
            for (int r = 0; r < reps; r++) {
                ++ticket;
                const double a = now_us();
                hipExtLaunchKernelGGL(probe, dim3(1), dim3(1024), 0, s, t0, t1, 0, c.flags, (const uint32_t *)h_work, ka,
                                      (const uint4 *)d_tab, h_row, d_done, (volatile uint32_t *)h_flag, ticket, d_sink);
                const double b = now_us();
                if (mode == 0 && (c.flags & F_FLAG)) {
                    while (*(volatile uint32_t *)h_flag != ticket) {
                    }
                    f.push_back(now_us() - a);
                    CHECK(hipStreamSynchronize(s));
                } else {
                    CHECK(hipStreamSynchronize(s));
                    if (mode == 1) y.push_back(now_us() - a);
                }
                if (mode == 0) {
                    l.push_back(b - a);
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, t0, t1));
                    k.push_back(ms * 1000.0);
                }
            }
        }
        auto med = [](std::vector<double> &v) {
            if (v.empty()) return 0.0;
            std::sort(v.begin(), v.end());
            return v[v.size() / 2];
        };
        printf("%-52s %10.2f %10.2f %10.2f %10.2f\n", c.name, med(k), med(l), med(f), med(y));
    }
    /* the same empty kernel launched plainly (no timestamps) */
    {
        std::vector<double> f;
        for (int r = 0; r < reps; r++) {
            ++ticket;
            const double a = now_us();
            hipLaunchKernelGGL(probe, dim3(1), dim3(1024), 0, s, (uint32_t)F_FLAG, (const uint32_t *)h_work, ka, (const uint4 *)d_tab,
                               h_row, d_done, (volatile uint32_t *)h_flag, ticket, d_sink);
            while (*(volatile uint32_t *)h_flag != ticket) {
            }
            f.push_back(now_us() - a);
            CHECK(hipStreamSynchronize(s));
        }
        std::sort(f.begin(), f.end());
        printf("%-52s %10s %10s %10.2f\n", "flag only, launched without timestamp events", "-", "-", f[f.size() / 2]);
    }
    /* code size: flag-only kernels with 1, 32, 128 KB of dead code */
    {
        auto run = [&](auto kern, const char *name) {
            std::vector<double> f;
            for (int r = 0; r < reps; r++) {
                ++ticket;
                const double a = now_us();
                hipLaunchKernelGGL(kern, dim3(1), dim3(1024), 0, s, 0u, (volatile uint32_t *)h_flag, ticket, d_sink);
                while (*(volatile uint32_t *)h_flag != ticket) {
                }
                f.push_back(now_us() - a);
                (void)hipStreamSynchronize(s);
            }
            std::sort(f.begin(), f.end());
            printf("%-52s %10s %10s %10.2f\n", name, "-", "-", f[f.size() / 2]);
        };
        auto run_args = [&](auto kern, auto args, const char *name) {
            std::vector<double> f;
            memset(&args, 3, sizeof args);
            for (int r = 0; r < reps; r++) {
                ++ticket;
                const double a = now_us();
                hipLaunchKernelGGL(kern, dim3(1), dim3(1024), 0, s, args, (volatile uint32_t *)h_flag, ticket, d_sink);
                while (*(volatile uint32_t *)h_flag != ticket) {
                }
                f.push_back(now_us() - a);
                (void)hipStreamSynchronize(s);
            }
            std::sort(f.begin(), f.end());
            printf("%-52s %10s %10s %10.2f\n", name, "-", "-", f[f.size() / 2]);
        };
        run_args(probe_args<4>, ArgsN<4>(), "flag only, 16 B of arguments by value");
        run_args(probe_args<80>, ArgsN<80>(), "flag only, 320 B of arguments by value");
        run_args(probe_args<640>, ArgsN<640>(), "flag only, 2560 B of arguments by value");
        run(probe_fat<1>, "flag only, 1 KB of dead code");
        run(probe_fat<32>, "flag only, 32 KB of dead code");
        run(probe_fat<128>, "flag only, 128 KB of dead code");
    }
    return 0;
}
