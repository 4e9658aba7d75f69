// How fast can two workgroups on DIFFERENT compute units exchange a word?  (fps_coop.hip / ffps_fly.hip pay ~1.5-2 us
// per pick for it through agent-scope atomics: the store is written through to memory, the polled load bypasses the L2.)
//
// Pairs of workgroups ping-pong a counter N times; thread 0 of each does the exchange.  Modes:
//   0  agent-scope atomic store + agent-scope atomic load        (what the samplers use today)
//   1  plain vector store (+ s_waitcnt) + SCALAR load with glc   (scalar loads have their own cache; glc goes to the L2)
//   2  agent-scope atomic store + scalar load with glc
//   3  plain vector store + vector load with sc0                 (expected to time out: sc0 loads may hit the CU's L1)
// Pair placement: "same" = blocks b and b + 8 (same XCD if block b runs on XCD b % 8), "next" = blocks 2p and 2p + 1
// (neighbouring XCDs).  Every poll is BOUNDED: a pair that never sees its partner's value gives up and is reported as
// a timeout -- the program cannot hang.  Prints ns per round trip (two one-way messages) and the XCC ids the pairs saw.
// Build: hipcc --offload-arch=gfx950 -O3 xcd_pingpong.hip -o xcd_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr unsigned kMaxPoll = 1u << 18;

template <int MODE>
__device__ __forceinline__ void put(unsigned long long *p, unsigned long long v) {
    if (MODE == 0 || MODE == 2) {
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *(volatile unsigned long long *)p = v;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
template <int MODE>
__device__ __forceinline__ unsigned long long get(const unsigned long long *p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 3) {
        unsigned long long v;
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// words: [pair][2]; res: [pair][4] = {cycles, timeouts, xcc of A, xcc of B}
template <int MODE>
__global__ void pingpong(unsigned long long *words, unsigned long long *res, int n, int same_xcd) {
    if (threadIdx.x != 0) return;
    int pair, side;
    if (same_xcd) { pair = (blockIdx.x & 7) + 8 * (blockIdx.x >> 4); side = (blockIdx.x >> 3) & 1; }
    else { pair = blockIdx.x >> 1; side = blockIdx.x & 1; }
    unsigned long long *mine = words + pair * 2 + side, *theirs = words + pair * 2 + (1 - side);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long fails = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();   // shader clock
    const unsigned long long w0 = wall_clock64();
    for (int i = 1; i <= n && fails == 0; ++i) {
        if (side == 0) put<MODE>(mine, (unsigned long long)i);
        unsigned spins = 0;
        while (get<MODE>(theirs) < (unsigned long long)i) {
            if (++spins > kMaxPoll) { fails = i; break; }
        }
        if (side == 1 && fails == 0) put<MODE>(mine, (unsigned long long)i);
    }
    const unsigned long long w1 = wall_clock64();
    (void)t0;
    res[(pair * 2 + side) * 4 + 0] = w1 - w0;
    res[(pair * 2 + side) * 4 + 1] = fails;
    res[(pair * 2 + side) * 4 + 2] = xcc & 0xF;
    // a side that gave up leaves a huge value so that its partner does not spin to its own bound for every round
    if (fails) put<0>(mine, ~0ull >> 1);
}

template <int MODE>
static void run(const char *name, int pairs, int n, int same) {
    unsigned long long *words, *res;
    CHECK(hipMalloc(&words, pairs * 2 * 8)); CHECK(hipMalloc(&res, pairs * 2 * 4 * 8));
    CHECK(hipMemset(words, 0, pairs * 2 * 8)); CHECK(hipMemset(res, 0, pairs * 2 * 4 * 8));
    hipLaunchKernelGGL(pingpong<MODE>, dim3(pairs * 2), dim3(64), 0, 0, words, res, n, same);
    CHECK(hipDeviceSynchronize());
    unsigned long long *h = (unsigned long long *)malloc(pairs * 2 * 4 * 8);
    CHECK(hipMemcpy(h, res, pairs * 2 * 4 * 8, hipMemcpyDeviceToHost));
    int dev = 0, wall_khz = 0;
    CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
    double sum = 0; int ok = 0, to = 0, samex = 0;
    for (int p = 0; p < pairs; ++p) {
        const unsigned long long *a = h + (p * 2) * 4, *b = h + (p * 2 + 1) * 4;
        if (a[1] || b[1]) { ++to; continue; }
        sum += (double)a[0] / n; ++ok;
        samex += a[2] == b[2];
    }
    printf("%-46s %-5s pairs %3d: ok %3d timeouts %3d  same-XCC pairs %3d  %8.1f ns per round trip\n", name, same ? "same" : "next", pairs, ok, to,
           samex, ok ? sum / ok / (wall_khz * 1e-6) : -1.0);
    free(h); CHECK(hipFree(words)); CHECK(hipFree(res));
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2000, pairs = argc > 2 ? atoi(argv[2]) : 64;
    for (int same = 1; same >= 0; --same) {
        run<0>("agent-scope store + agent-scope load", pairs, n, same);
        run<1>("plain store + scalar load glc", pairs, n, same);
        run<2>("agent-scope store + scalar load glc", pairs, n, same);
        run<3>("plain store + vector load sc0", pairs, n > 200 ? 200 : n, same);
    }
    return 0;
}
