// HBM write rate against the SHAPE of the writes: does a 128 x 128 fp32 tile of a [4096 x 4096] matrix (128 pieces of
// 512 bytes, 16 KB apart) reach the rate of a sequential fill?  The F-FPS matrix kernel (csrc/sqdist.hip) writes one
// direct and one mirrored tile per workgroup at 4.1 TB/s; a plain fill measures 6.3.
//   hipcc --offload-arch=gfx950 -O3 -o write_pattern write_pattern.hip && ./write_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 4096;

template <bool NT> __device__ __forceinline__ void st(f4 *p, f4 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// mode 0: sequential, a workgroup fills 64 KB in a row of the buffer
// mode 1: tile (bi, bj), tiles row-major: consecutive workgroups write neighbouring 512-byte pieces of the same rows
// mode 2: tile (bj, bi): consecutive workgroups write tiles stacked in a column
// mode 3: both (one direct + one mirrored tile per workgroup, half as many workgroups): the matrix kernel's pattern
// mode 4: RH rows x RW floats pieces (RH * RW = 16384), workgroups row-major over the pieces
template <bool NT>
__global__ __launch_bounds__(256) void wp_kernel(float *out, int mode, int RW, int frames, int xcd) {
    unsigned L = blockIdx.x;
    const int tid = threadIdx.x;
    const f4 v = {1.0f, 2.0f, 3.0f, (float)L};
    if (mode == 0) {
        f4 *p = (f4 *)(out + (size_t)L * 16384);
        for (int i = tid; i < 4096; i += 256) st<NT>(p + i, v);
        return;
    }
    const int T = N / 128;                                   // 32 tiles per side
    if (mode == 1 || mode == 2 || mode == 3) {
        const unsigned per = mode == 3 ? T * T / 2 : T * T;
        unsigned b = L / per, t = L % per;
        if (xcd) {                                           // frames of 8 consecutive workgroups ids -> one XCD each
            const unsigned G8 = 8u * per;
            b = 8u * (L / G8) + (L & 7u);
            t = (L % G8) >> 3;
        }
        int bi, bj;
        if (mode == 3) { bi = t / T; bj = t % T; if (bi >= T / 2) { bi -= T / 2; } bi *= 2; }   // any tile pair; rows 2*bi, 2*bi+1
        else { bi = t / T; bj = t % T; }
        float *base = out + (size_t)b * N * N;
        const int r0 = tid >> 5, c = (tid & 31) * 4;         // 8 rows x 512 bytes per pass
        if (mode == 1 || mode == 3) {
            float *tile = base + (size_t)(mode == 3 ? bi : bi) * 128 * N + bj * 128;
            for (int r = r0; r < 128; r += 8) st<NT>((f4 *)(tile + (size_t)r * N + c), v);
        }
        if (mode == 2 || mode == 3) {
            const int ri = mode == 3 ? bi + 1 : bi;
            float *tile = base + (size_t)bj * 128 * N + ri * 128;
            for (int r = r0; r < 128; r += 8) st<NT>((f4 *)(tile + (size_t)r * N + c), v);
        }
        return;
    }
    // mode 4: pieces of RH x RW
    const int RH = 16384 / RW, PW = N / RW;                  // pieces per row of pieces
    const unsigned per = (N / RH) * PW;
    const unsigned b = L / per, t = L % per;
    const int pi = t / PW, pj = t % PW;
    float *tile = out + (size_t)b * N * N + (size_t)pi * RH * N + pj * RW;
    const int q = RW / 4;                                    // float4 per row of the piece
    for (int i = tid; i < 4096; i += 256) st<NT>((f4 *)(tile + (size_t)(i / q) * N + (i % q) * 4), v);
}

int main() {
    const int frames = 64;                                   // 4.3 GB
    float *out;
    const size_t bytes = (size_t)frames * N * N * 4;
    if (hipMalloc(&out, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *name; int mode, RW, nt, xcd; } cases[] = {
        {"sequential 64 KB per workgroup", 0, 0, 1, 0}, {"sequential, plain stores", 0, 0, 0, 0},
        {"tiles 128x128 row-major (512 B pieces)", 1, 0, 1, 0}, {"tiles 128x128 column-major", 2, 0, 1, 0},
        {"direct + mirrored tile per workgroup", 3, 0, 1, 0}, {"direct + mirrored, plain stores", 3, 0, 0, 0},
        {"direct + mirrored, frame -> XCD", 3, 0, 1, 1}, {"tiles row-major, frame -> XCD", 1, 0, 1, 1},
        {"pieces 64 x 256 floats (1 KB)", 4, 256, 1, 0}, {"pieces 32 x 512 (2 KB)", 4, 512, 1, 0},
        {"pieces 16 x 1024 (4 KB)", 4, 1024, 1, 0}, {"pieces 4 x 4096 (16 KB = full rows)", 4, 4096, 1, 0},
        {"pieces 256 x 64 floats (256 B)", 4, 64, 1, 0},
    };
    for (auto &c : cases) {
        const unsigned per = c.mode == 3 ? 512 : 1024;
        const unsigned grid = frames * per;
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0, 0);
            if (c.nt) hipLaunchKernelGGL(wp_kernel<true>, dim3(grid), dim3(256), 0, 0, out, c.mode, c.RW, frames, c.xcd);
            else hipLaunchKernelGGL(wp_kernel<false>, dim3(grid), dim3(256), 0, 0, out, c.mode, c.RW, frames, c.xcd);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        printf("%-45s %7.3f ms  %6.2f TB/s\n", c.name, best, bytes / best * 1e-9);
        fflush(stdout);
    }
    return 0;
}
