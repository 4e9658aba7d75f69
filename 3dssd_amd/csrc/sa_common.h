// Shared device/host helpers for the gfx950 set-abstraction kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SA_OK 0
#define SA_ERR_INVALID (-1)   // bad shape / attribute (the reference's OP_REQUIRES checks)
#define SA_ERR_LAUNCH (-2)    // hipGetLastError() after a launch
#define SA_ERR_UNSUPPORTED (-3)
#define SA_ERR_PARTNERS (-4)  // a multi-workgroup sampler gave up waiting for its partner workgroups (sticky: sa_coop_error_state)

#define SA_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return SA_ERR_LAUNCH;       \
    } while (0)

// Kernel-selection knobs for A/B measurements.  The PRODUCT build reads nothing from the environment (the library is
// stateless and its behaviour a function of its arguments only): SA_KNOB(name, default) is the constant `default`.
// `make TUNE=1` (-DSA_TUNING_KNOBS, output lib3dssd_sa_tune.so, loaded by tools/ through 3dssd_amd.utils._native.LIB_PATH)
// makes it getenv(name), read once.
#ifdef SA_TUNING_KNOBS
#include <stdlib.h>
#define SA_KNOB(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define SA_KNOB(name, dflt) (dflt)
#endif

namespace sa {

// ---- sticky error word of the multi-workgroup samplers (fps_coop.hip, ffps_fly.hip; defined in hostutil.hip) ----------
// Their partner workgroups exchange words through memory and poll, which is only safe while all of them are resident.
// A workgroup whose bounded poll runs out no longer traps (that killed the process): it raises this word -- one int in
// pinned, device-visible host memory -- with a system-scope store and returns; everything that depends on it times out
// the same way within a few hundred ms, the launch completes with garbage in the affected frames, and the NEXT call
// of such a sampler (or sa_coop_error_state) reports SA_ERR_PARTNERS.  nullptr when the word could not be allocated.
int *coop_error_word();
constexpr int kCoopErrFps = 1, kCoopErrFfps = 2;

// Zero `bytes` (a multiple of 4, `p` 4-byte aligned) with a KERNEL on `stream` (hostutil.hip); hipSuccess or the launch
// error.  Not hipMemsetAsync: captured into a hipGraph, a memset node was not reliably ordered in front of the kernel
// node behind it on ROCm 7.2 -- the multi-workgroup sampler then read whatever the region held (scratch of a later stage
// of the previous replay) as its partners' exchange words and, where 16 bits of that matched the pick number, picked
// differently in 2-10 % of the replays, without any time-out (round 6, tools/verify_layers.py; round 4 saw the same with
// a memset BETWEEN two launches).  Kernel -> kernel edges are what every captured pass of this library relies on.
hipError_t zero_async(void *p, size_t bytes, hipStream_t stream);
__device__ __forceinline__ void coop_raise(int *word, int code) {
    // a plain system-scope store (not a fetch-or: an atomic RMW on host memory needs PCIe AtomicOps routing); two kinds of
    // failure in flight at once would leave the later code -- non-zero either way
    if (word) __hip_atomic_store(word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- DPP cross-lane moves (wave64, gfx9 encodings) ---------------------------------------
// quad_perm(1,0,3,2)=0xB1  quad_perm(2,3,0,1)=0x4E  row_half_mirror=0x141  row_mirror=0x140
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}

// The library is built with -fno-honor-nans, so these lower to single v_max_f32 / v_min_f32
// without the IEEE canonicalisation fmaxf() otherwise drags in; inputs are never NaN on this path.
__device__ __forceinline__ float fmax_nn(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float fmin_nn(float a, float b) { return __builtin_fminf(a, b); }

// all-reduce max inside each 16-lane row
__device__ __forceinline__ float row16_allmax(float x) {
    x = fmax_nn(x, dpp_mov<0xB1>(x));
    x = fmax_nn(x, dpp_mov<0x4E>(x));
    x = fmax_nn(x, dpp_mov<0x141>(x));
    x = fmax_nn(x, dpp_mov<0x140>(x));
    return x;
}
__device__ __forceinline__ unsigned row16_allmin_u32(unsigned x) {
    unsigned y;
    y = dpp_mov<0xB1>(x);  x = y < x ? y : x;
    y = dpp_mov<0x4E>(x);  x = y < x ? y : x;
    y = dpp_mov<0x141>(x); x = y < x ? y : x;
    y = dpp_mov<0x140>(x); x = y < x ? y : x;
    return x;
}

// all-reduce max over the 64 lanes of a wave: 4 DPP steps + the gfx950 row/half swaps.
__device__ __forceinline__ float wave_allmax(float x) {
    x = row16_allmax(x);
    {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = fmax_nn(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = fmax_nn(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return x;
}
__device__ __forceinline__ unsigned wave_allmin_u32(unsigned x) {
    x = row16_allmin_u32(x);
    {
        auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        x = r[0] < r[1] ? r[0] : r[1];
    }
    {
        auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        x = r[0] < r[1] ? r[0] : r[1];
    }
    return x;
}

// ---- bf16 split helpers --------------------------------------------------------------------
// round-to-nearest-even fp32 -> bf16 bit pattern (inputs finite)
__device__ __forceinline__ unsigned bf16_rne_bits(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
// x ~= hi + lo with hi, lo bf16: hi = rne(x), lo = rne(x - hi)
__device__ __forceinline__ void bf16_split(float x, unsigned &hi, unsigned &lo) {
    hi = bf16_rne_bits(x);
    float r = x - __uint_as_float(hi << 16);
    lo = bf16_rne_bits(r);
}

// ---- workgroup id -> position in the work list, XCD-aware ---------------------------------------------
// Block b of a launch is observed to run on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement": a
// speed hint, no contract).  Kernels that walk a list in frame order take their position from xcd_block() instead of
// blockIdx.x: of the G workgroups launched, the first A8 = roundup8(min(G, need)) take part (`need` = workgroups that
// have work in the first pass -- grids are sized for the densest row plan, the plan at hand is usually shorter -- so the
// work stays spread over all eight XCDs), and the A8 / 8 of them on one XCD get CONTIGUOUS positions: every pass over
// the list gives an XCD one contiguous eighth, a frame's points and features are fetched into one or two of the eight
// L2s instead of all of them.  Returns the position (a bijection of [0, A8)), -1 for a workgroup that sits out, and the
// stride between passes; the identity with stride G when A8 does not fit the grid.  A wrong guess about the placement
// only costs the saving.
__device__ __forceinline__ int xcd_block(int b, int G, int need, int &stride) {
    const int A8 = ((need < G ? need : G) + 7) & ~7;
    if (A8 > G || A8 == 0) { stride = G; return b; }
    stride = A8;
    return b >= A8 ? -1 : (b & 7) * (A8 >> 3) + (b >> 3);
}

}  // namespace sa
