// Farthest-point sampling of ONE frame by SEVERAL workgroups (cooperative launch), for the shapes whose
// per-frame state does not fit one CU: c-channel F-FPS on raw points (BASELINE.json configs[2]: n = 16384,
// c = 3 + 64 -> 4.4 MB per frame) and coordinate D-FPS beyond 16384 points (configs[4]: n = 65536).
//
// The single-workgroup kernels of fps.hip must re-read the frame from L2 in every one of the m-1 dependent
// iterations once it no longer fits registers/LDS (29 us per iteration at 4.4 MB and one CU's ~150 GB/s).  Here
// G = 2..16 workgroups of 1024 threads share a frame; each thread keeps its P points (all c channels) and their
// running minimum in VGPRs for the whole kernel, so an iteration touches memory only for
//   * the picked point's c channels (scalar loads, L2 hits), and
//   * one 8-byte slot per workgroup: {max value | pick number | tie key}, written with an agent-scope atomic
//     store and polled by the G-1 partners (agent-scope atomic loads) -- the cross-workgroup arg-max.
// Co-residency of the partners: an eager call uses hipLaunchCooperativeKernel (the runtime guarantees it and runs
// cooperative grids one at a time).  Cooperative launches cannot be captured into a hipGraph, so on a CAPTURING stream the
// same kernel is launched plainly (round 4): the grid of one launch never exceeds the device's
// resident capacity for this kernel (checked below, larger batches run as consecutive launches inside the call), and
// the CALLER guarantees that no two such launches from different streams are in flight together -- the staged
// executor (pipeline.py) issues every layer-1 sampling stage on its one sampler stream, and refuses graphs in its
// one-stream-per-slot mode for frames that need this kernel.  Why the restriction: partners of one frame sit on up to
// eight XCDs, each XCD places its workgroups in its own order, and with two such grids interleaved by the dispatcher
// XCD 0 can fill up with halves of grid X whose partners queue on XCD 1 behind halves of grid Y whose partners queue
// on XCD 0 behind X.  One grid at a time cannot do that (the oldest incomplete frame is first in every queue).  The
// plain launch under capture is therefore OPT-IN (capture_ok: the caller states that it keeps such launches on one
// stream -- the staged executor does); without it a capturing stream gets SA_ERR_UNSUPPORTED and the caller's
// single-workgroup kernels, which are safe on any number of streams.  The poll is bounded; a workgroup whose bound runs
// out raises the sticky error word (sa_common.h coop_raise) and returns instead of trapping the process: the launch ends,
// the frame's picks are garbage, and the next call -- or sa_coop_error_state() -- returns SA_ERR_PARTNERS.
// Tried and dropped (measured, MI355X): keeping the G partners of a frame on one XCD (ids of one residue class mod 8,
// verified in-kernel through HW_REG_XCC_ID) and exchanging the slots through that XCD's L2 instead of the
// device-coherent sc1 path.  A plain / sc0 store is not seen by an sc0 load outside threadgroup-split mode (poll
// times out), a returning atomic-OR poll serialises the 16 x G pollers on one line (64 us per pick), and
// buffer_inv sc1 + plain load costs 34 us per pick; the sc1 store / sc1 load pair below stays at 2.3 us.
//
// Semantics = tf_sampling_g.cu:123-178 exactly (same fmaf chain over the channels, same strict-maximum rule).
// The reference's tie-break among equal maxima is "lowest k mod 1024, then lowest k" (thread t owns k = t,
// t+1024, ... and the tree keeps the lower thread).  Workgroup g owns the points with k div 1024 in
// [g*P, (g+1)*P), thread t those with k mod 1024 == t, so inside a workgroup the fps.hip rules (first strict
// maximum per thread, lowest lane, lowest wave) already give that order; across workgroups the 16-bit key
// (k mod 1024) << 6 | (k div 1024) is compared explicitly.
#include <stdint.h>
#include <stdlib.h>

#include "sa_common.h"

namespace {

constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / 64;
constexpr float kInit = 1e38f;       // tf_sampling_g.cu:136
constexpr float kAbsent = -3.0e38f;  // slot of a thread that owns no point: never beats best = -1
constexpr unsigned kMaxSpin = 1u << 22;

template <int C, int P>
__global__ __launch_bounds__(kBlock) void fps_coop_kernel(int n, int m, int gshift, const float *__restrict__ inp,
                                                          unsigned long long *slots, int *__restrict__ out,
                                                          int out_stride, int idx_off, int *err_word, unsigned max_spin) {
    __shared__ float s_val[2][kWaves];
    __shared__ unsigned s_key[2][kWaves];
    __shared__ int s_old[2];
    const int G = 1 << gshift;
    const int f = blockIdx.x >> gshift, g = blockIdx.x & (G - 1);
    const float *p = inp + (size_t)f * n * C;
    int *o = out + (size_t)f * out_stride;
    unsigned long long *sl = slots + (size_t)f * 2 * G;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    float x[P][C], td[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int k = (g * P + j) * kBlock + t;
        const bool ok = k < n;
        const float *row = p + (size_t)(ok ? k : 0) * C;
#pragma unroll
        for (int l = 0; l < C; ++l) x[j][l] = row[l];
        td[j] = ok ? kInit : kAbsent;
    }
    int old = 0;                                       // tf_sampling_g.cu:130-133
    if (g == 0 && t == 0) o[0] = idx_off;

    for (int it = 1; it < m; ++it) {
        const float *po = p + (size_t)old * C;         // uniform address: scalar loads
        float d[P];
#pragma unroll
        for (int j = 0; j < P; ++j) d[j] = 0.0f;
#pragma unroll
        for (int l = 0; l < C; ++l) {
            const float q = po[l];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const float diff = x[j][l] - q;
                d[j] = __builtin_fmaf(diff, diff, d[j]);
            }
        }
        float best = -1.0f;                            // :141
        int bj = 0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const float t2 = sa::fmin_nn(d[j], td[j]); // :151-153
            td[j] = t2;
            const bool gt = t2 > best;                 // strict: first maximum wins, :154
            best = gt ? t2 : best;
            bj = gt ? j : bj;
        }
        const float wmax = sa::wave_allmax(best);
        const unsigned long long cand = __ballot(best == wmax);
        const int first = __builtin_ctzll(cand);
        const int par = it & 1;
        if (lane == first) {
            s_val[par][w] = wmax;
            s_key[par][w] = ((unsigned)t << 6) | (unsigned)(g * P + bj);
        }
        __syncthreads();
        const float v = s_val[par][lane & (kWaves - 1)];
        const unsigned ky = s_key[par][lane & (kWaves - 1)];
        const float M = sa::row16_allmax(v);
        const int ws = __builtin_ctzll(__ballot(v == M)) & (kWaves - 1);   // lowest wave holding the maximum
        // (readlane outside the branch: under `if (t == 0)` the compiler sinks the LDS read of ky into the
        // branch, where only lane 0 is active, and readlane of an inactive lane is undefined)
        const unsigned wkey = (unsigned)__builtin_amdgcn_readlane((int)ky, ws);
        if (t == 0) {
            const unsigned long long word =
                ((unsigned long long)__float_as_uint(M) << 32) | ((unsigned long long)(unsigned)it << 16) | wkey;
            __hip_atomic_store(sl + par * G + g, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ONE wave polls the G slots of this pick and hands the winner to the others through LDS + a second barrier
        // (round 5; until then every wave polled: 16 x G loads per poll round on the same lines of the coherence point).
        if (w == 0) {
            const unsigned long long *sp = sl + par * G + (lane & (G - 1));
            unsigned long long wv;
            unsigned spins = 0;
            int winner = -1;                           // -1: partners lost
            for (;;) {
                wv = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (((unsigned)wv >> 16) & 0xFFFFu) == (unsigned)it;
                if (__ballot(ok) == ~0ull) {
                    const float gv = __uint_as_float((unsigned)(wv >> 32));
                    const float GM = sa::row16_allmax(gv);
                    const unsigned gk = (gv == GM) ? ((unsigned)wv & 0xFFFFu) : 0xFFFFFFFFu;
                    const unsigned kmin = sa::row16_allmin_u32(gk);
                    winner = __builtin_amdgcn_readfirstlane((int)(((kmin & 63u) << 10) | (kmin >> 6)));
                    break;
                }
                if (++spins > max_spin) {              // partners lost: sticky error word, and every wave of the workgroup leaves
                    sa::coop_raise(err_word, sa::kCoopErrFps);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) s_old[par] = winner;
        }
        __syncthreads();
        old = s_old[par];
        if (old < 0) return;
        if (g == 0 && t == 0) o[it] = old + idx_off;
    }
}

struct Variant {
    const void *fn;
    int c, p, cap;    // cap: workgroups the device keeps resident (0 = not yet queried, <0 = unusable)
};

Variant g_variants[] = {
    {(const void *)fps_coop_kernel<3, 4>, 3, 4, 0},
    {(const void *)fps_coop_kernel<67, 1>, 67, 1, 0},
};

}  // namespace

// Returns SA_OK when the cooperative kernel was launched for all b frames; SA_ERR_UNSUPPORTED when this shape /
// stream state is not served (the caller then uses the single-workgroup kernels).  `temp` ([b,n] floats, the
// reference's scratch, tf_sampling.cpp:149-155) holds the slots.
// capture_ok: on a CAPTURING stream launch plainly (the caller keeps every such launch of the process on one stream);
// 0: a capturing stream gets SA_ERR_UNSUPPORTED.  orphan (tests only): launch one workgroup short, so that the last
// frame's partners wait in vain; max_spin: the poll bound (0 = the default).
static int fps_coop_launch(int b, int n, int c, int m, const float *inp, float *temp, int *out, int out_stride,
                           int idx_off, int capture_ok, int orphan, unsigned max_spin, hipStream_t stream) {
    static const int enabled = SA_KNOB("SA_FPS_COOP", 1);
    if (!enabled || !temp || ((uintptr_t)temp & 7) || m > 65535 || n <= kBlock) return SA_ERR_UNSUPPORTED;
    int *err_word = sa::coop_error_word();
    if (err_word && __atomic_load_n(err_word, __ATOMIC_RELAXED) != 0) return SA_ERR_PARTNERS;   // sticky: an earlier launch lost partners
    Variant *v = nullptr;
    for (auto &cand : g_variants)
        if (cand.c == c) v = &cand;
    if (!v) return SA_ERR_UNSUPPORTED;
    int gshift = 1;
    while (((size_t)kBlock * v->p << gshift) < (size_t)n) ++gshift;
    if (gshift > 4) return SA_ERR_UNSUPPORTED;         // 16 workgroups x 1024 threads x P points
    const int G = 1 << gshift;
    if ((size_t)n * sizeof(float) < (size_t)2 * G * sizeof(unsigned long long)) return SA_ERR_UNSUPPORTED;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return SA_ERR_UNSUPPORTED;
    }
    const bool capturing = cs != hipStreamCaptureStatusNone;   // -> plain launches (header comment), opt-in
    if (capturing && !capture_ok) return SA_ERR_UNSUPPORTED;
    if (capturing && !err_word) return SA_ERR_UNSUPPORTED;     // no error word (its allocation is refused during a capture): never run unchecked
    int cap = __atomic_load_n(&v->cap, __ATOMIC_ACQUIRE);
    if (cap == 0) {                                     // first use (two threads may both query: same answer)
        int dev = 0, cus = 0, per_cu = 0, coop = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v->fn, kBlock, 0) != hipSuccess)
            cap = -1;
        else
            cap = cus * per_cu > 0 ? cus * per_cu : -1;
        (void)hipGetLastError();
        __atomic_store_n(&v->cap, cap, __ATOMIC_RELEASE);
    }
    if (cap < G) return SA_ERR_UNSUPPORTED;
    const int per_launch = cap / G;                     // frames per cooperative launch
    if (max_spin == 0) max_spin = kMaxSpin;
    // every frame of the call has its OWN slots (2 G words), zeroed by one KERNEL (sa::zero_async: not a memset node, see
    // sa_common.h) in front of the first launch: the
    // consecutive launches of a large batch share nothing, so nothing depends on how a memset between two of them is
    // ordered (a shared region re-zeroed between the launches gave wrong picks in frames of the SECOND launch when the
    // call was replayed from a hipGraph beside other streams' kernels -- round 4, tools/archive/dbg_pipe.py)
    unsigned long long *slots0 = (unsigned long long *)temp;
#ifdef SA_COOP_MEMSET          // (variant builds only: the memset node of rounds 4-5, for tests/…does_not_read_what_its_exchange_words_held_before)
    if (hipMemsetAsync(slots0, 0, (size_t)b * 2 * G * sizeof(unsigned long long), stream) != hipSuccess) return SA_ERR_LAUNCH;
#else
    if (sa::zero_async(slots0, (size_t)b * 2 * G * sizeof(unsigned long long), stream) != hipSuccess) return SA_ERR_LAUNCH;
#endif
    for (int f0 = 0; f0 < b; f0 += per_launch) {
        const int nf = b - f0 < per_launch ? b - f0 : per_launch;
        unsigned long long *slots = slots0 + (size_t)f0 * 2 * G;
        const float *inp_f = inp + (size_t)f0 * n * c;
        int *out_f = out + (size_t)f0 * out_stride;
        void *args[] = {&n, &m, &gshift, &inp_f, &slots, &out_f, &out_stride, &idx_off, &err_word, &max_spin};
        static const int plain_knob = SA_KNOB("SA_FPS_COOP_PLAIN", 0);
        const bool plain = capturing || plain_knob || orphan;
        const int blocks = nf * G - ((orphan && f0 + nf >= b) ? 1 : 0);
        const hipError_t le = plain ? hipLaunchKernel(v->fn, dim3(blocks), dim3(kBlock), args, 0, stream)
                                    : hipLaunchCooperativeKernel(v->fn, dim3(blocks), dim3(kBlock), args, 0, stream);
        if (le != hipSuccess) {
            (void)hipGetLastError();
            return f0 == 0 ? SA_ERR_UNSUPPORTED : SA_ERR_LAUNCH;
        }
    }
    return SA_OK;
}

extern "C" int sa_fps_coop_ex(int b, int n, int c, int m, const float *inp, float *temp, int *out, int out_stride,
                              int idx_off, int capture_ok, hipStream_t stream) {
    return fps_coop_launch(b, n, c, m, inp, temp, out, out_stride, idx_off, capture_ok, 0, 0u, stream);
}

// TEST HOOK (tests/test_ops_gpu.py): the same launch with the last workgroup missing and a short poll bound, so that
// the partners of the last frame give up -- the sticky error word must then be raised and the process must live.
extern "C" int sa_debug_fps_coop_orphan(int b, int n, int c, int m, const float *inp, float *temp, int *out,
                                        unsigned max_spin, hipStream_t stream) {
    if (b <= 0 || n <= 0 || c <= 0 || m <= 0 || !inp || !temp || !out) return SA_ERR_INVALID;
    return fps_coop_launch(b, n, c, m, inp, temp, out, m, 0, 0, 1, max_spin ? max_spin : 20000u, stream);
}
