// Host-side helpers of the C ABI (no device code).
//   sa::coop_error_word / sa_coop_error_state: the sticky error word of the multi-workgroup samplers (sa_common.h).
//   sa_host_crc32c: CRC-32C (Castagnoli, reflected 0x82F63B78) as TensorFlow's tensor-bundle checkpoints use it for
//   block and tensor checksums (3dssd_amd/utils/tf_checkpoint.py verifies multi-megabyte tensors through this).
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "sa_common.h"

namespace sa {
int *coop_error_word() {
    // Allocated on first use and RETRIED until it succeeds (under call_once a first use inside a global-mode
    // stream capture -- where hipHostMalloc is refused -- left the word null for the life of the process, and a partner
    // loss then went unreported).  3dssd_amd.utils._native.lib() asks for it when the library is loaded on a GPU box.
    static std::atomic<int *> word{nullptr};
    static std::mutex mu;
    int *w = word.load(std::memory_order_acquire);
    if (w) return w;
    std::lock_guard<std::mutex> lock(mu);
    w = word.load(std::memory_order_acquire);
    if (w) return w;
    void *p = nullptr;
    // pinned + mapped: the device stores into it directly, the host reads it without a synchronisation
    if (hipHostMalloc(&p, 64, hipHostMallocMapped) == hipSuccess && p) {
        w = (int *)p;
        *w = 0;
        word.store(w, std::memory_order_release);
    } else {
        (void)hipGetLastError();
    }
    return w;
}

namespace {
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned *p, size_t nwords) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nwords; i += stride) {
        if (i + 4 <= nwords && ((uintptr_t)(p + i) & 15) == 0) {
            *(uint4 *)(p + i) = make_uint4(0u, 0u, 0u, 0u);
        } else {
            for (size_t j = i; j < nwords && j < i + 4; ++j) p[j] = 0u;
        }
    }
}
}  // namespace
hipError_t zero_async(void *p, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    const size_t nwords = bytes / 4;
    size_t blocks = (nwords + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned *)p, nwords);
    return hipGetLastError();
}
}  // namespace sa

// The sticky word: 0 = fine; bit 0 = a multi-workgroup D-FPS (fps_coop.hip), bit 1 = an on-the-fly F-FPS (ffps_fly.hip)
// gave up waiting for partner workgroups in some earlier launch -- the outputs of that launch are invalid.  reset != 0
// clears it after reading.  (No synchronisation: a launch still in flight may raise it later.)
extern "C" int sa_coop_error_state(int reset) {
    int *w = sa::coop_error_word();
    if (!w) return 0;
    const int v = __atomic_load_n(w, __ATOMIC_RELAXED);
    if (reset) __atomic_store_n(w, 0, __ATOMIC_RELAXED);
    return v;
}

namespace {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};
}  // namespace

extern "C" uint32_t sa_host_crc32c(const void *data, size_t len, uint32_t crc) {
    static const Crc32cTables tb;
    const uint8_t *p = (const uint8_t *)data;
    uint32_t c = ~crc;
    while (len && ((uintptr_t)p & 7)) { c = tb.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --len; }
    while (len >= 8) {                       // slicing-by-8
        uint64_t w;
        __builtin_memcpy(&w, p, 8);
        const uint32_t lo = (uint32_t)w ^ c, hi = (uint32_t)(w >> 32);
        c = tb.t[7][lo & 0xFF] ^ tb.t[6][(lo >> 8) & 0xFF] ^ tb.t[5][(lo >> 16) & 0xFF] ^ tb.t[4][lo >> 24] ^
            tb.t[3][hi & 0xFF] ^ tb.t[2][(hi >> 8) & 0xFF] ^ tb.t[1][(hi >> 16) & 0xFF] ^ tb.t[0][hi >> 24];
        p += 8; len -= 8;
    }
    while (len--) c = tb.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}
