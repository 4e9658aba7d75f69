// Host-side helpers of the C ABI (no device code).
//   sa_host_crc32c: CRC-32C (Castagnoli, reflected 0x82F63B78) as TensorFlow's tensor-bundle checkpoints use it for
//   block and tensor checksums (3dssd_amd/utils/tf_checkpoint.py verifies multi-megabyte tensors through this).
#include <stddef.h>
#include <stdint.h>

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
