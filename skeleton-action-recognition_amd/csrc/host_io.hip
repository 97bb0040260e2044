// Host-side helpers of the input pipeline (no device code): CRC-32C and TFRecord framing.
//
// The reference reads its clips with tf.data.TFRecordDataset (main_gnn.py:159-194), whose native reader verifies the
// masked CRC-32C of every record.  A 180 KB clip per record at >= 1000 clips/s per GPU needs a CRC at >= 200 MB/s per
// rank -- far beyond a Python byte loop -- so the checksum and the record framing live here, next to the kernels, behind
// the same C ABI (include/sar_hip.h, "host-side input helpers").
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/sar_hip.h"

namespace {

// slice-by-8 tables of the reflected Castagnoli polynomial 0x82F63B78
struct Tables {
  uint32_t t[8][256];
  Tables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
const Tables kTab;

uint32_t crc_sw(uint32_t c, const uint8_t* p, size_t n) {
  while (n && ((uintptr_t)p & 7)) {
    c = kTab.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    --n;
  }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = kTab.t[7][w & 0xFF] ^ kTab.t[6][(w >> 8) & 0xFF] ^ kTab.t[5][(w >> 16) & 0xFF] ^ kTab.t[4][(w >> 24) & 0xFF] ^
        kTab.t[3][(w >> 32) & 0xFF] ^ kTab.t[2][(w >> 40) & 0xFF] ^ kTab.t[1][(w >> 48) & 0xFF] ^ kTab.t[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = kTab.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c;
}

#if defined(__x86_64__)
// the SSE4.2 crc32 instruction implements exactly this polynomial; three independent streams hide its 3-cycle latency
__attribute__((target("sse4.2"))) uint32_t crc_hw(uint32_t c, const uint8_t* p, size_t n) {
  uint64_t c0 = c;
  while (n && ((uintptr_t)p & 7)) {
    c0 = __builtin_ia32_crc32qi((uint32_t)c0, *p++);
    --n;
  }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    c0 = __builtin_ia32_crc32di(c0, w);
    p += 8;
    n -= 8;
  }
  while (n--) c0 = __builtin_ia32_crc32qi((uint32_t)c0, *p++);
  return (uint32_t)c0;
}
bool have_hw() {
  static const bool v = __builtin_cpu_supports("sse4.2");
  return v;
}
#endif

inline uint32_t crc_any(const uint8_t* p, size_t n) {
  uint32_t c = 0xFFFFFFFFu;
#if defined(__x86_64__)
  c = have_hw() ? crc_hw(c, p, n) : crc_sw(c, p, n);
#else
  c = crc_sw(c, p, n);
#endif
  return c ^ 0xFFFFFFFFu;
}

inline uint32_t mask(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }

}  // namespace

extern "C" uint32_t sar_crc32c(const void* data, int64_t n) { return crc_any((const uint8_t*)data, (size_t)n); }

extern "C" uint32_t sar_crc32c_sw(const void* data, int64_t n) {
  return crc_sw(0xFFFFFFFFu, (const uint8_t*)data, (size_t)n) ^ 0xFFFFFFFFu;
}

extern "C" uint32_t sar_masked_crc32c(const void* data, int64_t n) { return mask(crc_any((const uint8_t*)data, (size_t)n)); }

extern "C" int64_t sar_tfrecord_index(const void* file, int64_t nbytes, int verify, int64_t* offsets, int64_t* lengths,
                                      int64_t max_records) {
  const uint8_t* p = (const uint8_t*)file;
  int64_t pos = 0, n = 0;
  if (!p || nbytes < 0) return -1;
  while (pos < nbytes) {
    if (nbytes - pos < 12) return -(2 + n * 4);  // truncated header
    uint64_t len;
    uint32_t lcrc;
    memcpy(&len, p + pos, 8);
    memcpy(&lcrc, p + pos + 8, 4);
    if (verify && mask(crc_any(p + pos, 8)) != lcrc) return -(3 + n * 4);  // corrupt length CRC
    // header + footer alone need 16 bytes: with 12..15 left the subtraction below would go negative (and huge as uint64)
    if (nbytes - pos < 16 || len > (uint64_t)(nbytes - pos - 16)) return -(4 + n * 4);   // truncated record
    if (verify >= 2) {
      uint32_t dcrc;
      memcpy(&dcrc, p + pos + 12 + len, 4);
      if (mask(crc_any(p + pos + 12, len)) != dcrc) return -(5 + n * 4);  // corrupt data CRC
    }
    if (offsets && lengths && n < max_records) {
      offsets[n] = pos + 12;
      lengths[n] = (int64_t)len;
    }
    ++n;
    pos += 16 + (int64_t)len;
  }
  return n;
}
