// Host-side helpers of the C ABI (no device code).
// hdrsky_crc32c: CRC-32C (Castagnoli, reflected 0x82F63B78), slicing-by-8 - the checksum TFRecord framing and the TF
// tensor-bundle checkpoint format use (tb_logging.py, tf_bundle.py); a 233 MB parameter buffer takes ~0.2 s.
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "hdrsky.h"

namespace {
uint32_t g_tab[8][256];
bool g_init = false;

void init_tables() {
  for (uint32_t n = 0; n < 256; ++n) {
    uint32_t c = n;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_tab[0][n] = c;
  }
  for (uint32_t n = 0; n < 256; ++n)
    for (int t = 1; t < 8; ++t) g_tab[t][n] = (g_tab[t - 1][n] >> 8) ^ g_tab[0][g_tab[t - 1][n] & 0xFF];
  g_init = true;
}
}  // namespace

extern "C" unsigned int hdrsky_crc32c(const void* data, size_t n, unsigned int crc) {
  if (!g_init) init_tables();
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~crc;
  while (n >= 8) {
    uint64_t w;
    std::memcpy(&w, p, 8);
    w ^= c;
    c = g_tab[7][w & 0xFF] ^ g_tab[6][(w >> 8) & 0xFF] ^ g_tab[5][(w >> 16) & 0xFF] ^ g_tab[4][(w >> 24) & 0xFF] ^
        g_tab[3][(w >> 32) & 0xFF] ^ g_tab[2][(w >> 40) & 0xFF] ^ g_tab[1][(w >> 48) & 0xFF] ^ g_tab[0][(w >> 56) & 0xFF];
    p += 8;
    n -= 8;
  }
  while (n--) c = g_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}

extern "C" int hdrsky_abi_version(void) { return HDRSKY_ABI_VERSION; }

extern "C" size_t hdrsky_sizeof(const char* name) {
  if (!name) return 0;
  if (!std::strcmp(name, "hdrsky_conv_desc")) return sizeof(hdrsky_conv_desc);
  if (!std::strcmp(name, "hdrsky_wgrad_job")) return sizeof(hdrsky_wgrad_job);
  if (!std::strcmp(name, "hdrsky_resconv_args")) return sizeof(hdrsky_resconv_args);
  return 0;
}
