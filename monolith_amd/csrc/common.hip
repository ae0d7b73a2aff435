// common.hip — error plumbing + workspace of libmonolith_hip.so.
#include <cstring>

#include "mono_common.h"

namespace mono {

std::string& last_error_ref() {
  static thread_local std::string err;
  return err;
}

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return 1;
}

int Workspace::ensure(size_t need) {
  if (need <= bytes) return 0;
  // grow geometrically so steady-state calls never allocate
  size_t want = bytes ? bytes : (size_t)1 << 20;
  while (want < need) want *= 2;
  if (ptr) {
    MONO_HIP(hipDeviceSynchronize());
    MONO_HIP(hipFree(ptr));
    ptr = nullptr;
    bytes = 0;
  }
  MONO_HIP(hipMalloc(&ptr, want));
  bytes = want;
  return 0;
}

// ---- pinned staging ring for small host -> device uploads ----
namespace {
constexpr int kUpSlots = 32;
constexpr size_t kUpSlotBytes = 64 * 1024;
struct UploadRing {
  char* base = nullptr;
  hipEvent_t ev[kUpSlots] = {};
  bool used[kUpSlots] = {};
  int next = 0;
};
UploadRing g_up;
}  // namespace

int upload_async(void* dst_dev, const void* src_host, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  if (bytes > kUpSlotBytes) {
    MONO_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
    return 0;
  }
  if (!g_up.base) {  // (one host thread per GPU; the ring lives until the process ends)
    MONO_HIP(hipHostMalloc((void**)&g_up.base, kUpSlots * kUpSlotBytes, hipHostMallocPortable));
    for (int i = 0; i < kUpSlots; ++i)
      MONO_HIP(hipEventCreateWithFlags(&g_up.ev[i], hipEventDisableTiming));
  }
  const int i = g_up.next;
  g_up.next = (i + 1) % kUpSlots;
  // the slot's previous copy (32 uploads ago) must have left the staging memory
  if (g_up.used[i]) MONO_HIP(hipEventSynchronize(g_up.ev[i]));
  char* slot = g_up.base + (size_t)i * kUpSlotBytes;
  memcpy(slot, src_host, bytes);
  MONO_HIP(hipMemcpyAsync(dst_dev, slot, bytes, hipMemcpyHostToDevice, st));
  MONO_HIP(hipEventRecord(g_up.ev[i], st));
  g_up.used[i] = true;
  return 0;
}

void Workspace::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  bytes = 0;
}

}  // namespace mono

namespace mono {
// CRC-32C (Castagnoli), slicing-by-8 tables built once. TFRecord framing uses the masked form.
static uint32_t g_crc_tab[8][256];
static bool g_crc_ready = false;
static void crc_init() {
  if (g_crc_ready) return;
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t)
      g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFF];
  g_crc_ready = true;
}
static uint32_t crc32c(const uint8_t* p, size_t n) {
  crc_init();
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_crc_tab[7][lo & 0xFF] ^ g_crc_tab[6][(lo >> 8) & 0xFF] ^ g_crc_tab[5][(lo >> 16) & 0xFF] ^
        g_crc_tab[4][lo >> 24] ^ g_crc_tab[3][hi & 0xFF] ^ g_crc_tab[2][(hi >> 8) & 0xFF] ^
        g_crc_tab[1][(hi >> 16) & 0xFF] ^ g_crc_tab[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 0xFF];
  return c ^ 0xFFFFFFFFu;
}
static uint32_t masked_crc(const uint8_t* p, size_t n) {
  const uint32_t c = crc32c(p, n);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}
}  // namespace mono

extern "C" {
const char* mono_last_error(void) { return mono::last_error_ref().c_str(); }
int mono_abi_version(void) { return 5; }

uint32_t mono_crc32c(const void* data, int64_t n) { return mono::crc32c((const uint8_t*)data, (size_t)n); }

int64_t mono_tfrecord_frame(const uint8_t* payloads, int64_t n, int64_t payload_len, uint8_t* out) {
  uint8_t* o = out;
  for (int64_t i = 0; i < n; ++i) {
    const uint8_t* p = payloads + i * payload_len;
    const uint64_t len = (uint64_t)payload_len;
    memcpy(o, &len, 8);
    const uint32_t c1 = mono::masked_crc(o, 8);
    memcpy(o + 8, &c1, 4);
    memcpy(o + 12, p, (size_t)payload_len);
    const uint32_t c2 = mono::masked_crc(p, (size_t)payload_len);
    memcpy(o + 12 + payload_len, &c2, 4);
    o += 16 + payload_len;
  }
  return (int64_t)(o - out);
}

int64_t mono_tfrecord_scan(const uint8_t* buf, int64_t len, int64_t* offsets, int64_t* lengths,
                           int64_t max_records, int verify) {
  int64_t pos = 0, k = 0;
  while (pos < len) {
    if (len - pos < 12) return mono::fail("tfrecord: truncated header at byte %lld", (long long)pos), -1;
    uint64_t n;
    uint32_t c;
    memcpy(&n, buf + pos, 8);
    memcpy(&c, buf + pos + 8, 4);
    if (verify && c != mono::masked_crc(buf + pos, 8))
      return mono::fail("tfrecord: length crc mismatch at byte %lld", (long long)pos), -1;
    if ((uint64_t)(len - pos - 12) < n + 4)
      return mono::fail("tfrecord: truncated record at byte %lld", (long long)pos), -1;
    memcpy(&c, buf + pos + 12 + n, 4);
    if (verify && c != mono::masked_crc(buf + pos + 12, (size_t)n))
      return mono::fail("tfrecord: data crc mismatch at byte %lld", (long long)pos), -1;
    if (k < max_records) {
      offsets[k] = pos + 12;
      lengths[k] = (int64_t)n;
    }
    ++k;
    pos += 16 + (int64_t)n;
  }
  return k;
}
}
