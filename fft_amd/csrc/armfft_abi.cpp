// armfft_abi.cpp -- the C ABI of libarmfft_mi355.so (include/arm_fft_mi355.h): host logic only.
//
// Mirrors the reference's init/execute interface (Include/dsp/transform_functions.h; bodies in
// Source/TransformFunctions/arm_cfft_init_*.c, arm_cfft_*.c, arm_rfft_fast_*.c) and routes every
// execute call to the HIP kernels.  There is NO CPU fallback: if no GPU is usable the execute
// functions fail loudly (arm_status / arm_fft_mi355_last_error, message on stderr).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "armfft_internal.h"

namespace armfft {

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char t_err[512] = "";

void set_error(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_err, sizeof(t_err), fmt, ap);
  va_end(ap);
  fprintf(stderr, "libarmfft_mi355: %s\n", t_err);
}

// last kernel(s) launched by this thread: two (format, arguments) records, formatted when asked for
struct KernelNote {
  const char *fmt = nullptr;
  int a[5] = {0, 0, 0, 0, 0};
};
static thread_local KernelNote t_note[2];
static thread_local char t_note_text[320] = "";

void note_kernel(const char *fmt, bool append, int a, int b, int c, int d, int e)
{
  KernelNote &n = t_note[append ? 1 : 0];
  n.fmt = fmt;
  n.a[0] = a; n.a[1] = b; n.a[2] = c; n.a[3] = d; n.a[4] = e;
  if (!append) t_note[1].fmt = nullptr;
}

bool hip_ok(hipError_t e, const char *what)
{
  if (e == hipSuccess) return true;
  set_error("%s failed: %s", what, hipGetErrorString(e));
  return false;
}

// ------------------------------------------------------------------------------------------
// host tables (what pTwiddle / pBitRevTable point to).  The reference's definitions live in the
// constant blob Source/CommonTables/arm_common_tables.c; here they are generated once per size:
//   f32  twiddle[2i] = cos(2 pi i/N), [2i+1] = sin(2 pi i/N)            (arm_common_tables.h:63-84)
//   rfft twiddle[2i] = sin(2 pi i/N), [2i+1] = cos(2 pi i/N), i < N/2   (:151-165)
//   q15/q31: 3N/4 (cos, sin) pairs * 2^15 / 2^31, +1.0 saturated; rule per arm_fft_mi355_set_twiddle_rounding:
//            0 floor, 1 nearest (both on the double value), 2 the fork's own generator (CMSIS/CMSIS.cpp:445-458
//            + :309-327), 3 (default) = floor up to 4096 points, the fork's generator at 8192 / 16384 (:88-130)
//   f32 bit-reversal index table: swap pairs (byte offsets) realising the radix-(2|4),8,8.. digit
//            reversal, same format arm_bitreversal_32 consumes (arm_bitreversal2.c:84-108)
//   fixed-point index table: plain bit-reversal swap pairs (arm_bitreversal2.c:123-148)
// ------------------------------------------------------------------------------------------
static std::mutex g_mu;
static int g_tw_rounding = 3;   // 3 = per size: see effective_rounding()

struct HostTables {
  std::vector<float> tw_f32, tw_rfft;
  std::vector<int16_t> tw_q15;
  std::vector<int32_t> tw_q31;
  std::vector<uint16_t> br_f32, br_fixed;
};
static HostTables g_host[kMaxLog + 2];

// The rule a size's table is made by.  Mode 3 (default): the sizes the reference's tests cover (16 ... 4096) keep floor --
// the fork's generator misses the reference's own q31 bound there (tests/test_oracle_golden.py), so those tables were not
// printed by it; 8192 and 16384 are the fork's additions and its generator is the only rule the mount holds for them.
static int effective_rounding(int mode, int logn) { return mode == 3 ? (logn >= 13 ? 2 : 0) : mode; }

static int tw_rounding(int logn)
{
  std::lock_guard<std::mutex> lk(g_mu);
  return effective_rounding(g_tw_rounding, logn);
}

static int ilog2u(uint32_t n)
{
  int l = 0;
  while ((1u << l) < n) l++;
  return l;
}
static bool is_pow2(uint32_t n) { return n && !(n & (n - 1)); }

static void exact_cs(uint32_t i, uint32_t N, double *c, double *s)
{
  if ((4u * i) % N == 0) {
    static const double cq[4] = {1, 0, -1, 0}, sq[4] = {0, 1, 0, -1};
    uint32_t q = (4u * i / N) & 3u;
    *c = cq[q];
    *s = sq[q];
  } else {
    double a = 2.0 * M_PI * (double)i / (double)N;
    *c = cos(a);
    *s = sin(a);
  }
}

static int64_t to_fixed(double v, double scale, int rounding, int64_t lo, int64_t hi)
{
  double s = v * scale;
  double r = rounding ? floor(s + 0.5) : floor(s);
  if (r > (double)hi) r = (double)hi;
  if (r < (double)lo) r = (double)lo;
  return (int64_t)r;
}

// Rule 2, the fork's own generator: twiddleCoefCalculator (CMSIS/CMSIS.cpp:445-458) feeds to_fixed (:309-327).
//   float c = cos(i * 2 * PI / (float)fft_len) with the FLOAT literal PI (Include/gnss_codes.h:33): int i*2, float product,
//   float division; cos / sin of that float, stored to float; to_fixed: float product with (float)pow(2, dyad), llround,
//   clamp (b > max -> max, b <= min -> min; the constants carry the q15 values in comments: swapped with the dyad),
//   negatives b = ~abs(b) = one lower than the rounded value.  No exact values at multiples of pi/2.
static void fork_cs(uint32_t i, uint32_t N, float *c, float *s)
{
  volatile float num = (float)(int)(i * 2u) * 3.14159265358979f;   // volatile: the two float roundings stay two
  volatile float arg = num / (float)N;
  *c = (float)cos((double)arg);
  *s = (float)sin((double)arg);
}
static int64_t fork_to_fixed(float f, int dyad, int64_t lo, int64_t hi)
{
  const float tmp = (float)pow(2.0f, dyad);
  volatile float prod = f * tmp;
  int64_t b = (int64_t)llround((double)prod);
  if (b > 0 && b > hi) return hi;
  if (b < 0 && b <= lo) return lo;
  if (b < 0) b = ~(-b);
  return b;
}

static uint32_t brev(uint32_t v, int bits)
{
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

// position of output bin k before the reordering (see scrambled_pos in armfft_f32.hip)
static uint32_t f32_pos_of_bin(uint32_t k, int logn)
{
  int l0 = logn % 3, n8 = logn / 3;
  uint32_t d0 = k & ((1u << l0) - 1), rest = k >> l0, low = 0;
  for (int s = 0; s < n8; s++) {
    low = (low << 3) | (rest & 7u);
    rest >>= 3;
  }
  return (d0 << (3 * n8)) | low;
}

static const float *host_tw_f32(int logn)
{
  std::lock_guard<std::mutex> lk(g_mu);
  auto &t = g_host[logn].tw_f32;
  if (t.empty()) {
    uint32_t N = 1u << logn;
    t.resize(2 * N);
    for (uint32_t i = 0; i < N; i++) {
      double a = 2.0 * M_PI * (double)i / (double)N;
      t[2 * i] = (float)cos(a);
      t[2 * i + 1] = (float)sin(a);
    }
  }
  return t.data();
}

static const float *host_tw_rfft(int logn_real)
{
  std::lock_guard<std::mutex> lk(g_mu);
  auto &t = g_host[logn_real].tw_rfft;
  if (t.empty()) {
    uint32_t N = 1u << logn_real;
    t.resize(N);
    for (uint32_t i = 0; i < N / 2; i++) {
      double a = 2.0 * M_PI * (double)i / (double)N;
      t[2 * i] = (float)sin(a);
      t[2 * i + 1] = (float)cos(a);
    }
  }
  return t.data();
}

static const int16_t *host_tw_q15(int logn)
{
  std::lock_guard<std::mutex> lk(g_mu);
  auto &t = g_host[logn].tw_q15;
  if (t.empty()) {
    uint32_t N = 1u << logn;
    t.resize(3 * N / 2);
    const int rule = effective_rounding(g_tw_rounding, logn);
    for (uint32_t i = 0; i < 3 * N / 4; i++) {
      if (rule == 2) {
        float c, s;
        fork_cs(i, N, &c, &s);
        t[2 * i] = (int16_t)fork_to_fixed(c, 15, -32768, 32767);
        t[2 * i + 1] = (int16_t)fork_to_fixed(s, 15, -32768, 32767);
        continue;
      }
      double c, s;
      exact_cs(i, N, &c, &s);
      t[2 * i] = (int16_t)to_fixed(c, 32768.0, rule, -32768, 32767);
      t[2 * i + 1] = (int16_t)to_fixed(s, 32768.0, rule, -32768, 32767);
    }
  }
  return t.data();
}

static const int32_t *host_tw_q31(int logn)
{
  std::lock_guard<std::mutex> lk(g_mu);
  auto &t = g_host[logn].tw_q31;
  if (t.empty()) {
    uint32_t N = 1u << logn;
    t.resize(3 * N / 2);
    const int rule = effective_rounding(g_tw_rounding, logn);
    for (uint32_t i = 0; i < 3 * N / 4; i++) {
      if (rule == 2) {
        float c, s;
        fork_cs(i, N, &c, &s);
        t[2 * i] = (int32_t)fork_to_fixed(c, 31, INT32_MIN, INT32_MAX);
        t[2 * i + 1] = (int32_t)fork_to_fixed(s, 31, INT32_MIN, INT32_MAX);
        continue;
      }
      double c, s;
      exact_cs(i, N, &c, &s);
      t[2 * i] = (int32_t)to_fixed(c, 2147483648.0, rule, INT32_MIN, INT32_MAX);
      t[2 * i + 1] = (int32_t)to_fixed(s, 2147483648.0, rule, INT32_MIN, INT32_MAX);
    }
  }
  return t.data();
}

// realCoefA/B of arm_rfft_q15/q31 (Include/arm_common_tables.h:271-275: 8192 scalars = 4096 pairs, the
// fork's 8192-point base).  Definitions are in the missing table blob; regenerated by the upstream rule
//   A[2i] = 0.5 (1 - sin(2 pi i / 8192)), A[2i+1] = -0.5 cos(..), B[2i] = 0.5 (1 + sin(..)), B[2i+1] = 0.5 cos(..)
// with round-half-away-from-zero to Q15 / Q31 and saturation (exact values at the quarter points).
constexpr int kRealCoefPairs = 4096;
struct RealCoef {
  std::vector<int16_t> a15, b15;
  std::vector<int32_t> a31, b31;
};
static const RealCoef &host_real_coef()
{
  static RealCoef rc;
  std::lock_guard<std::mutex> lk(g_mu);
  if (rc.a15.empty()) {
    rc.a15.resize(2 * kRealCoefPairs); rc.b15.resize(2 * kRealCoefPairs);
    rc.a31.resize(2 * kRealCoefPairs); rc.b31.resize(2 * kRealCoefPairs);
    auto fx = [](double v, double scale, double lo, double hi) { double r = round(v * scale); return r < lo ? lo : (r > hi ? hi : r); };
    for (int i = 0; i < kRealCoefPairs; i++) {
      double s, c;
      if (i == 0) { s = 0.0; c = 1.0; }
      else if (i == kRealCoefPairs / 2) { s = 1.0; c = 0.0; }
      else { s = sin(2.0 * M_PI * (double)i / (2.0 * kRealCoefPairs)); c = cos(2.0 * M_PI * (double)i / (2.0 * kRealCoefPairs)); }
      const double v[4] = {0.5 * (1.0 - s), 0.5 * (-1.0 * c), 0.5 * (1.0 + s), 0.5 * (1.0 * c)};
      rc.a15[2 * i] = (int16_t)fx(v[0], 32768.0, -32768.0, 32767.0);
      rc.a15[2 * i + 1] = (int16_t)fx(v[1], 32768.0, -32768.0, 32767.0);
      rc.b15[2 * i] = (int16_t)fx(v[2], 32768.0, -32768.0, 32767.0);
      rc.b15[2 * i + 1] = (int16_t)fx(v[3], 32768.0, -32768.0, 32767.0);
      rc.a31[2 * i] = (int32_t)fx(v[0], 2147483648.0, -2147483648.0, 2147483647.0);
      rc.a31[2 * i + 1] = (int32_t)fx(v[1], 2147483648.0, -2147483648.0, 2147483647.0);
      rc.b31[2 * i] = (int32_t)fx(v[2], 2147483648.0, -2147483648.0, 2147483647.0);
      rc.b31[2 * i + 1] = (int32_t)fx(v[3], 2147483648.0, -2147483648.0, 2147483647.0);
    }
  }
  return rc;
}

// Swap-pair tables use 16-bit byte offsets (8 bytes per complex f32/q31 point), so they can only
// describe N <= 8192; for 16384 the instance carries a NULL table and length 0 (the reference's
// own 16384 table is declared uint32_t and unusable through this field, arm_common_tables.h:265).
static const uint16_t *host_br_f32(int logn, uint16_t *len)
{
  std::lock_guard<std::mutex> lk(g_mu);
  if (logn > 13) { *len = 0; return nullptr; }
  auto &t = g_host[logn].br_f32;
  if (t.empty()) {
    uint32_t N = 1u << logn;
    std::vector<uint32_t> perm(N);  // out[k] = in[perm[k]]
    for (uint32_t k = 0; k < N; k++) perm[k] = f32_pos_of_bin(k, logn);
    std::vector<uint8_t> done(N, 0);
    for (uint32_t s = 0; s < N; s++) {
      if (done[s] || perm[s] == s) { done[s] = 1; continue; }
      uint32_t a = s;
      done[a] = 1;
      for (;;) {
        uint32_t b = perm[a];
        if (b == s) break;
        t.push_back((uint16_t)(8 * a));
        t.push_back((uint16_t)(8 * b));
        done[b] = 1;
        a = b;
      }
    }
    if (t.empty()) t.push_back(0), t.push_back(0);
  }
  *len = (uint16_t)t.size();
  return t.data();
}

// armBitRevTable of the fork's 16384-based legacy scheme (arm_common_tables.h:41, generator
// CMSIS/CMSIS.cpp:333-354): entry k = 13-bit reversal of k + 1, consumed by arm_bitreversal_f32
// (arm_bitreversal.c:46-105) with stride bitRevFactor.
static const uint16_t *host_br_radix2()
{
  std::lock_guard<std::mutex> lk(g_mu);
  static std::vector<uint16_t> t;
  if (t.empty()) {
    t.resize(4096);
    for (uint32_t k = 0; k < 4096; k++) t[k] = (uint16_t)brev(k + 1, 13);
  }
  return t.data();
}

static const uint16_t *host_br_fixed(int logn, uint16_t *len)
{
  std::lock_guard<std::mutex> lk(g_mu);
  if (logn > 13) { *len = 0; return nullptr; }
  auto &t = g_host[logn].br_fixed;
  if (t.empty()) {
    uint32_t N = 1u << logn;
    for (uint32_t i = 0; i < N; i++) {
      uint32_t j = brev(i, logn);
      if (i < j) {
        t.push_back((uint16_t)(8 * i));
        t.push_back((uint16_t)(8 * j));
      }
    }
  }
  *len = (uint16_t)t.size();
  return t.data();
}

// ------------------------------------------------------------------------------------------
// device plans: per GPU, per size, generated ON the device at first use and kept for the life
// of the process (a few hundred KiB in total).
// ------------------------------------------------------------------------------------------
constexpr int kMaxDev = 16;
struct DevPlans {
  float2 *tw_f32[kMaxLog + 2] = {};
  float2 *tw_rfft[kMaxLog + 2] = {};
  int16_t *tw_q15[kMaxLog + 2] = {};
  int32_t *tw_q31[kMaxLog + 2] = {};
  float2 *tw_lo[kMaxLargeLog + 1] = {};
  // realCoefA | realCoefB as the n_real-point transform reads them: entries i * twidCoefRModifier,
  // i < n_real/2, compacted per size so that consecutive lanes read consecutive entries
  int16_t *rc15[14] = {};
  int32_t *rc31[14] = {};
  void *stage = nullptr;  // grow-only staging buffers for the host-pointer entry points
  size_t stage_bytes = 0;
  void *stage2 = nullptr;
  size_t stage2_bytes = 0;
};
static DevPlans g_dev[kMaxDev];
static std::mutex g_dev_mu;

// Plan slots are written once (under g_dev_mu, with a release store once the table is complete) and
// never freed, so steady-state launches read them with one acquire load and take no lock.
template <typename T>
static inline T *slot_peek(T *const &slot) { return __atomic_load_n(&slot, __ATOMIC_ACQUIRE); }
template <typename T>
static inline void slot_publish(T *&slot, T *p) { __atomic_store_n(&slot, p, __ATOMIC_RELEASE); }

// _dev entry points: the kernels use 16-byte vector accesses on the data (armfft_f32.hip staged path,
// armfft_fixed.hip loads/stores), so device pointers must be 16-byte aligned (hipMalloc gives 256)
static bool aligned16(const void *p, const char *fn)
{
  if (((uintptr_t)p & 15u) == 0) return true;
  set_error("%s: device pointer %p is not 16-byte aligned", fn, p);
  return false;
}

// inverse arm_rfft_q15 / _q31: a transform's input is n + 2 scalars (68 bytes for 32 q15 points), read point by point, so a
// caller that slices a batch lands on word boundaries only: the kernels need the point's own alignment, not 16 bytes
static bool aligned_to(const void *p, unsigned bytes, const char *fn)
{
  if (((uintptr_t)p & (uintptr_t)(bytes - 1)) == 0) return true;
  set_error("%s: device pointer %p is not %u-byte aligned", fn, p, bytes);
  return false;
}

static bool current_device(int *dev)
{
  // the device count cannot change during the life of the process: ask once (a launch otherwise pays
  // for it every time, ~3 us)
  static std::atomic<int> n_cached{-1};
  int n = n_cached.load(std::memory_order_relaxed);
  if (n < 0) {
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) n = 0;
    n_cached.store(n, std::memory_order_relaxed);
  }
  if (n <= 0) {
    set_error("no HIP device available: libarmfft_mi355 has no CPU path");
    return false;
  }
  if (!hip_ok(hipGetDevice(dev), "hipGetDevice")) return false;
  if (*dev < 0 || *dev >= kMaxDev) {
    set_error("device index %d out of range", *dev);
    return false;
  }
  return true;
}

static const float2 *dev_tw_f32(int logn)
{
  int dev;
  if (!current_device(&dev)) return nullptr;
  float2 *&slot = g_dev[dev].tw_f32[logn];
  if (float2 *q = slot_peek(slot)) return q;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  if (slot) return slot;
  float2 *p = nullptr;
  uint32_t N = 1u << logn;
  if (!hip_ok(hipMalloc((void **)&p, sizeof(float2) * N), "hipMalloc(twiddle)")) return nullptr;
  if (!hip_ok(launch_gen_twiddle_f32(p, N, nullptr), "twiddle generation") || !hip_ok(hipStreamSynchronize(nullptr), "sync")) {
    (void)hipFree(p);
    return nullptr;
  }
  slot_publish(slot, p);
  return p;
}

static const float2 *dev_tw_rfft(int logn_real)
{
  int dev;
  if (!current_device(&dev)) return nullptr;
  float2 *&slot = g_dev[dev].tw_rfft[logn_real];
  if (float2 *q = slot_peek(slot)) return q;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  if (slot) return slot;
  float2 *p = nullptr;
  uint32_t N = 1u << logn_real;
  if (!hip_ok(hipMalloc((void **)&p, sizeof(float2) * (N / 2)), "hipMalloc(rfft twiddle)")) return nullptr;
  if (!hip_ok(launch_gen_twiddle_rfft(p, N, nullptr), "rfft twiddle generation") || !hip_ok(hipStreamSynchronize(nullptr), "sync")) {
    (void)hipFree(p);
    return nullptr;
  }
  slot_publish(slot, p);
  return p;
}

const float2 *dev_twiddle_f32(int logn) { return dev_tw_f32(logn); }
static const float2 *dev_tw_lo(int logn);
const float2 *dev_twiddle_lo(int logn) { return dev_tw_lo(logn); }
const float2 *dev_twiddle_f32_peek(int logn)
{
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev || logn < 0 || logn > kMaxLog) return nullptr;
  return slot_peek(g_dev[dev].tw_f32[logn]);
}
const float2 *dev_twiddle_lo_peek(int logn)
{
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev || logn < 0 || logn > kMaxLargeLog) return nullptr;
  return slot_peek(g_dev[dev].tw_lo[logn]);
}

static const float2 *dev_tw_lo(int logn)
{
  int dev;
  if (!current_device(&dev)) return nullptr;
  float2 *&slot = g_dev[dev].tw_lo[logn];
  if (float2 *q = slot_peek(slot)) return q;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  if (slot) return slot;
  float2 *p = nullptr;
  if (!hip_ok(hipMalloc((void **)&p, sizeof(float2) * 1024), "hipMalloc(lo twiddle)")) return nullptr;
  if (!hip_ok(launch_gen_twiddle_lo(p, logn, nullptr), "lo twiddle generation") || !hip_ok(hipStreamSynchronize(nullptr), "sync")) {
    (void)hipFree(p);
    return nullptr;
  }
  slot_publish(slot, p);
  return p;
}

// Fixed-point tables are generated on the device and then CHECKED against the host table the
// instance points to: a double-precision cos/sin that differs in the last bit between libm and
// the device could move a floor() across an integer (about 1e-7 per q31 entry), and bit-exact
// output needs bit-exact twiddles.  On any mismatch the host table is uploaded instead.
template <typename T>
static T *dev_tw_fixed(T *&slot, int logn, const T *host, hipError_t (*gen)(T *, uint32_t, int, hipStream_t), int rounding)
{
  if (slot) return slot;
  uint32_t N = 1u << logn;
  size_t count = 3 * (size_t)N / 2, bytes = count * sizeof(T);
  T *p = nullptr;
  if (!hip_ok(hipMalloc((void **)&p, bytes), "hipMalloc(fixed twiddle)")) return nullptr;
  std::vector<T> back(count);
  bool ok = hip_ok(gen(p, N, rounding, nullptr), "fixed twiddle generation") &&
            hip_ok(hipMemcpy(back.data(), p, bytes, hipMemcpyDeviceToHost), "hipMemcpy(D2H twiddle)");
  if (ok && memcmp(back.data(), host, bytes) != 0) ok = hip_ok(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice), "hipMemcpy(H2D twiddle)");
  if (!ok) {
    (void)hipFree(p);
    return nullptr;
  }
  slot_publish(slot, p);
  return p;
}

static const int16_t *dev_tw_q15(int logn)
{
  int dev;
  if (!current_device(&dev)) return nullptr;
  if (int16_t *q = slot_peek(g_dev[dev].tw_q15[logn])) return q;
  const int16_t *host = host_tw_q15(logn);
  const int rounding = tw_rounding(logn);
  std::lock_guard<std::mutex> lk(g_dev_mu);
  return dev_tw_fixed<int16_t>(g_dev[dev].tw_q15[logn], logn, host, launch_gen_twiddle_q15, rounding);
}

static const int32_t *dev_tw_q31(int logn)
{
  int dev;
  if (!current_device(&dev)) return nullptr;
  if (int32_t *q = slot_peek(g_dev[dev].tw_q31[logn])) return q;
  const int32_t *host = host_tw_q31(logn);
  const int rounding = tw_rounding(logn);
  std::lock_guard<std::mutex> lk(g_dev_mu);
  return dev_tw_fixed<int32_t>(g_dev[dev].tw_q31[logn], logn, host, launch_gen_twiddle_q31, rounding);
}

// device copies of realCoefA/B for n_real = 2^l: L = n_real/2 pairs of A (entries i * mod of the 4096-pair
// table, mod = 8192 / n_real = twidCoefRModifier), then L pairs of B
template <typename T>
static const T *dev_real_coef(T *&slot, const std::vector<T> &A, const std::vector<T> &B, int l)
{
  if (!slot) {
    const size_t L = (size_t)1 << (l - 1), mod = (size_t)8192 >> l;
    std::vector<T> h(4 * L);
    for (size_t i = 0; i < L; i++) {
      h[2 * i] = A[2 * i * mod];
      h[2 * i + 1] = A[2 * i * mod + 1];
      h[2 * L + 2 * i] = B[2 * i * mod];
      h[2 * L + 2 * i + 1] = B[2 * i * mod + 1];
    }
    T *p = nullptr;
    if (!hip_ok(hipMalloc((void **)&p, sizeof(T) * h.size()), "hipMalloc(realCoef)")) return nullptr;
    if (!hip_ok(hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice), "hipMemcpy(realCoef)")) {
      (void)hipFree(p);
      return nullptr;
    }
    slot_publish(slot, p);
  }
  return slot;
}
static const int16_t *dev_rc15(int l)
{
  int dev;
  if (!current_device(&dev)) return nullptr;
  if (int16_t *q = slot_peek(g_dev[dev].rc15[l])) return q;
  const RealCoef &rc = host_real_coef();
  std::lock_guard<std::mutex> lk(g_dev_mu);
  return dev_real_coef<int16_t>(g_dev[dev].rc15[l], rc.a15, rc.b15, l);
}
static const int32_t *dev_rc31(int l)
{
  int dev;
  if (!current_device(&dev)) return nullptr;
  if (int32_t *q = slot_peek(g_dev[dev].rc31[l])) return q;
  const RealCoef &rc = host_real_coef();
  std::lock_guard<std::mutex> lk(g_dev_mu);
  return dev_real_coef<int32_t>(g_dev[dev].rc31[l], rc.a31, rc.b31, l);
}

// n_real = 32 .. 8192 (arm_rfft_init_q15.c:178-370; the fork's 16384 entry reuses modifier 1 on the
// 8192-scalar tables, :152, and would read past them: refused)
static int rfft_fixed_log(uint32_t n_real)
{
  if (!is_pow2(n_real)) return -1;
  int l = ilog2u(n_real);
  return (l >= 5 && l <= 13) ? l : -1;
}

// staging for host-pointer calls: one pair of device buffers per GPU, so the calls of one GPU are serialised by that GPU's
// mutex; threads driving different GPUs do not wait for each other (these entry points are the compatibility path, not
// the fast path: pageable copies, synchronous)
static std::mutex g_stage_mu_dev[kMaxDev];
static std::mutex &stage_mu(int dev) { return g_stage_mu_dev[dev >= 0 && dev < kMaxDev ? dev : 0]; }
static void *stage_buffer(int dev, size_t bytes, int which)
{
  void *&p = which ? g_dev[dev].stage2 : g_dev[dev].stage;
  size_t &cap = which ? g_dev[dev].stage2_bytes : g_dev[dev].stage_bytes;
  if (cap < bytes) {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    if (!hip_ok(hipMalloc(&p, bytes), "hipMalloc(staging)")) return nullptr;
    cap = bytes;
  }
  return p;
}

// ------------------------------------------------------------------------------------------
// length checks
// ------------------------------------------------------------------------------------------
static int cfft_log(uint32_t n)
{
  if (!is_pow2(n)) return -1;
  int l = ilog2u(n);
  return (l >= kMinLog && l <= kMaxLog) ? l : -1;
}

template <typename S, typename T>
static arm_status host_batch(const S *inst, T *p, uint32_t batch, size_t elems_per_xform,
                             arm_status (*dev_fn)(const S *, T *, uint32_t, uint8_t, uint8_t, void *), uint8_t ifft,
                             uint8_t bitrev)
{
  if (!inst || !p) {
    set_error("null argument");
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if (batch == 0) return ARM_MATH_SUCCESS;
  int dev;
  if (!current_device(&dev)) return ARM_MATH_ARGUMENT_ERROR;
  std::lock_guard<std::mutex> lk(stage_mu(dev));
  size_t bytes = elems_per_xform * batch * sizeof(T);
  T *d = (T *)stage_buffer(dev, bytes, 0);
  if (!d) return ARM_MATH_ARGUMENT_ERROR;
  if (!hip_ok(hipMemcpy(d, p, bytes, hipMemcpyHostToDevice), "hipMemcpy(H2D)")) return ARM_MATH_ARGUMENT_ERROR;
  arm_status s = dev_fn(inst, d, batch, ifft, bitrev, nullptr);
  if (s != ARM_MATH_SUCCESS) return s;
  if (!hip_ok(hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost), "hipMemcpy(D2H)")) return ARM_MATH_ARGUMENT_ERROR;
  return ARM_MATH_SUCCESS;
}

}  // namespace armfft

using namespace armfft;

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

const char *arm_fft_mi355_version(void) { return "armfft_mi355 0.1 (gfx950)"; }
const char *arm_fft_mi355_last_error(void) { return t_err; }

const char *arm_fft_mi355_last_kernel(void)
{
  size_t used = 0;
  t_note_text[0] = 0;
  for (int i = 0; i < 2; i++) {
    const KernelNote &n = t_note[i];
    if (!n.fmt || used + 8 >= sizeof t_note_text) continue;
    if (i && used) used += (size_t)snprintf(t_note_text + used, sizeof t_note_text - used, " + ");
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wformat-nonliteral"
#pragma GCC diagnostic ignored "-Wformat-security"
    int w = snprintf(t_note_text + used, sizeof t_note_text - used, n.fmt, n.a[0], n.a[1], n.a[2], n.a[3], n.a[4]);
#pragma GCC diagnostic pop
    if (w > 0) used = (used + (size_t)w < sizeof t_note_text) ? used + (size_t)w : sizeof t_note_text - 1;
  }
  return t_note_text;
}

arm_status arm_fft_mi355_set_device(int device)
{
  return hip_ok(hipSetDevice(device), "hipSetDevice") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

arm_status arm_fft_mi355_set_twiddle_rounding(int mode)
{
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &h : g_host)
    if (!h.tw_q15.empty() || !h.tw_q31.empty()) {
      set_error("twiddle rounding must be chosen before the first fixed-point init");
      return ARM_MATH_ARGUMENT_ERROR;
    }
  if (mode < 0 || mode > 3) {
    set_error("twiddle rounding mode %d: 0 floor, 1 nearest, 2 the fork's generator, 3 per size (default)", mode);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  g_tw_rounding = mode;
  return ARM_MATH_SUCCESS;
}

arm_status arm_fft_mi355_warm(int kind, uint32_t fftLen)
{
  int l = cfft_log(kind == 3 ? fftLen / 2 : fftLen);
  if (l < 0) return ARM_MATH_ARGUMENT_ERROR;
  switch (kind) {
    case 0: return dev_tw_f32(l) ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
    case 1: return dev_tw_q15(l) ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
    case 2: return dev_tw_q31(l) ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
    case 3: return (dev_tw_f32(l) && dev_tw_rfft(l + 1)) ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
  }
  return ARM_MATH_ARGUMENT_ERROR;
}

// ---------------- f32 complex ----------------
arm_status arm_cfft_init_f32(arm_cfft_instance_f32 *S, uint16_t fftLen)
{
  if (!S) return ARM_MATH_ARGUMENT_ERROR;
  int l = cfft_log(fftLen);
  if (l < 0) return ARM_MATH_ARGUMENT_ERROR;  // S untouched, arm_cfft_init_f32.c:346-349
  S->fftLen = fftLen;
  S->pTwiddle = host_tw_f32(l);
  S->pBitRevTable = host_br_f32(l, &S->bitRevLength);
  return ARM_MATH_SUCCESS;
}

#define ARMFFT_INIT_N(N, T)                                                                   \
  arm_status arm_cfft_init_##N##_##T(arm_cfft_instance_##T *S) { return arm_cfft_init_##T(S, N); }
ARMFFT_INIT_N(4096, f32) ARMFFT_INIT_N(2048, f32) ARMFFT_INIT_N(1024, f32) ARMFFT_INIT_N(512, f32)
ARMFFT_INIT_N(256, f32) ARMFFT_INIT_N(128, f32) ARMFFT_INIT_N(64, f32) ARMFFT_INIT_N(32, f32) ARMFFT_INIT_N(16, f32)

arm_status arm_cfft_f32_dev(const arm_cfft_instance_f32 *S, float32_t *p, uint32_t batch, uint8_t ifft, uint8_t bitrev,
                            void *stream)
{
  if (!S || !p) { set_error("arm_cfft_f32_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (!aligned16(p, "arm_cfft_f32_dev")) return ARM_MATH_ARGUMENT_ERROR;
  int l = cfft_log(S->fftLen);
  if (l < 0) { set_error("arm_cfft_f32_dev: unsupported fftLen %u", (unsigned)S->fftLen); return ARM_MATH_ARGUMENT_ERROR; }
  const float2 *tw = dev_tw_f32(l);
  if (!tw) return ARM_MATH_ARGUMENT_ERROR;
  hipError_t e = launch_cfft_f32(l, reinterpret_cast<float2 *>(p), tw, batch, ifft != 0, bitrev ? 0 : 1, (hipStream_t)stream);
  return hip_ok(e, "cfft_f32 launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

arm_status arm_cfft_batch_f32(const arm_cfft_instance_f32 *S, float32_t *p, uint32_t batch, uint8_t ifft, uint8_t bitrev)
{
  if (!S || cfft_log(S->fftLen) < 0) { set_error("arm_cfft_batch_f32: unsupported instance"); return ARM_MATH_ARGUMENT_ERROR; }
  return host_batch<arm_cfft_instance_f32, float32_t>(S, p, batch, 2 * (size_t)S->fftLen, arm_cfft_f32_dev, ifft, bitrev);
}

void arm_cfft_f32(const arm_cfft_instance_f32 *S, float32_t *p1, uint8_t ifftFlag, uint8_t bitReverseFlag)
{
  if (!S || cfft_log(S->fftLen) < 0) return;  // silent no-op like the reference's switch (arm_cfft_f32.c:1263-1280)
  if (arm_cfft_batch_f32(S, p1, 1, ifftFlag, bitReverseFlag) != ARM_MATH_SUCCESS) {
    fprintf(stderr, "libarmfft_mi355: arm_cfft_f32 cannot run (%s)\n", t_err);
    abort();  // a void entry point has no other way to refuse to return wrong data
  }
}

// ---------------- f32 complex, deprecated radix-2 API (what fft_c32 calls, CMSIS/CMSIS.cpp:218-234) ----------------
// arm_cfft_radix2_init_f32.c:64-218: the fields are written BEFORE the length switch, so an
// unsupported length still leaves fftLen/flags/pTwiddle set and returns ARM_MATH_ARGUMENT_ERROR.
arm_status arm_cfft_radix2_init_f32(arm_cfft_radix2_instance_f32 *S, uint16_t fftLen, uint8_t ifftFlag, uint8_t bitReverseFlag)
{
  if (!S) return ARM_MATH_ARGUMENT_ERROR;
  S->fftLen = fftLen;
  S->pTwiddle = host_tw_f32(kMaxLog);  // the shared 16384-entry table, indexed with twidCoefModifier
  S->ifftFlag = ifftFlag;
  S->bitReverseFlag = bitReverseFlag;
  int l = cfft_log(fftLen);
  if (l < 0) return ARM_MATH_ARGUMENT_ERROR;
  const uint16_t mod = (uint16_t)(16384u >> l);
  S->twidCoefModifier = mod;
  S->bitRevFactor = mod;
  S->pBitRevTable = host_br_radix2() + (mod - 1);
  S->onebyfftLen = 1.0f / (float)fftLen;
  return ARM_MATH_SUCCESS;
}

arm_status arm_cfft_radix2_f32_dev(const arm_cfft_radix2_instance_f32 *S, float32_t *p, uint32_t batch, void *stream)
{
  if (!S || !p) { set_error("arm_cfft_radix2_f32_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (!aligned16(p, "arm_cfft_radix2_f32_dev")) return ARM_MATH_ARGUMENT_ERROR;
  int l = cfft_log(S->fftLen);
  if (l < 0) { set_error("arm_cfft_radix2_f32_dev: unsupported fftLen %u", (unsigned)S->fftLen); return ARM_MATH_ARGUMENT_ERROR; }
  const float2 *tw = dev_tw_f32(l);
  if (!tw) return ARM_MATH_ARGUMENT_ERROR;
  // bitReverseFlag == 0 leaves the radix-2 DIF order: plain bit reversal (arm_cfft_radix2_f32.c:64-88)
  hipError_t e = launch_cfft_f32(l, reinterpret_cast<float2 *>(p), tw, batch, S->ifftFlag != 0, S->bitReverseFlag ? 0 : 2,
                                 (hipStream_t)stream);
  return hip_ok(e, "cfft_radix2_f32 launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

static arm_status radix2_dev_adapter(const arm_cfft_radix2_instance_f32 *S, float32_t *p, uint32_t batch, uint8_t, uint8_t, void *stream)
{
  return arm_cfft_radix2_f32_dev(S, p, batch, stream);
}

arm_status arm_cfft_radix2_batch_f32(const arm_cfft_radix2_instance_f32 *S, float32_t *p, uint32_t batch)
{
  if (!S || cfft_log(S->fftLen) < 0) { set_error("arm_cfft_radix2_batch_f32: unsupported instance"); return ARM_MATH_ARGUMENT_ERROR; }
  return host_batch<arm_cfft_radix2_instance_f32, float32_t>(S, p, batch, 2 * (size_t)S->fftLen, radix2_dev_adapter, 0, 0);
}

void arm_cfft_radix2_f32(const arm_cfft_radix2_instance_f32 *S, float32_t *pSrc)
{
  if (!S || cfft_log(S->fftLen) < 0) return;
  if (arm_cfft_radix2_batch_f32(S, pSrc, 1) != ARM_MATH_SUCCESS) {
    fprintf(stderr, "libarmfft_mi355: arm_cfft_radix2_f32 cannot run (%s)\n", t_err);
    abort();
  }
}

// ---------------- fixed-point real FFT (arm_rfft_q15 / arm_rfft_q31) ----------------
// static complex instances the real instances point to (pCfft), one per size
static const arm_cfft_instance_q15 *cfft_inst_q15(int logl)
{
  static arm_cfft_instance_q15 inst[kMaxLog + 1];
  static std::once_flag once[kMaxLog + 1];
  std::call_once(once[logl], [&] { (void)arm_cfft_init_q15(&inst[logl], (uint16_t)(1u << logl)); });
  return &inst[logl];
}
static const arm_cfft_instance_q31 *cfft_inst_q31(int logl)
{
  static arm_cfft_instance_q31 inst[kMaxLog + 1];
  static std::once_flag once[kMaxLog + 1];
  std::call_once(once[logl], [&] { (void)arm_cfft_init_q31(&inst[logl], (uint16_t)(1u << logl)); });
  return &inst[logl];
}

// arm_rfft_init_q15.c:98-124 (RFFTINIT_Q15) + :440-500 (length switch)
arm_status arm_rfft_init_q15(arm_rfft_instance_q15 *S, uint32_t fftLenReal, uint32_t ifftFlagR, uint32_t bitReverseFlag)
{
  if (!S) return ARM_MATH_ARGUMENT_ERROR;
  const int l = rfft_fixed_log(fftLenReal);
  if (l < 0) return ARM_MATH_ARGUMENT_ERROR;
  const RealCoef &rc = host_real_coef();
  S->fftLenReal = fftLenReal;
  S->pTwiddleAReal = rc.a15.data();
  S->pTwiddleBReal = rc.b15.data();
  S->ifftFlagR = (uint8_t)ifftFlagR;
  S->bitReverseFlagR = (uint8_t)bitReverseFlag;
  S->twidCoefRModifier = 8192u >> l;
  S->pCfft = cfft_inst_q15(l - 1);
  return ARM_MATH_SUCCESS;
}
arm_status arm_rfft_init_q31(arm_rfft_instance_q31 *S, uint32_t fftLenReal, uint32_t ifftFlagR, uint32_t bitReverseFlag)
{
  if (!S) return ARM_MATH_ARGUMENT_ERROR;
  const int l = rfft_fixed_log(fftLenReal);
  if (l < 0) return ARM_MATH_ARGUMENT_ERROR;
  const RealCoef &rc = host_real_coef();
  S->fftLenReal = fftLenReal;
  S->pTwiddleAReal = rc.a31.data();
  S->pTwiddleBReal = rc.b31.data();
  S->ifftFlagR = (uint8_t)ifftFlagR;
  S->bitReverseFlagR = (uint8_t)bitReverseFlag;
  S->twidCoefRModifier = 8192u >> l;
  S->pCfft = cfft_inst_q31(l - 1);
  return ARM_MATH_SUCCESS;
}
#define ARMFFT_RFIX_INIT_N(N)                                                                                          \
  arm_status arm_rfft_init_##N##_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag) { return arm_rfft_init_q15(S, N, ifftFlagR, bitReverseFlag); } \
  arm_status arm_rfft_init_##N##_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag) { return arm_rfft_init_q31(S, N, ifftFlagR, bitReverseFlag); }
ARMFFT_RFIX_INIT_N(32) ARMFFT_RFIX_INIT_N(64) ARMFFT_RFIX_INIT_N(128) ARMFFT_RFIX_INIT_N(256) ARMFFT_RFIX_INIT_N(512)
ARMFFT_RFIX_INIT_N(1024) ARMFFT_RFIX_INIT_N(2048) ARMFFT_RFIX_INIT_N(4096) ARMFFT_RFIX_INIT_N(8192)

arm_status arm_rfft_q15_dev(const arm_rfft_instance_q15 *S, const q15_t *pSrc, q15_t *pDst, uint32_t batch, void *stream)
{
  if (!S || !pSrc || !pDst) { set_error("arm_rfft_q15_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (!(S->ifftFlagR ? aligned_to(pSrc, 4, "arm_rfft_q15_dev") : aligned16(pSrc, "arm_rfft_q15_dev")) || !aligned16(pDst, "arm_rfft_q15_dev"))
    return ARM_MATH_ARGUMENT_ERROR;
  const int l = rfft_fixed_log(S->fftLenReal);
  if (l < 0) { set_error("arm_rfft_q15_dev: unsupported fftLenReal %u", S->fftLenReal); return ARM_MATH_ARGUMENT_ERROR; }
  const int16_t *tw = dev_tw_q15(l - 1), *rc = dev_rc15(l);
  if (!tw || !rc) return ARM_MATH_ARGUMENT_ERROR;
  hipError_t e = launch_rfft_q15(l, pSrc, pDst, tw, rc, rc + ((size_t)1 << l), 1, batch, S->ifftFlagR != 0,
                                 S->bitReverseFlagR != 0, (hipStream_t)stream);
  return hip_ok(e, "rfft_q15 launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}
arm_status arm_rfft_q31_dev(const arm_rfft_instance_q31 *S, const q31_t *pSrc, q31_t *pDst, uint32_t batch, void *stream)
{
  if (!S || !pSrc || !pDst) { set_error("arm_rfft_q31_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (!(S->ifftFlagR ? aligned_to(pSrc, 8, "arm_rfft_q31_dev") : aligned16(pSrc, "arm_rfft_q31_dev")) || !aligned16(pDst, "arm_rfft_q31_dev"))
    return ARM_MATH_ARGUMENT_ERROR;
  const int l = rfft_fixed_log(S->fftLenReal);
  if (l < 0) { set_error("arm_rfft_q31_dev: unsupported fftLenReal %u", S->fftLenReal); return ARM_MATH_ARGUMENT_ERROR; }
  const int32_t *tw = dev_tw_q31(l - 1), *rc = dev_rc31(l);
  if (!tw || !rc) return ARM_MATH_ARGUMENT_ERROR;
  hipError_t e = launch_rfft_q31(l, pSrc, pDst, tw, rc, rc + ((size_t)1 << l), 1, batch, S->ifftFlagR != 0,
                                 S->bitReverseFlagR != 0, (hipStream_t)stream);
  return hip_ok(e, "rfft_q31 launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// host pointers.  Forward: n_real scalars in, 2*n_real out per transform; inverse: n_real + 2 in, n_real out.
extern "C++" {
template <typename S, typename T>
static arm_status rfft_fixed_host(const S *inst, const T *src, T *dst, uint32_t batch, arm_status (*dev_fn)(const S *, const T *, T *, uint32_t, void *))
{
  if (!inst || !src || !dst) { set_error("arm_rfft (fixed): null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (rfft_fixed_log(inst->fftLenReal) < 0) { set_error("arm_rfft (fixed): unsupported fftLenReal %u", inst->fftLenReal); return ARM_MATH_ARGUMENT_ERROR; }
  if (batch == 0) return ARM_MATH_SUCCESS;
  int dev;
  if (!current_device(&dev)) return ARM_MATH_ARGUMENT_ERROR;
  const size_t n = inst->fftLenReal;
  const size_t in_elems = (inst->ifftFlagR ? n + 2 : n) * batch, out_elems = (inst->ifftFlagR ? n : 2 * n) * batch;
  std::lock_guard<std::mutex> lk(stage_mu(dev));
  T *di = (T *)stage_buffer(dev, in_elems * sizeof(T), 0), *dout = (T *)stage_buffer(dev, out_elems * sizeof(T), 1);
  if (!di || !dout) return ARM_MATH_ARGUMENT_ERROR;
  if (!hip_ok(hipMemcpy(di, src, in_elems * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(H2D)")) return ARM_MATH_ARGUMENT_ERROR;
  arm_status s = dev_fn(inst, di, dout, batch, nullptr);
  if (s != ARM_MATH_SUCCESS) return s;
  if (!hip_ok(hipMemcpy(dst, dout, out_elems * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy(D2H)")) return ARM_MATH_ARGUMENT_ERROR;
  return ARM_MATH_SUCCESS;
}
}  // extern "C++"
arm_status arm_rfft_batch_q15(const arm_rfft_instance_q15 *S, const q15_t *pSrc, q15_t *pDst, uint32_t batch)
{
  return rfft_fixed_host<arm_rfft_instance_q15, q15_t>(S, pSrc, pDst, batch, arm_rfft_q15_dev);
}
arm_status arm_rfft_batch_q31(const arm_rfft_instance_q31 *S, const q31_t *pSrc, q31_t *pDst, uint32_t batch)
{
  return rfft_fixed_host<arm_rfft_instance_q31, q31_t>(S, pSrc, pDst, batch, arm_rfft_q31_dev);
}
// arm_rfft_q15.c:148-182 / arm_rfft_q31.c.  Unlike the reference, pSrc is left intact in the forward
// direction (the reference runs the complex FFT in place on it).
void arm_rfft_q15(const arm_rfft_instance_q15 *S, q15_t *pSrc, q15_t *pDst)
{
  if (!S || rfft_fixed_log(S->fftLenReal) < 0) return;
  if (arm_rfft_batch_q15(S, pSrc, pDst, 1) != ARM_MATH_SUCCESS) {
    fprintf(stderr, "libarmfft_mi355: arm_rfft_q15 cannot run (%s)\n", t_err);
    abort();
  }
}
void arm_rfft_q31(const arm_rfft_instance_q31 *S, q31_t *pSrc, q31_t *pDst)
{
  if (!S || rfft_fixed_log(S->fftLenReal) < 0) return;
  if (arm_rfft_batch_q31(S, pSrc, pDst, 1) != ARM_MATH_SUCCESS) {
    fprintf(stderr, "libarmfft_mi355: arm_rfft_q31 cannot run (%s)\n", t_err);
    abort();
  }
}

// ---------------- large f32 complex (four-step) ----------------
arm_status arm_cfft_init_u32_f32(arm_cfft_instance_u32_f32 *S, uint32_t fftLen)
{
  if (!S) return ARM_MATH_ARGUMENT_ERROR;
  if (!is_pow2(fftLen)) return ARM_MATH_ARGUMENT_ERROR;
  int l = ilog2u(fftLen);
  if (l < kMinLog || l > kMaxLargeLog) return ARM_MATH_ARGUMENT_ERROR;
  S->fftLen = fftLen;
  return ARM_MATH_SUCCESS;
}

arm_status arm_cfft_u32_f32_dev(const arm_cfft_instance_u32_f32 *S, float32_t *p, float32_t *work, uint32_t batch, uint8_t ifft,
                                void *stream)
{
  if (!S || !p) { set_error("arm_cfft_u32_f32_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (!aligned16(p, "arm_cfft_u32_f32_dev") || (work && !aligned16(work, "arm_cfft_u32_f32_dev"))) return ARM_MATH_ARGUMENT_ERROR;
  if (!is_pow2(S->fftLen)) { set_error("arm_cfft_u32_f32_dev: unsupported fftLen %u", S->fftLen); return ARM_MATH_ARGUMENT_ERROR; }
  int l = ilog2u(S->fftLen);
  if (l >= kMinLog && l <= kMaxLog) {
    const float2 *tw = dev_tw_f32(l);
    if (!tw) return ARM_MATH_ARGUMENT_ERROR;
    return hip_ok(launch_cfft_f32(l, reinterpret_cast<float2 *>(p), tw, batch, ifft != 0, 0, (hipStream_t)stream), "cfft_f32 launch")
               ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
  }
  if (l < kMinLargeLog || l > kMaxLargeLog) { set_error("arm_cfft_u32_f32_dev: unsupported fftLen %u", S->fftLen); return ARM_MATH_ARGUMENT_ERROR; }
  if (!work || work == p) { set_error("arm_cfft_u32_f32_dev: a separate workspace of batch*fftLen*8 bytes is required"); return ARM_MATH_ARGUMENT_ERROR; }
  const int l2 = l / 2, l1 = l - l2;
  const float2 *tw1 = dev_tw_f32(l1), *tw2 = dev_tw_f32(l2), *big = dev_tw_lo(l);
  if (!tw1 || !tw2 || !big) return ARM_MATH_ARGUMENT_ERROR;
  hipError_t e = launch_cfft_large_f32(l, reinterpret_cast<float2 *>(p), reinterpret_cast<float2 *>(work), tw1, tw2, big, batch,
                                       ifft != 0, (hipStream_t)stream);
  return hip_ok(e, "cfft_large_f32 launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

arm_status arm_cfft_u32_batch_f32(const arm_cfft_instance_u32_f32 *S, float32_t *p, uint32_t batch, uint8_t ifft)
{
  if (!S || !p) { set_error("arm_cfft_u32_batch_f32: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (!is_pow2(S->fftLen) || ilog2u(S->fftLen) < kMinLog || ilog2u(S->fftLen) > kMaxLargeLog) {
    set_error("arm_cfft_u32_batch_f32: unsupported fftLen %u", S->fftLen);  // before any staging is sized from it
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if (batch == 0) return ARM_MATH_SUCCESS;
  int dev;
  if (!current_device(&dev)) return ARM_MATH_ARGUMENT_ERROR;
  std::lock_guard<std::mutex> lk(stage_mu(dev));
  size_t bytes = (size_t)S->fftLen * 2 * sizeof(float) * batch;
  float *d = (float *)stage_buffer(dev, bytes, 0), *w = (float *)stage_buffer(dev, bytes, 1);
  if (!d || !w) return ARM_MATH_ARGUMENT_ERROR;
  if (!hip_ok(hipMemcpy(d, p, bytes, hipMemcpyHostToDevice), "hipMemcpy(H2D)")) return ARM_MATH_ARGUMENT_ERROR;
  arm_status s = arm_cfft_u32_f32_dev(S, d, w, batch, ifft, nullptr);
  if (s != ARM_MATH_SUCCESS) return s;
  if (!hip_ok(hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost), "hipMemcpy(D2H)")) return ARM_MATH_ARGUMENT_ERROR;
  return ARM_MATH_SUCCESS;
}

// ---------------- q15 complex ----------------
arm_status arm_cfft_init_q15(arm_cfft_instance_q15 *S, uint16_t fftLen)
{
  if (!S) return ARM_MATH_ARGUMENT_ERROR;
  int l = cfft_log(fftLen);
  if (l < 0) return ARM_MATH_ARGUMENT_ERROR;  // S untouched, arm_cfft_init_q15.c:346-349
  S->fftLen = fftLen;
  S->pTwiddle = host_tw_q15(l);
  S->pBitRevTable = host_br_fixed(l, &S->bitRevLength);
  return ARM_MATH_SUCCESS;
}
ARMFFT_INIT_N(16384, q15) ARMFFT_INIT_N(8192, q15) ARMFFT_INIT_N(4096, q15) ARMFFT_INIT_N(2048, q15)
ARMFFT_INIT_N(1024, q15) ARMFFT_INIT_N(512, q15) ARMFFT_INIT_N(256, q15) ARMFFT_INIT_N(128, q15)
ARMFFT_INIT_N(64, q15) ARMFFT_INIT_N(32, q15) ARMFFT_INIT_N(16, q15)

arm_status arm_cfft_q15_dev(const arm_cfft_instance_q15 *S, q15_t *p, uint32_t batch, uint8_t ifft, uint8_t bitrev, void *stream)
{
  if (!S || !p) { set_error("arm_cfft_q15_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (!aligned16(p, "arm_cfft_q15_dev")) return ARM_MATH_ARGUMENT_ERROR;
  int l = cfft_log(S->fftLen);
  if (l < 0) { set_error("arm_cfft_q15_dev: unsupported fftLen %u", (unsigned)S->fftLen); return ARM_MATH_ARGUMENT_ERROR; }
  const int16_t *tw = dev_tw_q15(l);
  if (!tw) return ARM_MATH_ARGUMENT_ERROR;
  hipError_t e = launch_cfft_q15(l, p, tw, batch, ifft != 0, bitrev != 0, (hipStream_t)stream);
  return hip_ok(e, "cfft_q15 launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

arm_status arm_cfft_batch_q15(const arm_cfft_instance_q15 *S, q15_t *p, uint32_t batch, uint8_t ifft, uint8_t bitrev)
{
  if (!S || cfft_log(S->fftLen) < 0) { set_error("arm_cfft_batch_q15: unsupported instance"); return ARM_MATH_ARGUMENT_ERROR; }
  return host_batch<arm_cfft_instance_q15, q15_t>(S, p, batch, 2 * (size_t)S->fftLen, arm_cfft_q15_dev, ifft, bitrev);
}

void arm_cfft_q15(const arm_cfft_instance_q15 *S, q15_t *p1, uint8_t ifftFlag, uint8_t bitReverseFlag)
{
  if (!S || cfft_log(S->fftLen) < 0) return;  // arm_cfft_q15.c:681-721: no default case
  if (arm_cfft_batch_q15(S, p1, 1, ifftFlag, bitReverseFlag) != ARM_MATH_SUCCESS) {
    fprintf(stderr, "libarmfft_mi355: arm_cfft_q15 cannot run (%s)\n", t_err);
    abort();
  }
}

// ---------------- q31 complex ----------------
arm_status arm_cfft_init_q31(arm_cfft_instance_q31 *S, uint16_t fftLen)
{
  if (!S) return ARM_MATH_ARGUMENT_ERROR;
  int l = cfft_log(fftLen);
  if (l < 0) return ARM_MATH_ARGUMENT_ERROR;  // S untouched, arm_cfft_init_q31.c:349-352
  S->fftLen = fftLen;
  S->pTwiddle = host_tw_q31(l);
  S->pBitRevTable = host_br_fixed(l, &S->bitRevLength);
  return ARM_MATH_SUCCESS;
}
ARMFFT_INIT_N(8192, q31) ARMFFT_INIT_N(4096, q31) ARMFFT_INIT_N(2048, q31) ARMFFT_INIT_N(1024, q31)
ARMFFT_INIT_N(512, q31) ARMFFT_INIT_N(256, q31) ARMFFT_INIT_N(128, q31) ARMFFT_INIT_N(64, q31)
ARMFFT_INIT_N(32, q31) ARMFFT_INIT_N(16, q31)

arm_status arm_cfft_q31_dev(const arm_cfft_instance_q31 *S, q31_t *p, uint32_t batch, uint8_t ifft, uint8_t bitrev, void *stream)
{
  if (!S || !p) { set_error("arm_cfft_q31_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (!aligned16(p, "arm_cfft_q31_dev")) return ARM_MATH_ARGUMENT_ERROR;
  int l = cfft_log(S->fftLen);
  if (l < 0) { set_error("arm_cfft_q31_dev: unsupported fftLen %u", (unsigned)S->fftLen); return ARM_MATH_ARGUMENT_ERROR; }
  const int32_t *tw = dev_tw_q31(l);
  if (!tw) return ARM_MATH_ARGUMENT_ERROR;
  hipError_t e = launch_cfft_q31(l, p, tw, batch, ifft != 0, bitrev != 0, (hipStream_t)stream);
  return hip_ok(e, "cfft_q31 launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

arm_status arm_cfft_batch_q31(const arm_cfft_instance_q31 *S, q31_t *p, uint32_t batch, uint8_t ifft, uint8_t bitrev)
{
  if (!S || cfft_log(S->fftLen) < 0) { set_error("arm_cfft_batch_q31: unsupported instance"); return ARM_MATH_ARGUMENT_ERROR; }
  return host_batch<arm_cfft_instance_q31, q31_t>(S, p, batch, 2 * (size_t)S->fftLen, arm_cfft_q31_dev, ifft, bitrev);
}

void arm_cfft_q31(const arm_cfft_instance_q31 *S, q31_t *p1, uint8_t ifftFlag, uint8_t bitReverseFlag)
{
  if (!S || cfft_log(S->fftLen) < 0) return;  // arm_cfft_q31.c:718-754: no default case
  if (arm_cfft_batch_q31(S, p1, 1, ifftFlag, bitReverseFlag) != ARM_MATH_SUCCESS) {
    fprintf(stderr, "libarmfft_mi355: arm_cfft_q31 cannot run (%s)\n", t_err);
    abort();
  }
}

// ---------------- f32 real ----------------
arm_status arm_rfft_fast_init_f32(arm_rfft_fast_instance_f32 *S, uint16_t fftLen)
{
  if (!S) return ARM_MATH_ARGUMENT_ERROR;  // arm_rfft_fast_init_f32.c:87
  int l = (fftLen >= 32 && is_pow2(fftLen)) ? cfft_log(fftLen / 2u) : -1;
  if (l < 0 || l + 1 > kMaxLog) return ARM_MATH_ARGUMENT_ERROR;  // S untouched, arm_rfft_fast_init_f32.c:366-368
  arm_status st = arm_cfft_init_f32(&S->Sint, (uint16_t)(fftLen / 2u));
  if (st != ARM_MATH_SUCCESS) return st;
  S->fftLenRFFT = fftLen;
  S->pTwiddleRFFT = host_tw_rfft(l + 1);
  return ARM_MATH_SUCCESS;
}
#define ARMFFT_RINIT_N(N) \
  arm_status arm_rfft_fast_init_##N##_f32(arm_rfft_fast_instance_f32 *S) { return arm_rfft_fast_init_f32(S, N); }
ARMFFT_RINIT_N(32) ARMFFT_RINIT_N(64) ARMFFT_RINIT_N(128) ARMFFT_RINIT_N(256) ARMFFT_RINIT_N(512) ARMFFT_RINIT_N(1024)
ARMFFT_RINIT_N(2048) ARMFFT_RINIT_N(4096)

static int rfft_log(const arm_rfft_fast_instance_f32 *S)
{
  if (!S || S->fftLenRFFT < 32 || !is_pow2(S->fftLenRFFT)) return -1;
  int l = ilog2u(S->fftLenRFFT);
  return (l - 1 >= kMinLog && l <= kMaxLog && S->Sint.fftLen == S->fftLenRFFT / 2) ? l : -1;
}

arm_status arm_rfft_fast_f32_dev(const arm_rfft_fast_instance_f32 *S, const float32_t *p, float32_t *pOut, uint32_t batch,
                                 uint8_t ifft, void *stream)
{
  if (!S || !p || !pOut) { set_error("arm_rfft_fast_f32_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (!aligned16(p, "arm_rfft_fast_f32_dev") || !aligned16(pOut, "arm_rfft_fast_f32_dev")) return ARM_MATH_ARGUMENT_ERROR;
  int l = rfft_log(S);
  if (l < 0) { set_error("arm_rfft_fast_f32_dev: unsupported instance"); return ARM_MATH_ARGUMENT_ERROR; }
  if (p == pOut) { set_error("arm_rfft_fast_f32_dev: input and output must be different buffers"); return ARM_MATH_ARGUMENT_ERROR; }
  const float2 *tw = dev_tw_f32(l - 1), *twr = dev_tw_rfft(l);
  if (!tw || !twr) return ARM_MATH_ARGUMENT_ERROR;
  hipError_t e = launch_rfft_f32(l, p, pOut, tw, twr, batch, ifft != 0, (hipStream_t)stream);
  return hip_ok(e, "rfft_f32 launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

arm_status arm_rfft_fast_batch_f32(const arm_rfft_fast_instance_f32 *S, float32_t *p, float32_t *pOut, uint32_t batch, uint8_t ifft)
{
  if (!S || !p || !pOut) { set_error("arm_rfft_fast_batch_f32: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (rfft_log(S) < 0) { set_error("arm_rfft_fast_batch_f32: unsupported instance"); return ARM_MATH_ARGUMENT_ERROR; }
  if (batch == 0) return ARM_MATH_SUCCESS;
  int dev;
  if (!current_device(&dev)) return ARM_MATH_ARGUMENT_ERROR;
  std::lock_guard<std::mutex> lk(stage_mu(dev));
  size_t bytes = (size_t)S->fftLenRFFT * batch * sizeof(float);
  float *din = (float *)stage_buffer(dev, bytes, 0), *dout = (float *)stage_buffer(dev, bytes, 1);
  if (!din || !dout) return ARM_MATH_ARGUMENT_ERROR;
  if (!hip_ok(hipMemcpy(din, p, bytes, hipMemcpyHostToDevice), "hipMemcpy(H2D)")) return ARM_MATH_ARGUMENT_ERROR;
  arm_status s = arm_rfft_fast_f32_dev(S, din, dout, batch, ifft, nullptr);
  if (s != ARM_MATH_SUCCESS) return s;
  if (!hip_ok(hipMemcpy(pOut, dout, bytes, hipMemcpyDeviceToHost), "hipMemcpy(D2H)")) return ARM_MATH_ARGUMENT_ERROR;
  return ARM_MATH_SUCCESS;
}

void arm_rfft_fast_f32(const arm_rfft_fast_instance_f32 *S, float32_t *p, float32_t *pOut, uint8_t ifftFlag)
{
  if (rfft_log(S) < 0) return;
  if (arm_rfft_fast_batch_f32(S, p, pOut, 1, ifftFlag) != ARM_MATH_SUCCESS) {
    fprintf(stderr, "libarmfft_mi355: arm_rfft_fast_f32 cannot run (%s)\n", t_err);
    abort();
  }
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// msb measurement wire format (msb_funcs.c:47-233): host-side bit packing, no GPU
// ------------------------------------------------------------------------------------------
namespace {
struct BitOut {
  uint8_t *buf;
  int cap_bits, pos;
  bool ok;
  // MSB-first, the low num_bits of val (msb_funcs.c:47-77)
  void put(uint64_t val, int num_bits)
  {
    for (int i = num_bits - 1; i >= 0; --i) {
      if (pos >= cap_bits) { ok = false; return; }
      const uint8_t mask = (uint8_t)(1u << (7 - (pos & 7)));
      if ((val >> i) & 1u) buf[pos >> 3] |= mask; else buf[pos >> 3] &= (uint8_t)~mask;
      ++pos;
    }
  }
};
struct BitIn {
  const uint8_t *buf;
  int size_bits, pos;
  // msb_funcs.c:14-43: a bit past the end reads as -1, which ORs all ones from that position up
  int64_t get(unsigned num_bit, bool is_signed)
  {
    uint64_t ans = 0;
    for (unsigned i = 0; i < num_bit; i++) {
      int64_t bit;
      if (pos >= size_bits) bit = -1;
      else { bit = (buf[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; }
      ans |= (uint64_t)bit << (num_bit - (i + 1));
    }
    if (is_signed && ((ans >> (num_bit - 1)) & 1u)) ans |= ~(uint64_t)0 << num_bit;
    return (int64_t)ans;
  }
};
// per-satellite fields (msb_funcs.c:108-113, :135-140): float division, truncation toward zero
void put_sat(BitOut &o, const bb_meas_sat_t &s, int phase_bits)
{
  o.put((uint64_t)(int64_t)(s.code_phase / 9.54e-7f), phase_bits);
  o.put((uint64_t)(int64_t)(s.doppler / 2.0f), 13);
  o.put((uint64_t)(int64_t)((s.cno - 28) / 4), 4);
}
}  // namespace

extern "C" int write_bb_msb(const bb_meas_t *m, uint8_t *bin_buff, int buff_size)
{
  if (!m || !bin_buff || buff_size <= 0 || m->num_sat < 0 || m->num_sat > 20) { set_error("write_bb_msb: bad argument"); return -1; }
  memset(bin_buff, 0, (size_t)buff_size);
  BitOut o{bin_buff, buff_size * 8, 0, true};
  uint64_t mask = 0;
  for (int i = 0; i < m->num_sat; i++)
    if (m->sats[i].constellation == SYS_GPS) mask |= (uint64_t)1 << ((m->sats[i].prn - 1) & 63);
  o.put(mask, 32);
  for (int i = 0; i < m->num_sat; i++)
    if (m->sats[i].constellation == SYS_GPS) put_sat(o, m->sats[i], 20);
  int gal = 0;
  for (int i = 0; i < m->num_sat; i++)
    if (m->sats[i].constellation == SYS_GAL) { gal = 1; break; }
  o.put((uint64_t)gal, 3);
  if (gal) {
    mask = 0;
    for (int i = 0; i < m->num_sat; i++)
      if (m->sats[i].constellation == SYS_GAL) mask |= (uint64_t)1 << ((m->sats[i].prn - 1) & 63);
    o.put(mask, 50);
    for (int i = 0; i < m->num_sat; i++)
      if (m->sats[i].constellation == SYS_GAL) put_sat(o, m->sats[i], 22);
  }
  if (!o.ok) { set_error("write_bb_msb: buffer of %d bytes is too small", buff_size); return -1; }
  return o.pos / 8;
}

extern "C" void read_bb_msb(uint8_t *bin_buff, int bin_size, bb_meas_t *m)
{
  if (!bin_buff || !m || bin_size < 0) { set_error("read_bb_msb: bad argument"); return; }
  BitIn in{bin_buff, bin_size * 8, 0};
  auto section = [&](int first, unsigned mask_bits, unsigned phase_bits, uint16_t constellation) {
    const uint64_t mask = (uint64_t)in.get(mask_bits, false);
    int prns[64], n = 0;
    for (unsigned i = 0; i < mask_bits; i++)
      if ((mask >> i) & 1u) prns[n++] = (int)i + 1;
    for (int i = 0; i < n; i++) {
      const int64_t code_phase = in.get(phase_bits, false), doppler = in.get(13, true), cno = in.get(4, false);
      if (first + i >= 20) continue;  // the reference would overrun sats[]
      bb_meas_sat_t &s = m->sats[first + i];
      s.time_tag = (double)m->time + m->sec;
      s.pseudorange = 0;
      s.code_phase = ((float)code_phase) * 9.54e-7f;
      s.cno = (float)(cno * 4 + 28);
      s.doppler = (float)(2 * doppler);
      s.prn = (uint16_t)prns[i];
      s.constellation = constellation;
    }
    return n;
  };
  m->num_sat = section(0, 32, 20, SYS_GPS);
  if (in.get(3, false) == 1) m->num_sat += section(m->num_sat, 50, 22, SYS_GAL);
}

extern "C" int gnss_acq_to_meas(const gnss_acq_result *best, int n, const float32_t *dopplers_hz, int n_dop, int fft_len, float ratio_min,
                                float cno_offset_db, uint16_t constellation, bb_meas_t *meas)
{
  if (!best || n < 0 || !dopplers_hz || n_dop <= 0 || fft_len <= 0 || !meas || meas->num_sat < 0 || meas->num_sat > 20) {
    set_error("gnss_acq_to_meas: bad argument");
    return -1;
  }
  int added = 0;
  for (int i = 0; i < n && meas->num_sat < 20; i++) {
    const gnss_acq_result &r = best[i];
    if (!(r.peak / r.second > ratio_min)) continue;                       // CMSIS.cpp:737
    if (r.doppler_bin < 0 || r.doppler_bin >= n_dop) { set_error("gnss_acq_to_meas: row %d has Doppler bin %d", i, r.doppler_bin); return -1; }
    bb_meas_sat_t &s = meas->sats[meas->num_sat++];
    s.prn = (uint16_t)r.prn;
    s.code_phase = r.code_phase_interp / (float)fft_len;                  // CMSIS.cpp:739
    s.doppler = dopplers_hz[r.doppler_bin];
    s.cno = r.snr_db + cno_offset_db;                                      // CMSIS.cpp:729, :741
    s.constellation = constellation;
    added++;
  }
  return added;
}
