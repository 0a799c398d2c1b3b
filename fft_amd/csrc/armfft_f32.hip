// armfft_f32.hip -- batched complex and real f32 FFT kernels for gfx950 (MI355X).
//
// Replaces the compute of arm_cfft_f32 (arm_cfft_f32.c:1243-1298: radix-8 butterflies
// arm_cfft_radix8_f32.c:51-291, radix-2/4 first stage arm_cfft_f32.c:846-1201, table bit reversal
// arm_bitreversal2.c:84-108) and arm_rfft_fast_f32 (arm_rfft_fast_f32.c:675-699, split/merge
// :316-462) with an LDS-resident Stockham transform (fft_core_f32.hpp): each transform is read
// from HBM once and written once (16 bytes per point), twiddles come from a device-generated,
// L2-resident table and never from HBM in steady state.
#include <atomic>

#include "armfft_internal.h"
#include "fft_core_f32.hpp"

namespace armfft {

// Position at which arm_cfft_f32 leaves bin k when bitReverseFlag == 0: digits of k are
// (d0 | radix-8 digits, least significant first) with d0 the radix-2/4 first-stage digit
// (arm_cfft_f32.c:1263-1280); the position has the same digits most significant first.
template <int LOGN>
__device__ __forceinline__ int scrambled_pos(int k)
{
  constexpr int L0 = LOGN % 3, N8 = LOGN / 3;
  int d0 = k & ((1 << L0) - 1);
  int rest = k >> L0;
  int low = 0;
#pragma unroll
  for (int s = 0; s < N8; s++) {
    low = (low << 3) | (rest & 7);
    rest >>= 3;
  }
  return (d0 << (3 * N8)) | low;
}

// Per-size tuning (same-box A/B measurements on MI355X; history in DESIGN.md 3.1):
//   ONESHOT   one transform per workgroup of N/16 threads, no loop: first-pass loads and last-pass
//             stores through a buffer descriptor (scalar leg offsets, one per-lane offset
//             register), twiddles from register-held powers (TwPow, LAZYMASK = passes served that
//             way, NESTED8 = its temp-free radix-8 form).  ~80-126 VGPRs, so 4 workgroups of 4096
//             points (8 of 2048, 16 of 1024) share a CU and overlap each other's load, compute
//             and store phases -- measured faster than persistent workgroups that prefetch their
//             next transform (which need 170-230 VGPRs): 4096 0.85 -> 0.78 ms per 65536.
//   MINW      __launch_bounds__ waves per SIMD (register budget 512 / MINW)
//   AUX       cache policy of the HBM accesses: 2 = nt (streamed once) where it measured faster
// Single-wave workgroups (64 threads) everywhere below 2048 points: a wave that never waits for
// another wave at a barrier streams best (same-box A/B: 16 points 5.65 -> 6.03 TB/s, 32: 5.23 ->
// 5.55, 64: 5.25 -> 5.45, 128: 5.49 -> 5.62).
// 256 and 512 points run the one-shot path that way (4 / 2 transforms per wave): every wave
// instruction then touches whole 128-byte lines (256) or one 512-byte run (512), so nt applies:
// 5.4 -> 5.8 TB/s and 5.3 -> 5.9 TB/s.  128 points stay on the plain kernel (8 transforms per
// wave, table twiddles): there a wave instruction covers 64-byte segments, nt costs 15%, the
// one-shot form without nt and LDS staging measured the same or slower.  16, 32 and 64 points are
// staged through LDS with 16-byte accesses (nt pays at 16 points only).
//   SPLIT     (one-shot sizes from 4096 points) the half-footprint exchange of fft_core_f32.hpp: real and imaginary parts
//             cross the workgroup one after the other through an LDS image of N floats, so twice as many workgroups share
//             a CU (LDS was the limit: 4096 points 4 -> 7 per CU, 16384 points 1 -> 2) and their HBM phases overlap each
//             other's passes.  Same-box A/B (tools/fft_lab2, 2 GiB batches): 4096 0.770 -> 0.722 ms (frac 0.70 -> 0.74),
//             8192 0.778 -> 0.743 (0.69 -> 0.72), 16384 1.126 -> 0.785 (0.48 -> 0.68; needs the 64-VGPR build, MINW 8,
//             for two 1024-thread workgroups per CU).  2048 points lose 3 % and 1024 gain 1 %: left as they were.
//   NESTED16  temp-free radix-16 twiddle products (see twpow_mul), SPLIT kernels only
struct TuneBase {
  static constexpr unsigned LAZYMASK = 0;
  static constexpr int MINW = 4, AUX = 0;
  static constexpr bool NESTED8 = false, NESTED16 = false, ONESHOT = false, SPLIT = false;
};
template <int LOGN> struct Tune : TuneBase {};
template <> struct Tune<4> : TuneBase { static constexpr int AUX = 2; };
template <> struct Tune<8> : TuneBase { static constexpr unsigned LAZYMASK = ~0u; static constexpr int AUX = 2; static constexpr bool ONESHOT = true; };
template <> struct Tune<9> : TuneBase { static constexpr unsigned LAZYMASK = ~0u; static constexpr int AUX = 2; static constexpr bool ONESHOT = true; };
template <> struct Tune<10> : TuneBase { static constexpr unsigned LAZYMASK = ~0u; static constexpr int AUX = 2; static constexpr bool ONESHOT = true; };
template <> struct Tune<11> : TuneBase { static constexpr unsigned LAZYMASK = ~0u; static constexpr int MINW = 3, AUX = 2; static constexpr bool ONESHOT = true; };
// Points per thread of the SPLIT sizes (Plan::E): 32 at 16384 points -- two 512-thread workgroups per CU at 124 VGPRs instead of two
// 1024-thread ones at 64: half the waves at every exchange barrier, twice the independent work between two barriers.  Same-box
// A/B: 16384 points 0.690 -> 0.705 of the roofline; 8192 points +-0 (0.72), 4096 points 0.745 -> 0.62 (20 bytes of spills at 128
// VGPRs, and seven two-wave workgroups per CU cover less than seven four-wave ones): those stay at 16.
#ifndef ARMFFT_F32_E12
#define ARMFFT_F32_E12 16
#endif
#ifndef ARMFFT_F32_E13
#define ARMFFT_F32_E13 16
#endif
template <> struct Tune<12> : TuneBase { static constexpr unsigned LAZYMASK = ~0u; static constexpr int AUX = 2; static constexpr bool ONESHOT = true, SPLIT = true, NESTED16 = true; };
template <> struct Tune<13> : TuneBase { static constexpr unsigned LAZYMASK = ~0u; static constexpr int AUX = 2; static constexpr bool ONESHOT = true, SPLIT = true; };
#ifndef ARMFFT_F32_E14
#define ARMFFT_F32_E14 32
#endif
template <> struct Tune<14> : TuneBase { static constexpr unsigned LAZYMASK = ~0u; static constexpr int MINW = 8 * 16 / ARMFFT_F32_E14, AUX = 2; static constexpr bool ONESHOT = true, SPLIT = true, NESTED16 = true; };

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Plan of the complex kernel: one-shot sizes below 4096 points use N/16 threads (at least one
// wave: 256 and 512 points put 4 / 2 transforms in it); the staged and plain kernels below 256
// points use single-wave workgroups too.
// (packed arithmetic everywhere but at 128, 512, 2048, 4096 and 8192 points (same-box A/B per size): see ArPk / ArSc)
template <int LOGN> constexpr int kSplitE = LOGN == 14 ? ARMFFT_F32_E14 : LOGN == 13 ? ARMFFT_F32_E13 : LOGN == 12 ? ARMFFT_F32_E12 : 16;   // points per thread
template <int LOGN> using CPlan = Plan<LOGN, LOGN >= 12 ? (1 << LOGN) / kSplitE<LOGN> : LOGN <= 7 ? 64 : (Tune<LOGN>::ONESHOT && LOGN < 12) ? ((1 << LOGN) / 16 < 64 ? 64 : (1 << LOGN) / 16) : 0, false, !(LOGN == 7 || LOGN == 9 || (LOGN >= 11 && LOGN <= 13)), kSplitE<LOGN>>;

// Buffer descriptor over one transform (gfx9 raw buffer: stride 0, range in bytes, 32-bit data
// format): address = base + scalar offset + per-lane offset, so the 16 legs of a thread share ONE
// per-lane offset register and differ only in a scalar -- no 64-bit address arithmetic in vector
// registers (2 VALU + a register pair per global access otherwise).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t transform_rsrc(const float2 *base, uint32_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(base), 0, (int)bytes, 0x00020000);
}

// AUX: cache policy of the access (0 default, 2 = nt: streamed once, keep it out of the way)
template <int LOGN, bool INV, int ORDER, int AUX = 0>
struct BufferStore {
  static constexpr bool SPLIT = true;
  __amdgpu_buffer_rsrc_t rsrc;
  float inv_n;
  // t = transform inside the descriptor (0 when it covers one transform); a store past the
  // descriptor's range (ragged last workgroup) is dropped by the hardware
  __device__ __forceinline__ void operator()(int t, int e0, int roff, float2 x) const
  {
    if (INV) x = make_float2(x.x * inv_n, -x.y * inv_n);
    u32x2 w;
    w[0] = __float_as_uint(x.x);
    w[1] = __float_as_uint(x.y);
    if constexpr (ORDER == 0) {
      __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, ((t << LOGN) + e0) * 8, roff * 8, AUX);
    } else {
      int e = e0 + roff;
      if (ORDER == 1) e = scrambled_pos<LOGN>(e);
      if (ORDER == 2) e = (int)(__brev((unsigned)e) >> (32 - LOGN));
      __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, ((t << LOGN) + e) * 8, 0, AUX);
    }
  }
};

template <int LOGN, bool INV>
__global__ __launch_bounds__(CPlan<LOGN>::T, Tune<LOGN>::MINW) void cfft_f32_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                                                 uint32_t batch, uint32_t ngroups, int order)
{
  using PL = CPlan<LOGN>;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const float inv_n = 1.0f / (float)PL::N;
  const TwTable twr{tw};
  const uint32_t g0 = blockIdx.x + blockIdx.y * gridDim.x;
  if constexpr (Tune<LOGN>::ONESHOT) {
    // one group of TPB transforms per workgroup, no loop: data loads first, then the twiddle bases
    // (needed from pass 1 on), everything through ONE buffer descriptor over the group's valid
    // transforms -- loads past it read zeros, stores past it are dropped, so the ragged last group
    // needs no branch
    if (g0 >= ngroups) return;
    const int tid = threadIdx.x;
    const uint32_t t0 = g0 * PL::TPB;
    const uint32_t nvalid = batch - t0 < (uint32_t)PL::TPB ? batch - t0 : (uint32_t)PL::TPB;
    const __amdgpu_buffer_rsrc_t rs = transform_rsrc(data + (size_t)t0 * PL::N, nvalid * PL::N * (uint32_t)sizeof(float2));
    float2 v[16];
    for_each_first_pass_input_split<PL>(tid, [&](int slot, int t, int j, int roff) {
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, ((t << LOGN) + j) * 8, roff * 8, Tune<LOGN>::AUX);
      float2 x = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
      if (INV) x.y = -x.y;
      v[slot] = x;
    });
    TwPow<PL, Tune<LOGN>::LAZYMASK, Tune<LOGN>::NESTED8> twp;
    twp.preload(tw, tid);
    if (order == 0) {
      BufferStore<LOGN, INV, 0, Tune<LOGN>::AUX> st{rs, inv_n};
      fft_block_preloaded<PL, false>(v, lds, twp, tid, st);
    } else if (order == 1) {
      BufferStore<LOGN, INV, 1, Tune<LOGN>::AUX> st{rs, inv_n};
      fft_block_preloaded<PL, false>(v, lds, twp, tid, st);
    } else {
      BufferStore<LOGN, INV, 2, Tune<LOGN>::AUX> st{rs, inv_n};
      fft_block_preloaded<PL, false>(v, lds, twp, tid, st);
    }
    return;
  }
  if constexpr (LOGN <= 6) {
    // 16 and 32 points (and 64, in single-wave workgroups): a thread's first-pass legs are 8 bytes at a 128/256-byte stride across
    // lanes, so the workgroup's transforms (one contiguous 32 KiB block) are staged through LDS
    // with 16-byte coalesced accesses instead, in and out
    const uint32_t g = g0;
    if (g >= ngroups) return;
    const int tid = threadIdx.x;
    const uint32_t t0 = g * PL::TPB;
    const uint32_t nvalid = (batch - t0 >= (uint32_t)PL::TPB ? (uint32_t)PL::TPB : batch - t0) * PL::N;  // points
    float2 *__restrict__ p = data + (size_t)t0 * PL::N;
    constexpr int PAIRS = PL::TPB * PL::N / 2;
    for (int i = tid; i < PAIRS; i += PL::T) {
      const int e2 = 2 * i, t = e2 >> LOGN, e = e2 & (PL::N - 1);
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((uint32_t)e2 < nvalid) { if (Tune<LOGN>::AUX == 2) { const f32x4 y = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p + e2)); x = make_float4(y[0], y[1], y[2], y[3]); } else x = *reinterpret_cast<const float4 *>(p + e2); }
      if (INV) { x.y = -x.y; x.w = -x.w; }
      *reinterpret_cast<float4 *>(&lds[PL::idx(t, e)]) = x;
    }
    __syncthreads();
    auto ld = [&](int t, int e) -> float2 { return lds[PL::idx(t, e)]; };
    auto st = [&](int t, int e, float2 v) {
      if (INV) v = make_float2(v.x * inv_n, -v.y * inv_n);
      if (order == 1) e = scrambled_pos<LOGN>(e);
      if (order == 2) e = (int)(__brev((unsigned)e) >> (32 - LOGN));
      lds[PL::idx(t, e)] = v;
    };
    fft_block_tw<PL, true, true>(lds, twr, tid, ld, st);
    __syncthreads();
    for (int i = tid; i < PAIRS; i += PL::T) {
      const int e2 = 2 * i, t = e2 >> LOGN, e = e2 & (PL::N - 1);
      if ((uint32_t)e2 < nvalid) { const float4 y = *reinterpret_cast<const float4 *>(&lds[PL::idx(t, e)]); if (Tune<LOGN>::AUX == 2) { f32x4 z; z[0] = y.x; z[1] = y.y; z[2] = y.z; z[3] = y.w; __builtin_nontemporal_store(z, reinterpret_cast<f32x4 *>(p + e2)); } else *reinterpret_cast<float4 *>(p + e2) = y; }
    }
    return;
  }
  if (g0 < ngroups) {
    const int tid = threadIdx.x;
    const uint32_t t0 = g0 * PL::TPB;
    float2 *__restrict__ p = data + (size_t)t0 * PL::N;     // uniform base, 32-bit per-lane offsets
    auto ld = [&](int t, int e) -> float2 {
      float2 v = make_float2(0.f, 0.f);
      if (t0 + t < batch) v = p[t * PL::N + e];
      if (INV) v.y = -v.y;
      return v;
    };
    auto st = [&](int t, int e, float2 v) {
      if (INV) v = make_float2(v.x * inv_n, -v.y * inv_n);
      if (order == 1) e = scrambled_pos<LOGN>(e);
      if (order == 2) e = (int)(__brev((unsigned)e) >> (32 - LOGN));
      if (t0 + t < batch) p[t * PL::N + e] = v;
    };
    fft_block_tw<PL, false, false>(lds, twr, tid, ld, st);
  }
}

// One-shot kernel with the half-footprint exchange (Tune::SPLIT sizes).  The output order is a template parameter here:
// with the three store forms in one kernel the register allocation is the worst of the three paths (112-123 VGPRs at 4096
// points instead of ~70), which halves the workgroups per CU the split exchange exists to double.
template <int LOGN, bool INV, int ORDER>
__global__ __launch_bounds__(CPlan<LOGN>::T, Tune<LOGN>::MINW) void cfft_f32_split_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw,
                                                                                       uint32_t batch)
{
  using PL = CPlan<LOGN>;
  static_assert(PL::TPB == 1, "one transform per workgroup");
  extern __shared__ __attribute__((aligned(16))) float ldsf[];
  const uint32_t g0 = blockIdx.x + blockIdx.y * gridDim.x;
  if (g0 >= batch) return;
  const int tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = transform_rsrc(data + (size_t)g0 * PL::N, PL::N * (uint32_t)sizeof(float2));
  float2 v[PL::E];
  for_each_first_pass_input_split<PL>(tid, [&](int slot, int, int j, int roff) {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, roff * 8, Tune<LOGN>::AUX);
    float2 x = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
    if (INV) x.y = -x.y;
    v[slot] = x;
  });
  BufferStore<LOGN, INV, ORDER, Tune<LOGN>::AUX> st{rs, 1.0f / (float)PL::N};
  fft_block_split<PL, Tune<LOGN>::NESTED8, Tune<LOGN>::NESTED16>(v, ldsf, tw, tid, st);
}

// ---- real FFT: N real points <-> packed half spectrum, through an N/2-point complex FFT ----
// LOGM = log2(N/2).  Forward: z = FFT(x as N/2 complex) to LDS, then the split of stage_rfft_f32
// (arm_rfft_fast_f32.c:316-402).  Inverse: merge_rfft_f32 (:405-462) fused into the load of the
// inverse complex FFT.
// Per-size tuning of the real-FFT kernel (LOGM = log2 of the complex length N/2): from 256 complex
// points on, the one-shot form of the complex kernel (TwPow, buffer addressing, nt policy) -- one
// transform per workgroup of M/16 threads from 1024 on, 4 / 2 transforms in one wave at 256 / 512
// (512 real points 5.0 -> 5.45 TB/s, 1024: 4.5 -> 5.48); below, the plain kernel.  (Staging the
// 32-point case through LDS like the 16-point complex kernel measured no gain: 3.7 TB/s both ways.)
// SUMSPLIT: the split / merge factors of a thread's 16 bins come from ONE per-lane table load per
// butterfly (bin e0) and angle sums with table entries at compile-time offsets (uniform address:
// scalar loads) instead of 16 per-lane loads; same-box A/B per size, real points: 1024 +5 %,
// 2048 +4.5 %, 4096 +3.6 %, 16384 +4 %, but 512 -3 % and 8192 -3 % (left on plain loads).
template <int LOGM> struct RTune { static constexpr int MINW = 4, AUX = 0; static constexpr bool NESTED8 = false, ONESHOT = false, SUMSPLIT = false; };
template <> struct RTune<8> { static constexpr int MINW = 4, AUX = 2; static constexpr bool NESTED8 = false, ONESHOT = true, SUMSPLIT = false; };
template <> struct RTune<9> { static constexpr int MINW = 4, AUX = 2; static constexpr bool NESTED8 = false, ONESHOT = true, SUMSPLIT = true; };
template <> struct RTune<10> { static constexpr int MINW = 4, AUX = 2; static constexpr bool NESTED8 = false, ONESHOT = true, SUMSPLIT = true; };
template <> struct RTune<11> { static constexpr int MINW = 4, AUX = 2; static constexpr bool NESTED8 = false, ONESHOT = true, SUMSPLIT = true; };
template <> struct RTune<12> { static constexpr int MINW = 4, AUX = 2; static constexpr bool NESTED8 = false, ONESHOT = true, SUMSPLIT = false; };
template <> struct RTune<13> { static constexpr int MINW = 4, AUX = 2; static constexpr bool NESTED8 = true, ONESHOT = true, SUMSPLIT = true; };

// (64 and 128 real points in single-wave workgroups as well: +3.5 % / +6 %)
template <int LOGM> using RPlan = Plan<LOGM, (LOGM == 5 || LOGM == 6 || LOGM == 8 || LOGM == 9) ? 64 : (LOGM == 10 || LOGM == 11) ? (1 << LOGM) / 16 : 0>;

// One transform per workgroup, no loop (RTune::ONESHOT).  Forward: every thread splits exactly the
// 16 bins its own last pass produced, so Z[k] never leaves its registers and only Z[M-k] is read
// back from LDS; the split factors are fetched before the transform.  Inverse: bins k and M-k are
// loaded raw (the mirrored one at a descending scalar offset) and merged in registers.
template <int LOGM, bool INV>
__device__ __forceinline__ void rfft_f32_oneshot(const float2 *__restrict__ in, float2 *__restrict__ out, float2 *lds,
                                                 const float2 *__restrict__ tw, const float2 *__restrict__ twr, uint32_t g,
                                                 uint32_t batch)
{
  using PL = RPlan<LOGM>;
  constexpr int M = PL::N;
  // one descriptor over the group's valid transforms (TPB = 1 from 1024 complex points on, 4 / 2 in
  // the single wave of 256 / 512): loads past it read zeros, stores past it are dropped
  const uint32_t t0 = g * PL::TPB;
  const uint32_t nvalid = batch - t0 < (uint32_t)PL::TPB ? batch - t0 : (uint32_t)PL::TPB;
  const uint32_t BYTES = nvalid * M * (uint32_t)sizeof(float2);
  constexpr int AUX = RTune<LOGM>::AUX;
  const int tid = threadIdx.x;
  auto load2 = [](const __amdgpu_buffer_rsrc_t &rs, int voff, int soff) -> float2 {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX);
    return make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
  };
  const __amdgpu_buffer_rsrc_t ri = transform_rsrc(in + (size_t)t0 * M, BYTES);
  const __amdgpu_buffer_rsrc_t ro = transform_rsrc(out + (size_t)t0 * M, BYTES);
  TwPow<PL, ~0u, RTune<LOGM>::NESTED8> twp;
  if constexpr (!INV) {
    float2 v[16];
    for_each_first_pass_input_split<PL>(tid, [&](int slot, int t, int j, int roff) { v[slot] = load2(ri, ((t << LOGM) + j) * 8, roff * 8); });
    twp.preload(tw, tid);
    float2 csr[16];  // split factors (sin, cos) of the 16 bins this thread's last pass produces
    // SUMSPLIT: ONE per-lane load (bin e0) per butterfly; the factor of bin e0 + roff is its angle
    // sum with the table entry `roff`, whose address is uniform -- a scalar load, not one more
    // vector-memory instruction in front of the stores
    float2 base;
    for_each_last_pass_output<PL>(tid, [&](int slot, int, int e0, int roff) {
      if (roff == 0 || !RTune<LOGM>::SUMSPLIT) {
        base = twr[e0 + roff];
        csr[slot] = base;
      } else {
        const float2 c = twr[roff];
        csr[slot] = make_float2(base.x * c.y + base.y * c.x, base.y * c.y - base.x * c.x);  // (sin, cos)(a + b)
      }
    });
    auto st = [&](int t, int e, float2 x) { lds[PL::idx(t, e)] = x; };
    fft_block_preloaded<PL, true>(v, lds, twp, tid, st);
    __syncthreads();
    // split: bin k from Z[k] (still in v[]) and Z[M-k]
    for_each_last_pass_output<PL>(tid, [&](int slot, int t, int e0, int roff) {
      const int k = e0 + roff;
      const float2 A = v[slot];
      const float2 B = lds[PL::idx(t, (M - k) & (M - 1))];
      const float2 c = csr[slot];
      const float a = B.x - A.x, b = B.y + A.y;
      float2 o;
      o.x = 0.5f * (A.x + B.x + c.x * a + c.y * b);
      o.y = 0.5f * (A.y - B.y + c.y * a - c.x * b);
      if (k == 0) o = make_float2(A.x + A.y, A.x - A.y);  // 0.5*((2 re) +- (2 im)), exact
      u32x2 w;
      w[0] = __float_as_uint(o.x);
      w[1] = __float_as_uint(o.y);
      __builtin_amdgcn_raw_buffer_store_b64(w, ro, ((t << LOGM) + e0) * 8, roff * 8, AUX);
      if ((slot & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    });
  } else {
    constexpr int R0 = 1 << PL::lr(0), S = M / R0, TOP = (R0 - 1) * S;  // first-pass legs: j + r*S, j < S
    float2 A[16], B[16], cm[16];
    for_each_first_pass_input_split<PL>(tid, [&](int slot, int t, int j, int roff) {
      A[slot] = load2(ri, ((t << LOGM) + j) * 8, roff * 8);
      B[slot] = load2(ri, ((t << LOGM) + M - j - TOP) * 8, (TOP - roff) * 8);   // bin M - e; e == 0 reads the point after the transform: unused
    });
    float2 base;
    for_each_first_pass_input_split<PL>(tid, [&](int slot, int, int j, int roff) {
      if (roff == 0 || !RTune<LOGM>::SUMSPLIT) {
        base = twr[j + roff];
        cm[slot] = base;
      } else {
        const float2 c = twr[roff];  // uniform address: scalar load
        cm[slot] = make_float2(base.x * c.y + base.y * c.x, base.y * c.y - base.x * c.x);
      }
    });
    twp.preload(tw, tid);
    float2 v[16];
    for_each_first_pass_input_split<PL>(tid, [&](int slot, int, int j, int roff) {
      const float2 a2 = A[slot], b2 = B[slot], c = cm[slot];
      float2 o;
      if ((j | roff) == 0) {
        o = make_float2(0.5f * (a2.x + a2.y), 0.5f * (a2.x - a2.y));
      } else {
        const float a = a2.x - b2.x, b = a2.y + b2.y;
        o.x = 0.5f * (a2.x + b2.x - c.x * a - c.y * b);
        o.y = 0.5f * (a2.y - b2.y + c.y * a - c.x * b);
      }
      o.y = -o.y;  // conjugate in (inverse through the forward kernel)
      v[slot] = o;
    });
    BufferStore<LOGM, true, 0, AUX> st{ro, 1.0f / (float)M};
    fft_block_preloaded<PL, false>(v, lds, twp, tid, st);
  }
}

template <int LOGM, bool INV>
__global__ __launch_bounds__(RPlan<LOGM>::T, RTune<LOGM>::MINW) void rfft_f32_kernel(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                                 const float2 *__restrict__ tw, const float2 *__restrict__ twr,
                                                                 uint32_t batch, uint32_t ngroups)
{
  using PL = RPlan<LOGM>;
  constexpr int M = PL::N;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  if constexpr (RTune<LOGM>::ONESHOT) {
    const uint32_t g = blockIdx.x + blockIdx.y * gridDim.x;
    if (g < ngroups) rfft_f32_oneshot<LOGM, INV>(in, out, lds, tw, twr, g, batch);
    return;
  }
  const int tid = threadIdx.x;
  if constexpr (LOGM == 4 && !INV) {
    // 32 real points, forward: a thread owns one whole 16-point transform, so its loads would be
    // 8 bytes at a 128-byte stride across lanes and its split stores 8 bytes per lane -- the
    // workgroup's contiguous block is staged through LDS with 16-byte accesses instead, and the
    // split handles two adjacent bins per lane (16-byte stores): 3.7 -> 4.5 TB/s
    const uint32_t g = blockIdx.x + blockIdx.y * gridDim.x;
    if (g >= ngroups) return;
    const uint32_t t0 = g * PL::TPB;
    const uint32_t nvalid = (batch - t0 >= (uint32_t)PL::TPB ? (uint32_t)PL::TPB : batch - t0) * M;  // points
    const float2 *__restrict__ pi = in + (size_t)t0 * M;
    float2 *__restrict__ po = out + (size_t)t0 * M;
    constexpr int PAIRS = PL::TPB * M / 2;
    for (int i = tid; i < PAIRS; i += PL::T) {
      const int e2 = 2 * i, t = e2 >> LOGM, e = e2 & (M - 1);
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((uint32_t)e2 < nvalid) { const f32x4 y = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(pi + e2)); x = make_float4(y[0], y[1], y[2], y[3]); }
      *reinterpret_cast<float4 *>(&lds[PL::idx(t, e)]) = x;
    }
    __syncthreads();
    auto ld = [&](int t, int e) -> float2 { return lds[PL::idx(t, e)]; };
    auto st = [&](int t, int e, float2 v) { lds[PL::idx(t, e)] = v; };
    fft_block<PL, true, true>(lds, tw, tid, ld, st);
    __syncthreads();
    for (int i = tid; i < PAIRS; i += PL::T) {
      const int e2 = 2 * i, t = e2 >> LOGM, k = e2 & (M - 1);  // bins k (even) and k + 1
      if ((uint32_t)e2 >= nvalid) continue;
      const float4 a4 = *reinterpret_cast<const float4 *>(&lds[PL::idx(t, k)]);
      const float2 A0 = make_float2(a4.x, a4.y), A1 = make_float2(a4.z, a4.w);
      const float2 B0 = lds[PL::idx(t, (M - k) & (M - 1))], B1 = lds[PL::idx(t, M - k - 1)];
      const float4 c4 = *reinterpret_cast<const float4 *>(twr + k);  // (sin, cos) of bins k, k + 1
      float4 o;
      {
        const float a = B0.x - A0.x, b = B0.y + A0.y;
        o.x = 0.5f * (A0.x + B0.x + c4.x * a + c4.y * b);
        o.y = 0.5f * (A0.y - B0.y + c4.y * a - c4.x * b);
        if (k == 0) { o.x = A0.x + A0.y; o.y = A0.x - A0.y; }  // 0.5*((2 re) +- (2 im)), exact
      }
      {
        const float a = B1.x - A1.x, b = B1.y + A1.y;
        o.z = 0.5f * (A1.x + B1.x + c4.z * a + c4.w * b);
        o.w = 0.5f * (A1.y - B1.y + c4.w * a - c4.z * b);
      }
      { f32x4 z; z[0] = o.x; z[1] = o.y; z[2] = o.z; z[3] = o.w; __builtin_nontemporal_store(z, reinterpret_cast<f32x4 *>(po + e2)); }
    }
    return;
  }
  {
    const uint32_t g = blockIdx.x + blockIdx.y * gridDim.x;
    if (g >= ngroups) return;
    const uint32_t t0 = g * PL::TPB;
    if (!INV) {
      auto ld = [&](int t, int e) -> float2 {
        const uint32_t tr = t0 + t;
        return tr < batch ? in[(size_t)tr * M + e] : make_float2(0.f, 0.f);
      };
      auto st = [&](int t, int e, float2 v) { lds[PL::idx(t, e)] = v; };
      // the split's (sin, cos) factors depend only on the thread: fetch them now, their latency
      // hides behind the whole transform
      constexpr int NSPLIT = PL::TPB * M / PL::T;  // = 16
      float2 csr[NSPLIT];
#pragma unroll
      for (int i = 0; i < NSPLIT; i++) csr[i] = twr[(tid + i * PL::T) & (M - 1)];
      fft_block<PL, false, true>(lds, tw, tid, ld, st);
      __syncthreads();
      // split: bin k from Z[k] and Z[M-k]
#pragma unroll
      for (int i = 0; i < NSPLIT; i++) {
        const int w = tid + i * PL::T;
        const int t = w >> LOGM, k = w & (M - 1);
        const uint32_t tr = t0 + t;
        const float2 A = lds[PL::idx(t, k)];
        float2 o;
        if (k == 0) {
          o = make_float2(A.x + A.y, A.x - A.y);  // 0.5*((2 re) +- (2 im)), exact
        } else {
          const float2 B = lds[PL::idx(t, M - k)];
          const float2 c = csr[i];  // (sin, cos)
          const float a = B.x - A.x, b = B.y + A.y;
          o.x = 0.5f * (A.x + B.x + c.x * a + c.y * b);
          o.y = 0.5f * (A.y - B.y + c.y * a - c.x * b);
        }
        if (tr < batch) out[(size_t)tr * M + k] = o;
      }
    } else {
      const float inv_m = 1.0f / (float)M;
      auto ld = [&](int t, int e) -> float2 {
        const uint32_t tr = t0 + t;
        if (tr >= batch) return make_float2(0.f, 0.f);
        const float2 A = in[(size_t)tr * M + e];
        float2 o;
        if (e == 0) {
          o = make_float2(0.5f * (A.x + A.y), 0.5f * (A.x - A.y));
        } else {
          const float2 B = in[(size_t)tr * M + (M - e)];
          const float2 c = twr[e];
          const float a = A.x - B.x, b = A.y + B.y;
          o.x = 0.5f * (A.x + B.x - c.x * a - c.y * b);
          o.y = 0.5f * (A.y - B.y + c.y * a - c.x * b);
        }
        o.y = -o.y;  // conjugate in (inverse through the forward kernel)
        return o;
      };
      auto st = [&](int t, int e, float2 v) {
        const uint32_t tr = t0 + t;
        if (tr < batch) out[(size_t)tr * M + e] = make_float2(v.x * inv_m, -v.y * inv_m);
      };
      fft_block<PL, false, false>(lds, tw, tid, ld, st);
    }
  }
}

// ---- on-device twiddle generation (double precision, exact octant reduction by sincospi) ----
__global__ void gen_twiddle_f32_kernel(float2 *tw, uint32_t n)
{
  uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  double s, c;
  sincospi(2.0 * (double)m / (double)n, &s, &c);
  tw[m] = make_float2((float)c, (float)(-s));
}

__global__ void gen_twiddle_rfft_kernel(float2 *tw, uint32_t n_real)
{
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_real / 2) return;
  double s, c;
  sincospi(2.0 * (double)k / (double)n_real, &s, &c);
  tw[k] = make_float2((float)s, (float)c);
}

hipError_t launch_gen_twiddle_f32(float2 *tw, uint32_t n, hipStream_t stream)
{
  gen_twiddle_f32_kernel<<<(n + 255) / 256, 256, 0, stream>>>(tw, n);
  return hipGetLastError();
}

hipError_t launch_gen_twiddle_rfft(float2 *tw, uint32_t n_real, hipStream_t stream)
{
  gen_twiddle_rfft_kernel<<<(n_real / 2 + 255) / 256, 256, 0, stream>>>(tw, n_real);
  return hipGetLastError();
}

// ---- launchers ----
static dim3 grid_for(uint32_t ngroups)
{
  // one workgroup per group of TPB transforms; fold very large batches into grid.y
  const uint32_t gx = ngroups < 65536u ? ngroups : 65536u;
  return dim3(gx, (ngroups + gx - 1) / gx, 1);
}

template <int LOGN>
static constexpr size_t cfft_lds_bytes()
{
  if constexpr (Tune<LOGN>::SPLIT) return SplitImage<CPlan<LOGN>>::LDS_BYTES;
  else return CPlan<LOGN>::LDS_BYTES;
}

template <int LOGN, bool INV>
static hipError_t launch_cfft_one(float2 *data, const float2 *tw, uint32_t batch, int order, hipStream_t stream)
{
  using PL = CPlan<LOGN>;
  static std::atomic<bool> attr_done[16];  // zero-initialised; a racing second hipFuncSetAttribute is harmless
  int dev = 0;
  (void)hipGetDevice(&dev);
  if constexpr (Tune<LOGN>::SPLIT) {
    static std::atomic<bool> split_attr_done[16];
    const size_t lds = SplitImage<PL>::LDS_BYTES;
    if (lds > 48 * 1024 && (dev < 0 || dev >= 16 || !split_attr_done[dev])) {   // beyond the cache's 16 slots: set it every time
      hipError_t e = hipFuncSetAttribute((const void *)cfft_f32_split_kernel<LOGN, INV, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void *)cfft_f32_split_kernel<LOGN, INV, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void *)cfft_f32_split_kernel<LOGN, INV, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) split_attr_done[dev] = true;
    }
    const dim3 grid = grid_for(batch);
    note_kernel(INV ? "cfft_f32_split_kernel<%d, true, %d>" : "cfft_f32_split_kernel<%d, false, %d>", false, LOGN, order);
    if (order == 0) cfft_f32_split_kernel<LOGN, INV, 0><<<grid, PL::T, lds, stream>>>(data, tw, batch);
    else if (order == 1) cfft_f32_split_kernel<LOGN, INV, 1><<<grid, PL::T, lds, stream>>>(data, tw, batch);
    else cfft_f32_split_kernel<LOGN, INV, 2><<<grid, PL::T, lds, stream>>>(data, tw, batch);
    return hipGetLastError();
  }
  auto kern = cfft_f32_kernel<LOGN, INV>;
  const size_t lds_bytes = cfft_lds_bytes<LOGN>();
  if (lds_bytes > 48 * 1024 && (dev < 0 || dev >= 16 || !attr_done[dev])) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  const uint32_t ngroups = (batch + PL::TPB - 1) / PL::TPB;
  dim3 grid = grid_for(ngroups);
  note_kernel(INV ? "cfft_f32_kernel<%d, true>" : "cfft_f32_kernel<%d, false>", false, LOGN);
  kern<<<grid, PL::T, lds_bytes, stream>>>(data, tw, batch, ngroups, order);
  return hipGetLastError();
}

static hipError_t launch_cfft_lane32(float2 *data, uint32_t batch, bool inverse, hipStream_t stream);   // below

hipError_t launch_cfft_f32(int logn, float2 *data, const float2 *tw, uint32_t batch, bool inverse, int order,
                           hipStream_t stream)
{
  if (batch == 0) return hipSuccess;
  if (logn == 5 && order == 0) return launch_cfft_lane32(data, batch, inverse, stream);
  switch (logn) {
#define C(L) \
  case L: return inverse ? launch_cfft_one<L, true>(data, tw, batch, order, stream) : launch_cfft_one<L, false>(data, tw, batch, order, stream);
    C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14)
#undef C
  }
  return hipErrorInvalidValue;
}

// 32 complex points, natural order, lane form: one lane per transform.  n = 2a + b: two 16-point butterflies, the odd half times
// W_32^q (constants in scalar register pairs), one radix-2 layer -- no exchange, no twiddle table; LDS only stages the 256-byte
// rows of a wave's 64 transforms (see rfft_f32_lane32_kernel below).  The two-pass LDS form ran at 0.72 of the roofline.
__device__ __forceinline__ float2 cmul_k(float2 a, float cr, float ci)
{
  const v2f w = {cr, ci};
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(tov(a)), "s"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(tov(a)), "s"(w), "v"(t));
  return tof(r);
}
__device__ constexpr float kCos32[9] = {1.0f, 0.98078528040323043058f, 0.92387953251128673848f, 0.83146961230254523567f, 0.70710678118654757274f,
                                        0.55557023301960228867f, 0.38268343236508983729f, 0.19509032201612833135f, 0.0f};
constexpr float cos32(int m) { return m <= 8 ? kCos32[m] : -kCos32[16 - m]; }   // m = 0 .. 16
constexpr float sin32(int m) { return m <= 8 ? kCos32[8 - m] : kCos32[m - 8]; }

template <bool INV>
__global__ __launch_bounds__(64) void cfft_f32_lane32_kernel(float2 *data, uint32_t batch)
{
  constexpr int N = 32, HALFB = N * 4 + 16, CPH = 8;   // a row is staged in two halves of 128 bytes (8 chunks of 16 bytes): whole lines, half the LDS
  __shared__ __attribute__((aligned(16))) unsigned char stage[64 * HALFB];
  const int lane = threadIdx.x;
  const uint32_t t0 = (blockIdx.x + blockIdx.y * gridDim.x) * 64u;
  if (t0 >= batch) return;
  const int nvalid = batch - t0 < 64u ? (int)(batch - t0) : 64;
  f32x4 *p = reinterpret_cast<f32x4 *>(data + (size_t)t0 * N);
  float2 e[16], o[16];
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int k = 0; k < CPH; k++) {
      const int v = lane + 64 * k, row = v / CPH, col = v % CPH;
      f32x4 raw = {0.f, 0.f, 0.f, 0.f};
      if (row < nvalid) raw = __builtin_nontemporal_load(p + row * 2 * CPH + h * CPH + col);
      *reinterpret_cast<f32x4 *>(stage + row * HALFB + col * 16) = raw;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CPH; k++) {
      const f32x4 r = *reinterpret_cast<const f32x4 *>(stage + lane * HALFB + k * 16);
      e[h * CPH + k] = make_float2(r[0], INV ? -r[1] : r[1]);
      o[h * CPH + k] = make_float2(r[2], INV ? -r[3] : r[3]);
    }
    __syncthreads();
  }
  butterfly<16, ArPk>(e);
  butterfly<16, ArPk>(o);
  float2 x[N];
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const float2 w = q == 0 ? o[0] : (q == 8 ? ArPk::mul_mi(o[8]) : cmul_k(o[q], cos32(q), -sin32(q)));
    x[q] = ArPk::cadd(e[q], w);
    x[q + 16] = ArPk::csub(e[q], w);
  }
  constexpr float inv_n = 1.0f / (float)N;
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int k = 0; k < CPH; k++) {
      const int i = 2 * (h * CPH + k);
      f32x4 w;
      w[0] = x[i].x; w[1] = x[i].y; w[2] = x[i + 1].x; w[3] = x[i + 1].y;
      if (INV) { w[0] *= inv_n; w[1] *= -inv_n; w[2] *= inv_n; w[3] *= -inv_n; }
      *reinterpret_cast<f32x4 *>(stage + lane * HALFB + k * 16) = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CPH; k++) {
      const int v = lane + 64 * k, row = v / CPH, col = v % CPH;
      if (row < nvalid) __builtin_nontemporal_store(*reinterpret_cast<const f32x4 *>(stage + row * HALFB + col * 16), p + row * 2 * CPH + h * CPH + col);
    }
    __syncthreads();
  }
}

static hipError_t launch_cfft_lane32(float2 *data, uint32_t batch, bool inverse, hipStream_t stream)
{
  const dim3 grid = grid_for((batch + 63u) / 64u);
  note_kernel(inverse ? "cfft_f32_lane32_kernel<true>" : "cfft_f32_lane32_kernel<false>", false);
  if (inverse) cfft_f32_lane32_kernel<true><<<grid, 64, 0, stream>>>(data, batch);
  else cfft_f32_lane32_kernel<false><<<grid, 64, 0, stream>>>(data, batch);
  return hipGetLastError();
}

// 32 real points, forward, lane form: ONE LANE PER TRANSFORM (the fixed-point real FFT's lane kernel, armfft_fixed.hip).  The 16
// complex points are one radix-16 butterfly on the lane's registers, both operands of every split bin are already there, the split
// factors have the same index in every lane (scalar loads), and LDS only turns the 128-byte rows of a wave's 64 transforms into
// lane-contiguous 16-byte accesses (rows padded by 16 bytes: conflict-free 16-byte row accesses).
__global__ __launch_bounds__(64) void rfft_f32_lane32_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ twr,
                                                             uint32_t batch)
{
  constexpr int M = 16, ROWB = M * 8 + 16, CPR = 8;   // 16-byte chunks per row
  __shared__ __attribute__((aligned(16))) unsigned char stage[64 * ROWB];
  const int lane = threadIdx.x;
  const uint32_t t0 = (blockIdx.x + blockIdx.y * gridDim.x) * 64u;
  if (t0 >= batch) return;
  const int nvalid = batch - t0 < 64u ? (int)(batch - t0) : 64;
  const f32x4 *__restrict__ pi = reinterpret_cast<const f32x4 *>(in + (size_t)t0 * M);
  f32x4 *__restrict__ po = reinterpret_cast<f32x4 *>(out + (size_t)t0 * M);
#pragma unroll
  for (int k = 0; k < CPR; k++) {
    const int v = lane + 64 * k, row = v / CPR, col = v % CPR;
    f32x4 raw = {0.f, 0.f, 0.f, 0.f};
    if (row < nvalid) raw = __builtin_nontemporal_load(pi + v);
    *reinterpret_cast<f32x4 *>(stage + row * ROWB + col * 16) = raw;
  }
  __syncthreads();
  float2 z[M];
#pragma unroll
  for (int k = 0; k < CPR; k++) {
    const f32x4 r = *reinterpret_cast<const f32x4 *>(stage + lane * ROWB + k * 16);
    z[2 * k] = make_float2(r[0], r[1]);
    z[2 * k + 1] = make_float2(r[2], r[3]);
  }
  __syncthreads();
  butterfly<16, ArPk>(z);
  // split of stage_rfft_f32 (arm_rfft_fast_f32.c:316-402), as in rfft_f32_kernel
  float2 o[M];
  o[0] = make_float2(z[0].x + z[0].y, z[0].x - z[0].y);
#pragma unroll
  for (int k = 1; k < M; k++) {
    const float2 A = z[k], B = z[M - k], c = twr[k];   // (sin, cos)
    const float a = B.x - A.x, b = B.y + A.y;
    o[k].x = 0.5f * (A.x + B.x + c.x * a + c.y * b);
    o[k].y = 0.5f * (A.y - B.y + c.y * a - c.x * b);
  }
#pragma unroll
  for (int k = 0; k < CPR; k++) {
    f32x4 w;
    w[0] = o[2 * k].x; w[1] = o[2 * k].y; w[2] = o[2 * k + 1].x; w[3] = o[2 * k + 1].y;
    *reinterpret_cast<f32x4 *>(stage + lane * ROWB + k * 16) = w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CPR; k++) {
    const int v = lane + 64 * k, row = v / CPR, col = v % CPR;
    if (row < nvalid) __builtin_nontemporal_store(*reinterpret_cast<const f32x4 *>(stage + row * ROWB + col * 16), po + v);
  }
}

template <int LOGM, bool INV>
static hipError_t launch_rfft_one(const float2 *in, float2 *out, const float2 *tw, const float2 *twr, uint32_t batch,
                                  hipStream_t stream)
{
  using PL = RPlan<LOGM>;
  static std::atomic<bool> attr_done[16];  // zero-initialised; a racing second hipFuncSetAttribute is harmless
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto kern = rfft_f32_kernel<LOGM, INV>;
  if (PL::LDS_BYTES > 48 * 1024 && (dev < 0 || dev >= 16 || !attr_done[dev])) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL::LDS_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  const uint32_t ngroups = (batch + PL::TPB - 1) / PL::TPB;
  dim3 grid = grid_for(ngroups);
  note_kernel(INV ? "rfft_f32_kernel<%d, true>" : "rfft_f32_kernel<%d, false>", false, LOGM);
  kern<<<grid, PL::T, PL::LDS_BYTES, stream>>>(in, out, tw, twr, batch, ngroups);
  return hipGetLastError();
}

hipError_t launch_rfft_f32(int logn_real, const float *in, float *out, const float2 *tw_c, const float2 *tw_r,
                           uint32_t batch, bool inverse, hipStream_t stream)
{
  if (batch == 0) return hipSuccess;
  const float2 *i2 = reinterpret_cast<const float2 *>(in);
  float2 *o2 = reinterpret_cast<float2 *>(out);
  if (logn_real == 5 && !inverse) {
    const uint32_t groups = (batch + 63u) / 64u;
    note_kernel("rfft_f32_lane32_kernel", false);
    rfft_f32_lane32_kernel<<<grid_for(groups), 64, 0, stream>>>(i2, o2, tw_r, batch);
    return hipGetLastError();
  }
  switch (logn_real - 1) {
#define C(L) \
  case L: return inverse ? launch_rfft_one<L, true>(i2, o2, tw_c, tw_r, batch, stream) : launch_rfft_one<L, false>(i2, o2, tw_c, tw_r, batch, stream);
    C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13)
#undef C
  }
  return hipErrorInvalidValue;
}

}  // namespace armfft
