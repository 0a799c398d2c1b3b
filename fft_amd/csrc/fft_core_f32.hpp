// fft_core_f32.hpp -- LDS-resident Stockham FFT core for gfx950 (device code, header-only).
//
// One workgroup of T threads transforms TPB complex sequences of N = 2^LOGN points at once, every
// thread owning E = 16 points per pass (T * 16 == TPB * N).  The transform is split into
// ceil(LOGN/4) passes of radix 2/4/8/16; between passes the data crosses the workgroup through
// LDS (padded against bank conflicts), the first pass reads its inputs through a caller-supplied
// load functor and the last pass hands its outputs to a store functor, so callers fuse whatever
// surrounds the FFT (conjugation/scaling, real-FFT split, Doppler wipe-off, spectrum multiply,
// peak search) without another trip through HBM.  Output is in natural order: the reordering that
// the reference does with arm_bitreversal_32 (arm_bitreversal2.c:84-108) is folded into the
// Stockham indexing.  Forward kernel only (e^{-j...}); inverse = conj-in / conj-out like
// arm_cfft_f32 (arm_cfft_f32.c:1252-1261, :1285-1297), done in the functors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace armfft {

// ---- complex arithmetic ----
// Two spellings of the same operations, chosen per plan (Plan::AR):
//   ArPk: packed fp32 (v_pk_*_f32: one instruction works on the (re, im) register pair).  The compiler forms packed
//         adds by itself but not the half-swapped, half-negated operand forms that complex multiplication and rotation
//         by -i need (it falls back to scalar instructions plus register moves), so those are spelled out: op_sel /
//         op_sel_hi pick which half of each source feeds the low / high lane, neg_lo / neg_hi negate there.  Half the
//         VALU instructions of the scalar form and ~20 fewer registers: the compute-bound GNSS kernel, the real FFTs
//         and most complex sizes gain (same-box A/B: GNSS 3.84 -> 3.50 ms, rfft 16384 0.62 -> 0.65 of the roofline,
//         cfft 1024 0.74 -> 0.77, 16384 0.68 -> 0.69);
//   ArSc: plain C++ on the components, scheduled freely by the compiler: the 4096- and 8192-point one-shot kernels
//         stream 5-7 % faster with it (0.749 against 0.681 of the roofline at 4096, whatever the workgroups per CU).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f tov(float2 a) { return __builtin_bit_cast(v2f, a); }
__device__ __forceinline__ float2 tof(v2f a) { return __builtin_bit_cast(float2, a); }

struct ArPk {
  static __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return tof(tov(a) + tov(b)); }
  static __device__ __forceinline__ float2 csub(float2 a, float2 b) { return tof(tov(a) - tov(b)); }
  // a + (-i) b = (a.x + b.y, a.y - b.x)
  static __device__ __forceinline__ float2 cadd_mi(float2 a, float2 b)
  {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(tov(a)), "v"(tov(b)));
    return tof(r);
  }
  // a - (-i) b = (a.x - b.y, a.y + b.x)
  static __device__ __forceinline__ float2 csub_mi(float2 a, float2 b)
  {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(tov(a)), "v"(tov(b)));
    return tof(r);
  }
  static __device__ __forceinline__ float2 cmul(float2 a, float2 w)
  {
    // t = (a.y w.y, a.y w.x);  r = (a.x w.x - t.x, a.x w.y + t.y)
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(tov(a)), "v"(tov(w)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(tov(a)), "v"(tov(w)), "v"(t));
    return tof(r);
  }
  static __device__ __forceinline__ float2 cmul_conj(float2 a, float2 w)  // a * conj(w) = (a.x w.x + a.y w.y, a.y w.x - a.x w.y)
  {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(tov(a)), "v"(tov(w)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(tov(a)), "v"(tov(w)), "v"(t));
    return tof(r);
  }
  static __device__ __forceinline__ float2 mul_mi(float2 a)   // * (-i) = (a.y, -a.x)
  {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(tov(a)));
    return tof(r);
  }
  // e + c * s for a real constant c
  static __device__ __forceinline__ float2 cfma_r(float2 s, float c, float2 e)
  {
    const v2f cc = {c, c};
    return tof(__builtin_elementwise_fma(tov(s), cc, tov(e)));
  }
  static __device__ __forceinline__ float2 scale(float2 a, float c)
  {
    const v2f cc = {c, c};
    return tof(tov(a) * cc);
  }
};

struct ArSc {
  static __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
  static __device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
  static __device__ __forceinline__ float2 cadd_mi(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }
  static __device__ __forceinline__ float2 csub_mi(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }
  static __device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
  static __device__ __forceinline__ float2 cmul_conj(float2 a, float2 w) { return make_float2(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y); }
  static __device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
  static __device__ __forceinline__ float2 cfma_r(float2 s, float c, float2 e) { return make_float2(e.x + c * s.x, e.y + c * s.y); }
  static __device__ __forceinline__ float2 scale(float2 a, float c) { return make_float2(a.x * c, a.y * c); }
};

// the kernels' own complex arithmetic (spectrum products, split / merge stages, big twiddles): packed
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return ArPk::cadd(a, b); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return ArPk::csub(a, b); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return ArPk::cmul(a, w); }
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 w) { return ArPk::cmul_conj(a, w); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return ArPk::mul_mi(a); }

// ---- in-register butterflies: forward DFT, natural order in and out ----
template <class AR>
__device__ __forceinline__ void fft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3)
{
  const float2 t0 = AR::cadd(a0, a2), t1 = AR::csub(a0, a2), t2 = AR::cadd(a1, a3), d = AR::csub(a1, a3);
  a0 = AR::cadd(t0, t2);
  a1 = AR::cadd_mi(t1, d);
  a2 = AR::csub(t0, t2);
  a3 = AR::csub_mi(t1, d);
}

template <int R, class AR = ArPk>
__device__ __forceinline__ void butterfly(float2 *v)
{
  static_assert(R == 2 || R == 4 || R == 8 || R == 16, "radix");
  if constexpr (R == 2) {
    const float2 t = v[0];
    v[0] = AR::cadd(t, v[1]);
    v[1] = AR::csub(t, v[1]);
  } else if constexpr (R == 4) {
    fft4<AR>(v[0], v[1], v[2], v[3]);
  } else if constexpr (R == 8) {
    const float c = 0.70710678118654752440f;
    fft4<AR>(v[0], v[2], v[4], v[6]);  // E0..E3
    fft4<AR>(v[1], v[3], v[5], v[7]);  // O0..O3
    // O1 W8^1 = c (x + y, y - x) = c * s1;  O2 W8^2 = -i O2;  O3 W8^3 = -c (x - y, x + y) = -c * s3
    const float2 s1 = AR::cadd_mi(v[3], v[3]), s3 = AR::csub_mi(v[7], v[7]);
    const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o2 = v[5];
    v[0] = AR::cadd(e0, o0);
    v[4] = AR::csub(e0, o0);
    v[1] = AR::cfma_r(s1, c, e1);
    v[5] = AR::cfma_r(s1, -c, e1);
    v[2] = AR::cadd_mi(e2, o2);
    v[6] = AR::csub_mi(e2, o2);
    v[3] = AR::cfma_r(s3, -c, e3);
    v[7] = AR::cfma_r(s3, c, e3);
  } else {
    // 16 = 4 x 4: X[q1 + 4 q2] = sum_b W16^(b q1) W4^(b q2) sum_a v[4a + b] W4^(a q1)
    const float c8 = 0.70710678118654752440f;
    const float c16 = 0.92387953251128675613f, s16 = 0.38268343236508977173f;
#pragma unroll
    for (int b = 0; b < 4; b++) fft4<AR>(v[b], v[4 + b], v[8 + b], v[12 + b]);  // v[4 q1 + b] = u_b[q1]
    // twiddle u_b[q1] by W16^(b q1)
    v[4 * 1 + 1] = AR::cmul(v[4 * 1 + 1], make_float2(c16, -s16));               // W16^1
    v[4 * 1 + 2] = AR::scale(AR::cadd_mi(v[4 * 1 + 2], v[4 * 1 + 2]), c8);       // W16^2 = c8 (x + y, y - x)
    v[4 * 1 + 3] = AR::cmul(v[4 * 1 + 3], make_float2(s16, -c16));               // W16^3
    v[4 * 2 + 1] = AR::scale(AR::cadd_mi(v[4 * 2 + 1], v[4 * 2 + 1]), c8);       // W16^2
    v[4 * 2 + 2] = AR::mul_mi(v[4 * 2 + 2]);                                     // W16^4
    v[4 * 2 + 3] = AR::scale(AR::csub_mi(v[4 * 2 + 3], v[4 * 2 + 3]), -c8);      // W16^6 = -c8 (x - y, x + y)
    v[4 * 3 + 1] = AR::cmul(v[4 * 3 + 1], make_float2(s16, -c16));               // W16^3
    v[4 * 3 + 2] = AR::scale(AR::csub_mi(v[4 * 3 + 2], v[4 * 3 + 2]), -c8);      // W16^6
    v[4 * 3 + 3] = AR::cmul(v[4 * 3 + 3], make_float2(-c16, s16));               // W16^9
    // second layer over b; result X[q1 + 4 q2] lands in v[4 q1 + q2] -> transpose to natural order
#pragma unroll
    for (int q1 = 0; q1 < 4; q1++) fft4<AR>(v[4 * q1 + 0], v[4 * q1 + 1], v[4 * q1 + 2], v[4 * q1 + 3]);
    float2 t;
#define ARMFFT_SWAP(i, j) t = v[i]; v[i] = v[j]; v[j] = t;
    ARMFFT_SWAP(1, 4) ARMFFT_SWAP(2, 8) ARMFFT_SWAP(3, 12) ARMFFT_SWAP(6, 9) ARMFFT_SWAP(7, 13) ARMFFT_SWAP(11, 14)
#undef ARMFFT_SWAP
  }
}

// ---- compile-time plan ----
// Plan<LOGN, TT, PP, PK, EE>: TT threads per workgroup (0 = default: 256 below 4096 points, N/16 above).
// PP: two LDS images used alternately (pass p writes image p & 1), which removes the barrier between
// a middle pass's reads and its writes -- for kernels that run few workgroups per CU and have the
// LDS to spare.
template <int LOGN_, int TT = 0, bool PP = false, bool PK = true, int EE = 16>
struct Plan {
  using AR = typename std::conditional<PK, ArPk, ArSc>::type;  // complex arithmetic of the passes (see ArPk / ArSc)
  static constexpr bool PINGPONG = PP;
  static constexpr int E = EE;  // points per thread and pass (16; 32 in the four-step's double tiles, split exchange only)
  static constexpr int LOGN = LOGN_;
  static constexpr int N = 1 << LOGN;
  static constexpr int NPASS = (LOGN + 3) / 4;
  static constexpr int BASE = LOGN / NPASS;
  static constexpr int EXTRA = LOGN % NPASS;
  // log2 radix of pass p: the first EXTRA passes are one bit wider
  static constexpr int lr(int p) { return BASE + (p < EXTRA ? 1 : 0); }
  // log2 of Ns (product of the radices of the passes before p)
  static constexpr int lns(int p) { return p == 0 ? 0 : lns(p - 1) + lr(p - 1); }
  // threads per workgroup and sequences per workgroup at 16 points per thread
  static constexpr int T = TT ? TT : ((N >= 4096) ? N / 16 : 256);
  static constexpr int TPB = (T * E) / N;
  static_assert(TPB >= 1 && TPB * N == T * E, "E points per thread");
  // LDS image: each sequence padded by 2 points per 16 (first-pass 16-byte stores and later
  // 8-byte reads conflict-free or 2-way) and skewed by 2 more points per sequence so that
  // accesses that walk ACROSS sequences (tile staging, transposes) spread over the banks.
  static constexpr int SEQ_STRIDE = N + (N >> 4) * 2 + (TPB > 1 ? 2 : 0);
  static constexpr int BUF_ELEMS = TPB * SEQ_STRIDE;
  static constexpr int LDS_ELEMS = (PP ? 2 : 1) * BUF_ELEMS;
  static constexpr size_t LDS_BYTES = (size_t)LDS_ELEMS * sizeof(float2);
  // LDS index of point e of sequence t
  static __device__ __forceinline__ int idx(int t, int e) { return t * SEQ_STRIDE + e + ((e >> 4) << 1); }
  // idx(t, e + c) - idx(t, e) for a compile-time c whose set bits are disjoint from e's (true for
  // the r << LNR / r << LNS strides of a pass): a constant, so it folds into the DS offset field
  static constexpr int delta(int c) { return c + ((c >> 4) << 1); }
};

// Load tag: the caller has already placed the first pass's inputs in v[] (same order the gather
// would have produced: butterfly q = 0..16/R-1 of thread tid is w = tid + q*T, leg r at v[q*R + r]
// holds point (w mod N/R) + r*N/R of sequence w / (N/R)).  Used by software-pipelined kernels
// that prefetch the next transform while the current one is in flight.
struct Preloaded {
  __device__ __forceinline__ float2 operator()(int, int) const { return make_float2(0.f, 0.f); }
};

// Calls f(slot, t, e) for the 16 first-pass inputs of thread tid: v[slot] = point e of sequence t.
template <class PL, class F>
__device__ __forceinline__ void for_each_first_pass_input(int tid, F f)
{
  constexpr int LR = PL::lr(0), R = 1 << LR, NB = PL::E / R, LNR = PL::LOGN - LR;
#pragma unroll
  for (int q = 0; q < NB; q++) {
    const int w = tid + q * PL::T;
    const int t = w >> LNR, j = w & ((1 << LNR) - 1);
#pragma unroll
    for (int r = 0; r < R; r++) f(q * R + r, t, j + (r << LNR));
  }
}

// Same, with point e given as j + roff: roff = r * N/R is a compile-time constant per leg, so a
// caller with a workgroup-uniform base pointer folds it into the (scalar) base and keeps a single
// per-lane offset register for all 16 loads.
template <class PL, class F>
__device__ __forceinline__ void for_each_first_pass_input_split(int tid, F f)
{
  constexpr int LR = PL::lr(0), R = 1 << LR, NB = PL::E / R, LNR = PL::LOGN - LR;
#pragma unroll
  for (int q = 0; q < NB; q++) {
    const int w = tid + q * PL::T;
    const int t = w >> LNR, j = w & ((1 << LNR) - 1);
#pragma unroll
    for (int r = 0; r < R; r++) f(q * R + r, t, j, r << LNR);
  }
}

// Calls f(slot, t, e0, roff) for the 16 last-pass outputs of thread tid: after the transform,
// v[slot] holds output bin e0 + roff of sequence t (roff = r * Ns: compile-time constant per leg).
template <class PL, class F>
__device__ __forceinline__ void for_each_last_pass_output(int tid, F f)
{
  constexpr int P = PL::NPASS - 1, LR = PL::lr(P), R = 1 << LR, NB = PL::E / R, LNS = PL::lns(P), NS = 1 << LNS, LNR = PL::LOGN - LR;
#pragma unroll
  for (int q = 0; q < NB; q++) {
    const int w = tid + q * PL::T;
    const int t = w >> LNR, j = w & ((1 << LNR) - 1);
    const int j0 = ((j >> LNS) << (LNS + LR)) + (j & (NS - 1));
#pragma unroll
    for (int r = 0; r < R; r++) f(q * R + r, t, j0, r << LNS);
  }
}

// ---- twiddle sources ----
// The pass code asks for twiddle `slot` (= q*(R-1) + r-1) of pass P, whose table index is idx.
// TwTable reads the device-generated table (L1/L2-resident, never HBM in steady state).
struct TwTable {
  static constexpr bool lazy(int) { return false; }
  const float2 *__restrict__ tw;
  // same set-up interface as TwPow, so that a kernel can pick its twiddle source per size
  __device__ __forceinline__ void preload(const float2 *__restrict__ table, int) { tw = table; }
  __device__ __forceinline__ void pin() {}
  template <int P>
  __device__ __forceinline__ float2 get(int /*slot*/, int idx) const { return tw[idx]; }
};

// TwRegs keeps the twiddles of the passes in HMASK (bit p = pass p) in registers: they depend only
// on the thread index, so a persistent workgroup loads them once and reuses them for every
// transform it processes.  The other passes read the table.
template <class PL, unsigned HMASK>
struct TwRegs {
  static constexpr bool lazy(int) { return false; }
  const float2 *__restrict__ tw;
  float2 w[PL::NPASS][PL::E - 1];

  template <int P>
  __device__ __forceinline__ void preload_pass(int tid)
  {
    if constexpr (P < PL::NPASS) {
      if constexpr (P > 0 && ((HMASK >> P) & 1u)) {
        constexpr int LOGN = PL::LOGN, LR = PL::lr(P), R = 1 << LR, NB = PL::E / R, LNS = PL::lns(P), NS = 1 << LNS, LNR = LOGN - LR;
#pragma unroll
        for (int q = 0; q < NB; q++) {
          const int j = (tid + q * PL::T) & ((1 << LNR) - 1);
          const int stepk = (j & (NS - 1)) << (LOGN - LNS - LR);
#pragma unroll
          for (int r = 1; r < R; r++) w[P][q * (R - 1) + r - 1] = tw[r * stepk];
        }
      }
      preload_pass<P + 1>(tid);
    }
  }
  __device__ __forceinline__ void preload(int tid) { preload_pass<0>(tid); }

  template <int P>
  __device__ __forceinline__ float2 get(int slot, int idx) const
  {
    if constexpr ((HMASK >> P) & 1u)
      return w[P][slot];
    else
      return tw[idx];
  }
};

// v[r] *= w^r, r < 2^LR, from the LR base powers bq[i] = w^(2^i) (see TwPow)
// NESTED8 / NESTED16: radix-8 / radix-16 passes multiply the data by the bases directly (v5 = (v5 w1) w4 ...): a few more
// complex multiplies (12 for 11, 32 for 26), the same number of roundings per leg or one more, and no product temporaries
// at all -- for kernels at the edge of their register budget.
template <int LR, bool NESTED8, bool NESTED16 = false, class AR = ArPk>
__device__ __forceinline__ void twpow_mul(const float2 *bq, float2 *v)
{
  auto cmul = [](float2 a, float2 w) { return AR::cmul(a, w); };
  if constexpr ((NESTED8 && LR == 3) || (NESTED16 && LR == 4)) {
#pragma unroll
    for (int r = 1; r < (1 << LR); r++) {
      if (r & 1) v[r] = cmul(v[r], bq[0]);
      if (r & 2) v[r] = cmul(v[r], bq[1]);
      if (r & 4) v[r] = cmul(v[r], bq[2]);
      if constexpr (LR == 4) {
        if (r & 8) v[r] = cmul(v[r], bq[3]);
      }
    }
    return;
  }
  float2 w[8];  // w[r], r < 8
  w[1] = bq[0];
  v[1] = cmul(v[1], w[1]);
  if constexpr (LR >= 2) {
    w[2] = bq[1];
    w[3] = cmul(w[1], w[2]);
    v[2] = cmul(v[2], w[2]);
    v[3] = cmul(v[3], w[3]);
  }
  if constexpr (LR >= 3) {
    w[4] = bq[2];
    v[4] = cmul(v[4], w[4]);
#pragma unroll
    for (int r = 1; r < 4; r++) {
      w[4 + r] = cmul(w[4], w[r]);
      v[4 + r] = cmul(v[4 + r], w[4 + r]);
    }
  }
  if constexpr (LR >= 4) {
    const float2 w8 = bq[3];
    v[8] = cmul(v[8], w8);
#pragma unroll
    for (int r = 1; r < 8; r++) v[8 + r] = cmul(v[8 + r], cmul(w8, w[r]));
  }
}

// TwPow: the R-1 twiddles of one butterfly are the powers w, w^2 .. w^(R-1) of ONE number
// w = W_N^stepk that depends only on the thread.  A persistent workgroup keeps w^1, w^2, w^4 (, w^8)
// of every butterfly in registers (correctly rounded table values, loaded once per kernel) and
// forms the other powers as products of at most three of them right where they are used: no
// twiddle load, no twiddle address math and no outstanding vector-memory operation inside the
// transform, which is what lets a prefetch of the NEXT transform stay in flight across all passes
// (gfx9 counts loads and stores in one in-order vmcnt: waiting for a twiddle would wait for every
// load issued before it).
template <class PL, unsigned MASK = ~0u, bool NESTED8 = false>
struct TwPow {
  // NESTED8: radix-8 passes multiply the data by w^1, w^2, w^4 directly (v5 = (v5 w1) w4 ...):
  // 12 complex multiplies instead of 11 and the same number of roundings, but no product
  // temporaries at all -- for kernels at the edge of their register budget.
  // MASK: the passes served this way (bit p = pass p); the others read the table like TwTable.
  static constexpr bool lazy(int p) { return (MASK >> p) & 1u; }
  const float2 *__restrict__ tw;
  template <int P>
  __device__ __forceinline__ float2 get(int /*slot*/, int idx) const { return tw[idx]; }

  static constexpr int nbase(int lr) { return lr; }
  float2 b[PL::NPASS][PL::E / 2];  // [pass][butterfly q * nbase + i]

  template <int P>
  __device__ __forceinline__ void preload_pass(int tid)
  {
    if constexpr (P < PL::NPASS) {
      if constexpr (P > 0 && lazy(P)) {
        constexpr int LOGN = PL::LOGN, LR = PL::lr(P), R = 1 << LR, NB = PL::E / R, LNS = PL::lns(P), NS = 1 << LNS, LNR = LOGN - LR;
        constexpr int NBASE = nbase(LR);
        static_assert(NB * NBASE <= PL::E / 2, "TwPow: too many butterflies per thread for this radix");
#pragma unroll
        for (int q = 0; q < NB; q++) {
          const int j = (tid + q * PL::T) & ((1 << LNR) - 1);
          const int stepk = (j & (NS - 1)) << (LOGN - LNS - LR);
#pragma unroll
          for (int i = 0; i < NBASE; i++) b[P][q * NBASE + i] = tw[stepk << i];
        }
      }
      preload_pass<P + 1>(tid);
    }
  }
  __device__ __forceinline__ void preload(const float2 *__restrict__ table, int tid)
  {
    tw = table;
    preload_pass<0>(tid);
  }

  // Call once per transform inside a persistent loop: makes the base values opaque to the
  // optimiser, which would otherwise hoist every product out of the loop and hold all R-1
  // twiddles of all passes in registers (spilling).
  template <int P = 1>
  __device__ __forceinline__ void pin()
  {
    if constexpr (P < PL::NPASS) {
      constexpr int CNT = lazy(P) ? (PL::E >> PL::lr(P)) * nbase(PL::lr(P)) : 0;
#pragma unroll
      for (int i = 0; i < CNT; i++) asm volatile("" : "+v"(b[P][i].x), "+v"(b[P][i].y));
      pin<P + 1>();
    }
  }

  // v[r] *= w^r for the R legs of butterfly q of pass P
  template <int P>
  __device__ __forceinline__ void mul(int q, float2 *v) const
  {
    constexpr int LR = PL::lr(P);
    twpow_mul<LR, NESTED8, false, typename PL::AR>(&b[P][q * nbase(LR)], v);
  }
};

// One pass.  P = pass index.  v holds the thread's 16 points: butterfly q (0..16/R-1), leg r at
// v[q*R + r].
// twc = this pass's twiddles (fetched during the previous pass), twn = filled with the next pass's
// twiddles right before this pass's closing barrier: v[] is dead by then, so the loads cost no
// registers, and their L1/L2 latency hides behind the barrier and the next gather instead of
// stalling every pass (workgroups of 8192/16384-point transforms are too few per CU to hide it).
//
// hook() runs in pass HP between the twiddle multiplies and the butterflies.  A persistent kernel
// issues the next transform's loads there -- ahead of this one's stores in the memory queues --
// so that their latency and transfer overlap the remaining passes and the store drain.  With a
// table twiddle source HP must be the last pass (a later twiddle wait would wait for the
// prefetch too: one in-order vmcnt); with TwPow any pass works and the choice is register budget.
// A Store functor that declares `static constexpr bool SPLIT = true` receives output bin e as
// (e0, roff) with roff = r * Ns a compile-time constant per leg (st(t, e0, roff, val)): a caller
// with a workgroup-uniform base folds roff into a scalar offset (buffer stores).
template <class S, class = void>
struct store_is_split : std::false_type {};
template <class S>
struct store_is_split<S, std::void_t<decltype(S::SPLIT)>> : std::true_type {};

// ... and one that declares `static constexpr bool SLOT = true` also gets the butterfly and leg it
// is storing: st(q, r, t, e0, roff, val) -- for per-butterfly state prepared by the caller.
template <class S, class = void>
struct store_has_slot : std::false_type {};
template <class S>
struct store_has_slot<S, std::void_t<decltype(S::SLOT)>> : std::true_type {};

struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};


template <class PL, int P, bool LDS_IN, bool LDS_OUT, int HP, class TW, class Load, class Store, class Hook>
__device__ __forceinline__ void fft_pass(float2 *v, float2 *lds, const TW &tw, int tid, Load &ld, Store &st, const float2 *twc, float2 *twn,
                                         Hook &hook)
{
  constexpr int LOGN = PL::LOGN, T = PL::T;
  constexpr int LR = PL::lr(P), R = 1 << LR, NB = PL::E / R;
  constexpr int LNS = PL::lns(P), NS = 1 << LNS;
  constexpr int LNR = LOGN - LR;  // log2(N/R): butterflies per transform
  constexpr bool FIRST = (P == 0), LAST = (P == PL::NPASS - 1);
  // ping-pong plans: pass p reads the image pass p - 1 wrote and writes the other one
  const float2 *src = lds + ((PL::PINGPONG && !FIRST && ((P - 1) & 1)) ? PL::BUF_ELEMS : 0);
  float2 *dst = lds + ((PL::PINGPONG && (P & 1)) ? PL::BUF_ELEMS : 0);

  // ---- gather ----
#pragma unroll
  for (int q = 0; q < NB; q++) {
    const int w = tid + q * T;
    const int t = w >> LNR, j = w & ((1 << LNR) - 1);
    const int base = PL::idx(t, j);  // one address per butterfly; the legs are compile-time offsets
#pragma unroll
    for (int r = 0; r < R; r++) {
      if constexpr (FIRST) {
        if constexpr (!std::is_same<Load, Preloaded>::value) v[q * R + r] = ld(t, j + (r << LNR));
      } else {
        v[q * R + r] = src[base + PL::delta(r << LNR)];
      }
    }
  }
  // ---- twiddle + butterfly ----
  if constexpr (!FIRST) {
#pragma unroll
    for (int q = 0; q < NB; q++) {
      if constexpr (TW::lazy(P)) {
        tw.template mul<P>(q, v + q * R);
      } else {
#pragma unroll
        for (int r = 1; r < R; r++) v[q * R + r] = PL::AR::cmul(v[q * R + r], twc[q * (R - 1) + r - 1]);
      }
    }
  }
  if constexpr (P == HP && !std::is_same<Hook, NoHook>::value) {
    // twc is dead: the hook's registers take its place (and must not be scheduled any earlier)
    __builtin_amdgcn_sched_barrier(0);
    hook();
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int q = 0; q < NB; q++) butterfly<R, typename PL::AR>(v + q * R);
  // ---- scatter ----
  // source and destination both in LDS: every read of this pass must finish before any write
  if ((!FIRST || LDS_IN) && (!LAST || LDS_OUT) && !(PL::PINGPONG && !FIRST && !LAST)) __syncthreads();
#pragma unroll
  for (int q = 0; q < NB; q++) {
    const int w = tid + q * T;
    const int t = w >> LNR, j = w & ((1 << LNR) - 1);
    const int k = j & (NS - 1);
    const int j0 = ((j >> LNS) << (LNS + LR)) + k;
    if (LAST) {
#pragma unroll
      for (int r = 0; r < R; r++) {
        if constexpr (store_has_slot<Store>::value) st(q, r, t, j0, r << LNS, v[q * R + r]);
        else if constexpr (store_is_split<Store>::value) st(t, j0, r << LNS, v[q * R + r]);
        else st(t, j0 + (r << LNS), v[q * R + r]);
      }
    } else if (FIRST && R >= 2) {
      // NS == 1: R consecutive points per butterfly -> 16-byte LDS stores
      const int base = PL::idx(t, j * R);
#pragma unroll
      for (int r = 0; r < R; r += 2) {
        *reinterpret_cast<float4 *>(&dst[base + PL::delta(r)]) =
            make_float4(v[q * R + r].x, v[q * R + r].y, v[q * R + r + 1].x, v[q * R + r + 1].y);
      }
    } else {
      const int base = PL::idx(t, j0);
#pragma unroll
      for (int r = 0; r < R; r++) dst[base + PL::delta(r << LNS)] = v[q * R + r];
    }
  }
  if constexpr (!LAST && !TW::lazy(LAST ? P : P + 1)) {
    // next pass's twiddles: W_{NS*R}^{r k} = tw[r * k * N/(NS*R)], k = butterfly index mod NS
    constexpr int LR1 = PL::lr(P + 1), R1 = 1 << LR1, NB1 = PL::E / R1, LNS1 = PL::lns(P + 1), NS1 = 1 << LNS1, LNR1 = LOGN - LR1;
#pragma unroll
    for (int q = 0; q < NB1; q++) {
      const int j = (tid + q * T) & ((1 << LNR1) - 1);
      const int stepk = (j & (NS1 - 1)) << (LOGN - LNS1 - LR1);
#pragma unroll
      for (int r = 1; r < R1; r++) twn[q * (R1 - 1) + r - 1] = tw.template get<P + 1>(q * (R1 - 1) + r - 1, r * stepk);
    }
  }
  if constexpr (!LAST) __syncthreads();
}

template <class PL, int P, bool LDS_IN, bool LDS_OUT, int HP, class TW, class Load, class Store, class Hook>
struct PassLoop {
  static __device__ __forceinline__ void run(float2 *v, float2 *lds, const TW &tw, int tid, Load &ld, Store &st, float2 *twa, float2 *twb,
                                             Hook &hook)
  {
    fft_pass<PL, P, LDS_IN, LDS_OUT, HP>(v, lds, tw, tid, ld, st, twa, twb, hook);
    if constexpr (P + 1 < PL::NPASS) PassLoop<PL, P + 1, LDS_IN, LDS_OUT, HP, TW, Load, Store, Hook>::run(v, lds, tw, tid, ld, st, twb, twa, hook);
  }
};

// Transform the workgroup's TPB sequences.  ld(t, e) returns point e of sequence t (t < TPB),
// st(t, e, val) receives output bin e of sequence t.  `lds` must hold PL::LDS_ELEMS float2
// (16-byte aligned) and must not be in use by other threads when the call starts; on return all
// st() calls of this thread have been issued (no trailing barrier).  LDS_IN / LDS_OUT: set when
// ld() reads / st() writes the workgroup's LDS (anywhere), so that the core separates them from
// its own LDS traffic with a barrier.
template <class PL, bool LDS_IN, bool LDS_OUT, int HP = -1, class TW, class Load, class Store, class Hook = NoHook>
__device__ __forceinline__ void fft_block_tw(float2 *lds, const TW &tw, int tid, Load ld, Store st, Hook hook = Hook())
{
  float2 v[PL::E], twa[PL::E - 1], twb[PL::E - 1];
  PassLoop<PL, 0, LDS_IN, LDS_OUT, HP, TW, Load, Store, Hook>::run(v, lds, tw, tid, ld, st, twa, twb, hook);
}

template <class PL, bool LDS_IN, bool LDS_OUT, class Load, class Store>
__device__ __forceinline__ void fft_block(float2 *lds, const float2 *__restrict__ tw, int tid, Load ld, Store st)
{
  TwTable t{tw};
  fft_block_tw<PL, LDS_IN, LDS_OUT>(lds, t, tid, ld, st);
}

// Same with the first pass's inputs already in v[16] (see Preloaded); hook runs in pass HP
// (default: the last), see fft_pass.  IMAGE_BUSY: other threads may still be reading the image when the call starts
// (a previous result aliased to it): the first pass then puts a barrier between its butterflies and its first write, so
// the caller needs none of its own and the first pass's arithmetic overlaps those readers.
template <class PL, bool LDS_OUT, int HP = PL::NPASS - 1, bool IMAGE_BUSY = false, class TW, class Store, class Hook = NoHook>
__device__ __forceinline__ void fft_block_preloaded(float2 *v, float2 *lds, const TW &tw, int tid, Store st, Hook hook = Hook())
{
  Preloaded ld;
  float2 twa[PL::E - 1], twb[PL::E - 1];
  PassLoop<PL, 0, IMAGE_BUSY, LDS_OUT, HP, TW, Preloaded, Store, Hook>::run(v, lds, tw, tid, ld, st, twa, twb, hook);
}

// ---- half-footprint exchange: real parts and imaginary parts cross the workgroup one after the other ----
// For transforms that fill the LDS (8192 / 16384 points: 64 / 128 KiB as float2) the number of workgroups per CU -- and
// with it the overlap of one workgroup's HBM phases with another's passes -- is set by the LDS image.  Here the image is
// N floats (+1 per 16 against bank conflicts: 68 KiB at 16384 points): every exchange between two passes runs twice
// through the same buffer, first the real parts, then the imaginary parts,
//     write re | barrier | read re | barrier | write im | barrier | read im | (barrier before the next exchange's writes)
// which doubles the DS instruction count (4-byte instead of 8-byte accesses) and the barriers, and doubles the workgroups
// per CU.  One transform per workgroup (TPB == 1), first-pass inputs preloaded.  Twiddles: the power bases of pass P + 1
// (see TwPow) are fetched from the table right before the exchange that precedes it and are dead after its multiplies, so
// at most 8 base values are live at a time instead of all passes' (32 registers at 16384 points): with the data's 32
// registers the kernel fits the 64-VGPR budget of two 1024-thread workgroups per CU.
template <class PL>
struct SplitImage {
  static_assert(PL::TPB == 1, "split exchange: one transform per workgroup");
#ifndef ARMFFT_SPLIT_PAD
#define ARMFFT_SPLIT_PAD 5
#endif
  // One float of padding per 2^PAD.  The exchange moves 4-byte words: ds_read_b32 / ds_write_b32 are served in two groups of 32
  // lanes over 32 banks (MI355X_MICROARCH.md, LDS), so the 32 consecutive points a group reads must be 32 consecutive words --
  // with a pad every 16 (rounds 1-2) every such access had one 2-way conflict per group: 40 % of all LDS cycles of the
  // 16384-point kernel were conflicts (profiles/r03_pmc_sq_cfft_f32_16384_before.txt), exactly what the bank model gives
  // (1.67 x the conflict-free cycles; a pad every 32: 1.17 x, and the first pass's stride-16 stores still spread over all banks).
  // Same-box A/B: 16384 points 0.675 -> 0.69 of the roofline, 4096 and 8192 unchanged (0.748, 0.70): these kernels wait for HBM.
  static constexpr int PAD = ARMFFT_SPLIT_PAD;
  static constexpr int ELEMS = PL::N + (PL::N >> PAD);
  static constexpr size_t LDS_BYTES = (size_t)ELEMS * sizeof(float);
  // butterfly w = tid + q*T of a pass with 2^LNR butterflies per transform: its sequence and its index in the sequence
  static __device__ __forceinline__ int seq(int, int) { return 0; }
  static __device__ __forceinline__ int bfly(int w, int lnr) { return w & ((1 << lnr) - 1); }
  static __device__ __forceinline__ int idx(int, int e) { return e + (e >> PAD); }
  static constexpr int delta(int c) { return c + (c >> PAD); }  // c's set bits disjoint from e's (leg strides)
};

// The same exchange for TPB sequences per workgroup held INTERLEAVED: butterfly w works on sequence w % TPB, so the lanes
// of a wave walk across the sequences first -- the order in which a column tile of a row-major matrix arrives from
// global memory (TPB adjacent columns = one contiguous run per row) and leaves for it.  A tile kernel can therefore load
// the first pass's inputs and store the last pass's outputs directly, with no staging copy through LDS.  Point e of
// sequence t lives at float (e + (e >> 4)) * TPB + t: 32 lanes cover 32 / TPB consecutive or padded-apart points of TPB
// adjacent sequences, conflict-free in every pass.
template <class PL>
struct SplitImageIL {
  static constexpr int LOGC = __builtin_ctz(PL::TPB);
  static_assert((1 << LOGC) == PL::TPB, "interleaved image: power-of-two sequences per workgroup");
  static constexpr int ELEMS = (PL::N + (PL::N >> 4)) * PL::TPB;
  static constexpr size_t LDS_BYTES = (size_t)ELEMS * sizeof(float);
  static __device__ __forceinline__ int seq(int w, int) { return w & (PL::TPB - 1); }
  static __device__ __forceinline__ int bfly(int w, int) { return w >> LOGC; }
  static __device__ __forceinline__ int idx(int t, int e) { return ((e + (e >> 4)) << LOGC) + t; }
  static constexpr int delta(int c) { return (c + (c >> 4)) << LOGC; }
};

// LATE_TW: the table `tw` is cheap to read (an LDS copy): every butterfly fetches its bases right where it multiplies,
// instead of one pass ahead across the exchange -- up to 8 fewer float2 live through the imaginary-part reads.
template <class PL, class IM, int P, bool NESTED8, bool NESTED16, bool LATE_TW, class Store>
__device__ __forceinline__ void fft_split_passes(float2 *v, float *ldsf, const float2 *__restrict__ tw, const float2 *bases, int tid, Store &st)
{
  constexpr int LOGN = PL::LOGN, T = PL::T;
  constexpr int LR = PL::lr(P), R = 1 << LR, NB = PL::E / R;
  constexpr int LNS = PL::lns(P), NS = 1 << LNS;
  constexpr int LNR = LOGN - LR;
  constexpr bool FIRST = (P == 0), LAST = (P == PL::NPASS - 1);
  if constexpr (!FIRST) {
#pragma unroll
    for (int q = 0; q < NB; q++) {
      if constexpr (LATE_TW) {
        const int j = IM::bfly(tid + q * T, LNR);
        const int stepk = (j & (NS - 1)) << (LOGN - LNS - LR);
        float2 bq[LR];
#pragma unroll
        for (int i = 0; i < LR; i++) bq[i] = tw[stepk << i];
        twpow_mul<LR, NESTED8, NESTED16, typename PL::AR>(bq, v + q * R);
      } else {
        twpow_mul<LR, NESTED8, NESTED16, typename PL::AR>(bases + q * LR, v + q * R);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NB; q++) butterfly<R, typename PL::AR>(v + q * R);
  if constexpr (LAST) {
#pragma unroll
    for (int q = 0; q < NB; q++) {
      const int t = IM::seq(tid + q * T, LNR), j = IM::bfly(tid + q * T, LNR);
      const int j0 = ((j >> LNS) << (LNS + LR)) + (j & (NS - 1));
#pragma unroll
      for (int r = 0; r < R; r++) {
        if constexpr (store_is_split<Store>::value) st(t, j0, r << LNS, v[q * R + r]);
        else st(t, j0 + (r << LNS), v[q * R + r]);
      }
    }
  } else {
    constexpr int LR1 = PL::lr(P + 1), R1 = 1 << LR1, NB1 = PL::E / R1, LNR1 = LOGN - LR1;
    // scatter / gather addresses: one per butterfly, the legs are compile-time offsets
    int wbase[NB], rbase[NB1];
#pragma unroll
    for (int q = 0; q < NB; q++) {
      const int t = IM::seq(tid + q * T, LNR), j = IM::bfly(tid + q * T, LNR);
      const int j0 = FIRST ? j * R : ((j >> LNS) << (LNS + LR)) + (j & (NS - 1));
      wbase[q] = IM::idx(t, j0);
    }
#pragma unroll
    for (int q = 0; q < NB1; q++) rbase[q] = IM::idx(IM::seq(tid + q * T, LNR1), IM::bfly(tid + q * T, LNR1));
    float2 w[PL::E];
    if constexpr (!FIRST) __syncthreads();  // the previous exchange's last reads are done
#pragma unroll
    for (int q = 0; q < NB; q++)
#pragma unroll
      for (int r = 0; r < R; r++) ldsf[wbase[q] + IM::delta(FIRST ? r : (r << LNS))] = v[q * R + r].x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NB1; q++)
#pragma unroll
      for (int r = 0; r < R1; r++) w[q * R1 + r].x = ldsf[rbase[q] + IM::delta(r << LNR1)];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NB; q++)
#pragma unroll
      for (int r = 0; r < R; r++) ldsf[wbase[q] + IM::delta(FIRST ? r : (r << LNS))] = v[q * R + r].y;
    // next pass's twiddle bases w^(2^i), w = W_N^stepk (TwPow::preload_pass): fetched once the outgoing data have left the registers
    // (v is dead here: the loads cost no extra registers), in flight across the last barrier and the imaginary-part reads
    constexpr int LNS1 = PL::lns(P + 1), NS1 = 1 << LNS1;
    float2 nb[NB1 * LR1];
    if constexpr (!LATE_TW) {
#pragma unroll
      for (int q = 0; q < NB1; q++) {
        const int j = IM::bfly(tid + q * T, LNR1);
        const int stepk = (j & (NS1 - 1)) << (LOGN - LNS1 - LR1);
#pragma unroll
        for (int i = 0; i < LR1; i++) nb[q * LR1 + i] = tw[stepk << i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NB1; q++)
#pragma unroll
      for (int r = 0; r < R1; r++) w[q * R1 + r].y = ldsf[rbase[q] + IM::delta(r << LNR1)];
#pragma unroll
    for (int i = 0; i < PL::E; i++) v[i] = w[i];
    fft_split_passes<PL, IM, P + 1, NESTED8, NESTED16, LATE_TW>(v, ldsf, tw, nb, tid, st);
  }
}

// v[16] = the thread's first-pass inputs (see Preloaded); ldsf = SplitImage<PL>::ELEMS floats; tw = table of N entries
template <class PL, bool NESTED8, bool NESTED16 = false, class Store>
__device__ __forceinline__ void fft_block_split(float2 *v, float *ldsf, const float2 *__restrict__ tw, int tid, Store st)
{
  fft_split_passes<PL, SplitImage<PL>, 0, NESTED8, NESTED16, false>(v, ldsf, tw, nullptr, tid, st);
}

// f(slot, t, e0, roff) for the 16 last-pass outputs of thread tid under image policy IM, in the order fft_split_passes hands
// them to its store functor: output slot is bin e0 + roff of sequence t
template <class PL, class IM, class F>
__device__ __forceinline__ void for_each_last_pass_output_im(int tid, F f)
{
  constexpr int P = PL::NPASS - 1, LR = PL::lr(P), R = 1 << LR, NB = PL::E / R, LNS = PL::lns(P), NS = 1 << LNS, LNR = PL::LOGN - LR;
#pragma unroll
  for (int q = 0; q < NB; q++) {
    const int t = IM::seq(tid + q * PL::T, LNR), j = IM::bfly(tid + q * PL::T, LNR);
    const int j0 = ((j >> LNS) << (LNS + LR)) + (j & (NS - 1));
#pragma unroll
    for (int r = 0; r < R; r++) f(q * R + r, t, j0, r << LNS);
  }
}

// Interleaved sequences (SplitImageIL): v[q*R + r] = point bfly + r * N/R of sequence (tid + q*T) % TPB, bfly = (tid + q*T) / TPB;
// ldsf = SplitImageIL<PL>::ELEMS floats
template <class PL, bool NESTED8, bool NESTED16 = false, bool LATE_TW = false, class Store>
__device__ __forceinline__ void fft_block_split_il(float2 *v, float *ldsf, const float2 *__restrict__ tw, int tid, Store st)
{
  fft_split_passes<PL, SplitImageIL<PL>, 0, NESTED8, NESTED16, LATE_TW>(v, ldsf, tw, nullptr, tid, st);
}

}  // namespace armfft
