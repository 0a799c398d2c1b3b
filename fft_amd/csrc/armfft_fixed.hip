// armfft_fixed.hip -- batched q15 / q31 complex FFT kernels for gfx950 (MI355X), bit-exact with
// the reference's generic-C fixed-point path.
//
// arm_cfft_q15 (arm_cfft_q15.c:671-726) and arm_cfft_q31 (arm_cfft_q31.c:708-759) are radix-4
// DIF transforms whose result depends on WHERE each stage scales and truncates, so the stage
// structure is reproduced exactly: a radix-2 pre-stage for N = 2*4^k (arm_cfft_radix4by2_*,
// arm_cfft_q15.c:736-931, arm_cfft_q31.c:767-885), then first / middle / last radix-4 stages
// (arm_cfft_radix4_q15.c:572-970 and :1434-1813, arm_cfft_radix4_q31.c:153-473 and :524-834) with
// the same shifts, saturations (__SSAT, Include/dsp/none.h:78-94) and truncating or rounded
// high-word multiplies (none.h:185-206).  Only the data movement is new: the transform lives in
// LDS, two radix-4 stages are fused in registers per LDS round trip, and the bit reversal of
// arm_bitreversal_16/_32 (arm_bitreversal2.c:84-148, plain bit reversal for the fixed tables) is
// folded into the final, coalesced store.
#include <atomic>
#include <type_traits>

#include "armfft_internal.h"

namespace armfft {

// ------------------------------------------------------------------------------------------
// element types: one complex point
// ------------------------------------------------------------------------------------------
struct Q15 {
  using scalar = int16_t;
  using word = uint32_t;     // packed {re, im}
  using tword = uint32_t;    // packed {cos, sin}
  struct cpx { int16_t re, im; };
  static __device__ __forceinline__ cpx unpack(word w) { return cpx{(int16_t)(w & 0xffffu), (int16_t)(w >> 16)}; }
  static __device__ __forceinline__ word pack(cpx c) { return (uint32_t)(uint16_t)c.re | ((uint32_t)(uint16_t)c.im << 16); }
};
struct Q31 {
  using scalar = int32_t;
  using word = int2;
  using tword = int2;
  struct cpx { int32_t re, im; };
  static __device__ __forceinline__ cpx unpack(word w) { return cpx{w.x, w.y}; }
  static __device__ __forceinline__ word pack(cpx c) { return make_int2(c.re, c.im); }
};

// 16-byte global accesses of streams that are touched once: non-temporal.  Same-box A/B against plain accesses, LDS form:
// cfft q15 16 / 64 / 256 points 0.71 / 0.725 / 0.72 -> 0.805 / 0.80 / 0.76 of the roofline, q31 16 points 0.71 -> 0.78, real
// transforms +1 ... +6 %.  The lane form keeps plain accesses (32 real q15 points 0.79 against 0.73 either way).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT = true>
__device__ __forceinline__ uint4 ld16_stream(const void *p)
{
  if constexpr (NT) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
  } else {
    return *reinterpret_cast<const uint4 *>(p);
  }
}
template <bool NT = true>
__device__ __forceinline__ void st16_stream(void *p, uint4 v)
{
  if constexpr (NT) {
    u32x4_t w;
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t *>(p));
  } else {
    *reinterpret_cast<uint4 *>(p) = v;
  }
}
__device__ __forceinline__ int ssat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wshl1(int32_t a) { return (int32_t)((uint32_t)a << 1); }
__device__ __forceinline__ int32_t mulhi32(int32_t a, int32_t b) { return __mulhi(a, b); }

// ------------------------------------------------------------------------------------------
// q15 radix-4 butterflies.  STAGE: 0 = first, 1 = middle, 2 = last.  x[0..3] are the points at
// i0, i0+n2, i0+2n2, i0+3n2; w1/w2/w3 the twiddles at ic, 2ic, 3ic.  Outputs go back in place:
// x[1] receives the "c" output (twiddle 2), x[2] the "b" output (twiddle 1), x[3] "d" (twiddle 3).
// ------------------------------------------------------------------------------------------
// Packed form: one complex point = one 32-bit register {re, im}; saturating adds are
// v_pk_add_i16/v_pk_sub_i16 with clamp (= __SSAT(a +- b, 16)), halvings v_pk_ashrrev_i16, the
// x -+ j*y cross terms a packed add and a packed sub of the half-swapped operand blended per half,
// and the first product of every complex multiply one v_dot2_i32_i16.  Every intermediate has the
// value the reference's scalar code gives it (sums of two halved q15 values fit 16 bits; the dot
// product |co*a + si*b| <= sqrt(2) * 2^30 fits 32 bits).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 to_v(Q15::cpx c) { return __builtin_bit_cast(s16x2, c); }
__device__ __forceinline__ Q15::cpx to_c(s16x2 v) { return __builtin_bit_cast(Q15::cpx, v); }
__device__ __forceinline__ s16x2 sat_add(s16x2 a, s16x2 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ s16x2 sat_sub(s16x2 a, s16x2 b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ s16x2 half(s16x2 a) { return a >> (short)1; }
__device__ __forceinline__ s16x2 swap_ri(s16x2 a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ s16x2 lo_hi(s16x2 lo, s16x2 hi) { return __builtin_shufflevector(lo, hi, 0, 3); }  // {lo.x, hi.y}
// a.x b.x + a.y b.y as the three-operand v_dot2_i32_i16 with the constant 0: through the builtin the compiler picks the
// two-operand v_dot2c_i32_i16, whose accumulator is its destination, and spends a v_mov_b32 on zeroing it first -- one of
// seven instructions of every complex multiply in kernels that issue vector instructions in every cycle (q15 8192 points)
__device__ __forceinline__ int dot2_q15(s16x2 a, s16x2 b)
{
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// forward: (co*x.re + si*x.im, co*x.im - si*x.re) >> 16;  inverse: (co*x.re - si*x.im, co*x.im + si*x.re) >> 16
template <bool INV>
__device__ __forceinline__ s16x2 cmul_q15(s16x2 x, s16x2 w)
{
  const int co = w.x, si = w.y, xr = x.x, xi = x.y;
  int re, im;
  if (!INV) {
    re = dot2_q15(w, x);
    im = co * xi - si * xr;
  } else {
    re = co * xr - si * xi;
    im = dot2_q15(w, swap_ri(x));  // co*x.im + si*x.re
  }
  s16x2 r;
  r.x = (short)(re >> 16);
  r.y = (short)(im >> 16);
  return r;
}

template <int STAGE, bool INV>
__device__ __forceinline__ void bfly_q15(Q15::cpx *x, Q15::cpx w1c, Q15::cpx w2c, Q15::cpx w3c)
{
  const s16x2 w1 = to_v(w1c), w2 = to_v(w2c), w3 = to_v(w3c);
  s16x2 x0 = to_v(x[0]), x1 = to_v(x[1]), x2 = to_v(x[2]), x3 = to_v(x[3]);
  if (STAGE == 0) {  // guard bits of the first stage (arm_cfft_radix4_q15.c:626-640)
    x0 = x0 >> (short)2; x1 = x1 >> (short)2; x2 = x2 >> (short)2; x3 = x3 >> (short)2;
  }
  s16x2 R = sat_add(x0, x2), S = sat_sub(x0, x2);
  s16x2 T = sat_add(x1, x3);
  s16x2 o0 = half(R) + half(T);
  if (STAGE == 1) o0 = half(o0);
  R = (STAGE == 0) ? sat_sub(R, T) : (half(R) - half(T));
  const s16x2 o1 = (STAGE == 2) ? R : cmul_q15<INV>(R, w2);
  T = sat_sub(x1, x3);
  // forward: R' = S - j*T... in components (S.re - T.im, S.im + T.re) and S' = (S.re + T.im, S.im - T.re); inverse swaps them
  s16x2 diff, sum;
  if (STAGE == 0) {
    const s16x2 Ts = swap_ri(T);
    diff = sat_sub(S, Ts);
    sum = sat_add(S, Ts);
  } else {
    const s16x2 Sh = half(S), Th = swap_ri(half(T));
    diff = Sh - Th;
    sum = Sh + Th;
  }
  const s16x2 Rn = INV ? lo_hi(sum, diff) : lo_hi(diff, sum);
  const s16x2 Sn = INV ? lo_hi(diff, sum) : lo_hi(sum, diff);
  x[0] = to_c(o0);
  x[1] = to_c(o1);
  x[2] = to_c((STAGE == 2) ? Sn : cmul_q15<INV>(Sn, w1));
  x[3] = to_c((STAGE == 2) ? Rn : cmul_q15<INV>(Rn, w3));
}

// radix-2 pre-stage of arm_cfft_radix4by2[_inverse]_q15 (arm_cfft_q15.c:786-804, :885-903)
template <bool INV>
__device__ __forceinline__ void by2_q15(Q15::cpx &a, Q15::cpx &b, Q15::cpx w)
{
  const int16_t c = w.re, s = w.im;
  int16_t xt = (int16_t)((a.re >> 1) - (b.re >> 1));
  a.re = (int16_t)(((a.re >> 1) + (b.re >> 1)) >> 1);
  int16_t yt = (int16_t)((a.im >> 1) - (b.im >> 1));
  a.im = (int16_t)(((b.im >> 1) + (a.im >> 1)) >> 1);
  if (!INV) {
    b.re = (int16_t)(((int16_t)(((int)xt * c) >> 16)) + ((int16_t)(((int)yt * s) >> 16)));
    b.im = (int16_t)(((int16_t)(((int)yt * c) >> 16)) - ((int16_t)(((int)xt * s) >> 16)));
  } else {
    b.re = (int16_t)(((int16_t)(((int)xt * c) >> 16)) - ((int16_t)(((int)yt * s) >> 16)));
    b.im = (int16_t)(((int16_t)(((int)yt * c) >> 16)) + ((int16_t)(((int)xt * s) >> 16)));
  }
}

// ------------------------------------------------------------------------------------------
// q31 radix-4 butterflies, same conventions
// ------------------------------------------------------------------------------------------
template <int STAGE, bool INV>
__device__ __forceinline__ void bfly_q31(Q31::cpx *x, Q31::cpx w1, Q31::cpx w2, Q31::cpx w3)
{
  if (STAGE == 2) {
    // last stage: plain wrapping adds (arm_cfft_radix4_q31.c:405-464 / :766-826)
    const int32_t xa = x[0].re, ya = x[0].im, xb = x[1].re, yb = x[1].im, xc = x[2].re, yc = x[2].im, xd = x[3].re, yd = x[3].im;
    x[0].re = wadd(wadd(wadd(xa, xb), xc), xd);
    x[0].im = wadd(wadd(wadd(ya, yb), yc), yd);
    x[1].re = wsub(wadd(wsub(xa, xb), xc), xd);
    x[1].im = wsub(wadd(wsub(ya, yb), yc), yd);
    if (!INV) {
      x[2].re = wsub(wsub(wadd(xa, yb), xc), yd);
      x[2].im = wadd(wsub(wsub(ya, xb), yc), xd);
      x[3].re = wadd(wsub(wsub(xa, yb), xc), yd);
      x[3].im = wsub(wsub(wadd(ya, xb), yc), xd);
    } else {
      x[2].re = wadd(wsub(wsub(xa, yb), xc), yd);
      x[2].im = wsub(wsub(wadd(ya, xb), yc), xd);
      x[3].re = wsub(wsub(wadd(xa, yb), xc), yd);
      x[3].im = wadd(wsub(wsub(ya, xb), yc), xd);
    }
    return;
  }
  int32_t a0r, a0i, a1r, a1i, a2r, a2i, a3r, a3i;
  if (STAGE == 0) {  // 4 guard bits (arm_cfft_radix4_q31.c:199-227)
    a0r = x[0].re >> 4; a0i = x[0].im >> 4; a1r = x[1].re >> 4; a1i = x[1].im >> 4;
    a2r = x[2].re >> 4; a2i = x[2].im >> 4; a3r = x[3].re >> 4; a3i = x[3].im >> 4;
  } else {
    a0r = x[0].re; a0i = x[0].im; a1r = x[1].re; a1i = x[1].im;
    a2r = x[2].re; a2i = x[2].im; a3r = x[3].re; a3i = x[3].im;
  }
  int32_t r1 = wadd(a0r, a2r), r2 = wsub(a0r, a2r);
  int32_t t1 = wadd(a1r, a3r);
  int32_t s1 = wadd(a0i, a2i), s2 = wsub(a0i, a2i);
  int32_t t2 = wadd(a1i, a3i);
  if (STAGE == 0) {
    x[0].re = wadd(r1, t1);
    x[0].im = wadd(s1, t2);
  } else {
    x[0].re = wadd(r1, t1) >> 2;
    x[0].im = wadd(s1, t2) >> 2;
  }
  r1 = wsub(r1, t1);
  s1 = wsub(s1, t2);
  t1 = wsub(a1i, a3i);
  t2 = wsub(a1r, a3r);
  int32_t o1r, o1i, o2r, o2i, o3r, o3i;
  if (!INV) {
    o1r = wadd(mulhi32(r1, w2.re), mulhi32(s1, w2.im));
    o1i = wsub(mulhi32(s1, w2.re), mulhi32(r1, w2.im));
    r1 = wadd(r2, t1); r2 = wsub(r2, t1);
    s1 = wsub(s2, t2); s2 = wadd(s2, t2);
    o2r = wadd(mulhi32(r1, w1.re), mulhi32(s1, w1.im));
    o2i = wsub(mulhi32(s1, w1.re), mulhi32(r1, w1.im));
    o3r = wadd(mulhi32(r2, w3.re), mulhi32(s2, w3.im));
    o3i = wsub(mulhi32(s2, w3.re), mulhi32(r2, w3.im));
  } else {
    o1r = wsub(mulhi32(r1, w2.re), mulhi32(s1, w2.im));
    o1i = wadd(mulhi32(s1, w2.re), mulhi32(r1, w2.im));
    r1 = wsub(r2, t1); r2 = wadd(r2, t1);
    s1 = wadd(s2, t2); s2 = wsub(s2, t2);
    o2r = wsub(mulhi32(r1, w1.re), mulhi32(s1, w1.im));
    o2i = wadd(mulhi32(s1, w1.re), mulhi32(r1, w1.im));
    o3r = wsub(mulhi32(r2, w3.re), mulhi32(s2, w3.im));
    o3i = wadd(mulhi32(s2, w3.re), mulhi32(r2, w3.im));
  }
  if (STAGE == 0) {
    x[1].re = wshl1(o1r); x[1].im = wshl1(o1i);
    x[2].re = wshl1(o2r); x[2].im = wshl1(o2i);
    x[3].re = wshl1(o3r); x[3].im = wshl1(o3i);
  } else {
    x[1].re = o1r >> 1; x[1].im = o1i >> 1;
    x[2].re = o2r >> 1; x[2].im = o2i >> 1;
    x[3].re = o3r >> 1; x[3].im = o3i >> 1;
  }
}

// arm_cfft_radix4by2[_inverse]_q31 pre-stage (arm_cfft_q31.c:778-798, :839-859): SMMULR/SMMLAR/SMMLSR
template <bool INV>
__device__ __forceinline__ void by2_q31(Q31::cpx &a, Q31::cpx &b, Q31::cpx w)
{
  const int32_t c = w.re, s = w.im;
  int32_t xt = wsub(a.re >> 2, b.re >> 2);
  a.re = wadd(a.re >> 2, b.re >> 2);
  int32_t yt = wsub(a.im >> 2, b.im >> 2);
  a.im = wadd(b.im >> 2, a.im >> 2);
  long long p0 = ((long long)xt * c + 0x80000000LL) >> 32;
  long long p1 = ((long long)yt * c + 0x80000000LL) >> 32;
  int32_t q0 = (int32_t)p0, q1 = (int32_t)p1;
  if (!INV) {
    q0 = (int32_t)(((long long)((unsigned long long)(long long)q0 << 32) + (long long)yt * s + 0x80000000LL) >> 32);
    q1 = (int32_t)(((long long)((unsigned long long)(long long)q1 << 32) - (long long)xt * s + 0x80000000LL) >> 32);
  } else {
    q0 = (int32_t)(((long long)((unsigned long long)(long long)q0 << 32) - (long long)yt * s + 0x80000000LL) >> 32);
    q1 = (int32_t)(((long long)((unsigned long long)(long long)q1 << 32) + (long long)xt * s + 0x80000000LL) >> 32);
  }
  b.re = wshl1(q0);
  b.im = wshl1(q1);
}

// dispatch by element type
template <class Q, int STAGE, bool INV>
__device__ __forceinline__ void bfly(typename Q::cpx *x, typename Q::cpx w1, typename Q::cpx w2, typename Q::cpx w3)
{
  if constexpr (sizeof(typename Q::scalar) == 2)
    bfly_q15<STAGE, INV>(x, w1, w2, w3);
  else
    bfly_q31<STAGE, INV>(x, w1, w2, w3);
}
template <class Q, bool INV>
__device__ __forceinline__ void by2(typename Q::cpx &a, typename Q::cpx &b, typename Q::cpx w)
{
  if constexpr (sizeof(typename Q::scalar) == 2)
    by2_q15<INV>(a, b, w);
  else
    by2_q31<INV>(a, b, w);
}
template <class Q>
__device__ __forceinline__ typename Q::cpx shl1(typename Q::cpx v)
{
  if constexpr (sizeof(typename Q::scalar) == 2)
    return typename Q::cpx{(int16_t)((uint16_t)v.re << 1), (int16_t)((uint16_t)v.im << 1)};
  else
    return typename Q::cpx{wshl1(v.re), wshl1(v.im)};
}

// ------------------------------------------------------------------------------------------
// plan: N = 2^LOGN, optional radix-2 pre-stage, sub-transform L = 4^M
// ------------------------------------------------------------------------------------------
template <int LOGN>
struct FixedPlan {
  static constexpr int N = 1 << LOGN;
  static constexpr bool BY2 = (LOGN & 1) != 0;
  static constexpr int LOGL = BY2 ? LOGN - 1 : LOGN;
  static constexpr int L = 1 << LOGL;
  static constexpr int M = LOGL / 2;             // radix-4 stages
  // 16 points per thread; single-wave workgroups below 2048 points (a wave that never waits for
  // another at a barrier streams best: q31 +4..9 %, q15 +4..7 % except 256 points -2.5 %)
  static constexpr int T = (N >= 1024) ? N / 16 : 64;
  static constexpr int TPB = (T * 16) / N;
  static constexpr int ELEMS = TPB * N;
  static constexpr int LDS_WORDS = ELEMS + (ELEMS >> 4);  // pad one point per 16
};

__device__ __forceinline__ int fpad(int e) { return e + (e >> 4); }

// stage class of radix-4 stage s of M: 0 first, 2 last, 1 middle
template <int S, int M>
struct StageKind { static constexpr int value = (S == 0) ? 0 : ((S == M - 1) ? 2 : 1); };

// One fused step over the 16 points of a thread.  S = first stage of the step, PAIR = two stages
// (S, S+1) or one.  r = group index inside the sub-transform (0 .. L/16-1), base = first point of
// the sub-transform in LDS, tw = table of the FULL transform (3N/4 entries), TWMOD0 = 1 or 2
// (arm_cfft_radix4by2 calls the butterflies with twidCoefModifier 2).
// TWL: optional LDS copy of every FOURTH table entry (tl[m] = tw[4m]): all stages after the first
// index the table in multiples of 4 (twidCoefModifier 4^S), so only stage 0 gathers from global memory.
struct NoTl {};
template <int MOD, class TWP, class TWL>
__device__ __forceinline__ auto tw_at(TWP tw, TWL tl, int ic)
{
  if constexpr (!std::is_same<TWL, NoTl>::value && (MOD % 4) == 0) return tl[ic >> 2];
  else return tw[ic];
}

template <class Q, int LOGL, int S, bool PAIR, bool INV, int TWMOD0, class TWP, class TWL>
__device__ __forceinline__ void fused_step(typename Q::word *lds, TWP tw, TWL tl, int base, int r)
{
  using cpx = typename Q::cpx;
  constexpr int M = LOGL / 2;
  constexpr int LOGN1 = LOGL - 2 * S;       // log2 n1 of stage S
  cpx x[16];
  if (PAIR) {
    constexpr int LOGQ = LOGN1 - 4;          // q = n1/16
    const int blk = r >> LOGQ, j = r & ((1 << LOGQ) - 1);
    const int e0 = base + (blk << LOGN1) + j;
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = Q::unpack(lds[fpad(e0 + (c << LOGQ))]);
    // stage S: for each b, butterfly over a (points c = 4a + b); twiddle index jj = j + b*q
    constexpr int MOD_S = TWMOD0 << (2 * S);
#pragma unroll
    for (int b = 0; b < 4; b++) {
      cpx y[4] = {x[b], x[4 + b], x[8 + b], x[12 + b]};
      const int ic = (j + (b << LOGQ)) * MOD_S;
      cpx w1{}, w2{}, w3{};
      if (StageKind<S, M>::value != 2) {
        w1 = Q::unpack(tw_at<MOD_S>(tw, tl, ic));
        w2 = Q::unpack(tw_at<MOD_S>(tw, tl, 2 * ic));
        w3 = Q::unpack(tw_at<MOD_S>(tw, tl, 3 * ic));
      }
      bfly<Q, StageKind<S, M>::value, INV>(y, w1, w2, w3);
      x[b] = y[0]; x[4 + b] = y[1]; x[8 + b] = y[2]; x[12 + b] = y[3];
    }
    // stage S+1: for each a, butterfly over b (points 4a + b); twiddle index j
    constexpr int MOD_S1 = TWMOD0 << (2 * (S + 1));
    cpx w1{}, w2{}, w3{};
    if (StageKind<S + 1, M>::value != 2) {
      const int ic = j * MOD_S1;
      w1 = Q::unpack(tw_at<MOD_S1>(tw, tl, ic));
      w2 = Q::unpack(tw_at<MOD_S1>(tw, tl, 2 * ic));
      w3 = Q::unpack(tw_at<MOD_S1>(tw, tl, 3 * ic));
    }
#pragma unroll
    for (int a = 0; a < 4; a++) bfly<Q, StageKind<S + 1, M>::value, INV>(&x[4 * a], w1, w2, w3);
#pragma unroll
    for (int c = 0; c < 16; c++) lds[fpad(e0 + (c << LOGQ))] = Q::pack(x[c]);
  } else {
    // single stage: 4 independent butterflies u = r + k*(L/16), k = 0..3
    constexpr int LOGN2 = LOGN1 - 2;
    constexpr int MOD_S = TWMOD0 << (2 * S);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int u = r + (k << (LOGL - 4));
      const int blk = u >> LOGN2, j = u & ((1 << LOGN2) - 1);
      const int e0 = base + (blk << LOGN1) + j;
      cpx y[4];
#pragma unroll
      for (int a = 0; a < 4; a++) y[a] = Q::unpack(lds[fpad(e0 + (a << LOGN2))]);
      cpx w1{}, w2{}, w3{};
      if (StageKind<S, M>::value != 2) {
        const int ic = j * MOD_S;
        w1 = Q::unpack(tw_at<MOD_S>(tw, tl, ic));
        w2 = Q::unpack(tw_at<MOD_S>(tw, tl, 2 * ic));
        w3 = Q::unpack(tw_at<MOD_S>(tw, tl, 3 * ic));
      }
      bfly<Q, StageKind<S, M>::value, INV>(y, w1, w2, w3);
#pragma unroll
      for (int a = 0; a < 4; a++) lds[fpad(e0 + (a << LOGN2))] = Q::pack(y[a]);
    }
  }
}

// all radix-4 stages from S on: a single stage first when the count is odd, then pairs
template <class Q, int LOGL, int S, bool INV, int TWMOD0, class TWP, class TWL = NoTl>
__device__ __forceinline__ void run_stages(typename Q::word *lds, TWP tw, int base, int r, TWL tl = TWL())
{
  constexpr int M = LOGL / 2;
  if constexpr (S < M) {
    constexpr bool PAIR = ((M - S) % 2) == 0;
    fused_step<Q, LOGL, S, PAIR, INV, TWMOD0>(lds, tw, tl, base, r);
    __syncthreads();
    run_stages<Q, LOGL, S + (PAIR ? 2 : 1), INV, TWMOD0>(lds, tw, base, r, tl);
  }
}

// One group of TPB transforms: global -> LDS, all stages, LDS -> global.  TWP = pointer to the twiddle
// table in global memory, TWL = optional LDS copy of every fourth entry (FixedTune::TW4).
template <class Q, int LOGN, bool INV, class TWP, class TWL = NoTl>
__device__ __forceinline__ void fixed_one_group(typename Q::word *__restrict__ data, TWP tw, typename Q::word *lds, uint32_t batch,
                                                uint32_t g, int bitrev, int tid, TWL tl = TWL())
{
  using PL = FixedPlan<LOGN>;
  using word = typename Q::word;
  using cpx = typename Q::cpx;
  const size_t gbase = (size_t)g * PL::ELEMS;  // first point of this workgroup's transforms
  const uint32_t nvalid = (batch - g * PL::TPB) >= (uint32_t)PL::TPB ? (uint32_t)PL::ELEMS : (batch - g * PL::TPB) * PL::N;

  // ---- global -> LDS, 16 bytes per lane ----
  constexpr int PER16 = 16 / sizeof(word);  // points per 16-byte access
  for (int v = tid; v < PL::ELEMS / PER16; v += PL::T) {
    const int e = v * PER16;
    if ((uint32_t)e < nvalid) {
      const uint4 raw = ld16_stream(data + gbase + e);
      const word *pw = reinterpret_cast<const word *>(&raw);
#pragma unroll
      for (int k = 0; k < PER16; k++) lds[fpad(e + k)] = pw[k];
    } else {
#pragma unroll
      for (int k = 0; k < PER16; k++) lds[fpad(e + k)] = word{};
    }
  }
  __syncthreads();

  // ---- radix-2 pre-stage for N = 2 * 4^k ----
  if constexpr (PL::BY2) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int u = tid + k * PL::T;                 // pair index over the workgroup
      const int t = u >> (LOGN - 1), i = u & (PL::L - 1);
      const int ea = t * PL::N + i, eb = ea + PL::L;
      cpx a = Q::unpack(lds[fpad(ea)]), b = Q::unpack(lds[fpad(eb)]);
      by2<Q, INV>(a, b, Q::unpack(tw[i]));
      lds[fpad(ea)] = Q::pack(a);
      lds[fpad(eb)] = Q::pack(b);
    }
    __syncthreads();
  }

  // ---- radix-4 stages on each sub-transform of length L ----
  {
    const int sub = tid >> (PL::LOGL - 4), r = tid & ((PL::L >> 4) - 1);
    run_stages<Q, PL::LOGL, 0, INV, PL::BY2 ? 2 : 1>(lds, tw, sub * PL::L, r, tl);
  }

  // ---- LDS -> global, bit reversal (over all LOGN bits) folded into the read, 16 B per lane ----
  for (int v = tid; v < PL::ELEMS / PER16; v += PL::T) {
    const int e = v * PER16;
    if ((uint32_t)e >= nvalid) continue;
    uint4 raw;
    word *pw = reinterpret_cast<word *>(&raw);
#pragma unroll
    for (int k = 0; k < PER16; k++) {
      const int idx = e + k, t = idx >> LOGN, i = idx & (PL::N - 1);
      const int src = bitrev ? (int)(__brev((unsigned)i) >> (32 - LOGN)) : i;
      cpx c = Q::unpack(lds[fpad(t * PL::N + src)]);
      if (PL::BY2) c = shl1<Q>(c);  // final << 1 of radix4by2 (arm_cfft_q15.c:814-831, arm_cfft_q31.c:807-824)
      pw[k] = Q::pack(c);
    }
    st16_stream(data + gbase + e, raw);
  }
}

// Per-size tuning.  TW4: every fourth entry of the 3N/4-entry twiddle table is copied to LDS first
// (3N/16 entries, 6 KiB at 4096 q31 points: still four workgroups per CU).  All stages after the first
// index the table in multiples of 4 (twidCoefModifier 4^S), so only stage 0 gathers its twiddles from
// global memory (a gather costs one L1 line per lane there).  Same-box A/B: q15 1024 +10 %, 2048 +17 %,
// 8192 +23 %, 16384 +47 %, 4096 3.45 TB/s (against 2.87 for an earlier persistent form that kept the
// WHOLE table in LDS; removed); q31 4096 +12 %, 8192 +15 %, 1024 / 2048 -2.5 % (left on global gathers).
template <class Q, int LOGN> struct FixedTune { static constexpr bool TW4 = false; };
template <int LOGN> struct FixedTune<Q15, LOGN> { static constexpr bool TW4 = (LOGN >= 10); };
template <int LOGN> struct FixedTune<Q31, LOGN> { static constexpr bool TW4 = (LOGN == 12 || LOGN == 13); };

template <class Q, int LOGN, bool INV>
__global__ __launch_bounds__(FixedPlan<LOGN>::T, 4) void cfft_fixed_kernel(typename Q::word *__restrict__ data,
                                                                          const typename Q::tword *__restrict__ tw, uint32_t batch,
                                                                          uint32_t ngroups, int bitrev)
{
  using PL = FixedPlan<LOGN>;
  using word = typename Q::word;
  using tword = typename Q::tword;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  word *lds = reinterpret_cast<word *>(lds_raw);
  const uint32_t g0 = blockIdx.x + blockIdx.y * gridDim.x;
  if (g0 >= ngroups) return;
  if constexpr (FixedTune<Q, LOGN>::TW4) {
    tword *tl = reinterpret_cast<tword *>(lds + PL::LDS_WORDS);
    for (int i = threadIdx.x; i < 3 * PL::N / 16; i += PL::T) tl[i] = tw[4 * i];
    // (the barrier after the data load inside fixed_one_group orders these writes before the stages)
    fixed_one_group<Q, LOGN, INV>(data, tw, lds, batch, g0, bitrev, threadIdx.x, (const tword *)tl);
  } else {
    fixed_one_group<Q, LOGN, INV>(data, tw, lds, batch, g0, bitrev, threadIdx.x);
  }
}

// ------------------------------------------------------------------------------------------
// Direct form (512 ... 16384 points, q31 also 256; bit-reversed output): the first step takes its 16 points straight from global
// memory into registers (point tid + c*N/16: every load instruction is one contiguous run across the workgroup) and the
// last step stores its 16 results straight to global memory -- thread tid transforms the 16-point block brev(tid) there,
// so result c lands at index brev4(c)*N/16 + tid, contiguous across the lanes again.  Against cfft_fixed_kernel that is
// half the LDS traffic (two of four round trips at 4096 points), two barriers fewer, and no address arithmetic for the
// staging copies; the stages, their order and every rounding step are the same functions (bfly, StageKind).
// LDS image: point p at word p + (p >> (LOGN-5)).  The last step reads blocks 2^(LOGN-5) points apart across 32 lanes:
// that stride becomes odd; the first step's contiguous runs stay contiguous; the middle steps pick which (blk, j) bits
// the low five lane bits carry so that their 32 lanes cover 32 banks too (dlane_map).  All offsets inside a step are
// compile-time constants (doff).
// N = 2 * 4^k (2048, 8192): points c and c + 8 of a thread are N/2 apart, so the radix-2 pre-stage of
// arm_cfft_radix4by2_* runs on the registers right after the loads, and points c, c+2, c+4, c+6 are a stage-0 butterfly of
// the thread's half: the first step is pre-stage + stage 0 of both halves.  The two half-length transforms then share
// the LDS image (blocks never straddle the halves), the twiddle modifier starts at 2, and the final << 1 goes into the
// last step's stores.  An odd number of stages between the first step and the last pair (8192) runs one single-stage
// step first.
// ------------------------------------------------------------------------------------------
namespace dtw {   // twiddle tables of the direct form's middle stages, see DirectPlan
constexpr int nbf(int logl, int s) { return 1 << (logl - 2 * s - 2); }
constexpr bool compact(int logl, int m, int s1, int s) { return s >= s1 && s <= m - 3 && nbf(logl, s) <= 64; }
constexpr bool any_quarter(int logl, int m, int s1, int s) { return s > m - 3 ? false : (!compact(logl, m, s1, s) || any_quarter(logl, m, s1, s + 1)); }
constexpr int coff(int logl, int m, int s1, int tl4, int s)
{
  return s <= s1 ? tl4 : coff(logl, m, s1, tl4, s - 1) + (compact(logl, m, s1, s - 1) ? 3 * nbf(logl, s - 1) : 0);
}
}  // namespace dtw

// CT: the late middle stages read compact twiddle tables (see below); chosen per element type and size (DirectTune)
template <int LOGN, bool CT = false>
struct DirectPlan {
  static_assert(LOGN >= 8 && LOGN <= 14, "direct form: 256 ... 16384 points");
  static constexpr bool BY2 = (LOGN & 1) != 0;
  static constexpr int LOGL = BY2 ? LOGN - 1 : LOGN;   // the radix-4 transforms' length
  static constexpr int TWMOD0 = BY2 ? 2 : 1;           // their twidCoefModifier at stage 0
  static constexpr int N = 1 << LOGN, T = N / 16, LOGT = LOGN - 4, M = LOGL / 2, SH = LOGN - 5;
  static constexpr int LDS_WORDS = N + (N >> SH);
  // 256 and 512 points take 16 / 32 threads: a one-wave workgroup holds 4 / 2 transforms, each with its own image
  static constexpr int TPB = T >= 64 ? 1 : 64 / T, WG = T * TPB;
  // distance between the images of a workgroup's transforms: at four transforms (16 lanes each) = 16 mod 64 words, so that
  // the same word of the four images falls into four different groups of 16 banks
  static constexpr int IMG = LDS_WORDS + (TPB == 4 ? (64 + 16 - LDS_WORDS % 64) % 64 : 0);
  static __device__ __forceinline__ int pad(int p) { return p + (p >> SH); }
  // offset of point e0 + (c << LOGQ) from pad(e0): exact whenever those points stay inside one aligned block of
  // 2^(LOGQ+4) points that e0 starts with a remainder below 2^LOGQ (see the steps)
  static constexpr int doff(int c, int logq) { return (c << logq) + ((c << logq) >> SH); }
  // ---- twiddles of the middle steps (stages S1 .. M-3) in LDS ----
  // A stage with few distinct butterflies (nbf <= 64: the last two middle stages) reads a COMPACT table of its own,
  // T[3 a + r - 1] = tw[r a MOD_S] for butterfly a: the three twiddles of a butterfly are adjacent, consecutive butterflies
  // three words apart -- conflict-free for any set of lanes.  Read from the every-fourth copy instead (index a MOD_S r / 4),
  // their few distinct addresses fall into one or two banks: 8- and 16-way conflicts, 58 % of all LDS cycles of the
  // 16384-point q15 kernel (profiles/r03_pmc_sq_cfft_q15_16384.txt).  The earlier middle stages (8192 and 16384 points only)
  // keep the every-fourth copy (strides 1 ... 12 words across the lanes).
  static constexpr int S1 = (BY2 || (M & 1)) ? 1 : 2;                                   // first middle stage
  static constexpr int nbf(int s) { return dtw::nbf(LOGL, s); }                         // distinct butterflies of stage s
  static constexpr bool compact(int s) { return CT && dtw::compact(LOGL, M, S1, s); }
  static constexpr int TL4 = (!CT || dtw::any_quarter(LOGL, M, S1, S1)) ? 3 * N / 16 : 0;   // entries of the every-fourth copy
  static constexpr int coff(int s) { return CT ? dtw::coff(LOGL, M, S1, TL4, s) : TL4; }      // first entry of stage s's compact table
  static constexpr int TL_ENTRIES = CT ? dtw::coff(LOGL, M, S1, TL4, M - 2 > S1 ? M - 2 : S1) : TL4;
};
// Same-box A/B of the compact tables, bit-exact either way: q15 4096 points 0.667 -> 0.694 of the roofline, 16384 0.515 -> 0.554;
// 1024 / 8192 points and every q31 size lose 2-5 % (the tables' gathered fill costs more than their reads save) and keep the
// every-fourth copy alone.
template <class Q, int LOGN> struct DirectTune { static constexpr bool COMPACT = false; };
template <> struct DirectTune<Q15, 12> { static constexpr bool COMPACT = true; };
template <> struct DirectTune<Q15, 14> { static constexpr bool COMPACT = true; };
template <class Q, int LOGN> using DPlan = DirectPlan<LOGN, DirectTune<Q, LOGN>::COMPACT>;
template <class TW> struct TlCompact { const TW *t; };
template <class T> struct is_compact : std::false_type {};
template <class TW> struct is_compact<TlCompact<TW>> : std::true_type {};

// (blk, j) of thread tid in a middle step over blocks of 2^LOGN1 points (j < 2^(LOGN1-4)): any bijection is allowed, the
// step is in place per thread.  The word address is (blk << LOGN1) + j + const + its own >> SH, so lane bit k (k < 5) is
// given to the (blk, j) bit that moves bank bit k: j bit k directly, blk bit i through the pad term at bank bit
// LOGN1 - SH + i.
template <int LOGN, int LOGN1>
__device__ __forceinline__ void dlane_map(int tid, int &blk, int &j)
{
  constexpr int LOGQ = LOGN1 - 4, SH = LOGN - 5, NB = LOGN - LOGN1, B0 = LOGN1 - SH;
  if constexpr (LOGQ >= 5 || NB == 0) {
    j = tid & ((1 << LOGQ) - 1);
    blk = tid >> LOGQ;
  } else if constexpr (B0 > 0) {
    static_assert(B0 + NB == 5 && B0 <= LOGQ, "lane bits 0..4 = j bits 0..B0-1, then every blk bit");
    j = (tid & ((1 << B0) - 1)) | ((tid >> 5) << B0);
    blk = (tid >> B0) & ((1 << NB) - 1);
  } else {
    constexpr int HB = 5 - LOGQ;  // top HB bits of blk from lane bits LOGQ..4
    static_assert(HB <= NB, "lane bits 0..4 = every j bit, then the top blk bits");
    j = tid & ((1 << LOGQ) - 1);
    blk = (((tid >> LOGQ) & ((1 << HB) - 1)) << (NB - HB)) | (tid >> 5);
  }
}

// Stage S on the 16 points x[c] = point e0 + (c << LOGQ) of a block of 2^LOGN1 = 2^(LOGL - 2S) points (LOGQ = LOGN1 - 4);
// j = e0's offset inside the block's first sixteenth: butterfly b over x[b], x[4+b], x[8+b], x[12+b].  TWS(ic) fetches
// table entry ic; TWMOD0 = the transform's twidCoefModifier at stage 0.
template <class Q, int LOGL, int S, bool INV, int TWMOD0, class TWS>
__device__ __forceinline__ void stage_butterflies(typename Q::cpx *x, int j, TWS tws)
{
  using cpx = typename Q::cpx;
  constexpr int M = LOGL / 2, LOGQ = LOGL - 2 * S - 4;
  constexpr int MOD_S = TWMOD0 << (2 * S);
#pragma unroll
  for (int b = 0; b < 4; b++) {
    cpx y[4] = {x[b], x[4 + b], x[8 + b], x[12 + b]};
    cpx w1{}, w2{}, w3{};
    if constexpr (StageKind<S, M>::value != 2) {
      if constexpr (is_compact<TWS>::value) {
        const int a3 = 3 * (j + (b << LOGQ));
        w1 = Q::unpack(tws.t[a3]);
        w2 = Q::unpack(tws.t[a3 + 1]);
        w3 = Q::unpack(tws.t[a3 + 2]);
      } else {
        const int ic = (j + (b << LOGQ)) * MOD_S;
        w1 = Q::unpack(tws(ic));
        w2 = Q::unpack(tws(2 * ic));
        w3 = Q::unpack(tws(3 * ic));
      }
    }
    bfly<Q, StageKind<S, M>::value, INV>(y, w1, w2, w3);
    x[b] = y[0]; x[4 + b] = y[1]; x[8 + b] = y[2]; x[12 + b] = y[3];
  }
}

// two fused stages S, S+1 on the same 16 points: stage S as above, then stage S+1's butterfly a over x[4a .. 4a+3]
template <class Q, int LOGL, int S, bool INV, int TWMOD0, class TWS, class TWS1>
__device__ __forceinline__ void pair_butterflies(typename Q::cpx *x, int j, TWS tws, TWS1 tws1)
{
  using cpx = typename Q::cpx;
  constexpr int M = LOGL / 2;
  constexpr int MOD_S1 = TWMOD0 << (2 * (S + 1));
  stage_butterflies<Q, LOGL, S, INV, TWMOD0>(x, j, tws);
  cpx w1{}, w2{}, w3{};
  if constexpr (StageKind<S + 1, M>::value != 2) {
    if constexpr (is_compact<TWS1>::value) {
      w1 = Q::unpack(tws1.t[3 * j]);
      w2 = Q::unpack(tws1.t[3 * j + 1]);
      w3 = Q::unpack(tws1.t[3 * j + 2]);
    } else {
      const int ic = j * MOD_S1;
      w1 = Q::unpack(tws1(ic));
      w2 = Q::unpack(tws1(2 * ic));
      w3 = Q::unpack(tws1(3 * ic));
    }
  }
#pragma unroll
  for (int a = 0; a < 4; a++) bfly<Q, StageKind<S + 1, M>::value, INV>(&x[4 * a], w1, w2, w3);
}

// middle steps (in place in LDS) from stage S up to, not including, the last pair: one single-stage step first when
// their number is odd, then pairs.  Blocks are counted over the whole N-point image (both halves of a 2 * 4^k transform).
template <class Q, int LOGN, int S, bool INV>
__device__ __forceinline__ void direct_middle(typename Q::word *lds, const typename Q::tword *tl, int tid)
{
  using PL = DPlan<Q, LOGN>;
  using cpx = typename Q::cpx;
  if constexpr (S < PL::M - 2) {
    constexpr bool PAIR = ((PL::M - 2 - S) % 2) == 0;
    constexpr int LOGN1 = PL::LOGL - 2 * S, LOGQ = LOGN1 - 4;
    int blk, j;
    dlane_map<LOGN, LOGN1>(tid, blk, j);
    const int e0 = (blk << LOGN1) + j;
    typename Q::word *base = lds + PL::pad(e0);
    cpx x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = Q::unpack(base[PL::doff(c, LOGQ)]);
    auto t4 = [&](int ic) { return tl[ic >> 2]; };
    const TlCompact<typename Q::tword> c0{tl + PL::coff(S)}, c1{tl + PL::coff(S + 1)};
    auto run = [&](auto twa, auto twb) {
      if constexpr (PAIR) pair_butterflies<Q, PL::LOGL, S, INV, PL::TWMOD0>(x, j, twa, twb);
      else stage_butterflies<Q, PL::LOGL, S, INV, PL::TWMOD0>(x, j, twa);
    };
    if constexpr (PL::compact(S)) run(c0, c1);                      // (a compact stage is followed by compact stages)
    else if constexpr (PAIR && PL::compact(S + 1)) run(t4, c1);
    else run(t4, t4);
#pragma unroll
    for (int c = 0; c < 16; c++) base[PL::doff(c, LOGQ)] = Q::pack(x[c]);
    __syncthreads();
    direct_middle<Q, LOGN, S + (PAIR ? 2 : 1), INV>(lds, tl, tid);
  }
}

template <class Q> struct DirectIo;
template <> struct DirectIo<Q15> {
  static __device__ __forceinline__ uint32_t load(const __amdgpu_buffer_rsrc_t &rs, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 2); }
  static __device__ __forceinline__ void store(uint32_t w, const __amdgpu_buffer_rsrc_t &rs, int voff, int soff) { __builtin_amdgcn_raw_buffer_store_b32(w, rs, voff, soff, 2); }
};
template <> struct DirectIo<Q31> {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ int2 load(const __amdgpu_buffer_rsrc_t &rs, int voff, int soff)
  {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 2);
    return make_int2((int)w[0], (int)w[1]);
  }
  static __device__ __forceinline__ void store(int2 v, const __amdgpu_buffer_rsrc_t &rs, int voff, int soff)
  {
    u32x2 w;
    w[0] = (unsigned)v.x;
    w[1] = (unsigned)v.y;
    __builtin_amdgcn_raw_buffer_store_b64(w, rs, voff, soff, 2);
  }
};

template <class Q, int LOGN, int S, class PL = DPlan<Q, LOGN>>
__device__ __forceinline__ void fill_compact(typename Q::tword *tl, const typename Q::tword *__restrict__ tw, int wtid)
{
  if constexpr (S <= PL::M - 3) {
    if constexpr (PL::compact(S)) {
      constexpr int MOD_S = PL::TWMOD0 << (2 * S);
      for (int i = wtid; i < 3 * PL::nbf(S); i += PL::WG) {
        const int a = i / 3, r = i - 3 * a + 1;
        tl[PL::coff(S) + i] = tw[r * a * MOD_S];
      }
    }
    fill_compact<Q, LOGN, S + 1, PL>(tl, tw, wtid);
  }
}

// The transform itself.  On entry x[c] = point tid + c*T; on return x[c] = the result whose index in bit-reversed
// (= natural) output order is brev4(c)*T + tid, before the final << 1 of the 2 * 4^k sizes (direct_out).  `lds` = this
// transform's image (PL::LDS_WORDS words), `tl` = the workgroup's room for 3N/16 table words, tid = thread inside the
// transform, wtid = thread inside the workgroup (PL::WG threads).  Ends without a barrier: the last step only reads.
template <class Q, int LOGN, bool INV>
__device__ __forceinline__ void direct_transform(typename Q::cpx *x, typename Q::word *lds, typename Q::tword *tl,
                                                 const typename Q::tword *__restrict__ tw, int tid, int wtid)
{
  using PL = DPlan<Q, LOGN>;
  using word = typename Q::word;
  using cpx = typename Q::cpx;
  constexpr int T = PL::T, M = PL::M;
  // every fourth table entry to LDS for the middle stages that read it (twidCoefModifier 4^S, S >= 1), compact tables for the others
  if constexpr (PL::TL4 > 0) {
    for (int i = wtid; i < PL::TL4; i += PL::WG) tl[i] = tw[4 * i];
  }
  fill_compact<Q, LOGN, PL::S1>(tl, tw, wtid);
  auto tg = [&](int ic) { return tw[ic]; };
  constexpr int S1 = PL::S1;  // first stage of the middle steps
  if constexpr (PL::BY2) {
    // radix-2 pre-stage on (point i, point i + N/2), i = tid + c*T, then stage 0 of both halves: butterfly k of half h
    // over x[8h + k], x[8h + k + 2], x[8h + k + 4], x[8h + k + 6] (points N/8 apart), twiddle index 2 (tid + k*T)
#pragma unroll
    for (int c = 0; c < 8; c++) by2<Q, INV>(x[c], x[c + 8], Q::unpack(tw[tid + c * T]));
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        cpx y[4] = {x[8 * h + k], x[8 * h + k + 2], x[8 * h + k + 4], x[8 * h + k + 6]};
        const int ic = 2 * (tid + k * T);
        bfly<Q, 0, INV>(y, Q::unpack(tw[ic]), Q::unpack(tw[2 * ic]), Q::unpack(tw[3 * ic]));
        x[8 * h + k] = y[0]; x[8 * h + k + 2] = y[1]; x[8 * h + k + 4] = y[2]; x[8 * h + k + 6] = y[3];
      }
    }
  } else if constexpr ((M & 1) == 0) {
    pair_butterflies<Q, LOGN, 0, INV, 1>(x, tid, tg, tg);
  } else {
    // single stage 0: butterfly k over points u + a*N/4, u = tid + k*T (x[k + 4a])
#pragma unroll
    for (int k = 0; k < 4; k++) {
      cpx y[4] = {x[k], x[4 + k], x[8 + k], x[12 + k]};
      const int ic = tid + k * T;
      bfly<Q, 0, INV>(y, Q::unpack(tw[ic]), Q::unpack(tw[2 * ic]), Q::unpack(tw[3 * ic]));
      x[k] = y[0]; x[4 + k] = y[1]; x[8 + k] = y[2]; x[12 + k] = y[3];
    }
  }
  {
    word *base = lds + PL::pad(tid);
#pragma unroll
    for (int c = 0; c < 16; c++) base[PL::doff(c, PL::LOGT)] = Q::pack(x[c]);
  }
  __syncthreads();

  // ---- middle steps, in place in LDS ----
  direct_middle<Q, LOGN, S1, INV>(lds, tl, tid);

  // ---- last step: block brev(tid) ----
  const int blk = (int)(__brev((unsigned)tid) >> (32 - PL::LOGT));
  const word *base = lds + PL::pad(blk << 4);
#pragma unroll
  for (int c = 0; c < 16; c++) x[c] = Q::unpack(base[PL::doff(c, 0)]);
  // stage M-2 of the pair has the uniform twiddles b * TWMOD0 * 4^(M-2) (j = 0): scalar loads from the global table
  pair_butterflies<Q, PL::LOGL, M - 2, INV, PL::TWMOD0>(x, 0, tg, tg);
}
// output index of result c of thread tid, in units of T
__device__ __forceinline__ constexpr int direct_rc(int c) { return ((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3); }
// result as arm_cfft_* leaves it: final << 1 of radix4by2 (arm_cfft_q15.c:814-831, arm_cfft_q31.c:807-824)
template <class Q, int LOGN>
__device__ __forceinline__ typename Q::cpx direct_out(typename Q::cpx v)
{
  return DirectPlan<LOGN>::BY2 ? shl1<Q>(v) : v;
}

template <class Q, int LOGN, bool INV>
__global__ __launch_bounds__(DirectPlan<LOGN>::WG) void cfft_fixed_direct_kernel(typename Q::word *__restrict__ data,
                                                                                const typename Q::tword *__restrict__ tw, uint32_t batch)
{
  using PL = DPlan<Q, LOGN>;
  using word = typename Q::word;
  using tword = typename Q::tword;
  using cpx = typename Q::cpx;
  constexpr int W = sizeof(word), T = PL::T, TPB = PL::TPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t g = blockIdx.x + blockIdx.y * gridDim.x;   // group of TPB transforms
  const uint32_t t0 = g * TPB;
  if (t0 >= batch) return;
  const int wtid = threadIdx.x, tid = wtid & (T - 1), tr = TPB == 1 ? 0 : wtid / T;
  word *lds = reinterpret_cast<word *>(lds_raw) + tr * PL::IMG;
  tword *tl = reinterpret_cast<tword *>(reinterpret_cast<word *>(lds_raw) + TPB * PL::IMG);
  // one descriptor over the group's valid transforms: loads past it read zeros, stores past it are dropped
  const uint32_t nvalid = batch - t0 < (uint32_t)TPB ? batch - t0 : (uint32_t)TPB;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(data + (size_t)t0 * PL::N, 0, (int)(nvalid * PL::N * W), 0x00020000);
  const int voff = (tr * PL::N + tid) * W;
  // first step: points tid + c*T from global memory; last step: results straight back to it
  cpx x[16];
#pragma unroll
  for (int c = 0; c < 16; c++) x[c] = Q::unpack(DirectIo<Q>::load(rs, voff, c * T * W));
  direct_transform<Q, LOGN, INV>(x, lds, tl, tw, tid, wtid);
#pragma unroll
  for (int c = 0; c < 16; c++) DirectIo<Q>::store(Q::pack(direct_out<Q, LOGN>(x[c])), rs, voff, direct_rc(c) * T * W);
}

// ------------------------------------------------------------------------------------------
// q31, 8192 / 16384 points: the direct form with a HALF-FOOTPRINT exchange (the f32 kernels' SPLIT, fft_core_f32.hpp).
// Above, a q31 image is N 8-byte words: 128 KiB at 16384 points, ONE 1024-thread workgroup per CU, whose load, transform and
// store phases then have nothing to overlap with (SQ_WAIT_INST_ANY = half of all wave-cycles, profiles/r02_pmc_sq_cfft_q31.txt).
// Here the points stay in REGISTERS from step to step and the image holds N 32-bit words: every exchange between two steps
// runs twice through it, real parts then imaginary parts (write re | barrier | read re | barrier | write im | barrier |
// read im), addressed exactly as the in-place steps above address their words -- a step's outputs leave through ITS map,
// the next step's inputs arrive through the next map, so the bank behaviour of both is what dlane_map arranged.  Three
// barriers more per exchange, half the image: two 1024-thread workgroups per CU at 16384 points, three 512-thread ones at
// 8192.  Stages, order and rounding are the same functions (bfly, by2, StageKind): the result does not change by a bit.
// Twiddles of the stages whose table index is a multiple of 2^TLS come from an LDS copy of every 2^TLS-th entry (TLS = 2:
// the quarter table, 12 KiB at 8192; TLS = 4 at 16384, 6 KiB -- the quarter table would push the workgroup past 80 KiB), the
// others from the global table (L2).
// ------------------------------------------------------------------------------------------
#ifndef ARMFFT_Q31_V14
#define ARMFFT_Q31_V14 2
#endif
#ifndef ARMFFT_Q31_V13
#define ARMFFT_Q31_V13 1
#endif
#ifndef ARMFFT_Q31_V12
#define ARMFFT_Q31_V12 1
#endif
template <int LOGN>
struct SplitQ31 {
  using PL = DirectPlan<LOGN>;
  static constexpr int TLS = LOGN >= 14 ? 4 : 2;
  static constexpr int TL_ENTRIES = (3 * PL::N / 4) >> TLS;
  static constexpr size_t LDS_BYTES = (size_t)PL::LDS_WORDS * sizeof(int32_t) + (size_t)TL_ENTRIES * sizeof(int2);
  // V: the plan's threads ("virtual threads", 16 points each) that one hardware thread runs side by side -- V = 2 halves the waves
  // every exchange barrier has to collect and doubles the independent work between two barriers.  Same-box A/B, bit-exact either
  // way: 16384 points V = 2 (two 512-thread workgroups per CU, 128 VGPRs) 0.515 -> 0.562 of the roofline, V = 4 0.54; 8192 points
  // V = 2 0.66 -> 0.633, 4096 points 0.66 -> 0.64-0.66: those keep V = 1.
  static constexpr int V = LOGN >= 14 ? ARMFFT_Q31_V14 : (LOGN == 13 ? ARMFFT_Q31_V13 : ARMFFT_Q31_V12);
  static constexpr int TT = PL::T / V;                                  // hardware threads of the workgroup
  static constexpr int MINW1 = LOGN >= 14 ? 8 : (LOGN == 13 ? 6 : 7);   // waves per SIMD the register budget is cut for (2 x 16 / 3 x 8 waves per CU ... per SIMD)
  static constexpr int MINW = MINW1 / V > 0 ? MINW1 / V : 1;
};

// one exchange: x[v][c] leaves as word baseA[v] + doff(c, LOGQA), x[v][c] returns as word baseB[v] + doff(c, LOGQB)
template <int LOGN, int LOGQA, int LOGQB, int V>
__device__ __forceinline__ void q31_exchange(int32_t *lds, const int (&baseA)[V], const int (&baseB)[V], Q31::cpx (&x)[V][16])
{
  using PL = DirectPlan<LOGN>;
#pragma unroll
  for (int v = 0; v < V; v++)
#pragma unroll
    for (int c = 0; c < 16; c++) lds[baseA[v] + PL::doff(c, LOGQA)] = x[v][c].re;
  __syncthreads();
#pragma unroll
  for (int v = 0; v < V; v++)
#pragma unroll
    for (int c = 0; c < 16; c++) x[v][c].re = lds[baseB[v] + PL::doff(c, LOGQB)];
  __syncthreads();
#pragma unroll
  for (int v = 0; v < V; v++)
#pragma unroll
    for (int c = 0; c < 16; c++) lds[baseA[v] + PL::doff(c, LOGQA)] = x[v][c].im;
  __syncthreads();
#pragma unroll
  for (int v = 0; v < V; v++)
#pragma unroll
    for (int c = 0; c < 16; c++) x[v][c].im = lds[baseB[v] + PL::doff(c, LOGQB)];
}

// steps from stage S on, the points in registers; (baseP, LOGQP) = where the previous step's outputs belong; vt[v] = the plan's thread
template <int LOGN, int S, bool INV, int LOGQP, bool FIRST, int V>
__device__ __forceinline__ void q31_split_steps(Q31::cpx (&x)[V][16], int32_t *lds, const int2 *tl, const int2 *__restrict__ tw,
                                                const int (&vt)[V], const int (&baseP)[V])
{
  using PL = DirectPlan<LOGN>;
  constexpr int TLS = SplitQ31<LOGN>::TLS;
  auto tg = [&](int ic) { return tw[ic]; };
  auto tld = [&](int ic) { return tl[ic >> TLS]; };
  if constexpr (!FIRST) __syncthreads();   // every thread has read the previous exchange's imaginary parts
  if constexpr (S < PL::M - 2) {
    constexpr bool PAIR = ((PL::M - 2 - S) % 2) == 0;
    constexpr int LOGN1 = PL::LOGL - 2 * S, LOGQ = LOGN1 - 4;
    constexpr bool LDS_S = ((PL::TWMOD0 << (2 * S)) & ((1 << TLS) - 1)) == 0;        // stage S's indices are multiples of 2^TLS
    int base[V], jv[V];
#pragma unroll
    for (int v = 0; v < V; v++) {
      int blk;
      dlane_map<LOGN, LOGN1>(vt[v], blk, jv[v]);
      base[v] = PL::pad((blk << LOGN1) + jv[v]);
    }
    q31_exchange<LOGN, LOGQP, LOGQ, V>(lds, baseP, base, x);
    // (compact tables for the last two middle stages, as in the q15 direct form, were measured here too: 4096 / 8192 points -2 %,
    // 16384 +-0 -- this kernel waits on its exchange barriers, not on the table reads)
#pragma unroll
    for (int v = 0; v < V; v++) {
      if constexpr (PAIR) {
        if constexpr (LDS_S) pair_butterflies<Q31, PL::LOGL, S, INV, PL::TWMOD0>(x[v], jv[v], tld, tld);
        else pair_butterflies<Q31, PL::LOGL, S, INV, PL::TWMOD0>(x[v], jv[v], tg, tld);       // stage S + 1: multiples of 4 * 2^TLS at least
      } else {
        if constexpr (LDS_S) stage_butterflies<Q31, PL::LOGL, S, INV, PL::TWMOD0>(x[v], jv[v], tld);
        else stage_butterflies<Q31, PL::LOGL, S, INV, PL::TWMOD0>(x[v], jv[v], tg);
      }
    }
    q31_split_steps<LOGN, S + (PAIR ? 2 : 1), INV, LOGQ, false, V>(x, lds, tl, tw, vt, base);
  } else {
    // last step: block brev(thread); stage M-2 of the pair has uniform twiddles (j = 0): scalar loads from the global table
    int base[V];
#pragma unroll
    for (int v = 0; v < V; v++) base[v] = PL::pad((int)(__brev((unsigned)vt[v]) >> (32 - PL::LOGT)) << 4);
    q31_exchange<LOGN, LOGQP, 0, V>(lds, baseP, base, x);
#pragma unroll
    for (int v = 0; v < V; v++) pair_butterflies<Q31, PL::LOGL, PL::M - 2, INV, PL::TWMOD0>(x[v], 0, tg, tg);
  }
}

template <int LOGN, bool INV>
__global__ __launch_bounds__(SplitQ31<LOGN>::TT, SplitQ31<LOGN>::MINW) void cfft_q31_split_kernel(int2 *__restrict__ data,
                                                                                              const int2 *__restrict__ tw, uint32_t batch)
{
  using PL = DirectPlan<LOGN>;
  using SP = SplitQ31<LOGN>;
  using cpx = Q31::cpx;
  static_assert(PL::TPB == 1 && LOGN >= 12 && LOGN <= 14, "one transform per workgroup");
  static_assert(LOGN != 14 || (1 << SP::TLS) <= 16, "stage 2 onwards indexes the table in multiples of 16");
  constexpr int W = sizeof(int2), T = PL::T, M = PL::M, V = SP::V, TT = SP::TT;
  static_assert(TT * V == T && TT >= 64, "whole waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t t0 = blockIdx.x + blockIdx.y * gridDim.x;
  if (t0 >= batch) return;
  int32_t *lds = reinterpret_cast<int32_t *>(lds_raw);
  int2 *tl = reinterpret_cast<int2 *>(lds + PL::LDS_WORDS);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(data + (size_t)t0 * PL::N, 0, (int)(PL::N * W), 0x00020000);
  int vt[V], base0[V];
#pragma unroll
  for (int v = 0; v < V; v++) vt[v] = (int)threadIdx.x + v * TT, base0[v] = PL::pad(vt[v]);
  cpx x[V][16];
#pragma unroll
  for (int v = 0; v < V; v++)
#pragma unroll
    for (int c = 0; c < 16; c++) x[v][c] = Q31::unpack(DirectIo<Q31>::load(rs, vt[v] * W, c * T * W));
  for (int i = threadIdx.x; i < SP::TL_ENTRIES; i += TT) tl[i] = tw[i << SP::TLS];
  // ---- first step on the loaded registers (as direct_transform) ----
  constexpr int S1 = (PL::BY2 || (M & 1)) ? 1 : 2;
#pragma unroll
  for (int v = 0; v < V; v++) {
    const int tid = vt[v];
    cpx *xv = x[v];
    if constexpr (PL::BY2) {
#pragma unroll
      for (int c = 0; c < 8; c++) by2<Q31, INV>(xv[c], xv[c + 8], Q31::unpack(tw[tid + c * T]));
#pragma unroll
      for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
          cpx y[4] = {xv[8 * h + k], xv[8 * h + k + 2], xv[8 * h + k + 4], xv[8 * h + k + 6]};
          const int ic = 2 * (tid + k * T);
          bfly<Q31, 0, INV>(y, Q31::unpack(tw[ic]), Q31::unpack(tw[2 * ic]), Q31::unpack(tw[3 * ic]));
          xv[8 * h + k] = y[0]; xv[8 * h + k + 2] = y[1]; xv[8 * h + k + 4] = y[2]; xv[8 * h + k + 6] = y[3];
        }
      }
    } else if constexpr ((M & 1) == 0) {
      auto tg = [&](int ic) { return tw[ic]; };
      pair_butterflies<Q31, LOGN, 0, INV, 1>(xv, tid, tg, tg);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        cpx y[4] = {xv[k], xv[4 + k], xv[8 + k], xv[12 + k]};
        const int ic = tid + k * T;
        bfly<Q31, 0, INV>(y, Q31::unpack(tw[ic]), Q31::unpack(tw[2 * ic]), Q31::unpack(tw[3 * ic]));
        xv[k] = y[0]; xv[4 + k] = y[1]; xv[8 + k] = y[2]; xv[12 + k] = y[3];
      }
    }
  }
  // ---- the remaining steps, an exchange before each ----
  q31_split_steps<LOGN, S1, INV, PL::LOGT, true, V>(x, lds, tl, tw, vt, base0);
#pragma unroll
  for (int v = 0; v < V; v++)
#pragma unroll
    for (int c = 0; c < 16; c++) DirectIo<Q31>::store(Q31::pack(direct_out<Q31, LOGN>(x[v][c])), rs, vt[v] * W, direct_rc(c) * T * W);
}

template <int LOGN, bool INV>
static hipError_t launch_q31_split(int2 *data, const int2 *tw, uint32_t batch, hipStream_t stream)
{
  constexpr size_t lds_bytes = SplitQ31<LOGN>::LDS_BYTES;
  static std::atomic<bool> attr_done[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto kern = cfft_q31_split_kernel<LOGN, INV>;
  if (lds_bytes > 48 * 1024 && (dev < 0 || dev >= 16 || !attr_done[dev])) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  const uint32_t gx = batch < 65536u ? batch : 65536u, gy = (batch + gx - 1) / gx;
  note_kernel(INV ? "cfft_q31_split_kernel<%d, true>" : "cfft_q31_split_kernel<%d, false>", false, LOGN);
  kern<<<dim3(gx, gy, 1), SplitQ31<LOGN>::TT, lds_bytes, stream>>>(data, tw, batch);
  return hipGetLastError();
}

// rocprof's spelling of a fixed-point kernel instance, for note_kernel (Q15 packs a point into one 32-bit word)
#define ARMFFT_KNAME_Q(base)                                                                                                   \
  (sizeof(typename Q::word) == 4 ? (INV ? base "<armfft::Q15, %d, true>" : base "<armfft::Q15, %d, false>")                     \
                                 : (INV ? base "<armfft::Q31, %d, true>" : base "<armfft::Q31, %d, false>"))

template <class Q, int LOGN, bool INV>
static hipError_t launch_fixed_direct(typename Q::word *data, const typename Q::tword *tw, uint32_t batch, hipStream_t stream)
{
  using PL = DPlan<Q, LOGN>;
  constexpr size_t lds_bytes = (size_t)PL::TPB * PL::IMG * sizeof(typename Q::word) + (size_t)PL::TL_ENTRIES * sizeof(typename Q::tword);
  static_assert(lds_bytes <= 160 * 1024, "direct form: the transforms of a workgroup and the quarter table in LDS");
  static std::atomic<bool> attr_done[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto kern = cfft_fixed_direct_kernel<Q, LOGN, INV>;
  if (lds_bytes > 48 * 1024 && (dev >= 16 || !attr_done[dev])) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if (dev < 16) attr_done[dev] = true;
  }
  const uint32_t groups = (batch + PL::TPB - 1) / PL::TPB;
  const uint32_t gx = groups < 65536u ? groups : 65536u, gy = (groups + gx - 1) / gx;
  note_kernel(ARMFFT_KNAME_Q("cfft_fixed_direct_kernel"), false, LOGN);
  kern<<<dim3(gx, gy, 1), PL::WG, lds_bytes, stream>>>(data, tw, batch);
  return hipGetLastError();
}

// ---- twiddle generation on the device (checked against the host table by the caller) ----
template <typename T>
__global__ void gen_twiddle_fixed_kernel(T *tw, uint32_t n, int rounding)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * n / 4) return;
  double s, c;
  if (rounding == 2) {
    // the fork's own generator (CMSIS/CMSIS.cpp:445-458 twiddleCoefCalculator + :309-327 to_fixed): float angle
    // (i*2)*PIf / (float)N, cos/sin of that float stored to float, FLOAT product with 2^dyad, round half away, clamps,
    // negatives one lower (b = ~|b|).  n is a power of two, so the float division is an exact scaling.
    const float num = __fmul_rn((float)(int)(i * 2u), 3.14159265358979f);
    const float arg = __fdiv_rn(num, (float)n);
    const float cf = (float)cos((double)arg), sf = (float)sin((double)arg);
    const float scale_f = sizeof(T) == 2 ? 32768.0f : 2147483648.0f;
    const long long hi = sizeof(T) == 2 ? 32767ll : 2147483647ll, lo = sizeof(T) == 2 ? -32768ll : -2147483648ll;
    long long b[2] = {llround((double)__fmul_rn(cf, scale_f)), llround((double)__fmul_rn(sf, scale_f))};
#pragma unroll
    for (int k = 0; k < 2; k++) {
      long long v = b[k];
      if (v > 0 && v > hi) v = hi;
      else if (v < 0 && v <= lo) v = lo;
      else if (v < 0) v = ~(-v);
      tw[2 * i + k] = (T)v;
    }
    return;
  }
  if ((4u * i) % n == 0) {
    const uint32_t q = (4u * i / n) & 3u;
    c = (q == 0) ? 1.0 : ((q == 2) ? -1.0 : 0.0);
    s = (q == 1) ? 1.0 : ((q == 3) ? -1.0 : 0.0);
  } else {
    sincospi(2.0 * (double)i / (double)n, &s, &c);
  }
  const double scale = sizeof(T) == 2 ? 32768.0 : 2147483648.0;
  const double hi = sizeof(T) == 2 ? 32767.0 : 2147483647.0, lo = -scale;
  double vc = c * scale, vs = s * scale;
  vc = rounding ? floor(vc + 0.5) : floor(vc);
  vs = rounding ? floor(vs + 0.5) : floor(vs);
  vc = fmin(fmax(vc, lo), hi);
  vs = fmin(fmax(vs, lo), hi);
  tw[2 * i] = (T)(long long)vc;
  tw[2 * i + 1] = (T)(long long)vs;
}

hipError_t launch_gen_twiddle_q15(int16_t *tw, uint32_t n, int rounding, hipStream_t stream)
{
  gen_twiddle_fixed_kernel<int16_t><<<(3 * n / 4 + 255) / 256, 256, 0, stream>>>(tw, n, rounding);
  return hipGetLastError();
}
hipError_t launch_gen_twiddle_q31(int32_t *tw, uint32_t n, int rounding, hipStream_t stream)
{
  gen_twiddle_fixed_kernel<int32_t><<<(3 * n / 4 + 255) / 256, 256, 0, stream>>>(tw, n, rounding);
  return hipGetLastError();
}

// ---- launchers ----
template <class Q, int LOGN, bool INV>
static hipError_t launch_fixed_one(typename Q::word *data, const typename Q::tword *tw, uint32_t batch, bool bitrev, hipStream_t stream)
{
  using PL = FixedPlan<LOGN>;
  constexpr size_t lds_bytes = (size_t)PL::LDS_WORDS * sizeof(typename Q::word) +
                               (FixedTune<Q, LOGN>::TW4 ? (size_t)(3 * PL::N / 16) * sizeof(typename Q::tword) : 0);
  static std::atomic<bool> attr_done[16];  // zero-initialised; a racing second hipFuncSetAttribute is harmless
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto kern = cfft_fixed_kernel<Q, LOGN, INV>;
  if (lds_bytes > 48 * 1024 && (dev < 0 || dev >= 16 || !attr_done[dev])) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  const uint32_t ngroups = (batch + PL::TPB - 1) / PL::TPB;
  const uint32_t gx = ngroups < 65536u ? ngroups : 65536u, gy = (ngroups + gx - 1) / gx;
  note_kernel(ARMFFT_KNAME_Q("cfft_fixed_kernel"), false, LOGN);
  kern<<<dim3(gx, gy, 1), PL::T, lds_bytes, stream>>>(data, tw, batch, ngroups, bitrev ? 1 : 0);
  return hipGetLastError();
}

template <class Q>
static hipError_t launch_fixed(int logn, typename Q::word *data, const typename Q::tword *tw, uint32_t batch, bool inverse, bool bitrev,
                               hipStream_t stream)
{
  if (batch == 0) return hipSuccess;
  if (bitrev) {
    if constexpr (sizeof(typename Q::word) == 8) {
      // q31, 4096 ... 16384 points: the half-footprint exchange (same-box A/B against the direct form above: 4096 0.654 ->
      // 0.660 of the roofline, 8192 0.625 -> 0.665, 16384 0.421 -> 0.526)
      if (logn >= 12 && logn <= 14) {
        int2 *d = reinterpret_cast<int2 *>(data);
        const int2 *t = reinterpret_cast<const int2 *>(tw);
        if (logn == 12) return inverse ? launch_q31_split<12, true>(d, t, batch, stream) : launch_q31_split<12, false>(d, t, batch, stream);
        if (logn == 13) return inverse ? launch_q31_split<13, true>(d, t, batch, stream) : launch_q31_split<13, false>(d, t, batch, stream);
        return inverse ? launch_q31_split<14, true>(d, t, batch, stream) : launch_q31_split<14, false>(d, t, batch, stream);
      }
    }
    switch (logn) {
#define D(L) \
  case L: return inverse ? launch_fixed_direct<Q, L, true>(data, tw, batch, stream) : launch_fixed_direct<Q, L, false>(data, tw, batch, stream);
      D(9) D(10) D(11) D(12) D(13) D(14)
#undef D
      case 8:  // 16 lanes per transform: 64-byte runs for q15 (0.61 against 0.72 of the roofline), 128-byte runs for q31 (0.73 against 0.63)
        if (sizeof(typename Q::word) == 8)
          return inverse ? launch_fixed_direct<Q, 8, true>(data, tw, batch, stream) : launch_fixed_direct<Q, 8, false>(data, tw, batch, stream);
        break;
    }
  }
  switch (logn) {
#define C(L) \
  case L: return inverse ? launch_fixed_one<Q, L, true>(data, tw, batch, bitrev, stream) : launch_fixed_one<Q, L, false>(data, tw, batch, bitrev, stream);
    C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14)
#undef C
  }
  return hipErrorInvalidValue;
}

hipError_t launch_cfft_q15(int logn, int16_t *data, const int16_t *tw, uint32_t batch, bool inverse, bool bitrev, hipStream_t stream)
{
  return launch_fixed<Q15>(logn, reinterpret_cast<uint32_t *>(data), reinterpret_cast<const uint32_t *>(tw), batch, inverse, bitrev, stream);
}

hipError_t launch_cfft_q31(int logn, int32_t *data, const int32_t *tw, uint32_t batch, bool inverse, bool bitrev, hipStream_t stream)
{
  return launch_fixed<Q31>(logn, reinterpret_cast<int2 *>(data), reinterpret_cast<const int2 *>(tw), batch, inverse, bitrev, stream);
}

// ------------------------------------------------------------------------------------------
// fixed-point REAL FFT: arm_rfft_q15 (arm_rfft_q15.c:148-182) / arm_rfft_q31 (arm_rfft_q31.c, the
// non-Neon body), generic-C split functions arm_split_rfft_* / arm_split_rifft_* restated bit for
// bit and fused around the in-LDS complex transform of L = n_real/2 points:
//   forward:  load L complex -> cfft -> split from LDS -> 2L complex words out (spectrum + mirror)
//   inverse:  split_rifft while loading (reads L+1 complex) -> inverse cfft -> saturating << 1 on store
// A/B = realCoefA/B tables (8192 scalars, 4096 pairs), mod = twidCoefRModifier.
// ------------------------------------------------------------------------------------------
// x * y + c on 64 bits (wrapping) as ONE v_mad_i64_i32.  Written out: with a coefficient that takes part in four products
// the compiler keeps its sign extension in a register pair and multiplies 64 x 64 bits -- three quarter-rate instructions
// and two additions per product instead of one (measured in the split of 512 real q31 points: 85 v_mul_lo_u32 and
// 40 v_mad_u64_u32 next to the 88 v_mad_i64_i32 it did find).
__device__ __forceinline__ long long mad64(int32_t x, int32_t y, long long c)
{
  long long r;
  unsigned long long carry;
  asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=&v"(r), "=s"(carry) : "v"(x), "v"(y), "v"(c));
  return r;
}
__device__ __forceinline__ long long hi_half(int32_t a) { return (long long)(((unsigned long long)(uint32_t)a << 32) | 0x80000000ULL); }  // a.5 in 32.32
__device__ __forceinline__ int32_t mult_r(int32_t x, int32_t y) { return (int32_t)(mad64(x, y, 0x80000000LL) >> 32); }
__device__ __forceinline__ int32_t macc_r(int32_t a, int32_t x, int32_t y) { return (int32_t)(mad64(x, y, hi_half(a)) >> 32); }
__device__ __forceinline__ int32_t msub_r(int32_t a, int32_t x, int32_t y)
{
  return (int32_t)((long long)((unsigned long long)hi_half(a) - (unsigned long long)mad64(x, y, 0)) >> 32);
}

// forward split of bin i (1 <= i < L) from s1 = X[i], s2 = X[L-i]; returns the word for position i;
// the mirrored position 2L - i holds (re, -im)
// (q15: the four products of a sum are two v_dot2_i32_i16; a product that is SUBTRACTED uses the complement of its
// coefficient, x * ~y = -x*y - x, and x goes in through the accumulator: everything is arithmetic modulo 2^32 like the
// reference's 32-bit sums, and ~y exists for every y, which -y does not)
__device__ __forceinline__ Q15::cpx split_fwd(Q15::cpx s1, Q15::cpx s2, Q15::cpx a, Q15::cpx b)
{
  const s16x2 v1 = to_v(s1), v2 = to_v(s2), va = to_v(a), vb = to_v(b);
  const s16x2 a_n = va ^ s16x2{0, -1};                 // {a.re, ~a.im}
  const s16x2 b_n = swap_ri(vb ^ s16x2{-1, 0});        // {b.im, ~b.re}
  // outR = s1.re a.re - s1.im a.im + s2.re b.re + s2.im b.im
  int32_t outR = __builtin_amdgcn_sdot2(v2, vb, (int)s1.im, false);
  outR = __builtin_amdgcn_sdot2(v1, a_n, outR, false);
  // outI = s2.re b.im - s2.im b.re + s1.im a.re + s1.re a.im
  int32_t outI = __builtin_amdgcn_sdot2(v1, swap_ri(va), (int)s2.im, false);
  outI = __builtin_amdgcn_sdot2(v2, b_n, outI, false);
  return Q15::cpx{(int16_t)(outR >> 16), (int16_t)(outI >> 16)};
}
__device__ __forceinline__ Q31::cpx split_fwd(Q31::cpx s1, Q31::cpx s2, Q31::cpx a, Q31::cpx b)
{
  const int32_t A1 = a.re, A2 = a.im, B1 = b.re;
  int32_t outR = mult_r(s1.re, A1);
  int32_t outI = mult_r(s1.re, A2);
  outR = msub_r(outR, s1.im, A2);
  outI = macc_r(outI, s1.im, A1);
  outR = msub_r(outR, s2.im, A2);
  outI = msub_r(outI, s2.im, B1);
  outR = macc_r(outR, s2.re, B1);
  outI = msub_r(outI, s2.re, A2);
  return Q31::cpx{outR, outI};
}
__device__ __forceinline__ Q15::cpx mirror(Q15::cpx v) { return Q15::cpx{v.re, (int16_t)(-(int)v.im)}; }
__device__ __forceinline__ Q31::cpx mirror(Q31::cpx v) { return Q31::cpx{v.re, (int32_t)(0u - (uint32_t)v.im)}; }
// DC / Nyquist words: ((re + im) >> 1, 0) and ((re - im) >> 1, 0)
__device__ __forceinline__ Q15::cpx dc_word(Q15::cpx x, bool nyq)
{
  const int v = nyq ? ((int)x.re - (int)x.im) : ((int)x.re + (int)x.im);
  return Q15::cpx{(int16_t)(v >> 1), 0};
}
__device__ __forceinline__ Q31::cpx dc_word(Q31::cpx x, bool nyq)
{
  const int32_t v = nyq ? wsub(x.re, x.im) : wadd(x.re, x.im);
  return Q31::cpx{v >> 1, 0};
}
// inverse split of point i (0 <= i < L) from s1 = X[i], s2 = X[L-i]
__device__ __forceinline__ Q15::cpx split_inv(Q15::cpx s1, Q15::cpx s2, Q15::cpx a, Q15::cpx b)
{
  const s16x2 v1 = to_v(s1), v2 = to_v(s2), va = to_v(a), vb = to_v(b);
  // outR = s2.re b.re - s2.im b.im + s1.re a.re + s1.im a.im
  int32_t outR = __builtin_amdgcn_sdot2(v2, vb ^ s16x2{0, -1}, (int)s2.im, false);
  outR = __builtin_amdgcn_sdot2(v1, va, outR, false);
  // outI = s1.im a.re - s1.re a.im - s2.re b.im - s2.im b.re
  int32_t outI = __builtin_amdgcn_sdot2(v1, swap_ri(va) ^ s16x2{-1, 0}, (int)s1.re + (int)s2.re + (int)s2.im, false);   // {~a.im, a.re}
  outI = __builtin_amdgcn_sdot2(v2, swap_ri(vb) ^ s16x2{-1, -1}, outI, false);                                 // {~b.im, ~b.re}
  return Q15::cpx{(int16_t)(outR >> 16), (int16_t)(outI >> 16)};
}
__device__ __forceinline__ Q31::cpx split_inv(Q31::cpx s1, Q31::cpx s2, Q31::cpx a, Q31::cpx b)
{
  const int32_t A1 = a.re, A2 = a.im, B1 = b.re;
  int32_t outR = mult_r(s1.re, A1);
  int32_t outI = mult_r(s1.re, (int32_t)(0u - (uint32_t)A2));
  outR = macc_r(outR, s1.im, A2);
  outI = macc_r(outI, s1.im, A1);
  outR = macc_r(outR, s2.im, A2);
  outI = msub_r(outI, s2.im, B1);
  outR = macc_r(outR, s2.re, B1);
  outI = macc_r(outI, s2.re, A2);
  return Q31::cpx{outR, outI};
}
// arm_shift_q15 / arm_shift_q31 by +1: saturating
__device__ __forceinline__ Q15::cpx sat_shl1(Q15::cpx v) { return Q15::cpx{(int16_t)ssat16((int)v.re << 1), (int16_t)ssat16((int)v.im << 1)}; }
__device__ __forceinline__ int32_t sat_dbl(int32_t x)
{
  const long long v = (long long)x * 2;
  return v > 2147483647LL ? 2147483647 : (v < -2147483648LL ? (int32_t)0x80000000 : (int32_t)v);
}
__device__ __forceinline__ Q31::cpx sat_shl1(Q31::cpx v) { return Q31::cpx{sat_dbl(v.re), sat_dbl(v.im)}; }

// LOGN = log2(L).  Forward: in = L words per transform, out = 2L words.  Inverse: in = L + 1 words per
// transform (stride L + 1), out = L words.
#ifndef ARMFFT_RFFT_MINW
#define ARMFFT_RFFT_MINW 4
#endif
template <class Q, int LOGN, bool INV>
__global__ __launch_bounds__(FixedPlan<LOGN>::T, ARMFFT_RFFT_MINW) void rfft_fixed_kernel(const typename Q::word *__restrict__ in,
                                                                          typename Q::word *__restrict__ out,
                                                                          const typename Q::tword *__restrict__ tw,
                                                                          const typename Q::tword *__restrict__ coefA,
                                                                          const typename Q::tword *__restrict__ coefB, int mod,
                                                                          uint32_t batch, uint32_t ngroups, int bitrev)
{
  using PL = FixedPlan<LOGN>;
  using word = typename Q::word;
  using cpx = typename Q::cpx;
  constexpr int L = PL::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  word *lds = reinterpret_cast<word *>(lds_raw);
  const int tid = threadIdx.x;
  const uint32_t g = blockIdx.x + blockIdx.y * gridDim.x;
  if (g >= ngroups) return;
  const uint32_t tr0 = g * PL::TPB;
  const uint32_t nvt = (batch - tr0) >= (uint32_t)PL::TPB ? (uint32_t)PL::TPB : (batch - tr0);  // valid transforms here
  typename Q::tword *tl4 = reinterpret_cast<typename Q::tword *>(lds + PL::LDS_WORDS);
  if constexpr (FixedTune<Q, LOGN>::TW4) {
    for (int i = tid; i < 3 * PL::N / 16; i += PL::T) tl4[i] = tw[4 * i];
  }

  // forward: the split's coefficients depend only on the thread (bins i0 .. i0 + SPER - 1 of step `it` of the store
  // loop below): fetched here, behind the data loads
  constexpr int SPER = (sizeof(word) == 8 || LOGN <= 9) ? 16 / sizeof(word) : 1, SITER = PL::ELEMS / SPER / PL::T;
  static_assert(SITER * SPER == 16 && SITER * SPER * PL::T == PL::ELEMS, "16 bins per thread");
  // (step `it` covers bins ((tid + it T) SPER mod L) + k: T SPER bins per step, so the steps share L / (T SPER) sets)
  constexpr int NSETS = (L > PL::T * SPER) ? L / (PL::T * SPER) : 1, NC = NSETS * SPER;
  static_assert(NC <= 16, "coefficient sets");
  typename Q::tword ca[INV ? 1 : NC], cb[INV ? 1 : NC];
  // ---- load ----
  if (!INV) {
    const word *__restrict__ src = in + (size_t)tr0 * L;
    constexpr int PER16 = 16 / sizeof(word);  // points per 16-byte access
    for (int v = tid; v < PL::ELEMS / PER16; v += PL::T) {
      const int e = v * PER16;
      uint4 raw = make_uint4(0u, 0u, 0u, 0u);
      if ((uint32_t)(e >> LOGN) < nvt) raw = ld16_stream(src + e);
      const word *pw = reinterpret_cast<const word *>(&raw);
#pragma unroll
      for (int k = 0; k < PER16; k++) lds[fpad(e + k)] = pw[k];
    }
#pragma unroll
    for (int set = 0; set < NSETS; set++) {
#pragma unroll
      for (int k = 0; k < SPER; k++) {
        const int i = (((tid + set * PL::T) * SPER) & (L - 1)) + k;
        ca[INV ? 0 : set * SPER + k] = coefA[i * mod];
        cb[INV ? 0 : set * SPER + k] = coefB[i * mod];
      }
    }
  } else {
    for (int e = tid; e < PL::ELEMS; e += PL::T) {
      const int t = e >> LOGN, i = e & (L - 1);
      word w{};
      if ((uint32_t)t < nvt) {
        const word *__restrict__ src = in + (size_t)(tr0 + t) * (L + 1);
        const cpx s1 = Q::unpack(src[i]), s2 = Q::unpack(src[L - i]);
        w = Q::pack(split_inv(s1, s2, Q::unpack(coefA[i * mod]), Q::unpack(coefB[i * mod])));
      }
      lds[fpad(e)] = w;
    }
  }
  __syncthreads();

  // ---- complex transform of L points, exactly as cfft_fixed_kernel ----
  if constexpr (PL::BY2) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int u = tid + k * PL::T;
      const int t = u >> (LOGN - 1), i = u & (PL::L - 1);
      const int ea = t * PL::N + i, eb = ea + PL::L;
      cpx a = Q::unpack(lds[fpad(ea)]), b = Q::unpack(lds[fpad(eb)]);
      by2<Q, INV>(a, b, Q::unpack(tw[i]));
      lds[fpad(ea)] = Q::pack(a);
      lds[fpad(eb)] = Q::pack(b);
    }
    __syncthreads();
  }
  {
    const int sub = tid >> (PL::LOGL - 4), r = tid & ((PL::L >> 4) - 1);
    if constexpr (FixedTune<Q, LOGN>::TW4) {
      // quarter table in LDS as in cfft_fixed_kernel; written before the barrier that follows the load phase
      run_stages<Q, PL::LOGL, 0, INV, PL::BY2 ? 2 : 1>(lds, tw, sub * PL::L, r, (const typename Q::tword *)tl4);
    } else {
      run_stages<Q, PL::LOGL, 0, INV, PL::BY2 ? 2 : 1>(lds, tw, sub * PL::L, r);
    }
  }
  // bin i of transform t as arm_cfft_* leaves it (bit reversal and the radix4by2 final << 1 folded in)
  auto X = [&](int t, int i) -> cpx {
    const int src = bitrev ? (int)(__brev((unsigned)i) >> (32 - LOGN)) : i;
    cpx c = Q::unpack(lds[fpad(t * L + src)]);
    if (PL::BY2) c = shl1<Q>(c);
    return c;
  };

  // ---- store ----
  if (!INV) {
    word *__restrict__ dst = out + (size_t)tr0 * (2 * L);
    // A thread splits SPER consecutive bins i0 .. i0+SPER-1 (16 bytes' worth) and stores them at position i0.  Their
    // conjugates belong at 2L-i0-SPER+1 .. 2L-i0, one word off 16-byte alignment, so the thread stores the ALIGNED
    // chunk [2L-i0-SPER, 2L-i0) instead: its own bins i0+1 .. i0+SPER-1 and bin i0+SPER, the first bin of the next lane
    // (wave shift; lane 63 takes lane 0's first bin of the next step; at the end of a transform it is the Nyquist word).
    // Stored straight from the lanes that split them, the mirrored words of a store instruction lay 16 bytes apart and
    // every 128-byte line of that half was written by partial stores of several instructions.
    auto from_next_lane = [&](word w, word lane0_next) -> word {
      constexpr int NW32 = sizeof(word) / 4;
      int *pw = reinterpret_cast<int *>(&w), *pn = reinterpret_cast<int *>(&lane0_next);
      word r;
      int *pr = reinterpret_cast<int *>(&r);
#pragma unroll
      for (int h = 0; h < NW32; h++) {
        const int sh = __builtin_amdgcn_update_dpp(0, pw[h], 0x130, 0xF, 0xF, false);   // wave_shl:1 -- lane l reads lane l+1
        const int first = __builtin_amdgcn_readfirstlane(pn[h]);   // (outside the select: every lane takes part)
        pr[h] = (tid & 63) == 63 ? first : sh;
      }
      return r;
    };
    // the SPER bins of step `it` into w[]
    auto split_step = [&](int it, word *w) {
      const int e = (tid + it * PL::T) * SPER, t = e >> LOGN, i0 = e & (L - 1);
#pragma unroll
      for (int k = 0; k < SPER; k++) {
        const int i = i0 + k, ci = INV ? 0 : (it % NSETS) * SPER + k;
        const cpx xi = X(t, i);
        const cpx o = split_fwd(xi, X(t, (L - i) & (L - 1)), Q::unpack(ca[ci]), Q::unpack(cb[ci]));
        w[k] = Q::pack(i == 0 ? xi : o);                        // bin 0: the raw X[0], for the DC and Nyquist words
      }
    };
    // step `it`: its bins w[] at position i0, the aligned mirrored chunk with `nxt` = bin i0 + SPER
    auto store_step = [&](int it, const word *w, word nxt) {
      const int e = (tid + it * PL::T) * SPER, t = e >> LOGN, i0 = e & (L - 1);
      if ((uint32_t)t >= nvt) return;
      word *__restrict__ d = dst + (size_t)t * (2 * L);
      cpx nx = mirror(Q::unpack(nxt));
      if (i0 + SPER == L) nx = dc_word(X(t, 0), true);
      uint4 raw, rawm;
      word *pw = reinterpret_cast<word *>(&raw), *pm = reinterpret_cast<word *>(&rawm);
      pm[0] = Q::pack(nx);
#pragma unroll
      for (int k = 0; k < SPER; k++) {
        const cpx c = Q::unpack(w[k]);
        pw[k] = (i0 + k == 0) ? Q::pack(dc_word(c, false)) : w[k];
        if (k > 0) pm[SPER - k] = Q::pack(mirror(c));
      }
      if constexpr (PL::T != 64) {
        // several waves (2048 complex points and up, natural-order output only -- the bit-reversed order takes the direct
        // form there): the next lane of lane 63 is another wave's; every lane stores its own conjugates
        if (i0 == 0) d[L] = Q::pack(dc_word(X(t, 0), true));
#pragma unroll
        for (int k = 0; k < SPER; k++)
          if (i0 + k != 0) d[2 * L - i0 - k] = Q::pack(mirror(Q::unpack(w[k])));
        if constexpr (SPER > 1) st16_stream(d + i0, raw);
        else d[i0] = pw[0];
      } else if constexpr (SPER > 1) {
        st16_stream(d + i0, raw);
        st16_stream(d + 2 * L - i0 - SPER, rawm);
      } else {
        d[i0] = pw[0];
        d[2 * L - i0 - 1] = pm[0];
      }
    };
    if constexpr (sizeof(word) == 4) {
      // q15: every step split first (16 registers), then the stores
      word ow[16];
#pragma unroll
      for (int it = 0; it < SITER; it++) split_step(it, ow + it * SPER);
#pragma unroll
      for (int it = 0; it < SITER; it++)
        store_step(it, ow + it * SPER, from_next_lane(ow[it * SPER], ow[(it + 1 < SITER ? it + 1 : it) * SPER]));
    } else {
      // q31 (the same in 32 registers spills): step `it` is split, then step it-1, which waited for this step's first bin, is stored
      word cur[SPER], prv[SPER];
#pragma unroll
      for (int it = 0; it <= SITER; it++) {
        if (it < SITER) split_step(it, cur);
        if (it > 0) store_step(it - 1, prv, from_next_lane(prv[0], it < SITER ? cur[0] : prv[0]));
#pragma unroll
        for (int k = 0; k < SPER; k++) prv[k] = cur[k];
      }
    }
  } else {
    word *__restrict__ dst = out + (size_t)tr0 * L;
    for (int e = tid; e < PL::ELEMS; e += PL::T) {
      const int t = e >> LOGN, i = e & (L - 1);
      if ((uint32_t)t < nvt) dst[e] = Q::pack(sat_shl1(X(t, i)));
    }
  }
}

// Direct form of the fixed-point real FFT (L = 512 ... 4096 complex points, bit-reversed cfft order = the
// reference's configuration): the complex transform is direct_transform of the complex kernel.
//   forward: L input words straight into the first step; the results go back into the LDS image in natural order (one
//            contiguous run per result index c across the lanes), and bin i = tid + c*T is split from X[i] and X[L-i]
//            -- contiguous and reverse-contiguous LDS reads -- and written twice (position i and, conjugated, 2L-i);
//   inverse: split_rifft of points i and L-i while loading (the mirrored one a descending contiguous run), transform,
//            saturating << 1 on the direct stores.
template <class Q, int LOGN, bool INV>
__global__ __launch_bounds__(DirectPlan<LOGN>::WG) void rfft_fixed_direct_kernel(const typename Q::word *__restrict__ in,
                                                                                typename Q::word *__restrict__ out,
                                                                                const typename Q::tword *__restrict__ tw,
                                                                                const typename Q::tword *__restrict__ coefA,
                                                                                const typename Q::tword *__restrict__ coefB, int mod,
                                                                                uint32_t batch)
{
  using PL = DPlan<Q, LOGN>;
  using word = typename Q::word;
  using tword = typename Q::tword;
  using cpx = typename Q::cpx;
  constexpr int W = sizeof(word), T = PL::T, L = PL::N, TPB = PL::TPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t g = blockIdx.x + blockIdx.y * gridDim.x;   // group of TPB transforms
  const uint32_t t0 = g * TPB;
  if (t0 >= batch) return;
  const int wtid = threadIdx.x, tid = wtid & (T - 1), tr = TPB == 1 ? 0 : wtid / T;
  word *lds = reinterpret_cast<word *>(lds_raw) + tr * PL::IMG;
  tword *tl = reinterpret_cast<tword *>(reinterpret_cast<word *>(lds_raw) + TPB * PL::IMG);
  const uint32_t nvalid = batch - t0 < (uint32_t)TPB ? batch - t0 : (uint32_t)TPB;
  cpx x[16];
  if (!INV) {
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<word *>(in) + (size_t)t0 * L, 0, (int)(nvalid * L * W), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)t0 * (2 * L), 0, (int)(nvalid * 2 * L * W), 0x00020000);
    const int vin = (tr * L + tid) * W, vout = tr * 2 * L * W;
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = Q::unpack(DirectIo<Q>::load(ri, vin, c * T * W));
    // the split's coefficients depend only on the thread: fetched now, behind the data loads, instead of after the
    // transform (same-box A/B: q15 1024 real points 0.42 -> 0.50 of the roofline, 4096 0.53 -> 0.57, 8192 0.51 -> 0.565;
    // q31 0.545 -> 0.58, 0.56 -> 0.585, 0.54 -> 0.565)
    tword ca[16], cb[16];
#pragma unroll
    for (int c = 0; c < 16; c++) {
      ca[c] = coefA[(tid + c * T) * mod];
      cb[c] = coefB[(tid + c * T) * mod];
    }
    direct_transform<Q, LOGN, false>(x, lds, tl, tw, tid, wtid);
    __syncthreads();  // every thread has read its last-step block: the image may be overwritten
#pragma unroll
    for (int c = 0; c < 16; c++) lds[direct_rc(c) * T + tid] = Q::pack(direct_out<Q, LOGN>(x[c]));
    __syncthreads();
    // The mirrored half as whole, aligned runs of T words: bins cT+1 .. cT+T-1 (lanes 1 .. T-1) land in
    // [2L-(c+1)T+1, 2L-cT), and the run's first word, position 2L-(c+1)T, is the mirror of bin (c+1)T -- lane 0's NEXT
    // result (the Nyquist word after the last one).  Written per c with the lanes' own bins, the runs would start one word
    // off the line and every line would be written by two instructions.
    const cpx x0 = Q::unpack(lds[0]);
    const int vmir = vout + (2 * L - 15 * T - (tid ? tid : T)) * W;   // mirror run m at + (15 - m) T words
    word prev{};
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const int i = tid + c * T;
      const cpx o = split_fwd(Q::unpack(lds[i]), Q::unpack(lds[(L - i) & (L - 1)]), Q::unpack(ca[c]), Q::unpack(cb[c]));
      const word ow = Q::pack(i == 0 ? dc_word(x0, false) : o), om = Q::pack(mirror(o));
      DirectIo<Q>::store(ow, ro, vout + tid * W, c * T * W);
      if (c > 0) DirectIo<Q>::store(tid ? prev : om, ro, vmir, (16 - c) * T * W);
      prev = om;
    }
    DirectIo<Q>::store(tid ? prev : Q::pack(dc_word(x0, true)), ro, vmir, 0);
  } else {
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<word *>(in) + (size_t)t0 * (L + 1), 0, (int)(nvalid * (L + 1) * W), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)t0 * L, 0, (int)(nvalid * L * W), 0x00020000);
    const int vin = tr * (L + 1) * W, vout = (tr * L + tid) * W;
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const int i = tid + c * T;
      const cpx s1 = Q::unpack(DirectIo<Q>::load(ri, vin + tid * W, c * T * W)), s2 = Q::unpack(DirectIo<Q>::load(ri, vin + (L - i) * W, 0));
      x[c] = split_inv(s1, s2, Q::unpack(coefA[i * mod]), Q::unpack(coefB[i * mod]));
    }
    direct_transform<Q, LOGN, true>(x, lds, tl, tw, tid, wtid);
#pragma unroll
    for (int c = 0; c < 16; c++) DirectIo<Q>::store(Q::pack(sat_shl1(direct_out<Q, LOGN>(x[c]))), ro, vout, direct_rc(c) * T * W);
  }
}

// Lane form of the fixed-point real FFT for 32 and 64 real points (L = 16 / 32 complex, bit-reversed cfft order = the
// reference's configuration): ONE LANE PER TRANSFORM.  All stages, the bit reversal and the split run on registers with
// compile-time indices; the twiddles and the split's coefficients have the same index in every lane, so they are scalar
// loads.  LDS only turns the rows of a wave's 64 transforms into lane-contiguous 16-byte global accesses (a row is padded
// by 16 bytes: the 16-byte row reads and writes of the lanes are conflict-free).  The LDS form above spends its
// instructions on per-element index arithmetic at these sizes (1221 vector instructions per 16 points at 32 real q15
// points); this one runs the butterflies and little else.
template <class Q, int LOGN, bool INV>
__global__ __launch_bounds__(64) void rfft_fixed_lane_kernel(const typename Q::word *__restrict__ in, typename Q::word *__restrict__ out,
                                                             const typename Q::tword *__restrict__ tw,
                                                             const typename Q::tword *__restrict__ coefA,
                                                             const typename Q::tword *__restrict__ coefB, int mod, uint32_t batch)
{
  static_assert(LOGN == 4 || LOGN == 5, "16 or 32 complex points");
  using word = typename Q::word;
  using cpx = typename Q::cpx;
  constexpr int L = 1 << LOGN, W = sizeof(word), PER16 = 16 / W, CPR = L / PER16;   // CPR: 16-byte chunks per row of L words
  constexpr int ROWB = L * W + 16;
  constexpr bool BY2 = (LOGN & 1) != 0;
  constexpr bool WHOLE = !INV && L * W < 128;   // forward output staged as whole rows of 2L words
  __shared__ __attribute__((aligned(16))) unsigned char stage[64 * (WHOLE ? 2 * L * W + 16 : ROWB)];
  const int lane = threadIdx.x;
  const uint32_t t0 = (blockIdx.x + blockIdx.y * gridDim.x) * 64u;
  if (t0 >= batch) return;
  const int nvalid = batch - t0 < 64u ? (int)(batch - t0) : 64;
  // rows of L words between global memory (chunk v = lane + 64k of the wave's 64 rows) and the lanes' registers
  auto rows_in = [&](const word *__restrict__ src, cpx *x) {
#pragma unroll
    for (int k = 0; k < CPR; k++) {
      const int v = lane + 64 * k, row = v / CPR, col = v % CPR;
      uint4 raw = make_uint4(0u, 0u, 0u, 0u);
      if (row < nvalid) raw = ld16_stream<false>(reinterpret_cast<const uint4 *>(src) + v);
      *reinterpret_cast<uint4 *>(stage + row * ROWB + col * 16) = raw;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CPR; k++) {
      const uint4 raw = *reinterpret_cast<const uint4 *>(stage + lane * ROWB + k * 16);
      const word *pw = reinterpret_cast<const word *>(&raw);
#pragma unroll
      for (int q = 0; q < PER16; q++) x[k * PER16 + q] = Q::unpack(pw[q]);
    }
    __syncthreads();
  };
  // RW words o[] of every lane to row (lane * rows_per_transform + h) of dst, rows of RW words
  auto rows_out = [&](word *__restrict__ dst, const word *o, auto rw, int rows_per_transform, int h) {
    constexpr int RW = decltype(rw)::value, CP = RW / PER16, RB = RW * W + 16;
#pragma unroll
    for (int k = 0; k < CP; k++) {
      uint4 raw;
      word *pw = reinterpret_cast<word *>(&raw);
#pragma unroll
      for (int q = 0; q < PER16; q++) pw[q] = o[k * PER16 + q];
      *reinterpret_cast<uint4 *>(stage + lane * RB + k * 16) = raw;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CP; k++) {
      const int v = lane + 64 * k, row = v / CP, col = v % CP;
      if (row < nvalid)
        st16_stream<false>(reinterpret_cast<uint4 *>(dst) + (size_t)(row * rows_per_transform + h) * CP + col, *reinterpret_cast<const uint4 *>(stage + row * RB + col * 16));
    }
    __syncthreads();
  };
  // the complex transform of arm_cfft_*: radix-2 pre-stage at 32 points, then stages 0 and 1 of the 16-point transforms
  auto tws = [&](int ic) { return tw[ic]; };
  auto transform = [&](cpx *x) {
    if constexpr (BY2) {
#pragma unroll
      for (int i = 0; i < 16; i++) by2<Q, INV>(x[i], x[i + 16], Q::unpack(tw[i]));
      pair_butterflies<Q, 4, 0, INV, 2>(x, 0, tws, tws);
      pair_butterflies<Q, 4, 0, INV, 2>(x + 16, 0, tws, tws);
    } else {
      pair_butterflies<Q, 4, 0, INV, 1>(x, 0, tws, tws);
    }
  };
  // bin i as arm_cfft_* leaves it: bit reversal over LOGN bits, the final << 1 of radix4by2
  auto bin = [&](const cpx *x, int i) -> cpx {
    const int src = (int)(__brev((unsigned)i) >> (32 - LOGN));
    return BY2 ? shl1<Q>(x[src]) : x[src];
  };

  cpx x[L];
  if (!INV) {
    rows_in(in + (size_t)t0 * L, x);
    transform(x);
    word o[2 * L];
    const cpx x0 = bin(x, 0);
    o[0] = Q::pack(dc_word(x0, false));
    o[L] = Q::pack(dc_word(x0, true));
#pragma unroll
    for (int i = 1; i < L; i++) {
      const cpx c = split_fwd(bin(x, i), bin(x, L - i), Q::unpack(coefA[i * mod]), Q::unpack(coefB[i * mod]));
      o[i] = Q::pack(c);
      o[2 * L - i] = Q::pack(mirror(c));
    }
    word *__restrict__ dst = out + (size_t)t0 * (2 * L);
    // a store instruction should cover whole 128-byte lines: the 2L words of a transform at once where L words are
    // half a line (32 real q15 points: 0.52 -> see DESIGN 3.3), else the two halves one after the other
    if constexpr (WHOLE) {
      rows_out(dst, o, std::integral_constant<int, 2 * L>(), 1, 0);
    } else {
      rows_out(dst, o, std::integral_constant<int, L>(), 2, 0);
      rows_out(dst, o + L, std::integral_constant<int, L>(), 2, 1);
    }
  } else {
    // L + 1 words per transform at a stride of L + 1 words (aligned to the word only): copied linearly, word by word;
    // a lane's row then starts at an odd word stride
    word *sw = reinterpret_cast<word *>(stage);
    const word *__restrict__ src = in + (size_t)t0 * (L + 1);
    const int nwords = nvalid * (L + 1);
#pragma unroll
    for (int k = 0; k <= L; k++) {
      const int w = lane + 64 * k;
      sw[w] = w < nwords ? src[w] : word{};
    }
    __syncthreads();
    cpx s[L + 1];
#pragma unroll
    for (int i = 0; i <= L; i++) s[i] = Q::unpack(sw[lane * (L + 1) + i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < L; i++) x[i] = split_inv(s[i], s[L - i], Q::unpack(coefA[i * mod]), Q::unpack(coefB[i * mod]));
    transform(x);
    word o[L];
#pragma unroll
    for (int i = 0; i < L; i++) o[i] = Q::pack(sat_shl1(bin(x, i)));
    rows_out(out + (size_t)t0 * L, o, std::integral_constant<int, L>(), 1, 0);
  }
}

template <class Q, int LOGN, bool INV>
static hipError_t launch_rfft_fixed_lane(const typename Q::word *in, typename Q::word *out, const typename Q::tword *tw,
                                         const typename Q::tword *A, const typename Q::tword *B, int mod, uint32_t batch, hipStream_t stream)
{
  const uint32_t groups = (batch + 63u) / 64u;
  const uint32_t gx = groups < 65536u ? groups : 65536u, gy = (groups + gx - 1) / gx;
  note_kernel(ARMFFT_KNAME_Q("rfft_fixed_lane_kernel"), false, LOGN);
  rfft_fixed_lane_kernel<Q, LOGN, INV><<<dim3(gx, gy, 1), 64, 0, stream>>>(in, out, tw, A, B, mod, batch);
  return hipGetLastError();
}

template <class Q, int LOGN, bool INV>
static hipError_t launch_rfft_fixed_direct(const typename Q::word *in, typename Q::word *out, const typename Q::tword *tw,
                                           const typename Q::tword *A, const typename Q::tword *B, int mod, uint32_t batch, hipStream_t stream)
{
  using PL = DPlan<Q, LOGN>;
  constexpr size_t lds_bytes = (size_t)PL::TPB * PL::IMG * sizeof(typename Q::word) + (size_t)PL::TL_ENTRIES * sizeof(typename Q::tword);
  static std::atomic<bool> attr_done[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto kern = rfft_fixed_direct_kernel<Q, LOGN, INV>;
  if (lds_bytes > 48 * 1024 && (dev >= 16 || !attr_done[dev])) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if (dev < 16) attr_done[dev] = true;
  }
  const uint32_t groups = (batch + PL::TPB - 1) / PL::TPB;
  const uint32_t gx = groups < 65536u ? groups : 65536u, gy = (groups + gx - 1) / gx;
  note_kernel(ARMFFT_KNAME_Q("rfft_fixed_direct_kernel"), false, LOGN);
  kern<<<dim3(gx, gy, 1), PL::WG, lds_bytes, stream>>>(in, out, tw, A, B, mod, batch);
  return hipGetLastError();
}

template <class Q, int LOGN, bool INV>
static hipError_t launch_rfft_fixed_one(const typename Q::word *in, typename Q::word *out, const typename Q::tword *tw,
                                        const typename Q::tword *A, const typename Q::tword *B, int mod, uint32_t batch, bool bitrev,
                                        hipStream_t stream)
{
  using PL = FixedPlan<LOGN>;
  constexpr size_t lds_bytes = (size_t)PL::LDS_WORDS * sizeof(typename Q::word) +
                               (FixedTune<Q, LOGN>::TW4 ? (size_t)(3 * PL::N / 16) * sizeof(typename Q::tword) : 0);
  static std::atomic<bool> attr_done[16];  // zero-initialised; a racing second hipFuncSetAttribute is harmless
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto kern = rfft_fixed_kernel<Q, LOGN, INV>;
  if (lds_bytes > 48 * 1024 && (dev < 0 || dev >= 16 || !attr_done[dev])) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) attr_done[dev] = true;
  }
  const uint32_t ngroups = (batch + PL::TPB - 1) / PL::TPB;
  const uint32_t gx = ngroups < 65536u ? ngroups : 65536u;
  note_kernel(ARMFFT_KNAME_Q("rfft_fixed_kernel"), false, LOGN);
  kern<<<dim3(gx, (ngroups + gx - 1) / gx, 1), PL::T, lds_bytes, stream>>>(in, out, tw, A, B, mod, batch, ngroups, bitrev ? 1 : 0);
  return hipGetLastError();
}

template <class Q>
static hipError_t launch_rfft_fixed(int logl, const typename Q::word *in, typename Q::word *out, const typename Q::tword *tw,
                                    const typename Q::tword *A, const typename Q::tword *B, int mod, uint32_t batch, bool inverse,
                                    bool bitrev, hipStream_t stream)
{
  if (batch == 0) return hipSuccess;
  if (bitrev) {
    switch (logl) {
#define D(L) \
  case L:    \
    return inverse ? launch_rfft_fixed_direct<Q, L, true>(in, out, tw, A, B, mod, batch, stream) \
                   : launch_rfft_fixed_direct<Q, L, false>(in, out, tw, A, B, mod, batch, stream);
#ifndef ARMFFT_NO_D9
      D(9)
#endif
      D(10) D(11) D(12)   // (256 complex points: 0.34 against 0.45 of the roofline, stays on the LDS form)
#undef D
#define D(L) \
  case L:    \
    return inverse ? launch_rfft_fixed_lane<Q, L, true>(in, out, tw, A, B, mod, batch, stream) \
                   : launch_rfft_fixed_lane<Q, L, false>(in, out, tw, A, B, mod, batch, stream);
      D(4)
    case 5:  // 64 real q31 points need 200 registers per lane in this form: 0.51 against 0.63 on the LDS form
      if (sizeof(typename Q::scalar) == 2)
        return inverse ? launch_rfft_fixed_lane<Q, 5, true>(in, out, tw, A, B, mod, batch, stream)
                       : launch_rfft_fixed_lane<Q, 5, false>(in, out, tw, A, B, mod, batch, stream);
      break;
#undef D
    }
  }
  switch (logl) {
#define C(L) \
  case L:    \
    return inverse ? launch_rfft_fixed_one<Q, L, true>(in, out, tw, A, B, mod, batch, bitrev, stream) \
                   : launch_rfft_fixed_one<Q, L, false>(in, out, tw, A, B, mod, batch, bitrev, stream);
    C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
#undef C
  }
  return hipErrorInvalidValue;
}

hipError_t launch_rfft_q15(int logn_real, const int16_t *in, int16_t *out, const int16_t *tw, const int16_t *A, const int16_t *B, int mod,
                           uint32_t batch, bool inverse, bool bitrev, hipStream_t stream)
{
  return launch_rfft_fixed<Q15>(logn_real - 1, reinterpret_cast<const uint32_t *>(in), reinterpret_cast<uint32_t *>(out),
                                reinterpret_cast<const uint32_t *>(tw), reinterpret_cast<const uint32_t *>(A),
                                reinterpret_cast<const uint32_t *>(B), mod, batch, inverse, bitrev, stream);
}

hipError_t launch_rfft_q31(int logn_real, const int32_t *in, int32_t *out, const int32_t *tw, const int32_t *A, const int32_t *B, int mod,
                           uint32_t batch, bool inverse, bool bitrev, hipStream_t stream)
{
  return launch_rfft_fixed<Q31>(logn_real - 1, reinterpret_cast<const int2 *>(in), reinterpret_cast<int2 *>(out),
                                reinterpret_cast<const int2 *>(tw), reinterpret_cast<const int2 *>(A), reinterpret_cast<const int2 *>(B),
                                mod, batch, inverse, bitrev, stream);
}

}  // namespace armfft
