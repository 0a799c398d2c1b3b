/*
 * armfft_oracle.c -- scalar-C restatement of the reference's transform hot path.
 * TEST INFRASTRUCTURE ONLY (see armfft_oracle.h).  Build: oracle/Makefile
 * (-O2 -ffp-contract=off so float results do not depend on FMA contraction).
 *
 * Reference paths are relative to the reference root (Source/TransformFunctions/...).
 */
#include "armfft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * Tables.  Include/arm_common_tables.h:63-130,151-165 declares them; the defining file
 * Source/CommonTables/arm_common_tables.c is absent from the mount, so they are regenerated.
 *   f32   : tw[2i] = (float)cos(2 pi i/N), tw[2i+1] = (float)sin(2 pi i/N), i < N
 *           (same rule as CMSIS/CMSIS.cpp:445-458; bit patterns agree with arm_mve_tables.c:48-51)
 *   rfft  : tw[2i] = (float)sin(2 pi i/N), tw[2i+1] = (float)cos(2 pi i/N), i < N/2
 *           (TW = i*exp(-2 pi i k/L), arm_rfft_fast_f32.c:362-373)
 *   q15/31: 3N/4 (cos, sin) pairs scaled by 2^15 / 2^31, +1.0 saturated to 0x7FFF / 0x7FFFFFFF.
 *           rounding = ORC_TW_FLOOR, ORC_TW_ROUND (both on the double value, exact at multiples of
 *           pi/2) or ORC_TW_FORK: the fork's own generator twiddleCoefCalculator + to_fixed
 *           (CMSIS/CMSIS.cpp:445-458, :309-327), the only table rule the mount holds in source
 *           and the one the fork's 8192/16384 tables were printed with (fork_twiddle below).
 * ------------------------------------------------------------------------------------------ */
void orc_twiddle_f32(uint32_t N, float *tw)
{
  for (uint32_t i = 0; i < N; i++) {
    double a = 2.0 * M_PI * (double)i / (double)N;
    tw[2 * i] = (float)cos(a);
    tw[2 * i + 1] = (float)sin(a);
  }
}

void orc_twiddle_rfft_f32(uint32_t N, float *tw)
{
  for (uint32_t i = 0; i < N / 2; i++) {
    double a = 2.0 * M_PI * (double)i / (double)N;
    tw[2 * i] = (float)sin(a);
    tw[2 * i + 1] = (float)cos(a);
  }
}

static int64_t fix_scale(double v, double scale, int rounding, int64_t lo, int64_t hi)
{
  double s = v * scale;
  double r = (rounding == ORC_TW_ROUND) ? floor(s + 0.5) : floor(s);
  if (r > (double)hi) r = (double)hi;
  if (r < (double)lo) r = (double)lo;
  return (int64_t)r;
}

/* exact values at multiples of pi/2 so that floor() never sees cos(pi/2) = 6e-17 or -1e-16 */
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

/* The fork's generator, restated operation by operation.
 *   twiddleCoefCalculator (CMSIS/CMSIS.cpp:452-456):
 *       float c = cos(i * 2 * PI / (float)fft_len), s = sin(same)
 *     PI is the FLOAT literal 3.14159265358979f (Include/gnss_codes.h:33, dsp/fast_math_functions.h:54),
 *     so i*2 is an int, (i*2)*PI a float product and the division a float division: the angle
 *     carries a relative error of up to ~9e-8 (PI's own float rounding included) before cos/sin
 *     ever see it.  cos/sin of that float, result stored to float (taken here as the correctly
 *     rounded float of the double function; an MSVC cosf that is not correctly rounded could
 *     differ in the last float bit -- nothing in the mount can settle that).
 *   to_fixed(f, e) (:309-327): tmp = (float)pow(2.0f, e); b = llround(f * tmp) -- a FLOAT product,
 *     so a q31 entry has 24 significant bits; clamp "b > 0 && b > max -> max", "b < 0 && b <= min
 *     -> min" (the constants 2147483647 / -2147483648 carry the comments "32767" / "-32768": the
 *     author swaps them with the dyad, and so does this restatement); then for b < 0:
 *     b = abs(b); b = ~b, i.e. -|b| - 1: every negative entry is ONE LOWER than the rounded value.
 *   No exact values at multiples of pi/2: cos(float(pi/2)) * 2^31 = -93.9 -> -94 -> -95. */
static int32_t fork_to_fixed(float f, int e, int64_t lo, int64_t hi)
{
  float tmp = (float)pow(2.0f, e);
  int64_t b = (int64_t)llround((double)(f * tmp));   /* llround(float): the float product, then round half away */
  if (b > 0 && b > hi) return (int32_t)hi;
  if (b < 0 && b <= lo) return (int32_t)lo;
  if (b < 0) {
    b = llabs(b);
    b = ~b;
  }
  return (int32_t)b;
}

static void fork_cs(uint32_t i, uint32_t N, float *c, float *s)
{
  const float PIf = 3.14159265358979f;
  volatile float num = (float)(int)(i * 2) * PIf;   /* volatile: no contraction / excess precision across the two float steps */
  volatile float arg = num / (float)N;
  *c = (float)cos((double)arg);
  *s = (float)sin((double)arg);
}

void orc_twiddle_q15(uint32_t N, int16_t *tw, int rounding)
{
  if (rounding == ORC_TW_FORK) {
    for (uint32_t i = 0; i < 3 * N / 4; i++) {
      float c, s;
      fork_cs(i, N, &c, &s);
      tw[2 * i] = (int16_t)fork_to_fixed(c, 15, -32768, 32767);
      tw[2 * i + 1] = (int16_t)fork_to_fixed(s, 15, -32768, 32767);
    }
    return;
  }
  for (uint32_t i = 0; i < 3 * N / 4; i++) {
    double c, s;
    exact_cs(i, N, &c, &s);
    tw[2 * i] = (int16_t)fix_scale(c, 32768.0, rounding, -32768, 32767);
    tw[2 * i + 1] = (int16_t)fix_scale(s, 32768.0, rounding, -32768, 32767);
  }
}

void orc_twiddle_q31(uint32_t N, int32_t *tw, int rounding)
{
  if (rounding == ORC_TW_FORK) {
    for (uint32_t i = 0; i < 3 * N / 4; i++) {
      float c, s;
      fork_cs(i, N, &c, &s);
      tw[2 * i] = fork_to_fixed(c, 31, INT32_MIN, INT32_MAX);
      tw[2 * i + 1] = fork_to_fixed(s, 31, INT32_MIN, INT32_MAX);
    }
    return;
  }
  for (uint32_t i = 0; i < 3 * N / 4; i++) {
    double c, s;
    exact_cs(i, N, &c, &s);
    tw[2 * i] = (int32_t)fix_scale(c, 2147483648.0, rounding, INT32_MIN, INT32_MAX);
    tw[2 * i + 1] = (int32_t)fix_scale(s, 2147483648.0, rounding, INT32_MIN, INT32_MAX);
  }
}

static uint32_t ilog2(uint32_t n)
{
  uint32_t l = 0;
  while ((1u << l) < n) l++;
  return l;
}

static int is_pow2(uint32_t n) { return n && !(n & (n - 1)); }

/* small per-size caches so repeated calls (batches, timing loops) do not regenerate tables */
#define ORC_MAXLOG 15
static float *g_tw_f32[ORC_MAXLOG + 1];
static float *g_tw_rfft[ORC_MAXLOG + 1];
static int16_t *g_tw_q15[3][ORC_MAXLOG + 1];
static int32_t *g_tw_q31[3][ORC_MAXLOG + 1];
static uint32_t *g_perm_f32[ORC_MAXLOG + 1];

static const float *tw_f32(uint32_t N)
{
  uint32_t l = ilog2(N);
  if (!g_tw_f32[l]) {
    float *t = (float *)malloc(sizeof(float) * 2 * N);
    orc_twiddle_f32(N, t);
    g_tw_f32[l] = t;
  }
  return g_tw_f32[l];
}
static const float *tw_rfft(uint32_t N)
{
  uint32_t l = ilog2(N);
  if (!g_tw_rfft[l]) {
    float *t = (float *)malloc(sizeof(float) * N);
    orc_twiddle_rfft_f32(N, t);
    g_tw_rfft[l] = t;
  }
  return g_tw_rfft[l];
}
static const int16_t *tw_q15(uint32_t N, int rounding)
{
  uint32_t l = ilog2(N);
  rounding = (rounding < 0 || rounding > 2) ? 0 : rounding;
  if (!g_tw_q15[rounding][l]) {
    int16_t *t = (int16_t *)malloc(sizeof(int16_t) * 3 * N / 2);
    orc_twiddle_q15(N, t, rounding);
    g_tw_q15[rounding][l] = t;
  }
  return g_tw_q15[rounding][l];
}
static const int32_t *tw_q31(uint32_t N, int rounding)
{
  uint32_t l = ilog2(N);
  rounding = (rounding < 0 || rounding > 2) ? 0 : rounding;
  if (!g_tw_q31[rounding][l]) {
    int32_t *t = (int32_t *)malloc(sizeof(int32_t) * 3 * N / 2);
    orc_twiddle_q31(N, t, rounding);
    g_tw_q31[rounding][l] = t;
  }
  return g_tw_q31[rounding][l];
}

/* ------------------------------------------------------------------------------------------
 * f32 complex FFT: arm_cfft_f32 (arm_cfft_f32.c:1243-1298)
 * ------------------------------------------------------------------------------------------ */

/* One radix-8 DIF butterfly over elements i0 + q*n2 (q = 0..7), sub-outputs left in natural
 * order at the same eight positions.  Follows the operation order of
 * arm_radix8_butterfly_f32 (arm_cfft_radix8_f32.c:87-138 twiddle-free column, :188-281
 * twiddled columns): co/si hold cos/sin of 1..7 times the column angle, applied as a
 * multiplication by (cos - j sin). */
static void r8_bfly(float *p, uint32_t i0, uint32_t n2, const float *co, const float *si)
{
  const float C81 = 0.70710678118f;
  float *x[8];
  for (int q = 0; q < 8; q++) x[q] = p + 2 * (i0 + (uint32_t)q * n2);

  float ar[4], br[4], ai[4], bi[4];
  for (int q = 0; q < 4; q++) {
    ar[q] = x[q][0] + x[q + 4][0];
    br[q] = x[q][0] - x[q + 4][0];
    ai[q] = x[q][1] + x[q + 4][1];
    bi[q] = x[q][1] - x[q + 4][1];
  }
  /* even half: 4-point DFT of a[] */
  float t1 = ar[0] - ar[2], r1 = ar[0] + ar[2];
  float r3 = ar[1] - ar[3], r2 = ar[1] + ar[3];
  float t2 = ai[0] - ai[2], s1 = ai[0] + ai[2];
  float s3 = ai[1] - ai[3], s2 = ai[1] + ai[3];
  float X0r = r1 + r2, X4r = r1 - r2;
  float X0i = s1 + s2, X4i = s1 - s2;
  float X2r = t1 + s3, X6r = t1 - s3;
  float X2i = t2 - r3, X6i = t2 + r3;
  /* odd half */
  float d1r = (br[1] - br[3]) * C81, d3r = (br[1] + br[3]) * C81;
  float d1i = (bi[1] - bi[3]) * C81, d3i = (bi[1] + bi[3]) * C81;
  float u1 = br[0] - d1r, r5 = br[0] + d1r;
  float r8 = br[2] - d3r, r7 = br[2] + d3r;
  float u2 = bi[0] - d1i, s5 = bi[0] + d1i;
  float s8 = bi[2] - d3i, s7 = bi[2] + d3i;
  float X1r = r5 + s7, X7r = r5 - s7;
  float X5r = u1 + s8, X3r = u1 - s8;
  float X1i = s5 - r7, X7i = s5 + r7;
  float X5i = u2 - r8, X3i = u2 + r8;

  float Xr[8] = {X0r, X1r, X2r, X3r, X4r, X5r, X6r, X7r};
  float Xi[8] = {X0i, X1i, X2i, X3i, X4i, X5i, X6i, X7i};
  x[0][0] = Xr[0];
  x[0][1] = Xi[0];
  if (co) {
    for (int q = 1; q < 8; q++) {
      float p1 = co[q] * Xr[q], p2 = si[q] * Xi[q], p3 = co[q] * Xi[q], p4 = si[q] * Xr[q];
      x[q][0] = p1 + p2;
      x[q][1] = p3 - p4;
    }
  } else {
    for (int q = 1; q < 8; q++) {
      x[q][0] = Xr[q];
      x[q][1] = Xi[q];
    }
  }
}

/* all radix-8 stages of a length-len block; tw is the table of the FULL transform and
 * mod its stride (arm_cfft_radix8_f32.c:51-291) */
static void r8_stages(float *p, uint32_t len, const float *tw, uint32_t mod)
{
  uint32_t n2 = len;
  do {
    uint32_t n1 = n2;
    n2 >>= 3;
    for (uint32_t i1 = 0; i1 < len; i1 += n1) r8_bfly(p, i1, n2, NULL, NULL);
    if (n2 < 8) break;
    uint32_t ia = 0;
    for (uint32_t j = 1; j < n2; j++) {
      float co[8], si[8];
      ia += mod;
      for (uint32_t q = 1; q < 8; q++) {
        co[q] = tw[2 * (q * ia)];
        si[q] = tw[2 * (q * ia) + 1];
      }
      for (uint32_t i1 = j; i1 < len; i1 += n1) r8_bfly(p, i1, n2, co, si);
    }
    mod <<= 3;
  } while (n2 > 7);
}

/* radix-2 first stage for N = 16, 128, 1024 (arm_cfft_radix8by2_f32, arm_cfft_f32.c:846-958):
 * col1 = a + b, col2 = (a - b) * W^n, with the second quarter using the -j symmetry of the
 * first-quarter twiddle instead of its own table entry (:923-931). */
static void r8by2(float *p, uint32_t N, const float *tw)
{
  uint32_t h = N / 2, q = N / 4;
  for (uint32_t n = 0; n < q; n++) {
    float *a = p + 2 * n, *b = p + 2 * (n + h), *c = p + 2 * (n + q), *d = p + 2 * (n + q + h);
    float twR = tw[2 * n], twI = tw[2 * n + 1];
    float a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1], c0 = c[0], c1 = c[1], d0 = d[0], d1 = d[1];
    a[0] = a0 + b0;
    a[1] = a1 + b1;
    float e0 = a0 - b0, e1 = a1 - b1;
    c[0] = c0 + d0;
    c[1] = c1 + d1;
    float f0 = d0 - c0, f1 = d1 - c1;
    float m0 = e0 * twR, m1 = e1 * twI, m2 = e1 * twR, m3 = e0 * twI;
    b[0] = m0 + m1;
    b[1] = m2 - m3;
    m0 = f0 * twI;
    m1 = f1 * twR;
    m2 = f1 * twI;
    m3 = f0 * twR;
    d[0] = m0 - m1;
    d[1] = m2 + m3;
  }
  r8_stages(p, h, tw, 2);
  r8_stages(p + 2 * h, h, tw, 2);
}

/* radix-4 first stage for N = 32, 256, 2048 (arm_cfft_radix8by4_f32, arm_cfft_f32.c:960-1201):
 * columns k = 0..3 receive X_k * W^(k n); rows n and N/4 - n are processed together and the
 * bottom row reuses the top row's twiddles through their symmetries (:1077-1132). */
static void r8by4(float *p, uint32_t N, const float *tw)
{
  uint32_t L = N / 4;
  float *c1 = p, *c2 = p + 2 * L, *c3 = p + 4 * L, *c4 = p + 6 * L;
  for (uint32_t n = 0; n <= L / 2; n++) {
    /* TOP row n */
    float *e1 = c1 + 2 * n, *e2 = c2 + 2 * n, *e3 = c3 + 2 * n, *e4 = c4 + 2 * n;
    float a0 = e1[0] + e3[0], s0 = e1[0] - e3[0], a1 = e1[1] + e3[1], s1 = e1[1] - e3[1];
    float t2r = s0 + e2[1] - e4[1], t2i = s1 - e2[0] + e4[0];
    float t3r = a0 - e2[0] - e4[0], t3i = a1 - e2[1] - e4[1];
    float t4r = s0 - e2[1] + e4[1], t4i = s1 + e2[0] - e4[0];
    float o1r = a0 + e2[0] + e4[0], o1i = a1 + e2[1] + e4[1];
    int have_bottom = (n >= 1 && n < L / 2);
    float b2r = 0, b2i = 0, b3r = 0, b3i = 0, b4r = 0, b4i = 0, q1r = 0, q1i = 0;
    float *g1 = NULL, *g2 = NULL, *g3 = NULL, *g4 = NULL;
    if (have_bottom) {
      /* BOTTOM row m = N/4 - n, read before the top row is stored (rows are distinct) */
      uint32_t m = L - n;
      g1 = c1 + 2 * m;
      g2 = c2 + 2 * m;
      g3 = c3 + 2 * m;
      g4 = c4 + 2 * m;
      float ba1 = g1[0] + g3[0], bs1 = g1[0] - g3[0]; /* p1ap3_1 / p1sp3_1: real parts */
      float ba0 = g1[1] + g3[1], bs0 = g1[1] - g3[1]; /* p1ap3_0 / p1sp3_0: imag parts */
      b2r = g2[1] - g4[1] + bs1;
      b2i = g1[1] - g3[1] - g2[0] + g4[0];
      b3r = ba1 - g2[0] - g4[0];
      b3i = ba0 - g2[1] - g4[1];
      b4r = g2[1] - g4[1] - bs1;
      b4i = g4[0] - g2[0] - bs0;
      q1i = ba0 + g2[1] + g4[1];
      q1r = ba1 + g2[0] + g4[0];
    }
    e1[0] = o1r;
    e1[1] = o1i;
    if (n == 0) {
      e2[0] = t2r;
      e2[1] = t2i;
      e3[0] = t3r;
      e3[1] = t3i;
      e4[0] = t4r;
      e4[1] = t4i;
      continue;
    }
    float twR, twI, m0, m1, m2, m3;
    if (have_bottom) {
      g1[1] = q1i;
      g1[0] = q1r;
    }
    /* COL 2: W^n ; bottom uses (twI - j twR) */
    twR = tw[2 * n];
    twI = tw[2 * n + 1];
    m0 = t2r * twR; m1 = t2i * twI; m2 = t2i * twR; m3 = t2r * twI;
    e2[0] = m0 + m1;
    e2[1] = m2 - m3;
    if (have_bottom) {
      m0 = b2i * twI; m1 = b2r * twR; m2 = b2r * twI; m3 = b2i * twR;
      g2[1] = m0 - m1;
      g2[0] = m2 + m3;
    }
    /* COL 3: W^2n ; bottom uses (-twR - j twI) */
    twR = tw[2 * (2 * n)];
    twI = tw[2 * (2 * n) + 1];
    m0 = t3r * twR; m1 = t3i * twI; m2 = t3i * twR; m3 = t3r * twI;
    e3[0] = m0 + m1;
    e3[1] = m2 - m3;
    if (have_bottom) {
      m0 = -b3i * twR; m1 = b3r * twI; m2 = b3r * twR; m3 = b3i * twI;
      g3[1] = m0 - m1;
      g3[0] = m3 - m2;
    }
    /* COL 4: W^3n ; bottom holds the NEGATED X3 and uses (twI + j twR) on it */
    twR = tw[2 * (3 * n)];
    twI = tw[2 * (3 * n) + 1];
    m0 = t4r * twR; m1 = t4i * twI; m2 = t4i * twR; m3 = t4r * twI;
    e4[0] = m0 + m1;
    e4[1] = m2 - m3;
    if (have_bottom) {
      m0 = b4i * twI; m1 = b4r * twR; m2 = b4r * twI; m3 = b4i * twR;
      g4[1] = m0 - m1;
      g4[0] = m2 + m3;
    }
  }
  r8_stages(c1, L, tw, 4);
  r8_stages(c2, L, tw, 4);
  r8_stages(c3, L, tw, 4);
  r8_stages(c4, L, tw, 4);
}

/* Digit reversal realised by arm_bitreversal_32 + armBitRevIndexTable<N>
 * (arm_bitreversal2.c:84-108): position with digits (d0 | d1 d2 ...), d0 the radix-2/4 first
 * stage digit (if any) and d1.. the radix-8 digits, most significant first, holds output bin
 * k = d0 + r0*(d1 + 8*(d2 + 8*(...))).  perm[k] = position. */
int orc_f32_perm(uint32_t N, uint32_t *perm)
{
  uint32_t lg = ilog2(N);
  if (!is_pow2(N) || lg < 4 || lg > 12) return -1;
  uint32_t r0 = 1u << (lg % 3); /* 1, 2 or 4 */
  uint32_t n8 = lg / 3;
  for (uint32_t pos = 0; pos < N; pos++) {
    uint32_t rest = N / r0;
    uint32_t d0 = pos / rest, low = pos % rest;
    uint32_t k = 0, w = 1;
    /* radix-8 digits of low, most significant first, become least significant of k */
    for (uint32_t s = 0; s < n8; s++) {
      uint32_t dig = (low >> (3 * (n8 - 1 - s))) & 7u;
      k += dig * w;
      w *= 8;
    }
    k = d0 + r0 * k;
    perm[k] = pos;
  }
  return 0;
}

static const uint32_t *perm_f32(uint32_t N)
{
  uint32_t l = ilog2(N);
  if (!g_perm_f32[l]) {
    uint32_t *t = (uint32_t *)malloc(sizeof(uint32_t) * N);
    orc_f32_perm(N, t);
    g_perm_f32[l] = t;
  }
  return g_perm_f32[l];
}

int orc_cfft_f32(float *p, uint32_t N, int ifft, int bitrev)
{
  if (!is_pow2(N) || N < 16 || N > 4096) return -1; /* arm_cfft_init_f32.c:300-350 */
  const float *tw = tw_f32(N);
  if (ifft) /* arm_cfft_f32.c:1252-1261 */
    for (uint32_t l = 0; l < N; l++) p[2 * l + 1] = -p[2 * l + 1];
  switch (ilog2(N) % 3) { /* arm_cfft_f32.c:1263-1280 */
    case 1: r8by2(p, N, tw); break;  /* 16, 128, 1024 */
    case 2: r8by4(p, N, tw); break;  /* 32, 256, 2048 */
    default: r8_stages(p, N, tw, 1); /* 64, 512, 4096 */
  }
  if (bitrev) {
    const uint32_t *perm = perm_f32(N);
    float *tmp = (float *)malloc(sizeof(float) * 2 * N);
    memcpy(tmp, p, sizeof(float) * 2 * N);
    for (uint32_t k = 0; k < N; k++) {
      p[2 * k] = tmp[2 * perm[k]];
      p[2 * k + 1] = tmp[2 * perm[k] + 1];
    }
    free(tmp);
  }
  if (ifft) { /* arm_cfft_f32.c:1285-1297 */
    float invL = 1.0f / (float)N;
    for (uint32_t l = 0; l < N; l++) {
      p[2 * l] *= invL;
      p[2 * l + 1] = -(p[2 * l + 1]) * invL;
    }
  }
  return 0;
}

int orc_cfft_f32_batch(float *p, uint32_t N, uint32_t batch, int ifft, int bitrev)
{
  for (uint32_t b = 0; b < batch; b++)
    if (orc_cfft_f32(p + (size_t)2 * N * b, N, ifft, bitrev)) return -1;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Real FFT: arm_rfft_fast_f32 (arm_rfft_fast_f32.c:675-699), stage_rfft_f32 (:316-402),
 * merge_rfft_f32 (:405-462).  N real samples -> packed {DC, Nyquist, X1 .. X(N/2-1)}.
 * ------------------------------------------------------------------------------------------ */
static void stage_rfft(const float *p, float *out, uint32_t N)
{
  const float *tw = tw_rfft(N);
  uint32_t M = N / 2;
  float t1a = p[0] + p[0], t1b = p[1] + p[1];
  out[0] = 0.5f * (t1a + t1b);
  out[1] = 0.5f * (t1a - t1b);
  for (uint32_t k = 1; k < M; k++) {
    float xAR = p[2 * k], xAI = p[2 * k + 1], xBR = p[2 * (M - k)], xBI = p[2 * (M - k) + 1];
    float twR = tw[2 * k], twI = tw[2 * k + 1];
    float a = xBR - xAR, b = xBI + xAI;
    float p0 = twR * a, p1 = twI * a, p2 = twR * b, p3 = twI * b;
    out[2 * k] = 0.5f * (xAR + xBR + p0 + p3);
    out[2 * k + 1] = 0.5f * (xAI - xBI + p1 - p2);
  }
}

static void merge_rfft(const float *p, float *out, uint32_t N)
{
  const float *tw = tw_rfft(N);
  uint32_t M = N / 2;
  out[0] = 0.5f * (p[0] + p[1]);
  out[1] = 0.5f * (p[0] - p[1]);
  for (uint32_t k = 1; k < M; k++) {
    float xAR = p[2 * k], xAI = p[2 * k + 1], xBR = p[2 * (M - k)], xBI = p[2 * (M - k) + 1];
    float twR = tw[2 * k], twI = tw[2 * k + 1];
    float a = xAR - xBR, b = xAI + xBI;
    float r = twR * a, s = twI * b, t = twI * a, u = twR * b;
    out[2 * k] = 0.5f * (xAR + xBR - r - s);
    out[2 * k + 1] = 0.5f * (xAI - xBI + t - u);
  }
}

int orc_rfft_fast_f32(float *p, float *out, uint32_t N, int ifft)
{
  if (!is_pow2(N) || N < 32 || N > 4096) return -1; /* arm_rfft_fast_init_f32.c:340-361 */
  if (ifft) {
    merge_rfft(p, out, N);
    return orc_cfft_f32(out, N / 2, 1, 1);
  }
  if (orc_cfft_f32(p, N / 2, 0, 1)) return -1; /* the caller's input buffer is clobbered (:694) */
  stage_rfft(p, out, N);
  return 0;
}

int orc_rfft_fast_f32_batch(float *p, float *out, uint32_t N, uint32_t batch, int ifft)
{
  for (uint32_t b = 0; b < batch; b++)
    if (orc_rfft_fast_f32(p + (size_t)N * b, out + (size_t)N * b, N, ifft)) return -1;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Radix-2 f32 (the FFT the fork's GNSS code calls, up to 16384):
 * arm_cfft_radix2_f32 (arm_cfft_radix2_f32.c:64-88), forward butterfly :227-271,
 * inverse :392-463 (1/N folded into the last stage), arm_bitreversal_f32
 * (arm_bitreversal.c:46-105) = plain bit reversal.
 * ------------------------------------------------------------------------------------------ */
static uint32_t bitrev_u32(uint32_t v, uint32_t bits)
{
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

static void bitrev_swap_f32(float *p, uint32_t N)
{
  uint32_t lg = ilog2(N);
  for (uint32_t i = 0; i < N; i++) {
    uint32_t j = bitrev_u32(i, lg);
    if (i < j) {
      float t0 = p[2 * i], t1 = p[2 * i + 1];
      p[2 * i] = p[2 * j];
      p[2 * i + 1] = p[2 * j + 1];
      p[2 * j] = t0;
      p[2 * j + 1] = t1;
    }
  }
}

int orc_cfft_radix2_f32(float *p, uint32_t N, int ifft, int bitrev)
{
  if (!is_pow2(N) || N < 16 || N > 16384) return -1; /* arm_cfft_radix2_init_f32.c:92-207 */
  const float *tw = tw_f32(N);
  uint32_t n2 = N, mod = 1;
  uint32_t last = ifft ? 2 : 1;
  for (uint32_t k = N; k > last; k >>= 1) {
    uint32_t n1 = n2;
    n2 >>= 1;
    uint32_t ia = 0;
    for (uint32_t j = 0; j < n2; j++) {
      float c = tw[2 * ia], s = tw[2 * ia + 1];
      ia += mod;
      for (uint32_t i = j; i < N; i += n1) {
        uint32_t l = i + n2;
        float a0 = p[2 * i] + p[2 * l], xt = p[2 * i] - p[2 * l];
        float yt = p[2 * i + 1] - p[2 * l + 1], a1 = p[2 * l + 1] + p[2 * i + 1];
        float p0 = xt * c, p1 = yt * s, p2 = yt * c, p3 = xt * s;
        p[2 * i] = a0;
        p[2 * i + 1] = a1;
        if (ifft) {
          p[2 * l] = p0 - p1;
          p[2 * l + 1] = p2 + p3;
        } else {
          p[2 * l] = p0 + p1;
          p[2 * l + 1] = p2 - p3;
        }
      }
    }
    mod <<= 1;
  }
  if (ifft) {
    float inv = 1.0f / (float)N; /* onebyfftLen, arm_cfft_radix2_init_f32.c */
    for (uint32_t i = 0; i < N; i += 2) {
      uint32_t l = i + 1;
      float a0 = p[2 * i] + p[2 * l], xt = p[2 * i] - p[2 * l];
      float a1 = p[2 * l + 1] + p[2 * i + 1], yt = p[2 * i + 1] - p[2 * l + 1];
      p[2 * i] = a0 * inv;
      p[2 * l] = xt * inv;
      p[2 * i + 1] = a1 * inv;
      p[2 * l + 1] = yt * inv;
    }
  }
  if (bitrev) bitrev_swap_f32(p, N);
  return 0;
}

/* float64 truth: iterative radix-2 DIT, twiddles by direct cos/sin per butterfly group */
int orc_cfft_f64(double *p, uint32_t N, int ifft)
{
  if (!is_pow2(N)) return -1;
  uint32_t lg = ilog2(N);
  for (uint32_t i = 0; i < N; i++) {
    uint32_t j = bitrev_u32(i, lg);
    if (i < j) {
      double t0 = p[2 * i], t1 = p[2 * i + 1];
      p[2 * i] = p[2 * j];
      p[2 * i + 1] = p[2 * j + 1];
      p[2 * j] = t0;
      p[2 * j + 1] = t1;
    }
  }
  double sgn = ifft ? 1.0 : -1.0;
  double *wr = (double *)malloc(sizeof(double) * (N / 2 + 1));
  double *wi = (double *)malloc(sizeof(double) * (N / 2 + 1));
  for (uint32_t k = 0; k < N / 2; k++) {
    double a = 2.0 * M_PI * (double)k / (double)N;
    wr[k] = cos(a);
    wi[k] = sgn * sin(a);
  }
  for (uint32_t len = 2; len <= N; len <<= 1) {
    uint32_t half = len / 2, step = N / len;
    for (uint32_t i = 0; i < N; i += len) {
      for (uint32_t j = 0; j < half; j++) {
        double c = wr[j * step], s = wi[j * step];
        double *a = p + 2 * (i + j), *b = p + 2 * (i + j + half);
        double tr = b[0] * c - b[1] * s, ti = b[0] * s + b[1] * c;
        b[0] = a[0] - tr;
        b[1] = a[1] - ti;
        a[0] += tr;
        a[1] += ti;
      }
    }
  }
  free(wr);
  free(wi);
  if (ifft) {
    double inv = 1.0 / (double)N;
    for (uint32_t i = 0; i < 2 * N; i++) p[i] *= inv;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * q15: arm_cfft_q15 (arm_cfft_q15.c:671-726), arm_cfft_radix4by2[_inverse]_q15 (:736-931),
 * arm_radix4_butterfly[_inverse]_q15 generic-C branch (arm_cfft_radix4_q15.c:572-970,
 * :1434-1813), arm_bitreversal_16 with the plain bit-reversal table (arm_bitreversal2.c:123-148).
 * inv selects the inverse butterflies (conjugated twiddles).  All shifts are arithmetic, ssat16
 * is the host __SSAT (Include/dsp/none.h:78-94), plain +/- results wrap on the 16-bit store.
 * ------------------------------------------------------------------------------------------ */
static inline int32_t ssat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

static void r4_q15(int16_t *p, uint32_t len, const int16_t *tw, uint32_t mod, int inv)
{
  int16_t R0, R1, S0, S1, T0, T1, U0, U1, Co1, Si1, Co2, Si2, Co3, Si3, out1, out2;
  uint32_t n1, n2 = len >> 2, ic = 0;

  /* first stage: inputs >> 2, saturating sums, products >> 16 (radix4_q15.c:599-727) */
  for (uint32_t i0 = 0; i0 < n2; i0++) {
    uint32_t i1 = i0 + n2, i2 = i1 + n2, i3 = i2 + n2;
    T0 = p[2 * i0] >> 2; T1 = p[2 * i0 + 1] >> 2;
    S0 = p[2 * i2] >> 2; S1 = p[2 * i2 + 1] >> 2;
    R0 = (int16_t)ssat16(T0 + S0); R1 = (int16_t)ssat16(T1 + S1);
    S0 = (int16_t)ssat16(T0 - S0); S1 = (int16_t)ssat16(T1 - S1);
    T0 = p[2 * i1] >> 2; T1 = p[2 * i1 + 1] >> 2;
    U0 = p[2 * i3] >> 2; U1 = p[2 * i3 + 1] >> 2;
    T0 = (int16_t)ssat16(T0 + U0); T1 = (int16_t)ssat16(T1 + U1);
    p[2 * i0] = (int16_t)((R0 >> 1) + (T0 >> 1));
    p[2 * i0 + 1] = (int16_t)((R1 >> 1) + (T1 >> 1));
    R0 = (int16_t)ssat16(R0 - T0); R1 = (int16_t)ssat16(R1 - T1);
    Co2 = tw[2 * (2 * ic)]; Si2 = tw[2 * (2 * ic) + 1];
    if (!inv) {
      out1 = (int16_t)((Co2 * R0 + Si2 * R1) >> 16);
      out2 = (int16_t)((-Si2 * R0 + Co2 * R1) >> 16);
    } else {
      out1 = (int16_t)((Co2 * R0 - Si2 * R1) >> 16);
      out2 = (int16_t)((Si2 * R0 + Co2 * R1) >> 16);
    }
    T0 = p[2 * i1] >> 2; T1 = p[2 * i1 + 1] >> 2;
    p[2 * i1] = out1; p[2 * i1 + 1] = out2;
    U0 = p[2 * i3] >> 2; U1 = p[2 * i3 + 1] >> 2;
    T0 = (int16_t)ssat16(T0 - U0); T1 = (int16_t)ssat16(T1 - U1);
    Co1 = tw[2 * ic]; Si1 = tw[2 * ic + 1];
    Co3 = tw[2 * (3 * ic)]; Si3 = tw[2 * (3 * ic) + 1];
    if (!inv) {
      R0 = (int16_t)ssat16((int32_t)S0 - T1); R1 = (int16_t)ssat16((int32_t)S1 + T0);
      S0 = (int16_t)ssat16((int32_t)S0 + T1); S1 = (int16_t)ssat16((int32_t)S1 - T0);
      out1 = (int16_t)((Si1 * S1 + Co1 * S0) >> 16);
      out2 = (int16_t)((-Si1 * S0 + Co1 * S1) >> 16);
      p[2 * i2] = out1; p[2 * i2 + 1] = out2;
      out1 = (int16_t)((Si3 * R1 + Co3 * R0) >> 16);
      out2 = (int16_t)((-Si3 * R0 + Co3 * R1) >> 16);
      p[2 * i3] = out1; p[2 * i3 + 1] = out2;
    } else {
      R0 = (int16_t)ssat16((int32_t)S0 + T1); R1 = (int16_t)ssat16((int32_t)S1 - T0);
      S0 = (int16_t)ssat16((int32_t)S0 - T1); S1 = (int16_t)ssat16((int32_t)S1 + T0);
      out1 = (int16_t)((Co1 * S0 - Si1 * S1) >> 16);
      out2 = (int16_t)((Si1 * S0 + Co1 * S1) >> 16);
      p[2 * i2] = out1; p[2 * i2 + 1] = out2;
      out1 = (int16_t)((Co3 * R0 - Si3 * R1) >> 16);
      out2 = (int16_t)((Si3 * R0 + Co3 * R1) >> 16);
      p[2 * i3] = out1; p[2 * i3 + 1] = out2;
    }
    ic += mod;
  }

  /* middle stages (radix4_q15.c:733-870) */
  mod <<= 2;
  for (uint32_t k = len / 4; k > 4; k >>= 2) {
    n1 = n2;
    n2 >>= 2;
    ic = 0;
    for (uint32_t j = 0; j < n2; j++) {
      Co1 = tw[2 * ic]; Si1 = tw[2 * ic + 1];
      Co2 = tw[2 * (2 * ic)]; Si2 = tw[2 * (2 * ic) + 1];
      Co3 = tw[2 * (3 * ic)]; Si3 = tw[2 * (3 * ic) + 1];
      ic += mod;
      for (uint32_t i0 = j; i0 < len; i0 += n1) {
        uint32_t i1 = i0 + n2, i2 = i1 + n2, i3 = i2 + n2;
        T0 = p[2 * i0]; T1 = p[2 * i0 + 1];
        S0 = p[2 * i2]; S1 = p[2 * i2 + 1];
        R0 = (int16_t)ssat16(T0 + S0); R1 = (int16_t)ssat16(T1 + S1);
        S0 = (int16_t)ssat16(T0 - S0); S1 = (int16_t)ssat16(T1 - S1);
        T0 = p[2 * i1]; T1 = p[2 * i1 + 1];
        U0 = p[2 * i3]; U1 = p[2 * i3 + 1];
        T0 = (int16_t)ssat16(T0 + U0); T1 = (int16_t)ssat16(T1 + U1);
        p[2 * i0] = (int16_t)(((R0 >> 1) + (T0 >> 1)) >> 1);
        p[2 * i0 + 1] = (int16_t)(((R1 >> 1) + (T1 >> 1)) >> 1);
        R0 = (int16_t)((R0 >> 1) - (T0 >> 1)); R1 = (int16_t)((R1 >> 1) - (T1 >> 1));
        if (!inv) {
          out1 = (int16_t)((Co2 * R0 + Si2 * R1) >> 16);
          out2 = (int16_t)((-Si2 * R0 + Co2 * R1) >> 16);
        } else {
          out1 = (int16_t)((Co2 * R0 - Si2 * R1) >> 16);
          out2 = (int16_t)((Si2 * R0 + Co2 * R1) >> 16);
        }
        T0 = p[2 * i1]; T1 = p[2 * i1 + 1];
        p[2 * i1] = out1; p[2 * i1 + 1] = out2;
        U0 = p[2 * i3]; U1 = p[2 * i3 + 1];
        T0 = (int16_t)ssat16(T0 - U0); T1 = (int16_t)ssat16(T1 - U1);
        if (!inv) {
          R0 = (int16_t)((S0 >> 1) - (T1 >> 1)); R1 = (int16_t)((S1 >> 1) + (T0 >> 1));
          S0 = (int16_t)((S0 >> 1) + (T1 >> 1)); S1 = (int16_t)((S1 >> 1) - (T0 >> 1));
          out1 = (int16_t)((Co1 * S0 + Si1 * S1) >> 16);
          out2 = (int16_t)((-Si1 * S0 + Co1 * S1) >> 16);
          p[2 * i2] = out1; p[2 * i2 + 1] = out2;
          out1 = (int16_t)((Si3 * R1 + Co3 * R0) >> 16);
          out2 = (int16_t)((-Si3 * R0 + Co3 * R1) >> 16);
          p[2 * i3] = out1; p[2 * i3 + 1] = out2;
        } else {
          R0 = (int16_t)((S0 >> 1) + (T1 >> 1)); R1 = (int16_t)((S1 >> 1) - (T0 >> 1));
          S0 = (int16_t)((S0 >> 1) - (T1 >> 1)); S1 = (int16_t)((S1 >> 1) + (T0 >> 1));
          out1 = (int16_t)((Co1 * S0 - Si1 * S1) >> 16);
          out2 = (int16_t)((Si1 * S0 + Co1 * S1) >> 16);
          p[2 * i2] = out1; p[2 * i2 + 1] = out2;
          out1 = (int16_t)((Co3 * R0 - Si3 * R1) >> 16);
          out2 = (int16_t)((Si3 * R0 + Co3 * R1) >> 16);
          p[2 * i3] = out1; p[2 * i3 + 1] = out2;
        }
      }
    }
    mod <<= 2;
  }

  /* last stage, twiddle-free (radix4_q15.c:879-961 / :1726-1805) */
  n1 = n2;
  n2 >>= 2;
  for (uint32_t i0 = 0; i0 <= len - n1; i0 += n1) {
    uint32_t i1 = i0 + n2, i2 = i1 + n2, i3 = i2 + n2;
    T0 = p[2 * i0]; T1 = p[2 * i0 + 1];
    S0 = p[2 * i2]; S1 = p[2 * i2 + 1];
    R0 = (int16_t)ssat16(T0 + S0); R1 = (int16_t)ssat16(T1 + S1);
    S0 = (int16_t)ssat16(T0 - S0); S1 = (int16_t)ssat16(T1 - S1);
    T0 = p[2 * i1]; T1 = p[2 * i1 + 1];
    U0 = p[2 * i3]; U1 = p[2 * i3 + 1];
    T0 = (int16_t)ssat16(T0 + U0); T1 = (int16_t)ssat16(T1 + U1);
    p[2 * i0] = (int16_t)((R0 >> 1) + (T0 >> 1));
    p[2 * i0 + 1] = (int16_t)((R1 >> 1) + (T1 >> 1));
    R0 = (int16_t)((R0 >> 1) - (T0 >> 1)); R1 = (int16_t)((R1 >> 1) - (T1 >> 1));
    T0 = p[2 * i1]; T1 = p[2 * i1 + 1];
    p[2 * i1] = R0; p[2 * i1 + 1] = R1;
    U0 = p[2 * i3]; U1 = p[2 * i3 + 1];
    T0 = (int16_t)ssat16(T0 - U0); T1 = (int16_t)ssat16(T1 - U1);
    if (!inv) {
      p[2 * i2] = (int16_t)((S0 >> 1) + (T1 >> 1)); p[2 * i2 + 1] = (int16_t)((S1 >> 1) - (T0 >> 1));
      p[2 * i3] = (int16_t)((S0 >> 1) - (T1 >> 1)); p[2 * i3 + 1] = (int16_t)((S1 >> 1) + (T0 >> 1));
    } else {
      p[2 * i2] = (int16_t)((S0 >> 1) - (T1 >> 1)); p[2 * i2 + 1] = (int16_t)((S1 >> 1) + (T0 >> 1));
      p[2 * i3] = (int16_t)((S0 >> 1) + (T1 >> 1)); p[2 * i3 + 1] = (int16_t)((S1 >> 1) - (T0 >> 1));
    }
  }
}

/* arm_cfft_q15.c:736-833 (forward) / :835-931 (inverse), generic-C branch */
static void r4by2_q15(int16_t *p, uint32_t N, const int16_t *tw, int inv)
{
  uint32_t n2 = N >> 1;
  for (uint32_t i = 0; i < n2; i++) {
    int16_t c = tw[2 * i], s = tw[2 * i + 1];
    uint32_t l = i + n2;
    int16_t xt = (int16_t)((p[2 * i] >> 1) - (p[2 * l] >> 1));
    p[2 * i] = (int16_t)(((p[2 * i] >> 1) + (p[2 * l] >> 1)) >> 1);
    int16_t yt = (int16_t)((p[2 * i + 1] >> 1) - (p[2 * l + 1] >> 1));
    p[2 * i + 1] = (int16_t)(((p[2 * l + 1] >> 1) + (p[2 * i + 1] >> 1)) >> 1);
    if (!inv) {
      p[2 * l] = (int16_t)(((int16_t)(((int32_t)xt * c) >> 16)) + ((int16_t)(((int32_t)yt * s) >> 16)));
      p[2 * l + 1] = (int16_t)(((int16_t)(((int32_t)yt * c) >> 16)) - ((int16_t)(((int32_t)xt * s) >> 16)));
    } else {
      p[2 * l] = (int16_t)(((int16_t)(((int32_t)xt * c) >> 16)) - ((int16_t)(((int32_t)yt * s) >> 16)));
      p[2 * l + 1] = (int16_t)(((int16_t)(((int32_t)yt * c) >> 16)) + ((int16_t)(((int32_t)xt * s) >> 16)));
    }
  }
  r4_q15(p, n2, tw, 2, inv);
  r4_q15(p + N, n2, tw, 2, inv);
  for (uint32_t i = 0; i < 2 * N; i++) p[i] = (int16_t)((uint16_t)p[i] << 1);
}

int orc_cfft_q15(int16_t *p, uint32_t N, int ifft, int bitrev, int rounding)
{
  if (!is_pow2(N) || N < 16 || N > 16384) return -1; /* arm_cfft_q15.c:681-721 */
  const int16_t *tw = tw_q15(N, rounding);
  uint32_t lg = ilog2(N);
  if (lg & 1)
    r4by2_q15(p, N, tw, ifft ? 1 : 0);
  else
    r4_q15(p, N, tw, 1, ifft ? 1 : 0);
  if (bitrev) { /* armBitRevIndexTable_fixed_<N> = pairs (i, bitrev(i)), i < bitrev(i) */
    for (uint32_t i = 0; i < N; i++) {
      uint32_t j = bitrev_u32(i, lg);
      if (i < j) {
        int16_t t0 = p[2 * i], t1 = p[2 * i + 1];
        p[2 * i] = p[2 * j];
        p[2 * i + 1] = p[2 * j + 1];
        p[2 * j] = t0;
        p[2 * j + 1] = t1;
      }
    }
  }
  return 0;
}

int orc_cfft_q15_batch(int16_t *p, uint32_t N, uint32_t batch, int ifft, int bitrev, int rounding)
{
  for (uint32_t b = 0; b < batch; b++)
    if (orc_cfft_q15(p + (size_t)2 * N * b, N, ifft, bitrev, rounding)) return -1;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * q31: arm_cfft_q31 (arm_cfft_q31.c:708-759), arm_cfft_radix4by2[_inverse]_q31 (:767-885),
 * arm_radix4_butterfly[_inverse]_q31 (arm_cfft_radix4_q31.c:153-473, :524-834); multiplies from
 * Include/dsp/none.h:185-206.  Additions wrap modulo 2^32 (done in uint32_t), >> is arithmetic.
 * ------------------------------------------------------------------------------------------ */
static inline int32_t mulhi(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * b) >> 32); }
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wshl1(int32_t a) { return (int32_t)((uint32_t)a << 1); }

static void r4_q31(int32_t *p, uint32_t len, const int32_t *tw, uint32_t mod, int inv)
{
  uint32_t n1, n2 = len >> 2, ia1 = 0;
  int32_t t1, t2, r1, r2, s1, s2, co1, co2, co3, si1, si2, si3;

  /* first stage: 4 guard bits, products >> 32 then << 1 (radix4_q31.c:187-282) */
  for (uint32_t i0 = 0; i0 < n2; i0++) {
    uint32_t i1 = i0 + n2, i2 = i1 + n2, i3 = i2 + n2;
    r1 = wadd(p[2 * i0] >> 4, p[2 * i2] >> 4);
    r2 = wsub(p[2 * i0] >> 4, p[2 * i2] >> 4);
    t1 = wadd(p[2 * i1] >> 4, p[2 * i3] >> 4);
    s1 = wadd(p[2 * i0 + 1] >> 4, p[2 * i2 + 1] >> 4);
    s2 = wsub(p[2 * i0 + 1] >> 4, p[2 * i2 + 1] >> 4);
    p[2 * i0] = wadd(r1, t1);
    r1 = wsub(r1, t1);
    t2 = wadd(p[2 * i1 + 1] >> 4, p[2 * i3 + 1] >> 4);
    p[2 * i0 + 1] = wadd(s1, t2);
    s1 = wsub(s1, t2);
    t1 = wsub(p[2 * i1 + 1] >> 4, p[2 * i3 + 1] >> 4);
    t2 = wsub(p[2 * i1] >> 4, p[2 * i3] >> 4);
    co2 = tw[2 * (2 * ia1)]; si2 = tw[2 * (2 * ia1) + 1];
    co1 = tw[2 * ia1]; si1 = tw[2 * ia1 + 1];
    co3 = tw[2 * (3 * ia1)]; si3 = tw[2 * (3 * ia1) + 1];
    if (!inv) {
      p[2 * i1] = wshl1(wadd(mulhi(r1, co2), mulhi(s1, si2)));
      p[2 * i1 + 1] = wshl1(wsub(mulhi(s1, co2), mulhi(r1, si2)));
      r1 = wadd(r2, t1); r2 = wsub(r2, t1);
      s1 = wsub(s2, t2); s2 = wadd(s2, t2);
      p[2 * i2] = wshl1(wadd(mulhi(r1, co1), mulhi(s1, si1)));
      p[2 * i2 + 1] = wshl1(wsub(mulhi(s1, co1), mulhi(r1, si1)));
      p[2 * i3] = wshl1(wadd(mulhi(r2, co3), mulhi(s2, si3)));
      p[2 * i3 + 1] = wshl1(wsub(mulhi(s2, co3), mulhi(r2, si3)));
    } else {
      p[2 * i1] = wshl1(wsub(mulhi(r1, co2), mulhi(s1, si2)));
      p[2 * i1 + 1] = wshl1(wadd(mulhi(s1, co2), mulhi(r1, si2)));
      r1 = wsub(r2, t1); r2 = wadd(r2, t1);
      s1 = wadd(s2, t2); s2 = wsub(s2, t2);
      p[2 * i2] = wshl1(wsub(mulhi(r1, co1), mulhi(s1, si1)));
      p[2 * i2 + 1] = wshl1(wadd(mulhi(s1, co1), mulhi(r1, si1)));
      p[2 * i3] = wshl1(wsub(mulhi(r2, co3), mulhi(s2, si3)));
      p[2 * i3 + 1] = wshl1(wadd(mulhi(s2, co3), mulhi(r2, si3)));
    }
    ia1 += mod;
  }

  /* middle stages (radix4_q31.c:294-395) */
  mod <<= 2;
  for (uint32_t k = len / 4; k > 4; k >>= 2) {
    n1 = n2;
    n2 >>= 2;
    ia1 = 0;
    for (uint32_t j = 0; j < n2; j++) {
      co1 = tw[2 * ia1]; si1 = tw[2 * ia1 + 1];
      co2 = tw[2 * (2 * ia1)]; si2 = tw[2 * (2 * ia1) + 1];
      co3 = tw[2 * (3 * ia1)]; si3 = tw[2 * (3 * ia1) + 1];
      ia1 += mod;
      for (uint32_t i0 = j; i0 < len; i0 += n1) {
        uint32_t i1 = i0 + n2, i2 = i1 + n2, i3 = i2 + n2;
        r1 = wadd(p[2 * i0], p[2 * i2]);
        r2 = wsub(p[2 * i0], p[2 * i2]);
        s1 = wadd(p[2 * i0 + 1], p[2 * i2 + 1]);
        s2 = wsub(p[2 * i0 + 1], p[2 * i2 + 1]);
        t1 = wadd(p[2 * i1], p[2 * i3]);
        p[2 * i0] = wadd(r1, t1) >> 2;
        r1 = wsub(r1, t1);
        t2 = wadd(p[2 * i1 + 1], p[2 * i3 + 1]);
        p[2 * i0 + 1] = wadd(s1, t2) >> 2;
        s1 = wsub(s1, t2);
        t1 = wsub(p[2 * i1 + 1], p[2 * i3 + 1]);
        t2 = wsub(p[2 * i1], p[2 * i3]);
        if (!inv) {
          p[2 * i1] = wadd(mulhi(r1, co2), mulhi(s1, si2)) >> 1;
          p[2 * i1 + 1] = wsub(mulhi(s1, co2), mulhi(r1, si2)) >> 1;
          r1 = wadd(r2, t1); r2 = wsub(r2, t1);
          s1 = wsub(s2, t2); s2 = wadd(s2, t2);
          p[2 * i2] = wadd(mulhi(r1, co1), mulhi(s1, si1)) >> 1;
          p[2 * i2 + 1] = wsub(mulhi(s1, co1), mulhi(r1, si1)) >> 1;
          p[2 * i3] = wadd(mulhi(r2, co3), mulhi(s2, si3)) >> 1;
          p[2 * i3 + 1] = wsub(mulhi(s2, co3), mulhi(r2, si3)) >> 1;
        } else {
          p[2 * i1] = wsub(mulhi(r1, co2), mulhi(s1, si2)) >> 1;
          p[2 * i1 + 1] = wadd(mulhi(s1, co2), mulhi(r1, si2)) >> 1;
          r1 = wsub(r2, t1); r2 = wadd(r2, t1);
          s1 = wadd(s2, t2); s2 = wsub(s2, t2);
          p[2 * i2] = wsub(mulhi(r1, co1), mulhi(s1, si1)) >> 1;
          p[2 * i2 + 1] = wadd(mulhi(s1, co1), mulhi(r1, si1)) >> 1;
          p[2 * i3] = wsub(mulhi(r2, co3), mulhi(s2, si3)) >> 1;
          p[2 * i3 + 1] = wadd(mulhi(s2, co3), mulhi(r2, si3)) >> 1;
        }
      }
    }
    mod <<= 2;
  }

  /* last stage: plain adds on 4 consecutive points (radix4_q31.c:405-464 / :766-826) */
  for (uint32_t g = 0; g < len / 4; g++) {
    int32_t *q = p + 8 * g;
    int32_t xa = q[0], ya = q[1], xb = q[2], yb = q[3], xc = q[4], yc = q[5], xd = q[6], yd = q[7];
    q[0] = wadd(wadd(wadd(xa, xb), xc), xd);
    q[1] = wadd(wadd(wadd(ya, yb), yc), yd);
    q[2] = wsub(wadd(wsub(xa, xb), xc), xd);
    q[3] = wsub(wadd(wsub(ya, yb), yc), yd);
    if (!inv) {
      q[4] = wsub(wsub(wadd(xa, yb), xc), yd);
      q[5] = wadd(wsub(wsub(ya, xb), yc), xd);
      q[6] = wadd(wsub(wsub(xa, yb), xc), yd);
      q[7] = wsub(wsub(wadd(ya, xb), yc), xd);
    } else {
      q[4] = wadd(wsub(wsub(xa, yb), xc), yd);
      q[5] = wsub(wsub(wadd(ya, xb), yc), xd);
      q[6] = wsub(wsub(wadd(xa, yb), xc), yd);
      q[7] = wadd(wsub(wsub(ya, xb), yc), xd);
    }
  }
}

/* arm_cfft_q31.c:767-826 / :828-885: rounded high-word multiplies (SMMULR/SMMLAR/SMMLSR) */
static void r4by2_q31(int32_t *p, uint32_t N, const int32_t *tw, int inv)
{
  uint32_t n2 = N >> 1;
  for (uint32_t i = 0; i < n2; i++) {
    int32_t c = tw[2 * i], s = tw[2 * i + 1];
    uint32_t l = i + n2;
    int32_t xt = wsub(p[2 * i] >> 2, p[2 * l] >> 2);
    p[2 * i] = wadd(p[2 * i] >> 2, p[2 * l] >> 2);
    int32_t yt = wsub(p[2 * i + 1] >> 2, p[2 * l + 1] >> 2);
    p[2 * i + 1] = wadd(p[2 * l + 1] >> 2, p[2 * i + 1] >> 2);
    int32_t p0 = (int32_t)(((int64_t)xt * c + 0x80000000LL) >> 32);
    int32_t p1 = (int32_t)(((int64_t)yt * c + 0x80000000LL) >> 32);
    if (!inv) {
      p0 = (int32_t)(((int64_t)((uint64_t)(int64_t)p0 << 32) + (int64_t)yt * s + 0x80000000LL) >> 32);
      p1 = (int32_t)(((int64_t)((uint64_t)(int64_t)p1 << 32) - (int64_t)xt * s + 0x80000000LL) >> 32);
    } else {
      p0 = (int32_t)(((int64_t)((uint64_t)(int64_t)p0 << 32) - (int64_t)yt * s + 0x80000000LL) >> 32);
      p1 = (int32_t)(((int64_t)((uint64_t)(int64_t)p1 << 32) + (int64_t)xt * s + 0x80000000LL) >> 32);
    }
    p[2 * l] = wshl1(p0);
    p[2 * l + 1] = wshl1(p1);
  }
  r4_q31(p, n2, tw, 2, inv);
  r4_q31(p + N, n2, tw, 2, inv);
  for (uint32_t i = 0; i < 2 * N; i++) p[i] = wshl1(p[i]);
}

int orc_cfft_q31(int32_t *p, uint32_t N, int ifft, int bitrev, int rounding)
{
  /* reference generic-C range is 16..4096 (arm_cfft_q31.c:718-754); 8192/16384 are the same
   * butterflies at the fork's larger table sizes (extension) */
  if (!is_pow2(N) || N < 16 || N > 16384) return -1;
  const int32_t *tw = tw_q31(N, rounding);
  uint32_t lg = ilog2(N);
  if (lg & 1)
    r4by2_q31(p, N, tw, ifft ? 1 : 0);
  else
    r4_q31(p, N, tw, 1, ifft ? 1 : 0);
  if (bitrev) {
    for (uint32_t i = 0; i < N; i++) {
      uint32_t j = bitrev_u32(i, lg);
      if (i < j) {
        int32_t t0 = p[2 * i], t1 = p[2 * i + 1];
        p[2 * i] = p[2 * j];
        p[2 * i + 1] = p[2 * j + 1];
        p[2 * j] = t0;
        p[2 * j + 1] = t1;
      }
    }
  }
  return 0;
}

int orc_cfft_q31_batch(int32_t *p, uint32_t N, uint32_t batch, int ifft, int bitrev, int rounding)
{
  for (uint32_t b = 0; b < batch; b++)
    if (orc_cfft_q31(p + (size_t)2 * N * b, N, ifft, bitrev, rounding)) return -1;
  return 0;
}
