// armfft_gnss.hip -- GNSS code-phase/Doppler acquisition for gfx950, fused around the LDS FFT core.
//
// Replaces the fork's per-cell recipe (read_ors, CMSIS/CMSIS.cpp:706-737: synth_gps_prn replica ->
// fft_c32(replica), fft_c32(signal), S*conj(R) -> fft_c32 inverse -> find_top2_peaks_cplx;
// coherent fold CMSIS.cpp:1154-1169) with ONE kernel per acquisition:
//
//   workgroup (scene, Doppler bin d):
//     wipe-off + fold   z[m] = u_d[m] * sum_b w_d[b] * x[b*spm + m]        (rows a15, a16)
//                       u_d[m] = e^{-j 2 pi f_d m / FS}, w_d[b] = e^{-j 2 pi f_d b spm / FS}
//     Z = FFT(z zero-padded to fft_len)                                    (a2/a9/a13)
//     for every PRN p:  y = IFFT(Z * conj(C_p)),  |y|                      (a12)
//                       top-2 peaks (separation pk_sep), SNR, 3-point interpolation   (a14)
//   C_p = FFT(C/A code p, sample-and-hold, zero Doppler), computed once per plan (a10, a11).
//
// Z stays in registers across the PRN loop, the code spectra (n_prn x fft_len x 8 B, 1 MiB for
// 32 PRNs at 4096) are L2-resident, the correlation surface never leaves the CU: per (PRN,
// Doppler) row only the 32-byte result tuple goes to HBM.  Bound: LDS bandwidth / fp32 VALU.
//
// find_top2_peaks_cplx (gnss_codes.c:1144-1166) is a sequential scan whose second peak depends on
// which index was the running best when each sample was seen; top2_scan() below reproduces it
// exactly (values and indices, ties included) with a workgroup prefix scan.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "armfft_internal.h"
#include "fft_core_f32.hpp"

namespace armfft {

// ------------------------------------------------------------------------------------------
// GPS L1 C/A code (getCode, gnss_codes.c:969-1030)
// ------------------------------------------------------------------------------------------
// G2 tap pairs (0-based stage indices) for PRN 1..37, gnss_codes.c:971-1008
__constant__ int8_t k_g2_taps[37][2] = {
    {1, 5}, {2, 6}, {3, 7}, {4, 8}, {0, 8}, {1, 9}, {0, 7}, {1, 8}, {2, 9}, {1, 2}, {2, 3}, {4, 5}, {5, 6},
    {6, 7}, {7, 8}, {8, 9}, {0, 3}, {1, 4}, {2, 5}, {3, 6}, {4, 7}, {5, 8}, {0, 2}, {3, 5}, {4, 6}, {5, 7},
    {6, 8}, {7, 9}, {0, 5}, {1, 6}, {2, 7}, {3, 8}, {4, 9}, {3, 9}, {0, 6}, {1, 7}, {3, 9}};

constexpr int kCodeLen = 1023;
constexpr float kRefPiF = 3.14159265358979f;   // the reference's PI: a FLOAT literal (Include/gnss_codes.h:33), 3.1415927410125732 as a double

// one sequence per workgroup at every size: N/16 threads
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int LOGN>
using GPlan = Plan<LOGN, (1 << LOGN) / 16>;

// chips[p][i] (+1 / -1 as int8), i < 1023.  One workgroup per PRN: thread 0 clocks the two 10-stage
// registers (all-ones start; G1 feedback stages {2,9}, G2 {1,2,5,7,8,9}) and records G1's output
// and G2's whole state per step, then all threads combine the PRN's two G2 taps.
__global__ void gnss_code_kernel(const int32_t *__restrict__ prns, int8_t *__restrict__ chips)
{
  __shared__ uint16_t g2_state[kCodeLen];
  __shared__ uint8_t g1_out[kCodeLen];
  const int prn = prns[blockIdx.x];
  if (threadIdx.x == 0) {
    uint32_t g1 = 0x3FF, g2 = 0x3FF;  // bit j = stage j
    for (int i = 0; i < kCodeLen; i++) {
      g1_out[i] = (g1 >> 9) & 1;
      g2_state[i] = (uint16_t)g2;
      const uint32_t f1 = ((g1 >> 2) ^ (g1 >> 9)) & 1;
      const uint32_t f2 = ((g2 >> 1) ^ (g2 >> 2) ^ (g2 >> 5) ^ (g2 >> 7) ^ (g2 >> 8) ^ (g2 >> 9)) & 1;
      g1 = ((g1 << 1) | f1) & 0x3FF;
      g2 = ((g2 << 1) | f2) & 0x3FF;
    }
  }
  __syncthreads();
  const int s0 = k_g2_taps[prn - 1][0], s1 = k_g2_taps[prn - 1][1];
  for (int i = threadIdx.x; i < kCodeLen; i += blockDim.x) {
    const int val = g1_out[i] ^ ((g2_state[i] >> s0) & 1) ^ ((g2_state[i] >> s1) & 1);
    chips[(size_t)blockIdx.x * kCodeLen + i] = val ? 1 : -1;  // logic 1 -> +1 (:1016-1018)
  }
}

// l5_generate_I from the epoch (gnss_codes.c:153-204): two 13-stage registers, all ones (:147-151),
// right shift with the feedback injected at stage 13 (:101-106; G1 taps 13,12,10,9,1 :90-94, G2
// taps 13,12,8,7,6,4,3,1 :95-99), chip = G1 stage 13 ^ two PRN-specific G2 stages (L5I_TAPS
// :115-124), 0 -> +1, 1 -> -1; the registers run on across the 10230-chip boundary (:194-197 wraps
// only the counter).  A tap number 0 reads as 0 (the reference shifts by 2^32 - 1 there, which the
// hosts it runs on take modulo 32).  Same structure as gnss_code_kernel: one thread clocks, all combine.
__constant__ uint8_t k_l5i_taps[38][2] = {
    {0, 0},
    {1, 5}, {2, 6}, {3, 7}, {4, 8}, {0, 8}, {1, 9}, {0, 7}, {1, 8}, {2, 9}, {1, 2},
    {2, 3}, {4, 5}, {5, 6}, {6, 7}, {7, 8}, {8, 9}, {0, 3}, {1, 4}, {2, 5}, {3, 6},
    {4, 7}, {5, 8}, {0, 2}, {3, 5}, {4, 6}, {5, 7}, {6, 8}, {7, 9}, {0, 5}, {1, 6},
    {2, 7}, {3, 8}, {4, 9}, {3, 9}, {0, 6}, {1, 7}, {3, 9},
};
constexpr int kL5Chunk = 2048;  // chips per round of the generator (LDS staging)
__global__ void gnss_l5_code_kernel(int prn, uint32_t n_chips, int8_t *__restrict__ out)
{
  __shared__ uint16_t g2_state[kL5Chunk];
  __shared__ uint8_t g1_out[kL5Chunk];
  __shared__ uint32_t carry[2];
  if (threadIdx.x == 0) { carry[0] = 0x1FFFu; carry[1] = 0x1FFFu; }
  const unsigned t1 = k_l5i_taps[prn][0], t2 = k_l5i_taps[prn][1];
  for (uint32_t base = 0; base < n_chips; base += kL5Chunk) {
    const uint32_t cnt = n_chips - base < (uint32_t)kL5Chunk ? n_chips - base : (uint32_t)kL5Chunk;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t g1 = carry[0], g2 = carry[1];
      for (uint32_t i = 0; i < cnt; i++) {
        g1_out[i] = (g1 >> 12) & 1u;
        g2_state[i] = (uint16_t)g2;
        const uint32_t f1 = ((g1 >> 12) ^ (g1 >> 11) ^ (g1 >> 9) ^ (g1 >> 8) ^ g1) & 1u;
        const uint32_t f2 = ((g2 >> 12) ^ (g2 >> 11) ^ (g2 >> 7) ^ (g2 >> 6) ^ (g2 >> 5) ^ (g2 >> 3) ^ (g2 >> 2) ^ g2) & 1u;
        g1 = ((g1 >> 1) | (f1 << 12)) & 0x1FFFu;
        g2 = ((g2 >> 1) | (f2 << 12)) & 0x1FFFu;
      }
      carry[0] = g1;
      carry[1] = g2;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
      const unsigned a = t1 ? (g2_state[i] >> (t1 - 1u)) & 1u : 0u, b = t2 ? (g2_state[i] >> (t2 - 1u)) & 1u : 0u;
      out[base + i] = (g1_out[i] ^ a ^ b) ? -1 : 1;
    }
  }
}

// synth_e1b_prn (gnss_codes.c:920-965) with cboc_e1b_weight (:253-260), boc_sign (:247-251) and code_chip_at
// (:265-270): Galileo E1-B CBOC(6,1,1/11) replica at fs = 4.092 MHz, chip rate 1.023 MHz, code and carrier phase
// starting at 0, data bit +1.  The reference advances BOTH phases by float recursion:
//   chips += 1.023e6f / 4.092e6f (= 0.25 exactly; every sum and the wrap at 4092 are exact in f32), so
//           chips(n) = (n mod 16368) / 4 in closed form;
//   phia  += dphia, rounded to f32 at every step -- order-dependent, so ONE thread walks the n steps (a 16384-step
//           dependent chain of f32 adds is ~60 us, once per plan) and leaves phia(n) in the output buffer's .x slot;
// then every thread forms its sample: amplitude chip * (w1 * s11 - w6 * s61) in f32, carrier (float)cos / (float)sin of
// the DOUBLE-evaluated f32 phase (sincosf_fast, :573-576).  One workgroup per code; codes = n_codes x 4092 chips (+1/-1,
// the E1B_Code[prn] rows a caller loads from the ICD's hex tables: hexstream_to_e1b_chips :276-289).
__global__ __launch_bounds__(256) void gnss_e1b_replica_kernel(const int8_t *__restrict__ codes, float doppler, uint32_t size, float2 *__restrict__ out)
{
  constexpr int L = 4092;
  const int8_t *ca = codes + (size_t)blockIdx.x * L;
  float2 *o = out + (size_t)blockIdx.x * size;
  const float PI_F = 3.14159265358979f;  // Include/gnss_codes.h:33
  const float fs_hz = 4.092e6f;
  if (threadIdx.x == 0) {
    const float dphia = 2.0f * PI_F * doppler / fs_hz;
    float phia = 0.0f;
    for (uint32_t n = 0; n < size; n++) {
      o[n].x = phia;
      phia = __fadd_rn(phia, dphia);
      if (phia > 1e9f || phia < -1e9f) phia = fmodf(phia, 2.0f * PI_F);
    }
  }
  __threadfence_block();
  __syncthreads();
  const float w1 = 0.9534625892455923f, w6 = 0.3015113445777636f;
  for (uint32_t n = threadIdx.x; n < size; n += blockDim.x) {
    const uint32_t q = n % (4u * L);              // chips = q / 4, exact
    const int idx = (int)(q >> 2);
    const float frac = (float)(q & 3u) * 0.25f;
    const int s11 = ((int)floorf(2.0f * 1 * frac) & 1) ? -1 : +1;
    const int s61 = ((int)floorf(2.0f * 6 * frac) & 1) ? -1 : +1;
    const float sca = __fsub_rn(__fmul_rn(w1, (float)s11), __fmul_rn(w6, (float)s61));
    const float ampa = __fmul_rn((float)ca[idx], sca);
    const double ph = (double)o[n].x;
    const float sa = (float)sin(ph), caph = (float)cos(ph);
    o[n] = make_float2(__fmul_rn(ampa, caph), __fmul_rn(ampa, sa));
  }
}

// probe of getCode: out[n] = chip[(n / spc) mod 1023], n < num * spc
__global__ void gnss_expand_code_kernel(const int8_t *__restrict__ chips, int32_t *__restrict__ out, int total, int spc)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < total) out[n] = chips[(n / spc) % kCodeLen];
}

// probe of synth_gps_prn / mix_prn (gnss_codes.c:1032-1071): replica[n] = code[n] e^{j (phi0 + n dth)}.  The reference
// advances a double phasor by one complex multiplication per sample and renormalises it (:1055-1061); here every sample
// evaluates its own phase in double -- the recursion's accumulated rounding (< 1e-12 over 16384 samples) is far below the
// float rounding of the result, so the two agree to the last float bit except on rounding ties.  dth and phi0 come from
// the host, computed with the reference's own constants (mix_prn uses the FLOAT literal PI of Include/gnss_codes.h:33,
// make_replica the double M_PI of quasi_pilot_tests.c:10-12).
__global__ void gnss_replica_kernel(const int8_t *__restrict__ chips, float2 *__restrict__ out, int size, int spc, double dth, double phi0)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= size) return;
  double s, c;
  sincos(phi0 + (double)n * dth, &s, &c);
  const double a = (double)chips[(n / spc) % kCodeLen];
  out[n] = make_float2((float)(a * c), (float)(a * s));
}

// mix_prn (gnss_codes.c:1032-1063) / make_replica (quasi_pilot_tests.c:153-181) on a caller-supplied code of `size` ints
__global__ void gnss_mix_code_kernel(const int32_t *__restrict__ code, float2 *__restrict__ out, int size, double dth, double phi0)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= size) return;
  double s, c;
  sincos(phi0 + (double)n * dth, &s, &c);
  const double a = (double)code[n];   // :1046 / :164
  out[n] = make_float2((float)(a * c), (float)(a * s));
}

// rotate_fwd (quasi_pilot_tests.c:138-151): out[i] = in[i - offset], indices wrapped once
__global__ void gnss_rotate_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ out, int size, int offset)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= size) return;
  const int j = i - offset;
  out[i] = in[j < 0 ? size + j : j];
}

// ------------------------------------------------------------------------------------------
// exact parallel form of find_top2_peaks (gnss_codes.c:1144-1190)
// ------------------------------------------------------------------------------------------
// Sequential rule, per sample x at index m with state (v1, i1, v2, i2):
//   x >= v1 : if (i1 != -1 && |m - i1| > sep) (v2, i2) = (v1, i1);  (v1, i1) = (x, m)
//   else if x > v2 && |m - i1| > sep : (v2, i2) = (x, m)
// => (v1, i1) = maximum, LAST index on ties;  the state (v1, i1) seen by sample m is the prefix
// maximum before m (last index on ties), so every sample yields at most one candidate for the
// second peak: (A) a record hands down the previous record if that lies > sep away, (B) a
// non-record more than sep away from the current record offers itself.  v2 = the largest
// candidate; on equal values an (A) candidate always overwrites (latest wins), a (B) candidate
// only takes over when strictly larger (earliest wins) -- encoded in the key below.
struct PrefixMax {
  float v;
  int i;
};
__device__ __forceinline__ PrefixMax pm_combine(PrefixMax earlier, PrefixMax later) { return later.v >= earlier.v ? later : earlier; }

// ---- cross-lane moves without the LDS crossbar (DPP, gfx9): a lane whose source lies outside its
// row of 16, or that the row mask excludes, keeps `old` ----
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float src)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xF, false));
}
constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118, kDppWaveShr1 = 0x138;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

struct Top2 {
  float v1, v2;
  int i1, i2;
  double sum;  // sum of all n magnitudes
};

constexpr float kNoPeak = -1e10f;  // the reference's initial value (gnss_codes.c:1148)

// second-peak candidate: larger value wins; on equal values the larger `ord` wins, with
//   ord(A) = 1<<31 | position of the record that hands down << 15 | index   (latest A wins, A beats B)
//   ord(B) =          (32767 - m) << 15 | m                                 (earliest B wins)
// ord == 0 means "no candidate".
struct Cand {
  float v;
  unsigned ord;
};
__device__ __forceinline__ Cand cand_better(Cand a, Cand b)
{
  const bool take_b = (b.ord != 0u) && ((a.ord == 0u) || (b.v > a.v) || (b.v == a.v && b.ord > a.ord));
  return take_b ? b : a;
}

// LDS scratch of top2_scan for T threads: per wave a prefix-maximum total, a lower bound, a candidate and a double sum
constexpr size_t top2_scratch_bytes(int T) { return (size_t)((T + 63) / 64) * 32 + 16; }

// T threads, thread tid owns samples [tid*CH, tid*CH + CH) of mag(m) (m < n valid; FULL: n == T*CH).
// `xch` = LDS scratch of top2_scratch_bytes(T) bytes not aliased with what mag() reads.  All threads return the result.
template <int T, int CH, bool FULL, class Mag>
__device__ __forceinline__ Top2 top2_scan(Mag mag, int n, int sep, int tid, void *xch)
{
  constexpr int NW = (T + 63) / 64;
  const int lane = tid & 63, wv = tid >> 6;
  const int m0 = tid * CH;
  float x[CH];
  PrefixMax loc{kNoPeak, -1};  // chunk maximum (last index on ties)
  float fsum = 0.f;
#pragma unroll
  for (int c = 0; c < CH; c++) {
    const int m = m0 + c;
    const bool ok = FULL || m < n;
    x[c] = ok ? mag(m) : kNoPeak;
    fsum += ok ? x[c] : 0.f;
    const bool rec = ok && x[c] >= loc.v;
    loc.v = rec ? x[c] : loc.v;
    loc.i = rec ? m : loc.i;
  }
  double sum = (double)fsum;
  // inclusive scan of the per-thread maxima over the wave, then over the waves
  // (DPP: shifts inside the rows of 16 lanes, then the row totals broadcast into the later rows;
  // a lane without a source receives the identity {kNoPeak, -1}, which pm_combine ignores)
  PrefixMax inc = loc;
#define ARMFFT_SCAN_STEP(CTRL, MASK)                                                                   \
  {                                                                                                    \
    const PrefixMax o{dpp_f<CTRL, MASK>(kNoPeak, inc.v), dpp_i<CTRL, MASK>(-1, inc.i)};                \
    inc = pm_combine(o, inc);                                                                          \
  }
  ARMFFT_SCAN_STEP(kDppShr1, 0xF) ARMFFT_SCAN_STEP(kDppShr2, 0xF) ARMFFT_SCAN_STEP(kDppShr4, 0xF) ARMFFT_SCAN_STEP(kDppShr8, 0xF)
  ARMFFT_SCAN_STEP(kDppBcast15, 0xA) ARMFFT_SCAN_STEP(kDppBcast31, 0xC)
#undef ARMFFT_SCAN_STEP
  const PrefixMax exc{dpp_f<kDppWaveShr1>(kNoPeak, inc.v), dpp_i<kDppWaveShr1>(-1, inc.i)};
  PrefixMax *wtot = reinterpret_cast<PrefixMax *>(xch);   // [NW]
  Cand *wcand = reinterpret_cast<Cand *>(wtot + NW);       // [NW]
  double *wsum = reinterpret_cast<double *>(wcand + NW);   // [NW]
  float *wlow = reinterpret_cast<float *>(wsum + NW);      // [NW]
  if (lane == 63) wtot[wv] = inc;
  __syncthreads();
  PrefixMax before{kNoPeak, -1}, total{kNoPeak, -1};
#pragma unroll
  for (int w = 0; w < NW; w++) {
    const PrefixMax t = wtot[w];
    if (w < wv) before = pm_combine(before, t);
    total = pm_combine(total, t);
  }
  PrefixMax run = pm_combine(before, exc);  // state (v1, i1) seen by this thread's first sample

  // the candidate sample m offers when the state before it is r (see the rule above)
  const unsigned usep2 = 2u * (unsigned)sep;
  auto offer = [&](float xv, int m, bool ok, PrefixMax r) -> Cand {
    const bool rec = ok && xv >= r.v;
    // |m - r.i| > sep  <=>  (unsigned)(m - r.i + sep) > 2 sep ; r.i == -1 (nothing seen yet) offers nothing
    const bool far = ((unsigned)(m - r.i + sep) > usep2) && r.i != -1;
    Cand k;
    k.v = rec ? r.v : xv;
    k.ord = rec ? ((1u << 31) | ((unsigned)m << 15) | (unsigned)r.i) : (((unsigned)(32767 - m) << 15) | (unsigned)m);
    if (!(ok && far)) k.ord = 0u;
    return k;
  };
  // Lower bound L of the runner-up: a chunk that starts more than sep after the maximum's position i1 sees the final
  // state (v1, i1) at every sample, none of its samples is a record (the maximum keeps the LAST index on ties), so its
  // own maximum is a real (B) candidate.  L = the largest of these over the workgroup (none: no bound).  A sample below L
  // cannot win -- as a (B) candidate its value is its own, as an (A) candidate it hands down a record that is not
  // larger than itself -- and equal values still take part (>=).
  float low = (loc.i != -1 && m0 > total.i + sep) ? loc.v : -__builtin_huge_valf();
  low = fmaxf(low, dpp_f<kDppXor1>(low, low));
  low = fmaxf(low, dpp_f<kDppXor2>(low, low));
  low = fmaxf(low, dpp_f<kDppHalfMirror>(low, low));
  low = fmaxf(low, dpp_f<kDppMirror>(low, low));
  low = fmaxf(low, __shfl_xor(low, 16));
  low = fmaxf(low, __shfl_xor(low, 32));
  if (lane == 0) wlow[wv] = low;
  __syncthreads();
  float lower = wlow[0];
#pragma unroll
  for (int w = 1; w < NW; w++) lower = fmaxf(lower, wlow[w]);
  // Walk the chunk: the state advances at every sample (3 operations), the offer is evaluated only in steps where some
  // lane of the wave holds a sample >= L; a wave none of whose chunk maxima reaches L skips the walk -- for a noise
  // row that leaves a handful of steps in one or two waves.
  Cand best{0.f, 0u};
  if (__any(loc.v >= lower)) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int m = m0 + c;
      const bool ok = FULL || m < n;
      if (__any(ok && x[c] >= lower)) best = cand_better(best, offer(x[c], m, ok, run));
      const bool rec = ok && x[c] >= run.v;
      run.v = rec ? x[c] : run.v;
      run.i = rec ? m : run.i;
    }
    // (butterfly inside the rows by DPP -- cand_better is a maximum over distinct keys, so any
    // pairing works --, the two cross-row steps through the LDS crossbar)
#define ARMFFT_RED_STEP(CTRL)                                                                                        \
  best = cand_better(best, Cand{dpp_f<CTRL>(best.v, best.v), (unsigned)dpp_i<CTRL>((int)best.ord, (int)best.ord)});
    ARMFFT_RED_STEP(kDppXor1) ARMFFT_RED_STEP(kDppXor2) ARMFFT_RED_STEP(kDppHalfMirror) ARMFFT_RED_STEP(kDppMirror)
#undef ARMFFT_RED_STEP
    best = cand_better(best, Cand{__shfl_xor(best.v, 16), __shfl_xor(best.ord, 16)});
    best = cand_better(best, Cand{__shfl_xor(best.v, 32), __shfl_xor(best.ord, 32)});
  }
  {
    auto dsum_step = [&](auto mover) {
      const long long bits = __double_as_longlong(sum);
      const int lo = mover((int)(bits & 0xffffffffll)), hi = mover((int)(bits >> 32));
      sum += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    dsum_step([](int w) { return dpp_i<kDppXor1>(w, w); });
    dsum_step([](int w) { return dpp_i<kDppXor2>(w, w); });
    dsum_step([](int w) { return dpp_i<kDppHalfMirror>(w, w); });
    dsum_step([](int w) { return dpp_i<kDppMirror>(w, w); });
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
  }
  if (lane == 0) {
    wcand[wv] = best;
    wsum[wv] = sum;
  }
  __syncthreads();
  best = Cand{0.f, 0u};
  sum = 0.0;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    best = cand_better(best, wcand[w]);
    sum += wsum[w];
  }
  Top2 r;
  r.v1 = total.v;
  r.i1 = total.i;
  r.sum = sum;
  r.v2 = best.ord ? best.v : kNoPeak;
  r.i2 = best.ord ? (int)(best.ord & 0x7FFFu) : -1;
  return r;
}

// A scanned row before its scalar arithmetic.  finish_row is a chain of ~160 dependent double-precision
// instructions (two divisions, a log10) that one lane would run per row while its workgroup waits; the fused
// kernels park the rows in LDS instead and finish up to kPendRows of them in one pass, one lane per row, so the
// chain is issued once per kPendRows rows (measured: 8 % of the GPS L1 acquisition).
struct RowRaw {
  float v1, v2;
  int i1, i2;
  double sum;
  float m2, m1, p1, p2;  // mag(i1 - 2), mag((i1 + n - 1) % n), mag((i1 + 1) % n), mag(i1 + 2); 0 where the index is out of range
};
constexpr int kPendRows = 64;  // one wave finishes them: writer (lane 0) and readers are the same wave
constexpr size_t pend_bytes() { return kPendRows * sizeof(RowRaw); }

template <class Mag>
__device__ __forceinline__ RowRaw gather_row(const Top2 &t, Mag mag, int n)
{
  RowRaw r;
  r.v1 = t.v1, r.v2 = t.v2, r.i1 = t.i1, r.i2 = t.i2, r.sum = t.sum;
  r.m2 = (t.i1 - 2 >= 0 && t.i1 - 2 < n) ? mag(t.i1 - 2) : 0.f;
  r.m1 = mag((t.i1 + n - 1) % n);
  r.p1 = mag((t.i1 + 1) % n);
  r.p2 = (t.i1 + 2 >= 0 && t.i1 + 2 < n) ? mag(t.i1 + 2) : 0.f;
  return r;
}

// compute_snr_cplx (gnss_codes.c:1110-1125) and InterpolateCodePhase (:1096-1108) on the finished row
__device__ __forceinline__ void finish_row(const RowRaw &t, int n, int dop_bin, int prn, gnss_acq_result *out)
{
  // the five samples around the peak leave the noise sum in index order, those outside the row are skipped;
  // where i1 - 1 / i1 + 1 lie inside the row the wrapped neighbours m1 / p1 are those samples, and mag(i1) is v1
  double off = t.sum;
  if (t.i1 - 2 >= 0 && t.i1 - 2 < n) off -= (double)t.m2;
  if (t.i1 - 1 >= 0 && t.i1 - 1 < n) off -= (double)t.m1;
  if (t.i1 >= 0 && t.i1 < n) off -= (double)t.v1;
  if (t.i1 + 1 >= 0 && t.i1 + 1 < n) off -= (double)t.p1;
  if (t.i1 + 2 >= 0 && t.i1 + 2 < n) off -= (double)t.p2;
  const double mean = off / (double)(n - 5);
  const double early = (double)t.m1, late = (double)t.p1, prompt = (double)t.v1;
  double xi, interp;
  if (late > early) {
    xi = 0.5 * (late - early) / (prompt - early);
    interp = (double)t.i1 + (xi / 3.0) * (2.0 * xi + 2.0);
  } else {
    xi = 0.5 * (late - early) / (prompt - late);
    interp = (double)t.i1 + (xi / 3.0) * (-2.0 * xi + 2.0);
  }
  gnss_acq_result r;
  r.peak = t.v1;
  r.second = t.v2;
  r.code_phase = t.i1;
  r.second_phase = t.i2;
  r.doppler_bin = dop_bin;
  r.prn = prn;
  r.snr_db = (float)(10.0 * log10(prompt / mean));
  r.code_phase_interp = (float)interp;
  *out = r;
}

// Row p of a (scene, Doppler bin) is parked by lane 0 (before a barrier of the caller); rows [p - p % kPendRows, p]
// are finished by the first lanes once the buffer is full or p is the last row.
template <class Mag>
__device__ __forceinline__ void park_row(RowRaw *pend, const Top2 &t, Mag mag, int n, int p, int tid)
{
  if (tid == 0) pend[p % kPendRows] = gather_row(t, mag, n);
}
__device__ __forceinline__ void flush_rows(const RowRaw *pend, int n, int p, int n_prn, int d, const int32_t *__restrict__ prns,
                                           gnss_acq_result *__restrict__ rows_of_scene, int n_dop, int tid)
{
  if (p % kPendRows != kPendRows - 1 && p != n_prn - 1) return;
  const int first = p - p % kPendRows;
  if (tid <= p - first) finish_row(pend[tid], n, d, prns[first + tid], rows_of_scene + (size_t)(first + tid) * n_dop + d);
}

// probe: find_top2_peaks_{real,cplx} on `batch` rows of n samples in global memory
template <bool CPLX>
__global__ __launch_bounds__(256) void gnss_top2_kernel(const float *__restrict__ data, int n, int sep, gnss_top2 *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) unsigned char xch[top2_scratch_bytes(256)];
  const float *row = data + (size_t)blockIdx.x * n * (CPLX ? 2 : 1);
  auto mag = [&](int m) -> float {
    if (CPLX) {
      const float re = row[2 * m], im = row[2 * m + 1];
      return sqrtf(re * re + im * im);
    }
    return row[m];
  };
  const Top2 t = top2_scan<256, 64, false>(mag, n, sep, threadIdx.x, xch);  // n <= 16384
  if (threadIdx.x == 0) {
    gnss_top2 r;
    r.val1 = t.v1;
    r.idx1 = t.i1;
    r.val2 = t.v2;
    r.idx2 = t.i2;
    out[blockIdx.x] = r;
  }
}

// probe: compute_snr_cplx / compute_snr_real (gnss_codes.c:1110-1142): 10 log10(peak / mean of the magnitudes more than
// two samples away from peak_loc), the mean over cov_size - 5 as in the reference; double sums (the reference adds the
// float magnitudes into a double one by one; here per-thread partial sums in double, combined through LDS)
template <bool CPLX>
__global__ __launch_bounds__(256) void gnss_snr_kernel(const float *__restrict__ data, int n, double peak, int peak_loc, double *__restrict__ out)
{
  __shared__ double part[256];
  double sum = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    float magn;
    if (CPLX) {
      const float re = data[2 * i], im = data[2 * i + 1];
      magn = sqrtf(re * re + im * im);
    } else {
      magn = data[i];
    }
    const int d = i - peak_loc;
    if ((d < 0 ? -d : d) > 2) sum += (double)magn;
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = part[0] / (double)(n - 5);
    out[0] = 10.0 * log10(peak / mean);
  }
}

// ------------------------------------------------------------------------------------------
// fine search: recover_prn_phase_deg_with_doppler (CMSIS/CMSIS.cpp:362-434) -- around an acquired (code offset, Doppler)
// the reference wipes the carrier at doppler + f, f = -1, 0, +1 Hz, with a FLOAT recursive phasor (no renormalisation,
// :391-404), correlates the wiped millisecond against the code at every offset within +-10 samples of the acquired one
// (double sums of float products, :407-416) and keeps the largest |sum| -- first one wins among equals, f outer / offset
// inner, both ascending -- returning atan2(im, re) in degrees, the offset, f and the power.
// One workgroup per satellite.  The three phasor recursions are sequential by construction (each step rounds to float),
// so three lanes run them, one each, with the reference's operations in the reference's order (no contraction) and leave
// the phasors in LDS; all threads then wipe (two float products and a float sum per component, as :396-397) and every wave
// takes its share of the 3 x 21 dot products.  The double sums are accumulated per lane and combined by a tree, not in
// the reference's index order: they agree with it to ~1e-13 relative, the offset / f decisions are the reference's.
// ------------------------------------------------------------------------------------------
constexpr int kRefineSpan = 10, kRefineOffs = 2 * kRefineSpan + 1, kRefineF = 3, kRefineMaxSize = 1023 * 4;   // static float iw[1023 * 4] (:392)
__host__ __device__ constexpr size_t refine_res_offset(int size) { return (((size_t)kRefineF * size * 8 + (size_t)size * 4) + 15) & ~(size_t)15; }
__host__ __device__ constexpr size_t refine_lds_bytes(int size) { return refine_res_offset(size) + (size_t)kRefineF * kRefineOffs * 2 * sizeof(double); }

__global__ __launch_bounds__(256) void gnss_refine_kernel(const float2 *__restrict__ sig, int sig_stride, const int32_t *__restrict__ codes, int size,
                                                          const int32_t *__restrict__ code_offsets, const double *__restrict__ dopplers,
                                                          double sample_rate, gnss_refine_result *__restrict__ out)
{
  // The reference's float expressions round after every product (MSVC / gcc without contraction).  Under the library's
  // -ffp-contract=fast the backend fuses a product into the sum that follows it whatever the source says (HIP's __fmul_rn is
  // a plain operator, a contract(off) pragma only clears per-instruction flags that =fast does not consult; seen: the
  // recursion drifts by 2e-6 over a millisecond).  rnd() hands the rounded product through an empty asm the combiner cannot
  // look into.
  auto rnd = [](float v) { asm volatile("" : "+v"(v)); return v; };
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float2 *wq = reinterpret_cast<float2 *>(lds_raw);                       // [3][size] (iw, qw) per f
  float *code = reinterpret_cast<float *>(wq + kRefineF * size);            // [size], the ints as the floats the products see (:414)
  double *res = reinterpret_cast<double *>(lds_raw + refine_res_offset(size));   // [63][2]
  const int tid = threadIdx.x, sat = blockIdx.x;
  const float2 *__restrict__ x = sig + (size_t)sat * sig_stride;
  const int32_t *__restrict__ prn = codes + (size_t)sat * size;
  const double fd = dopplers[sat];
  const int code_offset = code_offsets[sat];
  if (tid < kRefineF) {
    const float f = -1.0f + (float)tid;                                     // :384: f = -span; f <= +span + 1e-3f; f += step
    const double dtheta = 2.0 * (double)kRefPiF * (fd + (double)f) / sample_rate;   // :386 (PI is the float literal)
    const float c_inc = (float)cos(dtheta), s_inc = (float)sin(dtheta);
    float pc = 1.0f, ps = 0.0f;
    float2 *w = wq + tid * size;
    for (int n = 0; n < size; n++) {
      w[n] = make_float2(pc, ps);
      const float npc = rnd(pc * c_inc) - rnd(ps * s_inc);      // :399-400
      const float nps = rnd(pc * s_inc) + rnd(ps * c_inc);
      pc = npc;
      ps = nps;
    }
  }
  for (int n = tid; n < size; n += 256) code[n] = (float)prn[n];
  __syncthreads();
  for (int e = tid; e < kRefineF * size; e += 256) {
    const int n = e % size;
    const float2 p = wq[e], v = x[n];
    // (I + jQ) e^{-j theta}: iw = +I pc + Q ps, qw = -I ps + Q pc (:396-397), each product and sum rounded to float
    const float iw = rnd(+v.x * p.x) + rnd(v.y * p.y), qw = rnd(-v.x * p.y) + rnd(v.y * p.x);
    wq[e] = make_float2(iw, qw);
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  for (int combo = wave; combo < kRefineF * kRefineOffs; combo += 4) {
    const int fi = combo / kRefineOffs;
    int off = code_offset - kRefineSpan + combo % kRefineOffs;
    off = ((off % size) + size) % size;
    const float2 *w = wq + fi * size;
    double re = 0.0, im = 0.0;
    for (int n = lane; n < size; n += 64) {
      int idx = n - off;
      if (idx < 0) idx += size;                                              // :412-413
      const float c = code[idx];
      const float pr = rnd(w[n].x * c), pi = rnd(w[n].y * c);                // float product, double sum (:414-415)
      re += (double)pr;
      im += (double)pi;
    }
    for (int d = 32; d > 0; d >>= 1) {
      re += __shfl_xor(re, d);
      im += __shfl_xor(im, d);
    }
    if (lane == 0) {
      res[2 * combo] = re;
      res[2 * combo + 1] = im;
    }
  }
  __syncthreads();
  if (tid == 0) {
    // the reference's visiting order: f ascending, then off = 0 .. size-1 ascending over the offsets inside the window --
    // a window that wraps past either end of the code period (:378-380, :408-409) is visited low offsets first
    // the window exactly as :376-380 builds it
    float code_low = (float)code_offset - (float)kRefineSpan, code_high = (float)code_offset + (float)kRefineSpan;
    int wrap = 0;
    if (code_low < 0) { code_low += (float)size; wrap = 1; }
    if (code_high > (float)size) { code_high -= (float)size; wrap = 1; }
    double best_pow = -1.0, best_re = 0.0, best_im = 0.0, best_f = 0.0;
    int best_off = 0;
    const int lo = code_offset - kRefineSpan;
    for (int fi = 0; fi < kRefineF; fi++) {
      // offsets of the window in ascending order of the wrapped value
      int first = 0;   // index k (0..20) of the smallest wrapped offset
      {
        int min_off = size;
        for (int k = 0; k < kRefineOffs; k++) {
          const int o = (((lo + k) % size) + size) % size;
          if (o < min_off) { min_off = o; first = k; }
        }
      }
      for (int kk = 0; kk < kRefineOffs; kk++) {
        const int k = (first + kk) % kRefineOffs;
        const int combo = fi * kRefineOffs + k;
        const float off = (float)((((lo + k) % size) + size) % size);
        if (wrap == 0 && (off < code_low || off > code_high)) continue;        // :408
        if (wrap == 1 && (off > code_high && off < code_low)) continue;         // :409
        const double re = res[2 * combo], im = res[2 * combo + 1];
        const double pw = sqrt(re * re + im * im);
        if (pw > best_pow) {                                                  // :418: strict, the first of equals stays
          best_pow = pw;
          best_f = (double)(-1.0f + (float)fi);
          best_re = re;
          best_im = im;
          best_off = (((lo + k) % size) + size) % size;
        }
      }
    }
    gnss_refine_result r;
    r.phase_deg = atan2(best_im, best_re) * (double)(180.0f / kRefPiF);      // :432, R2D = (180.0f / 3.14159265358979f)
    r.pwr = (float)best_pow;
    r.doppler = (float)best_f;
    r.best_off = best_off;
    r.reserved = 0;
    out[sat] = r;
  }
}

// ------------------------------------------------------------------------------------------
// code spectra: C_p = FFT(code p sampled at spc samples per chip, spm samples, zero-padded)
// ------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(GPlan<LOGN>::T) void gnss_code_spectrum_kernel(const int8_t *__restrict__ chips,
                                                                                             float2 *__restrict__ spectra,
                                                                                             const float2 *__restrict__ tw, int spm, int spc)
{
  using PL = GPlan<LOGN>;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int8_t *c = chips + (size_t)blockIdx.x * kCodeLen;
  float2 *dst = spectra + (size_t)blockIdx.x * PL::N;
  auto ld = [&](int, int e) -> float2 { return make_float2(e < spm ? (float)c[(e / spc) % kCodeLen] : 0.f, 0.f); };
  auto st = [&](int, int k, float2 v) { dst[k] = v; };
  fft_block<PL, false, false>(lds, tw, threadIdx.x, ld, st);
}

// ------------------------------------------------------------------------------------------
// the fused acquisition kernel
// ------------------------------------------------------------------------------------------
// 2-bit I/Q samples of the fork's recordings, DecodeOrsIQCplx (CMSIS/CMSIS.cpp:465-485): one byte =
// two complex samples, bits [1:0] I0, [3:2] Q0, [5:4] I1, [7:6] Q1, code -> {+1, -1, +3, -3}.  Decoding
// inside the wipe-off load cuts the signal bytes 32x.
__host__ __device__ __forceinline__ size_t ors_bytes(size_t n_samples) { return (n_samples + 1) / 2; }
__device__ __forceinline__ float ors_level(unsigned code) { return (code & 2u ? 3.0f : 1.0f) * (code & 1u ? -1.0f : 1.0f); }
__device__ __forceinline__ float2 ors_sample(const uint8_t *__restrict__ data, int n)
{
  const unsigned byte = data[n >> 1], sh = (n & 1) * 4;
  return make_float2(ors_level((byte >> sh) & 3u), ors_level((byte >> (sh + 2)) & 3u));
}

__global__ void gnss_decode_ors_kernel(const uint8_t *__restrict__ data, uint32_t n_samples, float2 *__restrict__ out)
{
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < n_samples) out[n] = ors_sample(data, (int)n);
}

// Wipe-off + coherent fold of milliseconds [ms0, ms0 + count) + forward FFT; on return (after the
// trailing barrier) lds[PL::idx(0, k)] holds the spectrum in natural order.  The carrier phase is
// referenced to sample 0 of the scene, so groups of milliseconds stay phase-continuous.
// acc + a w, two packed multiply-adds: (a.x w.x, a.x w.y) + acc, then (-a.y w.y, a.y w.x) on top
__device__ __forceinline__ float2 cmac(float2 acc, float2 a, float2 w)
{
  v2f r, q;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(tov(a)), "v"(tov(w)), "v"(tov(acc)));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(q) : "v"(tov(a)), "v"(tov(w)), "v"(r));
  return tof(q);
}

template <class PL>
__device__ __forceinline__ void fold_and_fft(float2 *lds, float2 *wblk, const float2 *__restrict__ tw, const float2 *__restrict__ sig,
                                             const uint8_t *__restrict__ ors, int packed, double fd, double FS, int ms0, int count, int spm,
                                             int tid)
{
  constexpr int N = PL::N, T = PL::T;
  if (tid < count) {
    double cyc = fd * (double)(ms0 + tid) * (double)spm / FS;
    cyc -= floor(cyc);
    double s, c;
    sincospi(2.0 * cyc, &s, &c);
    wblk[tid] = make_float2((float)c, (float)(-s));
  }
  __syncthreads();
  // fold: the block phasors are the same for every point, so the blocks are the outer loop -- one phasor read, N/T
  // independent loads in flight and two packed multiply-adds per sample, the sums held in registers
  constexpr int PT = N / T;
  float2 acc[PT];
#pragma unroll
  for (int j = 0; j < PT; j++) acc[j] = make_float2(0.f, 0.f);
#pragma unroll 1
  for (int b = 0; b < count; b++) {
    const float2 w = wblk[b];
    const int base = (ms0 + b) * spm;
    float2 xv[PT];
#pragma unroll
    for (int j = 0; j < PT; j++) {
      const int e = tid + j * T;
      xv[j] = make_float2(0.f, 0.f);
      if (e < spm) xv[j] = packed ? ors_sample(ors, base + e) : sig[base + e];
    }
#pragma unroll
    for (int j = 0; j < PT; j++) acc[j] = cmac(acc[j], xv[j], w);
  }
#pragma unroll
  for (int j = 0; j < PT; j++) lds[PL::idx(0, tid + j * T)] = acc[j];
  // carrier wipe-off of the folded millisecond (every thread revisits its own points: no barrier)
  // one point at a time: the double-precision phase is register-hungry
#pragma unroll 1
  for (int e = tid; e < spm; e += T) {
    double cyc = fd * (double)e / FS;
    cyc -= floor(cyc);
    double s, c;
    sincospi(2.0 * cyc, &s, &c);
    lds[PL::idx(0, e)] = cmul(lds[PL::idx(0, e)], make_float2((float)c, (float)(-s)));
  }
  __syncthreads();
  {
    auto ld = [&](int, int e) -> float2 { return lds[PL::idx(0, e)]; };
    auto st = [&](int, int k, float2 v) { lds[PL::idx(0, k)] = v; };
    fft_block<PL, true, true>(lds, tw, tid, ld, st);
  }
  __syncthreads();
}

constexpr int kMaxMs = 64;

template <int LOGN>
__global__ __launch_bounds__(GPlan<LOGN>::T, GPlan<LOGN>::T >= 1024 ? 4 : 3) void gnss_acquire_kernel(
    const void *__restrict__ signal_any,  // [n_scenes][n_ms * spm] complex f32, or 2-bit packed (see ors_sample)
    int packed,
    const float2 *__restrict__ spectra,  // [n_prn][N]
    const float2 *__restrict__ tw, const float *__restrict__ dopplers, const int32_t *__restrict__ prns, int n_prn, int n_dop,
    int dop_begin, int n_ms, int spm, double fs_hz, int sep,
    gnss_acq_result *__restrict__ all)   // [n_scenes][n_prn][n_dop]
{
  using PL = GPlan<LOGN>;
  constexpr int N = PL::N, T = PL::T;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  // LDS map: [0, LDS_ELEMS) FFT image (the magnitude row aliases its head), then the exchange
  // scratch of top2_scan, then the per-block phasors w[b], then the parked rows
  unsigned char *xch = reinterpret_cast<unsigned char *>(lds + PL::LDS_ELEMS);
  float2 *wblk = reinterpret_cast<float2 *>(xch + top2_scratch_bytes(T));
  RowRaw *pend = reinterpret_cast<RowRaw *>(wblk + kMaxMs);
  float *magl = reinterpret_cast<float *>(lds);
  auto mag_idx = [](int m) { return m + ((m >> 4) << 2); };  // 16 floats + 4 pad: conflict-free 16-byte chunk reads

  const int tid = threadIdx.x;
  const int d = dop_begin + blockIdx.x, scene = blockIdx.y;
  const double FS = fs_hz;
  const double fd = (double)dopplers[d];
  const float2 *__restrict__ sig = reinterpret_cast<const float2 *>(signal_any) + (size_t)scene * n_ms * spm;
  const uint8_t *__restrict__ ors = reinterpret_cast<const uint8_t *>(signal_any) + (size_t)scene * ors_bytes(n_ms * spm);

  fold_and_fft<PL>(lds, wblk, tw, sig, ors, packed, fd, FS, 0, n_ms, spm, tid);
  // conj(Z) in the order the first pass consumes its inputs
  // ... scaled by 1/N here, once per Doppler bin instead of once per output of every PRN: a power
  // of two, so every later product and sum is the unscaled one times 2^-LOGN exactly
  const float inv_n = 1.0f / (float)N;
  float2 zc[16];
  for_each_first_pass_input<PL>(tid, [&](int slot, int, int e) {
    const float2 z = lds[PL::idx(0, e)];
    zc[slot] = make_float2(z.x * inv_n, -z.y * inv_n);
  });
  __syncthreads();

  // twiddles of the per-PRN transforms: powers of thread-constant bases held in registers for the
  // whole PRN loop (TwPow), no twiddle loads or address math per transform
  // (up to 4096 points; the larger sizes are register-bound and keep the table)
  typename std::conditional<(LOGN <= 12), TwPow<PL>, TwTable>::type twp;
  twp.preload(tw, tid);
  for (int p = 0; p < n_prn; p++) {
    int lt = tid;
    asm volatile("" : "+v"(lt));  // per-thread LDS address math stays inside the loop (else: spills)
    twp.pin();
    // the PRN's spectrum through a buffer descriptor: one per-lane offset register for the 16
    // legs, scalar leg offsets, no 64-bit address arithmetic in vector registers
    const __amdgpu_buffer_rsrc_t cs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(spectra + (size_t)p * N), 0, (int)(N * sizeof(float2)), 0x00020000);
    // IFFT(Z conj(C)) = conj(FFT(conj(Z) C)) / N; only the magnitude is used
    float2 v[16];
    for_each_first_pass_input_split<PL>(lt, [&](int slot, int, int j, int roff) {
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(cs, j * 8, roff * 8, 0);
      v[slot] = cmul(zc[slot], make_float2(__uint_as_float(w[0]), __uint_as_float(w[1])));
    });
    // (v_sqrt_f32: 1 ulp, no denormal rescue -- the magnitudes are O(1) after the 1/N scaling above)
    auto st = [&](int, int k, float2 y) { magl[mag_idx(k)] = __builtin_amdgcn_sqrtf(y.x * y.x + y.y * y.y); };
    // (the previous row may still be read by the lane that parks it: the first pass waits before its first write)
    fft_block_preloaded<PL, true, PL::NPASS - 1, true>(v, lds, twp, lt, st);
    __syncthreads();
    auto mag = [&](int m) -> float { return magl[mag_idx(m)]; };
    const Top2 t = top2_scan<T, 16, true>(mag, N, sep, lt, xch);
    park_row(pend, t, mag, N, p, lt);
    flush_rows(pend, N, p, n_prn, d, prns, all + (size_t)scene * n_prn * n_dop, n_dop, lt);
  }
}

// ------------------------------------------------------------------------------------------
// non-coherent and differential accumulation over groups of milliseconds
// ------------------------------------------------------------------------------------------
// The n_ms milliseconds of a scene are cut into n_groups groups; every group is folded coherently and
// correlated on its own, y_g = IFFT(FFT(z_g) conj(C_p)), and the rows are combined as
//   MODE 1 (non-coherent, the NCI loop of CMSIS/CMSIS.cpp:915-943):   r[m] = sum_g |y_g[m]|
//   MODE 2 (differential, test_quasi_diff_pilot CMSIS.cpp:1030-1062): r[m] = |sum_{g>=1} y_g[m] conj(y_{g-1}[m])|
// followed by find_top2_peaks_real / compute_snr_real / InterpolateCodePhase on r (CMSIS.cpp:948-951).
// Two kernels: the group spectra (conjugated) go through a caller-provided workspace, because one
// workgroup cannot hold n_groups spectra (32 KiB each at 4096 points) next to its FFT image.
template <int LOGN>
__global__ __launch_bounds__(GPlan<LOGN>::T, 4) void gnss_group_spectra_kernel(const void *__restrict__ signal_any, int packed,
                                                                              const float2 *__restrict__ tw, const float *__restrict__ dopplers,
                                                                              int dop_begin, int n_ms, int ms_per_group, int spm, double fs_hz,
                                                                              float2 *__restrict__ work)  // [scene][d_local][group][N]
{
  using PL = GPlan<LOGN>;
  constexpr int N = PL::N, T = PL::T;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  float2 *wblk = lds + PL::LDS_ELEMS;
  const int tid = threadIdx.x, g = blockIdx.z, scene = blockIdx.y, n_groups = gridDim.z;
  const double FS = fs_hz, fd = (double)dopplers[dop_begin + blockIdx.x];
  const float2 *__restrict__ sig = reinterpret_cast<const float2 *>(signal_any) + (size_t)scene * n_ms * spm;
  const uint8_t *__restrict__ ors = reinterpret_cast<const uint8_t *>(signal_any) + (size_t)scene * ors_bytes(n_ms * spm);
  fold_and_fft<PL>(lds, wblk, tw, sig, ors, packed, fd, FS, g * ms_per_group, ms_per_group, spm, tid);
  float2 *__restrict__ dst = work + (((size_t)scene * gridDim.x + blockIdx.x) * n_groups + g) * N;
  for (int k = tid; k < N; k += T) {
    const float2 z = lds[PL::idx(0, k)];
    dst[k] = make_float2(z.x, -z.y);
  }
}

// Last-pass store of the accumulation kernel: output `slot` of the thread is y = conj(Y) / N of one group; MODE 1 adds |y|,
// MODE 2 adds y conj(previous group's y) (mult / get_conj / add of gnss_codes.c:8-25) -- in registers, per slot.
template <int MODE, int R>   // R = legs per butterfly of the last pass: slot = butterfly * R + leg
struct AccStore {
  static constexpr bool SPLIT = true;
  static constexpr bool SLOT = true;
  float *acc;
  float2 *dacc, *prev;
  float inv_n;
  bool later;   // not the first group
  __device__ __forceinline__ void operator()(int q, int r, int, int, int, float2 Y) const
  {
    const int slot = q * R + r;
    const float2 y = make_float2(Y.x * inv_n, -Y.y * inv_n);
    if (MODE == 1) {
      acc[slot] += __builtin_amdgcn_sqrtf(y.x * y.x + y.y * y.y);
    } else {
      if (later) {
        dacc[slot].x += y.x * prev[slot].x - y.y * (-prev[slot].y);
        dacc[slot].y += y.x * (-prev[slot].y) + y.y * prev[slot].x;
      }
      prev[slot] = y;
    }
  }
};

// PRNs are taken kAccPrn at a time with the groups as the outer loop inside a chunk: a group's spectrum (32 KiB, from the
// workspace = HBM / Infinity Cache) is loaded once per chunk instead of once per PRN and held in registers next to the
// chunk's accumulators (per thread 16 code phases x kAccPrn PRNs).
// (measured, 64 scenes x 32 PRN x 41 bins x 10 groups: chunks of 2 / 1 -- 8.2 / 9.1 ms for MODE 1 / 2 -- against 8.5 with chunks of 1
// and spills with larger ones; the kernel is bound by its transforms, not by the workspace traffic)
template <int LOGN, int MODE> struct AccChunk { static constexpr int P = (LOGN <= 12 && MODE == 1) ? 2 : 1; };

template <int LOGN, int MODE>
__global__ __launch_bounds__(GPlan<LOGN>::T, GPlan<LOGN>::T >= 1024 ? 4 : 3) void gnss_accumulate_kernel(
    const float2 *__restrict__ work, const float2 *__restrict__ spectra, const float2 *__restrict__ tw, const int32_t *__restrict__ prns,
    int n_prn, int n_dop, int dop_begin, int n_groups, int sep, gnss_acq_result *__restrict__ all)
{
  using PL = GPlan<LOGN>;
  constexpr int N = PL::N, T = PL::T, P = AccChunk<LOGN, MODE>::P;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  unsigned char *xch = reinterpret_cast<unsigned char *>(lds + PL::LDS_ELEMS);
  RowRaw *pend = reinterpret_cast<RowRaw *>(xch + top2_scratch_bytes(T));
  float *magl = reinterpret_cast<float *>(lds);
  auto mag_idx = [](int m) { return m + ((m >> 4) << 2); };
  const int tid = threadIdx.x, d = dop_begin + blockIdx.x, scene = blockIdx.y;
  const float2 *__restrict__ zbase = work + ((size_t)scene * gridDim.x + blockIdx.x) * n_groups * N;
  const float inv_n = 1.0f / (float)N;
  const int m0 = tid * 16;
  typename std::conditional<(LOGN <= 12), TwPow<PL>, TwTable>::type twp;
  twp.preload(tw, tid);
  gnss_acq_result *rows = all + (size_t)scene * n_prn * n_dop;
  for (int p0 = 0; p0 < n_prn; p0 += P) {
    const int np = n_prn - p0 < P ? n_prn - p0 : P;
    // accumulators of the 16 code phases THIS thread's last pass produces (slot order of the transform): a group's row
    // never goes through LDS, only the accumulated row does, once per PRN
    float acc[P][16];       // MODE 1: running sums of magnitudes
    float2 dacc[MODE == 2 ? P : 1][16], prev[MODE == 2 ? P : 1][16];
#pragma unroll
    for (int pi = 0; pi < P; pi++) {
#pragma unroll
      for (int c = 0; c < 16; c++) {
        acc[pi][c] = 0.f;
        if (MODE == 2) { dacc[pi][c] = make_float2(0.f, 0.f); prev[pi][c] = make_float2(0.f, 0.f); }
      }
    }
    for (int g = 0; g < n_groups; g++) {
      int lt = tid;
      asm volatile("" : "+v"(lt));
      const __amdgpu_buffer_rsrc_t zs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(zbase + (size_t)g * N), 0, (int)(N * sizeof(float2)), 0x00020000);
      float2 zr[16];
      for_each_first_pass_input_split<PL>(lt, [&](int slot, int, int j, int roff) {
        const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(zs, j * 8, roff * 8, 0);
        zr[slot] = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
      });
#pragma unroll
      for (int pi = 0; pi < P; pi++) {
        if (pi < np) {
          twp.pin();
          const __amdgpu_buffer_rsrc_t cs =
              __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(spectra + (size_t)(p0 + pi) * N), 0, (int)(N * sizeof(float2)), 0x00020000);
          float2 v[16];
          for_each_first_pass_input_split<PL>(lt, [&](int slot, int, int j, int roff) {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(cs, j * 8, roff * 8, 0);
            v[slot] = cmul(zr[slot], make_float2(__uint_as_float(w[0]), __uint_as_float(w[1])));
          });
          // forward FFT of conj(Z) C = conj(y) * N: y = the output conjugated back and scaled
          AccStore<MODE, (1 << PL::lr(PL::NPASS - 1))> st{acc[pi], dacc[MODE == 2 ? pi : 0], prev[MODE == 2 ? pi : 0], inv_n, g > 0};
          // (the image may still be read by another wave's last pass of the previous transform: IMAGE_BUSY)
          fft_block_preloaded<PL, false, PL::NPASS - 1, true>(v, lds, twp, lt, st);
        }
      }
    }
#pragma unroll
    for (int pi = 0; pi < P; pi++) {
      if (pi < np) {
        const int p = p0 + pi;
        __syncthreads();  // the image (the last transform's reads, or the previous row's scan) is free
        for_each_last_pass_output<PL>(tid, [&](int slot, int, int e0, int roff) {
          magl[mag_idx(e0 + roff)] =
              (MODE == 1) ? acc[pi][slot] : sqrtf(dacc[MODE == 2 ? pi : 0][slot].x * dacc[MODE == 2 ? pi : 0][slot].x + dacc[MODE == 2 ? pi : 0][slot].y * dacc[MODE == 2 ? pi : 0][slot].y);
        });
        __syncthreads();
        auto mag = [&](int m) -> float { return magl[mag_idx(m)]; };
        const Top2 t = top2_scan<T, 16, true>(mag, N, sep, tid, xch);
        park_row(pend, t, mag, N, p, tid);
        flush_rows(pend, N, p, n_prn, d, prns, rows, n_dop, tid);
      }
    }
  }
}

// best Doppler bin per (scene, PRN): largest peak, lowest bin on ties (strict > scanning upwards)
__global__ void gnss_best_kernel(const gnss_acq_result *__restrict__ all, gnss_acq_result *__restrict__ best, int rows, int n_dop,
                                 int dop_begin, int dop_end)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const gnss_acq_result *a = all + (size_t)r * n_dop;
  gnss_acq_result b = a[dop_begin];
  for (int d = dop_begin + 1; d < dop_end; d++)
    if (a[d].peak > b.peak) b = a[d];
  best[r] = b;
}

// ------------------------------------------------------------------------------------------
// host side: plan + C ABI
// ------------------------------------------------------------------------------------------
}  // namespace armfft

using namespace armfft;

struct gnss_acq_plan {
  int device = 0;
  int logn = 0, fft_len = 0, spm = 0, n_ms = 0, spc = 0, n_prn = 0, n_dop = 0, pk_sep = 3;
  double fs_hz = 0.0;  // sample rate of the wipe-off: 1.023e6 * spc (C/A plans), 1000 * spm (replica plans)
  int32_t *prns_dev = nullptr;
  float *dops_dev = nullptr;
  int8_t *chips_dev = nullptr;
  float2 *spectra_dev = nullptr;
  const float2 *tw = nullptr;
  // staging for the host-pointer entry point
  float2 *sig_stage = nullptr;
  gnss_acq_result *all_stage = nullptr, *best_stage = nullptr;
  size_t stage_scenes = 0;
  std::mutex mu;
};

namespace {

// hipFuncSetAttribute once per kernel and device (the f32 / fixed / large launchers do the same): `done`
// is the caller's function-local static, one per template instantiation, zero-initialised; a racing
// second call is harmless.
template <class K>
hipError_t lds_attr_once(K kern, size_t bytes, std::atomic<bool> *done, int dev)
{
  if (bytes <= 48 * 1024 || (dev >= 0 && dev < 16 && done[dev].load(std::memory_order_relaxed))) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess && dev >= 0 && dev < 16) done[dev].store(true, std::memory_order_relaxed);
  return e;
}

template <int LOGN>
hipError_t launch_spectrum(const gnss_acq_plan *pl, hipStream_t s)
{
  using PL = GPlan<LOGN>;
  auto k = gnss_code_spectrum_kernel<LOGN>;
  static std::atomic<bool> done[16];
  hipError_t e = lds_attr_once(k, PL::LDS_BYTES, done, pl->device);
  if (e != hipSuccess) return e;
  k<<<pl->n_prn, PL::T, PL::LDS_BYTES, s>>>(pl->chips_dev, pl->spectra_dev, pl->tw, pl->spm, pl->spc);
  return hipGetLastError();
}

template <int LOGN>
hipError_t launch_acquire(const gnss_acq_plan *pl, const void *sig, int packed, uint32_t n_scenes, int d0, int d1, gnss_acq_result *all,
                          hipStream_t s)
{
  using PL = GPlan<LOGN>;
  const size_t lds = PL::LDS_BYTES + top2_scratch_bytes(PL::T) + kMaxMs * sizeof(float2) + pend_bytes();
  auto k = gnss_acquire_kernel<LOGN>;
  static std::atomic<bool> done[16];
  hipError_t e = lds_attr_once(k, lds, done, pl->device);
  if (e != hipSuccess) return e;
  note_kernel("gnss_acquire_kernel<%d>", false, LOGN);
  k<<<dim3(d1 - d0, n_scenes), PL::T, lds, s>>>(sig, packed, pl->spectra_dev, pl->tw, pl->dops_dev, pl->prns_dev, pl->n_prn, pl->n_dop, d0,
                                                pl->n_ms, pl->spm, pl->fs_hz, pl->pk_sep, all);
  return hipGetLastError();
}

template <int LOGN>
hipError_t launch_groups(const gnss_acq_plan *pl, const void *sig, int packed, uint32_t n_scenes, int d0, int d1, int mode, int n_groups,
                         float2 *work, gnss_acq_result *all, hipStream_t s)
{
  using PL = GPlan<LOGN>;
  const size_t lds_a = PL::LDS_BYTES + kMaxMs * sizeof(float2);
  const size_t lds_b = PL::LDS_BYTES + top2_scratch_bytes(PL::T) + pend_bytes();
  auto ka = gnss_group_spectra_kernel<LOGN>;
  static std::atomic<bool> done_a[16], done_b1[16], done_b2[16];
  hipError_t e = lds_attr_once(ka, lds_a, done_a, pl->device);
  if (e != hipSuccess) return e;
  note_kernel("gnss_group_spectra_kernel<%d>", false, LOGN);
  ka<<<dim3(d1 - d0, n_scenes, n_groups), PL::T, lds_a, s>>>(sig, packed, pl->tw, pl->dops_dev, d0, pl->n_ms, pl->n_ms / n_groups, pl->spm,
                                                             pl->fs_hz, work);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (mode == 1) {
    auto kb = gnss_accumulate_kernel<LOGN, 1>;
    if ((e = lds_attr_once(kb, lds_b, done_b1, pl->device)) != hipSuccess) return e;
    note_kernel("gnss_accumulate_kernel<%d, 1>", true, LOGN);
    kb<<<dim3(d1 - d0, n_scenes), PL::T, lds_b, s>>>(work, pl->spectra_dev, pl->tw, pl->prns_dev, pl->n_prn, pl->n_dop, d0, n_groups, pl->pk_sep, all);
  } else {
    auto kb = gnss_accumulate_kernel<LOGN, 2>;
    if ((e = lds_attr_once(kb, lds_b, done_b2, pl->device)) != hipSuccess) return e;
    note_kernel("gnss_accumulate_kernel<%d, 2>", true, LOGN);
    kb<<<dim3(d1 - d0, n_scenes), PL::T, lds_b, s>>>(work, pl->spectra_dev, pl->tw, pl->prns_dev, pl->n_prn, pl->n_dop, d0, n_groups, pl->pk_sep, all);
  }
  return hipGetLastError();
}

#define GNSS_DISPATCH(fn, logn, ...)                 \
  [&]() -> hipError_t {                              \
    switch (logn) {                                  \
      case 10: return fn<10>(__VA_ARGS__);           \
      case 11: return fn<11>(__VA_ARGS__);           \
      case 12: return fn<12>(__VA_ARGS__);           \
      case 13: return fn<13>(__VA_ARGS__);           \
      case 14: return fn<14>(__VA_ARGS__);           \
    }                                                \
    return hipErrorInvalidValue;                     \
  }()

bool plan_on_device(const gnss_acq_plan *pl, const char *who)
{
  int dev = -1;
  if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return false;
  if (dev != pl->device) {
    set_error("%s: the plan was created on device %d, the current device is %d", who, pl->device, dev);
    return false;
  }
  return true;
}

}  // namespace

extern "C" {

void gnss_acq_plan_destroy(gnss_acq_plan *pl)
{
  if (!pl) return;
  (void)hipFree(pl->prns_dev);
  (void)hipFree(pl->dops_dev);
  (void)hipFree(pl->chips_dev);
  (void)hipFree(pl->spectra_dev);
  (void)hipFree(pl->sig_stage);
  (void)hipFree(pl->all_stage);
  (void)hipFree(pl->best_stage);
  delete pl;
}

arm_status gnss_acq_plan_create(gnss_acq_plan **out, uint32_t fft_len, uint32_t samples_per_ms, uint32_t n_ms,
                                uint32_t samples_per_chip, const int32_t *prns, uint32_t n_prn, const float *dopplers_hz,
                                uint32_t n_dop)
{
  if (!out || !prns || !dopplers_hz) { set_error("gnss_acq_plan_create: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  *out = nullptr;
  int logn = -1;
  for (int l = 10; l <= 14; l++)
    if (fft_len == (1u << l)) logn = l;
  if (logn < 0) { set_error("gnss_acq_plan_create: fft_len %u is not a power of two in 1024..16384", fft_len); return ARM_MATH_ARGUMENT_ERROR; }
  if (samples_per_ms == 0 || samples_per_ms > fft_len || n_ms == 0 || n_ms > (uint32_t)kMaxMs || samples_per_chip == 0 ||
      n_prn == 0 || n_prn > 64 || n_dop == 0 || n_dop > 32768) {
    set_error("gnss_acq_plan_create: bad geometry (samples_per_ms %u <= fft_len, n_ms 1..%d, n_prn 1..64, n_dop 1..32768)",
              samples_per_ms, kMaxMs);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  for (uint32_t i = 0; i < n_prn; i++)
    if (prns[i] < 1 || prns[i] > 37) { set_error("gnss_acq_plan_create: PRN %d outside 1..37", prns[i]); return ARM_MATH_ARGUMENT_ERROR; }
  gnss_acq_plan *pl = new gnss_acq_plan;
  if (!hip_ok(hipGetDevice(&pl->device), "hipGetDevice")) { delete pl; return ARM_MATH_ARGUMENT_ERROR; }
  pl->logn = logn; pl->fft_len = (int)fft_len; pl->spm = (int)samples_per_ms; pl->n_ms = (int)n_ms; pl->spc = (int)samples_per_chip;
  pl->fs_hz = 1.023e6 * (double)samples_per_chip;
  pl->n_prn = (int)n_prn; pl->n_dop = (int)n_dop;
  pl->tw = dev_twiddle_f32(logn);
  bool ok = pl->tw != nullptr;
  ok = ok && hip_ok(hipMalloc((void **)&pl->prns_dev, sizeof(int32_t) * n_prn), "hipMalloc(prns)");
  ok = ok && hip_ok(hipMalloc((void **)&pl->dops_dev, sizeof(float) * n_dop), "hipMalloc(dopplers)");
  ok = ok && hip_ok(hipMalloc((void **)&pl->chips_dev, (size_t)n_prn * kCodeLen), "hipMalloc(chips)");
  ok = ok && hip_ok(hipMalloc((void **)&pl->spectra_dev, sizeof(float2) * (size_t)n_prn * fft_len), "hipMalloc(code spectra)");
  ok = ok && hip_ok(hipMemcpy(pl->prns_dev, prns, sizeof(int32_t) * n_prn, hipMemcpyHostToDevice), "hipMemcpy(prns)");
  ok = ok && hip_ok(hipMemcpy(pl->dops_dev, dopplers_hz, sizeof(float) * n_dop, hipMemcpyHostToDevice), "hipMemcpy(dopplers)");
  if (ok) {
    gnss_code_kernel<<<n_prn, 256, 0, nullptr>>>(pl->prns_dev, pl->chips_dev);
    ok = hip_ok(hipGetLastError(), "gnss_code_kernel") && hip_ok(GNSS_DISPATCH(launch_spectrum, logn, pl, nullptr), "code spectra") &&
         hip_ok(hipStreamSynchronize(nullptr), "plan synchronise");
  }
  if (!ok) { gnss_acq_plan_destroy(pl); return ARM_MATH_ARGUMENT_ERROR; }
  *out = pl;
  return ARM_MATH_SUCCESS;
}

arm_status gnss_acq_plan_set_peak_separation(gnss_acq_plan *pl, int32_t pk_sep)
{
  if (!pl || pk_sep < 0) { set_error("gnss_acq_plan_set_peak_separation: bad argument"); return ARM_MATH_ARGUMENT_ERROR; }
  pl->pk_sep = pk_sep;
  return ARM_MATH_SUCCESS;
}

const float *gnss_acq_plan_code_spectra_dev(const gnss_acq_plan *pl) { return pl ? reinterpret_cast<const float *>(pl->spectra_dev) : nullptr; }

static arm_status acquire_any(const gnss_acq_plan *pl, const void *signal_dev, int packed, uint32_t n_scenes, uint32_t dop_begin,
                              uint32_t dop_end, gnss_acq_result *all_dev, gnss_acq_result *best_dev, void *stream, const char *who)
{
  if (!pl || !signal_dev || !all_dev) { set_error("%s: null argument", who); return ARM_MATH_ARGUMENT_ERROR; }
  if (dop_begin > dop_end || dop_end > (uint32_t)pl->n_dop) {
    set_error("%s: Doppler range [%u, %u) outside the plan's %d bins", who, dop_begin, dop_end, pl->n_dop);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if (n_scenes > 65535) { set_error("%s: at most 65535 scenes per call", who); return ARM_MATH_ARGUMENT_ERROR; }
  if (n_scenes == 0 || dop_begin == dop_end) return ARM_MATH_SUCCESS;
  if (!plan_on_device(pl, who)) return ARM_MATH_ARGUMENT_ERROR;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = GNSS_DISPATCH(launch_acquire, pl->logn, pl, signal_dev, packed, n_scenes, (int)dop_begin, (int)dop_end, all_dev, s);
  if (!hip_ok(e, "gnss_acquire_kernel launch")) return ARM_MATH_ARGUMENT_ERROR;
  if (best_dev) {
    const int rows = (int)n_scenes * pl->n_prn;
    gnss_best_kernel<<<(rows + 255) / 256, 256, 0, s>>>(all_dev, best_dev, rows, pl->n_dop, (int)dop_begin, (int)dop_end);
    if (!hip_ok(hipGetLastError(), "gnss_best_kernel launch")) return ARM_MATH_ARGUMENT_ERROR;
  }
  return ARM_MATH_SUCCESS;
}

arm_status gnss_acquire_dev(const gnss_acq_plan *pl, const float *signal_dev, uint32_t n_scenes, uint32_t dop_begin, uint32_t dop_end,
                            gnss_acq_result *all_dev, gnss_acq_result *best_dev, void *stream)
{
  return acquire_any(pl, signal_dev, 0, n_scenes, dop_begin, dop_end, all_dev, best_dev, stream, "gnss_acquire_dev");
}

arm_status gnss_acquire_ors_dev(const gnss_acq_plan *pl, const uint8_t *ors_dev, uint32_t n_scenes, uint32_t dop_begin, uint32_t dop_end,
                                gnss_acq_result *all_dev, gnss_acq_result *best_dev, void *stream)
{
  return acquire_any(pl, ors_dev, 1, n_scenes, dop_begin, dop_end, all_dev, best_dev, stream, "gnss_acquire_ors_dev");
}

size_t gnss_acq_workspace_bytes(const gnss_acq_plan *pl, uint32_t n_scenes, uint32_t n_dop_bins, uint32_t n_groups)
{
  return pl ? (size_t)n_scenes * n_dop_bins * n_groups * (size_t)pl->fft_len * sizeof(float2) : 0;
}

arm_status gnss_acquire_groups_dev(const gnss_acq_plan *pl, const float *signal_dev, uint32_t n_scenes, uint32_t dop_begin, uint32_t dop_end,
                                   int32_t mode, uint32_t n_groups, void *work_dev, gnss_acq_result *all_dev, gnss_acq_result *best_dev,
                                   void *stream)
{
  const char *who = "gnss_acquire_groups_dev";
  if (!pl || !signal_dev || !all_dev || !work_dev) { set_error("%s: null argument", who); return ARM_MATH_ARGUMENT_ERROR; }
  if (mode != GNSS_ACQ_NONCOHERENT && mode != GNSS_ACQ_DIFFERENTIAL) { set_error("%s: mode must be GNSS_ACQ_NONCOHERENT or GNSS_ACQ_DIFFERENTIAL", who); return ARM_MATH_ARGUMENT_ERROR; }
  if (n_groups == 0 || n_groups > 65535 || (uint32_t)pl->n_ms % n_groups != 0 || (mode == GNSS_ACQ_DIFFERENTIAL && n_groups < 2)) {
    set_error("%s: n_groups must divide the plan's %d ms (and be >= 2 for the differential mode)", who, pl->n_ms);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if (dop_begin > dop_end || dop_end > (uint32_t)pl->n_dop) {
    set_error("%s: Doppler range [%u, %u) outside the plan's %d bins", who, dop_begin, dop_end, pl->n_dop);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if (n_scenes > 65535) { set_error("%s: at most 65535 scenes per call", who); return ARM_MATH_ARGUMENT_ERROR; }
  if (n_scenes == 0 || dop_begin == dop_end) return ARM_MATH_SUCCESS;
  if (!plan_on_device(pl, who)) return ARM_MATH_ARGUMENT_ERROR;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = GNSS_DISPATCH(launch_groups, pl->logn, pl, signal_dev, 0, n_scenes, (int)dop_begin, (int)dop_end, (int)mode, (int)n_groups,
                               reinterpret_cast<float2 *>(work_dev), all_dev, s);
  if (!hip_ok(e, "gnss group kernels launch")) return ARM_MATH_ARGUMENT_ERROR;
  if (best_dev) {
    const int rows = (int)n_scenes * pl->n_prn;
    gnss_best_kernel<<<(rows + 255) / 256, 256, 0, s>>>(all_dev, best_dev, rows, pl->n_dop, (int)dop_begin, (int)dop_end);
    if (!hip_ok(hipGetLastError(), "gnss_best_kernel launch")) return ARM_MATH_ARGUMENT_ERROR;
  }
  return ARM_MATH_SUCCESS;
}

arm_status gnss_decode_ors_iq_dev(const uint8_t *data_dev, uint32_t byte_length, float *iqs_dev, void *stream)
{
  if (!data_dev || !iqs_dev) { set_error("gnss_decode_ors_iq_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (byte_length == 0) return ARM_MATH_SUCCESS;
  if (byte_length > 0x7FFFFFFFu) { set_error("gnss_decode_ors_iq_dev: at most 2^31-1 bytes per call"); return ARM_MATH_ARGUMENT_ERROR; }
  const uint32_t n = 2 * byte_length;
  gnss_decode_ors_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(data_dev, n, reinterpret_cast<float2 *>(iqs_dev));
  return hip_ok(hipGetLastError(), "gnss_decode_ors_kernel launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// staging of the host-pointer entry points (caller holds pl->mu, the plan's device is current)
static bool ensure_stage(gnss_acq_plan *pl, uint32_t n_scenes)
{
  if (pl->stage_scenes >= n_scenes) return true;
  const size_t sig_elems = (size_t)n_scenes * pl->n_ms * pl->spm, rows = (size_t)n_scenes * pl->n_prn;
  (void)hipFree(pl->sig_stage); (void)hipFree(pl->all_stage); (void)hipFree(pl->best_stage);
  pl->sig_stage = nullptr; pl->all_stage = nullptr; pl->best_stage = nullptr; pl->stage_scenes = 0;
  if (!hip_ok(hipMalloc((void **)&pl->sig_stage, sizeof(float2) * sig_elems), "hipMalloc(signal staging)") ||
      !hip_ok(hipMalloc((void **)&pl->all_stage, sizeof(gnss_acq_result) * rows * pl->n_dop), "hipMalloc(cell staging)") ||
      !hip_ok(hipMalloc((void **)&pl->best_stage, sizeof(gnss_acq_result) * rows), "hipMalloc(result staging)"))
    return false;
  pl->stage_scenes = n_scenes;
  return true;
}

arm_status gnss_acquire(gnss_acq_plan *pl, const float *signal, uint32_t n_scenes, gnss_acq_result *best, gnss_acq_result *all)
{
  if (!pl || !signal || !best) { set_error("gnss_acquire: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (n_scenes == 0) return ARM_MATH_SUCCESS;
  if (!plan_on_device(pl, "gnss_acquire")) return ARM_MATH_ARGUMENT_ERROR;
  std::lock_guard<std::mutex> lk(pl->mu);
  const size_t sig_elems = (size_t)n_scenes * pl->n_ms * pl->spm, rows = (size_t)n_scenes * pl->n_prn;
  if (!ensure_stage(pl, n_scenes)) return ARM_MATH_ARGUMENT_ERROR;
  if (!hip_ok(hipMemcpy(pl->sig_stage, signal, sizeof(float2) * sig_elems, hipMemcpyHostToDevice), "hipMemcpy(H2D)")) return ARM_MATH_ARGUMENT_ERROR;
  arm_status st = gnss_acquire_dev(pl, reinterpret_cast<const float *>(pl->sig_stage), n_scenes, 0, (uint32_t)pl->n_dop, pl->all_stage,
                                   pl->best_stage, nullptr);
  if (st != ARM_MATH_SUCCESS) return st;
  if (!hip_ok(hipMemcpy(best, pl->best_stage, sizeof(gnss_acq_result) * rows, hipMemcpyDeviceToHost), "hipMemcpy(D2H)")) return ARM_MATH_ARGUMENT_ERROR;
  if (all && !hip_ok(hipMemcpy(all, pl->all_stage, sizeof(gnss_acq_result) * rows * pl->n_dop, hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
    return ARM_MATH_ARGUMENT_ERROR;
  return ARM_MATH_SUCCESS;
}

// Multi-GPU search without Python or RCCL: the C caller of the reference's read_ors loop (CMSIS/CMSIS.cpp:706-737) hands over
// one plan per GPU (each created with that GPU current; identical geometry, PRN list and Doppler grid).  Plan i searches the
// i-th contiguous share of the Doppler bins on ITS device from its own host thread (hipSetDevice is per thread); the signal
// is replicated to every device (327 KB per scene), nothing crosses between GPUs; the per-(scene, PRN) winners come back to
// the host and are merged by the rule of gnss_best_kernel -- largest peak, lowest Doppler bin on ties -- which makes the
// result identical to gnss_acquire on one plan.  Several plans may live on one device (they then share it).
arm_status gnss_acquire_multi(gnss_acq_plan *const *plans, uint32_t n_plans, const float *signal, uint32_t n_scenes, gnss_acq_result *best)
{
  const char *who = "gnss_acquire_multi";
  if (!plans || n_plans == 0 || !signal || !best) { set_error("%s: null argument", who); return ARM_MATH_ARGUMENT_ERROR; }
  if (n_plans > 64) { set_error("%s: at most 64 plans", who); return ARM_MATH_ARGUMENT_ERROR; }
  for (uint32_t i = 0; i < n_plans; i++) {
    const gnss_acq_plan *a = plans[0], *b = plans[i];
    if (!b) { set_error("%s: plan %u is null", who, i); return ARM_MATH_ARGUMENT_ERROR; }
    for (uint32_t j = 0; j < i; j++)
      if (plans[j] == b) { set_error("%s: plan %u is listed twice", who, i); return ARM_MATH_ARGUMENT_ERROR; }
    if (b->fft_len != a->fft_len || b->spm != a->spm || b->n_ms != a->n_ms || b->n_prn != a->n_prn || b->n_dop != a->n_dop || b->pk_sep != a->pk_sep) {
      set_error("%s: plan %u differs in geometry from plan 0", who, i);
      return ARM_MATH_ARGUMENT_ERROR;
    }
  }
  if (n_scenes == 0) return ARM_MATH_SUCCESS;
  if (n_scenes > 65535) { set_error("%s: at most 65535 scenes per call", who); return ARM_MATH_ARGUMENT_ERROR; }
  const int n_dop = plans[0]->n_dop;
  const size_t rows = (size_t)n_scenes * plans[0]->n_prn, sig_elems = (size_t)n_scenes * plans[0]->n_ms * plans[0]->spm;
  int prev_dev = 0;
  (void)hipGetDevice(&prev_dev);
  std::vector<std::vector<gnss_acq_result>> part(n_plans);
  std::vector<std::string> errs(n_plans);
  std::vector<std::thread> th;
  for (uint32_t i = 0; i < n_plans; i++) {
    // contiguous shares, the first n_dop % n_plans plans take one bin more (fft_amd/dist.py: doppler_range)
    const int q = n_dop / (int)n_plans, r = n_dop % (int)n_plans;
    const int d0 = (int)i * q + std::min((int)i, r), d1 = d0 + q + ((int)i < r ? 1 : 0);
    if (d1 <= d0) continue;  // more plans than bins: this one has nothing to search
    th.emplace_back([&, i, d0, d1] {
      gnss_acq_plan *pl = plans[i];
      bool ok = hip_ok(hipSetDevice(pl->device), "hipSetDevice");
      if (ok) {
        std::lock_guard<std::mutex> lk(pl->mu);
        ok = ensure_stage(pl, n_scenes) &&
             hip_ok(hipMemcpy(pl->sig_stage, signal, sizeof(float2) * sig_elems, hipMemcpyHostToDevice), "hipMemcpy(H2D)") &&
             gnss_acquire_dev(pl, reinterpret_cast<const float *>(pl->sig_stage), n_scenes, (uint32_t)d0, (uint32_t)d1, pl->all_stage, pl->best_stage,
                              nullptr) == ARM_MATH_SUCCESS;
        if (ok) {
          part[i].resize(rows);
          ok = hip_ok(hipMemcpy(part[i].data(), pl->best_stage, sizeof(gnss_acq_result) * rows, hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
        }
      }
      if (!ok) errs[i] = arm_fft_mi355_last_error();  // thread-local message, handed to the calling thread below
    });
  }
  for (auto &t : th) t.join();
  (void)hipSetDevice(prev_dev);
  for (uint32_t i = 0; i < n_plans; i++)
    if (!errs[i].empty()) { set_error("%s: plan %u (device %d): %s", who, i, plans[i]->device, errs[i].c_str()); return ARM_MATH_ARGUMENT_ERROR; }
  // merge: plans are in increasing bin order, so "strictly larger takes over" keeps the lowest bin on ties
  bool first = true;
  for (uint32_t i = 0; i < n_plans; i++) {
    if (part[i].empty()) continue;
    if (first) { memcpy(best, part[i].data(), sizeof(gnss_acq_result) * rows); first = false; continue; }
    for (size_t k = 0; k < rows; k++)
      if (part[i][k].peak > best[k].peak) best[k] = part[i][k];
  }
  return ARM_MATH_SUCCESS;
}

// ---- per-function probes (device pointers) ----
// C/A chips of PRN 1..37 on the current device: generated once per device (37 x 1023 bytes, kept for
// the life of the process) so that the probes below neither allocate nor free per call.
static arm_status chips_for(int prn, const int8_t **chips)
{
  if (prn < 1 || prn > 37) { set_error("PRN %d outside 1..37", prn); return ARM_MATH_ARGUMENT_ERROR; }
  static std::mutex mu;
  static int8_t *cache[16];
  int dev = -1;
  if (!hip_ok(hipGetDevice(&dev), "hipGetDevice") || dev < 0 || dev >= 16) { if (dev >= 16) set_error("device index %d out of range", dev); return ARM_MATH_ARGUMENT_ERROR; }
  std::lock_guard<std::mutex> lk(mu);
  if (!cache[dev]) {
    int32_t host_prns[37];
    for (int i = 0; i < 37; i++) host_prns[i] = i + 1;
    int32_t *p = nullptr;
    int8_t *c = nullptr;
    bool ok = hip_ok(hipMalloc((void **)&p, sizeof(host_prns)), "hipMalloc") && hip_ok(hipMalloc((void **)&c, 37 * (size_t)kCodeLen), "hipMalloc") &&
              hip_ok(hipMemcpy(p, host_prns, sizeof(host_prns), hipMemcpyHostToDevice), "hipMemcpy");
    if (ok) {
      gnss_code_kernel<<<37, 256, 0, nullptr>>>(p, c);
      ok = hip_ok(hipGetLastError(), "gnss_code_kernel") && hip_ok(hipStreamSynchronize(nullptr), "synchronise");
    }
    (void)hipFree(p);
    if (!ok) { (void)hipFree(c); return ARM_MATH_ARGUMENT_ERROR; }
    cache[dev] = c;
  }
  *chips = cache[dev] + (size_t)(prn - 1) * kCodeLen;
  return ARM_MATH_SUCCESS;
}

arm_status gnss_l5_generate_I_dev(int32_t prn, uint32_t n_chips, int8_t *out_dev)
{
  if (!out_dev) { set_error("gnss_l5_generate_I_dev: null output"); return ARM_MATH_ARGUMENT_ERROR; }
  if (prn <= 0 || prn > 37) { set_error("gnss_l5_generate_I_dev: PRN %d outside 1..37", prn); return ARM_MATH_ARGUMENT_ERROR; }  // :155
  if (n_chips == 0) return ARM_MATH_SUCCESS;
  gnss_l5_code_kernel<<<1, 256, 0, nullptr>>>(prn, n_chips, out_dev);
  return hip_ok(hipGetLastError(), "gnss_l5_code_kernel") && hip_ok(hipStreamSynchronize(nullptr), "synchronise") ? ARM_MATH_SUCCESS
                                                                                                                 : ARM_MATH_ARGUMENT_ERROR;
}

arm_status gnss_e1b_cboc_replica_dev(const int8_t *codes_dev, uint32_t n_codes, float doppler, uint32_t size, float *replica_dev, void *stream)
{
  if (!codes_dev || !replica_dev) { set_error("gnss_e1b_cboc_replica_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (n_codes > 65535) { set_error("gnss_e1b_cboc_replica_dev: at most 65535 codes per call"); return ARM_MATH_ARGUMENT_ERROR; }
  if (n_codes == 0 || size == 0) return ARM_MATH_SUCCESS;
  gnss_e1b_replica_kernel<<<n_codes, 256, 0, (hipStream_t)stream>>>(codes_dev, doppler, size, reinterpret_cast<float2 *>(replica_dev));
  return hip_ok(hipGetLastError(), "gnss_e1b_replica_kernel launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// hex_nibble + hexstream_to_e1b_chips / _e5a_chips / _e5a_qp_chips (gnss_codes.c:234-239, :276-325; they differ only in
// the code length): host-side text work on a few KiB, nothing for a GPU to do
int gnss_hexstream_to_chips(const char *hex_stream, int8_t *chips, int32_t code_len)
{
  if (!hex_stream || !chips || code_len <= 0) { set_error("gnss_hexstream_to_chips: bad argument"); return -1; }
  int32_t chip_idx = 0;
  for (const char *p = hex_stream; *p && chip_idx < code_len; ++p) {
    const int c = (unsigned char)*p;
    int v;
    if (c >= '0' && c <= '9') v = c - '0';
    else if (c >= 'A' && c <= 'F') v = 10 + (c - 'A');
    else if (c >= 'a' && c <= 'f') v = 10 + (c - 'a');
    else continue;  // separators are skipped
    for (int b = 3; b >= 0 && chip_idx < code_len; --b) chips[chip_idx++] = ((v >> b) & 1) ? (int8_t)-1 : (int8_t)+1;  // logic 1 -> -1 (:243-245)
  }
  return chip_idx == code_len ? 0 : -1;
}

arm_status gnss_acq_plan_create_replicas(gnss_acq_plan **out, uint32_t fft_len, uint32_t samples_per_ms, uint32_t n_ms,
                                         const float *replicas_dev, const int32_t *prn_labels, uint32_t n_prn, const float *dopplers_hz,
                                         uint32_t n_dop)
{
  if (!out || !replicas_dev || !prn_labels || !dopplers_hz) { set_error("gnss_acq_plan_create_replicas: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  *out = nullptr;
  int logn = -1;
  for (int l = 10; l <= 14; l++)
    if (fft_len == (1u << l)) logn = l;
  if (logn < 0) { set_error("gnss_acq_plan_create_replicas: fft_len %u is not a power of two in 1024..16384", fft_len); return ARM_MATH_ARGUMENT_ERROR; }
  if (samples_per_ms == 0 || samples_per_ms > fft_len || n_ms == 0 || n_ms > (uint32_t)kMaxMs || n_prn == 0 || n_prn > 64 || n_dop == 0 ||
      n_dop > 32768) {
    set_error("gnss_acq_plan_create_replicas: bad geometry (samples_per_ms %u <= fft_len, n_ms 1..%d, n_prn 1..64, n_dop 1..32768)",
              samples_per_ms, kMaxMs);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  gnss_acq_plan *pl = new gnss_acq_plan;
  if (!hip_ok(hipGetDevice(&pl->device), "hipGetDevice")) { delete pl; return ARM_MATH_ARGUMENT_ERROR; }
  pl->logn = logn; pl->fft_len = (int)fft_len; pl->spm = (int)samples_per_ms; pl->n_ms = (int)n_ms; pl->spc = 0;
  pl->fs_hz = 1000.0 * (double)samples_per_ms;   // one millisecond = samples_per_ms samples
  pl->n_prn = (int)n_prn; pl->n_dop = (int)n_dop;
  pl->tw = dev_twiddle_f32(logn);
  const size_t spec_bytes = sizeof(float2) * (size_t)n_prn * fft_len;
  bool ok = pl->tw != nullptr;
  ok = ok && hip_ok(hipMalloc((void **)&pl->prns_dev, sizeof(int32_t) * n_prn), "hipMalloc(prns)");
  ok = ok && hip_ok(hipMalloc((void **)&pl->dops_dev, sizeof(float) * n_dop), "hipMalloc(dopplers)");
  ok = ok && hip_ok(hipMalloc((void **)&pl->spectra_dev, spec_bytes), "hipMalloc(code spectra)");
  ok = ok && hip_ok(hipMemcpy(pl->prns_dev, prn_labels, sizeof(int32_t) * n_prn, hipMemcpyHostToDevice), "hipMemcpy(prns)");
  ok = ok && hip_ok(hipMemcpy(pl->dops_dev, dopplers_hz, sizeof(float) * n_dop, hipMemcpyHostToDevice), "hipMemcpy(dopplers)");
  // spectra = FFT(replica), the same forward transform the acquisition kernel runs
  ok = ok && hip_ok(hipMemcpyAsync(pl->spectra_dev, replicas_dev, spec_bytes, hipMemcpyDeviceToDevice, nullptr), "hipMemcpy(replicas)");
  ok = ok && hip_ok(launch_cfft_f32(logn, pl->spectra_dev, pl->tw, n_prn, false, 0, nullptr), "replica spectra") &&
       hip_ok(hipStreamSynchronize(nullptr), "plan synchronise");
  if (!ok) { gnss_acq_plan_destroy(pl); return ARM_MATH_ARGUMENT_ERROR; }
  *out = pl;
  return ARM_MATH_SUCCESS;
}

arm_status gnss_get_code_dev(int32_t num, int32_t samples_per_chip, int32_t prn, int32_t *out_dev)
{
  if (!out_dev || num < 0) { set_error("gnss_get_code_dev: bad argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (samples_per_chip < 1) samples_per_chip = 1;  // gnss_codes.c:1010
  if (num == 0) return ARM_MATH_SUCCESS;
  const int8_t *chips = nullptr;
  arm_status st = chips_for(prn, &chips);
  if (st != ARM_MATH_SUCCESS) return st;
  const int total = num * samples_per_chip;
  gnss_expand_code_kernel<<<(total + 255) / 256, 256, 0, nullptr>>>(chips, out_dev, total, samples_per_chip);
  bool ok = hip_ok(hipGetLastError(), "gnss_expand_code_kernel") && hip_ok(hipStreamSynchronize(nullptr), "synchronise");
  return ok ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

arm_status gnss_synth_gps_prn_dev(int32_t prn, float doppler, uint32_t size, float *replica_dev, int32_t samples_per_chip)
{
  if (!replica_dev || samples_per_chip < 1) { set_error("gnss_synth_gps_prn_dev: bad argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (size == 0) return ARM_MATH_SUCCESS;
  const int8_t *chips = nullptr;
  arm_status st = chips_for(prn, &chips);
  if (st != ARM_MATH_SUCCESS) return st;
  // mix_prn's own arithmetic (gnss_codes.c:1035-1042): FS = 1.023e6 * spc, dth = 2.0 * PI * f / FS with the FLOAT literal PI
  const double FS = 1.023e6 * samples_per_chip, dth = 2.0 * (double)kRefPiF * (double)doppler / FS;
  gnss_replica_kernel<<<(size + 255) / 256, 256, 0, nullptr>>>(chips, reinterpret_cast<float2 *>(replica_dev), (int)size, samples_per_chip,
                                                               dth, 0.0 * (double)kRefPiF / 180.0);
  bool ok = hip_ok(hipGetLastError(), "gnss_replica_kernel") && hip_ok(hipStreamSynchronize(nullptr), "synchronise");
  return ok ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// mix_prn (gnss_codes.c:1032-1063) on a caller-supplied code: signal[n] = code[n] e^{j (phase + 2 PI f n / FS)}, FS = 1.023e6 spc
arm_status gnss_mix_prn_dev(const int32_t *code_dev, double doppler_hz, double phase_deg, float *signal_dev, uint32_t size, int32_t spc,
                            void *stream)
{
  if (!code_dev || !signal_dev || spc < 1 || size > 0x7fffffffu) { set_error("gnss_mix_prn_dev: bad argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (size == 0) return ARM_MATH_SUCCESS;
  const double FS = 1.023e6 * spc;                                                   // :1035
  const double dth = 2.0 * (double)kRefPiF * doppler_hz / FS;                        // :1037, PI = 3.14159265358979f
  const double phi0 = phase_deg * (double)kRefPiF / 180.0;                           // :1041-1042
  gnss_mix_code_kernel<<<(size + 255) / 256, 256, 0, (hipStream_t)stream>>>(code_dev, reinterpret_cast<float2 *>(signal_dev), (int)size, dth, phi0);
  return hip_ok(hipGetLastError(), "gnss_mix_code_kernel") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// make_replica (quasi_pilot_tests.c:153-181): carrier at `doppler` on a caller-supplied code, initial phase 0.0001 degrees
arm_status gnss_make_replica_dev(const int32_t *code_dev, uint32_t size, float doppler, float samp_freq, float *out_dev, void *stream)
{
  if (!code_dev || !out_dev || size > 0x7fffffffu) { set_error("gnss_make_replica_dev: bad argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (size == 0) return ARM_MATH_SUCCESS;
  const double dth = 2.0 * M_PI * (double)doppler / (double)samp_freq;               // :157
  const double phi0 = 0.0001 * (double)M_PI / 180.0f;                                // :155, :160-161
  gnss_mix_code_kernel<<<(size + 255) / 256, 256, 0, (hipStream_t)stream>>>(code_dev, reinterpret_cast<float2 *>(out_dev), (int)size, dth, phi0);
  return hip_ok(hipGetLastError(), "gnss_mix_code_kernel") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// rotate_fwd (quasi_pilot_tests.c:138-151), out of place on device pointers; 0 <= offset <= size (the reference indexes outside
// the array for any other offset)
arm_status gnss_rotate_fwd_dev(const int32_t *in_dev, int32_t *out_dev, uint32_t size, int32_t offset, void *stream)
{
  if (!in_dev || !out_dev || in_dev == out_dev || size > 0x7fffffffu || offset < 0 || (uint32_t)offset > size) {
    set_error("gnss_rotate_fwd_dev: bad argument (0 <= offset <= size, out of place)");
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if (size == 0) return ARM_MATH_SUCCESS;
  gnss_rotate_kernel<<<(size + 255) / 256, 256, 0, (hipStream_t)stream>>>(in_dev, out_dev, (int)size, offset);
  return hip_ok(hipGetLastError(), "gnss_rotate_kernel") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// recover_prn_phase_deg_with_doppler (CMSIS/CMSIS.cpp:362-434) for n_sat satellites at once: satellite k searches
// signal_dev + k * signal_stride (size complex samples; stride 0 = every satellite searches the same millisecond) with
// code k (size ints) around code_offsets_dev[k] / dopplers_dev[k]
arm_status gnss_refine_dev(const float *signal_dev, uint32_t signal_stride, uint32_t size, const int32_t *codes_dev, const int32_t *code_offsets_dev,
                           const double *dopplers_dev, double sample_rate_hz, uint32_t n_sat, gnss_refine_result *out_dev, void *stream)
{
  if (!signal_dev || !codes_dev || !code_offsets_dev || !dopplers_dev || !out_dev) { set_error("gnss_refine_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  // the reference wipes into static float iw[1023 * 4], qw[1023 * 4] (:392): longer windows overrun them there, refused here
  if (size < (uint32_t)kRefineOffs || size > (uint32_t)kRefineMaxSize) {
    set_error("gnss_refine_dev: size %u outside %d..%d (the reference's static buffers hold 1023 * 4 samples)", size, kRefineOffs, kRefineMaxSize);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if (!(sample_rate_hz > 0.0)) { set_error("gnss_refine_dev: sample rate must be positive"); return ARM_MATH_ARGUMENT_ERROR; }
  if (n_sat == 0) return ARM_MATH_SUCCESS;
  int dev = -1;
  if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return ARM_MATH_ARGUMENT_ERROR;
  const size_t lds = refine_lds_bytes((int)size);
  static std::atomic<bool> done[16];
  if (!hip_ok(lds_attr_once(gnss_refine_kernel, refine_lds_bytes(kRefineMaxSize), done, dev), "gnss_refine_kernel attribute")) return ARM_MATH_ARGUMENT_ERROR;
  note_kernel("gnss_refine_kernel", false);
  gnss_refine_kernel<<<n_sat, 256, lds, (hipStream_t)stream>>>(reinterpret_cast<const float2 *>(signal_dev), (int)signal_stride, codes_dev, (int)size,
                                                             code_offsets_dev, dopplers_dev, sample_rate_hz, out_dev);
  return hip_ok(hipGetLastError(), "gnss_refine_kernel") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

arm_status gnss_find_top2_peaks_dev(const float *data_dev, int32_t is_complex, int32_t data_size, int32_t pk_sep, uint32_t batch,
                                    gnss_top2 *peaks_dev, void *stream)
{
  if (!data_dev || !peaks_dev) { set_error("gnss_find_top2_peaks_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  if (data_size <= 0 || data_size > 16384 || pk_sep < 0) { set_error("gnss_find_top2_peaks_dev: data_size must be 1..16384"); return ARM_MATH_ARGUMENT_ERROR; }
  if (batch == 0) return ARM_MATH_SUCCESS;
  if (is_complex)
    gnss_top2_kernel<true><<<batch, 256, 0, (hipStream_t)stream>>>(data_dev, data_size, pk_sep, peaks_dev);
  else
    gnss_top2_kernel<false><<<batch, 256, 0, (hipStream_t)stream>>>(data_dev, data_size, pk_sep, peaks_dev);
  return hip_ok(hipGetLastError(), "gnss_top2_kernel launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// ------------------------------------------------------------------------------------------
// the reference's own names (Include/gnss_codes.h:62-126), host pointers: stage, run the device probes above, copy
// back.  Void / value-returning entry points have no status channel: like arm_cfft_f32 they abort() with a message
// when the GPU cannot run them (find_top2_peaks_* return -1 instead, as their reference signature allows).
// ------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
  void *p = nullptr;
  explicit DevBuf(size_t bytes) { if (bytes && !hip_ok(hipMalloc(&p, bytes), "hipMalloc(staging)")) p = nullptr; }
  ~DevBuf() { if (p) (void)hipFree(p); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
};
[[noreturn]] void die(const char *who)
{
  fprintf(stderr, "libarmfft_mi355: %s cannot run (%s)\n", who, arm_fft_mi355_last_error());
  abort();
}
}  // namespace

// gnss_codes.c:969-1030
void getCode(int num, int samplesPerChip, const int prn, int *out)
{
  if (!out || num <= 0) return;
  if (samplesPerChip < 1) samplesPerChip = 1;  // :1010
  const size_t total = (size_t)num * (size_t)samplesPerChip;
  DevBuf d(total * sizeof(int32_t));
  if (!d.p || gnss_get_code_dev(num, samplesPerChip, prn, (int32_t *)d.p) != ARM_MATH_SUCCESS ||
      !hip_ok(hipMemcpy(out, d.p, total * sizeof(int32_t), hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
    die("getCode");
}

// gnss_codes.c:1066-1071 (getCode + mix_prn :1032-1063)
void synth_gps_prn(int prn, float doppler, size_t size, float32_t *replica, int spc)
{
  if (!replica || size == 0) return;
  DevBuf d(size * sizeof(float2));
  if (!d.p || size > 0xFFFFFFFFull || gnss_synth_gps_prn_dev(prn, doppler, (uint32_t)size, (float *)d.p, spc) != ARM_MATH_SUCCESS ||
      !hip_ok(hipMemcpy(replica, d.p, size * sizeof(float2), hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
    die("synth_gps_prn");
}

// gnss_codes.c:1032-1063 (external linkage there, although Include/gnss_codes.h does not declare it)
void mix_prn(const int32_t *prn_a, const double doppler_a_hz, const double phase_a_deg, float32_t *signal, int size, int spc)
{
  if (!prn_a || !signal || size <= 0) return;
  DevBuf c((size_t)size * sizeof(int32_t)), d((size_t)size * sizeof(float2));
  if (!c.p || !d.p || !hip_ok(hipMemcpy(c.p, prn_a, (size_t)size * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      gnss_mix_prn_dev((const int32_t *)c.p, doppler_a_hz, phase_a_deg, (float *)d.p, (uint32_t)size, spc, nullptr) != ARM_MATH_SUCCESS ||
      !hip_ok(hipMemcpy(signal, d.p, (size_t)size * sizeof(float2), hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
    die("mix_prn");
}

// quasi_pilot_tests.c:153-181 (Include/gnss_codes.h:110)
void make_replica(const int32_t *prn_a, float32_t *out_iandq, float doppler, int size, float samp_freq)
{
  if (!prn_a || !out_iandq || size <= 0) return;
  DevBuf c((size_t)size * sizeof(int32_t)), d((size_t)size * sizeof(float2));
  if (!c.p || !d.p || !hip_ok(hipMemcpy(c.p, prn_a, (size_t)size * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      gnss_make_replica_dev((const int32_t *)c.p, (uint32_t)size, doppler, samp_freq, (float *)d.p, nullptr) != ARM_MATH_SUCCESS ||
      !hip_ok(hipMemcpy(out_iandq, d.p, (size_t)size * sizeof(float2), hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
    die("make_replica");
}

// quasi_pilot_tests.c:138-151 (Include/gnss_codes.h:111), in place on the caller's array like the reference
void rotate_fwd(int array[], int size, int offset)
{
  if (!array || size <= 0) return;
  if (offset < 0 || offset > size) {   // the reference reads outside the array here; refuse and leave the array as it is
    set_error("rotate_fwd: offset %d outside 0..%d, array left unchanged", offset, size);
    return;
  }
  DevBuf a((size_t)size * sizeof(int32_t)), b((size_t)size * sizeof(int32_t));
  if (!a.p || !b.p || !hip_ok(hipMemcpy(a.p, array, (size_t)size * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      gnss_rotate_fwd_dev((const int32_t *)a.p, (int32_t *)b.p, (uint32_t)size, offset, nullptr) != ARM_MATH_SUCCESS ||
      !hip_ok(hipMemcpy(array, b.p, (size_t)size * sizeof(int32_t), hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
    die("rotate_fwd");
}

// CMSIS/CMSIS.cpp:362-434, the harness's own name and argument list (a C++ function there; exported with C linkage here)
double recover_prn_phase_deg_with_doppler(const float32_t *iandq, int SIZE, const int prn_a[], int code_offset, double doppler_a_hz,
                                          double sampe_rate_hz, int *best_offset_out, float *doppler_out, float *pwr)
{
  if (!iandq || !prn_a || !best_offset_out || !doppler_out || !pwr) { set_error("recover_prn_phase_deg_with_doppler: null argument"); die("recover_prn_phase_deg_with_doppler"); }
  const size_t n = (size_t)(SIZE > 0 ? SIZE : 0);
  DevBuf sig(n * sizeof(float2)), code(n * sizeof(int32_t)), par(sizeof(double) + sizeof(int32_t) * 2), res(sizeof(gnss_refine_result));
  struct { double dop; int32_t off; int32_t pad; } h = {doppler_a_hz, code_offset, 0};
  gnss_refine_result r;
  if (!sig.p || !code.p || !par.p || !res.p || !hip_ok(hipMemcpy(sig.p, iandq, n * sizeof(float2), hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      !hip_ok(hipMemcpy(code.p, prn_a, n * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      !hip_ok(hipMemcpy(par.p, &h, sizeof h, hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      gnss_refine_dev((const float *)sig.p, 0, (uint32_t)n, (const int32_t *)code.p, (const int32_t *)((char *)par.p + 8), (const double *)par.p,
                      sampe_rate_hz, 1, (gnss_refine_result *)res.p, nullptr) != ARM_MATH_SUCCESS ||
      !hip_ok(hipMemcpy(&r, res.p, sizeof r, hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
    die("recover_prn_phase_deg_with_doppler");
  *doppler_out = r.doppler;
  *best_offset_out = r.best_off;
  *pwr = r.pwr;
  return r.phase_deg;
}

// The step between the acquisition record and carr_phase (Include/msb_funcs.h:30, cycles): for every satellite of `meas` the
// fine search above around (code phase, Doppler) of the record on the first `size` samples of `signal` (host pointer),
// code = getCode(size / spc, spc, prn); carr_phase = phase_deg / 360.  details (optional) receives the search results.
int gnss_meas_refine(const float32_t *signal, uint32_t size, int32_t spc, double sample_rate_hz, int32_t fft_len, bb_meas_t *meas,
                     gnss_refine_result *details)
{
  if (!signal || !meas || spc < 1 || fft_len <= 0 || meas->num_sat < 0 || meas->num_sat > 20 || size % (uint32_t)spc) {
    set_error("gnss_meas_refine: bad argument");
    return -1;
  }
  const int ns = meas->num_sat;
  if (ns == 0) return 0;
  DevBuf sig((size_t)size * sizeof(float2)), codes((size_t)ns * size * sizeof(int32_t)), offs(ns * sizeof(int32_t)), dops(ns * sizeof(double)),
      res(ns * sizeof(gnss_refine_result));
  if (!sig.p || !codes.p || !offs.p || !dops.p || !res.p) return -1;
  int32_t h_off[20];
  double h_dop[20];
  for (int k = 0; k < ns; k++) {
    const bb_meas_sat_t &s = meas->sats[k];
    h_off[k] = (int32_t)lround((double)s.code_phase * (double)fft_len) % (int32_t)size;   // code_phase [ms] = samples / fft_len (gnss_acq_to_meas)
    if (h_off[k] < 0) h_off[k] += (int32_t)size;
    h_dop[k] = (double)s.doppler;
    if (gnss_get_code_dev((int32_t)(size / (uint32_t)spc), spc, (int32_t)s.prn, (int32_t *)codes.p + (size_t)k * size) != ARM_MATH_SUCCESS) return -1;
  }
  gnss_refine_result h_res[20];
  if (!hip_ok(hipMemcpy(sig.p, signal, (size_t)size * sizeof(float2), hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      !hip_ok(hipMemcpy(offs.p, h_off, ns * sizeof(int32_t), hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      !hip_ok(hipMemcpy(dops.p, h_dop, ns * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      gnss_refine_dev((const float *)sig.p, 0, size, (const int32_t *)codes.p, (const int32_t *)offs.p, (const double *)dops.p, sample_rate_hz, (uint32_t)ns,
                      (gnss_refine_result *)res.p, nullptr) != ARM_MATH_SUCCESS ||
      !hip_ok(hipMemcpy(h_res, res.p, ns * sizeof(gnss_refine_result), hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
    return -1;
  for (int k = 0; k < ns; k++) {
    meas->sats[k].carr_phase = (float)(h_res[k].phase_deg / 360.0);
    if (details) details[k] = h_res[k];
  }
  return ns;
}

static int top2_host(const float *data, int is_complex, int data_size, int pk_sep, top2_pks *peaks, FILE *fp_out)
{
  if (data_size <= 0 || !data || !peaks) return -1;  // :1145 / :1169
  // a negative pk_sep makes "abs(m - idx1) > pk_sep" true for every pair of distinct indices, exactly as pk_sep = 0 does
  // (m never equals idx1 when the test runs): the reference accepts it, and so does this wrapper
  if (pk_sep < 0) pk_sep = 0;
  if (fp_out) {  // the reference's debug dump of every magnitude (:1151, :1175): formatting only
    for (int m = 0; m < data_size; m++) {
      const float magn = is_complex ? sqrtf(data[2 * m] * data[2 * m] + data[2 * m + 1] * data[2 * m + 1]) : data[m];
      fprintf(fp_out, "%d, %f \n", m, magn);
    }
  }
  const size_t bytes = (size_t)data_size * (is_complex ? 2 : 1) * sizeof(float);
  DevBuf d(bytes), r(sizeof(gnss_top2));
  gnss_top2 t;
  if (!d.p || !r.p || !hip_ok(hipMemcpy(d.p, data, bytes, hipMemcpyHostToDevice), "hipMemcpy(H2D)") ||
      gnss_find_top2_peaks_dev((const float *)d.p, is_complex, data_size, pk_sep, 1, (gnss_top2 *)r.p, nullptr) != ARM_MATH_SUCCESS ||
      !hip_ok(hipMemcpy(&t, r.p, sizeof t, hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
    return -1;
  peaks->val1 = t.val1;
  peaks->idx1 = (size_t)t.idx1;  // -1 ("none") converts as in the reference's int -> size_t assignment (:1162-1165)
  peaks->val2 = t.val2;
  peaks->idx2 = (size_t)t.idx2;
  return 0;
}
// gnss_codes.c:1144-1166 / :1168-1190; data_size <= 16384 here
int find_top2_peaks_cplx(const float32_t *data, int data_size, int pk_sep, top2_pks *peaks, FILE *fp_out) { return top2_host(data, 1, data_size, pk_sep, peaks, fp_out); }
int find_top2_peaks_real(const float32_t *data, int data_size, int pk_sep, top2_pks *peaks, FILE *fp_out) { return top2_host(data, 0, data_size, pk_sep, peaks, fp_out); }

static double snr_host(const float *convol, int is_complex, int cov_size, double peak, int peak_loc, const char *who)
{
  if (cov_size <= 0) {
    // the reference's loop does not run and reads nothing (:1112 / :1129); what is left is scalar arithmetic on the
    // arguments: mean = 0 / (cov_size - 5), 10 log10(peak / mean) -- an infinity or a NaN, returned like the reference does
    const double mean = 0.0 / (double)(cov_size - 5);
    return 10.0 * log10(peak / mean);
  }
  if (!convol) { set_error("%s: null data", who); die(who); }
  const size_t bytes = (size_t)cov_size * (is_complex ? 2 : 1) * sizeof(float);
  DevBuf d(bytes), r(sizeof(double));
  double snr = 0.0;
  bool ok = d.p && r.p && hip_ok(hipMemcpy(d.p, convol, bytes, hipMemcpyHostToDevice), "hipMemcpy(H2D)");
  if (ok) {
    if (is_complex) gnss_snr_kernel<true><<<1, 256, 0, nullptr>>>((const float *)d.p, cov_size, peak, peak_loc, (double *)r.p);
    else gnss_snr_kernel<false><<<1, 256, 0, nullptr>>>((const float *)d.p, cov_size, peak, peak_loc, (double *)r.p);
    ok = hip_ok(hipGetLastError(), "gnss_snr_kernel") && hip_ok(hipMemcpy(&snr, r.p, sizeof snr, hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
  }
  if (!ok) die(who);
  return snr;
}
// gnss_codes.c:1110-1125 / :1127-1142
double compute_snr_cplx(float32_t *convol, int cov_size, double peak, int peak_loc) { return snr_host(convol, 1, cov_size, peak, peak_loc, "compute_snr_cplx"); }
double compute_snr_real(float32_t *convol, int cov_size, double peak, int peak_loc) { return snr_host(convol, 0, cov_size, peak, peak_loc, "compute_snr_real"); }

// gnss_codes.c:1096-1108: six floating-point operations on three doubles the caller already holds -- scalar host
// arithmetic, there is no data to send to a GPU (the fused kernel evaluates the same formula per row, finish_row)
double InterpolateCodePhase(uint32_t index, double earlyPower, double promptPower, double latePower)
{
  double interpolatedIndex;
  if (latePower > earlyPower) {
    interpolatedIndex = 0.5 * (latePower - earlyPower) / (promptPower - earlyPower);
    interpolatedIndex = index + (interpolatedIndex / 3.0) * (2.0 * interpolatedIndex + 2.0);
  } else {
    interpolatedIndex = 0.5 * (latePower - earlyPower) / (promptPower - latePower);
    interpolatedIndex = index + (interpolatedIndex / 3.0) * (-2.0 * interpolatedIndex + 2.0);
  }
  return interpolatedIndex;
}

}  // extern "C"
