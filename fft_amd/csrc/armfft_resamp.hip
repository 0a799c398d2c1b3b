// armfft_resamp.hip -- zero-phase windowed-sinc I/Q resampler for gfx950.
//
// Replaces iq_resamp_zp_init / iq_resamp_process_zero_phase (Source/TransformFunctions/up_sample.c:
// 31-44, :115-157; kernel iq_resamp_forward :65-104) behind up_sample_N_to_M and
// up_sample_10k_to_16k (gnss_codes.c:868-904): forward fractional-delay resample Fin -> Fout with
// an ntaps-tap Blackman-windowed sinc (reflection padding at the block edges), then the same
// kernel at identity rate on the reversed sequence, reversed back (zero phase).
//
// The reference runs two passes with a float intermediate y.  Here ONE kernel does both per tile:
// a workgroup computes y (rounded to float exactly as the reference stores it) for its 256 outputs
// plus the ntaps-1 halo into LDS, then applies the identity-rate pass.  The reversal folds into
// the indexing:  out[k] = sum_m h0[m] * y[reflect(k - m)] / sum_m h0[m],  m = -half .. half-1,
// h0 = the taps at zero fractional delay (reflect(N-1-a) = N-1-reflect(a)).  Arithmetic follows
// the reference: taps and accumulation in double, outputs rounded to float once per pass.
// HBM-bound in principle (8 B in, 8 B out per sample); double-precision sin() per tap makes it
// fp64-VALU-bound in practice -- the blocks are milliseconds long and few, this is not a hot loop
// of config 5 (not on its path at all) but of the E5a/L5 front end.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "armfft_internal.h"

namespace armfft {

constexpr int kResampTile = 256;
constexpr int kResampMaxTaps = 64;

__device__ __forceinline__ double rs_sinc_pi(double x)  // up_sample.c:16-20
{
  return fabs(x) < 1e-12 ? 1.0 : sin(M_PI * x) / (M_PI * x);
}
__device__ __forceinline__ double rs_blackman(int n, int half)  // up_sample.c:23-29
{
  const int N = 2 * half, k = n + half;
  const double phi = (2.0 * M_PI * k) / (double)(N - 1);
  return 0.42 - 0.5 * cos(phi) + 0.08 * cos(2.0 * phi);
}
__device__ __forceinline__ int rs_reflect(int idx, int N)  // up_sample.c:51-59
{
  if (N <= 1) return 0;
  while (idx < 0 || idx >= N) {
    if (idx < 0) idx = -idx - 1;
    if (idx >= N) idx = 2 * N - 1 - idx;
  }
  return idx;
}

// grid = (tiles, batch); block b of the batch: in + b*in_stride, out + b*out_stride (in samples)
__global__ __launch_bounds__(kResampTile) void resamp_zero_phase_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, int n_in,
                                                                        int n_out, size_t in_stride, size_t out_stride, double dt,
                                                                        int half, double cutoff)
{
  __shared__ double win[kResampMaxTaps];   // Blackman window per tap
  __shared__ double h0[kResampMaxTaps];    // taps of the identity-rate pass (frac = 0)
  __shared__ float2 ys[kResampTile + kResampMaxTaps];
  const int tid = threadIdx.x, ntaps = 2 * half;
  const float2 *__restrict__ x = in + (size_t)blockIdx.y * in_stride;
  float2 *__restrict__ o = out + (size_t)blockIdx.y * out_stride;
  if (tid < ntaps) {
    const int m = tid - half;
    const double w = rs_blackman(m, half);
    win[tid] = w;
    h0[tid] = (2.0 * cutoff) * rs_sinc_pi(2.0 * cutoff * (double)m) * w;
  }
  __syncthreads();
  // pass 1: y at reflect(j), j = k0 - half + 1 .. k0 + TILE - 1 + half
  const int k0 = blockIdx.x * kResampTile;
  for (int i = tid; i < kResampTile + ntaps - 1; i += kResampTile) {
    const int k = rs_reflect(k0 - half + 1 + i, n_out);
    const double t = (double)k * dt;
    const int ti = (int)floor(t);
    const double frac = t - (double)ti;
    double ai = 0.0, aq = 0.0, ws = 0.0;
    for (int m = -half; m < half; m++) {
      const double xx = (double)m - frac;
      double h = (2.0 * cutoff) * rs_sinc_pi(2.0 * cutoff * xx);
      h *= win[m + half];
      const float2 s = x[rs_reflect(ti + m, n_in)];
      ai += h * (double)s.x;
      aq += h * (double)s.y;
      ws += h;
    }
    const double inv = fabs(ws) > 1e-15 ? 1.0 / ws : 1.0;
    ys[i] = make_float2((float)(ai * inv), (float)(aq * inv));
  }
  __syncthreads();
  // pass 2 (identity rate on the reversed sequence, reversed back)
  const int k = k0 + tid;
  if (k < n_out) {
    double ai = 0.0, aq = 0.0, ws = 0.0;
    for (int m = -half; m < half; m++) {
      const double h = h0[m + half];
      // y[reflect(k - m)]: j = k - m -> LDS slot j - (k0 - half + 1)
      const float2 s = ys[tid - m + half - 1];
      ai += h * (double)s.x;
      aq += h * (double)s.y;
      ws += h;
    }
    const double inv = fabs(ws) > 1e-15 ? 1.0 / ws : 1.0;
    o[k] = make_float2((float)(ai * inv), (float)(aq * inv));
  }
}

}  // namespace armfft

using namespace armfft;

extern "C" {

// number of outputs iq_resamp_process_zero_phase would write (up_sample.c:121-125); 0 on bad arguments
uint32_t iq_resamp_zero_phase_count(double fin_hz, double fout_hz, int32_t ntaps, double cutoff, uint32_t n_in, uint32_t n_out_cap)
{
  if (ntaps < 4 || (ntaps & 1) || ntaps > kResampMaxTaps) return 0;   // up_sample.c:32 (+ this library's tap limit)
  if (!(cutoff > 0.0 && cutoff <= 0.5)) return 0;                     // :33
  if (n_in == 0 || !(fin_hz > 0.0) || !(fout_hz > 0.0)) return 0;
  long long n = llround((double)n_in * (fout_hz / fin_hz));
  if (n > (long long)n_out_cap) n = n_out_cap;
  if (n <= 0 || n > 0x7FFFFFFFLL || n_in > 0x7FFFFFFFu) return 0;
  return (uint32_t)n;
}

arm_status iq_resamp_zero_phase_dev(double fin_hz, double fout_hz, int32_t ntaps, double cutoff, const float32_t *in_dev, uint32_t n_in,
                                    float32_t *out_dev, uint32_t n_out_cap, uint32_t batch, uint32_t *n_out, void *stream)
{
  if (n_out) *n_out = 0;
  if (!in_dev || !out_dev) { set_error("iq_resamp_zero_phase_dev: null argument"); return ARM_MATH_ARGUMENT_ERROR; }
  const uint32_t n = iq_resamp_zero_phase_count(fin_hz, fout_hz, ntaps, cutoff, n_in, n_out_cap);
  if (n == 0) {
    set_error("iq_resamp_zero_phase_dev: bad arguments (ntaps even in 4..%d, 0 < cutoff <= 0.5, n_in > 0, capacity > 0)", kResampMaxTaps);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if (batch > 65535) { set_error("iq_resamp_zero_phase_dev: at most 65535 blocks per call"); return ARM_MATH_ARGUMENT_ERROR; }
  if (n_out) *n_out = n;
  if (batch == 0) return ARM_MATH_SUCCESS;
  const dim3 grid((n + kResampTile - 1) / kResampTile, batch);
  resamp_zero_phase_kernel<<<grid, kResampTile, 0, (hipStream_t)stream>>>(reinterpret_cast<const float2 *>(in_dev),
                                                                         reinterpret_cast<float2 *>(out_dev), (int)n_in, (int)n,
                                                                         (size_t)n_in, (size_t)n_out_cap, fin_hz / fout_hz, ntaps / 2, cutoff);
  return hip_ok(hipGetLastError(), "resamp_zero_phase_kernel launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// host pointers: one block, copies in and out; returns the number of samples written (0 on failure)
static uint32_t resamp_host(double fin, double fout, const float *in, uint32_t n_in, float *out, uint32_t cap, const char *who, int ntaps = 12,
                            double cutoff = 0.45)
{
  const uint32_t n = iq_resamp_zero_phase_count(fin, fout, ntaps, cutoff, n_in, cap);
  if (!in || !out || n == 0) return 0;
  float *din = nullptr, *dout = nullptr;
  bool ok = hip_ok(hipMalloc((void **)&din, sizeof(float) * 2 * n_in), "hipMalloc") && hip_ok(hipMalloc((void **)&dout, sizeof(float) * 2 * n), "hipMalloc");
  ok = ok && hip_ok(hipMemcpy(din, in, sizeof(float) * 2 * n_in, hipMemcpyHostToDevice), "hipMemcpy(H2D)");
  uint32_t got = 0;
  ok = ok && iq_resamp_zero_phase_dev(fin, fout, ntaps, cutoff, din, n_in, dout, n, 1, &got, nullptr) == ARM_MATH_SUCCESS;
  ok = ok && hip_ok(hipMemcpy(out, dout, sizeof(float) * 2 * n, hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
  (void)hipFree(din);
  (void)hipFree(dout);
  if (!ok) {
    fprintf(stderr, "libarmfft_mi355: %s cannot run (%s)\n", who, arm_fft_mi355_last_error());
    abort();  // void entry points of the reference: never return unprocessed data silently
  }
  return n;
}

// gnss_codes.c:868-884.  The capacity is llround(N_in * (M_out / N_in)) with the INTEGER quotient
// (:879), as in the reference: 10230 -> 16384 writes 10230 samples.
void up_sample_N_to_M(float32_t *in, int N_in, float32_t *out, int M_out)
{
  if (N_in <= 0 || M_out <= 0) { fprintf(stderr, "init failed\n"); return; }
  const long long cap = llround((double)((long long)N_in * (long long)(M_out / N_in)));
  (void)resamp_host((double)N_in, (double)M_out, in, (uint32_t)N_in, out, (uint32_t)cap, "up_sample_N_to_M");
}

// gnss_codes.c:886-904: 10230 samples at 10.23 MHz -> 16384 samples at 16.384 MHz
void up_sample_10k_to_16k(float32_t *in, float32_t *out)
{
  (void)resamp_host(10230e3, 16384e3, in, 10230, out, (uint32_t)(llround(10230 * (16384e3 / 10230e3)) + 8), "up_sample_10k_to_16k");
}

// Include/up_sample.h:20-24, bodies up_sample.c:31-48 (init / free) and :115-157 (process): the reference's own
// state object and calls, host pointers; the resampling itself runs on the GPU (one block per call).
iq_resamp_zp_t *iq_resamp_zp_init(double fin_hz, double fout_hz, int ntaps, double cutoff)
{
  if (ntaps < 4 || (ntaps & 1)) return nullptr;             // up_sample.c:32
  if (!(cutoff > 0.0 && cutoff <= 0.5)) return nullptr;     // :33
  iq_resamp_zp_t *st = (iq_resamp_zp_t *)calloc(1, sizeof(iq_resamp_zp_t));
  if (!st) return nullptr;
  st->fin = fin_hz;
  st->fout = fout_hz;
  st->dt = fin_hz / fout_hz;
  st->ntaps = ntaps;
  st->half = ntaps / 2;
  st->cutoff = cutoff;
  return st;
}

void iq_resamp_zp_free(iq_resamp_zp_t *st) { free(st); }

size_t iq_resamp_process_zero_phase(iq_resamp_zp_t *st, const float32_t *in, size_t Nin, float32_t *out, size_t Nout_cap)
{
  if (!st || !in || !out || Nin == 0 || Nout_cap == 0) return 0;  // up_sample.c:116-118
  if (st->ntaps > kResampMaxTaps) { set_error("iq_resamp_process_zero_phase: at most %d taps", kResampMaxTaps); return 0; }
  if (Nin > 0x7FFFFFFFu) return 0;
  const uint32_t cap = Nout_cap > 0x7FFFFFFFu ? 0x7FFFFFFFu : (uint32_t)Nout_cap;
  return resamp_host(st->fin, st->fout, in, (uint32_t)Nin, out, cap, "iq_resamp_process_zero_phase", st->ntaps, st->cutoff);
}

}  // extern "C"
