// armfft_internal.h -- declarations shared by the translation units of libarmfft_mi355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/arm_fft_mi355.h"

namespace armfft {

constexpr int kMinLog = 4;    // 16
constexpr int kMaxLog = 14;   // 16384: largest transform held whole in LDS (128 KiB of 160 KiB)
constexpr int kMinLargeLog = 15, kMaxLargeLog = 22;  // four-step through HBM

// ---- error plumbing (armfft_abi.cpp) ----
void set_error(const char *fmt, ...);
bool hip_ok(hipError_t e, const char *what);

// Name of the kernel the calling thread launched last, for arm_fft_mi355_last_kernel (bench.py puts it next to
// its roofline numbers so that the line can be matched to a rocprof trace).  fmt is a string LITERAL with up to five
// %d; nothing is formatted until somebody asks.  append: a second kernel of the same call (the four-step's pass 2).
void note_kernel(const char *fmt, bool append, int a = 0, int b = 0, int c = 0, int d = 0, int e = 0);

// per-device cached table exp(-2 pi i m / 2^logn), m < 2^logn (armfft_abi.cpp); nullptr + error on failure
const float2 *dev_twiddle_f32(int logn);
// three-level table of W_N for the four-step kernels (armfft_large.hip), same caching
const float2 *dev_twiddle_lo(int logn);
// the same tables if they already exist on the current device, else nullptr: never allocate, set no error
const float2 *dev_twiddle_f32_peek(int logn);
const float2 *dev_twiddle_lo_peek(int logn);

// ---- kernel launchers (no allocation, asynchronous on the stream) ----
// f32 complex, in place; tw = device table of N entries exp(-2 pi i m / N)
// order: 0 natural, 1 arm_cfft_f32's radix-8 digit-scrambled order, 2 plain bit reversal (radix-2 API)
hipError_t launch_cfft_f32(int logn, float2 *data, const float2 *tw, uint32_t batch, bool inverse, int order,
                           hipStream_t stream);
// f32 real: n_real = 2^logn_real points; tw_c = table of the n_real/2-point complex transform,
// tw_r = n_real/2 entries (sin, cos)(2 pi k / n_real)
hipError_t launch_rfft_f32(int logn_real, const float *in, float *out, const float2 *tw_c, const float2 *tw_r,
                           uint32_t batch, bool inverse, hipStream_t stream);
// fixed point, in place; tw = 3N/4 (cos, sin) pairs
hipError_t launch_cfft_q15(int logn, int16_t *data, const int16_t *tw, uint32_t batch, bool inverse, bool bitrev,
                           hipStream_t stream);
hipError_t launch_cfft_q31(int logn, int32_t *data, const int32_t *tw, uint32_t batch, bool inverse, bool bitrev,
                           hipStream_t stream);

// fixed-point real FFT, n_real = 2^logn_real (32..8192); tw = table of the n_real/2-point complex
// transform, A/B = realCoefA/B (8192 scalars), mod = 8192 / n_real.  Forward: in n_real scalars per
// transform, out 2*n_real.  Inverse: in n_real + 2 scalars per transform, out n_real.
hipError_t launch_rfft_q15(int logn_real, const int16_t *in, int16_t *out, const int16_t *tw, const int16_t *A, const int16_t *B, int mod,
                           uint32_t batch, bool inverse, bool bitrev, hipStream_t stream);
hipError_t launch_rfft_q31(int logn_real, const int32_t *in, int32_t *out, const int32_t *tw, const int32_t *A, const int32_t *B, int mod,
                           uint32_t batch, bool inverse, bool bitrev, hipStream_t stream);

// large complex f32 (2^15..2^22), four-step: data -> work -> data
hipError_t launch_cfft_large_f32(int logn, float2 *data, float2 *work, const float2 *tw1, const float2 *tw2,
                                 const float2 *tw_big, uint32_t batch, bool inverse, hipStream_t stream);
// three-level table of W_N, N = 2^logn, into a buffer of 1024 entries (layout: armfft_large.hip)
hipError_t launch_gen_twiddle_lo(float2 *tw, int logn, hipStream_t stream);

// on-device table generation
hipError_t launch_gen_twiddle_f32(float2 *tw, uint32_t n, hipStream_t stream);        // exp(-2 pi i m/n), m < n
hipError_t launch_gen_twiddle_rfft(float2 *tw, uint32_t n_real, hipStream_t stream);  // (sin, cos)(2 pi k/n_real), k < n_real/2
hipError_t launch_gen_twiddle_q15(int16_t *tw, uint32_t n, int rounding, hipStream_t stream);
hipError_t launch_gen_twiddle_q31(int32_t *tw, uint32_t n, int rounding, hipStream_t stream);

}  // namespace armfft
