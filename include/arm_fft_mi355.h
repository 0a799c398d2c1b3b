/*
 * arm_fft_mi355.h -- C ABI of libarmfft_mi355.so: the CMSIS-DSP transform hot path of
 * fy10000000/FFT, re-implemented as hand-written HIP kernels for AMD MI355X (gfx950).
 *
 * Drop-in surface: the structs and the init/execute functions below keep the names, layouts,
 * argument meaning and error behaviour of the reference's non-Neon/non-MVE C API
 * (reference file:line cited per declaration, paths relative to the reference root), so an
 * existing caller (CMSIS/CMSIS.cpp, the Testing/ suites) relinks unchanged.  Extensions
 * (suffixes _batch, _dev, prefix gnss_/arm_fft_mi355_) are new and return arm_status.
 *
 * Conventions
 *   - complex data is interleaved {re, im}; transform b of a batch starts at p + 2*fftLen*b
 *     (rfft: p + fftLenRFFT*b, pOut + fftLenRFFT*b);
 *   - host-pointer entry points copy in and out synchronously; *_dev entry points take device
 *     pointers and a hipStream_t (as void*), launch asynchronously and never allocate;
 *   - every device data pointer handed to a *_dev entry point must be 16-byte aligned (hipMalloc
 *     and torch allocations are; the kernels use 16-byte vector accesses): a misaligned pointer is
 *     refused with ARM_MATH_ARGUMENT_ERROR, nothing is launched;
 *   - the reference's own entry points return void (arm_cfft_f32, arm_cfft_q15, arm_cfft_q31,
 *     arm_rfft_fast_f32, arm_rfft_q15/q31, arm_cfft_radix2_f32, up_sample_*): when the GPU cannot
 *     run them (no device, a failed HIP call, staging memory exhausted) they print the reason to
 *     stderr and abort() rather than return unprocessed data.  A C caller that wants to handle such
 *     failures calls the status-returning *_batch_* twin (batch = 1) instead;
 *   - instances are caller-owned PODs; pTwiddle / pBitRevTable point to valid HOST tables owned
 *     by the library (static lifetime), device plans are cached inside the library per GPU.
 */
#ifndef ARM_FFT_MI355_H
#define ARM_FFT_MI355_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar types and status: Include/arm_math_types.h:334-350, :610-617 ---- */
typedef int16_t q15_t;
typedef int32_t q31_t;
typedef int64_t q63_t;
typedef float float32_t;

typedef enum {
  ARM_MATH_SUCCESS = 0,
  ARM_MATH_ARGUMENT_ERROR = -1,
  ARM_MATH_LENGTH_ERROR = -2,
  ARM_MATH_SIZE_MISMATCH = -3,
  ARM_MATH_NANINF = -4,
  ARM_MATH_SINGULAR = -5,
  ARM_MATH_TEST_FAILURE = -6,
  ARM_MATH_DECOMPOSITION_FAILURE = -7
} arm_status;

/* ---- instance structures ---- */
/* Include/dsp/transform_functions.h:416-430 */
typedef struct {
  uint16_t fftLen;
  const float32_t *pTwiddle;
  const uint16_t *pBitRevTable;
  uint16_t bitRevLength;
} arm_cfft_instance_f32;

/* Include/dsp/transform_functions.h:284-298 */
typedef struct {
  uint16_t fftLen;
  const q15_t *pTwiddle;
  const uint16_t *pBitRevTable;
  uint16_t bitRevLength;
} arm_cfft_instance_q15;

/* Include/dsp/transform_functions.h:352-366 */
typedef struct {
  uint16_t fftLen;
  const q31_t *pTwiddle;
  const uint16_t *pBitRevTable;
  uint16_t bitRevLength;
} arm_cfft_instance_q31;

/* Include/dsp/transform_functions.h:827-832 */
typedef struct {
  arm_cfft_instance_f32 Sint;
  uint16_t fftLenRFFT;
  const float32_t *pTwiddleRFFT;
} arm_rfft_fast_instance_f32;

/* ---- f32 complex FFT ---- */
/* Include/dsp/transform_functions.h:433-445; bodies Source/TransformFunctions/arm_cfft_init_f32.c:121-354.
 * Unsupported length -> ARM_MATH_ARGUMENT_ERROR.  Extension: 8192 and 16384 are accepted too
 * (the fork's stated goal, CMSIS/CMSIS.cpp:1-10; the reference stops at 4096). */
arm_status arm_cfft_init_4096_f32(arm_cfft_instance_f32 *S);
arm_status arm_cfft_init_2048_f32(arm_cfft_instance_f32 *S);
arm_status arm_cfft_init_1024_f32(arm_cfft_instance_f32 *S);
arm_status arm_cfft_init_512_f32(arm_cfft_instance_f32 *S);
arm_status arm_cfft_init_256_f32(arm_cfft_instance_f32 *S);
arm_status arm_cfft_init_128_f32(arm_cfft_instance_f32 *S);
arm_status arm_cfft_init_64_f32(arm_cfft_instance_f32 *S);
arm_status arm_cfft_init_32_f32(arm_cfft_instance_f32 *S);
arm_status arm_cfft_init_16_f32(arm_cfft_instance_f32 *S);
arm_status arm_cfft_init_f32(arm_cfft_instance_f32 *S, uint16_t fftLen);

/* Include/dsp/transform_functions.h:461-465; body arm_cfft_f32.c:1243-1298.  In place, forward
 * unscaled, inverse scaled by 1/fftLen; bitReverseFlag 1 = natural order (0 = the reference's
 * digit-scrambled order).  Silently does nothing for an unsupported fftLen, like the reference's
 * switch without default (:1263-1280); GPU failures are reported by arm_fft_mi355_last_error(). */
void arm_cfft_f32(const arm_cfft_instance_f32 *S, float32_t *p1, uint8_t ifftFlag, uint8_t bitReverseFlag);

/* extension: `batch` transforms back to back, host pointers */
arm_status arm_cfft_batch_f32(const arm_cfft_instance_f32 *S, float32_t *p1, uint32_t batch, uint8_t ifftFlag,
                              uint8_t bitReverseFlag);
/* extension: device pointers, asynchronous on `stream` (hipStream_t), no allocation */
arm_status arm_cfft_f32_dev(const arm_cfft_instance_f32 *S, float32_t *p1_dev, uint32_t batch, uint8_t ifftFlag,
                            uint8_t bitReverseFlag, void *stream);

/* ---- q15 complex FFT ---- */
/* Include/dsp/transform_functions.h:302-316; bodies arm_cfft_init_q15.c:117-352 (16..4096).
 * 8192 is accepted (declared by the fork at :303 but never defined); 16384 likewise. */
arm_status arm_cfft_init_16384_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_8192_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_4096_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_2048_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_1024_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_512_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_256_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_128_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_64_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_32_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_16_q15(arm_cfft_instance_q15 *S);
arm_status arm_cfft_init_q15(arm_cfft_instance_q15 *S, uint16_t fftLen);
/* Include/dsp/transform_functions.h:332-336; body arm_cfft_q15.c:671-726 (+ radix4by2 :736-931,
 * butterflies arm_cfft_radix4_q15.c:572-970, :1434-1813).  Output down-scaled by fftLen with the
 * per-stage shifts and saturations of the reference's generic-C branch reproduced bit for bit. */
void arm_cfft_q15(const arm_cfft_instance_q15 *S, q15_t *p1, uint8_t ifftFlag, uint8_t bitReverseFlag);
arm_status arm_cfft_batch_q15(const arm_cfft_instance_q15 *S, q15_t *p1, uint32_t batch, uint8_t ifftFlag,
                              uint8_t bitReverseFlag);
arm_status arm_cfft_q15_dev(const arm_cfft_instance_q15 *S, q15_t *p1_dev, uint32_t batch, uint8_t ifftFlag,
                            uint8_t bitReverseFlag, void *stream);

/* ---- q31 complex FFT ---- */
/* Include/dsp/transform_functions.h:369-382; bodies arm_cfft_init_q31.c:118-355 (16..4096).
 * Extension: 8192 and 16384 (same butterflies, the fork's larger tables). */
arm_status arm_cfft_init_8192_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_4096_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_2048_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_1024_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_512_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_256_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_128_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_64_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_32_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_16_q31(arm_cfft_instance_q31 *S);
arm_status arm_cfft_init_q31(arm_cfft_instance_q31 *S, uint16_t fftLen);
/* Include/dsp/transform_functions.h:396-400; body arm_cfft_q31.c:708-759 (+ radix4by2 :767-885,
 * butterflies arm_cfft_radix4_q31.c:153-473, :524-834, multiplies Include/dsp/none.h:185-206). */
void arm_cfft_q31(const arm_cfft_instance_q31 *S, q31_t *p1, uint8_t ifftFlag, uint8_t bitReverseFlag);
arm_status arm_cfft_batch_q31(const arm_cfft_instance_q31 *S, q31_t *p1, uint32_t batch, uint8_t ifftFlag,
                              uint8_t bitReverseFlag);
arm_status arm_cfft_q31_dev(const arm_cfft_instance_q31 *S, q31_t *p1_dev, uint32_t batch, uint8_t ifftFlag,
                            uint8_t bitReverseFlag, void *stream);

/* ---- f32 real FFT ---- */
/* Include/dsp/transform_functions.h:834-847; bodies arm_rfft_fast_init_f32.c:83-383 (32..4096).
 * Extension: 8192 and 16384 real points (needs the 4096/8192-point complex transform). */
arm_status arm_rfft_fast_init_32_f32(arm_rfft_fast_instance_f32 *S);
arm_status arm_rfft_fast_init_64_f32(arm_rfft_fast_instance_f32 *S);
arm_status arm_rfft_fast_init_128_f32(arm_rfft_fast_instance_f32 *S);
arm_status arm_rfft_fast_init_256_f32(arm_rfft_fast_instance_f32 *S);
arm_status arm_rfft_fast_init_512_f32(arm_rfft_fast_instance_f32 *S);
arm_status arm_rfft_fast_init_1024_f32(arm_rfft_fast_instance_f32 *S);
arm_status arm_rfft_fast_init_2048_f32(arm_rfft_fast_instance_f32 *S);
arm_status arm_rfft_fast_init_4096_f32(arm_rfft_fast_instance_f32 *S);
arm_status arm_rfft_fast_init_f32(arm_rfft_fast_instance_f32 *S, uint16_t fftLen);
/* Include/dsp/transform_functions.h:860-863; body arm_rfft_fast_f32.c:675-699 (stage :316-402,
 * merge :405-462).  Forward: fftLenRFFT reals -> {DC, Nyquist, X1..X(N/2-1)}; inverse the reverse.
 * The reference clobbers `p` in the forward direction; here `p` is left untouched. */
void arm_rfft_fast_f32(const arm_rfft_fast_instance_f32 *S, float32_t *p, float32_t *pOut, uint8_t ifftFlag);
arm_status arm_rfft_fast_batch_f32(const arm_rfft_fast_instance_f32 *S, float32_t *p, float32_t *pOut,
                                   uint32_t batch, uint8_t ifftFlag);
arm_status arm_rfft_fast_f32_dev(const arm_rfft_fast_instance_f32 *S, const float32_t *p_dev, float32_t *pOut_dev,
                                 uint32_t batch, uint8_t ifftFlag, void *stream);

/* ---- fixed-point real FFT ---- */
/* Include/dsp/transform_functions.h:513-526 (non-Neon, non-MVE layout) */
typedef struct {
  uint32_t fftLenReal;
  uint8_t ifftFlagR;
  uint8_t bitReverseFlagR;
  uint32_t twidCoefRModifier;
  const q15_t *pTwiddleAReal;
  const q15_t *pTwiddleBReal;
  const arm_cfft_instance_q15 *pCfft;
} arm_rfft_instance_q15;
/* same for q31 (transform_functions.h, arm_rfft_instance_q31) */
typedef struct {
  uint32_t fftLenReal;
  uint8_t ifftFlagR;
  uint8_t bitReverseFlagR;
  uint32_t twidCoefRModifier;
  const q31_t *pTwiddleAReal;
  const q31_t *pTwiddleBReal;
  const arm_cfft_instance_q31 *pCfft;
} arm_rfft_instance_q31;
/* arm_rfft_init_q15.c:98-124 and the length switch at its end; arm_rfft_init_q31.c likewise.
 * fftLenReal 32 .. 8192.  16384 is refused: the fork's entry (arm_rfft_init_q15.c:152, "fixme" in
 * arm_rfft_init_q31.c:150) reuses modifier 1 on the 8192-scalar realCoef tables and reads past them. */
arm_status arm_rfft_init_q15(arm_rfft_instance_q15 *S, uint32_t fftLenReal, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_q31(arm_rfft_instance_q31 *S, uint32_t fftLenReal, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_32_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_64_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_128_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_256_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_512_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_1024_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_2048_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_4096_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_8192_q15(arm_rfft_instance_q15 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_32_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_64_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_128_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_256_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_512_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_1024_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_2048_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_4096_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
arm_status arm_rfft_init_8192_q31(arm_rfft_instance_q31 *S, uint32_t ifftFlagR, uint32_t bitReverseFlag);
/* arm_rfft_q15.c:148-182 (split functions :352-405, :552-579), arm_rfft_q31.c (non-Neon body).
 * Forward (ifftFlagR 0): fftLenReal scalars in, 2*fftLenReal scalars out (spectrum and its mirror),
 * down-scaled as documented there.  Inverse: fftLenReal + 2 scalars read, fftLenReal written.
 * Bit-exact with the reference's generic-C branch; pSrc is NOT clobbered (the reference's forward
 * path transforms it in place). */
void arm_rfft_q15(const arm_rfft_instance_q15 *S, q15_t *pSrc, q15_t *pDst);
void arm_rfft_q31(const arm_rfft_instance_q31 *S, q31_t *pSrc, q31_t *pDst);
/* extensions: batch (transform b at pSrc + in_len*b, pDst + out_len*b with the lengths above) */
arm_status arm_rfft_batch_q15(const arm_rfft_instance_q15 *S, const q15_t *pSrc, q15_t *pDst, uint32_t batch);
arm_status arm_rfft_batch_q31(const arm_rfft_instance_q31 *S, const q31_t *pSrc, q31_t *pDst, uint32_t batch);
arm_status arm_rfft_q15_dev(const arm_rfft_instance_q15 *S, const q15_t *pSrc_dev, q15_t *pDst_dev, uint32_t batch, void *stream);
arm_status arm_rfft_q31_dev(const arm_rfft_instance_q31 *S, const q31_t *pSrc_dev, q31_t *pDst_dev, uint32_t batch, void *stream);

/* ---- f32 complex FFT, deprecated radix-2 API (what the fork's fft_c32 calls, CMSIS/CMSIS.cpp:218-234) ---- */
/* Include/dsp/transform_functions.h:217-227 */
typedef struct {
  uint16_t fftLen;
  uint8_t ifftFlag;
  uint8_t bitReverseFlag;
  const float32_t *pTwiddle;
  const uint16_t *pBitRevTable;
  uint16_t twidCoefModifier;
  uint16_t bitRevFactor;
  float32_t onebyfftLen;
} arm_cfft_radix2_instance_f32;
/* Include/dsp/transform_functions.h:231-235; body arm_cfft_radix2_init_f32.c:64-218 (16..16384 on the
 * fork's shared 16384 tables).  Like the reference, the fields are written even when the length
 * is refused with ARM_MATH_ARGUMENT_ERROR. */
arm_status arm_cfft_radix2_init_f32(arm_cfft_radix2_instance_f32 *S, uint16_t fftLen, uint8_t ifftFlag, uint8_t bitReverseFlag);
/* Include/dsp/transform_functions.h:238-240; body arm_cfft_radix2_f32.c:64-88 (butterflies :227-271,
 * :392-463, arm_bitreversal_f32 arm_bitreversal.c:46-105).  Inverse scaled by onebyfftLen;
 * bitReverseFlag 0 leaves plain bit-reversed order. */
void arm_cfft_radix2_f32(const arm_cfft_radix2_instance_f32 *S, float32_t *pSrc);
arm_status arm_cfft_radix2_batch_f32(const arm_cfft_radix2_instance_f32 *S, float32_t *pSrc, uint32_t batch);
arm_status arm_cfft_radix2_f32_dev(const arm_cfft_radix2_instance_f32 *S, float32_t *pSrc_dev, uint32_t batch, void *stream);

/* ---- large complex f32 FFT, 2^15 <= fftLen <= 2^22 (extension; BASELINE config 4) ----
 * No reference function runs these sizes: arm_cfft_f32 stops at 4096 (arm_cfft_f32.c:1263-1280) and
 * fftLen is a uint16_t in every reference instance (transform_functions.h:416-430), hence the
 * widened instance.  Four-step decomposition through HBM; needs a caller-provided device workspace
 * of the same size as the data (batch * fftLen * 8 bytes).  fftLen <= 16384 is routed to the
 * single-kernel path and ignores the workspace. */
typedef struct {
  uint32_t fftLen;
} arm_cfft_instance_u32_f32;
arm_status arm_cfft_init_u32_f32(arm_cfft_instance_u32_f32 *S, uint32_t fftLen);
arm_status arm_cfft_u32_f32_dev(const arm_cfft_instance_u32_f32 *S, float32_t *p1_dev, float32_t *work_dev, uint32_t batch,
                                uint8_t ifftFlag, void *stream);
/* Allocation-free variant of arm_cfft_u32_f32_dev.  No counterpart in the reference; it exists for callers that
 * cannot call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * The four-step intermediate goes to a workspace built into the library (a 64 MiB device array per GPU, held
 * until the process ends), so the call takes no workspace argument and calls no hipMalloc / hipFree or other
 * allocating routine.  Limits: 2^15 <= fftLen <= 2^22 and batch * fftLen <= 2^23 points (larger requests are
 * refused with ARM_MATH_ARGUMENT_ERROR and a message); arm_cfft_u32_f32_noalloc_prepare(fftLen) must have run
 * once on the device before (set-up: it may allocate the twiddle tables).  Every call on a GPU shares that GPU's
 * workspace: calls that overlap on one GPU (different streams without ordering) race; serialise them.
 * Results are bit-identical to arm_cfft_u32_f32_dev (same kernels). */
arm_status arm_cfft_u32_f32_noalloc_prepare(uint32_t fftLen);
arm_status arm_cfft_u32_f32_noalloc_dev(const arm_cfft_instance_u32_f32 *S, float32_t *p1_dev, uint32_t batch, uint8_t ifftFlag,
                                        void *stream);
/* host pointers; allocates its own device staging */
arm_status arm_cfft_u32_batch_f32(const arm_cfft_instance_u32_f32 *S, float32_t *p1, uint32_t batch, uint8_t ifftFlag);

/* ---- GNSS acquisition (BASELINE config 5) ----
 * The reference searches one (PRN, Doppler) cell row at a time on the CPU (read_ors,
 * CMSIS/CMSIS.cpp:706-737; coherent fold :1154-1169): replica = synth_gps_prn, three fft_c32,
 * S*conj(R), find_top2_peaks_cplx, compute_snr_cplx, InterpolateCodePhase.  Here one fused kernel
 * does every row of every scene; the tuple below is what the reference keeps per row. */
typedef struct {
  float32_t peak;            /* top2_pks.val1: largest correlation magnitude (IFFT scaled by 1/fft_len) */
  float32_t second;          /* top2_pks.val2 (-1e10 when there is none, gnss_codes.c:1148) */
  int32_t code_phase;        /* top2_pks.idx1, in samples */
  int32_t second_phase;      /* top2_pks.idx2 (-1 when there is none) */
  int32_t doppler_bin;       /* index into the plan's Doppler list */
  int32_t prn;
  float32_t snr_db;          /* compute_snr_cplx, gnss_codes.c:1110-1125 */
  float32_t code_phase_interp; /* InterpolateCodePhase, gnss_codes.c:1096-1108 */
} gnss_acq_result;

/* Include/gnss_codes.h:42-47 (top2_pks) with 32-bit indices */
typedef struct {
  float32_t val1;
  int32_t idx1;
  float32_t val2;
  int32_t idx2;
} gnss_top2;

typedef struct gnss_acq_plan gnss_acq_plan;

/* Plan = geometry + PRN list + Doppler grid; generates the C/A codes (getCode, gnss_codes.c:969-1030)
 * and their spectra on the current device.  fft_len in {1024..16384}; one millisecond is
 * samples_per_ms samples (FS = 1.023e6 * samples_per_chip) zero-padded to fft_len (CMSIS.cpp:1154-1160);
 * n_ms (<= 64) milliseconds are folded coherently; PRN 1..37 (<= 64 of them). */
arm_status gnss_acq_plan_create(gnss_acq_plan **plan, uint32_t fft_len, uint32_t samples_per_ms, uint32_t n_ms,
                                uint32_t samples_per_chip, const int32_t *prns, uint32_t n_prn, const float32_t *dopplers_hz,
                                uint32_t n_dop);
void gnss_acq_plan_destroy(gnss_acq_plan *plan);
/* minimum distance between the two reported peaks, default 3 (the pk_sep of CMSIS.cpp:731) */
arm_status gnss_acq_plan_set_peak_separation(gnss_acq_plan *plan, int32_t pk_sep);
/* device pointer to the n_prn x fft_len complex code spectra (read-only; for tests) */
const float32_t *gnss_acq_plan_code_spectra_dev(const gnss_acq_plan *plan);

/* signal_dev: n_scenes x (n_ms * samples_per_ms) complex f32 (interleaved I/Q, the c32 of
 * Include/cplx_types.h:9-12).  Searches Doppler bins [dop_begin, dop_end) of the plan (the unit a
 * multi-GPU caller shards on) and writes all_dev[(scene * n_prn + p) * n_dop + d] for those bins;
 * best_dev (optional) [scene * n_prn + p] = the bin with the largest peak among them, lowest bin
 * on ties.  Asynchronous on `stream`, allocates nothing. */
arm_status gnss_acquire_dev(const gnss_acq_plan *plan, const float32_t *signal_dev, uint32_t n_scenes, uint32_t dop_begin,
                            uint32_t dop_end, gnss_acq_result *all_dev, gnss_acq_result *best_dev, void *stream);
/* same on the fork's 2-bit packed recordings: ors_dev = n_scenes x
 * ceil(n_ms * samples_per_ms / 2) bytes, one byte = two complex samples decoded as DecodeOrsIQCplx does
 * (CMSIS/CMSIS.cpp:465-485: bits [1:0] I0, [3:2] Q0, [5:4] I1, [7:6] Q1, code -> {+1, -1, +3, -3}); the
 * decode happens inside the wipe-off load, the float signal never exists in memory. */
arm_status gnss_acquire_ors_dev(const gnss_acq_plan *plan, const uint8_t *ors_dev, uint32_t n_scenes, uint32_t dop_begin,
                                uint32_t dop_end, gnss_acq_result *all_dev, gnss_acq_result *best_dev, void *stream);
/* Non-coherent and differential accumulation.  The plan's n_ms milliseconds are cut
 * into n_groups groups (n_groups divides n_ms); each group is folded coherently and correlated, then
 *   GNSS_ACQ_NONCOHERENT:  r[m] = sum_g |y_g[m]|                        (NCI loop, CMSIS/CMSIS.cpp:915-943)
 *   GNSS_ACQ_DIFFERENTIAL: r[m] = |sum_{g>=1} y_g[m] conj(y_{g-1}[m])|  (test_quasi_diff_pilot, :1030-1062)
 * and find_top2_peaks_real / compute_snr_real / InterpolateCodePhase run on r (:948-951).  work_dev =
 * gnss_acq_workspace_bytes(plan, n_scenes, dop_end - dop_begin, n_groups) bytes of caller-owned device
 * memory (the group spectra); nothing is allocated. */
typedef enum { GNSS_ACQ_COHERENT = 0, GNSS_ACQ_NONCOHERENT = 1, GNSS_ACQ_DIFFERENTIAL = 2 } gnss_acq_mode;
size_t gnss_acq_workspace_bytes(const gnss_acq_plan *plan, uint32_t n_scenes, uint32_t n_dop_bins, uint32_t n_groups);
arm_status gnss_acquire_groups_dev(const gnss_acq_plan *plan, const float32_t *signal_dev, uint32_t n_scenes, uint32_t dop_begin,
                                   uint32_t dop_end, int32_t mode, uint32_t n_groups, void *work_dev, gnss_acq_result *all_dev,
                                   gnss_acq_result *best_dev, void *stream);
/* DecodeOrsIQCplx itself (CMSIS/CMSIS.cpp:465-485) on device pointers: 2 * byte_length complex samples out */
arm_status gnss_decode_ors_iq_dev(const uint8_t *data_dev, uint32_t byte_length, float32_t *iqs_dev, void *stream);
/* host pointers, all bins; `all` may be NULL */
arm_status gnss_acquire(gnss_acq_plan *plan, const float32_t *signal, uint32_t n_scenes, gnss_acq_result *best,
                        gnss_acq_result *all);

/* Multi-GPU search from a plain C host (no Python, no RCCL): plans[i] was created with GPU i current
 * (same geometry, PRN list and Doppler grid everywhere).  Plan i searches the i-th contiguous share of the Doppler
 * bins on its own device from its own host thread; the signal is copied to every device, nothing crosses between
 * GPUs, and the per-(scene, PRN) winners are merged on the host by the single-GPU rule (largest peak, lowest bin
 * on ties): best[scene * n_prn + p] equals what gnss_acquire returns on one plan.  Replaces the reference's serial
 * (prn, doppler) loop, CMSIS/CMSIS.cpp:706-737.  Two plans on ONE device are allowed (they share it). */
arm_status gnss_acquire_multi(gnss_acq_plan *const *plans, uint32_t n_plans, const float32_t *signal, uint32_t n_scenes,
                              gnss_acq_result *best);

/* Device-side counterparts of the reference's helpers, exported so each can be checked on its own:
 * getCode (gnss_codes.c:969-1030): out_dev[num * samples_per_chip] ints, +1/-1;
 * synth_gps_prn (gnss_codes.c:1066-1071, mix_prn :1032-1063): replica_dev[size] complex;
 * find_top2_peaks_real / _cplx (gnss_codes.c:1144-1190): `batch` rows of data_size samples,
 * same values AND indices as the sequential reference loop, ties included (data_size <= 16384). */
arm_status gnss_get_code_dev(int32_t num, int32_t samples_per_chip, int32_t prn, int32_t *out_dev);
arm_status gnss_synth_gps_prn_dev(int32_t prn, float32_t doppler, uint32_t size, float32_t *replica_dev, int32_t samples_per_chip);
arm_status gnss_find_top2_peaks_dev(const float32_t *data_dev, int32_t is_complex, int32_t data_size, int32_t pk_sep, uint32_t batch,
                                    gnss_top2 *peaks_dev, void *stream);
/* Carrier mixing on a CALLER-SUPPLIED code of `size` ints (one per sample, normally +1 / -1), device pointers,
 * asynchronous on `stream`, no allocation:
 *   gnss_mix_prn_dev       mix_prn (gnss_codes.c:1032-1063): signal[n] = code[n] e^{j (phase + 2 PI f n / FS)}, FS = 1.023e6 * spc,
 *                          with the reference's FLOAT literal PI (Include/gnss_codes.h:33) in both places;
 *   gnss_make_replica_dev  make_replica (quasi_pilot_tests.c:153-181, Include/gnss_codes.h:110): the same on a caller-given
 *                          sample rate, initial phase 0.0001 degrees, double M_PI;
 *   gnss_rotate_fwd_dev    rotate_fwd (quasi_pilot_tests.c:138-151): out[i] = in[i - offset] (wrapped), out of place,
 *                          0 <= offset <= size.
 * Every sample's phase is evaluated directly in double instead of by the reference's renormalised recursive phasor:
 * results agree with the reference to one float ulp (tests/golden/gnss_replica.npz holds make_replica outputs of the
 * reference's own code compiled here). */
arm_status gnss_mix_prn_dev(const int32_t *code_dev, double doppler_hz, double phase_deg, float32_t *signal_dev, uint32_t size, int32_t spc,
                            void *stream);
arm_status gnss_make_replica_dev(const int32_t *code_dev, uint32_t size, float32_t doppler, float32_t samp_freq, float32_t *out_dev,
                                 void *stream);
arm_status gnss_rotate_fwd_dev(const int32_t *in_dev, int32_t *out_dev, uint32_t size, int32_t offset, void *stream);

/* Fine search around an acquired cell: recover_prn_phase_deg_with_doppler (CMSIS/CMSIS.cpp:362-434).  For f = -1, 0, +1 Hz the
 * millisecond is wiped at doppler + f with the reference's FLOAT recursive phasor, correlated against the code at the 21
 * offsets within +-10 samples of code_offset (wrapping around the code period) and the largest |correlation| kept, the
 * first among equals in the reference's visiting order.  One result per satellite:
 *   phase_deg = atan2(im, re) * (180.0f / 3.14159265358979f), best_off = the offset, doppler = the winning f, pwr = |sum|.
 * signal_dev + k * signal_stride (complex samples; stride 0: all satellites search the same `size` samples), codes_dev =
 * n_sat x size ints (+1 / -1), size <= 1023 * 4 (the reference's static buffers, :392).  Asynchronous on `stream`, no allocation.
 * Parity unpinned by the reference (the harness cannot be compiled here, it holds no vector): oracle/gnss_oracle.c restates it. */
typedef struct {
  double phase_deg;
  float32_t pwr;
  float32_t doppler;
  int32_t best_off;
  int32_t reserved;
} gnss_refine_result;
arm_status gnss_refine_dev(const float32_t *signal_dev, uint32_t signal_stride, uint32_t size, const int32_t *codes_dev,
                           const int32_t *code_offsets_dev, const double *dopplers_dev, double sample_rate_hz, uint32_t n_sat,
                           gnss_refine_result *out_dev, void *stream);

/* The reference's own names for those helpers (Include/gnss_codes.h:62-126), host pointers; `c32` arrays are passed
 * as interleaved float32_t I/Q (Include/cplx_types.h:9: struct { float r, i; }, same memory).  Each stages its
 * arguments to the current device, runs the kernel above and copies the result back.
 *   getCode                 gnss_codes.c:969-1030   out[num * samplesPerChip] ints, +1 / -1
 *   synth_gps_prn           :1066-1071 (mix_prn :1032-1063)   replica[size] c32; carrier by direct evaluation of the
 *                           double phase instead of the reference's renormalised recursive phasor (<= 1.2e-7 apart)
 *   mix_prn                 :1032-1063 (external linkage in gnss_codes.c, not declared in gnss_codes.h)
 *   make_replica, rotate_fwd   quasi_pilot_tests.c:153-181, :138-151 (Include/gnss_codes.h:110-111); rotate_fwd leaves the array
 *                           unchanged (message in arm_fft_mi355_last_error) for an offset outside 0..size, where the
 *                           reference reads outside the array
 *   find_top2_peaks_cplx / _real   :1144-1190   same values and indices as the sequential loop, ties included;
 *                           data_size <= 16384 (the fork's largest correlation; -1 beyond it -- the reference has no limit);
 *                           a negative pk_sep behaves as 0, which is what the reference's comparison makes of it;
 *                           fp_out (optional) receives the reference's "m, magnitude" dump;
 *                           returns -1 on a bad argument or when the GPU cannot run (0 = success)
 *   compute_snr_cplx / _real       :1110-1142   10 log10(peak / mean off-peak magnitude); cov_size <= 0 returns what the
 *                           reference's arithmetic gives without reading data (an infinity or a NaN), it does not abort
 *   InterpolateCodePhase    :1096-1108   scalar formula on three doubles (host arithmetic: no data to move)
 * The void / double entry points abort() with a message when the GPU cannot run them (see Conventions). */
typedef struct {          /* Include/gnss_codes.h:42-47 */
  size_t idx1;
  float val1;
  size_t idx2;
  float val2;
} top2_pks;
void getCode(int num, int samplesPerChip, const int prn, int *out);
void synth_gps_prn(int prn, float doppler, size_t size, float32_t *replica, int spc);
void mix_prn(const int32_t *prn_a, const double doppler_a_hz, const double phase_a_deg, float32_t *signal, int size, int spc);
void make_replica(const int32_t *prn_a, float32_t *out_iandq, float doppler, int size, float samp_freq);
void rotate_fwd(int array[], int size, int offset);
int find_top2_peaks_cplx(const float32_t *data, int data_size, int pk_sep, top2_pks *peaks, FILE *fp_out);
int find_top2_peaks_real(const float32_t *data, int data_size, int pk_sep, top2_pks *peaks, FILE *fp_out);
double compute_snr_cplx(float32_t *convol, int cov_size, double peak, int peak_loc);
double compute_snr_real(float32_t *convol, int cov_size, double peak, int peak_loc);
double InterpolateCodePhase(uint32_t index, double earlyPower, double promptPower, double latePower);
/* CMSIS/CMSIS.cpp:362-434 under the harness's own name and argument list (iandq = c32[SIZE] as interleaved floats); aborts with a
 * message when the GPU cannot run it or SIZE is outside 21 .. 4092 */
double recover_prn_phase_deg_with_doppler(const float32_t *iandq, int SIZE, const int prn_a[], int code_offset, double doppler_a_hz,
                                          double sampe_rate_hz, int *best_offset_out, float *doppler_out, float *pwr);

/* ---- codes of any length through caller-supplied replicas ----
 * The fork correlates L5 / E5a codes (10230 chips at 10.23 Msps) by resampling signal and replica
 * 10230 -> 16384 per millisecond (up_sample_10k_to_16k, CMSIS/CMSIS.cpp:906, :934) and running the same
 * FFT correlation.  gnss_acq_plan_create_replicas builds a plan from the time-domain replicas
 * themselves -- n_prn x fft_len complex f32 on the device, one code period at zero Doppler,
 * zero-padded if samples_per_ms < fft_len -- so any code the caller can produce is searched by the
 * same fused kernel: gnss_acquire_dev / _groups_dev / gnss_acquire work unchanged on the plan
 * (signal = n_ms x samples_per_ms complex per scene, sample rate samples_per_ms kHz; prn_labels
 * only label the result rows).  For L5: gnss_l5_generate_I_dev -> complex (chip, 0) ->
 * iq_resamp_zero_phase_dev(10230e3, 16384e3, 12, 0.45) -> replicas; the received milliseconds go
 * through the same resampler call (batch = n_ms).  No counterpart as ONE function in the reference
 * (its L5 loop is inline in CMSIS.cpp:895-966). */
arm_status gnss_acq_plan_create_replicas(gnss_acq_plan **plan, uint32_t fft_len, uint32_t samples_per_ms, uint32_t n_ms,
                                         const float32_t *replicas_dev, const int32_t *prn_labels, uint32_t n_prn,
                                         const float32_t *dopplers_hz, uint32_t n_dop);
/* l5_generate_I(prn, out, n_chips, start_at_epoch = true, NULL) (gnss_codes.c:153-204): n_chips chips
 * (+1 / -1 as int8) of the fork's L5 I-channel generator for PRN 1..37, on the device.  The
 * registers are not reset at the 10230-chip boundary (as in the reference). */
arm_status gnss_l5_generate_I_dev(int32_t prn, uint32_t n_chips, int8_t *out_dev);

/* Galileo E1-B CBOC replica: synth_e1b_prn (gnss_codes.c:920-965) = code chip x
 * cboc_e1b_weight (:253-260: sqrt(10/11) BOC(1,1) - sqrt(1/11) BOC(6,1), boc_sign :247-251) x carrier at
 * `doppler`, fs = 4.092 MHz, code and carrier phase 0, data bit +1, with the reference's float phase
 * recursions.  The reference reads the code table from C:/work/Baseband/HEX_E1B.txt (:930), which is not
 * part of it; here the caller supplies the chips: codes_dev = n_codes x 4092 chips (+1 / -1 as int8,
 * the rows of E1B_Code[][]) on the device, replica_dev = n_codes x size complex f32.  With doppler = 0
 * and size = 4096 (4092 samples + the reference's zero padding is NOT added: sample n >= 4092 continues
 * the periodic code) the output feeds gnss_acq_plan_create_replicas.  Asynchronous on `stream`, no allocation. */
arm_status gnss_e1b_cboc_replica_dev(const int8_t *codes_dev, uint32_t n_codes, float32_t doppler, uint32_t size,
                                     float32_t *replica_dev, void *stream);
/* hexstream_to_e1b_chips / _e5a_chips / _e5a_qp_chips (gnss_codes.c:276-325; hex_nibble :234-239,
 * logic_to_chip_icd :243-245), generalised over the code length (4092 E1-B, 10230 E5a ...): the ICD's hex
 * text -> chips, MSB first within a nibble, logic 1 -> -1, any non-hex character skipped.  Host-side.
 * Returns 0, or -1 when the stream holds fewer than code_len bits. */
int gnss_hexstream_to_chips(const char *hex_stream, int8_t *chips, int32_t code_len);

/* ---- zero-phase I/Q resampler ----
 * Include/gnss_codes.h:97-99, bodies gnss_codes.c:868-904 over iq_resamp_zp_init /
 * iq_resamp_process_zero_phase [Include/up_sample.h:20-24, up_sample.c:31-157].  `in`/`out` are
 * the fork's c32 arrays (interleaved float I/Q, Include/cplx_types.h:9).  Same quirk as the
 * reference: up_sample_N_to_M caps the output at llround(N_in * (M_out / N_in)) with the INTEGER
 * quotient (gnss_codes.c:879), i.e. 10230 -> 16384 writes 10230 samples. */
void up_sample_N_to_M(float32_t *in, int N_in, float32_t *out, int M_out);
void up_sample_10k_to_16k(float32_t *in, float32_t *out);
/* extension: the general routine on device pointers.  `batch` blocks, block b at
 * in_dev + 2*n_in*b and out_dev + 2*n_out_cap*b; *n_out (optional) = samples written per block =
 * min(llround(n_in * fout/fin), n_out_cap) (up_sample.c:121-125).  ntaps even, 4..64;
 * 0 < cutoff <= 0.5 (up_sample.c:32-33).  Asynchronous on `stream`, allocates nothing. */
arm_status iq_resamp_zero_phase_dev(double fin_hz, double fout_hz, int32_t ntaps, double cutoff, const float32_t *in_dev, uint32_t n_in,
                                    float32_t *out_dev, uint32_t n_out_cap, uint32_t batch, uint32_t *n_out, void *stream);
uint32_t iq_resamp_zero_phase_count(double fin_hz, double fout_hz, int32_t ntaps, double cutoff, uint32_t n_in, uint32_t n_out_cap);
/* The reference's own resampler object and calls (Include/up_sample.h:11-24; bodies up_sample.c:31-48, :115-157),
 * host pointers, `in` / `out` as interleaved float32_t I/Q: init returns NULL for odd or < 4 taps or a cutoff outside
 * (0, 0.5]; process returns the number of samples written, min(llround(Nin * fout / fin), Nout_cap) (0 on a bad
 * argument); at most 64 taps here.  The filtering runs on the GPU (iq_resamp_zero_phase_dev, one block). */
typedef struct {
  double fin;     /* input sample rate (Hz) */
  double fout;    /* output sample rate (Hz) */
  double dt;      /* fin / fout */
  int ntaps;
  int half;
  double cutoff;
} iq_resamp_zp_t;
iq_resamp_zp_t *iq_resamp_zp_init(double fin_hz, double fout_hz, int ntaps, double cutoff);
size_t iq_resamp_process_zero_phase(iq_resamp_zp_t *st, const float32_t *in, size_t Nin, float32_t *out, size_t Nout_cap);
void iq_resamp_zp_free(iq_resamp_zp_t *st);

/* ---- measurement wire format "msb" (Include/msb_funcs.h, Source/TransformFunctions/msb_funcs.c) ----
 * The fork's downstream tools read acquisition results as a bit-packed record: a 32-bit GPS PRN
 * mask, then per GPS satellite (in the order of sats[]) code phase (20 bits, units of 9.54e-7 ms),
 * Doppler (13 bits signed, 2 Hz) and C/N0 (4 bits, (cno - 28) / 4); 3 bits "Galileo present", and
 * if so a 50-bit PRN mask and 22 + 13 + 4 bits per Galileo satellite (msb_funcs.c:92-145).
 * Host-side byte work on at most 20 satellites: plain C on the CPU, no GPU involved.  Structures
 * are the reference's (msb_funcs.h:16-41). */
#define SYS_GPS 0x01
#define SYS_GAL 0x02
typedef struct {
  double time_tag;
  float pseudorange;
  float code_phase;   /* ms */
  float doppler;      /* Hz */
  float carr_phase;
  float cno;          /* dB-Hz */
  uint16_t prn;
  uint16_t constellation;
} bb_meas_sat_t;
typedef struct {
  uint64_t time;
  double sec;
  int num_sat;
  bb_meas_sat_t sats[20];
} bb_meas_t;
/* msb_funcs.c:92-145: clears bin_buff[0 .. buff_size), packs, returns the number of WHOLE bytes
 * used (bit_pos / 8: trailing bits of a partial last byte are in the buffer but not counted, as
 * in the reference).  Extension: returns -1 instead of overrunning when buff_size is too small
 * or an argument is NULL / num_sat is outside 0..20 (the reference does not check). */
int write_bb_msb(const bb_meas_t *measurements, uint8_t *bin_buff, int buff_size);
/* msb_funcs.c:152-233 without its printf chatter: fills num_sat and sats[] (time_tag from the
 * caller-set time + sec, pseudorange 0, carr_phase untouched); bits past bin_size read as the
 * reference's error value (-1, i.e. all ones from that bit up).  More than 20 satellites in the
 * record are counted in num_sat but not stored (the reference would overrun sats[]). */
void read_bb_msb(uint8_t *bin_buff, int bin_size, bb_meas_t *measurements);
/* The acceptance step between acquisition and the record (CMSIS/CMSIS.cpp:737-748, :954-966):
 * every row of best[0 .. n) whose peak / second exceeds ratio_min is appended to meas->sats[]
 * (up to 20) with code_phase = code_phase_interp / fft_len [ms], doppler = dopplers_hz[bin]
 * (the signal's Doppler: the reference searches with the negated replica Doppler and reports the
 * negation, i.e. the same quantity),
 * cno = snr_db + cno_offset_db (35 in the GPS L1 flow, 0 in the L5 flow).  Returns the number
 * appended, -1 on a bad argument. */
int gnss_acq_to_meas(const gnss_acq_result *best, int n, const float32_t *dopplers_hz, int n_dop, int fft_len, float ratio_min,
                     float cno_offset_db, uint16_t constellation, bb_meas_t *meas);
/* carr_phase (msb_funcs.h:30, cycles) of every satellite in meas: the fine search above on the first `size` samples of `signal`
 * (HOST pointer, interleaved I/Q; size a multiple of spc, <= 4092) around the record's code phase (code_phase * fft_len samples)
 * and Doppler, code = getCode(size / spc, spc, prn); carr_phase = phase_deg / 360.  details (optional, num_sat entries) receives
 * the searches' results.  Returns the number of satellites refined, -1 on a bad argument or when the GPU cannot run. */
int gnss_meas_refine(const float32_t *signal, uint32_t size, int32_t spc, double sample_rate_hz, int32_t fft_len, bb_meas_t *meas,
                     gnss_refine_result *details);

/* ---- library services (no counterpart in the reference) ---- */
/* Select the GPU used by the calling thread's host-pointer entry points and build nothing else;
 * optional (device 0 / the current HIP device is used otherwise). */
arm_status arm_fft_mi355_set_device(int device);
/* Pre-build the device plan (on-device twiddle generation) for one size so that the first timed
 * call does not pay for it: kind 0 = f32, 1 = q15, 2 = q31, 3 = rfft f32. */
arm_status arm_fft_mi355_warm(int kind, uint32_t fftLen);
/* Message of the last failure on the calling thread ("" if none). */
const char *arm_fft_mi355_last_error(void);
/* Name(s) of the kernel(s) the calling thread's last transform or acquisition call launched, spelled as rocprofv3
 * prints them without the namespace and argument list, e.g. "cfft_f32_split_kernel<12, false, 0>" or
 * "fourstep_tile_kernel<11, 1, false, true, 32, 0> + fourstep_tile_kernel<11, 2, false, true, 16, 0>" ("" before the
 * first launch).  bench.py prints it next to its roofline figures so that a line can be matched to a trace. */
const char *arm_fft_mi355_last_kernel(void);
/* Rule the fixed-point twiddle tables (Include/arm_common_tables.h:88-130; definitions missing from the reference's
 * mount) are generated by.  Must be set before the first q15/q31 init; see DESIGN.md section 4.
 *   0  floor(x * 2^15 | 2^31) of the double value, +1.0 saturated (upstream's scalar tables, as recollected)
 *   1  round to nearest, the same otherwise (the two copies the mount holds: arm_mve_tables.c:3769, arm_neon_tables.c:8951)
 *   2  the fork's own generator, twiddleCoefCalculator + to_fixed (CMSIS/CMSIS.cpp:445-458, :309-327): FLOAT angle
 *      i*2*PI/(float)N, float cos / sin, float product with 2^dyad, llround, negatives one lower, clamps
 *   3  (default) per size: rule 0 up to 4096 points, rule 2 at 8192 / 16384 -- the sizes the fork added with that generator
 * Returns ARM_MATH_ARGUMENT_ERROR for another value or once a fixed-point table exists. */
arm_status arm_fft_mi355_set_twiddle_rounding(int mode);
const char *arm_fft_mi355_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ARM_FFT_MI355_H */
