/*
 * ref_shim.c -- thin entry points into the REAL reference code for oracle/_ref/libcmsis_ref.so.
 * TEST INFRASTRUCTURE ONLY; compiled only where /root/reference exists (oracle/Makefile `ref`).
 *
 * The reference's init functions pull every table of every precision through
 * Source/CommonTables/arm_const_structs.c, most of which are irrelevant here and all of which are
 * missing (arm_common_tables.c is not in the mount).  So instances are filled by hand with the
 * regenerated tables of ref_tables.c -- exactly what arm_cfft_init_<N>_* would copy out of the
 * const structs (arm_cfft_init_f32.c:116-136) -- and the reference's own execute functions run.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "arm_math_types.h"
#include "dsp/transform_functions.h"

#include "arm_common_tables.h" /* declarations only; definitions come from ref_tables.c */
extern int orc_ref_table_lengths_ok;

/* lengths: Include/arm_common_tables.h:203-263 */
static int inst_f32(arm_cfft_instance_f32 *S, uint32_t N)
{
  S->fftLen = (uint16_t)N;
  switch (N) {
#define C(N, L) case N: S->pTwiddle = twiddleCoef_##N; S->pBitRevTable = armBitRevIndexTable##N; S->bitRevLength = L; return 0;
    C(16, 20) C(32, 48) C(64, 56) C(128, 208) C(256, 440) C(512, 448) C(1024, 1800) C(2048, 3808) C(4096, 4032)
#undef C
  }
  return -1;
}

static int inst_q15(arm_cfft_instance_q15 *S, uint32_t N)
{
  S->fftLen = (uint16_t)N;
  switch (N) {
#define C(N, L) case N: S->pTwiddle = twiddleCoef_##N##_q15; S->pBitRevTable = armBitRevIndexTable_fixed_##N; S->bitRevLength = L; return 0;
    C(16, 12) C(32, 24) C(64, 56) C(128, 112) C(256, 240) C(512, 480) C(1024, 992) C(2048, 1984) C(4096, 4032) C(8192, 8128)
#undef C
  }
  return -1;
}

static int inst_q31(arm_cfft_instance_q31 *S, uint32_t N)
{
  S->fftLen = (uint16_t)N;
  switch (N) {
#define C(N, L) case N: S->pTwiddle = twiddleCoef_##N##_q31; S->pBitRevTable = armBitRevIndexTable_fixed_##N; S->bitRevLength = L; return 0;
    C(16, 12) C(32, 24) C(64, 56) C(128, 112) C(256, 240) C(512, 480) C(1024, 992) C(2048, 1984) C(4096, 4032)
#undef C
  }
  return -1;
}

int ref_tables_ok(void) { return orc_ref_table_lengths_ok; }

int ref_cfft_f32(float *p, uint32_t N, uint32_t batch, int ifft, int bitrev)
{
  arm_cfft_instance_f32 S;
  if (inst_f32(&S, N)) return -1;
  for (uint32_t b = 0; b < batch; b++) arm_cfft_f32(&S, p + (size_t)2 * N * b, (uint8_t)ifft, (uint8_t)bitrev);
  return 0;
}

int ref_cfft_q15(int16_t *p, uint32_t N, uint32_t batch, int ifft, int bitrev)
{
  arm_cfft_instance_q15 S;
  if (inst_q15(&S, N)) return -1;
  for (uint32_t b = 0; b < batch; b++) arm_cfft_q15(&S, p + (size_t)2 * N * b, (uint8_t)ifft, (uint8_t)bitrev);
  return 0;
}

int ref_cfft_q31(int32_t *p, uint32_t N, uint32_t batch, int ifft, int bitrev)
{
  arm_cfft_instance_q31 S;
  if (inst_q31(&S, N)) return -1;
  for (uint32_t b = 0; b < batch; b++) arm_cfft_q31(&S, p + (size_t)2 * N * b, (uint8_t)ifft, (uint8_t)bitrev);
  return 0;
}

int ref_rfft_fast_f32(float *p, float *out, uint32_t N, uint32_t batch, int ifft)
{
  arm_rfft_fast_instance_f32 S;
  if (inst_f32(&S.Sint, N / 2)) return -1;
  S.fftLenRFFT = (uint16_t)N;
  switch (N) {
#define C(N) case N: S.pTwiddleRFFT = twiddleCoef_rfft_##N; break;
    C(32) C(64) C(128) C(256) C(512) C(1024) C(2048) C(4096)
#undef C
    default: return -1;
  }
  for (uint32_t b = 0; b < batch; b++) arm_rfft_fast_f32(&S, p + (size_t)N * b, out + (size_t)N * b, (uint8_t)ifft);
  return 0;
}

/* the fork's radix-2 path (what fft_c32 calls, CMSIS/CMSIS.cpp:218-234); instance filled as
 * arm_cfft_radix2_init_f32.c:92-207 does for the 16384-based shared table */
int ref_cfft_radix2_f32(float *p, uint32_t N, uint32_t batch, int ifft, int bitrev)
{
  arm_cfft_radix2_instance_f32 S;
  if (N < 16 || N > 16384 || (N & (N - 1))) return -1;
  uint32_t mod = 16384u / N;
  S.fftLen = (uint16_t)N;
  S.ifftFlag = (uint8_t)ifft;
  S.bitReverseFlag = (uint8_t)bitrev;
  S.pTwiddle = twiddleCoef_16384;
  S.twidCoefModifier = (uint16_t)mod;
  S.bitRevFactor = (uint16_t)mod;
  S.pBitRevTable = &armBitRevTable[mod - 1];
  S.onebyfftLen = 1.0f / (float)N;
  for (uint32_t b = 0; b < batch; b++) arm_cfft_radix2_f32(&S, p + (size_t)2 * N * b);
  return 0;
}

/* ---- arm_rfft_q15 / arm_rfft_q31 (arm_rfft_q15.c:148-182, arm_rfft_q31.c): instances filled by hand
 * the way RFFTINIT_Q15/Q31 do (arm_rfft_init_q15.c:98-124), because arm_const_structs.c drags in
 * every table of the library ---- */
int ref_rfft_q15(int16_t *src, int16_t *dst, uint32_t n_real, int ifft, int bitrev)
{
  arm_cfft_instance_q15 C;
  arm_rfft_instance_q15 S;
  uint32_t mod = 8192u / n_real;
  if (n_real < 32 || n_real > 8192 || (n_real & (n_real - 1)) || inst_q15(&C, n_real / 2)) return -1;
  S.fftLenReal = n_real;
  S.pTwiddleAReal = realCoefAQ15;
  S.pTwiddleBReal = realCoefBQ15;
  S.ifftFlagR = (uint8_t)ifft;
  S.bitReverseFlagR = (uint8_t)bitrev;
  S.twidCoefRModifier = mod;
  S.pCfft = &C;
  arm_rfft_q15(&S, src, dst);
  return 0;
}

int ref_rfft_q31(int32_t *src, int32_t *dst, uint32_t n_real, int ifft, int bitrev)
{
  arm_cfft_instance_q31 C;
  arm_rfft_instance_q31 S;
  uint32_t mod = 8192u / n_real;
  if (n_real < 32 || n_real > 8192 || (n_real & (n_real - 1)) || inst_q31(&C, n_real / 2)) return -1;
  S.fftLenReal = n_real;
  S.pTwiddleAReal = realCoefAQ31;
  S.pTwiddleBReal = realCoefBQ31;
  S.ifftFlagR = (uint8_t)ifft;
  S.bitReverseFlagR = (uint8_t)bitrev;
  S.twidCoefRModifier = mod;
  S.pCfft = &C;
  arm_rfft_q31(&S, src, dst);
  return 0;
}
