/* fft_c32_caller.c -- a plain C host program built with gcc against include/arm_fft_mi355.h and
 * -larmfft_mi355: the "existing caller relinks unchanged" claim of INTEGRATION.md exercised without
 * Python or ctypes.  It runs, on a golden vector of the reference's own test patterns,
 *   (1) arm_cfft_init_f32 + arm_cfft_f32                 (Testing/Source/Tests/TransformCF32.cpp:10-46)
 *   (2) the fork's fft_c32 sequence, CMSIS/CMSIS.cpp:218-234: c32[] -> float[] copy,
 *       arm_cfft_radix2_init_f32, arm_cfft_radix2_f32, copy back -- forward, then inverse
 * and checks the results with the reference's thresholds (SNR >= 120 dB, |err| <= 8e-5 + 2e-5 |ref|).
 *
 * usage: fft_c32_caller <n> <input.f32> <expected_fft.f32>     (raw little-endian float32, 2*n each)
 * exit 0 = all checks passed.  tests/test_gpu_c_caller.py writes the files from tests/golden. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "arm_fft_mi355.h"

typedef struct { float r, i; } c32; /* Include/cplx_types.h:9-12 */

static int fft_c32(int size, c32 *w, int fwd) /* CMSIS/CMSIS.cpp:218-234 */
{
  float *data = (float *)malloc(sizeof(float) * 2 * (size_t)size);
  if (!data) return -1;
  for (int i = 0; i < size; i++) { data[2 * i] = w[i].r; data[2 * i + 1] = w[i].i; }
  arm_cfft_radix2_instance_f32 s;
  arm_status st = arm_cfft_radix2_init_f32(&s, (uint16_t)size, fwd ? 0 : 1, 1);
  if (st == ARM_MATH_SUCCESS) arm_cfft_radix2_f32(&s, data);
  for (int i = 0; i < size; i++) { w[i].r = data[2 * i]; w[i].i = data[2 * i + 1]; }
  free(data);
  return st == ARM_MATH_SUCCESS ? 0 : -1;
}

static int read_f32(const char *path, float *dst, size_t count)
{
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); return -1; }
  size_t got = fread(dst, sizeof(float), count, f);
  fclose(f);
  return got == count ? 0 : -1;
}

/* the reference's two assertions (Testing/FrameworkSource/Error.cpp: SNR and close-error) */
static int check(const char *what, const float *got, const float *ref, size_t count, double snr_min, double abs_e, double rel_e)
{
  double e = 0.0, d = 0.0, worst = 0.0;
  int bad = 0;
  for (size_t k = 0; k < count; k++) {
    const double diff = (double)got[k] - (double)ref[k];
    e += (double)ref[k] * ref[k];
    d += diff * diff;
    if (fabs(diff) > abs_e + rel_e * fabs((double)ref[k])) bad++;
    if (fabs(diff) > worst) worst = fabs(diff);
  }
  const double snr = d == 0.0 ? 1e5 : 10.0 * log10(e / d);
  printf("%-34s SNR %.1f dB, worst |err| %.3g, %d samples outside the bound\n", what, snr, worst, bad);
  return (snr >= snr_min && bad == 0) ? 0 : 1;
}

int main(int argc, char **argv)
{
  if (argc != 4) { fprintf(stderr, "usage: %s <n> <input.f32> <expected_fft.f32>\n", argv[0]); return 2; }
  const int n = atoi(argv[1]);
  if (n < 16 || n > 16384) return 2;
  const size_t count = 2 * (size_t)n;
  float *in = (float *)malloc(sizeof(float) * count), *want = (float *)malloc(sizeof(float) * count), *buf = (float *)malloc(sizeof(float) * count);
  c32 *w = (c32 *)malloc(sizeof(c32) * (size_t)n);
  if (!in || !want || !buf || !w) return 2;
  if (read_f32(argv[2], in, count) || read_f32(argv[3], want, count)) return 2;
  int fails = 0;

  /* (1) the modern API */
  arm_cfft_instance_f32 S;
  if (arm_cfft_init_f32(&S, (uint16_t)n) != ARM_MATH_SUCCESS) { fprintf(stderr, "arm_cfft_init_f32(%d) refused\n", n); return 1; }
  if (S.fftLen != n || !S.pTwiddle) { fprintf(stderr, "instance not filled\n"); return 1; }
  memcpy(buf, in, sizeof(float) * count);
  arm_cfft_f32(&S, buf, 0, 1);
  fails += check("arm_cfft_f32 forward", buf, want, count, 120.0, 8e-5, 2e-5);
  arm_cfft_f32(&S, buf, 1, 1);
  fails += check("arm_cfft_f32 inverse(forward(x))", buf, in, count, 120.0, 8e-5, 2e-5);

  /* (2) the fork's glue */
  for (int i = 0; i < n; i++) { w[i].r = in[2 * i]; w[i].i = in[2 * i + 1]; }
  if (fft_c32(n, w, 1)) { fprintf(stderr, "fft_c32 forward failed: %s\n", arm_fft_mi355_last_error()); return 1; }
  fails += check("fft_c32 forward (radix-2 API)", (const float *)w, want, count, 120.0, 8e-5, 2e-5);
  if (fft_c32(n, w, 0)) { fprintf(stderr, "fft_c32 inverse failed: %s\n", arm_fft_mi355_last_error()); return 1; }
  fails += check("fft_c32 inverse(forward(x))", (const float *)w, in, count, 120.0, 8e-5, 2e-5);

  /* an unsupported length is refused by init and leaves the data alone in execute, as in the reference */
  arm_cfft_instance_f32 B;
  memset(&B, 0, sizeof B);
  if (arm_cfft_init_f32(&B, 48) != ARM_MATH_ARGUMENT_ERROR || B.fftLen != 0) { fprintf(stderr, "init accepted length 48\n"); fails++; }
  printf("%s\n", fails ? "FAILED" : "all C-caller checks passed");
  free(in); free(want); free(buf); free(w);
  return fails ? 1 : 0;
}
