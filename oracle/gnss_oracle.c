/*
 * gnss_oracle.c -- scalar-C restatement of the fork's GNSS code-correlation path.
 * TEST INFRASTRUCTURE ONLY (see armfft_oracle.h).
 *
 * The reference holds no golden vector for this path ("parity unpinned by the reference"):
 * getCode is pinned by the public IS-GPS-200 first-10-chips table (tests/test_oracle_gnss.py),
 * everything else by synthetic scenes with known truth.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "armfft_oracle.h"

#define ORC_PI 3.14159265358979323846

/* GPS L1 C/A Gold code: getCode (gnss_codes.c:969-1030).  Two 10-stage LFSRs, all-ones start,
 * G1 feedback from stages {2,9}, G2 from {1,2,5,7,8,9} (0-based), chip = G1[9]^G2[s0]^G2[s1],
 * logic 1 -> +1, logic 0 -> -1 (:1016-1018), sample-and-hold by samples_per_chip. */
static const int k_sats[37][2] = {
    {1, 5}, {2, 6}, {3, 7}, {4, 8}, {0, 8}, {1, 9}, {0, 7}, {1, 8}, {2, 9}, {1, 2}, {2, 3}, {4, 5}, {5, 6},
    {6, 7}, {7, 8}, {8, 9}, {0, 3}, {1, 4}, {2, 5}, {3, 6}, {4, 7}, {5, 8}, {0, 2}, {3, 5}, {4, 6}, {5, 7},
    {6, 8}, {7, 9}, {0, 5}, {1, 6}, {2, 7}, {3, 8}, {4, 9}, {3, 9}, {0, 6}, {1, 7}, {3, 9}};

void orc_get_code(int num, int spc, int prn, int *out)
{
  int g1[10], g2[10];
  for (int i = 0; i < 10; i++) g1[i] = g2[i] = 1;
  int k = 0;
  if (spc < 1) spc = 1;
  for (int i = 0; i < num; i++) {
    int val = (g1[9] + g2[k_sats[prn - 1][0]] + g2[k_sats[prn - 1][1]]) % 2;
    int chip = val ? 1 : -1;
    int f1 = (g1[2] + g1[9]) % 2;
    int f2 = (g2[1] + g2[2] + g2[5] + g2[7] + g2[8] + g2[9]) % 2;
    for (int j = 9; j > 0; j--) {
      g1[j] = g1[j - 1];
      g2[j] = g2[j - 1];
    }
    g1[0] = f1;
    g2[0] = f2;
    for (int j = 0; j < spc; j++) out[k++] = chip;
  }
}

/* mix_prn (gnss_codes.c:1032-1063): replica = code * e^{j theta n}, double recursive phasor,
 * renormalised every sample, FS = 1.023e6 * spc. */
void orc_mix_prn(const int32_t *code, double doppler_hz, double phase_deg, orc_c32 *sig, int size, int spc)
{
  /* PI in gnss_codes.c is the FLOAT literal 3.14159265358979f of Include/gnss_codes.h:33 (promoted to double
   * 3.1415927410125732 in the expressions below), not M_PI: the replica's frequency is 2.8e-8 high */
  const double REF_PI = (double)3.14159265358979f;
  double FS = 1.023e6 * spc;
  double dth = 2.0 * REF_PI * doppler_hz / FS;                                   /* :1037 */
  double ci = cos(dth), si = sin(dth);
  double pc = cos(phase_deg * REF_PI / 180.0), ps = sin(phase_deg * REF_PI / 180.0);   /* :1041-1042 */
  for (int n = 0; n < size; n++) {
    double a = (double)code[n];
    sig[n].r = (float)(a * pc);
    sig[n].i = (float)(a * ps);
    double npc = pc * ci - ps * si, nps = pc * si + ps * ci;
    pc = npc;
    ps = nps;
    double na = 1.0 / sqrt(pc * pc + ps * ps);
    pc *= na;
    ps *= na;
  }
}

/* make_replica (quasi_pilot_tests.c:153-181): the same mixer on a caller-given sample rate, initial phase 0.0001
 * degrees, and -- unlike mix_prn -- the DOUBLE M_PI (quasi_pilot_tests.c:10-12); `quant` is a constant false there,
 * so quantize_pm13 / noise never run.  Pinned bit for bit by the reference's own object (oracle/_ref/libquasi_ref.so,
 * tests/test_oracle_gnss.py) and through tests/golden/gnss_replica.npz. */
void orc_make_replica(const int32_t *code, orc_c32 *out, float doppler, int size, float samp_freq)
{
  const double phase_deg = 0.0001;
  const double dth = 2.0 * ORC_PI * (double)doppler / (double)samp_freq;        /* :157 */
  const double ci = cos(dth), si = sin(dth);
  double pc = cos(phase_deg * (double)ORC_PI / 180.0f), ps = sin(phase_deg * (double)ORC_PI / 180.0f);   /* :160-161 */
  for (int n = 0; n < size; n++) {
    double a = (double)code[n];
    out[n].r = (float)(a * pc);
    out[n].i = (float)(a * ps);
    double npc = pc * ci - ps * si, nps = pc * si + ps * ci;
    pc = npc;
    ps = nps;
    double na = 1.0 / sqrt(pc * pc + ps * ps);
    pc *= na;
    ps *= na;
  }
}

/* rotate_fwd (quasi_pilot_tests.c:138-151); 0 <= offset <= size (anything else indexes outside the array there) */
int orc_rotate_fwd(int *array, int size, int offset)
{
  if (size <= 0 || offset < 0 || offset > size) return -1;
  int *tmp = (int *)malloc(sizeof(int) * (size_t)size);
  if (!tmp) return -1;
  for (int i = 0; i < size; i++) tmp[i] = (i - offset < 0) ? array[size + (i - offset)] : array[i - offset];
  memcpy(array, tmp, sizeof(int) * (size_t)size);
  free(tmp);
  return 0;
}

/* recover_prn_phase_deg_with_doppler (CMSIS/CMSIS.cpp:362-434), statement by statement: float recursive phasor
 * without renormalisation, float wipe products, double correlation sums in index order, the window and its wrap
 * flags as floats, strict ">" for the running best.  PI there is the FLOAT literal (Include/gnss_codes.h:33),
 * R2D = (180.0f / 3.14159265358979f) (CMSIS.cpp:26).  SIZE <= 4092 (static float iw[1023 * 4]).
 * Parity unpinned by the reference: the harness is Windows-only C++ and holds no vector. */
double orc_recover_prn_phase(const orc_c32 *iandq, int SIZE, const int *prn_a, int code_offset, double doppler_a_hz,
                             double sample_rate_hz, int *best_offset_out, float *doppler_out, float *pwr)
{
  const double REF_PI = (double)3.14159265358979f;
  const float R2D = (180.0f / 3.14159265358979f);
  double best_pow = -1.0, best_re = 0.0, best_im = 0.0, best_doppler = 0.0;
  int best_off = 0;
  if (SIZE <= 0 || SIZE > 1023 * 4) return 0.0 / 0.0;
  float code_search_span = 10;
  float code_low = code_offset - code_search_span;
  float code_high = code_offset + code_search_span;
  unsigned char wrap = 0;
  if (code_low < 0) { code_low += SIZE; wrap = 1; }
  if (code_high > SIZE) { code_high -= SIZE; wrap = 1; }
  float f_search_step_hz = 1.0;
  float f_search_span_hz = 1;
  float *iw = (float *)malloc(sizeof(float) * (size_t)SIZE), *qw = (float *)malloc(sizeof(float) * (size_t)SIZE);
  for (float f = -f_search_span_hz; f <= +f_search_span_hz + 1e-3f; f += f_search_step_hz) {
    double dtheta = 2.0 * REF_PI * (doppler_a_hz + f) / (double)sample_rate_hz;
    const float c_inc = (float)cos(dtheta);
    const float s_inc = (float)sin(dtheta);
    float pc = 1.0f, ps = 0.0f;
    for (int n = 0; n < SIZE; ++n) {
      float I = iandq[n].r, Q = iandq[n].i;
      iw[n] = +I * pc + Q * ps;
      qw[n] = -I * ps + Q * pc;
      float npc = pc * c_inc - ps * s_inc;
      float nps = pc * s_inc + ps * c_inc;
      pc = npc;
      ps = nps;
    }
    for (int off = 0; off < SIZE; ++off) {
      if (wrap == 0 && (off < code_low || off > code_high)) continue;
      if (wrap == 1 && (off > code_high && off < code_low)) continue;
      double re = 0.0, im = 0.0;
      for (int n = 0; n < SIZE; ++n) {
        int idx = n - off;
        if (idx < 0) idx += SIZE;
        re += iw[n] * prn_a[idx];
        im += qw[n] * prn_a[idx];
      }
      double p = sqrt(re * re + im * im);
      if (p > best_pow) {
        best_pow = p;
        best_doppler = (double)f;
        best_re = re;
        best_im = im;
        best_off = off;
      }
    }
  }
  free(iw);
  free(qw);
  *doppler_out = (float)best_doppler;
  *best_offset_out = best_off;
  *pwr = (float)best_pow;
  return atan2(best_im, best_re) * R2D;
}

/* synth_gps_prn (gnss_codes.c:1066-1071) */
void orc_synth_gps_prn(int prn, float doppler, int size, orc_c32 *replica, int spc)
{
  int *code = (int *)malloc(sizeof(int) * (size_t)(size + spc));
  orc_get_code(size / spc, spc, prn, code);
  orc_mix_prn(code, doppler, 0, replica, size, spc);
  free(code);
}

static inline float c32_mag(orc_c32 v) { return sqrtf(v.r * v.r + v.i * v.i); } /* gnss_codes.c:27-29 */

/* find_top2_peaks_cplx / _real (gnss_codes.c:1144-1190): sequential, ">=" keeps the LAST maximal
 * index, second peak must be > pk_sep away from the best AT THE TIME it is seen. */
static void top2_step(float magn, int m, int pk_sep, float *v1, int *i1, float *v2, int *i2)
{
  if (magn >= *v1) {
    if (*i1 != -1 && abs(m - *i1) > pk_sep) { *v2 = *v1; *i2 = *i1; }
    *v1 = magn;
    *i1 = m;
  } else if (magn > *v2) {
    if (magn != *v1 && abs(m - *i1) > pk_sep) { *v2 = magn; *i2 = m; }
  }
}

int orc_find_top2_peaks_cplx(const orc_c32 *data, int n, int pk_sep, orc_top2 *pk)
{
  if (n <= 0 || !data || !pk) return -1;
  float v1 = -1e10f, v2 = -1e10f;
  int i1 = -1, i2 = -1;
  for (int m = 0; m < n; m++) top2_step(c32_mag(data[m]), m, pk_sep, &v1, &i1, &v2, &i2);
  pk->val1 = v1; pk->idx1 = i1; pk->val2 = v2; pk->idx2 = i2;
  return 0;
}

int orc_find_top2_peaks_real(const float *data, int n, int pk_sep, orc_top2 *pk)
{
  if (n <= 0 || !data || !pk) return -1;
  float v1 = -1e10f, v2 = -1e10f;
  int i1 = -1, i2 = -1;
  for (int m = 0; m < n; m++) top2_step(data[m], m, pk_sep, &v1, &i1, &v2, &i2);
  pk->val1 = v1; pk->idx1 = i1; pk->val2 = v2; pk->idx2 = i2;
  return 0;
}

/* compute_snr_cplx (gnss_codes.c:1110-1125): 10 log10(peak / mean off-peak magnitude) */
double orc_compute_snr_cplx(const orc_c32 *conv, int n, double peak, int peak_loc)
{
  double sum = 0;
  for (int i = 0; i < n; i++) {
    float magn = c32_mag(conv[i]);
    if (abs(i - peak_loc) > 2) sum += magn;
  }
  double mean = sum / (n - 5);
  return 10.0 * log10(peak / mean);
}

/* compute_snr_real (gnss_codes.c:1127-1142): the same on real magnitudes */
double orc_compute_snr_real(const float *conv, int n, double peak, int peak_loc)
{
  double sum = 0;
  for (int i = 0; i < n; i++) {
    float magn = conv[i];
    if (abs(i - peak_loc) > 2) sum += magn;
  }
  double mean = sum / (n - 5);
  return 10.0 * log10(peak / mean);
}

/* InterpolateCodePhase (gnss_codes.c:1096-1108) */
double orc_interpolate_code_phase(uint32_t index, double early, double prompt, double late)
{
  double x;
  if (late > early) {
    x = 0.5 * (late - early) / (prompt - early);
    return index + (x / 3.0) * (2.0 * x + 2.0);
  }
  x = 0.5 * (late - early) / (prompt - late);
  return index + (x / 3.0) * (-2.0 * x + 2.0);
}

/* fft_c32 (CMSIS/CMSIS.cpp:218-234): the radix-2 FFT on interleaved c32 */
void orc_fft_c32(int size, orc_c32 *w, int fwd) { orc_cfft_radix2_f32((float *)w, (uint32_t)size, fwd ? 0 : 1, 1); }

/* One (prn, doppler) cell row exactly as read_ors does it (CMSIS/CMSIS.cpp:706-737): carrier at
 * -doppler on the REPLICA, FFT(replica), FFT(signal), S*conj(R), IFFT, top-2 peak search. */
int orc_acquire_cell_reference(const orc_c32 *signal, int size, int prn, float doppler, int spc, orc_top2 *pk,
                               float *mag_out)
{
  orc_c32 *repli = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)size);
  orc_c32 *signl = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)size);
  orc_c32 *prod = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)size);
  if (!repli || !signl || !prod) return -1;
  /* read_ors passes -prn2acq.doppler where .doppler = -(true Doppler); net: replica carrier = +f_d */
  orc_synth_gps_prn(prn, doppler, size, repli, spc);
  memcpy(signl, signal, sizeof(orc_c32) * (size_t)size);
  orc_fft_c32(size, repli, 1);
  orc_fft_c32(size, signl, 1);
  for (int i = 0; i < size; i++) { /* mult(signl, get_conj(repli)) */
    float ar = signl[i].r, ai = signl[i].i, br = repli[i].r, bi = -repli[i].i;
    prod[i].r = ar * br - ai * bi;
    prod[i].i = ar * bi + ai * br;
  }
  orc_fft_c32(size, prod, 0);
  orc_find_top2_peaks_cplx(prod, size, 3, pk);
  if (mag_out)
    for (int i = 0; i < size; i++) mag_out[i] = c32_mag(prod[i]);
  free(repli);
  free(signl);
  free(prod);
  return 0;
}

/* The fused form the GPU kernel implements:
 *   per Doppler bin: wipe-off on the SIGNAL, x[n] e^{-j 2 pi f n / FS} (the form of
 *     recover_prn_phase_deg_with_doppler, CMSIS/CMSIS.cpp:384-404, phase continuous over all n_ms),
 *     coherent fold of the n_ms one-ms blocks (linearity of CMSIS.cpp:1157-1169's frequency-domain
 *     sum), zero-pad samples_per_ms -> fft_len (CMSIS.cpp:1154-1160), ONE forward FFT;
 *   per PRN: multiply by conj(FFT(code, zero Doppler, zero-padded)), inverse FFT (1/N), |.|,
 *     find_top2_peaks (pk_sep 3), compute_snr, InterpolateCodePhase; best Doppler bin per PRN =
 *     largest val1 (lowest bin wins ties).
 * Phases are evaluated directly in double (no recursive phasor): the gate downstream is identical
 * PRN / code phase / Doppler bin, not bit-equal correlation values. */
/* core: `codes` = n_prn spectra of fft_len points (already transformed), FS = sample rate */
static int acquire_fused_core(const orc_c32 *signal, int n_ms, int spm, int fft_len, double FS, const orc_c32 *codes, const int *prns,
                              int n_prn, const float *dopplers, int n_dop, orc_acq_result *best, orc_acq_result *all)
{
  orc_c32 *z = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)fft_len);
  orc_c32 *prod = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)fft_len);
  if (!z || !prod) return -1;
  for (int p = 0; p < n_prn; p++) best[p].peak = -1.0f;
  for (int d = 0; d < n_dop; d++) {
    memset(z, 0, sizeof(orc_c32) * (size_t)fft_len);
    for (int b = 0; b < n_ms; b++) {
      for (int m = 0; m < spm; m++) {
        long n = (long)b * spm + m;
        double cyc = (double)dopplers[d] * (double)n / FS;
        cyc -= floor(cyc);
        double th = 2.0 * ORC_PI * cyc;
        float c = (float)cos(th), s = (float)sin(th);
        float I = signal[n].r, Q = signal[n].i;
        z[m].r += I * c + Q * s;
        z[m].i += -I * s + Q * c;
      }
    }
    orc_fft_c32(fft_len, z, 1);
    for (int p = 0; p < n_prn; p++) {
      const orc_c32 *c = codes + (size_t)p * fft_len;
      for (int i = 0; i < fft_len; i++) {
        float ar = z[i].r, ai = z[i].i, br = c[i].r, bi = -c[i].i;
        prod[i].r = ar * br - ai * bi;
        prod[i].i = ar * bi + ai * br;
      }
      orc_fft_c32(fft_len, prod, 0);
      orc_top2 pk;
      orc_find_top2_peaks_cplx(prod, fft_len, 3, &pk);
      orc_acq_result r;
      r.peak = pk.val1; r.second = pk.val2; r.code_phase = pk.idx1; r.second_phase = pk.idx2;
      r.doppler_bin = d; r.prn = prns[p];
      r.snr_db = (float)orc_compute_snr_cplx(prod, fft_len, pk.val1, pk.idx1);
      int e = (pk.idx1 + fft_len - 1) % fft_len, l = (pk.idx1 + 1) % fft_len;
      r.code_phase_interp = (float)orc_interpolate_code_phase((uint32_t)pk.idx1, c32_mag(prod[e]), pk.val1, c32_mag(prod[l]));
      if (all) all[(size_t)p * n_dop + d] = r;
      if (r.peak > best[p].peak) best[p] = r;
    }
  }
  free(z);
  free(prod);
  return 0;
}

int orc_acquire_fused(const orc_c32 *signal, int n_ms, int spm, int fft_len, int spc, const int *prns, int n_prn,
                      const float *dopplers, int n_dop, orc_acq_result *best, orc_acq_result *all)
{
  if (spm > fft_len || n_ms < 1) return -1;
  orc_c32 *codes = (orc_c32 *)calloc((size_t)n_prn * fft_len, sizeof(orc_c32));
  int *chips = (int *)malloc(sizeof(int) * (size_t)(fft_len + spc));
  if (!codes || !chips) return -1;
  for (int p = 0; p < n_prn; p++) {
    orc_c32 *c = codes + (size_t)p * fft_len;
    orc_get_code(spm / spc, spc, prns[p], chips);
    for (int n = 0; n < spm; n++) { c[n].r = (float)chips[n]; c[n].i = 0.0f; }
    orc_fft_c32(fft_len, c, 1);
  }
  int rc = acquire_fused_core(signal, n_ms, spm, fft_len, 1.023e6 * spc, codes, prns, n_prn, dopplers, n_dop, best, all);
  free(codes);
  free(chips);
  return rc;
}

/* The same with caller-supplied time-domain replicas (n_prn x fft_len complex, e.g. an L5 code
 * resampled 10230 -> 16384 the way CMSIS/CMSIS.cpp:906 prepares it, at zero Doppler); the sample
 * rate is samples_per_ms kHz. */
int orc_acquire_fused_replicas(const orc_c32 *signal, int n_ms, int spm, int fft_len, const orc_c32 *replicas, const int *prns, int n_prn,
                               const float *dopplers, int n_dop, orc_acq_result *best, orc_acq_result *all)
{
  if (spm > fft_len || n_ms < 1) return -1;
  orc_c32 *codes = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)n_prn * fft_len);
  if (!codes) return -1;
  memcpy(codes, replicas, sizeof(orc_c32) * (size_t)n_prn * fft_len);
  for (int p = 0; p < n_prn; p++) orc_fft_c32(fft_len, codes + (size_t)p * fft_len, 1);
  int rc = acquire_fused_core(signal, n_ms, spm, fft_len, 1000.0 * spm, codes, prns, n_prn, dopplers, n_dop, best, all);
  free(codes);
  return rc;
}

/* l5_generate_I (gnss_codes.c:153-204) from the epoch: two 13-stage LFSRs, all ones (:147-151),
 * right shift with the feedback injected at stage 13 (lfsr_step :101-106; g1_feedback :90-94 taps
 * 13,12,10,9,1; g2_feedback :95-99 taps 13,12,8,7,6,4,3,1), chip = G1 stage 13 XOR two PRN-specific
 * G2 stages (L5I_TAPS :115-124, stage_bit :130-132), 0 -> +1, 1 -> -1 (:180-181).  The registers
 * are NOT reset at the 10230-chip boundary (:194-197 only wraps the counter).  Quirk kept: a tap
 * number 0 makes stage_bit shift by (0 - 1u) = 2^32 - 1, undefined in C; on the x86/ARM hosts the
 * fork runs on the 32-bit shift count is taken modulo 32, i.e. >> 31, which reads 0 from the
 * zero-extended 16-bit state -- restated here as "tap 0 contributes 0".  Returns 0, -1 for an
 * unsupported PRN (:155). */
static const unsigned char k_l5i_taps[38][2] = {
    {0, 0},
    {1, 5}, {2, 6}, {3, 7}, {4, 8}, {0, 8}, {1, 9}, {0, 7}, {1, 8}, {2, 9}, {1, 2},
    {2, 3}, {4, 5}, {5, 6}, {6, 7}, {7, 8}, {8, 9}, {0, 3}, {1, 4}, {2, 5}, {3, 6},
    {4, 7}, {5, 8}, {0, 2}, {3, 5}, {4, 6}, {5, 7}, {6, 8}, {7, 9}, {0, 5}, {1, 6},
    {2, 7}, {3, 8}, {4, 9}, {3, 9}, {0, 6}, {1, 7}, {3, 9},
};
int orc_l5_generate_I(int prn, signed char *out, int n)
{
  if (prn <= 0 || prn > 37) return -1;
  unsigned g1 = 0x1FFFu, g2 = 0x1FFFu;
  const unsigned t1 = k_l5i_taps[prn][0], t2 = k_l5i_taps[prn][1];
  for (int i = 0; i < n; i++) {
    unsigned g1_out = (g1 >> 12) & 1u;
    unsigned a = t1 ? (g2 >> (t1 - 1u)) & 1u : 0u, b = t2 ? (g2 >> (t2 - 1u)) & 1u : 0u;
    unsigned chip = g1_out ^ a ^ b;
    out[i] = chip ? -1 : +1;
    unsigned f1 = ((g1 >> 12) ^ (g1 >> 11) ^ (g1 >> 9) ^ (g1 >> 8) ^ g1) & 1u;
    unsigned f2 = ((g2 >> 12) ^ (g2 >> 11) ^ (g2 >> 7) ^ (g2 >> 6) ^ (g2 >> 5) ^ (g2 >> 3) ^ (g2 >> 2) ^ g2) & 1u;
    g1 = ((g1 >> 1) | (f1 << 12)) & 0x1FFFu;
    g2 = ((g2 >> 1) | (f2 << 12)) & 0x1FFFu;
  }
  return 0;
}

/* compute_snr_real (gnss_codes.c:1127-1142) */
static double snr_real(const float *row, int n, double peak, int peak_loc)
{
  double sum = 0;
  for (int i = 0; i < n; i++)
    if (abs(i - peak_loc) > 2) sum += row[i];
  return 10.0 * log10(peak / (sum / (n - 5)));
}

/* Accumulation over groups of milliseconds: mode 1 = non-coherent sum of magnitudes
 * (NCI loop, CMSIS/CMSIS.cpp:915-943: nci_sum[i] += mag(up_prod[i])), mode 2 = differential
 * (test_quasi_diff_pilot, :1047-1062: diff_acc += y * conj(prev), then mag).  Groups are folded
 * coherently with the wipe-off of orc_acquire_fused; the row goes through find_top2_peaks_real,
 * compute_snr_real and InterpolateCodePhase (:948-951; neighbours taken circularly, the reference
 * indexes idx1 +- 1 without a wrap). */
int orc_acquire_groups(const orc_c32 *signal, int n_ms, int spm, int fft_len, int spc, const int *prns, int n_prn, const float *dopplers,
                       int n_dop, int mode, int n_groups, orc_acq_result *best, orc_acq_result *all)
{
  if (spm > fft_len || n_ms < 1 || n_groups < 1 || n_ms % n_groups || (mode != 1 && mode != 2) || (mode == 2 && n_groups < 2)) return -1;
  const int mpg = n_ms / n_groups;
  double FS = 1.023e6 * spc;
  orc_c32 *codes = (orc_c32 *)calloc((size_t)n_prn * fft_len, sizeof(orc_c32));
  orc_c32 *Z = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)fft_len * n_groups);
  orc_c32 *y = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)fft_len);
  orc_c32 *prev = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)fft_len);
  orc_c32 *dacc = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)fft_len);
  float *row = (float *)malloc(sizeof(float) * (size_t)fft_len);
  int *chips = (int *)malloc(sizeof(int) * (size_t)(fft_len + spc));
  if (!codes || !Z || !y || !prev || !dacc || !row || !chips) return -1;
  for (int p = 0; p < n_prn; p++) {
    orc_c32 *c = codes + (size_t)p * fft_len;
    orc_get_code(spm / spc, spc, prns[p], chips);
    for (int n = 0; n < spm; n++) { c[n].r = (float)chips[n]; c[n].i = 0.0f; }
    orc_fft_c32(fft_len, c, 1);
    best[p].peak = -1.0f;
  }
  for (int d = 0; d < n_dop; d++) {
    for (int g = 0; g < n_groups; g++) {
      orc_c32 *z = Z + (size_t)g * fft_len;
      memset(z, 0, sizeof(orc_c32) * (size_t)fft_len);
      for (int b = g * mpg; b < (g + 1) * mpg; b++)
        for (int m = 0; m < spm; m++) {
          long n = (long)b * spm + m;
          double cyc = (double)dopplers[d] * (double)n / FS;
          cyc -= floor(cyc);
          double th = 2.0 * ORC_PI * cyc;
          float c = (float)cos(th), s = (float)sin(th);
          float I = signal[n].r, Q = signal[n].i;
          z[m].r += I * c + Q * s;
          z[m].i += -I * s + Q * c;
        }
      orc_fft_c32(fft_len, z, 1);
    }
    for (int p = 0; p < n_prn; p++) {
      const orc_c32 *c = codes + (size_t)p * fft_len;
      memset(row, 0, sizeof(float) * (size_t)fft_len);
      memset(dacc, 0, sizeof(orc_c32) * (size_t)fft_len);
      for (int g = 0; g < n_groups; g++) {
        const orc_c32 *z = Z + (size_t)g * fft_len;
        for (int i = 0; i < fft_len; i++) {
          float ar = z[i].r, ai = z[i].i, br = c[i].r, bi = -c[i].i;
          y[i].r = ar * br - ai * bi;
          y[i].i = ar * bi + ai * br;
        }
        orc_fft_c32(fft_len, y, 0);
        if (mode == 1) {
          for (int i = 0; i < fft_len; i++) row[i] += c32_mag(y[i]);
        } else {
          if (g > 0)
            for (int i = 0; i < fft_len; i++) {
              float br = prev[i].r, bi = -prev[i].i;
              dacc[i].r += y[i].r * br - y[i].i * bi;
              dacc[i].i += y[i].r * bi + y[i].i * br;
            }
          memcpy(prev, y, sizeof(orc_c32) * (size_t)fft_len);
        }
      }
      if (mode == 2)
        for (int i = 0; i < fft_len; i++) row[i] = c32_mag(dacc[i]);
      orc_top2 pk;
      orc_find_top2_peaks_real(row, fft_len, 3, &pk);
      orc_acq_result r;
      r.peak = pk.val1; r.second = pk.val2; r.code_phase = pk.idx1; r.second_phase = pk.idx2;
      r.doppler_bin = d; r.prn = prns[p];
      r.snr_db = (float)snr_real(row, fft_len, pk.val1, pk.idx1);
      int e = (pk.idx1 + fft_len - 1) % fft_len, l = (pk.idx1 + 1) % fft_len;
      r.code_phase_interp = (float)orc_interpolate_code_phase((uint32_t)pk.idx1, row[e], pk.val1, row[l]);
      if (all) all[(size_t)p * n_dop + d] = r;
      if (r.peak > best[p].peak) best[p] = r;
    }
  }
  free(codes); free(Z); free(y); free(prev); free(dacc); free(row); free(chips);
  return 0;
}

/* DecodeOrsIQCplx (CMSIS/CMSIS.cpp:465-485): 2-bit I/Q, two complex samples per byte */
void orc_decode_ors_iq(const uint8_t *data, uint32_t byte_length, orc_c32 *iqs)
{
  static const int16_t tbl[4] = {1, -1, 3, -3};
  for (uint32_t i = 0; i < byte_length; i++) {
    uint8_t byte = data[i];
    iqs[2 * i].r = (float)tbl[byte & 0x03];
    iqs[2 * i].i = (float)tbl[(byte >> 2) & 0x03];
    iqs[2 * i + 1].r = (float)tbl[(byte >> 4) & 0x03];
    iqs[2 * i + 1].i = (float)tbl[(byte >> 6) & 0x03];
  }
}

/* ---- resampler: iq_resamp_zp_* (up_sample.c:16-157) behind up_sample_N_to_M / up_sample_10k_to_16k
 * (gnss_codes.c:868-904) ---- */
static double sinc_pi(double x) { return fabs(x) < 1e-12 ? 1.0 : sin(ORC_PI * x) / (ORC_PI * x); } /* up_sample.c:16-20 */
static double blackman_sym(int n, int half)                                                        /* :23-29 */
{
  int N = 2 * half, k = n + half;
  double phi = (2.0 * ORC_PI * k) / (double)(N - 1);
  return 0.42 - 0.5 * cos(phi) + 0.08 * cos(2.0 * phi);
}
static int reflect_index(int idx, int N) /* :51-59 */
{
  if (N <= 1) return 0;
  while (idx < 0 || idx >= N) {
    if (idx < 0) idx = -idx - 1;
    if (idx >= N) idx = 2 * N - 1 - idx;
  }
  return idx;
}
static void resamp_forward(double dt, int half, double c, const orc_c32 *in, int nin, orc_c32 *out, int nout) /* :65-104 */
{
  for (int k = 0; k < nout; k++) {
    double t = (double)k * dt;
    int ti = (int)floor(t);
    double frac = t - (double)ti, ai = 0, aq = 0, ws = 0;
    for (int m = -half; m < half; m++) {
      double x = (double)m - frac;
      double h = (2.0 * c) * sinc_pi(2.0 * c * x);
      h *= blackman_sym(m, half);
      int idx = reflect_index(ti + m, nin);
      ai += h * (double)in[idx].r;
      aq += h * (double)in[idx].i;
      ws += h;
    }
    double inv = fabs(ws) > 1e-15 ? 1.0 / ws : 1.0;
    out[k].r = (float)(ai * inv);
    out[k].i = (float)(aq * inv);
  }
}

/* iq_resamp_zp_init + iq_resamp_process_zero_phase (up_sample.c:31-44, :115-157): forward
 * resample fin -> fout, reverse, identity-rate pass of the same kernel, reverse.  Returns the
 * number of outputs written = min(llround(n_in * fout / fin), cap); 0 on bad arguments. */
long orc_iq_resamp_zero_phase(double fin, double fout, int ntaps, double cutoff, const orc_c32 *in, long n_in, orc_c32 *out,
                              long cap)
{
  if (ntaps < 4 || (ntaps & 1)) return 0;
  if (!(cutoff > 0.0 && cutoff <= 0.5)) return 0;
  if (!in || !out || n_in <= 0) return 0;
  const int half = ntaps / 2;
  long n_out = llround((double)n_in * (fout / fin));
  if (n_out > cap) n_out = cap;
  if (n_out <= 0) return 0;
  orc_c32 *y = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)n_out);
  orc_c32 *yr = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)n_out);
  orc_c32 *yr2 = (orc_c32 *)malloc(sizeof(orc_c32) * (size_t)n_out);
  if (!y || !yr || !yr2) return 0;
  resamp_forward(fin / fout, half, cutoff, in, (int)n_in, y, (int)n_out);
  for (long k = 0; k < n_out; k++) yr[k] = y[n_out - 1 - k];
  resamp_forward(1.0, half, cutoff, yr, (int)n_out, yr2, (int)n_out);
  for (long k = 0; k < n_out; k++) out[k] = yr2[n_out - 1 - k];
  free(y);
  free(yr);
  free(yr2);
  return n_out;
}

/* up_sample_N_to_M (gnss_codes.c:868-884): 12 taps, cutoff 0.45, rates N_in and M_out; the cap is
 * llround(N_in * (M_out / N_in)) with the INTEGER quotient (:879), so 10230 -> 16384 yields only
 * 10230 outputs there.  Returns the number written. */
long orc_up_sample_n_to_m(const orc_c32 *in, int n_in, orc_c32 *out, int m_out)
{
  if (n_in <= 0) return 0;
  long cap = llround((double)((long)n_in * (long)(m_out / n_in)));
  return orc_iq_resamp_zero_phase((double)n_in, (double)m_out, 12, 0.45, in, n_in, out, cap);
}

/* up_sample_10k_to_16k (gnss_codes.c:886-904): 10230 samples at 10.23 MHz -> 16384 at 16.384 MHz */
long orc_up_sample_10k_to_16k(const orc_c32 *in, orc_c32 *out)
{
  return orc_iq_resamp_zero_phase(10230e3, 16384e3, 12, 0.45, in, 10230, out, llround(10230 * (16384e3 / 10230e3)) + 8);
}

/* ---- Galileo E1-B CBOC replica (parity unpinned: the reference holds no vector
 * for it and its code tables HEX_E1B.txt are outside the mount) ---- */
/* hex_nibble + hexstream_to_e1b_chips (gnss_codes.c:234-239, :276-289; the E5a / E5-QP twins :299-325
 * differ only in the length): hex digits -> chips, MSB first within a nibble, logic 1 -> -1, logic 0 -> +1
 * (logic_to_chip_icd :243-245); any other character is skipped.  0 on success, -1 if the stream is short. */
int orc_hexstream_to_chips(const char *hex_stream, signed char *chips, int code_len)
{
  int chip_idx = 0;
  for (const char *p = hex_stream; *p && chip_idx < code_len; ++p) {
    const int c = (unsigned char)*p;
    int v;
    if (c >= '0' && c <= '9') v = c - '0';
    else if (c >= 'A' && c <= 'F') v = 10 + (c - 'A');
    else if (c >= 'a' && c <= 'f') v = 10 + (c - 'a');
    else continue;
    for (int b = 3; b >= 0 && chip_idx < code_len; --b) chips[chip_idx++] = ((v >> b) & 1) ? (signed char)-1 : (signed char)+1;
  }
  return chip_idx == code_len ? 0 : -1;
}

/* boc_sign (gnss_codes.c:247-251) and cboc_e1b_weight (:253-260): sqrt(10/11) BOC(1,1) minus sqrt(1/11) BOC(6,1) */
static int boc_sign_f(int m, float u)
{
  if (m <= 0) return +1;
  int k = (int)floorf(2.0f * m * u);
  return (k & 1) ? -1 : +1;
}
float orc_cboc_e1b_weight(float u)
{
  const float w1 = 0.9534625892455923f, w6 = 0.3015113445777636f;
  return w1 * (float)boc_sign_f(1, u) - w6 * (float)boc_sign_f(6, u);
}

/* synth_e1b_prn (gnss_codes.c:920-965) with the code table handed in (`ca` = the 4092 chips of one PRN,
 * E1B_Code[prn] in the reference): fs 4.092 MHz, chip rate 1.023 MHz, code phase and carrier phase start at 0,
 * FLOAT recursions for both, carrier through the double sin/cos of sincosf_fast (:573-576). */
void orc_synth_e1b_prn(const signed char *ca, float doppler, size_t N, orc_c32 *out)
{
  const float PI_F = 3.14159265358979f; /* Include/gnss_codes.h:33 */
  const float fs_hz = 4.092e6f, code_rate_cps = 1.023e6f;
  const int L = 4092;
  float dchips = code_rate_cps / fs_hz;
  float dphia = 2.0f * (float)PI_F * (doppler) / fs_hz;
  float chips = 0.0f, phia = 0.0f;
  for (size_t n = 0; n < N; ++n) {
    float frac = chips - floorf(chips);
    int idx = (int)floorf(chips); /* code_chip_at :265-270 */
    if (idx >= L) idx -= L;
    else if (idx < 0) idx += L;
    float ampa = (float)ca[idx] * orc_cboc_e1b_weight(frac);
    float sa = (float)sin(phia), caph = (float)cos(phia);
    out[n].r = ampa * caph;
    out[n].i = ampa * sa;
    chips += dchips;
    if (chips >= L) chips -= L;
    else if (chips < 0) chips += L;
    phia += dphia;
    if (phia > 1e9f || phia < -1e9f) phia = fmodf(phia, 2.0f * PI_F);
  }
}
