/*
 * ref_shim_resamp.c -- entry point into the reference's OWN resampler (Source/TransformFunctions/
 * up_sample.c, compiled where it lies; see Makefile target `ref`).  TEST INFRASTRUCTURE ONLY.
 * up_sample.c is self-contained C (its only includes are up_sample.h and cplx_types.h).
 */
#include <stddef.h>

#include "up_sample.h"

long ref_iq_resamp_zero_phase(double fin, double fout, int ntaps, double cutoff, const float *in, long n_in, float *out, long cap)
{
  iq_resamp_zp_t *st = iq_resamp_zp_init(fin, fout, ntaps, cutoff);
  if (!st) return 0;
  size_t n = iq_resamp_process_zero_phase(st, (const c32 *)in, (size_t)n_in, (c32 *)out, (size_t)cap);
  iq_resamp_zp_free(st);
  return (long)n;
}
