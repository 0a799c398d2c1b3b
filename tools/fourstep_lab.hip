// fourstep_lab: phase time stamps of the four-step tile kernels (library sources built with -DARMFFT_LAB_STAMPS).
// tools only; build: see tools/build_fourstep_lab.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "arm_fft_mi355.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
namespace armfft { extern __device__ unsigned long long *g_lab_stamps; }

int main(int argc, char **argv)
{
  const unsigned batch = argc > 1 ? atoi(argv[1]) : 8;
  const size_t n = 1u << 22;
  float *data, *work;
  CK(hipMalloc(&data, n * 8 * batch)); CK(hipMalloc(&work, n * 8 * batch + 64));
  CK(hipMemset(data, 0, n * 8 * batch));
  arm_cfft_instance_u32_f32 S;
  if (arm_cfft_init_u32_f32(&S, (uint32_t)n) != ARM_MATH_SUCCESS) { printf("init failed: %s\n", arm_fft_mi355_last_error()); return 1; }
  const unsigned nwg = 256 * batch;
  unsigned long long *stamps; CK(hipMalloc(&stamps, (size_t)nwg * 64));
  unsigned long long *null = nullptr;
  for (int i = 0; i < 10; i++) arm_cfft_u32_f32_dev(&S, data, work, batch, 0, nullptr);
  CK(hipDeviceSynchronize());
  // stamps of pass 2 overwrite those of pass 1 (same grid): run with "pass" selecting which one to keep is not possible
  // through the C ABI, so both passes write and the second wins; argv[2] = 1 keeps pass 1 by pointing the stamps away
  CK(hipMemcpyToSymbol(HIP_SYMBOL(armfft::g_lab_stamps), &stamps, sizeof stamps));
  arm_cfft_u32_f32_dev(&S, data, work, batch, 0, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipMemcpyToSymbol(HIP_SYMBOL(armfft::g_lab_stamps), &null, sizeof null));
  std::vector<unsigned long long> h((size_t)nwg * 8);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t00 = ~0ull;
  for (unsigned b = 0; b < nwg; b++) t00 = std::min(t00, h[b * 8]);
  // s_memtime ticks at 100 MHz: 10 ns
  printf("# wg xcc cu/simd-bits start issue loads_done fft_done stores_done (us from first start)\n");
  for (unsigned b = 0; b < nwg; b++) {
    const unsigned long long *s = &h[b * 8];
    const unsigned hw = (unsigned)s[5], xcc = (unsigned)(s[5] >> 32) & 15;
    if (b < 40 || (b % 97) == 0)
      printf("%5u x%u cu%2u se%u  %8.2f %8.2f %8.2f %8.2f %8.2f\n", b, xcc, (hw >> 8) & 15, (hw >> 13) & 7, (s[0] - t00) * 0.01, (s[1] - t00) * 0.01,
             (s[2] - t00) * 0.01, (s[3] - t00) * 0.01, (s[4] - t00) * 0.01);
  }
  double a[4] = {0, 0, 0, 0};
  for (unsigned b = 0; b < nwg; b++) { const unsigned long long *s = &h[b * 8]; a[0] += s[2] - s[0]; a[1] += s[3] - s[2]; a[2] += s[4] - s[3]; a[3] += s[4] - s[0]; }
  printf("mean per workgroup: load %.2f us, fft+store issue %.2f us, store drain %.2f us, total %.2f us\n", a[0] / nwg * 0.01, a[1] / nwg * 0.01, a[2] / nwg * 0.01, a[3] / nwg * 0.01);
  unsigned long long tend = 0; for (unsigned b = 0; b < nwg; b++) tend = std::max(tend, h[b * 8 + 4]);
  printf("kernel span %.2f us\n", (tend - t00) * 0.01);
  return 0;
}
