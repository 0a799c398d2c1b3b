// fft_lab2.hip -- A/B harness around the LIBRARY's own batched f32 kernels (included from fft_amd/csrc) plus
// experimental variants; all arms run interleaved in one process, medians over rounds (same-box A/B).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ifft_amd/csrc -Iinclude tools/fft_lab2.hip -o tools/fft_lab2
#include "../fft_amd/csrc/armfft_f32.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>
using namespace armfft;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
namespace armfft { void set_error(const char *, ...) {} bool hip_ok(hipError_t e, const char *) { return e == hipSuccess; } const float2 *dev_twiddle_f32(int) { return nullptr; } }

#include "fft_lab2_variants.hpp"

struct Arm { std::string name; std::function<void(float2 *)> launch; std::vector<float> ms; double err; };

int main(int argc, char **argv)
{
  const int logn = argc > 1 ? atoi(argv[1]) : 12;
  const uint32_t N = 1u << logn;
  const size_t bytes = 2ull << 30;
  const uint32_t batch = (uint32_t)(bytes / (N * sizeof(float2)));
  float2 *d, *ref, *tw, *src;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&ref, bytes)); CK(hipMalloc(&src, bytes)); CK(hipMalloc(&tw, N * sizeof(float2)));
  {
    std::vector<float2> h((size_t)batch * N);
    uint32_t s = 12345;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x.x = (int)(s >> 8) / 8388608.0f - 1.0f; s = s * 1664525u + 1013904223u; x.y = (int)(s >> 8) / 8388608.0f - 1.0f; }
    CK(hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice));
  }
  CK(launch_gen_twiddle_f32(tw, N, 0));
  if (argc > 2 && !strcmp(argv[2], "diag")) { CK(hipMemcpy(d, src, bytes, hipMemcpyDeviceToDevice)); run_diag(tw, batch, d); return 0; }
  std::vector<Arm> arms;
  arms.push_back({"library kernel", [&](float2 *p) { CK(launch_cfft_f32(logn, p, tw, batch, false, 0, 0)); }, {}, 0});
  add_variants(arms, logn, tw, batch);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 9, reps = 3;
  std::vector<float2> href(N * 4), hout(N * 4);
  for (int round = -1; round < rounds; round++) {
    for (size_t i = 0; i < arms.size(); i++) {
      if (round < 0) {
        CK(hipMemcpy(d, src, bytes, hipMemcpyDeviceToDevice));
        arms[i].launch(d);
        CK(hipDeviceSynchronize());
        if (i == 0) CK(hipMemcpy(ref, d, bytes, hipMemcpyDeviceToDevice));
        double num = 0, den = 0;
        for (int q = 0; q < 4; q++) {
          size_t off = (size_t)((q * 21845u + 3) % batch) * N;
          if (q == 3) off = (size_t)(batch - 1) * N;
          CK(hipMemcpy(href.data() + q * N, ref + off, N * sizeof(float2), hipMemcpyDeviceToHost));
          CK(hipMemcpy(hout.data() + q * N, d + off, N * sizeof(float2), hipMemcpyDeviceToHost));
        }
        for (size_t k = 0; k < href.size(); k++) {
          double dx = href[k].x - hout[k].x, dy = href[k].y - hout[k].y;
          num += dx * dx + dy * dy; den += (double)href[k].x * href[k].x + (double)href[k].y * href[k].y;
        }
        arms[i].err = sqrt(num / den);
        continue;
      }
      // timing runs on the evolving data of d scaled back each round by the copy
      CK(hipMemcpyAsync(d, src, bytes, hipMemcpyDeviceToDevice, 0));
      arms[i].launch(d);
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; r++) arms[i].launch(d);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      arms[i].ms.push_back(t / reps);
    }
  }
  for (auto &a : arms) {
    std::sort(a.ms.begin(), a.ms.end());
    printf("%-44s relL2 %.1e  median %.4f ms (min %.4f max %.4f) %5.0f GB/s frac %.3f\n", a.name.c_str(), a.err, a.ms[rounds / 2], a.ms[0], a.ms[rounds - 1],
           2.0 * bytes / a.ms[rounds / 2] / 1e6, 2.0 * bytes / a.ms[rounds / 2] / 1e6 / 8000.0);
  }
  return 0;
}
