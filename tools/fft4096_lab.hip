// fft4096_lab.hip -- A/B harness for variants of the batched 4096-point f32 kernel (MI355X).
// All variants run interleaved in one process (rule 24) on random data; outputs are compared to
// variant 0 (the library's kernel shape, parity-tested against the oracle by tests/).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ifft_amd/csrc tools/fft4096_lab.hip -o tools/fft4096_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "fft_core_f32.hpp"
using namespace armfft;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LOGN = 12, N = 4096;
using PL = Plan<LOGN>;

// V0: one workgroup per transform (library shape)
template <int MINW>
__global__ __launch_bounds__(256, MINW) void v0(float2 *__restrict__ data, const float2 *__restrict__ tw, uint32_t batch)
{
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const uint32_t b = blockIdx.x;
  auto ld = [&](int, int e) { return data[(size_t)b * N + e]; };
  auto st = [&](int, int e, float2 v) { data[(size_t)b * N + e] = v; };
  fft_block<PL, false, false>(lds, tw, threadIdx.x, ld, st);
}

// V1: persistent; twiddles of the passes in HMASK live in registers; everything derived from the
// thread index is made opaque per iteration so that the compiler does not hoist address math.
template <int MINW, unsigned HMASK>
__global__ __launch_bounds__(256, MINW) void v1(float2 *__restrict__ data, const float2 *__restrict__ tw, uint32_t batch)
{
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  TwRegs<PL, HMASK> twr;
  twr.tw = tw;
  twr.preload(threadIdx.x);
  for (uint32_t b = blockIdx.x; b < batch; b += gridDim.x) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    float2 *__restrict__ p = data + (size_t)b * N;
    auto ld = [&](int, int e) { return p[e]; };
    auto st = [&](int, int e, float2 v) { p[e] = v; };
    fft_block_tw<PL, false, false>(lds, twr, tid, ld, st);
    __syncthreads();
  }
}

// V2: V1 + register prefetch of the next transform
template <int MINW, unsigned HMASK>
__global__ __launch_bounds__(256, MINW) void v2(float2 *__restrict__ data, const float2 *__restrict__ tw, uint32_t batch)
{
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  TwRegs<PL, HMASK> twr;
  twr.tw = tw;
  twr.preload(threadIdx.x);
  uint32_t b = blockIdx.x;
  if (b >= batch) return;
  float2 nxt[16];
#pragma unroll
  for (int r = 0; r < 16; r++) nxt[r] = data[(size_t)b * N + threadIdx.x + 256 * r];
  while (true) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = nxt[r];
    const uint32_t nb = b + gridDim.x;
    if (nb < batch) {
      const float2 *__restrict__ pn = data + (size_t)nb * N;
#pragma unroll
      for (int r = 0; r < 16; r++) nxt[r] = pn[tid + 256 * r];
    }
    float2 *__restrict__ p = data + (size_t)b * N;
    auto st = [&](int, int e, float2 x) { p[e] = x; };
    fft_block_preloaded<PL, false>(v, lds, twr, tid, st);
    if (nb >= batch) break;
    b = nb;
    __syncthreads();
  }
}

struct Variant { const char *name; void (*k)(float2 *, const float2 *, uint32_t); int blocks_per_cu; };

int main(int argc, char **argv)
{
  const uint32_t batch = 65536;
  const size_t bytes = (size_t)batch * N * sizeof(float2);
  float2 *d, *ref, *tw, *src;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&ref, bytes)); CK(hipMalloc(&src, bytes)); CK(hipMalloc(&tw, N * sizeof(float2)));
  {
    std::vector<float2> h((size_t)batch * N);
    uint32_t s = 12345;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x.x = (int)(s >> 8) / 8388608.0f - 1.0f; s = s * 1664525u + 1013904223u; x.y = (int)(s >> 8) / 8388608.0f - 1.0f; }
    CK(hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice));
    std::vector<float2> t(N);
    for (int m = 0; m < N; m++) { double a = -2.0 * M_PI * m / N; t[m] = make_float2((float)cos(a), (float)sin(a)); }
    CK(hipMemcpy(tw, t.data(), N * sizeof(float2), hipMemcpyHostToDevice));
  }
  std::vector<Variant> vs = {
      {"v0 1wg/xform minw4", v0<4>, 0},
      {"v1 persist h000 minw4 x4", v1<4, 0u>, 4},
      {"v1 persist h100 minw4 x4", v1<4, 4u>, 4},
      {"v1 persist h100 minw3 x3", v1<3, 4u>, 3},
      {"v1 persist h110 minw4 x4", v1<4, 6u>, 4},
      {"v1 persist h110 minw3 x3", v1<3, 6u>, 3},
      {"v1 persist h110 minw2 x2", v1<2, 6u>, 2},
      {"v2 prefetch h100 minw3 x3", v2<3, 4u>, 3},
      {"v2 prefetch h110 minw3 x3", v2<3, 6u>, 3},
      {"v2 prefetch h110 minw2 x2", v2<2, 6u>, 2},
  };
  for (auto &v : vs) CK(hipFuncSetAttribute((const void *)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL::LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<std::vector<float>> ms(vs.size());
  const int rounds = 7;
  std::vector<float2> href(4096 * 4), hout(4096 * 4);
  for (int round = 0; round < rounds; round++) {
    for (size_t i = 0; i < vs.size(); i++) {
      CK(hipMemcpy(d, src, bytes, hipMemcpyDeviceToDevice));
      const int grid = vs[i].blocks_per_cu ? 256 * vs[i].blocks_per_cu : (int)batch;
      CK(hipEventRecord(e0));
      vs[i].k<<<grid, 256, PL::LDS_BYTES>>>(d, tw, batch);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      ms[i].push_back(t);
      if (round == 0) {
        if (i == 0) CK(hipMemcpy(ref, d, bytes, hipMemcpyDeviceToDevice));
        // compare 4 transforms spread over the batch
        double num = 0, den = 0;
        for (int q = 0; q < 4; q++) {
          size_t off = (size_t)(q * 21845 % batch) * N;
          CK(hipMemcpy(href.data() + q * N, ref + off, N * sizeof(float2), hipMemcpyDeviceToHost));
          CK(hipMemcpy(hout.data() + q * N, d + off, N * sizeof(float2), hipMemcpyDeviceToHost));
        }
        for (size_t k = 0; k < href.size(); k++) {
          double dx = href[k].x - hout[k].x, dy = href[k].y - hout[k].y;
          num += dx * dx + dy * dy; den += (double)href[k].x * href[k].x + (double)href[k].y * href[k].y;
        }
        printf("%-34s relL2 vs v0 %.2e\n", vs[i].name, sqrt(num / den));
      }
    }
  }
  for (size_t i = 0; i < vs.size(); i++) {
    std::sort(ms[i].begin(), ms[i].end());
    printf("%-34s median %.3f ms  min %.3f ms  -> %.0f GB/s (median)\n", vs[i].name, ms[i][rounds / 2], ms[i][0], 2.0 * bytes / ms[i][rounds / 2] / 1e6);
  }
  return 0;
}
