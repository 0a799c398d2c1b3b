// membench.hip -- HBM streaming ceilings for the access shapes the FFT kernels use (MI355X).
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void copy16(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy8(const float2 *__restrict__ in, float2 *__restrict__ out, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// in-place scale, 16 B / lane
__global__ void inplace16(float4 *__restrict__ d, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = d[i]; v.x *= 1.0001f; d[i] = v;
  }
}
// FFT-shaped: one 256-thread block per 4096-point transform, thread t reads points t + 256 r
// (8 B each), holds 16 in registers, writes them back to the same places.
template <int W>
__global__ __launch_bounds__(256) void fftshape(float2 *__restrict__ d, uint32_t batch)
{
  for (uint32_t b = blockIdx.x; b < batch; b += gridDim.x) {
    float2 *p = d + (size_t)b * 4096;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = p[threadIdx.x + 256 * r];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r].x *= 1.0001f;
#pragma unroll
    for (int r = 0; r < 16; r++) p[threadIdx.x + 256 * r] = v[r];
  }
}
// same with 16 B per lane (thread t handles points 2t, 2t+1 of rows r = 0..7, 128.. )
__global__ __launch_bounds__(256) void fftshape16(float4 *__restrict__ d, uint32_t batch)
{
  for (uint32_t b = blockIdx.x; b < batch; b += gridDim.x) {
    float4 *p = d + (size_t)b * 2048;
    float4 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = p[threadIdx.x + 256 * r];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r].x *= 1.0001f;
#pragma unroll
    for (int r = 0; r < 8; r++) p[threadIdx.x + 256 * r] = v[r];
  }
}

template <class F>
static double timeit(F f, int reps = 10)
{
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int i = 0; i < reps; i++) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main()
{
  const size_t bytes = (size_t)2 << 30;  // 2 GiB
  float *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  const uint32_t batch = 65536;
  double t;
  for (int grid : {2048, 8192, 65536}) {
    t = timeit([&] { copy16<<<grid, 256>>>((float4 *)a, (float4 *)b, bytes / 16); });
    printf("copy16   grid %6d: %.3f ms  %.0f GB/s (read+write)\n", grid, t, 2.0 * bytes / t / 1e6);
    t = timeit([&] { copy8<<<grid, 256>>>((float2 *)a, (float2 *)b, bytes / 8); });
    printf("copy8    grid %6d: %.3f ms  %.0f GB/s\n", grid, t, 2.0 * bytes / t / 1e6);
    t = timeit([&] { inplace16<<<grid, 256>>>((float4 *)a, bytes / 16); });
    printf("inplace16 grid %6d: %.3f ms  %.0f GB/s\n", grid, t, 2.0 * bytes / t / 1e6);
  }
  for (int grid : {1024, 2048, 4096, 65536}) {
    t = timeit([&] { fftshape<0><<<grid, 256>>>((float2 *)a, batch); });
    printf("fftshape8  grid %6d: %.3f ms  %.0f GB/s\n", grid, t, 2.0 * bytes / t / 1e6);
    t = timeit([&] { fftshape16<<<grid, 256>>>((float4 *)a, batch); });
    printf("fftshape16 grid %6d: %.3f ms  %.0f GB/s\n", grid, t, 2.0 * bytes / t / 1e6);
  }
  return 0;
}
