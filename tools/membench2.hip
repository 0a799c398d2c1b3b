// membench2.hip -- where is the HBM streaming ceiling of this MI355X, and which access shape reaches it?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy16u(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n)
{
  // each block handles contiguous chunks of 256*U float4
  for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&in[base + u * 256 + threadIdx.x]) : in[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], &out[base + u * 256 + threadIdx.x]); else out[base + u * 256 + threadIdx.x] = v[u]; }
  }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void rmw16u(f4 *__restrict__ d, size_t n)
{
  for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&d[base + u * 256 + threadIdx.x]) : d[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u].x *= 1.0001f; if (NT) __builtin_nontemporal_store(v[u], &d[base + u * 256 + threadIdx.x]); else d[base + u * 256 + threadIdx.x] = v[u]; }
  }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void rmw8u(f2 *__restrict__ d, size_t n)
{
  for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
    f2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&d[base + u * 256 + threadIdx.x]) : d[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u].x *= 1.0001f; if (NT) __builtin_nontemporal_store(v[u], &d[base + u * 256 + threadIdx.x]); else d[base + u * 256 + threadIdx.x] = v[u]; }
  }
}
template <int U>
__global__ __launch_bounds__(256) void read16u(const float4 *__restrict__ in, float *sink, size_t n)
{
  float acc = 0;
  for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = in[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) *sink = acc;
}
template <int U>
__global__ __launch_bounds__(256) void write16u(float4 *__restrict__ out, size_t n)
{
  for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
#pragma unroll
    for (int u = 0; u < U; u++) out[base + u * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, (float)u);
  }
}

template <class F>
static double timeit(F f, int reps = 9)
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
  const size_t bytes = (size_t)2 << 30;
  float *a, *b, *sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  const size_t n16 = bytes / 16, n8 = bytes / 8;
  double t;
#define RUN(label, call, traffic) t = timeit([&] { call; }); printf("%-40s %.3f ms %6.0f GB/s\n", label, t, (traffic) / t / 1e6);
  for (int grid : {512, 1024, 2048, 4096, 16384}) {
    printf("-- grid %d\n", grid);
    RUN("read16 U8", (read16u<8><<<grid, 256>>>((float4 *)a, sink, n16)), 1.0 * bytes)
    RUN("write16 U8", (write16u<8><<<grid, 256>>>((float4 *)b, n16)), 1.0 * bytes)
    RUN("copy16 U4", (copy16u<4, false><<<grid, 256>>>((f4 *)a, (f4 *)b, n16)), 2.0 * bytes)
    RUN("copy16 U8", (copy16u<8, false><<<grid, 256>>>((f4 *)a, (f4 *)b, n16)), 2.0 * bytes)
    RUN("copy16 U8 nt", (copy16u<8, true><<<grid, 256>>>((f4 *)a, (f4 *)b, n16)), 2.0 * bytes)
    RUN("rmw16 U8", (rmw16u<8, false><<<grid, 256>>>((f4 *)a, n16)), 2.0 * bytes)
    RUN("rmw16 U8 nt", (rmw16u<8, true><<<grid, 256>>>((f4 *)a, n16)), 2.0 * bytes)
    RUN("rmw8 U16", (rmw8u<16, false><<<grid, 256>>>((f2 *)a, n8)), 2.0 * bytes)
    RUN("rmw8 U16 nt", (rmw8u<16, true><<<grid, 256>>>((f2 *)a, n8)), 2.0 * bytes)
  }
  return 0;
}
