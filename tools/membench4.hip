// membench4: what does this box stream at?  The plainest copy forms (one 16-byte element per thread, huge grids),
// unrolled block-contiguous forms, grid-stride persistent forms, read-only and write-only ceilings, in place and
// out of place, several footprints.  tools only; hipcc --offload-arch=gfx950 -O3 -o tools/membench4 tools/membench4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// U elements per thread, block-contiguous: element (blockIdx * U + u) * blockDim + threadIdx
template <class V, int U, bool NT>
__global__ void copy_block(const V *__restrict__ src, V *__restrict__ dst, size_t n)
{
  const size_t base = (size_t)blockIdx.x * U * blockDim.x + threadIdx.x;
  V v[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const size_t i = base + (size_t)u * blockDim.x;
    if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const size_t i = base + (size_t)u * blockDim.x;
    if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
  }
}

// grid-stride persistent
template <class V, int U, bool NT>
__global__ void copy_stride(const V *__restrict__ src, V *__restrict__ dst, size_t n)
{
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += step * U) {
    V v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t i = i0 + u * step; if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i]; }
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t i = i0 + u * step; if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; } }
  }
}

template <class V, int U, bool NT>
__global__ void read_only(const V *__restrict__ src, float *__restrict__ sink, size_t n)
{
  const size_t base = (size_t)blockIdx.x * U * blockDim.x + threadIdx.x;
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const size_t i = base + (size_t)u * blockDim.x;
    if (i < n) { V v = NT ? __builtin_nontemporal_load(src + i) : src[i]; acc += v[0]; }
  }
  if (acc == 1.2345e-30f) sink[0] = acc;
}

template <class V, int U, bool NT>
__global__ void write_only(V *__restrict__ dst, size_t n)
{
  const size_t base = (size_t)blockIdx.x * U * blockDim.x + threadIdx.x;
  V v;
  for (int k = 0; k < (int)(sizeof(V) / 4); k++) v[k] = (float)threadIdx.x;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const size_t i = base + (size_t)u * blockDim.x;
    if (i < n) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
  }
}

template <class F>
static float timeit(F f, int reps = 10)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) f();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / reps;
}

template <class V, int U, bool NT>
static void run_block(const char *name, float *a, float *b, size_t bytes, int bs, bool inplace)
{
  const size_t n = bytes / sizeof(V);
  const unsigned grid = (unsigned)((n + (size_t)bs * U - 1) / ((size_t)bs * U));
  float ms = timeit([&] { copy_block<V, U, NT><<<grid, bs>>>((const V *)a, (V *)(inplace ? a : b), n); });
  printf("%-10s W%2zu U%d bs%4d %s %s %5zu MiB: %.3f ms %6.0f GB/s\n", name, sizeof(V), U, bs, NT ? "nt " : "def", inplace ? "inplace" : "copy   ", bytes >> 20, ms,
         2.0 * bytes / ms / 1e6);
}

template <class V, int U, bool NT>
static void run_stride(float *a, float *b, size_t bytes, int bs, int grid, bool inplace)
{
  const size_t n = bytes / sizeof(V);
  float ms = timeit([&] { copy_stride<V, U, NT><<<grid, bs>>>((const V *)a, (V *)(inplace ? a : b), n); });
  printf("stride     W%2zu U%d bs%4d grid %5d %s %s %5zu MiB: %.3f ms %6.0f GB/s\n", sizeof(V), U, bs, grid, NT ? "nt " : "def", inplace ? "inplace" : "copy   ", bytes >> 20, ms,
         2.0 * bytes / ms / 1e6);
}

int main()
{
  const size_t maxb = 4ull << 30;
  float *a, *b; CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb));
  CK(hipMemset(a, 1, maxb)); CK(hipMemset(b, 2, maxb));
  for (size_t bytes : {512ull << 20, 2ull << 30, 4ull << 30}) {
    for (int bs : {256, 512, 1024}) {
      run_block<f32x4, 1, false>("block", a, b, bytes, bs, false);
      run_block<f32x4, 1, true>("block", a, b, bytes, bs, false);
      run_block<f32x4, 2, false>("block", a, b, bytes, bs, false);
      run_block<f32x4, 4, false>("block", a, b, bytes, bs, false);
      run_block<f32x4, 4, true>("block", a, b, bytes, bs, false);
      run_block<f32x4, 8, false>("block", a, b, bytes, bs, false);
      run_block<f32x4, 8, true>("block", a, b, bytes, bs, false);
      run_block<f32x4, 1, false>("block", a, b, bytes, bs, true);
      run_block<f32x4, 4, true>("block", a, b, bytes, bs, true);
      run_block<f32x4, 8, true>("block", a, b, bytes, bs, true);
      run_block<f32x2, 4, true>("block", a, b, bytes, bs, false);
      run_block<f32x2, 16, true>("block", a, b, bytes, bs, false);
      run_block<f32x2, 16, true>("block", a, b, bytes, bs, true);
      run_block<f32x2, 16, false>("block", a, b, bytes, bs, true);
    }
    for (int grid : {256, 512, 1024, 2048, 4096, 8192}) {
      run_stride<f32x4, 1, false>(a, b, bytes, 256, grid, false);
      run_stride<f32x4, 4, false>(a, b, bytes, 256, grid, false);
      run_stride<f32x4, 4, true>(a, b, bytes, 256, grid, false);
      run_stride<f32x4, 4, true>(a, b, bytes, 256, grid, true);
      run_stride<f32x4, 4, false>(a, b, bytes, 1024, grid / 4 > 0 ? grid / 4 : 1, false);
    }
    {
      const size_t n = bytes / 16;
      for (int bs : {256, 1024}) {
        float ms = timeit([&] { read_only<f32x4, 4, false><<<(unsigned)(n / (bs * 4)), bs>>>((const f32x4 *)a, b, n); });
        printf("read-only  W16 U4 bs%4d def %5zu MiB: %.3f ms %6.0f GB/s\n", bs, bytes >> 20, ms, 1.0 * bytes / ms / 1e6);
        ms = timeit([&] { read_only<f32x4, 4, true><<<(unsigned)(n / (bs * 4)), bs>>>((const f32x4 *)a, b, n); });
        printf("read-only  W16 U4 bs%4d nt  %5zu MiB: %.3f ms %6.0f GB/s\n", bs, bytes >> 20, ms, 1.0 * bytes / ms / 1e6);
        ms = timeit([&] { write_only<f32x4, 4, false><<<(unsigned)(n / (bs * 4)), bs>>>((f32x4 *)b, n); });
        printf("write-only W16 U4 bs%4d def %5zu MiB: %.3f ms %6.0f GB/s\n", bs, bytes >> 20, ms, 1.0 * bytes / ms / 1e6);
        ms = timeit([&] { write_only<f32x4, 4, true><<<(unsigned)(n / (bs * 4)), bs>>>((f32x4 *)b, n); });
        printf("write-only W16 U4 bs%4d nt  %5zu MiB: %.3f ms %6.0f GB/s\n", bs, bytes >> 20, ms, 1.0 * bytes / ms / 1e6);
      }
    }
    fflush(stdout);
  }
  return 0;
}
