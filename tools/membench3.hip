// membench3: streaming ceilings with the non-temporal policy (tools only; hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MAP: 0 = grid-stride (chunk b, b + grid, ...), 1 = blocked (workgroup b owns a contiguous range),
// 2 = XCD-blocked (the 8 XCDs interleave 32 KiB chunks inside a moving window of grid chunks -- same as 0 --
// but consecutive rounds of one workgroup are adjacent chunks of 8)
template <int AUX, bool INPLACE, int W, int MAP = 0>  // W = dwords per lane (2 or 4); persistent over 32 KiB chunks
__global__ __launch_bounds__(256) void stream_kernel(const unsigned *__restrict__ src, unsigned *__restrict__ dst, size_t n_chunks)
{
  constexpr int CH = 32768;  // bytes per chunk = one 4096-point transform
  const size_t per = (n_chunks + gridDim.x - 1) / gridDim.x;
  for (size_t k = 0; k < per; k++) {
    size_t c;
    if (MAP == 0) c = blockIdx.x + k * gridDim.x;
    else if (MAP == 1) c = (size_t)blockIdx.x * per + k;
    else c = ((size_t)blockIdx.x * 8 + (k & 7)) + (k >> 3) * 8 * gridDim.x;
    if (c >= n_chunks) break;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)src + c * CH), 0, CH, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)((char *)(INPLACE ? (unsigned *)src : dst) + c * CH), 0, CH, 0x00020000);
    constexpr int PER = CH / (256 * W * 4);
    if constexpr (W == 4) {
      u32x4 v[PER];
#pragma unroll
      for (int i = 0; i < PER; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, i * 4096, AUX);
#pragma unroll
      for (int i = 0; i < PER; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b128(v[i], rd, threadIdx.x * 16, i * 4096, AUX); }
    } else {
      u32x2 v[PER];
#pragma unroll
      for (int i = 0; i < PER; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, threadIdx.x * 8, i * 2048, AUX);
#pragma unroll
      for (int i = 0; i < PER; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], rd, threadIdx.x * 8, i * 2048, AUX); }
    }
  }
}

template <int AUX, bool INPLACE, int W, int MAP = 0>
void run(const char *name, unsigned *a, unsigned *b, size_t bytes, int grid)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t chunks = bytes / 32768;
  for (int i = 0; i < 3; i++) stream_kernel<AUX, INPLACE, W, MAP><<<grid, 256>>>(a, b, chunks);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; i++) stream_kernel<AUX, INPLACE, W, MAP><<<grid, 256>>>(a, b, chunks);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  printf("%-28s grid %6d: %.3f ms  %.0f GB/s (read+write)\n", name, grid, ms, 2.0 * bytes / ms / 1e6);
}

int main()
{
  const size_t bytes = 2ull << 30;
  unsigned *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  for (int grid : {512, 1024, 2048, 65536}) {
    run<0, false, 4>("copy16 default", a, b, bytes, grid);
    run<2, false, 4>("copy16 nt", a, b, bytes, grid);
    run<0, true, 4>("inplace16 default", a, b, bytes, grid);
    run<2, true, 4>("inplace16 nt", a, b, bytes, grid);
    run<0, true, 2>("inplace8 default", a, b, bytes, grid);
    run<2, true, 2>("inplace8 nt", a, b, bytes, grid);
    run<2, true, 2, 1>("inplace8 nt blocked", a, b, bytes, grid);
    run<2, true, 2, 2>("inplace8 nt runs-of-8", a, b, bytes, grid);
  }
  return 0;
}
