// membench5: the one-shot FFT access shape (one workgroup streams one 32 KiB chunk in place: all loads, then all
// stores) under the knobs the FFT kernel could turn: threads per chunk, bytes per lane, cache policy of loads and of
// stores separately, workgroups per CU (capped through a dynamic LDS allocation), chunk size.
// tools only; hipcc --offload-arch=gfx950 -O3 -o tools/membench5 tools/membench5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int T, int W, int CH, int AUXL, int AUXS, bool INPLACE, int FMAS = 0>
__global__ __launch_bounds__(T) void chunk_kernel(const unsigned *__restrict__ src, unsigned *__restrict__ dst, int touch_lds)
{
  extern __shared__ unsigned lds[];
  const size_t c = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)src + c * CH), 0, CH, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)((char *)(INPLACE ? (unsigned *)src : dst) + c * CH), 0, CH, 0x00020000);
  constexpr int PER = CH / (T * W);
  if (touch_lds) lds[threadIdx.x] = 0;  // keeps the allocation alive
  if constexpr (W == 16) {
    u32x4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, i * T * 16, AUXL);
#pragma unroll
    for (int i = 0; i < PER; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b128(v[i], rd, threadIdx.x * 16, i * T * 16, AUXS); }
  } else {
    u32x2 v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, threadIdx.x * 8, i * T * 8, AUXL);
    if constexpr (FMAS > 0) {
      // dummy arithmetic on the loaded values: FMAS dependent-free fma instructions per lane
      float a[8];
#pragma unroll
      for (int k = 0; k < 8; k++) a[k] = __uint_as_float(v[k % PER][0]);
#pragma unroll 1
      for (int it = 0; it < FMAS / 8; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = __builtin_fmaf(a[k], 1.0000001f, 1e-9f);
      }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; k++) sum += a[k];
      if (sum == 1.2345e-31f) v[0][1] += 1;
    }
#pragma unroll
    for (int i = 0; i < PER; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], rd, threadIdx.x * 8, i * T * 8, AUXS); }
  }
}

static float *g_a, *g_b;
static const size_t BYTES = 2ull << 30;
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
struct Cfg { std::string name; std::function<void()> launch; std::vector<float> ms; };
static std::vector<Cfg> g_cfgs;

template <int T, int W, int CH, int AUXL, int AUXS, bool INPLACE, int FMAS = 0>
static void add(int lds_kb)
{
  auto k = chunk_kernel<T, W, CH, AUXL, AUXS, INPLACE, FMAS>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const unsigned grid = (unsigned)(BYTES / CH);
  char buf[160];
  snprintf(buf, sizeof buf, "T%4d W%2d chunk %3d KiB auxL %d auxS %d %s lds %3d KiB fma %4d", T, W, CH >> 10, AUXL, AUXS, INPLACE ? "inplace" : "copy   ", lds_kb, FMAS);
  g_cfgs.push_back({buf, [=] { k<<<grid, T, lds_kb * 1024>>>((const unsigned *)g_a, (unsigned *)g_b, 0); }, {}});
}

int main()
{
  CK(hipMalloc(&g_a, BYTES)); CK(hipMalloc(&g_b, BYTES));
  CK(hipMemset(g_a, 1, BYTES)); CK(hipMemset(g_b, 2, BYTES));
  // 16384-point shape: one 128 KiB chunk per 1024-thread workgroup, 1 or 2 workgroups per CU, with a compute phase
  add<1024, 8, 131072, 2, 2, true, 0>(144);
  add<1024, 8, 131072, 2, 2, true, 2048>(144);
  add<1024, 8, 131072, 2, 2, true, 4096>(144);
  add<1024, 8, 131072, 2, 2, true, 0>(76);
  add<1024, 8, 131072, 2, 2, true, 2048>(76);
  add<1024, 8, 131072, 2, 2, true, 4096>(76);
  add<1024, 8, 131072, 2, 2, true, 6144>(76);
  // 8192-point shape: 64 KiB per 512-thread workgroup, 2 or 4 workgroups per CU
  add<512, 8, 65536, 2, 2, true, 0>(76);
  add<512, 8, 65536, 2, 2, true, 1536>(76);
  add<512, 8, 65536, 2, 2, true, 3072>(76);
  add<512, 8, 65536, 2, 2, true, 0>(38);
  add<512, 8, 65536, 2, 2, true, 1536>(38);
  add<512, 8, 65536, 2, 2, true, 3072>(38);
  add<512, 8, 65536, 2, 2, true, 4096>(38);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; i++) g_cfgs[0].launch();   // clocks
  const int rounds = 9, reps = 4;
  for (int r = 0; r < rounds; r++)
    for (auto &c : g_cfgs) {
      c.launch();
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; i++) c.launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      c.ms.push_back(ms / reps);
    }
  for (auto &c : g_cfgs) {
    std::sort(c.ms.begin(), c.ms.end());
    printf("%s: median %.3f ms (min %.3f max %.3f) %6.0f GB/s\n", c.name.c_str(), c.ms[rounds / 2], c.ms[0], c.ms[rounds - 1], 2.0 * BYTES / c.ms[rounds / 2] / 1e6);
  }
  return 0;
}
