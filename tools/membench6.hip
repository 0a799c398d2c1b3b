// membench6: 32 KiB in-place chunks, loads by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction) issued by
// LW waves of the workgroup (the extra ones exit after the landing barrier), stores by the 4 "compute" waves as 16 x 8 B
// per lane (the FFT's store shape) or 8 x 16 B.  Compared with the plain register path of membench5.
// tools only; hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/membench6 tools/membench6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void *lds_ptr_t;

// LW loader waves (LW * 64 threads launch); waves >= 4 exit after the barrier
template <int LW, int AUXL, int AUXS, int SW>
__global__ __launch_bounds__(LW * 64) void dma_chunk_kernel(unsigned *__restrict__ data)
{
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  constexpr int CH = 32768;
  const size_t c = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((char *)data + c * CH), 0, CH, 0x00020000);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int PER = 32 / LW;  // 1 KiB pieces per wave
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int piece = wave * PER + i;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + piece * 256), 16, lane * 16, piece * 1024, 0, AUXL);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wave >= 4) return;
  const int tid = threadIdx.x;
  if constexpr (SW == 8) {
    u32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = *reinterpret_cast<const u32x2 *>(lds + (tid + i * 256) * 2);
#pragma unroll
    for (int i = 0; i < 16; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], rs, tid * 8, i * 2048, AUXS); }
  } else {
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const u32x4 *>(lds + (tid + i * 256) * 4);
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b128(v[i], rs, tid * 16, i * 4096, AUXS); }
  }
}

// reference: plain register path, T256, 16 x 8 B
template <int AUX>
__global__ __launch_bounds__(256) void reg_chunk_kernel(unsigned *__restrict__ data)
{
  constexpr int CH = 32768;
  const size_t c = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((char *)data + c * CH), 0, CH, 0x00020000);
  u32x2 v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, threadIdx.x * 8, i * 2048, AUX);
#pragma unroll
  for (int i = 0; i < 16; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], rs, threadIdx.x * 8, i * 2048, AUX); }
}


// register-path loaders: LW waves load (32/LW KiB each, 8 B per lane) and pass the chunk through LDS; SW waves store.
// LW = 16, SW = 4: many loaders that exit early; LW = 4, SW = 16: few loaders, many storers (all alive to the end)
template <int LW, int SW, int AUX>
__global__ __launch_bounds__((LW > SW ? LW : SW) * 64) void split_chunk_kernel(unsigned *__restrict__ data)
{
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  constexpr int CH = 32768;
  const size_t c = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((char *)data + c * CH), 0, CH, 0x00020000);
  const int wave = threadIdx.x >> 6;
  const int tid = threadIdx.x;
  if (wave < LW) {
    constexpr int PER = 4096 / (LW * 64);   // 8-byte elements per thread
    u32x2 v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, tid * 8, i * LW * 512, AUX);
#pragma unroll
    for (int i = 0; i < PER; i++) *reinterpret_cast<u32x2 *>(lds + (tid + i * LW * 64) * 2) = v[i];
  }
  __syncthreads();
  if (wave >= SW) return;
  {
    constexpr int PER = 4096 / (SW * 64);
    u32x2 v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) v[i] = *reinterpret_cast<const u32x2 *>(lds + (tid + i * SW * 64) * 2);
#pragma unroll
    for (int i = 0; i < PER; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], rs, tid * 8, i * SW * 512, AUX); }
  }
}

static unsigned *g_a;
static const size_t BYTES = 2ull << 30;
struct Cfg { std::string name; std::function<void()> launch; std::vector<float> ms; };
static std::vector<Cfg> g_cfgs;

template <int LW, int AUXL, int AUXS, int SW>
static void add(int lds_kb)
{
  auto k = dma_chunk_kernel<LW, AUXL, AUXS, SW>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  char buf[160];
  snprintf(buf, sizeof buf, "dma loaders %2d waves auxL %d auxS %d store %2d B lds %3d KiB", LW, AUXL, AUXS, SW, lds_kb);
  g_cfgs.push_back({buf, [=] { k<<<(unsigned)(BYTES / 32768), LW * 64, lds_kb * 1024>>>(g_a); }, {}});
}

template <int LW, int SW>
static void add_split(int lds_kb)
{
  auto k = split_chunk_kernel<LW, SW, 2>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  char buf[160];
  snprintf(buf, sizeof buf, "split: %2d load waves, %2d store waves, lds %3d KiB", LW, SW, lds_kb);
  g_cfgs.push_back({buf, [=] { k<<<(unsigned)(BYTES / 32768), (LW > SW ? LW : SW) * 64, lds_kb * 1024>>>(g_a); }, {}});
}

int main()
{
  CK(hipMalloc(&g_a, BYTES));
  CK(hipMemset(g_a, 1, BYTES));
  {
    auto k = reg_chunk_kernel<2>;
    g_cfgs.push_back({"register path T256 nt", [=] { k<<<(unsigned)(BYTES / 32768), 256, 36 * 1024>>>(g_a); }, {}});
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    g_cfgs.push_back({"register path T256 nt lds 76", [=] { k<<<(unsigned)(BYTES / 32768), 256, 76 * 1024>>>(g_a); }, {}});
  }
  for (int lds_kb : {36, 52, 76}) {
    add_split<4, 4>(lds_kb);
    add_split<16, 4>(lds_kb);
    add_split<4, 16>(lds_kb);
    add_split<16, 16>(lds_kb);
    add_split<8, 4>(lds_kb);
    add_split<8, 8>(lds_kb);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; i++) g_cfgs[0].launch();
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
    printf("%-64s median %.3f ms (min %.3f max %.3f) %6.0f GB/s\n", c.name.c_str(), c.ms[rounds / 2], c.ms[0], c.ms[rounds - 1], 2.0 * BYTES / c.ms[rounds / 2] / 1e6);
  }
  // sanity: every word was incremented the same number of times in its first lane-word
  return 0;
}
