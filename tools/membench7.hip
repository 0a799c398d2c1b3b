// membench7: the four-step passes' access shapes without the arithmetic.  8 transforms of 2^22 complex f32 points as
// 2048 x 2048 matrices; a workgroup of 1024 threads moves one tile of 2048 rows x C columns (E x 8 bytes per lane,
// E = 2 C: 16 for the 2048 x 8 tiles, 32 for the 2048 x 16 tiles).
//   mode 1: read the tile strided (runs of C*8 bytes, 16 KiB apart), write it as one contiguous block
//   mode 2: read one contiguous block, write the tile strided
//   mode 0: contiguous both ways (ceiling)
// knobs: workgroups per CU (through the LDS allocation), cache policy, tile -> workgroup mapping (plain, or the pair
// mapping of the library: workgroups b and b+8 -- same XCD -- take the two tiles that share 128-byte lines).
// tools only; hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/membench7 tools/membench7.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int C, int AUX, bool PAIRED, int E = 2 * C>
__global__ __launch_bounds__(1024) void tile_kernel(const unsigned *__restrict__ src, unsigned *__restrict__ dst, int touch)
{
  extern __shared__ unsigned lds[];
  if (touch) lds[threadIdx.x] = 0;
  constexpr unsigned ROW = 2048, TILES = ROW / C, NFULL = 2048u * 2048u;
  unsigned b = blockIdx.x;
  unsigned item = b;
  if (PAIRED) {
    const unsigned x = b & 7u, s = b >> 3;
    item = 2u * ((s >> 1) * 8u + x) + (s & 1u);
  }
  const unsigned tile = item % TILES, xf = item / TILES;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(src + (size_t)xf * NFULL * 2), 0, NFULL * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(dst + (size_t)xf * NFULL * 2), 0, NFULL * 8, 0x00020000);
  const unsigned t = threadIdx.x;
  const unsigned strided = ((t / C) * ROW + tile * C + t % C) * 8u, step_s = (1024u / C) * ROW * 8u;
  const unsigned linear = ((tile * (ROW * C)) + t) * 8u, step_l = 1024u * 8u;
  u32x2 v[E];
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < E; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, strided, i * step_s, AUX);
#pragma unroll
    for (int i = 0; i < E; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], rd, linear, i * step_l, AUX); }
  } else if (MODE == 2) {
#pragma unroll
    for (int i = 0; i < E; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, linear, i * step_l, AUX);
#pragma unroll
    for (int i = 0; i < E; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], rd, strided, i * step_s, AUX); }
  } else {
#pragma unroll
    for (int i = 0; i < E; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, linear, i * step_l, AUX);
#pragma unroll
    for (int i = 0; i < E; i++) { v[i][0] += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], rd, linear, i * step_l, AUX); }
  }
}

static unsigned *g_a, *g_b;
static size_t BYTES = 8ull * 2048 * 2048 * 8;  // 256 MiB (8 transforms); argv[1] = number of transforms
static unsigned g_nx = 8;
struct Cfg { std::string name; std::function<void()> launch; std::vector<float> ms; };
static std::vector<Cfg> g_cfgs;

template <int MODE, int C, int AUX, bool PAIRED>
static void add(int lds_kb)
{
  auto k = tile_kernel<MODE, C, AUX, PAIRED>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const unsigned grid = g_nx * 2048u / C;
  char buf[160];
  snprintf(buf, sizeof buf, "mode %d C %2d aux %d %s lds %3d KiB", MODE, C, AUX, PAIRED ? "paired" : "plain ", lds_kb);
  g_cfgs.push_back({buf, [=] { k<<<grid, 1024, lds_kb * 1024>>>((const unsigned *)g_a, (unsigned *)g_b, 0); }, {}});
}

int main(int argc, char **argv)
{
  if (argc > 1) { g_nx = (unsigned)atoi(argv[1]); BYTES = (size_t)g_nx * 2048 * 2048 * 8; }
  printf("%u transforms: %zu MiB per buffer\n", g_nx, BYTES >> 20);
  CK(hipMalloc(&g_a, BYTES)); CK(hipMalloc(&g_b, BYTES));
  CK(hipMemset(g_a, 1, BYTES)); CK(hipMemset(g_b, 2, BYTES));
  for (int lds_kb : {150, 72}) {
    add<0, 8, 0, false>(lds_kb);
    add<0, 8, 2, false>(lds_kb);
    add<1, 8, 0, false>(lds_kb);
    add<1, 8, 0, true>(lds_kb);
    add<1, 8, 2, true>(lds_kb);
    add<2, 8, 0, false>(lds_kb);
    add<2, 8, 0, true>(lds_kb);
    add<2, 8, 2, true>(lds_kb);
  }
  add<1, 16, 0, false>(150);
  add<1, 16, 2, false>(150);
  add<2, 16, 0, false>(150);
  add<2, 16, 2, false>(150);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 50; i++) g_cfgs[0].launch();
  const int rounds = 7, reps = 6;
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
    printf("%s: median %.4f ms (min %.4f max %.4f) %6.0f GB/s\n", c.name.c_str(), c.ms[rounds / 2], c.ms[0], c.ms[rounds - 1], 2.0 * BYTES / c.ms[rounds / 2] / 1e6);
  }
  return 0;
}
