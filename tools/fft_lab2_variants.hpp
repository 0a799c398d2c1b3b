// experimental arms of fft_lab2 (tools only)
#pragma once

// P1: persistent workgroups (grid = WGPCU * 256), the next transform's 16 loads per thread in flight in registers
// while the current one is transformed; loads issued in pass HP.  Items are taken grid-stride, so the chunks in
// flight at any time are neighbours, as with the one-shot kernel.
template <int LOGN, int MINW, int HP, int AUX>
__global__ __launch_bounds__((1 << LOGN) / 16, MINW) void p1_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw, uint32_t batch)
{
  using PL = Plan<LOGN, (1 << LOGN) / 16>;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const uint32_t first = blockIdx.x, stride = gridDim.x;
  if (first >= batch) return;
  TwPow<PL, ~0u, false> twp;
  twp.preload(tw, threadIdx.x);
  float2 pre[16];
  auto issue = [&](uint32_t item, int t) {
    const bool live = item < batch;
    const __amdgpu_buffer_rsrc_t rs = transform_rsrc(data + (size_t)(live ? item : 0) * PL::N, live ? PL::N * 8u : 0u);
    for_each_first_pass_input_split<PL>(t, [&](int slot, int, int j, int roff) {
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, roff * 8, AUX);
      pre[slot] = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
    });
  };
  issue(first, threadIdx.x);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (uint32_t item = first; item < batch; item += stride) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    twp.pin();
    float2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = pre[i];
    auto hook = [&]() { issue(item + stride, tid); };
    const __amdgpu_buffer_rsrc_t ro = transform_rsrc(data + (size_t)item * PL::N, PL::N * 8u);
    BufferStore<LOGN, false, 0, AUX> st{ro, 1.0f};
    fft_block_preloaded<PL, false, HP>(v, lds, twp, tid, st, hook);
    __syncthreads();
  }
}

template <class A, int LOGN, int MINW, int HP, int AUX>
static void add_p1(std::vector<A> &arms, const float2 *tw, uint32_t batch, int wgpcu, int lds_extra_kb = 0)
{
  using PL = Plan<LOGN, (1 << LOGN) / 16>;
  auto k = p1_kernel<LOGN, MINW, HP, AUX>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  char nm[96];
  snprintf(nm, sizeof nm, "P1 persist regs minw%d hp%d aux%d x%d lds+%d", MINW, HP, AUX, wgpcu, lds_extra_kb);
  const size_t ldsb = PL::LDS_BYTES + lds_extra_kb * 1024;
  arms.push_back({nm, [=](float2 *p) { k<<<256 * wgpcu, PL::T, ldsb>>>(p, tw, batch); }, {}, 0});
}


// ---- diagnostic builds: s_memtime stamps per phase (wave 0 of each workgroup), never timed as an arm ----
__device__ __forceinline__ uint64_t stamp() { return __builtin_amdgcn_s_memtime(); }

template <int LOGN, int AUX>
__global__ __launch_bounds__((1 << LOGN) / 16, 4) void d_oneshot_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw, uint32_t batch, unsigned long long *__restrict__ out)
{
  using PL = Plan<LOGN, (1 << LOGN) / 16>;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const uint64_t t0 = stamp();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x;
  const uint32_t g0 = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs = transform_rsrc(data + (size_t)g0 * PL::N, PL::N * 8u);
  float2 v[16];
  for_each_first_pass_input_split<PL>(tid, [&](int slot, int, int j, int roff) {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, roff * 8, AUX);
    v[slot] = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
  });
  TwPow<PL, ~0u, false> twp;
  twp.preload(tw, tid);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const uint64_t t1 = stamp();
  BufferStore<LOGN, false, 0, AUX> st{rs, 1.0f};
  Preloaded ld;
  NoHook hook;
  float2 twa[15], twb[15];
  fft_pass<PL, 0, false, false, -1>(v, lds, twp, tid, ld, st, twa, twb, hook);
  const uint64_t t2 = stamp();
  fft_pass<PL, 1, false, false, -1>(v, lds, twp, tid, ld, st, twa, twb, hook);
  const uint64_t t3 = stamp();
  fft_pass<PL, 2, false, false, -1>(v, lds, twp, tid, ld, st, twa, twb, hook);
  const uint64_t t4 = stamp();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const uint64_t t5 = stamp();
  if (tid == 0) {
    unsigned long long *o = out + (size_t)g0 * 8;
    o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t5 - t4; o[5] = t5 - t0; o[6] = r0; o[7] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int LOGN, int MINW, int HP, int AUX>
__global__ __launch_bounds__((1 << LOGN) / 16, MINW) void d_p1_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw, uint32_t batch, unsigned long long *__restrict__ out)
{
  using PL = Plan<LOGN, (1 << LOGN) / 16>;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const uint32_t first = blockIdx.x, stride = gridDim.x;
  if (first >= batch) return;
  TwPow<PL, ~0u, false> twp;
  twp.preload(tw, threadIdx.x);
  float2 pre[16];
  auto issue = [&](uint32_t item, int t) {
    const bool live = item < batch;
    const __amdgpu_buffer_rsrc_t rs = transform_rsrc(data + (size_t)(live ? item : 0) * PL::N, live ? PL::N * 8u : 0u);
    for_each_first_pass_input_split<PL>(t, [&](int slot, int, int j, int roff) {
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, roff * 8, AUX);
      pre[slot] = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
    });
  };
  issue(first, threadIdx.x);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  uint64_t acc[6] = {0, 0, 0, 0, 0, 0};
  const uint64_t tstart = stamp();
  const uint64_t rstart = __builtin_amdgcn_s_memrealtime();
  for (uint32_t item = first; item < batch; item += stride) {
    const uint64_t t0 = stamp();
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    twp.pin();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    const uint64_t t1 = stamp();
    float2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = pre[i];
    auto hook = [&]() { issue(item + stride, tid); };
    const __amdgpu_buffer_rsrc_t ro = transform_rsrc(data + (size_t)item * PL::N, PL::N * 8u);
    BufferStore<LOGN, false, 0, AUX> st{ro, 1.0f};
    Preloaded ld;
    float2 twa[15], twb[15];
    fft_pass<PL, 0, false, false, HP>(v, lds, twp, tid, ld, st, twa, twb, hook);
    const uint64_t t2 = stamp();
    fft_pass<PL, 1, false, false, HP>(v, lds, twp, tid, ld, st, twa, twb, hook);
    const uint64_t t3 = stamp();
    fft_pass<PL, 2, false, false, HP>(v, lds, twp, tid, ld, st, twa, twb, hook);
    const uint64_t t4 = stamp();
    __syncthreads();
    const uint64_t t5 = stamp();
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4;
  }
  const uint64_t tend = stamp();
  if (threadIdx.x == 0) {
    unsigned long long *o = out + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 5; i++) o[i] = acc[i];
    o[5] = tend - tstart; o[6] = rstart; o[7] = __builtin_amdgcn_s_memrealtime();
  }
}

static void report_diag(const char *name, unsigned long long *dout, size_t nwg, double per_wg_items)
{
  std::vector<unsigned long long> h(nwg * 8);
  CK(hipMemcpy(h.data(), dout, nwg * 64, hipMemcpyDeviceToHost));
  double s[6] = {0, 0, 0, 0, 0, 0}, rsum = 0;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (size_t w = 0; w < nwg; w++) { for (int i = 0; i < 6; i++) s[i] += (double)h[w * 8 + i]; tmin = std::min(tmin, h[w * 8 + 6]); tmax = std::max(tmax, h[w * 8 + 7]); rsum += (double)(h[w * 8 + 7] - h[w * 8 + 6]); }
  const double d = (double)nwg * per_wg_items;
  printf("%-36s cycles per transform: loadwait %.0f  pass0 %.0f  pass1 %.0f  pass2+storeissue %.0f  tail %.0f  total %.0f   (kernel span %.1f us; s_memtime clock %.0f MHz)\n", name, s[0] / d, s[1] / d, s[2] / d,
         s[3] / d, s[4] / d, s[5] / d, (double)(tmax - tmin) / 100.0, s[5] / rsum * 100.0);
}

static void run_diag(const float2 *tw, uint32_t batch, float2 *d)
{
  unsigned long long *dout;
  CK(hipMalloc(&dout, (size_t)batch * 64));
  using PL = Plan<12, 256>;
  for (int kb : {37, 76, 148}) {
    auto k = d_oneshot_kernel<12, 2>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int r = 0; r < 2; r++) k<<<batch, 256, kb * 1024>>>(d, tw, batch, dout);
    CK(hipDeviceSynchronize());
    char nm[64]; snprintf(nm, sizeof nm, "one-shot, %d wg/cu", 160 / kb);
    report_diag(nm, dout, batch, 1.0);
  }
  {
    auto k = d_p1_kernel<12, 2, 0, 2>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int x : {1, 2}) {
      for (int r = 0; r < 2; r++) k<<<256 * x, 256, (x == 1 ? 148 : 76) * 1024>>>(d, tw, batch, dout);
      CK(hipDeviceSynchronize());
      char nm[64]; snprintf(nm, sizeof nm, "P1 hp0 x%d", x);
      report_diag(nm, dout, 256 * x, (double)batch / (256 * x));
    }
  }
  CK(hipFree(dout));
}


// O2: the library's one-shot kernel with knobs: waves per SIMD assumed by the register allocator, ping-pong LDS
template <int LOGN, int MINW, bool PP, int AUX>
__global__ __launch_bounds__((1 << LOGN) / 16, MINW) void o2_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw, uint32_t batch)
{
  using PL = Plan<LOGN, (1 << LOGN) / 16, PP>;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int tid = threadIdx.x;
  const uint32_t g0 = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs = transform_rsrc(data + (size_t)g0 * PL::N, PL::N * 8u);
  float2 v[16];
  for_each_first_pass_input_split<PL>(tid, [&](int slot, int, int j, int roff) {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, roff * 8, AUX);
    v[slot] = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
  });
  TwPow<PL, ~0u, false> twp;
  twp.preload(tw, tid);
  BufferStore<LOGN, false, 0, AUX> st{rs, 1.0f};
  fft_block_preloaded<PL, false>(v, lds, twp, tid, st);
}

template <class A, int LOGN, int MINW, bool PP, int AUX>
static void add_o2(std::vector<A> &arms, const float2 *tw, uint32_t batch, int lds_kb)
{
  auto k = o2_kernel<LOGN, MINW, PP, AUX>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  char nm[96];
  snprintf(nm, sizeof nm, "O2 one-shot minw%d pp%d aux%d lds %d KiB", MINW, (int)PP, AUX, lds_kb);
  arms.push_back({nm, [=](float2 *p) { k<<<batch, (1 << LOGN) / 16, lds_kb * 1024>>>(p, tw, batch); }, {}, 0});
}


// O3: one-shot kernel for 8192 / 16384 points with the half-footprint (re / im) exchange of fft_core_f32.hpp
template <int LOGN, int MINW, bool NESTED8, int AUX, bool NESTED16 = false>
__global__ __launch_bounds__((1 << LOGN) / 16, MINW) void o3_kernel(float2 *__restrict__ data, const float2 *__restrict__ tw, uint32_t batch)
{
  using PL = Plan<LOGN, (1 << LOGN) / 16>;
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int tid = threadIdx.x;
  const uint32_t g0 = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rs = transform_rsrc(data + (size_t)g0 * PL::N, PL::N * 8u);
  float2 v[16];
  for_each_first_pass_input_split<PL>(tid, [&](int slot, int, int j, int roff) {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, j * 8, roff * 8, AUX);
    v[slot] = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
  });
  BufferStore<LOGN, false, 0, AUX> st{rs, 1.0f};
  fft_block_split<PL, NESTED8, NESTED16>(v, reinterpret_cast<float *>(lds), tw, tid, st);
}

template <class A, int LOGN, int MINW, bool NESTED8, int AUX, bool NESTED16 = false>
static void add_o3(std::vector<A> &arms, const float2 *tw, uint32_t batch, int lds_kb)
{
  auto k = o3_kernel<LOGN, MINW, NESTED8, AUX, NESTED16>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  char nm[96];
  snprintf(nm, sizeof nm, "O3 split-exchange minw%d nested8 %d nested16 %d aux%d lds %d KiB", MINW, (int)NESTED8, (int)NESTED16, AUX, lds_kb);
  arms.push_back({nm, [=](float2 *p) { k<<<batch, (1 << LOGN) / 16, lds_kb * 1024>>>(p, tw, batch); }, {}, 0});
}

template <class A>
static void add_variants(std::vector<A> &arms, int logn, const float2 *tw, uint32_t batch)
{
  if (logn == 11) {
    add_o3<A, 11, 4, false, 2>(arms, tw, batch, 9);
    add_o3<A, 11, 4, true, 2>(arms, tw, batch, 9);
    add_o3<A, 11, 4, false, 2>(arms, tw, batch, 12);
    add_o3<A, 11, 4, false, 2>(arms, tw, batch, 18);
    add_o3<A, 11, 4, false, 0>(arms, tw, batch, 9);
  }
  if (logn == 10) {
    add_o3<A, 10, 4, false, 2>(arms, tw, batch, 5);
    add_o3<A, 10, 4, true, 2>(arms, tw, batch, 5);
    add_o3<A, 10, 4, false, 2>(arms, tw, batch, 8);
    add_o3<A, 10, 4, false, 0>(arms, tw, batch, 5);
  }
  if (logn == 14) {
    add_o3<A, 14, 8, true, 2>(arms, tw, batch, 70);
    add_o3<A, 14, 8, false, 2>(arms, tw, batch, 70);
    add_o3<A, 14, 8, true, 2, true>(arms, tw, batch, 70);
    add_o3<A, 14, 8, false, 2, true>(arms, tw, batch, 70);
    add_o3<A, 14, 8, true, 0, true>(arms, tw, batch, 70);
  }
  if (logn == 13) {
    add_o3<A, 13, 4, true, 2>(arms, tw, batch, 36);
    add_o3<A, 13, 4, false, 2>(arms, tw, batch, 36);
    add_o3<A, 13, 6, false, 2>(arms, tw, batch, 36);
    add_o3<A, 13, 8, false, 2>(arms, tw, batch, 36);
    add_o3<A, 13, 8, true, 2>(arms, tw, batch, 36);
    add_o3<A, 13, 4, false, 2>(arms, tw, batch, 50);
  }
  if (logn == 12) {
    add_o3<A, 12, 4, false, 2>(arms, tw, batch, 18);
    add_o3<A, 12, 7, false, 2>(arms, tw, batch, 18);
    add_o3<A, 12, 4, false, 2>(arms, tw, batch, 22);
    add_o3<A, 12, 4, false, 2>(arms, tw, batch, 26);
    add_o3<A, 12, 4, false, 2>(arms, tw, batch, 32);
    add_o3<A, 12, 4, false, 2, true>(arms, tw, batch, 18);
    add_o3<A, 12, 4, false, 0>(arms, tw, batch, 18);
    add_o3<A, 12, 5, false, 2>(arms, tw, batch, 18);
    auto k = cfft_f32_kernel<12, false>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int kb : {76}) {
      char nm[64];
      snprintf(nm, sizeof nm, "library kernel, lds %d KiB (%d wg/cu)", kb, 160 / kb);
      arms.push_back({nm, [=](float2 *p) { k<<<batch, 256, kb * 1024>>>(p, tw, batch, batch, 0); }, {}, 0});
    }
  }
}
