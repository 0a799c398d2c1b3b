// armfft_large.hip -- complex f32 FFT for N = 2^15 .. 2^22: four-step decomposition through HBM.
//
// No reference function runs these sizes (arm_cfft_f32 stops at 4096, arm_cfft_f32.c:1263-1280;
// the fork's radix-2 path at 16384, arm_cfft_radix2_init_f32.c:92-116; fftLen is a uint16_t); this
// is BASELINE config 4 ("large-N fp32 cfft 2^22 points four-step HBM decomposition"); the tests
// check it against a float64 transform.
//
//   N = N1 * N2,  n = n1*N2 + n2,  k = k1 + N1*k2
//   X[k1 + N1 k2] = sum_n2 W_N2^(n2 k2) * W_N^(n2 k1) * sum_n1 x[n1 N2 + n2] W_N1^(n1 k1)
//
// pass 1 (per tile of C adjacent columns n2): the N1 x C tile (C*8-byte runs) straight into registers,
//   N1-point FFT of every column, write row n2 of the N2 x N1 intermediate B[n2][k1] to the workspace
//   (blocked layout from N = 2^17: see RowStore);
// pass 2 (per tile of C adjacent columns k1 of B): the N2 x C tile into registers, times W_N^(n2 k1)
//   (formed while the loads are in flight), N2-point FFTs, and straight to X[k1 + N1 k2] (C*8-byte runs).
// Every point crosses HBM twice in each direction: 32*N bytes against the compulsory 16*N, so
// the ceiling under SURVEY 8(d)'s accounting is 50 % of the HBM roofline; both are reported.  The tile
// shapes themselves stream below the contiguous rate (tools/membench7, 8 x 2^22 points, no arithmetic:
// contiguous copy 5.9 TB/s, pass 1's shape 4.6, pass 2's 4.8), which puts the two passes' floor at
// 0.23 ms per 8 transforms = 0.29 of the roofline under 8(d).
// One workgroup = 1024 threads x 16 points = C sequences (C = 16384/N1), 72 KiB of LDS, two per CU.
#include <atomic>

#include "armfft_internal.h"
#include "fft_core_f32.hpp"

namespace armfft {

// Big twiddle W_N^m, m < N <= 2^22, as the product of three correctly rounded table entries,
//   W_N^m = a[m & 127] * b[(m >> 7) & 127] * c[m >> 14]      (a: W_N^i, b: W_N^(128 i), c: W_N^(16384 i)),
// 128 + 128 + 256 entries = 4 KiB, copied to LDS by every pass-2 workgroup: the lookups are LDS
// reads, issued while the tile's global loads are still in flight.  The table
// buffer keeps its historical size of 1 << kLoLog entries.
constexpr int kLoLog = 10;
constexpr int kBigA = 0, kBigB = 128, kBigC = 256, kBigEntries = 512;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xform_rsrc(const float2 *base, size_t n_points)
{
  // gfx9 raw buffer: stride 0, range in bytes, 32-bit data format (<= 32 MiB per transform here)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(base), 0, (int)(n_points * sizeof(float2)), 0x00020000);
}
// Cache policy of the two passes (aux of the buffer instructions, 2 = nt): the input streams in and the output streams
// out without taking room in the 256 MiB Infinity Cache, the intermediate is written cacheable -- pass 2 finds part of it
// there -- and read for the last time.  Same-box A/B against all-default: 2^22 x 8 0.272 -> 0.235 ms, 2^20 x 32 0.226 ->
// 0.174, 2^16 x 1024 (512 MiB) 0.41 against 0.35 of the roofline, 2^22 x 16 / x 32 0.281 / 0.280 against 0.261 / 0.273.
// Beyond 1 GiB pass 2 of the 2^22-point transforms keeps the default policy both ways (POL = 1: 2^22 x 64 1.90 against 1.92 ms).
constexpr int NT_L1 = 2, NT_S1 = 0, NT_L2 = 2, NT_S2 = 2;
template <int AUX = 0>
__device__ __forceinline__ float2 buf_load(const __amdgpu_buffer_rsrc_t &rs, uint32_t voff, uint32_t soff)
{
  const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, AUX);
  return make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store(const __amdgpu_buffer_rsrc_t &rs, uint32_t voff, uint32_t soff, float2 x)
{
  u32x2 w;
  w[0] = __float_as_uint(x.x);
  w[1] = __float_as_uint(x.y);
  __builtin_amdgcn_raw_buffer_store_b64(w, rs, (int)voff, (int)soff, AUX);
}

// MODE 1 store: output bin k = e0 + roff of sequence seq goes to the intermediate B[seq][k].  Layout of B: row-major
// (rows of 2^LOGA points), or -- `tiled` -- the order in which pass 2 consumes it: blocks of C2 = 16384 / n_seq adjacent
// columns k, each block holding its rows one after the other ([k / C2][seq][k % C2], 16384 points = 128 KiB per
// block).  Pass 2 then reads each of its tiles as ONE contiguous block instead of n_seq runs of
// C2 * 8 = 64 bytes; this pass writes runs of C * C2 * 8 = 512 bytes instead of whole rows.  The
// leg offset roff = r * Ns is a multiple of C2 whenever the launcher selects `tiled`, so it still
// folds into the scalar offset: (k / C2) * 16384 = (e0 / C2) * 16384 + roff * n_seq.
template <int LOGA, bool TILED>
struct RowStore {
  static constexpr bool SPLIT = true;
  __amdgpu_buffer_rsrc_t rs;
  uint32_t c0;
  int log_seq;   // log2 of the number of sequences (rows of B)
  int log_blk;   // log2 of the points in one pass-2 tile (= one block)
  __device__ __forceinline__ void operator()(int t, int e0, int roff, float2 v) const
  {
    const uint32_t seq = c0 + (uint32_t)t;
    const int lc2 = log_blk - log_seq;
    const uint32_t row_major = (seq << LOGA) + (uint32_t)e0;
    const uint32_t blocked = (((uint32_t)e0 >> lc2) << log_blk) + (seq << lc2) + ((uint32_t)e0 & ((1u << lc2) - 1u));
    const uint32_t voff = (TILED ? blocked : row_major) * 8u;
    const uint32_t soff = (TILED ? (uint32_t)roff << log_seq : (uint32_t)roff) * 8u;
    buf_store<NT_S1>(rs, voff, soff, v);
  }
};

// One-shot tile kernel: one workgroup of 1024 threads per tile, two resident per CU (72 KiB of LDS, 64 VGPRs), so one
// tile's loads and stores overlap the other's passes.  The A x C tile arrives in exactly the order the interleaved
// split exchange wants (fft_core_f32.hpp, SplitImageIL: lane -> column first, then row), so the first pass takes its
// inputs straight from the loads and the last pass stores straight to global memory: no staging copies, and the only
// LDS traffic is the exchanges between passes (real parts, then imaginary parts).
// MODE 1: tile of the input in -> FFT -> rows of the intermediate out.  Workgroups b and b + 8 -- same XCD, dispatched
//   side by side -- take the two tiles that share 128-byte lines (`paired`): the second reader of a line finds it in that
//   XCD's L2 (tools/membench7: strided reads 4.0 -> 4.6 TB/s with it; strided writes lose 8 %, so MODE 2 maps plainly).
// MODE 2: tile of the intermediate in -> * W_N^(row * column) -> FFT -> tile of the output out.  The twiddles are formed
//   from the LDS copy of the three-level table while the tile's loads are in flight.
template <bool INV, int AUX>
struct TileStore {
  static constexpr bool SPLIT = true;
  __amdgpu_buffer_rsrc_t rs;
  uint32_t other, c0;
  float scale;
  __device__ __forceinline__ void operator()(int t, int e0, int roff, float2 x) const
  {
    if (INV) x = make_float2(x.x * scale, -x.y * scale);
    buf_store<AUX>(rs, ((uint32_t)e0 * other + c0 + (uint32_t)t) * 8u, (uint32_t)roff * other * 8u, x);
  }
};

// E = points per thread: 16 (tile = 16384 points, two workgroups per CU) or 32 (tile = 32768 points, one workgroup per CU;
// used for 2048-point sub-transforms, where it doubles the strided side's runs from 64 to 128 bytes -- whole lines).
template <int LOGA, int MODE, bool INV, bool TILED, int E, int POL = 0>
__global__ __launch_bounds__(1024, E == 16 ? 8 : 4) void fourstep_tile_kernel(const float2 *__restrict__ in, float2 *__restrict__ out,
                                                                            const float2 *__restrict__ tw,      // 2^LOGA-point table
                                                                            const float2 *__restrict__ tw_big,  // three-level table of W_N
                                                                            int log_other, uint32_t tiles_per_xform, uint32_t n_items, int paired,
                                                                            int log_blk,   // MODE 1, blocked B: log2 of pass 2's tile
                                                                            float scale)
{
  using PL = Plan<LOGA, 1024, false, true, E>;
  using IM = SplitImageIL<PL>;
  constexpr int A = PL::N, C = PL::TPB, LOGC = IM::LOGC, LOGTILE = E == 16 ? 14 : 15;
  static_assert(A * C == E * 1024, "E points per thread");
  constexpr int R0 = 1 << PL::lr(0), NB0 = E / R0;
  extern __shared__ __attribute__((aligned(16))) float ldsf[];
  const uint32_t other = 1u << log_other;
  const size_t nfull = (size_t)A << log_other;
  uint32_t item = blockIdx.x;
  if (paired) {
    const uint32_t x = blockIdx.x & 7u, s = blockIdx.x >> 3;   // XCD, slot on that XCD
    item = 2u * ((s >> 1) * 8u + x) + (s & 1u);
  }
  if (item >= n_items) return;
  const int tid = threadIdx.x;
  const uint32_t tile = item % tiles_per_xform, xf = item / tiles_per_xform;
  const uint32_t c0 = tile * C;
  // leg i of thread t: row t / C + i * 1024 / C, column c0 + t % C (or, from a blocked intermediate, element t + 1024 i of
  // the tile's contiguous block: the same element); butterfly q, leg r of the first pass is leg i = q + r * NB0
  // Vector loads return in order: the two table entries a thread carries to LDS are requested BEFORE the tile, so that
  // their copy -- and, in pass 2, the barrier behind it and the twiddle products -- does not wait for the tile's data.
  static_assert(A / 2 <= 1024, "one table entry per thread");
  float2 twl_r = make_float2(0.f, 0.f), big_r = make_float2(0.f, 0.f);
  if (tid < A / 2) twl_r = tw[tid];
  if (MODE == 2 && tid < kBigEntries) big_r = tw_big[tid];
  float2 v[E];
  const __amdgpu_buffer_rsrc_t rs = xform_rsrc(in + (size_t)xf * nfull, nfull);
  constexpr bool linear_in = (MODE == 2) && TILED;
  const uint32_t voff = (linear_in ? (tile << LOGTILE) + (uint32_t)tid : ((uint32_t)tid >> LOGC) * other + c0 + ((uint32_t)tid & (C - 1))) * 8u;
  const uint32_t step = linear_in ? 1024u * 8u : (1024u >> LOGC) * other * 8u;
#pragma unroll
  for (int i = 0; i < E; i++) v[(i % NB0) * R0 + i / NB0] = buf_load<(MODE == 1 ? NT_L1 : (POL == 1 ? 0 : NT_L2))>(rs, voff, i * step);
  // The passes' twiddle bases are entries below A/2 of the A-point table: an LDS copy, so that no pass waits for a
  // global load queued behind the other resident workgroup's tile traffic (first read after the first exchange's barriers)
  float2 *big = reinterpret_cast<float2 *>(ldsf + IM::ELEMS);
  float2 *twl = big + (MODE == 2 ? kBigEntries : 0);
  if (tid < A / 2) twl[tid] = twl_r;
  if (MODE == 1) {
    if (INV) {
#pragma unroll
      for (int i = 0; i < E; i++) v[i].y = -v[i].y;  // inverse = conj(FFT(conj x)) / N
    }
  } else {
    // B[row][col] *= W_N^(row * col): row = t / C + i * 1024 / C, col = c0 + t % C; the exponent advances by a constant
    // per leg and stays below N
    if (tid < kBigEntries) big[tid] = big_r;
    __syncthreads();
    const uint32_t col = c0 + ((uint32_t)tid & (C - 1));
    uint32_t m = ((uint32_t)tid >> LOGC) * col;
    const uint32_t dm = (1024u >> LOGC) * col;
#pragma unroll
    for (int i = 0; i < E; i++) {
      const float2 w = cmul(cmul(big[kBigA + (m & 127u)], big[kBigB + ((m >> 7) & 127u)]), big[kBigC + (m >> 14)]);
      float2 &x = v[(i % NB0) * R0 + i / NB0];
      x = cmul(x, w);
      m += dm;
    }
  }
  const __amdgpu_buffer_rsrc_t ro = xform_rsrc(out + (size_t)xf * nfull, nfull);
  if constexpr (MODE == 1 && C >= 16) {
    // 16 or more sequences per tile: the lanes of a wave run across the sequences, so direct stores of B[seq][k] would be
    // 8-byte pieces 2^LOGA * 8 bytes (or one pass-2 tile row) apart.  The results go through the image once more, this time
    // sequence-major (row stride A + PADT floats: the lanes' column walk stays conflict-free), and leave with the lanes
    // along k: element m = tid + 1024 i = bin m % A of sequence m / A, whole 512-byte runs per wave.
    constexpr int PADT = C >= 32 ? 1 : 32 / C, RS = A + PADT;
    static_assert(C * RS <= IM::ELEMS, "the transposed tile fits the image");
    float2 o[E];
    int n = 0;
    auto cap = [&](int, int, float2 x) { o[n++] = x; };
    fft_block_split_il<PL, E == 16, E == 16, true>(v, ldsf, twl, tid, cap);
    float2 y[E];
    __syncthreads();  // the last pass has read the image
    for_each_last_pass_output_im<PL, IM>(tid, [&](int slot, int t, int e0, int roff) { ldsf[t * RS + e0 + roff] = o[slot].x; });
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; i++) y[i].x = ldsf[((tid + i * 1024) >> LOGA) * RS + ((tid + i * 1024) & (A - 1))];
    __syncthreads();
    for_each_last_pass_output_im<PL, IM>(tid, [&](int slot, int t, int e0, int roff) { ldsf[t * RS + e0 + roff] = o[slot].y; });
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; i++) y[i].y = ldsf[((tid + i * 1024) >> LOGA) * RS + ((tid + i * 1024) & (A - 1))];
    // B[seq][k], row-major or blocked (RowStore's layouts)
    const int lc2 = log_blk - log_other;
#pragma unroll
    for (int i = 0; i < E; i++) {
      const uint32_t seq = c0 + (uint32_t)((tid + i * 1024) >> LOGA), kk = (uint32_t)((tid + i * 1024) & (A - 1));
      const uint32_t row_major = (seq << LOGA) + kk;
      const uint32_t blocked = ((kk >> lc2) << log_blk) + (seq << lc2) + (kk & ((1u << lc2) - 1u));
      buf_store<NT_S1>(ro, (TILED ? blocked : row_major) * 8u, 0, y[i]);
    }
  } else if constexpr (MODE == 1) {
    RowStore<LOGA, TILED> st{ro, c0, log_other, log_blk};
    fft_block_split_il<PL, E == 16, E == 16, true>(v, ldsf, twl, tid, st);
  } else {
    TileStore<INV, (POL == 1 ? 0 : NT_S2)> st{ro, other, c0, scale};
    fft_block_split_il<PL, E == 16, E == 16, true>(v, ldsf, twl, tid, st);
  }
}

template <int LOGA, int MODE, bool INV, bool TILED, int E, int POL = 0>
static hipError_t launch_pass(const float2 *in, float2 *out, const float2 *tw, const float2 *tw_big, int log_other, uint32_t batch,
                              int log_blk, float scale, hipStream_t stream)
{
  using PL = Plan<LOGA, 1024, false, true, E>;
  const size_t lds_bytes = SplitImageIL<PL>::LDS_BYTES + sizeof(float2) * ((MODE == 2 ? kBigEntries : 0) + PL::N / 2);
  static std::atomic<bool> attr_done[16];  // zero-initialised; a racing second hipFuncSetAttribute is harmless
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto kern = fourstep_tile_kernel<LOGA, MODE, INV, TILED, E, POL>;
  if (dev >= 16 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if (dev < 16) attr_done[dev] = true;
  }
  const uint32_t tiles = (1u << log_other) / PL::TPB;
  const uint64_t n_items = (uint64_t)tiles * batch;
  if (n_items > 0x7fffffffull) return hipErrorInvalidValue;
  // pair mapping (64-byte runs only): whole pairs, and a grid that is a multiple of 16 (two slots on each of the 8 XCDs)
  const int paired = (MODE == 1 && PL::TPB == 8 && tiles % 2 == 0 && n_items % 16 == 0) ? 1 : 0;
  note_kernel(INV ? (TILED ? "fourstep_tile_kernel<%d, %d, true, true, %d, %d>" : "fourstep_tile_kernel<%d, %d, true, false, %d, %d>")
                  : (TILED ? "fourstep_tile_kernel<%d, %d, false, true, %d, %d>" : "fourstep_tile_kernel<%d, %d, false, false, %d, %d>"),
              MODE == 2, LOGA, MODE, E, POL);
  kern<<<(uint32_t)n_items, 1024, lds_bytes, stream>>>(in, out, tw, tw_big, log_other, tiles, (uint32_t)n_items, paired, log_blk, scale);
  return hipGetLastError();
}

__global__ void gen_twiddle_lo_kernel(float2 *tw, int logn)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << kLoLog)) return;
  // entry i of the three-level table: exponent i, 128 (i - 128), 16384 (i - 256); rest unused
  uint64_t m = 0;
  if (i < kBigB) m = i;
  else if (i < kBigC) m = (uint64_t)(i - kBigB) << 7;
  else if (i < kBigEntries) m = (uint64_t)(i - kBigC) << 14;
  m &= (1ull << logn) - 1;
  double s, c;
  sincospi(2.0 * (double)m / (double)(1ull << logn), &s, &c);
  tw[i] = make_float2((float)c, (float)(-s));
}

hipError_t launch_gen_twiddle_lo(float2 *tw, int logn, hipStream_t stream)
{
  gen_twiddle_lo_kernel<<<(1 << kLoLog) / 256, 256, 0, stream>>>(tw, logn);
  return hipGetLastError();
}

// Layout of the intermediate (see RowStore): blocked whenever pass 1's last-pass leg stride Ns is a
// multiple of pass 2's tile width C2 = 16384 / N2, which holds for N1 >= 512, i.e. N >= 2^17:
// pass 1 (LOGA = log2 N1) is blocked from LOGA = 9, pass 2 (LOGA = log2 N2) from LOGA = 9 and,
// for N = 2^17 only, at LOGA = 8.
// points per thread of the kernel that runs 2^loga-point sub-transforms (see fourstep_tile_kernel): 32 only for 2048-point
// sub-transforms, and there by measurement (2^22 points, same box):
//   pass 1 (strided reads) wants the 128-byte runs of the 2048 x 16 tiles as soon as they fill the chip twice (batch 1: 128
//   such tiles, 0.20 of the roofline against 0.225 with 2048 x 8 tiles);
//   pass 2 reads the intermediate, part of which is still in the Infinity Cache while the whole job is small (see NT_*): up to
//   2^25 points (256 MiB) two 2048 x 8 workgroups per CU overlap better than one 2048 x 16 (batch 4: 0.316 against 0.264,
//   batch 8: 0.316 against 0.285, batch 12: equal), beyond that the longer runs of the strided writes win (batch 64: 0.279
//   against 0.259).
static int tile_e(int loga, int mode, int logn, uint32_t batch)
{
  if (loga != 11) return 16;
  const uint64_t points = (uint64_t)batch << logn;
  if (mode == 2) return points <= (1ull << 25) ? 16 : 32;
  return (points >> 15) >= 512 ? 32 : 16;
}

template <int MODE, bool INV>
static hipError_t launch_pass_log(int loga, int e32, const float2 *in, float2 *out, const float2 *tw, const float2 *tw_big, int log_other,
                                  uint32_t batch, int log_blk, float scale, bool tiled, hipStream_t stream)
{
  switch (loga) {
#define C(L, T, E) case L: return launch_pass<L, MODE, INV, T, E>(in, out, tw, tw_big, log_other, batch, log_blk, scale, stream);
    C(7, false, 16)
    case 8:
      if (MODE == 2 && tiled) return launch_pass<8, MODE, INV, (MODE == 2), 16>(in, out, tw, tw_big, log_other, batch, log_blk, scale, stream);
      return launch_pass<8, MODE, INV, false, 16>(in, out, tw, tw_big, log_other, batch, log_blk, scale, stream);
    C(9, true, 16) C(10, true, 16)
    case 11:
      if (e32 && MODE == 2 && ((uint64_t)batch << (11 + log_other)) > (1ull << 27))  // beyond 1 GiB: default cache policy
        return launch_pass<11, MODE, INV, true, 32, (MODE == 2 ? 1 : 0)>(in, out, tw, tw_big, log_other, batch, log_blk, scale, stream);
      if (e32) return launch_pass<11, MODE, INV, true, 32>(in, out, tw, tw_big, log_other, batch, log_blk, scale, stream);
      return launch_pass<11, MODE, INV, true, 16>(in, out, tw, tw_big, log_other, batch, log_blk, scale, stream);
#undef C
  }
  return hipErrorInvalidValue;
}

// data -> work (pass 1) -> data (pass 2).  tw1/tw2: tables of the N1- and N2-point transforms.
hipError_t launch_cfft_large_f32(int logn, float2 *data, float2 *work, const float2 *tw1, const float2 *tw2,
                                 const float2 *tw_big, uint32_t batch, bool inverse, hipStream_t stream)
{
  if (batch == 0) return hipSuccess;
  const int log2_ = logn / 2, log1 = logn - log2_;  // N1 >= N2
  const float scale = 1.0f / (float)(1ull << logn);
  const bool tiled = logn >= 17;
  const int e1 = tile_e(log1, 1, logn, batch), e2 = tile_e(log2_, 2, logn, batch);
  const int log_blk = e2 == 32 ? 15 : 14;  // the block of the blocked intermediate = pass 2's tile
  hipError_t e;
  if (!inverse) {
    e = launch_pass_log<1, false>(log1, e1 == 32, data, work, tw1, tw_big, log2_, batch, log_blk, 1.0f, tiled, stream);
    if (e != hipSuccess) return e;
    return launch_pass_log<2, false>(log2_, e2 == 32, work, data, tw2, tw_big, log1, batch, log_blk, 1.0f, tiled, stream);
  }
  e = launch_pass_log<1, true>(log1, e1 == 32, data, work, tw1, tw_big, log2_, batch, log_blk, 1.0f, tiled, stream);
  if (e != hipSuccess) return e;
  return launch_pass_log<2, true>(log2_, e2 == 32, work, data, tw2, tw_big, log1, batch, log_blk, scale, tiled, stream);
}

}  // namespace armfft

// ------------------------------------------------------------------------------------------
// allocation-free variant of arm_cfft_u32_f32_dev (no counterpart in the reference; for callers that
// cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid it)
// ------------------------------------------------------------------------------------------
// arm_cfft_u32_f32_dev needs a caller-provided workspace of the data's size for the intermediate of the
// four-step decomposition.  Here that workspace is a file-scope device array: 64 MiB per GPU, held from
// the moment the code object is loaded on that GPU until the process ends, shared by every call on that
// GPU (calls that overlap on one GPU race on it; one stream, or events between streams, order them).
// Pass 1 overwrites every word of the part pass 2 then reads, so nothing needs clearing between calls.
// In host code `armfft_noalloc_scratch` is only a handle to the symbol, not a device address: the per-device
// address comes from hipGetSymbolAddress (each GPU holds its own copy at its own address) and only that
// pointer is ever handed to kernels.
namespace {
constexpr size_t kNoAllocScratchBytes = (size_t)64 << 20;                         // 2^23 complex f32 points
constexpr uint64_t kNoAllocMaxPoints = kNoAllocScratchBytes / sizeof(float2);
}  // namespace
static __device__ __align__(256) unsigned char armfft_noalloc_scratch[kNoAllocScratchBytes];

namespace {
std::atomic<float2 *> g_scratch_ptr[16];   // per device, resolved once with that device current
// per (device, log2 fftLen): arm_cfft_u32_f32_noalloc_prepare ran to the end -- tables built, scratch address resolved, every
// kernel of that size launched once (so its attributes are set).  Tables that arm_cfft_u32_f32_dev happened to create for
// the same sub-sizes are NOT proof of that: the allocation-free call asks this flag, nothing else.
std::atomic<bool> g_noalloc_prepared[16][armfft::kMaxLargeLog + 1];

float2 *noalloc_scratch(int dev)
{
  using namespace armfft;
  if (dev < 0 || dev >= 16) { set_error("device index %d out of range", dev); return nullptr; }
  float2 *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (p) return p;
  void *q = nullptr;
  if (!hip_ok(hipGetSymbolAddress(&q, HIP_SYMBOL(armfft_noalloc_scratch)), "hipGetSymbolAddress(scratch)") || !q) return nullptr;
  g_scratch_ptr[dev].store(reinterpret_cast<float2 *>(q), std::memory_order_release);   // same value from any racing thread
  return reinterpret_cast<float2 *>(q);
}
}  // namespace

extern "C" {

// set-up (may allocate: the twiddle tables of this size on the current device): builds the tables, resolves the
// scratch address and runs one forward and one inverse transform inside the scratch array so that every
// per-kernel attribute is set before the first allocation-free call
arm_status arm_cfft_u32_f32_noalloc_prepare(uint32_t fftLen)
{
  using namespace armfft;
  if (fftLen == 0 || (fftLen & (fftLen - 1))) { set_error("arm_cfft_u32_f32_noalloc_prepare: fftLen %u is not a power of two", fftLen); return ARM_MATH_ARGUMENT_ERROR; }
  int l = 0;
  while ((1u << l) < fftLen) l++;
  if (l < kMinLargeLog || l > kMaxLargeLog) { set_error("arm_cfft_u32_f32_noalloc_prepare: fftLen %u outside 2^%d..2^%d", fftLen, kMinLargeLog, kMaxLargeLog); return ARM_MATH_ARGUMENT_ERROR; }
  int dev = 0;
  if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return ARM_MATH_ARGUMENT_ERROR;
  float2 *scr = noalloc_scratch(dev);
  if (!scr) return ARM_MATH_ARGUMENT_ERROR;
  const int l2 = l / 2, l1 = l - l2;
  const float2 *tw1 = dev_twiddle_f32(l1), *tw2 = dev_twiddle_f32(l2), *big = dev_twiddle_lo(l);
  if (!tw1 || !tw2 || !big) return ARM_MATH_ARGUMENT_ERROR;
  // data = first half of the scratch array, workspace = second half (2^22 points x 8 bytes = 32 MiB each at most)
  float2 *work = scr + kNoAllocMaxPoints / 2;
  bool ok = hip_ok(hipMemsetAsync(scr, 0, sizeof(float2) * (size_t)fftLen, nullptr), "hipMemsetAsync(scratch)") &&
            hip_ok(launch_cfft_large_f32(l, scr, work, tw1, tw2, big, 1, false, nullptr), "four-step warm-up (forward)") &&
            hip_ok(launch_cfft_large_f32(l, scr, work, tw1, tw2, big, 1, true, nullptr), "four-step warm-up (inverse)") &&
            hip_ok(hipStreamSynchronize(nullptr), "synchronise");
  if (ok) g_noalloc_prepared[dev][l].store(true, std::memory_order_release);
  return ok ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

// The allocation-free call itself: no hipMalloc / hipFree / allocating routine on any path.  Requires
// arm_cfft_u32_f32_noalloc_prepare(S->fftLen) earlier on this device, batch * fftLen <= 2^23 points.
arm_status arm_cfft_u32_f32_noalloc_dev(const arm_cfft_instance_u32_f32 *S, float32_t *p, uint32_t batch, uint8_t ifft, void *stream)
{
  using namespace armfft;
  const char *who = "arm_cfft_u32_f32_noalloc_dev";
  if (!S || !p) { set_error("%s: null argument", who); return ARM_MATH_ARGUMENT_ERROR; }
  if (((uintptr_t)p & 15u) != 0) { set_error("%s: device pointer %p is not 16-byte aligned", who, (void *)p); return ARM_MATH_ARGUMENT_ERROR; }
  const uint32_t n = S->fftLen;
  int l = 0;
  while (l < 31 && (1u << l) < n) l++;
  if (n == 0 || (n & (n - 1)) || l < kMinLargeLog || l > kMaxLargeLog) {
    set_error("%s: fftLen %u is not a power of two in 2^%d..2^%d (smaller transforms need no workspace: arm_cfft_f32_dev)", who, n, kMinLargeLog, kMaxLargeLog);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if ((uint64_t)batch * n > kNoAllocMaxPoints) {
    set_error("%s: batch %u x fftLen %u = %llu points exceeds the built-in workspace of %llu points (64 MiB); split the batch or use arm_cfft_u32_f32_dev", who,
              batch, n, (unsigned long long)batch * n, (unsigned long long)kNoAllocMaxPoints);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  if (batch == 0) return ARM_MATH_SUCCESS;
  int dev = 0;
  if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return ARM_MATH_ARGUMENT_ERROR;
  // everything below only READS what prepare left behind: no hipGetSymbolAddress, no hipFuncSetAttribute on this path
  const bool prepared = dev >= 0 && dev < 16 && g_noalloc_prepared[dev][l].load(std::memory_order_acquire);
  float2 *scr = prepared ? g_scratch_ptr[dev].load(std::memory_order_acquire) : nullptr;
  const int l2 = l / 2, l1 = l - l2;
  const float2 *tw1 = dev_twiddle_f32_peek(l1), *tw2 = dev_twiddle_f32_peek(l2), *big = dev_twiddle_lo_peek(l);
  if (!prepared || !scr || !tw1 || !tw2 || !big) {
    set_error("%s: no plan for fftLen %u on device %d: call arm_cfft_u32_f32_noalloc_prepare(%u) once first (this entry point never allocates)", who, n, dev, n);
    return ARM_MATH_ARGUMENT_ERROR;
  }
  hipError_t e = launch_cfft_large_f32(l, reinterpret_cast<float2 *>(p), scr, tw1, tw2, big, batch, ifft != 0, (hipStream_t)stream);
  return hip_ok(e, "cfft_large_f32 launch") ? ARM_MATH_SUCCESS : ARM_MATH_ARGUMENT_ERROR;
}

}  // extern "C"
