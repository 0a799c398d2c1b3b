// valubench: issue rate of the integer VALU instructions the fixed-point butterflies are made of, against v_fma_f32.
// One workgroup of 256 threads per CU (one wave per SIMD), 8 independent chains per lane, wall-clock cycles per
// wave instruction per SIMD from s_memtime.  tools only; hipcc --offload-arch=gfx950 -O3 -o tools/valubench tools/valubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define BENCH(NAME, ASM)                                                                                      \
  __global__ __launch_bounds__(256) void k_##NAME(int *out, long long *cyc, int seed)                         \
  {                                                                                                           \
    int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    int b = seed * 77 + 12345, c = seed * 31 + 999;                                                           \
    const long long t0 = __builtin_readcyclecounter();                                                        \
    for (int it = 0; it < 1024; it++) {                                                                       \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                   : "v"(b), "v"(c));                                                                         \
    }                                                                                                         \
    const long long t1 = __builtin_readcyclecounter();                                                        \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                              \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
  }

#define A_MULHI(i) "v_mul_hi_i32 %" #i ", %" #i ", %8\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MAD24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n"
#define A_MUL24(i) "v_mul_i32_i24 %" #i ", %" #i ", %8\n"
#define A_MULHI24(i) "v_mul_hi_i32_i24 %" #i ", %" #i ", %8\n"
#define A_DOT2(i) "v_dot2_i32_i16 %" #i ", %" #i ", %8, %9\n"
#define A_PKADD(i) "v_pk_add_i16 %" #i ", %" #i ", %8 clamp\n"
#define A_PKASHR(i) "v_pk_ashrrev_i16 %" #i ", 1, %" #i "\n"
#define A_PKMUL(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define A_PKMAD(i) "v_pk_mad_i16 %" #i ", %" #i ", %8, %9\n"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 1, %" #i "\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_MADI16(i) "v_mad_i32_i16 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define A_BFE(i) "v_bfe_i32 %" #i ", %" #i ", 16, 16\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 16\n"

BENCH(mul_hi_i32, A_MULHI)
BENCH(mul_lo_u32, A_MULLO)
BENCH(fma_f32, A_FMA)
BENCH(add_u32, A_ADD)
BENCH(add3_u32, A_ADD3)
BENCH(mad_i32_i24, A_MAD24)
BENCH(mul_i32_i24, A_MUL24)
BENCH(mul_hi_i32_i24, A_MULHI24)
BENCH(dot2_i32_i16, A_DOT2)
BENCH(pk_add_i16_clamp, A_PKADD)
BENCH(pk_ashrrev_i16, A_PKASHR)
BENCH(pk_mul_lo_u16, A_PKMUL)
BENCH(pk_mad_i16, A_PKMAD)
BENCH(ashrrev_i32, A_ASHR)
BENCH(perm_b32, A_PERM)
BENCH(mad_i32_i16, A_MADI16)
BENCH(lshl_add_u32, A_LSHLADD)
BENCH(bfe_i32, A_BFE)
BENCH(alignbit_b32, A_ALIGNBIT)

// packed fp32 on register pairs
#define PKBENCH(NAME, INSTR)                                                                                   \
  __global__ __launch_bounds__(256) void k_##NAME(int *out, long long *cyc, int seed)                         \
  {                                                                                                           \
    typedef float v2f __attribute__((ext_vector_type(2)));                                                    \
    v2f a0 = {(float)seed, 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, a4 = a0 * 11.f, a5 = a0 * 13.f, a6 = a0 * 17.f, a7 = a0 * 19.f; \
    v2f b = {1.0001f, 0.9999f}, c = {1e-9f, 2e-9f};                                                            \
    const long long t0 = __builtin_readcyclecounter();                                                        \
    for (int it = 0; it < 1024; it++) {                                                                       \
      asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                   : "v"(b), "v"(c));                                                                         \
    }                                                                                                         \
    const long long t1 = __builtin_readcyclecounter();                                                        \
    const v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                       \
    out[blockIdx.x * 256 + threadIdx.x] = (int)(s.x + s.y);                                                   \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
  }
#define A_PKFMA32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PKADD32(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define A_PKMUL32(i) "v_pk_mul_f32 %" #i ", %" #i ", %8 op_sel:[1,1] op_sel_hi:[1,0]\n"
PKBENCH(pk_fma_f32, A_PKFMA32)
PKBENCH(pk_add_f32, A_PKADD32)
PKBENCH(pk_mul_f32_opsel, A_PKMUL32)

// 64-bit multiply-add: v_mad_i64_i32 (64-bit accumulator pair)
__global__ __launch_bounds__(256) void k_mad_i64_i32(int *out, long long *cyc, int seed)
{
  long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  int b = seed * 77 + 12345, c = seed * 31 + 999;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 1024; it++) {
#define M(i) "v_mad_i64_i32 %" #i ", vcc, %8, %9, %" #i "\n"
    asm volatile(M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b), "v"(c)
                 : "vcc");
#undef M
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = (int)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
static void run(const char *name, K k, int *out, long long *cyc)
{
  k<<<256, 256>>>(out, cyc, 3);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  k<<<256, 256>>>(out, cyc, 3);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[256];
  CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < 256; i++) s += (double)h[i];
  // s_memtime ticks at a constant 100 MHz; report the event time per instruction too
  printf("%-20s %8.3f us  -> %6.2f ns per wave instruction per SIMD (%.1f cycles at 2.4 GHz)   counter ticks avg %.0f\n", name, ms * 1e3,
         ms * 1e6 / (1024.0 * 16), ms * 1e6 / (1024.0 * 16) * 2.4, s / 256);
}

int main()
{
  int *out; long long *cyc;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  for (int rep = 0; rep < 2; rep++) {
#define R(n) run(#n, k_##n, out, cyc);
    R(pk_fma_f32) R(pk_add_f32) R(pk_mul_f32_opsel) R(fma_f32) R(add_u32) R(add3_u32) R(ashrrev_i32) R(lshl_add_u32) R(bfe_i32) R(alignbit_b32) R(perm_b32)
    R(mul_hi_i32) R(mul_lo_u32) R(mad_i64_i32) R(mad_i32_i24) R(mul_i32_i24) R(mul_hi_i32_i24) R(mad_i32_i16)
    R(dot2_i32_i16) R(pk_add_i16_clamp) R(pk_ashrrev_i16) R(pk_mul_lo_u16) R(pk_mad_i16)
  }
  return 0;
}
