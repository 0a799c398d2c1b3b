// Issue rate of the vector instructions the kernels are made of, in wave64 instructions per second and SIMD.  Measurement tool only: nothing in the library uses it.
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate [waves per SIMD]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

constexpr int kIters = 8192;  // iterations of eight independent instructions
constexpr int kChains = 8;

struct AddU32 {
    static constexpr const char* name = "v_add_u32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct XorB32 {
    static constexpr const char* name = "v_xor_b32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct AddF32 {
    static constexpr const char* name = "v_add_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct SubI32 {
    static constexpr const char* name = "v_sub_i32 (vop3)";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_sub_i32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct AddU32E64 {
    static constexpr const char* name = "v_add_u32_e64";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct AddU32Sgpr {
    static constexpr const char* name = "v_add_u32 v, s, v";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "s"(b)); }
};
struct FmacF32 {
    static constexpr const char* name = "v_fmac_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a) : "v"(b)); }
};
struct MulF32 {
    static constexpr const char* name = "v_mul_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct AshrI32 {
    static constexpr const char* name = "v_ashrrev_i32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a)); }
};
struct MaxF32 {
    static constexpr const char* name = "v_max_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct MovB32 {
    static constexpr const char* name = "v_mov_b32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b)); }
};
struct Cndmask {
    static constexpr const char* name = "v_cndmask_b32 (vcc)";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); }
};
struct CmpGt {
    static constexpr const char* name = "v_cmp_gt_f32 (vcc)";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc"); }
};
struct PkAddF32 {
    static constexpr const char* name = "v_pk_add_f32";
    __device__ static void op(int&, int, long long& w) { asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(w)); }
};
struct PkMulF32 {
    static constexpr const char* name = "v_pk_mul_f32";
    __device__ static void op(int&, int, long long& w) { asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(w)); }
};
struct Mad32_24 {
    static constexpr const char* name = "v_mad_i32_i24";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a) : "v"(b)); }
};
struct CndmaskSet {
    static constexpr const char* name = "v_cndmask_b32 (vcc set)";
    static constexpr bool kSetVcc = true;
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); }
};
struct CndmaskSgpr {
    static constexpr const char* name = "v_cndmask_b32 (s[n:n+1])";
    __device__ static void op(int& a, int b, long long&) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(0x5555aaaa5555aaaaull));
    }
};
struct CmpSgpr {
    static constexpr const char* name = "v_cmp_gt_f32 (s[n:n+1])";
    __device__ static void op(int& a, int b, long long&) {
        unsigned long long m;
        asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    }
};
struct SubF32 {
    static constexpr const char* name = "v_sub_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct SubU32 {
    static constexpr const char* name = "v_sub_u32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct MulF32Sgpr {
    static constexpr const char* name = "v_mul_f32 v, s, v";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "s"(b)); }
};
struct MulF32Const {
    static constexpr const char* name = "v_mul_f32 v, 0.5, v";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a)); }
};
struct MulF32Lit {
    static constexpr const char* name = "v_mul_f32 v, literal, v";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mul_f32 %0, 0x3f3504f3, %0" : "+v"(a)); }
};
struct MinF32 {
    static constexpr const char* name = "v_min_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct MaxI32 {
    static constexpr const char* name = "v_max_i32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct Max3F32 {
    static constexpr const char* name = "v_max3_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); }
};
struct AndB32 {
    static constexpr const char* name = "v_and_b32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct LshlB32 {
    static constexpr const char* name = "v_lshlrev_b32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a)); }
};
struct LshlAdd {
    static constexpr const char* name = "v_lshl_add_u32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b)); }
};
struct Add3 {
    static constexpr const char* name = "v_add3_u32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); }
};
struct AddCo {
    static constexpr const char* name = "v_add_co_u32 (vcc)";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a) : "v"(b) : "vcc"); }
};
struct CvtF32I32 {
    static constexpr const char* name = "v_cvt_f32_i32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a)); }
};
struct AddF32Diff {
    static constexpr const char* name = "v_add_f32 d, a, b";
    __device__ static void op(int& a, int b, long long& w) {
        int d;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    }
};
struct CndmaskE64Vcc {
    static constexpr const char* name = "v_cndmask_b32_e64 .. vcc";
    static constexpr bool kSetVcc = true;
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); }
};
struct CmpCndVcc {
    static constexpr const char* name = "v_cmp vcc + v_cndmask vcc (pair)";
    __device__ static void op(int& a, int b, long long&) {
        asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
    }
};
struct CmpCndSgpr {
    static constexpr const char* name = "v_cmp s[] + v_cndmask s[] (pair)";
    __device__ static void op(int& a, int b, long long&) {
        unsigned long long m;
        asm volatile("v_cmp_gt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a), "=&s"(m) : "v"(b));
    }
};
struct AddcVcc {
    static constexpr const char* name = "v_addc_co_u32 (vcc in/out)";
    static constexpr bool kSetVcc = true;
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc"); }
};
struct CmpCnd2Vcc {
    static constexpr const char* name = "v_cmp vcc + 2 v_cndmask vcc";
    __device__ static void op(int& a, int b, long long& w) {
        int& i = reinterpret_cast<int*>(&w)[0];
        asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(a), "+v"(i) : "v"(b) : "vcc");
    }
};
struct CmpCnd4Vcc {
    static constexpr const char* name = "v_cmp vcc + 4 v_cndmask vcc";
    __device__ static void op(int& a, int b, long long& w) {
        int& i = reinterpret_cast<int*>(&w)[0];
        int& k = reinterpret_cast<int*>(&w)[1];
        asm volatile("v_cmp_gt_f32 vcc, %0, %3\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc\n\t"
                     "v_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(a), "+v"(i), "+v"(k) : "v"(b) : "vcc");
    }
};
struct CmpCnd2Sgpr {
    static constexpr const char* name = "v_cmp s[] + 2 v_cndmask s[]";
    __device__ static void op(int& a, int b, long long& w) {
        int& i = reinterpret_cast<int*>(&w)[0];
        unsigned long long m;
        asm volatile("v_cmp_gt_f32_e64 %2, %0, %3\n\tv_cndmask_b32_e64 %0, %0, %3, %2\n\tv_cndmask_b32_e64 %1, %1, %3, %2" : "+v"(a), "+v"(i), "=&s"(m) : "v"(b));
    }
};
struct CmpCnd2E64Vcc {
    static constexpr const char* name = "v_cmp vcc + 2 v_cndmask_e64 vcc";
    __device__ static void op(int& a, int b, long long& w) {
        int& i = reinterpret_cast<int*>(&w)[0];
        asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, vcc\n\tv_cndmask_b32_e64 %1, %1, %2, vcc" : "+v"(a), "+v"(i) : "v"(b) : "vcc");
    }
};
struct CmpCnd2Spaced {
    static constexpr const char* name = "v_cmp vcc, cnd, 4 adds, cnd";
    __device__ static void op(int& a, int b, long long& w) {
        int& i = reinterpret_cast<int*>(&w)[0];
        int& k = reinterpret_cast<int*>(&w)[1];
        asm volatile("v_cmp_gt_f32 vcc, %0, %3\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\t"
                     "v_add_u32 %2, %2, %3\n\tv_add_u32 %2, %2, %3\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(a), "+v"(i), "+v"(k) : "v"(b) : "vcc");
    }
};
struct Mul24Sdwa {
    static constexpr const char* name = "v_mul_i32_i24_sdwa";
    __device__ static void op(int& a, int b, long long&) {
        asm volatile("v_mul_i32_i24_sdwa %0, sext(%0), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(a) : "v"(b));
    }
};
struct MovSdwa {
    static constexpr const char* name = "v_mov_b32_sdwa (sext word)";
    __device__ static void op(int& a, int b, long long&) {
        asm volatile("v_mov_b32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(a));
    }
};
struct BfeI32 {
    static constexpr const char* name = "v_bfe_i32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(a)); }
};
struct PkSubI16 {
    static constexpr const char* name = "v_pk_sub_i16";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct Bfi {
    static constexpr const char* name = "v_bfi_b32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a) : "v"(b)); }
};
struct MadI32I16 {
    static constexpr const char* name = "v_mad_i32_i16";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mad_i32_i16 %0, %0, %1, %1" : "+v"(a) : "v"(b)); }
};
struct SqrtF32 {
    static constexpr const char* name = "v_sqrt_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_sqrt_f32 %0, %0" : "+v"(a)); }
};
struct RcpF32 {
    static constexpr const char* name = "v_rcp_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a)); }
};
struct MovDppShr {
    static constexpr const char* name = "v_mov_b32_dpp row_shr:1";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); }
};
struct MovDppBcast {
    static constexpr const char* name = "v_mov_b32_dpp row_bcast:15";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a) : "v"(b)); }
};
struct MovDppWave {
    static constexpr const char* name = "v_mov_b32_dpp wave_shr:1";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); }
};
struct MaxDpp {
    static constexpr const char* name = "v_max_f32_dpp quad_perm";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_max_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); }
};
struct MulHi {
    static constexpr const char* name = "v_mul_hi_i32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct MulLo {
    static constexpr const char* name = "v_mul_lo_u32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct Mad64 {
    static constexpr const char* name = "v_mad_i64_i32";
    __device__ static void op(int& a, int b, long long& w) {
        long long carry;
        asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(w), "=s"(carry) : "v"(a), "v"(b));
    }
};
struct Mul24 {
    static constexpr const char* name = "v_mul_i32_i24";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct MulHi24 {
    static constexpr const char* name = "v_mul_hi_i32_i24";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_mul_hi_i32_i24 %0, %0, %1" : "+v"(a) : "v"(b)); }
};
struct Dot2 {
    static constexpr const char* name = "v_dot2_i32_i16";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_dot2_i32_i16 %0, %0, %1, 0" : "+v"(a) : "v"(b)); }
};
struct PkAddI16 {
    static constexpr const char* name = "v_pk_add_i16 clamp";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(a) : "v"(b)); }
};
struct PkAshr {
    static constexpr const char* name = "v_pk_ashrrev_i16";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_pk_ashrrev_i16 %0, 1, %0" : "+v"(a)); }
};
struct AddI32Clamp {
    static constexpr const char* name = "v_add_i32 clamp";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a) : "v"(b)); }
};
struct Perm {
    static constexpr const char* name = "v_perm_b32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); }
};
struct Ashr64 {
    static constexpr const char* name = "v_ashrrev_i64";
    __device__ static void op(int&, int, long long& w) { asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(w)); }
};
struct FmaF32 {
    static constexpr const char* name = "v_fma_f32";
    __device__ static void op(int& a, int b, long long&) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); }
};
struct PkFmaF32 {
    static constexpr const char* name = "v_pk_fma_f32";
    __device__ static void op(int&, int, long long& w) { asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w)); }
};
struct FmaF64 {
    static constexpr const char* name = "v_fma_f64";
    __device__ static void op(int&, int, long long& w) { asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(w)); }
};

template <class OP, class = void> struct set_vcc { static constexpr bool value = false; };
template <class OP> struct set_vcc<OP, decltype((void)OP::kSetVcc)> { static constexpr bool value = true; };

template <class OP>
__global__ __launch_bounds__(256) void rate_kernel(int* out, int seed) {
    int a[kChains];
    long long w[kChains];
    for (int c = 0; c < kChains; ++c) {
        a[c] = seed + threadIdx.x * (c + 3);
        w[c] = (long long)a[c] * 77;
    }
    const int b = seed * 31 + 7;
    if constexpr (set_vcc<OP>::value) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(a[0]), "v"(b) : "vcc");
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int c = 0; c < kChains; ++c) OP::op(a[c], b, w[c]);
    }
    int s = 0;
    for (int c = 0; c < kChains; ++c) s += a[c] + (int)w[c] + (int)(w[c] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class OP>
double run(int* out, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    rate_kernel<OP><<<blocks, 256>>>(out, 5);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) rate_kernel<OP><<<blocks, 256>>>(out, 5 + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // wave instructions per second and SIMD (1024 SIMDs): blocks * 4 waves * iters * chains * 20 launches
    const double insts = double(blocks) * 4 * kIters * kChains * 20;
    return insts / (ms * 1e-3) / 1024.0;
}

int main(int argc, char** argv) {
    const int per_simd = argc > 1 ? std::atoi(argv[1]) : 8;  // waves per SIMD: workgroups of four waves per CU
    const int blocks = 256 * per_simd;
    int* out;
    CHECK(hipMalloc(&out, size_t(blocks) * 256 * sizeof(int)));
    std::printf("%d waves per SIMD\n", per_simd);
#define ROW(OP)                                                                                         \
    {                                                                                                   \
        const double r = run<OP>(out, blocks);                                                          \
        std::printf("%-22s %8.1f M wave-inst/s/SIMD  %5.2f cycles at 2.4 GHz\n", OP::name, r / 1e6, 2400e6 / r); \
    }
    ROW(AddU32) ROW(AddU32) ROW(AddU32E64) ROW(AddU32Sgpr) ROW(XorB32) ROW(AshrI32) ROW(MovB32) ROW(AndB32) ROW(LshlB32) ROW(SubU32) ROW(LshlAdd) ROW(Add3) ROW(AddCo) ROW(CvtF32I32)
    ROW(Cndmask) ROW(CndmaskSet) ROW(CndmaskE64Vcc) ROW(CndmaskSgpr) ROW(CmpCndVcc) ROW(CmpCndSgpr) ROW(CmpCnd2Vcc) ROW(CmpCnd4Vcc) ROW(CmpCnd2Sgpr) ROW(CmpCnd2E64Vcc) ROW(CmpCnd2Spaced) ROW(AddcVcc) ROW(CmpGt) ROW(CmpSgpr) ROW(MinF32) ROW(MaxI32) ROW(Max3F32)
    ROW(SubF32) ROW(AddF32Diff) ROW(MulF32Sgpr) ROW(MulF32Const) ROW(MulF32Lit)
    ROW(AddF32) ROW(MulF32) ROW(FmacF32) ROW(MaxF32) ROW(FmaF32) ROW(PkAddF32) ROW(PkMulF32) ROW(PkFmaF32) ROW(FmaF64)
    ROW(SubI32) ROW(AddI32Clamp) ROW(MulHi) ROW(MulLo) ROW(Mad64) ROW(Mul24) ROW(MulHi24) ROW(Mad32_24) ROW(Dot2)
    ROW(SqrtF32) ROW(RcpF32) ROW(MovDppShr) ROW(MovDppBcast) ROW(MovDppWave) ROW(MaxDpp) ROW(Mul24Sdwa) ROW(MovSdwa) ROW(BfeI32) ROW(MadI32I16) ROW(PkSubI16) ROW(Bfi) ROW(PkAddI16) ROW(PkAshr) ROW(Perm) ROW(Ashr64)
    CHECK(hipFree(out));
    return 0;
}
