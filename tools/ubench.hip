// ubench.hip -- VALU instruction-throughput microbenchmark for gfx950 (measurement tool, not product code).
// Every case issues the same instruction on 8 independent register sets, 4 waves per SIMD on every CU, and
// reports issue cycles per wave-instruction per SIMD (assuming the reported clock).  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL_D(NAME, ASM)                                                        \
    __global__ __launch_bounds__(256) void NAME(double* out, double seed) {        \
        double a[8], b = seed, c = seed * 0.5;                                     \
        for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                 \
        for (int it = 0; it < ITERS; ++it) {                                       \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                          \
                asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));                   \
        }                                                                          \
        double s = 0;                                                              \
        for (int i = 0; i < 8; ++i) s += a[i];                                     \
        if (s == 12345.678) out[0] = s;                                            \
    }
#define KERNEL_F(NAME, ASM)                                                        \
    __global__ __launch_bounds__(256) void NAME(double* out, double seed) {        \
        float a[8], b = (float)seed, c = (float)seed * 0.5f;                       \
        for (int i = 0; i < 8; ++i) a[i] = (float)seed + i + threadIdx.x;          \
        for (int it = 0; it < ITERS; ++it) {                                       \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                          \
                asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));                   \
        }                                                                          \
        float s = 0;                                                               \
        for (int i = 0; i < 8; ++i) s += a[i];                                     \
        if (s == 12345.678f) out[0] = s;                                           \
    }
// mixed: double accumulators with a float side register
#define KERNEL_DF(NAME, ASM)                                                       \
    __global__ __launch_bounds__(256) void NAME(double* out, double seed) {        \
        double a[8];                                                               \
        float f[8];                                                                \
        for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; f[i] = (float)a[i]; } \
        for (int it = 0; it < ITERS; ++it) {                                       \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                          \
                asm volatile(ASM : "+v"(a[i]), "+v"(f[i]));                        \
        }                                                                          \
        double s = 0;                                                              \
        for (int i = 0; i < 8; ++i) s += a[i] + f[i];                              \
        if (s == 12345.678) out[0] = s;                                            \
    }

KERNEL_D(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_D(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_D(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_D(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_D(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_D(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_D(k_mov_b64, "v_mov_b64 %0, %1")
KERNEL_D(k_cmp_f64, "v_cmp_gt_f64 vcc, %0, %1")
KERNEL_D(k_max_f64, "v_max_f64 %0, %0, %1")
KERNEL_D(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL_F(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_F(k_pk_fma_f32x, "v_fmac_f32 %0, %1, %2")
KERNEL_F(k_rsq_f32, "v_rsq_f32 %0, %0")
KERNEL_F(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_F(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL_F(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL_F(k_cndmask_2src, "v_cndmask_b32 %0, %1, %2, vcc")
KERNEL_F(k_bfi_b32, "v_bfi_b32 %0, %1, %2, %0")
KERNEL_F(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
KERNEL_F(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL_F(k_ashr_i32, "v_ashrrev_i32 %0, 31, %0")
KERNEL_F(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL_F(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
KERNEL_F(k_cmp_f32, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL_D(k_min_f64, "v_min_f64 %0, %0, %1")
KERNEL_D(k_cmp_e64_f64, "v_cmp_gt_f64_e64 s[10:11], %0, %1")
KERNEL_D(k_fmac_f64, "v_fmac_f64 %0, %1, %2")
KERNEL_DF(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
KERNEL_DF(k_cvt_f32_f64, "v_cvt_f32_f64 %1, %0")

// packed float32: the operands are 64-bit VGPR pairs (two floats per lane); KERNEL_D's double registers serve as the pairs
KERNEL_D(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_D(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL_D(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL_F(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL_F(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL_F(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL_F(k_fmaak_f32, "v_fmaak_f32 %0, %0, %1, 0x3f000000")
KERNEL_F(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")

typedef void (*kern_t)(double*, double);
struct Case { const char* name; kern_t k; };

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    const int cus = p.multiProcessorCount;
    printf("device %s CUs=%d clock=%.3f GHz\n", p.name, cus, ghz);
    double* d;
    hipMalloc(&d, 64);
    Case cases[] = {{"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64},
                    {"v_rsq_f64", k_rsq_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_sqrt_f64", k_sqrt_f64},
                    {"v_mov_b64", k_mov_b64}, {"v_cmp_gt_f64", k_cmp_f64}, {"v_max_f64", k_max_f64},
                    {"v_lshl_add_u64", k_lshl_add_u64}, {"v_fma_f32", k_fma_f32}, {"v_fmac_f32", k_pk_fma_f32x},
                    {"v_rsq_f32", k_rsq_f32}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov_b32}, {"v_cndmask_e64_sgpr", k_cndmask_e64}, {"v_cndmask_2src", k_cndmask_2src},
                    {"v_bfi_b32", k_bfi_b32}, {"v_and_or_b32", k_and_or_b32}, {"v_xor_b32", k_xor_b32}, {"v_ashrrev_i32", k_ashr_i32},
                    {"v_max_f32", k_max_f32}, {"v_med3_f32", k_med3_f32}, {"v_cmp_gt_f32", k_cmp_f32}, {"v_min_f64", k_min_f64},
                    {"v_cmp_gt_f64_e64", k_cmp_e64_f64}, {"v_fmac_f64", k_fmac_f64},
                    {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64},
                    {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32},
                    {"v_sqrt_f32", k_sqrt_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_mul_f32", k_mul_f32}, {"v_add_f32", k_add_f32},
                    {"v_fmaak_f32", k_fmaak_f32}, {"v_mov_b32_dpp", k_mov_dpp}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = cus * 4;  // 4 blocks x 4 waves per CU = 4 waves per SIMD
    for (auto& c : cases) {
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, d, 1.25);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, d, 1.25);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = 4.0 * ITERS * 8;  // 4 waves per SIMD
        const double cyc = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
        printf("%-16s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at %.2f GHz nominal)\n", c.name, ms, cyc, ghz);
    }
    return 0;
}
