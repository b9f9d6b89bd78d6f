// tools/clock_probe.hip -- what clock does the shader run at WHILE a heavy kernel runs?  (measurement tool)
// One wave sleeps N x `s_sleep 127` (64 x 127 shader clocks each, ISA manual) and reads s_memrealtime (constant 100 MHz) around
// it: shader cycles / wall time = the effective shader clock, with no profiler and no SMI.  Also reads s_memtime (clock64) to see
// which clock that counter follows on gfx950.  Run alone (idle clock) and next to a float64-FMA burner / a store burner.
//   hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o tools/clock_probe && tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void probe(uint64_t* out, int iters) {
    const uint64_t t0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
    const uint64_t t1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; }
}
__global__ void burn_f64(double* out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9, d = a + 1.0;
    for (int i = 0; i < iters; ++i) { a = fma(a, b, c); d = fma(d, b, c); a = fma(a, b, d); d = fma(d, b, a); }
    if (a + d == 12345.678) out[0] = a;
}
__global__ void burn_store(float* out, size_t n, int reps) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store((float)r, out + i);
}

int main() {
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clockRate %d kHz wallClockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate, wall_khz);
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    uint64_t* d; CK(hipMalloc(&d, 64 * 16));
    double* dd; CK(hipMalloc(&dd, 8));
    const size_t nst = (size_t)4 << 30; float* st; CK(hipMalloc(&st, nst * 4));
    const int iters = 600;   // 600 x 127 x 64 = 4.9e6 shader clocks ~ 2-2.6 ms
    auto run_probes = [&](const char* label, int count) -> int {
        for (int k = 0; k < count; ++k) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, sb, d + 2 * k, iters);
        CK(hipStreamSynchronize(sb));
        std::vector<uint64_t> h(2 * count);
        CK(hipMemcpy(h.data(), d, 16 * count, hipMemcpyDeviceToHost));
        for (int k = 0; k < count; ++k) {
            const double secs = (double)h[2 * k] / (wall_khz * 1e3);
            printf("  %-28s probe %d: %.3f ms  s_sleep clock %.0f MHz (x127.5: %.0f)  clock64/wall = %.4f\n", label, k, secs * 1e3,
                   iters * 127.0 * 64.0 / secs / 1e6, iters * 127.5 * 64.0 / secs / 1e6, (double)h[2 * k + 1] / (double)h[2 * k]);
        }
        return 0;
    };
    run_probes("idle (cold)", 4);
    run_probes("idle (again)", 4);
    // float64 FMA burner: 256 CUs x 8 waves, ~60 ms
    hipLaunchKernelGGL(burn_f64, dim3(p.multiProcessorCount * 8), dim3(256), 0, sa, dd, 1 << 21);
    run_probes("next to float64 FMA burner", 8);
    CK(hipStreamSynchronize(sa));
    hipLaunchKernelGGL(burn_store, dim3(p.multiProcessorCount * 8), dim3(256), 0, sa, st, nst, 8);
    run_probes("next to store burner", 8);
    CK(hipStreamSynchronize(sa));
    hipLaunchKernelGGL(burn_f64, dim3(p.multiProcessorCount * 4), dim3(256), 0, sa, dd, 1 << 21);
    hipLaunchKernelGGL(burn_store, dim3(p.multiProcessorCount * 4), dim3(256), 0, sb, st, nst, 8);
    hipStream_t sc; CK(hipStreamCreate(&sc));
    { hipStream_t keep = sb; sb = sc; run_probes("next to both", 8); sb = keep; }
    CK(hipDeviceSynchronize());
    run_probes("idle (after)", 4);
    return 0;
}
