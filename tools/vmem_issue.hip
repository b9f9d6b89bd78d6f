// vmem_issue.hip -- how many cycles does one CU need per wave64 global store / load instruction of 4, 8 or 16 bytes per lane?
// (measurement tool).  Every wave hammers its own small, cache-resident window, so HBM is out of the picture: what remains is
// the per-CU address/data path.  Build: hipcc --offload-arch=gfx950 -O3 tools/vmem_issue.hip -o tools/vmem_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VEC, bool STORE, int PLANES>
__global__ __launch_bounds__(256) void k(float* buf, int iters, size_t plane_stride, float* sink) {
    // wave-private window: PLANES regions ("planes") of ROWS rows x 64 lanes x VEC floats
    constexpr int ROWS = 2;
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float* base = buf + wave * (size_t)(ROWS * 64 * VEC) + (size_t)lane * VEC;
    typedef float vt __attribute__((ext_vector_type(VEC)));
    vt acc = 0;
    for (int it = 0; it < iters; ++it) {
        const int r = it & (ROWS - 1);
#pragma unroll
        for (int p = 0; p < PLANES; ++p) {
            vt* q = reinterpret_cast<vt*>(base + (size_t)p * plane_stride + (size_t)r * 64 * VEC);
            if (STORE) { vt v = acc + (float)it; *q = v; }
            else acc += __builtin_nontemporal_load(q);
        }
    }
    if (!STORE) { float s = 0; for (int e = 0; e < VEC; ++e) s += acc[e]; if (s == 123.456f) *sink = s; }
}

template <int VEC, bool STORE, int PLANES> void run(float* buf, size_t plane_stride, float* sink, int blocks, int iters, const char* name) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<VEC, STORE, PLANES>), dim3(blocks), dim3(256), 0, 0, buf, 64, plane_stride, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k<VEC, STORE, PLANES>), dim3(blocks), dim3(256), 0, 0, buf, iters, plane_stride, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double instr_per_cu = (double)blocks * 4 * iters * PLANES / 256.0;
    const double bytes = (double)blocks * 4 * iters * PLANES * 64 * VEC * 4;
    printf("%-28s %8.3f ms  %7.1f ns per wave-instr per CU (%5.1f cycles @1.9GHz)  %8.1f GB/s through the CUs\n", name, ms,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 1.9, bytes / ms / 1e6);
}

int main() {
    const int blocks = 256 * 8;   // 8 workgroups (32 waves) per CU
    const size_t plane_stride = (size_t)blocks * 4 * 2 * 64 * 4 + 4096;
    float* buf; float* sink;
    CHECK(hipMalloc(&buf, plane_stride * 11 * sizeof(float)));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, plane_stride * 11 * sizeof(float)));
    const int it = 4000;
    run<1, true, 1>(buf, plane_stride, sink, blocks, it, "store dword, 1 plane");
    run<2, true, 1>(buf, plane_stride, sink, blocks, it, "store dwordx2, 1 plane");
    run<4, true, 1>(buf, plane_stride, sink, blocks, it, "store dwordx4, 1 plane");
    run<1, true, 11>(buf, plane_stride, sink, blocks, it / 4, "store dword, 11 planes");
    run<4, true, 11>(buf, plane_stride, sink, blocks, it / 4, "store dwordx4, 11 planes");
    run<1, false, 1>(buf, plane_stride, sink, blocks, it, "load dword, 1 plane");
    run<2, false, 1>(buf, plane_stride, sink, blocks, it, "load dwordx2, 1 plane");
    run<4, false, 1>(buf, plane_stride, sink, blocks, it, "load dwordx4, 1 plane");
    return 0;
}
