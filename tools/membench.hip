// membench.hip -- HBM streaming ceilings for the terrain kernel's access pattern (measurement tool, not product code).
// 1 float32 plane in, K float32 planes out, N x N pixels:
//   linear   every lane loads a float4 and stores it to the K planes (grid-stride, perfectly linear streams)
//   tiled    the terrain kernel's pattern without its math: 256-thread workgroups own 256 x TH tiles (XCD-aware order), the
//            tile is staged in LDS with 16-byte loads, then every wave stores one 256-byte row segment per plane and row
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench ; run: tools/membench [N] [K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Planes { float* p[16]; };

template <int K>
__global__ __launch_bounds__(256) void linear_kernel(const float4* in, Planes out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = in[i];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float4 w = v;
            w.x += k;
            reinterpret_cast<float4*>(out.p[k])[i] = w;
        }
    }
}

template <int K, int TH>
__global__ __launch_bounds__(256) void tiled_kernel(const float* in, Planes out, int n, int tiles_x, int ntiles, int grid8) {
    __shared__ __attribute__((aligned(16))) float tile[TH * 256];
    const int b = blockIdx.x;
    const int logical = (b & 7) * grid8 + (b >> 3);
    if (logical >= ntiles) return;
    const int ty = logical / tiles_x, tx = logical - ty * tiles_x;
    const size_t x0 = (size_t)tx * 256, y0 = (size_t)ty * TH;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < TH * 64; idx += 256) {
        const int r = idx >> 6, v = idx & 63;
        const size_t gy = y0 + r, gx = x0 + 4 * v;
        float4 val = make_float4(0, 0, 0, 0);
        if (gy < (size_t)n && gx + 4 <= (size_t)n) val = *reinterpret_cast<const float4*>(in + gy * n + gx);
        *reinterpret_cast<float4*>(&tile[r * 256 + 4 * v]) = val;
    }
    __syncthreads();
    if (x0 + tid >= (size_t)n) return;
    for (int r = 0; r < TH; ++r) {
        const size_t gy = y0 + r;
        if (gy >= (size_t)n) break;
        const float z = tile[r * 256 + tid];
#pragma unroll
        for (int k = 0; k < K; ++k) out.p[k][gy * n + x0 + tid] = z + k;
    }
}

// same tile, but rows leave as 1 KiB float4 stores: wave w stores rows w, w+4, ... (64 lanes x 16 B = one full tile row)
template <int K, int TH>
__global__ __launch_bounds__(256) void tiled4_kernel(const float* in, Planes out, int n, int tiles_x, int ntiles, int grid8) {
    __shared__ __attribute__((aligned(16))) float tile[TH * 256];
    const int b = blockIdx.x;
    const int logical = (b & 7) * grid8 + (b >> 3);
    if (logical >= ntiles) return;
    const int ty = logical / tiles_x, tx = logical - ty * tiles_x;
    const size_t x0 = (size_t)tx * 256, y0 = (size_t)ty * TH;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < TH * 64; idx += 256) {
        const int r = idx >> 6, v = idx & 63;
        const size_t gy = y0 + r, gx = x0 + 4 * v;
        float4 val = make_float4(0, 0, 0, 0);
        if (gy < (size_t)n && gx + 4 <= (size_t)n) val = *reinterpret_cast<const float4*>(in + gy * n + gx);
        *reinterpret_cast<float4*>(&tile[r * 256 + 4 * v]) = val;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < TH; r += 4) {
        const size_t gy = y0 + r, gx = x0 + 4 * lane;
        if (gy >= (size_t)n || gx + 4 > (size_t)n) continue;
        float4 z = *reinterpret_cast<float4*>(&tile[r * 256 + 4 * lane]);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float4 w = z;
            w.x += k;
            *reinterpret_cast<float4*>(out.p[k] + gy * n + gx) = w;
        }
    }
}

template <typename F> float time_ms(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

template <int K> void run(int n) {
    const size_t px = (size_t)n * n;
    float* in; CHECK(hipMalloc(&in, px * 4));
    CHECK(hipMemset(in, 0, px * 4));
    Planes out;
    for (int k = 0; k < K; ++k) CHECK(hipMalloc(&out.p[k], px * 4));
    const double gb = (double)px * 4 * (1 + K) / 1e9;
    float ms = time_ms([&] { hipLaunchKernelGGL((linear_kernel<K>), dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<const float4*>(in), out, px / 4); });
    printf("K=%2d linear            %8.3f ms  %7.1f GB/s\n", K, ms, gb / ms * 1e3);
    auto tiled = [&](auto kern, int TH, const char* name) {
        const int tx = (n + 255) / 256, ty = (n + TH - 1) / TH, nt = tx * ty, g8 = (nt + 7) / 8;
        float t = time_ms([&] { hipLaunchKernelGGL(kern, dim3(g8 * 8), dim3(256), 0, 0, in, out, n, tx, nt, g8); });
        printf("K=%2d %-17s %8.3f ms  %7.1f GB/s\n", K, name, t, gb / t * 1e3);
    };
    tiled(tiled_kernel<K, 32>, 32, "tiled 256x32");
    tiled(tiled_kernel<K, 16>, 16, "tiled 256x16");
    tiled(tiled_kernel<K, 8>, 8, "tiled 256x8");
    tiled(tiled4_kernel<K, 32>, 32, "tiled4 256x32");
    tiled(tiled4_kernel<K, 16>, 16, "tiled4 256x16");
    hipFree(in);
    for (int k = 0; k < K; ++k) hipFree(out.p[k]);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 40000;
    run<1>(n);
    run<2>(n);
    run<11>(n);
    return 0;
}
