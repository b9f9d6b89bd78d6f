// membench2.hip -- store-pattern study for the terrain kernel (measurement tool, not product code).
// 1 float32 plane in, 11 float32 planes out, N x N pixels; every variant moves the same 48 B/pixel.  Knobs:
//   TW / TH      tile width (columns per workgroup pass) / height
//   MODE 0       direct: lane = column, one 256-B row segment per wave, plane and row (the terrain kernel's direct sink)
//   MODE 1       rows as 1 KiB float4 stores, whole tile staged first (no per-row barrier) -- upper bound of row-wise stores
//   MODE 2       per-row LDS transpose with one barrier per row (the staged sink)
//   NT           nontemporal stores
//   OCC          workgroups per CU forced through LDS padding (3 = the terrain kernel's occupancy)
//   WORK         dependent float64 FMAs per pixel row in MODE 0/2 (emulates the math between the stores)
//   ORDER 0/1    tile order: XCD-aware row-major / XCD-aware column-major (consecutive tiles run down a column)
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench2.hip -o tools/membench2 ; run: tools/membench2 [N]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int K = 11;
struct Planes { float* p[K]; };

template <bool NT> __device__ __forceinline__ void st1(float* p, float v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT> __device__ __forceinline__ void st4(float* p, float4 v) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 w = {v.x, v.y, v.z, v.w};
    if (NT) __builtin_nontemporal_store(w, reinterpret_cast<v4*>(p)); else *reinterpret_cast<v4*>(p) = w;
}

template <int TH, int MODE, bool NT, int PADKB, int WORK, int ORDER>
__global__ __launch_bounds__(256) void pat_kernel(const float* in, Planes out, int n, int tiles_x, int tiles_y, int ntiles, int grid8) {
    constexpr int STAGE = (MODE == 1) ? TH * 256 * 1 : 256;
    __shared__ __attribute__((aligned(16))) float tile[TH * 256];
    __shared__ __attribute__((aligned(16))) float stage[(MODE == 2) ? 2 * K * 256 : 4];
    __shared__ float pad[PADKB * 256 + 1];
    const int b = blockIdx.x;
    int logical = (b & 7) * grid8 + (b >> 3);
    if (logical >= ntiles) return;
    int ty, tx;
    if (ORDER == 0) { ty = logical / tiles_x; tx = logical - ty * tiles_x; }
    else { tx = logical / tiles_y; ty = logical - tx * tiles_y; }
    const size_t x0 = (size_t)tx * 256, y0 = (size_t)ty * TH;
    const int tid = threadIdx.x;
    if (PADKB && tid == 1000) pad[n & 255] = 1.0f;  // keep the padding alive
    for (int idx = tid; idx < TH * 64; idx += 256) {
        const int r = idx >> 6, v = idx & 63;
        const size_t gy = y0 + r, gx = x0 + 4 * v;
        float4 val = make_float4(0, 0, 0, 0);
        if (gy < (size_t)n && gx + 4 <= (size_t)n) val = *reinterpret_cast<const float4*>(in + gy * n + gx);
        *reinterpret_cast<float4*>(&tile[r * 256 + 4 * v]) = val;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    if (MODE == 1) {
        for (int r = wave; r < TH; r += 4) {
            const size_t gy = y0 + r, gx = x0 + 4 * lane;
            if (gy >= (size_t)n || gx + 4 > (size_t)n) continue;
            float4 z = *reinterpret_cast<float4*>(&tile[r * 256 + 4 * lane]);
#pragma unroll
            for (int k = 0; k < K; ++k) { float4 w = z; w.x += k; st4<NT>(out.p[k] + gy * n + gx, w); }
        }
        return;
    }
    const bool colok = x0 + tid < (size_t)n;
    for (int r = 0; r < TH; ++r) {
        const size_t gy = y0 + r;
        if (gy >= (size_t)n) break;
        double z = tile[r * 256 + tid];
#pragma unroll 8
        for (int w = 0; w < WORK; ++w) z = __builtin_fma(z, 1.0000001, 1e-9);
        const float zf = (float)z;
        if (MODE == 0) {
            if (colok) {
#pragma unroll
                for (int k = 0; k < K; ++k) st1<NT>(out.p[k] + gy * n + x0 + tid, zf + k);
            }
        } else {
            float* cur = stage + (r & 1) * (K * 256);
#pragma unroll
            for (int k = 0; k < K; ++k) cur[k * 256 + tid] = zf + k;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int j0 = 0; j0 < K; j0 += 4) {
                const int j = j0 + wave;
                if (j < K && x0 + 4 * lane < (size_t)n) {
                    const float4 v = *reinterpret_cast<const float4*>(cur + j * 256 + 4 * lane);
                    st4<NT>(out.p[j] + gy * n + x0 + 4 * lane, v);
                }
            }
        }
    }
}

template <typename F> float time_ms(F launch, int reps = 4) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

static float* g_in; static Planes g_out; static int g_n;
template <int TH, int MODE, bool NT, int PADKB, int WORK, int ORDER> void run(const char* name) {
    const int n = g_n;
    const int tx = (n + 255) / 256, ty = (n + TH - 1) / TH, nt = tx * ty, g8 = (nt + 7) / 8;
    float t = time_ms([&] { hipLaunchKernelGGL((pat_kernel<TH, MODE, NT, PADKB, WORK, ORDER>), dim3(g8 * 8), dim3(256), 0, 0, g_in, g_out, n, tx, ty, nt, g8); });
    const double gb = (double)n * n * 4 * (1 + K) / 1e9;
    printf("%-44s TH=%2d mode=%d nt=%d padKB=%2d work=%3d order=%d  %8.3f ms  %7.1f GB/s\n", name, TH, MODE, (int)NT, PADKB, WORK, ORDER, t, gb / t * 1e3);
    fflush(stdout);
}

int main(int argc, char** argv) {
    g_n = argc > 1 ? atoi(argv[1]) : 40000;
    const size_t px = (size_t)g_n * g_n;
    CHECK(hipMalloc(&g_in, px * 4));
    CHECK(hipMemset(g_in, 0, px * 4));
    for (int k = 0; k < K; ++k) CHECK(hipMalloc(&g_out.p[k], px * 4));
    // occupancy: PADKB chosen so that tile + stage + pad ~ 52 KB -> 3 workgroups per CU
    run<32, 0, false, 0, 0, 0>("direct, full occupancy");
    run<16, 0, false, 0, 0, 0>("direct, full occupancy");
    run<32, 0, false, 20, 0, 0>("direct, 3 WG/CU");
    run<16, 0, false, 36, 0, 0>("direct, 3 WG/CU");
    run<32, 0, false, 20, 200, 0>("direct, 3 WG/CU, 200 fma/row");
    run<16, 0, false, 36, 200, 0>("direct, 3 WG/CU, 200 fma/row");
    run<32, 0, true, 20, 200, 0>("direct nt, 3 WG/CU, 200 fma/row");
    run<32, 0, false, 20, 200, 1>("direct, 3 WG/CU, 200 fma/row, column-major tiles");
    run<16, 0, false, 36, 200, 1>("direct, 3 WG/CU, 200 fma/row, column-major tiles");
    run<32, 0, true, 20, 200, 1>("direct nt, 3 WG/CU, 200 fma, col-major");
    run<32, 1, false, 0, 0, 0>("1KiB rows after staging whole tile, full occ");
    run<16, 1, false, 0, 0, 0>("1KiB rows after staging whole tile, full occ");
    run<32, 1, false, 20, 0, 0>("1KiB rows whole tile, 3 WG/CU");
    run<16, 1, false, 36, 0, 0>("1KiB rows whole tile, 3 WG/CU");
    run<16, 1, true, 36, 0, 0>("1KiB rows whole tile nt, 3 WG/CU");
    run<16, 1, false, 36, 0, 1>("1KiB rows whole tile, 3 WG/CU, col-major");
    run<16, 2, false, 14, 0, 0>("row barrier 1KiB, 3 WG/CU");
    run<16, 2, false, 14, 200, 0>("row barrier 1KiB, 3 WG/CU, 200 fma/row");
    run<16, 2, true, 14, 200, 0>("row barrier 1KiB nt, 3 WG/CU, 200 fma/row");
    run<16, 2, false, 14, 200, 1>("row barrier 1KiB, 3 WG/CU, 200 fma, col-major");
    run<32, 0, false, 20, 150, 0>("direct, 3 WG/CU, 150 fma/row");
    run<32, 0, false, 20, 250, 0>("direct, 3 WG/CU, 250 fma/row");
    return 0;
}
