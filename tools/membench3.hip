// membench3.hip -- store-form study for the fused terrain kernel, round 3 (measurement tool, not product code).
// Emulates the kernel's structure: 256 x TH tiles (+ 2-row / 4-column halo) staged in LDS with nontemporal float4 loads, one
// lane per column marching down the rows, WORK float64 FMAs per row in 4 independent chains, 11 float32 planes written
// (48 B/pixel of algorithmic traffic).  What varies is how a row's 11 values per lane reach HBM:
//   FORM 0  direct: one global_store_dword (nt sc1) per plane and row -- 256 contiguous bytes per wave instruction (shipped form)
//   FORM 1  wave-private LDS transpose, one ROW at a time: lanes 16q..16q+15 write plane (4j + q) of the row as float4 -> one
//           global_store_dwordx4 carries 4 planes x 256 B; 3 instead of 11 store instructions per row, no workgroup barrier
//   FORM 2  wave-private LDS transpose over 4 ROWS of one plane: a dwordx4 store carries 4 rows x 256 B of one plane
//   SYNC n  s_barrier every n rows (0 = none): keeps the four waves of a workgroup on the same raster row so that their 256-B
//           segments of a 1 KiB row piece reach the memory controller together
//   OCC     workgroups per CU forced through LDS padding
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench3.hip -o tools/membench3 ; run: tools/membench3 [N]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int K = 11;
constexpr int PITCH = 264;
struct Planes { float* p[K]; };

__device__ __forceinline__ void st1(float* base, uint32_t off, float v) {
    uint64_t t;
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dword %1, %2, %0 nt sc1" : "=&s"(t) : "v"(off), "v"(v), "s"(base) : "memory");
}
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4(float* p, f4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
}

template <int TH, int FORM, int SYNC, int PADKB, int WORK, int ORDER>
__global__ __launch_bounds__(256) void pat_kernel(const float* in, Planes out, int n, int tiles_x, int tiles_y, int ntiles, int grid8) {
    __shared__ __attribute__((aligned(16))) float tile[(TH + 4) * PITCH];
    constexpr int STG = FORM == 1 ? 4 * K * 64 : (FORM == 2 ? 4 * 4 * K * 64 : 4);
    __shared__ __attribute__((aligned(16))) float stage[STG];
    __shared__ float pad[PADKB * 256 + 1];
    const int b = blockIdx.x;
    int logical;
    if (ORDER == 2) logical = b;                       // natural order: the 8 XCDs interleave along a tile row
    else logical = (b & 7) * grid8 + (b >> 3);         // XCD bands (shipped)
    if (logical >= ntiles) return;
    int ty, tx;
    if (ORDER == 1) { tx = logical / tiles_y; ty = logical - tx * tiles_y; }
    else { ty = logical / tiles_x; tx = logical - ty * tiles_x; }
    const int64_t x0 = (int64_t)tx * 256, y0 = (int64_t)ty * TH;
    const int tid = threadIdx.x;
    if (PADKB && tid == 1000) pad[n & 255] = 1.0f;
    typedef float v4 __attribute__((ext_vector_type(4)));
    for (int idx = tid; idx < (TH + 4) * (PITCH / 4); idx += 256) {
        const int r = idx / (PITCH / 4), v = idx - r * (PITCH / 4);
        const int64_t gy = y0 - 2 + r, gx = x0 - 4 + 4 * v;
        v4 val = {0, 0, 0, 0};
        if (gy >= 0 && gy < n && gx >= 0 && gx + 4 <= n) val = __builtin_nontemporal_load(reinterpret_cast<const v4*>(in + gy * n + gx));
        *reinterpret_cast<v4*>(&tile[r * PITCH + 4 * v]) = val;
    }
    __syncthreads();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool colok = x0 + tid < n;
    const int nrows = (n - y0) < TH ? (int)(n - y0) : TH;
    const uint64_t org_u = (uint64_t)(y0 * n + x0);
    const int64_t org = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(org_u >> 32)) << 32) |
                                  (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)org_u));
    float* base[K];
#pragma unroll
    for (int k = 0; k < K; ++k) base[k] = out.p[k] + org;
    float* wst = stage + wave * (STG / 4);   // this wave's staging area
    const int q4 = lane >> 4, c4 = lane & 15;
    const bool ok4 = x0 + wave * 64 + 4 * c4 + 3 < n;
    float* pj[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float* pl = out.p[0];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) pl = (4 * j + q4 == kk) ? out.p[kk] : pl;
        pj[j] = pl + org + wave * 64 + 4 * c4;
    }
    uint32_t off = tid * 4;
    const uint32_t rowb = (uint32_t)n * 4;
    double c0 = 1e-9, c1 = 2e-9, c2 = 3e-9, c3 = 4e-9;
    for (int r = 0; r < nrows; ++r) {
        const float* row = tile + (r + 2) * PITCH + 4 + tid;
        double z0 = row[0], z1 = row[-1], z2 = row[1], z3 = row[-PITCH];
#pragma unroll 4
        for (int w = 0; w < WORK / 4; ++w) {
            z0 = __builtin_fma(z0, 1.0000001, c0); z1 = __builtin_fma(z1, 1.0000001, c1);
            z2 = __builtin_fma(z2, 1.0000001, c2); z3 = __builtin_fma(z3, 1.0000001, c3);
        }
        const float zf = (float)((z0 + z1) + (z2 + z3));
        if (FORM == 0) {
            if (colok) {
#pragma unroll
                for (int k = 0; k < K; ++k) st1(base[k], off, zf + k);
            }
            off += rowb;
        } else if (FORM == 1) {
            // row r: stage [plane][64 columns] for this wave, read back as float4: lane (q = lane >> 4, c = lane & 15)
            // takes plane 4 j + q, columns 4 c .. 4 c + 3 (per-lane plane pointers pj[], advanced one raster row per row)
#pragma unroll
            for (int k = 0; k < K; ++k) wst[k * 64 + lane] = zf + k;
#pragma unroll
            for (int j = 0; j < (K + 3) / 4; ++j) {
                if (4 * j + q4 < K) {
                    const f4 v = *reinterpret_cast<const f4*>(wst + (4 * j + q4) * 64 + 4 * c4);
                    if (ok4) st4(pj[j], v);
                }
                pj[j] += n;
            }
        } else {
            // 4 rows of every plane per flush: stage [row & 3][plane][64], then lane (q, c) stores row q, columns 4 c ..
#pragma unroll
            for (int k = 0; k < K; ++k) wst[((r & 3) * K + k) * 64 + lane] = zf + k;
            if ((r & 3) == 3 || r == nrows - 1) {
                const int q = lane >> 4, c = lane & 15;
                const int rr = (r & ~3) + q;
                if (rr <= r && x0 + wave * 64 + 4 * c + 3 < n) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const f4 v = *reinterpret_cast<const f4*>(wst + (q * K + k) * 64 + 4 * c);
                        st4(out.p[k] + (y0 + rr) * n + x0 + wave * 64 + 4 * c, v);
                    }
                }
            }
        }
        if (SYNC > 0 && (r % SYNC) == SYNC - 1) __builtin_amdgcn_s_barrier();
    }
}

// Streaming strips: every WAVE owns a 64-column strip and marches down BH rows on its own -- no workgroup barrier, no tile
// load phase.  Rows arrive through LDS-DMA (global_load_lds_dwordx4: no staging registers) in blocks of 16 into a per-wave
// ring of 32 rows x 72 columns (the 64 columns + 4 on either side); the block after the one being marched is in flight the
// whole time.  vmcnt is counted: the plane stores issued after a block's loads stay in flight (vmcnt(63): gfx9 VMEM completes
// in order, so "at most 63 outstanding" means the loads -- older than the last 63 stores -- have landed).
template <int BH, int WORK, int PADKB>
__global__ __launch_bounds__(256) void strip_kernel(const float* in, Planes out, int n, int strips, int bands) {
    constexpr int RP = 72;            // ring pitch (floats)
    constexpr int RING = 32;
    __shared__ __attribute__((aligned(16))) float ring[4 * RING * RP];
    __shared__ float pad[PADKB * 256 + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (PADKB && tid == 1000) pad[n & 255] = 1.0f;
    const int task = blockIdx.x * 4 + wave;          // strip-major inside a band: neighbouring strips run together
    if (task >= strips * bands) return;
    const int band = task / strips, strip = task - band * strips;
    const int64_t x0 = 64 + (int64_t)strip * 64, y0 = 32 + (int64_t)band * BH;   // (interior only: the model skips the frame)
    float* my = ring + wave * (RING * RP);
    // lane -> (row, 16-byte column quad) of a 16-row block: slot t = 64 i + lane, row = t / 18, quad = t % 18
    auto issue = [&](int64_t row0, int half) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int t = 64 * i + lane;
            const int r = t / 18, q = t - r * 18;
            if (t < 288) {
                const float* src = in + (row0 + r) * n + (x0 - 4) + 4 * q;
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(my + half * 16 * RP + 64 * i * 4), 16, 0, 0);
            }
        }
    };
    issue(y0 - 2, 0);
    issue(y0 + 14, 1);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // first block landed, second in flight
    float* base[K];
    const int64_t org = y0 * n + x0;
#pragma unroll
    for (int k = 0; k < K; ++k) base[k] = out.p[k] + org;
    uint32_t off = lane * 4;
    const uint32_t rowb = (uint32_t)n * 4;
    double c0 = 1e-9, c1 = 2e-9, c2 = 3e-9, c3 = 4e-9;
    for (int r = 0; r < BH; ++r) {
        // ring row of raster row y0 + r (block b = (r + 2) / 16 sits in half b & 1)
        const int rr = (r + 2) & (RING - 1);
        if (((r + 2) & 15) == 0) {
            // rows y0 + r .. + 15 are needed from here on: their block was issued 16 rows ago; refill the half just left
            asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            if (r + 14 < BH + 2) issue(y0 + r + 16 - 2 + 2, ((r + 2) >> 4) + 1 & 1);
        }
        const float* row = my + rr * RP + 4 + lane;
        const float* up = my + ((rr + RING - 1) & (RING - 1)) * RP + 4 + lane;
        double z0 = row[0], z1 = row[-1], z2 = row[1], z3 = up[0];
#pragma unroll 4
        for (int w = 0; w < WORK / 4; ++w) {
            z0 = __builtin_fma(z0, 1.0000001, c0); z1 = __builtin_fma(z1, 1.0000001, c1);
            z2 = __builtin_fma(z2, 1.0000001, c2); z3 = __builtin_fma(z3, 1.0000001, c3);
        }
        const float zf = (float)((z0 + z1) + (z2 + z3));
#pragma unroll
        for (int k = 0; k < K; ++k) st1(base[k], off, zf + k);
        off += rowb;
    }
}

// Ceilings of the memory system itself: linear float4 streams, grid-stride, nothing else.  MODE 0 write-only (11 planes),
// 1 read-only (1 plane, 11 passes' worth of bytes is not needed: reports its own GB/s), 2 copy 1 -> 1, 3 the kernel's mix (1 read : 11 written)
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const f4* in, Planes out, size_t n4, float* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f4 v = {1.f, 2.f, 3.f, 4.f};
        if (MODE != 0) v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (MODE == 1) { acc += v; continue; }
        const int np = MODE == 2 ? 1 : K;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (k < np) {
                f4* dst = reinterpret_cast<f4*>(out.p[k]) + i;
                if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(dst), "v"(v) : "memory");
                else *dst = v;
            }
    }
    if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
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
template <int TH, int FORM, int SYNC, int PADKB, int WORK, int ORDER> void run(const char* name) {
    const int n = g_n;
    const int tx = (n + 255) / 256, ty = (n + TH - 1) / TH, nt = tx * ty, g8 = (nt + 7) / 8;
    float t = time_ms([&] { hipLaunchKernelGGL((pat_kernel<TH, FORM, SYNC, PADKB, WORK, ORDER>), dim3(g8 * 8), dim3(256), 0, 0, g_in, g_out, n, tx, ty, nt, g8); });
    const double gb = (double)n * n * 4 * (1 + K) / 1e9;
    printf("%-34s TH=%2d form=%d sync=%d padKB=%2d work=%3d order=%d  %8.3f ms  %7.1f GB/s\n", name, TH, FORM, SYNC, PADKB, WORK, ORDER, t, gb / t * 1e3);
    fflush(stdout);
}

template <int BH, int WORK, int PADKB> void run_strip(const char* name) {
    const int n = g_n;
    const int strips = (n - 128) / 64, bands = (n - 64) / BH;
    const int tasks = strips * bands, blocks = (tasks + 3) / 4;
    float t = time_ms([&] { hipLaunchKernelGGL((strip_kernel<BH, WORK, PADKB>), dim3(blocks), dim3(256), 0, 0, g_in, g_out, n, strips, bands); });
    const double gb = (double)strips * 64 * bands * BH * 4 * (1 + K) / 1e9;
    printf("%-34s BH=%4d padKB=%2d work=%3d  %8.3f ms  %7.1f GB/s  (scaled to the full raster: %7.3f ms)\n", name, BH, PADKB, WORK, t, gb / t * 1e3,
           t * ((double)n * n) / ((double)strips * 64 * bands * BH));
    fflush(stdout);
}

int main(int argc, char** argv) {
    g_n = argc > 1 ? atoi(argv[1]) : 40000;
    const size_t px = (size_t)g_n * g_n;
    CHECK(hipMalloc(&g_in, px * 4));
    CHECK(hipMemset(g_in, 0, px * 4));
    for (int k = 0; k < K; ++k) CHECK(hipMalloc(&g_out.p[k], px * 4));
    const bool only_strips = argc > 2 && argv[2][0] == 's';
    if (!only_strips) {
        float* sink; CHECK(hipMalloc(&sink, 64));
        const size_t n4 = px / 4;
        const char* nm[4] = {"write-only 11 planes", "read-only 1 plane", "copy 1 -> 1", "read 1 : write 11"};
        const double gbs[4] = {px * 4.0 * K, px * 4.0, px * 8.0, px * 4.0 * (K + 1)};
        for (int blocks : {256 * 8, 256 * 32}) {
            float t;
#define STREAM(M, NTF) t = time_ms([&] { hipLaunchKernelGGL((stream_kernel<M, NTF>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const f4*>(g_in), g_out, n4, sink); }); \
            printf("stream %-22s nt=%d blocks=%5d  %8.3f ms  %7.1f GB/s\n", nm[M], (int)NTF, blocks, t, gbs[M] / t / 1e6);
            STREAM(0, false) STREAM(0, true) STREAM(1, false) STREAM(1, true) STREAM(2, false) STREAM(2, true) STREAM(3, false) STREAM(3, true)
        }
        fflush(stdout);
    }
    // LDS per workgroup: tile 38 KB (TH 32) / 21 KB (TH 16); pads chosen for 3 workgroups per CU (~52 KB each)
    if (only_strips) goto strips;
    // --- no math: what the store form alone streams
    run<32, 0, 0, 12, 0, 0>("direct");
    run<32, 0, 4, 12, 0, 0>("direct, sync 4");
    run<32, 0, 8, 12, 0, 0>("direct, sync 8");
    run<32, 1, 0, 1, 0, 0>("4 planes / dwordx4");
    run<32, 2, 0, 0, 0, 0>("4 rows / dwordx4 (LDS 81 KB: 1 WG/CU)");
    run<16, 0, 0, 30, 0, 0>("direct");
    run<16, 1, 0, 19, 0, 0>("4 planes / dwordx4");
    run<16, 2, 0, 0, 0, 0>("4 rows / dwordx4 (LDS 65 KB: 2 WG/CU)");
    run<32, 0, 0, 12, 0, 2>("direct, natural tile order");
    run<32, 0, 0, 12, 0, 1>("direct, column-major tiles");
    // --- with the kernel's amount of math between the stores
    run<32, 0, 0, 12, 160, 0>("direct");
    run<32, 0, 1, 12, 160, 0>("direct, sync 1");
    run<32, 0, 2, 12, 160, 0>("direct, sync 2");
    run<32, 0, 4, 12, 160, 0>("direct, sync 4");
    run<32, 0, 8, 12, 160, 0>("direct, sync 8");
    run<32, 1, 0, 1, 160, 0>("4 planes / dwordx4");
    run<32, 1, 4, 1, 160, 0>("4 planes / dwordx4, sync 4");
    run<32, 2, 0, 0, 160, 0>("4 rows / dwordx4 (1 WG/CU)");
    run<16, 0, 0, 30, 160, 0>("direct");
    run<16, 0, 4, 30, 160, 0>("direct, sync 4");
    run<16, 1, 0, 19, 160, 0>("4 planes / dwordx4");
    run<16, 2, 0, 0, 160, 0>("4 rows / dwordx4 (2 WG/CU)");
    run<32, 0, 0, 12, 160, 2>("direct, natural tile order");
    run<32, 0, 4, 12, 160, 2>("direct, natural order, sync 4");
    run<32, 0, 0, 12, 120, 0>("direct");
    run<32, 0, 4, 12, 120, 0>("direct, sync 4");
    run<32, 1, 0, 1, 120, 0>("4 planes / dwordx4");
    run<32, 0, 0, 12, 200, 0>("direct");
    run<32, 1, 0, 1, 200, 0>("4 planes / dwordx4");
strips:
    run<32, 0, 0, 12, 200, 0>("direct (reference)");
    run<32, 0, 0, 12, 160, 0>("direct (reference)");
    run<32, 0, 0, 12, 120, 0>("direct (reference)");
    // --- wave-autonomous streaming strips (LDS 36 KB per workgroup + pad: 3 workgroups per CU at pad 12)
    run_strip<512, 0, 12>("strips, LDS-DMA ring");
    run_strip<512, 120, 12>("strips, LDS-DMA ring");
    run_strip<512, 160, 12>("strips, LDS-DMA ring");
    run_strip<512, 200, 12>("strips, LDS-DMA ring");
    run_strip<2048, 160, 12>("strips, LDS-DMA ring");
    run_strip<2048, 200, 12>("strips, LDS-DMA ring");
    run_strip<128, 200, 12>("strips, LDS-DMA ring");
    run_strip<512, 200, 0>("strips, LDS-DMA ring, 4 WG/CU");
    run_strip<512, 160, 0>("strips, LDS-DMA ring, 4 WG/CU");
    run<32, 0, 0, 0, 200, 0>("direct, 4 WG/CU");
    run<32, 0, 0, 0, 160, 0>("direct, 4 WG/CU");
    return 0;
}
