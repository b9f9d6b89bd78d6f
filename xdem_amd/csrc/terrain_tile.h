// terrain_tile.h -- fused terrain-attribute kernel for gfx950 (MI355X): kernel templates and launchers, instantiated per
// (DEM dtype, output dtype) pair by terrain_ff.hip / terrain_dd.hip / terrain_fd.hip / terrain_df.hip (four translation
// units so that they compile in parallel); terrain.hip holds the dtype dispatch.
//
// One pass over the DEM produces every requested attribute (slope, aspect, hillshade, the curvatures,
// TPI, TRI): 4 B read + 4 B written per attribute per pixel -- an HBM-bound job, no MFMA (no dense
// contraction anywhere).  Layout of one workgroup (256 threads = 4 wave64):
//
//   * tile = 256 columns x TH rows of output; the DEM patch with its halo (TH + 2*HALO rows, 264 columns)
//     is staged in LDS with 16-byte coalesced row-major global loads (one float4 per lane, 1 KiB per wave
//     instruction); pixels outside the raster become NaN while staging, so the stencil needs no edge code;
//   * each thread owns ONE column and marches down the tile rows (terrain_math.h: march_column) holding a
//     rotating register window of per-row partial sums: LDS reads are lane-consecutive (conflict-free),
//     every store is a 256-byte contiguous row segment per wave and plane;
//   * tiles are handed to workgroups in an XCD-aware order: hardware round-robins consecutive workgroup
//     ids over the 8 XCDs, so logical tile = (id % 8) * (n/8) + id / 8 gives every XCD one contiguous band
//     of the raster and halo rows/columns shared by neighbouring tiles hit the same 4 MiB L2.
//
// Replaces xdem/terrain/surfit.py:1197-1305 (_get_surface_attributes), xdem/terrain/window.py:926-1002
// (_get_windowed_indexes, TPI/TRI) and the post-steps of xdem/terrain/terrain.py:586-596.
#pragma once
#include <stdlib.h>

#include "common.h"
#include "terrain_math.h"

namespace xd {

constexpr int TILE_W = 256;
constexpr int XPAD = 4;
constexpr int PITCH = TILE_W + 2 * XPAD;

template <typename TIN, typename TOUT> struct TileArgs {
    const TIN* dem;
    int64_t H, W, stride, halo_top, halo_bottom;
    int tiles_x, tiles_y, ntiles, grid8;  // grid8 = padded grid / 8
    int vec_ok;
    int sync_n, order;      // options "terrain_sync" / "terrain_order"
    // frame mode (frame != 0): only the tiles OUTSIDE the tile rectangle [fr_tx0, fr_tx1) x [fr_ty0, fr_ty1) -- the streaming
    // kernel below covers that interior; ntiles then counts the frame tiles
    int frame, fr_tx0, fr_tx1, fr_ty0, fr_ty1;
    int nplanes;            // requested planes of this launch (staged stores)
    TOUT* compact[N_ATTR];  // ... their pointers in ascending attribute order
    TerrainParams P;
    Planes<TOUT> out;
};

// Output rows leave the workgroup as 1 KiB row stores: every thread parks its pixel of each plane in an LDS staging row
// ([plane][256 columns], double-buffered over the output rows), one barrier later wave w reads back planes w, w+4, ... as
// float4 (64 lanes x 16 B = the 256 columns of the tile row) and stores them with one global_store_dwordx4 each -- a
// quarter of the store instructions of the direct form and 1 KiB contiguous per instruction instead of 256 B
// (tools/membench.hip: 11.65 ms vs 12.35-13.4 ms for the 76.8 GB of the 40000^2 headline case).  One barrier per row
// suffices: a wave that passes barrier i+1 has finished its reads of buffer i&1 (they feed its stores), so buffer i&1 is
// free for row i+2.  Needs float32 planes, W % 4 == 0 and 16-byte aligned planes (else the direct sink is launched).
template <uint32_t CMASK> struct StagedSink {
    typedef float out_t;
    static constexpr int NPL = CMASK ? __builtin_popcount(CMASK) : N_ATTR;
    float* mine;        // LDS: this thread's column in staging buffer 0, plane slot 0
    float* cur;
    const float* stage; // LDS: staging base
    float* const* compact;  // kernarg: plane pointers in slot order
    const int* slot;        // kernarg: attribute -> slot (runtime masks)
    int nplanes;
    int64_t org_off;    // element offset of the tile origin in a plane (wave-uniform)
    uint32_t row_bytes; // W * 4
    bool cols_ok;       // this lane's 4 columns (flush role) lie inside the raster
    int lane, wave;
    __device__ __forceinline__ void begin_row(int i) { cur = mine + (i & 1) * (nplanes * TILE_W); }
    template <int K> __device__ __forceinline__ void put(float v) {
        const int sl = CMASK ? __builtin_popcount(CMASK & ((1u << K) - 1u)) : slot[K];
        cur[sl * TILE_W] = v;
    }
    __device__ __forceinline__ void end_row(int i) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float* buf = stage + (i & 1) * (nplanes * TILE_W) + 4 * lane;
        const uint32_t off = (uint32_t)i * row_bytes + 16u * (uint32_t)lane;
#pragma unroll
        for (int j0 = 0; j0 < NPL; j0 += 4) {
            const int j = j0 + wave;
            if (j < nplanes) {
                const float4 v = *reinterpret_cast<const float4*>(buf + j * TILE_W);
                if (cols_ok) *reinterpret_cast<float4*>(reinterpret_cast<char*>(compact[j] + org_off) + off) = v;
            }
        }
    }
};

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, typename TOUT, int TH, int STORE, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void terrain_tile_kernel(const TileArgs<TIN, TOUT> a) {
    constexpr int HALO = Halo<FIT>::v;
    constexpr int VEC = 16 / sizeof(TIN);
    constexpr int NV = PITCH / VEC;
    __shared__ __attribute__((aligned(16))) TIN tile[(TH + 2 * HALO) * PITCH];

    // XCD-aware tile order (see file header)
    const int b = blockIdx.x;
    const int logical = (a.order | a.frame) ? b : (b & 7) * a.grid8 + (b >> 3);
    if (logical >= a.ntiles) return;
    int ty = logical / a.tiles_x, tx = logical - ty * a.tiles_x;
    if (a.frame) {  // frame tiles in order: the tile rows above the interior, its left / right flanks, the tile rows below
        int f = logical;
        const int top = a.tiles_x * a.fr_ty0, midw = a.fr_tx0 + (a.tiles_x - a.fr_tx1), mid = midw * (a.fr_ty1 - a.fr_ty0);
        if (f >= top) {
            f -= top;
            if (f < mid) {
                ty = a.fr_ty0 + f / midw;
                const int j = f - (f / midw) * midw;
                tx = j < a.fr_tx0 ? j : a.fr_tx1 + (j - a.fr_tx0);
            } else {
                f -= mid;
                ty = a.fr_ty1 + f / a.tiles_x;
                tx = f - (f / a.tiles_x) * a.tiles_x;
            }
        }
    }
    const int64_t x0 = (int64_t)tx * TILE_W, y0 = (int64_t)ty * TH;
    const int n_out = (int)((a.H - y0) < TH ? (a.H - y0) : TH);
    const int rows = n_out + 2 * HALO;
    const int tid = threadIdx.x;
    const TIN nan_in = (TIN)NAN;

    typedef TIN vec_t __attribute__((ext_vector_type(VEC)));
    for (int idx = tid; idx < rows * NV; idx += 256) {
        const int r = idx / NV, v = idx - r * NV;
        const int64_t gy = y0 - HALO + r;
        const int64_t gx = x0 - XPAD + (int64_t)v * VEC;
        const bool rowok = (gy >= -a.halo_top) && (gy < a.H + a.halo_bottom);
        const TIN* src = a.dem + (gy + a.halo_top) * a.stride + gx;
        vec_t val;
#if defined(XD_NOLOAD)  // (measurement builds: what does the tile load phase cost?  synthetic pixels, no global loads)
        if (true) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) val[e] = (TIN)(1000.0f + 0.37f * (float)((gx + e) & 1023) + 0.21f * (float)(gy & 1023) + 0.01f * (float)(((gx + e) * gy) & 255));
        } else
#endif
        if (rowok && a.vec_ok && gx >= 0 && gx + VEC <= a.W) {
#if defined(XD_PLAINLOAD)  // (measurement builds)
            val = *reinterpret_cast<const vec_t*>(src);
#else
            // streaming hint: a tile's rows are used once (the rows shared with the tile row below come back from HBM anyway,
            // FETCH_SIZE = 1.125 x the raster); 1.1 % faster (15.08 vs 15.25 ms in one session)
            val = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(src));
#endif
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int64_t x = gx + e;
                val[e] = (rowok && x >= 0 && x < a.W) ? src[e] : nan_in;
            }
        }
        *reinterpret_cast<vec_t*>(&tile[r * PITCH + v * VEC]) = val;
    }
    __syncthreads();

    // wave-uniform tile origin (readfirstlane pins it into SGPRs so the plane pointers below stay scalar)
    const uint64_t org_u = (uint64_t)(y0 * a.W + x0);
    // (the builtin returns a signed int: go through uint32_t, or a low half >= 2^31 sign-extends over the high half)
    const int64_t org_off = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(org_u >> 32)) << 32) |
                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)org_u));
    if constexpr (STORE == 1) {
        typedef StagedSink<SP::CMASK> sink_t;
        __shared__ __attribute__((aligned(16))) float stage[2 * sink_t::NPL * TILE_W];
        sink_t sk;
        sk.stage = stage;
        sk.mine = stage + tid;
        sk.compact = reinterpret_cast<float* const*>(a.compact);
        sk.slot = a.P.slot;
        sk.nplanes = SP::CMASK ? sink_t::NPL : a.nplanes;
        sk.org_off = org_off;
        sk.row_bytes = (uint32_t)(a.W * sizeof(float));
        sk.lane = tid & 63;
        sk.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        sk.cols_ok = x0 + 4 * (tid & 63) < a.W;
        // every thread marches (the row barrier needs all four waves); columns beyond the raster only see NaN padding
        march_column<FIT, CURV, WIN, SP, TIN, sink_t>(tile + XPAD + tid, PITCH, n_out, a.P, sk);
    } else {
        {
            // Every thread marches (the optional row barrier needs all four waves): a thread whose column lies beyond the raster
            // takes the raster's last column instead and stores the same values to the same addresses as that column's own
            // thread -- duplicate identical stores, only in the last tile of a tile row, and no predicate anywhere.
            const int64_t last = a.W - 1 - x0;
            const int ct = (int)((int64_t)tid < last ? (int64_t)tid : last);
            DirectSink<TOUT> sk;
#pragma unroll
            for (int k = 0; k < N_ATTR; ++k) sk.org.p[k] = a.out.p[k] + org_off;
            sk.o0 = (uint32_t)(ct * sizeof(TOUT));
            sk.ostride = (uint32_t)(a.W * sizeof(TOUT));
            sk.sync_n = (uint32_t)a.sync_n;
            march_column<FIT, CURV, WIN, SP, TIN, DirectSink<TOUT>>(tile + XPAD + ct, PITCH, n_out, a.P, sk);
        }
    }
}

// ---- streaming strips (float32 rasters, the specialised attribute sets) ----------------------------------------------------
// The tile kernel above pays, per 32-row tile, a load phase the whole workgroup waits for (global loads -> LDS -> barrier:
// the three workgroups of a CU start together and stay in step, so those bubbles line up), 4 halo rows of partial sums and
// 12 % re-read input.  Here every WAVE owns a 64-column strip and marches down a band of BH rows on its own: rows arrive by
// LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) in blocks of 16 into a per-wave ring of 32 rows x
// 72 columns (the strip + 4 columns either side: 16-byte quads), the block after the one being marched always in flight;
// no workgroup barrier anywhere, halo rows only at the two ends of a band.  tools/membench3.hip (same structure, float64
// FMAs standing in for the attribute math): 14.9 -> 13.5 ms at the kernel's amount of math, 12.2 ms without any.
// The ring is filled from inside the raster only, so this kernel covers the raster's INTERIOR (whole 256-column x 32-row
// tiles whose windows and 4-column pads stay inside); the frame of edge tiles goes to the tile kernel in frame mode.
// VMEM bookkeeping: a block's loads are waited for with a COUNTED s_waitcnt -- the plane stores issued after them (inline asm,
// invisible to the compiler) stay in flight: gfx9 VMEM operations of a wave retire in order, so "at most N outstanding" with
// N <= the number of stores issued since the loads means the loads have landed.  N = min(63, 11 rows x planes per row).
struct StripArgs {
    const float* dem;
    int64_t H, W, stride, halo_top;
    int64_t xi0, yi0, yi1;      // interior: columns from xi0 (groups of 256), rows [yi0, yi1)
    int groups_x, ngroups, grid8, order;
    int nbands;             // bands of BH rows (order 3 deals the strip groups of one column band to consecutive workgroups)
    uint32_t perm_mul;      // order 2: workgroup b takes strip group (b * perm_mul) mod (grid8 * 8), perm_mul coprime to that
    int safe_wait;          // option "terrain_ring_wait" = 1: s_waitcnt vmcnt(0) instead of the counted wait (test switch)
    TerrainParams P;
    Planes<float> out;
};

constexpr int RING_ROWS = 32, RING_PITCH = 72;
#ifndef XD_RING_BLOCK   // (measurement builds: -DXD_RING_BLOCK(n)=8 or 16 for every set)
#define XD_RING_BLOCK(nplanes) ((nplanes) <= 5 ? 8 : 16)
#endif

typedef const __attribute__((address_space(3))) float* lds_cfloat_ptr;   // 32-bit LDS address (a generic pointer costs 64-bit adds)
// The ring holds RING_ROWS tile rows in RING_ROWS / BLK blocks of BLK rows; a block is refilled as soon as no path reads its rows
// any more (the oldest row a step r reads is r - 4: the Florinsky window, the cold path's reference-order sums, the TPI / TRI
// re-read) with the rows RING_ROWS further down.  BLK = 16 (rounds 3-5: two halves) issues a refill 11 rows ahead of its first
// use; BLK = 8 (round 6: four quarters) 19 rows ahead -- the 1-3-plane sets march a row in ~300 issue cycles, 11 rows are 1.7 us
// at 1.9 GHz and their loads were NOT back in time (measurement build without refills: slope alone 3.58 -> 3.11 ms, the issue
// bound; profiles/r06_small_sets_bound.txt); the eleven-plane sets take ~800 cycles per row and never waited.
template <int NPL, int BLK> struct RowsRing {
    static constexpr int NBLK = RING_ROWS / BLK;
    static constexpr int QUADS = BLK * RING_PITCH / 4;     // float4 per block: 288 (4.5 wave instructions) / 144 (2.25)
    static constexpr int LOADS = (QUADS + 63) / 64;        // wave instructions per block
    lds_cfloat_ptr mine;    // LDS: this lane's column in ring row 0
    const float* gsrc;      // global: first pixel (column x0 - 4) of tile row 0
    float* ring;            // LDS: this wave's ring
    int64_t stride;
    int nrows, lane;        // nrows: bit 30 = test switch "terrain_ring_wait" (drain every VMEM operation instead of counting: the check of the
                            // counted form) -- packed into a value that is live anyway: the headline kernel has no scalar register to spare
                            // (a spilled plane pointer would come back through v_readlane, which the inline-asm stores must not follow)
    static constexpr int SAFE_BIT = 1 << 30;
    __device__ __forceinline__ int n_rows() const { return nrows & (SAFE_BIT - 1); }
    __device__ __forceinline__ lds_cfloat_ptr ptr(int t) const { return mine + (t & (RING_ROWS - 1)) * RING_PITCH; }
    __device__ __forceinline__ void issue(int k) const {   // block k = tile rows [BLK k, BLK k + BLK) -> ring block k mod NBLK
        float* dst = ring + (k & (NBLK - 1)) * (BLK * RING_PITCH);
        // block base on the scalar unit, per-lane element offsets in 24-bit multiplies (stride < 2^24: launch_stream checks)
        const float* base = gsrc + (int64_t)(BLK * k) * stride;
        const int last = n_rows() - 1 - BLK * k;   // (rows past the band's last: duplicates of it, never read)
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int t = 64 * i + lane;
            int r = t / (RING_PITCH / 4);
            const int q = t - r * (RING_PITCH / 4);
            r = r < last ? r : last;
            if (t < QUADS)
                __builtin_amdgcn_global_load_lds(base + (__umul24((uint32_t)r, (uint32_t)stride) + 4u * (uint32_t)q),
                                                 (__attribute__((address_space(3))) void*)(dst + 64 * i * 4), 16, 0, 0);
        }
    }
    // The counted waits below are only as good as the schedule they count on.  Block b >= NBLK is issued at the march step r with
    // r mod BLK == REFILL_AT in which block b - NBLK died, and first read at step BLK b - 1 (the prefetch of tile row BLK b):
    // ROWS_BETWEEN steps later, every one of which emits one output row (r >= BLK + REFILL_AT > 2 HALO: past the band's lead-in) and
    // an output row is NPL unpredicated plane stores -- the cold paths only ADD stores -- so at least ROWS_BETWEEN * NPL VMEM
    // operations are younger than the block's loads, and gfx9 retires a wave's VMEM operations in order: "at most N outstanding" with
    // N <= ROWS_BETWEEN * NPL means the loads have landed.  (Loads of blocks issued in between are younger as well, but not every band
    // issues them -- its last blocks have no successors --, so they are not counted on.)  The blocks of the kernel's prologue (0 ..
    // NBLK - 1, all issued: a band holds at least 68 tile rows) are waited for with what is certainly younger than them: the later
    // prologue blocks' loads and the rows stored since step 4.  The constants are tied together here so that an edit of the schedule
    // fails to compile instead of reading stale ring rows; test switch "terrain_ring_wait" = 1 replaces every counted wait by
    // vmcnt(0) (GPU test: identical planes).
    static constexpr int REFILL_AT = 4;                                  // oldest row still read at step r is r - 4 (Florinsky window)
    static constexpr int ROWS_BETWEEN = BLK * (NBLK - 1) - 1 - REFILL_AT;   // 11 (BLK 16) / 19 (BLK 8) output rows between a block's issue and its first read
    static constexpr int N_WAIT = ROWS_BETWEEN * NPL < 63 ? ROWS_BETWEEN * NPL : 63;
    static constexpr int n_prologue(int b) { return (LOADS * (NBLK - 1 - b) + NPL * (BLK * b - 1 - REFILL_AT)) < 63 ? (LOADS * (NBLK - 1 - b) + NPL * (BLK * b - 1 - REFILL_AT)) : 63; }
    static_assert(BLK == 8 || BLK == 16, "two halves or four quarters");
    static_assert(RING_ROWS % BLK == 0 && NBLK >= 2 && NBLK <= 4, "ring blocks");
    static_assert(REFILL_AT >= 4 && REFILL_AT < BLK - 1, "a block is refilled only once no path reads its rows any more");
    static_assert(N_WAIT >= 1 && N_WAIT <= ROWS_BETWEEN * NPL, "the counted wait may not exceed the stores issued since the refill");
    static_assert(BLK - 1 - REFILL_AT >= 1, "prologue block 1 is read after at least one stored row");
    __device__ __forceinline__ void prologue() const {   // every block of the ring in flight, block 0 landed
        bool all = true;
#pragma unroll
        for (int k = 0; k < NBLK; ++k) {
            if (BLK * k < n_rows()) issue(k);
            else all = false;
        }
        if (all) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS * (NBLK - 1)) : "memory");   // block 0 landed, the others in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __device__ __forceinline__ void step(int r) const {
        // about to read tile row r + 1, the first of its block b
        if (((r + 1) & (BLK - 1)) == 0) {
            const int b = (r + 1) / BLK;
            if (nrows & SAFE_BIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (b >= NBLK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_WAIT) : "memory");
            else if (b == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_prologue(1)) : "memory");
            else if (b == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_prologue(NBLK > 2 ? 2 : 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_prologue(NBLK > 3 ? 3 : 1)) : "memory");
        }
        // the block that held tile rows below BLK (r / BLK) is dead from here on: refill it with the rows RING_ROWS further down
        if ((r & (BLK - 1)) == REFILL_AT && r >= BLK + REFILL_AT) {
            const int k = r / BLK - 1 + NBLK;
            if (BLK * k < n_rows()) issue(k);
        }
    }
};

template <int FIT, bool CURV, bool WIN, class SP, int BH, int MINW = 1, int RBLK = 16>
__global__ __launch_bounds__(256, MINW) void terrain_strip_kernel(const StripArgs a) {
    constexpr int HALO = Halo<FIT>::v;
    constexpr int NPL = __builtin_popcount(SP::CMASK);
    static_assert(SP::CMASK != 0 && BH % 32 == 0, "specialised attribute sets only");
    __shared__ __attribute__((aligned(16))) float ring[4 * RING_ROWS * RING_PITCH];
    const int b = blockIdx.x;
    // order 0 (default): one band of strip groups per XCD, neighbouring groups share an L2; 1: natural order; 2 / 3: measurement
    // forms that spread the groups in flight over the raster -- 2 a multiplicative permutation of the group index, 3 column-major
    // (consecutive workgroups of an XCD walk DOWN a 256-column band: their row streams lie a band height apart instead of 1 KiB)
    int logical = (a.order == 1 || a.order == 2) ? b : (b & 7) * a.grid8 + (b >> 3);
    if (a.order == 2) logical = (int)(((uint64_t)(uint32_t)b * a.perm_mul) % (uint32_t)(a.grid8 * 8));
    if (logical >= a.ngroups) return;
    int band = logical / a.groups_x, gx = logical - band * a.groups_x;
    if (a.order == 3) { gx = logical / a.nbands; band = logical - gx * a.nbands; }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t x0 = a.xi0 + ((int64_t)gx * 4 + wave) * 64, y0 = a.yi0 + (int64_t)band * BH;
    const int n_out = (int)((a.yi1 - y0) < BH ? (a.yi1 - y0) : BH);
    typedef RowsRing<NPL, RBLK> ring_t;
    ring_t rows;
    rows.ring = ring + wave * (RING_ROWS * RING_PITCH);
    rows.mine = (lds_cfloat_ptr)(rows.ring + 4 + lane);
    rows.gsrc = a.dem + (y0 - HALO + a.halo_top) * a.stride + (x0 - 4);
    rows.stride = a.stride;
    rows.nrows = (n_out + 2 * HALO) | (a.safe_wait ? ring_t::SAFE_BIT : 0);
    rows.lane = lane;
    rows.prologue();
    const uint64_t org_u = (uint64_t)(y0 * a.W + x0);
    const int64_t org_off = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(org_u >> 32)) << 32) |
                                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)org_u));
    // (plane pointers stay in scalar registers for the whole kernel -- none is spilled to VGPR lanes, which the CPU test suite
    // checks on the compiled code -- so the stores read them directly: DirectSink<float, false>.  One instantiation does spill a
    // scalar pair -- Florinsky with directional curvatures parks the saved exec mask of its cold path in a VGPR lane --: its stores
    // take the plane pointer through the scalar copy, which is safe whatever was restored: the same ISA test checks that every
    // kernel with a v_readlane has the copy in front of every store)
    constexpr bool PTR_COPY = FIT == 2 && CURV && SP::DIR == 1;
    DirectSink<float, PTR_COPY> sk;
#pragma unroll
    for (int k = 0; k < N_ATTR; ++k) sk.org.p[k] = a.out.p[k] + org_off;
    sk.o0 = (uint32_t)(lane * sizeof(float));
    sk.ostride = (uint32_t)(a.W * sizeof(float));
#if defined(XD_STRIP_SYNC)   // (measurement builds: workgroup barrier every XD_STRIP_SYNC output rows -- the four strips of a group then write
    sk.sync_n = XD_STRIP_SYNC;   // the same plane rows at about the same time; legal: the four waves march the same number of rows)
#endif
    march_rows<FIT, CURV, WIN, SP, float, DirectSink<float, PTR_COPY>, ring_t>(rows, n_out, a.P, sk);
}

// TPI / TRI / roughness for an arbitrary odd window (the reference default 3 is handled by the fused kernels above).
// Per pixel the window is summed in float64 in row-major order -- the order of the flattened footprint the reference's
// generic_filter callbacks receive (xdem/terrain/window.py:67-308) -- whatever the route:
//  * window_lds_kernel (round 4): a workgroup of 256 threads owns WIN_TW x WIN_TH output pixels and stages the
//    (WIN_TH + w - 1) x (WIN_TW + w - 1) patch of the DEM in LDS once (coalesced row loads, NaN outside the raster and its
//    halo rows); every tap is then a conflict-free LDS read (lanes = consecutive columns) instead of a global load through
//    L1 / L2 -- w^2 of them per pixel -- and only the requested indexes are computed (compile-time flags: TPI alone skips the
//    difference / fused-multiply-add / min / max chains).  Windows whose patch exceeds WIN_LDS_BYTES (w > ~90) and
//    option "terrain_window_lds" = 0 take
//  * window_generic_kernel: one thread per pixel, taps through L1 / L2 (the form of rounds 1-3; the check of the other).
constexpr int WIN_TW = 64, WIN_TH = 16;
constexpr size_t WIN_LDS_BYTES = 64 * 1024;

template <typename TIN, typename TOUT, bool DO_TPI, bool DO_TRI, bool DO_ROUGH>
__device__ __forceinline__ void window_accumulate(const TIN* win /* top-left tap */, int pitch, int w, double c, int tri_wilson,
                                                  double& sum, double& acc, TIN& mx, TIN& mn, bool& has_nan) {
    for (int dy = 0; dy < w; ++dy) {
        const TIN* row = win + dy * pitch;
        for (int dx = 0; dx < w; ++dx) {
            const TIN t = row[dx];
            const double v = (double)t;
            if (DO_TPI) sum += v;
            if (DO_ROUGH) {
                has_nan |= (t != t);
                mx = t > mx ? t : mx;
                mn = t < mn ? t : mn;
            }
            if (DO_TRI) {
                const double d = fabs(v - c);
                acc = tri_wilson ? (acc + d) : fma(d, d, acc);
            }
        }
    }
}

template <typename TIN, typename TOUT, bool DO_TPI, bool DO_TRI, bool DO_ROUGH>
__global__ __launch_bounds__(256) void window_lds_kernel(const TIN* __restrict__ dem, int64_t H, int64_t W, int64_t stride,
                                                          int64_t halo_top, int64_t halo_bottom, int w, int tri_wilson,
                                                          TOUT* __restrict__ tpi, TOUT* __restrict__ tri, TOUT* __restrict__ rough) {
    extern __shared__ __attribute__((aligned(16))) unsigned char win_smem[];
    TIN* tile = reinterpret_cast<TIN*>(win_smem);
    const int h = w / 2;
    const int pitch = WIN_TW + w - 1, prow = WIN_TH + w - 1;
    const int64_t x0 = (int64_t)blockIdx.x * WIN_TW, y0 = (int64_t)blockIdx.y * WIN_TH;
    const TIN nan_in = (TIN)NAN;
    for (int idx = threadIdx.x; idx < prow * pitch; idx += 256) {
        const int r = idx / pitch, cidx = idx - r * pitch;
        const int64_t gy = y0 - h + r, gx = x0 - h + cidx;
        const bool ok = (gy >= -halo_top) && (gy < H + halo_bottom) && gx >= 0 && gx < W;
        tile[idx] = ok ? dem[(gy + halo_top) * stride + gx] : nan_in;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
    const int64_t x = x0 + lx;
    if (x >= W) return;
    const double nn = (double)(w * w - 1);
#pragma unroll 1
    for (int ly = ly0; ly < WIN_TH; ly += 4) {
        const int64_t y = y0 + ly;
        if (y >= H) break;
        const double c = (double)tile[(ly + h) * pitch + lx + h];
        double sum = 0.0, acc = 0.0;
        TIN mx = -(TIN)INFINITY, mn = (TIN)INFINITY;
        bool has_nan = false;
        window_accumulate<TIN, TOUT, DO_TPI, DO_TRI, DO_ROUGH>(tile + ly * pitch + lx, pitch, w, c, tri_wilson, sum, acc, mx, mn, has_nan);
        const int64_t o = y * W + x;
        if (DO_TPI) tpi[o] = (TOUT)(c - (sum - c) / nn);
        if (DO_TRI) tri[o] = (TOUT)(tri_wilson ? acc / nn : sqrt(acc));
        if (DO_ROUGH) rough[o] = has_nan ? (TOUT)NAN : (TOUT)((double)mx - (double)mn);
    }
}

template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void window_generic_kernel(const TIN* dem, int64_t H, int64_t W, int64_t stride,
                                                              int64_t halo_top, int64_t halo_bottom, int w,
                                                              int tri_wilson, TOUT* tpi, TOUT* tri, TOUT* rough) {
    const int64_t x = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t y = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int h = w / 2;
    const double c = (double)dem[(y + halo_top) * stride + x];
    double sum = 0.0, acc = 0.0, mx = -INFINITY, mn = INFINITY;
    bool has_nan = false;
    for (int dy = -h; dy <= h; ++dy) {
        const int64_t yy = y + dy;
        const bool rowok = (yy >= -halo_top) && (yy < H + halo_bottom);
        for (int dx = -h; dx <= h; ++dx) {
            const int64_t xx = x + dx;
            const double v = (rowok && xx >= 0 && xx < W) ? (double)dem[(yy + halo_top) * stride + xx] : (double)NAN;
            sum += v;
            has_nan |= (v != v);
            mx = v > mx ? v : mx;
            mn = v < mn ? v : mn;
            const double d = fabs(v - c);
            acc = tri_wilson ? (acc + d) : fma(d, d, acc);
        }
    }
    const double nn = (double)(w * w - 1);
    const int64_t o = y * W + x;
    if (tpi) tpi[o] = (TOUT)(c - (sum - c) / nn);
    if (tri) tri[o] = (TOUT)(tri_wilson ? acc / nn : sqrt(acc));
    if (rough) rough[o] = has_nan ? (TOUT)NAN : (TOUT)(mx - mn);
}

// Launcher of the two forms above; returns XDEMHIP_OK or an error code
template <typename TIN, typename TOUT>
static int launch_window(xdemhip_ctx* ctx, const TerrainLaunch& L, uint32_t win) {
    const TIN* dem = static_cast<const TIN*>(L.dem);
    TOUT* tpi = (win & A_TPI) ? static_cast<TOUT*>(L.planes[P_TPI]) : nullptr;
    TOUT* tri = (win & A_TRI) ? static_cast<TOUT*>(L.planes[P_TRI]) : nullptr;
    TOUT* rough = (win & A_ROUGH) ? static_cast<TOUT*>(L.planes[P_ROUGH]) : nullptr;
    const int w = L.window_size, wilson = (int)(L.tri_method == XDEMHIP_TRI_WILSON);
    const size_t lds = (size_t)(WIN_TH + w - 1) * (size_t)(WIN_TW + w - 1) * sizeof(TIN);
    const int64_t gy_lds = (L.H + WIN_TH - 1) / WIN_TH, gy_gen = (L.H + 3) / 4;
    if (ctx->terrain_window_lds != 0 && lds <= WIN_LDS_BYTES && gy_lds <= 65535) {
        const dim3 grid((unsigned)((L.W + WIN_TW - 1) / WIN_TW), (unsigned)gy_lds);
        const int which = (tpi ? 1 : 0) | (tri ? 2 : 0) | (rough ? 4 : 0);
#define XD_WIN(T, R, O)                                                                                                       \
    do {                                                                                                                      \
        if (lds > 48 * 1024) XD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&window_lds_kernel<TIN, TOUT, T, R, O>), \
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIN_LDS_BYTES)); \
        hipLaunchKernelGGL((window_lds_kernel<TIN, TOUT, T, R, O>), grid, dim3(256), lds, ctx->stream, dem, L.H, L.W, L.row_stride, \
                           L.halo_top, L.halo_bottom, w, wilson, tpi, tri, rough);                                            \
    } while (0)
        switch (which) {
            case 1: XD_WIN(true, false, false); break;
            case 2: XD_WIN(false, true, false); break;
            case 3: XD_WIN(true, true, false); break;
            case 4: XD_WIN(false, false, true); break;
            case 5: XD_WIN(true, false, true); break;
            case 6: XD_WIN(false, true, true); break;
            default: XD_WIN(true, true, true); break;
        }
#undef XD_WIN
        XD_HIP_CHECK(ctx, hipGetLastError());
        return XDEMHIP_OK;
    }
    if (gy_gen > 65535) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "raster too tall for one launch of the window kernel");
    dim3 grid((unsigned)((L.W + 63) / 64), (unsigned)gy_gen);
    hipLaunchKernelGGL((window_generic_kernel<TIN, TOUT>), grid, dim3(256), 0, ctx->stream, dem, L.H, L.W, L.row_stride, L.halo_top,
                       L.halo_bottom, w, wilson, tpi, tri, rough);
    XD_HIP_CHECK(ctx, hipGetLastError());
    return XDEMHIP_OK;
}

static void fill_params(const TerrainLaunch& L, TerrainParams& P) {
    const double res = L.resolution;
    double c1 = 8.0, cxx = 1.0, cxy = 4.0;
    if (L.surface_fit == XDEMHIP_FIT_ZEVENBERGTHORNE) { c1 = 2.0; cxx = 1.0; cxy = 4.0; }
    if (L.surface_fit == XDEMHIP_FIT_FLORINSKY) { c1 = 420.0; cxx = 35.0; cxy = 100.0; }
    // same divider expressions as xdem/terrain/surfit.py:281-302, inverted once in double
    P.s1 = 1.0 / (c1 * res);
    P.sxx = 1.0 / (cxx * (res * res));
    P.sxy = 1.0 / (cxy * (res * res));
    const double deg = 0.017453292519943295;
    const double az = (360.0 - L.hs_az) * deg;  // np.deg2rad(360 - azimuth), surfit.py:614
    const double alt = L.hs_alt * deg;
    P.hs_sin_alt = 254.0 * sin(alt);
    P.hs_kx = 254.0 * (-cos(alt) * L.hs_z * cos(az));
    P.hs_ky = 254.0 * (cos(alt) * L.hs_z * sin(az));
    P.hs_zf2 = L.hs_z * L.hs_z;
    fill_ref_weights(L.surface_fit, res, P.wref);
    P.mask = L.attr_mask;
    P.curv_directional = L.curv_method == XDEMHIP_CURV_DIRECTIONAL;
    P.tri_wilson = L.tri_method == XDEMHIP_TRI_WILSON;
    P.degrees = L.degrees;
    P.hs_clip = L.hs_unclipped ? 0 : 1;
}

struct FrameRect { int tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0; };  // interior tile rectangle left out in frame mode (empty: all tiles)

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, typename TOUT, int TH, int STORE, int MINW = 1>
static int launch_tiles(xdemhip_ctx* ctx, const TerrainLaunch& L, uint32_t mask, unsigned dyn_lds = 0, FrameRect fr = FrameRect()) {
    TileArgs<TIN, TOUT> a;
    a.dem = static_cast<const TIN*>(L.dem);
    a.H = L.H; a.W = L.W; a.stride = L.row_stride; a.halo_top = L.halo_top; a.halo_bottom = L.halo_bottom;
    const int64_t tx = (L.W + TILE_W - 1) / TILE_W, ty = (L.H + TH - 1) / TH;
    // (a HIP dispatch carries the total work-item count in 32 bits: 2^24 tiles x 256 threads is the most one launch holds --
    // 1.4e11 pixels, beyond any device memory)
    if (tx * ty > (int64_t)0xfffff0) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "raster too large for one launch");
    a.tiles_x = (int)tx; a.tiles_y = (int)ty; a.ntiles = (int)(tx * ty);
    a.frame = (fr.tx1 > fr.tx0 && fr.ty1 > fr.ty0) ? 1 : 0;
    a.fr_tx0 = fr.tx0; a.fr_tx1 = fr.tx1; a.fr_ty0 = fr.ty0; a.fr_ty1 = fr.ty1;
    if (a.frame) a.ntiles -= (fr.tx1 - fr.tx0) * (fr.ty1 - fr.ty0);
    if (a.ntiles == 0) return XDEMHIP_OK;
    a.grid8 = (a.ntiles + 7) / 8;
    constexpr int VEC = 16 / sizeof(TIN);
    a.vec_ok = ((reinterpret_cast<uintptr_t>(L.dem) & 15) == 0) && (L.row_stride % VEC == 0);
    a.sync_n = ctx->terrain_sync;
    a.order = ctx->terrain_order;
    fill_params(L, a.P);
    a.P.mask = mask;
    a.nplanes = 0;
    for (int k = 0; k < N_ATTR; ++k) {
        a.out.p[k] = static_cast<TOUT*>(L.planes[k]);
        a.compact[k] = nullptr;
        a.P.slot[k] = 0;
    }
    for (int k = 0; k < N_ATTR; ++k)
        if (mask & (1u << k)) {
            a.P.slot[k] = a.nplanes;
            a.compact[a.nplanes++] = static_cast<TOUT*>(L.planes[k]);
        }
    hipLaunchKernelGGL((terrain_tile_kernel<FIT, CURV, WIN, SP, TIN, TOUT, TH, STORE, MINW>), dim3(a.grid8 * 8), dim3(256), dyn_lds,
                       ctx->stream, a);
    XD_HIP_CHECK(ctx, hipGetLastError());
    return XDEMHIP_OK;
}

// Tile height and store form.  Default: direct stores (one 256-byte row segment per wave, plane and row) from 32-row
// (float32 DEM) / 16-row (float64 DEM) tiles.  The staged form (StagedSink: rows leave as 1 KiB float4 stores after an
// LDS transpose with one workgroup barrier per row; float32 planes, W % 4 == 0, 16-byte aligned planes) is kept behind
// context option "terrain_store" = 1: measured 5-8 % SLOWER than the direct form on MI355X (profiles/README.md, r02 --
// the per-row barrier costs more than the wider stores return); "terrain_rows" picks the tile height in measurement builds.
template <typename TOUT> static bool staged_ok(const TerrainLaunch& L, uint32_t mask) {
    if (sizeof(TOUT) != 4 || (L.W & 3)) return false;
    for (int k = 0; k < N_ATTR; ++k)
        if ((mask & (1u << k)) && (reinterpret_cast<uintptr_t>(L.planes[k]) & 15)) return false;
    return true;
}

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, typename TOUT, bool ALLSHAPES = false>
static int launch_shaped(xdemhip_ctx* ctx, const TerrainLaunch& L, uint32_t mask) {
    constexpr int TH_DIRECT = (sizeof(TIN) == 4) ? 32 : 16;
#ifdef XD_EXPERIMENT   // (the staged store form measured 5-8 % slower: measurement builds only -- the product does not carry its kernels)
    if constexpr (sizeof(TOUT) == 4) {
        if ((ctx->terrain_store == 1) && staged_ok<TOUT>(L, mask)) {
            if constexpr (ALLSHAPES) {  // measurement builds: option "terrain_rows"
                if (ctx->terrain_rows == 32) return launch_tiles<FIT, CURV, WIN, SP, TIN, TOUT, 32, 1>(ctx, L, mask);
                if (ctx->terrain_rows == 24) return launch_tiles<FIT, CURV, WIN, SP, TIN, TOUT, 24, 1>(ctx, L, mask);
            }
            return launch_tiles<FIT, CURV, WIN, SP, TIN, TOUT, 16, 1>(ctx, L, mask);
        }
    }
#endif
    if constexpr (ALLSHAPES && sizeof(TIN) == 4) {
        if (ctx->terrain_rows == 16) return launch_tiles<FIT, CURV, WIN, SP, TIN, TOUT, 16, 0>(ctx, L, mask);
        if (ctx->terrain_rows == 24) return launch_tiles<FIT, CURV, WIN, SP, TIN, TOUT, 24, 0>(ctx, L, mask);
        // occupancy experiments: 132 = register cap of 4 waves / SIMD (128 VGPRs); 232 / 332 = the default kernel held to 2 / 1
        // workgroups per CU by unused dynamic LDS
        if (ctx->terrain_rows == 132) return launch_tiles<FIT, CURV, WIN, SP, TIN, TOUT, 32, 0, 4>(ctx, L, mask);
        if (ctx->terrain_rows == 232) return launch_tiles<FIT, CURV, WIN, SP, TIN, TOUT, 32, 0>(ctx, L, mask, 40 * 1024);
        if (ctx->terrain_rows == 332) return launch_tiles<FIT, CURV, WIN, SP, TIN, TOUT, 32, 0>(ctx, L, mask, 90 * 1024);
    }
    return launch_tiles<FIT, CURV, WIN, SP, TIN, TOUT, TH_DIRECT, 0>(ctx, L, mask);
}

// Streaming route for float32 in / float32 out and the specialised attribute sets: the interior by terrain_strip_kernel, the
// frame of edge tiles by the tile kernel.  Returns 1 if it took the launch, 0 if the raster does not qualify (caller falls
// back to the tile kernel for everything), < 0 on error.  Option "terrain_stream": 0 off, 1 on (default), 128 / 256 / 512 =
// band height (measurement).
template <int FIT, bool CURV, bool WIN, class SP>
static int launch_stream(xdemhip_ctx* ctx, const TerrainLaunch& L, uint32_t mask) {
    constexpr int HALO = Halo<FIT>::v;
    constexpr int TH = 32;
    if (ctx->terrain_stream == 0 || ctx->terrain_store == 1 || ctx->terrain_rows != 0) return 0;
    if ((reinterpret_cast<uintptr_t>(L.dem) & 15) || (L.row_stride & 3)) return 0;   // 16-byte quads of the LDS-DMA
    if (L.row_stride >= ((int64_t)1 << 24) - 64) return 0;                               // 24-bit row offsets inside a block
    FrameRect fr;
    fr.tx0 = 1;
    fr.tx1 = (int)((L.W - 4) / TILE_W);
    fr.ty0 = L.halo_top >= HALO ? 0 : 1;
    const int64_t ylim = (L.H + L.halo_bottom - HALO) < L.H ? (L.H + L.halo_bottom - HALO) : L.H;
    fr.ty1 = (int)(ylim / TH);
    if (fr.tx1 - fr.tx0 < 1 || fr.ty1 - fr.ty0 < 2) return 0;
    if ((int64_t)(fr.tx1 - fr.tx0) * (fr.ty1 - fr.ty0) < 64) return 0;   // small rasters: one launch of the tile kernel
    StripArgs a;
    a.dem = static_cast<const float*>(L.dem);
    a.H = L.H; a.W = L.W; a.stride = L.row_stride; a.halo_top = L.halo_top;
    a.xi0 = (int64_t)fr.tx0 * TILE_W; a.yi0 = (int64_t)fr.ty0 * TH; a.yi1 = (int64_t)fr.ty1 * TH;
    a.groups_x = fr.tx1 - fr.tx0;
    a.order = ctx->terrain_order;
    fill_params(L, a.P);
    a.P.mask = mask;
    for (int k = 0; k < N_ATTR; ++k) { a.out.p[k] = static_cast<float*>(L.planes[k]); a.P.slot[k] = 0; }
    const int bh = ctx->terrain_stream >= 64 ? ctx->terrain_stream : 128;
    const bool dbg_no_strips = ctx->terrain_stream == 3, dbg_no_frame = ctx->terrain_stream == 2;  // (debug: one of the two launches only)
    const int64_t bands = (a.yi1 - a.yi0 + bh - 1) / bh;
    if (bands * a.groups_x > (int64_t)0xfffff0) return 0;
    a.ngroups = (int)(bands * a.groups_x);
    a.grid8 = (a.ngroups + 7) / 8;
    a.nbands = (int)bands;
    a.safe_wait = ctx->terrain_ring_wait;
    {   // multiplier of order 2: near the golden section of the grid, coprime to it
        const uint32_t n = (uint32_t)a.grid8 * 8u;
        uint32_t m = (uint32_t)(0.6180339887 * (double)n) | 1u;
        auto gcd = [](uint32_t x, uint32_t y) { while (y) { const uint32_t t = x % y; x = y; y = t; } return x; };
        while (m > 1 && gcd(m, n) != 1) m += 2;
        a.perm_mul = m;
    }
    const dim3 grid(a.grid8 * 8), block(256);
    // four waves per SIMD for every streaming kernel (four workgroups of 36 KB LDS per CU): the Florinsky sets with curvatures get
    // there since round 6 (forward-accumulated stencil sums + the TPI / TRI window re-read from LDS: 164 -> 127 VGPRs, no scratch:
    // the CPU suite checks the compiled kernels); the register allocator is told so -- left alone it takes 129
    constexpr int MINW = (FIT == 2 && CURV && SP::F64TAIL != 2) ? 1 : 4;   // (the mixed tail of option terrain_math = 0 would spill two registers)
    // ring blocks: the sets of up to five planes march a row in a few hundred cycles and need their refills further ahead (RowsRing)
    constexpr int RBLK = XD_RING_BLOCK(__builtin_popcount(SP::CMASK));
    if (dbg_no_strips) {}
    else if (bh == 128) hipLaunchKernelGGL((terrain_strip_kernel<FIT, CURV, WIN, SP, 128, MINW, RBLK>), grid, block, 0, ctx->stream, a);
    else if (bh == 256) hipLaunchKernelGGL((terrain_strip_kernel<FIT, CURV, WIN, SP, 256, MINW, RBLK>), grid, block, 0, ctx->stream, a);
    else if (bh == 512) hipLaunchKernelGGL((terrain_strip_kernel<FIT, CURV, WIN, SP, 512, MINW, RBLK>), grid, block, 0, ctx->stream, a);
    else return xd_fail(ctx, XDEMHIP_EINVAL, "terrain_stream: 0, 1, 128, 256 or 512");
    XD_HIP_CHECK(ctx, hipGetLastError());
    const int rc = dbg_no_frame ? XDEMHIP_OK : launch_tiles<FIT, CURV, WIN, SP, float, float, TH, 0>(ctx, L, mask, 0, fr);
    return rc == XDEMHIP_OK ? 1 : rc;
}

template <typename TIN, typename TOUT>
static int launch_typed(xdemhip_ctx* ctx, const TerrainLaunch& L) {
    const uint32_t surf = L.attr_mask & ~A_ANY_WIN;
    const uint32_t win = L.attr_mask & A_ANY_WIN;
    const bool fuse_win = win && L.window_size == 3;
    const uint32_t mask = surf | (fuse_win ? win : 0u);
    const bool curv = (surf & A_ANY_CURV) != 0;
    int rc = XDEMHIP_OK;
    if (mask) {
        const int fit = surf ? L.surface_fit : XDEMHIP_FIT_ZEVENBERGTHORNE;  // window-only: cheapest 3x3 march
        // Compile-time specialised kernels for the headline configurations (reference defaults: geometric
        // curvatures, degrees, Riley TRI, z_factor 1): all attribute branches fold away -> one schedulable basic block.
        // (an unclipped hillshade -- the engine-boundary call -- is known to the float64 tail of the runtime-mask kernels only)
        const bool defaults_dir = L.degrees && L.tri_method == XDEMHIP_TRI_RILEY && L.hs_z == 1.0 && !L.hs_unclipped;  // ... with either curvature method
        const bool defaults = L.curv_method == XDEMHIP_CURV_GEOMETRIC && defaults_dir;
        // option "terrain_math" = 1: float64 attribute math for float32 rasters too (other dtype pairs always use it);
        // 2 (default) / 0: lean / mixed tail of the specialised float32 kernels (the runtime-mask kernels keep the mixed tail)
        constexpr bool FF = SameT<TIN, float>::v && SameT<TOUT, float>::v;
        const bool f64tail = FF && (ctx->terrain_math == 1 || L.hs_unclipped);
#ifdef XD_EXPERIMENT
        constexpr bool ALLSHAPES = FF;
#else
        constexpr bool ALLSHAPES = false;
#endif
        if constexpr (FF) {
            // float32 in / float32 out, specialised sets: lean tail (option "terrain_math" = 2, the default) or the mixed tail of
            // round 2 (0); the streaming route (interior by wave-autonomous strips + frame by tiles) where the raster qualifies
            const bool fl = defaults && mask == MASK_FULL11 && fit == XDEMHIP_FIT_FLORINSKY;
            const bool zt = defaults && mask == MASK_FULL11 && fit == XDEMHIP_FIT_ZEVENBERGTHORNE;
            const bool hn = defaults && mask == MASK_SAH_WIN && fit == XDEMHIP_FIT_HORN;
            // round 5: the same eleven planes with curv_method="directional" (the other value users pass: terrain.py:694-747) were
            // the runtime-mask tile kernel's (0.60 of the roofline); they now have the full set's route
            const bool dirc = defaults_dir && L.curv_method == XDEMHIP_CURV_DIRECTIONAL && mask == MASK_FULL11;
            const bool fld = dirc && fit == XDEMHIP_FIT_FLORINSKY, ztd = dirc && fit == XDEMHIP_FIT_ZEVENBERGTHORNE;
#define XD_SPECIALISED(LV)                                                                                                   \
    do {                                                                                                                     \
        int took = 0;                                                                                                        \
        if (fl) took = launch_stream<2, true, true, Spec<MASK_FULL11, 0, 1, 0, 1, LV>>(ctx, L, mask);                        \
        else if (zt) took = launch_stream<1, true, true, Spec<MASK_FULL11, 0, 1, 0, 1, LV>>(ctx, L, mask);                   \
        else if (hn) took = launch_stream<0, false, true, Spec<MASK_SAH_WIN, 0, 1, 0, 1, LV>>(ctx, L, mask);                 \
        else if (fld) took = launch_stream<2, true, true, Spec<MASK_FULL11, 1, 1, 0, 1, LV>>(ctx, L, mask);                  \
        else if (ztd) took = launch_stream<1, true, true, Spec<MASK_FULL11, 1, 1, 0, 1, LV>>(ctx, L, mask);                  \
        if (took != 0) return took < 0 ? took : XDEMHIP_OK;                                                                  \
        if (fl) return launch_shaped<2, true, true, Spec<MASK_FULL11, 0, 1, 0, 1, LV>, TIN, TOUT, ALLSHAPES>(ctx, L, mask);  \
        if (zt) return launch_shaped<1, true, true, Spec<MASK_FULL11, 0, 1, 0, 1, LV>, TIN, TOUT>(ctx, L, mask);             \
        if (hn) return launch_shaped<0, false, true, Spec<MASK_SAH_WIN, 0, 1, 0, 1, LV>, TIN, TOUT>(ctx, L, mask);           \
        if (fld) return launch_shaped<2, true, true, Spec<MASK_FULL11, 1, 1, 0, 1, LV>, TIN, TOUT>(ctx, L, mask);            \
        if (ztd) return launch_shaped<1, true, true, Spec<MASK_FULL11, 1, 1, 0, 1, LV>, TIN, TOUT>(ctx, L, mask);            \
    } while (0)
            if (ctx->terrain_math == 2) XD_SPECIALISED(2);
            else if (ctx->terrain_math == 0) XD_SPECIALISED(0);
#undef XD_SPECIALISED
            // Round 4: the SMALL surface sets -- DEM.slope(), slope + aspect, a hillshade, the three together -- for every fit, with
            // reference defaults (degrees, z_factor 1): compile-time masks + lean tail + the streaming route, like the full set.
            // With 8-16 bytes per pixel instead of 48 these launches are bound by instruction issue, so the folded attribute
            // branches and the float32 scale factors matter more here than for the eleven planes.
            // (win == 0: a windowed index of another window size still has its own launch below -- these macros return)
            if (win == 0 && ctx->terrain_math == 2 && L.degrees && !L.hs_unclipped && L.hs_z == 1.0 && (mask & ~(A_SLOPE | A_ASPECT | A_HILLSHADE)) == 0) {
#define XD_SMALL(F, M)                                                                                                       \
    do {                                                                                                                     \
        const int took = launch_stream<F, false, false, Spec<M, 0, 1, 0, 1, 2>>(ctx, L, mask);                              \
        if (took != 0) return took < 0 ? took : XDEMHIP_OK;                                                                  \
        return launch_shaped<F, false, false, Spec<M, 0, 1, 0, 1, 2>, TIN, TOUT>(ctx, L, mask);                             \
    } while (0)
#define XD_SMALL_FITS(M)                                                                                                     \
    do {                                                                                                                     \
        if (fit == XDEMHIP_FIT_HORN) XD_SMALL(0, M);                                                                         \
        else if (fit == XDEMHIP_FIT_ZEVENBERGTHORNE) XD_SMALL(1, M);                                                         \
        else XD_SMALL(2, M);                                                                                                 \
    } while (0)
                if (mask == A_SLOPE) XD_SMALL_FITS(A_SLOPE);
                else if (mask == (A_SLOPE | A_ASPECT)) XD_SMALL_FITS(A_SLOPE | A_ASPECT);
                else if (mask == A_HILLSHADE) XD_SMALL_FITS(A_HILLSHADE);
                else if (mask == (A_SLOPE | A_ASPECT | A_HILLSHADE)) XD_SMALL_FITS(A_SLOPE | A_ASPECT | A_HILLSHADE);
#undef XD_SMALL_FITS
#undef XD_SMALL
            }
        } else if (!f64tail) {
            if (defaults && mask == MASK_FULL11 && fit == XDEMHIP_FIT_FLORINSKY)
                return launch_shaped<2, true, true, Spec<MASK_FULL11, 0, 1, 0, 1>, TIN, TOUT, ALLSHAPES>(ctx, L, mask);
            if (defaults && mask == MASK_FULL11 && fit == XDEMHIP_FIT_ZEVENBERGTHORNE)
                return launch_shaped<1, true, true, Spec<MASK_FULL11, 0, 1, 0, 1>, TIN, TOUT>(ctx, L, mask);
            if (defaults && mask == MASK_SAH_WIN && fit == XDEMHIP_FIT_HORN)
                return launch_shaped<0, false, true, Spec<MASK_SAH_WIN, 0, 1, 0, 1>, TIN, TOUT>(ctx, L, mask);
        }
        if constexpr (FF) {
            if (f64tail && defaults && mask == MASK_FULL11 && fit == XDEMHIP_FIT_FLORINSKY)
                return launch_shaped<2, true, true, Spec<MASK_FULL11, 0, 1, 0, 1, 1>, TIN, TOUT, ALLSHAPES>(ctx, L, mask);
        }
#define XD_GO(F, C, Wn)                                                                             \
    do {                                                                                            \
        if constexpr (FF) {                                                                         \
            rc = f64tail ? launch_shaped<F, C, Wn, SpecRuntimeF64, TIN, TOUT>(ctx, L, mask)         \
                         : launch_shaped<F, C, Wn, SpecRuntime, TIN, TOUT>(ctx, L, mask);           \
        } else {                                                                                    \
            rc = launch_shaped<F, C, Wn, SpecRuntime, TIN, TOUT>(ctx, L, mask);                     \
        }                                                                                           \
    } while (0)
        if (fit == XDEMHIP_FIT_HORN) { if (fuse_win) XD_GO(0, false, true); else XD_GO(0, false, false); }
        else if (fit == XDEMHIP_FIT_ZEVENBERGTHORNE) {
            if (curv) { if (fuse_win) XD_GO(1, true, true); else XD_GO(1, true, false); }
            else { if (fuse_win) XD_GO(1, false, true); else XD_GO(1, false, false); }
        } else {
            if (curv) { if (fuse_win) XD_GO(2, true, true); else XD_GO(2, true, false); }
            else { if (fuse_win) XD_GO(2, false, true); else XD_GO(2, false, false); }
        }
#undef XD_GO
        if (rc != XDEMHIP_OK) return rc;
    }
    if (win && !fuse_win) {
        rc = launch_window<TIN, TOUT>(ctx, L, win);
        if (rc != XDEMHIP_OK) return rc;
    }
    return XDEMHIP_OK;
}

}  // namespace xd
