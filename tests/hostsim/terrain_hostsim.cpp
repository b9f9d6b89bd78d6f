// Host-compiled numerics harness for xdem_amd/csrc/terrain_math.h (TEST INFRASTRUCTURE ONLY).
// Emulates what one workgroup of terrain.hip does -- stage a NaN-padded tile, march every column -- with
// the very same march_column<> template the GPU kernel instantiates, so the stencil algebra, the window
// rotation, the NaN rule and the division-free float64 formulas can be checked against the oracle on a
// machine without a GPU.  Never linked into libxdemhip.so.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../xdem_amd/csrc/terrain_math.h"
#include "../../xdem_amd/csrc/terrain_nonfinite.h"

using namespace xd;

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, typename TOUT>
static void run(const TIN* dem, int64_t H, int64_t W, int64_t halo_top, int64_t halo_bottom, int TH,
                const TerrainParams& P, const Planes<TOUT>& out) {
    constexpr int HALO = Halo<FIT>::v;
    const int XPAD = 4, TW = 256, PITCH = TW + 2 * XPAD;
    std::vector<TIN> tile((size_t)(TH + 2 * HALO) * PITCH);
    for (int64_t y0 = 0; y0 < H; y0 += TH)
        for (int64_t x0 = 0; x0 < W; x0 += TW) {
            const int n_out = (int)std::min<int64_t>(TH, H - y0);
            for (int r = 0; r < n_out + 2 * HALO; ++r)
                for (int c = 0; c < PITCH; ++c) {
                    const int64_t gy = y0 - HALO + r, gx = x0 - XPAD + c;
                    const bool ok = gy >= -halo_top && gy < H + halo_bottom && gx >= 0 && gx < W;
                    tile[(size_t)r * PITCH + c] = ok ? dem[(gy + halo_top) * W + gx] : (TIN)NAN;
                }
            for (int t = 0; t < TW && x0 + t < W; ++t)
                {
                    DirectSink<TOUT> sk;
                    for (int k = 0; k < N_ATTR; ++k) sk.org.p[k] = out.p[k] + (y0 * W + x0);
                    sk.o0 = (uint32_t)(t * sizeof(TOUT));
                    sk.ostride = (uint32_t)(W * sizeof(TOUT));
                    march_column<FIT, CURV, WIN, SP, TIN, DirectSink<TOUT>>(tile.data() + XPAD + t, PITCH, n_out, P, sk);
                }
        }
}

static int g_tail = 2;  // tail of the specialised float32 kernels: 2 lean (the library's default), 0 mixed (option "terrain_math")
extern "C" void hostsim_set_tail(int level) { g_tail = level; }

template <typename TIN, typename TOUT>
static int go(const void* dem, int64_t H, int64_t W, int64_t ht, int64_t hb, int TH, int fit, const TerrainParams& P,
              void* const* planes) {
    Planes<TOUT> out;
    for (int k = 0; k < N_ATTR; ++k) out.p[k] = static_cast<TOUT*>(planes[k]);
    const TIN* d = static_cast<const TIN*>(dem);
    const bool curv = (P.mask & A_ANY_CURV) != 0, win = (P.mask & A_ANY_WIN) != 0;
    // exercise the compile-time specialised instantiations exactly when the GPU launcher would pick them
    if (P.mask == MASK_FULL11 && !P.curv_directional && P.degrees && !P.tri_wilson && fit != 0 && P.hs_zf2 == 1.0) {
        if (g_tail == 2) {
            if (fit == 2) run<2, true, true, Spec<MASK_FULL11, 0, 1, 0, 1, 2>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
            else run<1, true, true, Spec<MASK_FULL11, 0, 1, 0, 1, 2>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
        } else {
            if (fit == 2) run<2, true, true, Spec<MASK_FULL11, 0, 1, 0, 1>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
            else run<1, true, true, Spec<MASK_FULL11, 0, 1, 0, 1>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
        }
        return 0;
    }
    if (P.mask == MASK_FULL11 && P.curv_directional && P.degrees && !P.tri_wilson && fit != 0 && P.hs_zf2 == 1.0) {
        if (g_tail == 2) {
            if (fit == 2) run<2, true, true, Spec<MASK_FULL11, 1, 1, 0, 1, 2>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
            else run<1, true, true, Spec<MASK_FULL11, 1, 1, 0, 1, 2>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
        } else {
            if (fit == 2) run<2, true, true, Spec<MASK_FULL11, 1, 1, 0, 1>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
            else run<1, true, true, Spec<MASK_FULL11, 1, 1, 0, 1>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
        }
        return 0;
    }
    if (P.mask == MASK_SAH_WIN && P.degrees && !P.tri_wilson && fit == 0 && P.hs_zf2 == 1.0) {
        if (g_tail == 2) run<0, false, true, Spec<MASK_SAH_WIN, 0, 1, 0, 1, 2>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
        else run<0, false, true, Spec<MASK_SAH_WIN, 0, 1, 0, 1>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);
        return 0;
    }
    // the small first-derivative sets of terrain_tile.h's XD_SMALL (lean tail, compile-time masks), every fit
    if (g_tail == 2 && !win && !curv && P.degrees && P.hs_zf2 == 1.0 && sizeof(TIN) == 4 && sizeof(TOUT) == 4) {
#define SMALL(M)                                                                                                   \
    do {                                                                                                           \
        if (fit == 0) run<0, false, false, Spec<M, 0, 1, 0, 1, 2>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);        \
        else if (fit == 1) run<1, false, false, Spec<M, 0, 1, 0, 1, 2>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);   \
        else run<2, false, false, Spec<M, 0, 1, 0, 1, 2>, TIN, TOUT>(d, H, W, ht, hb, TH, P, out);                 \
        return 0;                                                                                                  \
    } while (0)
        if (P.mask == A_SLOPE) SMALL(A_SLOPE);
        if (P.mask == (A_SLOPE | A_ASPECT)) SMALL(A_SLOPE | A_ASPECT);
        if (P.mask == A_HILLSHADE) SMALL(A_HILLSHADE);
        if (P.mask == (A_SLOPE | A_ASPECT | A_HILLSHADE)) SMALL(A_SLOPE | A_ASPECT | A_HILLSHADE);
#undef SMALL
    }
#define GO(F, C, Wn) run<F, C, Wn, SpecRuntime, TIN, TOUT>(d, H, W, ht, hb, TH, P, out)
    if (fit == 0) { if (win) GO(0, false, true); else GO(0, false, false); }
    else if (fit == 1) { if (curv) { if (win) GO(1, true, true); else GO(1, true, false); } else { if (win) GO(1, false, true); else GO(1, false, false); } }
    else { if (curv) { if (win) GO(2, true, true); else GO(2, true, false); } else { if (win) GO(2, false, true); else GO(2, false, false); } }
#undef GO
    return 0;
}

extern "C" int hostsim_terrain(const void* dem, int dem_dtype, int64_t H, int64_t W, int64_t halo_top,
                               int64_t halo_bottom, int tile_rows, double resolution, int fit, int curv_dir,
                               uint32_t mask, int tri_wilson, double hs_alt, double hs_az, double hs_z, int degrees,
                               int out_dtype, void* const* planes12) {
    TerrainParams P;
    double c1 = 8, cxx = 1, cxy = 4;
    if (fit == 1) { c1 = 2; }
    if (fit == 2) { c1 = 420; cxx = 35; cxy = 100; }
    P.s1 = 1.0 / (c1 * resolution);
    P.sxx = 1.0 / (cxx * (resolution * resolution));
    P.sxy = 1.0 / (cxy * (resolution * resolution));
    const double deg = 0.017453292519943295;
    const double az = (360.0 - hs_az) * deg, alt = hs_alt * deg;
    P.hs_sin_alt = 254.0 * sin(alt);  // (the factor 254 of the hillshade is folded into the sun coefficients, as in terrain.hip)
    P.hs_kx = 254.0 * (-cos(alt) * hs_z * cos(az));
    P.hs_ky = 254.0 * (cos(alt) * hs_z * sin(az));
    P.hs_zf2 = hs_z * hs_z;
    fill_ref_weights(fit, resolution, P.wref);
    P.mask = mask; P.curv_directional = curv_dir; P.tri_wilson = tri_wilson; P.degrees = degrees;
    if (dem_dtype == 0 && out_dtype == 0) return go<float, float>(dem, H, W, halo_top, halo_bottom, tile_rows, fit, P, planes12);
    if (dem_dtype == 1 && out_dtype == 1) return go<double, double>(dem, H, W, halo_top, halo_bottom, tile_rows, fit, P, planes12);
    if (dem_dtype == 0 && out_dtype == 1) return go<float, double>(dem, H, W, halo_top, halo_bottom, tile_rows, fit, P, planes12);
    return go<double, float>(dem, H, W, halo_top, halo_bottom, tile_rows, fit, P, planes12);
}

// The Numba engine's rule for +-Inf pixels (xdem_amd/csrc/terrain_nonfinite.h) applied on top of planes hostsim_terrain has
// filled -- what terrain_nonfinite.hip's kernel does per pixel under option "terrain_nonfinite" = 1.
template <typename TIN, typename TOUT>
static void nf_go(const void* dem, int64_t H, int64_t W, int64_t ht, int64_t hb, const NfParams& P, void* const* planes) {
    NfPlanes<TOUT> out;
    for (int k = 0; k < 10; ++k) out.p[k] = static_cast<TOUT*>(planes[k]);
    for (int64_t r = 0; r < H; ++r)
        for (int64_t c = 0; c < W; ++c) nf_pixel<TIN, TOUT>(static_cast<const TIN*>(dem), r, c, H, W, W, ht, hb, P, out);
}
extern "C" int hostsim_nonfinite(const void* dem, int dem_dtype, int64_t H, int64_t W, int64_t halo_top, int64_t halo_bottom,
                                 double resolution, int fit, int curv_dir, uint32_t mask, double hs_alt, double hs_az, double hs_z,
                                 int degrees, int out_dtype, void* const* planes12) {
    NfParams P;
    nf_fill_params(P, fit, curv_dir, resolution, hs_alt, hs_az, hs_z, degrees, mask & 0x3ffu);
    if (!P.mask) return 0;
    if (dem_dtype == 0 && out_dtype == 0) nf_go<float, float>(dem, H, W, halo_top, halo_bottom, P, planes12);
    else if (dem_dtype == 1 && out_dtype == 1) nf_go<double, double>(dem, H, W, halo_top, halo_bottom, P, planes12);
    else if (dem_dtype == 0 && out_dtype == 1) nf_go<float, double>(dem, H, W, halo_top, halo_bottom, P, planes12);
    else nf_go<double, float>(dem, H, W, halo_top, halo_bottom, P, planes12);
    return 0;
}
