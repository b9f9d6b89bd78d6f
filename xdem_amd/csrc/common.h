// common.h -- context object and helpers shared by the C-ABI translation units of libxdemhip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/xdemhip.h"

struct xdemhip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // stream work is enqueued on (own_stream or the caller's)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    int num_cu = 256;
    std::string err;
};

#define XD_HIP_CHECK(ctx, expr)                                                                   \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            char _b[512];                                                                         \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (ctx)->err = _b;                                                                      \
            return XDEMHIP_EHIP;                                                                  \
        }                                                                                         \
    } while (0)

inline int xd_fail(xdemhip_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// Launchers implemented in the kernel translation units.
namespace xd {
struct TerrainLaunch {
    const void* dem;     // device pointer to buffer row 0
    int dem_dtype, out_dtype;
    int64_t H, W, row_stride, halo_top, halo_bottom;
    double resolution;
    int surface_fit, curv_method, tri_method, window_size, degrees;
    uint32_t attr_mask;
    double hs_alt, hs_az, hs_z;
    void* planes[12];    // device pointers by attribute bit (null when not requested)
};
int launch_terrain(xdemhip_ctx* ctx, const TerrainLaunch& L);
}  // namespace xd
