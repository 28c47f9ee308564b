// usc_host.h - host-side helpers shared by the translation units of libunscreen_hip.so.
#pragma once
#include <math.h>
#include <stddef.h>

#include "usc_device.h"

namespace usc {

// error plumbing (defined in usc_pixel.hip; message readable through usc_last_error())
__attribute__((visibility("hidden"))) int fail(int code, const char* fmt, ...);
__attribute__((visibility("hidden"))) int check_launch(const char* what);

// How cv::resize(INTER_LINEAR, 8-bit) treats a (src -> dst) size pair.
struct ResizeGeom { int mode; double scale_x, scale_y; };
inline ResizeGeom resize_geom(int sh, int sw, int dh, int dw) {
    ResizeGeom g;
    g.scale_x = 1. / ((double)dw / sw);
    g.scale_y = 1. / ((double)dh / sh);
    g.mode = RS_LINEAR;
    if (sh == dh && sw == dw) g.mode = RS_COPY;
    else {
        int ix = (int)lrint(g.scale_x), iy = (int)lrint(g.scale_y);
        if (fabs(g.scale_x - ix) < 2.220446049250313e-16 && fabs(g.scale_y - iy) < 2.220446049250313e-16 && ix == 2 && iy == 2)
            g.mode = RS_AREA2;        // OpenCV reroutes exact 2x bilinear decimation to the 2x2 box average
    }
    return g;
}

inline bool too_big(int H, int W) { return (long long)H * W >= (1ll << 31); }   // kernels index pixels with 32 bits

inline int grid_for(size_t work_items, int threads) {
    size_t blocks = (work_items + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;       // 16 workgroups per CU, grid-stride beyond that
    return (int)blocks;
}

}  // namespace usc
