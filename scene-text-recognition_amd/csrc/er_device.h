// er_device.h -- device helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include <float.h>
#include <math.h>
#include <stdint.h>

namespace str_er {

// cv::threshold(..., THRESH_OTSU): the scan of OpenCV 4.x's getThreshVal_Otsu_8u over a 256-bin
// histogram of n pixels -- one lane, f64, the operation order of the original.
__device__ __forceinline__ int otsu_from_hist(const uint32_t *hist, double n)
{
    double       mu = 0;
    const double scale = 1. / n;
    for (int i = 0; i < 256; ++i) mu += i * (double)hist[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; ++i) {
        const double p_i = hist[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1. - q1;
        if (fmin(q1, q2) < (double)FLT_EPSILON || fmax(q1, q2) > 1. - (double)FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return (int)max_val;
}

// cv::resize INTER_LINEAR 8UC1 geometry (OpenCV 4.x semantics; the CPU restatement the tests compare with follows the same statement)
struct ResizeGeom {
    int    sw, sh, dw, dh;
    int    mode;          // 0 copy, 1 exact 2x2 area, 2 fixed-point bilinear
    double scale_x, scale_y;
};

__device__ __forceinline__ ResizeGeom resize_geom(int sw, int sh, int dw, int dh)
{
    ResizeGeom g;
    g.sw = sw; g.sh = sh; g.dw = dw; g.dh = dh;
    g.scale_x = 1.0 / ((double)dw / sw);
    g.scale_y = 1.0 / ((double)dh / sh);
    if (dw == sw && dh == sh) { g.mode = 0; return g; }
    const int isx = (int)rint(g.scale_x), isy = (int)rint(g.scale_y);
    const bool fast = fabs(g.scale_x - isx) < DBL_EPSILON && fabs(g.scale_y - isy) < DBL_EPSILON;
    g.mode = (fast && isx == 2 && isy == 2) ? 1 : 2;
    return g;
}

} // namespace str_er
