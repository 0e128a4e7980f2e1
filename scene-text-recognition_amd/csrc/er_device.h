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

} // namespace str_er
