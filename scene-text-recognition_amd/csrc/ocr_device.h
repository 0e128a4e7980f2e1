// ocr_device.h -- device helpers shared by ocr_kernels.hip and track_kernels.hip: where a box of a scoring / colour call lies.
#pragma once
#include <hip/hip_runtime.h>

#include "ocr_kernels.h"

namespace str_er {

constexpr int OCR_WAVES = 4;           // boxes in flight per workgroup (a wave per box)
constexpr int OCR_BIG_PX = 4096;       // boxes above this many pixels are spread over many workgroups: a launch lasts as long as its longest wave,
constexpr int OCR_BIG_CAP = 4095;      // and one wave needs ~7 us per 1000 pixels -- the largest boxes of a batch are 50 times the average one
constexpr int OCR_BIG_PARTS = 32;      // row ranges a big box is cut into

// device memory, and said so: a pointer read out of a descriptor is a generic one to the compiler -- flat loads, which count on both wait counters
// and drag every LDS wait of the loop along
typedef const __attribute__((address_space(1))) uint8_t *GlobalBytes;

// boxes of a launch: the host's number, or -- a launch sized before the host knew it (OcrSrc::n_dev) -- the device's count, at most the number sized for
__device__ __forceinline__ int ocr_count(const OcrSrc &s, int n) { return s.n_dev ? (int)min((uint32_t)n, *s.n_dev) : n; }

struct OcrBox { GlobalBytes roi; int stride, inv, bw, bh; };

__device__ __forceinline__ OcrBox ocr_box(const OcrSrc &s, int bi)
{
    OcrBox b;
    if (s.recs) {
        const CandRec   &cd = s.recs[s.list[bi]];
        const PlaneDesc &pd = s.planes[cd.plane];
        b.bw = cd.w; b.bh = cd.h; b.stride = pd.stride; b.inv = pd.invert;
        b.roi = (GlobalBytes)(pd.pix + (size_t)cd.y * pd.stride + cd.x);
    } else {
        const int32_t *q = s.boxes + 4 * (size_t)bi;
        b.bw = q[2]; b.bh = q[3]; b.stride = s.stride; b.inv = s.inv;
        b.roi = (GlobalBytes)(s.plane + (size_t)q[1] * s.stride + q[0]);
    }
    return b;
}

} // namespace str_er
