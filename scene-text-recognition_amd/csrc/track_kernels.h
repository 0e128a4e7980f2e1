// track_kernels.h -- calc_color + er_track on the classified candidates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "er_kernels.h"

namespace str_er {

// What er_track leaves on an ER (inc/ER.h:57-61 color1-3, center) plus whether it ended up in all_er.
struct TrackRec {
    double   color1, color2, color3;
    int32_t  cx, cy;
    uint32_t tracked;
    uint32_t reserved;
};
static_assert(sizeof(TrackRec) == 40, "TrackRec must match str_er_track");

// The three bytes calc_color reads per pixel: planar (pipeline: Y, Cr, Cb planes of the level) or
// interleaved (cv::Mat 8UC3: step 3).
struct ColorSrc {
    const uint8_t *c0, *c1, *c2;
    int32_t        step;
    int64_t        stride;
};

// pipeline: every strong / weak candidate of the batch (cls == 0 gets a zero record)
void launch_calc_color_batch(hipStream_t s, const BatchDev &b, TrackRec *tr);
// single-stage API: n boxes on one mask plane
void launch_calc_color_boxes(hipStream_t s, const uint8_t *mask, int mstride, ColorSrc col, const int32_t *boxes, int n, TrackRec *tr);
// ranges[2g], ranges[2g+1] = candidate range of image g (planes g*ppg .. (g+1)*ppg-1)
void launch_group_ranges(hipStream_t s, const BatchDev &b, int planes_per_group, int n_groups, uint32_t *ranges);
// er_track per image; list = scratch of as many words as there are candidates
void launch_er_track(hipStream_t s, const CandRec *cands, TrackRec *tr, uint32_t *list, const uint32_t *ranges, int n_groups);

} // namespace str_er
