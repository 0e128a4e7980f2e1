// track_kernels.h -- calc_color + er_track on the classified candidates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "er_kernels.h"
#include "ocr_kernels.h"

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

// calc_color for n boxes (src: explicit boxes on one mask plane with the colour image `col`, or records -- then the colour image is the
// Y, Cr, Cb planes of the record's pyramid level and tr is indexed by the record, not by the box).  scratch: calc_color_scratch_bytes(n).
// Records that are no box of the call keep what tr holds (the pipeline zeroes tr first).
size_t calc_color_scratch_bytes(size_t n);
void launch_calc_color(hipStream_t s, const OcrSrc &src, const ColorSrc &col, int n, TrackRec *tr, uint8_t *scratch);
// ranges[2g], ranges[2g+1] = candidate range of image g (planes g*ppg .. (g+1)*ppg-1)
void launch_group_ranges(hipStream_t s, const BatchDev &b, int planes_per_group, int n_groups, uint32_t *ranges);
// er_track per image; list = scratch of as many words as there are candidates
void launch_er_track(hipStream_t s, const CandRec *cands, TrackRec *tr, uint32_t *list, const uint32_t *ranges, int n_groups);

// ---- er_grouping, the data-parallel half -------------------------------------
// Per image g (candidate range ranges[2g..2g+1]); every array below is indexed from the image's `lo`:
//   sorted[lo + k], k < n_sorted[g]   all_er after the stable sort by center.x (ties: candidate order) and,
//                                     if inner_sup, ERFilter::inner_suppression -- candidate indices
//   pairs[pair_off[g] ...]            (i << 16 | j), i < j positions in sorted[], for which the rule of
//                                     src/ER.cpp:631-644 holds, in the order the reference's double loop meets them
struct GroupBufs {
    uint32_t *tmp_a, *tmp_b;       // scratch, one word per candidate each
    uint32_t *sorted;              // one word per candidate
    uint32_t *row_cnt;             // one word per candidate: pairs of row i, then their offset
    uint32_t *n_sorted;            // per image
    uint32_t *pair_off;            // per image + 1 (the last one is the total)
    uint32_t *pairs;               // capacity pair_cap
    uint32_t  pair_cap;
};
void launch_group_prepare(hipStream_t s, const CandRec *cands, const TrackRec *tr, const uint32_t *ranges, int n_groups, int inner_sup,
                          const GroupBufs &g);
void launch_group_pairs_count(hipStream_t s, const CandRec *cands, const TrackRec *tr, const uint32_t *ranges, int n_groups, const GroupBufs &g);
void launch_group_pairs_fill(hipStream_t s, const CandRec *cands, const TrackRec *tr, const uint32_t *ranges, int n_groups, const GroupBufs &g);

} // namespace str_er
