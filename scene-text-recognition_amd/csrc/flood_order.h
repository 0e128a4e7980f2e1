// flood_order.h -- the order in which the reference's flood (er_tree_extract, src/ER.cpp:240-374) makes the pixels of a plane
// accessible, computed on the host for the few planes whose NMS has a sibling tie that can change the pool (er_kernels.hip,
// k_nms).  See flood_order.cpp for why this one step runs on a host core.
#pragma once
#include <cstdint>

namespace str_er {

// pix: the plane (w x h, `stride` bytes per row), `invert` 0 or 0xFF, qscale = float(1/THRESH_STEP), hi = 255/THRESH_STEP + 1.
// watch[0..n_watch): pixels whose stamps are wanted -- the walk stops once all of them are stamped; n_watch = 0xFFFFFFFF: all pixels.
// stamp[w*h] must be zero on entry; on return stamp[p] = 1-based position of p in the order of first access (0: not reached).
void flood_order_host(const uint8_t *pix, int w, int h, int64_t stride, int invert, float qscale, int hi, const uint32_t *watch,
                      uint32_t n_watch, uint32_t *stamp);

} // namespace str_er
