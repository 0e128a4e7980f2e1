// flood_order.h -- the order in which the reference's flood (er_tree_extract, src/ER.cpp:240-374) makes the pixels of a plane
// accessible, computed on the host for the few planes whose NMS has a sibling tie that can change the pool (er_kernels.hip,
// k_nms).  See flood_order.cpp for why this one step runs on a host core.
#pragma once
#include <cstddef>
#include <cstdint>

namespace str_er {

// pix: the plane (w x h, `stride` bytes per row), `invert` 0 or 0xFF, qscale = float(1/THRESH_STEP), hi = 255/THRESH_STEP + 1.
// The stamp of a pixel = its 1-based position in the order of first access (0: not reached).
// n_watch = 0xFFFFFFFF: stamp[w*h] (zero on entry) receives the stamp of every pixel.
// otherwise: watch[0..n_watch) are the pixels whose stamps are wanted, stamp[j] (zero on entry) receives the stamp of watch[j],
// and the walk stops as soon as all of them are stamped.
// group (optional, watch mode): group[j] = id of the set watch[j] belongs to (the parent its node competes for).  Only the member
// entered LAST matters per set, so the walk stops once every set has at most one unstamped member; those keep 0xFFFFFFFF.
void flood_order_host(const uint8_t *pix, int w, int h, int64_t stride, int invert, float qscale, int hi, const uint32_t *watch,
                      uint32_t n_watch, uint32_t *stamp, const uint32_t *group = nullptr);

// The walks of a batch's tie planes run on a small process-wide pool of host threads (started at the first use, never more than
// flood_walk_threads() of them however many contexts and planes there are; the calling thread works along).  fn(k) is called once for
// every k < n; an exception inside it is caught.  Returns 0, -1 if a call ran out of memory, -2 if it failed otherwise.
int flood_walks_run(size_t n, void (*fn)(size_t k, void *arg), void *arg);
int flood_walk_threads();
// hardware threads this process can keep busy: hardware_concurrency() cut down to the container's CPU quota (cgroup cpu.max) if there is one
int flood_host_cpus();

} // namespace str_er
