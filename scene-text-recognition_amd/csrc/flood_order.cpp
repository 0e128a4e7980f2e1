// flood_order.cpp -- replay of the ORDER of the reference's flood on a host core.
//
// Where two or more child chains compete for a parent in non_maximum_supression (src/ER.cpp:416-462) the reference's winner is
// the child its flood entered last (children are prepended when merged, src/ER.cpp:183-185).  That order is the visiting order
// of a depth-first walk with a fixed neighbour order over the pixel grid -- an inherently sequential computation (ordered
// depth-first search is P-complete): every step depends on the marks the previous step left.  The GPU version of the same walk
// (k_flood_order, er_kernels.hip) spends ~0.8 us per pixel on one lane, a host core ~10-20 ns, and the walk is needed for roughly one
// plane in a thousand (single two-way ties whose two outcomes give different pools), so this is the one piece of the NMS that
// runs where the reference's own flood runs: on the host -- like the greedy line assignment of er_grouping (er_group.cpp).
// Nothing is built here (the component tree, the kept nodes, the pools all come from the GPU); the walk only stamps pixels.
//
// Same start pixel, same edge order (right, bottom, left, top), same LIFO buckets per level, same "priority == highest_level
// means empty" rule as src/ER.cpp:254-345: pixels at the sentinel level are marked but never popped.
#include "flood_order.h"

#include <cmath>
#include <vector>

namespace str_er {

void flood_order_host(const uint8_t *pix, int w, int h, int64_t stride, int invert, float qscale, int hi, const uint32_t *watch,
                      uint32_t n_watch, uint32_t *stamp)
{
    const uint32_t n = (uint32_t)w * (uint32_t)h;
    if (n == 0) return;
    // quantised levels (src/ER.cpp:250: 8U -> 8U convertTo with scale 1/step = round-half-even of float(p) * float(1/step))
    uint16_t lut[256];
    for (int v = 0; v < 256; ++v) lut[v] = (uint16_t)std::lrintf((float)v * qscale);
    std::vector<uint16_t> lv(n);
    for (int y = 0; y < h; ++y) {
        const uint8_t *row = pix + (size_t)y * stride;
        uint16_t      *o = lv.data() + (size_t)y * w;
        for (int x = 0; x < w; ++x) o[x] = lut[row[x] ^ invert];
    }
    constexpr uint32_t WATCH = 0x80000000u;
    uint32_t remaining = 0xFFFFFFFFu;
    if (n_watch != 0xFFFFFFFFu) {
        remaining = 0;
        for (uint32_t i = 0; i < n_watch; ++i)
            if (watch[i] < n && stamp[watch[i]] != WATCH) { stamp[watch[i]] = WATCH; ++remaining; }
        if (remaining == 0) return;
    }
    std::vector<std::vector<uint32_t>> bucket((size_t)hi + 1);      // entries: pixel << 3 | next edge
    uint32_t priority = (uint32_t)hi, counter = 0;
    uint32_t cur = 0, edge = 0, cl = lv[0];
    auto mark = [&](uint32_t p) {
        if (stamp[p] == WATCH) --remaining;
        stamp[p] = ++counter;
    };
    mark(0);
    while (remaining != 0) {
        const uint32_t x = cur % (uint32_t)w;
        bool descended = false;
        for (; edge < 4; ++edge) {
            uint32_t q;
            switch (edge) {
            case 0: q = (x + 1 < (uint32_t)w) ? cur + 1 : cur; break;
            case 1: q = (cur + (uint32_t)w < n) ? cur + (uint32_t)w : cur; break;
            case 2: q = (x > 0) ? cur - 1 : cur; break;
            default: q = (cur >= (uint32_t)w) ? cur - (uint32_t)w : cur; break;
            }
            const uint32_t s = stamp[q];
            if (q == cur || (s != 0 && s != WATCH)) continue;
            mark(q);
            const uint32_t l = lv[q];
            if (l >= cl) {
                if (l < (uint32_t)hi) bucket[l].push_back(q << 3);      // (the bucket of the sentinel level is never popped)
                if (l < priority) priority = l;
            } else {
                if (cl < (uint32_t)hi) bucket[cl].push_back((cur << 3) | (edge + 1));
                if (cl < priority) priority = cl;
                cur = q; cl = l; edge = 0;
                descended = true;
                break;
            }
        }
        if (descended) continue;
        if (priority == (uint32_t)hi) break;
        const uint32_t v = bucket[priority].back();
        bucket[priority].pop_back();
        cur = v >> 3; edge = v & 7u; cl = priority;
        while (priority < (uint32_t)hi && bucket[priority].empty()) ++priority;
    }
    if (n_watch != 0xFFFFFFFFu)
        for (uint32_t i = 0; i < n_watch; ++i)
            if (watch[i] < n && stamp[watch[i]] == WATCH) stamp[watch[i]] = 0;      // not reached (sealed off by sentinel pixels)
}

} // namespace str_er
