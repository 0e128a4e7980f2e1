// Host-only check of flood_order_host's watch mode (compiled and run by tests/test_host_cpp.py): the walk that stops early must give the watched
// pixels the stamps of the complete walk, and with groups may leave out exactly the member of a set that the complete walk reaches last.
// (The complete walk itself is checked against the oracle's flood in tests/test_flood_order.py; on a GPU box the tie tests use the watch mode.)
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <set>
#include <vector>

#include "flood_order.h"

using namespace str_er;

int main()
{
    std::mt19937_64 rng(20260928);
    long checked = 0;
    for (int trial = 0; trial < 400; ++trial) {
        const int w = 1 + (int)(rng() % 97), h = 1 + (int)(rng() % 61), pad = (int)(rng() % 5), stride = w + pad;
        static const int steps[7] = {1, 2, 4, 5, 8, 9, 16};
        const int step = steps[rng() % 7], hi = 255 / step + 1, invert = (rng() & 1) ? 255 : 0;
        const float qscale = 1.0f / (float)step;
        std::vector<uint8_t> pix((size_t)stride * h);
        const int kind = (int)(rng() % 4);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < stride; ++x) {
                uint8_t v;
                if (kind == 0) v = (uint8_t)rng();                                                     // noise
                else if (kind == 1) v = (uint8_t)(((x / 7 + y / 5) & 1) ? 200 + rng() % 40 : 20 + rng() % 30);   // blocks
                else if (kind == 2) v = (uint8_t)((x * 255) / std::max(1, w - 1));                     // ramp
                else v = (uint8_t)((rng() % 16 == 0) ? rng() : 128);                                   // flat with speckles
                pix[(size_t)y * stride + x] = v;
            }
        const uint32_t n = (uint32_t)w * h;
        std::vector<uint32_t> full(n, 0);
        flood_order_host(pix.data(), w, h, stride, invert, qscale, hi, nullptr, 0xFFFFFFFFu, full.data());
        // distinct watched pixels
        const uint32_t nw = 1 + (uint32_t)(rng() % std::min<uint32_t>(n, 12));
        std::set<uint32_t> chosen;
        while (chosen.size() < nw) chosen.insert((uint32_t)(rng() % n));
        std::vector<uint32_t> watch(chosen.begin(), chosen.end());
        std::shuffle(watch.begin(), watch.end(), rng);
        // 1. no groups: every watched pixel that the complete walk reaches carries the same stamp
        {
            std::vector<uint32_t> st(nw, 0);
            flood_order_host(pix.data(), w, h, stride, invert, qscale, hi, watch.data(), nw, st.data(), nullptr);
            for (uint32_t j = 0; j < nw; ++j)
                if (st[j] != full[watch[j]]) { printf("trial %d: watch %u: %u, complete walk %u\n", trial, watch[j], st[j], full[watch[j]]); return 1; }
            checked += nw;
        }
        // 2. groups: per set at most one member without a stamp (0xFFFFFFFF), and it is the one the complete walk reaches last
        {
            std::vector<uint32_t> group(nw), st(nw, 0);
            const uint32_t ng = 1 + (uint32_t)(rng() % nw);
            for (uint32_t j = 0; j < nw; ++j) group[j] = 1000 + (uint32_t)(rng() % ng);
            flood_order_host(pix.data(), w, h, stride, invert, qscale, hi, watch.data(), nw, st.data(), group.data());
            for (uint32_t g = 1000; g < 1000 + ng; ++g) {
                uint32_t open = 0, last_stamped = 0, open_full = 0;
                for (uint32_t j = 0; j < nw; ++j) {
                    if (group[j] != g) continue;
                    if (st[j] == 0xFFFFFFFFu) { ++open; open_full = full[watch[j]]; }
                    else {
                        if (st[j] != full[watch[j]]) { printf("trial %d group %u: watch %u: %u, complete walk %u\n", trial, g, watch[j], st[j], full[watch[j]]); return 1; }
                        last_stamped = std::max(last_stamped, st[j]);
                    }
                }
                // (pixels the complete walk never reaches have stamp 0 there: a walk that ends with the stacks empty leaves such members open too)
                if (open > 1) {
                    uint32_t unreachable = 0;
                    for (uint32_t j = 0; j < nw; ++j) if (group[j] == g && st[j] == 0xFFFFFFFFu && full[watch[j]] == 0) ++unreachable;
                    if (open - unreachable > 1) { printf("trial %d group %u: %u members without a stamp\n", trial, g, open); return 1; }
                } else if (open == 1 && open_full != 0 && open_full < last_stamped) {
                    printf("trial %d group %u: the member left out (%u) is not the last one (%u)\n", trial, g, open_full, last_stamped);
                    return 1;
                }
            }
            checked += nw;
        }
    }
    printf("flood watch ok (%ld stamps)\n", checked);
    return 0;
}
