// example_text_detect.cpp -- the reference's image_mode hot path (src/utils.cpp:29-54 ->
// ERFilter::text_detect, src/ER.cpp:33-60) written against er_filter_hip.hpp.
//
//   g++ -std=c++17 -O2 example_text_detect.cpp -I../../include -L../lib -lstr_er_hip -o example_text_detect
//   ./example_text_detect strong.classifier weak.classifier frame.bgr 640 480
//
// frame.bgr is a raw interleaved 8-bit BGR dump.  Prints one line per plane and one per strong/weak ER,
// then exercises the staged calls (er_tree_extract -> non_maximum_supression -> classify) on plane 0,
// checks that they give the same pool as the fused call, and runs er_track + er_grouping on the result.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "er_filter_hip.hpp"

using namespace str_er_host;

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: %s strong.classifier weak.classifier frame.bgr width height\n", argv[0]); return 2; }
    const int w = std::atoi(argv[4]), h = std::atoi(argv[5]);
    std::vector<uint8_t> bgr((size_t)w * h * 3);
    {
        std::ifstream f(argv[3], std::ios::binary);
        if (!f.read(reinterpret_cast<char *>(bgr.data()), (std::streamsize)bgr.size())) { std::fprintf(stderr, "short read\n"); return 2; }
    }
    try {
        // src/main.cpp:22-24 with the macros of inc/utils.h:6-11
        ERFilter er_filter(8, 120, 900000, 2, 0.7, 0.15, w, h, 1);
        er_filter.set_stc(argv[1]);
        er_filter.set_wtc(argv[2]);

        Image8 src(bgr.data(), w, h, (int64_t)w * 3, 3);
        std::vector<ERTree> trees;
        ERs root;
        std::vector<ERs> all, pool, strong, weak;
        std::vector<double> times = er_filter.text_detect(src, trees, root, all, pool, strong, weak);
        for (size_t i = 0; i < root.size(); ++i)
            std::printf("plane %zu kept %zu pool %zu strong %zu weak %zu root_level %d root_area %d\n", i, trees[i].nodes.size(),
                        pool[i].size(), strong[i].size(), weak[i].size(), root[i]->level, root[i]->area);
        for (size_t i = 0; i < root.size(); ++i) {
            for (ER *e : strong[i]) std::printf("S %zu %d %d %d %d %d %u %.17g\n", i, e->bound.x, e->bound.y, e->bound.width, e->bound.height, e->area, e->key, e->score_strong);
            for (ER *e : weak[i]) std::printf("W %zu %d %d %d %d %d %u %.17g\n", i, e->bound.x, e->bound.y, e->bound.width, e->bound.height, e->area, e->key, e->score_weak);
        }
        std::printf("times extract %.6f nms %.6f classify %.6f total %.6f\n", times[0], times[1], times[2], times[6]);

        // staged calls on plane 0, like the reference's direct callers (src/utils.cpp:680-684)
        std::vector<std::vector<uint8_t>> channels;
        er_filter.compute_channels(src, channels);
        Image8 plane(channels[0].data(), w, h, w, 1);
        ERTree t0;
        ER *r0 = er_filter.er_tree_extract(plane, t0);
        ERs all0, pool0, strong0, weak0;
        er_filter.non_maximum_supression(t0, all0, pool0, plane);
        er_filter.classify(pool0, strong0, weak0, plane);
        bool same = r0 && pool0.size() == pool[0].size() && strong0.size() == strong[0].size() && weak0.size() == weak[0].size();
        for (size_t i = 0; same && i < pool0.size(); ++i) same = pool0[i]->key == pool[0][i]->key && pool0[i]->area == pool[0][i]->area;
        // the rest of text_detect (src/ER.cpp:62-69): er_track, then er_grouping(tracked, text, false, true)
        std::vector<uint8_t> ycrcb((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; ++i) { ycrcb[3 * i] = channels[0][i]; ycrcb[3 * i + 1] = channels[1][i]; ycrcb[3 * i + 2] = channels[2][i]; }
        std::vector<Image8> chan;
        for (int i = 0; i < 6; ++i) chan.emplace_back(channels[i].data(), w, h, w, 1);
        ERs tracked;
        er_filter.er_track(strong, weak, tracked, chan, Image8(ycrcb.data(), w, h, (int64_t)w * 3, 3));
        std::vector<Text> text;
        er_filter.er_grouping(tracked, text, false, true);
        std::printf("tracked %zu lines %zu\n", tracked.size(), text.size());
        for (const Text &t : text) {
            std::printf("T %.17g %d %d %d %d :", t.slope, t.box.x, t.box.y, t.box.width, t.box.height);
            for (const ER *e : t.ers) std::printf(" %d/%u", e->ch, e->key);
            std::printf("\n");
        }
        std::printf("staged == fused on plane 0: %s\n", same ? "yes" : "NO");
        return same ? 0 : 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
}
