// example_video_stream.cpp -- the reference's video_mode loop (src/utils.cpp:59-140: `cap >> frame` -> text_detect per frame)
// written against the ingest stream of include/str_er.h: frames go straight into page-locked staging buffers, several
// batches are in flight, results come back in order.
//
//   g++ -std=c++17 -O2 example_video_stream.cpp -I../../include -L../lib -lstr_er_hip -o example_video_stream
//   ./example_video_stream strong.classifier weak.classifier frames.bgr 640 480 <n_frames> <frames_per_batch>
//
// frames.bgr is a raw dump of n_frames interleaved 8-bit BGR frames.  Prints one line per frame (pooled / strong / weak /
// tracked candidates, text lines) and checks the stream against plain str_er_detect_bgr calls.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "str_er.h"

static void die(const char *what, const char *msg) { std::fprintf(stderr, "error: %s: %s\n", what, msg); std::exit(3); }

struct FrameStats { int pool = 0, strong = 0, weak = 0, tracked = 0, lines = 0; };

static void collect(const str_er_result *r, int first_frame, std::vector<FrameStats> &out)
{
    int32_t nc = 0, nt = 0, nx = 0;
    const str_er_cand  *c = str_er_result_cands(r, &nc);
    const str_er_track *t = str_er_result_tracks(r, &nt);
    const str_er_text  *x = str_er_result_texts(r, &nx);
    for (int i = 0; i < nc; ++i) {
        FrameStats &s = out[(size_t)first_frame + c[i].frame];
        ++s.pool;
        s.strong += c[i].cls == STR_ER_CLS_STRONG;
        s.weak += c[i].cls == STR_ER_CLS_WEAK;
        s.tracked += t[i].tracked != 0;
    }
    for (int i = 0; i < nx; ++i) ++out[(size_t)first_frame + x[i].frame].lines;
}

int main(int argc, char **argv)
{
    if (argc < 8) { std::fprintf(stderr, "usage: %s strong weak frames.bgr width height n_frames frames_per_batch\n", argv[0]); return 2; }
    const int w = std::atoi(argv[4]), h = std::atoi(argv[5]), n = std::atoi(argv[6]), fpb = std::atoi(argv[7]);
    const size_t fb = (size_t)w * h * 3;
    std::vector<uint8_t> video(fb * (size_t)n);
    {
        std::ifstream f(argv[3], std::ios::binary);
        if (!f.read(reinterpret_cast<char *>(video.data()), (std::streamsize)video.size())) die("read", "short file");
    }
    const uint32_t stages = STR_ER_STAGE_ALL | STR_ER_STAGE_TRACK | STR_ER_STAGE_GROUP | STR_ER_GROUP_INNER_SUP;
    str_er_params p;
    str_er_default_params(&p);
    p.max_width = w; p.max_height = h; p.max_frames = fpb;

    // ---- the stream: 3 batches in flight
    str_er_stream *st = nullptr;
    if (str_er_stream_create(&p, 3, &st) != STR_ER_OK) die("str_er_stream_create", str_er_last_error(nullptr));
    if (str_er_stream_load_cascade(st, STR_ER_CASCADE_STRONG, argv[1]) != STR_ER_OK ||
        str_er_stream_load_cascade(st, STR_ER_CASCADE_WEAK, argv[2]) != STR_ER_OK)
        die("load_cascade", str_er_stream_last_error(st));
    std::vector<FrameStats> got((size_t)n), want((size_t)n);
    std::vector<int> first_of_ticket(1, 0);
    auto drain_one = [&]() {
        str_er_result *r = nullptr;
        uint64_t ticket = 0;
        if (str_er_stream_next(st, &r, &ticket) != STR_ER_OK) die("str_er_stream_next", str_er_stream_last_error(st));
        collect(r, first_of_ticket[(size_t)ticket], got);
        str_er_result_free(r);
    };
    for (int f0 = 0; f0 < n; f0 += fpb) {
        if (str_er_stream_pending(st) == str_er_stream_depth(st)) drain_one();
        int32_t slot; uint8_t *buf; int64_t cap;
        if (str_er_stream_acquire(st, &slot, &buf, &cap) != STR_ER_OK) die("acquire", str_er_stream_last_error(st));
        const int k = std::min(fpb, n - f0);
        std::memcpy(buf, video.data() + (size_t)f0 * fb, (size_t)k * fb);     // a decoder would write here directly
        uint64_t ticket = 0;
        if (str_er_stream_submit(st, slot, w, h, 3 * w, (int64_t)fb, k, stages, &ticket) != STR_ER_OK) die("submit", str_er_stream_last_error(st));
        if (first_of_ticket.size() <= ticket) first_of_ticket.resize((size_t)ticket + 1);
        first_of_ticket[(size_t)ticket] = f0;
    }
    while (str_er_stream_pending(st)) drain_one();
    str_er_stream_destroy(st);

    // ---- the same frames through plain calls
    str_er_ctx *ctx = nullptr;
    if (str_er_create(&p, &ctx) != STR_ER_OK) die("str_er_create", str_er_last_error(nullptr));
    str_er_load_cascade(ctx, STR_ER_CASCADE_STRONG, argv[1]);
    str_er_load_cascade(ctx, STR_ER_CASCADE_WEAK, argv[2]);
    for (int f0 = 0; f0 < n; f0 += fpb) {
        const int k = std::min(fpb, n - f0);
        str_er_result *r = nullptr;
        if (str_er_detect_bgr(ctx, video.data() + (size_t)f0 * fb, w, h, 3 * w, (int64_t)fb, k, STR_ER_MEM_HOST, stages, &r) != STR_ER_OK)
            die("str_er_detect_bgr", str_er_last_error(ctx));
        collect(r, f0, want);
        str_er_result_free(r);
    }
    str_er_destroy(ctx);

    bool same = true;
    for (int i = 0; i < n; ++i) {
        std::printf("frame %d pool %d strong %d weak %d tracked %d lines %d\n", i, got[(size_t)i].pool, got[(size_t)i].strong, got[(size_t)i].weak,
                    got[(size_t)i].tracked, got[(size_t)i].lines);
        same = same && !std::memcmp(&got[(size_t)i], &want[(size_t)i], sizeof(FrameStats));
    }
    std::printf("stream == direct calls: %s\n", same ? "yes" : "NO");
    return same ? 0 : 1;
}
