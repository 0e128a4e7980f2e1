// er_filter_hip.hpp -- C++ host-side mirror of the reference's `class ERFilter` hot-path
// surface (inc/ER.h:110-136), implemented on top of the C ABI in include/str_er.h.
//
// The reference keeps its host code in C++ and so does this: the class below has the same
// constructor arguments, public members and method names as the reference for the path
//
//     text_detect -> compute_channels -> er_tree_extract -> non_maximum_supression -> classify
//
// but takes plain 8-bit buffers instead of cv::Mat (OpenCV is not a dependency of this
// library).  A maintainer of the reference wraps `cv::Mat::data/step` in the `Image8` view
// below (see INTEGRATION.md for the exact patch).  `struct ER` / `ERs` keep the reference's
// field names (inc/ER.h:42-82) so downstream code (er_track, er_grouping, er_ocr,
// src/ER.cpp:532-786) compiles against it unchanged.
//
// Header-only; link with -lstr_er_hip.  Errors are reported the way the reference does for
// CV_Assert (src/ER.cpp:242): by throwing (std::runtime_error instead of cv::Exception).
#pragma once

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/str_er.h"

namespace str_er_host {

// Borrowed view of an 8-bit image: what cv::Mat gives through .data/.cols/.rows/.step.
struct Image8 {
    const uint8_t *data = nullptr;
    int cols = 0, rows = 0;
    int64_t step = 0;      // bytes per row
    int channels = 1;      // 1 (8UC1 plane) or 3 (8UC3, BGR)
    Image8() = default;
    Image8(const uint8_t *d, int c, int r, int64_t s, int ch) : data(d), cols(c), rows(r), step(s), channels(ch) {}
};

struct Rect { int x = 0, y = 0, width = 0, height = 0; int area() const { return width * height; } };
struct Point { int x = 0, y = 0; };

// Field-for-field the hot-path part of the reference's struct ER (inc/ER.h:42-80).
struct ER {
    int pixel = 0, level = 0, x = 0, y = 0;
    int area = 0;
    Rect bound;
    bool done = false;
    double stability = 0;
    ER *parent = nullptr, *child = nullptr, *next = nullptr;
    Point center;                                  // set by er_track (src/ER.cpp:542)
    double color1 = 0, color2 = 0, color3 = 0;     // set by er_track -> calc_color
    int ch = 0;
    // set by classify on pooled ERs (not in the reference, where strong/weak are separate lists)
    double score_strong = -DBL_MAX, score_weak = 0;
    uint32_t key = 0;
};
typedef std::vector<ER *> ERs;

// struct Text (inc/ER.h:84-97), the part er_grouping fills
struct Text {
    ERs ers;
    double slope = 0;
    Rect box;
};

// Owns the nodes of one plane's tree (the reference leaks them in image_mode and frees them
// with ERFilter::er_delete in video_mode, src/ER.cpp:194-233, src/utils.cpp:212-213).
struct ERTree {
    std::vector<ER> nodes;
    ER *root = nullptr;
};

class AdaBoostHandle {   // stands for `AdaBoost *stc, *wtc` (inc/ER.h:117-118): a loaded cascade file
public:
    explicit AdaBoostHandle(std::string path) : path_(std::move(path)) {}
    const std::string &path() const { return path_; }
private:
    std::string path_;
};

class ERFilter {
public:
    // inc/ER.h:113 -- same argument order and defaults
    ERFilter(int thresh_step = 2, int min_area = 100, int max_area = 100000, int stability_t = 2,
             double overlap_coef = 0.7, double min_ocr_prob = 0.01, int max_width = 1920, int max_height = 1080,
             int max_frames = 1, int device = 0)
        : MIN_OCR_PROB(min_ocr_prob)
    {
        str_er_params p;
        str_er_default_params(&p);
        p.thresh_step = thresh_step; p.min_area = min_area; p.max_area = max_area; p.stability_t = stability_t;
        p.overlap_coef = overlap_coef; p.max_width = max_width; p.max_height = max_height; p.max_frames = max_frames;
        p.device = device;
        str_er_ctx *c = nullptr;
        const int rc = str_er_create(&p, &c);
        if (rc != STR_ER_OK) throw std::runtime_error(std::string("str_er_create: ") + str_er_last_error(nullptr));
        ctx_.reset(c, str_er_destroy);
    }

    //! modules (inc/ER.h:116-119): assigning a cascade file loads it onto the GPU
    std::shared_ptr<AdaBoostHandle> stc, wtc;
    void set_stc(const std::string &file) { load(STR_ER_CASCADE_STRONG, file); stc = std::make_shared<AdaBoostHandle>(file); }
    void set_wtc(const std::string &file) { load(STR_ER_CASCADE_WEAK, file); wtc = std::make_shared<AdaBoostHandle>(file); }

    void set_thresh_step(int t) { check(str_er_set_thresh_step(ctx_.get(), t)); }   // src/ER.cpp:21-24
    void set_min_area(int m) { check(str_er_set_min_area(ctx_.get(), m)); }         // src/ER.cpp:27-30

    // ERFilter::text_detect up to classify (src/ER.cpp:33-60).  root/pool/strong/weak are resized to
    // the number of channels exactly like the reference does (:42-46); `all` stays empty unless
    // GET_ALL_ER semantics are wanted (inc/ER.h:24).  Returns the 7-slot times vector (:99-110).
    std::vector<double> text_detect(const Image8 &src, std::vector<ERTree> &trees, ERs &root, std::vector<ERs> &all,
                                    std::vector<ERs> &pool, std::vector<ERs> &strong, std::vector<ERs> &weak)
    {
        if (src.channels != 3) throw std::runtime_error("text_detect expects an 8UC3 BGR image");
        str_er_result *r = nullptr;
        check(str_er_detect_bgr(ctx_.get(), src.data, src.cols, src.rows, src.step, src.step * (int64_t)src.rows, 1,
                                STR_ER_MEM_HOST, STR_ER_STAGE_ALL | STR_ER_WANT_NODES, &r));
        std::unique_ptr<str_er_result, void (*)(str_er_result *)> guard(r, str_er_result_free);
        const int n = str_er_result_n_planes(r);
        trees.assign(n, ERTree());
        root.assign(n, nullptr); all.assign(n, ERs()); pool.assign(n, ERs()); strong.assign(n, ERs()); weak.assign(n, ERs());
        for (int i = 0; i < n; ++i) unpack_plane(r, i, trees[i], pool[i], strong[i], weak[i]), root[i] = trees[i].root;
        const double *t = str_er_result_times(r);
        return std::vector<double>(t, t + 7);
    }

    // ER* ERFilter::er_tree_extract(Mat input) (src/ER.cpp:240-374); input must be 8UC1 (:242)
    ER *er_tree_extract(const Image8 &input, ERTree &tree)
    {
        if (input.channels != 1) throw std::runtime_error("er_tree_extract: input.type() == CV_8UC1");
        str_er_result *r = nullptr;
        check(str_er_detect_planes(ctx_.get(), input.data, input.cols, input.rows, input.step, 0, 1, STR_ER_MEM_HOST,
                                   STR_ER_STAGE_EXTRACT | STR_ER_WANT_NODES, &r));
        std::unique_ptr<str_er_result, void (*)(str_er_result *)> guard(r, str_er_result_free);
        ERs p, s, w;
        unpack_plane(r, 0, tree, p, s, w);
        return tree.root;
    }

    // void ERFilter::non_maximum_supression(ER *er, ERs &all, ERs &pool, Mat input) (src/ER.cpp:416-505);
    // `tree` is the table er_tree_extract filled.  pool comes back in ascending key order.
    void non_maximum_supression(ERTree &tree, ERs &all, ERs &pool, const Image8 &input, int *ambiguous = nullptr)
    {
        (void)all;
        std::vector<str_er_node> tab(tree.nodes.size());
        for (size_t i = 0; i < tab.size(); ++i) {
            const ER &e = tree.nodes[i];
            str_er_node &n = tab[i];
            n.key = e.key; n.parent = e.parent ? (int32_t)(e.parent - tree.nodes.data()) : (int32_t)i; n.area = e.area;
            n.x = (uint16_t)e.bound.x; n.y = (uint16_t)e.bound.y; n.w = (uint16_t)e.bound.width; n.h = (uint16_t)e.bound.height;
            n.level = (uint8_t)e.level; n.flags = (&e == tree.root) ? 1 : 0; n.reserved = 0;
            if (&e == tree.root) n.parent = (int32_t)i;
        }
        std::vector<int32_t> idx(tab.size());
        int32_t np = 0, amb = 0;
        // the table is in key order, so sibling ties are decided on the plane itself, by the reference's flood order
        if (input.data && input.channels == 1)
            check(str_er_nms_tree_plane(ctx_.get(), tab.data(), (int32_t)tab.size(), input.data, input.cols, input.rows, input.step,
                                        idx.data(), (int32_t)idx.size(), &np, &amb));
        else
            check(str_er_nms_tree(ctx_.get(), tab.data(), (int32_t)tab.size(), input.rows, input.cols, idx.data(),
                                  (int32_t)idx.size(), &np, &amb));
        pool.clear();
        for (int i = 0; i < np; ++i) pool.push_back(&tree.nodes[idx[i]]);
        if (ambiguous) *ambiguous = amb;
    }

    // void ERFilter::classify(ERs &pool, ERs &strong, ERs &weak, Mat input) (src/ER.cpp:507-528)
    void classify(ERs &pool, ERs &strong, ERs &weak, const Image8 &input)
    {
        const int n = (int)pool.size();
        std::vector<int32_t> boxes(4 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            boxes[4 * i] = pool[i]->bound.x; boxes[4 * i + 1] = pool[i]->bound.y;
            boxes[4 * i + 2] = pool[i]->bound.width; boxes[4 * i + 3] = pool[i]->bound.height;
        }
        std::vector<uint8_t> cls(n);
        std::vector<double> ss(n), sw(n);
        check(str_er_classify_boxes(ctx_.get(), input.data, input.cols, input.rows, input.step, boxes.data(), n, cls.data(),
                                    ss.data(), sw.data()));
        for (int i = 0; i < n; ++i) {
            pool[i]->score_strong = ss[i]; pool[i]->score_weak = sw[i];
            if (cls[i] == STR_ER_CLS_STRONG) strong.push_back(pool[i]);
            else if (cls[i] == STR_ER_CLS_WEAK) weak.push_back(pool[i]);
        }
    }

    // void ERFilter::compute_channels(Mat &src, Mat &YCrcb, vector<Mat> &channels) (src/ER.cpp:114-128):
    // six tightly packed planes [Y, Cr, Cb, 255-Y, 255-Cr, 255-Cb]
    void compute_channels(const Image8 &src, std::vector<std::vector<uint8_t>> &channels)
    {
        std::vector<uint8_t> six((size_t)6 * src.cols * src.rows);
        check(str_er_compute_channels(ctx_.get(), src.data, src.cols, src.rows, src.step, six.data()));
        const size_t n = (size_t)src.cols * src.rows;
        channels.assign(6, std::vector<uint8_t>());
        for (int i = 0; i < 6; ++i) channels[i].assign(six.begin() + i * n, six.begin() + (i + 1) * n);
    }

    // vector<double> ERFilter::make_LBP_hist(Mat input, N = 2, normalize_size = 24) (src/ER.cpp:789-816)
    std::vector<double> make_LBP_hist(const Image8 &input)
    {
        const int32_t box[4] = {0, 0, input.cols, input.rows};
        std::vector<double> h(1024);
        check(str_er_lbp_hist(ctx_.get(), input.data, input.cols, input.rows, input.step, box, 1, h.data(), nullptr));
        return h;
    }

    // Mat ERFilter::calc_LBP(Mat input, const int size = 24) (inc/ER.h:134, src/ER.cpp:819-845): the 24 x 24 code map, row-major
    std::vector<uint8_t> calc_LBP(const Image8 &input)
    {
        const int32_t box[4] = {0, 0, input.cols, input.rows};
        std::vector<uint8_t> lbp(24 * 24);
        check(str_er_calc_lbp(ctx_.get(), input.data, input.cols, input.rows, input.step, box, 1, lbp.data()));
        return lbp;
    }

    // void ERFilter::er_track(vector<ERs> &strong, vector<ERs> &weak, ERs &all_er, vector<Mat> &channel, Mat Ycrcb)
    // (src/ER.cpp:530-590).  channel[i] is the plane strong[i] / weak[i] came from, Ycrcb the 8UC3 image
    // compute_channels made.  all_er = the strong ERs (channel order), then the tracked weak ones in channel / list
    // order (the reference appends them in the order its nested loops find them; er_grouping sorts all_er anyway).
    void er_track(std::vector<ERs> &strong, std::vector<ERs> &weak, ERs &all_er, const std::vector<Image8> &channel, const Image8 &Ycrcb)
    {
        if (Ycrcb.channels != 3) throw std::runtime_error("er_track: Ycrcb must be 8UC3");
        std::vector<ER *> ers;
        std::vector<str_er_cand> cands;
        std::vector<double> colors;
        for (size_t i = 0; i < strong.size(); ++i)
            for (int pass = 0; pass < 2; ++pass) {
                ERs &list = pass == 0 ? strong[i] : weak[i];
                if (list.empty()) continue;
                std::vector<int32_t> boxes;
                for (ER *e : list) { boxes.push_back(e->bound.x); boxes.push_back(e->bound.y); boxes.push_back(e->bound.width); boxes.push_back(e->bound.height); }
                std::vector<double> col(3 * list.size());
                check(str_er_calc_color(ctx_.get(), channel[i].data, channel[i].cols, channel[i].rows, channel[i].step, Ycrcb.data, Ycrcb.cols,
                                        Ycrcb.rows, Ycrcb.step, boxes.data(), (int32_t)list.size(), col.data()));
                for (size_t k = 0; k < list.size(); ++k) {
                    ER *e = list[k];
                    e->color1 = col[3 * k]; e->color2 = col[3 * k + 1]; e->color3 = col[3 * k + 2];
                    e->center.x = e->bound.x + e->bound.width / 2; e->center.y = e->bound.y + e->bound.height / 2;
                    e->ch = (int)i;
                    str_er_cand c{};
                    c.ch = (uint8_t)i; c.cls = pass == 0 ? STR_ER_CLS_STRONG : STR_ER_CLS_WEAK;
                    c.x = (uint16_t)e->bound.x; c.y = (uint16_t)e->bound.y; c.w = (uint16_t)e->bound.width; c.h = (uint16_t)e->bound.height;
                    c.area = (uint32_t)e->area; c.key = e->key;
                    ers.push_back(e); cands.push_back(c);
                    colors.insert(colors.end(), col.begin() + 3 * k, col.begin() + 3 * k + 3);
                }
            }
        std::vector<uint8_t> tracked(ers.size());
        check(str_er_er_track(ctx_.get(), cands.data(), colors.data(), (int32_t)ers.size(), tracked.data(), nullptr, nullptr));
        for (size_t k = 0; k < ers.size(); ++k) if (cands[k].cls == STR_ER_CLS_STRONG) all_er.push_back(ers[k]);
        for (size_t k = 0; k < ers.size(); ++k) if (cands[k].cls == STR_ER_CLS_WEAK && tracked[k]) all_er.push_back(ers[k]);
    }

    // void ERFilter::er_grouping(ERs &all_er, vector<Text> &text, bool overlap_sup, bool inner_sup) (src/ER.cpp:612-692).
    // all_er comes back sorted by center.x (and inner-suppressed); bound / center of ERs that overlap_suppression merged
    // into are updated in place as in the reference, and the ERs it merged away are erased.
    void er_grouping(ERs &all_er, std::vector<Text> &text, bool overlap_sup = false, bool inner_sup = false)
    {
        // ties in center.x are broken by candidate order = (channel, key), whatever order all_er arrives in
        std::sort(all_er.begin(), all_er.end(), [](const ER *a, const ER *b) { return a->ch != b->ch ? a->ch < b->ch : a->key < b->key; });
        const int32_t n = (int32_t)all_er.size();
        std::vector<str_er_cand> cands((size_t)n);
        std::vector<str_er_track> tr((size_t)n);
        for (int32_t k = 0; k < n; ++k) {
            const ER *e = all_er[(size_t)k];
            str_er_cand c{};
            c.x = (uint16_t)e->bound.x; c.y = (uint16_t)e->bound.y; c.w = (uint16_t)e->bound.width; c.h = (uint16_t)e->bound.height;
            c.area = (uint32_t)e->area; c.ch = (uint8_t)e->ch; c.key = e->key;
            cands[(size_t)k] = c;
            str_er_track t{};
            t.color1 = e->color1; t.color2 = e->color2; t.color3 = e->color3; t.cx = e->center.x; t.cy = e->center.y; t.tracked = 1;
            tr[(size_t)k] = t;
        }
        str_er_result *r = nullptr;
        check(str_er_er_grouping(ctx_.get(), cands.data(), tr.data(), n, overlap_sup ? 1 : 0, inner_sup ? 1 : 0, &r));
        std::unique_ptr<str_er_result, void (*)(str_er_result *)> guard(r, str_er_result_free);
        int32_t nt = 0, ne = 0, nb = 0;
        const str_er_text   *tx = str_er_result_texts(r, &nt);
        const int32_t       *te = str_er_result_text_ers(r, &ne);
        const str_er_gbound *gb = str_er_result_group_bounds(r, &nb);
        for (int32_t k = 0; k < nb; ++k) {
            ER *e = all_er[(size_t)k];
            e->bound.x = gb[k].x; e->bound.y = gb[k].y; e->bound.width = gb[k].w; e->bound.height = gb[k].h;
            e->center.x = gb[k].cx; e->center.y = gb[k].cy;
        }
        (void)ne;
        const ERs in(all_er);
        for (int32_t i = 0; i < nt; ++i) {
            Text t;
            for (int32_t k = 0; k < tx[i].count; ++k) t.ers.push_back(in[(size_t)te[tx[i].first + k]]);
            t.slope = tx[i].slope;
            t.box.x = tx[i].x; t.box.y = tx[i].y; t.box.width = tx[i].w; t.box.height = tx[i].h;
            text.push_back(t);
        }
        int32_t na = 0;
        const int32_t *ga = str_er_result_group_all(r, &na);       // all_er: sorted, inner-suppressed
        all_er.clear();
        for (int32_t k = 0; k < na; ++k) all_er.push_back(in[(size_t)ga[k]]);
    }

    // void ERFilter::er_delete(ER *er) (src/ER.cpp:194-233): the table owns the nodes
    void er_delete(ERTree &tree) { tree.nodes.clear(); tree.root = nullptr; }

    str_er_ctx *handle() const { return ctx_.get(); }

private:
    double MIN_OCR_PROB;
    std::shared_ptr<str_er_ctx> ctx_;

    void check(int rc) const
    {
        if (rc != STR_ER_OK) throw std::runtime_error(std::string(str_er_strerror(rc)) + ": " + str_er_last_error(ctx_.get()));
    }
    void load(int which, const std::string &file)
    {
        // the reference prints and returns false (src/adaboost.cpp:877-881); callers ignore it (src/main.cpp:23-24).
        check(str_er_load_cascade(ctx_.get(), which, file.c_str()));
    }

    // node table -> intrusive tree with the reference's parent/child/next links; children are
    // prepended like er_merge does (src/ER.cpp:183-185), in ascending key order overall.
    static void unpack_plane(const str_er_result *r, int plane, ERTree &tree, ERs &pool, ERs &strong, ERs &weak)
    {
        int32_t nn = 0, nc = 0;
        const str_er_node *nodes = str_er_result_plane_nodes(r, plane, &nn);
        const str_er_cand *cands = str_er_result_plane_cands(r, plane, &nc);
        str_er_plane_info info;
        str_er_result_plane_info(r, plane, &info);
        tree.nodes.assign((size_t)nn, ER());
        tree.root = nn ? &tree.nodes[info.root] : nullptr;
        for (int i = 0; i < nn; ++i) {
            ER &e = tree.nodes[i];
            const str_er_node &n = nodes[i];
            e.level = n.level; e.area = n.area; e.key = n.key; e.ch = info.ch;
            e.bound.x = n.x; e.bound.y = n.y; e.bound.width = n.w; e.bound.height = n.h;
            e.pixel = (int)n.key; e.x = (int)(n.key % (uint32_t)info.width); e.y = (int)(n.key / (uint32_t)info.width);
        }
        for (int i = nn - 1; i >= 0; --i) {       // descending index + prepend = ascending child lists
            if (i == info.root) continue;
            ER &e = tree.nodes[i];
            ER &p = tree.nodes[nodes[i].parent];
            e.parent = &p; e.next = p.child; p.child = &e;
        }
        for (int i = 0; i < nc; ++i) {
            const str_er_cand &c = cands[i];
            if (c.node < 0) continue;
            ER *e = &tree.nodes[c.node];
            e->score_strong = c.score_strong; e->score_weak = c.score_weak;
            pool.push_back(e);
            if (c.cls == STR_ER_CLS_STRONG) strong.push_back(e);
            else if (c.cls == STR_ER_CLS_WEAK) weak.push_back(e);
        }
    }
};

// `class OCR` of the reference (inc/OCR.h:26-53), the part on the config-3 path: the constructor loads the
// libsvm model (OCR::OCR, src/OCR.cpp:19-22) and chain_run scores one ER (src/OCR.cpp:67-140).  It shares the
// ERFilter's context (the reference hangs the OCR object off ERFilter::ocr, inc/ER.h:119).
class OCR {
public:
    OCR(ERFilter &filter, const char *svm_file_name, int _img_L = 30, int _feature_L = 15) : ctx_(filter.handle())
    {
        if (_img_L != 30 || _feature_L != 15) throw std::runtime_error("OCR: only img_L = 30, feature_L = 15 (src/main.cpp:25) are built");
        if (str_er_load_svm_model(ctx_, svm_file_name, 8 * _feature_L * _feature_L) != STR_ER_OK)
            throw std::runtime_error(std::string("svm_load_model: ") + str_er_last_error(ctx_));
    }

    // double OCR::chain_run(Mat src, int thresh, double slope): src = channel(bound).  Returns table[label] + prob.
    // `thresh` is ignored exactly as in the reference (THRESH_OTSU overrides it); |slope| > 0.01 rotates the
    // binarised ROI by atan2(slope, 1) first (rotate_mat, src/OCR.cpp:73-78).
    double chain_run(const Image8 &plane, const Rect &bound, int /*thresh*/, double slope)
    {
        const int32_t box[4] = {bound.x, bound.y, bound.width, bound.height};
        int32_t label = 0;
        double  prob = 0;
        const int rc = str_er_ocr_chain_run_slope(ctx_, plane.data, plane.cols, plane.rows, plane.step, box, &slope, 1, &label, &prob, nullptr);
        if (rc != STR_ER_OK) throw std::runtime_error(std::string("chain_run: ") + str_er_last_error(ctx_));
        static const char *table = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz&()";   // src/OCR.cpp:10
        return (label >= 0 && label < 65 ? table[label] : '?') + prob;
    }

private:
    str_er_ctx *ctx_;
};

} // namespace str_er_host
