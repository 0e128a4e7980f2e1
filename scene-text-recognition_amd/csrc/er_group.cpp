// er_group.cpp -- see er_group.h.  Every std::sort by center.x of the reference is a STABLE sort here: the reference's
// is unstable, i.e. its order among equal center.x is unspecified, and this is the build's definition of it.
#include "er_group.h"

#include <algorithm>
#include <cmath>

namespace str_er {
namespace {

void sort_by_cx(const std::vector<GroupEr> &ers, std::vector<int32_t> &idx)
{
    std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return ers[(size_t)a].cx < ers[(size_t)b].cx; });
}

// ERFilter::overlap_suppression (src/ER.cpp:925-964)
void overlap_suppression(std::vector<GroupEr> &ers, std::vector<int32_t> &pool)
{
    std::vector<char> merged(pool.size(), 0);
    for (size_t i = 0; i < pool.size(); ++i)
        for (size_t j = i + 1; j < pool.size(); ++j) {
            if (merged[j]) continue;
            GroupEr       &a = ers[(size_t)pool[i]];
            const GroupEr &b = ers[(size_t)pool[j]];
            const int ix = std::max(a.x, b.x), iy = std::max(a.y, b.y);
            int       iw = std::min(a.x + a.w, b.x + b.w) - ix, ih = std::min(a.y + a.h, b.y + b.h) - iy;
            if (iw <= 0 || ih <= 0) iw = ih = 0;                                   // cv::Rect & : empty
            const int ux = std::min(a.x, b.x), uy = std::min(a.y, b.y);
            const int uw = std::max(a.x + a.w, b.x + b.w) - ux, uh = std::max(a.y + a.h, b.y + b.h) - uy;
            if ((double)(iw * ih) / (double)(uw * uh) > 0.5) {
                merged[j] = 1;
                const int x = (int)((a.x + b.x) * 0.5), y = (int)((a.y + b.y) * 0.5);
                const int width = (int)((a.w + b.w) * 0.5), height = (int)((a.h + b.h) * 0.5);
                a.x = x; a.y = y; a.h = height; a.w = width;
                a.cx = (int)(x + a.w * 0.5);
                a.cy = (int)(y + a.h * 0.5);
            }
        }
    size_t m = 0;
    for (size_t i = 0; i < pool.size(); ++i) if (!merged[i]) pool[m++] = pool[i];
    pool.resize(m);
}

// ERFilter::inner_suppression (src/ER.cpp:893-922)
void inner_suppression(const std::vector<GroupEr> &ers, std::vector<int32_t> &pool)
{
    std::vector<char> del(pool.size(), 0);
    for (size_t i = 0; i < pool.size(); ++i)
        for (size_t j = 0; j < pool.size(); ++j) {
            const GroupEr &a = ers[(size_t)pool[i]], &b = ers[(size_t)pool[j]];
            const double   dx = a.cx - b.cx, dy = a.cy - b.cy;
            if (std::sqrt(dx * dx + dy * dy) < 0.2 * std::max(a.w, a.h) && a.x <= b.x && a.y <= b.y && a.x + a.w >= b.x + b.w &&
                a.y + a.h >= b.y + b.h && (double)(a.w * a.h) / (double)(b.w * b.h) > 2.0)
                del[j] = 1;
        }
    size_t m = 0;
    for (size_t i = 0; i < pool.size(); ++i) if (!del[i]) pool[m++] = pool[i];
    pool.resize(m);
}

// fitline_avgslope (src/ER.cpp:1361-1389) over the bottom-right corners
double fitline_avgslope(const std::vector<int32_t> &px, const std::vector<int32_t> &py)
{
    const size_t n = px.size();
    if (n <= 2) return 0;
    const double epsilon = 0.07;
    double       slope = .0;
    for (size_t i = 0; i + 2 < n; ++i) {
        const double s12 = (double)(py[i] - py[i + 1]) / (px[i] - px[i + 1]);
        const double s23 = (double)(py[i + 1] - py[i + 2]) / (px[i + 1] - px[i + 2]);
        const double s13 = (double)(py[i] - py[i + 2]) / (px[i] - px[i + 2]);
        if (std::fabs(s12 - s23) < epsilon && std::fabs(s23 - s13) < epsilon && std::fabs(s12 - s13) < epsilon) slope += (s12 + s23 + s13) / 3;
        else if (std::fabs(s12) < std::fabs(s23) && std::fabs(s12) < std::fabs(s13)) slope += s12;
        else if (std::fabs(s23) < std::fabs(s12) && std::fabs(s23) < std::fabs(s13)) slope += s23;
        else if (std::fabs(s13) < std::fabs(s12) && std::fabs(s13) < std::fabs(s23)) slope += s13;
    }
    slope /= (double)(n - 2);
    return slope;
}

} // namespace

void sort_and_overlap_suppress(std::vector<GroupEr> &ers, std::vector<int32_t> &order)
{
    sort_by_cx(ers, order);                 // :614
    overlap_suppression(ers, order);        // :616-617
}

void group_lines(std::vector<GroupEr> &ers, const uint32_t *pairs, size_t n_pairs, std::vector<TextLine> &lines)
{
    lines.clear();
    std::vector<int32_t> gi(ers.size(), -1);
    for (size_t p = 0; p < n_pairs; ++p) {                                          // :627-664
        const int32_t i = (int32_t)(pairs[p] >> 16), j = (int32_t)(pairs[p] & 0xFFFFu);
        if (gi[(size_t)i] == -1 && gi[(size_t)j] == -1) {
            gi[(size_t)i] = gi[(size_t)j] = (int32_t)lines.size();
            lines.emplace_back();
            lines.back().ers.push_back(i);
            lines.back().ers.push_back(j);
        } else if (gi[(size_t)j] != -1) {
            gi[(size_t)i] = gi[(size_t)j];
            lines[(size_t)gi[(size_t)i]].ers.push_back(i);
        } else {
            gi[(size_t)j] = gi[(size_t)i];
            lines[(size_t)gi[(size_t)j]].ers.push_back(j);
        }
    }
    for (TextLine &t : lines) {                                                     // :666-691
        sort_by_cx(ers, t.ers);
        std::vector<int32_t> tmp(t.ers);
        overlap_suppression(ers, tmp);
        inner_suppression(ers, tmp);
        std::vector<int32_t> px(tmp.size()), py(tmp.size());
        for (size_t k = 0; k < tmp.size(); ++k) { px[k] = ers[(size_t)tmp[k]].x + ers[(size_t)tmp[k]].w; py[k] = ers[(size_t)tmp[k]].y + ers[(size_t)tmp[k]].h; }
        t.slope = fitline_avgslope(px, py);
        const GroupEr &f = ers[(size_t)t.ers.front()];
        int x0 = f.x, y0 = f.y, x1 = f.x + f.w, y1 = f.y + f.h;
        for (int32_t k : t.ers) {
            const GroupEr &e = ers[(size_t)k];
            x0 = std::min(x0, e.x); y0 = std::min(y0, e.y); x1 = std::max(x1, e.x + e.w); y1 = std::max(y1, e.y + e.h);
        }
        t.box[0] = x0; t.box[1] = y0; t.box[2] = x1 - x0; t.box[3] = y1 - y0;
    }
}

} // namespace str_er
