/* TEST INFRASTRUCTURE ONLY -- see er_group_oracle.h. */
#include "er_group_oracle.h"
#include "er_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

void ero_calc_color(const uint8_t *mask_plane, int mstride, int bx, int by, int bw, int bh,
                    const uint8_t *color_img, int cstride, double out[3])
{
    const uint8_t *roi = mask_plane + (size_t)by * mstride + bx;
    const int th = ero_otsu_threshold(roi, mstride, bw, bh, 1);        /* threshold(255-img, img, 128, 255, THRESH_OTSU) */
    int    count = 0;
    double color1 = 0, color2 = 0, color3 = 0;
    for (int i = 0; i < bh; ++i) {
        const uint8_t *ptr = roi + (size_t)i * mstride;
        const uint8_t *color_ptr = color_img + (size_t)i * cstride;    /* row i of the whole image (:1404) */
        for (int j = 0, k = 0; j < bw; ++j, k += 3)
            if ((255 - ptr[j]) > th) {
                ++count;
                color1 += color_ptr[k];
                color2 += color_ptr[k + 1];
                color3 += color_ptr[k + 2];
            }
    }
    out[0] = color1 / count;
    out[1] = color2 / count;
    out[2] = color3 / count;
}

int ero_er_track(ero_er *ers, int n, int *order)
{
    for (int i = 0; i < n; ++i) { ers[i].cx = ers[i].x + ers[i].w / 2; ers[i].cy = ers[i].y + ers[i].h / 2; }
    int na = 0;
    for (int i = 0; i < n; ++i) if (ers[i].cls == 1) order[na++] = i;
    uint8_t *tracked = (uint8_t *)calloc((size_t)n + 1, 1);
    for (int i = 0; i < na; ++i) {
        const ero_er *s = &ers[order[i]];
        for (int k = 0; k < n; ++k) {
            if (ers[k].cls != 2 || tracked[k]) continue;
            const ero_er *w = &ers[k];
            if (abs(s->cx - w->cx) + abs(s->cy - w->cy) < imax(s->w, s->h) << 1 &&
                abs(s->h - w->h) < imin(s->h, w->h) &&
                abs(s->w - w->w) < (s->w + w->w) >> 1 &&
                fabs(s->color1 - w->color1) < 25 &&
                fabs(s->color2 - w->color2) < 25 &&
                fabs(s->color3 - w->color3) < 25 &&
                abs(s->area - w->area) < imin(s->area, w->area) * 3) {
                tracked[k] = 1;
                order[na++] = k;
            }
        }
    }
    free(tracked);
    return na;
}

/* ---- er_grouping ------------------------------------------------------------------------ */
static void stable_sort_by_cx(const ero_er *ers, int *idx, int n)
{
    for (int i = 1; i < n; ++i) {                 /* insertion sort: stable, n is small */
        const int v = idx[i];
        int j = i - 1;
        while (j >= 0 && ers[idx[j]].cx > ers[v].cx) { idx[j + 1] = idx[j]; --j; }
        idx[j + 1] = v;
    }
}

static int rect_area(int w, int h) { return w * h; }

/* inner_suppression (src/ER.cpp:893-922) on the list idx[0..*n) */
static void inner_suppression(const ero_er *ers, int *idx, int *n)
{
    const double T1 = 2.0, T2 = 0.2;
    uint8_t *del = (uint8_t *)calloc((size_t)*n + 1, 1);
    for (int i = 0; i < *n; ++i)
        for (int j = 0; j < *n; ++j) {
            const ero_er *a = &ers[idx[i]], *b = &ers[idx[j]];
            const double dx = a->cx - b->cx, dy = a->cy - b->cy;
            if (sqrt(dx * dx + dy * dy) < T2 * imax(a->w, a->h))            /* cv::norm(Point) */
                if (a->x <= b->x && a->y <= b->y && a->x + a->w >= b->x + b->w && a->y + a->h >= b->y + b->h &&
                    (double)rect_area(a->w, a->h) / (double)rect_area(b->w, b->h) > T1)
                    del[j] = 1;
        }
    int m = 0;
    for (int i = 0; i < *n; ++i) if (!del[i]) idx[m++] = idx[i];
    *n = m;
    free(del);
}

/* overlap_suppression (src/ER.cpp:925-964): merges j into i (averaged bound, recomputed center) in place */
static void overlap_suppression(ero_er *ers, int *idx, int *n)
{
    uint8_t *merged = (uint8_t *)calloc((size_t)*n + 1, 1);
    for (int i = 0; i < *n; ++i)
        for (int j = i + 1; j < *n; ++j) {
            if (merged[j]) continue;
            ero_er *a = &ers[idx[i]];
            const ero_er *b = &ers[idx[j]];
            /* cv::Rect & and | */
            const int ix = imax(a->x, b->x), iy = imax(a->y, b->y);
            int iw = imin(a->x + a->w, b->x + b->w) - ix, ih = imin(a->y + a->h, b->y + b->h) - iy;
            if (iw <= 0 || ih <= 0) { iw = 0; ih = 0; }
            const int ux = imin(a->x, b->x), uy = imin(a->y, b->y);
            const int uw = imax(a->x + a->w, b->x + b->w) - ux, uh = imax(a->y + a->h, b->y + b->h) - uy;
            if ((double)rect_area(iw, ih) / (double)rect_area(uw, uh) > 0.5) {
                merged[j] = 1;
                const int x = (int)((a->x + b->x) * 0.5), y = (int)((a->y + b->y) * 0.5);
                const int width = (int)((a->w + b->w) * 0.5), height = (int)((a->h + b->h) * 0.5);
                a->x = x; a->y = y; a->h = height; a->w = width;
                a->cx = (int)(x + a->w * 0.5);
                a->cy = (int)(y + a->h * 0.5);
            }
        }
    int m = 0;
    for (int i = 0; i < *n; ++i) if (!merged[i]) idx[m++] = idx[i];
    *n = m;
    free(merged);
}

double ero_fitline_avgslope(const int *px, const int *py, int n)
{
    if (n <= 2) return 0;
    const double epsilon = 0.07;
    double slope = .0;
    for (int i = 0; i < n - 2; ++i) {
        const double slope12 = (double)(py[i + 0] - py[i + 1]) / (px[i + 0] - px[i + 1]);
        const double slope23 = (double)(py[i + 1] - py[i + 2]) / (px[i + 1] - px[i + 2]);
        const double slope13 = (double)(py[i + 0] - py[i + 2]) / (px[i + 0] - px[i + 2]);
        if (fabs(slope12 - slope23) < epsilon && fabs(slope23 - slope13) < epsilon && fabs(slope12 - slope13) < epsilon)
            slope += (slope12 + slope23 + slope13) / 3;
        else if (fabs(slope12) < fabs(slope23) && fabs(slope12) < fabs(slope13))
            slope += slope12;
        else if (fabs(slope23) < fabs(slope12) && fabs(slope23) < fabs(slope13))
            slope += slope23;
        else if (fabs(slope13) < fabs(slope12) && fabs(slope13) < fabs(slope23))
            slope += slope13;
    }
    slope /= (n - 2);
    return slope;
}

typedef struct { int *v; int n, cap; } ivec;
static void ivec_push(ivec *a, int x)
{
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 8; a->v = (int *)realloc(a->v, sizeof(int) * (size_t)a->cap); }
    a->v[a->n++] = x;
}

int ero_er_grouping(ero_er *ers, int n, int overlap_sup, int inner_sup, int *all_idx, int *n_all,
                    int *line_first, int *line_count, double *slope, int *box, int line_cap, int *member, int member_cap)
{
    int m = n;
    for (int i = 0; i < n; ++i) all_idx[i] = i;
    stable_sort_by_cx(ers, all_idx, m);                                    /* :614 */
    if (overlap_sup) overlap_suppression(ers, all_idx, &m);
    if (inner_sup) inner_suppression(ers, all_idx, &m);
    *n_all = m;

    int  *gi = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    ivec *text = NULL;
    int   n_text = 0, text_cap = 0;
    for (int i = 0; i < m; ++i) gi[i] = -1;
    for (int i = 0; i < m; ++i) {
        const ero_er *a = &ers[all_idx[i]];
        for (int j = i + 1; j < m; ++j) {
            const ero_er *b = &ers[all_idx[j]];
            if (abs(a->cx - b->cx) < imax(a->w, b->w) * 3.0 &&
                abs(a->cy - b->cy) < (a->h + b->h) * 0.25 &&
                abs(a->h - b->h) < imin(a->h, b->h) &&
                abs(a->w - b->w) < imin(a->h, b->h * 2) &&
                fabs(a->color1 - b->color1) < 25 &&
                fabs(a->color2 - b->color2) < 25 &&
                fabs(a->color3 - b->color3) < 25 &&
                abs(a->area - b->area) < imin(a->area, b->area) * 4) {
                if (gi[i] == -1 && gi[j] == -1) {
                    gi[i] = gi[j] = n_text;
                    if (n_text == text_cap) { text_cap = text_cap ? 2 * text_cap : 8; text = (ivec *)realloc(text, sizeof(ivec) * (size_t)text_cap); }
                    text[n_text].v = NULL; text[n_text].n = text[n_text].cap = 0;
                    ivec_push(&text[n_text], all_idx[i]);
                    ivec_push(&text[n_text], all_idx[j]);
                    ++n_text;
                } else if (gi[j] != -1) {
                    gi[i] = gi[j];
                    ivec_push(&text[gi[i]], all_idx[i]);
                } else {
                    gi[j] = gi[i];
                    ivec_push(&text[gi[j]], all_idx[j]);
                }
            }
        }
    }
    int rc = n_text, used = 0;
    if (n_text > line_cap) rc = -1;
    for (int t = 0; t < n_text && rc >= 0; ++t) {
        ivec *tx = &text[t];
        stable_sort_by_cx(ers, tx->v, tx->n);                               /* :668 */
        int *tmp = (int *)malloc(sizeof(int) * (size_t)tx->n), tn = tx->n;
        memcpy(tmp, tx->v, sizeof(int) * (size_t)tn);
        overlap_suppression(ers, tmp, &tn);
        inner_suppression(ers, tmp, &tn);
        int *px = (int *)malloc(sizeof(int) * (size_t)(tn + 1)), *py = (int *)malloc(sizeof(int) * (size_t)(tn + 1));
        for (int j = 0; j < tn; ++j) { px[j] = ers[tmp[j]].x + ers[tmp[j]].w; py[j] = ers[tmp[j]].y + ers[tmp[j]].h; }   /* bound.br() */
        slope[t] = ero_fitline_avgslope(px, py, tn);
        free(px); free(py); free(tmp);
        if (used + tx->n > member_cap) { rc = -1; break; }
        line_first[t] = used; line_count[t] = tx->n;
        int x0 = ers[tx->v[0]].x, y0 = ers[tx->v[0]].y, x1 = x0 + ers[tx->v[0]].w, y1 = y0 + ers[tx->v[0]].h;
        for (int j = 0; j < tx->n; ++j) {
            const ero_er *e = &ers[tx->v[j]];
            member[used++] = tx->v[j];
            x0 = imin(x0, e->x); y0 = imin(y0, e->y); x1 = imax(x1, e->x + e->w); y1 = imax(y1, e->y + e->h);
        }
        box[4 * t] = x0; box[4 * t + 1] = y0; box[4 * t + 2] = x1 - x0; box[4 * t + 3] = y1 - y0;
    }
    for (int t = 0; t < n_text; ++t) free(text[t].v);
    free(text); free(gi);
    return rc;
}
