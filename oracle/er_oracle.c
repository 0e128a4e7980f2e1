/*
 * er_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle); see er_oracle.h.
 *
 * Plain C restatement of the reference's extremal-region hot path.  Every
 * function cites the reference lines it follows (paths relative to
 * /root/reference).  Nothing here is used by the product library.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math).
 */
#define _GNU_SOURCE /* qsort_r */
#include "er_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* OpenCV primitives, restated (unpinned: OpenCV is not in this image)      */
/* ======================================================================== */

/* cvRound(float): round half to even (lrintf in the default FP environment). */
static int round_half_even_f(float v) { return (int)lrintf(v); }

int ero_highest_level(int step) { return 255 / step + 1; } /* src/ER.cpp:247 */

/* src/ER.cpp:250 `input_clone /= THRESH_STEP` == convertTo(self,-1,1.0/step):
 * 8U->8U with a float scale, saturate_cast<uchar>(cvRound(p*a)). */
void ero_quant_lut(int step, uint8_t lut[256])
{
    const float a = (float)(1.0 / (double)step);
    for (int p = 0; p < 256; ++p) {
        int r = round_half_even_f((float)p * a);
        if (r < 0) r = 0;
        if (r > 255) r = 255;
        lut[p] = (uint8_t)r;
    }
}

static int clip_row(int v, int n) { return v < 0 ? 0 : (v < n ? v : n - 1); }

/* cv::resize, INTER_LINEAR, 8UC1 (called from OCR::ARAN, src/OCR.cpp:401).
 * Restates OpenCV 4.x resize.cpp: same-size copy; exact 2x2 decimation takes
 * the INTER_AREA fast path; otherwise fixed-point bilinear with 11-bit
 * coefficients (INTER_RESIZE_COEF_BITS), horizontal pass into int, vertical
 * pass ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2)>>2.                        */
void ero_resize_linear_u8(const uint8_t *src, int sstride, int sw, int sh,
                          uint8_t *dst, int dstride, int dw, int dh)
{
    if (dw <= 0 || dh <= 0) return;
    if (dw == sw && dh == sh) {
        for (int y = 0; y < sh; ++y) memcpy(dst + (size_t)y * dstride, src + (size_t)y * sstride, (size_t)sw);
        return;
    }
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1.0 / inv_sx, scale_y = 1.0 / inv_sy;
    {
        int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
        int fast = fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON;
        if (fast && isx == 2 && isy == 2) {
            for (int y = 0; y < dh; ++y) {
                const uint8_t *r0 = src + (size_t)(2 * y) * sstride, *r1 = r0 + sstride;
                for (int x = 0; x < dw; ++x)
                    dst[(size_t)y * dstride + x] =
                        (uint8_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + 2) >> 2);
            }
            return;
        }
    }
    int   *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *xa   = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int   sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
        xofs[dx] = sx;
        xa[2 * dx]     = (short)round_half_even_f((1.f - fx) * 2048.f);
        xa[2 * dx + 1] = (short)round_half_even_f(fx * 2048.f);
    }
    int *row0 = (int *)malloc(sizeof(int) * (size_t)dw), *row1 = (int *)malloc(sizeof(int) * (size_t)dw);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int   sy = (int)floorf(fy);
        fy -= (float)sy;
        const short b0 = (short)round_half_even_f((1.f - fy) * 2048.f);
        const short b1 = (short)round_half_even_f(fy * 2048.f);
        const uint8_t *s0 = src + (size_t)clip_row(sy, sh) * sstride;
        const uint8_t *s1 = src + (size_t)clip_row(sy + 1, sh) * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            const int sx1 = sx + 1 < sw ? sx + 1 : sx; /* coefficient is 0 there */
            row0[dx] = s0[sx] * xa[2 * dx] + s0[sx1] * xa[2 * dx + 1];
            row1[dx] = s1[sx] * xa[2 * dx] + s1[sx1] * xa[2 * dx + 1];
        }
        for (int dx = 0; dx < dw; ++dx) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)dy * dstride + dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(row0); free(row1); free(xofs); free(xa);
}

static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* src/ER.cpp:114-128.  OpenCV 8-bit BGR2YCrCb, yuv_shift 14:
 * Y=(1868B+9617G+4899R+8192)>>14, Cr=((R-Y)*11682+(128<<14)+8192)>>14,
 * Cb=((B-Y)*9241+(128<<14)+8192)>>14; then the three inversions 255-x.      */
void ero_compute_channels(const uint8_t *bgr, int stride, int w, int h, uint8_t *planes6)
{
    const size_t n = (size_t)w * h;
    for (int y = 0; y < h; ++y) {
        const uint8_t *p = bgr + (size_t)y * stride;
        for (int x = 0; x < w; ++x) {
            const int B = p[3 * x], G = p[3 * x + 1], R = p[3 * x + 2];
            const int Y  = (1868 * B + 9617 * G + 4899 * R + 8192) >> 14;
            const int Cr = ((R - Y) * 11682 + (128 << 14) + 8192) >> 14;
            const int Cb = ((B - Y) * 9241 + (128 << 14) + 8192) >> 14;
            const size_t i = (size_t)y * w + x;
            const uint8_t y8 = sat_u8(Y), cr8 = sat_u8(Cr), cb8 = sat_u8(Cb);
            planes6[0 * n + i] = y8;
            planes6[1 * n + i] = cr8;
            planes6[2 * n + i] = cb8;
            planes6[3 * n + i] = (uint8_t)(255 - y8);
            planes6[4 * n + i] = (uint8_t)(255 - cr8);
            planes6[5 * n + i] = (uint8_t)(255 - cb8);
        }
    }
}

/* NV12 ingest -- BUILD-DEFINED, no reference counterpart (the reference decodes to BGR: src/utils.cpp:31, 109), like the pyramid.
 * y: w*h luma bytes (y_stride per row); uv: interleaved Cb,Cr pairs, (w/2) pairs x (h/2) rows (uv_stride per row).
 * planes3 = [Y, Cr, Cb], each w*h contiguous: Y is the luma byte, Cr(x,y) = V(x/2, y/2), Cb(x,y) = U(x/2, y/2).            */
void ero_nv12_to_ycrcb(const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride, int w, int h, uint8_t *planes3)
{
    const size_t n = (size_t)w * h;
    for (int r = 0; r < h; ++r)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)r * w + x;
            planes3[i] = y[(size_t)r * y_stride + x];
            planes3[n + i] = uv[(size_t)(r / 2) * uv_stride + 2 * (x / 2) + 1];
            planes3[2 * n + i] = uv[(size_t)(r / 2) * uv_stride + 2 * (x / 2)];
        }
}

void ero_pyr_dims(int w0, int h0, int level, int *w, int *h)
{
    const double s = pow(2.0, -0.5 * level);
    int ww = (int)floor(w0 * s + 0.5), hh = (int)floor(h0 * s + 0.5);
    *w = ww < 1 ? 1 : ww;
    *h = hh < 1 ? 1 : hh;
}

/* ======================================================================== */
/* Component tree: the flood of src/ER.cpp:240-413                          */
/* ======================================================================== */

typedef struct fnode {
    int level, area;
    int x0, y0, x1, y1;      /* inclusive bbox (cv::Rect x,y,x+w-1,y+h-1)    */
    int sx, sy;               /* seed pixel coords (ER::x, ER::y)             */
    int parent, child, next;
    int key, npix, nsub;
    int alive;
} fnode;

typedef struct fvec { fnode *a; int n, cap; } fvec;

static int fv_new(fvec *v, int level, int x, int y)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 1024;
        v->a = (fnode *)realloc(v->a, sizeof(fnode) * (size_t)v->cap);
    }
    fnode *e = &v->a[v->n];
    /* ER::ER (src/ER.cpp:6-10): area starts at 1, bound = Rect(x,y,1,1) */
    e->level = level; e->area = 1;
    e->x0 = e->x1 = x; e->y0 = e->y1 = y; e->sx = x; e->sy = y;
    e->parent = e->child = e->next = -1;
    e->key = INT_MAX; e->npix = 0; e->nsub = 1; e->alive = 1;
    return v->n++;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* er_merge, src/ER.cpp:153-191 */
static void f_merge(fvec *v, int pi, int ci, int min_area, int *dead_branch)
{
    fnode *p = &v->a[pi], *c = &v->a[ci];
    p->area += c->area; p->npix += c->npix; p->nsub += c->nsub;
    p->x0 = imin(p->x0, c->x0); p->x1 = imax(p->x1, c->x1);
    p->y0 = imin(p->y0, c->y0); p->y1 = imax(p->y1, c->y1);
    if (c->area <= min_area) {
        int nc = c->child;
        if (nc != -1) { /* src/ER.cpp:169-178; unreachable in practice */
            ++*dead_branch;
            int last = nc;
            while (v->a[last].next != -1) last = v->a[last].next;
            v->a[last].next = p->child;
            p->child = nc;
            v->a[nc].parent = pi;
        }
        c->alive = 0;
    } else {
        c->next = p->child; /* children are prepended */
        p->child = ci;
        c->parent = pi;
    }
}

static void tree_from_flood(fvec *v, int root, ero_tree *out)
{
    /* pre-order walk in list order; renumber */
    int *map = (int *)malloc(sizeof(int) * (size_t)v->n);
    int *stack = (int *)malloc(sizeof(int) * (size_t)v->n);
    int *order = (int *)malloc(sizeof(int) * (size_t)v->n);
    for (int i = 0; i < v->n; ++i) map[i] = -1;
    int sp = 0, cnt = 0;
    stack[sp++] = root;
    while (sp) {
        int i = stack[--sp];
        map[i] = cnt; order[cnt++] = i;
        /* push siblings chain of the child list in reverse so the first child pops first */
        int tmpn = 0;
        for (int c = v->a[i].child; c != -1; c = v->a[c].next) ++tmpn;
        int base = sp; sp += tmpn;
        int k = 0;
        for (int c = v->a[i].child; c != -1; c = v->a[c].next, ++k) stack[base + tmpn - 1 - k] = c;
    }
    out->nodes = (ero_node *)malloc(sizeof(ero_node) * (size_t)(cnt ? cnt : 1));
    out->n_nodes = cnt;
    out->root = 0;
    for (int k = 0; k < cnt; ++k) {
        const fnode *e = &v->a[order[k]];
        ero_node *o = &out->nodes[k];
        o->level = e->level; o->area = e->area;
        o->x = e->x0; o->y = e->y0; o->w = e->x1 - e->x0 + 1; o->h = e->y1 - e->y0 + 1;
        o->parent = (order[k] == root || e->parent < 0) ? -1 : map[e->parent];
        o->child = e->child < 0 ? -1 : map[e->child];
        o->next = e->next < 0 ? -1 : map[e->next];
        o->key = e->key; o->npix = e->npix; o->nsub = e->nsub;
    }
    free(map); free(stack); free(order);
}

int ero_tree_extract(const uint8_t *img, int stride, int w, int h,
                     int thresh_step, int min_area, ero_tree *out)
{
    if (!img || !out || w <= 0 || h <= 0 || thresh_step <= 0) return -1;
    memset(out, 0, sizeof(*out));
    const int    hi = ero_highest_level(thresh_step);
    const size_t n = (size_t)w * h;
    uint8_t lut[256];
    ero_quant_lut(thresh_step, lut);
    uint8_t *q = (uint8_t *)malloc(n);                  /* input_clone / step     */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) q[(size_t)y * w + x] = lut[img[(size_t)y * stride + x]];
    uint8_t *acc = (uint8_t *)calloc(n, 1);             /* pixel_accessible       */
    int *nxt = (int *)malloc(sizeof(int) * n);          /* bucket stacks (LIFO)   */
    uint8_t *edg = (uint8_t *)malloc(n);
    int head[258];
    for (int i = 0; i < 258; ++i) head[i] = -1;
    fvec v = {0, 0, 0};
    int *stk = (int *)malloc(sizeof(int) * 300);        /* component stack <= 258 */
    int sp = 0;
    int dead = 0;
    long long created = 0;

    int prio = hi;
    stk[sp++] = fv_new(&v, 256, 0, 0);                  /* dummy, src/ER.cpp:263  */
    int cur = 0, edge = 0, lvl = q[0];
    acc[0] = 1;
    int x, y;

descend:                                                /* step_3, src/ER.cpp:273 */
    x = cur % w; y = cur / w;
    stk[sp++] = fv_new(&v, lvl, x, y); ++created;
    /* seed pixel bookkeeping only (ER::pixel is never read downstream) */
    for (;;) {
        for (; edge < 4; ++edge) {                      /* src/ER.cpp:294-332     */
            int nb = cur;
            switch (edge) {
            case 0: if (x + 1 < w) nb = cur + 1; break; /* right  */
            case 1: if (y + 1 < h) nb = cur + w; break; /* bottom */
            case 2: if (x > 0) nb = cur - 1; break;     /* left   */
            default: if (y > 0) nb = cur - w; break;    /* top    */
            }
            if (nb != cur && !acc[nb]) {
                acc[nb] = 1;
                const int nl = q[nb];
                if (nl >= lvl) {
                    nxt[nb] = head[nl]; head[nl] = nb; edg[nb] = 0;
                    if (nl < prio) prio = nl;
                } else {
                    nxt[cur] = head[lvl]; head[lvl] = cur; edg[cur] = (uint8_t)(edge + 1);
                    if (lvl < prio) prio = lvl;
                    cur = nb; lvl = nl; edge = 0;
                    goto descend;
                }
            }
        }
        {   /* er_accumulate, src/ER.cpp:131-151 */
            fnode *t = &v.a[stk[sp - 1]];
            t->area++; t->npix++;
            t->x0 = imin(t->x0, x); t->x1 = imax(t->x1, x);
            t->y0 = imin(t->y0, y); t->y1 = imax(t->y1, y);
            if (cur < t->key) t->key = cur;
        }
        if (prio == hi) break;                          /* src/ER.cpp:343-347     */
        const int np = head[prio];
        head[prio] = nxt[np];
        const int ne = edg[np];
        const int ng = q[np];
        while (prio < hi && head[prio] == -1) ++prio;   /* src/ER.cpp:357-358     */
        cur = np; edge = ne; x = cur % w; y = cur / w;
        if (ng != lvl) {
            lvl = ng;
            /* process_stack, src/ER.cpp:377-413 */
            for (;;) {
                const int top = stk[--sp];
                const int second = stk[sp - 1];
                if (ng < v.a[second].level) {
                    const int nn = fv_new(&v, ng, v.a[top].sx, v.a[top].sy); ++created;
                    stk[sp++] = nn;
                    f_merge(&v, nn, top, min_area, &dead);
                    break;
                }
                f_merge(&v, second, top, min_area, &dead);
                if (!(ng > v.a[stk[sp - 1]].level)) break;
            }
        }
    }
    const int root = stk[sp - 1];
    tree_from_flood(&v, root, out);
    out->n_created = created;
    out->dead_branch = dead;
    free(v.a); free(stk); free(q); free(acc); free(nxt); free(edg);
    return 0;
}

void ero_tree_free(ero_tree *t)
{
    if (t && t->nodes) { free(t->nodes); t->nodes = NULL; t->n_nodes = 0; }
}

/* ------------------------------------------------------------------------ */
/* Brute force: canonical node set straight from the definition */
/* ------------------------------------------------------------------------ */
typedef struct bnode { int level, npix, x0, y0, x1, y1, key, parent, nsub, rep; } bnode;

static int cmp_key(const void *a, const void *b, void *ctx)
{
    const ero_node *n = (const ero_node *)ctx;
    int ka = n[*(const int *)a].key, kb = n[*(const int *)b].key;
    if (ka != kb) return ka < kb ? -1 : 1;
    int la = n[*(const int *)a].level, lb = n[*(const int *)b].level;
    return la < lb ? -1 : (la > lb);
}

int ero_tree_bruteforce(const uint8_t *img, int stride, int w, int h,
                        int thresh_step, int min_area, ero_tree *out)
{
    if (!img || !out || w <= 0 || h <= 0 || thresh_step <= 0) return -1;
    memset(out, 0, sizeof(*out));
    const int hi = ero_highest_level(thresh_step);
    const size_t n = (size_t)w * h;
    uint8_t lut[256];
    ero_quant_lut(thresh_step, lut);
    uint8_t *q = (uint8_t *)malloc(n);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) q[(size_t)y * w + x] = lut[img[(size_t)y * stride + x]];
    /* start rules */
    int start = -1;
    if (q[0] < hi) start = 0;
    else if (w > 1 && q[1] < hi) start = 1;
    else if (h > 1 && q[w] < hi) start = w;
    if (start < 0) {
        out->nodes = (ero_node *)calloc(1, sizeof(ero_node));
        out->n_nodes = 1; out->root = 0; out->n_created = 1;
        ero_node *o = &out->nodes[0];
        o->level = q[0]; o->area = 2; o->x = 0; o->y = 0; o->w = 1; o->h = 1;
        o->parent = o->child = o->next = -1; o->key = 0; o->npix = 1; o->nsub = 1;
        free(q);
        return 0;
    }
    /* S = 4-connected component of {q < hi} containing start */
    uint8_t *inS = (uint8_t *)calloc(n, 1);
    int *queue = (int *)malloc(sizeof(int) * n);
    {
        int qh = 0, qt = 0;
        queue[qt++] = start; inS[start] = 1;
        while (qh < qt) {
            int p = queue[qh++], px = p % w, py = p / w;
            int nb[4] = { px + 1 < w ? p + 1 : -1, py + 1 < h ? p + w : -1, px > 0 ? p - 1 : -1, py > 0 ? p - w : -1 };
            for (int k = 0; k < 4; ++k)
                if (nb[k] >= 0 && !inS[nb[k]] && q[nb[k]] < hi) { inS[nb[k]] = 1; queue[qt++] = nb[k]; }
        }
    }
    int maxl = 0;
    for (size_t i = 0; i < n; ++i) if (inS[i] && q[i] > maxl) maxl = q[i];
    /* label every threshold independently */
    int **lab = (int **)calloc((size_t)maxl + 1, sizeof(int *));
    int *nodeof_base = (int *)calloc((size_t)maxl + 2, sizeof(int));
    bnode *bn = NULL; int nb_n = 0, nb_cap = 0;
    int **comp2node = (int **)calloc((size_t)maxl + 1, sizeof(int *));
    for (int t = 0; t <= maxl; ++t) {
        int *L = (int *)malloc(sizeof(int) * n);
        for (size_t i = 0; i < n; ++i) L[i] = -1;
        int ncomp = 0;
        int c2n_cap = 64; int *c2n = (int *)malloc(sizeof(int) * (size_t)c2n_cap);
        for (size_t s = 0; s < n; ++s) {
            if (!inS[s] || q[s] > t || L[s] != -1) continue;
            int qh = 0, qt = 0, has_t = 0, key = INT_MAX, npix = 0;
            int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
            queue[qt++] = (int)s; L[s] = ncomp;
            while (qh < qt) {
                int p = queue[qh++], px = p % w, py = p / w;
                ++npix;
                if (q[p] == t) { has_t = 1; if (p < key) key = p; }
                x0 = imin(x0, px); x1 = imax(x1, px); y0 = imin(y0, py); y1 = imax(y1, py);
                int nbp[4] = { px + 1 < w ? p + 1 : -1, py + 1 < h ? p + w : -1, px > 0 ? p - 1 : -1, py > 0 ? p - w : -1 };
                for (int k = 0; k < 4; ++k) {
                    int r = nbp[k];
                    if (r >= 0 && inS[r] && q[r] <= t && L[r] == -1) { L[r] = ncomp; queue[qt++] = r; }
                }
            }
            if (ncomp == c2n_cap) { c2n_cap *= 2; c2n = (int *)realloc(c2n, sizeof(int) * (size_t)c2n_cap); }
            c2n[ncomp] = -1;
            if (has_t) {
                if (nb_n == nb_cap) { nb_cap = nb_cap ? nb_cap * 2 : 1024; bn = (bnode *)realloc(bn, sizeof(bnode) * (size_t)nb_cap); }
                bnode *b = &bn[nb_n];
                b->level = t; b->npix = npix; b->x0 = x0; b->y0 = y0; b->x1 = x1; b->y1 = y1;
                b->key = key; b->parent = -1; b->nsub = 1; b->rep = (int)s;
                c2n[ncomp] = nb_n++;
            }
            ++ncomp;
        }
        lab[t] = L; comp2node[t] = c2n;
    }
    (void)nodeof_base;
    /* parents: first higher threshold whose component (containing rep) is a node */
    for (int i = 0; i < nb_n; ++i) {
        for (int t = bn[i].level + 1; t <= maxl; ++t) {
            int c = lab[t][bn[i].rep];
            if (comp2node[t][c] >= 0) { bn[i].parent = comp2node[t][c]; break; }
        }
    }
    /* nodes were created in increasing level order -> children precede parents */
    for (int i = 0; i < nb_n; ++i) if (bn[i].parent >= 0) bn[bn[i].parent].nsub += bn[i].nsub;
    int rootb = 0;
    for (int i = 0; i < nb_n; ++i) if (bn[i].parent < 0) rootb = i; /* exactly one: S is connected */
    /* prune: keep area > min_area, plus the root */
    int *map = (int *)malloc(sizeof(int) * (size_t)nb_n);
    int cnt = 0;
    for (int i = 0; i < nb_n; ++i) {
        int area = bn[i].npix + bn[i].nsub;
        map[i] = (area > min_area || i == rootb) ? cnt++ : -1;
    }
    out->nodes = (ero_node *)calloc((size_t)cnt, sizeof(ero_node));
    out->n_nodes = cnt; out->n_created = nb_n;
    for (int i = 0; i < nb_n; ++i) {
        if (map[i] < 0) continue;
        ero_node *o = &out->nodes[map[i]];
        o->level = bn[i].level; o->area = bn[i].npix + bn[i].nsub;
        o->x = bn[i].x0; o->y = bn[i].y0; o->w = bn[i].x1 - bn[i].x0 + 1; o->h = bn[i].y1 - bn[i].y0 + 1;
        o->key = bn[i].key; o->npix = bn[i].npix; o->nsub = bn[i].nsub;
        o->child = -1; o->next = -1;
        int p = bn[i].parent;
        while (p >= 0 && map[p] < 0) p = bn[p].parent; /* never loops: areas grow upward */
        o->parent = p < 0 ? -1 : map[p];
    }
    out->root = map[rootb];
    /* child lists in ascending key order */
    {
        int *idx = (int *)malloc(sizeof(int) * (size_t)cnt);
        for (int i = 0; i < cnt; ++i) idx[i] = i;
        qsort_r(idx, (size_t)cnt, sizeof(int), cmp_key, out->nodes);
        for (int k = cnt - 1; k >= 0; --k) { /* prepend in descending order -> ascending lists */
            int i = idx[k], p = out->nodes[i].parent;
            if (p >= 0) { out->nodes[i].next = out->nodes[p].child; out->nodes[p].child = i; }
        }
        free(idx);
    }
    for (int t = 0; t <= maxl; ++t) { free(lab[t]); free(comp2node[t]); }
    free(lab); free(comp2node); free(nodeof_base); free(bn); free(map);
    free(q); free(inS); free(queue);
    return 0;
}

/* ======================================================================== */
/* NMS, src/ER.cpp:416-505                                                  */
/* ======================================================================== */
int ero_nms(const ero_tree *t, int rows, int cols, const ero_params *p,
            int sibling_mode, int32_t **pool_out, int32_t *n_pool, int32_t *ambiguous_out)
{
    if (!t || !p || !pool_out || !n_pool) return -1;
    const int n = t->n_nodes;
    const ero_node *nd = t->nodes;
    /* children arrays in the requested order */
    int *cstart = (int *)calloc((size_t)n + 1, sizeof(int));
    int *clist = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    for (int i = 0; i < n; ++i)
        for (int c = nd[i].child; c != -1; c = nd[c].next) cstart[i + 1]++;
    for (int i = 0; i < n; ++i) cstart[i + 1] += cstart[i];
    for (int i = 0; i < n; ++i) {
        int k = cstart[i];
        for (int c = nd[i].child; c != -1; c = nd[c].next) clist[k++] = c;
        if (sibling_mode != 0) {
            qsort_r(clist + cstart[i], (size_t)(k - cstart[i]), sizeof(int), cmp_key, (void *)nd);
            if (sibling_mode == 2)
                for (int a = cstart[i], b = k - 1; a < b; ++a, --b) { int tmp = clist[a]; clist[a] = clist[b]; clist[b] = tmp; }
        }
    }
    /* post-order (children in list order, each subtree completely before the next sibling) */
    int *order = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    int *st = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    int *it = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    int sp = 0, no = 0;
    st[sp] = t->root; it[sp] = cstart[t->root]; ++sp;
    while (sp) {
        int i = st[sp - 1];
        if (it[sp - 1] < cstart[i + 1]) { int c = clist[it[sp - 1]++]; st[sp] = c; it[sp] = cstart[c]; ++sp; }
        else { order[no++] = i; --sp; }
    }
    uint8_t *done = (uint8_t *)calloc((size_t)(n ? n : 1), 1);
    int32_t *pool = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    int *chain = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    double *stab = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    int np = 0, amb = 0;
    const int T = p->stability_t;
    for (int k = 0; k < no; ++k) {
        const int X = order[k];
        if (done[X]) continue;
        int len = 0, par = X;
        const int ax = nd[X].w * nd[X].h; /* (root->bound & parent->bound).area(): X's box lies inside */
        for (;;) {
            const int ap = nd[par].w * nd[par].h;
            const int pass = (double)ax / (double)ap > p->overlap_coef;
            if (!(pass && !done[par])) {
                if (pass && done[par] && len > 0 && par != chain[len - 1]) ++amb;
                break;
            }
            done[par] = 1;
            chain[len++] = par;
            par = (nd[par].parent < 0) ? par : nd[par].parent; /* root->parent = root (:424) */
        }
        if (len >= 1 + T) {
            const int m = len - T;
            for (int i = 0; i < m; ++i) {
                const int a = nd[chain[i]].w * nd[chain[i]].h, b = nd[chain[i + T]].w * nd[chain[i + T]].h;
                stab[i] = (double)a / (double)(b - a);
            }
            int mx = 0;
            for (int i = 1; i < m; ++i) {
                if (stab[i] > stab[mx]) mx = i;
                else if (stab[i] == stab[mx])
                    mx = (nd[chain[i]].w * nd[chain[i]].h < nd[chain[mx]].w * nd[chain[mx]].h) ? i : mx;
            }
            const ero_node *e = &nd[chain[mx]];
            const double ar = (double)e->w / (double)e->h;
            if (ar < 2.0 && ar > 0.10 && e->area < p->max_area && e->area > p->min_area &&
                e->h < rows * 0.8 && e->w < cols * 0.8)
                pool[np++] = chain[mx];
        }
    }
    *pool_out = pool; *n_pool = np;
    if (ambiguous_out) *ambiguous_out = amb;
    free(cstart); free(clist); free(order); free(st); free(it); free(done); free(chain); free(stab);
    return 0;
}

/* ======================================================================== */
/* classify chain                                                           */
/* ======================================================================== */

/* OCR::ARAN with L = 26, para = 0.5 (src/OCR.cpp:394-430) */
void ero_aran26(const uint8_t *roi, int stride, int w, int h, uint8_t tile[26 * 26])
{
    const int L = 26;
    const double R1 = (w > h) ? (double)h / w : (double)w / h;
    int dw, dh;
    if (w > h) { dw = L; dh = (int)(L * pow(R1, 0.5)); }
    else       { dw = (int)(L * pow(R1, 0.5)); dh = L; }
    memset(tile, 0, (size_t)L * L);
    if (dw <= 0 || dh <= 0) return; /* cv::resize would throw; callers never get here via NMS */
    uint8_t tmp[26 * 26];
    ero_resize_linear_u8(roi, stride, w, h, tmp, dw, dw, dh);
    if (dw > dh) {
        const int off = (int)round((double)((L - dh) / 2));
        for (int i = 0; i < dh; ++i) memcpy(tile + (size_t)(i + off) * L, tmp + (size_t)i * dw, (size_t)dw);
    } else {
        const int off = (int)round((double)((L - dw) / 2));
        for (int i = 0; i < dh; ++i) memcpy(tile + (size_t)i * L + off, tmp + (size_t)i * dw, (size_t)dw);
    }
}

/* Size OCR::ARAN resizes to (src/OCR.cpp:396-397), and a self-test used by tests/: the HIP
 * kernel computes (int)(26*sqrt(R1)) where the reference writes (int)(26*pow(R1,0.5)); this
 * counts the (w,h) pairs up to maxdim for which this libm's pow() disagrees with sqrt(). */
void ero_aran_dims(int w, int h, int *dw, int *dh)
{
    const double R1 = (w > h) ? (double)h / w : (double)w / h;
    if (w > h) { *dw = 26; *dh = (int)(26 * pow(R1, 0.5)); }
    else       { *dw = (int)(26 * pow(R1, 0.5)); *dh = 26; }
}

long ero_selftest_pow_vs_sqrt(int maxdim)
{
    long bad = 0;
    for (int w = 1; w <= maxdim; ++w)
        for (int h = 1; h <= maxdim; ++h) {
            volatile double r = (w > h) ? (double)h / w : (double)w / h;
            if ((int)(26 * pow(r, 0.5)) != (int)(26 * sqrt(r))) ++bad;
        }
    return bad;
}

/* ERFilter::calc_LBP (src/ER.cpp:819-845): neighbour offsets are written with
 * `size` (24) although the tile is 26 wide -- kept as is. */
void ero_lbp24(const uint8_t tile[26 * 26], uint8_t lbp[24 * 24])
{
    const int size = 24;
    static const int sgn[8][2] = { {-1, -1}, {-1, 0}, {-1, 1}, {0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1} };
    for (int i = 0; i < size; ++i) {
        const uint8_t *pin = tile + (i + 1) * 26 + 1;
        for (int j = 0; j < size; ++j) {
            int v[8], sum = 0;
            for (int k = 0; k < 8; ++k) { v[k] = pin[j + sgn[k][0] * size + sgn[k][1]]; sum += v[k]; }
            const double thresh = sum / 8.0;
            int code = 0;
            for (int k = 0; k < 8; ++k) code += (v[k] > thresh) << k;
            lbp[i * size + j] = (uint8_t)code;
        }
    }
}

/* ERFilter::make_LBP_hist with N = 2, normalize_size = 24 (src/ER.cpp:789-816) */
void ero_lbp_hist(const uint8_t *roi, int stride, int w, int h, double hist[1024])
{
    uint8_t tile[26 * 26], lbp[24 * 24];
    ero_aran26(roi, stride, w, h, tile);
    ero_lbp24(tile, lbp);
    for (int i = 0; i < 1024; ++i) hist[i] = 0.0;
    for (int m = 0; m < 2; ++m)
        for (int nn = 0; nn < 2; ++nn)
            for (int i = 0; i < 12; ++i)
                for (int j = 0; j < 12; ++j)
                    hist[m * 2 * 256 + nn * 256 + lbp[(m * 12 + i) * 24 + nn * 12 + j]] += 1.0;
}

/* ---- cascade ------------------------------------------------------------ */
struct ero_cascade {
    int real;            /* 1 REAL, 0 DISCRETE */
    int n_stages;
    int *stage_n;
    int *stage_thresh;   /* (int)stod(...) truncation, src/adaboost.cpp:919 */
    int n_stumps;
    double *weight;
    int *dim;
    double *thresh, *cp, *cn; /* DISCRETE: cp = dir */
};

static char *next_token(char **cursor)
{
    char *s = *cursor;
    while (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\n') ++s;
    if (!*s) { *cursor = s; return NULL; }
    char *e = s;
    while (*e && *e != ' ' && *e != '\t' && *e != '\r' && *e != '\n') ++e;
    if (*e) { *e = 0; ++e; }
    *cursor = e;
    return s;
}

static int parse_num(const char *tok, double *v)
{
    char *end = NULL;
    *v = strtod(tok, &end);
    return end != tok;
}

static ero_cascade *cascade_parse(char *text)
{
    ero_cascade *c = (ero_cascade *)calloc(1, sizeof(*c));
    c->real = 1;
    char *cur = text, *tok;
    int cap_st = 0;
    tok = next_token(&cur);
    if (tok && !strcmp(tok, "boost_type")) { tok = next_token(&cur); c->real = !(tok && !strcmp(tok, "DISCRETE")); tok = next_token(&cur); }
    if (tok && !strcmp(tok, "base_type")) { tok = next_token(&cur); tok = next_token(&cur); }
    if (tok && !strcmp(tok, "num_of_iter")) {
        for (;;) {
            tok = next_token(&cur);
            double v;
            if (!tok || !parse_num(tok, &v)) break;
            if (c->n_stages == cap_st) { cap_st = cap_st ? cap_st * 2 : 8; c->stage_n = (int *)realloc(c->stage_n, sizeof(int) * (size_t)cap_st); }
            c->stage_n[c->n_stages++] = (int)v;
        }
    }
    c->stage_thresh = (int *)calloc((size_t)(c->n_stages ? c->n_stages : 1), sizeof(int));
    if (tok && !strcmp(tok, "threshold")) {
        for (int j = 0; j < c->n_stages; ++j) {
            tok = next_token(&cur);
            double v = 0;
            if (tok) parse_num(tok, &v);
            c->stage_thresh[j] = (int)v;
        }
    }
    /* remaining tokens: rows of `weight p0 p1 p2 [p3]` */
    const int per = c->real ? 5 : 4;
    int cap = 0;
    for (;;) {
        double vals[5];
        int k = 0;
        for (; k < per; ++k) {
            tok = next_token(&cur);
            if (!tok || !parse_num(tok, &vals[k])) break;
        }
        if (k < per) break;
        if (c->n_stumps == cap) {
            cap = cap ? cap * 2 : 1024;
            c->weight = (double *)realloc(c->weight, sizeof(double) * (size_t)cap);
            c->dim = (int *)realloc(c->dim, sizeof(int) * (size_t)cap);
            c->thresh = (double *)realloc(c->thresh, sizeof(double) * (size_t)cap);
            c->cp = (double *)realloc(c->cp, sizeof(double) * (size_t)cap);
            c->cn = (double *)realloc(c->cn, sizeof(double) * (size_t)cap);
        }
        const int i = c->n_stumps++;
        c->weight[i] = vals[0];
        c->dim[i] = (int)vals[1];
        if (c->real) { c->thresh[i] = vals[2]; c->cp[i] = vals[3]; c->cn[i] = vals[4]; }
        else         { c->cp[i] = (double)(int)vals[2]; c->thresh[i] = vals[3]; c->cn[i] = 0; }
    }
    return c;
}

ero_cascade *ero_cascade_load(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
    buf[sz] = 0;
    fclose(f);
    ero_cascade *c = cascade_parse(buf);
    free(buf);
    return c;
}

void ero_cascade_free(ero_cascade *c)
{
    if (!c) return;
    free(c->stage_n); free(c->stage_thresh); free(c->weight); free(c->dim);
    free(c->thresh); free(c->cp); free(c->cn); free(c);
}

int ero_cascade_n_stages(const ero_cascade *c) { return c ? c->n_stages : 0; }
int ero_cascade_n_stumps(const ero_cascade *c) { return c ? c->n_stumps : 0; }

/* CascadeBoost::predict (src/adaboost.cpp:507-542) */
double ero_cascade_predict(const ero_cascade *c, const double fv[1024])
{
    double score = 0;
    int off = 0;
    for (int i = 0; i < c->n_stages; ++i) {
        score = 0;
        for (int j = off; j < off + c->stage_n[i] && j < c->n_stumps; ++j) {
            if (c->real) score += (fv[c->dim[j]] < c->thresh[j]) ? c->cp[j] : c->cn[j];
            else {
                const double dir = c->cp[j];
                score += ((fv[c->dim[j]] * dir < c->thresh[j] * dir) ? 1 : -1) * c->weight[j];
            }
        }
        if (score < c->stage_thresh[i]) return -DBL_MAX;
        off += c->stage_n[i];
    }
    return score;
}

/* ERFilter::classify (src/ER.cpp:507-528) */
void ero_classify(const uint8_t *plane, int stride, const int32_t *boxes, int n,
                  const ero_cascade *strong, const ero_cascade *weak,
                  uint8_t *cls, double *s_strong, double *s_weak)
{
    double fv[1024];
    for (int i = 0; i < n; ++i) {
        const int x = boxes[4 * i], y = boxes[4 * i + 1], w = boxes[4 * i + 2], h = boxes[4 * i + 3];
        ero_lbp_hist(plane + (size_t)y * stride + x, stride, w, h, fv);
        const double ss = ero_cascade_predict(strong, fv);
        double sw = 0;
        int c = 0;
        if (ss > -DBL_MAX) c = 1;
        else {
            sw = ero_cascade_predict(weak, fv);
            if (sw > -DBL_MAX) c = 2;
        }
        if (cls) cls[i] = (uint8_t)c;
        if (s_strong) s_strong[i] = ss;
        if (s_weak) s_weak[i] = sw;
    }
}

/* ======================================================================== */
/* OCR scorer, feature half (src/OCR.cpp:67-218, 254-357) -- unpinned        */
/* ======================================================================== */

/* cv::threshold(..., THRESH_OTSU): getThreshVal_Otsu_8u of OpenCV 4.x (imgproc/thresh.cpp).
 * `invert` applies 255 - p first (chain_run thresholds 255 - src, src/OCR.cpp:72). */
int ero_otsu_threshold(const uint8_t *img, int stride, int w, int h, int invert)
{
    int hist[256] = {0};
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) hist[invert ? 255 - img[(size_t)y * stride + x] : img[(size_t)y * stride + x]]++;
    double mu = 0, scale = 1. / ((double)w * h);
    for (int i = 0; i < 256; ++i) mu += i * (double)hist[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; ++i) {
        double p_i = hist[i] * scale, q2, mu2, sigma;
        mu1 *= q1;
        q1 += p_i;
        q2 = 1. - q1;
        const double mn = q1 < q2 ? q1 : q2, mx = q1 > q2 ? q1 : q2;
        if (mn < FLT_EPSILON || mx > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        mu2 = (mu - q1 * mu1) / q2;
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return (int)max_val;
}

/* chain_run lines 72 + 79: Otsu-binarise 255-roi (dst = v > thresh ? 255 : 0), then ARAN(img_L = 30) */
void ero_ocr_normalise(const uint8_t *roi, int stride, int w, int h, uint8_t img30[30 * 30])
{
    const int L = 30;
    const int th = ero_otsu_threshold(roi, stride, w, h, 1);
    uint8_t *bin = (uint8_t *)malloc((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) bin[(size_t)y * w + x] = (255 - roi[(size_t)y * stride + x]) > th ? 255 : 0;
    const double R1 = (w > h) ? (double)h / w : (double)w / h;
    int dw, dh;
    if (w > h) { dw = L; dh = (int)(L * pow(R1, 0.5)); }
    else       { dw = (int)(L * pow(R1, 0.5)); dh = L; }
    memset(img30, 0, (size_t)L * L);
    if (dw > 0 && dh > 0) {
        uint8_t tmp[30 * 30];
        ero_resize_linear_u8(bin, w, w, h, tmp, dw, dw, dh);
        if (dw > dh) {
            const int off = (L - dh) / 2;
            for (int i = 0; i < dh; ++i) memcpy(img30 + (size_t)(i + off) * L, tmp + (size_t)i * dw, (size_t)dw);
        } else {
            const int off = (L - dw) / 2;
            for (int i = 0; i < dh; ++i) memcpy(img30 + (size_t)i * L + off, tmp + (size_t)i * dw, (size_t)dw);
        }
    }
    free(bin);
}

/* OCR::rotate_mat (src/OCR.cpp:254-357): rotate about ((cols-1)/2, (rows-1)/2) by rad with a
 * hand-written bilinear tap; with `crop` the rows that would show the slanted top/bottom edges are
 * cut (crop_height may be negative for rad < 0, which grows the canvas instead -- kept).  The last
 * row/column of the canvas are never written (loops are exclusive), nor is the first row when
 * cropping.  *dst is malloc'ed (dw x dh, tight); returns 0.                                      */
int ero_rotate_mat(const uint8_t *src, int w, int h, double rad, int crop, uint8_t **dst, int *dw, int *dh)
{
    const int x0 = (int)((w - 1) / 2.0), y0 = (int)((h - 1) / 2.0);
    const int cx[4] = {0 - x0, (w - 1) - x0, (w - 1) - x0, 0 - x0};
    const int cy[4] = {0 - y0, 0 - y0, (h - 1) - y0, (h - 1) - y0};
    int nx[4], ny[4];
    for (int k = 0; k < 4; ++k) {
        nx[k] = (int)round(cx[k] * cos(rad) - cy[k] * sin(rad));
        ny[k] = (int)round(cx[k] * sin(rad) + cy[k] * cos(rad));
    }
    int max_x = nx[0], max_y = ny[0], min_x = nx[0], min_y = ny[0];
    for (int k = 1; k < 4; ++k) {
        if (nx[k] > max_x) max_x = nx[k];
        if (ny[k] > max_y) max_y = ny[k];
        if (nx[k] < min_x) min_x = nx[k];
        if (ny[k] < min_y) min_y = ny[k];
    }
    int ch = 0;
    if (crop) {
        ch = (int)((nx[1] - nx[0]) * tan(rad) * 0.5);
        if (max_y - min_y + 1 - 2 * ch <= 0) return ero_rotate_mat(src, w, h, rad, 0, dst, dw, dh);
    }
    const int rw = max_x - min_x + 1, rh = max_y - min_y + 1 - 2 * ch;
    uint8_t  *tmp = (uint8_t *)calloc((size_t)rw * rh, 1);
    for (int i = min_y + ch; i < max_y - ch; ++i) {
        uint8_t *t = tmp + (size_t)(i - min_y - ch) * rw;
        for (int j = min_x; j < max_x; ++j) {
            const double new_j = cos(rad) * j - sin(rad) * (i - ch) + x0;
            const double new_i = sin(rad) * j + cos(rad) * (i - ch) + y0;
            if (!(new_i > 0 && new_j > 0 && new_i < h - 1 && new_j < w - 1)) continue;
            if (crop && !(i > (min_y + ch) && i < (max_y - ch))) continue;
            const uint8_t *sp = src + (size_t)(int)new_i * w + (int)new_j;
            if (new_i == floor(new_i) && new_j == floor(new_j)) t[j - min_x] = sp[0];
            else {
                const double alpha = new_i - floor(new_i), beta = new_j - floor(new_j);
                const uint8_t A = sp[0], B = sp[1], C = sp[w], D = sp[w + 1];
                t[j - min_x] = (uint8_t)round((1 - alpha) * (1 - beta) * A + (1 - alpha) * beta * B + alpha * (1 - beta) * C + alpha * beta * D);
            }
        }
    }
    *dst = tmp; *dw = rw; *dh = rh;
    return 0;
}

/* chain_run lines 72-79 with a text-line slope: Otsu-binarise 255-roi, rotate_mat(atan2(slope,1), crop)
 * when |slope| > 0.01, then ARAN(30).                                                                */
void ero_ocr_normalise_slope(const uint8_t *roi, int stride, int w, int h, double slope, uint8_t img30[30 * 30])
{
    const int L = 30;
    if (!(fabs(slope) > 0.01)) { ero_ocr_normalise(roi, stride, w, h, img30); return; }
    const int th = ero_otsu_threshold(roi, stride, w, h, 1);
    uint8_t *bin = (uint8_t *)malloc((size_t)w * h), *rot = NULL;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) bin[(size_t)y * w + x] = (255 - roi[(size_t)y * stride + x]) > th ? 255 : 0;
    int rw = 0, rh = 0;
    ero_rotate_mat(bin, w, h, atan2(slope, 1), 1, &rot, &rw, &rh);
    free(bin);
    const double R1 = (rw > rh) ? (double)rh / rw : (double)rw / rh;
    int dw, dh;
    if (rw > rh) { dw = L; dh = (int)(L * pow(R1, 0.5)); }
    else         { dw = (int)(L * pow(R1, 0.5)); dh = L; }
    memset(img30, 0, (size_t)L * L);
    if (dw > 0 && dh > 0) {
        uint8_t tmp[30 * 30];
        ero_resize_linear_u8(rot, rw, rw, rh, tmp, dw, dw, dh);
        if (dw > dh) {
            const int off = (L - dh) / 2;
            for (int i = 0; i < dh; ++i) memcpy(img30 + (size_t)(i + off) * L, tmp + (size_t)i * dw, (size_t)dw);
        } else {
            const int off = (L - dw) / 2;
            for (int i = 0; i < dh; ++i) memcpy(img30 + (size_t)i * L + off, tmp + (size_t)i * dw, (size_t)dw);
        }
    }
    free(rot);
}

/* OCR::chain_code_direction(p1 = next point, p2 = current point), src/OCR.cpp:602-622 */
static int chain_dir(int nx, int ny, int cx, int cy)
{
    if (nx < cx && ny == cy) return 0;
    if (nx < cx && ny < cy) return 1;
    if (nx == cx && ny < cy) return 2;
    if (nx > cx && ny < cy) return 3;
    if (nx > cx && ny == cy) return 4;
    if (nx > cx && ny > cy) return 5;
    if (nx == cx && ny > cy) return 6;
    if (nx < cx && ny > cy) return 7;
    return -1;
}

/* cv::findContours(RETR_LIST, CHAIN_APPROX_NONE) restated from OpenCV's contours.cpp (Suzuki-Abe border
 * following, cvFindNextContour + icvFetchContour) fused with the loop of extract_feature
 * (src/OCR.cpp:155-168): every border is closed, single-point borders are skipped, and for each
 * consecutive pair (p_j, p_j+1) the bitmap of direction(p_j+1 relative to p_j) gets 255 at p_j. */
void ero_chain_bitmaps(const uint8_t img30[30 * 30], uint8_t maps[8 * 30 * 30])
{
    enum { L = 30, W = L + 2 };
    signed char f[W * W];                       /* 0/1 image with a one-pixel zero frame */
    memset(f, 0, sizeof(f));
    for (int y = 0; y < L; ++y)
        for (int x = 0; x < L; ++x) f[(y + 1) * W + x + 1] = img30[y * L + x] ? 1 : 0;
    memset(maps, 0, 8 * L * L);
    /* 8 neighbours, counter-clockwise on screen starting east (CV_INIT_3X3_DELTAS) */
    const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int delta[8];
    for (int k = 0; k < 8; ++k) delta[k] = dy[k] * W + dx[k];
    static int px[4 * W * W], py[4 * W * W];
    for (int y = 1; y <= L; ++y) {
        int prev = 0;
        for (int x = 1; x <= L + 1; ++x) {
            int p = f[y * W + x];
            if (p == prev) continue;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) { prev = p; continue; }
                is_hole = 1;
            }
            /* ---- icvFetchContour from (x - is_hole, y) ---- */
            const int i0 = y * W + x - is_hole;
            const int nbd = 2;
            int s_end = is_hole ? 0 : 4, s = s_end, i1 = i0, n = 0;
            do { s = (s - 1) & 7; i1 = i0 + delta[s]; } while (f[i1] == 0 && s != s_end);
            int cx = x - is_hole, cy = y;
            if (s == s_end) {                   /* isolated pixel */
                f[i0] = (signed char)(nbd | -128);
                px[n] = cx; py[n] = cy; ++n;
            } else {
                int i3 = i0;
                for (;;) {
                    int i4;
                    s_end = s;
                    for (;;) { i4 = i3 + delta[++s & 7]; if (f[i4] != 0) break; }
                    s &= 7;
                    if ((unsigned)(s - 1) < (unsigned)s_end) f[i3] = (signed char)(nbd | -128);
                    else if (f[i3] == 1) f[i3] = (signed char)nbd;
                    px[n] = cx; py[n] = cy; ++n;
                    cx += dx[s]; cy += dy[s];
                    if (i4 == i0 && i3 == i1) break;
                    i3 = i4;
                    s = (s + 4) & 7;
                }
            }
            if (n > 1)
                for (int j = 0; j < n; ++j) {
                    const int jn = (j + 1) % n;          /* contour closed with its first point (:163) */
                    const int d = chain_dir(px[jn] - 1, py[jn] - 1, px[j] - 1, py[j] - 1);
                    if (d >= 0) maps[d * L * L + (py[j] - 1) * L + (px[j] - 1)] = 255;
                }
            p = f[y * W + x];
            prev = p;
        }
    }
}

static int reflect101(int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i; return i; }

void ero_chain_features(const uint8_t *roi, int stride, int w, int h, uint8_t q[1800])
{
    ero_chain_features_slope(roi, stride, w, h, 0.0, q);
}

void ero_chain_features_slope(const uint8_t *roi, int stride, int w, int h, double slope, uint8_t q[1800])
{
    enum { L = 30, F = 15 };
    uint8_t img[L * L], maps[8 * L * L];
    ero_ocr_normalise_slope(roi, stride, w, h, slope, img);
    ero_chain_bitmaps(img, maps);
    static const int kg[7] = {8, 28, 56, 72, 56, 28, 8};   /* getGaussianKernel(7, sigma<=0) in 8.8 fixed point */
    for (int c = 0; c < 8; ++c) {
        const uint8_t *m = maps + c * L * L;
        int hrow[L * L];
        for (int y = 0; y < L; ++y)
            for (int x = 0; x < L; ++x) {
                int s = 0;
                for (int k = -3; k <= 3; ++k) s += m[y * L + reflect101(x + k, L)] * kg[k + 3];
                hrow[y * L + x] = s;                         /* ufixedpoint16, 8 fractional bits */
            }
        uint8_t blur[L * L];
        int mn = 255, mx = 0;
        for (int y = 0; y < L; ++y)
            for (int x = 0; x < L; ++x) {
                int s = 0;
                for (int k = -3; k <= 3; ++k) s += hrow[reflect101(y + k, L) * L + x] * kg[k + 3];
                const int v = (s + (1 << 15)) >> 16;         /* round to nearest, 16 fractional bits */
                blur[y * L + x] = (uint8_t)(v > 255 ? 255 : v);
                if (blur[y * L + x] < mn) mn = blur[y * L + x];
                if (blur[y * L + x] > mx) mx = blur[y * L + x];
            }
        /* cv::normalize(NORM_MINMAX, 0..255) = convertTo(CV_8U, scale, shift) in float */
        const double scale = (mx - mn) > DBL_EPSILON ? 255.0 / (mx - mn) : 0.0, shift = 0.0 - mn * scale;
        const float a = (float)scale, b = (float)shift;
        uint8_t nrm[L * L];
        for (int i = 0; i < L * L; ++i) {
            const float v = (float)blur[i] * a + b;
            int r = (int)lrintf(v);
            nrm[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
        for (int y = 0; y < F; ++y)                          /* resize 30 -> 15: exact 2x, INTER_AREA path */
            for (int x = 0; x < F; ++x)
                q[c * F * F + y * F + x] = (uint8_t)((nrm[(2 * y) * L + 2 * x] + nrm[(2 * y) * L + 2 * x + 1] +
                                                      nrm[(2 * y + 1) * L + 2 * x] + nrm[(2 * y + 1) * L + 2 * x + 1] + 2) >> 2);
    }
}
