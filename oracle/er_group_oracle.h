/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's candidate post-processing
 * (SURVEY 8(f) rows 1-2): calc_color, er_track, er_grouping and its helpers.  Nothing in the
 * product may include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Parity status: the per-pixel part of calc_color uses cv::threshold(THRESH_OTSU) (restated in
 * er_oracle.c, unpinned); everything else here is plain integer / f64 logic of the reference
 * (src/ER.cpp) with no OpenCV algorithm behind it, but the reference cannot be compiled in this
 * image (it includes opencv2/opencv.hpp), so these functions are checked by hand-made cases only.
 */
#ifndef ER_GROUP_ORACLE_H
#define ER_GROUP_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* the fields of struct ER (inc/ER.h:42-80) that er_track / er_grouping read or write */
typedef struct ero_er {
    int32_t x, y, w, h;        /* bound */
    int32_t cx, cy;            /* center (set by er_track, src/ER.cpp:542/552) */
    int32_t area;
    int32_t ch;                /* channel the ER came from */
    int32_t cls;               /* 1 = came in through strong[ch], 2 = through weak[ch] */
    int32_t id;                /* caller's tag, carried along */
    double  color1, color2, color3;
} ero_er;

/* calc_color (src/ER.cpp:1391-1419): Otsu-binarise 255 - mask(bound), average the three bytes of
 * color_img under the mask -- color_img is indexed from ITS OWN row 0 / column 0 (color_img.ptr(i),
 * k = 3j), not from bound; kept.  count == 0 gives 0.0/0 (NaN).                                   */
void ero_calc_color(const uint8_t *mask_plane, int mstride, int bx, int by, int bw, int bh,
                    const uint8_t *color_img, int cstride, double out[3]);

/* er_track (src/ER.cpp:530-590) on the ERs of one image: ers[] in the order strong[0], weak[0] ...
 * do not matter -- strong ones (cls 1) are taken in array order, then weak ones (cls 2, array order
 * = the m, n loops) join while the rule at :575-587 links them to anything already in all_er.
 * Sets cx, cy.  order[] receives the indices of all_er in the reference's order; returns its size. */
int ero_er_track(ero_er *ers, int n, int *order);

/* ERFilter::er_grouping(all_er, text, overlap_sup, inner_sup) (src/ER.cpp:612-692) with its helpers
 * inner_suppression (:893-922), overlap_suppression (:925-964) and fitline_avgslope (:1361-1389).
 * ers[0..n) is all_er (cx, cy, colours set); every std::sort by center.x of the reference is made
 * STABLE here (ties keep their current order) -- the reference's sort is unstable, so its tie order is
 * unspecified; this is the build's definition.  overlap_suppression rewrites bound/center of the ERs it
 * merges into, in place (they are shared between lines), exactly as the reference does.
 *
 * Outputs: *n_all / all_idx = all_er after the sort (and suppressions), as indices into ers[];
 * lines: line_first[k], line_count[k] slice member[] (indices into ers[], in the line's sorted order),
 * slope[k], box[4k..] = union of the members' bounds at that moment (the !DO_OCR branch, :684-690).
 * member_cap / line_cap bound the arrays; returns the number of lines or -1 on overflow.          */
int ero_er_grouping(ero_er *ers, int n, int overlap_sup, int inner_sup, int *all_idx, int *n_all,
                    int *line_first, int *line_count, double *slope, int *box, int line_cap, int *member, int member_cap);

double ero_fitline_avgslope(const int *px, const int *py, int n);

#ifdef __cplusplus
}
#endif
#endif
