/*
 * er_oracle.h -- TEST INFRASTRUCTURE ONLY (parity oracle).
 *
 * Plain-C, single-threaded CPU restatement of the extremal-region hot path of
 * HsiehYiChia/Scene-text-recognition (compute_channels -> er_tree_extract ->
 * non_maximum_supression -> classify).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (include/str_er.h, libstr_er_hip.so) never links, imports or calls it.
 *
 * PARITY STATUS
 *   - cascade scoring (ero_cascade_*): PINNED against the real reference
 *     (src/adaboost.cpp compiled unmodified into oracle/_ref, see Makefile).
 *   - component tree / NMS / ARAN+LBP / YCrCb: "parity unpinned".  The
 *     reference's src/ER.cpp and src/OCR.cpp need OpenCV, which this image does
 *     not have, and the reference ships no tests or golden vectors
 *     (SURVEY.md section 4).  The restatement follows the cited lines and is
 *     checked against the known-answer runs recorded in SURVEY.md Appendix B
 *     and against an independent brute-force component-tree labelling.
 *   - four OpenCV primitives are restated from OpenCV 4.x semantics
 *     (Mat::convertTo rounding, resize INTER_LINEAR 8UC1, cvtColor BGR2YCrCb);
 *     each is a named function below so it can be re-pinned later.
 */
#ifndef ER_ORACLE_H
#define ER_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One node of the (pruned) component tree, flat-array form of the reference's
 * struct ER (inc/ER.h:42-80).  Indices are positions in the node array. */
typedef struct ero_node {
    int32_t level;        /* quantised grey level t                           */
    int32_t area;         /* reference "quirk" area: |C| + #nodes in subtree  */
    int32_t x, y, w, h;   /* cv::Rect bound                                   */
    int32_t parent;       /* index of parent, -1 for the root                 */
    int32_t child;        /* first child (reference list order), -1 if none   */
    int32_t next;         /* next sibling, -1 if none                         */
    int32_t key;          /* canonical id: min linear index of the pixels of
                             level == `level` inside the component            */
    int32_t npix;         /* |C| (true pixel count), for diagnostics          */
    int32_t nsub;         /* number of tree nodes in the subtree incl. itself,
                             pruned ones included                             */
} ero_node;

typedef struct ero_tree {
    ero_node *nodes;
    int32_t   n_nodes;       /* kept nodes                                    */
    int32_t   root;          /* index of the root in nodes[]                  */
    int64_t   n_created;     /* every node the flood created (pruned incl.)   */
    int32_t   dead_branch;   /* times the reference's dead re-parenting branch
                                (src/ER.cpp:169-178) would have run           */
} ero_tree;

typedef struct ero_params {
    int32_t thresh_step;     /* THRESH_STEP   (inc/utils.h:6, default 8)      */
    int32_t min_area;        /* MIN_AREA      (120)                           */
    int32_t max_area;        /* MAX_AREA      (900000)                        */
    int32_t stability_t;     /* STABILITY_T   (2)                             */
    double  overlap_coef;    /* OVERLAP_COEF  (0.7)                           */
} ero_params;

/* ---- OpenCV primitives restated (unpinned) ------------------------------ */
/* Mat /= step on 8U: saturate_cast<uchar>(cvRound(float(p)*float(1/step))),
 * round-half-to-even (src/ER.cpp:250). */
void ero_quant_lut(int step, uint8_t lut[256]);
/* highest_level = 255/step + 1 (src/ER.cpp:247). */
int  ero_highest_level(int step);
/* cv::resize(src,dst,Size(dw,dh)) INTER_LINEAR on 8UC1 (src/OCR.cpp:401). */
void ero_resize_linear_u8(const uint8_t *src, int sstride, int sw, int sh,
                          uint8_t *dst, int dstride, int dw, int dh);
/* cv::cvtColor(BGR2YCrCb)+split+inversions (src/ER.cpp:114-128).
 * planes[k] for k in Y,Cr,Cb,255-Y,255-Cr,255-Cb; each w*h contiguous. */
void ero_compute_channels(const uint8_t *bgr, int stride, int w, int h,
                          uint8_t *planes6);

/* NV12 ingest (build-defined, no reference counterpart): planes3 = [Y, Cr, Cb]; chroma replicated over its 2x2 block. */
void ero_nv12_to_ycrcb(const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride, int w, int h, uint8_t *planes3);

/* ---- component tree (src/ER.cpp:240-413) -------------------------------- */
/* Nister-Stewenius flood exactly as the reference runs it, including the
 * level-`highest_level` sentinel behaviour and on-merge pruning.            */
int  ero_tree_extract(const uint8_t *img, int stride, int w, int h,
                      int thresh_step, int min_area, ero_tree *out);
void ero_tree_free(ero_tree *t);

/* Independent brute-force labelling of the canonical node set:
 * same output type, children order = ascending key.                         */
int  ero_tree_bruteforce(const uint8_t *img, int stride, int w, int h,
                         int thresh_step, int min_area, ero_tree *out);

/* ---- NMS (src/ER.cpp:416-505) ------------------------------------------- */
/* sibling_mode 0: reference list order; 1: ascending key; 2: descending key.
 * pool_out: malloc'd array of node indices in visiting order.
 * ambiguous_out (may be NULL): number of nodes at which >=2 child chains
 * competed for the parent (result depends on sibling order there).          */
int  ero_nms(const ero_tree *t, int rows, int cols, const ero_params *p,
             int sibling_mode, int32_t **pool_out, int32_t *n_pool,
             int32_t *ambiguous_out);

/* ---- classify chain (src/ER.cpp:507-528, 789-845; src/OCR.cpp:394-430) --- */
/* ARAN(L=26) of a w*h ROI into a zeroed 26*26 tile. */
void ero_aran26(const uint8_t *roi, int stride, int w, int h, uint8_t tile[26 * 26]);
void ero_aran_dims(int w, int h, int *dw, int *dh);
long ero_selftest_pow_vs_sqrt(int maxdim);
/* calc_LBP on the 26*26 tile -> 24*24 codes (stride-24-on-26 quirk kept). */
void ero_lbp24(const uint8_t tile[26 * 26], uint8_t lbp[24 * 24]);
/* make_LBP_hist: 2x2 blocks x 256 bins, counts as doubles. */
void ero_lbp_hist(const uint8_t *roi, int stride, int w, int h, double hist[1024]);

typedef struct ero_cascade ero_cascade;
/* CascadeBoost::load_classifier (src/adaboost.cpp:873-951). NULL on failure. */
ero_cascade *ero_cascade_load(const char *path);
void         ero_cascade_free(ero_cascade *c);
int          ero_cascade_n_stages(const ero_cascade *c);
int          ero_cascade_n_stumps(const ero_cascade *c);
/* CascadeBoost::predict REAL branch (src/adaboost.cpp:526-541):
 * returns -DBL_MAX on rejection, else the last stage's score. */
double       ero_cascade_predict(const ero_cascade *c, const double fv[1024]);

/* classify (src/ER.cpp:507-528) for a pool of boxes on one plane.
 * cls[i]: 1 strong, 2 weak, 0 dropped.  s_strong/s_weak: predict() results
 * (s_weak is only evaluated when strong rejected; otherwise set to 0).      */
void ero_classify(const uint8_t *plane, int stride,
                  const int32_t *boxes_xywh, int n,
                  const ero_cascade *strong, const ero_cascade *weak,
                  uint8_t *cls, double *s_strong, double *s_weak);

/* ---- OCR scorer, feature half (config 3) -- "parity unpinned" ------------
 * OCR::chain_run up to the svm call (src/OCR.cpp:67-91); the plain entry points are slope == 0 (no rotation):
 *   threshold(255 - roi, THRESH_OTSU)  ->  ARAN(30)  ->  extract_feature (src/OCR.cpp:144-218):
 *   findContours(RETR_LIST, CHAIN_APPROX_NONE) -> 8 direction bitmaps -> GaussianBlur 7x7 ->
 *   normalize(0,255,MINMAX) -> resize 15x15 -> q[1800] (feature value = q/255.0).
 * cv::threshold/Otsu, cv::findContours (Suzuki-Abe border following), cv::GaussianBlur (8-bit
 * fixed-point kernel 8,28,56,72,56,28,8 / 256, BORDER_REFLECT_101) and cv::normalize are restated
 * from OpenCV 4.x; none of it can be pinned here.                                            */
int  ero_otsu_threshold(const uint8_t *img, int stride, int w, int h, int invert);
/* binary 0/255 image of the inverted ROI, ARAN-normalised to 30x30 (zero padded) */
void ero_ocr_normalise(const uint8_t *roi, int stride, int w, int h, uint8_t img30[30 * 30]);
/* the eight 30x30 direction bitmaps (0/255) of extract_feature, before blurring */
void ero_chain_bitmaps(const uint8_t img30[30 * 30], uint8_t maps[8 * 30 * 30]);
/* the complete feature vector: q[8*15*15] */
void ero_chain_features(const uint8_t *roi, int stride, int w, int h, uint8_t q[1800]);
/* the same with the text line's slope (src/OCR.cpp:73-78, rotate_mat :254-357) */
int  ero_rotate_mat(const uint8_t *src, int w, int h, double rad, int crop, uint8_t **dst, int *dw, int *dh);
void ero_ocr_normalise_slope(const uint8_t *roi, int stride, int w, int h, double slope, uint8_t img30[30 * 30]);
void ero_chain_features_slope(const uint8_t *roi, int stride, int w, int h, double slope, uint8_t q[1800]);

/* ---- build-defined pyramid (no reference counterpart) -- */
/* Level k plane size: (lround(w0*2^(-k/2)), lround(h0*2^(-k/2))), min 1.
 * Level k (k>=1) is DEFINED as ero_resize_linear_u8(level k-1 -> dims k).    */
void ero_pyr_dims(int w0, int h0, int level, int *w, int *h);

#ifdef __cplusplus
}
#endif
#endif /* ER_ORACLE_H */
