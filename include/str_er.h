/*
 * str_er.h -- C ABI of libstr_er_hip.so: the MI355X (gfx950) implementation of the
 * extremal-region scene-text detection hot path of HsiehYiChia/Scene-text-recognition.
 *
 * The boundary is the public surface of the reference's `class ERFilter`
 * (inc/ER.h:110-136) for the per-plane hot loop of ERFilter::text_detect
 * (src/ER.cpp:42-60):
 *
 *     compute_channels -> er_tree_extract -> non_maximum_supression -> classify
 *
 * and, behind further stage flags, what text_detect does with the classified ERs (src/ER.cpp:62-72): calc_color + er_track,
 * er_grouping, the chain-code / SVM scoring of er_ocr; plus a frame-ingest stream (str_er_stream_*).
 *
 * Plain pointers and sizes only; no C++/torch types cross this header.  Every entry
 * point returns 0 (STR_ER_OK) or a negative STR_ER_E* code and never throws.  All
 * per-pixel work runs in hand-written HIP kernels; there is no CPU fallback -- if no
 * gfx950 device is usable, str_er_create fails with STR_ER_EHIP.
 *
 * Paths below are relative to the reference checkout (/root/reference).
 */
#ifndef STR_ER_H
#define STR_ER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: str_er_params::sibling_order 0 means the reference's exact flood order (it was the largest-key rule, now 2);
 *    the library no longer edits the process environment when it is loaded (str_er_runtime_hint).                  */
#define STR_ER_ABI_VERSION 2

/* ---- error codes (reference: loaders print+return false, src/adaboost.cpp:877-881;
 *      CV_Assert throws on non-8UC1, src/ER.cpp:242) ------------------------------ */
#define STR_ER_OK          0
#define STR_ER_EINVAL     (-1)  /* bad argument (NULL, non-positive size, bad mask ...)    */
#define STR_ER_ENOMEM     (-2)  /* host or device allocation failed                         */
#define STR_ER_EHIP       (-3)  /* HIP runtime error / no usable device / kernel failure    */
#define STR_ER_EIO        (-4)  /* classifier file could not be opened                      */
#define STR_ER_EFORMAT    (-5)  /* classifier text could not be parsed                      */
#define STR_ER_ESTATE     (-6)  /* call needs cascades that are not loaded                  */
#define STR_ER_ECAPACITY  (-7)  /* input exceeds the capacity the context was created with  */

/* which cascade: ERFilter::stc / ERFilter::wtc (inc/ER.h:117-118) */
#define STR_ER_CASCADE_STRONG 0
#define STR_ER_CASCADE_WEAK   1

/* where an input buffer lives */
#define STR_ER_MEM_HOST   0
#define STR_ER_MEM_DEVICE 1     /* pointer is HIP device memory on params.device */

/* stage mask for str_er_detect_* (each stage needs the previous ones) */
#define STR_ER_STAGE_EXTRACT  1u   /* er_tree_extract          src/ER.cpp:240-374 */
#define STR_ER_STAGE_NMS      2u   /* non_maximum_supression   src/ER.cpp:416-505 */
#define STR_ER_STAGE_CLASSIFY 4u   /* classify                 src/ER.cpp:507-528 */
#define STR_ER_STAGE_ALL      7u
#define STR_ER_STAGE_OCR      8u   /* config 3: OCR::chain_run (slope 0) on every strong/weak ER; needs an SVM model */
/* output option */
#define STR_ER_WANT_NODES     16u  /* also return the kept-node table of every plane */
/* the rest of text_detect (src/ER.cpp:62-72), BGR frames only */
#define STR_ER_STAGE_TRACK    32u  /* calc_color + ERFilter::er_track on the strong/weak ERs of every image (src/ER.cpp:530-590) */
#define STR_ER_STAGE_GROUP    64u  /* ERFilter::er_grouping(tracked, text, false, false) (src/ER.cpp:612-692); needs STR_ER_STAGE_TRACK */
#define STR_ER_GROUP_INNER_SUP 128u /* ... with inner_sup = true, as text_detect calls it when DO_OCR is defined (src/ER.cpp:69) */
#define STR_ER_GROUP_OVERLAP_SUP 512u /* ... with overlap_sup = true, as video_mode calls it (er_grouping(tracked, text, true, true), src/utils.cpp:196) */
#define STR_ER_STAGE_OCR_LINES 256u /* er_ocr's per-line scoring (src/ER.cpp:695-747): chain_run with the line's slope on every member;
                                      needs STR_ER_STAGE_GROUP + an SVM model */

/* candidate class: which list of text_detect() the ER landed in (src/ER.cpp:516-526) */
#define STR_ER_CLS_POOL   0   /* pooled by NMS, rejected by both cascades */
#define STR_ER_CLS_STRONG 1
#define STR_ER_CLS_WEAK   2

typedef struct str_er_ctx    str_er_ctx;
typedef struct str_er_result str_er_result;

/* Constructor arguments of ERFilter (inc/ER.h:113, src/main.cpp:22, macros
 * inc/utils.h:6-11) plus the build's own batching/pyramid/capacity knobs.          */
typedef struct str_er_params {
    int32_t  thresh_step;     /* THRESH_STEP   default 8                              */
    int32_t  min_area;        /* MIN_AREA      default 120                            */
    int32_t  max_area;        /* MAX_AREA      default 900000                         */
    int32_t  stability_t;     /* STABILITY_T   default 2                              */
    double   overlap_coef;    /* OVERLAP_COEF  default 0.7                            */
    int32_t  n_pyr_levels;    /* 1 = native resolution only (the reference);
                                 level k>=1 of Y/Cr/Cb is resize_linear(level k-1)
                                 to (lround(w*2^(-k/2)), lround(h*2^(-k/2)));
                                 inverted channels are 255 - that level               */
    uint32_t channel_mask;    /* bit i = plane i of [Y,Cr,Cb,255-Y,255-Cr,255-Cb]
                                 (src/ER.cpp:122-127); 0x3F = the reference           */
    int32_t  device;          /* HIP device ordinal                                   */
    int32_t  max_width;       /* capacity: largest frame / plane width                */
    int32_t  max_height;      /* capacity: largest frame / plane height               */
    int32_t  max_frames;      /* capacity: frames per str_er_detect_bgr call; the
                                 per-plane entry points accept up to
                                 max_frames * popcount(channel_mask) * n_pyr_levels   */
    int32_t  kept_cap;        /* per-plane capacity of the kept-node table.  0 (both
                                 this and pool_cap) = by the plane's size: padded
                                 pixels / 64 + 512, at most max(4096, w*h/64)         */
    int32_t  pool_cap;        /* per-plane capacity of the NMS pool, 0 = kept_cap/4
                                 (at least 256); a plane that needs more fails with
                                 STR_ER_ECAPACITY, the message names the plane        */
    int32_t  sibling_order;   /* what decides NMS where two or more child chains
                                 compete for a parent:
                                 0 = the reference's own order -- the child whose
                                     basin its flood entered last, src/ER.cpp:183-185,
                                     416-462; found by replaying the flood on the GPU
                                     for the planes that have such a tie (exact);
                                 1 = child with the smallest key, 2 = largest key
                                     (canonical rules, no replay; the plane is only
                                     flagged through str_er_plane_info::ambiguous)    */
    void    *stream;          /* hipStream_t to enqueue on; NULL = private stream     */
} str_er_params;

/* One kept node of a plane's component tree: flat form of struct ER (inc/ER.h:42-80). */
typedef struct str_er_node {
    uint32_t key;        /* canonical id: min linear pixel index (y*w+x) over the pixels
                            of level == `level` inside the component                   */
    int32_t  parent;     /* index in this plane's node table; the root points to itself
                            (non_maximum_supression sets root->parent=root, :424)      */
    int32_t  area;       /* ER::area as the reference computes it: pixel count +
                            number of tree nodes in the subtree (ctor starts at 1)     */
    uint16_t x, y, w, h; /* ER::bound                                                  */
    uint8_t  level;      /* ER::level (quantised grey level)                           */
    uint8_t  flags;      /* bit0: root                                                 */
    uint16_t reserved;
} str_er_node;           /* 24 bytes */

/* One NMS survivor (member of `pool`, and of `strong`/`weak` if cls says so). */
typedef struct str_er_cand {
    uint32_t frame;      /* frame index inside the call (0 for the per-plane calls)    */
    uint8_t  ch;         /* plane index 0..5 as in src/ER.cpp:122-127 (ER::ch)         */
    uint8_t  pyr;        /* pyramid level                                              */
    uint8_t  level;      /* ER::level                                                  */
    uint8_t  cls;        /* STR_ER_CLS_*                                               */
    uint16_t x, y, w, h; /* ER::bound in the plane's own coordinates                   */
    uint32_t area;       /* ER::area (reference semantics)                             */
    uint32_t key;        /* canonical id, see str_er_node                              */
    int32_t  node;       /* index in the plane's kept-node table                       */
    uint32_t plane;      /* plane index inside the result                              */
    double   score_strong; /* stc->predict(fv): last stage score or -DBL_MAX           */
    double   score_weak;   /* wtc->predict(fv) if the strong cascade rejected, else 0  */
} str_er_cand;           /* 48 bytes */

typedef struct str_er_plane_info {
    uint32_t frame;
    uint8_t  ch, pyr, reserved0, reserved1;
    int32_t  width, height;
    int32_t  n_created;   /* tree nodes before pruning (all (t,C) pairs)              */
    int32_t  n_kept;      /* nodes with area > MIN_AREA, plus the root                */
    int32_t  n_pool, n_strong, n_weak;
    int32_t  ambiguous;   /* informational: #nodes where >=2 child chains competed
                             in the first NMS pass (the reference's answer depends on
                             its flood's sibling order there).  With
                             sibling_order = 0 those ties were decided exactly by a
                             replay of the reference's flood; 0 = no tie at all       */
    int32_t  root;        /* index of the root in the kept-node table                 */
} str_er_plane_info;

/* STR_ER_STAGE_TRACK / STR_ER_STAGE_GROUP records (described at str_er_result_tracks / str_er_result_texts) */
typedef struct str_er_track {
    double   color1, color2, color3;     /* NaN where the Otsu mask is empty (0.0 / 0 in calc_color) */
    int32_t  cx, cy;
    uint32_t tracked;
    uint32_t reserved;
} str_er_track;
typedef struct str_er_text {
    uint32_t frame;
    uint8_t  pyr, reserved0, reserved1, reserved2;
    int32_t  first, count;
    double   slope;
    int32_t  x, y, w, h;
} str_er_text;
typedef struct str_er_gbound {
    int32_t x, y, w, h, cx, cy;
} str_er_gbound;

/* ---- lifetime --------------------------------------------------------------------- */
/* Fills the reference's own defaults (src/main.cpp:22): 8,120,900000,2,0.7; six
 * planes, one level; capacity 1920x1080x8 frames.                                     */
void str_er_default_params(str_er_params *p);
int  str_er_create(const str_er_params *p, str_er_ctx **out);
void str_er_destroy(str_er_ctx *ctx);
/* Text of the last error on this context ("" if none). ctx may be NULL for create errors. */
const char *str_er_last_error(const str_er_ctx *ctx);
const char *str_er_strerror(int code);
int  str_er_abi_version(void);
/* Settings of the HIP runtime this library works best with, as "NAME=value" (space separated if several): a context uses
 * two HIP streams (three until round 6, and with STR_ER_PRIO_STREAM) and hosts keep several contexts in flight, which the runtime's default of 4 hardware queues serialises
 * (about 8 % in bench.py).  The runtime reads them when it initialises, so they must be in the environment before the
 * process's first HIP call.  The library never sets them by itself; str_er_apply_runtime_hint() does, for a host that
 * opts in: returns 1 if it set something, 0 if the host's environment already decides, < 0 on error.                  */
/* Exact NMS sibling ties (sibling_order = 0): how many planes of this context's calls so far needed the reference's flood order
 * walked on a host core, the host time those walks took in all (ms, summed over planes), and how many host threads the
 * library's process-wide pool for them has at most (the CPUs the process may use -- hardware threads cut down to a container's cgroup CPU quota --, at least 1,
 * at most 64; STR_ER_WALK_THREADS overrides).  Any pointer may be NULL. */
/* (The pool's threads are detached and live as long as the process: do not dlclose() the library once a tie has been walked.) */
int  str_er_tie_stats(const str_er_ctx *ctx, uint64_t *planes_walked, double *walk_ms_total, int32_t *host_threads);
/* What the component-tree passes of this context's last detect call worked on: node records the tile kernel exported (32 bytes each; what
 * k_group_merge / k_resolve / k_reduce read and write), pixel pairs across tile borders (k_seam: two 16-bit seam entries each) and tiles.
 * Measurement aid (bench.py prices the passes against the HBM roofline with it); the reference has no counterpart.  Any pointer may be NULL. */
int  str_er_last_tree_stats(const str_er_ctx *ctx, uint64_t *records, uint64_t *seam_pairs, uint64_t *tiles);
/* The tile trees of the chroma planes (few levels per tile) are built by a second tile kernel, k_tile_tree2 (level by level on bit masks); a tile with
 * more levels / nodes than it takes is handed back to k_tile_tree.  Since the context was created: tiles given to k_tile_tree2, tiles it handed
 * back.  Measurement aid; results do not depend on which kernel built a tile's tree.  Any pointer may be NULL. */
int  str_er_tile2_stats(const str_er_ctx *ctx, uint64_t *tiles, uint64_t *handed_back);
/* STR_ER_STAGE_OCR is enqueued right behind classify, sized from the previous batch of the context and working on the device's own count of strong / weak
 * ERs (the reference's call site, src/ER.cpp:728-735, has no barrier between the two either); a batch with more ERs than guessed, or whose candidates an NMS
 * tie pass re-made, is scored again after the counters were read.  Since the context was created: batches whose early scores were used, batches scored again.
 * Measurement aid; results do not depend on it.  Any pointer may be NULL. */
int  str_er_ocr_stage_stats(const str_er_ctx *ctx, uint64_t *scored_early, uint64_t *scored_again);
const char *str_er_runtime_hint(void);
int  str_er_apply_runtime_hint(void);
/* Several contexts of a process keep batches in flight on one GPU (bench.py: six).  The GPU shares itself evenly among their kernels, so equal batches
 * that started together finish together -- and then all wait together for whatever their host side does next (the flood order walk of an NMS sibling tie,
 * about 5 ms for a 1080p plane: src/ER.cpp:416-505 depends on the flood's order), with the GPU idle meanwhile: a trace of six 48-frame batches in flight
 * showed no kernel running for 14 % of the time.  With n > 0, at most n detect calls of the process have their batch's kernels on the GPU at a time: a
 * call takes a slot before it enqueues and gives it back when its kernels are done, BEFORE the host-side work that follows, so the waiting calls' kernels
 * run during that work.  Calls of a frame or two (<= 96 planes) do not take part.  n = 0 (the default): no limit.  Process-wide; returns the old value.
 * A scheduling aid only: results do not depend on it.  (No reference counterpart: text_detect is one frame at a time.) */
int  str_er_set_batch_slots(int n);

/* ERFilter::set_thresh_step / set_min_area (src/ER.cpp:21-30) */
int str_er_set_thresh_step(str_er_ctx *ctx, int32_t t);
/* MIN_OCR_PROB, the last constructor argument of ERFilter (inc/ER.h:113; src/main.cpp:22 passes 0.15, the default here) */
int str_er_set_min_ocr_prob(str_er_ctx *ctx, double min_ocr_prob);
int str_er_set_min_area(str_er_ctx *ctx, int32_t m);

/* ---- models: CascadeBoost::load_classifier (src/adaboost.cpp:873-951) --------------- */
int str_er_load_cascade(str_er_ctx *ctx, int which, const char *path);
int str_er_load_cascade_mem(str_er_ctx *ctx, int which, const char *text, size_t len);
/* n_stages / n_stumps of a loaded cascade (0 if not loaded) */
int str_er_cascade_info(const str_er_ctx *ctx, int which, int32_t *n_stages, int32_t *n_stumps);

/* ---- the hot path ------------------------------------------------------------------ */
/* Whole loop of ERFilter::text_detect up to and including classify
 * (src/ER.cpp:39-60) for n_frames interleaved-BGR 8UC3 frames of w*h pixels
 * (stride = bytes per row, frame_pitch = bytes between frames): compute_channels,
 * optional pyramid, then per plane extract -> NMS -> classify.                        */
int str_er_detect_bgr(str_er_ctx *ctx, const uint8_t *bgr, int32_t w, int32_t h,
                      int64_t stride, int64_t frame_pitch, int32_t n_frames,
                      int mem_kind, uint32_t stages, str_er_result **out);

/* The same for NV12 frames -- what a video decoder delivers: a luma plane of h rows, then one interleaved chroma plane
 * (Cb, Cr, Cb, Cr ...) of h / 2 rows, `stride` bytes per row both; w and h even.  The reference has no such input (it decodes to
 * BGR: cv::imread / `cap >> frame`, src/utils.cpp:31, 109); the conversion is BUILD-DEFINED like the pyramid: Y = the luma byte,
 * Cr(x, y) = V(x/2, y/2), Cb(x, y) = U(x/2, y/2) -- the decoder's samples are the channel values, chroma replicated over its 2 x 2
 * block (oracle: ero_nv12_to_ycrcb).  Everything after the three planes is the path of str_er_detect_bgr.  Half the bytes of a BGR
 * frame cross the host link.                                                                                                     */
int str_er_detect_nv12(str_er_ctx *ctx, const uint8_t *nv12, int32_t w, int32_t h,
                       int64_t stride, int64_t frame_pitch, int32_t n_frames,
                       int mem_kind, uint32_t stages, str_er_result **out);

/* The same for a SUBSET of the logical planes of every frame: plane_select[level * n_channels + k] != 0 selects the
 * k-th channel of the context's channel_mask at that pyramid level (n_select = n_pyr_levels * popcount(channel_mask)).
 * Channels and the pyramid are built on the device as far as the deepest selected level; the result holds the selected
 * planes only, in the usual order.  This is how one large frame is split over GPUs plane by plane without a
 * host copy of any plane.  Not with STR_ER_STAGE_TRACK / _GROUP (they read every plane of an image).                        */
int str_er_detect_bgr_planes(str_er_ctx *ctx, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch,
                             int32_t n_frames, int mem_kind, uint32_t stages, const uint8_t *plane_select,
                             int32_t n_select, str_er_result **out);

/* The loop body (src/ER.cpp:52-59) for n_planes independent 8UC1 planes of w*h
 * pixels (plane_pitch = bytes between planes).  This is what the reference's direct
 * callers use (src/utils.cpp:680-684, 716-720, 763-769, 940-947, 1386-1388).        */
int str_er_detect_planes(str_er_ctx *ctx, const uint8_t *planes, int32_t w, int32_t h,
                         int64_t stride, int64_t plane_pitch, int32_t n_planes,
                         int mem_kind, uint32_t stages, str_er_result **out);

/* ---- single stages, one per remaining ERFilter method ------------------------------ */
/* ERFilter::compute_channels (src/ER.cpp:114-128): planes6 receives the six w*h
 * planes [Y,Cr,Cb,255-Y,255-Cr,255-Cb], each tightly packed (host memory).           */
int str_er_compute_channels(str_er_ctx *ctx, const uint8_t *bgr, int32_t w, int32_t h,
                            int64_t stride, uint8_t *planes6);

/* ERFilter::classify (src/ER.cpp:507-528) on caller-supplied boxes of one host plane:
 * boxes_xywh[4*i..4*i+3] = ER::bound.  cls/score arrays receive n entries.           */
int str_er_classify_boxes(str_er_ctx *ctx, const uint8_t *plane, int32_t w, int32_t h,
                          int64_t stride, const int32_t *boxes_xywh, int32_t n,
                          uint8_t *cls, double *score_strong, double *score_weak);

/* ERFilter::make_LBP_hist(input, 2, 24) (src/ER.cpp:789-816) for n boxes of one host
 * plane: hist receives n*1024 doubles; tiles26 (optional, may be NULL) receives the
 * n ARAN(26) tiles (src/OCR.cpp:394-430), 676 bytes each.                            */
int str_er_lbp_hist(str_er_ctx *ctx, const uint8_t *plane, int32_t w, int32_t h,
                    int64_t stride, const int32_t *boxes_xywh, int32_t n,
                    double *hist, uint8_t *tiles26);

/* ERFilter::calc_LBP(input, 24) (inc/ER.h:134, src/ER.cpp:819-845; caller OCR::lbp_run,
 * src/OCR.cpp:37-39) for n boxes of one host plane: lbp24 receives the n 24x24 Mean-LBP
 * code maps, 576 bytes each, row-major -- the Mat the reference returns, stride-24-on-26
 * addressing included.                                                                */
int str_er_calc_lbp(str_er_ctx *ctx, const uint8_t *plane, int32_t w, int32_t h,
                    int64_t stride, const int32_t *boxes_xywh, int32_t n, uint8_t *lbp24);

/* AdaBoost::predict(vector<double> fv) (inc/adaboost.h:131; CascadeBoost::predict,
 * src/adaboost.cpp:507-542) for n caller-supplied 1024-element feature vectors:
 * out[i] = last stage score, or -DBL_MAX if a stage rejected.                       */
int str_er_cascade_predict(str_er_ctx *ctx, int which, const double *fv, int32_t n, double *out);

/* ---- OCR scorer, SVM half (config 3) --------------------------------------
 * svm_load_model (src/svm.cpp:2767-2982; OCR::OCR, src/OCR.cpp:19-22): a libsvm C-SVC / RBF text model
 * with probability information (probA/probB).  dim = feature dimension (1800 = 8 x 15 x 15 for the
 * reference's chain-code features, src/OCR.cpp:203-216); must exceed the largest SV index.  2 <= nr_class <= 125
 * (the reference's model has 65): STR_ER_EFORMAT otherwise.                                                     */
int str_er_load_svm_model(str_er_ctx *ctx, const char *path, int32_t dim);
int str_er_load_svm_model_mem(str_er_ctx *ctx, const char *text, size_t len, int32_t dim);
int str_er_svm_info(const str_er_ctx *ctx, int32_t *nr_class, int32_t *total_sv, int32_t *dim);
/* How the loaded model's kernel matrix is computed for vectors that come from boxes (chain_run, STR_ER_STAGE_OCR / _OCR_LINES; measurement aid, any pointer
 * may be NULL): *bytes = 1 if the support vectors are 8-bit numerators over 255 -- as the reference's are, being feature vectors of its training set
 * (src/OCR.cpp:211) -- and |x - sv|^2 is an exact integer from 8-bit matrix instructions, 0 if each f32 value goes as three bf16 pieces; *class_sums = 1 if
 * the decision values are summed per class as dense f64 products (models with more than 8 support vectors a class), 0 if per vector. */
int str_er_svm_forms(const str_er_ctx *ctx, int32_t *bytes, int32_t *class_sums);
/* svm_predict_probability (inc/svm.h:88, src/svm.cpp:2592-2629) for n dense feature vectors x[n][dim]
 * (zeros = absent svm_nodes): label[i] = model->label[argmax], prob[i][nr_class]; dec (optional, may
 * be NULL) receives the nr_class*(nr_class-1)/2 decision values of svm_predict_values.              */
int str_er_svm_predict_probability(str_er_ctx *ctx, const double *x, int32_t n, int32_t dim, int32_t *label, double *prob,
                                   double *dec);

/* OCR::chain_run(Mat src, int thresh, double slope) (src/OCR.cpp:67-140) for n boxes (ER::bound) of one host
 * plane (the channel the ER came from), with slope == 0: Otsu-binarise 255-roi, ARAN(30), chain-code
 * features (src/OCR.cpp:144-218), then
 * svm_predict_probability.  label[i] = class label, prob[i] = pv[label] (chain_run returns
 * table[label] + prob); q_out (optional) receives the 1800 feature bytes of every box (value = q/255);
 * label/prob may be NULL to get the features only (no SVM model needed then).                        */
int str_er_ocr_chain_run(str_er_ctx *ctx, const uint8_t *plane, int32_t w, int32_t h, int64_t stride, const int32_t *boxes_xywh,
                         int32_t n, int32_t *label, double *prob, uint8_t *q_out);

/* calc_color(ER*, Mat mask_channel, Mat color_img) (src/ER.cpp:1391-1419) for n boxes (ER::bound) of one host
 * mask plane and one host 3-byte interleaved colour image (the Ycrcb Mat): colors[i][0..2] = ER::color1..3.
 * The colour image is read from its own row 0 / column 0 for every box, as the reference does (:1404).
 * It must be at least as large as the largest box.                                                    */
int str_er_calc_color(str_er_ctx *ctx, const uint8_t *mask_plane, int32_t w, int32_t h, int64_t stride, const uint8_t *color_img,
                      int32_t cw, int32_t ch, int64_t cstride, const int32_t *boxes_xywh, int32_t n, double *colors);

/* ERFilter::er_track (src/ER.cpp:530-590) on caller-supplied ERs of ONE image: cands[i].{x,y,w,h,area,cls}
 * (cls 1 = from strong[], 2 = from weak[], others ignored) and colors[i][3] (from str_er_calc_color);
 * tracked[i] = 1 for the members of all_er, cx/cy (optional) receive ER::center.                     */
int str_er_er_track(str_er_ctx *ctx, const str_er_cand *cands, const double *colors, int32_t n, uint8_t *tracked, int32_t *cx,
                    int32_t *cy);

/* ERFilter::er_grouping(all_er, text, false, inner_sup) (src/ER.cpp:612-692) on caller-supplied ERs of ONE image:
 * cands[i].{x,y,w,h,area} and tracks[i].{color1-3, cx, cy, tracked} (all_er = the ones with tracked != 0).  The
 * result holds copies of cands and tracks plus the lines (str_er_result_texts / _text_ers / _group_bounds);
 * free it with str_er_result_free.  overlap_sup = true (video_mode without DO_OCR asks for it,
 * src/utils.cpp:196): sort + overlap_suppression (src/ER.cpp:614-617, sequential -- a merge rewrites the survivor's box) run
 * on the host, the survivors then take the same GPU stages; the merged boxes are in str_er_result_gbounds, the surviving
 * list (all_er as the reference leaves it) in str_er_result_group_all.                                    */
int str_er_er_grouping(str_er_ctx *ctx, const str_er_cand *cands, const str_er_track *tracks, int32_t n, int overlap_sup,
                       int inner_sup, str_er_result **out);

/* The same with the text line's slope per box (Text::slope, src/ER.cpp:731): where |slope[i]| > 0.01 the
 * binarised ROI goes through OCR::rotate_mat(atan2(slope, 1), crop = true) (src/OCR.cpp:73-78, 254-357)
 * before ARAN.  slope == NULL means all zero.  A non-finite slope is STR_ER_EINVAL.                   */
int str_er_ocr_chain_run_slope(str_er_ctx *ctx, const uint8_t *plane, int32_t w, int32_t h, int64_t stride,
                               const int32_t *boxes_xywh, const double *slope, int32_t n, int32_t *label, double *prob,
                               uint8_t *q_out);

/* ERFilter::non_maximum_supression (src/ER.cpp:416-505) on a caller-supplied kept tree
 * (parent indices; root points to itself or -1).  pool_idx receives up to cap node
 * indices in ascending key order; *n_pool the count; *ambiguous as in plane_info.
 * Sibling ties with sibling_order = 0: the TABLE ORDER is the child-list order -- of
 * the children of one parent the one listed first is the first of ER::child / ER::next
 * (src/ER.cpp:183-185), i.e. the one the reference's post-order walk visits first.     */
int str_er_nms_tree(str_er_ctx *ctx, const str_er_node *nodes, int32_t n_nodes,
                    int32_t rows, int32_t cols, int32_t *pool_idx, int32_t cap,
                    int32_t *n_pool, int32_t *ambiguous);

/* The same for a table that came from str_er_detect_planes / er_tree_extract (ordered
 * by key, which says nothing about the flood): the plane itself is passed -- as the
 * reference's signature does (`Mat input`, src/ER.cpp:416) -- and sibling ties are
 * decided by replaying the reference's flood on it (sibling_order = 0).  node.key
 * must be the canonical key (min pixel index of the node's own level).                */
int str_er_nms_tree_plane(str_er_ctx *ctx, const str_er_node *nodes, int32_t n_nodes,
                          const uint8_t *plane, int32_t cols, int32_t rows, int64_t stride,
                          int32_t *pool_idx, int32_t cap, int32_t *n_pool, int32_t *ambiguous);

/* The order in which the reference's flood (er_tree_extract, src/ER.cpp:240-374) first reaches the pixels
 * of a host plane: stamp[y*w+x] = 1-based position, 0 = never reached (sealed off by sentinel-level
 * pixels).  This is the walk that decides NMS sibling ties (sibling_order = 0); it runs on the calling
 * thread, needs no context and no GPU, and is exported so that the tie-break can be checked on its own. */
int str_er_flood_order(const uint8_t *plane, int32_t w, int32_t h, int64_t stride, int32_t thresh_step, uint32_t *stamp);

/* Build-defined pyramid primitive (no reference counterpart): fixed-point bilinear
 * resize of one host plane, same arithmetic as cv::resize INTER_LINEAR 8UC1.         */
int str_er_resize_plane(str_er_ctx *ctx, const uint8_t *src, int32_t sw, int32_t sh,
                        int64_t sstride, uint8_t *dst, int32_t dw, int32_t dh);

/* ---- results (owned by the library until str_er_result_free) ----------------------- */
int32_t str_er_result_n_planes(const str_er_result *r);
int     str_er_result_plane_info(const str_er_result *r, int32_t plane, str_er_plane_info *info);
/* All plane records of the call as one array (n = str_er_result_n_planes). */
const str_er_plane_info *str_er_result_plane_infos(const str_er_result *r, int32_t *n);
/* All candidates of the call, ordered by (plane, key). */
const str_er_cand *str_er_result_cands(const str_er_result *r, int32_t *n);
/* Candidates of one plane (a slice of the array above). */
const str_er_cand *str_er_result_plane_cands(const str_er_result *r, int32_t plane, int32_t *n);
/* With STR_ER_STAGE_OCR: per candidate of str_er_result_cands(), the class label chosen by
 * svm_predict_probability and its probability (what OCR::chain_run returns as table[label] + prob,
 * src/OCR.cpp:139); label -1 / prob 0 for candidates that are neither strong nor weak.  NULL otherwise. */
const int32_t *str_er_result_ocr_labels(const str_er_result *r, int32_t *n);
const double  *str_er_result_ocr_probs(const str_er_result *r, int32_t *n);
/* With STR_ER_STAGE_TRACK: per candidate of str_er_result_cands(), what er_track leaves on the ER
 * (ER::color1-3, ER::center) and whether it is in `tracked` (all_er of src/ER.cpp:530).  An image is
 * one frame at one pyramid level (the reference has only level 0): its strong ERs and every weak ER the
 * rule at :575-587 ties, directly or through other tracked ERs, to one of them.  all_er's ORDER (strong
 * lists, then weak ones as found) is not reproduced -- er_grouping sorts it first thing (:614).
 * Records of pool-only candidates (cls 0) are all zero.  NULL without the stage.                   */
const str_er_track *str_er_result_tracks(const str_er_result *r, int32_t *n);
/* With STR_ER_STAGE_GROUP: the text lines (`vector<Text>` of src/ER.cpp:612) of every image, images in plane
 * order.  A line's members are str_er_result_text_ers()[first .. first+count): indices into
 * str_er_result_cands(), in the line's order (sorted by center.x); as in the reference an ER can sit in more
 * than one line, and more than once in a line (:650-661 never merges two lines).  slope = fitline_avgslope of
 * the members that survive overlap_ and inner_suppression; box = union of the members' bounds
 * (the reference fills Text::box here only without DO_OCR, :684-690).
 * Where the reference calls std::sort on center.x (:614, :668) this library sorts STABLY, ties in candidate
 * order / current line order: the reference's order among equal center.x is unspecified (unstable sort over a
 * traversal-dependent input order), so lines that hinge on such ties may legitimately differ from it.
 * overlap_suppression rewrites bound and center of the ERs it merges into (:945-955, shared by all lines):
 * str_er_result_group_bounds() has every candidate's bound and center as er_grouping leaves them.        */
const str_er_text   *str_er_result_texts(const str_er_result *r, int32_t *n);
const int32_t       *str_er_result_text_ers(const str_er_result *r, int32_t *n);
const str_er_gbound *str_er_result_group_bounds(const str_er_result *r, int32_t *n);
/* all_er as er_grouping leaves it (sorted by center.x, minus inner_suppression's victims): candidate indices, images concatenated */
const int32_t       *str_er_result_group_all(const str_er_result *r, int32_t *n);
/* With STR_ER_STAGE_OCR_LINES: the first half of ERFilter::er_ocr (src/ER.cpp:695-747) on the lines of
 * str_er_result_texts().  Parallel to str_er_result_text_ers(): label / prob = what OCR::chain_run(channel[er->ch](er->bound),
 * level * THRESH_STEP, text.slope) returns for that member (bound as er_grouping left it), kept = 1 if the member survives the
 * 0.95-overlap deletion (:700-721) and prob >= MIN_OCR_PROB (:737-741).  Per line: alive = at least 2 members kept (:743-747).
 * The word graph, feedback verification and spelling correction that follow in er_ocr are not part of this library.        */
const int32_t *str_er_result_line_labels(const str_er_result *r, int32_t *n);
const double  *str_er_result_line_probs(const str_er_result *r, int32_t *n);
const uint8_t *str_er_result_line_kept(const str_er_result *r, int32_t *n);
const uint8_t *str_er_result_text_alive(const str_er_result *r, int32_t *n);
/* Kept-node table of one plane, ascending (key, level); NULL unless STR_ER_WANT_NODES. */
const str_er_node *str_er_result_plane_nodes(const str_er_result *r, int32_t plane, int32_t *n);
/* times[7] = {extract, nms, classify, track, group, ocr, total} seconds, the contract of
 * ERFilter::text_detect's return value (src/ER.cpp:99-110); track, group and ocr are filled by their stages.
 * extract/nms/classify are GPU stage times for the whole batch (HIP events).          */
const double *str_er_result_times(const str_er_result *r);
/* Device copy of the candidate array (for an RCCL gather without a host round trip):
 * copies min(n, cap) records to dst_dev on the context's stream and synchronises.     */
int  str_er_result_cands_to_device(str_er_ctx *ctx, const str_er_result *r, void *dst_dev,
                                   int32_t cap, int32_t *n);
void str_er_result_free(str_er_result *r);

/* ---- one plane in strips over several GPUs (no reference counterpart) -----------------------
 * The level-0 planes of ONE frame cut into n_strips bands of tile rows.  Every participant calls str_er_strip_extract with the
 * frame and its strip number: compute_channels, the tile trees of the strip and the seams inside it, for every channel of the
 * context; the blob holds the strip's node records and the node of every pixel of its first and last row -- plain bytes, to be
 * brought to the plane's owner by any transport.  The owner calls str_er_strip_merge with the frame and all n_strips blobs (in
 * strip order): records behind one another, seams across the cuts joined, then the usual passes; the result is the one
 * str_er_detect_bgr gives for the level-0 planes of the frame (same planes, same records).  Equal parameters on all participants;
 * a context with a pyramid strips its level-0 planes only (the smaller planes are dealt out whole: str_er_detect_bgr_planes).
 * A blob is checked before it is used (sizes against the headers, rows against the cut, every node id against its strip's record
 * count): a damaged or foreign one gives STR_ER_EFORMAT, never an out-of-range device access.
 *   _extract      *blob is malloc'ed host memory (str_er_strip_free)
 *   _extract_dev  *d_blob is a device buffer of the context, valid until its next strip call: with str_er_comm_allgather_bytes
 *                 (STR_ER_MEM_DEVICE in and out) over an RCCL communicator a blob goes from GPU to GPU without touching a host
 *   _merge        host blobs, all channels;  _merge_ex: blobs in host or device memory (blob_kind), and plane_select (one byte per
 *                 channel of the context, or NULL = all): the channels THIS owner puts together -- different planes of one frame
 *                 can have different owners                                                                                     */
int  str_er_strip_extract(str_er_ctx *ctx, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int mem_kind,
                          int32_t strip, int32_t n_strips, void **blob, int64_t *blob_bytes);
int  str_er_strip_extract_dev(str_er_ctx *ctx, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int mem_kind,
                              int32_t strip, int32_t n_strips, const void **d_blob, int64_t *blob_bytes);
void str_er_strip_free(void *blob);
int  str_er_strip_merge(str_er_ctx *ctx, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int mem_kind,
                        const void *const *blobs, const int64_t *blob_bytes, int32_t n_strips, uint32_t stages,
                        str_er_result **out);
int  str_er_strip_merge_ex(str_er_ctx *ctx, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int mem_kind,
                           const void *const *blobs, const int64_t *blob_bytes, int blob_kind, int32_t n_strips,
                           const uint8_t *plane_select, uint32_t stages, str_er_result **out);

/* ---- multi-GPU: the one exchange of the path ------------------------------------------
 * One process per GPU; frames (or planes) are dealt out to the ranks and only the candidate records travel, where the
 * reference's er_track reads the strong / weak lists of every plane (src/ER.cpp:63):
 *     all_gather(my count) -> all_gather(records padded to the largest count) -> padding dropped, frame offsets added.
 * A communicator is an RCCL communicator (librccl.so is loaded on first use; ncclAllGather over xGMI on a stream of its own)
 * or a member of an in-process group that exchanges through host memory (tests without a GPU; one thread per rank).     */
typedef struct str_er_comm str_er_comm;
typedef struct str_er_comm_group str_er_comm_group;
/* RCCL: rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the others by any means; all ranks then create. */
int  str_er_comm_unique_id(void *id128);
int  str_er_comm_create(int32_t device, int32_t rank, int32_t world, const void *id128, str_er_comm **out);
/* in-process group of `world` ranks */
int  str_er_comm_local_group(int32_t world, str_er_comm_group **out);
int  str_er_comm_create_local(str_er_comm_group *g, int32_t rank, str_er_comm **out);
void str_er_comm_local_group_free(str_er_comm_group *g);
void str_er_comm_destroy(str_er_comm *c);
int32_t str_er_comm_rank(const str_er_comm *c);
int32_t str_er_comm_world(const str_er_comm *c);
const char *str_er_comm_last_error(const str_er_comm *c);
/* Collective: every rank passes its records (host memory) and the number to add to their `frame` field; every rank receives
 * all records ordered by rank (*all, free with str_er_gather_free) and, if counts != NULL, the per-rank counts.            */
int  str_er_gather_cands(str_er_comm *c, const str_er_cand *local, int32_t n_local, uint32_t frame_offset,
                         str_er_cand **all, int32_t *n_all, int32_t *counts);
/* The same for the candidates of ctx's last detect call, taken from the device array they are still in: no host hop on
 * the sending side (RCCL communicators only).                                                                             */
int  str_er_gather_last(str_er_comm *c, str_er_ctx *ctx, uint32_t frame_offset, str_er_cand **all, int32_t *n_all,
                        int32_t *counts);
void str_er_gather_free(str_er_cand *p);
/* Collective: a variable-length all-gather of plain bytes (the strip blobs of str_er_strip_extract).  `local` is host or device memory
 * (in_kind).  out_kind STR_ER_MEM_HOST: *all is malloc'ed (str_er_comm_free) and holds the contributions back to back; STR_ER_MEM_DEVICE:
 * *all points into a device buffer of the communicator, valid until its next collective -- with an RCCL communicator and device input
 * the bytes never touch the host.  starts[k] / sizes[k] (world entries each): where rank k's bytes are in *all.
 * A rank whose arguments are bad still joins the exchange of the sizes: every rank then returns an error, none is left waiting.      */
int  str_er_comm_allgather_bytes(str_er_comm *c, const void *local, int64_t n_local, int in_kind, int out_kind,
                                 void **all, int64_t *starts, int64_t *sizes);
void str_er_comm_free(void *p);

/* ---- introspection / measurement --------------------------------------------------- */
/* Per-kernel-group GPU time of the LAST detect call, measured with HIP events on the
 * context's stream.  names[i] are static strings.  Returns the number of groups.      */
int str_er_last_profile(const str_er_ctx *ctx, const char **names, double *ms, int32_t cap);
/* Enable (1) / disable (0) the per-group events above (default 0: two events per call). */
int str_er_set_profiling(str_er_ctx *ctx, int enable);
/* Bytes of device workspace held by the context. */
int64_t str_er_workspace_bytes(const str_er_ctx *ctx);

/* ---- frame ingest ------------------------------------------------------------------
 * The reference takes its frames from the host (cv::imread / `cap >> frame`, src/utils.cpp:31, 59-82, 109) and calls
 * text_detect on each.  A str_er_stream keeps `depth` batches in flight: it owns `depth` contexts (same parameters,
 * each with its own HIP stream and workspace), one page-locked staging buffer per context and a worker thread per
 * context, so the upload of one batch overlaps the kernels of the others.  Producer loop:
 *
 *     str_er_stream_acquire(s, &slot, &buf, &cap);     // a pinned buffer of max_frames * max_width * max_height * 3 bytes
 *     ... decode / copy up to max_frames BGR frames into buf ...
 *     str_er_stream_submit(s, slot, w, h, stride, frame_pitch, n_frames, stages, &ticket);
 *     if (str_er_stream_pending(s) == depth) str_er_stream_next(s, &result, &ticket);   // oldest first; blocks until it is done
 *
 * acquire never blocks: with every buffer in flight it returns STR_ER_ESTATE (collect a result first).  Results come
 * back in submission order and are freed with str_er_result_free.  Models are loaded into every context with
 * str_er_stream_load_cascade (other per-context calls: str_er_stream_context).  One producer/consumer thread at a time. */
typedef struct str_er_stream str_er_stream;
int         str_er_stream_create(const str_er_params *p, int32_t depth, str_er_stream **out);
void        str_er_stream_destroy(str_er_stream *s);
int32_t     str_er_stream_depth(const str_er_stream *s);
str_er_ctx *str_er_stream_context(str_er_stream *s, int32_t i);
const char *str_er_stream_last_error(const str_er_stream *s);
int         str_er_stream_load_cascade(str_er_stream *s, int which, const char *path);
int         str_er_stream_acquire(str_er_stream *s, int32_t *slot, uint8_t **buffer, int64_t *capacity);
int         str_er_stream_submit(str_er_stream *s, int32_t slot, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch,
                                 int32_t n_frames, uint32_t stages, uint64_t *ticket);
/* ... the staging buffer holds NV12 frames (str_er_detect_nv12): stride >= w, a frame is h + h/2 rows */
int         str_er_stream_submit_nv12(str_er_stream *s, int32_t slot, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch,
                                      int32_t n_frames, uint32_t stages, uint64_t *ticket);
/* convenience: acquire + copy the frames in (one extra host copy) + submit */
int         str_er_stream_submit_copy(str_er_stream *s, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch,
                                      int32_t n_frames, uint32_t stages, uint64_t *ticket);
int         str_er_stream_next(str_er_stream *s, str_er_result **out, uint64_t *ticket);
int32_t     str_er_stream_pending(str_er_stream *s);

#ifdef __cplusplus
}
#endif
#endif /* STR_ER_H */
