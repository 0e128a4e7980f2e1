// ocr_kernels.h -- device model + launchers of the OCR scorer of config 3 (ocr_kernels.hip):
// OCR::chain_run (src/OCR.cpp:67-140) = chain-code features + svm_predict_probability (src/svm.cpp:2592-2629).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "er_kernels.h"

namespace str_er {

// a class pair i < j as k_svm_couple reads it: first support vector and count of both classes
struct alignas(16) SvmPair { int32_t sa, na, sb, nb, ci, cj, pad0, pad1; };

struct SvmDev {
    int32_t k, l, l_pad, dim, dpad;
    int32_t kc;             // row length of coef_t: k - 1 rounded up to 64 (64 or 128)
    double  gamma;
    const float  *sv;       // [l_pad x dpad] dense, zero padded
    const uint16_t *svq;    // [3][l_pad x dq] bf16: sv = svq[0] + svq[1] + svq[2] exactly (the f32's 24 significant bits in three pieces)
    int32_t dq;             // dim rounded up to 64: row length of svq and of OcrBuf::xq
    // models whose support vectors are 8-bit numerators over 255 (the reference's are: svm_tables.h svm_sv_bytes), else null
    const uint8_t *sv8;     // [l_pad x dq8] numerator ^ 0x80, padding 0x80
    const int32_t *sv8s;    // [l_pad x 2] sum of the numerators, sum of their squares
    int32_t dq8;            // dim rounded up to 128: row length of sv8 and of OcrBuf::x8
    const double *svnorm;   // [l_pad]
    const double *coef;     // [(k-1) x l]            sv_coef as libsvm stores it (kept for the layout tests)
    const double *coef_t;   // [l_pad x kc]              coef_t[q][b] = sv_coef[b][q], zero padded: one coalesced row per support vector
    const double *rho, *probA, *probB;   // [k(k-1)/2]
    const int32_t *label, *nsv, *start;  // [k]
    // k <= 65 only: coef_rows[i][h][r][b], r < mp, b < 64 -- h = 0: sv_coef[b][start[i] + r] (class i's r-th support vector against class b + 1), zero for
    // r >= nr_sv[i]; h = 1: sv_coef[i][start[b + 1] + r] (class b + 1's r-th support vector against class i), zero for r >= nr_sv[b + 1].  msv = the largest
    // nr_sv, mp = svm_rows_per_class(msv)
    const double *coef_rows;
    int32_t msv, mp;
    const SvmPair *pairs;                // [k(k-1)/2 rounded up to 512]  every class pair i < j, in libsvm's order
};

// Where the boxes of a call come from: explicit boxes on one device plane (single-stage API), or records
// (box = recs[list[i]].x,y,w,h on plane planes[recs[list[i]].plane]); rot (optional) = OCR::rotate_mat per box.
struct OcrSrc {
    const uint8_t   *plane;
    int32_t          stride, inv;
    const int32_t   *boxes;
    const CandRec   *recs;
    const uint32_t  *list;
    const PlaneDesc *planes;
    const RotGeom   *rot;
    // null, or the number of boxes as the DEVICE knows it (k_ocr_list's count): the launches are then sized for the `n` handed to the launcher, an upper
    // bound the host guessed, and the kernels work on min(n, *n_dev) boxes -- no read of the counters between classify and the scorer (str_er_api.cpp)
    const uint32_t  *n_dev;
};

// Scratch of one scoring call (all device pointers; carve them out of one allocation with ocr_layout()).
struct OcrBuf {
    uint32_t *hist;      // [n x 256]       histogram of 255 - roi                       (features)
    int32_t  *thresh;    // [n]             Otsu threshold                               (features)
    uint32_t *big;       // [1 + 4095]      count, then the boxes whose histogram is spread over many workgroups (features)
    uint8_t  *q;         // [n x 1800] or null: the features as the reference's 8-bit image (q / 255.0 = the svm input)
    float    *xf;        // [n_pad x dpad]  svm input, f32, zero padded                  (svm, vectors given as doubles)
    uint16_t *xq;        // [n_pad x dq]    svm input times 255 -- the features' 8-bit numerators -- as bf16, zero padded (svm, vectors from boxes)
    double   *xnorm;     // [n_pad]         |x|^2                                        (svm)
    uint8_t  *x8;        // [n_pad x dq8] or null: the numerators ^ 0x80, padding 0x80 (svm, models with SvmDev::sv8)
    int32_t  *x8s;       // [n_pad x 2] or null: sum of the numerators, sum of their squares
    double   *kv;        // [n_pad x l_pad] RBF kernel values                            (svm)
    double   *av;        // [n_pad x k x 64] or null: per class c and slot j the sum of sv_coef[j][q] K[q] over class c's support vectors (svm, svm_uses_class_sums())
    double   *dec;       // [n x k(k-1)/2] or null: decision values
    double   *prob;      // [n x k] or null: class probabilities
    int32_t  *label;     // [n]
    double   *pbest;     // [n]             probability of the predicted label (pv[label], src/OCR.cpp:92-93)
    const uint32_t *n_dev; // null, or the device's count of vectors (see OcrSrc::n_dev); set by the caller after ocr_layout()
    size_t    bytes;     // total size of the carve-up
};

// compute units of the device (read once per process, thread-safe)
int ocr_n_cu();

// Models with many support vectors a class (the reference's shape: 120 training samples a class) take the decision values in two steps -- k_svm_decide sums
// coefficient x kernel value per (class, other class) as one dense product per class (f64 MFMA, 64 vectors a wave), k_svm_couple adds the two halves of a
// pair -- instead of k_svm_couple walking the coefficient table per vector
inline bool svm_uses_class_sums(const SvmDev &m) { return m.k <= 65 && m.kc == 64 && m.mp > 8; }

// Offsets are applied to `base` (may be null to size the allocation only: read .bytes).
OcrBuf ocr_layout(uint8_t *base, size_t n, const SvmDev *m /* null: features only */, bool want_q, bool want_dec, bool want_prob);

// indices of the strong / weak candidates of the batch (n_cands = the host's copy of *b.total_cands), in candidate order (deterministic):
// hdr[0] = their number, hdr[16 .. 272) = scratch, the list itself from hdr + OCR_LIST_HDR (room for n_cands entries)
constexpr int OCR_LIST_HDR = 272;
void launch_ocr_list(hipStream_t s, const BatchDev &b, uint32_t n_cands, uint32_t *hdr);
// the same for the candidates named in from[0 .. *from_n) (device memory; in the order of that list): the planes an NMS tie pass re-made
void launch_ocr_list_from(hipStream_t s, const BatchDev &b, const uint32_t *from, const uint32_t *from_n, uint32_t *hdr);

// Otsu threshold of 255 - roi for n boxes (cv::threshold(..., THRESH_OTSU)): hist [n x 256], big [1 + 4095] (scratch), thresh [n]
void launch_box_thresholds(hipStream_t s, const OcrSrc &src, int n, uint32_t *hist, uint32_t *big, int32_t *thresh);

// chain-code features of n boxes: Otsu of 255 - roi, ARAN(30), direction bitmaps, 7x7 Gaussian, min-max, 2x2 decimation
// -> buf.q (if not null) and buf.xq / buf.xnorm (if m is not null)
void launch_ocr_features(hipStream_t s, const OcrSrc &src, int n, const OcrBuf &buf, const SvmDev *m);

// svm_predict_probability for the n rows of buf.xf (numerators = false) or buf.xq (true) and buf.xnorm -> buf.label, buf.pbest (+ buf.prob, buf.dec if not null)
void launch_svm_score(hipStream_t s, int n, const OcrBuf &buf, const SvmDev &m, bool numerators);
// ... its two halves: the RBF kernel matrix buf.kv; decision values + coupling from buf.kv
void launch_svm_kernel(hipStream_t s, int n, const OcrBuf &buf, const SvmDev &m, bool numerators);
void launch_svm_couple(hipStream_t s, int n, const OcrBuf &buf, const SvmDev &m);
// (svm_rows_per_class() and the builders of SvmDev::svq / coef_rows: svm_tables.h)

// f64 feature vectors (API entry str_er_svm_predict_probability) -> buf.xf / buf.xnorm
void launch_svm_prep(hipStream_t s, const double *x, int n, int dim, const OcrBuf &buf, const SvmDev &m);

} // namespace str_er
