// svm_tables.h -- the tables the SVM model loader lays out for k_svm_kernel_q and k_svm_couple (ocr_kernels.h), as plain host functions:
// no HIP in here, so that tests/test_host_cpp.py can check them on a machine without a GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace str_er {

// rows per class of the per-class coefficient rows for a model whose largest class has msv support vectors (the kernel's builds: 5 exactly -- the
// shape of the reference's training set -- or eights)
inline int svm_rows_per_class(int msv) { return msv == 5 ? 5 : (msv + 7) / 8 * 8; }

// An f32 as three bf16 values whose sum is the f32, exactly: the top 8 significant bits, the next 8, the last 8.  Each cut is a truncation of the
// remainder (so the pieces share the sign), each remainder is exact in f32.  (A piece 2^-126 or so below the value's exponent would be subnormal
// and is cut to zero by the same mask: values below 2^-110 lose bits that no dot product of 8-bit numerators can see.)
inline void split_bf16x3(float v, uint16_t out[3])
{
    float r = v;
    for (int pl = 0; pl < 3; ++pl) {
        uint32_t u;
        memcpy(&u, &r, 4);
        u &= 0xFFFF0000u;
        float piece;
        memcpy(&piece, &u, 4);
        out[pl] = (uint16_t)(u >> 16);
        r -= piece;
    }
}

// svq[pl][i][j], i < l_pad, j < dq: piece pl of support vector i's feature j (sv: [l x dpad] f32, zero padded rows)
inline std::vector<uint16_t> svm_sv_planes(const std::vector<float> &sv, int l, int l_pad, int dim, int dpad, int dq)
{
    std::vector<uint16_t> svq((size_t)3 * l_pad * dq, 0);
    for (int i = 0; i < l; ++i)
        for (int j = 0; j < dim; ++j) {
            uint16_t p[3];
            split_bf16x3(sv[(size_t)i * dpad + j], p);
            for (int pl = 0; pl < 3; ++pl) svq[((size_t)pl * l_pad + i) * dq + j] = p[pl];
        }
    return svq;
}

// The reference's support vectors ARE feature vectors of its training set: 8-bit numerators over 255 (src/OCR.cpp:211, src/utils.cpp:1478-1541), written to the model
// with 8 significant digits.  A model all of whose values are such multiples of 1 / 255 gets its support vectors as bytes: |x - sv|^2 is then an EXACT integer
// over 255^2, from one 8-bit matrix instruction per 32 features (k_svm_kernel_i8).  sv8[i][j], i < l_pad, j < dq8: numerator ^ 0x80 (= numerator - 128 as a signed
// byte; padding: numerator 0); svs[2 i] = sum of the numerators, svs[2 i + 1] = sum of their squares.  Returns false (and leaves the outputs alone) for any other model.
inline bool svm_sv_bytes(const std::vector<double> &sv_exact, int l, int l_pad, int dim, int dq8, std::vector<uint8_t> &sv8, std::vector<int32_t> &svs)
{
    std::vector<uint8_t> out((size_t)l_pad * dq8, 0x80);
    std::vector<int32_t> sums((size_t)2 * l_pad, 0);
    for (int i = 0; i < l; ++i)
        for (int j = 0; j < dim; ++j) {
            const double v = sv_exact[(size_t)i * dim + j] * 255.0;
            const double q = v < 0 ? -1.0 : (double)(long)(v + 0.5);
            if (q < 0 || q > 255 || v - q > 1e-3 || q - v > 1e-3) return false;
            out[(size_t)i * dq8 + j] = (uint8_t)((int)q ^ 0x80);
            sums[2 * (size_t)i] += (int32_t)q; sums[2 * (size_t)i + 1] += (int32_t)q * (int32_t)q;
        }
    sv8.swap(out); svs.swap(sums);
    return true;
}

// coef_rows[i][h][r][b], r < mp, b < 64 (k <= 65 classes): h = 0: sv_coef[b][start[i] + r], zero for r >= nsv[i]; h = 1: sv_coef[i][start[b + 1] + r],
// zero for r >= nsv[b + 1] -- coef is libsvm's [(k - 1) x l]
inline std::vector<double> svm_coef_rows(const std::vector<double> &coef, const std::vector<int32_t> &start, const std::vector<int32_t> &nsv, int k, int l, int mp)
{
    std::vector<double> rows((size_t)k * 2 * mp * 64, 0.0);
    for (int i = 0; i < k; ++i)
        for (int b = 0; b + 1 < k; ++b) {                     // b = the row of sv_coef, and the second class b + 1
            double *r1 = &rows[((size_t)(2 * i) * mp) * 64 + b], *r2 = r1 + (size_t)mp * 64;
            for (int r = 0; r < nsv[i]; ++r) r1[(size_t)r * 64] = coef[(size_t)b * l + start[i] + r];
            if (i < k - 1) for (int r = 0; r < nsv[b + 1]; ++r) r2[(size_t)r * 64] = coef[(size_t)i * l + start[b + 1] + r];
        }
    return rows;
}

} // namespace str_er
