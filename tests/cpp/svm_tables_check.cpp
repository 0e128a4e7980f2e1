// Host-only check of scene-text-recognition_amd/csrc/svm_tables.h (compiled and run by tests/test_host_cpp.py): the three-piece bf16 split is exact,
// the planes and the per-class coefficient rows hold what their definitions say.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "svm_tables.h"

using namespace str_er;

static float bf16_to_float(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main()
{
    std::mt19937_64 rng(12345);
    // 1. split_bf16x3: pieces sum to the value exactly (pieces added smallest first in double: every partial sum is exact)
    long bad = 0;
    auto check = [&](float v) {
        uint16_t p[3];
        split_bf16x3(v, p);
        const double s = (double)bf16_to_float(p[2]) + (double)bf16_to_float(p[1]) + (double)bf16_to_float(p[0]);
        if (s != (double)v) ++bad;
        // each piece has at most 8 significant bits and the sign of the value (or is zero)
        for (int i = 0; i < 3; ++i) { const float f = bf16_to_float(p[i]); if (f != 0.f && (f < 0) != (v < 0)) ++bad; }
    };
    const float edge[] = {0.f, -0.f, 1.f, -1.f, 1.f / 255.f, 254.f / 255.f, 0.99999994f, 1.0000001f, 3.4028235e38f, -3.4028235e38f, 1e-30f, -7.3e-20f, 0.5f, 0.33333334f};
    for (float v : edge) check(v);
    std::uniform_int_distribution<uint32_t> bits(0, 0xFFFFFFFFu);
    for (int i = 0; i < 2000000; ++i) {
        uint32_t u = bits(rng);
        const uint32_t e = (u >> 23) & 0xFF;
        if (e == 0xFF || e < 40) continue;                      // no inf / nan; values whose third piece would be subnormal are cut (header comment)
        float v; memcpy(&v, &u, 4);
        check(v);
    }
    std::uniform_real_distribution<double> unit(0.0, 1.0);
    for (int i = 0; i < 200000; ++i) check((float)(floor(unit(rng) * 256.0) / 255.0));      // the features' own values
    if (bad) { printf("split_bf16x3: %ld mismatches\n", bad); return 1; }

    // 2. svm_rows_per_class
    const int exp_rows[][2] = {{1, 8}, {4, 8}, {5, 5}, {6, 8}, {8, 8}, {9, 16}, {16, 16}, {17, 24}, {600, 600}};
    for (auto &e : exp_rows) if (svm_rows_per_class(e[0]) != e[1]) { printf("svm_rows_per_class(%d) = %d\n", e[0], svm_rows_per_class(e[0])); return 1; }

    // 3. planes and coefficient rows of random small models (with empty classes) against their definitions
    for (int trial = 0; trial < 40; ++trial) {
        const int k = 2 + (int)(rng() % 64), dim = 1 + (int)(rng() % 200), dpad = (dim + 15) / 16 * 16, dq = (dim + 63) / 64 * 64;
        std::vector<int32_t> nsv(k), start(k);
        int l = 0, msv = 0;
        for (int i = 0; i < k; ++i) { nsv[i] = (int)(rng() % 7); if (trial % 3 == 0 && i == k - 1) nsv[i] = 0; start[i] = l; l += nsv[i]; msv = std::max(msv, nsv[i]); }
        if (l == 0) { nsv[0] = 1; l = 1; msv = 1; for (int i = 1; i < k; ++i) start[i] = 1; }
        const int l_pad = (l + 63) / 64 * 64, mp = svm_rows_per_class(msv);
        std::vector<float> sv((size_t)l_pad * dpad, 0.f);
        for (int i = 0; i < l; ++i) for (int j = 0; j < dim; ++j) sv[(size_t)i * dpad + j] = (float)(floor(unit(rng) * 256.0) / 255.0);
        std::vector<double> coef((size_t)(k - 1) * l);
        for (auto &c : coef) c = unit(rng) * 4.0 - 2.0;
        const std::vector<uint16_t> svq = svm_sv_planes(sv, l, l_pad, dim, dpad, dq);
        if (svq.size() != (size_t)3 * l_pad * dq) { printf("svq size\n"); return 1; }
        for (int i = 0; i < l_pad; ++i)
            for (int j = 0; j < dq; ++j) {
                const double s = (double)bf16_to_float(svq[((size_t)2 * l_pad + i) * dq + j]) + (double)bf16_to_float(svq[((size_t)1 * l_pad + i) * dq + j]) +
                                 (double)bf16_to_float(svq[((size_t)0 * l_pad + i) * dq + j]);
                const double want = (i < l && j < dim) ? (double)sv[(size_t)i * dpad + j] : 0.0;
                if (s != want) { printf("svq[%d][%d]\n", i, j); return 1; }
            }
        const std::vector<double> rows = svm_coef_rows(coef, start, nsv, k, l, mp);
        if (rows.size() != (size_t)k * 2 * mp * 64) { printf("rows size\n"); return 1; }
        for (int i = 0; i < k; ++i)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < mp; ++r)
                    for (int b = 0; b < 64; ++b) {
                        double want = 0.0;
                        if (b + 1 < k) {
                            if (h == 0 && r < nsv[i]) want = coef[(size_t)b * l + start[i] + r];
                            if (h == 1 && i < k - 1 && r < nsv[b + 1]) want = coef[(size_t)i * l + start[b + 1] + r];
                        }
                        if (rows[(((size_t)(2 * i + h)) * mp + r) * 64 + b] != want) { printf("rows[%d][%d][%d][%d] (k %d)\n", i, h, r, b, k); return 1; }
                    }
        // a pair's decision value from the rows == libsvm's sum (same order: class i's support vectors, then class j's)
        std::vector<double> K(l);
        for (auto &v : K) v = unit(rng);
        for (int i = 0; i < k; ++i)
            for (int j = i + 1; j < k; ++j) {
                double a = 0, bsum = 0;
                for (int q = 0; q < nsv[i]; ++q) a += coef[(size_t)(j - 1) * l + start[i] + q] * K[start[i] + q];
                for (int q = 0; q < nsv[j]; ++q) a += coef[(size_t)i * l + start[j] + q] * K[start[j] + q];
                for (int r = 0; r < mp; ++r) bsum += rows[(((size_t)(2 * i)) * mp + r) * 64 + (j - 1)] * K[std::min(start[i] + r, l - 1)];
                for (int r = 0; r < mp; ++r) bsum += rows[(((size_t)(2 * i + 1)) * mp + r) * 64 + (j - 1)] * K[std::min(start[j] + r, l - 1)];
                if (a != bsum) { printf("pair (%d, %d) of %d: %.17g vs %.17g\n", i, j, k, a, bsum); return 1; }
            }
    }
    // 4. svm_sv_bytes: numerators over 255 written with 8 significant digits (svm_save_model: "%.8g") and read back come out as the bytes, their sums and
    // sums of squares; a model with one value that is no such multiple is refused
    for (int trial = 0; trial < 20; ++trial) {
        const int l = 1 + (int)(rng() % 9), dim = 40 + (int)(rng() % 60), l_pad = (l + 63) / 64 * 64, dq8 = (dim + 127) / 128 * 128;
        std::vector<int> q((size_t)l * dim);
        std::vector<double> sv((size_t)l * dim);
        for (size_t i = 0; i < q.size(); ++i) {
            q[i] = (int)(rng() % 4 == 0 ? rng() % 256 : 0);
            char txt[32];
            snprintf(txt, sizeof txt, "%.8g", q[i] / 255.0);
            sv[i] = strtod(txt, nullptr);
        }
        std::vector<uint8_t> b8; std::vector<int32_t> sums;
        if (!svm_sv_bytes(sv, l, l_pad, dim, dq8, b8, sums)) { printf("svm_sv_bytes refused a model of numerators\n"); return 1; }
        for (int i = 0; i < l_pad; ++i) {
            int s1 = 0, s2 = 0;
            for (int j = 0; j < dq8; ++j) {
                const int want = i < l && j < dim ? q[(size_t)i * dim + j] : 0;
                if ((b8[(size_t)i * dq8 + j] ^ 0x80) != want) { printf("svm_sv_bytes: byte (%d, %d)\n", i, j); return 1; }
                s1 += want; s2 += want * want;
            }
            if (sums[2 * (size_t)i] != s1 || sums[2 * (size_t)i + 1] != s2) { printf("svm_sv_bytes: sums of row %d\n", i); return 1; }
        }
        sv[rng() % sv.size()] = 0.5003;
        std::vector<uint8_t> keep = b8;
        if (svm_sv_bytes(sv, l, l_pad, dim, dq8, b8, sums) || b8 != keep) { printf("svm_sv_bytes took a value that is no multiple of 1 / 255\n"); return 1; }
    }
    printf("svm tables ok\n");
    return 0;
}
