/* svm_oracle.c -- TEST INFRASTRUCTURE ONLY; see svm_oracle.h. */
#define _GNU_SOURCE
#include "svm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct ero_svm {
    int     k, l, max_index;
    double  gamma;
    double *rho, *probA, *probB;   /* k(k-1)/2 */
    int    *label, *nsv;           /* k */
    double *coef;                  /* (k-1) x l, row-major: coef[j*l + i] = sv_coef[j][i] */
    int    *sv_ptr;                /* l+1, CSR */
    int    *sv_idx;
    double *sv_val;
};

static int read_doubles(char *s, double *out, int n)
{
    int got = 0;
    char *end;
    while (got < n) {
        double v = strtod(s, &end);
        if (end == s) break;
        out[got++] = v;
        s = end;
    }
    return got;
}

ero_svm *ero_svm_load(const char *path)   /* src/svm.cpp:2767-2982 */
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    ero_svm *m = (ero_svm *)calloc(1, sizeof(*m));
    char *line = NULL;
    size_t cap = 0;
    int ok_type = 0, ok_kernel = 0;
    while (getline(&line, &cap, f) > 0) {
        char key[64];
        if (sscanf(line, "%63s", key) != 1) continue;
        char *rest = line + strlen(key);
        const int np = m->k * (m->k - 1) / 2;
        if (!strcmp(key, "svm_type")) ok_type = strstr(rest, "c_svc") != NULL;
        else if (!strcmp(key, "kernel_type")) ok_kernel = strstr(rest, "rbf") != NULL;
        else if (!strcmp(key, "gamma")) m->gamma = strtod(rest, NULL);
        else if (!strcmp(key, "nr_class")) m->k = atoi(rest);
        else if (!strcmp(key, "total_sv")) m->l = atoi(rest);
        else if (!strcmp(key, "rho")) { m->rho = (double *)calloc((size_t)np, 8); read_doubles(rest, m->rho, np); }
        else if (!strcmp(key, "probA")) { m->probA = (double *)calloc((size_t)np, 8); read_doubles(rest, m->probA, np); }
        else if (!strcmp(key, "probB")) { m->probB = (double *)calloc((size_t)np, 8); read_doubles(rest, m->probB, np); }
        else if (!strcmp(key, "label")) {
            m->label = (int *)calloc((size_t)m->k, sizeof(int));
            double *t = (double *)calloc((size_t)m->k, 8);
            read_doubles(rest, t, m->k);
            for (int i = 0; i < m->k; ++i) m->label[i] = (int)t[i];
            free(t);
        } else if (!strcmp(key, "nr_sv")) {
            m->nsv = (int *)calloc((size_t)m->k, sizeof(int));
            double *t = (double *)calloc((size_t)m->k, 8);
            read_doubles(rest, t, m->k);
            for (int i = 0; i < m->k; ++i) m->nsv[i] = (int)t[i];
            free(t);
        } else if (!strcmp(key, "SV")) break;
    }
    if (!ok_type || !ok_kernel || m->k < 2 || m->l < 1 || !m->rho || !m->probA || !m->probB || !m->label || !m->nsv) {
        fclose(f); free(line); ero_svm_free(m); return NULL;
    }
    m->coef = (double *)calloc((size_t)(m->k - 1) * m->l, 8);
    m->sv_ptr = (int *)calloc((size_t)m->l + 1, sizeof(int));
    int cap_nz = 1 << 16, nz = 0;
    m->sv_idx = (int *)malloc(sizeof(int) * (size_t)cap_nz);
    m->sv_val = (double *)malloc(8 * (size_t)cap_nz);
    for (int i = 0; i < m->l; ++i) {
        if (getline(&line, &cap, f) <= 0) { fclose(f); free(line); ero_svm_free(m); return NULL; }
        char *s = line, *end;
        for (int j = 0; j < m->k - 1; ++j) { m->coef[(size_t)j * m->l + i] = strtod(s, &end); s = end; }
        m->sv_ptr[i] = nz;
        for (;;) {
            long idx = strtol(s, &end, 10);
            if (end == s || *end != ':') break;
            s = end + 1;
            double v = strtod(s, &end);
            s = end;
            if (nz == cap_nz) {
                cap_nz *= 2;
                m->sv_idx = (int *)realloc(m->sv_idx, sizeof(int) * (size_t)cap_nz);
                m->sv_val = (double *)realloc(m->sv_val, 8 * (size_t)cap_nz);
            }
            m->sv_idx[nz] = (int)idx; m->sv_val[nz] = v; ++nz;
            if ((int)idx > m->max_index) m->max_index = (int)idx;
        }
    }
    m->sv_ptr[m->l] = nz;
    fclose(f); free(line);
    return m;
}

void ero_svm_free(ero_svm *m)
{
    if (!m) return;
    free(m->rho); free(m->probA); free(m->probB); free(m->label); free(m->nsv);
    free(m->coef); free(m->sv_ptr); free(m->sv_idx); free(m->sv_val); free(m);
}

int    ero_svm_nr_class(const ero_svm *m) { return m->k; }
int    ero_svm_total_sv(const ero_svm *m) { return m->l; }
int    ero_svm_max_index(const ero_svm *m) { return m->max_index; }
double ero_svm_gamma(const ero_svm *m) { return m->gamma; }

/* Kernel::k_function, RBF branch (src/svm.cpp:326-366): merge of two index-sorted sparse lists */
static double k_rbf(const int *xi, const double *xv, int xn, const int *yi, const double *yv, int yn, double gamma)
{
    double sum = 0;
    int a = 0, b = 0;
    while (a < xn && b < yn) {
        if (xi[a] == yi[b]) { double d = xv[a] - yv[b]; sum += d * d; ++a; ++b; }
        else if (xi[a] > yi[b]) { sum += yv[b] * yv[b]; ++b; }
        else { sum += xv[a] * xv[a]; ++a; }
    }
    while (a < xn) { sum += xv[a] * xv[a]; ++a; }
    while (b < yn) { sum += yv[b] * yv[b]; ++b; }
    return exp(-gamma * sum);
}

static double sigmoid_predict(double dv, double A, double B)   /* src/svm.cpp:1818-1826 */
{
    double fApB = dv * A + B;
    if (fApB >= 0) return exp(-fApB) / (1.0 + exp(-fApB));
    return 1.0 / (1 + exp(fApB));
}

/* multiclass_probability, src/svm.cpp:1829-1890 (r is k x k row-major) */
static void multiclass_probability(int k, const double *r, double *p)
{
    int t, j, iter = 0, max_iter = k > 100 ? k : 100;
    double *Q = (double *)malloc(8 * (size_t)k * k), *Qp = (double *)malloc(8 * (size_t)k);
    double pQp, eps = 0.005 / k;
    for (t = 0; t < k; t++) {
        p[t] = 1.0 / k;
        Q[t * k + t] = 0;
        for (j = 0; j < t; j++) { Q[t * k + t] += r[j * k + t] * r[j * k + t]; Q[t * k + j] = Q[j * k + t]; }
        for (j = t + 1; j < k; j++) { Q[t * k + t] += r[j * k + t] * r[j * k + t]; Q[t * k + j] = -r[j * k + t] * r[t * k + j]; }
    }
    for (iter = 0; iter < max_iter; iter++) {
        pQp = 0;
        for (t = 0; t < k; t++) {
            Qp[t] = 0;
            for (j = 0; j < k; j++) Qp[t] += Q[t * k + j] * p[j];
            pQp += p[t] * Qp[t];
        }
        double max_error = 0;
        for (t = 0; t < k; t++) { double e = fabs(Qp[t] - pQp); if (e > max_error) max_error = e; }
        if (max_error < eps) break;
        for (t = 0; t < k; t++) {
            double diff = (-Qp[t] + pQp) / Q[t * k + t];
            p[t] += diff;
            pQp = (pQp + diff * (diff * Q[t * k + t] + 2 * Qp[t])) / (1 + diff) / (1 + diff);
            for (j = 0; j < k; j++) { Qp[j] = (Qp[j] + diff * Q[t * k + j]) / (1 + diff); p[j] /= (1 + diff); }
        }
    }
    free(Q); free(Qp);
}

int ero_svm_predict_probability(const ero_svm *m, const double *x, int dim, double *dec, double *prob)
{
    const int k = m->k, l = m->l;
    /* the sparse list OCR::extract_feature builds: non-zero entries in index order (src/OCR.cpp:203-216) */
    int *xi = (int *)malloc(sizeof(int) * (size_t)(dim > 0 ? dim : 1));
    double *xv = (double *)malloc(8 * (size_t)(dim > 0 ? dim : 1));
    int xn = 0;
    for (int i = 0; i < dim; ++i) if (x[i] != 0) { xi[xn] = i; xv[xn] = x[i]; ++xn; }
    double *kv = (double *)malloc(8 * (size_t)l);
    for (int i = 0; i < l; ++i)
        kv[i] = k_rbf(xi, xv, xn, m->sv_idx + m->sv_ptr[i], m->sv_val + m->sv_ptr[i], m->sv_ptr[i + 1] - m->sv_ptr[i], m->gamma);
    int *start = (int *)malloc(sizeof(int) * (size_t)k);
    start[0] = 0;
    for (int i = 1; i < k; ++i) start[i] = start[i - 1] + m->nsv[i - 1];
    int p = 0;
    for (int i = 0; i < k; ++i)                       /* svm_predict_values, src/svm.cpp:2539-2566 */
        for (int j = i + 1; j < k; ++j) {
            double sum = 0;
            const int si = start[i], sj = start[j], ci = m->nsv[i], cj = m->nsv[j];
            const double *coef1 = m->coef + (size_t)(j - 1) * l, *coef2 = m->coef + (size_t)i * l;
            for (int q = 0; q < ci; ++q) sum += coef1[si + q] * kv[si + q];
            for (int q = 0; q < cj; ++q) sum += coef2[sj + q] * kv[sj + q];
            sum -= m->rho[p];
            dec[p++] = sum;
        }
    double *r = (double *)calloc((size_t)k * k, 8);
    const double min_prob = 1e-7;
    p = 0;
    for (int i = 0; i < k; ++i)                       /* svm_predict_probability, src/svm.cpp:2603-2611 */
        for (int j = i + 1; j < k; ++j) {
            double v = sigmoid_predict(dec[p], m->probA[p], m->probB[p]);
            v = v > min_prob ? v : min_prob;
            v = v < 1 - min_prob ? v : 1 - min_prob;
            r[i * k + j] = v; r[j * k + i] = 1 - v;
            ++p;
        }
    multiclass_probability(k, r, prob);
    int best = 0;
    for (int i = 1; i < k; ++i) if (prob[i] > prob[best]) best = i;
    const int lab = m->label[best];
    free(xi); free(xv); free(kv); free(start); free(r);
    return lab;
}
