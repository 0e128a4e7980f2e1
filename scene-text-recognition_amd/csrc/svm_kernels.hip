// svm_kernels.hip -- batched libsvm inference for the OCR scorer of config 3:
// svm_predict_probability (src/svm.cpp:2592-2629) for N feature vectors at once.
//
//   k_svm_prep    f64 features -> padded f32 matrix + squared norms
//   k_svm_kernel  K[n][i] = exp(-gamma * ||x_n - sv_i||^2)  via  ||x||^2 + ||sv||^2 - 2 x.sv, the x.sv
//                 part is a dense [N x 1800] x [1800 x l] GEMM on the matrix cores
//                 (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate -- exact f32 FMA chains).  This is the
//                 one step of the reference that is a real dense contraction, hence the one MFMA use.
//   k_svm_decide  k(k-1)/2 pairwise decision values per vector, summed in the reference's order
//                 (svm_predict_values, src/svm.cpp:2539-2566), f64
//   k_svm_prob    Platt sigmoid (:1818-1826) + Wu-Lin-Weng pairwise coupling (:1829-1890), one wave per
//                 vector, f64, same operation order as the reference
#include <hip/hip_runtime.h>

#include <float.h>
#include <stdint.h>

#include "svm_kernels.h"

namespace str_er {

typedef float float16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_svm_prep(const double *__restrict__ x, int n, int dim, float *__restrict__ xf, int dpad,
                                                  double *__restrict__ xnorm)
{
    // one wave per vector
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n) return;
    double s = 0;
    for (int i = lane; i < dpad; i += 64) {
        const double d = i < dim ? x[(size_t)v * dim + i] : 0.0;
        xf[(size_t)v * dpad + i] = (float)d;
        s += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) xnorm[v] = s;
}

// C = X . SV^T with 64x64 block tiles, one 32x32 MFMA tile per wave, K-step 16 staged through LDS.
constexpr int GT = 64, GK = 16;

__global__ __launch_bounds__(256) void k_svm_kernel(const float *__restrict__ xf, const double *__restrict__ xnorm, int n_pad,
                                                    const float *__restrict__ sv, const double *__restrict__ svnorm, int l_pad,
                                                    int dpad, double gamma, double *__restrict__ kv)
{
    __shared__ float As[GK][GT + 1];
    __shared__ float Bs[GK][GT + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    const int lrow = tid >> 2, lk = (tid & 3) * 4;          // this lane stages 4 consecutive k of one row
    float16v acc = {0};
    for (int k0 = 0; k0 < dpad; k0 += GK) {
        const float4 a = *reinterpret_cast<const float4 *>(xf + (size_t)(m0 + lrow) * dpad + k0 + lk);
        const float4 b = *reinterpret_cast<const float4 *>(sv + (size_t)(n0 + lrow) * dpad + k0 + lk);
        __syncthreads();
        As[lk][lrow] = a.x; As[lk + 1][lrow] = a.y; As[lk + 2][lrow] = a.z; As[lk + 3][lrow] = a.w;
        Bs[lk][lrow] = b.x; Bs[lk + 1][lrow] = b.y; Bs[lk + 2][lrow] = b.z; Bs[lk + 3][lrow] = b.w;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const float av = As[kk + (lane >> 5)][wm + (lane & 31)];
            const float bv = Bs[kk + (lane >> 5)][wn + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    // 32x32 accumulator layout: element i of lane L is row 8*(i/4) + 4*(L/32) + i%4, column L%32
    const int col = n0 + wn + (lane & 31);
    const double sn = svnorm[col];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        double d2 = xnorm[row] + sn - 2.0 * (double)acc[i];
        d2 = d2 > 0 ? d2 : 0;
        kv[(size_t)row * l_pad + col] = exp(-gamma * d2);
    }
}

// svm_predict_values (src/svm.cpp:2539-2566): one lane per class pair, sums in the reference's order.
__global__ __launch_bounds__(256) void k_svm_decide(const double *__restrict__ kv, int l_pad, SvmDev m, double *__restrict__ dec)
{
    const int v = blockIdx.x;
    const int np = m.k * (m.k - 1) / 2;
    const double *kr = kv + (size_t)v * l_pad;
    for (int p = threadIdx.x; p < np; p += blockDim.x) {
        const int i = m.pair_i[p], j = m.pair_j[p];
        const int si = m.start[i], sj = m.start[j], ci = m.nsv[i], cj = m.nsv[j];
        const double *coef1 = m.coef + (size_t)(j - 1) * m.l, *coef2 = m.coef + (size_t)i * m.l;
        double sum = 0;
        for (int q = 0; q < ci; ++q) sum += coef1[si + q] * kr[si + q];
        for (int q = 0; q < cj; ++q) sum += coef2[sj + q] * kr[sj + q];
        dec[(size_t)v * np + p] = sum - m.rho[p];
    }
}

__device__ __forceinline__ double bcast(double v, int src)
{
    const unsigned long long u = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// sigmoid_predict + multiclass_probability (src/svm.cpp:1818-1890), one wave per vector.
// Lane L owns classes L and L+64 (k <= 128).  Every scalar of the reference's loops (Qp[t], Q[t][t], pQp,
// diff) is broadcast so all lanes apply the same operations in the same order as the sequential code.
__global__ __launch_bounds__(64) void k_svm_prob(const double *__restrict__ dec, SvmDev m, double *__restrict__ prob,
                                                 int32_t *__restrict__ label)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int k = m.k, np = k * (k - 1) / 2;
    double *V = sm;            // np pairwise probabilities r[i][j], i < j
    double *Q = sm + np;       // k x k
    const int v = blockIdx.x, lane = threadIdx.x;
    const double min_prob = 1e-7;
    for (int p = lane; p < np; p += 64) {
        const double fApB = dec[(size_t)v * np + p] * m.probA[p] + m.probB[p];
        double s = fApB >= 0 ? exp(-fApB) / (1.0 + exp(-fApB)) : 1.0 / (1 + exp(fApB));
        s = s > min_prob ? s : min_prob;
        s = s < 1 - min_prob ? s : 1 - min_prob;
        V[p] = s;
    }
    __syncthreads();
    // r(i,j): V for i<j, 1-V for i>j.  pair index of (i<j): i*k - i*(i+1)/2 + (j-i-1)
#define RIJ(i, j) ((i) < (j) ? V[(i) * k - (i) * ((i) + 1) / 2 + ((j) - (i) - 1)] : 1.0 - V[(j) * k - (j) * ((j) + 1) / 2 + ((i) - (j) - 1)])
    for (int t = lane; t < k; t += 64) {
        double qtt = 0;
        for (int j = 0; j < t; ++j) { const double r = RIJ(j, t); qtt += r * r; }
        for (int j = t + 1; j < k; ++j) { const double r = RIJ(j, t); qtt += r * r; Q[t * k + j] = -r * RIJ(t, j); }
        Q[t * k + t] = qtt;
    }
    __syncthreads();
    for (int t = lane; t < k; t += 64)
        for (int j = 0; j < t; ++j) Q[t * k + j] = Q[j * k + t];
    __syncthreads();
#undef RIJ
    const bool has2 = lane + 64 < k;
    double p0 = 1.0 / k, p1 = 1.0 / k, q0 = 0, q1 = 0;       // p[lane], p[lane+64], Qp[lane], Qp[lane+64]
    const int max_iter = k > 100 ? k : 100;
    const double eps = 0.005 / k;
    for (int iter = 0; iter < max_iter; ++iter) {
        q0 = 0; q1 = 0;
        for (int j = 0; j < k; ++j) {
            const double pj = j < 64 ? bcast(p0, j) : bcast(p1, j - 64);
            if (lane < k) q0 += Q[lane * k + j] * pj;
            if (has2) q1 += Q[(lane + 64) * k + j] * pj;
        }
        double pQp = 0;
        for (int t = 0; t < k; ++t) pQp += t < 64 ? bcast(p0 * q0, t) : bcast(p1 * q1, t - 64);
        double err = lane < k ? fabs(q0 - pQp) : 0.0;
        if (has2) err = fmax(err, fabs(q1 - pQp));
        for (int o = 32; o > 0; o >>= 1) err = fmax(err, __shfl_xor(err, o));
        if (err < eps) break;
        for (int t = 0; t < k; ++t) {
            const double Qpt = t < 64 ? bcast(q0, t) : bcast(q1, t - 64);
            const double Qtt = Q[t * k + t];
            const double diff = (-Qpt + pQp) / Qtt;
            if (t == lane) p0 += diff;
            if (t == lane + 64) p1 += diff;
            pQp = (pQp + diff * (diff * Qtt + 2 * Qpt)) / (1 + diff) / (1 + diff);
            if (lane < k) { q0 = (q0 + diff * Q[t * k + lane]) / (1 + diff); p0 /= (1 + diff); }
            if (has2) { q1 = (q1 + diff * Q[t * k + lane + 64]) / (1 + diff); p1 /= (1 + diff); }
        }
    }
    if (lane < k) prob[(size_t)v * k + lane] = p0;
    if (has2) prob[(size_t)v * k + lane + 64] = p1;
    // argmax with the reference's tie rule (first maximum, src/svm.cpp:2614-2617)
    double best = lane < k ? p0 : -1.0;
    int    bi = lane;
    if (has2 && p1 > best) { best = p1; bi = lane + 64; }
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int    oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) label[v] = m.label[bi];
}

void launch_svm_predict(hipStream_t s, const double *x, int n, int dim, float *xf, double *xnorm, int n_pad, double *kv, double *dec,
                        double *prob, int32_t *label, const SvmDev &m)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_svm_prep, dim3((n + 3) / 4), dim3(256), 0, s, x, n, dim, xf, m.dpad, xnorm);
    hipLaunchKernelGGL(k_svm_kernel, dim3(m.l_pad / GT, n_pad / GT), dim3(256), 0, s, xf, xnorm, n_pad, m.sv, m.svnorm, m.l_pad, m.dpad,
                       m.gamma, kv);
    hipLaunchKernelGGL(k_svm_decide, dim3(n), dim3(256), 0, s, kv, m.l_pad, m, dec);
    const size_t lds = sizeof(double) * ((size_t)m.k * (m.k - 1) / 2 + (size_t)m.k * m.k);
    hipLaunchKernelGGL(k_svm_prob, dim3(n), dim3(64), lds, s, dec, m, prob, label);
}

} // namespace str_er
