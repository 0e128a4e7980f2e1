// svm_kernels.h -- device model + launcher of the batched libsvm inference (svm_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace str_er {

struct SvmDev {
    int32_t k, l, l_pad, dim, dpad;
    double  gamma;
    const float  *sv;       // [l_pad x dpad] dense, zero padded
    const double *svnorm;   // [l_pad]
    const double *coef;     // [(k-1) x l]
    const double *rho, *probA, *probB;   // [k(k-1)/2]
    const int32_t *label, *nsv, *start;  // [k]
    const int32_t *pair_i, *pair_j;      // [k(k-1)/2]
};

// x: device [n x dim] f64; scratch: xf [n_pad x dpad] f32, xnorm [n_pad], kv [n_pad x l_pad], dec [n x pairs];
// outputs prob [n x k], label [n].  n_pad = n rounded up to 64 (scratch rows beyond n must be zero / readable).
void launch_svm_predict(hipStream_t s, const double *x, int n, int dim, float *xf, double *xnorm, int n_pad, double *kv, double *dec,
                        double *prob, int32_t *label, const SvmDev &m);

} // namespace str_er
