/*
 * svm_oracle.h -- TEST INFRASTRUCTURE ONLY (parity oracle for SURVEY 8a row a14).
 *
 * Plain-C restatement of libsvm 3.21 inference as vendored by the reference:
 * svm_load_model (src/svm.cpp:2767-2982), Kernel::k_function RBF (:316-373),
 * svm_predict_values (:2501-2575), sigmoid_predict (:1818-1826),
 * multiclass_probability (:1829-1890), svm_predict_probability (:2592-2629).
 *
 * PINNED: oracle/_ref/libref_svm.so is the reference's own src/svm.cpp compiled
 * unmodified; tests compare this restatement against it bit for bit, and
 * tests/golden/svm_vectors.npz stores its outputs.  The model is synthetic
 * (classifier/OCR.model is missing from the checkout): it was minted with the
 * reference's own svm-train and the flags of src/utils.cpp:1547-1551.
 */
#ifndef SVM_ORACLE_H
#define SVM_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ero_svm ero_svm;

ero_svm *ero_svm_load(const char *path);            /* NULL on failure; only c_svc + rbf with probA/probB */
void     ero_svm_free(ero_svm *m);
int      ero_svm_nr_class(const ero_svm *m);
int      ero_svm_total_sv(const ero_svm *m);
int      ero_svm_max_index(const ero_svm *m);       /* largest feature index used by any SV */
double   ero_svm_gamma(const ero_svm *m);
/* x: dense feature vector x[0..dim-1] (feature index i is x[i]; zeros are skipped exactly like the
 * sparse svm_node list the reference builds in OCR::extract_feature, src/OCR.cpp:203-216).
 * dec: k(k-1)/2 decision values, prob: k probabilities.  Returns model->label[argmax prob]. */
int      ero_svm_predict_probability(const ero_svm *m, const double *x, int dim, double *dec, double *prob);

#ifdef __cplusplus
}
#endif
#endif
