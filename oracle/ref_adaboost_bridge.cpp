/*
 * ref_adaboost_bridge.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * A few extern "C" entry points around the REAL reference classes
 * CascadeBoost::load_classifier / CascadeBoost::predict
 * (/root/reference/src/adaboost.cpp:873-951, 507-542), so tests can call the
 * unmodified reference cascade through ctypes.  This file contains no
 * reference code; oracle/Makefile compiles it TOGETHER WITH the reference's
 * own src/adaboost.cpp (from where it lies under /root/reference) into
 * oracle/_ref/libref_adaboost.so.  oracle/_ref/ is git-ignored.
 */
#include "adaboost.h" /* -I/root/reference/inc */

#include <vector>

extern "C" {

void *ref_cascade_load(const char *path)
{
    CascadeBoost *c = new CascadeBoost();
    if (!c->load_classifier(path)) { delete c; return nullptr; }
    return c;
}

int ref_cascade_n_stumps(void *h) { return static_cast<CascadeBoost *>(h)->get_num_iter(); }

double ref_cascade_predict(void *h, const double *fv, int n)
{
    std::vector<double> v(fv, fv + n);
    return static_cast<CascadeBoost *>(h)->predict(v);
}

void ref_cascade_free(void *h) { delete static_cast<CascadeBoost *>(h); }

} /* extern "C" */
