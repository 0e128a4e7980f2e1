// api_models.cpp -- the C ABI, part 2: the models -- cascade text (CascadeBoost::load_classifier), libsvm text model, and the entry points that run them on explicit inputs (predict, svm_predict_probability, OCR::chain_run)
#include "str_er_ctx.h"
#include "svm_tables.h"

namespace {

// ---- cascade text (format: SURVEY.md Appendix C; CascadeBoost::load_classifier,
// ---- src/adaboost.cpp:873-951) ---------------------------------------------------------------
bool parse_number(const std::string &t, double &v)
{
    if (t.empty()) return false;
    char *end = nullptr;
    v = std::strtod(t.c_str(), &end);
    return end != t.c_str();
}

// (int) of a parsed number, as the reference's (int)stod(...) -- but a value an int cannot hold (inf, nan, 1e99: undefined behaviour in the
// reference, found by the fuzz loop under UBSan) is a format error here
bool to_int(double v, int32_t &out)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return false;
    out = (int32_t)v;
    return true;
}

} // namespace

int str_er_host::parse_cascade(str_er_ctx *c, HostCascade &hc, const char *text, size_t len)
{
    std::vector<std::string> tok;
    {
        size_t i = 0;
        while (i < len) {
            while (i < len && (text[i] == ' ' || text[i] == '\t' || text[i] == '\r' || text[i] == '\n')) ++i;
            size_t j = i;
            while (j < len && !(text[j] == ' ' || text[j] == '\t' || text[j] == '\r' || text[j] == '\n')) ++j;
            if (j > i) tok.emplace_back(text + i, j - i);
            i = j;
        }
    }
    size_t k = 0;
    auto next = [&]() -> const std::string * { return k < tok.size() ? &tok[k++] : nullptr; };
    HostCascade n;
    const std::string *t = next();
    if (!t || *t != "boost_type") return fail(c, STR_ER_EFORMAT, "cascade: missing boost_type");
    t = next();
    if (!t) return fail(c, STR_ER_EFORMAT, "cascade: truncated header");
    n.real = (*t != "DISCRETE");
    t = next();
    if (!t || *t != "base_type") return fail(c, STR_ER_EFORMAT, "cascade: missing base_type");
    t = next();
    t = next();
    if (!t || *t != "num_of_iter") return fail(c, STR_ER_EFORMAT, "cascade: missing num_of_iter");
    for (;;) {
        t = next();
        double v;
        if (!t || !parse_number(*t, v)) break;
        int32_t iv;
        if (!to_int(v, iv)) return fail(c, STR_ER_EFORMAT, "cascade: stage size is not an integer");
        n.stage_n.push_back(iv);
    }
    if (!t || *t != "threshold" || n.stage_n.empty()) return fail(c, STR_ER_EFORMAT, "cascade: missing threshold");
    for (size_t j = 0; j < n.stage_n.size(); ++j) {
        t = next();
        double v;
        if (!t || !parse_number(*t, v)) return fail(c, STR_ER_EFORMAT, "cascade: short threshold list");
        int32_t iv;
        if (!to_int(v, iv)) return fail(c, STR_ER_EFORMAT, "cascade: stage threshold outside the range of int");
        n.stage_thresh.push_back(iv); // (int)stod(...), src/adaboost.cpp:919
    }
    const int per = n.real ? 5 : 4;
    for (;;) {
        double v[5];
        int got = 0;
        for (; got < per; ++got) {
            t = next();
            if (!t || !parse_number(*t, v[got])) break;
        }
        if (got == 0) break;
        if (got < per) return fail(c, STR_ER_EFORMAT, "cascade: incomplete stump row");
        int32_t d;
        if (!to_int(v[1], d) || d < 0 || d >= 1024) return fail(c, STR_ER_EFORMAT, "cascade: feature index outside the 1024-bin histogram");
        n.dim.push_back((uint16_t)d);
        if (n.real) { n.thr.push_back(v[2]); n.dir.push_back(1.0); n.vp.push_back(v[3]); n.vn.push_back(v[4]); }
        else {
            int32_t dr;
            if (!to_int(v[2], dr)) return fail(c, STR_ER_EFORMAT, "cascade: stump direction is not an integer");
            n.dir.push_back((double)dr); n.thr.push_back(v[3]); n.vp.push_back(1.0 * v[0]); n.vn.push_back(-1.0 * v[0]);
        }
    }
    long long total = 0;
    for (int32_t s : n.stage_n) { if (s < 0) return fail(c, STR_ER_EFORMAT, "cascade: negative stage size"); total += s; }
    if (total > (long long)n.dim.size()) return fail(c, STR_ER_EFORMAT, "cascade: fewer stump rows than num_of_iter announces");
    // upload: one blob
    const size_t ns = n.dim.size(), nst = n.stage_n.size();
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 16); return o; };
    const size_t o_thr = take(ns * 8), o_dir = take(ns * 8), o_vp = take(ns * 8), o_vn = take(ns * 8), o_dim = take(ns * 2),
                 o_sn = take(nst * 4), o_st = take(nst * 4), o_rec = take(ns * sizeof(StumpRec)), o_w = take(ns * 4), o_ab = take(ns * 16);
    std::vector<uint8_t> blob(off ? off : 16);
    std::memcpy(&blob[o_thr], n.thr.data(), ns * 8); std::memcpy(&blob[o_dir], n.dir.data(), ns * 8);
    std::memcpy(&blob[o_vp], n.vp.data(), ns * 8); std::memcpy(&blob[o_vn], n.vn.data(), ns * 8);
    std::memcpy(&blob[o_dim], n.dim.data(), ns * 2); std::memcpy(&blob[o_sn], n.stage_n.data(), nst * 4);
    std::memcpy(&blob[o_st], n.stage_thresh.data(), nst * 4);
    int32_t all_unit = 1;
    for (size_t i = 0; i < ns; ++i) {
        StumpRec r;
        r.dim = n.dim[i]; r.thr = n.thr[i]; r.vp = n.vp[i]; r.vn = n.vn[i];
        r.mode = n.dir[i] == 1.0 ? 0 : (n.dir[i] == -1.0 ? 1 : 2);
        std::memcpy(&blob[o_rec + i * sizeof(StumpRec)], &r, sizeof(r));
        // integer form for 8-bit counts: (h < T) ? A : B
        double T = 0, A = n.vn[i], B = n.vn[i];
        if (r.mode == 0) {            // h < thr  <=>  h < ceil(thr)
            A = n.vp[i]; B = n.vn[i];
            T = std::isnan(n.thr[i]) ? 0.0 : std::ceil(n.thr[i]);
        } else if (r.mode == 1) {     // h > thr  <=>  !(h < floor(thr)+1)
            A = n.vn[i]; B = n.vp[i];
            T = std::isnan(n.thr[i]) ? 1e9 : std::floor(n.thr[i]) + 1.0;   // NaN: h > NaN is false -> always vn = A
        } else all_unit = 0;
        const uint32_t Ti = (uint32_t)std::min(std::max(T, 0.0), 300.0);
        const uint32_t wv = (uint32_t)n.dim[i] | (Ti << 10);
        std::memcpy(&blob[o_w + i * 4], &wv, 4);
        std::memcpy(&blob[o_ab + i * 16], &A, 8); std::memcpy(&blob[o_ab + i * 16 + 8], &B, 8);
    }
    void *d = nullptr;
    HIP_TRY(c, hipMalloc(&d, blob.size()));
    hipError_t e = hipMemcpy(d, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail(c, STR_ER_EHIP, std::string("cascade upload: ") + hipGetErrorString(e)); }
    if (hc.d_blob) (void)hipFree(hc.d_blob);
    n.d_blob = d;
    const uint8_t *b = static_cast<const uint8_t *>(d);
    n.dev.thr = reinterpret_cast<const double *>(b + o_thr); n.dev.dir = reinterpret_cast<const double *>(b + o_dir);
    n.dev.vp = reinterpret_cast<const double *>(b + o_vp); n.dev.vn = reinterpret_cast<const double *>(b + o_vn);
    n.dev.dim = reinterpret_cast<const uint16_t *>(b + o_dim);
    n.dev.rec = reinterpret_cast<const StumpRec *>(b + o_rec);
    n.dev.w = reinterpret_cast<const uint32_t *>(b + o_w); n.dev.ab = reinterpret_cast<const double *>(b + o_ab);
    n.dev.all_unit = all_unit;
    n.dev.stage_n = reinterpret_cast<const int32_t *>(b + o_sn); n.dev.stage_thresh = reinterpret_cast<const int32_t *>(b + o_st);
    n.dev.n_stages = (int32_t)nst; n.dev.n_stumps = (int32_t)ns;
    n.dev.max_stage = 0;
    for (int32_t s : n.stage_n) n.dev.max_stage = std::max(n.dev.max_stage, s);
    n.loaded = true;
    hc = std::move(n);
    return STR_ER_OK;
}

extern "C" {

int str_er_load_cascade_mem(str_er_ctx *c, int which, const char *text, size_t len)
try {
    if (!c) return STR_ER_EINVAL;
    if (!text || (which != STR_ER_CASCADE_STRONG && which != STR_ER_CASCADE_WEAK)) return fail(c, STR_ER_EINVAL, "bad cascade argument");
    HIP_TRY(c, hipSetDevice(c->prm.device));
    return parse_cascade(c, c->casc[which], text, len);
} ABI_GUARD(c)

int str_er_load_cascade(str_er_ctx *c, int which, const char *path)
try {
    if (!c) return STR_ER_EINVAL;
    if (!path) return fail(c, STR_ER_EINVAL, "null path");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(c, STR_ER_EIO, std::string("cannot open ") + path);
    std::string buf;
    char tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, n);
    std::fclose(f);
    return str_er_load_cascade_mem(c, which, buf.data(), buf.size());
} ABI_GUARD(c)

int str_er_cascade_info(const str_er_ctx *c, int which, int32_t *n_stages, int32_t *n_stumps)
try {
    if (!c || (which != 0 && which != 1)) return STR_ER_EINVAL;
    if (n_stages) *n_stages = c->casc[which].loaded ? c->casc[which].dev.n_stages : 0;
    if (n_stumps) *n_stumps = c->casc[which].loaded ? c->casc[which].dev.n_stumps : 0;
    return STR_ER_OK;
} ABI_GUARD(const_cast<str_er_ctx *>(c))

int str_er_cascade_predict(str_er_ctx *c, int which, const double *fv, int32_t n, double *out)
try {
    if (!c) return STR_ER_EINVAL;
    if ((which != 0 && which != 1) || n < 0 || (n > 0 && (!fv || !out))) return fail(c, STR_ER_EINVAL, "bad arguments");
    if (!c->casc[which].loaded) return fail(c, STR_ER_ESTATE, "cascade not loaded");
    if (n == 0) return STR_ER_OK;
    HIP_TRY(c, hipSetDevice(c->prm.device));
    const size_t in_b = 8192 * (size_t)n, total = in_b + 8 * (size_t)n;
    int rc = ensure_scratch(c, total);
    if (rc != STR_ER_OK) return rc;
    uint8_t *s = static_cast<uint8_t *>(c->d_scratch);
    HIP_TRY(c, hipMemcpyAsync(s, fv, in_b, hipMemcpyHostToDevice, c->stream));
    launch_cascade_fv(c->stream, reinterpret_cast<const double *>(s), n, reinterpret_cast<double *>(s + in_b), c->casc[which].dev);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, s + in_b, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, wait_stream(c, c->stream));
    return STR_ER_OK;
} ABI_GUARD(c)

// ---- libsvm text model (svm_save_model format, src/svm.cpp:2641-2736; reader :2767-2982) -------------
int str_er_load_svm_model_mem(str_er_ctx *c, const char *text, size_t len, int32_t dim)
try {
    if (!c) return STR_ER_EINVAL;
    if (!text || dim < 1) return fail(c, STR_ER_EINVAL, "bad arguments");
    HIP_TRY(c, hipSetDevice(c->prm.device));
    std::string buf(text, len);
    size_t pos = 0;
    auto next_line = [&](std::string &out) -> bool {
        if (pos >= buf.size()) return false;
        size_t e = buf.find('\n', pos);
        if (e == std::string::npos) e = buf.size();
        out.assign(buf, pos, e - pos);
        pos = e + 1;
        return true;
    };
    auto numbers = [](const std::string &s, size_t from, std::vector<double> &out) {
        const char *p = s.c_str() + from;
        char *end;
        for (;;) { double v = std::strtod(p, &end); if (end == p) break; out.push_back(v); p = end; }
    };
    int k = 0, l = 0;
    double gamma = 0;
    bool ok_type = false, ok_kernel = false, have_sv = false;
    std::vector<double> rho, pa, pb, lab, nsv;
    std::string line;
    while (next_line(line)) {
        const size_t sp = line.find(' ');
        const std::string key = line.substr(0, sp);
        const size_t from = sp == std::string::npos ? line.size() : sp;
        if (key == "svm_type") ok_type = line.find("c_svc") != std::string::npos;
        else if (key == "kernel_type") ok_kernel = line.find("rbf") != std::string::npos;
        else if (key == "gamma") gamma = std::strtod(line.c_str() + from, nullptr);
        else if (key == "nr_class") { const long v = std::strtol(line.c_str() + from, nullptr, 10); k = v < 0 || v > 125 ? -1 : (int)v; }      // (k_svm_couple keeps k (k - 1) / 2 + 3 k doubles in 64 KB of LDS)
        // (every support vector is a line of the text: a count beyond the text's length is a damaged header, not a table to allocate)
        else if (key == "total_sv") { const long v = std::strtol(line.c_str() + from, nullptr, 10); l = v < 0 || (unsigned long)v > len || v > (1L << 28) ? -1 : (int)v; }
        else if (key == "rho") numbers(line, from, rho);
        else if (key == "probA") numbers(line, from, pa);
        else if (key == "probB") numbers(line, from, pb);
        else if (key == "label") numbers(line, from, lab);
        else if (key == "nr_sv") numbers(line, from, nsv);
        else if (key == "SV") { have_sv = true; break; }
    }
    if (!ok_type || !ok_kernel) return fail(c, STR_ER_EFORMAT, "svm model: only svm_type c_svc with kernel_type rbf is supported");
    const int np = k * (k - 1) / 2;
    if (!have_sv || k < 2 || k > 125 || l < 1 || (int)rho.size() != np || (int)pa.size() != np || (int)pb.size() != np ||
        (int)lab.size() != k || (int)nsv.size() != k)
        return fail(c, STR_ER_EFORMAT, "svm model: incomplete header (need nr_class<=125, rho, label, probA, probB, nr_sv)");
    const int dpad = (int)align_up((size_t)dim, 16), l_pad = (int)align_up((size_t)l, 64);
    std::vector<float> sv((size_t)l_pad * dpad, 0.f);
    std::vector<double> sv_exact((size_t)l * dim, 0.0);      // as parsed (the byte form of the vectors is decided on these)
    std::vector<double> svnorm(l_pad, 0.0), coef((size_t)(k - 1) * l, 0.0);
    for (int i = 0; i < l; ++i) {
        if (!next_line(line)) return fail(c, STR_ER_EFORMAT, "svm model: fewer SV lines than total_sv");
        const char *p = line.c_str();
        char *end;
        for (int j = 0; j < k - 1; ++j) { coef[(size_t)j * l + i] = std::strtod(p, &end); if (end == p) return fail(c, STR_ER_EFORMAT, "svm model: bad SV line"); p = end; }
        double nrm = 0;
        for (;;) {
            const long idx = std::strtol(p, &end, 10);
            if (end == p || *end != ':') break;
            p = end + 1;
            const double v = std::strtod(p, &end);
            p = end;
            if (idx < 0 || idx >= dim) return fail(c, STR_ER_EFORMAT, "svm model: SV feature index outside [0, dim)");
            sv[(size_t)i * dpad + idx] = (float)v;
            sv_exact[(size_t)i * dim + idx] = v;
            nrm += (double)(float)v * (double)(float)v;       // (of the f32 value the kernels multiply with: |x - sv|^2 = |x|^2 + |sv|^2 - 2 x.sv stays consistent)
        }
        svnorm[i] = nrm;
    }
    std::vector<int32_t> ilab(k), insv(k), start(k);
    int tot = 0;
    for (int i = 0; i < k; ++i) {
        if (!to_int(lab[i], ilab[i]) || !to_int(nsv[i], insv[i]) || insv[i] < 0 || insv[i] > l) return fail(c, STR_ER_EFORMAT, "svm model: label / nr_sv entries are not counts");
        start[i] = tot; tot += insv[i];
    }
    if (tot != l) return fail(c, STR_ER_EFORMAT, "svm model: nr_sv does not add up to total_sv");
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // coef_t[q][b] = sv_coef[b][q]: the decision values read one coalesced row per support vector (k_svm_couple)
    const int kc = (int)align_up((size_t)(k - 1), 64);
    std::vector<double> coef_t((size_t)l_pad * kc, 0.0);      // (l_pad rows: the walk reads eight rows at a time)
    for (int j = 0; j < k - 1; ++j) for (int i = 0; i < l; ++i) coef_t[(size_t)i * kc + j] = coef[(size_t)j * l + i];
    // what a lane of k_svm_couple needs about its class pair in one 32-byte record; padded to whole passes of the wave with pairs that have no support vectors
    const size_t np_pad = align_up((size_t)np, 512);
    std::vector<SvmPair> pairs(np_pad, SvmPair{0, 0, 0, 0, 0, 1, 0, 0});
    for (int i = 0, p = 0; i < k; ++i)
        for (int j = i + 1; j < k; ++j, ++p)       // (an empty last class starts at l: keep the idle loads inside the tables)
            pairs[(size_t)p] = SvmPair{std::min(start[i], l - 1), insv[i], std::min(start[j], l - 1), insv[j], i, j, 0, 0};
    const size_t o_pij = take(np_pad * sizeof(SvmPair));
    // k <= 65: the coefficients as k_svm_couple's row passes read them (ocr_kernels.h)
    int msv = 0;
    for (int i = 0; i < k; ++i) msv = std::max(msv, insv[i]);
    const int mp = svm_rows_per_class(msv);
    std::vector<double> rows;
    if (k <= 65) {
        if ((size_t)k * 2 * mp * 64 * 8 > ((size_t)1 << 30))
            return fail(c, STR_ER_EFORMAT, "svm model: a class with so many support vectors that the per-class coefficient rows exceed 1 GB");
        rows = svm_coef_rows(coef, start, insv, k, l, mp);
    }
    const size_t o_rows = take(rows.size() * 8);
    // the support vectors in three bf16 pieces (k_svm_kernel_q; svm_tables.h)
    const int dq = (int)align_up((size_t)dim, 64);
    const std::vector<uint16_t> svq = svm_sv_planes(sv, l, l_pad, dim, dpad, dq);
    const size_t o_svq = take(svq.size() * 2);
    // ... and as bytes, if they are 8-bit numerators over 255 (the reference's models: k_svm_kernel_i8)
    const int dq8 = (int)align_up((size_t)dim, 128);
    std::vector<uint8_t> sv8; std::vector<int32_t> sv8s;
    const bool bytes_ok = svm_sv_bytes(sv_exact, l, l_pad, dim, dq8, sv8, sv8s);
    const size_t o_sv8 = take(sv8.size()), o_sv8s = take(sv8s.size() * 4);
    const size_t o_sv = take(sv.size() * 4), o_nrm = take(svnorm.size() * 8), o_coef = take(coef.size() * 8), o_coeft = take(coef_t.size() * 8), o_rho = take(np * 8),
                 o_pa = take(np * 8), o_pb = take(np * 8), o_lab = take(k * 4), o_nsv = take(k * 4), o_start = take(k * 4);
    std::vector<uint8_t> blob(off);
    std::memcpy(&blob[o_svq], svq.data(), svq.size() * 2);
    if (bytes_ok) { std::memcpy(&blob[o_sv8], sv8.data(), sv8.size()); std::memcpy(&blob[o_sv8s], sv8s.data(), sv8s.size() * 4); }
    std::memcpy(&blob[o_sv], sv.data(), sv.size() * 4); std::memcpy(&blob[o_nrm], svnorm.data(), svnorm.size() * 8);
    std::memcpy(&blob[o_pij], pairs.data(), np_pad * sizeof(SvmPair));
    if (!rows.empty()) std::memcpy(&blob[o_rows], rows.data(), rows.size() * 8);
    std::memcpy(&blob[o_coef], coef.data(), coef.size() * 8); std::memcpy(&blob[o_coeft], coef_t.data(), coef_t.size() * 8); std::memcpy(&blob[o_rho], rho.data(), np * 8);
    std::memcpy(&blob[o_pa], pa.data(), np * 8); std::memcpy(&blob[o_pb], pb.data(), np * 8);
    std::memcpy(&blob[o_lab], ilab.data(), k * 4); std::memcpy(&blob[o_nsv], insv.data(), k * 4); std::memcpy(&blob[o_start], start.data(), k * 4);
    void *d = nullptr;
    HIP_TRY(c, hipMalloc(&d, blob.size()));
    hipError_t e = hipMemcpy(d, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail(c, STR_ER_EHIP, std::string("svm upload: ") + hipGetErrorString(e)); }
    if (c->d_svm_blob) (void)hipFree(c->d_svm_blob);
    c->d_svm_blob = d;
    const uint8_t *b = static_cast<const uint8_t *>(d);
    SvmDev m{};
    m.k = k; m.l = l; m.l_pad = l_pad; m.dim = dim; m.dpad = dpad; m.gamma = gamma;
    m.svq = reinterpret_cast<const uint16_t *>(b + o_svq); m.dq = dq;
    m.sv8 = bytes_ok ? b + o_sv8 : nullptr; m.sv8s = bytes_ok ? reinterpret_cast<const int32_t *>(b + o_sv8s) : nullptr; m.dq8 = dq8;
    m.sv = reinterpret_cast<const float *>(b + o_sv); m.svnorm = reinterpret_cast<const double *>(b + o_nrm);
    m.pairs = reinterpret_cast<const SvmPair *>(b + o_pij);
    m.coef_rows = rows.empty() ? nullptr : reinterpret_cast<const double *>(b + o_rows); m.msv = msv; m.mp = mp;
    m.kc = kc; m.coef_t = reinterpret_cast<const double *>(b + o_coeft);
    m.coef = reinterpret_cast<const double *>(b + o_coef); m.rho = reinterpret_cast<const double *>(b + o_rho);
    m.probA = reinterpret_cast<const double *>(b + o_pa); m.probB = reinterpret_cast<const double *>(b + o_pb);
    m.label = reinterpret_cast<const int32_t *>(b + o_lab); m.nsv = reinterpret_cast<const int32_t *>(b + o_nsv);
    m.start = reinterpret_cast<const int32_t *>(b + o_start);
    c->svm = m;
    c->svm_loaded = true;
    return STR_ER_OK;
} ABI_GUARD(c)

int str_er_load_svm_model(str_er_ctx *c, const char *path, int32_t dim)
try {
    if (!c) return STR_ER_EINVAL;
    if (!path) return fail(c, STR_ER_EINVAL, "null path");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(c, STR_ER_EIO, std::string("cannot open ") + path);      // reference: svm_load_model returns NULL (src/svm.cpp:2878-2879)
    std::string buf;
    char tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, n);
    std::fclose(f);
    return str_er_load_svm_model_mem(c, buf.data(), buf.size(), dim);
} ABI_GUARD(c)

int str_er_svm_info(const str_er_ctx *c, int32_t *nr_class, int32_t *total_sv, int32_t *dim)
try {
    if (!c) return STR_ER_EINVAL;
    if (nr_class) *nr_class = c->svm_loaded ? c->svm.k : 0;
    if (total_sv) *total_sv = c->svm_loaded ? c->svm.l : 0;
    if (dim) *dim = c->svm_loaded ? c->svm.dim : 0;
    return STR_ER_OK;
} ABI_GUARD(const_cast<str_er_ctx *>(c))

int str_er_svm_forms(const str_er_ctx *c, int32_t *bytes, int32_t *class_sums)
try {
    if (!c) return STR_ER_EINVAL;
    if (bytes) *bytes = c->svm_loaded && c->svm.sv8 ? 1 : 0;
    if (class_sums) *class_sums = c->svm_loaded && svm_uses_class_sums(c->svm) ? 1 : 0;
    return STR_ER_OK;
} ABI_GUARD(const_cast<str_er_ctx *>(c))

int str_er_svm_predict_probability(str_er_ctx *c, const double *x, int32_t n, int32_t dim, int32_t *label, double *prob, double *dec)
try {
    if (!c) return STR_ER_EINVAL;
    if (n < 0 || (n > 0 && (!x || !label || !prob))) return fail(c, STR_ER_EINVAL, "bad arguments");
    if (!c->svm_loaded) return fail(c, STR_ER_ESTATE, "svm model not loaded");
    if (dim != c->svm.dim) return fail(c, STR_ER_EINVAL, "feature dimension differs from the one the model was loaded with");
    if (n == 0) return STR_ER_OK;
    HIP_TRY(c, hipSetDevice(c->prm.device));
    const SvmDev &m = c->svm;
    const size_t np = (size_t)m.k * (m.k - 1) / 2;
    const size_t o_x = 0, o_buf = align_up((size_t)n * dim * 8, 256);
    int rc = ensure_scratch(c, o_buf + ocr_layout(nullptr, (size_t)n, &m, false, dec != nullptr, true).bytes);
    if (rc != STR_ER_OK) return rc;
    uint8_t *s = static_cast<uint8_t *>(c->d_scratch);
    const OcrBuf buf = ocr_layout(s + o_buf, (size_t)n, &m, false, dec != nullptr, true);
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpyAsync(s + o_x, x, (size_t)n * dim * 8, hipMemcpyHostToDevice, st));
    launch_svm_prep(st, reinterpret_cast<const double *>(s + o_x), n, dim, buf, m);
    launch_svm_score(st, n, buf, m, false);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(prob, buf.prob, (size_t)n * m.k * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(label, buf.label, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (dec) HIP_TRY(c, hipMemcpyAsync(dec, buf.dec, (size_t)n * np * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, wait_stream(c, st));
    return STR_ER_OK;
} ABI_GUARD(c)

int str_er_ocr_chain_run(str_er_ctx *c, const uint8_t *plane, int32_t w, int32_t h, int64_t stride, const int32_t *boxes, int32_t n,
                         int32_t *label, double *prob, uint8_t *q_out)
try {
    return str_er_ocr_chain_run_slope(c, plane, w, h, stride, boxes, nullptr, n, label, prob, q_out);
} ABI_GUARD(c)

int str_er_ocr_chain_run_slope(str_er_ctx *c, const uint8_t *plane, int32_t w, int32_t h, int64_t stride, const int32_t *boxes,
                               const double *slope, int32_t n, int32_t *label, double *prob, uint8_t *q_out)
try {
    if (!c) return STR_ER_EINVAL;
    if (!plane || w < 1 || h < 1 || stride < w || n < 0 || (n > 0 && !boxes)) return fail(c, STR_ER_EINVAL, "bad arguments");
    const bool want_svm = label != nullptr || prob != nullptr;
    if (want_svm && (!label || !prob)) return fail(c, STR_ER_EINVAL, "label and prob must be given together");
    if (want_svm && !c->svm_loaded) return fail(c, STR_ER_ESTATE, "svm model not loaded");
    if (want_svm && c->svm.dim != 1800) return fail(c, STR_ER_ESTATE, "chain_run needs a model loaded with dim = 1800 (8 x 15 x 15)");
    for (int i = 0; i < n; ++i) {
        const int32_t *b = boxes + 4 * (size_t)i;
        if (b[2] < 1 || b[3] < 1 || b[0] < 0 || b[1] < 0 || (int64_t)b[0] + b[2] > w || (int64_t)b[1] + b[3] > h)
            return fail(c, STR_ER_EINVAL, "box " + std::to_string(i) + " outside the plane");
    }
    if (n == 0) return STR_ER_OK;
    HIP_TRY(c, hipSetDevice(c->prm.device));
    if ((size_t)w * h > c->pix_bytes) return fail(c, STR_ER_ECAPACITY, "plane larger than the context capacity");
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpy2DAsync(c->d_pix, (size_t)w, plane, (size_t)stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, st));
    const SvmDev *m = want_svm ? &c->svm : nullptr;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_rot = take(slope ? sizeof(RotGeom) * (size_t)n : 0), o_box = take(16 * (size_t)n), o_buf = take(0);
    int rc = ensure_scratch(c, o_buf + ocr_layout(nullptr, (size_t)n, m, true, false, false).bytes);
    if (rc != STR_ER_OK) return rc;
    uint8_t *s = static_cast<uint8_t *>(c->d_scratch);
    const OcrBuf buf = ocr_layout(s + o_buf, (size_t)n, m, true, false, false);
    HIP_TRY(c, hipMemcpyAsync(s + o_box, boxes, 16 * (size_t)n, hipMemcpyHostToDevice, st));
    std::vector<RotGeom> rot;
    if (slope) {
        rot.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            if (!std::isfinite(slope[i])) return fail(c, STR_ER_EINVAL, "slope " + std::to_string(i) + " is not finite");
            rot[(size_t)i] = make_rot_geom(boxes[4 * (size_t)i + 2], boxes[4 * (size_t)i + 3], slope[i]);
        }
        HIP_TRY(c, hipMemcpyAsync(s + o_rot, rot.data(), sizeof(RotGeom) * (size_t)n, hipMemcpyHostToDevice, st));
    }
    OcrSrc src{};
    src.plane = c->d_pix; src.stride = w; src.inv = 0; src.boxes = reinterpret_cast<const int32_t *>(s + o_box);
    src.rot = slope ? reinterpret_cast<const RotGeom *>(s + o_rot) : nullptr;
    launch_ocr_features(st, src, n, buf, m);
    if (want_svm) {
        // prob = pv[label]; the reference indexes pv with the label itself (src/OCR.cpp:92-93), which k_svm_couple does too (the arg max's entry only for labels 0 .. k - 1 in order)
        launch_svm_score(st, n, buf, *m, true);
        HIP_TRY(c, hipMemcpyAsync(prob, buf.pbest, (size_t)n * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(label, buf.label, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(c, hipGetLastError());
    if (q_out) HIP_TRY(c, hipMemcpyAsync(q_out, buf.q, 1800 * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, wait_stream(c, st));
    return STR_ER_OK;
} ABI_GUARD(c)

int str_er_set_min_ocr_prob(str_er_ctx *c, double p)
try {
    if (!c) return STR_ER_EINVAL;
    if (!(p >= 0.0 && p <= 1.0)) return fail(c, STR_ER_EINVAL, "min_ocr_prob must be in [0, 1]");
    c->min_ocr_prob = p;
    return STR_ER_OK;
} ABI_GUARD(c)

} // extern "C"
