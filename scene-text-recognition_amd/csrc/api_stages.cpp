// api_stages.cpp -- the C ABI, part 4: the single-stage entry points (compute_channels, LBP / classify on explicit boxes, NMS of an uploaded tree, the flood order, resize, er_grouping, calc_color, er_track)
#include "str_er_ctx.h"

extern "C" {

int str_er_compute_channels(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, uint8_t *planes6)
try {
    if (!c) return STR_ER_EINVAL;
    if (!bgr || !planes6 || w < 1 || h < 1 || stride < (int64_t)w * 3) return fail(c, STR_ER_EINVAL, "bad arguments");
    HIP_TRY(c, hipSetDevice(c->prm.device));
    const size_t n = (size_t)w * h;
    if (n * 3 > c->in_bytes || n * 6 > c->pix_bytes) return fail(c, STR_ER_ECAPACITY, "frame larger than the context capacity");
    HIP_TRY(c, hipMemcpy2DAsync(c->d_in, (size_t)w * 3, bgr, (size_t)stride, (size_t)w * 3, (size_t)h, hipMemcpyHostToDevice, c->stream));
    uint8_t *d = c->d_pix;
    launch_bgr_to_ycrcb(c->stream, c->d_in, w, h, (int64_t)w * 3, 0, 1, d, d + n, d + 2 * n, w, 0);
    launch_invert(c->stream, d, d + 3 * n, 3 * n);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(planes6, d, 6 * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, wait_stream(c, c->stream));
    return STR_ER_OK;
} ABI_GUARD(c)

static int boxes_call(str_er_ctx *c, const uint8_t *plane, int32_t w, int32_t h, int64_t stride, const int32_t *boxes, int32_t n,
                      double *hist, uint8_t *tiles, uint8_t *cls, double *ss, double *sw, bool cascades, uint8_t *codes = nullptr)
{
    if (!c) return STR_ER_EINVAL;
    if (!plane || w < 1 || h < 1 || stride < w || n < 0 || (n > 0 && !boxes)) return fail(c, STR_ER_EINVAL, "bad arguments");
    if (cascades && !(c->casc[0].loaded && c->casc[1].loaded)) return fail(c, STR_ER_ESTATE, "classify needs both cascades");
    if (cascades && (!cls || !ss || !sw)) return fail(c, STR_ER_EINVAL, "null output");
    for (int i = 0; i < n; ++i) {
        const int32_t *b = boxes + 4 * (size_t)i;
        if (b[2] < 1 || b[3] < 1 || b[0] < 0 || b[1] < 0 || (int64_t)b[0] + b[2] > w || (int64_t)b[1] + b[3] > h)
            return fail(c, STR_ER_EINVAL, "box " + std::to_string(i) + " outside the plane");
    }
    if (n == 0) return STR_ER_OK;
    HIP_TRY(c, hipSetDevice(c->prm.device));
    const size_t np = (size_t)w * h;
    if (np > c->pix_bytes) return fail(c, STR_ER_ECAPACITY, "plane larger than the context capacity");
    HIP_TRY(c, hipMemcpy2DAsync(c->d_pix, (size_t)w, plane, (size_t)stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, c->stream));
    const size_t o_box = 0, o_hist = align_up(16 * (size_t)n, 256), o_tile = o_hist + 8192 * (size_t)n,
                 o_cls = align_up(o_tile + 676 * (size_t)n, 256), o_ss = align_up(o_cls + (size_t)n, 256), o_sw = o_ss + 8 * (size_t)n,
                 o_code = align_up(o_sw + 8 * (size_t)n, 256), total = o_code + 576 * (size_t)n;
    int rc = ensure_scratch(c, total);
    if (rc != STR_ER_OK) return rc;
    uint8_t *s = static_cast<uint8_t *>(c->d_scratch);
    HIP_TRY(c, hipMemcpyAsync(s + o_box, boxes, 16 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    launch_lbp_boxes(c->stream, c->d_pix, w, h, w, reinterpret_cast<const int32_t *>(s + o_box), n,
                     hist ? reinterpret_cast<double *>(s + o_hist) : nullptr, tiles ? s + o_tile : nullptr, codes ? s + o_code : nullptr, s + o_cls,
                     reinterpret_cast<double *>(s + o_ss), reinterpret_cast<double *>(s + o_sw), c->casc[0].dev, c->casc[1].dev,
                     cascades ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
    if (hist) HIP_TRY(c, hipMemcpyAsync(hist, s + o_hist, 8192 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (tiles) HIP_TRY(c, hipMemcpyAsync(tiles, s + o_tile, 676 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (codes) HIP_TRY(c, hipMemcpyAsync(codes, s + o_code, 576 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (cascades) {
        HIP_TRY(c, hipMemcpyAsync(cls, s + o_cls, (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(ss, s + o_ss, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(sw, s + o_sw, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, wait_stream(c, c->stream));
    return STR_ER_OK;
}

int str_er_classify_boxes(str_er_ctx *c, const uint8_t *plane, int32_t w, int32_t h, int64_t stride, const int32_t *boxes,
                          int32_t n, uint8_t *cls, double *score_strong, double *score_weak)
try {
    return boxes_call(c, plane, w, h, stride, boxes, n, nullptr, nullptr, cls, score_strong, score_weak, true);
} ABI_GUARD(c)

int str_er_lbp_hist(str_er_ctx *c, const uint8_t *plane, int32_t w, int32_t h, int64_t stride, const int32_t *boxes, int32_t n,
                    double *hist, uint8_t *tiles26)
try {
    if (c && !hist) return fail(c, STR_ER_EINVAL, "null hist");
    return boxes_call(c, plane, w, h, stride, boxes, n, hist, tiles26, nullptr, nullptr, nullptr, false);
} ABI_GUARD(c)

int str_er_calc_lbp(str_er_ctx *c, const uint8_t *plane, int32_t w, int32_t h, int64_t stride, const int32_t *boxes, int32_t n, uint8_t *lbp24)
try {
    if (c && !lbp24) return fail(c, STR_ER_EINVAL, "null lbp24");
    return boxes_call(c, plane, w, h, stride, boxes, n, nullptr, nullptr, nullptr, nullptr, nullptr, false, lbp24);
} ABI_GUARD(c)

static int nms_tree_impl(str_er_ctx *c, const str_er_node *nodes, int32_t n_nodes, const uint8_t *plane, int64_t stride, int32_t rows, int32_t cols,
                         int32_t *pool_idx, int32_t cap, int32_t *n_pool, int32_t *ambiguous)
{
    if (!c) return STR_ER_EINVAL;
    if (!nodes || n_nodes < 1 || rows < 1 || cols < 1 || !n_pool || (cap > 0 && !pool_idx) || cap < 0 || (plane && stride < cols))
        return fail(c, STR_ER_EINVAL, "bad arguments");
    if (plane && (size_t)rows * (size_t)cols > c->pix_bytes) return fail(c, STR_ER_ECAPACITY, "plane larger than the context capacity");
    if (!c->auto_caps && n_nodes > c->kept_cap) return fail(c, STR_ER_ECAPACITY, "tree larger than kept_cap");
    HIP_TRY(c, hipSetDevice(c->prm.device));
    std::vector<uint32_t> key(n_nodes), area(n_nodes); std::vector<int32_t> par(n_nodes);
    std::vector<uint16_t> box(4 * (size_t)n_nodes); std::vector<uint8_t> lev(n_nodes);
    int root = -1, maxl = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const str_er_node &n = nodes[i];
        int p = n.parent;
        if (p < 0 || p == i) { if (root >= 0) return fail(c, STR_ER_EINVAL, "tree has more than one root"); root = i; p = i; }
        if (p >= n_nodes) return fail(c, STR_ER_EINVAL, "parent index out of range");
        if (n.w < 1 || n.h < 1) return fail(c, STR_ER_EINVAL, "empty box");
        if (plane && n.key >= (uint32_t)rows * (uint32_t)cols) return fail(c, STR_ER_EINVAL, "node key outside the plane");
        key[i] = n.key; area[i] = (uint32_t)n.area; par[i] = p; lev[i] = n.level;
        box[4 * (size_t)i] = n.x; box[4 * (size_t)i + 1] = n.y; box[4 * (size_t)i + 2] = n.w; box[4 * (size_t)i + 3] = n.h;
        maxl = std::max(maxl, (int)n.level);
    }
    if (root < 0) return fail(c, STR_ER_EINVAL, "tree has no root");
    for (int i = 0; i < n_nodes; ++i)
        if (i != root && lev[par[i]] <= lev[i]) return fail(c, STR_ER_EINVAL, "parent level must exceed child level");
    hipStream_t s = c->stream;
    Batch b;
    add_plane(b, c->d_pix, cols, rows, cols, 0, 0, 0, 0);
    b.kept_floor = b.pool_floor = (uint32_t)n_nodes;        // (the imported tree is the plane's kept-node table)
    assign_tables(b, c);
    if (b.kept > c->kept_total || b.pool > c->pool_total) {
        const int rct = alloc_tables(c, std::max(c->kept_total, b.kept), std::max(c->pool_total, b.pool));
        if (rct != STR_ER_OK) return rct;
    }
    std::memcpy(c->h_planes, b.planes.data(), sizeof(PlaneDesc));
    PlaneCtr pc{};
    pc.n_kept = (uint32_t)n_nodes; pc.root_slot = (uint32_t)root; pc.max_level = (uint32_t)maxl;
    c->h_ctr[0] = pc;
    c->zero_clean_bytes = 0;          // (the counter block is written here, not through upload_layout)
    c->planes_on_device = 0;
    HIP_TRY(c, hipMemcpyAsync(c->d_planes, c->h_planes, sizeof(PlaneDesc), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_ctr, c->h_ctr, sizeof(PlaneCtr), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->ka.key, key.data(), 4 * (size_t)n_nodes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->ka.area, area.data(), 4 * (size_t)n_nodes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->ka.parent, par.data(), 4 * (size_t)n_nodes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->ka.box, box.data(), 8 * (size_t)n_nodes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->ka.level, lev.data(), (size_t)n_nodes, hipMemcpyHostToDevice, s));
    if (plane) HIP_TRY(c, hipMemcpy2DAsync(c->d_pix, (size_t)cols, plane, (size_t)stride, (size_t)cols, (size_t)rows, hipMemcpyHostToDevice, s));
    HIP_TRY(c, wait_stream(c, s)); // host vectors go out of scope after this call
    BatchDev bd = make_batchdev(c, b);
    bd.n_seam_blocks = 0;
    const DetectParams dp = make_dp(c);
    launch_nms(s, bd, dp, /*use_index_order=*/plane == nullptr);
    if (plane) launch_nms_alt(s, bd, dp, c->d_alt_list);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(PlaneCtr), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, wait_stream(c, s));
    const uint32_t n_amb = c->h_ctr[0].n_amb;
    if (plane && c->prm.sibling_order == 0 && c->h_ctr[0].n_rel) {      // ties: the reference's flood order decides (k_flood_order)
        bool replayed = false;
        const int rcr = resolve_sibling_ties(c, b, bd, dp, replayed, /*from_tree=*/true);
        if (rcr != STR_ER_OK) return rcr;
        if (c->prio) HIP_TRY(c, wait_stream(c, c->prio));       // the tie pass ran there
        HIP_TRY(c, hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(PlaneCtr), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, wait_stream(c, s));
    }
    if (c->h_ctr[0].overflow & 2u) return fail(c, STR_ER_ECAPACITY, "NMS pool overflow: raise pool_cap");
    const int np = (int)c->h_ctr[0].n_pool;
    *n_pool = np;
    if (ambiguous) *ambiguous = (int32_t)n_amb;
    const int ncopy = std::min(np, cap);
    if (ncopy > 0) {
        HIP_TRY(c, hipMemcpyAsync(pool_idx, c->d_pool, 4 * (size_t)ncopy, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, wait_stream(c, s));
    }
    return STR_ER_OK;
}

int str_er_nms_tree(str_er_ctx *c, const str_er_node *nodes, int32_t n_nodes, int32_t rows, int32_t cols, int32_t *pool_idx,
                    int32_t cap, int32_t *n_pool, int32_t *ambiguous)
try {
    return nms_tree_impl(c, nodes, n_nodes, nullptr, 0, rows, cols, pool_idx, cap, n_pool, ambiguous);
} ABI_GUARD(c)

int str_er_nms_tree_plane(str_er_ctx *c, const str_er_node *nodes, int32_t n_nodes, const uint8_t *plane, int32_t cols, int32_t rows,
                          int64_t stride, int32_t *pool_idx, int32_t cap, int32_t *n_pool, int32_t *ambiguous)
try {
    if (!plane) return c ? fail(c, STR_ER_EINVAL, "null plane") : STR_ER_EINVAL;
    return nms_tree_impl(c, nodes, n_nodes, plane, stride, rows, cols, pool_idx, cap, n_pool, ambiguous);
} ABI_GUARD(c)

int str_er_flood_order(const uint8_t *plane, int32_t w, int32_t h, int64_t stride, int32_t thresh_step, uint32_t *stamp)
{
    if (!plane || !stamp || w < 1 || h < 1 || stride < w || thresh_step < 1 || thresh_step > 255 || (int64_t)w * h > (1 << 24)) return STR_ER_EINVAL;
    std::memset(stamp, 0, 4 * (size_t)w * h);
    flood_order_host(plane, w, h, stride, 0, (float)(1.0 / (double)thresh_step), 255 / thresh_step + 1, nullptr, 0xFFFFFFFFu, stamp);
    return STR_ER_OK;
}

int str_er_resize_plane(str_er_ctx *c, const uint8_t *src, int32_t sw, int32_t sh, int64_t sstride, uint8_t *dst, int32_t dw,
                        int32_t dh)
try {
    if (!c) return STR_ER_EINVAL;
    if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1 || sstride < sw) return fail(c, STR_ER_EINVAL, "bad arguments");
    HIP_TRY(c, hipSetDevice(c->prm.device));
    const size_t ns = (size_t)sw * sh, nd = (size_t)dw * dh;
    if (ns > c->in_bytes || nd > c->pix_bytes) return fail(c, STR_ER_ECAPACITY, "plane larger than the context capacity");
    HIP_TRY(c, hipMemcpy2DAsync(c->d_in, (size_t)sw, src, (size_t)sstride, (size_t)sw, (size_t)sh, hipMemcpyHostToDevice, c->stream));
    launch_resize(c->stream, c->d_in, sw, sh, sw, 0, 0, c->d_pix, dw, dh, dw, 0, 0, 1, 1);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(dst, c->d_pix, nd, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, wait_stream(c, c->stream));
    return STR_ER_OK;
} ABI_GUARD(c)

// ---- results ---------------------------------------------------------------------------------
int32_t str_er_result_n_planes(const str_er_result *r) { return r ? (int32_t)r->planes.size() : 0; }

int str_er_er_grouping(str_er_ctx *c, const str_er_cand *cands, const str_er_track *tracks, int32_t n, int overlap_sup, int inner_sup,
                       str_er_result **out)
try {
    if (!c) return STR_ER_EINVAL;
    if (!out || n < 0 || (n > 0 && (!cands || !tracks))) return fail(c, STR_ER_EINVAL, "bad arguments");
    *out = nullptr;
    HIP_TRY(c, hipSetDevice(c->prm.device));
    str_er_result *r = new (std::nothrow) str_er_result();
    if (!r) return fail(c, STR_ER_ENOMEM, "result allocation");
    r->cands.assign(cands, cands + n);
    r->tracks.assign(tracks, tracks + n);
    r->have_tracks = true;
    r->cand_off.assign(2, 0); r->cand_off[1] = (uint32_t)n;
    r->planes.resize(1);
    std::memset(&r->planes[0], 0, sizeof(str_er_plane_info));
    r->planes[0].n_pool = n; r->planes[0].root = -1;
    int rc = STR_ER_OK;
    if (n > 0 && overlap_sup) {
        rc = group_phase_overlap(c, std::vector<uint32_t>{0u, (uint32_t)n}, inner_sup != 0, r);
    } else if (n > 0) {
        const size_t o_c = 0, o_tr = align_up(sizeof(CandRec) * (size_t)n, 256);
        rc = ensure_scratch(c, o_tr + sizeof(TrackRec) * (size_t)n);
        if (rc == STR_ER_OK) {
            uint8_t *sc = static_cast<uint8_t *>(c->d_scratch);
            hipError_t e = hipMemcpyAsync(sc + o_c, cands, sizeof(CandRec) * (size_t)n, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(sc + o_tr, tracks, sizeof(TrackRec) * (size_t)n, hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) rc = fail(c, STR_ER_EHIP, hipGetErrorString(e));
            else rc = group_phase(c, reinterpret_cast<const CandRec *>(sc + o_c), reinterpret_cast<const TrackRec *>(sc + o_tr),
                                  std::vector<uint32_t>{0u, (uint32_t)n}, inner_sup != 0, r);
        }
    } else {
        r->have_texts = true;
    }
    if (rc != STR_ER_OK) { delete r; return rc; }
    *out = r;
    return STR_ER_OK;
} ABI_GUARD(c)

int str_er_calc_color(str_er_ctx *c, const uint8_t *mask_plane, int32_t w, int32_t h, int64_t stride, const uint8_t *color_img, int32_t cw,
                      int32_t ch, int64_t cstride, const int32_t *boxes, int32_t n, double *colors)
try {
    if (!c) return STR_ER_EINVAL;
    if (!mask_plane || !color_img || w < 1 || h < 1 || stride < w || cw < 1 || ch < 1 || cstride < (int64_t)cw * 3 || n < 0 ||
        (n > 0 && (!boxes || !colors)))
        return fail(c, STR_ER_EINVAL, "bad arguments");
    for (int i = 0; i < n; ++i) {
        const int32_t *b = boxes + 4 * (size_t)i;
        if (b[2] < 1 || b[3] < 1 || b[0] < 0 || b[1] < 0 || (int64_t)b[0] + b[2] > w || (int64_t)b[1] + b[3] > h)
            return fail(c, STR_ER_EINVAL, "box " + std::to_string(i) + " outside the plane");
        if (b[2] > cw || b[3] > ch) return fail(c, STR_ER_EINVAL, "box " + std::to_string(i) + " larger than the colour image");
    }
    if (n == 0) return STR_ER_OK;
    HIP_TRY(c, hipSetDevice(c->prm.device));
    if ((size_t)w * h > c->pix_bytes || (size_t)cw * 3 * ch > c->in_bytes) return fail(c, STR_ER_ECAPACITY, "image larger than the context capacity");
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpy2DAsync(c->d_pix, (size_t)w, mask_plane, (size_t)stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpy2DAsync(c->d_in, (size_t)cw * 3, color_img, (size_t)cstride, (size_t)cw * 3, (size_t)ch, hipMemcpyHostToDevice, st));
    const size_t o_box = 0, o_tr = align_up(16 * (size_t)n, 256), o_cs = align_up(o_tr + sizeof(TrackRec) * (size_t)n, 256);
    int rc = ensure_scratch(c, o_cs + calc_color_scratch_bytes((size_t)n));
    if (rc != STR_ER_OK) return rc;
    uint8_t *sc = static_cast<uint8_t *>(c->d_scratch);
    HIP_TRY(c, hipMemcpyAsync(sc + o_box, boxes, 16 * (size_t)n, hipMemcpyHostToDevice, st));
    ColorSrc col{c->d_in, c->d_in + 1, c->d_in + 2, 3, (int64_t)cw * 3};
    OcrSrc src{};
    src.plane = c->d_pix; src.stride = w; src.inv = 0; src.boxes = reinterpret_cast<const int32_t *>(sc + o_box);
    launch_calc_color(st, src, col, n, reinterpret_cast<TrackRec *>(sc + o_tr), sc + o_cs);
    HIP_TRY(c, hipGetLastError());
    std::vector<TrackRec> tr((size_t)n);
    HIP_TRY(c, hipMemcpyAsync(tr.data(), sc + o_tr, sizeof(TrackRec) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, wait_stream(c, st));
    for (int i = 0; i < n; ++i) { colors[3 * (size_t)i] = tr[(size_t)i].color1; colors[3 * (size_t)i + 1] = tr[(size_t)i].color2; colors[3 * (size_t)i + 2] = tr[(size_t)i].color3; }
    return STR_ER_OK;
} ABI_GUARD(c)

int str_er_er_track(str_er_ctx *c, const str_er_cand *cands, const double *colors, int32_t n, uint8_t *tracked, int32_t *cx, int32_t *cy)
try {
    if (!c) return STR_ER_EINVAL;
    if (n < 0 || (n > 0 && (!cands || !colors || !tracked))) return fail(c, STR_ER_EINVAL, "bad arguments");
    if (n == 0) return STR_ER_OK;
    HIP_TRY(c, hipSetDevice(c->prm.device));
    hipStream_t st = c->stream;
    const size_t o_c = 0, o_tr = align_up(sizeof(CandRec) * (size_t)n, 256), o_list = align_up(o_tr + sizeof(TrackRec) * (size_t)n, 256),
                 o_rng = align_up(o_list + 4 * (size_t)n, 256);
    int rc = ensure_scratch(c, o_rng + 64);
    if (rc != STR_ER_OK) return rc;
    uint8_t *sc = static_cast<uint8_t *>(c->d_scratch);
    std::vector<TrackRec> tr((size_t)n);
    for (int i = 0; i < n; ++i) {
        TrackRec t{};
        t.color1 = colors[3 * (size_t)i]; t.color2 = colors[3 * (size_t)i + 1]; t.color3 = colors[3 * (size_t)i + 2];
        tr[(size_t)i] = t;
    }
    const uint32_t rng[2] = {0u, (uint32_t)n};
    HIP_TRY(c, hipMemcpyAsync(sc + o_c, cands, sizeof(CandRec) * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(sc + o_tr, tr.data(), sizeof(TrackRec) * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(sc + o_rng, rng, sizeof(rng), hipMemcpyHostToDevice, st));
    launch_er_track(st, reinterpret_cast<const CandRec *>(sc + o_c), reinterpret_cast<TrackRec *>(sc + o_tr),
                    reinterpret_cast<uint32_t *>(sc + o_list), reinterpret_cast<const uint32_t *>(sc + o_rng), 1);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(tr.data(), sc + o_tr, sizeof(TrackRec) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, wait_stream(c, st));
    for (int i = 0; i < n; ++i) {
        tracked[i] = (uint8_t)tr[(size_t)i].tracked;
        if (cx) cx[i] = tr[(size_t)i].cx;
        if (cy) cy[i] = tr[(size_t)i].cy;
    }
    return STR_ER_OK;
} ABI_GUARD(c)

} // extern "C"
