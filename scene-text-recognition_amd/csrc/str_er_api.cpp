// str_er_api.cpp -- implementation of the C ABI in include/str_er.h.
//
// Host side of the hot path: owns the device workspace, lays out one batch of planes
// (frames x channels x pyramid levels), enqueues the gfx950 kernels of er_kernels.hip on
// one HIP stream with no host synchronisation in between, then copies back the per-plane
// counters and the packed candidate records.  No computation of the path happens on the
// host; if the device or a kernel fails the call fails (there is no CPU fallback).
#include "str_er_ctx.h"
#include <condition_variable>
#include <mutex>

thread_local std::string g_create_error;

namespace str_er_host {

// Node records: every plane gets `share` records per padded pixel (+ a floor for tiny planes), never more than one per pixel.
size_t plane_node_cap(int tiles, double share)
{
    const size_t px = (size_t)tiles * TILE_PX;
    return std::min<size_t>(px, (size_t)std::ceil((double)px * share) + 256);
}
void assign_node_records(Batch &b, double share)
{
    b.nodes = 0;
    for (PlaneDesc &d : b.planes) {
        d.node_base = (uint32_t)b.nodes;
        d.node_cap = (uint32_t)plane_node_cap(d.tiles_x * d.tiles_y, share);
        b.nodes += d.node_cap;
    }
}

// Kept-node and pool tables.  Explicit caps (str_er_params) are given to every plane; by default a plane's share follows its padded
// pixel count -- a 240 x 135 pyramid level does not need the table of a 1920 x 1080 plane -- and grows with the context's shares.
// groups of gx x gy tiles per plane (k_group_merge), numbered batch-wide
void assign_groups(Batch &b, int gx, int gy)
{
    // (a half-given or oversized group shape -- STR_ER_GROUP_X without _Y, a negative value, more than 64 tiles -- means NO grouping: with
    // group_x > 0 and group_y = 0 k_seam divided by zero on the device)
    if (gx <= 0 || gy <= 0 || gx * gy > 64) gx = gy = 0;
    b.group_x = gx; b.group_y = gy; b.n_groups = 0;
    for (PlaneDesc &d : b.planes) {
        d.group_base = b.n_groups;
        if (gx > 0 && gy > 0) b.n_groups += (uint32_t)((d.tiles_x + gx - 1) / gx) * (uint32_t)((d.tiles_y + gy - 1) / gy);
    }
}

void assign_tables(Batch &b, const str_er_ctx *c)
{
    b.kept = b.pool = 0;
    for (PlaneDesc &d : b.planes) {
        const size_t px = (size_t)d.tiles_x * d.tiles_y * TILE_PX;
        if (c->auto_caps) {
            d.kept_cap = (uint32_t)std::min<size_t>(px + 1, (size_t)std::ceil((double)px * c->kept_share) + 512);
            d.kept_cap = std::max(d.kept_cap, b.kept_floor);
            d.pool_cap = (uint32_t)std::min<size_t>(px + 1, (size_t)std::ceil((double)px * c->pool_share) + 256);
            d.pool_cap = std::max(d.pool_cap, b.pool_floor);
        } else {
            d.kept_cap = (uint32_t)c->kept_cap;
            d.pool_cap = (uint32_t)c->pool_cap;
        }
        d.kept_base = (uint32_t)b.kept; b.kept += d.kept_cap;
        d.pool_base = (uint32_t)b.pool; b.pool += d.pool_cap;
    }
}

// (Re)allocate the tables for KP kept nodes and PP pooled ERs in all.
int alloc_tables(str_er_ctx *c, size_t KP, size_t PP)
{
    auto re = [&](auto *&p, size_t n) -> int {
        using T = std::remove_reference_t<decltype(*p)>;
        if (p) { (void)hipFree(p); p = nullptr; }
        void *v = nullptr;
        const size_t bytes = std::max<size_t>(n * sizeof(T), 256);
        if (hipMalloc(&v, bytes) != hipSuccess) return fail(c, STR_ER_ENOMEM, "hipMalloc (kept-node / pool tables, " + std::to_string(bytes) + " bytes)");
        p = static_cast<T *>(v);
        c->ws_bytes += (int64_t)bytes;
        return STR_ER_OK;
    };
    c->ws_bytes -= c->table_bytes;
    const int64_t before = c->ws_bytes;
    c->kept_total = c->pool_total = 0; c->table_bytes = 0;
    int rc = STR_ER_OK;
#define T_(p, n) if (rc == STR_ER_OK) rc = re(p, n)
    T_(c->ka.node, KP); T_(c->ka.key, KP); T_(c->ka.area, KP); T_(c->ka.parent, KP); T_(c->ka.box, 4 * KP); T_(c->ka.level, KP);
    T_(c->ka.start, KP); T_(c->ka.ncand, KP); T_(c->ka.best, KP); T_(c->ka.perm, KP);
    T_(c->d_pool, PP); T_(c->d_pool_tmp, PP); T_(c->d_cands, PP); T_(c->d_cand_plane, PP); T_(c->d_cands2, PP); T_(c->d_cand_plane2, PP);
    T_(c->d_redo, PP + 1); T_(c->d_track, PP); T_(c->d_track_list, PP);
#undef T_
    c->table_bytes = c->ws_bytes - before;
    if (rc != STR_ER_OK) return rc;
    c->kept_total = KP; c->pool_total = PP;
    return STR_ER_OK;
}

BatchDev make_batchdev(str_er_ctx *c, const Batch &b)
{
    BatchDev d{};
    d.planes = c->d_planes; d.ctr = c->d_ctr; d.n_planes = (int32_t)b.planes.size();
    d.n_tiles = b.n_tiles; d.n_pairs = b.n_pairs; d.n_node_blocks = c->n_node_blocks; d.nb_plane = c->d_nb_plane;
    d.tile_plane = c->d_tile_plane; d.seam_block_plane = c->d_sb_plane; d.seam_block_first = c->d_sb_first;
    d.n_seam_blocks = (uint32_t)c->h_sb_plane.size();
    d.tile_nrec = c->d_tile_nrec; d.group_done = c->d_group_done; d.group_plane = c->d_group_plane; d.undone_list = c->d_undone; d.undone_count = c->d_total ? c->d_total + 2 : nullptr; d.n_groups = b.n_groups; d.group_x = b.group_x; d.group_y = b.group_y;
    d.na = c->na; d.ka = c->ka; d.tile_nbase = c->d_tile_nbase; d.seam = c->d_seam; d.pool = c->d_pool; d.pool_tmp = c->d_pool_tmp;
    d.cands = c->d_cands; d.total_cands = c->d_total; d.cand_plane = c->d_cand_plane; d.watch = c->d_watch; d.wstamp = c->d_wstamp; d.wparent = c->d_wparent;
    return d;
}

// The tile trees of a batch: the planes with few levels per tile (the chroma planes of a frame: 2.2 levels and ~15 nodes per tile on text-like
// content) go to k_tile_tree2, the others to k_tile_tree; tiles k_tile_tree2 does not take (too many levels / nodes / records) come back in a
// list that k_tile_tree then walks.  Results do not depend on who built a tile's tree (tests: STR_ER_TILE2=0 / 1 / 2 give the same records).
static bool plane_takes_t2(const str_er_ctx *c, const PlaneDesc &pd, const DetectParams &dp)
{
    if (c->t2_mode == 0 || !c->tile_sparse) return false;                    // (noise-like batches: hundreds of nodes per tile)
    if (dp.hi > 32 || dp.hi < 2 || (dp.thresh_step & (dp.thresh_step - 1)) != 0) return false;      // levels as 5 bits; quantisation by shifts
    if (c->t2_mode == 2) return true;
    return c->t2_backoff == 0 && pd.ch % 3 != 0;
}
static int launch_tile_trees(str_er_ctx *c, const Batch &b, const BatchDev &bd, const DetectParams &dp)
{
    hipStream_t s = c->stream;
    std::vector<uint32_t> key;
    key.reserve(b.planes.size() * 3 + 1);
    bool any = false;
    for (const PlaneDesc &pd : b.planes) {
        const bool t = plane_takes_t2(c, pd, dp);
        any |= t;
        key.push_back((uint32_t)pd.w); key.push_back((uint32_t)pd.h); key.push_back(t ? 1u : 0u);
    }
    if (!any) { launch_tile_tree(s, bd, dp, c->tile_sparse); c->n_t2_tiles = 0; return STR_ER_OK; }
    if (key != c->t2_key) {
        c->t2_key.clear();
        c->h_t1_list.clear(); c->h_t2_pairs.clear();
        uint32_t n2 = 0;
        for (const PlaneDesc &pd : b.planes) {
            if (plane_takes_t2(c, pd, dp)) {
                for (int ty = 0; ty < pd.tiles_y; ++ty)
                    for (int tx = 0; tx < pd.tiles_x; tx += 2)
                        c->h_t2_pairs.push_back((pd.tile_base + (uint32_t)(ty * pd.tiles_x + tx)) | (tx + 1 < pd.tiles_x ? 0x80000000u : 0u));
                n2 += (uint32_t)(pd.tiles_x * pd.tiles_y);
            } else
                for (uint32_t k = 0; k < (uint32_t)(pd.tiles_x * pd.tiles_y); ++k) c->h_t1_list.push_back(pd.tile_base + k);
        }
        if (!c->h_t1_list.empty()) HIP_TRY(c, hipMemcpyAsync(c->d_t1_list, c->h_t1_list.data(), 4 * c->h_t1_list.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->d_t2_pairs, c->h_t2_pairs.data(), 4 * c->h_t2_pairs.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(c, wait_stream(c, s));   // (pageable host vectors)
        c->n_t2_tiles = n2;
        c->t2_key = key;
    }
    if ((int)b.planes.size() <= SPEC_PLANES && c->side && c->ev_fork && c->ev_join) {
        // a call of a frame or two: the two kernels side by side (a wave of k_tile_tree2 is one long dependent chain -- 70 us for a pair of tiles whatever
        // else the GPU does --, and a frame's tiles do not fill the GPU: one after the other they took 130 us, k_tile_tree alone on all tiles 100)
        HIP_TRY(c, hipEventRecord(c->ev_fork, s));
        HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
        launch_tile_tree2(c->side, bd, dp, c->d_t2_pairs, (uint32_t)c->h_t2_pairs.size(), c->d_fb_list, c->d_total + 1);
        HIP_TRY(c, hipEventRecord(c->ev_join, c->side));
        launch_tile_tree(s, bd, dp, c->tile_sparse, c->d_t1_list, (uint32_t)c->h_t1_list.size());
        HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
        launch_tile_tree_fb(s, bd, dp, c->tile_sparse, c->d_fb_list, c->d_total + 1, std::min<uint32_t>(c->n_t2_tiles, 256u));
        return STR_ER_OK;
    }
    // the small kernel first: its waves are the longer ones (a dependent chain per pair of tiles), the big kernel's workgroups fill in behind
    launch_tile_tree2(s, bd, dp, c->d_t2_pairs, (uint32_t)c->h_t2_pairs.size(), c->d_fb_list, c->d_total + 1);
    rec(c, "tile_tree2");
    launch_tile_tree(s, bd, dp, c->tile_sparse, c->d_t1_list, (uint32_t)c->h_t1_list.size());
    launch_tile_tree_fb(s, bd, dp, c->tile_sparse, c->d_fb_list, c->d_total + 1, std::min<uint32_t>(c->n_t2_tiles, 256u));
    return STR_ER_OK;
}

void rec(str_er_ctx *c, const char *name, hipStream_t on, bool always)
{
    if (!always && !c->profiling) return;
    static_assert(str_er_ctx::MAX_EV >= 32, "the fullest call (BGR prologue, both tile kernels, track, group, both scorer stages) records 27 events");
    if (c->n_ev < str_er_ctx::MAX_EV) {
        (void)hipEventRecord(c->ev[c->n_ev], on ? on : c->stream);
        c->profile.emplace_back(name, 0.0);
        ++c->n_ev;
    }
}

// OCR::rotate_mat's canvas for a w x h box (src/OCR.cpp:256-290): corner rounding, crop height and the
// fall-back to the uncropped canvas, evaluated with the host libm exactly as the reference does.
RotGeom make_rot_geom(int w, int h, double slope)
{
    RotGeom g;
    std::memset(&g, 0, sizeof(g));
    if (!(std::fabs(slope) > 0.01)) return g;
    const double rad = std::atan2(slope, 1.0);
    const int    x0 = (int)((w - 1) / 2.0), y0 = (int)((h - 1) / 2.0);
    const int    cx[4] = {0 - x0, (w - 1) - x0, (w - 1) - x0, 0 - x0}, cy[4] = {0 - y0, 0 - y0, (h - 1) - y0, (h - 1) - y0};
    int          nx[4], ny[4];
    for (int k = 0; k < 4; ++k) {
        nx[k] = (int)std::round(cx[k] * std::cos(rad) - cy[k] * std::sin(rad));
        ny[k] = (int)std::round(cx[k] * std::sin(rad) + cy[k] * std::cos(rad));
    }
    g.max_x = std::max(std::max(nx[0], nx[1]), std::max(nx[2], nx[3]));
    g.max_y = std::max(std::max(ny[0], ny[1]), std::max(ny[2], ny[3]));
    g.min_x = std::min(std::min(nx[0], nx[1]), std::min(nx[2], nx[3]));
    g.min_y = std::min(std::min(ny[0], ny[1]), std::min(ny[2], ny[3]));
    g.on = 1;
    g.crop = 1;
    g.ch = (int)((nx[1] - nx[0]) * std::tan(rad) * 0.5);
    if (g.max_y - g.min_y + 1 - 2 * g.ch <= 0) { g.crop = 0; g.ch = 0; }
    g.rw = g.max_x - g.min_x + 1;
    g.rh = g.max_y - g.min_y + 1 - 2 * g.ch;
    g.x0 = x0; g.y0 = y0;
    g.c = std::cos(rad); g.s = std::sin(rad);
    return g;
}


// er_ocr's per-line scoring (src/ER.cpp:695-747) on the lines group_phase left in r: every member is scored by
// OCR::chain_run on its (possibly merged) bound with the line's slope -- one batched launch for all members of all lines --
// then per line, last line first as the reference iterates: members whose boxes overlap by more than 0.95 lose the smaller one
// (:700-721), members below MIN_OCR_PROB go (:737-741), a line with fewer than 2 members left is dropped (:743-747).
int line_ocr_phase(str_er_ctx *c, const PlaneDesc *d_planes, str_er_result *r)
{
    const size_t n_m = r->text_ers.size();
    r->have_line_ocr = true;
    r->line_label.assign(n_m, -1);
    r->line_prob.assign(n_m, 0.0);
    r->line_kept.assign(n_m, 0);
    r->text_alive.assign(r->texts.size(), 0);
    if (n_m == 0) return STR_ER_OK;
    const SvmDev &m = c->svm;
    std::vector<CandRec> recs(n_m);
    std::vector<RotGeom> rot(n_m);
    std::vector<uint32_t> ident(n_m);
    for (size_t t = 0; t < r->texts.size(); ++t) {
        const str_er_text &tx = r->texts[t];
        for (int32_t k = 0; k < tx.count; ++k) {
            const size_t  j = (size_t)tx.first + (size_t)k;
            const int32_t ci = r->text_ers[j];
            CandRec       rc;
            std::memcpy(&rc, &r->cands[(size_t)ci], sizeof(CandRec));
            const str_er_gbound &g = r->gbounds[(size_t)ci];
            rc.x = (uint16_t)g.x; rc.y = (uint16_t)g.y; rc.w = (uint16_t)g.w; rc.h = (uint16_t)g.h;
            recs[j] = rc;
            rot[j] = make_rot_geom(g.w, g.h, tx.slope);
            ident[j] = (uint32_t)j;
        }
    }
    hipStream_t s = c->stream;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_rec = take(sizeof(CandRec) * n_m), o_rot = take(sizeof(RotGeom) * n_m), o_list = take(4 * n_m), o_buf = take(0);
    int rc2 = ensure_scratch(c, o_buf + ocr_layout(nullptr, n_m, &m, false, false, false).bytes);
    if (rc2 != STR_ER_OK) return rc2;
    uint8_t *sc = static_cast<uint8_t *>(c->d_scratch);
    const OcrBuf buf = ocr_layout(sc + o_buf, n_m, &m, false, false, false);
    HIP_TRY(c, hipMemcpyAsync(sc + o_rec, recs.data(), sizeof(CandRec) * n_m, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(sc + o_rot, rot.data(), sizeof(RotGeom) * n_m, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(sc + o_list, ident.data(), 4 * n_m, hipMemcpyHostToDevice, s));
    OcrSrc src{};
    src.recs = reinterpret_cast<const CandRec *>(sc + o_rec); src.list = reinterpret_cast<const uint32_t *>(sc + o_list); src.planes = d_planes;
    src.rot = reinterpret_cast<const RotGeom *>(sc + o_rot);
    rec(c, "line_ocr_host_gap");
    launch_ocr_features(s, src, (int)n_m, buf, &m);
    rec(c, "line_ocr_features");
    launch_svm_kernel(s, (int)n_m, buf, m, true);
    rec(c, "line_svm_kernel");
    launch_svm_couple(s, (int)n_m, buf, m);
    rec(c, "line_svm_couple");
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, wait_stream(c, s));          // (wait, then copy into pageable memory: see run_batch's OCR stage)
    HIP_TRY(c, hipMemcpyAsync(r->line_label.data(), buf.label, 4 * n_m, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(r->line_prob.data(), buf.pbest, 8 * n_m, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, wait_stream(c, s));
    for (size_t t = r->texts.size(); t-- > 0;) {
        const str_er_text &tx = r->texts[t];
        const size_t       f = (size_t)tx.first, n = (size_t)tx.count;
        std::vector<char>  del(n, 0);
        for (size_t a = 0; a < n; ++a)
            for (size_t bq = a + 1; bq < n; ++bq) {
                const str_er_gbound &A = r->gbounds[(size_t)r->text_ers[f + a]], &B = r->gbounds[(size_t)r->text_ers[f + bq]];
                const int ix = std::max(A.x, B.x), iy = std::max(A.y, B.y);
                int       iw = std::min(A.x + A.w, B.x + B.w) - ix, ih = std::min(A.y + A.h, B.y + B.h) - iy;
                if (iw <= 0 || ih <= 0) iw = ih = 0;
                const int    ux = std::min(A.x, B.x), uy = std::min(A.y, B.y);
                const double overlap_area = (double)(iw * ih);
                const double union_area = (double)((std::max(A.x + A.w, B.x + B.w) - ux) * (std::max(A.y + A.h, B.y + B.h) - uy));
                if (overlap_area / union_area > 0.95) {
                    if (A.w * A.h > B.w * B.h) del[bq] = 1;
                    else del[a] = 1;
                }
            }
        size_t left = 0;
        for (size_t a = 0; a < n; ++a) {
            const bool keep = !del[a] && !(r->line_prob[f + a] < c->min_ocr_prob);
            r->line_kept[f + a] = keep ? 1 : 0;
            left += keep ? 1 : 0;
        }
        r->text_alive[t] = left >= 2 ? 1 : 0;      // min_pass_ocr (:698)
    }
    return STR_ER_OK;
}

// er_grouping for the images of a call.  img[g] = candidate range [lo, hi) of image g (on the device: d_cands / d_track
// hold the same records the host has in r->cands / r->tracks).  GPU: sort ranks, inner_suppression flags, pair list;
// host: the greedy line assignment and the per-line steps (er_group.cpp).
int group_phase(str_er_ctx *c, const CandRec *d_cands, const TrackRec *d_track, const std::vector<uint32_t> &img, bool inner_sup,
                str_er_result *r, bool presorted)
{
    const int    G = (int)(img.size() / 2);
    const size_t n_c = r->cands.size();
    r->have_texts = true;
    r->gbounds.resize(n_c);
    for (size_t i = 0; i < n_c; ++i) {
        const str_er_cand &cd = r->cands[i];
        str_er_gbound &gb = r->gbounds[i];
        gb.x = cd.x; gb.y = cd.y; gb.w = cd.w; gb.h = cd.h; gb.cx = r->tracks[i].cx; gb.cy = r->tracks[i].cy;
    }
    if (G == 0 || n_c == 0) return STR_ER_OK;
    for (int g = 0; g < G; ++g)
        if (img[2 * g + 1] - img[2 * g] > 65535u) return fail(c, STR_ER_ECAPACITY, "more than 65535 candidates in one image");
    hipStream_t s = c->stream;
    const size_t words = 4 * n_c + 4 * (size_t)G + 64;
    if (words > c->group_words) {          // (a quarter more than needed: the candidate count differs from batch to batch, and hipFree / hipMalloc wait for the whole device)
        const size_t get = words + words / 4;
        if (c->d_group) { (void)hipFree(c->d_group); c->d_group = nullptr; c->group_words = 0; }
        if (hipMalloc(reinterpret_cast<void **>(&c->d_group), 4 * get) != hipSuccess) return fail(c, STR_ER_ENOMEM, "hipMalloc (grouping workspace)");
        c->group_words = get;
    }
    auto grow_pairs = [&](size_t cap) -> int {
        if (cap <= c->group_pair_cap) return STR_ER_OK;
        cap += cap / 4;
        if (c->d_group_pairs) { (void)hipFree(c->d_group_pairs); c->d_group_pairs = nullptr; c->group_pair_cap = 0; }
        if (hipMalloc(reinterpret_cast<void **>(&c->d_group_pairs), 4 * cap) != hipSuccess) return fail(c, STR_ER_ENOMEM, "hipMalloc (pair list)");
        c->group_pair_cap = cap;
        return STR_ER_OK;
    };
    int rc = grow_pairs(std::max<size_t>(8 * n_c, 4096));
    if (rc != STR_ER_OK) return rc;
    GroupBufs gb{};
    gb.tmp_a = c->d_group; gb.tmp_b = gb.tmp_a + n_c; gb.sorted = gb.tmp_b + n_c; gb.row_cnt = gb.sorted + n_c;
    gb.n_sorted = gb.row_cnt + n_c; gb.pair_off = gb.n_sorted + G;
    uint32_t *d_rng = gb.pair_off + G + 1;
    gb.pairs = c->d_group_pairs; gb.pair_cap = (uint32_t)std::min<size_t>(c->group_pair_cap, 0xFFFFFFFFu);
    HIP_TRY(c, hipMemcpyAsync(d_rng, img.data(), 4 * img.size(), hipMemcpyHostToDevice, s));
    launch_group_prepare(s, d_cands, d_track, d_rng, G, (inner_sup ? 1 : 0) | (presorted ? 2 : 0), gb);
    launch_group_pairs_count(s, d_cands, d_track, d_rng, G, gb);
    launch_group_pairs_fill(s, d_cands, d_track, d_rng, G, gb);
    HIP_TRY(c, hipGetLastError());
    std::vector<uint32_t> n_sorted((size_t)G), pair_off((size_t)G + 1), sorted(n_c);
    HIP_TRY(c, wait_stream(c, s));          // (wait, then copy into pageable memory: see run_batch's OCR stage)
    HIP_TRY(c, hipMemcpyAsync(n_sorted.data(), gb.n_sorted, 4 * (size_t)G, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(pair_off.data(), gb.pair_off, 4 * ((size_t)G + 1), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(sorted.data(), gb.sorted, 4 * n_c, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, wait_stream(c, s));
    const size_t n_pairs = pair_off[(size_t)G];
    if (n_pairs > c->group_pair_cap) {               // the optimistic buffer was too small: the fill kernel did nothing
        rc = grow_pairs(n_pairs + n_pairs / 4);
        if (rc != STR_ER_OK) return rc;
        gb.pairs = c->d_group_pairs; gb.pair_cap = (uint32_t)std::min<size_t>(c->group_pair_cap, 0xFFFFFFFFu);
        launch_group_pairs_fill(s, d_cands, d_track, d_rng, G, gb);
        HIP_TRY(c, hipGetLastError());
    }
    std::vector<uint32_t> pairs(n_pairs);
    if (n_pairs) HIP_TRY(c, hipMemcpyAsync(pairs.data(), gb.pairs, 4 * n_pairs, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, wait_stream(c, s));

    std::vector<GroupEr>  ers;
    std::vector<TextLine> lines;
    for (int g = 0; g < G; ++g) {
        const uint32_t lo = img[2 * g], m = n_sorted[(size_t)g];
        if (m == 0) continue;
        ers.resize(m);
        for (uint32_t k = 0; k < m; ++k) {
            const str_er_gbound &b = r->gbounds[sorted[lo + k]];
            ers[k] = GroupEr{b.x, b.y, b.w, b.h, b.cx, b.cy};
        }
        for (uint32_t k = 0; k < m; ++k) r->group_all.push_back((int32_t)sorted[lo + k]);
        group_lines(ers, pairs.data() + pair_off[(size_t)g], pair_off[(size_t)g + 1] - pair_off[(size_t)g], lines);
        for (uint32_t k = 0; k < m; ++k) {
            str_er_gbound &b = r->gbounds[sorted[lo + k]];
            b.x = ers[k].x; b.y = ers[k].y; b.w = ers[k].w; b.h = ers[k].h; b.cx = ers[k].cx; b.cy = ers[k].cy;
        }
        const str_er_cand &first = r->cands[lo];
        for (const TextLine &t : lines) {
            str_er_text tx{};
            tx.frame = first.frame; tx.pyr = first.pyr;
            tx.first = (int32_t)r->text_ers.size(); tx.count = (int32_t)t.ers.size();
            tx.slope = t.slope; tx.x = t.box[0]; tx.y = t.box[1]; tx.w = t.box[2]; tx.h = t.box[3];
            for (int32_t k : t.ers) r->text_ers.push_back((int32_t)sorted[lo + (uint32_t)k]);
            r->texts.push_back(tx);
        }
    }
    return STR_ER_OK;
}

// Exact NMS where the reference's answer depends on its flood's sibling order (DetectParams::sibling_order == 0): for every plane
// whose first NMS pass met a tie, replay the reference's flood on the GPU (k_flood_order) and repeat the plane's NMS with the ties
// decided by the replayed order.  h_ctr holds the counters of the first pass.  Planes go in rounds that fit the scratch buffer.
int resolve_sibling_ties(str_er_ctx *c, const Batch &b, const BatchDev &bd, const DetectParams &dp, bool &replayed, bool from_tree)
{
    // (from_tree: str_er_nms_tree_plane -- no batch ran, nothing was exported)
    replayed = false;
    std::vector<int> amb;
    for (int i = 0; i < (int)b.planes.size(); ++i)
        if (c->h_ctr[i].n_rel) amb.push_back(i);         // ties that can change the pool (k_nms); the others need no decision
    if (amb.empty()) return STR_ER_OK;
    // (the context's own stream is idle here -- the caller has just synchronised it -- and other contexts keep the GPU's queues
    // full: on the high-priority stream these few small operations do not wait behind their tile kernels)
    hipStream_t s = c->prio ? c->prio : c->stream;
    const auto plane_px = [&](int i) { return (size_t)b.planes[i].w * b.planes[i].h; };
    const auto dense = [&](int i) { return c->h_ctr[i].n_watch > (uint32_t)NMS_WATCH_CAP; };    // watch list overflowed: a stamp per pixel

    // ---- device scratch: the GPU walk needs 9 bytes per pixel, the host walk only room for dense stamp arrays -------------
    auto scratch_need = [&](int i) -> size_t {
        if (c->replay_on_gpu) return replay_scratch_bytes(b.planes[i].w, b.planes[i].h);
        return dense(i) ? ((plane_px(i) * 4 + 255) / 256) * 256 : 0;
    };
    size_t largest = 0, total = 0;
    for (int i : amb) { largest = std::max(largest, scratch_need(i)); total += scratch_need(i); }
    const size_t want = std::max(largest, std::min<size_t>(total, (size_t)1 << 30));
    if (want > c->replay_bytes) {
        // (grown in steps of at least 2 x: hipFree / hipMalloc wait for the whole device -- every other context's kernels included; a rocprofv3 timeline of
        // six batches in flight showed one such call as a 64 ms hole in the GPU's work)
        const size_t get = std::max(want, 2 * c->replay_bytes);
        if (c->d_replay) { (void)hipFree(c->d_replay); c->d_replay = nullptr; c->replay_bytes = 0; }
        if (hipMalloc(reinterpret_cast<void **>(&c->d_replay), get) != hipSuccess) return fail(c, STR_ER_ENOMEM, "hipMalloc (flood replay scratch)");
        c->replay_bytes = get;
    }

    // Planes go in rounds that fit the scratch (one round unless dozens of planes have dense stamps).  A round of the host walk is
    // two device round trips: planes + watch lists down, [walk], stamps up + the tie pass of the NMS; everything the device reads or
    // writes on the host side lives in one page-locked arena (copies to pageable memory go through bounce buffers under a lock).
    size_t at = 0;
    while (at < amb.size()) {
        std::vector<ReplayItem> items;
        size_t pos = 0, hneed = ((sizeof(ReplayItem) * amb.size() + 255) / 256) * 256;
        std::vector<size_t> hoff;
        while (at < amb.size() && (items.empty() || pos + scratch_need(amb[at]) <= c->replay_bytes)) {
            const int i = amb[at++];
            ReplayItem it{};
            it.plane = (uint32_t)i; it.off = pos;
            items.push_back(it);
            pos += scratch_need(i);
            hoff.push_back(hneed);
            // (every term a multiple of 256: the export kernel places a plane at 256 * pad_ -- a dense stamp area of w * h * 4 bytes with
            // w * h % 64 != 0 used to leave the NEXT plane's offset unaligned, and the device then wrote where the walk did not read)
            hneed += ((plane_px(i) + 255) / 256) * 256 + ((3 * 4 * (size_t)NMS_WATCH_CAP + 255) / 256) * 256 +
                     (dense(i) && !c->replay_on_gpu ? ((plane_px(i) * 4 + 255) / 256) * 256 : 0);
        }
        for (size_t o : hoff) if (o % 256 != 0) return fail(c, STR_ER_ESTATE, "tie plane arena: unaligned plane offset");
        const size_t m = items.size();
        if (hneed > c->h_replay_bytes) {
            // (the arena's need follows the number of tie planes of a batch, which differs from batch to batch: room for 8 planes of the context's size at once --
            // at most 64 MB -- then doubling; a page-locked allocation is a 4 ms call that the other contexts' copies queue behind)
            const size_t plane_bytes = (((size_t)c->prm.max_width * c->prm.max_height + 255) / 256) * 256 + ((3 * 4 * (size_t)NMS_WATCH_CAP + 255) / 256) * 256;
            const size_t get = std::max(hneed, std::max(2 * c->h_replay_bytes, std::min<size_t>(8 * plane_bytes, (size_t)64 << 20)));
            if (c->h_replay) { (void)hipHostFree(c->h_replay); c->h_replay = nullptr; c->h_replay_bytes = 0; }
            if (hipHostMalloc(reinterpret_cast<void **>(&c->h_replay), get, hipHostMallocMapped) != hipSuccess)
                return fail(c, STR_ER_ENOMEM, "hipHostMalloc (flood order walk staging)");
            c->h_replay_bytes = get;
        }
        ReplayItem *h_items = reinterpret_cast<ReplayItem *>(c->h_replay);
        // which planes the device has already put into host memory (k_export_tie_planes: the first TIE_SLOTS tie planes of the batch)
        std::vector<int> slot_of(m, -1);
        if (!c->replay_on_gpu && c->h_tie && !from_tree)
            for (size_t k = 0; k < m; ++k)
                for (uint32_t q = 0; q < std::min<uint32_t>(*c->h_tie_count, (uint32_t)c->n_tie_slots); ++q) if (c->h_tie_plane[q] == items[k].plane) slot_of[k] = (int)q;
        // (pad_: where the export kernel puts a plane that has no slot, in units of 256 bytes from the arena's start)
        for (size_t k = 0; k < m; ++k) items[k].pad_ = (c->replay_on_gpu || slot_of[k] >= 0) ? 0xFFFFFFFFu : (uint32_t)(hoff[k] / 256);
        std::memcpy(h_items, items.data(), sizeof(ReplayItem) * m);
        HIP_TRY(c, hipMemcpyAsync(c->d_replay_items, h_items, sizeof(ReplayItem) * m, hipMemcpyHostToDevice, s));
        if (c->replay_on_gpu) {
            launch_flood_order(s, bd, dp, c->d_replay_items, (int)m, c->d_replay);
        } else {
            struct HostPlane { uint8_t *pix; uint32_t *watch, *group, *stamp; uint32_t n_watch; };
            std::vector<HostPlane> hp(m);
            bool need_sync = false;
            for (size_t k = 0; k < m; ++k) {
                const int        i = (int)items[k].plane;
                HostPlane       &h = hp[k];
                h.pix = c->h_replay + hoff[k];
                h.watch = reinterpret_cast<uint32_t *>(h.pix + ((plane_px(i) + 255) / 256) * 256);
                h.group = h.watch + NMS_WATCH_CAP;
                h.stamp = h.group + NMS_WATCH_CAP;          // NMS_WATCH_CAP entries, or w * h when dense
                h.n_watch = c->h_ctr[i].n_watch;
                if (slot_of[k] >= 0) {
                    h.pix = c->h_tie + (size_t)slot_of[k] * c->tie_slot_bytes;
                    uint32_t *wl = reinterpret_cast<uint32_t *>(h.pix + ((plane_px(i) + 255) / 256) * 256);
                    if (!dense(i)) { std::memcpy(h.watch, wl, 4 * (size_t)h.n_watch); std::memcpy(h.group, wl + NMS_WATCH_CAP, 4 * (size_t)h.n_watch); }
                    continue;
                }
                need_sync = true;
            }
            // the planes without a slot: one launch writes them (and their watch lists) into the arena
            if (need_sync) { launch_export_listed_planes(s, bd, c->d_replay_items, (int)m, c->h_replay); HIP_TRY(c, hipGetLastError()); }
            if (need_sync) HIP_TRY(c, wait_stream(c, s));
            const auto tw0 = std::chrono::steady_clock::now();
            auto walk = [&](size_t k) {
                const int        i = (int)items[k].plane;
                const PlaneDesc &pd = b.planes[i];
                HostPlane       &h = hp[k];
                if (!dense(i)) {
                    std::memset(h.stamp, 0, 4 * (size_t)NMS_WATCH_CAP);
                    flood_order_host(h.pix, pd.w, pd.h, pd.w, pd.invert, dp.qscale, dp.hi, h.watch, h.n_watch, h.stamp, h.group);
                } else {
                    std::memset(h.stamp, 0, 4 * plane_px(i));
                    flood_order_host(h.pix, pd.w, pd.h, pd.w, pd.invert, dp.qscale, dp.hi, nullptr, 0xFFFFFFFFu, h.stamp);
                }
            };
            // (a bounded, process-wide pool: however many contexts have tie planes at the moment, at most flood_walk_threads() host
            // threads walk; an allocation failure inside a walk comes back as an error code, nothing is thrown across the C ABI)
            std::vector<double> walk_ms(m, 0.0);
            struct WalkArg { decltype(walk) *fn; std::vector<double> *ms; } wa{&walk, &walk_ms};
            const int wrc = flood_walks_run(m, [](size_t k, void *a) {
                WalkArg *w = static_cast<WalkArg *>(a);
                const auto t0 = std::chrono::steady_clock::now();
                (*w->fn)(k);
                (*w->ms)[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }, &wa);
            if (wrc != 0) return fail(c, wrc == -1 ? STR_ER_ENOMEM : STR_ER_EHIP, "flood order walk failed on the host (out of memory?)");
            for (double v : walk_ms) c->walk_ms_total += v;
            if (c->dbg_stats)
                std::fprintf(stderr, "[str_er] flood order walk: %zu plane(s), %.1f ms on the host\n", m,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count());
            for (size_t k = 0; k < m; ++k) {
                const int i = (int)items[k].plane;
                if (!dense(i)) {
                    if (hp[k].n_watch) HIP_TRY(c, hipMemcpyAsync(c->d_wstamp + (size_t)i * NMS_WATCH_CAP, hp[k].stamp, 4 * (size_t)hp[k].n_watch, hipMemcpyHostToDevice, s));
                } else {
                    HIP_TRY(c, hipMemcpyAsync(c->d_replay + items[k].off, hp[k].stamp, 4 * plane_px(i), hipMemcpyHostToDevice, s));
                }
            }
        }
        launch_nms_resolve(s, bd, dp, c->d_replay_items, (int)m, c->d_replay);
        HIP_TRY(c, hipGetLastError());
        c->n_replayed += m;
        // (the arena and the item table are reused by the next round; the last round is left to the caller's synchronisation)
        if (at < amb.size()) HIP_TRY(c, wait_stream(c, s));
    }
    replayed = true;
    return STR_ER_OK;
}

// er_grouping(all_er, text, overlap_sup = true, inner_sup) (src/ER.cpp:612-692; video_mode calls it so, src/utils.cpp:196): sort and
// overlap_suppression are sequential (a merge rewrites the survivor's box, which the next comparison reads), so they run on the host over the
// records the result already holds; the survivors -- with their rewritten boxes, in list order -- then go through the same GPU stages as
// before (inner_suppression flags, pair rule) and the host's greedy line assignment.  Indices in the result refer to r's candidates.
int group_phase_overlap(str_er_ctx *c, const std::vector<uint32_t> &img, bool inner_sup, str_er_result *r)
{
    const size_t n_c = r->cands.size();
    str_er_result tmp;
    std::vector<int32_t> orig;                          // candidate index in r of every record of tmp
    std::vector<uint32_t> img2;
    for (size_t g = 0; g + 1 < img.size(); g += 2) {
        std::vector<GroupEr> ers;
        std::vector<int32_t> order, src;
        for (uint32_t i = img[g]; i < img[g + 1]; ++i)
            if (r->tracks[i].tracked) {
                const str_er_cand &cd = r->cands[i];
                ers.push_back(GroupEr{cd.x, cd.y, cd.w, cd.h, r->tracks[i].cx, r->tracks[i].cy});
                order.push_back((int32_t)src.size());
                src.push_back((int32_t)i);
            }
        sort_and_overlap_suppress(ers, order);
        img2.push_back((uint32_t)tmp.cands.size());
        for (int32_t k : order) {
            str_er_cand  cd = r->cands[(size_t)src[(size_t)k]];
            str_er_track tk = r->tracks[(size_t)src[(size_t)k]];
            const GroupEr &e = ers[(size_t)k];
            if (e.x < 0 || e.y < 0 || e.w < 0 || e.h < 0 || e.x > 65535 || e.y > 65535 || e.w > 65535 || e.h > 65535) return fail(c, STR_ER_EINVAL, "box out of range");
            cd.x = (uint16_t)e.x; cd.y = (uint16_t)e.y; cd.w = (uint16_t)e.w; cd.h = (uint16_t)e.h;
            tk.cx = e.cx; tk.cy = e.cy; tk.tracked = 1;
            tmp.cands.push_back(cd); tmp.tracks.push_back(tk);
            orig.push_back(src[(size_t)k]);
        }
        img2.push_back((uint32_t)tmp.cands.size());
    }
    r->have_texts = true;
    r->gbounds.resize(n_c);
    for (size_t i = 0; i < n_c; ++i) {
        const str_er_cand &cd = r->cands[i];
        r->gbounds[i] = str_er_gbound{cd.x, cd.y, cd.w, cd.h, r->tracks[i].cx, r->tracks[i].cy};
    }
    const size_t m = tmp.cands.size();
    if (m == 0) return STR_ER_OK;
    const size_t o_tr = align_up(sizeof(CandRec) * m, 256);
    int rc = ensure_scratch(c, o_tr + sizeof(TrackRec) * m);
    if (rc != STR_ER_OK) return rc;
    uint8_t *sc = static_cast<uint8_t *>(c->d_scratch);
    HIP_TRY(c, hipMemcpyAsync(sc, tmp.cands.data(), sizeof(CandRec) * m, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(sc + o_tr, tmp.tracks.data(), sizeof(TrackRec) * m, hipMemcpyHostToDevice, c->stream));
    rc = group_phase(c, reinterpret_cast<const CandRec *>(sc), reinterpret_cast<const TrackRec *>(sc + o_tr), img2, inner_sup, &tmp, /*presorted=*/true);
    if (rc != STR_ER_OK) return rc;
    // back to r's numbering
    r->texts = tmp.texts;
    r->text_ers.clear();
    for (int32_t k : tmp.text_ers) r->text_ers.push_back(orig[(size_t)k]);
    r->group_all.clear();
    for (int32_t k : tmp.group_all) r->group_all.push_back(orig[(size_t)k]);
    for (size_t k = 0; k < m; ++k) r->gbounds[(size_t)orig[k]] = tmp.gbounds[k];
    return STR_ER_OK;
}

// Plane descriptors, zeroed counters and the tile / seam-block lookup tables of a laid-out batch go to the device.
int upload_layout(str_er_ctx *c, Batch &b)
{
    hipStream_t s = c->stream;
    const int np = (int)b.planes.size();
    {   // workgroups of the per-record kernels: by plane size, the largest plane node_blocks of them
        size_t most_tiles = 1;
        for (const PlaneDesc &pd : b.planes) most_tiles = std::max(most_tiles, (size_t)pd.tiles_x * pd.tiles_y);
        uint32_t nb = std::min<uint32_t>(std::max<uint32_t>(c->node_blocks, 1u), 255u);
        for (;;) {
            uint32_t at = 0;
            for (PlaneDesc &pd : b.planes) {
                const size_t t = (size_t)pd.tiles_x * pd.tiles_y;
                pd.nb_count = (uint8_t)std::max<size_t>(1, (t * nb + most_tiles - 1) / most_tiles);
                pd.nb_base = at;
                at += pd.nb_count;
            }
            c->n_node_blocks = at;
            if (at <= (size_t)c->max_planes * str_er_ctx::NB_PLANE_SHARE || nb == 1) break;        // (the workgroup -> plane table holds NB_PLANE_SHARE entries a plane)
            nb = std::max<uint32_t>(1, nb / 2);
        }
    }
    // (the descriptors of a video's frames are the same call after call -- planes in the context's own pool, same shares of the tables: uploaded when they change)
    if (c->planes_on_device != np || std::memcmp(c->h_planes, b.planes.data(), sizeof(PlaneDesc) * np) != 0) {
        c->planes_on_device = 0;
        std::memcpy(c->h_planes, b.planes.data(), sizeof(PlaneDesc) * np);
        HIP_TRY(c, hipMemcpyAsync(c->d_planes, c->h_planes, sizeof(PlaneDesc) * np, hipMemcpyHostToDevice, s));
        c->planes_on_device = np;
    }
    // counters of the batch (candidates, tiles k_tile_tree2 handed back), plane counters, the groups' done flags
    // (a whole number of 256-byte pieces: one fill kernel, not two.  The last call of the context usually left the block zeroed -- a fill in front of the tile
    // kernels is ~20 us of a 1-frame call, with the queue switches around it)
    {
        const size_t need = align_up(c->zero_gd_off + b.n_groups, 256);
        if (c->zero_clean_bytes < need) HIP_TRY(c, hipMemsetAsync(c->d_zero, 0, need, s));
        c->zero_clean_bytes = 0;
    }
    {   // tile -> plane and seam-block -> (plane, first pair) tables; re-uploaded only when the layout changes
        std::vector<uint32_t> key;
        key.reserve(np * 2 + 1);
        key.push_back((uint32_t)np);
        key.push_back(c->node_blocks);
        key.push_back((uint32_t)b.group_x * 256u + (uint32_t)b.group_y);
        for (const PlaneDesc &pd : b.planes) { key.push_back((uint32_t)pd.w); key.push_back((uint32_t)pd.h); }
        if (key != c->layout_key) {
            c->layout_key.clear();          // the tables are being rebuilt: a failure below must not leave the old key naming them
            c->h_tile_plane.clear(); c->h_sb_plane.clear(); c->h_sb_first.clear(); c->h_nb_plane.clear();
            for (int i = 0; i < np; ++i) c->h_nb_plane.insert(c->h_nb_plane.end(), (size_t)b.planes[i].nb_count, (uint16_t)i);
            HIP_TRY(c, hipMemcpyAsync(c->d_nb_plane, c->h_nb_plane.data(), 2 * c->h_nb_plane.size(), hipMemcpyHostToDevice, s));
            c->h_group_plane.clear();
            if (b.group_x > 0 && b.group_y > 0)
                for (int i = 0; i < np; ++i) {
                    const PlaneDesc &pd = b.planes[i];
                    c->h_group_plane.insert(c->h_group_plane.end(), (size_t)((pd.tiles_x + b.group_x - 1) / b.group_x) * ((pd.tiles_y + b.group_y - 1) / b.group_y), (uint16_t)i);
                }
            if (!c->h_group_plane.empty()) HIP_TRY(c, hipMemcpyAsync(c->d_group_plane, c->h_group_plane.data(), 2 * c->h_group_plane.size(), hipMemcpyHostToDevice, s));
            // ... and the groups by class of plane: the chroma planes' first (run_batch launches k_group_merge once per class)
            c->h_group_list.clear(); c->n_groups_small = 0;
            if (b.group_x > 0 && b.group_y > 0) {
                for (int pass = 0; pass < 2; ++pass) {
                    for (int i = 0; i < np; ++i) {
                        const PlaneDesc &pd = b.planes[i];
                        if ((pd.ch % 3 != 0) != (pass == 0)) continue;
                        const uint32_t ng = (uint32_t)((pd.tiles_x + b.group_x - 1) / b.group_x) * (uint32_t)((pd.tiles_y + b.group_y - 1) / b.group_y);
                        for (uint32_t g = 0; g < ng; ++g) c->h_group_list.push_back(pd.group_base + g);
                    }
                    if (pass == 0) c->n_groups_small = (uint32_t)c->h_group_list.size();
                }
                if (!c->h_group_list.empty()) HIP_TRY(c, hipMemcpyAsync(c->d_group_list, c->h_group_list.data(), 4 * c->h_group_list.size(), hipMemcpyHostToDevice, s));
            }
            for (int i = 0; i < np; ++i) {
                const PlaneDesc &pd = b.planes[i];
                c->h_tile_plane.insert(c->h_tile_plane.end(), (size_t)pd.tiles_x * pd.tiles_y, (uint16_t)i);
                // k_seam's workgroups: 1024 consecutive pixel pairs each.  With groups of tiles only the seams BETWEEN groups are k_seam's (the inner ones:
                // k_group_merge, or k_seam_undone for a group it left alone) -- three quarters of all pairs get no workgroup at all.  A seam's last workgroup
                // reaches into the pairs behind it: inner ones (skipped by the kernel) or outer ones seen twice (a connect is idempotent).
                auto blocks = [&](uint32_t lo, uint32_t hi) {
                    for (uint32_t f0 = lo; f0 < hi; f0 += (uint32_t)SEAM_BLOCK) { c->h_sb_plane.push_back((uint16_t)i); c->h_sb_first.push_back(f0); }
                };
                if (b.group_x > 0 && b.group_y > 0) {
                    for (int j = b.group_y - 1; j + 1 < pd.tiles_y; j += b.group_y) blocks((uint32_t)j * (uint32_t)pd.w, (uint32_t)(j + 1) * (uint32_t)pd.w);
                    for (int k = b.group_x - 1; k + 1 < pd.tiles_x; k += b.group_x) blocks(pd.n_hpairs + (uint32_t)k * (uint32_t)pd.h, pd.n_hpairs + (uint32_t)(k + 1) * (uint32_t)pd.h);
                } else blocks(0, pd.n_pairs);
            }
            if (c->h_sb_plane.size() > c->sb_slots) return fail(c, STR_ER_ECAPACITY, "seam block table capacity exceeded");
            HIP_TRY(c, hipMemcpyAsync(c->d_tile_plane, c->h_tile_plane.data(), 2 * c->h_tile_plane.size(), hipMemcpyHostToDevice, s));
            if (!c->h_sb_plane.empty()) {
                HIP_TRY(c, hipMemcpyAsync(c->d_sb_plane, c->h_sb_plane.data(), 2 * c->h_sb_plane.size(), hipMemcpyHostToDevice, s));
                HIP_TRY(c, hipMemcpyAsync(c->d_sb_first, c->h_sb_first.data(), 4 * c->h_sb_first.size(), hipMemcpyHostToDevice, s));
            }
            HIP_TRY(c, wait_stream(c, s));   // pageable host vectors: make sure the copies are done
            c->layout_key = key;
        }
    }
    return STR_ER_OK;
}

// Enqueue extract -> NMS -> classify for a laid-out batch and build the result.
// import_trees (optional): the tile trees were built elsewhere (strips of a plane extracted by other GPUs) -- instead of running
// k_tile_tree / k_seam the hook puts node records and counters in place on the context's stream.
// STR_ER_STAGE_OCR: chain_run on the strong / weak ERs of the batch (src/ER.cpp:728-735 calls it per ER).  `cap` = the number of ERs the launches and the
// scratch are sized for; the kernels work on min(cap, the device's own count) -- so the stage can be enqueued right behind classify with a guessed cap, or
// after the host has read the counters with the exact number.  The results travel to the context's page-locked block: count | list | labels | probabilities.
// (two regions: the scores of the batch, and -- while those wait to be read -- the scores of the planes an NMS tie pass re-made)
struct OcrPinned { uint32_t *count; uint32_t *list; int32_t *label; double *prob; };
static OcrPinned ocr_pinned(const str_er_ctx *c, size_t cap, int region = 0)
{
    OcrPinned p;
    uint8_t  *base = c->h_ocr + (size_t)region * (64 + 16 * c->h_ocr_cap);
    p.count = reinterpret_cast<uint32_t *>(base);
    p.prob = reinterpret_cast<double *>(base + 64);
    p.list = reinterpret_cast<uint32_t *>(base + 64 + 8 * cap);
    p.label = reinterpret_cast<int32_t *>(base + 64 + 12 * cap);
    return p;
}
static int ocr_stage(str_er_ctx *c, const BatchDev &bd, size_t cap, hipStream_t s, bool again, const uint32_t *from = nullptr, const uint32_t *from_n = nullptr, int region = 0)
{
    static const char *const names[2][4] = {{"ocr_host_gap", "ocr_features", "svm_kernel", "svm_couple"}, {"ocr_again_host_gap", "ocr_again_features", "ocr_again_svm_kernel", "ocr_again_svm_couple"}};
    const char *const *nm = names[again ? 1 : 0];
    if (cap > c->h_ocr_cap) {
        const size_t want = cap + cap / 4;
        if (c->h_ocr) { (void)hipHostFree(c->h_ocr); c->h_ocr = nullptr; c->h_ocr_cap = 0; }
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_ocr), 2 * (64 + 16 * want)));
        c->h_ocr_cap = want;
    }
    const SvmDev &m = c->svm;
    const size_t  n_cands = c->pool_total;        // (the list has room for every candidate the tables hold)
    const size_t  o_list = 0, o_buf = align_up(4 * (n_cands + OCR_LIST_HDR), 256);
    const int     rc = ensure_scratch(c, o_buf + ocr_layout(nullptr, cap, &m, false, false, false).bytes);
    if (rc != STR_ER_OK) return rc;
    uint8_t  *sc = static_cast<uint8_t *>(c->d_scratch);
    OcrBuf    buf = ocr_layout(sc + o_buf, cap, &m, false, false, false);
    uint32_t *d_list = reinterpret_cast<uint32_t *>(sc + o_list);
    buf.n_dev = d_list;                          // (hdr[0]: k_ocr_list's count)
    rec(c, nm[0]);          // (what lies between classify and the scorer: a round trip to the host when the counters were read first)
    if (from) launch_ocr_list_from(s, bd, from, from_n, d_list);       // (the candidates of the planes a tie pass re-made)
    else launch_ocr_list(s, bd, (uint32_t)n_cands, d_list);
    OcrSrc src{};
    src.recs = bd.cands; src.list = d_list + OCR_LIST_HDR; src.planes = bd.planes; src.n_dev = d_list;
    launch_ocr_features(s, src, (int)cap, buf, &m);
    rec(c, nm[1]);
    launch_svm_kernel(s, (int)cap, buf, m, true);
    rec(c, nm[2]);
    launch_svm_couple(s, (int)cap, buf, m);
    rec(c, nm[3]);
    HIP_TRY(c, hipGetLastError());
    const OcrPinned hp = ocr_pinned(c, cap, region);
    HIP_TRY(c, hipMemcpyAsync(hp.count, d_list, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(hp.list, d_list + OCR_LIST_HDR, 4 * cap, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(hp.label, buf.label, 4 * cap, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(hp.prob, buf.pbest, 8 * cap, hipMemcpyDeviceToHost, s));
    return STR_ER_OK;
}

// str_er_set_batch_slots (include/str_er.h): at most g_slot_cap large batches have their kernels on the GPU at a time
static std::mutex              g_slot_mu;
static std::condition_variable g_slot_cv;
static int                     g_slot_cap = 0, g_slot_used = 0;
struct BatchSlot {
    bool held = false;
    void take()
    {
        std::unique_lock<std::mutex> lk(g_slot_mu);
        if (g_slot_cap <= 0) return;
        g_slot_cv.wait(lk, [] { return g_slot_cap <= 0 || g_slot_used < g_slot_cap; });
        if (g_slot_cap > 0) { ++g_slot_used; held = true; }
    }
    void give()
    {
        if (!held) return;
        { std::lock_guard<std::mutex> lk(g_slot_mu); --g_slot_used; }
        held = false;
        g_slot_cv.notify_one();
    }
    ~BatchSlot() { give(); }
};

int run_batch(str_er_ctx *c, const Batch &b_in, uint32_t stages, str_er_result **out,
              std::chrono::steady_clock::time_point t_start, bool pre_recorded, const ImportHook *import_trees, int attempt)
{
    // A batch whose planes outgrow their shares of the tables is laid out again with larger shares and repeated.  Every repeat raises a
    // share (or fails), and a share stops at one entry per pixel: the repeats end; `attempt` only guards against a slip in that argument.
    if (attempt > 24) return fail(c, STR_ER_ECAPACITY, "the batch was repeated 24 times with growing tables and still does not fit (internal error)");
    c->last_valid = false;             // (str_er_gather_last: the candidate array is being rewritten, or re-allocated)
    struct SpinGuard { str_er_ctx *c; int old; ~SpinGuard() { c->wait_spin_us = old; } } spin_guard{c, c->wait_spin_us};
    c->wait_spin_us = (int)b_in.planes.size() <= SPEC_PLANES ? 2000 : 300;      // (for this call only: the guard puts the default back)
    Batch b = b_in;
    BatchSlot slot;                    // (str_er_set_batch_slots: given back when the batch's kernels are done, below, or on any way out)
    if ((int)b_in.planes.size() > SPEC_PLANES) slot.take();
    // tiles are joined in two steps: groups of tiles in LDS (k_group_merge), then the groups through the global passes.  Text-like
    // batches (small tile kernel: ~14 records per tile) take 8 x 4 tiles per group (a call of a frame or two: 4 x 4) with room for 2048 records,
    // noise-like ones (~250 records per tile) 2 x 5.
    // (Round 6, tools/dev_groups.sh: ALONE on the GPU the 4 x 4 / 1024-record kernel is the faster one -- 0.41 against 0.48 ms per 48-frame batch, 50 against 81 us
    // per frame: 64 KB of LDS leave two workgroups a compute unit -- but with six batches in flight the larger groups win: 9.9-10.5 k -> 10.8-11.2 k frames/s on
    // pyr3x8, +0.5 % on native6, +1 % at 4K.  The other batches' kernels fill what the group kernel leaves idle, and what it saves -- groups of the luma pyramid that
    // no longer overflow the table and take the global passes whole, a third fewer records and pairs on the groups' outer borders -- are device-scope atomics, which
    // the batches in flight share.)
    const bool grouped = !import_trees && c->group_mode != 0;
    const bool big_groups = c->tile_sparse && (int)b_in.planes.size() > SPEC_PLANES;
    {
        int gx = c->tile_sparse ? (big_groups ? 8 : 4) : 2, gy = c->tile_sparse ? 4 : 5;
        if (c->dbg_group[0] > 0) { gx = c->dbg_group[0]; gy = c->dbg_group[1]; }
        if (grouped) assign_groups(b, gx, gy); else assign_groups(b, 0, 0);
    }
    assign_node_records(b, c->node_share);
    assign_tables(b, c);
    const int ev_entry = pre_recorded ? c->n_ev : -1;
    const int np = (int)b.planes.size();
    if (np == 0) return fail(c, STR_ER_EINVAL, "no planes");
    if (np > c->max_planes) return fail(c, STR_ER_ECAPACITY, "more planes than the context was created for");
    if (b.slots > c->slots) return fail(c, STR_ER_ECAPACITY, "planes exceed the pixel capacity of the context");
    if (b.seam > c->seam_slots) return fail(c, STR_ER_ECAPACITY, "seam map capacity exceeded");
    if (b.kept > c->kept_total || b.pool > c->pool_total) {     // (many tiny planes: the per-plane floors add up; or the shares have grown)
        const int rct = alloc_tables(c, std::max(c->kept_total, b.kept + b.kept / 8), std::max(c->pool_total, b.pool + b.pool / 8));
        if (rct != STR_ER_OK) return rct;
    }
    if (b.n_tiles > c->tile_slots) return fail(c, STR_ER_ECAPACITY, "tile table capacity exceeded");
    if (b.nodes > c->node_slots) {       // (a layout with many tiny planes: the per-plane floor adds up)
        const int rcn = alloc_node_records(c, b.nodes + b.nodes / 8);
        if (rcn != STR_ER_OK) return rcn;
    }
    if ((stages & STR_ER_STAGE_CLASSIFY) && !(c->casc[0].loaded && c->casc[1].loaded))
        return fail(c, STR_ER_ESTATE, "classify needs both cascades (str_er_load_cascade)");
    if (!(stages & STR_ER_STAGE_EXTRACT)) return fail(c, STR_ER_EINVAL, "stages must include STR_ER_STAGE_EXTRACT");
    if ((stages & STR_ER_STAGE_CLASSIFY) && !(stages & STR_ER_STAGE_NMS))
        return fail(c, STR_ER_EINVAL, "STR_ER_STAGE_CLASSIFY needs STR_ER_STAGE_NMS");
    if ((stages & STR_ER_STAGE_OCR) && !(stages & STR_ER_STAGE_CLASSIFY)) return fail(c, STR_ER_EINVAL, "STR_ER_STAGE_OCR needs STR_ER_STAGE_CLASSIFY");
    if ((stages & STR_ER_STAGE_OCR) && !(c->svm_loaded && c->svm.dim == 1800))
        return fail(c, STR_ER_ESTATE, "STR_ER_STAGE_OCR needs an SVM model loaded with dim = 1800 (str_er_load_svm_model)");

    if ((stages & STR_ER_STAGE_TRACK) && !(stages & STR_ER_STAGE_CLASSIFY)) return fail(c, STR_ER_EINVAL, "STR_ER_STAGE_TRACK needs STR_ER_STAGE_CLASSIFY");
    if ((stages & (STR_ER_STAGE_GROUP | STR_ER_GROUP_INNER_SUP | STR_ER_GROUP_OVERLAP_SUP)) && !(stages & STR_ER_STAGE_TRACK)) return fail(c, STR_ER_EINVAL, "STR_ER_STAGE_GROUP needs STR_ER_STAGE_TRACK");
    if ((stages & (STR_ER_GROUP_INNER_SUP | STR_ER_GROUP_OVERLAP_SUP)) && !(stages & STR_ER_STAGE_GROUP)) return fail(c, STR_ER_EINVAL, "STR_ER_GROUP_INNER_SUP / _OVERLAP_SUP modify STR_ER_STAGE_GROUP");
    if ((stages & STR_ER_STAGE_OCR_LINES) && !(stages & STR_ER_STAGE_GROUP)) return fail(c, STR_ER_EINVAL, "STR_ER_STAGE_OCR_LINES needs STR_ER_STAGE_GROUP");
    if ((stages & STR_ER_STAGE_OCR_LINES) && !(c->svm_loaded && c->svm.dim == 1800))
        return fail(c, STR_ER_ESTATE, "STR_ER_STAGE_OCR_LINES needs an SVM model loaded with dim = 1800 (str_er_load_svm_model)");
    if ((stages & STR_ER_STAGE_TRACK) && b.planes_per_image <= 0)
        return fail(c, STR_ER_EINVAL, "STR_ER_STAGE_TRACK needs BGR frames (calc_color reads the YCrCb image)");

    const DetectParams dp = make_dp(c);
    hipStream_t s = c->stream;
    { const int rcu = upload_layout(c, b); if (rcu != STR_ER_OK) return rcu; }
    BatchDev bd = make_batchdev(c, b);
    if (!pre_recorded) { c->n_ev = 0; c->profile.clear(); rec(c, "begin", nullptr, true); }

    // developer aid (tools/dev_exposed.py): STR_ER_DEBUG_STOP_AFTER=n ends the call (with an error) behind stage n -- 0 channels + pyramid, 1 tile trees, 2 group,
    // 3 seam, 4 resolve, 5 accumulate --: what a stage costs with several batches in flight is the difference of two such runs
    static const int dbg_stop = [] { const char *e = std::getenv("STR_ER_DEBUG_STOP_AFTER"); return e ? std::atoi(e) : -1; }();
#define DBG_STOP(n) do { if (dbg_stop == (n)) { (void)wait_stream(c, s); return fail(c, STR_ER_ESTATE, "STR_ER_DEBUG_STOP_AFTER"); } } while (0)
    DBG_STOP(0);
    if (import_trees) {
        const int rci = (*import_trees)(b, bd);
        if (rci != STR_ER_OK) return rci;
        rec(c, "tile_tree");
    } else {
        { const int rct = launch_tile_trees(c, b, bd, dp); if (rct != STR_ER_OK) return rct; }
        rec(c, "tile_tree");
    }
    DBG_STOP(1);
    if (c->dbg_tile_only) {     // developer aid (see STR_ER_STOP_AFTER in er_kernels.hip): time the tile kernel alone
        float ms = 0;
        (void)wait_stream(c, s);
        (void)hipEventElapsedTime(&ms, c->ev[c->n_ev - 2], c->ev[c->n_ev - 1]);
        std::fprintf(stderr, "[str_er] tile_tree alone: %.4f ms\n", ms);
        return fail(c, STR_ER_ESTATE, "STR_ER_DEBUG_TILE_ONLY is set");
    }
    if (grouped && b.n_groups) {
        const int variant = c->dbg_group[2] >= 0 ? c->dbg_group[2] : (c->tile_sparse ? 4 : 6);       // (measured, tools/dev_groups.sh: 2048 records, 1024 lanes on text-like batches)
        // Large text-like batches: a launch per class of planes.  The groups of the chroma planes hold ~150 records: 512 slots and 256 lanes (16 KB of LDS) -- a
        // group that does overflow is k_seam_undone's.  With six batches in flight k_group_merge cost the line its WHOLE isolated time (tools/dev_exposed.py with
        // STR_ER_DEBUG_STOP_AFTER: 0.30 of 0.33 ms per 32-frame batch, where seam / resolve / reduce cost half of theirs): the tile kernels hold all 160 KB of a
        // compute unit's LDS, and a workgroup that wants 64 KB and 16 wave slots waits until four of theirs have left it.  13.0 -> 13.3 k frames/s.
        if (big_groups && c->dbg_group[2] < 0 && c->n_groups_small && c->h_group_list.size() == b.n_groups) {
            launch_group_merge(s, bd, 0, c->d_group_list, c->n_groups_small);
            launch_group_merge(s, bd, variant, c->d_group_list + c->n_groups_small, b.n_groups - c->n_groups_small);
        } else launch_group_merge(s, bd, variant);
    }
    rec(c, "group");
    DBG_STOP(2);
    if (!import_trees) launch_seam(s, bd, !c->tile_sparse);
    rec(c, "seam");
    DBG_STOP(3);
    launch_resolve(s, bd);                            rec(c, "resolve");
    DBG_STOP(4);
    launch_reduce(s, bd);                             rec(c, "accumulate");
    DBG_STOP(5);
    launch_root(s, bd, dp);
    launch_select(s, bd, dp);
    launch_kept(s, bd, dp);                           rec(c, "select", nullptr, true);
    const int i_extract = c->n_ev - 1;
    if (stages & STR_ER_STAGE_NMS) launch_nms(s, bd, dp);
    rec(c, "nms", nullptr, true);
    const int i_nms = c->n_ev - 1;
    const bool alt_pass = (stages & STR_ER_STAGE_NMS) && c->prm.sibling_order == 0;
    if (alt_pass) {       // beside classify: it only decides whether a tie needs the flood order walk
        HIP_TRY(c, hipEventRecord(c->ev_fork, s));
        HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
        launch_nms_alt(c->side, bd, dp, c->d_alt_list);
        if (c->h_tie && !c->replay_on_gpu) {
            *c->h_tie_count = 0;        // (the previous batch of this context is done: nothing on the device touches it any more)
            launch_export_tie_planes(c->side, bd, c->h_tie, c->tie_slot_bytes, c->n_tie_slots, c->d_tie_slot_plane, c->h_tie_count, c->h_tie_plane);
        }
        HIP_TRY(c, hipEventRecord(c->ev_join, c->side));
    }
    // everything after NMS reads the pools: enqueued once; if sibling ties had to be decided by a flood order walk, the planes whose
    // pool that changed are classified again (the others keep their records).  calc_color + er_track run when the candidates are final
    // and the host knows how many of them are strong / weak (below): the colour pass is sized for exactly those boxes.
    if (stages & STR_ER_STAGE_NMS) {
        launch_cand_prefix(s, bd);
        launch_classify(s, bd, dp, c->casc[0].dev, c->casc[1].dev, (stages & STR_ER_STAGE_CLASSIFY) ? 1 : 0, nullptr, nullptr, np <= SPEC_PLANES);
    }
    rec(c, "classify", nullptr, true);
    const int i_cls = c->n_ev - 1;
    int       i_trk = -1;
    // the scorer of STR_ER_STAGE_OCR behind classify, sized from the last batch (ocr_stage): no trip to the host between the two, like the reference's
    // call site (src/ER.cpp:728-735)
    size_t ocr_cap = 0, ocr_cap2 = 0;
    bool   cands_remade = false;
    if ((stages & STR_ER_STAGE_OCR) && c->ocr_spec && c->ocr_last_n > 0 && c->pool_total) {
        ocr_cap = align_up(c->ocr_last_n + c->ocr_last_n / 8 + 256, 128);
        const int rco = ocr_stage(c, bd, ocr_cap, s, false);
        if (rco != STR_ER_OK) return rco;
    }
    if (alt_pass) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
    HIP_TRY(c, hipGetLastError());
    const CandRec *spec_src = nullptr;
    uint32_t       spec_n = 0;
    if ((stages & STR_ER_STAGE_NMS) && np <= SPEC_PLANES && c->pool_total && c->last_total <= SPEC_CANDS) {
        spec_src = c->d_cands; spec_n = (uint32_t)std::min<size_t>(SPEC_CANDS, c->pool_total);
    }
    if (np <= SPEC_PLANES) {
        // a call of a frame or two: counters and (if they are likely to fit) the candidate records go to the page-locked block in one launch
        launch_results_to_host(s, c->d_zero, c->h_zero, 256 + sizeof(PlaneCtr) * np, spec_src, spec_src ? c->h_cands_spec : nullptr, spec_n, c->d_total);
        HIP_TRY(c, hipGetLastError());
    } else {
        HIP_TRY(c, hipMemcpyAsync(c->h_zero, c->d_zero, 256 + sizeof(PlaneCtr) * np, hipMemcpyDeviceToHost, s));        // (counters of the batch + plane counters: one block)
    }
    HIP_TRY(c, wait_stream(c, s));
    slot.give();                       // the batch's kernels are done: what follows on the host (counters, tie walks, results) leaves the GPU to the other calls
    if (c->n_t2_tiles && !import_trees) {     // tiles k_tile_tree2 handed back: if they are many, the chroma planes stay with k_tile_tree for a while
        const uint32_t fbn = c->h_total[1];
        c->t2_tiles_total += c->n_t2_tiles; c->t2_fb_total += fbn;
        if (c->t2_mode == 1 && (uint64_t)fbn * 8u > c->n_t2_tiles) c->t2_backoff = 32;
    } else if (c->t2_backoff > 0) --c->t2_backoff;
    {   // what the tree passes of this batch worked on (str_er_last_tree_stats: bench.py prices them against the HBM roofline)
        uint64_t recs = 0, pairs = 0, tiles = 0;
        for (int i = 0; i < np; ++i) {
            const PlaneDesc &pd = b.planes[i];
            if (!(c->h_ctr[i].overflow & 8u)) recs += c->h_ctr[i].n_nodes;
            pairs += (uint64_t)(pd.tiles_y > 0 ? pd.tiles_y - 1 : 0) * (uint64_t)pd.w + (uint64_t)(pd.tiles_x > 0 ? pd.tiles_x - 1 : 0) * (uint64_t)pd.h;
            tiles += (uint64_t)pd.tiles_x * (uint64_t)pd.tiles_y;
        }
        c->last_tree_records = recs; c->last_tree_pairs = pairs; c->last_tree_tiles = tiles;
    }
    {   // a plane ran out of node records: the counters say how many it wanted -- grow the share and do the batch again
        double need = 0;
        for (int i = 0; i < np; ++i)
            if (c->h_ctr[i].overflow & 8u)
                need = std::max(need, (double)c->h_ctr[i].n_nodes / (double)((size_t)b.planes[i].tiles_x * b.planes[i].tiles_y * TILE_PX));
        if (need > 0) {
            if (c->node_share >= 1.0) return fail(c, STR_ER_ECAPACITY, "node records exhausted at one record per pixel (internal error)");
            c->node_share = std::min(1.0, std::max(c->node_share * 1.5, need * 1.25));
            const size_t want = (size_t)std::ceil((double)c->slots * c->node_share) + 256 * (size_t)c->max_planes;
            if (want > c->node_slots) {
                const int rcn = alloc_node_records(c, want);
                if (rcn != STR_ER_OK) return rcn;
            }
            if (ev_entry >= 0) { c->n_ev = ev_entry; c->profile.resize((size_t)ev_entry); }
            return run_batch(c, b_in, stages, out, t_start, pre_recorded, import_trees, attempt + 1);
        }
    }
    // the same for the kept-node table (the counter says how many nodes the plane has) and the pool (it does not: double).  Checked here and
    // once more after the tie pass, whose pools can be larger than the first pass's.
    auto grow_tables = [&](bool &again) -> int {
        again = false;
        if (!c->auto_caps) return STR_ER_OK;
        double need_k = 0;
        bool   more_pool = false;
        for (int i = 0; i < np; ++i) {
            const double px = (double)((size_t)b.planes[i].tiles_x * b.planes[i].tiles_y * TILE_PX);
            if (c->h_ctr[i].overflow & 1u) need_k = std::max(need_k, (double)c->h_ctr[i].n_kept / px);
            else if (c->h_ctr[i].overflow & 2u) more_pool = true;
        }
        if (need_k == 0 && !more_pool) return STR_ER_OK;
        const double kept0 = c->kept_share, pool0 = c->pool_share;
        if (need_k > 0) c->kept_share = std::min(1.0, std::max(c->kept_share * 1.5, need_k * 1.25));
        if (more_pool) {
            // (the pool is a subset of the kept nodes: past the kept share it is the kept table that has to grow with it)
            c->pool_share = std::min(1.0, c->pool_share * 2.0);
            if (c->pool_share > c->kept_share) c->kept_share = c->pool_share;
        }
        if (c->kept_share == kept0 && c->pool_share == pool0)
            return fail(c, STR_ER_ECAPACITY, "kept-node / pool tables exhausted at one entry per pixel (internal error)");
        again = true;
        return STR_ER_OK;
    };
    {
        bool again = false;
        const int rcg = grow_tables(again);
        if (rcg != STR_ER_OK) return rcg;
        if (again) {
            if (ev_entry >= 0) { c->n_ev = ev_entry; c->profile.resize((size_t)ev_entry); }
            return run_batch(c, b_in, stages, out, t_start, pre_recorded, import_trees, attempt + 1);      // (re-laid out, tables re-allocated on entry)
        }
    }
    if ((stages & STR_ER_STAGE_NMS) && c->prm.sibling_order == 0) {
        bool replayed = false;
        const auto tr0 = std::chrono::steady_clock::now();
        const int rcr = resolve_sibling_ties(c, b, bd, dp, replayed);
        if (rcr != STR_ER_OK) return rcr;
        const auto tr1 = std::chrono::steady_clock::now();
        if (replayed && c->dbg_stats) std::fprintf(stderr, "[str_er] tie resolution (copies + walk + NMS pass): %.1f ms\n", std::chrono::duration<double, std::milli>(tr1 - tr0).count());
        if (replayed) {
            cands_remade = true;
            hipStream_t sp = c->prio ? c->prio : s;         // same stream as the tie pass: ordered behind it
            const CandRec *first = c->d_cands;
            std::swap(c->d_cands, c->d_cands2);
            std::swap(c->d_cand_plane, c->d_cand_plane2);
            bd = make_batchdev(c, b);
            uint32_t *n_redo = c->d_redo + c->pool_total;
            launch_cand_reprefix(sp, bd, first, c->d_redo, n_redo);
            launch_classify(sp, bd, dp, c->casc[0].dev, c->casc[1].dev, (stages & STR_ER_STAGE_CLASSIFY) ? 1 : 0, c->d_redo, n_redo, true);
            if (ocr_cap && *ocr_pinned(c, ocr_cap).count <= ocr_cap) {
                // the batch was scored behind classify: the candidates of the re-made planes are scored behind THEIR classify, again without a trip to the
                // host -- as many as the device lists (the scratch is sized for the whole batch)
                ocr_cap2 = ocr_cap;
                const int rco = ocr_stage(c, bd, ocr_cap2, sp, true, c->d_redo, n_redo, 1);
                if (rco != STR_ER_OK) return rco;
            }
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(c->h_zero, c->d_zero, 256 + sizeof(PlaneCtr) * np, hipMemcpyDeviceToHost, sp));
            HIP_TRY(c, wait_stream(c, sp));
            if (c->dbg_stats) std::fprintf(stderr, "[str_er] classify again after the tie pass: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr1).count());
            bool again = false;             // (a pool of the tie pass did not fit: same remedy as above)
            const int rcg = grow_tables(again);
            if (rcg != STR_ER_OK) return rcg;
            if (again) {
                std::swap(c->d_cands, c->d_cands2);         // (back to the buffers the first pass writes)
                std::swap(c->d_cand_plane, c->d_cand_plane2);
                if (ev_entry >= 0) { c->n_ev = ev_entry; c->profile.resize((size_t)ev_entry); }
                return run_batch(c, b_in, stages, out, t_start, pre_recorded, import_trees, attempt + 1);
            }
        }
    }

    if (c->dbg_stats) {     // developer aid: how many nodes left the tiles
        unsigned long long tot = 0, created = 0;
        int n_tied = 0, n_rel = 0;
        for (int i = 0; i < np; ++i) { created += c->h_ctr[i].n_created; tot += c->h_ctr[i].n_nodes; n_tied += c->h_ctr[i].n_amb != 0; n_rel += c->h_ctr[i].n_rel != 0; }
        std::fprintf(stderr, "[str_er] planes %d, with NMS sibling ties %d, with ties that can change the pool (flood replayed) %d\n", np, n_tied, n_rel);
        for (int i = 0; i < np; ++i)
            if (c->h_ctr[i].n_amb)
                std::fprintf(stderr, "[str_er]   tie plane %d (frame %u ch %d pyr %d, %dx%d): ties %u relevant %u, last tie node: %u contenders, kept %u pool %u\n", i,
                             b.planes[i].frame, b.planes[i].ch, b.planes[i].pyr, b.planes[i].w, b.planes[i].h, c->h_ctr[i].n_amb, c->h_ctr[i].n_rel,
                             c->h_ctr[i].tie_nc, c->h_ctr[i].n_kept, c->h_ctr[i].n_pool);
        std::fprintf(stderr, "[str_er] tiles %u exported nodes %llu (%.1f per tile; share %.3f of %.3f) created %llu seam pairs %u\n", b.n_tiles, tot,
                     (double)tot / b.n_tiles, (double)tot / (double)b.slots, c->node_share, created, b.n_pairs);
    }
    for (int i = 0; i < np; ++i) {
        if (c->h_ctr[i].overflow & 1u)
            return fail(c, STR_ER_ECAPACITY, "kept-node table overflow: plane " + std::to_string(i) + " has " +
                        std::to_string(c->h_ctr[i].n_kept) + " kept nodes, its share of the table is " + std::to_string(b.planes[i].kept_cap) + " (set kept_cap)");
        if (c->h_ctr[i].overflow & 2u) return fail(c, STR_ER_ECAPACITY, "NMS pool overflow: raise pool_cap");
    }
    {
        uint32_t most = 0;
        for (int i = 0; i < np; ++i) most = std::max(most, c->h_ctr[i].n_nodes);
        // (more lanes than this in flight only queue up behind the same hot parent words: measured on noise, 256 workgroups per
        // plane made k_resolve 4x slower than 12, 32 10 % slower; text-like batches -- the ones the small tile kernel runs on -- were
        // fastest with 24 while the kernels combined a wave's targets in a loop.  Round 6, with the per-wave tables: a batch of a thousand planes does not care
        // (24 / 48 / 96: resolve + accumulate 0.43 / 0.41 / 0.43 ms), but a wave of k_reduce walks its chains towards the root one sweep step after the other, so
        // a call of a frame or two wants its few big planes spread over many waves (one 1080p frame: accumulate 117 / 72 / 73 us, resolve 36 / 28 / 22) and a 4K
        // batch a little (accumulate 0.45 / 0.39 / 0.33, resolve 0.26 / 0.29 / 0.32 -- more waves, less combining))
        const uint32_t cap = c->node_blocks_cap ? c->node_blocks_cap : (c->tile_sparse ? (np <= 96 ? 96u : 48u) : 12u);
        // (in powers of two: the number is part of the layout key, and a video whose frames ask for 79, 98, 85 ... would upload its tables again call after call)
        uint32_t want = 4;
        while (want < (most + 255) / 256 && want < 256u) want *= 2;
        c->node_blocks = std::min<uint32_t>(cap, want);
    }
    if (c->tile_mode == 0 && b.n_tiles) {      // text-like frames make a few dozen nodes per tile, noise several hundred
        unsigned long long created = 0;
        for (int i = 0; i < np; ++i) created += c->h_ctr[i].n_created;
        const double per_tile = (double)created / (double)b.n_tiles;
        if (per_tile > 320.0) c->tile_sparse = false;
        else if (per_tile < 240.0) c->tile_sparse = true;
    }
    str_er_result *r = new (std::nothrow) str_er_result();
    if (!r) return fail(c, STR_ER_ENOMEM, "result allocation");
    const uint32_t total = *c->h_total;
    c->last_total = total; c->last_valid = (stages & STR_ER_STAGE_NMS) != 0;
    r->cands.resize(total);
    r->cand_off.assign(np + 1, 0);
    r->planes.resize(np);
    if (total && spec_src == c->d_cands && total <= spec_n)        // (same buffer as at the time of the copy: no tie pass re-made the records)
        std::memcpy(r->cands.data(), c->h_cands_spec, sizeof(CandRec) * (size_t)total);
    else if (total)
        if (hipMemcpyAsync(r->cands.data(), c->d_cands, sizeof(CandRec) * (size_t)total, hipMemcpyDeviceToHost, s) != hipSuccess) {
            delete r; return fail(c, STR_ER_EHIP, "candidate copy failed");
        }
    if ((stages & STR_ER_STAGE_TRACK)) {
        // calc_color + er_track on the final candidates: the strong / weak ones are listed, their boxes' Otsu thresholds and masked colour means
        // computed a wave per box (big boxes by many workgroups), then er_track per image
        size_t n_cls = 0;
        for (int i = 0; i < np; ++i) n_cls += c->h_ctr[i].n_strong + c->h_ctr[i].n_weak;
        rec(c, "track_host_gap", nullptr, true);
        if (total) {
            const int    n_img = np / b.planes_per_image;
            const size_t o_list = 0, o_cs = align_up(4 * ((size_t)total + OCR_LIST_HDR) + 256, 256);
            const int    rcs = ensure_scratch(c, o_cs + calc_color_scratch_bytes(n_cls));
            if (rcs != STR_ER_OK) { delete r; return rcs; }
            uint8_t  *sc = static_cast<uint8_t *>(c->d_scratch);
            uint32_t *d_list = reinterpret_cast<uint32_t *>(sc + o_list);
            if (hipMemsetAsync(c->d_track, 0, sizeof(TrackRec) * (size_t)total, s) != hipSuccess) { delete r; return fail(c, STR_ER_EHIP, "track reset failed"); }
            if (n_cls) {
                launch_ocr_list(s, bd, (uint32_t)total, d_list);
                OcrSrc src{};
                src.recs = bd.cands; src.list = d_list + OCR_LIST_HDR; src.planes = bd.planes;
                launch_calc_color(s, src, ColorSrc{}, (int)n_cls, c->d_track, sc + o_cs);
            }
            launch_group_ranges(s, bd, b.planes_per_image, n_img, c->d_ranges);
            launch_er_track(s, bd.cands, c->d_track, c->d_track_list, c->d_ranges, n_img);
        }
        rec(c, "track", nullptr, true);
        i_trk = c->n_ev - 1;
        r->tracks.resize(total);
        r->have_tracks = true;
        static_assert(sizeof(str_er_track) == sizeof(TrackRec), "track record layout");
        if (total && (wait_stream(c, s) != hipSuccess ||          // (wait, then copy: see the OCR stage below)
                      hipMemcpyAsync(r->tracks.data(), c->d_track, sizeof(TrackRec) * (size_t)total, hipMemcpyDeviceToHost, s) != hipSuccess)) {
            delete r; return fail(c, STR_ER_EHIP, "track copy failed");
        }
    }
    double t_group_s = 0;
    if (stages & STR_ER_STAGE_GROUP) {
        const auto tg0 = std::chrono::steady_clock::now();
        if (wait_stream(c, s) != hipSuccess) { delete r; return fail(c, STR_ER_EHIP, "sync before grouping"); }
        std::vector<uint32_t> img;
        const int n_img = np / b.planes_per_image;
        uint32_t off2 = 0;
        for (int g = 0; g < n_img; ++g) {
            img.push_back(off2);
            for (int k = 0; k < b.planes_per_image; ++k) off2 += c->h_ctr[g * b.planes_per_image + k].n_pool;
            img.push_back(off2);
        }
        const int rcg = (stages & STR_ER_GROUP_OVERLAP_SUP) ? group_phase_overlap(c, img, (stages & STR_ER_GROUP_INNER_SUP) != 0, r)
                                                            : group_phase(c, c->d_cands, c->d_track, img, (stages & STR_ER_GROUP_INNER_SUP) != 0, r);
        if (rcg != STR_ER_OK) { delete r; return rcg; }
        t_group_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count();
    }
    const auto t_ocr0 = std::chrono::steady_clock::now();
    if (stages & STR_ER_STAGE_OCR_LINES) {
        const int rcl = line_ocr_phase(c, c->d_planes, r);
        if (rcl != STR_ER_OK) { delete r; return rcl; }
    }
    if (stages & STR_ER_STAGE_OCR) r->have_ocr = true;      // (an empty table, not a missing one, when there are no candidates)
    if ((stages & STR_ER_STAGE_OCR) && total) {
        size_t n_ocr = 0;
        for (int i = 0; i < np; ++i) n_ocr += c->h_ctr[i].n_strong + c->h_ctr[i].n_weak;
        r->ocr_label.assign(total, -1);
        r->ocr_prob.assign(total, 0.0);
        r->have_ocr = true;
        // scored behind classify already?  Only if the guess covered the batch.  If an NMS tie pass re-made the candidates of a few planes since, the scores of
        // the other planes are still good (their records were moved, PlaneCtr::cand_base_old -> cand_base) and only the re-made planes are scored again
        const uint32_t early_n = ocr_cap ? *ocr_pinned(c, ocr_cap).count : 0u;
        const bool     early = ocr_cap != 0 && early_n <= ocr_cap && (cands_remade || early_n == n_ocr);
        auto scatter = [&](size_t cap, size_t n, int region = 0) {
            const OcrPinned hp = ocr_pinned(c, cap, region);
            for (size_t i = 0; i < n; ++i) {
                if (hp.list[i] >= total) continue;
                r->ocr_label[hp.list[i]] = hp.label[i];
                r->ocr_prob[hp.list[i]] = hp.prob[i];
            }
        };
        auto run_stage = [&](size_t cap, size_t expect, const uint32_t *from, const uint32_t *from_n) -> int {
            bd = make_batchdev(c, b);
            hipStream_t so = (cands_remade && c->prio) ? c->prio : s;        // (behind the tie pass and its classify)
            const int rc2 = ocr_stage(c, bd, cap, so, ocr_cap != 0, from, from_n);
            if (rc2 != STR_ER_OK) return rc2;
            if (wait_stream(c, so) != hipSuccess) return fail(c, STR_ER_EHIP, "OCR stage failed");
            if (*ocr_pinned(c, cap).count != expect)
                return fail(c, STR_ER_EHIP, "OCR stage: the device listed a different number of strong / weak ERs than the plane counters say (internal error)");
            return STR_ER_OK;
        };
        if (ocr_cap) { if (early && !cands_remade) ++c->n_ocr_spec; else ++c->n_ocr_redo; }
        if (early && !cands_remade) {
            scatter(ocr_cap, n_ocr);
        } else if (early) {
            const OcrPinned hp = ocr_pinned(c, ocr_cap);
            size_t n2 = 0;
            int    pl = 0;
            for (int i = 0; i < np; ++i) if (c->h_ctr[i].pool_changed) n2 += c->h_ctr[i].n_strong + c->h_ctr[i].n_weak;
            for (size_t i = 0; i < early_n; ++i) {                // (the list is in candidate order: the planes come by)
                const uint32_t old = hp.list[i];
                while (pl + 1 < np && c->h_ctr[pl + 1].cand_base_old <= old) ++pl;
                const PlaneCtr &pc = c->h_ctr[pl];
                if (pc.pool_changed || old < pc.cand_base_old) continue;
                const uint32_t now = old - pc.cand_base_old + pc.cand_base;
                if (now >= total) continue;
                r->ocr_label[now] = hp.label[i];
                r->ocr_prob[now] = hp.prob[i];
            }
            if (n2 && ocr_cap2 && n2 <= ocr_cap2 && *ocr_pinned(c, ocr_cap2, 1).count == n2) {
                scatter(ocr_cap2, n2, 1);
            } else if (n2) {
                const int rc2 = run_stage(n2, n2, c->d_redo, c->d_redo + c->pool_total);
                if (rc2 != STR_ER_OK) { delete r; return rc2; }
                scatter(n2, n2);
            }
        } else if (n_ocr) {
            // the slow way: the host knows the number now
            const int rc2 = run_stage(n_ocr, n_ocr, nullptr, nullptr);
            if (rc2 != STR_ER_OK) { delete r; return rc2; }
            scatter(n_ocr, n_ocr);
        }
        c->ocr_last_n = n_ocr;
    }
    const double t_ocr_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ocr0).count();
    const bool want_nodes = (stages & STR_ER_WANT_NODES) != 0;
    if (want_nodes) {
        r->nodes.resize(np);
        for (int i = 0; i < np; ++i) {
            const PlaneDesc &pd = b.planes[i];
            const uint32_t nk = c->h_ctr[i].n_kept;
            r->nodes[i].resize(nk);
            // gather the SoA kept arrays into records on the host side of the copy
            std::vector<uint32_t> key(nk), area(nk); std::vector<int32_t> par(nk); std::vector<uint16_t> box(4 * (size_t)nk);
            std::vector<uint8_t> lev(nk);
            hipError_t e = hipSuccess;
            if (nk) {
                e = hipMemcpyAsync(key.data(), c->ka.key + pd.kept_base, 4 * (size_t)nk, hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipMemcpyAsync(area.data(), c->ka.area + pd.kept_base, 4 * (size_t)nk, hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipMemcpyAsync(par.data(), c->ka.parent + pd.kept_base, 4 * (size_t)nk, hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipMemcpyAsync(box.data(), c->ka.box + 4 * (size_t)pd.kept_base, 8 * (size_t)nk, hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipMemcpyAsync(lev.data(), c->ka.level + pd.kept_base, (size_t)nk, hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = wait_stream(c, s);
            }
            if (e != hipSuccess) { delete r; return fail(c, STR_ER_EHIP, std::string("node copy: ") + hipGetErrorString(e)); }
            // order by (key, level) so the table is deterministic; remap parents and the root
            std::vector<uint32_t> order(nk), rank(nk);
            for (uint32_t k = 0; k < nk; ++k) order[k] = k;
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t bb) {
                return key[a] != key[bb] ? key[a] < key[bb] : lev[a] < lev[bb]; });
            for (uint32_t k = 0; k < nk; ++k) rank[order[k]] = k;
            for (uint32_t k = 0; k < nk; ++k) {
                const uint32_t o = order[k];
                str_er_node &n = r->nodes[i][k];
                n.key = key[o]; n.parent = (int32_t)rank[(uint32_t)par[o]]; n.area = (int32_t)area[o];
                n.x = box[4 * o]; n.y = box[4 * o + 1]; n.w = box[4 * o + 2]; n.h = box[4 * o + 3];
                n.level = lev[o]; n.flags = (o == c->h_ctr[i].root_slot) ? 1 : 0; n.reserved = 0;
            }
            c->h_ctr[i].root_slot = nk ? rank[c->h_ctr[i].root_slot] : 0;
        }
    }
    HIP_TRY(c, wait_stream(c, s));

    uint32_t off = 0;
    for (int i = 0; i < np; ++i) {
        const PlaneDesc &pd = b.planes[i];
        const PlaneCtr &pc = c->h_ctr[i];
        str_er_plane_info &pi = r->planes[i];
        pi.frame = pd.frame; pi.ch = pd.ch; pi.pyr = pd.pyr; pi.reserved0 = pi.reserved1 = 0;
        pi.width = pd.w; pi.height = pd.h;
        pi.n_created = (int32_t)pc.n_created; pi.n_kept = (int32_t)pc.n_kept;
        pi.n_pool = (int32_t)pc.n_pool; pi.n_strong = (int32_t)pc.n_strong; pi.n_weak = (int32_t)pc.n_weak;
        pi.ambiguous = (int32_t)pc.n_amb; pi.root = want_nodes ? (int32_t)pc.root_slot : -1;
        r->cand_off[i] = off;
        off += pc.n_pool;
    }
    r->cand_off[np] = off;
    if (want_nodes) {
        // candidates carry the device kept slot; translate to the sorted table through (key, level)
        for (int i = 0; i < np; ++i) {
            const auto &tbl = r->nodes[i];
            for (uint32_t k = r->cand_off[i]; k < r->cand_off[i + 1]; ++k) {
                str_er_cand &cd = r->cands[k];
                auto it = std::lower_bound(tbl.begin(), tbl.end(), cd, [](const str_er_node &n, const str_er_cand &q) {
                    return n.key != q.key ? n.key < q.key : n.level < q.level; });
                cd.node = (it != tbl.end() && it->key == cd.key && it->level == cd.level) ? (int32_t)(it - tbl.begin()) : -1;
            }
        }
        r->have_nodes = true;
    } else {
        for (auto &cd : r->cands) cd.node = -1;
    }

    float ms = 0;
    double stage_s[3] = {0, 0, 0};
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[i_extract]) == hipSuccess) stage_s[0] = ms * 1e-3;
    if (hipEventElapsedTime(&ms, c->ev[i_extract], c->ev[i_nms]) == hipSuccess) stage_s[1] = ms * 1e-3;
    if (hipEventElapsedTime(&ms, c->ev[i_nms], c->ev[i_cls]) == hipSuccess) stage_s[2] = ms * 1e-3;
    for (int i = 1; i < c->n_ev; ++i)
        if (hipEventElapsedTime(&ms, c->ev[i - 1], c->ev[i]) == hipSuccess) c->profile[i].second = ms;
    r->times[0] = stage_s[0]; r->times[1] = stage_s[1]; r->times[2] = stage_s[2];
    if ((stages & STR_ER_STAGE_TRACK) && i_trk > 0 && hipEventElapsedTime(&ms, c->ev[i_trk - 1], c->ev[i_trk]) == hipSuccess) r->times[3] = ms * 1e-3;
    if (stages & (STR_ER_STAGE_OCR | STR_ER_STAGE_OCR_LINES)) r->times[5] = t_ocr_s;
    r->times[4] = t_group_s;
    r->times[6] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    *out = r;
    {   // everything of this batch has been read: the counter block is zeroed for the context's next batch now, behind the caller's back (upload_layout)
        const size_t need = align_up(c->zero_gd_off + b.n_groups, 256);
        if (hipMemsetAsync(c->d_zero, 0, need, s) == hipSuccess) c->zero_clean_bytes = need;
    }
    return STR_ER_OK;
}

int stage_input(str_er_ctx *c, const uint8_t *src, size_t bytes, int mem_kind, const uint8_t **dev)
{
    if (mem_kind == STR_ER_MEM_DEVICE) { *dev = src; return STR_ER_OK; }
    if (bytes > c->in_bytes) return fail(c, STR_ER_ECAPACITY, "host input larger than the staging buffer");
    HIP_TRY(c, hipMemcpyAsync(c->d_in, src, bytes, hipMemcpyHostToDevice, c->stream));
    *dev = c->d_in;
    return STR_ER_OK;
}

} // namespace str_er_host

extern "C" {

int str_er_abi_version(void) { return STR_ER_ABI_VERSION; }

// A context uses three HIP streams (main, alt NMS pass, tie pass) and applications keep several contexts in flight; on the runtime's
// default of 4 hardware queues the long single-workgroup kernels of one context's tie pass end up in front of another context's
// tile kernel (bench.py: 4940 -> 5330 frames/s with 16).  The HIP runtime reads the variable when it initialises, so it has to be
// in the environment before the process's first HIP call.  The library never touches the environment by itself: the host either
// exports what str_er_runtime_hint() names or calls str_er_apply_runtime_hint() -- an explicit opt-in -- before it initialises HIP.
int str_er_tie_stats(const str_er_ctx *c, uint64_t *planes_walked, double *walk_ms_total, int32_t *host_threads)
try {
    if (!c) return STR_ER_EINVAL;
    if (planes_walked) *planes_walked = c->n_replayed;
    if (walk_ms_total) *walk_ms_total = c->walk_ms_total;
    if (host_threads) *host_threads = flood_walk_threads();
    return STR_ER_OK;
} ABI_GUARD(const_cast<str_er_ctx *>(c))

int str_er_last_tree_stats(const str_er_ctx *c, uint64_t *records, uint64_t *seam_pairs, uint64_t *tiles)
try {
    if (!c) return STR_ER_EINVAL;
    if (records) *records = c->last_tree_records;
    if (seam_pairs) *seam_pairs = c->last_tree_pairs;
    if (tiles) *tiles = c->last_tree_tiles;
    return STR_ER_OK;
} ABI_GUARD(const_cast<str_er_ctx *>(c))

int str_er_tile2_stats(const str_er_ctx *c, uint64_t *tiles, uint64_t *handed_back)
try {
    if (!c) return STR_ER_EINVAL;
    if (tiles) *tiles = c->t2_tiles_total;
    if (handed_back) *handed_back = c->t2_fb_total;
    return STR_ER_OK;
} ABI_GUARD(const_cast<str_er_ctx *>(c))

const char *str_er_runtime_hint(void) { return "GPU_MAX_HW_QUEUES=16"; }
int str_er_set_batch_slots(int n)
{
    std::lock_guard<std::mutex> lk(g_slot_mu);
    const int old = g_slot_cap;
    g_slot_cap = n > 0 ? n : 0;
    g_slot_cv.notify_all();
    return old;
}
int str_er_ocr_stage_stats(const str_er_ctx *c, uint64_t *scored_early, uint64_t *scored_again)
{
    if (!c) return STR_ER_EINVAL;
    if (scored_early) *scored_early = c->n_ocr_spec;
    if (scored_again) *scored_again = c->n_ocr_redo;
    return STR_ER_OK;
}

int str_er_apply_runtime_hint(void)
{
    if (getenv("GPU_MAX_HW_QUEUES")) return 0;
    return setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0) == 0 ? 1 : STR_ER_EINVAL;
}

const char *str_er_strerror(int code)
{
    switch (code) {
    case STR_ER_OK: return "ok";
    case STR_ER_EINVAL: return "invalid argument";
    case STR_ER_ENOMEM: return "out of memory";
    case STR_ER_EHIP: return "HIP runtime error";
    case STR_ER_EIO: return "I/O error";
    case STR_ER_EFORMAT: return "bad classifier format";
    case STR_ER_ESTATE: return "cascades not loaded";
    case STR_ER_ECAPACITY: return "capacity exceeded";
    default: return "unknown error";
    }
}

void str_er_default_params(str_er_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->thresh_step = 8; p->min_area = 120; p->max_area = 900000; p->stability_t = 2; p->overlap_coef = 0.7;
    p->n_pyr_levels = 1; p->channel_mask = 0x3F; p->device = 0;
    p->max_width = 1920; p->max_height = 1080; p->max_frames = 8;
    p->kept_cap = 0; p->pool_cap = 0; p->sibling_order = 0; p->stream = nullptr;
}

const char *str_er_last_error(const str_er_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void str_er_destroy(str_er_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->prm.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_strip_out) (void)hipFree(c->d_strip_out);
    if (c->d_strip_in) (void)hipFree(c->d_strip_in);
    if (c->d_replay) (void)hipFree(c->d_replay);
    if (c->h_replay) (void)hipHostFree(c->h_replay);
    if (c->h_tie) (void)hipHostFree(c->h_tie);
    if (c->na.rec) (void)hipFree(c->na.rec);
    if (c->na.aux) (void)hipFree(c->na.aux);
    for (void *p : {(void *)c->ka.node, (void *)c->ka.key, (void *)c->ka.area, (void *)c->ka.parent, (void *)c->ka.box, (void *)c->ka.level, (void *)c->ka.start,
                    (void *)c->ka.ncand, (void *)c->ka.best, (void *)c->ka.perm, (void *)c->d_pool, (void *)c->d_pool_tmp, (void *)c->d_cands, (void *)c->d_cand_plane, (void *)c->d_cands2,
                    (void *)c->d_cand_plane2, (void *)c->d_redo, (void *)c->d_track, (void *)c->d_track_list})
        if (p) (void)hipFree(p);
    if (c->d_group) (void)hipFree(c->d_group);
    if (c->d_group_pairs) (void)hipFree(c->d_group_pairs);
    for (auto &hc : c->casc) if (hc.d_blob) (void)hipFree(hc.d_blob);
    if (c->d_svm_blob) (void)hipFree(c->d_svm_blob);
    if (c->h_planes) (void)hipHostFree(c->h_planes);
    if (c->h_zero) (void)hipHostFree(c->h_zero);
    if (c->h_cands_spec) (void)hipHostFree(c->h_cands_spec);
    if (c->h_ocr) (void)hipHostFree(c->h_ocr);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->prio) { (void)hipStreamSynchronize(c->prio); (void)hipStreamDestroy(c->prio); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int str_er_create(const str_er_params *p, str_er_ctx **out)
{
    if (!p || !out) return fail(nullptr, STR_ER_EINVAL, "null argument");
    *out = nullptr;
    if (p->thresh_step < 1 || p->thresh_step > 255) return fail(nullptr, STR_ER_EINVAL, "thresh_step must be in [1,255]");
    if (p->stability_t < 0 || p->stability_t > 255) return fail(nullptr, STR_ER_EINVAL, "stability_t must be in [0,255]");
    if (p->n_pyr_levels < 1 || p->n_pyr_levels > 32) return fail(nullptr, STR_ER_EINVAL, "n_pyr_levels must be in [1,32]");
    if (!(p->channel_mask & 0x3F) || (p->channel_mask & ~0x3Fu)) return fail(nullptr, STR_ER_EINVAL, "channel_mask must select planes 0..5");
    if (p->max_width < 1 || p->max_height < 1 || p->max_frames < 1) return fail(nullptr, STR_ER_EINVAL, "capacity must be positive");
    if (p->max_width > 65535 || p->max_height > 65535) return fail(nullptr, STR_ER_EINVAL, "planes are limited to 65535 x 65535");
    {   // node ids are 24-bit (PAR_ID in er_kernels.hip): tiles * 2048 of one plane must stay below 2^24
        const size_t padded = (size_t)((p->max_width + TILE_W - 1) / TILE_W) * ((p->max_height + TILE_H - 1) / TILE_H) * TILE_PX;
        if (padded > (1u << 24)) return fail(nullptr, STR_ER_EINVAL, "planes are limited to 2^24 pixels (after padding to 64x32 tiles)");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, STR_ER_EHIP, "no HIP device available: this library has no CPU path");
    if (p->device < 0 || p->device >= ndev) return fail(nullptr, STR_ER_EINVAL, "device ordinal out of range");
    if (hipSetDevice(p->device) != hipSuccess) return fail(nullptr, STR_ER_EHIP, "hipSetDevice failed");

    str_er_ctx *c = new (std::nothrow) str_er_ctx();
    if (!c) return fail(nullptr, STR_ER_ENOMEM, "context allocation");
    c->prm = *p;
    if (const char *tk = std::getenv("STR_ER_TILE_KERNEL")) {      // developer switch: force one size of the tile kernel
        if (!std::strcmp(tk, "sparse")) { c->tile_mode = 1; c->tile_sparse = true; }
        else if (!std::strcmp(tk, "dense")) { c->tile_mode = 2; c->tile_sparse = false; }
    }
    if (const char *g = std::getenv("STR_ER_GROUPS")) c->group_mode = std::atoi(g) == 0 ? 0 : -1;      // developer switch: 0 = tiles joined by the global passes only
    if (const char *g = std::getenv("STR_ER_GROUP_X")) c->dbg_group[0] = std::atoi(g);
    if (const char *g = std::getenv("STR_ER_GROUP_Y")) c->dbg_group[1] = std::atoi(g);
    if (const char *g = std::getenv("STR_ER_GROUP_KERNEL")) c->dbg_group[2] = std::atoi(g);
    if (const char *t2 = std::getenv("STR_ER_TILE2")) c->t2_mode = std::max(0, std::min(2, std::atoi(t2)));      // developer switch: 0 k_tile_tree only, 2 k_tile_tree2 on every plane
    c->dbg_tile_only = std::getenv("STR_ER_DEBUG_TILE_ONLY") != nullptr;
    if (c->dbg_tile_only) c->profiling = true;
    c->dbg_stats = std::getenv("STR_ER_DEBUG_STATS") != nullptr;
    if (const char *os = std::getenv("STR_ER_OCR_SPEC")) c->ocr_spec = std::atoi(os) != 0;      // developer switch: 0 = the scorer is sized after the counters were read
    if (const char *nb = std::getenv("STR_ER_NODE_BLOCKS")) c->node_blocks_cap = (uint32_t)std::max(1, std::atoi(nb));
    if (const char *rp = std::getenv("STR_ER_REPLAY")) c->replay_on_gpu = !std::strcmp(rp, "gpu");
    for (int i = 0; i < 6; ++i) if (p->channel_mask & (1u << i)) c->chans.push_back(i);
    c->ppf = (int)c->chans.size() * p->n_pyr_levels;
    c->max_planes = c->ppf * p->max_frames;
    if (c->max_planes > 65535) { delete c; return fail(nullptr, STR_ER_ECAPACITY, "more than 65535 planes per call (max_frames x channels x levels)"); }
    size_t px_frame = 0, phys_frame = 0;
    for (int l = 0; l < p->n_pyr_levels; ++l) {
        int w, h; pyr_dims(p->max_width, p->max_height, l, w, h);
        px_frame += (size_t)((w + TILE_W - 1) / TILE_W) * ((h + TILE_H - 1) / TILE_H) * TILE_PX * c->chans.size();
        phys_frame += 3 * align_up((size_t)align_up(w, 64) * h, 256);
    }
    c->slots = px_frame * p->max_frames;
    if (c->slots >= 0xFFFF0000ull) { delete c; return fail(nullptr, STR_ER_ECAPACITY, "batch too large: more than 2^32 pixels per call"); }
    const size_t plane_px = (size_t)p->max_width * p->max_height;
    c->kept_cap = p->kept_cap > 0 ? p->kept_cap : (int)std::max<size_t>(4096, plane_px / 64);
    c->pool_cap = p->pool_cap > 0 ? p->pool_cap : std::max(256, c->kept_cap / 4);
    c->auto_caps = p->kept_cap <= 0 && p->pool_cap <= 0;
    c->seam_slots = c->slots / 8 + 4096;

    int rc = STR_ER_OK;
    auto A = [&](int r) { if (rc == STR_ER_OK && r != STR_ER_OK) rc = r; };
    if (p->stream) c->stream = static_cast<hipStream_t>(p->stream);
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(nullptr, STR_ER_EHIP, "hipStreamCreate failed"); }
        c->own_stream = true;
    }
    for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { A(fail(nullptr, STR_ER_EHIP, "hipEventCreate failed")); break; }
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);       // (numerically lower = higher priority)
        // (Round 6: no longer created unless STR_ER_PRIO_STREAM is set -- the tie pass then runs on the main stream.  A third stream per context is a third hardware
        // queue: with six contexts the line gains 1 % without it, seven are 2 % better than six, and the cliff eight contexts fell off -- 13.4 -> 10.9 k frames/s at 24
        // streams -- is gone: 8 / 9 / 10 contexts 13.65 / 13.69 / 13.77 k.)
        if (std::getenv("STR_ER_PRIO_STREAM") && hipStreamCreateWithPriority(&c->prio, hipStreamNonBlocking, hi) != hipSuccess) A(fail(nullptr, STR_ER_EHIP, "priority stream creation failed"));
    }
    if (p->sibling_order == 0) {
        c->tie_slot_bytes = ((plane_px + 255) / 256) * 256 + 2 * 4 * (size_t)NMS_WATCH_CAP + 256;
        c->n_tie_slots = (int)std::min<size_t>((size_t)TIE_SLOTS, std::max<size_t>(4, ((size_t)64 << 20) / c->tie_slot_bytes));
        const size_t tb = (size_t)c->n_tie_slots * c->tie_slot_bytes + 4 * (size_t)TIE_SLOTS + 64;
        if (hipHostMalloc(reinterpret_cast<void **>(&c->h_tie), tb, hipHostMallocMapped) != hipSuccess) A(fail(nullptr, STR_ER_ENOMEM, "hipHostMalloc (tie plane export)"));
        else {
            c->h_tie_plane = reinterpret_cast<uint32_t *>(c->h_tie + (size_t)c->n_tie_slots * c->tie_slot_bytes);
            c->h_tie_count = c->h_tie_plane + TIE_SLOTS;
        }
    }
    c->spin_wait = std::getenv("STR_ER_SPIN_WAIT") != nullptr;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)
        A(fail(nullptr, STR_ER_EHIP, "side stream creation failed"));
    c->in_bytes = std::max((size_t)p->max_frames * plane_px * 3, c->slots);
    c->pix_bytes = std::max(phys_frame * p->max_frames, c->slots + 4096);
    // kept-node and pool tables: by default a plane's share follows its (padded) pixel count (assign_tables); explicit caps are given to
    // every plane
    const size_t S = c->slots;
    const size_t KP = c->auto_caps ? (size_t)std::ceil((double)S * c->kept_share) + 512 * (size_t)c->max_planes : (size_t)c->max_planes * c->kept_cap;
    const size_t PP = c->auto_caps ? (size_t)std::ceil((double)S * c->pool_share) + 256 * (size_t)c->max_planes : (size_t)c->max_planes * c->pool_cap;
    A(dev_alloc(c, c->d_in, c->in_bytes));
    A(dev_alloc(c, c->d_pix, c->pix_bytes));
    A(dev_alloc(c, c->d_planes, (size_t)c->max_planes));
    A(alloc_node_records(c, (size_t)std::ceil((double)S * c->node_share) + 256 * (size_t)c->max_planes));
    A(alloc_tables(c, KP, PP));
    A(dev_alloc(c, c->d_seam, c->seam_slots));
    c->tile_slots = c->slots / TILE_PX + 16;
    c->sb_slots = c->seam_slots / (2 * (size_t)std::min(SEAM_BLOCK, 256)) + (size_t)c->max_planes + 16;
    A(dev_alloc(c, c->d_tile_plane, c->tile_slots)); A(dev_alloc(c, c->d_sb_plane, c->sb_slots)); A(dev_alloc(c, c->d_sb_first, c->sb_slots));
    A(dev_alloc(c, c->d_tile_nbase, c->tile_slots));
    A(dev_alloc(c, c->d_t1_list, c->tile_slots)); A(dev_alloc(c, c->d_t2_pairs, c->tile_slots)); A(dev_alloc(c, c->d_fb_list, c->tile_slots));
    A(dev_alloc(c, c->d_nb_plane, (size_t)c->max_planes * str_er_ctx::NB_PLANE_SHARE));
    A(dev_alloc(c, c->d_tile_nrec, c->tile_slots)); A(dev_alloc(c, c->d_group_plane, c->tile_slots)); A(dev_alloc(c, c->d_undone, c->tile_slots)); A(dev_alloc(c, c->d_group_list, c->tile_slots));
    A(dev_alloc(c, c->d_ranges, 2 * (size_t)c->max_planes + 2));
    {   // what a batch starts from zero -- the candidate / handed-back-tile counters, the plane counters, the groups' done flags -- is ONE block: one
        // memset per batch, and the counters come back with one copy (a call of one frame is a chain of ~30 operations: each costs 5 - 10 us)
        c->zero_gd_off = 256 + align_up(sizeof(PlaneCtr) * (size_t)c->max_planes, 256);
        A(dev_alloc(c, c->d_zero, align_up(c->zero_gd_off + c->tile_slots, 256)));
        if (c->d_zero) {
            c->d_total = reinterpret_cast<uint32_t *>(c->d_zero); c->d_ctr = reinterpret_cast<PlaneCtr *>(c->d_zero + 256);
            c->d_group_done = c->d_zero + c->zero_gd_off;
        }
    }
    A(dev_alloc(c, c->d_watch, (size_t)c->max_planes * NMS_WATCH_CAP)); A(dev_alloc(c, c->d_wstamp, (size_t)c->max_planes * NMS_WATCH_CAP)); A(dev_alloc(c, c->d_wparent, (size_t)c->max_planes * NMS_WATCH_CAP));
    A(dev_alloc(c, c->d_replay_items, (size_t)c->max_planes));
    A(dev_alloc(c, c->d_tie_slot_plane, (size_t)TIE_SLOTS));
    A(dev_alloc(c, c->d_alt_list, (size_t)NMS_ALT_CAP));
    A(dev_alloc(c, c->d_strip_flag, (size_t)1));
    if (rc == STR_ER_OK) {
        if (hipHostMalloc(reinterpret_cast<void **>(&c->h_planes), sizeof(PlaneDesc) * c->max_planes) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&c->h_zero), 256 + sizeof(PlaneCtr) * c->max_planes) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&c->h_cands_spec), sizeof(CandRec) * SPEC_CANDS) != hipSuccess)
            rc = fail(nullptr, STR_ER_ENOMEM, "hipHostMalloc failed");
    }
    if (rc == STR_ER_OK) { c->h_total = reinterpret_cast<uint32_t *>(c->h_zero); c->h_ctr = reinterpret_cast<PlaneCtr *>(c->h_zero + 256); }
    if (rc != STR_ER_OK) { std::string keep = g_create_error.empty() ? c->err : g_create_error; str_er_destroy(c); g_create_error = keep; return rc; }
    *out = c;
    return STR_ER_OK;
}

int str_er_set_thresh_step(str_er_ctx *c, int32_t t)
try {
    if (!c) return STR_ER_EINVAL;
    if (t < 1 || t > 255) return fail(c, STR_ER_EINVAL, "thresh_step must be in [1,255]");
    c->prm.thresh_step = t;
    return STR_ER_OK;
} ABI_GUARD(c)

int str_er_set_min_area(str_er_ctx *c, int32_t m)
try {
    if (!c) return STR_ER_EINVAL;
    c->prm.min_area = m;
    return STR_ER_OK;
} ABI_GUARD(c)

static int detect_bgr_impl(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch, int32_t n_frames,
                           int mem_kind, uint32_t stages, const uint8_t *plane_select, str_er_result **out, bool nv12 = false);

int str_er_detect_bgr(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch,
                      int32_t n_frames, int mem_kind, uint32_t stages, str_er_result **out)
try {
    return detect_bgr_impl(c, bgr, w, h, stride, frame_pitch, n_frames, mem_kind, stages, nullptr, out);
} ABI_GUARD(c)

int str_er_detect_bgr_planes(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch,
                             int32_t n_frames, int mem_kind, uint32_t stages, const uint8_t *plane_select, int32_t n_select,
                             str_er_result **out)
try {
    if (!c) return STR_ER_EINVAL;
    if (!plane_select || n_select != c->ppf) return fail(c, STR_ER_EINVAL, "plane_select needs one flag per logical plane of a frame (levels x channels of the context)");
    if (stages & (STR_ER_STAGE_TRACK | STR_ER_STAGE_GROUP | STR_ER_STAGE_OCR_LINES))
        return fail(c, STR_ER_EINVAL, "er_track / er_grouping read every plane of an image: not with a plane subset");
    bool any = false;
    for (int i = 0; i < n_select; ++i) any |= plane_select[i] != 0;
    if (!any) return fail(c, STR_ER_EINVAL, "plane_select selects nothing");
    return detect_bgr_impl(c, bgr, w, h, stride, frame_pitch, n_frames, mem_kind, stages, plane_select, out);
} ABI_GUARD(c)

int str_er_detect_nv12(str_er_ctx *c, const uint8_t *nv12, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch,
                       int32_t n_frames, int mem_kind, uint32_t stages, str_er_result **out)
try {
    return detect_bgr_impl(c, nv12, w, h, stride, frame_pitch, n_frames, mem_kind, stages, nullptr, out, /*nv12=*/true);
} ABI_GUARD(c)

// (nv12: `bgr` is a luma plane of h rows followed by the interleaved chroma plane of h / 2 rows, `stride` bytes per row both)
static int detect_bgr_impl(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int64_t frame_pitch, int32_t n_frames,
                           int mem_kind, uint32_t stages, const uint8_t *plane_select, str_er_result **out, bool nv12)
{
    if (!c) return STR_ER_EINVAL;
    const int64_t row_bytes = nv12 ? (int64_t)w : (int64_t)w * 3;                 // bytes of a source row
    const int64_t src_rows = nv12 ? (int64_t)h + h / 2 : (int64_t)h;              // rows of a source frame
    if (!bgr || !out || w < 1 || h < 1 || n_frames < 1 || stride < row_bytes) return fail(c, STR_ER_EINVAL, "bad frame arguments");
    if (nv12 && ((w | h) & 1)) return fail(c, STR_ER_EINVAL, "NV12 frames have even width and height");
    if (n_frames > 1 && frame_pitch < stride * src_rows) return fail(c, STR_ER_EINVAL, "frame_pitch smaller than a frame");
    if (w > c->prm.max_width || h > c->prm.max_height || n_frames > c->prm.max_frames)
        return fail(c, STR_ER_ECAPACITY, "frame larger than / more frames than the context capacity");
    *out = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->prm.device));
    const uint8_t *dbgr = nullptr;
    int64_t dstride = stride, dpitch = frame_pitch;
    if (mem_kind == STR_ER_MEM_HOST) {
        // pack rows tightly while staging
        dstride = row_bytes; dpitch = dstride * src_rows;
        if ((size_t)dpitch * n_frames > c->in_bytes) return fail(c, STR_ER_ECAPACITY, "staging buffer too small");
        if (stride == dstride && (n_frames == 1 || frame_pitch == dpitch))      // already tight: one linear copy
            HIP_TRY(c, hipMemcpyAsync(c->d_in, bgr, (size_t)dpitch * n_frames, hipMemcpyHostToDevice, c->stream));
        else
            for (int f = 0; f < n_frames; ++f)
                HIP_TRY(c, hipMemcpy2DAsync(c->d_in + (size_t)f * dpitch, (size_t)dstride, bgr + (size_t)f * frame_pitch, (size_t)stride,
                                            (size_t)row_bytes, (size_t)src_rows, hipMemcpyHostToDevice, c->stream));
        dbgr = c->d_in;
    } else if (mem_kind == STR_ER_MEM_DEVICE) dbgr = bgr;
    else return fail(c, STR_ER_EINVAL, "bad mem_kind");

    // physical planes: per level, [Y, Cr, Cb], row stride padded to 64 bytes (a plane subset builds the pyramid only as far
    // down as its deepest selected level)
    int nl = c->prm.n_pyr_levels;
    if (plane_select) {
        int deepest = 0;
        for (int l = 0; l < nl; ++l)
            for (size_t k = 0; k < c->chans.size(); ++k) if (plane_select[(size_t)l * c->chans.size() + k]) deepest = l;
        nl = deepest + 1;
    }
    std::vector<PlaneGeom> geo(nl);
    size_t frame_bytes = 0;
    for (int l = 0; l < nl; ++l) {
        pyr_dims(w, h, l, geo[l].w, geo[l].h);
        geo[l].stride = (int)align_up(geo[l].w, 64);
        geo[l].off = frame_bytes;
        frame_bytes += 3 * align_up((size_t)geo[l].stride * geo[l].h, 256);
    }
    if (frame_bytes * n_frames > c->pix_bytes) return fail(c, STR_ER_ECAPACITY, "plane pool too small");
    auto plane_sz = [&](int l) { return align_up((size_t)geo[l].stride * geo[l].h, 256); };
    c->n_ev = 0; c->profile.clear(); rec(c, "begin", nullptr, true);
    const hipStream_t ws = c->stream;
    if (nv12)
        launch_nv12_to_ycrcb(ws, dbgr, w, h, dstride, dpitch, n_frames, c->d_pix + geo[0].off, c->d_pix + geo[0].off + plane_sz(0),
                             c->d_pix + geo[0].off + 2 * plane_sz(0), geo[0].stride, (int64_t)frame_bytes);
    else
        launch_bgr_to_ycrcb(ws, dbgr, w, h, dstride, dpitch, n_frames, c->d_pix + geo[0].off, c->d_pix + geo[0].off + plane_sz(0),
                            c->d_pix + geo[0].off + 2 * plane_sz(0), geo[0].stride, (int64_t)frame_bytes);
    rec(c, "channels", ws);
    for (int l = 1; l < nl; ++l)
        launch_resize(ws, c->d_pix + geo[l - 1].off, geo[l - 1].w, geo[l - 1].h, geo[l - 1].stride, (int64_t)plane_sz(l - 1),
                      (int64_t)frame_bytes, c->d_pix + geo[l].off, geo[l].w, geo[l].h, geo[l].stride, (int64_t)plane_sz(l),
                      (int64_t)frame_bytes, 3, n_frames);
    rec(c, "pyramid", ws);

    Batch b;
    for (int f = 0; f < n_frames; ++f)
        for (int l = 0; l < nl; ++l)
            for (size_t k = 0; k < c->chans.size(); ++k) {
                if (plane_select && !plane_select[(size_t)l * c->chans.size() + k]) continue;
                const int ch = c->chans[k];
                const uint8_t *pix = c->d_pix + (size_t)f * frame_bytes + geo[l].off + (size_t)(ch % 3) * plane_sz(l);
                add_plane(b, pix, geo[l].w, geo[l].h, geo[l].stride, ch >= 3, (uint32_t)f, ch, l);
                b.planes.back().color_pitch = (uint32_t)plane_sz(l);
            }
    b.planes_per_image = plane_select ? 0 : (int)c->chans.size();
    return run_batch(c, b, stages, out, t0, true);
}

int str_er_detect_planes(str_er_ctx *c, const uint8_t *planes, int32_t w, int32_t h, int64_t stride, int64_t plane_pitch,
                         int32_t n_planes, int mem_kind, uint32_t stages, str_er_result **out)
try {
    if (!c) return STR_ER_EINVAL;
    if (!planes || !out || w < 1 || h < 1 || n_planes < 1 || stride < w) return fail(c, STR_ER_EINVAL, "bad plane arguments");
    if (n_planes > 1 && plane_pitch < stride * (int64_t)h) return fail(c, STR_ER_EINVAL, "plane_pitch smaller than a plane");
    if (w > c->prm.max_width || h > c->prm.max_height || n_planes > c->max_planes)
        return fail(c, STR_ER_ECAPACITY, "plane larger than / more planes than the context capacity");
    if (stride > 0x7FFFFFFF) return fail(c, STR_ER_EINVAL, "stride too large");
    *out = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->prm.device));
    const uint8_t *dp = nullptr;
    int dstride = (int)stride;
    int64_t dpitch = plane_pitch;
    if (mem_kind == STR_ER_MEM_HOST) {
        dstride = (int)align_up(w, 64); dpitch = (int64_t)align_up((size_t)dstride * h, 256);
        if ((size_t)dpitch * n_planes > c->pix_bytes) return fail(c, STR_ER_ECAPACITY, "plane pool too small");
        for (int i = 0; i < n_planes; ++i)
            HIP_TRY(c, hipMemcpy2DAsync(c->d_pix + (size_t)i * dpitch, (size_t)dstride, planes + (size_t)i * plane_pitch, (size_t)stride,
                                        (size_t)w, (size_t)h, hipMemcpyHostToDevice, c->stream));
        dp = c->d_pix;
    } else if (mem_kind == STR_ER_MEM_DEVICE) dp = planes;
    else return fail(c, STR_ER_EINVAL, "bad mem_kind");
    Batch b;
    for (int i = 0; i < n_planes; ++i)
        add_plane(b, dp + (size_t)i * dpitch, w, h, dstride, 0, 0, i & 255, 0);
    return run_batch(c, b, stages, out, t0, false);
} ABI_GUARD(c)

int str_er_internal_last_cands(str_er_ctx *c, const void **d_cands, uint32_t *n, int *device)
try {
    if (!c || !c->last_valid) return STR_ER_ESTATE;
    *d_cands = c->d_cands; *n = c->last_total; *device = c->prm.device;
    return STR_ER_OK;
} ABI_GUARD(c)

int str_er_result_plane_info(const str_er_result *r, int32_t plane, str_er_plane_info *info)
{
    if (!r || !info || plane < 0 || plane >= (int32_t)r->planes.size()) return STR_ER_EINVAL;
    *info = r->planes[plane];
    return STR_ER_OK;
}

const str_er_plane_info *str_er_result_plane_infos(const str_er_result *r, int32_t *n)
{
    if (!r) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->planes.size();
    return r->planes.data();
}

const str_er_cand *str_er_result_cands(const str_er_result *r, int32_t *n)
{
    if (!r) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->cands.size();
    return r->cands.data();
}

const str_er_cand *str_er_result_plane_cands(const str_er_result *r, int32_t plane, int32_t *n)
{
    if (!r || plane < 0 || plane >= (int32_t)r->planes.size()) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)(r->cand_off[plane + 1] - r->cand_off[plane]);
    return r->cands.data() + r->cand_off[plane];
}

const str_er_node *str_er_result_plane_nodes(const str_er_result *r, int32_t plane, int32_t *n)
{
    if (!r || !r->have_nodes || plane < 0 || plane >= (int32_t)r->planes.size()) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->nodes[plane].size();
    return r->nodes[plane].data();
}

const int32_t *str_er_result_ocr_labels(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_ocr) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->ocr_label.size();
    static const int32_t none = 0;
    return r->ocr_label.empty() ? &none : r->ocr_label.data();
}

const double *str_er_result_ocr_probs(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_ocr) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->ocr_prob.size();
    static const double none = 0;
    return r->ocr_prob.empty() ? &none : r->ocr_prob.data();
}

const str_er_track *str_er_result_tracks(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_tracks) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->tracks.size();
    static const str_er_track none{};
    return r->tracks.empty() ? &none : r->tracks.data();     // never NULL once the stage has run
}

const str_er_text *str_er_result_texts(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_texts) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->texts.size();
    static const str_er_text none{};
    return r->texts.empty() ? &none : r->texts.data();
}

const int32_t *str_er_result_text_ers(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_texts) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->text_ers.size();
    static const int32_t none = 0;
    return r->text_ers.empty() ? &none : r->text_ers.data();
}

const int32_t *str_er_result_group_all(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_texts) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->group_all.size();
    static const int32_t none = 0;
    return r->group_all.empty() ? &none : r->group_all.data();
}

const int32_t *str_er_result_line_labels(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_line_ocr) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->line_label.size();
    static const int32_t none = 0;
    return r->line_label.empty() ? &none : r->line_label.data();
}

const double *str_er_result_line_probs(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_line_ocr) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->line_prob.size();
    static const double none = 0;
    return r->line_prob.empty() ? &none : r->line_prob.data();
}

const uint8_t *str_er_result_line_kept(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_line_ocr) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->line_kept.size();
    static const uint8_t none = 0;
    return r->line_kept.empty() ? &none : r->line_kept.data();
}

const uint8_t *str_er_result_text_alive(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_line_ocr) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->text_alive.size();
    static const uint8_t none = 0;
    return r->text_alive.empty() ? &none : r->text_alive.data();
}

const str_er_gbound *str_er_result_group_bounds(const str_er_result *r, int32_t *n)
{
    if (!r || !r->have_texts) { if (n) *n = 0; return nullptr; }
    if (n) *n = (int32_t)r->gbounds.size();
    static const str_er_gbound none{};
    return r->gbounds.empty() ? &none : r->gbounds.data();
}

const double *str_er_result_times(const str_er_result *r) { return r ? r->times : nullptr; }

int str_er_result_cands_to_device(str_er_ctx *c, const str_er_result *r, void *dst_dev, int32_t cap, int32_t *n)
try {
    if (!c) return STR_ER_EINVAL;
    if (!r || !dst_dev || cap < 0 || !n) return fail(c, STR_ER_EINVAL, "bad arguments");
    HIP_TRY(c, hipSetDevice(c->prm.device));
    const int32_t m = std::min<int32_t>((int32_t)r->cands.size(), cap);
    *n = (int32_t)r->cands.size();
    if (m > 0) {
        HIP_TRY(c, hipMemcpyAsync(dst_dev, r->cands.data(), sizeof(str_er_cand) * (size_t)m, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, wait_stream(c, c->stream));
    }
    return STR_ER_OK;
} ABI_GUARD(c)

void str_er_result_free(str_er_result *r) { delete r; }

int str_er_last_profile(const str_er_ctx *c, const char **names, double *ms, int32_t cap)
try {
    if (!c || !c->profiling) return 0;          // (without profiling the few events of a call are not one per kernel group)
    int k = 0;
    for (size_t i = 1; i < c->profile.size(); ++i, ++k)
        if (k < cap) { if (names) names[k] = c->profile[i].first; if (ms) ms[k] = c->profile[i].second; }
    return k;
} ABI_GUARD(const_cast<str_er_ctx *>(c))

int str_er_set_profiling(str_er_ctx *c, int enable)
try {
    if (!c) return STR_ER_EINVAL;
    c->profiling = enable != 0;
    return STR_ER_OK;
} ABI_GUARD(c)

int64_t str_er_workspace_bytes(const str_er_ctx *c) { return c ? c->ws_bytes : 0; }

} // extern "C"
