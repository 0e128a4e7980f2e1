// api_strips.cpp -- the C ABI, part 3: one plane in horizontal strips over several GPUs
#include "str_er_ctx.h"

extern "C" {

// =================================================================================================
// one plane in horizontal strips over several GPUs.
//
// Whole planes are the unit the path shards by (8(e)); one large frame has few of them and the three level-0 planes bound the
// speed-up (3840x2160, 12 levels: 6.0x on 8 GPUs).  The tile kernel is the largest part of the work and has no data flow between
// tiles, so a plane can be cut into strips of tile rows: every GPU builds the tile trees of its strip and joins the seams INSIDE the
// strip (str_er_strip_extract); what crosses the wire is the strip's node records (32 bytes per exported node: about a quarter of
// the strip's pixel bytes on text-like frames) and the node of every pixel of its first / last row.  The owner of the plane puts the
// strips' records behind one another, makes the ids plane-wide, joins the pixel pairs across every cut with the same connect as any
// other seam and carries on with the usual passes -- resolve, accumulate, prune, NMS, classify (str_er_strip_merge).  The node
// set of a component tree does not depend on the order in which tiles are joined, so the result is that of the unsplit plane.
// Strips are cut from the level-0 planes (in a pyramid context too: its smaller planes are dealt out whole, str_er_detect_bgr_planes).
// The blob is assembled ON THE DEVICE and can stay there: with an RCCL communicator (str_er_comm_allgather_bytes) it goes from the
// extracting GPU's memory into the owner's without touching a host; the host-memory entry points copy it once.
//
//   blob = StripHeader | StripPlane x n_planes | per plane: records | per plane: node of every pixel of the first row (if the plane
//          goes on above), of the last row (if it goes on below) -- sections start on 256-byte boundaries
// =================================================================================================
extern "C++" {          // (helpers with C++ types, inside the file's extern "C" block)
namespace {

constexpr uint32_t STRIP_MAGIC = 0x50525453u;      // "STRP"
constexpr uint32_t STRIP_VERSION = 2;
struct StripHeader { uint32_t magic, version, w, h, strip, n_strips, row0 /* plane row of the records' row 0 */, rows, n_planes, thresh_step, channel_mask, reserved; };
struct StripPlane { uint32_t ch, n_nodes, n_walls, start_node, has_top, has_bot; };      // (records: ids, keys and rows local to the strip)
struct StripLayout { size_t head = 0, total = 0; std::vector<size_t> rec, top, bot; };

StripLayout strip_layout(const std::vector<StripPlane> &sp, uint32_t w)
{
    StripLayout L;
    const size_t n = sp.size();
    L.head = sizeof(StripHeader) + n * sizeof(StripPlane);
    size_t at = align_up(L.head, 256);
    L.rec.resize(n); L.top.resize(n); L.bot.resize(n);
    for (size_t k = 0; k < n; ++k) { L.rec[k] = at; at = align_up(at + (size_t)sp[k].n_nodes * sizeof(NodeRec), 256); }
    for (size_t k = 0; k < n; ++k) {
        L.top[k] = at; if (sp[k].has_top) at = align_up(at + 4 * (size_t)w, 256);
        L.bot[k] = at; if (sp[k].has_bot) at = align_up(at + 4 * (size_t)w, 256);
    }
    L.total = at;
    return L;
}

// channels of one BGR frame into the level-0 planes of the pixel pool (what str_er_detect_bgr does for level 0)
int frame_to_planes(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int mem_kind, int &pstride, size_t &psize)
{
    const uint8_t *dbgr = nullptr;
    int64_t dstride = stride;
    if (mem_kind == STR_ER_MEM_HOST) {
        dstride = (int64_t)w * 3;
        if ((size_t)dstride * h > c->in_bytes) return fail(c, STR_ER_ECAPACITY, "staging buffer too small");
        HIP_TRY(c, hipMemcpy2DAsync(c->d_in, (size_t)dstride, bgr, (size_t)stride, (size_t)w * 3, (size_t)h, hipMemcpyHostToDevice, c->stream));
        dbgr = c->d_in;
    } else if (mem_kind == STR_ER_MEM_DEVICE) dbgr = bgr;
    else return fail(c, STR_ER_EINVAL, "bad mem_kind");
    pstride = (int)align_up((size_t)w, 64);
    psize = align_up((size_t)pstride * h, 256);
    if (3 * psize > c->pix_bytes) return fail(c, STR_ER_ECAPACITY, "plane pool too small");
    launch_bgr_to_ycrcb(c->stream, dbgr, w, h, dstride, dstride * h, 1, c->d_pix, c->d_pix + psize, c->d_pix + 2 * psize, pstride, (int64_t)(3 * psize));
    return STR_ER_OK;
}

int ensure_strip_buf(str_er_ctx *c, uint8_t *&p, size_t &cap, size_t need)
{
    if (need <= cap) return STR_ER_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    need += need / 4;
    if (hipMalloc(reinterpret_cast<void **>(&p), need) != hipSuccess) return fail(c, STR_ER_ENOMEM, "hipMalloc (strip blob)");
    cap = need;
    return STR_ER_OK;
}

} // namespace
} // extern "C++"

int str_er_strip_extract_dev(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int mem_kind, int32_t strip, int32_t n_strips,
                             const void **d_blob, int64_t *blob_bytes)
try {
    if (!c) return STR_ER_EINVAL;
    if (!bgr || !d_blob || !blob_bytes || w < 1 || h < 1 || stride < (int64_t)w * 3 || n_strips < 1 || strip < 0 || strip >= n_strips)
        return fail(c, STR_ER_EINVAL, "bad strip arguments");
    if (w > c->prm.max_width || h > c->prm.max_height) return fail(c, STR_ER_ECAPACITY, "frame larger than the context capacity");
    *d_blob = nullptr; *blob_bytes = 0;
    HIP_TRY(c, hipSetDevice(c->prm.device));
    c->last_valid = false;
    int pstride = 0; size_t psize = 0;
    int rc = frame_to_planes(c, bgr, w, h, stride, mem_kind, pstride, psize);
    if (rc != STR_ER_OK) return rc;
    const int ty_all = (h + TILE_H - 1) / TILE_H;
    const int t0 = (int)((int64_t)strip * ty_all / n_strips), t1 = (int)((int64_t)(strip + 1) * ty_all / n_strips);
    const int r0 = t0 * TILE_H, r1 = std::min(h, t1 * TILE_H), rows = std::max(0, r1 - r0);
    const size_t npl = c->chans.size();
    std::vector<StripPlane> sp(npl);
    for (size_t k = 0; k < npl; ++k) { sp[k] = StripPlane{}; sp[k].ch = (uint32_t)c->chans[k]; sp[k].start_node = NONE; }
    hipStream_t s = c->stream;
    Batch b;
    const bool ptop = rows > 0 && r0 > 0, pbot = rows > 0 && r1 < h;
    if (rows > 0) {
        // A strip is laid out as a plane with a PHANTOM tile row above and / or below wherever the plane goes on: the strip's first /
        // last row is then an ordinary tile seam -- its nodes stay open and their ids are in the seam map -- and the tile kernel needs to
        // know nothing about strips (no strip flags, no second code path in the largest kernel of the step).  Below, the phantom row is simply
        // past the image (one tile row more than the height needs).  Above, the strip is copied behind TILE_H rows of pixels at the
        // sentinel level, which the flood never enters -- hence the restriction to thresh_steps that have such a level.
        const size_t pad_plane = align_up((size_t)pstride * (size_t)(rows + TILE_H), 256);
        if (ptop) {
            if ((int)std::lrintf(255.0f * (float)(1.0 / (double)c->prm.thresh_step)) != 255 / c->prm.thresh_step + 1)
                return fail(c, STR_ER_EINVAL, "strips need a thresh_step whose top level is the sentinel level (2, 4, 8, 16 ...: round(255/step) = 255/step + 1)");
            rc = ensure_scratch(c, pad_plane * npl);
            if (rc != STR_ER_OK) return rc;
        }
        for (size_t k = 0; k < npl; ++k) {
            const int ch = c->chans[k];
            const uint8_t *src = c->d_pix + (size_t)(ch % 3) * psize + (size_t)r0 * pstride;
            const uint8_t *lay = src;
            if (ptop) {
                uint8_t *dst = static_cast<uint8_t *>(c->d_scratch) + k * pad_plane;
                HIP_TRY(c, hipMemsetAsync(dst, ch >= 3 ? 0x00 : 0xFF, (size_t)TILE_H * pstride, s));      // (inverted channels read pixel ^ 0xFF)
                HIP_TRY(c, hipMemcpyAsync(dst + (size_t)TILE_H * pstride, src, (size_t)rows * pstride, hipMemcpyDeviceToDevice, s));
                lay = dst;
            }
            add_plane(b, lay, w, rows + (ptop ? TILE_H : 0) + (pbot ? TILE_H : 0), pstride, ch >= 3, 0, ch, 0);
            PlaneDesc &pd = b.planes.back();
            pd.h = rows + (ptop ? TILE_H : 0);                   // (the phantom row below is simply past the image)
            pd.n_pairs = pd.n_hpairs + (uint32_t)pd.h * (uint32_t)(pd.tiles_x - 1);
        }
        const DetectParams dp = make_dp(c);
        for (int attempt = 0;; ++attempt) {
            assign_node_records(b, c->node_share);
            if (b.nodes > c->node_slots) { rc = alloc_node_records(c, b.nodes + b.nodes / 8); if (rc != STR_ER_OK) return rc; }
            if (b.seam > c->seam_slots || b.n_tiles > c->tile_slots || b.slots > c->slots) return fail(c, STR_ER_ECAPACITY, "strip exceeds the context capacity");
            c->layout_key.clear();                 // (a strip's layout is not a frame's: never reuse the cached tables for it)
            rc = upload_layout(c, b);
            c->layout_key.clear();
            if (rc != STR_ER_OK) return rc;
            const BatchDev bd = make_batchdev(c, b);
            // the seam row of the phantom tile row below lies past the image: no tile writes it, so it is blanked here ("wall")
            if (pbot)
                for (const PlaneDesc &pd : b.planes)
                    HIP_TRY(c, hipMemsetAsync(c->d_seam + pd.seam_base + (size_t)(2 * (pd.tiles_y - 2) + 1) * w, 0xFF, 2 * (size_t)w, s));
            launch_tile_tree(s, bd, dp, c->tile_sparse);
            launch_seam(s, bd, !c->tile_sparse);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(PlaneCtr) * npl, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, wait_stream(c, s));
            double need = 0;
            for (size_t k = 0; k < npl; ++k)
                if (c->h_ctr[k].overflow & 8u) need = std::max(need, (double)c->h_ctr[k].n_nodes / (double)((size_t)b.planes[k].tiles_x * b.planes[k].tiles_y * TILE_PX));
            if (need == 0) break;
            if (c->node_share >= 1.0 || attempt > 24) return fail(c, STR_ER_ECAPACITY, "node records exhausted at one record per pixel (internal error)");
            c->node_share = std::min(1.0, std::max(c->node_share * 1.5, need * 1.25));
        }
        for (size_t k = 0; k < npl; ++k) {
            const PlaneCtr &pc = c->h_ctr[k];
            sp[k].n_nodes = pc.n_nodes; sp[k].n_walls = pc.n_walls; sp[k].start_node = r0 == 0 ? pc.start_node : NONE;
            sp[k].has_top = ptop; sp[k].has_bot = pbot;
        }
    }
    // what leaves the GPU, put together on the GPU: the records, and the node of every pixel of the first / last row (seam map:
    // index in the tile's records; tile_nbase: the tile's first record)
    const StripLayout L = strip_layout(sp, (uint32_t)w);
    rc = ensure_strip_buf(c, c->d_strip_out, c->strip_out_cap, L.total);
    if (rc != STR_ER_OK) return rc;
    std::vector<uint8_t> head(L.head);
    const StripHeader hd{STRIP_MAGIC, STRIP_VERSION, (uint32_t)w, (uint32_t)h, (uint32_t)strip, (uint32_t)n_strips, (uint32_t)std::max(0, r0 - (r0 > 0 ? TILE_H : 0)), (uint32_t)rows,
                         (uint32_t)npl, (uint32_t)c->prm.thresh_step, c->prm.channel_mask, 0u};
    std::memcpy(head.data(), &hd, sizeof(hd));
    std::memcpy(head.data() + sizeof(hd), sp.data(), npl * sizeof(StripPlane));
    HIP_TRY(c, hipMemcpyAsync(c->d_strip_out, head.data(), L.head, hipMemcpyHostToDevice, s));
    for (size_t k = 0; k < npl && rows > 0; ++k) {
        const PlaneDesc &pd = b.planes[k];
        if (sp[k].n_nodes) HIP_TRY(c, hipMemcpyAsync(c->d_strip_out + L.rec[k], c->na.rec + pd.node_base, (size_t)sp[k].n_nodes * sizeof(NodeRec), hipMemcpyDeviceToDevice, s));
        // seam map: boundary j holds pixel row (j+1)*TILE_H - 1 at [2j * w, +w) and pixel row (j+1)*TILE_H at [(2j+1) * w, +w)
        const int jt = 0, jb = pd.tiles_y - 2;               // the seams under the phantom row above / over the phantom row below
        if (ptop) launch_strip_border_ids(s, c->d_seam + pd.seam_base + (size_t)(2 * jt + 1) * w, c->d_tile_nbase + pd.tile_base + (size_t)(jt + 1) * pd.tiles_x, w,
                                          reinterpret_cast<uint32_t *>(c->d_strip_out + L.top[k]));
        if (pbot) launch_strip_border_ids(s, c->d_seam + pd.seam_base + (size_t)(2 * jb) * w, c->d_tile_nbase + pd.tile_base + (size_t)jb * pd.tiles_x, w,
                                          reinterpret_cast<uint32_t *>(c->d_strip_out + L.bot[k]));
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, wait_stream(c, s));        // (also: `head` is pageable memory)
    *d_blob = c->d_strip_out; *blob_bytes = (int64_t)L.total;
    return STR_ER_OK;
} ABI_GUARD(c)

int str_er_strip_extract(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int mem_kind, int32_t strip, int32_t n_strips,
                         void **blob, int64_t *blob_bytes)
try {
    if (!c) return STR_ER_EINVAL;
    if (!blob || !blob_bytes) return fail(c, STR_ER_EINVAL, "bad strip arguments");
    *blob = nullptr; *blob_bytes = 0;
    const void *d = nullptr;
    int64_t n = 0;
    const int rc = str_er_strip_extract_dev(c, bgr, w, h, stride, mem_kind, strip, n_strips, &d, &n);
    if (rc != STR_ER_OK) return rc;
    uint8_t *out = static_cast<uint8_t *>(std::malloc((size_t)n));
    if (!out) return fail(c, STR_ER_ENOMEM, "strip blob allocation");
    if (hipMemcpy(out, d, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) { std::free(out); return fail(c, STR_ER_EHIP, "strip blob download"); }
    *blob = out; *blob_bytes = n;
    return STR_ER_OK;
} ABI_GUARD(c)

void str_er_strip_free(void *blob) { std::free(blob); }

int str_er_strip_merge_ex(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int mem_kind, const void *const *blobs,
                          const int64_t *blob_bytes, int blob_kind, int32_t n_strips, const uint8_t *plane_select, uint32_t stages, str_er_result **out)
try {
    if (!c) return STR_ER_EINVAL;
    if (!bgr || !blobs || !blob_bytes || !out || w < 1 || h < 1 || stride < (int64_t)w * 3 || n_strips < 1 ||
        (blob_kind != STR_ER_MEM_HOST && blob_kind != STR_ER_MEM_DEVICE))
        return fail(c, STR_ER_EINVAL, "bad strip arguments");
    if (w > c->prm.max_width || h > c->prm.max_height) return fail(c, STR_ER_ECAPACITY, "frame larger than the context capacity");
    *out = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->prm.device));
    const size_t npl = c->chans.size();
    // the planes this call puts together (the others' strips are skipped: another owner merges them)
    std::vector<size_t> sel;
    for (size_t k = 0; k < npl; ++k) if (!plane_select || plane_select[k]) sel.push_back(k);
    if (sel.empty()) return fail(c, STR_ER_EINVAL, "plane_select selects no plane");
    // ---- the blobs: on the device (as they are, or uploaded once); their headers on the host, checked before anything is trusted ----
    const size_t head_bytes = sizeof(StripHeader) + npl * sizeof(StripPlane);
    struct View { StripHeader hd; std::vector<StripPlane> sp; StripLayout L; const uint8_t *d; };
    std::vector<View> view((size_t)n_strips);
    size_t up_total = 0;
    for (int i = 0; i < n_strips; ++i) {
        if (!blobs[i] || blob_bytes[i] < (int64_t)head_bytes) return fail(c, STR_ER_EFORMAT, "strip blob too short");
        up_total += align_up((size_t)blob_bytes[i], 256);
    }
    if (blob_kind == STR_ER_MEM_HOST) {
        const int rcb = ensure_strip_buf(c, c->d_strip_in, c->strip_in_cap, up_total);
        if (rcb != STR_ER_OK) return rcb;
    }
    size_t up_at = 0;
    const int ty_all = (h + TILE_H - 1) / TILE_H;
    for (int i = 0; i < n_strips; ++i) {
        View &v = view[(size_t)i];
        std::vector<uint8_t> head(head_bytes);
        if (blob_kind == STR_ER_MEM_HOST) {
            std::memcpy(head.data(), blobs[i], head_bytes);
            HIP_TRY(c, hipMemcpyAsync(c->d_strip_in + up_at, blobs[i], (size_t)blob_bytes[i], hipMemcpyHostToDevice, c->stream));
            v.d = c->d_strip_in + up_at;
            up_at += align_up((size_t)blob_bytes[i], 256);
        } else {
            HIP_TRY(c, hipMemcpy(head.data(), blobs[i], head_bytes, hipMemcpyDeviceToHost));
            v.d = static_cast<const uint8_t *>(blobs[i]);
        }
        std::memcpy(&v.hd, head.data(), sizeof(StripHeader));
        const StripHeader &hd = v.hd;
        if (hd.magic != STRIP_MAGIC || hd.version != STRIP_VERSION || hd.w != (uint32_t)w || hd.h != (uint32_t)h || hd.n_strips != (uint32_t)n_strips ||
            hd.strip != (uint32_t)i || hd.n_planes != npl || hd.thresh_step != (uint32_t)c->prm.thresh_step || hd.channel_mask != c->prm.channel_mask)
            return fail(c, STR_ER_EFORMAT, "strip blob does not belong to this frame / context (strip " + std::to_string(i) + ")");
        // the rows the strip claims are the rows this cut gives it
        const int s0 = (int)((int64_t)i * ty_all / n_strips), s1 = (int)((int64_t)(i + 1) * ty_all / n_strips);
        const int r0 = s0 * TILE_H, r1 = std::min(h, s1 * TILE_H), rows = std::max(0, r1 - r0);
        if (hd.rows != (uint32_t)rows || hd.row0 != (uint32_t)std::max(0, r0 - (r0 > 0 ? TILE_H : 0)))
            return fail(c, STR_ER_EFORMAT, "strip blob: rows do not match the cut of the frame (strip " + std::to_string(i) + ")");
        v.sp.resize(npl);
        std::memcpy(v.sp.data(), head.data() + sizeof(StripHeader), npl * sizeof(StripPlane));
        for (size_t k = 0; k < npl; ++k) {
            const StripPlane &p = v.sp[k];
            const bool top = rows > 0 && r0 > 0, bot = rows > 0 && r1 < h;
            if (p.ch != (uint32_t)c->chans[k] || p.n_nodes >= (1u << 24) || (p.start_node != NONE && p.start_node >= p.n_nodes) ||
                (p.has_top != 0) != top || (p.has_bot != 0) != bot || (rows == 0 && p.n_nodes != 0))
                return fail(c, STR_ER_EFORMAT, "strip blob: inconsistent plane header (strip " + std::to_string(i) + ", plane " + std::to_string(k) + ")");
        }
        v.L = strip_layout(v.sp, (uint32_t)w);
        if ((size_t)blob_bytes[i] != v.L.total) return fail(c, STR_ER_EFORMAT, "strip blob: size does not match its headers (strip " + std::to_string(i) + ")");
    }
    int pstride = 0; size_t psize = 0;
    c->n_ev = 0; c->profile.clear(); rec(c, "begin", nullptr, true);
    int rc = frame_to_planes(c, bgr, w, h, stride, mem_kind, pstride, psize);
    if (rc != STR_ER_OK) return rc;
    rec(c, "channels");
    Batch b;
    for (size_t k : sel) {
        const int ch = c->chans[k];
        add_plane(b, c->d_pix + (size_t)(ch % 3) * psize, w, h, pstride, ch >= 3, 0, ch, 0);
        b.planes.back().color_pitch = (uint32_t)psize;
    }
    b.planes_per_image = sel.size() == npl ? (int)npl : 0;       // (er_track / calc_color need all channels of the frame)
    // the records of all strips of a plane must fit the plane's share
    const size_t ns = sel.size();
    std::vector<std::vector<uint32_t>> base(ns, std::vector<uint32_t>((size_t)n_strips + 1, 0));
    double need = 0;
    for (size_t j = 0; j < ns; ++j) {
        for (int i = 0; i < n_strips; ++i) base[j][(size_t)i + 1] = base[j][(size_t)i] + view[(size_t)i].sp[sel[j]].n_nodes;
        need = std::max(need, (double)base[j][(size_t)n_strips] / (double)((size_t)b.planes[j].tiles_x * b.planes[j].tiles_y * TILE_PX));
        if (base[j][(size_t)n_strips] >= (1u << 24)) return fail(c, STR_ER_ECAPACITY, "more than 2^24 node records in one plane");
    }
    if (need > c->node_share) c->node_share = std::min(1.0, need * 1.05);
    HIP_TRY(c, hipMemsetAsync(c->d_strip_flag, 0, sizeof(uint32_t), c->stream));
    const ImportHook hook = [&](const Batch &bb, const BatchDev &bd) -> int {
        hipStream_t s = c->stream;
        for (size_t j = 0; j < ns; ++j) {
            const size_t k = sel[j];
            const PlaneDesc &pd = bb.planes[j];
            if (base[j][(size_t)n_strips] > pd.node_cap) return fail(c, STR_ER_ECAPACITY, "strip records exceed the plane's share (internal error)");
            PlaneCtr pc{};
            pc.n_nodes = base[j][(size_t)n_strips];
            pc.start_node = NONE;
            for (int i = 0; i < n_strips; ++i) {
                const View &v = view[(size_t)i];
                const StripPlane &p = v.sp[k];
                pc.n_walls += p.n_walls;
                if (p.start_node != NONE) pc.start_node = p.start_node + base[j][(size_t)i];
                if (!p.n_nodes) continue;
                NodeRec *dst = bd.na.rec + pd.node_base + base[j][(size_t)i];
                HIP_TRY(c, hipMemcpyAsync(dst, v.d + v.L.rec[k], (size_t)p.n_nodes * sizeof(NodeRec), hipMemcpyDeviceToDevice, s));
                // (ids, keys and rows from strip-local to plane-wide; a parent id outside the strip's records raises the flag)
                launch_rebase_records(s, dst, bd.na.aux + pd.node_base + base[j][(size_t)i], p.n_nodes, base[j][(size_t)i], v.hd.row0 * (uint32_t)w, v.hd.row0, (uint32_t)w, (uint32_t)h,
                                      c->d_strip_flag);
            }
            c->h_ctr[j] = pc;
            HIP_TRY(c, hipMemcpyAsync(c->d_ctr + j, c->h_ctr + j, sizeof(PlaneCtr), hipMemcpyHostToDevice, s));
            // (before anything walks the parent chains: levels consistent, no cycles -- a damaged blob must end in EFORMAT, not in a hang)
            launch_check_forest(s, bd.na.rec + pd.node_base, pc.n_nodes, c->d_strip_flag);
            // pixel pairs across the cuts (strips without rows have no borders: the cut is between the nearest strips that have)
            for (int lo = 0; lo + 1 < n_strips; ++lo) {
                if (!view[(size_t)lo].sp[k].has_bot) continue;
                int hi = lo + 1;
                while (hi < n_strips && !view[(size_t)hi].sp[k].has_top) ++hi;
                if (hi >= n_strips) continue;
                launch_connect_cut(s, bd.na.rec + pd.node_base, reinterpret_cast<const uint32_t *>(view[(size_t)lo].d + view[(size_t)lo].L.bot[k]),
                                   reinterpret_cast<const uint32_t *>(view[(size_t)hi].d + view[(size_t)hi].L.top[k]), (uint32_t)w, base[j][(size_t)lo],
                                   view[(size_t)lo].sp[k].n_nodes, base[j][(size_t)hi], view[(size_t)hi].sp[k].n_nodes, c->d_strip_flag);
            }
        }
        HIP_TRY(c, hipGetLastError());
        uint32_t flag = 0;
        HIP_TRY(c, hipMemcpyAsync(&flag, c->d_strip_flag, sizeof(flag), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, wait_stream(c, s));       // h_ctr is about to be reused for the counters coming back
        if (flag) return fail(c, STR_ER_EFORMAT, "strip blob: damaged node records (an id outside its strip's records, a box or key outside the plane, inconsistent levels or a cycle of parents)");
        return STR_ER_OK;
    };
    return run_batch(c, b, stages, out, t0, true, &hook);
} ABI_GUARD(c)

int str_er_strip_merge(str_er_ctx *c, const uint8_t *bgr, int32_t w, int32_t h, int64_t stride, int mem_kind, const void *const *blobs,
                       const int64_t *blob_bytes, int32_t n_strips, uint32_t stages, str_er_result **out)
try {
    return str_er_strip_merge_ex(c, bgr, w, h, stride, mem_kind, blobs, blob_bytes, STR_ER_MEM_HOST, n_strips, nullptr, stages, out);
} ABI_GUARD(c)

} // extern "C"
