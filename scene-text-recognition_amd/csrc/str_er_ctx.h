// str_er_ctx.h -- INTERNAL: the context / result structs and the small host helpers shared by the translation units of the C ABI
// (str_er_api.cpp: contexts, batches, the detect entry points, results; api_models.cpp: cascade / libsvm models and the OCR entry points;
//  api_strips.cpp: one plane in strips over several GPUs; api_stages.cpp: the single-stage entry points).  Not installed, not part of the ABI.
// The helpers in the unnamed namespace are small and private to each translation unit; what one unit defines for the others is declared in str_er_host.
#pragma once
#include "../../include/str_er.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "er_kernels.h"
#include "ocr_kernels.h"
#include "track_kernels.h"
#include "er_group.h"
#include "flood_order.h"
#include <functional>
#include <thread>

using namespace str_er;

extern thread_local std::string g_create_error;      // the error text of a failed str_er_create (defined in str_er_api.cpp)

namespace str_er_host {

constexpr int TIE_SLOTS = 16;      // at most so many planes per batch the device hands to the host for the flood order walk without a round trip
                                   // (a context has as many slots as fit 64 MB of page-locked memory, at least 4: str_er_ctx::n_tie_slots)

struct HostCascade {
    bool loaded = false;
    bool real = true;
    std::vector<int32_t> stage_n, stage_thresh;
    std::vector<uint16_t> dim;
    std::vector<double> thr, dir, vp, vn;
    void *d_blob = nullptr;
    CascadeDev dev{};
};

static_assert(sizeof(str_er_node) == 24, "node layout");
static_assert(sizeof(CandRec) == sizeof(str_er_cand), "cand layout");

struct PlaneGeom { int w, h, stride; size_t off; }; // physical planes of one pyramid level

} // namespace str_er_host
using namespace str_er_host;

struct str_er_result {
    std::vector<str_er_plane_info> planes;
    std::vector<str_er_cand> cands;
    std::vector<uint32_t> cand_off;          // n_planes + 1
    std::vector<std::vector<str_er_node>> nodes;
    bool have_nodes = false;
    std::vector<int32_t> ocr_label;
    std::vector<double> ocr_prob;
    bool have_ocr = false;
    std::vector<str_er_track> tracks;
    bool have_tracks = false;
    std::vector<str_er_text> texts;
    std::vector<int32_t> text_ers;
    std::vector<str_er_gbound> gbounds;
    std::vector<int32_t> group_all;
    std::vector<int32_t> line_label;
    std::vector<double> line_prob;
    std::vector<uint8_t> line_kept, text_alive;
    bool have_line_ocr = false;
    bool have_texts = false;
    double times[7] = {0, 0, 0, 0, 0, 0, 0};
};

struct str_er_ctx {
    str_er_params prm{};
    std::string err;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;          // the opposite-rule NMS pass runs here, beside classify
    hipStream_t prio = nullptr;          // high priority: the few small operations that settle an NMS tie (they would queue behind other contexts' big kernels)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool own_stream = false;
    int ppf = 0;                     // logical planes per frame
    std::vector<int> chans;          // channel indices selected by the mask
    size_t slots = 0;                // node slots (== plane pixels) the workspace can hold
    int max_planes = 0;
    static constexpr size_t NB_PLANE_SHARE = 64;     // entries of the workgroup -> plane table (d_nb_plane) per plane of the capacity

    int kept_cap = 0, pool_cap = 0;   // per plane: the most a plane may get
    bool auto_caps = true;            // (neither was given: every plane gets a share of the tables by its pixel count)
    double kept_share = 1.0 / 64, pool_share = 1.0 / 256;   // ... kept nodes / pooled ERs per padded pixel; grown -- and the batch repeated -- on overflow
    size_t kept_total = 0, pool_total = 0;      // entries of the kept-node / pool arrays
    int64_t table_bytes = 0;
    double min_ocr_prob = 0.15;       // MIN_OCR_PROBABILITY (inc/utils.h), the ERFilter constructor's last argument
    bool   tile_sparse = true;        // which size of k_tile_tree the next batch uses (er_kernels.hip: FOLD_CAP_SPARSE / _DENSE)
    uint64_t last_tree_records = 0, last_tree_pairs = 0, last_tree_tiles = 0;     // of the last batch (str_er_last_tree_stats)
    bool   spin_wait = false;          // STR_ER_SPIN_WAIT=1: always hipStreamSynchronize (busy-waits on a core), see wait_stream
    int    wait_spin_us = 300;         // how long wait_stream polls before it sleeps between polls (run_batch: 2 ms for a call of a frame or two)
    bool   dbg_tile_only = false, dbg_stats = false;   // developer aids (STR_ER_DEBUG_TILE_ONLY / _STATS), read once at create
    int    tile_mode = 0;             // 0 auto (from the node density of the previous batch), 1 sparse, 2 dense (STR_ER_TILE_KERNEL)
    int64_t ws_bytes = 0;

    // device workspace
    uint8_t *d_in = nullptr;  size_t in_bytes = 0;    // staging for host inputs
    uint8_t *d_pix = nullptr; size_t pix_bytes = 0;   // physical planes (Y,Cr,Cb per level)
    PlaneDesc *d_planes = nullptr;
    PlaneCtr *d_ctr = nullptr;
    uint8_t *d_zero = nullptr, *h_zero = nullptr; size_t zero_gd_off = 0;      // the block a batch zeroes: d_total | d_ctr | d_group_done (and its page-locked mirror: h_total | h_ctr)
    size_t zero_clean_bytes = 0;      // so many bytes of d_zero are zero already: filled behind the last call's results (run_batch), off the next call's critical path
    int planes_on_device = 0;         // plane descriptors in d_planes = the first so many of h_planes (0: none)
    NodeArrays na{};
    KeptArrays ka{};
    uint16_t *d_seam = nullptr; size_t seam_slots = 0;
    size_t node_slots = 0;            // node records allocated (NodeArrays::rec / aux)
    uint32_t node_blocks_cap = 0;     // workgroups per plane in k_resolve / k_reduce; 0 = by the frames' content (STR_ER_NODE_BLOCKS sets it)
    uint32_t node_blocks = 12;        // workgroups per plane of the per-record kernels: from the record counts of the previous batch
    double node_share = 0.06;         // records per padded plane pixel (S-text needs 0.006, S-noise 0.09); grown -- and the batch repeated -- when a plane runs out
    uint16_t *d_tile_plane = nullptr, *d_sb_plane = nullptr; uint32_t *d_sb_first = nullptr; size_t sb_slots = 0;
    std::vector<uint16_t> h_tile_plane, h_sb_plane; std::vector<uint32_t> h_sb_first;
    std::vector<uint32_t> layout_key;   // (w,h,...) of the batch whose tables are on the device
    uint32_t *d_tile_nbase = nullptr; size_t tile_slots = 0;
    // the batch's tiles split between the two tile kernels (launch_tile_trees): the planes k_tile_tree2 takes (chroma: few levels per tile) as
    // pairs of tiles, the others as a list for k_tile_tree, and the tiles k_tile_tree2 hands back (their count: d_total[1])
    int       t2_mode = 1;                            // STR_ER_TILE2: 0 off, 1 the chroma planes (ch % 3 != 0), 2 every plane
    int       t2_backoff = 0;                         // batches for which the chroma planes stay with k_tile_tree (the last batch handed too many tiles back)
    uint32_t *d_t1_list = nullptr, *d_t2_pairs = nullptr, *d_fb_list = nullptr;
    std::vector<uint32_t> h_t1_list, h_t2_pairs, t2_key;
    uint32_t  n_t2_tiles = 0;                         // tiles of the planes k_tile_tree2 takes in the current lists
    uint64_t  t2_tiles_total = 0, t2_fb_total = 0;    // statistics (str_er_tile2_stats)
    uint32_t *d_pool = nullptr, *d_pool_tmp = nullptr;
    CandRec *d_cands = nullptr, *d_cands2 = nullptr;      // (second set: the layout after an NMS tie pass changed pools, then swapped)
    uint32_t *d_redo = nullptr;                          // candidates to classify again + their count (last word)
    TrackRec *d_track = nullptr; uint32_t *d_track_list = nullptr, *d_ranges = nullptr;   // STR_ER_STAGE_TRACK
    uint32_t *d_group = nullptr, *d_group_pairs = nullptr; size_t group_words = 0, group_pair_cap = 0;   // STR_ER_STAGE_GROUP, grown on demand
    uint32_t *d_total = nullptr;
    uint32_t *d_wparent = nullptr;
    // tie planes exported by the device itself (k_export_tie_planes): TIE_SLOTS x tie_slot_bytes of page-locked, device-addressable memory,
    // then the slot -> plane table and the slot counter
    uint8_t *h_tie = nullptr; size_t tie_slot_bytes = 0; int n_tie_slots = 0; uint32_t *h_tie_plane = nullptr, *h_tie_count = nullptr;
    uint8_t *h_replay = nullptr; size_t h_replay_bytes = 0;   // page-locked: the planes (and watch lists) the flood order walk reads
    uint32_t *d_watch = nullptr, *d_wstamp = nullptr; // NMS: watched key pixels per plane (k_nms -> flood order walk) and their stamps (-> k_nms)
    ReplayItem *d_replay_items = nullptr;
    uint32_t *d_alt_list = nullptr;                   // planes of the opposite-rule NMS pass (k_alt_list)
    uint32_t *d_tie_slot_plane = nullptr;             // plane of every tie slot of the batch (k_tie_slots -> k_export_tie_planes)
    uint8_t *d_replay = nullptr; size_t replay_bytes = 0;   // flood-replay scratch, allocated the first time a plane has sibling ties
    uint32_t last_total = 0; bool last_valid = false;   // candidates of the last detect call, still in d_cands (str_er_gather_last)
    uint64_t n_replayed = 0;                          // planes whose NMS ties were decided by a flood replay (statistics)
    double   walk_ms_total = 0;                       // host time those walks took, summed over planes (statistics)
    uint64_t n_batches = 0;
    bool replay_on_gpu = false;                       // STR_ER_REPLAY=gpu: walk the flood with k_flood_order instead of a host core
    uint16_t *d_cand_plane = nullptr, *d_cand_plane2 = nullptr;
    void *d_scratch = nullptr; size_t scratch_bytes = 0;
    uint8_t *d_strip_out = nullptr, *d_strip_in = nullptr; size_t strip_out_cap = 0, strip_in_cap = 0;   // strip blobs: made here / uploaded for a merge
    uint32_t *d_strip_flag = nullptr;                 // a strip blob named a node outside its records
    uint16_t *d_nb_plane = nullptr; std::vector<uint16_t> h_nb_plane; uint32_t n_node_blocks = 0;      // plane of every workgroup of the per-record kernels
    uint16_t *d_tile_nrec = nullptr;                  // records per tile (k_tile_tree -> k_group_merge)
    uint16_t *d_group_plane = nullptr; std::vector<uint16_t> h_group_plane;      // plane of every group of tiles
    uint32_t *d_group_list = nullptr; std::vector<uint32_t> h_group_list; uint32_t n_groups_small = 0;      // the groups by class of plane: first those of the chroma planes (few records a tile: k_group_merge with a small table), then the others
    uint32_t *d_undone = nullptr;                     // the groups k_group_merge left alone, listed on the device (their number: d_total[2]) for k_seam_undone
    uint8_t  *d_group_done = nullptr;                 // per group of tiles: joined in LDS (k_group_merge -> k_seam)
    int       dbg_group[3] = {0, 0, -1};              // developer knobs STR_ER_GROUP_X / _Y / _KERNEL
    int       group_mode = -1;                        // STR_ER_GROUPS: -1 automatic (4 x 4 tiles with the small tile kernel, 2 x 5 with the big one), 0 off
    std::vector<void *> allocs;

    // pinned host mirrors
    PlaneDesc *h_planes = nullptr;
    PlaneCtr *h_ctr = nullptr;
    uint32_t *h_total = nullptr;
    CandRec  *h_cands_spec = nullptr;                 // small calls: the first SPEC_CANDS candidate records come back WITH the counters (run_batch)
    // STR_ER_STAGE_OCR behind classify, before the host has read a counter (run_batch): the launches are sized for a little more than the last batch's number of
    // strong / weak ERs and work on the device's own count; the results come back with the counters in one page-locked block
    // (count | list | labels | probabilities).  A batch with more ERs than guessed, or whose candidates an NMS tie pass re-made, is scored again the slow way.
    bool      ocr_spec = true;                        // STR_ER_OCR_SPEC=0 turns it off (developer switch)
    size_t    ocr_last_n = 0;                         // strong + weak ERs of the last batch scored
    uint8_t  *h_ocr = nullptr; size_t h_ocr_cap = 0;  // page-locked results for up to so many ERs
    uint64_t  n_ocr_spec = 0, n_ocr_redo = 0;         // statistics: batches scored behind classify / scored again

    HostCascade casc[2];
    bool svm_loaded = false;
    SvmDev svm{};
    void *d_svm_blob = nullptr;
    static constexpr int MAX_EV = 32;
    hipEvent_t ev[MAX_EV]{};
    int n_ev = 0;
    bool profiling = false;
    std::vector<std::pair<const char *, double>> profile;
};

namespace {

int fail(str_er_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

// Waiting for a stream.  hipStreamSynchronize busy-waits (so does hipEventSynchronize on a hipEventBlockingSync event, measured): with a batch in
// flight on each of six contexts that is six host cores spinning -- and the GPU boxes grant a process 16 (cgroup quota), which the flood order walks
// of the NMS ties need (round 4: the S-ties bench leg, 86 ms of walks per batch on 16 pool threads + 6 spinning waiters = throttled).  So: poll for
// ~300 us, then sleep between polls.  A latency call (<= SPEC_PLANES planes: a frame or two, under a millisecond of GPU work, one wait at its end) polls
// for 2 ms instead: the 100 us naps added 0.14 ms to most one-frame calls (0.80 ms when the wait happened to end inside the polling, 0.94 otherwise).
// A call of a frame or two (<= SPEC_PLANES planes) is a latency call: its candidate records -- a thousand per 1920 x 1080 frame -- are copied to page-locked
// memory right behind the counters, before the host knows how many there are; if they all fit (and no NMS tie pass re-made them) the second trip to the
// device -- counters, THEN as many records as they say, into pageable memory -- is saved: about 0.1 of a 0.9 ms call.
constexpr uint32_t SPEC_CANDS = 8192;
constexpr int      SPEC_PLANES = 96;

static hipError_t wait_stream(str_er_ctx *c, hipStream_t s)
{
    if (!c || c->spin_wait) return hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(c->wait_spin_us)) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail((ctx), (e_ == hipErrorOutOfMemory) ? STR_ER_ENOMEM : STR_ER_EHIP,          \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

template <typename T> int dev_alloc(str_er_ctx *c, T *&p, size_t n)
{
    void *v = nullptr;
    const size_t bytes = std::max<size_t>(n * sizeof(T), 256);
    hipError_t e = hipMalloc(&v, bytes);
    if (e != hipSuccess) return fail(c, STR_ER_ENOMEM, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    c->allocs.push_back(v);
    c->ws_bytes += (int64_t)bytes;
    p = static_cast<T *>(v);
    return STR_ER_OK;
}

// The node records (32 B + 2 x 4 B per record) are the one part of the workspace whose need depends on the frames' content: they
// are allocated for `node_share` records per pixel and re-allocated larger when a batch overflows them (run_batch).
int alloc_node_records(str_er_ctx *c, size_t n)
{
    if (c->na.rec) { (void)hipFree(c->na.rec); c->ws_bytes -= (int64_t)(c->node_slots * sizeof(NodeRec)); c->na.rec = nullptr; }
    if (c->na.aux) { (void)hipFree(c->na.aux); c->ws_bytes -= (int64_t)(c->node_slots * 8); c->na.aux = nullptr; c->na.arr = nullptr; }
    c->node_slots = 0;
    if (hipMalloc(reinterpret_cast<void **>(&c->na.rec), n * sizeof(NodeRec)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->na.aux), n * 8) != hipSuccess)
        return fail(c, STR_ER_ENOMEM, "hipMalloc (node records, " + std::to_string(n * 40) + " bytes)");
    c->na.arr = c->na.aux + n;
    c->node_slots = n;
    c->ws_bytes += (int64_t)(n * 40);
    return STR_ER_OK;
}

int ensure_scratch(str_er_ctx *c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return STR_ER_OK;
    // (a quarter more than asked for, and at least 1.5 x what there was: the need follows the batch's content -- how many ERs are scored, how many lines have
    // members -- and hipFree / hipMalloc wait for the whole device, every other context's kernels included)
    bytes = std::max(bytes + bytes / 4, c->scratch_bytes + c->scratch_bytes / 2);
    if (c->d_scratch) { (void)hipFree(c->d_scratch); c->d_scratch = nullptr; c->scratch_bytes = 0; }
    hipError_t e = hipMalloc(&c->d_scratch, bytes);
    if (e != hipSuccess) return fail(c, STR_ER_ENOMEM, std::string("hipMalloc scratch: ") + hipGetErrorString(e));
    c->scratch_bytes = bytes;
    return STR_ER_OK;
}

void pyr_dims(int w0, int h0, int level, int &w, int &h)
{
    const double s = std::pow(2.0, -0.5 * level);
    w = std::max(1, (int)std::floor(w0 * s + 0.5));
    h = std::max(1, (int)std::floor(h0 * s + 0.5));
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

DetectParams make_dp(const str_er_ctx *c)
{
    DetectParams d{};
    d.thresh_step = c->prm.thresh_step; d.min_area = c->prm.min_area; d.max_area = c->prm.max_area;
    d.stability_t = c->prm.stability_t; d.overlap_coef = c->prm.overlap_coef;
    d.hi = 255 / c->prm.thresh_step + 1;
    d.qscale = (float)(1.0 / (double)c->prm.thresh_step);
    d.kept_cap = c->kept_cap; d.pool_cap = c->pool_cap; d.sibling_order = c->prm.sibling_order;
    return d;
}

} // namespace

namespace str_er_host {
// ---- batch layout -------------------------------------------------------------------------------
struct Batch {
    std::vector<PlaneDesc> planes;
    uint32_t n_tiles = 0, n_pairs = 0;
    size_t slots = 0, seam = 0, nodes = 0;      // padded pixels, seam entries, node records
    size_t kept = 0, pool = 0;                  // entries of the kept-node / pool arrays handed to the planes
    uint32_t kept_floor = 0, pool_floor = 0;    // str_er_nms_tree: the plane's tables must hold the imported tree
    int planes_per_image = 0;       // BGR frames: planes of one (frame, pyramid level), consecutive in `planes`; 0 = no colour image
    uint32_t n_groups = 0; int group_x = 0, group_y = 0;       // k_group_merge: groups of group_x x group_y tiles (0: none); assign_groups()
};

inline void add_plane(Batch &b, const uint8_t *pix, int w, int h, int stride, int invert, uint32_t frame, int ch, int pyr)
{
    PlaneDesc d{};
    d.pix = pix; d.w = w; d.h = h; d.stride = stride; d.invert = invert ? 0xFF : 0;
    d.tiles_x = (w + TILE_W - 1) / TILE_W; d.tiles_y = (h + TILE_H - 1) / TILE_H;
    d.tile_base = b.n_tiles; b.n_tiles += (uint32_t)d.tiles_x * d.tiles_y;
    d.n_hpairs = (uint32_t)w * (d.tiles_y - 1);
    d.n_pairs = d.n_hpairs + (uint32_t)h * (d.tiles_x - 1);
    d.pair_base = b.n_pairs; b.n_pairs += d.n_pairs;
    b.slots += (size_t)d.tiles_x * d.tiles_y * TILE_PX;      // (node records are laid out by assign_node_records)
    d.seam_base = (uint32_t)b.seam; b.seam += 2 * (size_t)d.n_pairs;
    d.frame = frame; d.ch = (uint8_t)ch; d.pyr = (uint8_t)pyr;
    b.planes.push_back(d);
}

// ---- defined in str_er_api.cpp, used by the other translation units ------------------------------------------------------------------
using ImportHook = std::function<int(const Batch &, const BatchDev &)>;
void assign_node_records(Batch &b, double share);
void assign_tables(Batch &b, const str_er_ctx *c);
int alloc_tables(str_er_ctx *c, size_t KP, size_t PP);
BatchDev make_batchdev(str_er_ctx *c, const Batch &b);
// an event behind what was enqueued so far.  `always`: one of the few a call needs for str_er_result_times (begin, end of extraction, NMS, classify, track);
// the others -- one per kernel group -- are recorded only while str_er_set_profiling is on: an event between two kernels costs ~7 us of stream time
// (1-frame calls: a dozen of them were a tenth of the call)
void rec(str_er_ctx *c, const char *name, hipStream_t on = nullptr, bool always = false);
RotGeom make_rot_geom(int w, int h, double slope);
int group_phase(str_er_ctx *c, const CandRec *d_cands, const TrackRec *d_track, const std::vector<uint32_t> &img, bool inner_sup, str_er_result *r,
                bool presorted = false);
int resolve_sibling_ties(str_er_ctx *c, const Batch &b, const BatchDev &bd, const DetectParams &dp, bool &replayed, bool from_tree = false);
int group_phase_overlap(str_er_ctx *c, const std::vector<uint32_t> &img, bool inner_sup, str_er_result *r);
int upload_layout(str_er_ctx *c, Batch &b);
int run_batch(str_er_ctx *c, const Batch &b_in, uint32_t stages, str_er_result **out, std::chrono::steady_clock::time_point t_start, bool pre_recorded,
              const ImportHook *import_trees = nullptr, int attempt = 0);
int stage_input(str_er_ctx *c, const uint8_t *src, size_t bytes, int mem_kind, const uint8_t **dev);
// ---- defined in api_models.cpp
int parse_cascade(str_er_ctx *c, HostCascade &hc, const char *text, size_t len);
} // namespace str_er_host

// =================================================================================================
// C ABI
// =================================================================================================
// Nothing is thrown across the C ABI: every entry point that takes a context is a function-try-block (the std::vector / std::string work behind
// them -- parsers of untrusted bytes, per-batch tables -- can run out of memory).
static int abi_caught(str_er_ctx *c, int code, const char *what)
{
    if (!c) return code;
    try { c->err = what; } catch (...) { }
    return code;
}
#define ABI_GUARD(ctx)                                                                                   \
    catch (const std::bad_alloc &) { return abi_caught((ctx), STR_ER_ENOMEM, "out of host memory"); }      \
    catch (const std::length_error &) { return abi_caught((ctx), STR_ER_ENOMEM, "out of host memory (container size)"); } \
    catch (...) { return abi_caught((ctx), STR_ER_EHIP, "internal error (exception)"); }
