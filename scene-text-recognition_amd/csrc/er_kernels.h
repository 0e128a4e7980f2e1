// er_kernels.h -- host-callable launchers of the gfx950 kernels in er_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include "er_types.h"

namespace str_er {

struct BatchDev {
    const PlaneDesc *planes;   // device array [n_planes]
    PlaneCtr        *ctr;      // device array [n_planes]
    int32_t          n_planes;
    uint32_t         n_tiles;  // batch-wide
    uint32_t         n_pairs;  // batch-wide seam pixel pairs
    uint32_t         n_node_blocks; // workgroups of the per-record kernels, batch-wide
    const uint16_t  *nb_plane;   // plane of every such workgroup (a plane has PlaneDesc::nb_count of them, from nb_base on)
    NodeArrays       na;
    KeptArrays       ka;
    const uint16_t  *tile_plane;        // plane of every tile
    const uint16_t  *seam_block_plane;  // plane of every k_seam workgroup
    const uint32_t  *seam_block_first;  // its first pair inside that plane
    uint32_t         n_seam_blocks;
    uint32_t        *tile_nbase; // plane-local id of every tile's first node record (NONE: the plane ran out of records)
    uint16_t        *tile_nrec;  // ... and how many records the tile has
    uint8_t         *group_done; // per group of tiles: k_group_merge has joined its inner seams
    uint32_t        *undone_list, *undone_count;    // the groups k_group_merge left alone (too many records): their inner seams are joined by k_seam_undone
    const uint16_t  *group_plane; // plane of every group
    uint32_t         n_groups;   // batch-wide
    int32_t          group_x, group_y;   // tiles per group (0: no grouping in this batch)
    uint16_t        *seam;     // node of every tile-border pixel: index inside its tile's records (0xFFFF: wall)
    uint32_t        *pool;     // kept slots chosen by NMS, ascending key
    uint32_t        *pool_tmp;
    CandRec         *cands;    // packed, ordered by (plane, key)
    uint32_t        *total_cands;
    uint16_t        *cand_plane; // plane of every packed candidate
    uint32_t        *watch;      // [n_planes x NMS_WATCH_CAP] key pixels of the children that may compete for a parent
    uint32_t        *wstamp;     // ... and, after the flood order walk, their stamps (order of first access; 0xFFFFFFFF: later than all stamped ones)
    uint32_t        *wparent;    // ... and the kept slot of the parent each one may compete for (the walk stops once every parent's order is decided)
};

// compute_channels (src/ER.cpp:114-128): interleaved BGR -> Y, Cr, Cb planes.
void launch_bgr_to_ycrcb(hipStream_t s, const uint8_t *bgr, int w, int h, int64_t stride,
                         int64_t frame_pitch, int n_frames, uint8_t *y, uint8_t *cr, uint8_t *cb,
                         int dstride, int64_t dst_frame_pitch);
// NV12 (luma plane, then interleaved Cb/Cr at half resolution, same row stride) -> the same three planes; chroma replicated 2 x 2
void launch_nv12_to_ycrcb(hipStream_t s, const uint8_t *nv12, int w, int h, int64_t stride, int64_t frame_pitch, int n_frames, uint8_t *y, uint8_t *cr,
                          uint8_t *cb, int dstride, int64_t dst_frame_pitch);
// 255 - x for the single-stage compute_channels entry point.
void launch_invert(hipStream_t s, const uint8_t *src, uint8_t *dst, size_t n);
// cv::resize INTER_LINEAR 8UC1 semantics; z planes with the given pitches.
void launch_resize(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, int64_t splane_pitch,
                   int64_t sframe_pitch, uint8_t *dst, int dw, int dh, int dstride, int64_t dplane_pitch,
                   int64_t dframe_pitch, int planes_per_frame, int n_frames);

// sparse: the small-LDS / high-occupancy size of the kernel (text-like frames); dense: the big one (noise-like frames)
// list: the tiles to work on (batch-wide tile numbers, device), nullptr = all tiles of the batch
void launch_tile_tree(hipStream_t s, const BatchDev &b, const DetectParams &p, bool sparse, const uint32_t *list = nullptr, uint32_t n = 0);
// the tiles k_tile_tree2 handed back (list / count on the device): `grid` workgroups walk the list
void launch_tile_tree_fb(hipStream_t s, const BatchDev &b, const DetectParams &p, bool sparse, const uint32_t *list, const uint32_t *count, uint32_t grid);
// k_tile_tree2 (er_tile_tree2.inl): the level-by-level / bit-mask form for planes with few levels per tile; one wave per pair of horizontally
// adjacent tiles (entry: first tile | 1 << 31 if the tile to its right takes part); tiles it does not take end up in fb_list
void launch_tile_tree2(hipStream_t s, const BatchDev &b, const DetectParams &p, const uint32_t *pairs, uint32_t n_pairs, uint32_t *fb_list, uint32_t *fb_count);
// the tiles of every group of BatchDev::group_x x group_y tiles joined in LDS, in place (variant: LDS capacity / lanes, see er_kernels.hip)
void launch_group_merge(hipStream_t s, const BatchDev &b, int variant, const uint32_t *glist = nullptr, uint32_t n = 0);
void launch_seam(hipStream_t s, const BatchDev &b, bool xcd_affine);
void launch_resolve(hipStream_t s, const BatchDev &b);
// strips of a plane extracted elsewhere: make a strip's record ids plane-wide; join pixel pairs (plane-local ids) across a cut
void launch_rebase_records(hipStream_t s, NodeRec *rec, uint32_t *aux, uint32_t n, uint32_t delta, uint32_t key_add, uint32_t y_add, uint32_t w, uint32_t h,
                           uint32_t *bad);
void launch_check_forest(hipStream_t s, NodeRec *plane_rec, uint32_t n, uint32_t *bad);
void launch_connect_cut(hipStream_t s, NodeRec *plane_rec, const uint32_t *bot, const uint32_t *top, uint32_t w, uint32_t base_lo, uint32_t n_lo, uint32_t base_hi,
                        uint32_t n_hi, uint32_t *bad);
void launch_strip_border_ids(hipStream_t s, const uint16_t *seam_row, const uint32_t *tile_nbase_row, int w, uint32_t *out);
// er_merge's accumulation (src/ER.cpp:153-165) for the whole batch in ONE launch: a node pushes its totals to its parent when its
// last open child has pushed (dependency counters, agent-scope release/acquire)
void launch_reduce(hipStream_t s, const BatchDev &b);
void launch_root(hipStream_t s, const BatchDev &b, const DetectParams &p);
void launch_select(hipStream_t s, const BatchDev &b, const DetectParams &p);
void launch_kept(hipStream_t s, const BatchDev &b, const DetectParams &p);
// pass 0: every plane; sibling ties by key (exact mode: provisional, planes with ties are counted in n_amb and get a watch list).
// use_index_order: exact mode on an uploaded tree -- the table order is the child-list order.
void launch_nms(hipStream_t s, const BatchDev &b, const DetectParams &p, bool use_index_order = false);
void launch_nms_alt(hipStream_t s, const BatchDev &b, const DetectParams &p, uint32_t *alt_list /* NMS_ALT_CAP words, device */);
// planes with ties left (PlaneCtr::n_rel) -> host-addressable memory: pixels, watch keys, watch parents per slot (see er_kernels.hip)
void launch_export_listed_planes(hipStream_t s, const BatchDev &b, const ReplayItem *items, int n_items, uint8_t *host_buf /* page-locked, device-addressable */);
void launch_export_tie_planes(hipStream_t s, const BatchDev &b, uint8_t *host_buf, size_t slot_bytes, int n_slots, uint32_t *slot_plane_dev /* n_slots words, device */,
                              uint32_t *count, uint32_t *slot_plane);
// exact mode, planes with sibling ties: replay the reference's flood (src/ER.cpp:240-374) to stamp every pixel with the order in
// which it becomes accessible, then NMS again with the ties decided by those stamps.  scratch: see ReplayItem.
size_t replay_scratch_bytes(int w, int h);
void launch_flood_order(hipStream_t s, const BatchDev &b, const DetectParams &p, const ReplayItem *items, int n_items, uint8_t *scratch);
void launch_nms_resolve(hipStream_t s, const BatchDev &b, const DetectParams &p, const ReplayItem *items, int n_items, const uint8_t *scratch);
void launch_cand_prefix(hipStream_t s, const BatchDev &b);
// classify (src/ER.cpp:507-528) over the packed pool of the batch.
void launch_classify(hipStream_t s, const BatchDev &b, const DetectParams &p, CascadeDev strong,
                     CascadeDev weak, int run_cascades, const uint32_t *list = nullptr, const uint32_t *n_list = nullptr, bool few = false);      // (few: a call of a frame or two -- 16 candidates a workgroup instead of 64)
// after the NMS tie pass: new candidate offsets (b.cands / b.cand_plane = the second set of buffers), the records of unchanged planes
// moved over from `from`, the candidates of the changed planes listed in redo[0 .. *n_redo) for launch_classify
// (calls of a frame or two) the counter block and the first candidate records to device-addressable host memory, one launch
void launch_results_to_host(hipStream_t s, const void *ctr_block, void *to_ctr, size_t ctr_bytes, const CandRec *cands, CandRec *to_cands, uint32_t cap_cands,
                            const uint32_t *total_cands);
void launch_cand_reprefix(hipStream_t s, const BatchDev &b, const CandRec *from, uint32_t *redo, uint32_t *n_redo);

// Stand-alone classify chain on explicit boxes of one device plane (single-stage API).
void launch_lbp_boxes(hipStream_t s, const uint8_t *plane, int w, int h, int stride, const int32_t *boxes,
                      int n, double *hist /*n*1024 or null*/, uint8_t *tiles /*n*676 or null*/, uint8_t *codes /*n*576 or null*/, uint8_t *cls,
                      double *s_strong, double *s_weak, CascadeDev strong, CascadeDev weak, int run_cascades);

// CascadeBoost::predict on explicit feature vectors (n x 1024 doubles).
void launch_cascade_fv(hipStream_t s, const double *fv, int n, double *out, CascadeDev c);

} // namespace str_er
