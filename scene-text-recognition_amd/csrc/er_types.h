// er_types.h -- structures shared by the host API (str_er_api.cpp) and the gfx950
// kernels (er_kernels.hip).  Internal; the public boundary is include/str_er.h.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace str_er {

// Tile of the in-LDS component-tree kernel.  64 pixels wide so a tile row is one
// 64-byte scanline segment and the per-node column set fits one 64-bit mask;
// 32 rows (row set = one 32-bit mask, 256 lanes) or, with -DSTR_ER_TILE_H=64, 64 rows
// (64-bit row mask, 512 lanes: a third fewer seam pixels, half as many tiles).
#ifndef STR_ER_TILE_H
#define STR_ER_TILE_H 32
#endif
static_assert(STR_ER_TILE_H == 32 || STR_ER_TILE_H == 64, "tile height is 32 or 64");
constexpr int      TILE_W   = 64;
constexpr int      TILE_H   = STR_ER_TILE_H;
constexpr int      TILE_PX  = TILE_W * TILE_H;   // 2048 / 4096
constexpr int      TILE_PPT = 8;                 // consecutive pixels per lane
constexpr int      TILE_THREADS = TILE_PX / TILE_PPT; // 256 / 512: 4 / 8 wavefronts of 64
constexpr uint32_t NONE     = 0xFFFFFFFFu;
// (512 since round 6, 1024 before: with several batches in flight a workgroup gets onto a compute unit in the holes the tile kernels' workgroups -- 256 lanes,
// 20 KB of LDS, all of a unit's wave slots and LDS between eight of them -- leave behind: 1024 lanes need four holes at once.  256 / 512 / 1024: 13.42 / 13.44 / 13.29 k frames/s)
#ifndef STR_ER_SEAM_BLOCK
#define STR_ER_SEAM_BLOCK 512
#endif
constexpr int      SEAM_BLOCK = STR_ER_SEAM_BLOCK;          // seam pixel pairs (= lanes) per k_seam workgroup; the host lists (plane, first pair) per workgroup

// One logical plane = one (frame, channel, pyramid level): the unit the reference
// loops over at src/ER.cpp:50-60.  Inverted channels (255-x, src/ER.cpp:125-127) share
// the physical plane of their source and set `invert`.
struct PlaneDesc {
    const uint8_t *pix;     // device pointer, top-left pixel of the physical plane
    int32_t  w, h, stride;
    int32_t  invert;        // 0 or 0xFF (xor mask)
    int32_t  tiles_x, tiles_y;
    uint32_t tile_base;     // first tile of this plane in the batch-wide tile numbering
    uint32_t pair_base;     // first seam pixel-pair of this plane (batch-wide numbering)
    uint32_t n_hpairs;      // w * (tiles_y - 1)
    uint32_t n_pairs;       // n_hpairs + h * (tiles_x - 1)
    uint32_t node_base;     // first record of this plane in the node records (capacity node_cap)
    uint32_t seam_base;     // offset in the seam map (u16 units)
    uint32_t kept_base;     // offset in the kept-node arrays (capacity kept_cap)
    uint32_t pool_base;     // offset in the pool arrays (capacity pool_cap)
    uint32_t frame;
    uint8_t  ch, pyr;
    uint8_t  nb_count;      // workgroups of the per-record kernels (k_resolve, k_reduce, k_select) this plane gets
    uint8_t  pad1;
    uint32_t color_pitch;   // BGR frames: bytes between the Y, Cr, Cb planes of this level (pix - (ch % 3) * color_pitch is Y); 0 = no colour image
    uint32_t node_cap;      // node records this plane may use (a share of its pixel count; overflow -> the host grows the share and repeats)
    uint32_t kept_cap;      // entries of the kept-node arrays / of the pool arrays that belong to this plane (by default a share of its pixel
    uint32_t pool_cap;      // count: a 240 x 135 pyramid level does not need the table of a 1920 x 1080 plane)
    uint32_t group_base;    // first group of tiles of this plane (k_group_merge; batch-wide numbering, for the batch's group size)
    uint32_t nb_base;       // ... and the first of them
};
static_assert(sizeof(PlaneDesc) == 96 && offsetof(PlaneDesc, node_cap) == 76, "PlaneDesc layout (host and device)");

// Per-plane device counters, zeroed before every batch.
struct PlaneCtr {
    uint32_t n_nodes;       // node records handed out to the plane's tiles (may exceed node_cap: then overflow bit 3 is set)
    uint32_t n_walls;       // in-image pixels at the sentinel level
    uint32_t start_node;    // node of the flood's start pixel, NONE if none (A.2)
    uint32_t root_node;     // root of the start pixel's tree
    uint32_t n_kept;
    uint32_t n_pool;
    uint32_t n_amb;
    uint32_t overflow;      // bit0 kept table, bit1 pool, bit3 node records
    uint32_t n_strong;
    uint32_t n_weak;
    uint32_t cand_base;     // exclusive prefix of n_pool over planes
    uint32_t n_created;     // live nodes inside the start pixel's tree
    uint32_t root_slot;     // kept slot of the root
    uint32_t max_level;     // highest level among kept nodes (bounds the NMS sweep)
    uint32_t n_watch;       // NMS: children that may compete for their parent (their key pixels are what the flood replay must reach)
    uint32_t n_rel;         // NMS: ties that can change the pool (n_amb counts all ties); > 0 -> the plane's flood is replayed
    uint32_t tie_node;      // NMS: kept slot of a node with a tie and the number of its contenders (meaningful when n_amb == 1)
    uint32_t tie_nc;
    uint32_t pool_changed;  // NMS tie pass: the plane's pool is not the first pass's pool (its candidates have to be classified again)
    uint32_t cand_base_old; // candidate offset of the plane before the tie pass (k_cand_reprefix)
};
static_assert(sizeof(PlaneCtr) == 80, "PlaneCtr is mirrored in pinned host memory");

// One plane whose NMS has to be decided by the reference's flood order (k_flood_order, then k_nms pass 1).
struct ReplayItem {
    uint32_t plane;
    uint32_t pad_;
    uint64_t off;           // byte offset of the plane's scratch (stamp u32[w*h], link u32[w*h], level u8[w*h]) in the replay buffer
};
constexpr int NMS_ALT_CAP = 64;      // planes per batch the opposite-rule NMS pass can take
constexpr int NMS_WATCH_CAP = 256;   // watched key pixels per plane; more -> the replay floods the whole plane

// One tree node that left its tile ("exported"): 32 bytes, written by k_tile_tree with two 16-byte stores, index =
// PlaneDesc::node_base + plane-local id.  Ids are handed out densely per plane (PlaneCtr::n_nodes), a tile at a time.
struct alignas(16) NodeRec {
    uint32_t par;           // NONE (tree root) or level of the parent << 24 | plane-local id of the parent; CAS-ed by k_seam
    uint32_t key;           // bits 0..23 min linear pixel index of the node's own-level pixels; bits 24..31 the node's level
    // (cnt | nod and each corner of the box are 8-byte aligned pairs: k_reduce moves them with ONE 64-bit device-scope operation each --
    // those operations, not bytes, are what the accumulation costs)
    uint32_t cnt;           // pixels: own -> subtree total
    uint32_t nod;           // bits 0..23 nodes: 1 -> subtree total (pruned ones included); bits 24.. NODE_* flags
    uint32_t x0, y0, x1, y1; // bbox: own -> subtree (atomicMin / atomicMax)
};
static_assert(offsetof(NodeRec, cnt) == 8 && offsetof(NodeRec, x0) == 16 && offsetof(NodeRec, x1) == 24, "64-bit pairs");
static_assert(sizeof(NodeRec) == 32, "two dwordx4 stores");
constexpr uint32_t NODE_DEAD = 1u << 24;     // unified into another node of the same level (k_resolve)
constexpr uint32_t NODE_CLOSED = 2u << 24;   // never touches a seam: totals were final in the tile, never pushes
// open nodes: the sides of their TILE their component lies on (after k_group_merge: of their group of tiles)
constexpr uint32_t NODE_SIDE_T = 4u << 24, NODE_SIDE_B = 8u << 24, NODE_SIDE_L = 16u << 24, NODE_SIDE_R = 32u << 24;
constexpr uint32_t NODE_SIDES = NODE_SIDE_T | NODE_SIDE_B | NODE_SIDE_L | NODE_SIDE_R;
constexpr uint32_t NODE_CNT = 0xFFFFFFu;
struct NodeArrays {
    NodeRec  *rec;
    uint32_t *aux;          // per node: its open children that push their totals (counted by k_resolve, constant through k_reduce), then node id -> kept slot (k_select -> k_kept)
    uint32_t *arr;          // per node: how many of those children have pushed so far (zeroed by k_resolve, counted by k_reduce)
};

// Kept-node storage (index = PlaneDesc::kept_base + slot).
struct KeptArrays {
    uint32_t *node;         // plane-local node id
    uint32_t *key;
    uint32_t *area;         // reference "quirk" area = cnt + nod
    int32_t  *parent;       // kept slot of the parent; root -> itself
    uint16_t *box;          // 4 per slot: x, y, w, h
    uint8_t  *level;
    uint32_t *start;        // NMS: chain start of the chain that owns this node
    uint32_t *ncand;        // NMS: number of child chains that want this node
    unsigned long long *best; // NMS: (order key << 32 | child slot), minimum wins
    uint32_t *perm;         // NMS, planes whose scratch does not fit LDS: the plane's kept slots in level order
};

// Cascade tables on the device (CascadeBoost, inc/adaboost.h:158-185).
// One stump, 32 bytes, laid out so a wave-uniform read of a whole stump is one s_load_dwordx8.
struct StumpRec {
    int32_t dim;
    int32_t mode;   // 0: fv < thr ? vp : vn (dir = +1)   1: fv > thr ? vp : vn (dir = -1)   2: general, see dir[]
    double  thr, vp, vn;
};

struct CascadeDev {
    const StumpRec *rec;    // same stumps as the arrays below, array-of-structures
    // integer form for 8-bit histogram counts h (valid when every dir is +1 or -1):
    //   value = (h < T) ? ab[2i] : ab[2i+1],  w[i] = dim | T << 10
    const uint32_t *w;
    const double   *ab;
    int32_t         all_unit;
    const uint16_t *dim;
    const double   *thr;
    const double   *dir;    // +1 for REAL; DecisionStump::dir for DISCRETE
    const double   *vp;     // value when fv*dir <  thr*dir (REAL: cp; DISCRETE: +weight)
    const double   *vn;     // value otherwise              (REAL: cn; DISCRETE: -weight)
    const int32_t  *stage_n;
    const int32_t  *stage_thresh;
    int32_t n_stages;
    int32_t n_stumps;
    int32_t max_stage;
};

struct DetectParams {
    int32_t thresh_step, min_area, max_area, stability_t;
    double  overlap_coef;
    int32_t hi;             // highest_level = 255/step + 1 (src/ER.cpp:247)
    float   qscale;         // float(1.0/step): the convertTo scale (src/ER.cpp:250)
    int32_t kept_cap, pool_cap;
    int32_t sibling_order;  // 0 exact (flood order / table order), 1 smallest key, 2 largest key
};

// Matches str_er_cand in include/str_er.h (48 bytes).
// OCR::rotate_mat (src/OCR.cpp:254-357) for one box: canvas size and mapping, computed on the host
// (libm cos/sin/tan/round, exactly as the reference evaluates them); on == 0 means "not rotated".
struct RotGeom {
    int32_t on, crop;
    int32_t rw, rh;            // canvas (the image ARAN then sees)
    int32_t min_x, min_y, max_x, max_y, ch, x0, y0;
    int32_t pad_;
    double  c, s;              // cos(rad), sin(rad)
};

struct CandRec {
    uint32_t frame;
    uint8_t  ch, pyr, level, cls;
    uint16_t x, y, w, h;
    uint32_t area, key;
    int32_t  node;
    uint32_t plane;
    double   score_strong, score_weak;
};
static_assert(sizeof(CandRec) == 48, "CandRec must match str_er_cand");

} // namespace str_er
