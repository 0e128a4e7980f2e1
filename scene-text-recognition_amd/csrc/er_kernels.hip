// er_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the extremal-region
// hot path.  One kernel per function of the reference's per-plane loop
// (src/ER.cpp:50-60); citations are to /root/reference.
//
//   k_bgr_to_ycrcb   compute_channels                 src/ER.cpp:114-128
//   k_resize         cv::resize (pyramid + ARAN)      src/OCR.cpp:401
//   k_tile_tree      er_tree_extract inside a tile    src/ER.cpp:240-413
//   k_seam           ... across tile borders
//   k_resolve / k_accumulate / k_root / k_select / k_kept
//                    er_accumulate + er_merge + pruning   src/ER.cpp:131-191
//   k_nms            non_maximum_supression           src/ER.cpp:416-505
//   k_classify       classify -> make_LBP_hist -> calc_LBP -> ARAN -> CascadeBoost::predict
//                                                     src/ER.cpp:507-528, 789-845
//
// The component tree is NOT computed by the reference's sequential flood.  Each
// 64x32 tile builds its tree in LDS with a lock-free "connect" (merge of two sorted
// root paths, LDS compare-and-swap); tile trees are then joined along the seams with
// the same connect on global memory (agent-scope atomics).  The node set, levels,
// boxes and areas of a component tree do not depend on the order in which it is
// built, so the result equals the flood's (tests/ check this bit for bit).
//
// Compile with -ffp-contract=off: the resize coefficients must be computed with the
// same IEEE operations as the host statement of cv::resize.
#include <hip/hip_runtime.h>

#include <float.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "er_device.h"
#include "er_kernels.h"
#include "tile2_body.h"

namespace str_er {

// One translation unit, six parts (device code without relocatable linking: helpers, macros and the developer trace buffers are shared):
#include "er_planes.inl"        // compute_channels, NV12 ingest, cv::resize
#include "er_tile_tree.inl"     // k_tile_tree
#include "er_tile_tree2.inl"    // k_tile_tree2 (chroma / few-level planes)
#include "er_tree_passes.inl"   // k_group_merge, k_seam, strips, k_resolve, k_reduce, k_root / k_select / k_kept
#include "er_nms.inl"           // k_nms and everything around NMS ties
#include "er_classify.inl"      // k_classify, k_lbp_boxes, k_cascade_fv

} // namespace str_er
