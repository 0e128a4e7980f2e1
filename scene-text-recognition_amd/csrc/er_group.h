// er_group.h -- the sequential half of ERFilter::er_grouping (src/ER.cpp:612-692): the greedy line assignment over
// the pair list the GPU produced, then per line the sort, overlap_suppression (:925-964), inner_suppression
// (:893-922), fitline_avgslope (:1361-1389) and the line's box.  Host code, as in the reference.
#pragma once
#include <cstdint>
#include <vector>

namespace str_er {

// bound + center of one ER of all_er; overlap_suppression rewrites them in place (the reference mutates the ER
// objects, which lines share)
struct GroupEr {
    int32_t x, y, w, h, cx, cy;
};

struct TextLine {
    std::vector<int32_t> ers;      // positions in all_er (the sorted list), in the line's own sorted order
    double  slope = 0;
    int32_t box[4] = {0, 0, 0, 0}; // union of the members' bounds (src/ER.cpp:684-690)
};

// ers: all_er after sort (+ inner_suppression); pairs: (i << 16 | j), i < j, in the reference's visiting order.
void group_lines(std::vector<GroupEr> &ers, const uint32_t *pairs, size_t n_pairs, std::vector<TextLine> &lines);

// er_grouping(all_er, text, overlap_sup = true, ...): the first two statements of src/ER.cpp:612-617 on the host -- `order` (indices into
// ers, in all_er's order) is sorted by center.x (stably) and overlap_suppression merges boxes into the survivors (rewriting their bound /
// center in `ers`) and drops the merged ones.  The list is NOT sorted again afterwards (the reference does not either).
void sort_and_overlap_suppress(std::vector<GroupEr> &ers, std::vector<int32_t> &order);

} // namespace str_er
