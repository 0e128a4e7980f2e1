// er_nms.inl -- part of er_kernels.hip (included there, inside namespace str_er; not a translation unit of its own): non_maximum_supression: k_nms, the opposite-rule pass, tie planes, the GPU flood-order replay, candidate packing.
// ------------------------------------------------------------------------------------
// non_maximum_supression (src/ER.cpp:416-505), one workgroup per plane.
//
// The reference walks the tree in post-order and lets every not-yet-claimed node X
// climb while bboxarea(X)/bboxarea(parent) > OVERLAP_COEF and the parent is unclaimed.
// Equivalent bottom-up form: start(P) = start(c) for the child c whose chain passes the
// overlap test on P, or P itself if no child chain does.  If two or more child chains
// pass, the reference's winner is the first of them in P's child list -- children are
// prepended when they are merged (src/ER.cpp:183-185), i.e. the child whose basin its
// flood ENTERED LAST (a basin is flooded completely once entered, so the merge order of
// sibling basins is their entry order).  That order is an artefact of the sequential
// flood and cannot be derived locally, so:
//   pass 0 (all planes) decides ties by key (largest / smallest, DetectParams::sibling_order)
//          and counts them (n_amb).  No tie -> the result does not depend on any order.
//          A tie at a node X whose box covers so much of the plane that every chain able to claim X or an ancestor of X
//          starts at a node failing the size filter of src/ER.cpp:489-490 (w < 0.8 cols && h < 0.8 rows) cannot change the
//          pool: such a start has bbox area > OVERLAP_COEF * area(X) >= 0.64 rows cols, so it and all chain members above it
//          are too big to be accepted, whoever wins; chains with acceptable starts never reach X.  Only the other ties count
//          in n_rel ("relevant") -- ties between background-sized regions are frequent on noisy frames, relevant ones are rare.
//   pass "alt" (exact mode, planes whose only tie is one relevant two-way tie): the NMS again under the opposite key rule.
//          Below the tie nothing is a choice, and if the alt pass meets the same single tie and no other, the two passes are
//          the only two outcomes there are; if their pools are equal the tie does not matter and n_rel is cleared.
//   exact mode (sibling_order 0): for the planes with relevant ties left k_flood_order replays the
//          reference's flood and stamps every pixel with the order in which it became
//          accessible; pass 1 repeats the NMS of those planes with ties decided by the
//          stamp of each child's key pixel (any pixel of a basin would do: the access
//          intervals of sibling basins are disjoint) -- largest stamp = entered last = wins.
//   uploaded trees (str_er_nms_tree): the table order is the child-list order.
// ------------------------------------------------------------------------------------
constexpr int NMS_THREADS = 1024;
constexpr int NMS_SORT_CAP = 4096;       // pooled ERs of a plane whose keys are ranked out of LDS
constexpr int NMS_LDS_CAP = 4096;        // kept nodes of a plane whose NMS scratch lives in LDS (16 bytes each)

// order word of a child in a tie: the smallest one wins
enum { NMS_ORD_KEY_MAX = 0, NMS_ORD_KEY_MIN = 1, NMS_ORD_INDEX = 2, NMS_ORD_STAMP = 3 };
enum { NMS_PASS_FIRST = 0, NMS_PASS_ALT = 1, NMS_PASS_STAMP = 2 };

__global__ __launch_bounds__(NMS_THREADS) void k_nms(BatchDev b, DetectParams prm, const ReplayItem *items, const uint8_t *scratch, int ord_mode, int pass)
{
    __shared__ uint32_t s_npool, s_alt_amb, s_alt_node, s_alt_nc, s_alt_diff;
    __shared__ uint32_t s_levels[8];
#ifdef STR_ER_WG_TRACE
#define NMS_MARK(i) do { if (threadIdx.x == 0 && pass == NMS_PASS_FIRST && blockIdx.x < 128u) g_wg_trace[128 + blockIdx.x][(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NMS_MARK(i) do { } while (0)
#endif
    NMS_MARK(0);
    const bool       pass1 = pass != NMS_PASS_FIRST;          // a repeat: the plane's counters stay as the first pass left them
    const bool       alt = pass == NMS_PASS_ALT;
    // (the opposite-rule pass runs on the handful of planes k_alt_list found, `scratch` = its list: one workgroup per listed plane instead
    // of one per plane of the batch, of which all but a few returned at once)
    const uint32_t   pi_ = pass == NMS_PASS_STAMP ? items[blockIdx.x].plane : alt ? reinterpret_cast<const uint32_t *>(scratch)[blockIdx.x] : blockIdx.x;
    if (pi_ == NONE) return;
    const int        pi = (int)pi_;
    PlaneCtr        &c = b.ctr[pi];
    if (alt && !(c.n_rel != 0 && c.n_amb == 1 && c.tie_nc == 2)) return;
    const PlaneDesc &pd = b.planes[pi];
    const size_t     kb = pd.kept_base, pb = pd.pool_base;
    if (c.n_kept > pd.kept_cap) return;         // (see k_kept)
    const uint32_t   K = c.n_kept;
    const int        tid = threadIdx.x;
    const uint8_t   *klev = b.ka.level + kb;
    const uint16_t  *kbox = b.ka.box + 4 * kb;
    // chain starts, proposal counts and best proposals: in LDS when the plane's kept nodes fit (they do on everything but noise-like
    // full-size planes) -- the level loop below is one dependent atomic / load round trip after another on these three
    __shared__ uint32_t s_nstart[NMS_LDS_CAP], s_nncand[NMS_LDS_CAP];
    __shared__ unsigned long long s_nbest[NMS_LDS_CAP];
    const bool       in_lds = c.n_kept <= (uint32_t)NMS_LDS_CAP;
    uint32_t        *kstart = in_lds ? s_nstart : b.ka.start + kb;
    uint32_t        *kncand = in_lds ? s_nncand : b.ka.ncand + kb;
    unsigned long long *kbest = in_lds ? s_nbest : b.ka.best + kb;
    // ... and, for the chain walks, the parents and the box areas (w * h) beside them
    __shared__ int32_t s_npar[NMS_LDS_CAP], s_narea[NMS_LDS_CAP];
    const int32_t   *kpar = in_lds ? s_npar : b.ka.parent + kb;
    auto barea = [&](uint32_t i) -> int { return in_lds ? s_narea[i] : (int)kbox[4 * i + 2] * (int)kbox[4 * i + 3]; };
    const uint32_t  *kkey = b.ka.key + kb;
    const int        maxl = (int)c.max_level;
    const uint32_t   root = c.root_slot;
    const double     rel_area = 0.8 * (double)pd.w * 0.8 * (double)pd.h * (1.0 + 1e-9);   // OVERLAP_COEF * area(X) below this: the tie at X is relevant
    // stamps of the flood order walk: per watched key (the usual case) or, when the plane had more candidates than the watch
    // list holds, one per pixel
    __shared__ uint32_t s_wkey[NMS_WATCH_CAP], s_wstamp[NMS_WATCH_CAP];
    const uint32_t   n_watch = c.n_watch;
    const bool       sparse = n_watch <= (uint32_t)NMS_WATCH_CAP;
    const uint32_t  *stamp = (pass == NMS_PASS_STAMP && !sparse) ? reinterpret_cast<const uint32_t *>(scratch + items[blockIdx.x].off) : nullptr;
    if (pass == NMS_PASS_STAMP && sparse)
        for (uint32_t i = threadIdx.x; i < n_watch; i += NMS_THREADS) {
            s_wkey[i] = b.watch[(size_t)pi * NMS_WATCH_CAP + i];
            s_wstamp[i] = b.wstamp[(size_t)pi * NMS_WATCH_CAP + i];
        }

    for (int i = tid; i < 8; i += NMS_THREADS) s_levels[i] = 0;
    if (tid == 0) { s_npool = 0; s_alt_amb = 0; s_alt_node = NONE; s_alt_nc = 0; s_alt_diff = 0; }
    __syncthreads();
    // LDS planes (all but noise-like full-size ones): the static facts of the <= 4 nodes a thread owns -- level, parent, key, its box area and the
    // parent's -- stay in registers, and the box area of every chain start sits beside the start.  Round 4, traced in place (tools/dev_nms_trace.py): the level
    // loop was 70 % of the kernel -- 5.4 k cycles a level on the largest plane of a frame -- because every level re-read the levels of ALL nodes from memory
    // and walked generic pointers; the chain evaluation, one thread per chain with a division per step, another 20 %.
    constexpr int NPT = NMS_LDS_CAP / NMS_THREADS;
    __shared__ int s_nsa[NMS_LDS_CAP];                   // box area of kstart[i]
    int      lev_r[NPT], area_r[NPT], parea_r[NPT];
    uint32_t par_r[NPT], key_r[NPT];
    // (the nodes are handed out in level order -- a counting sort in LDS: the nodes of one level then sit in neighbouring lanes of one or two of a thread's
    // four turns, and a level costs the waves that have nodes there one pass of the loop body instead of four)
    __shared__ uint32_t s_lcur[256];
    __shared__ uint16_t s_perm[NMS_LDS_CAP];
    uint32_t node_r[NPT];
    // (planes too big for LDS -- a 3840 x 2160 level-0 plane keeps 10-20 k nodes -- are handed out in level order too, from a list in memory: every pass
    // below then touches a node once, where each level of the level loop used to test the level byte of ALL nodes: 1.03 ms for the largest plane of a
    // 4K batch, a quarter of that batch's GPU time)
    uint32_t *kperm = b.ka.perm + kb;
    if (tid < 256) s_lcur[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < K; i += NMS_THREADS) {
        kstart[i] = i; kncand[i] = 0; kbest[i] = ~0ull;
        if (in_lds) { s_npar[i] = b.ka.parent[kb + i]; s_nsa[i] = s_narea[i] = (int)kbox[4 * i + 2] * (int)kbox[4 * i + 3]; }
        atomicAdd(&s_lcur[klev[i]], 1u);
        atomicOr(&s_levels[klev[i] >> 5], 1u << (klev[i] & 31));
    }
    __syncthreads();
    {
        if (tid < 64) {              // counts -> first positions: four levels a lane, a scan over the wave
            const uint32_t c0 = s_lcur[4 * tid], c1 = s_lcur[4 * tid + 1], c2 = s_lcur[4 * tid + 2], c3 = s_lcur[4 * tid + 3], tot = c0 + c1 + c2 + c3;
            uint32_t incl = tot;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if (tid >= d) incl += o; }
            const uint32_t e = incl - tot;
            s_lcur[4 * tid] = e; s_lcur[4 * tid + 1] = e + c0; s_lcur[4 * tid + 2] = e + c0 + c1; s_lcur[4 * tid + 3] = e + c0 + c1 + c2;
        }
        __syncthreads();
        if (in_lds) for (uint32_t i = tid; i < K; i += NMS_THREADS) s_perm[atomicAdd(&s_lcur[klev[i]], 1u)] = (uint16_t)i;
        else for (uint32_t i = tid; i < K; i += NMS_THREADS) kperm[atomicAdd(&s_lcur[klev[i]], 1u)] = i;
        __syncthreads();          // (s_lcur[l] is now the END of level l's stretch of the list = the start of level l + 1's)
    }
    if (in_lds) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const uint32_t pos = (uint32_t)tid + (uint32_t)k * NMS_THREADS;
            lev_r[k] = -1; node_r[k] = 0; par_r[k] = 0; key_r[k] = 0; area_r[k] = 1; parea_r[k] = 1;
            if (pos < K) {
                const uint32_t i = s_perm[pos];
                node_r[k] = i; lev_r[k] = klev[i]; par_r[k] = (uint32_t)s_npar[i]; key_r[k] = kkey[i]; area_r[k] = s_narea[i]; parea_r[k] = s_narea[par_r[k]];
            }
        }
    }
    // (double)as / (double)ap > overlap_coef, the reference's test (src/ER.cpp:452), without the division unless the quotient is within 1e-9 of the coefficient
    auto ratio_gt = [&](int as, int ap) -> bool {
        const double x = (double)as, y = (double)ap, d = x - prm.overlap_coef * y;
        if (fabs(d) > 1e-9 * y) return d > 0.0;
        return x / y > prm.overlap_coef;
    };
    NMS_MARK(1);

    for (int t = 0; t <= maxl; ++t) {
        if (!((s_levels[t >> 5] >> (t & 31)) & 1)) continue;      // no kept node at this level
        // settle the nodes of level t (all their children, at lower levels, have proposed), then
        // let them propose to their parents; one barrier per level is enough because a node only
        // reads what lower levels wrote and only writes to higher levels
        if (in_lds) {
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                if (lev_r[k] != t) continue;
                const uint32_t i = node_r[k];
                const uint32_t nc = s_nncand[i];
                const uint32_t child = (uint32_t)(s_nbest[i] & 0xFFFFFFFFull);
                uint32_t s = i;
                int      as = area_r[k];
                if (nc) {
                    s = s_nstart[child]; as = s_nsa[child];
                    s_nstart[i] = s; s_nsa[i] = as;
                    if (nc > 1 && !pass1) {
                        atomicAdd(&c.n_amb, 1u);
                        c.tie_node = i; c.tie_nc = nc;
                        if ((double)area_r[k] * prm.overlap_coef < rel_area) atomicAdd(&c.n_rel, 1u);
                    }
                    if (nc > 1 && alt) { atomicAdd(&s_alt_amb, 1u); s_alt_node = i; s_alt_nc = nc; }
                }
                if (i == root) continue;
                const uint32_t P = par_r[k];
                const int      ap = parea_r[k];
                if (ratio_gt(as, ap)) {
                    atomicAdd(&s_nncand[P], 1u);
                    uint32_t ord;
                    if (ord_mode == NMS_ORD_STAMP) {                             // entered last = first in the child list
                        uint32_t st = 0;
                        if (!sparse) st = stamp[key_r[k]];
                        else if (ratio_gt(area_r[k], ap)) {                      // (only such children are watched)
                            for (uint32_t j = 0; j < n_watch; ++j) if (s_wkey[j] == key_r[k]) { st = s_wstamp[j]; break; }
                        }
                        ord = ~st;
                    }
                    else if (ord_mode == NMS_ORD_INDEX) ord = i;
                    else ord = ord_mode == NMS_ORD_KEY_MAX ? ~key_r[k] : key_r[k];
                    atomicMin(&s_nbest[P], ((unsigned long long)ord << 32) | i);
                }
            }
        } else
        for (uint32_t pos = (t ? s_lcur[t - 1] : 0u) + (uint32_t)tid, pend = s_lcur[t]; pos < pend; pos += NMS_THREADS) {
            const uint32_t i = kperm[pos];
            uint32_t s = i;
            const uint32_t nc = LD_AGENT(&kncand[i]);
            if (nc) {
                const uint32_t child = (uint32_t)(LD_AGENT(&kbest[i]) & 0xFFFFFFFFull);
                s = kstart[child];
                kstart[i] = s;
                if (nc > 1 && !pass1) {
                    atomicAdd(&c.n_amb, 1u);
                    c.tie_node = i; c.tie_nc = nc;
                    if ((double)(barea(i)) * prm.overlap_coef < rel_area) atomicAdd(&c.n_rel, 1u);
                }
                if (nc > 1 && alt) { atomicAdd(&s_alt_amb, 1u); s_alt_node = i; s_alt_nc = nc; }
            }
            if (i == root) continue;
            const uint32_t P = (uint32_t)kpar[i];
            const int as = barea(s);
            const int ap = barea(P);
            if ((double)as / (double)ap > prm.overlap_coef) {
                atomicAdd(&kncand[P], 1u);
                uint32_t ord;
                if (ord_mode == NMS_ORD_STAMP) {                             // entered last = first in the child list
                    uint32_t st = 0;
                    if (!sparse) st = stamp[kkey[i]];
                    else if ((double)(barea(i)) / (double)ap > prm.overlap_coef) {   // (only such children are watched)
                        const uint32_t key = kkey[i];
                        for (uint32_t j = 0; j < n_watch; ++j) if (s_wkey[j] == key) { st = s_wstamp[j]; break; }
                    }
                    ord = ~st;
                }
                else if (ord_mode == NMS_ORD_INDEX) ord = i;
                else ord = ord_mode == NMS_ORD_KEY_MAX ? ~kkey[i] : kkey[i];
                atomicMin(&kbest[P], ((unsigned long long)ord << 32) | i);
            }
        }
        __syncthreads();
    }

    NMS_MARK(2);
    // planes with ties (exact mode only needs it): the key pixels the flood replay has to reach.  Which ties occur can depend
    // on how lower ties were decided, so the list holds every child that COULD compete whatever the order: a chain start
    // lies inside its child's box, so only children whose own box covers more than OVERLAP_COEF of the parent's can pass, and
    // a parent needs two of them.
    if (!pass1 && prm.sibling_order == 0 && ord_mode != NMS_ORD_INDEX && LD_AGENT(&c.n_rel) != 0) {
        for (uint32_t i = tid; i < K; i += NMS_THREADS) kncand[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < K; i += NMS_THREADS) {
            if (i == root) continue;
            const uint32_t P = (uint32_t)kpar[i];
            const int ai = barea(i), ap = barea(P);
            if ((double)ai / (double)ap > prm.overlap_coef) atomicAdd(&kncand[P], 1u);
        }
        __syncthreads();
        for (uint32_t i = tid; i < K; i += NMS_THREADS) {
            if (i == root) continue;
            const uint32_t P = (uint32_t)kpar[i];
            const int ai = barea(i), ap = barea(P);
            if ((double)ai / (double)ap > prm.overlap_coef && kncand[P] > 1 && (double)ap * prm.overlap_coef < rel_area) {
                const uint32_t at = atomicAdd(&c.n_watch, 1u);
                if (at < (uint32_t)NMS_WATCH_CAP) { b.watch[(size_t)pi * NMS_WATCH_CAP + at] = kkey[i]; b.wparent[(size_t)pi * NMS_WATCH_CAP + at] = P; }
            }
        }
        __syncthreads();
    }

    NMS_MARK(3);
    // evaluate every chain from its start (src/ER.cpp:464-497)
    const int T = prm.stability_t;
    if (in_lds) {
        // Every member of a chain whose T-th ancestor is still in the chain has a stability of its own: all of them at once, a division each; the chain's
        // winner -- largest stability, then smallest box, then lowest (src/ER.cpp:470-484 keeps the earlier one; along a chain the boxes only grow, so
        // "lowest level" decides both) -- by two rounds of LDS atomics on the slot of the chain's start.  A stability is positive or +inf: 0 = "none yet".
        for (uint32_t i = tid; i < K; i += NMS_THREADS) { s_nbest[i] = 0ull; s_nncand[i] = 0xFFFFFFFFu; }
        __syncthreads();
        unsigned long long st_r[NPT];
        uint32_t           X_r[NPT];
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const uint32_t i = node_r[k];
            st_r[k] = 0ull; X_r[k] = 0;
            if (lev_r[k] < 0) continue;
            const uint32_t X = s_nstart[i];
            uint32_t       anc = i;
            bool           ok = true;
            for (int j = 0; j < T; ++j) { if (anc == root) { ok = false; break; } anc = (uint32_t)s_npar[anc]; }
            if (!ok || s_nstart[anc] != X) continue;
            const int    a = area_r[k], bb = s_narea[anc];
            const double st = (double)a / (double)(bb - a);   // 0 denominator -> +inf, as in the reference
            st_r[k] = (unsigned long long)__double_as_longlong(st); X_r[k] = X;
            atomicMax(&s_nbest[X], st_r[k]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPT; ++k)
            if (st_r[k] != 0ull && st_r[k] == s_nbest[X_r[k]]) atomicMin(&s_nncand[X_r[k]], ((uint32_t)lev_r[k] << 16) | node_r[k]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const uint32_t best = node_r[k];
            if (st_r[k] == 0ull || st_r[k] != s_nbest[X_r[k]] || s_nncand[X_r[k]] != (((uint32_t)lev_r[k] << 16) | best)) continue;
            const int    bw = kbox[4 * best + 2], bh = kbox[4 * best + 3];
            const double ar = (double)bw / (double)bh;
            const int    area = (int)b.ka.area[kb + best];
            if (ar < 2.0 && ar > 0.10 && area < prm.max_area && area > prm.min_area && bh < pd.h * 0.8 && bw < pd.w * 0.8) {
                const uint32_t slot = atomicAdd(&s_npool, 1u);
                if (slot < pd.pool_cap) b.pool_tmp[pb + slot] = best;
            }
        }
    } else {
        // the same three rounds on the tables in memory (kbest: the chain's largest stability, in the slot of its start; kncand: the level of the lowest
        // member that has it -- levels grow along a chain, so the level names the member).  A member's stability is worked out again in each round: T
        // dependent loads, against a table of 8 bytes per kept node
        for (uint32_t i = tid; i < K; i += NMS_THREADS) { kbest[i] = 0ull; kncand[i] = 0xFFFFFFFFu; }
        __syncthreads();
        auto stab = [&](uint32_t i, uint32_t &X) -> unsigned long long {
            X = kstart[i];
            uint32_t anc = i;
            for (int j = 0; j < T; ++j) { if (anc == root) return 0ull; anc = (uint32_t)kpar[anc]; }
            if (kstart[anc] != X) return 0ull;
            const int    a = barea(i), bb = barea(anc);
            const double st = (double)a / (double)(bb - a);   // 0 denominator -> +inf, as in the reference
            return (unsigned long long)__double_as_longlong(st);
        };
        for (uint32_t i = tid; i < K; i += NMS_THREADS) {
            uint32_t X;
            const unsigned long long st = stab(i, X);
            if (st != 0ull) atomicMax(&kbest[X], st);
        }
        __syncthreads();
        for (uint32_t i = tid; i < K; i += NMS_THREADS) {
            uint32_t X;
            const unsigned long long st = stab(i, X);
            if (st != 0ull && st == LD_AGENT(&kbest[X])) atomicMin(&kncand[X], (uint32_t)klev[i]);
        }
        __syncthreads();
        for (uint32_t i = tid; i < K; i += NMS_THREADS) {
            uint32_t X;
            const unsigned long long st = stab(i, X);
            if (st == 0ull || st != LD_AGENT(&kbest[X]) || LD_AGENT(&kncand[X]) != (uint32_t)klev[i]) continue;
            const uint32_t best = i;
            const int    bw = kbox[4 * best + 2], bh = kbox[4 * best + 3];
            const double ar = (double)bw / (double)bh;
            const int    area = (int)b.ka.area[kb + best];
            if (ar < 2.0 && ar > 0.10 && area < prm.max_area && area > prm.min_area && bh < pd.h * 0.8 && bw < pd.w * 0.8) {
                const uint32_t slot = atomicAdd(&s_npool, 1u);
                if (slot < pd.pool_cap) b.pool_tmp[pb + slot] = best;
            }
        }
    }
    __syncthreads();
    NMS_MARK(4);
    uint32_t np = s_npool;
    if (np > pd.pool_cap) {
        if (tid == 0) atomicOr(&c.overflow, 2u);
        np = pd.pool_cap;
    }
    // order the pool by key (keys are unique inside a plane): rank = number of smaller keys.  The keys are staged in LDS
    // first -- ranking straight from the tables is two dependent global loads per comparison, the longest part of the kernel
    __shared__ uint32_t s_keys[NMS_SORT_CAP];
    // (the alt pass only compares: is its pool the first pass's pool?)
    // (the tie pass notes whether its pool differs from the first pass's: only then the plane's candidates are classified again)
    const bool cmp = pass == NMS_PASS_STAMP;
    if ((alt || cmp) && np != c.n_pool) s_alt_diff = 1;
    if (np <= (uint32_t)NMS_SORT_CAP) {
        for (uint32_t i = tid; i < np; i += NMS_THREADS) s_keys[i] = kkey[b.pool_tmp[pb + i]];
        __syncthreads();
        for (uint32_t i = tid; i < np; i += NMS_THREADS) {
            const uint32_t mk = s_keys[i];
            uint32_t       rank = 0;
            for (uint32_t j = 0; j < np; ++j) rank += s_keys[j] < mk;
            if (cmp && rank < c.n_pool && b.pool[pb + rank] != b.pool_tmp[pb + i]) s_alt_diff = 1;
            if (!alt) b.pool[pb + rank] = b.pool_tmp[pb + i];
            else if (rank >= c.n_pool || b.pool[pb + rank] != b.pool_tmp[pb + i]) s_alt_diff = 1;
        }
    } else {
        for (uint32_t i = tid; i < np; i += NMS_THREADS) {
            const uint32_t me = b.pool_tmp[pb + i], mk = kkey[me];
            uint32_t       rank = 0;
            for (uint32_t j = 0; j < np; ++j) rank += kkey[b.pool_tmp[pb + j]] < mk;
            if (cmp && rank < c.n_pool && b.pool[pb + rank] != me) s_alt_diff = 1;
            if (!alt) b.pool[pb + rank] = me;
            else if (rank >= c.n_pool || b.pool[pb + rank] != me) s_alt_diff = 1;
        }
    }
    if (alt) {
        __syncthreads();
        // the same single two-way tie and no other one, and the same pool: whichever child the reference's flood entered last,
        // the pool is this one
        if (tid == 0 && s_alt_diff == 0 && s_alt_amb == 1 && s_alt_node == c.tie_node && s_alt_nc == 2) { c.n_rel = 0; c.n_watch = 0; }
        return;
    }
    if (cmp) __syncthreads();
    NMS_MARK(5);
    if (threadIdx.x == 0 && pass == NMS_PASS_FIRST && blockIdx.x < 128u) { NMS_MARK(6); }
#ifdef STR_ER_WG_TRACE
    if (threadIdx.x == 0 && pass == NMS_PASS_FIRST && blockIdx.x < 128u) { g_wg_trace[128 + blockIdx.x][7] = K; g_wg_trace[128 + blockIdx.x][8] = (unsigned long long)maxl; g_wg_trace[128 + blockIdx.x][9] = np; }
#endif
    if (tid == 0) {
        c.n_pool = np;
        if (cmp) c.pool_changed = s_alt_diff;
    }
}

void launch_nms(hipStream_t s, const BatchDev &b, const DetectParams &p, bool use_index_order)
{
    if (!b.n_planes) return;
    const int mode = (use_index_order && p.sibling_order == 0) ? NMS_ORD_INDEX : (p.sibling_order == 1 ? NMS_ORD_KEY_MIN : NMS_ORD_KEY_MAX);
    hipLaunchKernelGGL(k_nms, dim3(b.n_planes), dim3(NMS_THREADS), 0, s, b, p, (const ReplayItem *)nullptr, (const uint8_t *)nullptr, mode, (int)NMS_PASS_FIRST);
}

// exact mode: planes whose only tie is one two-way tie are tried under the opposite rule; equal pools settle them (n_rel = 0).
// Touches only NMS scratch and the counters n_rel / n_watch, so it may run beside the kernels that consume the pools.
// the planes whose only tie is one relevant two-way tie (the only ones the opposite-rule pass can settle), at most NMS_ALT_CAP of them; a
// plane that finds no room keeps its tie for the flood order walk
__global__ __launch_bounds__(1024) void k_alt_list(BatchDev b, uint32_t *list)
{
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    for (int i = threadIdx.x; i < NMS_ALT_CAP; i += blockDim.x) list[i] = NONE;
    __syncthreads();
    for (int pi = threadIdx.x; pi < b.n_planes; pi += blockDim.x) {
        const PlaneCtr &c = b.ctr[pi];
        if (c.n_rel != 0 && c.n_amb == 1 && c.tie_nc == 2) {
            const uint32_t at = atomicAdd(&s_n, 1u);
            if (at < (uint32_t)NMS_ALT_CAP) list[at] = (uint32_t)pi;
        }
    }
}

void launch_nms_alt(hipStream_t s, const BatchDev &b, const DetectParams &p, uint32_t *alt_list)
{
    if (!b.n_planes || p.sibling_order != 0) return;
    hipLaunchKernelGGL(k_alt_list, dim3(1), dim3(1024), 0, s, b, alt_list);
    hipLaunchKernelGGL(k_nms, dim3(NMS_ALT_CAP), dim3(NMS_THREADS), 0, s, b, p, (const ReplayItem *)nullptr, reinterpret_cast<const uint8_t *>(alt_list), (int)NMS_ORD_KEY_MIN,
                       (int)NMS_PASS_ALT);
}

void launch_nms_resolve(hipStream_t s, const BatchDev &b, const DetectParams &p, const ReplayItem *items, int n_items, const uint8_t *scratch)
{
    if (n_items <= 0) return;
    hipLaunchKernelGGL(k_nms, dim3(n_items), dim3(NMS_THREADS), 0, s, b, p, items, scratch, (int)NMS_ORD_STAMP, (int)NMS_PASS_STAMP);
}

// The planes whose ties need the flood order walk, handed to the host without a round trip: straight after the opposite-rule pass
// every such plane (rare: about one in a thousand) is written into page-locked host memory the device can address -- pixels,
// then its watch list (keys, parents) -- so that it is already there when the host learns, from the plane counters, that it
// needs it.  slot_plane[slot] = plane index; planes beyond n_slots are fetched by an explicit copy later.
// Two launches: one workgroup hands the (few) tie planes of the batch their slots, then EXPORT_SPLIT workgroups per slot copy the plane
// with 16-byte stores -- a 1920 x 1080 plane crosses the host link in ~0.1 ms.  (Round 2: one 1024-lane workgroup per plane of the batch,
// the one with work pushing its 2 MB through dword stores: 0.9 ms per batch, the second-largest kernel of the profile.)
constexpr int EXPORT_SPLIT = 32;
__global__ __launch_bounds__(1024) void k_tie_slots(BatchDev b, size_t slot_bytes, int n_slots, uint32_t *slot_plane_dev, uint32_t *count, uint32_t *slot_plane)
{
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    for (int i = threadIdx.x; i < n_slots; i += blockDim.x) slot_plane_dev[i] = NONE;
    __syncthreads();
    for (int pi = threadIdx.x; pi < b.n_planes; pi += blockDim.x) {
        if (b.ctr[pi].n_rel == 0) continue;
        const uint32_t slot = atomicAdd(&s_n, 1u);
        if (slot >= (uint32_t)n_slots) continue;
        const PlaneDesc &pd = b.planes[pi];
        const bool fits = (size_t)pd.w * pd.h + 2 * 4 * (size_t)NMS_WATCH_CAP + 256 <= slot_bytes;
        slot_plane_dev[slot] = fits ? (uint32_t)pi : NONE;
        slot_plane[slot] = fits ? (uint32_t)pi : NONE;          // (host memory: read by the host after the batch's synchronisation)
    }
    __syncthreads();
    if (threadIdx.x == 0) *count = s_n;
}

// one plane (packed, w bytes per row) and behind it, 256-byte aligned, its watch list (keys, then parents) into host memory; the rows are
// dealt to the gridDim.x workgroups that share the plane
__device__ __forceinline__ void export_plane(const BatchDev &b, uint32_t pi, uint8_t *dst)
{
    const PlaneCtr  &c = b.ctr[pi];
    const PlaneDesc &pd = b.planes[pi];
    const size_t n = (size_t)pd.w * pd.h;
    // 16-byte / 4-byte / single-byte moves as the geometry allows (the planes the library builds have 64-byte aligned rows)
    const uintptr_t a = reinterpret_cast<uintptr_t>(pd.pix) | (uintptr_t)pd.stride | (uintptr_t)pd.w;
    for (int y = blockIdx.x; y < pd.h; y += gridDim.x) {
        const uint8_t *src = pd.pix + (size_t)y * pd.stride;
        uint8_t       *d = dst + (size_t)y * pd.w;
        if ((a & 15) == 0)
            for (int x = threadIdx.x; x < pd.w / 16; x += blockDim.x) reinterpret_cast<uint4 *>(d)[x] = reinterpret_cast<const uint4 *>(src)[x];
        else if ((a & 3) == 0)
            for (int x = threadIdx.x; x < pd.w / 4; x += blockDim.x) reinterpret_cast<uint32_t *>(d)[x] = reinterpret_cast<const uint32_t *>(src)[x];
        else
            for (int x = threadIdx.x; x < pd.w; x += blockDim.x) d[x] = src[x];
    }
    if (blockIdx.x == 0) {
        uint32_t *wl = reinterpret_cast<uint32_t *>(dst + ((n + 255) / 256) * 256);
        const uint32_t nw = min(c.n_watch, (uint32_t)NMS_WATCH_CAP);
        for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) {
            wl[i] = b.watch[(size_t)pi * NMS_WATCH_CAP + i];
            wl[NMS_WATCH_CAP + i] = b.wparent[(size_t)pi * NMS_WATCH_CAP + i];
        }
    }
}

__global__ __launch_bounds__(256) void k_export_tie_planes(BatchDev b, uint8_t *host_buf, size_t slot_bytes, const uint32_t *slot_plane_dev)
{
    const uint32_t pi = slot_plane_dev[blockIdx.y];
    if (pi != NONE) export_plane(b, pi, host_buf + (size_t)blockIdx.y * slot_bytes);
}

// ... and the tie planes that found no slot (a batch with more than a handful: tie-rich content), once the host knows which they are:
// item k goes to host_buf + 256 * items[k].pad_ (0xFFFFFFFF: already exported).  A pitched hipMemcpy2DAsync per plane took milliseconds each.
__global__ __launch_bounds__(256) void k_export_listed_planes(BatchDev b, const ReplayItem *items, uint8_t *host_buf)
{
    const ReplayItem it = items[blockIdx.y];
    if (it.pad_ != 0xFFFFFFFFu) export_plane(b, it.plane, host_buf + (size_t)it.pad_ * 256);
}

void launch_export_listed_planes(hipStream_t s, const BatchDev &b, const ReplayItem *items, int n_items, uint8_t *host_buf)
{
    if (n_items > 0) hipLaunchKernelGGL(k_export_listed_planes, dim3(EXPORT_SPLIT, n_items), dim3(256), 0, s, b, items, host_buf);
}

void launch_export_tie_planes(hipStream_t s, const BatchDev &b, uint8_t *host_buf, size_t slot_bytes, int n_slots, uint32_t *slot_plane_dev, uint32_t *count,
                              uint32_t *slot_plane)
{
    if (!b.n_planes || !host_buf || n_slots <= 0) return;
    hipLaunchKernelGGL(k_tie_slots, dim3(1), dim3(1024), 0, s, b, slot_bytes, n_slots, slot_plane_dev, count, slot_plane);
    hipLaunchKernelGGL(k_export_tie_planes, dim3(EXPORT_SPLIT, n_slots), dim3(256), 0, s, b, host_buf, slot_bytes, slot_plane_dev);
}

// ------------------------------------------------------------------------------------
// Replay of the reference's flood (er_tree_extract, src/ER.cpp:240-374) for the planes whose NMS has sibling ties: same
// start pixel, same edge order (right, bottom, left, top), same 256 LIFO buckets, same "priority == highest_level means
// empty" rule -- but nothing is built, every pixel is only stamped with the order in which it is marked accessible.
// The flood is inherently sequential: ONE lane per plane walks it (the other lanes only prepare the scratch), which costs
// about a microsecond per pixel.  It runs only for planes where the reference's own answer depends on this order
// (about 1 plane in 100 on photographs, none on the synthetic bench frames) and stops as soon as every watched key pixel
// has its stamp.
//   scratch per plane: stamp u32[n] (0 = not accessible yet; WATCH = not accessible, watched), link u32[n] (bucket lists:
//   next entry << 3 | edge), level u8[n].
// ------------------------------------------------------------------------------------
constexpr int      FLOOD_THREADS = 1024;
constexpr uint32_t FLOOD_WATCH = 0x80000000u;
constexpr uint32_t FLOOD_NIL = 0x1FFFFFFFu;

size_t replay_scratch_bytes(int w, int h)
{
    const size_t n = (size_t)w * h;
    return ((9 * n + 255) / 256) * 256 + 256;
}

__global__ __launch_bounds__(FLOOD_THREADS) void k_flood_order(BatchDev b, DetectParams prm, const ReplayItem *items, uint8_t *scratch)
{
    __shared__ uint32_t s_head[257];
    const ReplayItem it = items[blockIdx.x];
    const PlaneDesc &pd = b.planes[it.plane];
    const PlaneCtr  &c = b.ctr[it.plane];
    const int        w = pd.w, h = pd.h;
    const uint32_t   n = (uint32_t)w * (uint32_t)h;
    uint32_t *stamp = reinterpret_cast<uint32_t *>(scratch + it.off);
    uint32_t *link = stamp + n;
    uint8_t  *lv = reinterpret_cast<uint8_t *>(link + n);
    const uint32_t hi = (uint32_t)prm.hi;
    for (uint32_t i = threadIdx.x; i < n; i += FLOOD_THREADS) {
        const uint32_t y = i / (uint32_t)w, x = i - y * (uint32_t)w;
        const uint32_t q = (uint32_t)__float2int_rn((float)(pd.pix[(size_t)y * pd.stride + x] ^ pd.invert) * prm.qscale);   // src/ER.cpp:250
        lv[i] = (uint8_t)min(q, 255u);
        stamp[i] = 0;
    }
    for (uint32_t i = threadIdx.x; i < 257; i += FLOOD_THREADS) s_head[i] = FLOOD_NIL;
    __syncthreads();
    const uint32_t n_watch = c.n_watch;
    const bool     watching = n_watch <= (uint32_t)NMS_WATCH_CAP;
    if (watching) for (uint32_t i = threadIdx.x; i < n_watch; i += FLOOD_THREADS) stamp[b.watch[(size_t)it.plane * NMS_WATCH_CAP + i]] = FLOOD_WATCH;
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x != 0) return;

    uint32_t remaining = watching ? n_watch : 0xFFFFFFFFu;     // distinct pixels: keys are unique inside a plane
    uint32_t counter = 0;
    uint32_t priority = hi;
    uint32_t cur = 0, edge = 0;
    uint32_t cl = lv[0];                       // (levels >= hi can only be == hi for step >= 2; for step 1 hi = 256 is never reached)
    {
        const uint32_t old = stamp[0];
        stamp[0] = ++counter;
        if (old == FLOOD_WATCH) --remaining;
    }
    while (remaining != 0) {
        // 4. explore the remaining edges of the current pixel
        const uint32_t x = cur % (uint32_t)w;
        uint32_t nb[4], st[4], nl[4];
        nb[0] = (x + 1 < (uint32_t)w) ? cur + 1 : cur;
        nb[1] = (cur + (uint32_t)w < n) ? cur + (uint32_t)w : cur;
        nb[2] = (x > 0) ? cur - 1 : cur;
        nb[3] = (cur >= (uint32_t)w) ? cur - (uint32_t)w : cur;
#pragma unroll
        for (int e = 0; e < 4; ++e) { st[e] = stamp[nb[e]]; nl[e] = lv[nb[e]]; }       // eight loads in flight
        bool descended = false;
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            if (e < edge || descended) continue;
            const uint32_t q = nb[e];
            if (q == cur || (st[e] != 0 && st[e] != FLOOD_WATCH)) continue;
            stamp[q] = ++counter;
            if (st[e] == FLOOD_WATCH) --remaining;
            const uint32_t l = nl[e];
            if (l >= cl) {
                if (l < hi) { link[q] = (s_head[l] << 3); s_head[l] = q; }        // (bucket `hi` is never popped: src/ER.cpp:343)
                if (l < priority) priority = l;
            } else {
                if (cl < hi) { link[cur] = (s_head[cl] << 3) | (e + 1u); s_head[cl] = cur; }
                if (cl < priority) priority = cl;
                cur = q; cl = l; edge = 0;
                descended = true;
            }
        }
        if (remaining == 0) break;
        if (descended) continue;
        // 5./6. the current pixel is done; pop the lowest boundary pixel
        if (priority == hi) break;
        cur = s_head[priority];
        const uint32_t v = link[cur];
        edge = v & 7u;
        s_head[priority] = v >> 3;
        cl = priority;
        while (priority < hi && s_head[priority] == FLOOD_NIL) ++priority;
    }
    // what the NMS pass reads: the stamps of the watched pixels (a watched pixel the walk never reached keeps the mark: 0)
    if (watching)
        for (uint32_t i = 0; i < n_watch; ++i) {
            const uint32_t v = stamp[b.watch[(size_t)it.plane * NMS_WATCH_CAP + i]];
            b.wstamp[(size_t)it.plane * NMS_WATCH_CAP + i] = v == FLOOD_WATCH ? 0u : v;
        }
}

void launch_flood_order(hipStream_t s, const BatchDev &b, const DetectParams &p, const ReplayItem *items, int n_items, uint8_t *scratch)
{
    if (n_items <= 0) return;
    hipLaunchKernelGGL(k_flood_order, dim3(n_items), dim3(FLOOD_THREADS), 0, s, b, p, items, scratch);
}

// exclusive prefix of the pool sizes: where every plane's candidates go in the packed array;
// also the candidate -> plane table, so k_classify does not have to search.
__global__ __launch_bounds__(1024) void k_cand_prefix(BatchDev b)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < b.n_planes; base += 1024) {
        const int      i = base + tid;
        const uint32_t n = i < b.n_planes ? b.ctr[i].n_pool : 0;
        const uint32_t incl = wave_incl_scan(n);
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t off = s_carry, tot = 0;
        for (int k = 0; k < 16; ++k) { if (k < wv) off += s_w[k]; tot += s_w[k]; }
        const uint32_t mine = off + incl - n;
        if (i < b.n_planes) {
            b.ctr[i].cand_base = mine;
            b.ctr[i].n_strong = 0; b.ctr[i].n_weak = 0;     // k_classify counts into these
        }
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    if (tid == 0) *b.total_cands = s_carry;
    // candidate -> plane, a wave per plane (one lane writing a plane's few hundred entries one after the other was most of this kernel)
    for (int i = wv; i < b.n_planes; i += 16) {
        const uint32_t n = b.ctr[i].n_pool, base = b.ctr[i].cand_base;
        for (uint32_t k = lane; k < n; k += 64) b.cand_plane[base + k] = (uint16_t)i;
    }
}

void launch_cand_prefix(hipStream_t s, const BatchDev &b)
{
    hipLaunchKernelGGL(k_cand_prefix, dim3(1), dim3(1024), 0, s, b);
}

// After the NMS tie pass (exact sibling ties): the pools of a few planes may have changed.  New candidate offsets for all planes
// (b.cands / b.cand_plane are the SECOND set of buffers); the candidates of the changed planes go on the list of k_classify, the
// records of all other planes are moved over from the first set (k_cand_move).
__global__ __launch_bounds__(1024) void k_cand_reprefix(BatchDev b, uint32_t *redo, uint32_t *n_redo)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_carry = 0; *n_redo = 0; }
    __syncthreads();
    for (int base = 0; base < b.n_planes; base += 1024) {
        const int      i = base + tid;
        const uint32_t n = i < b.n_planes ? b.ctr[i].n_pool : 0;
        const uint32_t incl = wave_incl_scan(n);
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t off = s_carry, tot = 0;
        for (int k = 0; k < 16; ++k) { if (k < wv) off += s_w[k]; tot += s_w[k]; }
        const uint32_t mine = off + incl - n;
        if (i < b.n_planes) {
            PlaneCtr &c = b.ctr[i];
            c.cand_base_old = c.cand_base;
            c.cand_base = mine;
            if (c.pool_changed) {
                c.n_strong = 0; c.n_weak = 0;
                const uint32_t at = atomicAdd(n_redo, n);
                for (uint32_t k = 0; k < n; ++k) redo[at + k] = mine + k;
            }
        }
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    if (tid == 0) *b.total_cands = s_carry;
    for (int i = wv; i < b.n_planes; i += 16) {         // candidate -> plane, a wave per plane
        const uint32_t n = b.ctr[i].n_pool, base = b.ctr[i].cand_base;
        for (uint32_t k = lane; k < n; k += 64) b.cand_plane[base + k] = (uint16_t)i;
    }
}

__global__ __launch_bounds__(256) void k_cand_move(BatchDev b, const CandRec *__restrict__ from)
{
    static_assert(sizeof(CandRec) % 16 == 0, "candidate records are moved as 16-byte words");
    constexpr uint32_t Q = sizeof(CandRec) / 16;
    for (int pi = blockIdx.x; pi < b.n_planes; pi += gridDim.x) {
        const PlaneCtr &c = b.ctr[pi];
        if (c.pool_changed) continue;
        const uint4 *src = reinterpret_cast<const uint4 *>(from + c.cand_base_old);
        uint4       *dst = reinterpret_cast<uint4 *>(b.cands + c.cand_base);
        for (uint32_t k = threadIdx.x; k < c.n_pool * Q; k += blockDim.x) dst[k] = src[k];
    }
}

// The end of a call of a frame or two: the batch's counter block and its first candidate records written to page-locked host memory by ONE launch (the
// two copies they replace were two launches of the runtime's copy kernel with ~10 us of queue switching around each: a twentieth of such a call).
// to_ctr / to_cands are device-addressable host memory; n_words = the counter block in 16-byte words; at most cap_cands records.
__global__ __launch_bounds__(256) void k_results_to_host(const uint4 *__restrict__ ctr_block, uint4 *__restrict__ to_ctr, uint32_t n_words, const CandRec *__restrict__ cands,
                                                        CandRec *__restrict__ to_cands, uint32_t cap_cands, const uint32_t *__restrict__ total_cands)
{
    static_assert(sizeof(CandRec) % 16 == 0, "candidate records are moved as 16-byte words");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (uint32_t i = t; i < n_words; i += nt) to_ctr[i] = ctr_block[i];
    if (to_cands) {
        const uint32_t n = min(*total_cands, cap_cands) * (uint32_t)(sizeof(CandRec) / 16);
        const uint4   *src = reinterpret_cast<const uint4 *>(cands);
        uint4         *dst = reinterpret_cast<uint4 *>(to_cands);
        for (uint32_t i = t; i < n; i += nt) dst[i] = src[i];
    }
}

void launch_results_to_host(hipStream_t s, const void *ctr_block, void *to_ctr, size_t ctr_bytes, const CandRec *cands, CandRec *to_cands, uint32_t cap_cands,
                            const uint32_t *total_cands)
{
    hipLaunchKernelGGL(k_results_to_host, dim3(32), dim3(256), 0, s, static_cast<const uint4 *>(ctr_block), static_cast<uint4 *>(to_ctr), (uint32_t)((ctr_bytes + 15) / 16), cands,
                       to_cands, cap_cands, total_cands);
}

void launch_cand_reprefix(hipStream_t s, const BatchDev &b, const CandRec *from, uint32_t *redo, uint32_t *n_redo)
{
    hipLaunchKernelGGL(k_cand_reprefix, dim3(1), dim3(1024), 0, s, b, redo, n_redo);
    hipLaunchKernelGGL(k_cand_move, dim3(256), dim3(256), 0, s, b, from);
}
