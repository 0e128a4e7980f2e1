// er_tree_passes.inl -- part of er_kernels.hip (included there, inside namespace str_er; not a translation unit of its own): joining the tiles: k_group_merge, k_seam, strips, k_resolve, k_reduce, k_root / k_select / k_kept.
// ------------------------------------------------------------------------------------
// Component tree, part 1b: the tiles of a GROUP (GX x GY tiles) joined in LDS, in place.
//
// The global passes (k_seam, k_resolve, k_reduce) work on device-scope atomics, a few hundred picoseconds per record, and every
// node that touches any tile border goes through them.  Most of those nodes only touch a seam towards a NEIGHBOURING tile and are
// complete a tile or two further on.  One workgroup per group loads the records of its tiles (a few hundred: text-like 4 x 8 tiles
// ~ 430, noise 2 x 5 ~ 2300), joins the pixel pairs of the seams INSIDE the group with the same connect on LDS words, hands the
// statistics of unified nodes to their survivors (k_resolve's job), and folds every node whose component does not reach the group's
// OUTER border into its parent (k_reduce's job) -- what is left for the global passes are the nodes on the outer border: a quarter
// (4 x 8) or a third (2 x 5) of before.  Everything stays where it is: survivors keep their record, unified nodes are marked
// NODE_DEAD (k_resolve skips them), folded ones NODE_CLOSED (they never push again), so the seam map and every id stay valid.
// A group with more records than fit LDS is left alone (group_done stays 0: k_seam joins its inner seams as before).
// ------------------------------------------------------------------------------------
constexpr int GROUP_MAX_TILES = 64;

template <int CAP, int GROUP_THREADS>
__global__ __launch_bounds__(GROUP_THREADS) void k_group_merge(BatchDev b, const uint32_t *glist)
{
    const uint32_t group = glist ? glist[blockIdx.x] : blockIdx.x;       // (a launch per class of planes takes its groups from a list)
    __shared__ uint32_t s_par[CAP], s_cnt[CAP], s_nod[CAP], s_key[CAP], s_x0[CAP], s_y0[CAP], s_x1[CAP], s_y1[CAP];
    __shared__ uint32_t s_toff[GROUP_MAX_TILES + 1], s_tbase[GROUP_MAX_TILES];
    __shared__ uint32_t s_levels[8];
    const int       tid = threadIdx.x;
#ifdef STR_ER_WG_TRACE
#define GM_MARK(i) do { if (tid == 0 && blockIdx.x % 149u == 0u && blockIdx.x / 149u < 96u) g_wg_trace[288 + blockIdx.x / 149u][(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GM_MARK(i) do { } while (0)
#endif
    const int       GX = b.group_x, GY = b.group_y;
    const int       pi = b.group_plane[group];
    const PlaneDesc pd = b.planes[pi];
    const int       groups_x = (pd.tiles_x + GX - 1) / GX;
    const uint32_t  gl = group - pd.group_base;
    const int       tx0 = (int)(gl % (uint32_t)groups_x) * GX, ty0 = (int)(gl / (uint32_t)groups_x) * GY;
    const int       gw = min(GX, pd.tiles_x - tx0), gh = min(GY, pd.tiles_y - ty0);
    const int       nt = gw * gh;
    if (nt < 2) return;
    if (tid < 8) s_levels[tid] = 0;
    if (tid < 64) {
        // the tiles' record ranges: one lane per tile (first wave), offsets by a wave scan -- a lane walking the tiles one after the other
        // spends a global round trip per tile before anybody else can start
        uint32_t nb = 0, cnt = 0;
        if (tid < nt) {
            const uint32_t tile = pd.tile_base + (uint32_t)(ty0 + tid / gw) * pd.tiles_x + (uint32_t)(tx0 + tid % gw);
            nb = b.tile_nbase[tile];
            cnt = b.tile_nrec[tile];
        }
        const uint32_t incl = wave_incl_scan(cnt);
        const bool bad = __any(tid < nt && nb == NONE);
        if (tid < nt) { s_toff[tid] = incl - cnt; s_tbase[tid] = nb; }
        if (tid == nt - 1) s_toff[nt] = (!bad && incl <= (uint32_t)CAP) ? incl : NONE;
    }
    __syncthreads();
    const uint32_t N = s_toff[nt];
    if (N == NONE) {            // too many records for the table (or a tile without records of its own: an overflowing plane): left alone, and listed for k_seam_undone
        if (tid == 0) b.undone_list[atomicAdd(b.undone_count, 1u)] = group;
        return;
    }
    if (N == 0) return;
    GM_MARK(1);
    NodeRec *const nr = b.na.rec + pd.node_base;
    auto tile_of = [&](uint32_t i) -> int {         // (records are grouped by tile: the tile whose range holds local index i)
        int lo = 0, hi = nt - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_toff[mid] <= i) lo = mid; else hi = mid - 1; }
        return lo;
    };
    // ---- load; parent ids become local; only the sides on the group's OUTER border stay ----
    for (uint32_t i = tid; i < N; i += GROUP_THREADS) {
        const int      t = tile_of(i);
        const uint32_t gid = s_tbase[t] + (i - s_toff[t]);
        const uint4    a = reinterpret_cast<const uint4 *>(nr + gid)[0], c = reinterpret_cast<const uint4 *>(nr + gid)[1];
        const int      ix = t % gw, iy = t / gw;
        const uint32_t inner = (ix + 1 < gw ? NODE_SIDE_R : 0u) | (ix > 0 ? NODE_SIDE_L : 0u) | (iy + 1 < gh ? NODE_SIDE_B : 0u) | (iy > 0 ? NODE_SIDE_T : 0u);
        s_par[i] = a.x == NONE ? NONE : PAR_MAKE(PAR_LVL(a.x), PAR_ID(a.x) - s_tbase[t] + s_toff[t]);
        s_key[i] = a.y; s_cnt[i] = a.z; s_nod[i] = a.w & ~inner;
        s_x0[i] = c.x; s_y0[i] = c.y; s_x1[i] = c.z; s_y1[i] = c.w;
        atomicOr(&s_levels[(a.y >> 24) >> 5], 1u << ((a.y >> 24) & 31u));
    }
    __syncthreads();
    GM_MARK(2);
    // ---- the pixel pairs of the inner seams (same connect as node_connect, on LDS words) ----
    auto lfind = [&](uint32_t &a, uint32_t la) -> uint32_t {
        uint32_t wa = LD_WG(&s_par[a]);
        while (wa != NONE && PAR_LVL(wa) == la) {
            const uint32_t nx = PAR_ID(wa);
            const uint32_t w2 = LD_WG(&s_par[nx]);
            if (w2 != NONE && PAR_LVL(w2) == la) s_par[a] = w2;       // path halving, same node
            a = nx; wa = w2;
        }
        return wa;
    };
    auto lconnect = [&](uint32_t a, uint32_t bb) {
        uint32_t la = s_key[a] >> 24, lb = s_key[bb] >> 24;
        for (;;) {
            uint32_t wa = lfind(a, la);
            uint32_t wb = lfind(bb, lb);
            if (a == bb) return;
            if (la > lb || (la == lb && a < bb)) { uint32_t t; t = a; a = bb; bb = t; t = la; la = lb; lb = t; t = wa; wa = wb; wb = t; }
            if (la == lb || wa == NONE || PAR_LVL(wa) > lb) {
                const uint32_t old = atomicCAS(&s_par[a], wa, PAR_MAKE(lb, bb));
                if (old != wa) continue;
                if (wa == NONE) return;
            }
            a = PAR_ID(wa); la = PAR_LVL(wa);
        }
    };
    {
        const uint16_t *seam = b.seam + pd.seam_base;
        const uint32_t  n_hp = (uint32_t)((gh - 1) * gw * TILE_W), n_vp = (uint32_t)((gw - 1) * gh * TILE_H);
        const size_t    voff = 2u * (size_t)pd.w * (pd.tiles_y - 1);
        for (uint32_t p0 = 0; p0 < n_hp + n_vp; p0 += GROUP_THREADS) {
            const uint32_t p = p0 + (uint32_t)tid;
            uint32_t a = NONE, bb = NONE;
            if (p < n_hp) {
                const int iy = (int)(p / (uint32_t)(gw * TILE_W)), xg = (int)(p % (uint32_t)(gw * TILE_W));      // boundary under tile row iy; column inside the group
                const int x = tx0 * TILE_W + xg, j = ty0 + iy;
                if (x < pd.w) {
                    const uint32_t ea = seam[((size_t)j * 2) * pd.w + x], eb = seam[((size_t)j * 2 + 1) * pd.w + x];
                    if (ea != 0xFFFFu && eb != 0xFFFFu) { a = s_toff[iy * gw + xg / TILE_W] + ea; bb = s_toff[(iy + 1) * gw + xg / TILE_W] + eb; }
                }
            } else if (p < n_hp + n_vp) {
                const uint32_t q = p - n_hp;
                const int ix = (int)(q / (uint32_t)(gh * TILE_H)), yg = (int)(q % (uint32_t)(gh * TILE_H));
                const int y = ty0 * TILE_H + yg, k = tx0 + ix;
                if (y < pd.h) {
                    const uint32_t ea = seam[voff + ((size_t)k * 2) * pd.h + y], eb = seam[voff + ((size_t)k * 2 + 1) * pd.h + y];
                    if (ea != 0xFFFFu && eb != 0xFFFFu) { a = s_toff[(yg / TILE_H) * gw + ix] + ea; bb = s_toff[(yg / TILE_H) * gw + ix + 1] + eb; }
                }
            }
            // (neighbouring lanes very often carry the same pair -- a flat region along the seam: the first lane of such a run connects.
            //  Round 4, traced in place -- tools/dev_group_trace.py: this step is half of a workgroup's 40 k cycles -- and tried: connecting only the pairs
            //  that are a local minimum of max(level a, level b) along their tile's side -- the heavier of two neighbouring pairs is the heaviest edge of a
            //  cycle whose other edges stay; parity green, ~20x fewer connects -- and fetching four rounds of seam entries ahead: neither moved it, here or
            //  in k_seam.  The step is as long as its longest connects, the ones that merge two deep root paths; their number is not what costs.)
            {
                const uint32_t pa = __shfl_up(a, 1), pb = __shfl_up(bb, 1);
                const bool dup = (tid & 63) != 0 && pa == a && pb == bb;
                if (a != NONE && !dup) lconnect(a, bb);
            }
        }
    }
    __syncthreads();
    GM_MARK(3);
    // ---- unified nodes hand their own statistics to the surviving level root; the others get a canonical parent (k_resolve) ----
    for (uint32_t i = tid; i < N; i += GROUP_THREADS) {
        const uint32_t l = s_key[i] >> 24, w = s_par[i];
        if (w == NONE) continue;
        uint32_t q = PAR_ID(w);
        const uint32_t lq = PAR_LVL(w);
        for (;;) { const uint32_t w2 = s_par[q]; if (w2 == NONE || PAR_LVL(w2) != lq) break; q = PAR_ID(w2); }
        if (lq == l) {
            const uint32_t f = s_nod[i];
            atomicAdd(&s_cnt[q], s_cnt[i]);
            atomicAdd(&s_nod[q], (f & NODE_CNT) - 1u);          // (its folded descendants; the node itself is the survivor's)
            atomicOr(&s_nod[q], f & NODE_SIDES);
            atomicMin(&s_x0[q], s_x0[i]); atomicMin(&s_y0[q], s_y0[i]); atomicMax(&s_x1[q], s_x1[i]); atomicMax(&s_y1[q], s_y1[i]);
            atomicMin(&s_key[q], s_key[i]);                     // same level: the top byte is equal, the minimum is over the pixel index
            atomicOr(&s_nod[i], NODE_DEAD);
        }
        // (a unified node's parent word names its survivor from now on: whoever reads it later walks one hop)
        if (q != PAR_ID(w)) s_par[i] = PAR_MAKE(lq, q);
    }
    __syncthreads();
    GM_MARK(4);
    // ---- bottom-up over the levels: a node whose component reaches the group's outer border passes its sides on to its parent; any
    // other node is complete -- it adds its totals to its parent (k_reduce) and is closed ----
    for (int wd = 0; wd < 8; ++wd) {
        uint32_t pm = s_levels[wd];
        while (pm) {
            const uint32_t t = (uint32_t)wd * 32u + (uint32_t)__ffs((int)pm) - 1u;
            pm &= pm - 1u;
            for (uint32_t i = tid; i < N; i += GROUP_THREADS) {
                if ((s_key[i] >> 24) != t) continue;
                const uint32_t f = s_nod[i], w = s_par[i];
                if ((f & (NODE_DEAD | NODE_CLOSED)) || w == NONE) continue;
                const uint32_t q = PAR_ID(w);
                if (f & NODE_SIDES) atomicOr(&s_nod[q], f & NODE_SIDES);
                else {
                    atomicAdd(&s_cnt[q], s_cnt[i]);
                    atomicAdd(&s_nod[q], f & NODE_CNT);
                    atomicMin(&s_x0[q], s_x0[i]); atomicMin(&s_y0[q], s_y0[i]); atomicMax(&s_x1[q], s_x1[i]); atomicMax(&s_y1[q], s_y1[i]);
                    s_nod[i] = f | NODE_CLOSED;
                }
            }
            __syncthreads();
        }
    }
    GM_MARK(5);
    // ---- back to the records, in place (ids global again) ----
    for (uint32_t i = tid; i < N; i += GROUP_THREADS) {
        const int      t = tile_of(i);
        const uint32_t gid = s_tbase[t] + (i - s_toff[t]);
        uint32_t       w = s_par[i];
        if (w != NONE) { const uint32_t q = PAR_ID(w); const int tq = tile_of(q); w = PAR_MAKE(PAR_LVL(w), s_tbase[tq] + (q - s_toff[tq])); }
        uint4 *dst = reinterpret_cast<uint4 *>(nr + gid);
        dst[0] = make_uint4(w, s_key[i], s_cnt[i], s_nod[i]);
        dst[1] = make_uint4(s_x0[i], s_y0[i], s_x1[i], s_y1[i]);
    }
    if (tid == 0) b.group_done[group] = 1;
    GM_MARK(6);
}

// variant: LDS for 512 / 1024 / 2048 / 3072 records per group, 256 / 512 / 1024 lanes; glist / n: the groups of this launch (nullptr: all b.n_groups)
void launch_group_merge(hipStream_t s, const BatchDev &b, int variant, const uint32_t *glist, uint32_t n)
{
    if (!b.n_groups || b.group_x <= 0 || b.group_y <= 0 || b.group_x * b.group_y > GROUP_MAX_TILES) return;
    if (!glist) n = b.n_groups;
    if (!n) return;
    switch (variant) {
    case 0: hipLaunchKernelGGL((k_group_merge<512, 256>), dim3(n), dim3(256), 0, s, b, glist); break;
    case 1: hipLaunchKernelGGL((k_group_merge<1024, 256>), dim3(n), dim3(256), 0, s, b, glist); break;
    case 2: hipLaunchKernelGGL((k_group_merge<1024, 512>), dim3(n), dim3(512), 0, s, b, glist); break;
    case 3: hipLaunchKernelGGL((k_group_merge<2048, 512>), dim3(n), dim3(512), 0, s, b, glist); break;
    case 4: hipLaunchKernelGGL((k_group_merge<2048, 1024>), dim3(n), dim3(1024), 0, s, b, glist); break;
    case 5: hipLaunchKernelGGL((k_group_merge<3072, 1024>), dim3(n), dim3(1024), 0, s, b, glist); break;
    // 2528 records: 32 B each + the tile tables = 64 of the 1280-byte LDS granules, so TWO workgroups fit a CU
    case 6: hipLaunchKernelGGL((k_group_merge<2528, 1024>), dim3(n), dim3(1024), 0, s, b, glist); break;
    default: hipLaunchKernelGGL((k_group_merge<2528, 512>), dim3(n), dim3(512), 0, s, b, glist); break;
    }
}

// ------------------------------------------------------------------------------------
// Component tree, part 2: join the tile trees along every seam.  Same connect as in
// the tile kernel, on the global node arrays, with agent-scope atomics (the per-XCD L2s
// are not coherent with each other, so every access to `par` that may race goes through
// an agent-scope atomic).  Levels are immutable here and read with plain loads.
// ------------------------------------------------------------------------------------
#ifdef STR_ER_SEAM_PROF
// Developer aid: -DSTR_ER_SEAM_PROF counts the work of k_seam; read with str_er_debug_seam_counts().
__device__ unsigned long long g_seam_cnt[8];
#define SCNT(i, v) atomicAdd(&g_seam_cnt[i], (unsigned long long)(v))
extern "C" void str_er_debug_seam_counts(unsigned long long *out8, int reset)
{
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_seam_cnt), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seam_cnt), z, sizeof(z));
    }
}
#else
#define SCNT(i, v) do { } while (0)
#endif

// (node records are 8 dwords: the parent word of node i is at rec[i].par; `nr` below is the plane's first record)
__device__ __forceinline__ uint32_t node_find(NodeRec *nr, uint32_t &a, uint32_t la)
{
    uint32_t wa = LD_AGENT(&nr[a].par);
    while (wa != NONE && PAR_LVL(wa) == la) {
        SCNT(2, 1);
        const uint32_t nx = PAR_ID(wa);
        const uint32_t w2 = LD_AGENT(&nr[nx].par);
        if (w2 != NONE && PAR_LVL(w2) == la) ST_AGENT(&nr[a].par, w2);   // path halving, same node
        a = nx;
        wa = w2;
    }
    return wa;
}

__device__ __forceinline__ void node_connect(NodeRec *nr, uint32_t a, uint32_t b)
{
    uint32_t la = nr[a].key >> 24, lb = nr[b].key >> 24;       // levels are immutable: plain loads
    SCNT(0, 1);
    if (la == lb) SCNT(5, 1);
    for (;;) {
        SCNT(1, 1);
        uint32_t wa = node_find(nr, a, la);
        uint32_t wb = node_find(nr, b, lb);
        if (a == b) return;
        if (la > lb || (la == lb && a < b)) {
            uint32_t t;
            t = a; a = b; b = t;
            t = la; la = lb; lb = t;
            t = wa; wa = wb; wb = t;
        }
        if (la == lb || wa == NONE || PAR_LVL(wa) > lb) {
            const uint32_t old = atomicCAS(&nr[a].par, wa, PAR_MAKE(lb, b));
            SCNT(3, 1);
            if (old != wa) { SCNT(4, 1); continue; }
            if (wa == NONE) return;
            a = PAR_ID(wa);
            la = PAR_LVL(wa);
        } else {
            a = PAR_ID(wa);
            la = PAR_LVL(wa);
        }
    }
}

__global__ __launch_bounds__(SEAM_BLOCK) void k_seam(BatchDev b, int xcd_affine)
{
    // a block never straddles two planes: the host lists (plane, first pair) per block
    // Workgroups are dealt to the 8 XCDs round-robin; renumber them so that consecutive seam blocks (= one plane's
    // seams) run on ONE XCD and the plane's parent words stay in that XCD's L2 (for speed only: every access that can
    // race is agent-scope anyway).  Measured: this helps noise-like frames (seam 1.56 -> 1.17 ms per 8 frames), where every
    // node is touched by few connects, and hurts text-like ones (0.58 -> 0.86 ms per 32), where the connects of a plane pile up
    // on a few hot background nodes -- so the host asks for it together with the big size of the tile kernel.
    const uint32_t per = (b.n_seam_blocks + 7u) / 8u;
    const uint32_t vb = xcd_affine ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    if (vb >= b.n_seam_blocks) return;
    const int        pi = b.seam_block_plane[vb];
    const PlaneDesc &pd = b.planes[pi];
    const uint32_t   i = b.seam_block_first[vb] + threadIdx.x;
    uint32_t         na = NONE, nbn = NONE;
    if (i < pd.n_pairs) {
        // an entry of the seam map is the node's index inside its tile's records; the tile's first record is tile_nbase
        const uint16_t *seam = b.seam + pd.seam_base;
        const uint32_t *tnb = b.tile_nbase + pd.tile_base;
        uint32_t la, lb, ta, tb;
        // (a seam inside a group of tiles is not this kernel's: k_group_merge has joined it in LDS, or listed the group for k_seam_undone -- the host does
        // not even launch workgroups for the pair ranges that hold inner seams only, launch_seam's table)
        const uint32_t GX = (uint32_t)b.group_x, GY = (uint32_t)b.group_y;
        bool inner = false;
        if (i < pd.n_hpairs) {
            const uint32_t j = i / pd.w, x = i - j * pd.w;
            ta = j * pd.tiles_x + x / (uint32_t)TILE_W; tb = ta + pd.tiles_x;
            inner = GX && (j + 1u) % GY != 0u;
            la = lb = 0xFFFFu;
            if (!inner) {
                la = seam[((size_t)j * 2) * pd.w + x];
                lb = seam[((size_t)j * 2 + 1) * pd.w + x];
            }
        } else {
            const uint32_t i2 = i - pd.n_hpairs;
            const uint32_t k = i2 / pd.h, y = i2 - k * pd.h;
            const size_t   voff = 2u * (size_t)pd.w * (pd.tiles_y - 1);
            ta = (y / (uint32_t)TILE_H) * pd.tiles_x + k; tb = ta + 1;
            inner = GX && (k + 1u) % GX != 0u;
            la = lb = 0xFFFFu;
            if (!inner) {
                la = seam[voff + ((size_t)k * 2) * pd.h + y];
                lb = seam[voff + ((size_t)k * 2 + 1) * pd.h + y];
            }
        }
        if (la != 0xFFFFu && lb != 0xFFFFu) {
            const uint32_t ba = tnb[ta], bb = tnb[tb];
            if (ba != NONE && bb != NONE) { na = ba + la; nbn = bb + lb; }
        }
    }
    // neighbouring lanes very often carry the same pair (a flat region crossing the seam):
    // only the first lane of a run does the work.
    const uint32_t pa = __shfl_up(na, 1), pb = __shfl_up(nbn, 1);
    const bool     dup = (threadIdx.x & 63) != 0 && pa == na && pb == nbn;
    if (i < pd.n_pairs) SCNT(6, 1);
    // ... and the same pair keeps coming back further along the seam (background | speckle | background ...): a connect is
    // idempotent, so only the first lane of the block that brings a pair does it (open-addressing set in LDS).
    __shared__ unsigned long long s_seen[2 * SEAM_BLOCK];
    __shared__ uint32_t s_n;
    s_seen[threadIdx.x] = ~0ull; s_seen[threadIdx.x + SEAM_BLOCK] = ~0ull;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    bool mine = !(na == NONE || nbn == NONE || dup);
    if (mine) {
        const unsigned long long key = (unsigned long long)na | ((unsigned long long)nbn << 32);
        constexpr uint32_t HMASK = 2u * (uint32_t)SEAM_BLOCK - 1u;      // table of 2 * SEAM_BLOCK slots (a power of two)
        uint32_t h = ((na * 0x9E3779B1u ^ nbn * 0x85EBCA77u) >> 16) & HMASK;
        for (;;) {
            const unsigned long long old = atomicCAS(&s_seen[h], ~0ull, key);
            if (old == ~0ull) break;
            if (old == key) { mine = false; break; }
            h = (h + 1u) & HMASK;
        }
    }
    // The survivors (typically a tenth of the lanes, scattered over all 16 waves) are packed into the first waves: the other
    // waves retire at once and make room for the next workgroups, so more connects -- chains of dependent fabric round
    // trips -- are in flight per CU.
    // (the list lives in the set's memory -- the set is done with after a barrier: 16 KB of LDS per workgroup instead of 24, and it
    // is the LDS that limits how many workgroups, each down to a wave or two by now, share a CU)
    __syncthreads();
    uint32_t *const s_pa = reinterpret_cast<uint32_t *>(s_seen), *const s_pb = s_pa + SEAM_BLOCK;
    if (mine) { const uint32_t at = atomicAdd(&s_n, 1u); s_pa[at] = na; s_pb[at] = nbn; }
    __syncthreads();
    if (threadIdx.x >= s_n) return;
#ifdef STR_ER_ABL_SEAM
    return;
#endif
    node_connect(b.na.rec + pd.node_base, s_pa[threadIdx.x], s_pb[threadIdx.x]);
}

// The inner seams of the groups k_group_merge left alone (more records than its table holds: noise-like content; none on text-like batches), joined on the
// global records like any other seam.  A fixed grid walks the list: k_group_merge wrote it, so its length is only known on the device.
__global__ __launch_bounds__(SEAM_BLOCK) void k_seam_undone(BatchDev b)
{
    const uint32_t n = *b.undone_count;
    const int      GX = b.group_x, GY = b.group_y;
    for (uint32_t u = blockIdx.x; u < n; u += gridDim.x) {
        const uint32_t  g = b.undone_list[u];
        const int       pi = b.group_plane[g];
        const PlaneDesc &pd = b.planes[pi];
        const int       groups_x = (pd.tiles_x + GX - 1) / GX;
        const uint32_t  gl = g - pd.group_base;
        const int       tx0 = (int)(gl % (uint32_t)groups_x) * GX, ty0 = (int)(gl / (uint32_t)groups_x) * GY;
        const int       gw = min(GX, pd.tiles_x - tx0), gh = min(GY, pd.tiles_y - ty0);
        const uint16_t *seam = b.seam + pd.seam_base;
        const uint32_t *tnb = b.tile_nbase + pd.tile_base;
        const uint32_t  n_hp = (uint32_t)((gh - 1) * gw * TILE_W), n_vp = (uint32_t)((gw - 1) * gh * TILE_H);
        const size_t    voff = 2u * (size_t)pd.w * (pd.tiles_y - 1);
        for (uint32_t p0 = 0; p0 < n_hp + n_vp; p0 += (uint32_t)SEAM_BLOCK) {
            const uint32_t p = p0 + threadIdx.x;
            uint32_t na = NONE, nbn = NONE;
            if (p < n_hp) {
                const int iy = (int)(p / (uint32_t)(gw * TILE_W)), xg = (int)(p % (uint32_t)(gw * TILE_W));      // boundary under tile row iy; column inside the group
                const int x = tx0 * TILE_W + xg, j = ty0 + iy;
                if (x < pd.w) {
                    const uint32_t ea = seam[((size_t)j * 2) * pd.w + x], eb = seam[((size_t)j * 2 + 1) * pd.w + x];
                    const uint32_t ta = (uint32_t)j * pd.tiles_x + (uint32_t)(x / TILE_W), ba = tnb[ta], bb = tnb[ta + pd.tiles_x];
                    if (ea != 0xFFFFu && eb != 0xFFFFu && ba != NONE && bb != NONE) { na = ba + ea; nbn = bb + eb; }
                }
            } else if (p < n_hp + n_vp) {
                const uint32_t q = p - n_hp;
                const int ix = (int)(q / (uint32_t)(gh * TILE_H)), yg = (int)(q % (uint32_t)(gh * TILE_H));
                const int y = ty0 * TILE_H + yg, k = tx0 + ix;
                if (y < pd.h) {
                    const uint32_t ea = seam[voff + ((size_t)k * 2) * pd.h + y], eb = seam[voff + ((size_t)k * 2 + 1) * pd.h + y];
                    const uint32_t ta = (uint32_t)(y / TILE_H) * pd.tiles_x + (uint32_t)k, ba = tnb[ta], bb = tnb[ta + 1];
                    if (ea != 0xFFFFu && eb != 0xFFFFu && ba != NONE && bb != NONE) { na = ba + ea; nbn = bb + eb; }
                }
            }
            const uint32_t pa = __shfl_up(na, 1), pb = __shfl_up(nbn, 1);
            const bool     dup = (threadIdx.x & 63) != 0 && pa == na && pb == nbn;
            if (na != NONE && nbn != NONE && !dup) node_connect(b.na.rec + pd.node_base, na, nbn);
        }
    }
}

void launch_seam(hipStream_t s, const BatchDev &b, bool xcd_affine)
{
    if (b.n_seam_blocks) hipLaunchKernelGGL(k_seam, dim3((b.n_seam_blocks + 7u) / 8u * 8u), dim3(SEAM_BLOCK), 0, s, b, xcd_affine ? 1 : 0);
    if (b.n_groups && b.group_x > 0 && b.group_y > 0 && b.undone_list)
        hipLaunchKernelGGL(k_seam_undone, dim3(b.n_groups < 2048u ? b.n_groups : 2048u), dim3(SEAM_BLOCK), 0, s, b);
}

// ---- a plane put together from strips that other GPUs extracted ---------------------------------------------
// The records of a strip arrive with ids, keys and rows local to the strip (a strip is extracted like a plane of its own: the tile
// kernel knows nothing of the rows above it).  `delta` makes the ids those of the whole plane (the strip's records sit behind those
// of the strips above it), key_add = first row * width and y_add = first row put keys and boxes into the whole plane's coordinates.
__global__ __launch_bounds__(256) void k_rebase_records(NodeRec *rec, uint32_t *aux, uint32_t n, uint32_t delta, uint32_t key_add, uint32_t y_add, uint32_t w,
                                                        uint32_t h, uint32_t *bad)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        NodeRec r = rec[i];
        if (r.par != NONE) {
            // (the records came from another process: a parent outside the strip's records must not become a device index)
            if (PAR_ID(r.par) >= n) { atomicOr(bad, 1u); r.par = NONE; }
            else r.par = PAR_MAKE(PAR_LVL(r.par), PAR_ID(r.par) + delta);
        }
        // ... nor a box or a key outside the plane: k_kept hands the box to k_classify, which reads the plane's pixels over it, and the NMS tie
        // pass indexes its per-pixel stamps with the key.  A bad record is flagged (the merge fails with EFORMAT) and made harmless.
        const uint32_t key = (r.key & 0xFFFFFFu) + key_add;
        const bool     ok = r.x0 <= r.x1 && r.x1 < w && r.y0 <= r.y1 && r.y1 < h - min(h, y_add) && key < w * h;
        if (!ok) { atomicOr(bad, 1u); r.x0 = r.x1 = r.y0 = r.y1 = 0; r.key &= 0xFF000000u; }
        else { r.key += key_add; r.y0 += y_add; r.y1 += y_add; }      // (bits 0..23; the plane has fewer than 2^24 pixels, the level byte is not reached)
        rec[i] = r;
        aux[i] = 0;
    }
}
// The forest the strips' records form is checked before anything walks it (the records came from another process): a parent word must name a record
// of a level not below the node's own, carry that record's level, and the parent chains must END -- a cycle would keep every find, resolve and
// accumulate loop downstream spinning for ever (found by tools/san_fuzz.py: a damaged blob hung the merge).  Brent's cycle detection per record,
// O(chain length); an offending record is cut loose (parent NONE) and the flag raised: the merge then fails with EFORMAT and nothing hangs.
__global__ __launch_bounds__(256) void k_check_forest(NodeRec *rec, uint32_t n, uint32_t *bad)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t w = LD_AGENT(&rec[i].par);
        if (w == NONE) continue;
        bool ok = PAR_ID(w) < n && PAR_LVL(w) == (rec[PAR_ID(w)].key >> 24) && PAR_LVL(w) >= (rec[i].key >> 24);
        if (ok) {
            // Levels never fall along a parent chain (checked link by link above, for every record), so a cycle through record i stays at i's level:
            // the walk ends where the level rises.  Same-level chains of a real blob are redirects between the records of one flat region, one per tile it
            // crosses at most; a chain longer than any plane has tiles is damage as well -- and bounds the walk of a blob built to be slow.
            const uint32_t lv = rec[i].key >> 24;
            uint32_t tortoise = i, hare = PAR_ID(w), power = 1, lam = 1, hops = 0;
            while (PAR_LVL(w) == lv) {
                if (hare == tortoise || ++hops > (1u << 16)) { ok = false; break; }
                const uint32_t wh = LD_AGENT(&rec[hare].par);
                if (wh == NONE || PAR_ID(wh) >= n || PAR_LVL(wh) != lv) break;
                if (power == lam) { tortoise = hare; power *= 2; lam = 0; }
                hare = PAR_ID(wh);
                ++lam;
            }
        }
        if (!ok) { ST_AGENT(&rec[i].par, NONE); atomicOr(bad, 1u); }
    }
}
// ... and the pixel pairs across the cut between two strips are joined like any other seam: bot[x] / top[x] = strip-local node of pixel x of
// the last row above / the first row below the cut (NONE: a wall).  Neighbouring lanes very often carry the same pair (a flat region
// along the cut): only the first lane of such a run connects.
__global__ __launch_bounds__(256) void k_connect_cut(NodeRec *nr, const uint32_t *bot, const uint32_t *top, uint32_t w, uint32_t base_lo, uint32_t n_lo,
                                                     uint32_t base_hi, uint32_t n_hi, uint32_t *bad)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = NONE, b = NONE;
    if (x < w) {
        a = bot[x]; b = top[x];
        if ((a != NONE && a >= n_lo) || (b != NONE && b >= n_hi)) { atomicOr(bad, 1u); a = b = NONE; }
    }
    const uint32_t pa = __shfl_up(a, 1), pb = __shfl_up(b, 1);
    const bool dup = (threadIdx.x & 63) != 0 && pa == a && pb == b;
    if (a != NONE && b != NONE && !dup) node_connect(nr, a + base_lo, b + base_hi);
}
// node (strip-local record index) of every pixel of one border row of a strip: seam map entry + first record of the pixel's tile
__global__ __launch_bounds__(256) void k_strip_border_ids(const uint16_t *seam_row, const uint32_t *tile_nbase_row, int w, uint32_t *out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    const uint32_t e = seam_row[x], nb = tile_nbase_row[x / TILE_W];
    out[x] = (e == 0xFFFFu || nb == NONE) ? NONE : nb + e;
}
void launch_rebase_records(hipStream_t s, NodeRec *rec, uint32_t *aux, uint32_t n, uint32_t delta, uint32_t key_add, uint32_t y_add, uint32_t w, uint32_t h,
                           uint32_t *bad)
{
    if (n) hipLaunchKernelGGL(k_rebase_records, dim3((n + 255) / 256 < 4096u ? (n + 255) / 256 : 4096u), dim3(256), 0, s, rec, aux, n, delta, key_add, y_add, w, h, bad);
}
void launch_check_forest(hipStream_t s, NodeRec *plane_rec, uint32_t n, uint32_t *bad)
{
    if (n) hipLaunchKernelGGL(k_check_forest, dim3((n + 255) / 256 < 4096u ? (n + 255) / 256 : 4096u), dim3(256), 0, s, plane_rec, n, bad);
}
void launch_connect_cut(hipStream_t s, NodeRec *plane_rec, const uint32_t *bot, const uint32_t *top, uint32_t w, uint32_t base_lo, uint32_t n_lo, uint32_t base_hi,
                        uint32_t n_hi, uint32_t *bad)
{
    if (w) hipLaunchKernelGGL(k_connect_cut, dim3((w + 255) / 256), dim3(256), 0, s, plane_rec, bot, top, w, base_lo, n_lo, base_hi, n_hi, bad);
}
void launch_strip_border_ids(hipStream_t s, const uint16_t *seam_row, const uint32_t *tile_nbase_row, int w, uint32_t *out)
{
    if (w > 0) hipLaunchKernelGGL(k_strip_border_ids, dim3((w + 255) / 256), dim3(256), 0, s, seam_row, tile_nbase_row, w, out);
}

// ------------------------------------------------------------------------------------
// Part 3: per-node passes over the plane's records.  Grid = (NODE_BLOCKS, planes); a block strides over its plane's nodes.
// ------------------------------------------------------------------------------------
// (a plane gets workgroups of 256 lanes by its size -- PlaneDesc::nb_count of them, BatchDev::nb_plane lists the plane of every workgroup:
// the largest plane as many as the record counts of the previous batch ask for, a 240 x 135 pyramid level ONE; with the same number for
// every plane (round 2) a pyr3x8 batch launched 27 000 workgroups per pass, most of them for planes with a hundred records)

__device__ __forceinline__ uint32_t plane_nodes(const BatchDev &b, int pi)
{
    // a plane that ran out of records has holes in them: nothing downstream touches it (the host repeats the batch with more)
    return (b.ctr[pi].overflow & 8u) ? 0u : b.ctr[pi].n_nodes;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor(v, o));
    return v;
}

// What the lanes of a wave send to the same record is combined before it goes out -- the pieces a seam cut a big node into all hand over to ONE
// survivor, the children of a big node all count into (k_resolve) and push into (k_reduce) ONE parent, and without combining these few hot records
// serialise the kernels.  Until round 6 the combining was a loop over the wave's distinct targets, a leader lane and seven butterfly reductions per
// target: 64 passes for a wave of 64 different targets, in k_reduce each with two dependent trips to the coherence point inside (timing-only builds:
// the loops were 52 of k_resolve's 90 us and 45 of k_reduce's 165 us on one 1080p frame, 0.25 and 0.2 ms of a 4K batch).  Now every wave has a
// small table in LDS (128 slots keyed by the target): all lanes claim their target's slot with ONE compare-and-swap instruction (the first lane of a
// target becomes its owner, lanes whose slot belongs to another target keep their contribution to themselves), add into it with LDS atomics, the
// owners read the sums back, clear the slot and issue the global atomics -- owners and loners side by side in ONE set of instructions.  No barriers:
// a wave's LDS operations are carried out in program order.
#ifdef STR_ER_HOTSTAT
// developer counters (tools/dev_hotstat.py): pieces handed to one survivor, children pushed into one parent
__device__ uint32_t g_hand[1u << 23];
__device__ uint32_t g_hotstat[16];
#endif
constexpr int WTAB = 128;
struct WaveTab {
    uint32_t key[WTAB];                       // target record (plane-local id), NONE: free
    uint32_t c[WTAB], nd[WTAB], x0[WTAB], y0[WTAB], x1[WTAB], y1[WTAB], ky[WTAB], k[WTAB];
};
__device__ __forceinline__ void wtab_clear(WaveTab &t, int i)
{
    t.key[i] = NONE; t.c[i] = 0; t.nd[i] = 0; t.x0[i] = 0xFFFFFFFFu; t.y0[i] = 0xFFFFFFFFu; t.x1[i] = 0; t.y1[i] = 0; t.ky[i] = 0xFFFFFFFFu; t.k[i] = 0;
}
__device__ __forceinline__ uint32_t wtab_hash(uint32_t target) { return (target * 2654435761u) >> 25; }
static_assert(WTAB == 128, "wtab_hash takes the top 7 bits");

// Nodes that were unified into another node of the same level hand their own statistics to the surviving level root;
// surviving nodes get a canonical parent (the parent node's level root).  Every node that will push its totals to a parent
// -- open, alive, not a tree root -- is counted in the parent's dependency counter (aux).
__global__ __launch_bounds__(256) void k_resolve(BatchDev b)
{
    __shared__ WaveTab s_tab[4];
    const int       pi = b.nb_plane[blockIdx.x];
    const uint32_t  bi = blockIdx.x - b.planes[pi].nb_base, nbp = b.planes[pi].nb_count;     // this plane's workgroups: bi of nbp
    const uint32_t  n = plane_nodes(b, pi);
    NodeRec        *nr = b.na.rec + b.planes[pi].node_base;
    uint32_t       *aux = b.na.aux + b.planes[pi].node_base;
    uint32_t       *arr = b.na.arr + b.planes[pi].node_base;
    const int       lane = threadIdx.x & 63;
    WaveTab        &tab = s_tab[threadIdx.x >> 6];
    wtab_clear(tab, lane); wtab_clear(tab, lane + 64);
    for (uint32_t x0 = bi * blockDim.x + (threadIdx.x & ~63u); x0 < n; x0 += nbp * blockDim.x) {
        const uint32_t  x = x0 + (uint32_t)lane;
        uint32_t        push_to = NONE;         // the parent this node will push its totals to
        uint32_t        hand_to = NONE;         // the surviving level root this (unified) node hands its own statistics to
        uint32_t        c = 0, nd = 0, bx0 = 0xFFFFFFFFu, by0 = 0xFFFFFFFFu, bx1 = 0, by1 = 0, ky = 0xFFFFFFFFu;
        if (x < n) {
            arr[x] = 0;                          // k_reduce's arrival counter (every record is visited exactly once here)
            const NodeRec   me = nr[x];          // (plain loads: k_seam's writes are visible since the kernel boundary, and a parent
            const uint32_t  l = me.key >> 24;    //  word rewritten by a lane of THIS kernel points to the same node either way)
            const uint32_t  w = me.par;
            if (me.nod & NODE_DEAD) {
                // unified and handed over inside its group of tiles already (k_group_merge)
            } else if (w != NONE && PAR_LVL(w) == l) {
                uint32_t r = PAR_ID(w);
                for (;;) {
                    const uint32_t w2 = nr[r].par;
                    if (w2 == NONE || PAR_LVL(w2) != l) break;
                    r = PAR_ID(w2);
                }
                hand_to = r;
                c = me.cnt; nd = (me.nod & NODE_CNT) - 1u;      // (its folded descendants; the node itself is the survivor's)
                bx0 = me.x0; by0 = me.y0; bx1 = me.x1; by1 = me.y1; ky = me.key;
                atomicOr(&nr[x].nod, NODE_DEAD);
            } else if (w != NONE) {
                uint32_t       q = PAR_ID(w);
                const uint32_t lq = PAR_LVL(w);
                for (;;) {
                    const uint32_t w2 = nr[q].par;
                    if (w2 == NONE || PAR_LVL(w2) != lq) break;
                    q = PAR_ID(w2);
                }
                if (q != PAR_ID(w)) nr[x].par = PAR_MAKE(lq, q);
                if (!(me.nod & NODE_CLOSED)) push_to = q;                 // closed nodes never push (their totals are final)
            }
        }
        // (a record hands over or counts, never both; a target may be handed to by some lanes and counted into by others: one slot)
        const bool     hands = hand_to != NONE;
        const uint32_t tgt = hands ? hand_to : push_to;
#ifdef STR_ER_HOTSTAT
        if (hands) atomicAdd(&g_hand[(b.planes[pi].node_base + hand_to) & ((1u << 23) - 1u)], 1u);
#endif
        uint32_t       kk = push_to != NONE ? 1u : 0u;
        if (tgt != NONE) {
            const uint32_t h = wtab_hash(tgt);
            const uint32_t old = atomicCAS(&tab.key[h], NONE, tgt);
            if (old == NONE || old == tgt) {
                if (hands) {
                    atomicAdd(&tab.c[h], c); atomicAdd(&tab.nd[h], nd);
                    atomicMin(&tab.x0[h], bx0); atomicMin(&tab.y0[h], by0); atomicMax(&tab.x1[h], bx1); atomicMax(&tab.y1[h], by1);
                    atomicMin(&tab.ky[h], ky);
                } else atomicAdd(&tab.k[h], 1u);
                if (old == NONE) {               // the owner: the sums of everybody who shares the target (their LDS operations precede these reads)
                    c = tab.c[h]; nd = tab.nd[h]; bx0 = tab.x0[h]; by0 = tab.y0[h]; bx1 = tab.x1[h]; by1 = tab.y1[h]; ky = tab.ky[h]; kk = tab.k[h];
                    wtab_clear(tab, (int)h);
                } else { ky = 0xFFFFFFFFu; kk = 0; }      // a member: the owner sends its share
            }
            if (ky != 0xFFFFFFFFu) {             // something is handed over (a piece always brings its key)
                atomicAdd(&nr[tgt].cnt, c);
                if (nd) atomicAdd(&nr[tgt].nod, nd);
                atomicMin(&nr[tgt].x0, bx0); atomicMin(&nr[tgt].y0, by0); atomicMax(&nr[tgt].x1, bx1); atomicMax(&nr[tgt].y1, by1);
                atomicMin(&nr[tgt].key, ky);          // same level: the top byte is equal, the minimum is over the pixel index
            }
            if (kk) atomicAdd(&aux[tgt], kk);
        }
    }
}

void launch_resolve(hipStream_t s, const BatchDev &b)
{
    if (!b.n_planes) return;
    hipLaunchKernelGGL(k_resolve, dim3(b.n_node_blocks), dim3(256), 0, s, b);
}

// er_merge's accumulation (src/ER.cpp:153-165): every live open node adds its (final) totals to its parent.  One launch for
// the whole tree: aux[q] is the number of children of q that push (k_resolve; constant here), arr[q] how many of them have.  A node nobody pushes into
// is taken by the lane that meets it in the node sweep; a node with children belongs to the lane whose push completed the count, and that lane carries on
// towards the root.  Every word that changes -- totals, arrival counters -- is only ever touched with
// agent-scope atomics, which are performed at the device's coherence point (the per-XCD L2s are not coherent with each other), so
// no cache has to be written back or invalidated: the ordering "my pushes, then my arrival" / "the arrival, then my reads" only
// needs the lane to wait for its own outstanding operations (a workgroup-scope fence = s_waitcnt; an agent-scope acquire /
// release would write back and invalidate the whole L2 per node -- measured: 20 ms instead of 0.5 per batch).
// (Round 1 launched once per level: ~32 dependent launches per batch -- the whole cost of the step on small batches and on noise.)
#define RMW_AGENT(op, p, v) __hip_atomic_fetch_##op((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// The pushes are RETURNING atomics and node_arrive makes the counter decrement depend on what they returned: a returned value
// comes from the coherence point, so the push has been performed there before the decrement is even issued.  (Waiting for the
// acknowledgement of non-returning atomics -- s_waitcnt vmcnt(0), a workgroup-scope release -- is not enough: measured, subtree
// totals came out short now and then.)
// (pixels and nodes are neighbours in the record: one 64-bit add -- neither half can carry, a plane has fewer than 2^24 pixels and nodes)
__device__ __forceinline__ uint32_t node_push(NodeRec *dst, uint32_t c, uint32_t nd, uint32_t bx0, uint32_t by0, uint32_t bx1, uint32_t by1)
{
    const unsigned long long cn = RMW_AGENT(add, static_cast<unsigned long long *>(__builtin_assume_aligned(&dst->cnt, 8)), (unsigned long long)c | ((unsigned long long)nd << 32));
    uint32_t r = (uint32_t)cn | (uint32_t)(cn >> 32);
    r |= RMW_AGENT(min, &dst->x0, bx0); r |= RMW_AGENT(min, &dst->y0, by0);
    r |= RMW_AGENT(max, &dst->x1, bx1); r |= RMW_AGENT(max, &dst->y1, by1);
    return r;
}
// a node's totals once all its children have pushed: three 64-bit device-scope loads (pixels | nodes, the two corners of the box)
__device__ __forceinline__ void node_totals(const NodeRec *n, uint32_t &c, uint32_t &nodw, uint32_t &bx0, uint32_t &by0, uint32_t &bx1, uint32_t &by1)
{
    const unsigned long long cn = LD_AGENT(static_cast<const unsigned long long *>(__builtin_assume_aligned(&n->cnt, 8)));
    const unsigned long long a = LD_AGENT(static_cast<const unsigned long long *>(__builtin_assume_aligned(&n->x0, 8))), z = LD_AGENT(static_cast<const unsigned long long *>(__builtin_assume_aligned(&n->x1, 8)));
    c = (uint32_t)cn; nodw = (uint32_t)(cn >> 32);
    bx0 = (uint32_t)a; by0 = (uint32_t)(a >> 32); bx1 = (uint32_t)z; by1 = (uint32_t)(z >> 32);
}
// children done: `k` of them just pushed into a node (`pushed` = what node_push returned) that waits for `expect` of them in all; true if these were the
// last ones: the caller now owns the node.  (Until round 4 the children counted the parent's counter DOWN and the one that reached 0 then had to CLAIM the
// node with a compare-and-swap, against the sweep lane that might meet the 0 at the same moment: a fourth dependent trip to the coherence point per level of
// every chain.  Counting the arrivals up in a word of their own leaves k_resolve's count untouched: a childless node is simply one whose count is 0 -- the
// sweep takes it without any atomic --, and the arrival that completes the count owns the parent: totals, push, arrival -- three trips per level.)
__device__ __forceinline__ bool node_arrive(uint32_t *arrived, uint32_t expect, uint32_t k, uint32_t pushed)
{
    asm volatile("" : "+v"(k) : "v"(pushed) : "memory");
    uint32_t won = __hip_atomic_fetch_add(arrived, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + k == expect ? 1u : 0u;
    asm volatile("" : "+v"(won) : : "memory");       // the reads of the node's totals are issued after the arrival has returned
    return won != 0;
}

__global__ __launch_bounds__(256) void k_reduce(BatchDev b)
{
    __shared__ WaveTab s_tab[4];
    const int       pi = b.nb_plane[blockIdx.x];
    const uint32_t  bi = blockIdx.x - b.planes[pi].nb_base, nbp = b.planes[pi].nb_count;
    const uint32_t  n = plane_nodes(b, pi);
    NodeRec        *nr = b.na.rec + b.planes[pi].node_base;
    const uint32_t *aux = b.na.aux + b.planes[pi].node_base;       // (constant here: plain loads)
    uint32_t       *arr = b.na.arr + b.planes[pi].node_base;
    const int       lane = threadIdx.x & 63;
    WaveTab        &tab = s_tab[threadIdx.x >> 6];
    wtab_clear(tab, lane); wtab_clear(tab, lane + 64);
    for (uint32_t x0 = bi * blockDim.x + (threadIdx.x & ~63u); x0 < n; x0 += nbp * blockDim.x) {
        const uint32_t x = x0 + lane;
        bool     act = false;
        uint32_t q = NONE, c = 0, nd = 0, bx0 = 0xFFFFFFFFu, by0 = 0xFFFFFFFFu, bx1 = 0, by1 = 0;
        if (x < n) {
            const uint32_t w = nr[x].par, f = nr[x].nod;          // parent and flags are final since k_resolve
            // (a node nobody pushes into is ready; the others are taken by their last child)
            act = w != NONE && !(f & (NODE_DEAD | NODE_CLOSED)) && aux[x] == 0u;
#ifdef STR_ER_HOTSTAT
            {
                const uint32_t hd = g_hand[(b.planes[pi].node_base + x) & ((1u << 23) - 1u)], ch = aux[x];
                g_hand[(b.planes[pi].node_base + x) & ((1u << 23) - 1u)] = 0;
                if (hd) { atomicMax(&g_hotstat[0], hd); atomicAdd(&g_hotstat[1], hd); if (hd >= 16) atomicAdd(&g_hotstat[2], 1u); if (hd >= 128) atomicAdd(&g_hotstat[3], 1u); atomicAdd(&g_hotstat[9], 1u); }
                if (ch) { atomicMax(&g_hotstat[4], ch); atomicAdd(&g_hotstat[5], ch); if (ch >= 16) atomicAdd(&g_hotstat[6], 1u); if (ch >= 128) atomicAdd(&g_hotstat[7], 1u); if (ch >= 1024) atomicAdd(&g_hotstat[8], 1u); atomicAdd(&g_hotstat[10], 1u); }
                atomicAdd(&g_hotstat[11], 1u);
                if (hd >= 16) atomicAdd(&g_hotstat[12], hd);
                if (ch >= 16) atomicAdd(&g_hotstat[13], ch);
                if (ch == 1) atomicAdd(&g_hotstat[14], 1u);
            }
#endif
            if (act) {
                q = PAR_ID(w);
                node_totals(nr + x, c, nd, bx0, by0, bx1, by1);
                nd &= NODE_CNT;
            }
        }
        // first step: the lanes of the wave that share a parent combine in the wave's table (above): one set of atomics and one arrival per
        // distinct parent and wave -- the background node of a plane has thousands of such children --, all parents of the wave side by side
        uint32_t k = 0;
        if (act) {
            k = 1;
            const uint32_t h = wtab_hash(q);
            const uint32_t old = atomicCAS(&tab.key[h], NONE, q);
            if (old == NONE || old == q) {
                atomicAdd(&tab.c[h], c); atomicAdd(&tab.nd[h], nd);
                atomicMin(&tab.x0[h], bx0); atomicMin(&tab.y0[h], by0); atomicMax(&tab.x1[h], bx1); atomicMax(&tab.y1[h], by1);
                atomicAdd(&tab.k[h], 1u);
                if (old == NONE) {
                    c = tab.c[h]; nd = tab.nd[h]; bx0 = tab.x0[h]; by0 = tab.y0[h]; bx1 = tab.x1[h]; by1 = tab.y1[h]; k = tab.k[h];
                    wtab_clear(tab, (int)h);
                } else k = 0;                    // (the owner pushes this lane's share)
            }
        }
        // The lane carries (c, nd, box) -- the totals of k children -- into node g and, whenever that completes g, on towards the root.
        // A node that waits for exactly the children the lane brings (k of k: every node with ONE pushing child, most of a text-like tree) is
        // completed without any atomic: nobody else writes its record, so its totals are its own statistics (a plain load, requested a step ahead
        // together with its child count) plus what the lane carries, stored back as they are -- one trip to memory per level of such a chain
        // instead of three (push, arrival, totals).  Timing-only builds: the upward walks were 107 of the kernel's 165 us on one 1080p frame.
        if (k) {
            uint32_t g = q;
            uint32_t ex = aux[g];
            NodeRec  rec = nr[g];                // (par and flags are final since k_resolve; cnt / nod / box only used where nobody else pushes into g)
            for (;;) {
                const uint32_t w = rec.par;
                const uint32_t gn = w != NONE ? PAR_ID(w) : g;
                const uint32_t ex_n = aux[gn];   // the next node's, requested before this one's atomics are waited for
                const NodeRec  rec_n = nr[gn];
                if (ex == k) {
                    c += rec.cnt; nd += rec.nod & NODE_CNT;
                    bx0 = min(bx0, rec.x0); by0 = min(by0, rec.y0); bx1 = max(bx1, rec.x1); by1 = max(by1, rec.y1);
                    unsigned long long *d = reinterpret_cast<unsigned long long *>(__builtin_assume_aligned(&nr[g].cnt, 8));
                    d[0] = (unsigned long long)c | ((unsigned long long)((rec.nod & ~NODE_CNT) | nd) << 32);
                    d[1] = (unsigned long long)bx0 | ((unsigned long long)by0 << 32);
                    d[2] = (unsigned long long)bx1 | ((unsigned long long)by1 << 32);
                } else {
                    if (!node_arrive(&arr[g], ex, k, node_push(nr + g, c, nd, bx0, by0, bx1, by1))) break;
                    if (w == NONE) break;
                    node_totals(nr + g, c, nd, bx0, by0, bx1, by1);
                    nd &= NODE_CNT;
                }
                if (w == NONE || (rec.nod & (NODE_DEAD | NODE_CLOSED))) break;       // a tree root (or a node that never pushes)
                g = gn; ex = ex_n; rec = rec_n; k = 1;
            }
        }
    }
}

#ifdef STR_ER_HOTSTAT
extern "C" __attribute__((visibility("default"))) int str_er_debug_hotstat(uint32_t *out)
{
    uint32_t z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hotstat), sizeof z) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_hotstat), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
void launch_reduce(hipStream_t s, const BatchDev &b)
{
    if (!b.n_planes) return;
    hipLaunchKernelGGL(k_reduce, dim3(b.n_node_blocks), dim3(256), 0, s, b);
}

// Root of the tree that holds the flood's start pixel (er_stack.back(), src/ER.cpp:346).
// If the start pixel and its two candidates are all at the sentinel level the reference
// returns one childless node {level hi, area 2, bound (0,0,1,1)}.
__global__ void k_root(BatchDev b, DetectParams prm)
{
    const int pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= b.n_planes) return;
    PlaneCtr       &c = b.ctr[pi];
    const PlaneDesc &pd = b.planes[pi];
    const NodeRec  *nr = b.na.rec + pd.node_base;
    uint32_t        x = c.start_node;
    if (x == NONE) {
        const size_t kb = pd.kept_base;
        b.ka.node[kb] = NONE;
        b.ka.key[kb] = 0;
        b.ka.area[kb] = 2;
        b.ka.parent[kb] = 0;
        b.ka.box[4 * kb + 0] = 0; b.ka.box[4 * kb + 1] = 0; b.ka.box[4 * kb + 2] = 1; b.ka.box[4 * kb + 3] = 1;
        b.ka.level[kb] = (uint8_t)prm.hi;
        c.root_node = NONE;
        c.n_kept = 1;
        c.root_slot = 0;
        c.n_created = 1;
        c.max_level = prm.hi;
        return;
    }
    for (;;) { const uint32_t w = nr[x].par; if (w == NONE || PAR_LVL(w) != (nr[x].key >> 24)) break; x = PAR_ID(w); }
    for (;;) { const uint32_t w = nr[x].par; if (w == NONE) break; x = PAR_ID(w); }
    c.root_node = x;
    c.n_created = nr[x].nod & NODE_CNT;
    c.max_level = nr[x].key >> 24;
}

void launch_root(hipStream_t s, const BatchDev &b, const DetectParams &p)
{
    if (!b.n_planes) return;
    hipLaunchKernelGGL(k_root, dim3((b.n_planes + 63) / 64), dim3(64), 0, s, b, p);
}

// Pruning (src/ER.cpp:167-180): a node survives iff area > MIN_AREA (area = pixels +
// nodes of the subtree, because ER::ER starts area at 1), plus the root.  Nodes of other
// trees (regions sealed off by sentinel-level pixels) were never visited by the flood.
__global__ __launch_bounds__(256) void k_select(BatchDev b, DetectParams prm)
{
    const int       pi = b.nb_plane[blockIdx.x];
    const uint32_t  bi = blockIdx.x - b.planes[pi].nb_base, nbp = b.planes[pi].nb_count;
    PlaneCtr       &c = b.ctr[pi];
    const uint32_t  root = c.root_node;
    if (root == NONE) return;
    const PlaneDesc &pd = b.planes[pi];
    const uint32_t  n = plane_nodes(b, pi);
    const NodeRec  *nr = b.na.rec + pd.node_base;
    uint32_t       *aux = b.na.aux + pd.node_base;
    const bool      walls = c.n_walls != 0;
    for (uint32_t x = bi * blockDim.x + threadIdx.x; x < n; x += nbp * blockDim.x) {
        const uint32_t f = nr[x].nod;
        if (f & NODE_DEAD) continue;
        if (x != root) {
            const uint32_t area = nr[x].cnt + (f & NODE_CNT);
            if ((int64_t)area <= (int64_t)prm.min_area) continue;
            if (walls) {
                uint32_t y = x;
                for (;;) { const uint32_t w = nr[y].par; if (w == NONE) break; y = PAR_ID(w); }
                if (y != root) continue;
            }
        }
        const uint32_t slot = atomicAdd(&c.n_kept, 1u);
        if (slot < pd.kept_cap) {
            b.ka.node[pd.kept_base + slot] = x;
            aux[x] = slot;
        } else {
            atomicOr(&c.overflow, 1u);
        }
    }
}

void launch_select(hipStream_t s, const BatchDev &b, const DetectParams &p)
{
    if (!b.n_planes) return;
    hipLaunchKernelGGL(k_select, dim3(b.n_node_blocks), dim3(256), 0, s, b, p);
}

// Kept-node records (flat form of struct ER, inc/ER.h:42-80).
__global__ __launch_bounds__(256) void k_kept(BatchDev b, DetectParams prm)
{
    const int       pi = blockIdx.y;
    PlaneCtr       &c = b.ctr[pi];
    if (c.root_node == NONE) return;
    const PlaneDesc &pd = b.planes[pi];
    if (c.n_kept > pd.kept_cap) return;         // table overflow (flagged by k_select): parents may be missing -- the host grows the table or fails
    const uint32_t  n = c.n_kept;
    const size_t    kb = pd.kept_base;
    const NodeRec  *nr = b.na.rec + pd.node_base;
    const uint32_t *aux = b.na.aux + pd.node_base;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const uint32_t x = b.ka.node[kb + s];
        const NodeRec  r = nr[x];
        b.ka.key[kb + s] = r.key & 0xFFFFFFu;
        b.ka.area[kb + s] = r.cnt + (r.nod & NODE_CNT);
        b.ka.level[kb + s] = (uint8_t)(r.key >> 24);
        b.ka.box[4 * (kb + s) + 0] = (uint16_t)r.x0;
        b.ka.box[4 * (kb + s) + 1] = (uint16_t)r.y0;
        b.ka.box[4 * (kb + s) + 2] = (uint16_t)(r.x1 - r.x0 + 1);
        b.ka.box[4 * (kb + s) + 3] = (uint16_t)(r.y1 - r.y0 + 1);
        if (x == c.root_node) {
            b.ka.parent[kb + s] = (int32_t)s;
            c.root_slot = s;
        } else {
            b.ka.parent[kb + s] = (int32_t)aux[PAR_ID(r.par)];
        }
    }
}

void launch_kept(hipStream_t s, const BatchDev &b, const DetectParams &p)
{
    if (!b.n_planes) return;
    hipLaunchKernelGGL(k_kept, dim3(16, b.n_planes), dim3(256), 0, s, b, p);
}
