// er_tile_tree.inl -- part of er_kernels.hip (included there, inside namespace str_er; not a translation unit of its own): the component tree of one 64 x 32 tile (k_tile_tree).
// ------------------------------------------------------------------------------------
// Component tree, part 1: one workgroup builds the tree of one 64x32 tile in LDS.
//
// LDS state per pixel p:  s_lev[p]  quantised level (0xFFFF = wall: outside the image or
//                                   at the sentinel level the reference never floods)
//                         s_par[p]  NONE, or (level of q << 16 | q): q is a pixel of the
//                                   same node (same level, q < p) or of the parent node.
// A pixel whose s_par is NONE or points to a higher level is the "level root" of its
// node; the level root of node (t, C) ends up being the smallest-index pixel of level
// t in C, which is also the node's canonical key.
// ------------------------------------------------------------------------------------
constexpr uint32_t WALL = 0xFFFFu;
// Global parent words: NONE, or (level of parent << 24 | parent id) -- ids are < 2^24 (planes are
// limited to 2^24 pixels), and having the level in the word saves the dependent lvl[] load on every
// hop of a find.
#define PAR_ID(w)  ((w) & 0xFFFFFFu)
#define PAR_LVL(w) ((w) >> 24)
#define PAR_MAKE(l, id) (((uint32_t)(l) << 24) | (uint32_t)(id))

// LDS placement of pixel p: one unused word after every 32 pixels ("skewed"), slot = p + p / 32.  When every
// lane touches the k-th of its 8 consecutive pixels (p = 8 * lane + k) a half-wave then hits 32 different
// banks instead of 4, and -- unlike a transposed layout -- the map is monotonic: slots compare like pixels
// (the canonical level root stays "the smallest one"), a lane's 8 pixels are 8 consecutive slots, the pixel
// below is always +66.  So the whole kernel works in slot numbers (all stored pointers are slots) and only
// converts back, SLOT_PIXEL, where a pixel position is needed.
constexpr int TILE_SLOTS = TILE_PX + TILE_PX / 32;      // 2112
constexpr int TILE_WS = TILE_W + TILE_W / 32;           // 66: slot distance of vertically adjacent pixels
#define SLOT_PIXEL(q) ((q) - (q) / 33u)
#define LX(q)  (q)
#define OWN(k) (p0 + (uint32_t)(k))

// Developer aid: build with -DSTR_ER_PHASE_PROF to accumulate per-phase cycle counts of
// k_tile_tree (lane 0 of every block) into g_tile_phase[]; read with str_er_debug_phase_cycles().
#ifdef STR_ER_PHASE_PROF
__device__ unsigned long long g_tile_phase[16];
#define PHASE_MARK(i)                                                                  \
    do {                                                                               \
        if (threadIdx.x == 0) {                                                        \
            const unsigned long long t_now = wall_clock64();                           \
            atomicAdd(&g_tile_phase[i], t_now - t_prev);                               \
            t_prev = t_now;                                                            \
        }                                                                              \
    } while (0)
#define PHASE_INIT() unsigned long long t_prev = wall_clock64()
#ifdef STR_ER_COUNT_PROF
#define CNT(i, v) atomicAdd(&g_tile_phase[8 + (i)], (unsigned long long)(v))
#else
#define CNT(i, v) do { } while (0)
#endif
#else
#define CNT(i, v) do { } while (0)
#if defined(STR_ER_WG_TRACE)
// Developer aid: -DSTR_ER_WG_TRACE: every 997th workgroup of k_tile_tree notes s_memtime at its start (slot 15) and behind every phase, lane 0 only, straight
// into g_wg_trace (one 8-byte store each, no atomics, nothing kept in registers or LDS: occupancy as in the product); tools/dev_wg_trace.py prints the phases'
// share of a workgroup's LIFETIME -- which, the kernel needing every workgroup a CU can hold, is what its throughput follows.
__device__ unsigned long long g_wg_trace[512][16];
#define PHASE_INIT() const bool tr_on = threadIdx.x == 0 && blockIdx.x % 997u == 0u && blockIdx.x / 997u < 512u; \
    if (tr_on) g_wg_trace[blockIdx.x / 997u][15] = __builtin_amdgcn_s_memtime()
#define PHASE_MARK(i) do { if (tr_on) g_wg_trace[blockIdx.x / 997u][(i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" void str_er_debug_wg_trace(unsigned long long *out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_trace), sizeof(unsigned long long) * 512 * 16);
    if (reset) { static unsigned long long z[512 * 16]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace), z, sizeof(z)); }
}
#elif defined(STR_ER_STOP_AFTER)
// Developer aid: -DSTR_ER_STOP_AFTER=n ends k_tile_tree after phase n (0 load .. 6 seam map) so that the cost of each
// phase can be read off as a difference of kernel times; only meaningful with STR_ER_DEBUG_TILE_ONLY=1 (str_er_api.cpp).
#define PHASE_MARK(i) do { if ((i) == STR_ER_STOP_AFTER) return; } while (0)
#define PHASE_INIT() do { } while (0)
#else
#define PHASE_MARK(i) do { } while (0)
#define PHASE_INIT() do { } while (0)
#endif
#endif

// Level root of pixel a (level la), with path halving: every same-level hop re-points the
// pixel at its grandparent.  Only non-roots are rewritten, and only with another pixel of
// the same node, so racing with the CAS in connect_pass (which targets level roots) is benign.
__device__ __forceinline__ uint32_t tile_find(uint32_t *s_par, uint32_t &a, uint32_t la)
{
    uint32_t wa = LD_WG(&s_par[LX(a)]);
    // (NONE reads as level 0xFFFF, which no pixel has: the level test covers it)
    while ((wa >> 16) == la) {
        const uint32_t nx = wa & 0xFFFFu;
        const uint32_t w2 = LD_WG(&s_par[LX(nx)]);
        if ((w2 >> 16) == la) s_par[LX(a)] = w2;
        a = nx;
        wa = w2;
        CNT(2, 1);
    }
    return wa;
}

// Join pixels a and b (4-neighbours, both not walls): when the edge is done, the root paths of a and b are merged into one path
// sorted by level.  Lock-free; every change is one CAS on the parent word of a level root, conditional on the value that was read.
// This is ONE pass -- find both level roots, then link the lower one under the other, or climb -- written with a single branch
// (around the CAS) besides the finds, everything else is selects: the kernel is bound by instruction issue, scalar
// bookkeeping of divergent branches included (round 2: 3.19 -> 3.11 ms per 32 text frames, 9.8 -> 8.5 on noise, against the same
// pass as nested ifs).  Returns whether the edge still needs passes.
__device__ __forceinline__ bool connect_pass(uint32_t *s_par, uint32_t &a, uint32_t &b, uint32_t &la, uint32_t &lb)
{
    CNT(1, 1);
    uint32_t       wa = tile_find(s_par, a, la);
    const uint32_t wb = tile_find(s_par, b, lb);
    const bool     same = a == b;
    {
        // (selects, not a branch around moves: 5 vector instructions instead of 9 + the scalar bookkeeping of the branch)
        const bool     sw = la > lb || (la == lb && a < b);
        const uint32_t a2 = sw ? b : a, b2 = sw ? a : b, la2 = sw ? lb : la, lb2 = sw ? la : lb;
        wa = sw ? wb : wa;
        a = a2; b = b2; la = la2; lb = lb2;
    }
    // now a must end up below b: either in the same node (equal levels, a > b) or as a descendant.  If a's current parent is
    // higher than b (or there is none: NONE reads as level 0xFFFF), b slots in between; otherwise climb
    const bool link = !same && (la == lb || (wa >> 16) > lb);
    uint32_t   old = wa;
    if (link) { old = atomicCAS(&s_par[LX(a)], wa, (lb << 16) | b); CNT(3, 1); }
    const bool ok = old == wa;                  // (a lane that climbs has ok = true as well)
    // linked under b, or climbing: carry on with a's (former) parent; a lost CAS repeats the pass with the same pair
    if (!same && ok) { a = wa & 0xFFFFu; la = wa >> 16; }
    return !(same || (link && ok && wa == NONE));
}

// The edge list of the connect round (see k_tile_tree): entry = slot of the edge's second (horizontal: the edge is (left of p, p),
// one slot back, two across the unused word after every 32 pixels: bit 14) or first (vertical, bit 15 set: (p, pixel below p)) pixel.  A connect
// takes anything from one to a dozen passes, so a lane takes its next edge as soon as it is done with one.  Every WAVE owns a contiguous
// quarter of the list and hands its entries, in list order, to whichever of its lanes are idle (ballot + mbcnt: no LDS traffic, no
// barrier): a wave leaves after ~(passes of its quarter) / 64 iterations instead of after the passes of its unluckiest lane.
// (Other hand-out schemes and pass orders, measured and not adopted: NOTES.md "Connect loop")
#ifdef STR_ER_CONNECT_CNT
// Developer aid (-DSTR_ER_CONNECT_CNT, tools/dev_connect_cnt.py): per-wave counts of the hand-written connect loop.  Round 4, text-like luma tile:
// 201 edges, 17.4 iterations, 33 + 19 rounds of the two walking loops per wave -- a round is paid by the whole wave whenever one lane is not at
// its level root yet, 2.5 rounds per iteration, about as many instructions as the passes themselves.  Naming the edges by RUN HEADS instead of
// pixels (32-bit entries; the listing lane knows its pieces' heads, the run coming in from the left and the pieces of the lane below) brought that
// to 27 + 17 and cost more in the listing loops than it saved: 1.832 against 1.812 ms per 32 text frames, noise 5.15 against 4.70 -- not adopted.
__device__ unsigned long long g_connect_cnt[8];      // waves, loop iterations, walk rounds (a), (b), edges
extern "C" void str_er_debug_connect_counts(unsigned long long *out8, int reset)
{
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_connect_cnt), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_connect_cnt), z, sizeof(z)); }
}
#endif
// LDS byte address of an object in shared memory (what a ds_* instruction takes)
template <class T>
__device__ __forceinline__ uint32_t lds_addr(const T *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) T *)p;
}

__device__ __forceinline__ void tile_connect_list(uint32_t *s_par, const uint16_t *s_lev, const uint16_t *s_elist, uint32_t n_edges_)
{
    constexpr uint32_t NW = TILE_THREADS / 64;
    const uint32_t n_edges = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_edges_);
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t w0 = (wv * n_edges) / NW, m = ((wv + 1u) * n_edges) / NW - w0;     // the wave's share: entries [w0, w0 + m)
#ifndef STR_ER_CONNECT_CXX
    // ---- the loop below, written by hand (round 4).  The kernel runs at the knee of vector issue (a SIMD's 8 waves keep its vector unit ~85 % busy:
    // tools/issue_caps.hip -- plain 32-bit vector instructions issue at 0.22-0.24 per cycle and SIMD, only v_mov / add / sub / and / or / xor / lshrrev at
    // 0.37-0.41) and of the scalar unit, and the compiler's loop spent 49 vector + ~45 scalar instructions per iteration: lane masks kept as 0 / 1 in
    // vector registers and compared back, exec saved / restored / branched around every `if`.  Here: the two ends of an edge are KEYS,
    // (level << 16) | slot -- the very word a parent pointer holds --, one compare of the keys with their slot halves flipped orders them, the CAS
    // writes the other key as it is, level tests are 16-bit sub-word compares (SDWA), the lanes' states are masks in scalar registers, and exec is
    // simply set: 19 vector instructions per pass + 14 for a hand-out.
    // Needs s_par at LDS address 0 (a key's low half << 2 is then the address); checked here, folded away by the compiler.
    static_assert(TILE_SLOTS <= 4096 && TILE_WS == 66, "edge entry: 12-bit slot, codes for + 1 / + 2 / + 66");
    if (lds_addr(s_par) != 0u) __builtin_trap();
    {
        uint32_t ka, kb, wa, wb, aa, ab, t0, t1, t2;
        unsigned long long busy, m1, m2, m3, m4, sx;
        uint32_t cur, tmp;
#ifdef STR_ER_CONNECT_CNT
        uint32_t n_it = 0, n_ha = 0, n_hb = 0;      // developer aid: loop iterations / rounds of the two walks, per wave (tools/dev_connect_cnt.py)
#endif
        asm volatile(
            "s_mov_b64 %[sx], exec\n"
            "s_mov_b64 %[busy], 0\n"
            "s_mov_b32 %[cur], 0\n"
            "LOOP_%=:\n"
#ifdef STR_ER_CONNECT_CNT
            "s_add_u32 %[n_it], %[n_it], 1\n"
#endif
            // ---- hand the next entries of the wave's share to its idle lanes (list order, ballot + mbcnt)
            "s_cmp_ge_u32 %[cur], %[m]\n"
            "s_cbranch_scc1 NOHAND_%=\n"
            "s_not_b64 vcc, %[busy]\n"                     // idle lanes (SCC: any)
            "s_cbranch_scc0 NOHAND_%=\n"
            "v_mbcnt_lo_u32_b32 %[t0], vcc_lo, 0\n"
            "v_mbcnt_hi_u32_b32 %[t0], vcc_hi, %[t0]\n"
            "v_add_u32 %[t0], %[cur], %[t0]\n"
            "v_cmp_gt_u32_e64 %[m1], %[m], %[t0]\n"
            "s_and_b64 %[m1], %[m1], vcc\n"                 // the lanes that take an entry
            "s_bcnt1_i32_b64 %[tmp], vcc\n"
            "s_add_u32 %[cur], %[cur], %[tmp]\n"
            "s_or_b64 %[busy], %[busy], %[m1]\n"
            "s_mov_b64 exec, %[m1]\n"
            "v_lshl_add_u32 %[t1], %[t0], 1, %[elist]\n"
            "ds_read_u16 %[t2], %[t1]\n"
            "s_waitcnt lgkmcnt(0)\n"
            // entry: slot of the edge's first pixel | code << 12; second pixel = first + 1 (code 0), + 2 (1: across the unused word), + 66 (2: below)
            "v_and_b32 %[aa], 0xfff, %[t2]\n"
            "v_lshrrev_b32 %[t0], 12, %[t2]\n"
            "v_lshrrev_b32 %[t1], 1, %[t0]\n"
            "v_mad_u32_u24 %[t0], %[t1], 63, %[t0]\n"
            "v_add3_u32 %[ab], %[aa], %[t0], 1\n"
            "v_lshl_add_u32 %[t0], %[aa], 1, %[lev]\n"
            "v_lshl_add_u32 %[t1], %[ab], 1, %[lev]\n"
            "ds_read_u16 %[t0], %[t0]\n"
            "ds_read_u16 %[t1], %[t1]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_lshl_or_b32 %[ka], %[t0], 16, %[aa]\n"
            "v_lshl_or_b32 %[kb], %[t1], 16, %[ab]\n"
            "NOHAND_%=:\n"
            "s_cmp_eq_u64 %[busy], 0\n"
            "s_cbranch_scc1 DONE_%=\n"
            "s_mov_b64 exec, %[busy]\n"
            // ---- one pass: the level roots of both ends ...
            "v_lshlrev_b32_sdwa %[aa], %[two], %[ka] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
            "v_lshlrev_b32_sdwa %[ab], %[two], %[kb] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
            "ds_read_b32 %[wa], %[aa]\n"
            "ds_read_b32 %[wb], %[ab]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_cmp_eq_u32_sdwa %[m1], %[wa], %[ka] src0_sel:WORD_1 src1_sel:WORD_1\n"      // same level: not the root yet (NONE reads as level 0xFFFF)
            "v_cmp_eq_u32_sdwa %[m2], %[wb], %[kb] src0_sel:WORD_1 src1_sel:WORD_1\n"
            "s_or_b64 %[m3], %[m1], %[m2]\n"
            "s_cbranch_scc0 ROOTS_%=\n"
            // (walks with path halving: a pixel is re-pointed at its grandparent while the grandparent is of the same level; only non-roots are
            // rewritten, and only with a pixel of the same node, so a race with the CAS below -- which targets roots -- is benign)
            "s_mov_b64 exec, %[m1]\n"
            "s_cbranch_execz HOPB_%=\n"
            "HOPA_%=:\n"
#ifdef STR_ER_CONNECT_CNT
            "s_add_u32 %[n_ha], %[n_ha], 1\n"
#endif
            "v_lshlrev_b32_sdwa %[t0], %[two], %[wa] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
            "ds_read_b32 %[t1], %[t0]\n"
            "v_mov_b32 %[ka], %[wa]\n"
            "v_mov_b32 %[t2], %[aa]\n"
            "v_mov_b32 %[aa], %[t0]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_mov_b32 %[wa], %[t1]\n"
            "v_cmp_eq_u32_sdwa vcc, %[t1], %[ka] src0_sel:WORD_1 src1_sel:WORD_1\n"
            "s_and_b64 exec, exec, vcc\n"
            "ds_write_b32 %[t2], %[t1]\n"
            "s_cbranch_execnz HOPA_%=\n"
            "HOPB_%=:\n"
            "s_mov_b64 exec, %[m2]\n"
            "s_cbranch_execz HOPX_%=\n"
            "HOPBL_%=:\n"
#ifdef STR_ER_CONNECT_CNT
            "s_add_u32 %[n_hb], %[n_hb], 1\n"
#endif
            "v_lshlrev_b32_sdwa %[t0], %[two], %[wb] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
            "ds_read_b32 %[t1], %[t0]\n"
            "v_mov_b32 %[kb], %[wb]\n"
            "v_mov_b32 %[t2], %[ab]\n"
            "v_mov_b32 %[ab], %[t0]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_mov_b32 %[wb], %[t1]\n"
            "v_cmp_eq_u32_sdwa vcc, %[t1], %[kb] src0_sel:WORD_1 src1_sel:WORD_1\n"
            "s_and_b64 exec, exec, vcc\n"
            "ds_write_b32 %[t2], %[t1]\n"
            "s_cbranch_execnz HOPBL_%=\n"
            "HOPX_%=:\n"
            "s_mov_b64 exec, %[busy]\n"
            "ROOTS_%=:\n"
            // ... then the lower root (lower level; same level: larger slot -- the smallest pixel stays the node's root) goes under the other one, or climbs
            // (gfx940 family: a vector instruction that reads an SGPR / VCC written by a vector compare needs two instructions in between)
            "v_xor_b32 %[t0], 0xffff, %[ka]\n"
            "v_xor_b32 %[t1], 0xffff, %[kb]\n"
            "v_cmp_gt_u32 vcc, %[t0], %[t1]\n"              // a is the higher one: swap
            "v_cmp_ne_u32_e64 %[m1], %[ka], %[kb]\n"        // not yet one node
            "s_nop 0\n"
            "v_cndmask_b32 %[t2], %[ka], %[kb], vcc\n"      // lo
            "v_cndmask_b32 %[kb], %[kb], %[ka], vcc\n"      // hi
            "v_xor_b32 %[t1], %[t2], %[kb]\n"
            "v_cndmask_b32 %[t0], %[wa], %[wb], vcc\n"      // parent word of lo
            "v_cmp_gt_u32_e64 %[m2], %[c64k], %[t1]\n"      // equal levels: the same node
            "v_cndmask_b32 %[aa], %[aa], %[ab], vcc\n"      // address of lo
            "v_or_b32 %[t1], 0xffff, %[kb]\n"
            "v_cmp_gt_u32_e64 %[m3], %[t0], %[t1]\n"        // lo's parent is above hi (or there is none): hi slots in between
            "s_or_b64 %[m2], %[m2], %[m3]\n"
            "s_and_b64 %[m2], %[m2], %[m1]\n"               // link
            "s_mov_b64 exec, %[m2]\n"
            "ds_cmpst_rtn_b32 %[t1], %[aa], %[t0], %[kb]\n"
            "v_cmp_eq_u32_e64 %[m4], -1, %[t0]\n"           // lo had no parent: the edge is done once linked
            "s_waitcnt lgkmcnt(0)\n"
            "v_cmp_eq_u32_e64 %[m3], %[t1], %[t0]\n"        // linked
            "s_mov_b64 exec, %[busy]\n"
            "s_and_b64 %[m4], %[m4], %[m3]\n"               // (m3, m4 were written under exec = link lanes: zero elsewhere)
            "s_orn2_b64 vcc, %[m3], %[m2]\n"                // linked, or climbing: carry on with lo's (former) parent; a lost CAS repeats the pair
            "s_and_b64 vcc, vcc, %[m1]\n"
            "v_cndmask_b32 %[ka], %[t2], %[t0], vcc\n"
            "s_andn2_b64 %[busy], %[m1], %[m4]\n"
            "s_branch LOOP_%=\n"
            "DONE_%=:\n"
            "s_mov_b64 exec, %[sx]\n"
            : [ka] "=&v"(ka), [kb] "=&v"(kb), [wa] "=&v"(wa), [wb] "=&v"(wb), [aa] "=&v"(aa), [ab] "=&v"(ab), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
              [busy] "=&s"(busy), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4), [sx] "=&s"(sx), [cur] "=&s"(cur), [tmp] "=&s"(tmp)
#ifdef STR_ER_CONNECT_CNT
              , [n_it] "+s"(n_it), [n_ha] "+s"(n_ha), [n_hb] "+s"(n_hb)
#endif
            : [m] "s"(m), [elist] "s"(lds_addr(s_elist) + 2u * w0), [lev] "s"(lds_addr(s_lev)), [two] "v"(2u), [c64k] "s"(0x10000u)
            : "vcc", "scc", "memory");
#ifdef STR_ER_CONNECT_CNT
        if ((threadIdx.x & 63) == 0) { atomicAdd(&g_connect_cnt[0], 1ull); atomicAdd(&g_connect_cnt[1], (unsigned long long)n_it); atomicAdd(&g_connect_cnt[2], (unsigned long long)n_ha);
                                        atomicAdd(&g_connect_cnt[3], (unsigned long long)n_hb); atomicAdd(&g_connect_cnt[4], (unsigned long long)m); }
#endif
    }
#else
    uint32_t       a = 0, b = 0, la = 0, lb = 0;
    uint32_t       cur = 0;                                                           // wave-uniform cursor
    // which lanes have an edge in hand: a wave-uniform 64-bit mask kept in scalar registers (a per-lane flag costs a vector compare wherever
    // the wave needs to know "is anybody idle / busy")
    unsigned long long busy = 0;
    for (;;) {
        if (cur < m && ~busy != 0ull) {
            const unsigned long long idle = ~busy;
            const uint32_t c = cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const bool     tk = __builtin_amdgcn_inverse_ballot_w64(idle) && c < m;
            if (tk) {
                const uint32_t e = s_elist[w0 + c], t = e >> 12;
                a = e & 0xFFFu;
                b = a + 1u + t + 63u * (t >> 1);
                la = s_lev[LX(a)]; lb = s_lev[LX(b)];
                CNT(0, 1);
            }
            busy |= __builtin_amdgcn_ballot_w64(tk);
            cur += (uint32_t)__popcll(idle);
        }
        if (busy == 0ull) break;
        bool more = false;
        if (__builtin_amdgcn_inverse_ballot_w64(busy)) more = connect_pass(s_par, a, b, la, lb);
        busy = __builtin_amdgcn_ballot_w64(more);
    }
#endif
}

// Orders a wave's own LDS accesses around a point (no instruction: the hardware keeps a wave's LDS operations in order; this keeps the compiler from
// moving them across).
#define WAVE_SYNC()                                               \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)

// Data-parallel-primitive moves: the source lane is named in the instruction (no LDS-pipe bpermute, no address register).  A "row" is
// 16 lanes = two tile rows of 8 lanes; a lane whose source lies outside its row keeps `v` (the callers ignore those lanes).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t keep, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)keep, (int)v, CTRL, ROW_MASK, 0xF, false);
}
#define LANE_M1(v) dpp_mov<0x111>((v), (v))      // row_shr:1 -- the value of lane - 1
#define LANE_M2(v) dpp_mov<0x112>((v), (v))
#define LANE_M4(v) dpp_mov<0x114>((v), (v))
#define LANE_P1(v) dpp_mov<0x101>((v), (v))      // row_shl:1 -- the value of lane + 1

// All-reduce over the 8 lanes of a tile row (butterfly: lane ^ 1, lane ^ 2, then lane <-> 7 - lane): every lane ends up with the result.
// The operation and the lane exchange are ONE instruction (v_min_u32_dpp ...): written out, because the compiler turned "move with DPP, then
// combine" into copy + v_mov_b32_dpp + operation -- three vector instructions per step, eighteen steps in the statistics phase of every tile --
// and this kernel is bound by the number of instructions it issues.  (s_nop 1: a DPP source written by the previous vector instruction needs two
// wait states; the compiler does not see into the asm.)
#define DPP_FUSED(OPNAME, CTRL, v)                                                                              \
    ({ uint32_t r_; asm("s_nop 1\n\t" OPNAME "_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(r_) : "v"(v)); r_; })
#define ROW8_ALLREDUCE(v, OPNAME)                              \
    do {                                                       \
        v = DPP_FUSED(OPNAME, "quad_perm:[1,0,3,2]", v);       \
        v = DPP_FUSED(OPNAME, "quad_perm:[2,3,0,1]", v);       \
        v = DPP_FUSED(OPNAME, "row_half_mirror", v);           \
    } while (0)
#define OP_ADD(a, b) ((a) + (b))
#define OP_OR(a, b)  ((a) | (b))
#define OP_MIN(a, b) min((a), (b))

// Inclusive prefix sum over the wave in six DPP adds: shifts by 1, 2, 4, 8 inside every row of 16 lanes (a lane whose source is outside the
// row adds 0), then lane 15 of rows 0 and 2 is added to rows 1 and 3 (row_bcast:15) and lane 31 to rows 2 and 3 (row_bcast:31).
// (Round 2 used six ds_bpermute shuffles, each with a select: 3 scans per tile.)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += dpp_mov<0x111>(0u, v);
    v += dpp_mov<0x112>(0u, v);
    v += dpp_mov<0x114>(0u, v);
    v += dpp_mov<0x118>(0u, v);
    v += dpp_mov<0x142, 0xA>(0u, v);
    v += dpp_mov<0x143, 0xC>(0u, v);
    return v;
}

// Block-wide exclusive prefix sum of one value per lane (256 lanes = 4 waves).
// Returns the lane's offset; *total receives the block sum.  Contains two barriers.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wsum, uint32_t *total)
{
    const int      tid = threadIdx.x;
    const uint32_t incl = wave_incl_scan(v);
    __syncthreads();                     // s_wsum may still be read from the previous scan
    if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < TILE_THREADS / 64; ++i) {
        if (i < (tid >> 6)) off += s_wsum[i];
        tot += s_wsum[i];
    }
    *total = tot;
    return off + incl - v;
}

// per-node statistics in LDS: w0 (pixels | nodes | open), the set of tile rows, the set of tile columns
typedef std::conditional<(TILE_H > 32), unsigned long long, uint32_t>::type rowmask_t;
constexpr int      ROW_WORDS = (int)sizeof(rowmask_t) / 4;
constexpr int      NODE_WORDS = 3 + ROW_WORDS;                   // s_work words per node
constexpr int      CNT_BITS = TILE_PX > 2048 ? 13 : 12;          // a tile has up to TILE_PX pixels / nodes
constexpr uint32_t CNT_MASK = (1u << CNT_BITS) - 1u;
constexpr int      SLOT_BITS = TILE_PX > 2048 ? 13 : 12;         // export list entry: slot | node << SLOT_BITS | level << (SLOT_BITS + A_BITS)
constexpr int      A_BITS = TILE_H > 32 ? 11 : 10;
__device__ __forceinline__ int row_lo(uint32_t m) { return __ffs((int)m) - 1; }
__device__ __forceinline__ int row_hi(uint32_t m) { return 31 - __clz((int)m); }
__device__ __forceinline__ int row_lo(unsigned long long m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ int row_hi(unsigned long long m) { return 63 - __clzll((long long)m); }
// The tile kernel exists in two sizes.  FOLD_CAP = how many nodes a tile may have and still fold its closed nodes in LDS
// (more: every node is exported and the global passes do the folding); it sets the size of s_work and with it how many
// workgroups fit a CU: 880 nodes = 26.2 KB of LDS = 21 of the 1280-byte granules LDS is handed out in -> 6 workgroups (24 waves) per
// CU; 480 nodes = 19.9 KB = 16 granules -> 8 per CU, 13 % faster on text-like frames.  Frames that are mostly noise have ~860 nodes
// per tile and need the big one (with 480 the global accumulate pass quadruples).  The host picks per batch from the node density of
// the previous batch (str_er_api.cpp).
constexpr int FOLD_CAP_DENSE = TILE_H > 32 ? 1408 : 880;    // 21 (42) LDS granules of 1280 B: 6 (3) workgroups per CU
constexpr int FOLD_CAP_SPARSE = TILE_H > 32 ? 1024 : 480;   // 16 (32) granules: 8 (4) workgroups per CU

// ------------------------------------------------------------------------------------
// k_tile_tree: the component tree of one 64x32 tile, built from PIECES.  A piece is a maximal run of equal-level pixels inside a
// lane's 8 pixels; its first pixel is its head.  A lane knows its pieces as two bit sets (walls, run starts); everything after the load
// phase works on pieces instead of pixels: the edges that need a connect follow from the bit sets of the lane and of the lane below, and
// the phases after the connects (flatten, statistics) loop over the lane's pieces -- on text-like frames a lane holds 1.8 pieces on
// average and the fullest lane of a wave 4.4, where a loop over the lane's 8 pixels always costs 8 rounds.
// The kernel is bound by instruction issue (vector AND scalar: every divergent branch is scalar bookkeeping), not by HBM (1 byte per
// pixel) nor by LDS bandwidth: what made it faster in round 2 was fewer instructions per wave, 4435 -> 3270 (vector 2105 -> 1749, scalar
// 1960 -> 1213, LDS 370 -> 308), for 3.73 -> 3.03 ms per 32 text frames.
// (Launch shapes, prefetching and the uniform-tile short cut, measured and not adopted: NOTES.md "Tile kernel as a whole")
// Per channel (32 frames, pyr3x8): luma 0.98 ms, Cr 0.50, Cb 0.50 -- a chroma tile has a sixth of a luma tile's nodes and costs half: what a
// tile costs is mostly what EVERY tile costs (load phase 22 % of a chroma tile, building the edge list, the reductions of the statistics pass).
// ------------------------------------------------------------------------------------
#define LEVK(k) ((((k) < 4 ? lev_lo : lev_hi) >> (8 * ((k) & 3))) & 0xFFu)
// (the body is a function of the tile: k_tile_tree takes the tile from its workgroup number or from a list of tiles -- the planes k_tile_tree2 does
// not take --, k_tile_tree_fb walks the list of tiles k_tile_tree2 handed back)
template <int FOLD_CAP>
__device__ __forceinline__ void tile_tree_body(const BatchDev &b, const DetectParams &prm, const uint32_t tile_no)
{
    constexpr int WORK_WORDS = NODE_WORDS * FOLD_CAP;
    constexpr int STAT_CHUNK = FOLD_CAP < 512 ? FOLD_CAP : 512;   // dense tiles: nodes whose statistics are accumulated per pass
    // the edge list: at most 63 horizontal edges per row and 32 vertical ones per pair of rows (local minima, see below); behind it the levels of
    // every wave's first row (3 words per lane), which the wave above needs
    constexpr int ELIST_CAP = TILE_H * (TILE_W - 1) + (TILE_H - 1) * (TILE_W / 2);
    constexpr int ROWLV_AT = (ELIST_CAP + 1) / 2;                 // word offset in s_work
    static_assert(ROWLV_AT + 3 * 8 * (TILE_THREADS / 64) <= WORK_WORDS, "edge list + first-row levels must fit s_work");
    // (ONE object in shared memory, the parent words first: the connect loop takes "slot << 2" as the LDS address of a parent word, so s_par must lie at
    // LDS address 0 -- which a kernel's only shared object does; tile_connect_list checks it)
    struct TileLds {
        uint32_t par[TILE_SLOTS];
        uint32_t work[WORK_WORDS] __attribute__((aligned(8)));     // edge worklist + lane masks, later the per-node statistics
        uint16_t lev[TILE_SLOTS];      // levels; once the connects are done the same array holds the dense node id of every level-root pixel
        uint32_t wsum[TILE_THREADS / 64];
        uint32_t walls, start, nbase;
        uint32_t present[8];           // which levels have a node in this tile (big kernel)
    };
    __shared__ TileLds s_lds;
    uint32_t (&s_par)[TILE_SLOTS] = s_lds.par;
    uint32_t (&s_work)[WORK_WORDS] = s_lds.work;
    uint16_t (&s_lev)[TILE_SLOTS] = s_lds.lev;
    uint16_t *const     s_nid = s_lev;
    uint32_t (&s_wsum)[TILE_THREADS / 64] = s_lds.wsum;
    uint32_t &s_walls = s_lds.walls, &s_start = s_lds.start, &s_nbase = s_lds.nbase;
    uint32_t (&s_present)[8] = s_lds.present;
    // Small kernel: the fold (closed nodes add their totals to their parents, bottom-up over the levels) is done by ONE wave over a list of the
    // tile's level roots sorted by level -- see "fold" below.  The list is made by counting: s_hist[l] = roots at level l (counted where the
    // roots are found), turned into start offsets between the two barriers of the id scan, used as cursors where the ids are handed out.
    // It lives in the tail of s_work (never touched by the edge list), the list behind the statistics.
    constexpr bool W0FOLD = FOLD_CAP == FOLD_CAP_SPARSE;
    constexpr int  HIST_WORDS = 256;
    constexpr int  HIST_AT = WORK_WORDS - HIST_WORDS;
    static_assert(!W0FOLD || ROWLV_AT + 3 * 8 * (TILE_THREADS / 64) <= HIST_AT, "level histogram must lie behind the edge list");
    uint32_t *const s_hist = s_work + HIST_AT;

    const int       tid = threadIdx.x;
    const int       pi = b.tile_plane[tile_no];
    const PlaneDesc pd = b.planes[pi];
    const uint32_t  tl = tile_no - pd.tile_base;
    const int       tx = tl % pd.tiles_x, ty = tl / pd.tiles_x;
    const int       ox = tx * TILE_W, oy = ty * TILE_H;
    const int       ly = tid >> 3, lx = (tid & 7) * TILE_PPT;
    const uint32_t  p0 = (uint32_t)tid * TILE_PPT + ((uint32_t)tid >> 2);   // slot of the lane's first pixel
    const int       gx = ox + lx, gy = oy + ly;

    if (tid == 0) s_walls = 0;
    if (W0FOLD) { for (int i = tid; i < HIST_WORDS; i += TILE_THREADS) s_hist[i] = 0; }
    else if (tid < 8) s_present[tid] = 0;
    PHASE_INIT();

    // ---- load 8 consecutive pixels of one scanline, quantise (src/ER.cpp:250) ----------
    uint32_t lev_lo = 0, lev_hi = 0;    // the 8 levels, one byte each (walls: 0, see wallm)
    uint32_t wallm = 0, startm = 0;     // bit k: pixel k is a wall / starts a run of equal level (bit 0: unless it continues the run of the pixel to its left)
    bool     left_wall;                 // the pixel left of the lane's first one is a wall (or the tile's edge)
    uint32_t *const s_rowlv = s_work + ROWLV_AT;
    {
        uint32_t lev[TILE_PPT];
        int      nvalid = 0;
        // (written without branches per pixel: selects and bit operations only -- a branch costs scalar instructions whether or not a
        // lane takes it, and this kernel is bound by instruction issue)
        uint32_t vx = 0, vy = 0;            // the 8 pixels, bytes 0-3 and 4-7
        if (gy < pd.h && gx < pd.w) {
            const uint8_t *row = pd.pix + (size_t)gy * pd.stride + gx;
            nvalid = min(TILE_PPT, pd.w - gx);
            if (nvalid == TILE_PPT && (reinterpret_cast<uintptr_t>(row) & 7) == 0) {
                const uint2 v = *reinterpret_cast<const uint2 *>(row);
                vx = v.x; vy = v.y;
            } else {
#pragma unroll 1
                for (int k = 0; k < nvalid; ++k) {
                    const uint32_t bv = row[k];
                    if (k < 4) vx |= bv << (8 * k); else vy |= bv << (8 * (k - 4));
                }
            }
        }
        const uint32_t inv = (uint32_t)pd.invert * 0x01010101u;
        vx ^= inv; vy ^= inv;
        const uint32_t invalidm = ~((1u << nvalid) - 1u);
        uint32_t head = p0;
        uint32_t first_lev, last_lev;       // levels of the lane's pixels 0 and 7 (WALL: a wall)
        if (prm.hi <= 0x7F && (prm.thresh_step & (prm.thresh_step - 1)) == 0) {
            // thresh_step 4, 8, 16, ...: the float product p * float(1 / step) is exact, so rint_half_even of it is the integer
            //   t + ((r + (t & 1) + step / 2 - 1) >> s)   with t = p >> s, r = p mod step, step = 2^s
            // -- four pixels per instruction, a byte each (levels <= 64).  Walls, levels without the walls, run starts: byte-parallel as well
            // (bit 7 of (x | 0x80) - y is set iff x >= y for bytes below 0x80); the 8 flags are gathered from bit 7 of the 8 bytes with shifts.
            constexpr uint32_t B1 = 0x01010101u, H7 = 0x80808080u;
            const uint32_t sft = 31u - (uint32_t)__clz(prm.thresh_step);
            const uint32_t Mt = (0xFFu >> sft) * B1, Mr = ((1u << sft) - 1u) * B1, Cr = ((1u << (sft - 1u)) - 1u) * B1, HIb = (uint32_t)prm.hi * B1;
            // (pixels outside the image read as 255: the sentinel level, a wall)
            vx |= nvalid >= 4 ? 0u : 0xFFFFFFFFu << (8 * nvalid);
            vy |= nvalid >= 8 ? 0u : (nvalid <= 4 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (nvalid - 4)));
            auto quant = [&](uint32_t v) -> uint32_t {
                const uint32_t t = (v >> sft) & Mt, r = v & Mr;
                return t + (((r + (t & B1) + Cr) >> sft) & B1);
            };
            auto gather8 = [](uint32_t lo7, uint32_t hi7) -> uint32_t {      // bit 7 of the 8 bytes of (lo7, hi7) -> bits 0 .. 7
                uint32_t z = (lo7 >> 7) | (hi7 >> 3);
                z |= z >> 7;
                z |= z >> 14;
                return z & 0xFFu;
            };
            auto nz7 = [](uint32_t d) -> uint32_t { return (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u; };      // bit 7 of the bytes that are not 0
            const uint32_t qx = quant(vx), qy = quant(vy);
            const uint32_t wx7 = ((qx | H7) - HIb) & H7, wy7 = ((qy | H7) - HIb) & H7;       // bit 7: that pixel is a wall
            const uint32_t wbx = wx7 | (wx7 - (wx7 >> 7)), wby = wy7 | (wy7 - (wy7 >> 7));     // 0xFF in the bytes of walls
            wallm = gather8(wx7, wy7);
            lev_lo = qx & ~wbx; lev_hi = qy & ~wby;
            const uint32_t ex = qx | wbx, ey = qy | wby;                                     // the levels, walls as 0xFF
            // pixel k > 0 starts a run iff it is no wall and differs from pixel k - 1
            startm = gather8(nz7(ex ^ (ex << 8)) & ~wx7, nz7(ey ^ ((ey << 8) | (ex >> 24))) & ~wy7) & 0xFEu;
            // s_lev: 16 bits per pixel, walls 0xFFFF -- level byte and wall byte interleaved
            typedef uint32_t lev4_t __attribute__((ext_vector_type(4), aligned(2)));
            lev4_t pk;
            pk.x = __builtin_amdgcn_perm(wbx, ex, 0x05010400u); pk.y = __builtin_amdgcn_perm(wbx, ex, 0x07030602u);
            pk.z = __builtin_amdgcn_perm(wby, ey, 0x05010400u); pk.w = __builtin_amdgcn_perm(wby, ey, 0x07030602u);
            *reinterpret_cast<lev4_t *>(&s_lev[OWN(0)]) = pk;
            // runs: every pixel of a run points at the run's first pixel
            const uint32_t contm = ~(startm | wallm);          // bit k (k > 0): pixel k continues the run of pixel k - 1
#pragma unroll
            for (int k = 1; k < TILE_PPT; ++k) {
                const bool     same = ((contm >> k) & 1u) != 0;
                const uint32_t q = ((k < 4 ? ex : ey) >> (8 * (k & 3))) & 0xFFu;
                s_par[OWN(k)] = same ? ((q << 16) | head) : NONE;
                head = same ? head : p0 + k;
            }
            first_lev = (ex & 0xFFu) == 0xFFu ? WALL : (ex & 0xFFu);
            last_lev = (ey >> 24) == 0xFFu ? WALL : (ey >> 24);
        } else {
    #pragma unroll
            for (int k = 0; k < TILE_PPT; ++k) {
                const uint32_t q0 = (uint32_t)__float2int_rn((float)(((k < 4 ? vx : vy) >> (8 * (k & 3))) & 0xFFu) * prm.qscale);
                const bool     wall = q0 >= (uint32_t)prm.hi || ((invalidm >> k) & 1u);
                const uint32_t q = wall ? WALL : q0;
                lev[k] = q;
                s_lev[OWN(k)] = (uint16_t)q;
                wallm |= (wall ? 1u : 0u) << k;
                if (k < 4) lev_lo |= (wall ? 0u : q0) << (8 * (k & 3)); else lev_hi |= (wall ? 0u : q0) << (8 * (k & 3));
                if (k > 0) {
                    // runs: inside the lane's own 8 pixels, equal-level neighbours are one node; every pixel of a run points at the
                    // run's first pixel
                    const bool same = !wall && q == lev[k > 0 ? k - 1 : 0];
                    startm |= ((wall || same) ? 0u : 1u) << k;
                    s_par[OWN(k)] = same ? ((q << 16) | head) : NONE;
                    head = same ? head : p0 + k;
                }
            }
            first_lev = lev[0]; last_lev = lev[TILE_PPT - 1];
        }
        // ... and across the lane boundary: if the lane's first pixel continues the run of the pixel to
        // its left, it points at the head of that run.  The 8 lanes of a tile row are neighbours in the
        // wave; a lane that is one single run and itself continues leftwards forwards the head it got.
        uint32_t left_lev = LANE_M1(last_lev);
        if (lx == 0) left_lev = WALL;
        left_wall = left_lev == WALL;
        const bool joins = first_lev != WALL && first_lev == left_lev;
        if (first_lev != WALL && !joins) startm |= 1u;
        if ((ly & 7) == 0) {        // a wave's first row: the last row of the wave above reads it from LDS (the other rows are exchanged by shuffles)
            uint32_t *d = s_rowlv + 3 * ((tid >> 6) * 8 + (tid & 7));
            d[0] = lev_lo; d[1] = lev_hi; d[2] = wallm;
        }
        {
            uint32_t val = head;                       // head of the lane's last run
            bool     pass = joins && head == p0;
            // (lane - o by DPP; where that lane is in another tile row -- or outside the 16-lane DPP row -- `pass` is already false)
            { const uint32_t lv = LANE_M1(val), lp = LANE_M1((uint32_t)pass); if (pass) { val = lv; pass = lp != 0; } }
            { const uint32_t lv = LANE_M2(val), lp = LANE_M2((uint32_t)pass); if (pass) { val = lv; pass = lp != 0; } }
            { const uint32_t lv = LANE_M4(val), lp = LANE_M4((uint32_t)pass); if (pass) { val = lv; pass = lp != 0; } }
            const uint32_t t = LANE_M1(val);
            s_par[OWN(0)] = joins ? ((first_lev << 16) | t) : NONE;
        }
        const uint32_t walls = (uint32_t)__popc(wallm & ((1u << nvalid) - 1u));    // pixels of the image at the sentinel level
        __syncthreads();
        if (walls) atomicAdd(&s_walls, walls);
    }
    PHASE_MARK(0);

    // ---- connect the in-tile edges: one list, one round -----------------------------------------------------------------------
    // Which pixel pairs need a connect.  The component tree of the tile is the tree of ANY spanning subgraph of its pixel grid that holds a
    // minimum spanning forest for the edge weight max(level, level): the components of {level <= t} are those of the edges of weight <= t, and
    // an edge that is the largest of a cycle (under any fixed total order refining the weight) is in no minimum spanning forest.  The cycles
    // used here are the 2 x 2 pixel blocks, the order is (weight, then equal-level horizontal < other horizontal < vertical, then position):
    //   * horizontal edges inside a run of equal level are never dropped -- they cost nothing, the run pointers of the load phase are them;
    //   * the other horizontal edges are never the largest of a block (a vertical edge of the block weighs at least as much and ranks higher);
    //   * the vertical edge of column x, weight w(x) = max(level above, level below), is the largest of the block to its left iff
    //     w(x) >= w(x-1) and of the block to its right iff w(x) > w(x+1) (blocks with a wall in them are no cycles: w = infinity there).
    // So a pair of rows is joined at the LOCAL MINIMA of w -- the leftmost column of a plateau -- and nowhere else: 29 % of the vertical
    // edges the rule "wherever a run starts in either row" (round 2) listed on text-like planes, 34 % on noise, and what is left is within
    // a few percent of a spanning forest (text-like Y plane: 823 edges for 762 pieces).  Checked in tools/sim_tile.cpp (same trees).
    uint16_t *const s_elist = reinterpret_cast<uint16_t *>(s_work);     // one 16-bit entry per edge
    {
        const uint32_t hmask = startm & ~((wallm << 1) | (left_wall ? 1u : 0u)) & 0xFFu;
        uint32_t       vmask = 0;
        // the row below: lane + 8 of the wave, or -- for a wave's last row -- the first row of the next wave, from LDS
        uint32_t b_lo = __shfl_down(lev_lo, 8), b_hi = __shfl_down(lev_hi, 8), b_wall = __shfl_down(wallm, 8);
        if ((ly & 7) == 7 && ly + 1 < TILE_H) {
            const uint32_t *d = s_rowlv + 3 * (((tid >> 6) + 1) * 8 + (tid & 7));
            b_lo = d[0]; b_hi = d[1]; b_wall = d[2];
        }
        if (ly + 1 >= TILE_H) b_wall = 0xFFu;
        const uint32_t nowall = ~(wallm | b_wall) & 0xFFu;
        if (prm.hi <= 0x7F) {
            // levels are < 0x7F: eight columns at a time, one byte each, 0x7F = infinity (a wall in either row)
            constexpr uint32_t H = 0x80808080u;
            const uint32_t wm = wallm | b_wall;
            const uint32_t inf_lo = (((wm & 0xFu) * 0x00204081u) & 0x01010101u) * 0x7Fu, inf_hi = ((((wm >> 4) & 0xFu) * 0x00204081u) & 0x01010101u) * 0x7Fu;
            auto bmax = [](uint32_t x, uint32_t y) {
                const uint32_t m = ((((x | 0x80808080u) - y) & 0x80808080u) >> 7) * 0xFFu;      // 0xFF in the bytes where x >= y
                return (x & m) | (y & ~m);
            };
            const uint32_t w_lo = bmax(lev_lo, b_lo) | inf_lo, w_hi = bmax(lev_hi, b_hi) | inf_hi;
            uint32_t lw = LANE_M1(w_hi), rw = LANE_P1(w_lo);
            if (lx == 0) lw = 0x7F7F7F7Fu;
            if (lx == TILE_W - TILE_PPT) rw = 0x7F7F7F7Fu;
            const uint32_t prev_lo = (w_lo << 8) | (lw >> 24), prev_hi = (w_hi << 8) | (w_lo >> 24);
            const uint32_t next_lo = (w_lo >> 8) | (w_hi << 24), next_hi = (w_hi >> 8) | (rw << 24);
            // w < prev and w <= next (bit 7 of (x | H) - y is set iff x >= y)
            const uint32_t k_lo = ~((w_lo | H) - prev_lo) & ((next_lo | H) - w_lo) & H;
            const uint32_t k_hi = ~((w_hi | H) - prev_hi) & ((next_hi | H) - w_hi) & H;
            vmask = ((((k_lo >> 7) * 0x01020408u) >> 24) & 0xFu) | ((((k_hi >> 7) * 0x01020408u) >> 20) & 0xF0u);
            vmask &= nowall;
        } else {
            // thresh_step 1 and 2: levels up to 255, column by column
            constexpr uint32_t INF = 0x1FFu;
            uint32_t w[TILE_PPT];
#pragma unroll
            for (int k = 0; k < TILE_PPT; ++k) {
                const uint32_t bl = ((k < 4 ? b_lo : b_hi) >> (8 * (k & 3))) & 0xFFu;
                w[k] = ((nowall >> k) & 1u) ? max(LEVK(k), bl) : INF;
            }
            uint32_t lw = LANE_M1(w[TILE_PPT - 1]), rw = LANE_P1(w[0]);
            if (lx == 0) lw = INF;
            if (lx == TILE_W - TILE_PPT) rw = INF;
#pragma unroll
            for (int k = 0; k < TILE_PPT; ++k) {
                const uint32_t pv = k > 0 ? w[k > 0 ? k - 1 : 0] : lw, nx = k < TILE_PPT - 1 ? w[k < TILE_PPT - 1 ? k + 1 : 0] : rw;
                vmask |= (w[k] != INF && w[k] < pv && w[k] <= nx ? 1u : 0u) << k;
            }
        }
        uint32_t n_edges;
        uint32_t off = block_excl_scan(__popc(hmask) + __popc(vmask), s_wsum, &n_edges);
        uint32_t em = hmask | (vmask << 8);
        while (em) {
            const int k = __ffs((int)em) - 1;
            em &= em - 1u;
            // (bit 14: the left neighbour lies across the unused word, a lane's first pixel at a multiple of 32)
            // entry = slot of the edge's FIRST pixel (left / upper) | code << 12: the second one is 1 (code 0), 2 (1: the left neighbour lies across
            // the unused word, a lane's first pixel at a multiple of 32) or TILE_WS = 66 (2: the pixel below) slots further on
            const uint32_t cross = (k == 0 && (tid & 3) == 0) ? 1u : 0u;
            s_elist[off++] = (uint16_t)(k < 8 ? (p0 + k - 1u - cross) | (cross << 12) : (p0 + k - 8) | 0x2000u);
        }
        __syncthreads();
        tile_connect_list(s_par, s_lev, s_elist, n_edges);
        __syncthreads();
        PHASE_MARK(1);
        PHASE_MARK(2);
    }

    // ---- flatten + level roots, one pass over the lane's pieces.  The head of a piece that is not a level root is pointed straight at
    // its level root (the other pixels of a piece point at the head or, where a find halved a path, at some pixel further up in the same
    // node); the parent word of a level root is made to point at the parent node's level root.  No barrier in between: a walk follows
    // same-level words and stops at a word of another level, and neither kind of rewrite changes the level in a word.
    const uint32_t headm = (startm | (~wallm & 1u)) & 0xFFu;      // (a lane's first pixel heads a piece also when it continues a run)
    const uint32_t stopm = headm | wallm | 0x100u;
    uint32_t rootmask = 0;
    uint32_t first_root = NONE;         // level root of the lane's first piece (the statistics pass samples it)
    {
        uint32_t m = headm;
        while (m) {
            const int k = __ffs((int)m) - 1;
            m &= m - 1u;
            const uint32_t p = p0 + k;
            const uint32_t l = s_lev[LX(p)];
            const uint32_t w = LD_WG(&s_par[LX(p)]);
            if ((w >> 16) == l) {           // (NONE reads as level 0xFFFF: never a pixel's level)
                uint32_t r = w & 0xFFFFu;
                for (;;) {
                    const uint32_t w2 = LD_WG(&s_par[LX(r)]);
                    if ((w2 >> 16) != l) break;
                    r = w2 & 0xFFFFu;
                }
                s_par[LX(p)] = (l << 16) | r;
                if (first_root == NONE) first_root = r;
            } else {
                rootmask |= 1u << k;
                if (first_root == NONE) first_root = p;
                if (W0FOLD) atomicAdd(&s_hist[l & 0xFFu], 1u);
                else atomicOr(&s_present[(l >> 5) & 7u], 1u << (l & 31u));
                if constexpr (!W0FOLD) {        // (small kernel: done per NODE further down, a lane per node instead of a round of this loop)
                    if (w == NONE) continue;
                    uint32_t q = w & 0xFFFFu;
                    for (;;) {
                        const uint32_t wq = LD_WG(&s_par[LX(q)]);
                        if ((wq >> 16) != (w >> 16)) break;
                        q = wq & 0xFFFFu;
                    }
                    s_par[LX(p)] = (w & 0xFFFF0000u) | q;
                }
            }
        }
    }
    PHASE_MARK(3);

    // ---- the flood's start pixel: pixel 0, else pixel 1, else pixel w (read the
    // levels now: s_lev is about to be reused for node ids) -----------------------------------------
    if (tid == 0) {
        if (s_walls) atomicAdd(&b.ctr[pi].n_walls, s_walls);
        uint32_t sr = NONE;
        if (tl == 0) {
            int sp = -1;
            if (s_lev[LX(0)] != WALL) sp = 0;
            else if (pd.w > 1 && s_lev[LX(1)] != WALL) sp = 1;
            else if (pd.h > 1 && s_lev[LX(TILE_WS)] != WALL) sp = TILE_WS;
            if (sp >= 0) {
                const uint32_t l = s_lev[LX(sp)];
                sr = (uint32_t)sp;
                for (;;) {
                    const uint32_t w = LD_WG(&s_par[LX(sr)]);
                    if (w == NONE || (w >> 16) != l) break;
                    sr = w & 0xFFFFu;
                }
            }
        }
        s_start = sr;
    }
    // ---- dense ids for ALL level roots of the tile.  Big kernel: in pixel order (a block-wide scan).  Small kernel: in LEVEL order -- the
    // id of a root is its place in the list of the tile's roots sorted by level: s_hist[l] (roots at level l, counted in the loop above)
    // becomes the place of level l's first root (first wave, one scan over the levels), every root takes the next place of its level.
    // The statistics arrays are indexed by these ids, so the nodes of one level are neighbours there, and entry i of the list describes
    // node i: everything from here to the export works on nodes (a lane per node, ~90 of them in a text-like tile), not on pixels.
    uint32_t total_all;
    uint32_t aid0 = 0;
    if constexpr (W0FOLD) {
        __syncthreads();
        if (tid < 64) {
            uint32_t carry = 0;
            for (int base = 0; base < prm.hi; base += 64) {      // (levels 0 .. hi - 1: a root is no wall)
                const uint32_t c = s_hist[base + tid], in = wave_incl_scan(c);
                s_hist[base + tid] = carry + in - c;
                carry += (uint32_t)__builtin_amdgcn_readlane((int)in, 63);
            }
            if (tid == 0) s_wsum[0] = carry;
        }
        __syncthreads();
        total_all = s_wsum[0];
    } else {
        aid0 = block_excl_scan(__popc(rootmask), s_wsum, &total_all);
    }
    // (a lane holds half a level root on average on text-like frames: the loops over "the lane's roots" below run over the set bits)
    auto lev_of = [&](int k) -> uint32_t { return ((k < 4 ? lev_lo : lev_hi) >> (8 * (k & 3))) & 0xFFu; };
    // fold path of the small kernel: statistics [0, NODE_WORDS n), the list (a word per node) behind them, the level cursors in the tail.
    // list entry: slot of the level root (12 bits) | level << 12 | id of the parent << 20 (ORDER_NOPAR: none)
    constexpr uint32_t ORDER_NOPAR = 0x1FFu;
    static_assert(!W0FOLD || (TILE_SLOTS <= 4096 && (NODE_WORDS + 1) * 2 * TILE_THREADS >= HIST_AT), "list entry: 12-bit slots; the per-node passes take two nodes per lane");
    const uint32_t n_even_all = (total_all + 1u) & ~1u;
    const bool     w0fold = W0FOLD && (uint32_t)(NODE_WORDS + 1) * n_even_all <= (uint32_t)HIST_AT && total_all < ORDER_NOPAR;
    uint32_t *const s_order = s_work + NODE_WORDS * n_even_all;
    {
        uint32_t m = rootmask, id = aid0;
        while (m) {
            const int k = __ffs((int)m) - 1;
            m &= m - 1u;
            if constexpr (W0FOLD) {
                const uint32_t l = lev_of(k), pos = atomicAdd(&s_hist[l], 1u);
                s_nid[OWN(k)] = (uint16_t)pos;
                if (w0fold) s_order[pos] = (p0 + (uint32_t)k) | (l << 12);
            } else {
                s_nid[OWN(k)] = (uint16_t)id++;
            }
        }
    }
    __syncthreads();
    PHASE_MARK(4);

    NodeRec *const nrec = b.na.rec + pd.node_base;
    uint32_t total = 0;     // nodes exported by this tile
    // The exported nodes of a tile are consecutive records of the plane, handed out with one atomic per tile (the ids depend on
    // the order in which tiles finish; nothing downstream does -- results are ordered by key).  A plane that runs out of records
    // flags it and exports nothing from this tile: the host grows the share and repeats the batch.
    auto take_records = [&](uint32_t n) {
        uint32_t at = atomicAdd(&b.ctr[pi].n_nodes, n);
        if (at + n > pd.node_cap) { atomicOr(&b.ctr[pi].overflow, 8u); at = NONE; }
        s_nbase = at;
    };
    auto put_record = [&](uint32_t id, uint32_t par, uint32_t cnt, uint32_t nod_flags, uint32_t key_lvl, uint32_t x0, uint32_t y0,
                          uint32_t x1, uint32_t y1) {
        uint4 *dst = reinterpret_cast<uint4 *>(nrec + id);
        dst[0] = make_uint4(par, key_lvl, cnt, nod_flags);
        dst[1] = make_uint4(x0, y0, x1, y1);
        b.na.aux[pd.node_base + id] = 0;           // dependency counter of k_resolve / k_reduce
    };
    // small kernel: the parent word of level root p made to point at the parent node's level root (the big kernel does it while flattening)
    auto fix_parent = [&](uint32_t p) -> uint32_t {
        const uint32_t w = s_par[LX(p)];
        if (w == NONE) return NONE;
        uint32_t q = w & 0xFFFFu;
        for (;;) {
            const uint32_t wq = LD_WG(&s_par[LX(q)]);
            if ((wq >> 16) != (w >> 16)) break;
            q = wq & 0xFFFFu;
        }
        const uint32_t nw = (w & 0xFFFF0000u) | q;
        s_par[LX(p)] = nw;
        return nw;
    };
    // dense id of the node of the piece headed by p
    auto piece_node = [&](uint32_t p, bool isroot) -> uint32_t { return s_nid[LX(isroot ? p : (s_par[LX(p)] & 0xFFFFu))]; };

    if (W0FOLD ? w0fold : total_all <= (uint32_t)FOLD_CAP) {
        // ---- fold path.  Statistics of every node of the tile live in LDS:
        //   s_w0[a]  = pixels (CNT_BITS bits) | nodes (CNT_BITS bits) | open (bit 31)
        //   s_row[a] = set of tile rows, s_col[a] = set of tile columns the component touches.
        // "open" = the component reaches a pixel that has a neighbour in another tile, so seam
        // merging may still change it.  Everything else ("closed") is final inside this tile: a
        // closed node adds its totals to its parent here in LDS and is exported only if the
        // reference would keep it (area > MIN_AREA); the thousands of small speckle nodes never
        // reach global memory, yet they are counted (ER::area includes the node count).
        // The three arrays are packed for the tile's own node count n (not FOLD_CAP): what is left
        // of s_work behind them holds the export list further down.
        const uint32_t      n_even = (total_all + 1u) & ~1u;
        uint32_t           *s_w0 = s_work;                                   // [n_even]
        rowmask_t          *s_row = reinterpret_cast<rowmask_t *>(s_work + n_even);                  // [n_even]
        unsigned long long *s_col = reinterpret_cast<unsigned long long *>(s_work + (1 + ROW_WORDS) * n_even); // [n_even]
        uint32_t           *s_exp = s_work + NODE_WORDS * n_even;            // [NODE_WORDS * (FOLD_CAP - n_even)]
        for (uint32_t i = tid; i < total_all; i += TILE_THREADS) {
            s_w0[i] = 0; s_row[i] = 0; s_col[i] = 0ull;
            if (W0FOLD) {
                // the parent's id joins the list entry: what the fold needs, looked up by all lanes here instead of by the one wave that
                // folds, level after level
                const uint32_t en = s_order[i], w = fix_parent(en & 0xFFFu);
                s_order[i] = en | ((w == NONE ? ORDER_NOPAR : (uint32_t)s_nid[LX(w & 0xFFFFu)]) << 20);
            }
        }
        __syncthreads();
        // One set of LDS atomics per piece -- except for the pieces of the row's two HOT nodes.  A text-like tile is two or three big
        // nodes (the background levels) and dozens of speckles: hundreds of pieces add to the same three words, and LDS atomics of a wave
        // that hit one address are carried out one lane after the other (measured: this pass took 21 % of the kernel for 10 % of its
        // instructions).  So every tile row (8 lanes) picks two nodes -- the node of its first piece and of the first piece of another node
        // (sampled from the lanes' first pieces) -- whose pieces are summed in registers, reduced over the row with three DPP steps and
        // added by ONE lane: at most 32 atomics per word and tile for a hot node.  All other pieces take the atomics below.
        // The piece headed by the node's level root also carries the node itself (+1 in the node field), a piece with a pixel on a seam
        // carries the side bits (OR-ed separately: an add could carry).
        {
            const bool top = ly == 0 && ty > 0, bot = ly == TILE_H - 1 && ty + 1 < pd.tiles_y;
            const bool lef = lx == 0 && tx > 0, rig = lx == TILE_W - TILE_PPT && tx + 1 < pd.tiles_x;
            const uint32_t chunk = (uint32_t)tid & 7u;
            // (the big kernel is picked for batches that are mostly noise: a row's pieces are all different nodes there, nothing is hot)
            constexpr bool HOT = FOLD_CAP == FOLD_CAP_SPARSE;
            uint32_t h1 = NONE, h2 = NONE;            // the row's hot level roots (slots; NONE: none)
            if (HOT) {
                uint32_t d = first_root == NONE ? 0xFFFFFFFFu : ((chunk << 16) | first_root);
                ROW8_ALLREDUCE(d, "v_min_u32");
                h1 = d == 0xFFFFFFFFu ? NONE : (d & 0xFFFFu);
                d = (first_root == NONE || first_root == h1) ? 0xFFFFFFFFu : ((chunk << 16) | first_root);
                ROW8_ALLREDUCE(d, "v_min_u32");
                h2 = d == 0xFFFFFFFFu ? NONE : (d & 0xFFFFu);
            }
            // per hot node: pixels (7 bits) | is-the-root-piece (bit 7); node 1 in bits 0..15, node 2 in 16..31.  hsides: the tile sides
            // (top 1, bottom 2, left 4, right 8) its pieces lie on, node 1 in bits 0..3, node 2 in bits 4..7
            uint32_t acc = 0, col1 = 0, col2 = 0, hsides = 0;
            const uint32_t side_tb = (top ? 1u : 0u) | (bot ? 2u : 0u);
            uint32_t m = headm;
            while (m) {
                const int k = __ffs((int)m) - 1;
                m &= m - 1u;
                const uint32_t len = (uint32_t)__ffs((int)(stopm >> (k + 1)));      // distance to the next head, wall or the lane's end
                const bool     isroot = ((rootmask >> k) & 1u) != 0;
                // the sides of the tile the piece lies on ("open": the node can still change when the tiles are joined; which sides, because
                // joining goes in two steps -- groups of tiles first, k_group_merge, and a seam inside a group is no border any more)
                const uint32_t sd = side_tb | ((lef && k == 0) ? 4u : 0u) | ((rig && k + (int)len == TILE_PPT) ? 8u : 0u);
                const uint32_t r = isroot ? p0 + (uint32_t)k : (s_par[LX(p0 + (uint32_t)k)] & 0xFFFFu);
                const uint32_t cb = ((1u << len) - 1u) << k;
                const uint32_t add = len | (isroot ? 0x80u : 0u);
                if (r == h1) { acc += add; col1 |= cb; hsides |= sd; }
                else if (r == h2) { acc += add << 16; col2 |= cb; hsides |= sd << 4; }
                else {
                    const uint32_t id = s_nid[LX(r)];
                    atomicAdd(&s_w0[id], len + (isroot ? 1u << CNT_BITS : 0u));
                    atomicOr(&s_row[id], (rowmask_t)1 << ly);
                    // (the lane's 8 columns lie in one half of the 64-bit column set)
                    atomicOr(reinterpret_cast<uint32_t *>(&s_col[id]) + (lx >> 5), cb << (lx & 31));
                    if (sd) atomicOr(&s_w0[id], sd << 28);
                }
            }
          if (HOT) {
            ROW8_ALLREDUCE(acc, "v_add_u32");
            ROW8_ALLREDUCE(hsides, "v_or_b32");
            // column sets: the 4 lanes of a quad own the 4 bytes of one dword (lanes 0-3: columns 0-31, lanes 4-7: columns 32-63)
            col1 <<= 8u * (chunk & 3u); col2 <<= 8u * (chunk & 3u);
            col1 = DPP_FUSED("v_or_b32", "quad_perm:[1,0,3,2]", col1); col1 = DPP_FUSED("v_or_b32", "quad_perm:[2,3,0,1]", col1);
            col2 = DPP_FUSED("v_or_b32", "quad_perm:[1,0,3,2]", col2); col2 = DPP_FUSED("v_or_b32", "quad_perm:[2,3,0,1]", col2);
            const uint32_t oth1 = dpp_mov<0x141>(col1, col1), oth2 = dpp_mov<0x141>(col2, col2);     // the other quad's dword
            // lane 0 of the row adds node 1 (its own dword is the low one), lane 4 node 2 (its own dword is the high one)
            const uint32_t my_h = chunk == 0 ? h1 : h2, my_acc = chunk == 0 ? (acc & 0xFFFFu) : (acc >> 16);
            const uint32_t my_lo = chunk == 0 ? col1 : oth2, my_hi = chunk == 0 ? oth1 : col2;
            if ((chunk & 3u) == 0 && my_h != NONE && (my_acc & 0x7Fu) != 0) {
                const uint32_t id = s_nid[LX(my_h)];
                atomicAdd(&s_w0[id], (my_acc & 0x7Fu) + ((my_acc & 0x80u) ? 1u << CNT_BITS : 0u));
                atomicOr(&s_row[id], (rowmask_t)1 << ly);
                if (my_lo) atomicOr(reinterpret_cast<uint32_t *>(&s_col[id]), my_lo);
                if (my_hi) atomicOr(reinterpret_cast<uint32_t *>(&s_col[id]) + 1, my_hi);
                const uint32_t my_sd = chunk == 0 ? (hsides & 0xFu) : (hsides >> 4);
                if (my_sd) atomicOr(&s_w0[id], my_sd << 28);
            }
          }
        }
        __syncthreads();
        PHASE_MARK(5);
      if (W0FOLD) {
        // bottom-up over the levels present in the tile (children are at lower levels than parents), by the first wave alone: the roots of a
        // level are consecutive entries of the sorted list, one lane each.  A text-like tile has ~90 nodes on ~7 levels: the owners' form
        // below has all four waves compare their pixels' levels with every level and meet at a barrier per level for a handful of
        // nodes each (301 instructions per wave); here three waves go straight to the barrier behind the fold.  The wave's LDS operations
        // are carried out in the order it issues them, so a level sees the sums of the levels below without any barrier.
        if (tid < 64) {
            // (measured in place, tools/dev_wg_trace.py: the fold is 13-14 % of a luma workgroup's LIFETIME for 3 % of its instructions -- a chain over the
            // levels by one wave while three wait.  Raising the wave's priority for it (s_setprio 3, also for the scan of the level counts) shortened it by
            // 6 % and the kernel not at all: not adopted.)
            uint32_t begin = 0;
            for (int base = 0; base < prm.hi; base += 64) {
                const uint32_t end = s_hist[base + tid];                 // cursor of level base + lane = where its entries end
                const uint32_t prev = (uint32_t)__shfl_up((int)end, 1);          // the level below (lane 0: see `begin`)
                unsigned long long pm = __ballot(end != (tid == 0 ? begin : prev));
                while (pm) {
                    const int      l = __ffsll((long long)pm) - 1;
                    pm &= pm - 1ull;
                    const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)end, l);
                    for (uint32_t e = begin + (uint32_t)tid; e < e1; e += 64u) {
                        // (everything a node needs is requested at once -- ONE trip to LDS per level instead of three dependent ones: the fold is a chain
                        // over the levels, 14 % of a luma workgroup's lifetime, measured in place: tools/dev_wg_trace.py)
                        const uint32_t           a = e, en = s_order[e], v = s_w0[a];
                        const rowmask_t          rw = s_row[a];
                        const unsigned long long cl = s_col[a];
                        const uint32_t           pa = en >> 20;
                        if (pa == ORDER_NOPAR) continue;
                        if (v >> 28) atomicOr(&s_w0[pa], v & 0xF0000000u);       // open: so is the parent, on the same sides
                        else {
                            atomicAdd(&s_w0[pa], v & ((1u << (2 * CNT_BITS)) - 1u));
                            atomicOr(&s_row[pa], rw);
                            atomicOr(&s_col[pa], cl);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    begin = e1;
                }
                begin = (uint32_t)__builtin_amdgcn_readlane((int)end, 63);
            }
        }
        __syncthreads();
      } else {
        uint32_t rootspread = 0;        // rootmask with pixel k at bit 8 (k & 3) + 4 (k >> 2)
#pragma unroll
        for (int k = 0; k < TILE_PPT; ++k) rootspread |= ((rootmask >> k) & 1u) << (8 * (k & 3) + 4 * (k >> 2));
        // bottom-up over the levels present in the tile: children are at lower levels than parents
        // (only the levels that occur: one barrier per level)
        for (int wd = 0; wd < 8; ++wd) {
          uint32_t pm = s_present[wd];
          while (pm) {
            const uint32_t t = (uint32_t)wd * 32u + (uint32_t)__ffs((int)pm) - 1u;
            pm &= pm - 1u;
            // the lane's roots at level t: compare all 8 level bytes at once (0x80 in every byte of x that is zero), then keep the roots --
            // a tile has a few hundred nodes on six levels, so most lanes have nothing to do at a given level
            const uint32_t tt = t * 0x01010101u;
            const uint32_t x0 = lev_lo ^ tt, x1 = lev_hi ^ tt;
            const uint32_t z0 = ~(((x0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x0 | 0x7F7F7F7Fu), z1 = ~(((x1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x1 | 0x7F7F7F7Fu);
            uint32_t       hit = ((z0 >> 7) | (z1 >> 3)) & rootspread;       // bit 8 j (pixel j < 4), bit 8 j + 4 (pixel 4 + j)
            while (hit) {
                const int pos = __ffs((int)hit) - 1;
                hit &= hit - 1u;
                const int k = (pos >> 3) | (pos & 4);
                const uint32_t a = aid0 + (uint32_t)__popc(rootmask & ((1u << k) - 1u));
                const uint32_t w = s_par[OWN(k)];
                if (w == NONE) continue;
                const uint32_t pa = s_nid[LX(w & 0xFFFFu)];
                const uint32_t v = s_w0[a];
                if (v >> 28) atomicOr(&s_w0[pa], v & 0xF0000000u);       // open: so is the parent, on the same sides
                else {
                    atomicAdd(&s_w0[pa], v & ((1u << (2 * CNT_BITS)) - 1u));
                    atomicOr(&s_row[pa], s_row[a]);
                    atomicOr(&s_col[pa], s_col[a]);
                }
            }
            __syncthreads();
          }
        }
      }
        PHASE_MARK(7);
        // (the scan's barriers separate the last reads of s_nid as "all-node id" from the rewrite)
        // One exported node: everything it needs is in LDS except its own level and whether it is open.
        auto export_node = [&](uint32_t nbase, uint32_t p, uint32_t a, uint32_t l) {
            uint32_t q = s_par[LX(p)], ql = 0;
            if (q != NONE) { ql = (q >> 16) & 0xFFu; q &= 0xFFFFu; }
            while (q != NONE && s_nid[LX(q)] == 0xFFFFu) {      // only the start pixel's node can need this
                const uint32_t w2 = s_par[LX(q)];
                if (w2 == NONE) q = NONE;
                else { ql = (w2 >> 16) & 0xFFu; q = w2 & 0xFFFFu; }
            }
            const uint32_t v = s_w0[a];
            const unsigned long long cm = s_col[a];
            const rowmask_t          rm = s_row[a];
            const uint32_t px = SLOT_PIXEL(p);
            put_record(nbase + s_nid[LX(p)], (q == NONE) ? NONE : PAR_MAKE(ql, nbase + s_nid[LX(q)]), v & CNT_MASK,
                       ((v >> CNT_BITS) & CNT_MASK) | ((v >> 28) ? (v >> 28) << 26 : NODE_CLOSED),       // (NODE_SIDE_T .. _R = bits 26..29)
                       (uint32_t)((oy + (int)(px >> 6)) * pd.w + ox + (int)(px & 63u)) | (l << 24),
                       ox + __ffsll((long long)cm) - 1, oy + row_lo(rm), ox + 63 - __clzll((long long)cm), oy + row_hi(rm));
        };
        if constexpr (W0FOLD) {
            // which nodes leave the tile: open ones, closed ones the reference keeps, tile roots and the node of the flood's start pixel.
            // A lane per node (two rounds for a tile with more than 256); the exported ones get consecutive record ids in list order.
            const uint32_t sroot = s_start;
            uint32_t       en[2] = {0u, 0u}, keep = 0, cnt = 0;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t i = (uint32_t)tid + (uint32_t)(r * TILE_THREADS);
                if (i < total_all) {
                    en[r] = s_order[i];
                    const uint32_t v = s_w0[i];
                    const uint32_t area = (v & CNT_MASK) + ((v >> CNT_BITS) & CNT_MASK);
                    if ((v >> 28) != 0 || (int64_t)area > (int64_t)prm.min_area || (en[r] >> 20) == ORDER_NOPAR || (en[r] & 0xFFFu) == sroot) {
                        keep |= 1u << r;
                        ++cnt;
                    }
                }
            }
            const uint32_t eid0 = block_excl_scan(cnt, s_wsum, &total);
            if (tid == 0) take_records(total);
            // (the scan's barriers separate the last reads of s_nid as "all-node id" from the rewrite)
            {
                uint32_t id = eid0;
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if ((uint32_t)tid + (uint32_t)(r * TILE_THREADS) < total_all) s_nid[LX(en[r] & 0xFFFu)] = (uint16_t)(((keep >> r) & 1u) ? id++ : 0xFFFFu);
            }
            __syncthreads();
            const uint32_t nbase = s_nbase;
            if (nbase != NONE) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if ((keep >> r) & 1u) export_node(nbase, en[r] & 0xFFFu, (uint32_t)tid + (uint32_t)(r * TILE_THREADS), (en[r] >> 12) & 0xFFu);
            }
        } else {
            // which nodes leave the tile: open ones, closed ones the reference keeps, tile roots and the
            // node of the flood's start pixel
            uint32_t expmask = 0;
            {
                uint32_t m = rootmask, id = aid0;
                const uint32_t sroot = s_start;
                while (m) {
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const uint32_t v = s_w0[id++];
                    const uint32_t area = (v & CNT_MASK) + ((v >> CNT_BITS) & CNT_MASK);
                    const bool     open = (v >> 28) != 0;
                    if (open || (int64_t)area > (int64_t)prm.min_area || s_par[OWN(k)] == NONE || p0 + k == sroot) expmask |= 1u << k;
                }
            }
            const uint32_t eid0 = block_excl_scan(__popc(expmask), s_wsum, &total);
            if (tid == 0) take_records(total);
            // The exported nodes are listed behind the statistics (slot | a << SLOT_BITS | level << (SLOT_BITS + A_BITS))
            // and written out one per lane; a tile too full for the list writes them from the owners.
            const bool listed = (uint32_t)NODE_WORDS * n_even + total <= (uint32_t)NODE_WORDS * (uint32_t)FOLD_CAP;
            {
                uint32_t m = rootmask, id = eid0, aid = aid0;
                while (m) {
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const uint32_t a = aid++;
                    if ((expmask >> k) & 1) {
                        if (listed) s_exp[id] = (p0 + k) | (a << SLOT_BITS) | (lev_of(k) << (SLOT_BITS + A_BITS));
                        s_nid[OWN(k)] = (uint16_t)id++;
                    } else {
                        s_nid[OWN(k)] = (uint16_t)0xFFFFu;
                    }
                }
            }
            __syncthreads();
            const uint32_t nbase = s_nbase;
            if (nbase == NONE) {
                // no records: nothing leaves this tile
            } else if (listed) {
                for (uint32_t e = tid; e < total; e += TILE_THREADS) {
                    const uint32_t w = s_exp[e];
                    export_node(nbase, w & ((1u << SLOT_BITS) - 1u), (w >> SLOT_BITS) & ((1u << A_BITS) - 1u), (w >> (SLOT_BITS + A_BITS)) & 0xFFu);
                }
            } else {
                uint32_t aid = aid0;
    #pragma unroll 1
                for (int k = 0; k < TILE_PPT; ++k) {
                    if (!((rootmask >> k) & 1)) continue;
                    const uint32_t a = aid++;
                    if ((expmask >> k) & 1) export_node(nbase, p0 + k, a, lev_of(k));
                }
            }
        }
    } else {
        // ---- dense tile (more than FOLD_CAP nodes): export every node with its own statistics,
        // STAT_CHUNK nodes per pass; the global passes do all the accumulation.
        total = total_all;
        if (tid == 0) take_records(total);
        if constexpr (W0FOLD) {
            for (uint32_t m = rootmask; m; m &= m - 1u) fix_parent(p0 + (uint32_t)__ffs((int)m) - 1u);
        }
        uint32_t            *s_cnt = s_work;                       // [STAT_CHUNK]
        rowmask_t           *s_row = reinterpret_cast<rowmask_t *>(s_work + STAT_CHUNK);          // [STAT_CHUNK]
        unsigned long long  *s_col = reinterpret_cast<unsigned long long *>(s_work + (1 + ROW_WORDS) * STAT_CHUNK); // [STAT_CHUNK]
        for (uint32_t c0 = 0; c0 < total; c0 += STAT_CHUNK) {
            for (int i = tid; i < NODE_WORDS * STAT_CHUNK; i += TILE_THREADS) s_work[i] = 0;
            __syncthreads();
            {   // one set of atomics per piece of the lane
                uint32_t m = headm;
                while (m) {
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const uint32_t id = piece_node(p0 + k, ((rootmask >> k) & 1u) != 0) - c0;
                    if (id >= (uint32_t)STAT_CHUNK) continue;
                    const uint32_t len = (uint32_t)__ffs((int)(stopm >> (k + 1)));
                    atomicAdd(&s_cnt[id], len);
                    atomicOr(&s_row[id], (rowmask_t)1 << ly);
                    atomicOr(reinterpret_cast<uint32_t *>(&s_col[id]) + (lx >> 5), ((1u << len) - 1u) << ((lx & 31) + k));
                }
            }
            __syncthreads();
            const uint32_t nbase = s_nbase;
#pragma unroll 1
            for (int k = 0; k < TILE_PPT; ++k) {
                if (!((rootmask >> k) & 1) || nbase == NONE) continue;
                const uint32_t p = p0 + k;
                const uint32_t li = (uint32_t)s_nid[LX(p)] - c0;
                if (li >= (uint32_t)STAT_CHUNK) continue;
                const uint32_t w = s_par[LX(p)];
                const unsigned long long cm = s_col[li];
                const rowmask_t          rm = s_row[li];
                // (no fold, so nothing is known about sides: a node of a dense tile counts as lying on all four)
                put_record(nbase + s_nid[LX(p)], (w == NONE) ? NONE : PAR_MAKE((w >> 16) & 0xFFu, nbase + s_nid[LX(w & 0xFFFFu)]), s_cnt[li], 1u | NODE_SIDES,
                           (uint32_t)(gy * pd.w + gx + k) | (lev_of(k) << 24),
                           ox + __ffsll((long long)cm) - 1, oy + row_lo(rm), ox + 63 - __clzll((long long)cm), oy + row_hi(rm));
            }
            __syncthreads();
        }
    }
    if (!(W0FOLD && w0fold)) __syncthreads();       // (the small kernel's fold path has read s_nbase behind a barrier already and writes no LDS after it)
    const uint32_t nbase = s_nbase;
    if (tid == 0) {
        b.tile_nbase[tile_no] = nbase;
        b.tile_nrec[tile_no] = (uint16_t)total;
        if (tl == 0) b.ctr[pi].start_node = (s_start == NONE || nbase == NONE) ? NONE : nbase + s_nid[LX(s_start)];
    }
    PHASE_MARK(13);

    // ---- node of every tile-border pixel, for the seam pass: its index inside this tile's records (16 bits; the seam
    // kernel adds tile_nbase) ---------------------------------------------------------------
    // seam layout per plane: for every horizontal tile boundary j (1..tiles_y-1) two rows
    // of w entries (pixel row j*TH-1, then j*TH); then for every vertical boundary i two
    // columns of h entries (pixel column i*TW-1, then i*TW).
    // The 64 pixels of the tile's top row belong to 8 lanes (the first 8 of wave 0), those of the bottom row to the last 8 of the last wave.
    // Had the owners written them, 8 lanes of a wave would walk through 8 pixels each while 56 watch (the kernel is bound by the number of
    // instructions its waves issue, whatever the lanes do): instead every lane of the wave takes ONE pixel -- it fetches the owner's bit
    // sets with one shuffle -- and the wave writes the row with one store (round 3: 289 -> 95 instructions per wave for this phase).
    {
        uint16_t *seam = b.seam + pd.seam_base;
        const uint32_t voff = 2u * pd.w * (pd.tiles_y - 1);
        // record (inside the tile) of the node of pixel k of the lane whose first slot is q0: the level root of the piece it lies in
        auto node_of = [&](uint32_t wm, uint32_t hm, uint32_t rm, uint32_t q0, int k) -> uint16_t {
            if (((wm >> k) & 1u) || nbase == NONE) return (uint16_t)0xFFFFu;
            const int      hk = 31 - __clz((int)(hm & ((2u << k) - 1u)));       // head of the piece
            const uint32_t hp = q0 + (uint32_t)hk;
            return s_nid[LX(((rm >> hk) & 1u) ? hp : (s_par[LX(hp)] & 0xFFFFu))];
        };
        const int  wv = tid >> 6, lane = tid & 63;
        const bool do_top = wv == 0 && ty > 0, do_bot = wv == TILE_THREADS / 64 - 1 && ty + 1 < pd.tiles_y;      // (wave-uniform)
        const uint32_t sets = wallm | (headm << 8) | (rootmask << 16);
        if (do_top || do_bot) {
            const int      owner = (do_top ? 0 : 64 - TILE_W / TILE_PPT) + (lane >> 3);      // lane of this wave that holds the pixel
            const uint32_t os = (uint32_t)__shfl((int)sets, owner);
            const uint32_t otid = (uint32_t)(tid & ~63) + (uint32_t)owner;
            const int      col = ox + lane, row = do_top ? oy : oy + TILE_H - 1;
            if (col < pd.w && row < pd.h)
                seam[(do_top ? ((size_t)(ty - 1) * 2 + 1) : ((size_t)ty * 2)) * pd.w + col] =
                    node_of(os & 0xFFu, (os >> 8) & 0xFFu, os >> 16, otid * TILE_PPT + (otid >> 2), lane & 7);
        }
        const bool lef = lx == 0 && tx > 0, rig = lx == TILE_W - TILE_PPT && tx + 1 < pd.tiles_x;
        if ((lef || rig) && gy < pd.h && (lef ? gx : gx + TILE_PPT - 1) < pd.w)
            seam[voff + (lef ? ((size_t)(tx - 1) * 2 + 1) : ((size_t)tx * 2)) * pd.h + gy] = node_of(wallm, headm, rootmask, p0, lef ? 0 : TILE_PPT - 1);
    }
    PHASE_MARK(6);
}

template <int FOLD_CAP>
__global__ __launch_bounds__(TILE_THREADS, (FOLD_CAP == FOLD_CAP_SPARSE ? 8 : 6)) void k_tile_tree(BatchDev b, DetectParams prm, const uint32_t *list)
{
    tile_tree_body<FOLD_CAP>(b, prm, list ? list[blockIdx.x] : blockIdx.x);
}
// the tiles k_tile_tree2 handed back (more levels / nodes / records than it takes): their number is known on the device only, so a fixed number of
// workgroups walks the list
template <int FOLD_CAP>
__global__ __launch_bounds__(TILE_THREADS, (FOLD_CAP == FOLD_CAP_SPARSE ? 8 : 6)) void k_tile_tree_fb(BatchDev b, DetectParams prm, const uint32_t *list, const uint32_t *count)
{
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        tile_tree_body<FOLD_CAP>(b, prm, list[i]);
        __syncthreads();
    }
}

#ifdef STR_ER_PHASE_PROF
extern "C" void str_er_debug_phase_cycles(unsigned long long *out16, int reset)
{
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tile_phase), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_phase), z, sizeof(z));
    }
}
#endif

// list == nullptr: every tile of the batch; else the n listed tiles
void launch_tile_tree(hipStream_t s, const BatchDev &b, const DetectParams &p, bool sparse, const uint32_t *list, uint32_t n)
{
    const uint32_t grid = list ? n : b.n_tiles;
    if (!grid) return;
    if (sparse) hipLaunchKernelGGL(k_tile_tree<FOLD_CAP_SPARSE>, dim3(grid), dim3(TILE_THREADS), 0, s, b, p, list);
    else        hipLaunchKernelGGL(k_tile_tree<FOLD_CAP_DENSE>, dim3(grid), dim3(TILE_THREADS), 0, s, b, p, list);
}

void launch_tile_tree_fb(hipStream_t s, const BatchDev &b, const DetectParams &p, bool sparse, const uint32_t *list, const uint32_t *count, uint32_t grid)
{
    if (!grid) return;
    if (sparse) hipLaunchKernelGGL(k_tile_tree_fb<FOLD_CAP_SPARSE>, dim3(grid), dim3(TILE_THREADS), 0, s, b, p, list, count);
    else        hipLaunchKernelGGL(k_tile_tree_fb<FOLD_CAP_DENSE>, dim3(grid), dim3(TILE_THREADS), 0, s, b, p, list, count);
}
