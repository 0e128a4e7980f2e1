// tile2_body.h -- the component tree of a 64 x 32 tile, second form (k_tile_tree2): level by level on BIT MASKS.
//
// What it computes is what k_tile_tree computes (er_tile_tree.inl; semantics: /root/reference/src/ER.cpp:131-191, 240-413): the nodes of the
// tile's component tree that leave the tile -- open ones (their component reaches a side of the tile that has a neighbouring tile), closed
// ones the reference keeps (area > MIN_AREA), tile roots, the node of the flood's start pixel -- as 32-byte records with the totals of their
// closed descendants folded in, plus the seam map.  How it computes it is different: k_tile_tree joins PIECES with a lock-free union-find
// in LDS (cost: ~1000 vector instructions per 512 pixels whatever the tile holds); this one is for tiles with FEW levels and FEW nodes
// (the chroma planes: 2.2 levels and ~15 nodes per tile), where almost all of that is fixed cost.
//
// One wave = two horizontally adjacent tiles; a lane = one tile ROW (lanes 0..31 tile A, 32..63 tile B), a row of pixels = a 64-bit mask.
//   for every level t present, ascending:   M = {level <= t} (one byte-parallel compare of the packed levels),  E = M \ M(previous level)
//     while E has pixels no component of this level has claimed:
//        seed  = the first of them in raster order  (= the node's key pixel: the smallest own-level pixel of its component)
//        F     = the component of M that holds the seed: flood on masks.  Inside a row a run is filled by ONE add (the carry runs
//                through the run) -- both directions with a bit reversal; between rows the masks move by one lane (DPP).  A big
//                component of an earlier node (BG) joins in one step once it is touched.
//        the node (t, F): every component of a lower level inside F is complete (they were built first), so nothing ever climbs:
//                pixels  = |F \ O|   (O: pixels of OPEN nodes so far -- those push their own totals later, k_reduce)
//                nodes   = 1 + |N & F \ O|   (N: one marked pixel per node so far)
//                box     = box of F \ O,  sides = the tile sides F lies on,  parent of the pending exported nodes inside F := this node
// No union-find, no per-pixel LDS state, no barriers (one wave), no atomics but the reservation of the records.
//
// The algorithm is written ONCE against an execution policy W: on the device W's vector types are plain per-lane scalars and its cross-lane
// operations are DPP / swizzle / readlane (er_tile_tree2.inl); on the host they are 64-element arrays executed in lock step
// (tests/cpp/tile2_model_check.cpp), where the very same source is checked tile by tile against a brute-force component tree and its
// vector operations are counted.  Hence the style: no per-lane `if` / `?:` (W::sel), per-lane loops as `while (W::any(..))` with masked
// updates, wave-uniform branches only.
#pragma once
#include <stdint.h>

#include "er_kernels.h"
#include "er_types.h"

#ifdef __HIPCC__
#define T2_FN static __device__ __forceinline__
#else
#define T2_FN static inline
#endif

namespace str_er {
namespace t2 {

// (the host check raises the three limits, -DSTR_ER_T2_REC_CAP=.. etc., to run the algorithm on tiles the kernel would hand back)
#ifndef STR_ER_T2_REC_CAP
#define STR_ER_T2_REC_CAP 64
#define STR_ER_T2_MAX_LEVELS 12
#define STR_ER_T2_MAX_STEPS 160
#endif
constexpr int      REC_CAP = STR_ER_T2_REC_CAP;          // exported records of a tile held in LDS; a tile with more goes to the fall-back list (k_tile_tree)
constexpr int      MAX_LEVELS = STR_ER_T2_MAX_LEVELS;    // levels in the pair of tiles; more -> fall-back
constexpr int      MAX_STEPS = STR_ER_T2_MAX_STEPS;      // node steps of the pair; more -> fall-back
constexpr uint32_t BG_MIN = 192;          // a component of at least this many pixels is remembered as BG
constexpr uint32_t NOREC = 0xFFFFu;

// what a launch works on besides the batch
struct Args {
    const uint32_t *pairs;      // entry: first tile of the pair (batch-wide tile number) | 1 << 31 if the tile to its right takes part
    uint32_t        n_pairs;
    uint32_t       *fb_list;    // tiles handed back to k_tile_tree (batch-wide tile numbers)
    uint32_t       *fb_count;
};

template <class W>
struct Body {
    typedef typename W::u32  u32;
    typedef typename W::u64  u64;
    typedef typename W::mask mask;

    // pixels of the lane's row whose level is >= c (1 <= c <= 64): qh holds the row's 64 levels, one byte each, with bit 7 set, so a
    // byte of qh - c * 0x01010101 keeps bit 7 exactly when level >= c; the eight flags of eight pixels are gathered with two shift-ors
    T2_FN u64 mask_ge(const u32 (&qh)[16], uint32_t c)
    {
        const uint32_t cc = c * 0x01010101u;
        u32            acc[2] = {W::bc(0u), W::bc(0u)};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const u32 g0 = qh[2 * g] - cc, g1 = qh[2 * g + 1] - cc;
            u32       z = W::and_or(g1 >> 3, 0x10101010u, (g0 >> 7) & 0x01010101u);
            z = W::lshl_or(z, 7, z);
            z = W::lshl_or(z, 14, z);
            acc[g >> 2] = W::lshl_or(W::bfe(z, 21, 8), 8 * (g & 3), acc[g >> 2]);
        }
        return W::mk64(acc[0], acc[1]);
    }

    // the runs of m (a row's mask) that hold a bit of s (s a subset of m), whole: upwards the carry of m + s runs from the lowest seed of a run to
    // its end, downwards the same on the reversed words
    T2_FN u64 hfill(u64 m, u64 mr, u64 s)
    {
        const u64 x = m + s;
        const u64 up = W::bfi64(x, s, m);             // (x & s) | (~x & m)
        const u64 sr = W::brev64(s), xr = mr + sr;
        const u64 dn = W::brev64(W::bfi64(xr, sr, mr));
        return up | dn;
    }

    // every column of f continued up and down through m, over the tile's 32 rows: Kogge-Stone over the lanes inside the 16-row blocks a DPP row shift
    // reaches (4 steps each way; afterwards p is the AND of m from the block's first / last row up to the lane's row), then across the blocks' border
    // with what row 15 (going down) / row 16 (going up) ended with
    T2_FN u64 vfill(W &w, u64 f, u64 m, mask lower_block)
    {
        u64 g = f, p = m;
#pragma unroll
        for (int k = 1; k < 16; k *= 2) { g = W::and_or64(w.rows_down(g, k), p, g); p = p & w.rows_down(p, k); }
        g = W::and_or64(W::sel64(lower_block, w.row15_of_upper(g), W::bc64(0ull)), p, g);
        u64 g2 = f;
        p = m;
#pragma unroll
        for (int k = 1; k < 16; k *= 2) { g2 = W::and_or64(w.rows_up(g2, k), p, g2); p = p & w.rows_up(p, k); }
        g2 = W::and_or64(W::sel64(lower_block, W::bc64(0ull), w.row16_of_lower(g2)), p, g2);
        return g | g2;
    }

    T2_FN void run(W &w, const BatchDev &b, const DetectParams &prm, const Args &a, uint32_t pair)
    {
        const uint32_t  entry = a.pairs[pair];
        const uint32_t  tileA = entry & 0x7FFFFFFFu;
        const bool      hasB = (entry >> 31) != 0u;
        const int       pi = b.tile_plane[tileA];
        const PlaneDesc pd = b.planes[pi];
        const uint32_t  tl = tileA - pd.tile_base;
        const int       tx0 = (int)(tl % (uint32_t)pd.tiles_x), ty = (int)(tl / (uint32_t)pd.tiles_x);
        const int       oy = ty * TILE_H;
        const u32       lane = w.lane();
        const u32       half = lane >> 5, row = lane & 31u;
        const mask      isB = half != 0u;
        const mask      live = hasB ? W::all() : !isB;
        const u32       ox = (W::bc((uint32_t)tx0) + half) * (uint32_t)TILE_W;
        const u32       gy = W::bc((uint32_t)oy) + row;
        const mask      rowvalid = live & (gy < (uint32_t)pd.h);
        // columns of the tile inside the image: 64, fewer in the plane's last tile column
        const u32       wleft = W::bc((uint32_t)pd.w) - ox;
        const u32       ncols = W::sel(wleft < 64u, wleft, W::bc(64u));
        const bool      supported = prm.hi <= 32 && prm.hi >= 2 && (prm.thresh_step & (prm.thresh_step - 1)) == 0 && TILE_H == 32;

        W::mark(0);
        // ---- load the row, quantise (src/ER.cpp:250), pack: qh = level | 0x80 per byte; pixels outside the image read as 255 = the sentinel level
        u32 qh[16];
        {
            const bool fast = (pd.stride & 15) == 0 && (reinterpret_cast<uintptr_t>(pd.pix) & 15u) == 0u && (tx0 + (hasB ? 2 : 1)) * TILE_W <= pd.stride;
            w.load_row(pd.pix, gy * (uint32_t)pd.stride + ox, rowvalid, ncols, fast, qh);
            const uint32_t inv = (uint32_t)pd.invert * 0x01010101u;
            const bool     ragged = W::any(!rowvalid | (ncols < 64u));
            const uint32_t sft = 31u - (uint32_t)__builtin_clz((unsigned)(prm.thresh_step | 1));
            const uint32_t B1 = 0x01010101u;
            const uint32_t Mt = (0xFFu >> sft) * B1, Mr = ((1u << sft) - 1u) * B1, Cr = (sft ? ((1u << (sft - 1u)) - 1u) : 0u) * B1;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                u32 v = qh[k] ^ inv;
                if (ragged) {
                    // bytes of this dword that lie inside the image: ncols - 4k, clamped to 0 .. 4
                    const u32 nv = W::sel(rowvalid, W::min_u(W::max_i(ncols - (uint32_t)(4 * k), 0), 4u), W::bc(0u));
                    v = v | W::sel(nv >= 4u, W::bc(0u), W::bc(0xFFFFFFFFu) << (nv * 8u));
                }
                // thresh_step 2^s: rint_half_even(p / step) = t + ((r + (t & 1) + step / 2 - 1) >> s), t = p >> s, r = p mod step -- four pixels at once
                const u32 t = (v >> sft) & Mt, r = v & Mr;
                qh[k] = (t + (((r + (t & B1) + Cr) >> sft) & B1)) | 0x80808080u;
            }
        }
        W::mark(1);
        // walls: the sentinel level hi = 255 / step + 1 (never flooded) and everything outside the image
        const u64 Wm = mask_ge(qh, (uint32_t)prm.hi);
        u64       inimg;
        {
            const u64 colm = W::sel64(ncols >= 64u, W::bc64(~0ull), W::shl64(W::bc64(1ull), ncols & 63u) - 1ull);
            inimg = W::sel64(rowvalid, colm, W::bc64(0ull));
        }
        // which levels occur (both tiles; a wall may add a level that is not there, which costs one empty turn of the loop)
        uint32_t present;
        {
            u32 p = W::bc(0u);
#pragma unroll
            for (int k = 0; k < 16; ++k) p = W::onehot4_or(qh[k], p);
            present = w.wave_or(p);
            if (prm.hi < 32) present &= (1u << prm.hi) - 1u;
        }

        // ---- state
        u64 Mprev = W::bc64(0ull), O = W::bc64(0ull), N = W::bc64(0ull), R = W::bc64(0ull), BG = W::bc64(0ull);
        u32 nrec = W::bc(0u);                    // records of the lane's tile so far (half-uniform)
        u32 startid = W::bc(NOREC);              // record of the start pixel's node
        mask dead = !live;                       // the half takes no (further) part: no tile, or handed to the fall-back
        mask fb = W::none();                     // ... handed to the fall-back
        u32 topid[2] = {W::bc(NOREC), W::bc(NOREC)}, botid[2] = {W::bc(NOREC), W::bc(NOREC)};      // lane = COLUMN: record of the tile's top / bottom row pixels
        u32 leftid = W::bc(NOREC), rightid = W::bc(NOREC);                                            // lane = row: record of the row's first / last pixel
        const u32 rowflag = W::sel(row == 0u, W::bc(1u), W::bc(0u)) | W::sel(row == 31u, W::bc(2u), W::bc(0u));
        // the sides that have a neighbouring tile (top 1, bottom 2, left 4, right 8)
        const u32 sidemask = W::bc((ty > 0 ? 1u : 0u) | (ty + 1 < pd.tiles_y ? 2u : 0u)) | W::sel((W::bc((uint32_t)tx0) + half) > 0u, W::bc(4u), W::bc(0u)) |
                             W::sel((W::bc((uint32_t)tx0) + half + 1u) < (uint32_t)pd.tiles_x, W::bc(8u), W::bc(0u));
        // the pixels of the lane's row that lie on such a side
        const u64 SP = W::sel64(((sidemask & 1u) != 0u) & (row == 0u), W::bc64(~0ull), W::bc64(0ull)) | W::sel64(((sidemask & 2u) != 0u) & (row == 31u), W::bc64(~0ull), W::bc64(0ull)) |
                       W::mk64((sidemask >> 2) & 1u, (sidemask >> 3) << 31);
        const u32 npix = w.half_sum(W::popc64(inimg & ~Wm));          // flooded pixels of the tile
        const u32 nwall = w.half_sum(W::popc64(inimg & Wm));           // in-image pixels at the sentinel level
        const mask haswalls = w.half_sum(W::popc64(Wm)) != 0u;       // (outside-the-image pixels count: a ragged tile takes the general root test)
        // the flood's start pixel: pixel 0, else pixel 1, else pixel w -- in the plane's first tile
        u64 spm = W::bc64(0ull);
        const bool has_start = tl == 0u;
        if (has_start) {
            const uint64_t w0 = w.read_lane64(Wm, 0), w1 = pd.h > 1 ? w.read_lane64(Wm, 1) : ~0ull;
            int sr = -1, sc = 0;
            if (!(w0 & 1ull)) { sr = 0; sc = 0; }
            else if (pd.w > 1 && !(w0 & 2ull)) { sr = 0; sc = 1; }
            else if (pd.h > 1 && !(w1 & 1ull)) { sr = 1; sc = 0; }
            if (sr >= 0) spm = W::sel64(lane == (uint32_t)sr, W::bc64(1ull << sc), W::bc64(0ull));
        }
        if (!supported || __builtin_popcount(present) > MAX_LEVELS) { fb = live; dead = W::all(); present = 0; }
        // Isolated runs (below) may be marked in bulk where a small closed node can be neither a tile root nor the start pixel's node: no walls in
        // the tile (then anything smaller than the tile borders on a flooded pixel) and at least two rows of it inside the image
        const mask bulk_ok = !haswalls & W::sel_half(isB, 1u, has_start ? 0u : 1u) != 0u & (W::bc((uint32_t)(pd.h - oy)) >= 2u);

        uint32_t steps = 0;
        while (present) {
            W::mark(2);
            const uint32_t t = (uint32_t)__builtin_ctz(present);
            present &= present - 1u;
            const u64 M = W::sel64(dead, W::bc64(0ull), ~mask_ge(qh, t + 1u) & ~Wm);
            const u64 E = M & ~Mprev;
            Mprev = M;
            u64       U = E;
            if (!W::any(U != 0ull)) continue;
            const u64 Mr = W::brev64(M);
            const u64 Mup = W::sel64(row != 0u, M, W::bc64(0ull)), Mdn = W::sel64(row != 31u, M, W::bc64(0ull));
            {
                // ---- speckles in bulk.  A run of M with no pixel of M above or below it is a whole component; if it holds an own-level pixel it is a
                // node, closed unless it lies on an open side, and with 2 * length <= MIN_AREA too small to be kept (area = pixels + nodes <= 2 * pixels):
                // such a node leaves nothing behind but its mark in N.  Three quarters of the nodes of a text-like chroma tile are of this kind.
                const u64  V = W::sel64(row != 0u, w.row_above(M), W::bc64(0ull)) | W::sel64(row != 31u, w.row_below(M), W::bc64(0ull));
                const u64  touched = hfill(M, Mr, M & (V | SP));
                const mask rowok = bulk_ok & !W::gt_i64(W::popc64(M) * 2u, prm.min_area);
                const u64  iso = W::sel64(rowok, M & ~touched, W::bc64(0ull));
                // (the runs of `iso` are whole runs of M.  The mark goes to the run's first OWN-level pixel -- where the carry of M + s starts --: the run's
                // first pixel may be a lower node's mark)
                const u64  sd = U & iso;
                const u64  upf = W::bfi64(M + sd, sd, M);
                const u64  sdr = W::brev64(sd);
                const u64  Rn = upf | W::brev64(W::bfi64(Mr + sdr, sdr, Mr));
                N = N | (upf & ~(upf << 1));
                U = U & ~Rn;
            }
            while (W::any(U != 0ull)) {
                if (++steps > (uint32_t)MAX_STEPS) { fb = fb | (live & !dead); dead = W::all(); present = 0; break; }
                W::mark(3);
                // ---- the seed: first unclaimed own-level pixel in raster order = the node's key pixel
                const uint64_t ub = W::ballot(U != 0ull);
                const uint32_t ubA = (uint32_t)ub, ubB = (uint32_t)(ub >> 32);
                const uint32_t rA = ubA ? (uint32_t)__builtin_ctz(ubA) : 63u, rB = ubB ? (uint32_t)__builtin_ctz(ubB) : 63u;     // (63: no row of the half)
                const mask     act = W::sel_half(isB, ubB, ubA) != 0u;
                const u64      seed = W::sel64(row == W::sel_half(isB, rB, rA), U & (W::bc64(0ull) - U), W::bc64(0ull));
                W::mark(4);
                // ---- the component of M that holds it
                u64  F = hfill(M, Mr, seed);
                mask merged;
                int  rounds = 0;
                {
                    bool force = false;
                    // (a seed in a long run is a seed in something big: no point in creeping row by row first)
                    const bool wide = W::any(W::popc64(F) >= 24u);
                    merged = W::none();
                    bool bg_on = W::any((BG != 0ull) & act);
                    for (;;) {
                        if (bg_on) {
                            // a remembered component that F has reached joins whole (it is connected and inside M)
                            const uint64_t hit = W::ballot((F & BG) != 0ull);
                            if (hit) {
                                const mask hm = W::sel_half(isB, (uint32_t)(hit >> 32), (uint32_t)hit) != 0u;
                                const u64  add = W::sel64(hm, BG & ~F, W::bc64(0ull));
                                if (W::any(add != 0ull)) { F = F | add; force = true; }
                                BG = W::sel64(hm, W::bc64(0ull), BG);       // (inside F now: F is the remembered component from here on)
                                merged = merged | hm;
                                bg_on = W::any((BG != 0ull) & act);
                            }
                        }
                        const u64 nb = W::bfi64(F, W::bc64(0ull), W::and_or64(w.row_above(F), Mup, w.row_below(F) & Mdn));     // vertical neighbours inside M, not in F yet
                        if (!force && !W::any(nb != 0ull)) break;
                        F = hfill(M, Mr, F | nb);
                        W::stat(0, 1);
                        force = false;
                        if (++rounds >= 4 || (wide && rounds <= 2)) {
                            // still growing after four rounds of one row each (or big from the start; measured on text-like chroma planes with the
                            // host model: without this step 58 rounds per pair of tiles instead of 19, the thresholds hardly matter): every column of F up and down through M as far
                            // as it goes (log steps over the lanes), then the runs again -- a component that spans the tile takes 2 or 3 such rounds, not 30
                            W::stat(1, 1);
                            const u64 Fv = vfill(w, F, M, (row & 16u) != 0u);
                            if (W::any((Fv & ~F) != 0ull)) F = hfill(M, Mr, F | Fv);
                        }
                    }
                }
                W::mark(5);
                W::stat(2, rounds);
                U = U & ~F;
                // ---- the node (t, F)
                const u64 X = F & ~O;
                u32       w1 = W::popc64(X) | (W::popc64(X & N) << 16);
                w1 = w.half_sum(w1);
                const u32      cnt = w1 & 0xFFFFu, nodc = (w1 >> 16) + 1u;
                const uint64_t ob = W::ballot((F & SP) != 0ull);
                const mask     open = act & (W::sel_half(isB, (uint32_t)(ob >> 32), (uint32_t)ob) != 0u);
                const u32  area = cnt + nodc;
                const mask small = !(W::gt_i64(area, prm.min_area));
                // a closed small node leaves the tile only as a tile root (nothing flooded borders on it) or as the start pixel's node
                mask isroot = act & !open & (cnt == npix);
                if (W::any(act & !open & small & haswalls)) {
                    // (row 0 / 31 of a half would read the other half's rows)
                    const u64 adjv = W::sel64(row != 0u, w.row_above(F), W::bc64(0ull)) | W::sel64(row != 31u, w.row_below(F), W::bc64(0ull));
                    const u64 nbr = (adjv | (F << 1) | (F >> 1)) & ~F & ~Wm;
                    const u32 anyn = w.half_or(W::sel(nbr != 0ull, W::bc(1u), W::bc(0u)));
                    isroot = isroot | (act & !open & (anyn == 0u));
                }
                mask isstart = W::none();
                if (has_start) isstart = act & (w.half_or(W::sel((F & E & spm) != 0ull, W::bc(1u), W::bc(0u))) != 0u);
                const mask exported = act & (open | !small | isroot | isstart);
                N = N | seed;
                O = O | W::sel64(open, F, W::bc64(0ull));
                if (W::any(exported)) {
                    W::mark(6);
                    const u32 id = nrec;
                    nrec = nrec + W::sel(exported, W::bc(1u), W::bc(0u));
                    // the sides of the tile the component lies on
                    const u32 sf = w.half_or(W::sel(F != 0ull, rowflag, W::bc(0u)) | ((W::lo(F) & 1u) << 2) | ((W::hi(F) >> 31) << 3)) & sidemask;
                    const mask over = exported & (id >= (uint32_t)REC_CAP);
                    if (W::any(over)) { fb = fb | over; dead = dead | over; }
                    const mask ex = exported & !over;
                    // the key pixel: row << 6 | column
                    const uint64_t sdA = w.read_lane64(seed, (int)(rA & 31u)), sdB = w.read_lane64(seed, (int)(32u + (rB & 31u)));
                    const u32      sp = W::sel_half(isB, (rB << 6) | (sdB ? (uint32_t)__builtin_ctzll(sdB) : 0u), (rA << 6) | (sdA ? (uint32_t)__builtin_ctzll(sdA) : 0u));
                    // box of F \ O
                    const u32      cl = w.half_or(W::lo(X)), ch = w.half_or(W::hi(X));
                    const u64      cols = W::mk64(cl, ch);
                    const uint64_t rb = W::ballot(X != 0ull);
                    const u32      rows = W::sel_half(isB, (uint32_t)(rb >> 32), (uint32_t)rb);
                    // (a half that exports nothing computes on zeros here; nothing of it is used)
                    const u32      x0 = ox + W::ffs64(cols), x1 = ox + W::fls64(cols);
                    const u32      y0 = W::ffs32(rows) + (uint32_t)oy, y1 = W::fls32(rows) + (uint32_t)oy;
                    const u32      key = ((W::bc((uint32_t)oy) + (sp >> 6)) * (uint32_t)pd.w + ox + (sp & 63u)) | (t << 24);
                    const u32      flags = W::sel(open, sf << 26, W::bc(NODE_CLOSED));
                    {
                        u32 f[8] = {W::bc(NONE), key, cnt, nodc | flags, x0, y0, x1, y1};
                        w.rec_write(half, id, f, ex & (row == 0u));
                        w.idmap_write(half, sp & 2047u, id, ex & (row == 0u));
                    }
                    startid = W::sel(ex & isstart, id, startid);
                    // the pending exported nodes inside F get this node as their parent
                    {
                        u64       pend = W::sel64(ex, R & F, W::bc64(0ull));
                        const u32 pw = (t << 24) | id;
                        while (W::any(pend != 0ull)) {
                            const mask pm = pend != 0ull;
                            const u32  c = W::ffs64(pend | W::sel64(pm, W::bc64(0ull), W::bc64(1ull)));
                            const u32  cid = w.idmap_read(half, (row << 6) | c, pm);
                            w.rec_set_par(half, cid, pw, pm);
                            pend = pend & (pend - 1ull);
                        }
                        R = W::sel64(ex, (R & ~F) | seed, R);
                    }
                    // seam map: the node's own pixels on the tile's border rows / columns
                    const u64 own = W::sel64(ex, F & E, W::bc64(0ull));
                    if (W::any(ex & ((sf & 3u) != 0u))) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t idh = w.read_lane(id, 32 * h);
                            const uint64_t o0 = w.read_lane64(own, 32 * h), o1 = w.read_lane64(own, 32 * h + 31);
                            topid[h] = W::sel(W::lanes_of(o0), W::bc(idh), topid[h]);
                            botid[h] = W::sel(W::lanes_of(o1), W::bc(idh), botid[h]);
                        }
                    }
                    leftid = W::sel((W::lo(own) & 1u) != 0u, id, leftid);
                    rightid = W::sel((W::hi(own) >> 31) != 0u, id, rightid);
                }
                BG = W::sel64(act & ((cnt >= BG_MIN) | merged) & !dead, F, BG);
                // a half that was handed back stops here
                if (W::any(dead & (U != 0ull))) { U = W::sel64(dead, W::bc64(0ull), U); Mprev = W::sel64(dead, W::bc64(~0ull), Mprev); }
            }
        }

        W::mark(7);
        // ---- the tiles' records: one reservation for the pair, tile A's first
        const uint64_t fbm = W::ballot(fb), livem = W::ballot(live);
        const bool     fbA = (fbm & 1ull) != 0, fbB = (fbm >> 32 & 1ull) != 0, liveB = (livem >> 32 & 1ull) != 0;
        if (fbA | fbB) {
            const uint32_t n = (fbA ? 1u : 0u) + (fbB ? 1u : 0u);
            const uint32_t at = w.atomic_add(a.fb_count, n);
            if (fbA) w.store_scalar(a.fb_list + at, tileA);
            if (fbB) w.store_scalar(a.fb_list + at + (fbA ? 1u : 0u), tileA + 1u);
        }
        const mask     out = live & !fb;                       // halves that export
        const uint32_t nA = fbA ? 0u : w.read_lane(nrec, 0), nB = (!liveB || fbB) ? 0u : w.read_lane(nrec, 32);
        uint32_t       base = 0;
        bool           norec = false;
        if (nA + nB) {
            base = w.atomic_add(&b.ctr[pi].n_nodes, nA + nB);
            if (base + nA + nB > pd.node_cap) { w.atomic_or(&b.ctr[pi].overflow, 8u); norec = true; }
        }
        const u32 nbase = norec ? W::bc(NONE) : W::sel(isB, W::bc(base + nA), W::bc(base));
        {
            NodeRec *const nrecs = b.na.rec + pd.node_base;
            uint32_t *const aux = b.na.aux + pd.node_base;
            const uint32_t  most = nA > nB ? nA : nB;
            for (uint32_t r0 = 0; r0 < most && !norec; r0 += 32u) {
                const u32  i = row + r0;
                const mask m = out & (i < nrec);
                u32        f[8];
                w.rec_read(half, i, f, m);
                f[0] = W::sel(f[0] == NONE, f[0], f[0] + nbase);
                w.store_rec(nrecs, nbase + i, f, m);
                w.store_u32(aux, nbase + i, W::bc(0u), m);
            }
        }
        {
            const mask lead = out & (row == 0u);
            w.store_u32(b.tile_nbase, W::bc(tileA) + half, nbase, lead);
            w.store_u16(b.tile_nrec, W::bc(tileA) + half, nrec, lead);
            if (has_start && !fbA) w.store_scalar(&b.ctr[pi].start_node, (w.read_lane(startid, 0) == NOREC || norec) ? NONE : base + w.read_lane(startid, 0));
            const uint32_t wallsA = fbA ? 0u : w.read_lane(nwall, 0), wallsB = (!liveB || fbB) ? 0u : w.read_lane(nwall, 32);
            if (wallsA + wallsB) w.atomic_add(&b.ctr[pi].n_walls, wallsA + wallsB);
        }
        // ---- seam map (layout: er_tile_tree.inl): per horizontal tile boundary two rows of w entries, then per vertical boundary two columns of h entries
        {
            uint16_t *const seam = b.seam + pd.seam_base;
            const uint32_t  voff = 2u * (uint32_t)pd.w * (uint32_t)(pd.tiles_y - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool on = h == 0 ? !fbA : (liveB && !fbB);
                if (!on) continue;
                const uint32_t oxh = (uint32_t)(tx0 + h) * (uint32_t)TILE_W;
                const u32      col = W::bc(oxh) + lane;
                const mask     inw = col < (uint32_t)pd.w;
                if (ty > 0) {
                    const mask wl = W::lanes_of(w.read_lane64(Wm, 32 * h));
                    w.store_u16(seam, W::bc((uint32_t)((ty - 1) * 2 + 1) * (uint32_t)pd.w) + col, W::sel(wl | norec_mask(norec), W::bc(NOREC), topid[h]), inw);
                }
                if (ty + 1 < pd.tiles_y) {
                    const mask wl = W::lanes_of(w.read_lane64(Wm, 32 * h + 31));
                    w.store_u16(seam, W::bc((uint32_t)(ty * 2) * (uint32_t)pd.w) + col, W::sel(wl | norec_mask(norec), W::bc(NOREC), botid[h]), inw);
                }
            }
            const u32  txv = W::bc((uint32_t)tx0) + half;
            const mask inh = out & (gy < (uint32_t)pd.h);
            {
                const mask m = inh & (txv > 0u);
                w.store_u16(seam, W::bc(voff) + ((txv - 1u) * 2u + 1u) * (uint32_t)pd.h + gy, W::sel(((W::lo(Wm) & 1u) != 0u) | norec_mask(norec), W::bc(NOREC), leftid), m);
            }
            {
                const mask m = inh & ((txv + 1u) < (uint32_t)pd.tiles_x);
                w.store_u16(seam, W::bc(voff) + (txv * 2u) * (uint32_t)pd.h + gy, W::sel(((W::hi(Wm) >> 31) != 0u) | norec_mask(norec), W::bc(NOREC), rightid), m);
            }
        }
    }

    T2_FN mask norec_mask(bool norec) { return norec ? W::all() : W::none(); }
};

} // namespace t2
} // namespace str_er
