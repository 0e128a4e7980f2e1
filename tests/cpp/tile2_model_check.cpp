// tile2_model_check.cpp -- the second tile kernel's algorithm (scene-text-recognition_amd/csrc/tile2_body.h) executed on the HOST, lane by lane in
// lock step, and compared tile by tile with a brute-force component tree of the same tile (components of {level <= t} by breadth-first search,
// nodes / parents / totals by their definition).  The body is the very source the device kernel compiles (er_tile_tree2.inl gives it per-lane
// scalars and DPP; here its vector types are 64-element arrays), so what is checked here is the algorithm the GPU runs, not a model of it.
// Also counts the body's vector operations per tile (a 64-bit operation as two): the figure the kernel's time follows.
//
//   g++ -O2 -std=c++17 -I scene-text-recognition_amd/csrc -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tests/cpp/tile2_model_check.cpp -o /tmp/t2check && /tmp/t2check [rounds] [plane.lev W H]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <queue>
#include <vector>

#include "tile2_body.h"

using namespace str_er;

static unsigned long long g_ops = 0;      // vector operations (32-bit units)
static unsigned long long g_phase_ops[8], g_phase_n[8], g_mark_at = 0;
static int g_phase = 0;
static unsigned long long g_stat[32];

template <class T> struct V {
    T v[64];
    V() {}
    V(T s) { for (int i = 0; i < 64; ++i) v[i] = s; }
};
struct MaskV {
    uint64_t b;
};
#define VOP(op)                                                                                                                           \
    template <class T> V<T> operator op(const V<T> &a, const V<T> &b) { V<T> r; for (int i = 0; i < 64; ++i) r.v[i] = (T)(a.v[i] op b.v[i]); g_ops += sizeof(T) / 4; return r; } \
    template <class T, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> V<T> operator op(const V<T> &a, S b) { V<T> r; for (int i = 0; i < 64; ++i) r.v[i] = (T)(a.v[i] op (T)b); g_ops += sizeof(T) / 4; return r; } \
    template <class T, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> V<T> operator op(S a, const V<T> &b) { V<T> r; for (int i = 0; i < 64; ++i) r.v[i] = (T)((T)a op b.v[i]); g_ops += sizeof(T) / 4; return r; }
VOP(+) VOP(-) VOP(&) VOP(|) VOP(^) VOP(*)
#undef VOP
template <class T> V<T> operator~(const V<T> &a) { V<T> r; for (int i = 0; i < 64; ++i) r.v[i] = (T)~a.v[i]; g_ops += sizeof(T) / 4; return r; }
template <class T> V<T> operator<<(const V<T> &a, int s) { V<T> r; for (int i = 0; i < 64; ++i) r.v[i] = (T)(a.v[i] << s); g_ops += sizeof(T) / 4; return r; }
template <class T> V<T> operator>>(const V<T> &a, int s) { V<T> r; for (int i = 0; i < 64; ++i) r.v[i] = (T)(a.v[i] >> s); g_ops += sizeof(T) / 4; return r; }
template <class T> V<T> operator<<(const V<T> &a, uint32_t s) { return a << (int)s; }
template <class T> V<T> operator>>(const V<T> &a, uint32_t s) { return a >> (int)s; }
template <class T> V<T> operator<<(const V<T> &a, const V<uint32_t> &s) { V<T> r; for (int i = 0; i < 64; ++i) r.v[i] = (T)(a.v[i] << (s.v[i] & (8 * sizeof(T) - 1))); g_ops += sizeof(T) / 4; return r; }
#define VCMP(op)                                                                                                                          \
    template <class T> MaskV operator op(const V<T> &a, const V<T> &b) { MaskV r{0}; for (int i = 0; i < 64; ++i) r.b |= (uint64_t)(a.v[i] op b.v[i]) << i; g_ops += 1; return r; } \
    template <class T, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> MaskV operator op(const V<T> &a, S b) { MaskV r{0}; for (int i = 0; i < 64; ++i) r.b |= (uint64_t)(a.v[i] op (T)b) << i; g_ops += 1; return r; }
VCMP(==) VCMP(!=) VCMP(<) VCMP(>) VCMP(<=) VCMP(>=)
#undef VCMP
static MaskV operator&(MaskV a, MaskV b) { return MaskV{a.b & b.b}; }
static MaskV operator|(MaskV a, MaskV b) { return MaskV{a.b | b.b}; }
static MaskV operator!(MaskV a) { return MaskV{~a.b}; }

// what the body writes to "global memory"
struct HostWave {
    typedef V<uint32_t> u32;
    typedef V<uint64_t> u64;
    typedef MaskV       mask;
    uint32_t lds_rec[2][t2::REC_CAP][8];
    uint16_t lds_idmap[2][2048];     // (bytes on the device, where REC_CAP <= 256)
    bool     junk_neighbours = true;      // rows beyond a half's first / last row read the other half's rows, like the hardware's wave shift

    static void stat(int i, int add) { if (i == 2) { ++g_stat[8 + std::min(add, 15)]; } else g_stat[i] += add; }
    static void mark(int ph) { g_phase_ops[g_phase] += g_ops - g_mark_at; g_mark_at = g_ops; g_phase = ph; ++g_phase_n[ph]; }
    u32 lane() const { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (uint32_t)i; return r; }
    static u32 bc(uint32_t s) { return u32(s); }
    static u64 bc64(uint64_t s) { return u64(s); }
    static mask all() { return MaskV{~0ull}; }
    static mask none() { return MaskV{0ull}; }
    static bool any(mask m) { return m.b != 0; }
    static uint64_t ballot(mask m) { g_ops += 1; return m.b; }
    static mask lanes_of(uint64_t bits) { return MaskV{bits}; }
    static u32 sel(mask m, const u32 &a, const u32 &b) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (m.b >> i & 1) ? a.v[i] : b.v[i]; g_ops += 1; return r; }
    static u64 sel64(mask m, const u64 &a, const u64 &b) { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = (m.b >> i & 1) ? a.v[i] : b.v[i]; g_ops += 2; return r; }
    static u32 sel_half(mask isB, uint32_t vb, uint32_t va) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (isB.b >> i & 1) ? vb : va; g_ops += 2; return r; }
    static u32 and_or(const u32 &a, uint32_t m, const u32 &c) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (a.v[i] & m) | c.v[i]; g_ops += 1; return r; }
    static u64 and_or64(const u64 &a, const u64 &m, const u64 &c) { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = (a.v[i] & m.v[i]) | c.v[i]; g_ops += 2; return r; }
    static u32 lshl_or(const u32 &a, int s, const u32 &c) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (a.v[i] << s) | c.v[i]; g_ops += 1; return r; }
    static u32 bfe(const u32 &a, int off, int wd) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (a.v[i] >> off) & ((1u << wd) - 1u); g_ops += 1; return r; }
    static u64 bfi64(const u64 &x, const u64 &a, const u64 &b) { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = (x.v[i] & a.v[i]) | (~x.v[i] & b.v[i]); g_ops += 2; return r; }
    static u64 brev64(const u64 &a) { u64 r; for (int i = 0; i < 64; ++i) { uint64_t x = a.v[i], y = 0; for (int k = 0; k < 64; ++k) y |= (x >> k & 1ull) << (63 - k); r.v[i] = y; } g_ops += 2; return r; }
    static u64 mk64(const u32 &lo, const u32 &hi) { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = (uint64_t)lo.v[i] | (uint64_t)hi.v[i] << 32; return r; }
    static u32 lo(const u64 &a) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (uint32_t)a.v[i]; return r; }
    static u32 hi(const u64 &a) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (uint32_t)(a.v[i] >> 32); return r; }
    static u64 shl64(const u64 &a, const u32 &s) { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] << (s.v[i] & 63u); g_ops += 1; return r; }
    static u32 popc64(const u64 &a) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (uint32_t)__builtin_popcountll(a.v[i]); g_ops += 2; return r; }
    static u32 ffs64(const u64 &a) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] ? (uint32_t)__builtin_ctzll(a.v[i]) : 0xFFFFFFFFu; g_ops += 4; return r; }
    static u32 fls64(const u64 &a) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] ? 63u - (uint32_t)__builtin_clzll(a.v[i]) : 0xFFFFFFFFu; g_ops += 4; return r; }
    static u32 ffs32(const u32 &a) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] ? (uint32_t)__builtin_ctz(a.v[i]) : 0xFFFFFFFFu; g_ops += 1; return r; }
    static u32 fls32(const u32 &a) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] ? 31u - (uint32_t)__builtin_clz(a.v[i]) : 0xFFFFFFFFu; g_ops += 2; return r; }
    static u32 min_u(const u32 &a, uint32_t b) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = std::min(a.v[i], b); g_ops += 1; return r; }
    static u32 max_i(const u32 &a, int b) { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (uint32_t)std::max((int)a.v[i], b); g_ops += 1; return r; }
    static mask gt_i64(const u32 &a, int32_t b) { MaskV r{0}; for (int i = 0; i < 64; ++i) r.b |= (uint64_t)((int64_t)a.v[i] > (int64_t)b) << i; g_ops += 1; return r; }
    static u32 onehot4_or(const u32 &q, const u32 &acc) { u32 r; for (int i = 0; i < 64; ++i) { uint32_t x = acc.v[i]; for (int k = 0; k < 4; ++k) x |= 1u << ((q.v[i] >> (8 * k)) & 31u); r.v[i] = x; } g_ops += 6; return r; }
    // cross-lane
    u64 row_above(const u64 &a) const { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = i == 0 ? (junk_neighbours ? 0x5555AAAA5555AAAAull : 0ull) : a.v[i - 1]; g_ops += 2; return r; }
    u64 row_below(const u64 &a) const { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = i == 63 ? (junk_neighbours ? 0xAAAA5555AAAA5555ull : 0ull) : a.v[i + 1]; g_ops += 2; return r; }
    // the value of the row k above / below inside the 16-lane DPP row; 0 where that leaves the row
    u64 rows_down(const u64 &a, int k) const { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = (i % 16) >= k ? a.v[i - k] : 0ull; g_ops += 2; return r; }
    u64 rows_up(const u64 &a, int k) const { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = (i % 16) + k < 16 ? a.v[i + k] : 0ull; g_ops += 2; return r; }
    // the tile's row 15 as seen from rows 16 .. 31 / row 16 as seen from rows 0 .. 15 (other lanes: anything)
    u64 row15_of_upper(const u64 &a) const { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[(i & 32) + 15]; g_ops += 2; return r; }
    u64 row16_of_lower(const u64 &a) const { u64 r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[(i & 32) + 16]; g_ops += 6; return r; }
    template <class F> static u32 half_red(const u32 &a, F f) { u32 r; for (int h = 0; h < 2; ++h) { uint32_t x = a.v[32 * h]; for (int i = 1; i < 32; ++i) x = f(x, a.v[32 * h + i]); for (int i = 0; i < 32; ++i) r.v[32 * h + i] = x; } g_ops += 6; return r; }
    u32 half_sum(const u32 &a) const { return half_red(a, [](uint32_t x, uint32_t y) { return x + y; }); }
    u32 half_or(const u32 &a) const { return half_red(a, [](uint32_t x, uint32_t y) { return x | y; }); }
    u32 half_min(const u32 &a) const { return half_red(a, [](uint32_t x, uint32_t y) { return std::min(x, y); }); }
    uint32_t wave_or(const u32 &a) const { uint32_t x = 0; for (int i = 0; i < 64; ++i) x |= a.v[i]; g_ops += 7; return x; }
    uint32_t read_lane(const u32 &a, int l) const { g_ops += 1; return a.v[l]; }
    uint64_t read_lane64(const u64 &a, int l) const { g_ops += 2; return a.v[l]; }
    // memory
    void load_row(const uint8_t *base, const u32 &off, mask rowvalid, const u32 &ncols, bool fast, u32 (&out)[16]) const
    {
        for (int k = 0; k < 16; ++k)
            for (int i = 0; i < 64; ++i) {
                uint32_t x = 0;
                if (rowvalid.b >> i & 1)
                    for (int j = 0; j < 4; ++j)
                        if (fast || (uint32_t)(4 * k + j) < ncols.v[i]) x |= (uint32_t)base[(size_t)off.v[i] + 4 * k + j] << (8 * j);
                out[k].v[i] = x;
            }
        g_ops += 4;
    }
    void rec_write(const u32 &half, const u32 &id, const u32 (&f)[8], mask m) { for (int i = 0; i < 64; ++i) if (m.b >> i & 1) for (int k = 0; k < 8; ++k) lds_rec[half.v[i]][id.v[i]][k] = f[k].v[i]; g_ops += 2; }
    void rec_read(const u32 &half, const u32 &id, u32 (&f)[8], mask m) const { for (int i = 0; i < 64; ++i) for (int k = 0; k < 8; ++k) f[k].v[i] = (m.b >> i & 1) ? lds_rec[half.v[i]][id.v[i]][k] : 0u; g_ops += 2; }
    void rec_set_par(const u32 &half, const u32 &id, const u32 &val, mask m) { for (int i = 0; i < 64; ++i) if (m.b >> i & 1) { if (id.v[i] >= (uint32_t)t2::REC_CAP) { puts("rec_set_par: id out of range"); exit(2); } lds_rec[half.v[i]][id.v[i]][0] = val.v[i]; } g_ops += 2; }
    void idmap_write(const u32 &half, const u32 &pix, const u32 &id, mask m) { for (int i = 0; i < 64; ++i) if (m.b >> i & 1) lds_idmap[half.v[i]][pix.v[i]] = (uint16_t)id.v[i]; g_ops += 2; }
    u32 idmap_read(const u32 &half, const u32 &pix, mask m) const { u32 r; for (int i = 0; i < 64; ++i) r.v[i] = (m.b >> i & 1) ? lds_idmap[half.v[i]][pix.v[i] & 2047u] : 0u; g_ops += 2; return r; }
    uint32_t atomic_add(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p += v; return o; }
    void atomic_or(uint32_t *p, uint32_t v) { *p |= v; }
    template <class T> void store_scalar(T *p, T v) { *p = v; }
    void store_u16(uint16_t *p, const u32 &idx, const u32 &val, mask m) { for (int i = 0; i < 64; ++i) if (m.b >> i & 1) p[idx.v[i]] = (uint16_t)val.v[i]; g_ops += 1; }
    void store_u32(uint32_t *p, const u32 &idx, const u32 &val, mask m) { for (int i = 0; i < 64; ++i) if (m.b >> i & 1) p[idx.v[i]] = val.v[i]; g_ops += 1; }
    void store_rec(NodeRec *p, const u32 &idx, const u32 (&f)[8], mask m) { for (int i = 0; i < 64; ++i) if (m.b >> i & 1) { uint32_t t[8]; for (int k = 0; k < 8; ++k) t[k] = f[k].v[i]; std::memcpy(&p[idx.v[i]], t, 32); } g_ops += 2; }
};

// ---------------------------------------------------------------- brute force
struct RefNode {
    uint32_t level, key;            // key: plane-linear index of the smallest own-level pixel
    int      parent = -1;           // index into the tile's node list
    uint32_t own = 0, total = 0, nodes_sub = 1;
    uint32_t sides = 0;
    int      x0 = 1 << 30, y0 = 1 << 30, x1 = -1, y1 = -1;       // box of the whole component
    std::vector<int> children;
    // folded (what a record holds)
    uint32_t f_cnt = 0, f_nod = 0;
    int      fx0, fy0, fx1, fy1;
    bool     exported = false;
};

struct Plane {
    int w, h, stride, invert, step, hi, min_area;
    std::vector<uint8_t> pix;
    int tiles_x, tiles_y;
    int level(int x, int y) const { return (int)lrintf((float)(pix[(size_t)y * stride + x] ^ invert) * (float)(1.0 / step)); }
};

struct TileRef {
    std::vector<RefNode> nodes;
    std::map<std::pair<int, int>, int> node_of_pixel;      // (x, y) plane coordinates -> node index
    int start_node = -1;
};

static TileRef brute(const Plane &P, int tx, int ty)
{
    TileRef R;
    const int ox = tx * 64, oy = ty * 32, tw = std::min(64, P.w - ox), th = std::min(32, P.h - oy);
    std::vector<int> lev(tw * th);
    std::vector<int> present;
    for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x) {
            const int l = P.level(ox + x, oy + y);
            lev[y * tw + x] = l >= P.hi ? -1 : l;
            if (l < P.hi) present.push_back(l);
        }
    std::sort(present.begin(), present.end());
    present.erase(std::unique(present.begin(), present.end()), present.end());
    std::vector<int> top(tw * th, -1);      // node index of the highest node built so far whose component holds the pixel
    const uint32_t smask = (ty > 0 ? 1u : 0u) | (ty + 1 < P.tiles_y ? 2u : 0u) | (tx > 0 ? 4u : 0u) | (tx + 1 < P.tiles_x ? 8u : 0u);
    for (int t : present) {
        std::vector<int> lab(tw * th, -1);
        for (int s = 0; s < tw * th; ++s) {
            if (lev[s] < 0 || lev[s] > t || lab[s] >= 0) continue;
            std::vector<int> comp;
            std::queue<int>  q;
            q.push(s); lab[s] = s;
            while (!q.empty()) {
                const int p = q.front(); q.pop();
                comp.push_back(p);
                const int x = p % tw, y = p / tw;
                const int nb[4] = {x > 0 ? p - 1 : -1, x + 1 < tw ? p + 1 : -1, y > 0 ? p - tw : -1, y + 1 < th ? p + tw : -1};
                for (int n : nb) if (n >= 0 && lev[n] >= 0 && lev[n] <= t && lab[n] < 0) { lab[n] = s; q.push(n); }
            }
            bool has_own = false;
            for (int p : comp) has_own |= lev[p] == t;
            if (!has_own) continue;
            RefNode n;
            n.level = (uint32_t)t; n.key = 0xFFFFFFFFu;
            const int me = (int)R.nodes.size();
            std::vector<int> kids;
            for (int p : comp) {
                const int x = p % tw, y = p / tw;
                if (lev[p] == t) { ++n.own; n.key = std::min(n.key, (uint32_t)((oy + y) * P.w + ox + x)); R.node_of_pixel[{ox + x, oy + y}] = me; }
                ++n.total;
                n.x0 = std::min(n.x0, ox + x); n.x1 = std::max(n.x1, ox + x); n.y0 = std::min(n.y0, oy + y); n.y1 = std::max(n.y1, oy + y);
                if (y == 0) n.sides |= 1u; if (y == 31) n.sides |= 2u; if (x == 0) n.sides |= 4u; if (x == 63) n.sides |= 8u;
                if (top[p] >= 0 && top[p] != me) { if (std::find(kids.begin(), kids.end(), top[p]) == kids.end()) kids.push_back(top[p]); }
                top[p] = me;
            }
            n.sides &= smask;
            n.children = kids;
            R.nodes.push_back(n);
            for (int k : kids) R.nodes[k].parent = me;
        }
    }
    // totals, bottom-up (children have smaller indices)
    for (size_t i = 0; i < R.nodes.size(); ++i) {
        RefNode &n = R.nodes[i];
        n.nodes_sub = 1;
        for (int k : n.children) n.nodes_sub += R.nodes[k].nodes_sub;
    }
    // start pixel (tile 0 only)
    if (tx == 0 && ty == 0) {
        auto ok = [&](int x, int y) { return x < P.w && y < P.h && P.level(x, y) < P.hi; };
        int sx = -1, sy = -1;
        if (ok(0, 0)) { sx = 0; sy = 0; }
        else if (P.w > 1 && ok(1, 0)) { sx = 1; sy = 0; }
        else if (P.h > 1 && ok(0, 1)) { sx = 0; sy = 1; }
        if (sx >= 0) R.start_node = R.node_of_pixel[{sx, sy}];
    }
    // folded records: own pixels + closed children (whole subtrees), computed from the component minus its open children's components
    for (size_t i = 0; i < R.nodes.size(); ++i) {
        RefNode &n = R.nodes[i];
        n.f_cnt = n.total; n.f_nod = n.nodes_sub;
        for (int k : n.children) if (R.nodes[k].sides) { n.f_cnt -= R.nodes[k].total; n.f_nod -= R.nodes[k].nodes_sub; }
        const bool open = n.sides != 0;
        n.exported = open || (int64_t)(n.total + n.nodes_sub) > (int64_t)P.min_area || n.parent < 0 || (int)i == R.start_node;
    }
    // boxes of the folded part: pixels of the component that are in no open child's component
    for (size_t i = 0; i < R.nodes.size(); ++i) {
        RefNode &n = R.nodes[i];
        n.fx0 = n.fy0 = 1 << 30; n.fx1 = n.fy1 = -1;
    }
    {
        // every pixel belongs to the folded part of exactly the nodes on its ancestor chain up to (and including) the first OPEN node above... walk it
        for (auto &kv : R.node_of_pixel) {
            int n = kv.second;
            const int x = kv.first.first, y = kv.first.second;
            for (;;) {
                RefNode &nd = R.nodes[n];
                nd.fx0 = std::min(nd.fx0, x); nd.fx1 = std::max(nd.fx1, x); nd.fy0 = std::min(nd.fy0, y); nd.fy1 = std::max(nd.fy1, y);
                if (nd.sides || nd.parent < 0) break;      // an open node's pixels are not folded into its parent
                n = nd.parent;
            }
        }
    }
    return R;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

static Plane make_plane(int kind, int w, int h, int step, int min_area, bool invert)
{
    Plane P;
    P.w = w; P.h = h; P.step = step; P.hi = 255 / step + 1; P.min_area = min_area; P.invert = invert ? 0xFF : 0;
    P.stride = (kind & 8) ? w : (w + 63) / 64 * 64;
    P.tiles_x = (w + 63) / 64; P.tiles_y = (h + 31) / 32;
    P.pix.assign((size_t)P.stride * h + 64, 0);
    const int base = 100 + (int)(rnd() % 60);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int v = base;
            switch (kind & 7) {
            case 0: v = base + (int)(rnd() % 9) - 4; break;                                              // speckles around a level boundary
            case 1: v = base + ((x / 7 + y / 5) % 3) * step + (int)(rnd() % 5) - 2; break;               // blocks
            case 2: v = (int)(rnd() % 256); break;                                                       // noise (hits the sentinel)
            case 3: v = base + (x * 3 * step) / std::max(w, 1) + (rnd() % 16 == 0 ? step : 0); break;     // ramp + bright speckles
            case 4: v = ((x ^ y) & 4) ? base : base + step + (int)(rnd() % 3); break;                    // checker blocks
            case 5: v = (rnd() % 50 == 0) ? 255 : base + (int)(rnd() % (2 * step)); break;               // walls sprinkled in
            case 6: v = ((x % 9) == 4 || (y % 11) == 5) ? 254 : base + ((x / 9 + y / 11) % 2) * step; break;      // a lattice of walls: closed cells (tile roots)
            default: v = base + ((y & 1) ? step : 0) + ((x % 13) == 0 ? -step : 0); break;               // stripes
            }
            P.pix[(size_t)y * P.stride + x] = (uint8_t)std::min(255, std::max(0, v));
        }
    for (size_t i = 0; i < P.pix.size(); ++i) if ((int)(i % P.stride) >= w) P.pix[i] = (uint8_t)rnd();     // padding holds garbage
    return P;
}

struct RunOut {
    std::vector<NodeRec>  rec;
    std::vector<uint32_t> aux, tile_nbase, fb_list;
    std::vector<uint16_t> tile_nrec, seam;
    PlaneCtr ctr;
    uint32_t fb_count = 0;
};

static void run_body(const Plane &P, RunOut &o, unsigned long long *ops_out)
{
    PlaneDesc pd{};
    pd.pix = P.pix.data(); pd.w = P.w; pd.h = P.h; pd.stride = P.stride; pd.invert = P.invert;
    pd.tiles_x = P.tiles_x; pd.tiles_y = P.tiles_y; pd.tile_base = 0; pd.node_base = 0; pd.seam_base = 0;
    pd.node_cap = (uint32_t)(P.tiles_x * P.tiles_y * 2048);
    const uint32_t n_tiles = (uint32_t)(P.tiles_x * P.tiles_y);
    o.rec.assign(pd.node_cap, NodeRec{}); o.aux.assign(pd.node_cap, 0xDEADBEEFu);
    o.tile_nbase.assign(n_tiles, 0xABABABABu); o.tile_nrec.assign(n_tiles, 0xABAB); o.fb_list.assign(n_tiles, 0);
    o.seam.assign((size_t)2 * P.w * std::max(0, P.tiles_y - 1) + (size_t)2 * P.h * std::max(0, P.tiles_x - 1) + 8, 0xEEEE);
    std::memset(&o.ctr, 0, sizeof(o.ctr));
    std::vector<uint16_t> tile_plane(n_tiles, 0);
    BatchDev b{};
    b.planes = &pd; b.ctr = &o.ctr; b.n_planes = 1; b.n_tiles = n_tiles; b.tile_plane = tile_plane.data();
    b.na.rec = o.rec.data(); b.na.aux = o.aux.data(); b.tile_nbase = o.tile_nbase.data(); b.tile_nrec = o.tile_nrec.data(); b.seam = o.seam.data();
    DetectParams prm{};
    prm.thresh_step = P.step; prm.min_area = P.min_area; prm.hi = P.hi;
    std::vector<uint32_t> pairs;
    for (int ty = 0; ty < P.tiles_y; ++ty)
        for (int tx = 0; tx < P.tiles_x; tx += 2) pairs.push_back((uint32_t)(ty * P.tiles_x + tx) | (tx + 1 < P.tiles_x ? 0x80000000u : 0u));
    t2::Args a{pairs.data(), (uint32_t)pairs.size(), o.fb_list.data(), &o.fb_count};
    const unsigned long long ops0 = g_ops;
    for (uint32_t p = 0; p < pairs.size(); ++p) {
        HostWave w;
        std::memset(w.lds_idmap, 0xCD, sizeof(w.lds_idmap));
        t2::Body<HostWave>::run(w, b, prm, a, p);
    }
    if (ops_out) *ops_out = g_ops - ops0;
}

static int check_plane(const Plane &P, const char *what, unsigned long long *ops_out, unsigned *fb_out)
{
    RunOut o;
    run_body(P, o, ops_out);
    std::vector<char> is_fb((size_t)P.tiles_x * P.tiles_y, 0);
    for (uint32_t i = 0; i < o.fb_count; ++i) is_fb[o.fb_list[i]] = 1;
    if (fb_out) *fb_out = o.fb_count;
    int errors = 0;
    auto fail = [&](int tx, int ty, const char *msg, long a = 0, long b2 = 0) {
        if (errors < 12) printf("  [%s %dx%d step %d] tile (%d,%d): %s (%ld, %ld)\n", what, P.w, P.h, P.step, tx, ty, msg, a, b2);
        ++errors;
    };
    uint32_t total_recs = 0, walls = 0;
    const uint32_t voff = 2u * P.w * (P.tiles_y - 1);
    for (int ty = 0; ty < P.tiles_y; ++ty)
        for (int tx = 0; tx < P.tiles_x; ++tx) {
            const int tile = ty * P.tiles_x + tx;
            if (is_fb[tile]) continue;
            const TileRef R = brute(P, tx, ty);
            for (int y = ty * 32; y < std::min(P.h, ty * 32 + 32); ++y) for (int x = tx * 64; x < std::min(P.w, tx * 64 + 64); ++x) walls += P.level(x, y) >= P.hi;
            std::map<std::pair<uint32_t, uint32_t>, int> want;       // (level, key) -> node
            for (size_t i = 0; i < R.nodes.size(); ++i) if (R.nodes[i].exported) want[{R.nodes[i].level, R.nodes[i].key}] = (int)i;
            const uint32_t nb = o.tile_nbase[tile], nr = o.tile_nrec[tile];
            total_recs += nr;
            if (nr != want.size()) { fail(tx, ty, "record count", nr, (long)want.size()); continue; }
            std::map<int, uint32_t> rec_of_node;
            for (uint32_t r = 0; r < nr; ++r) {
                const NodeRec &q = o.rec[nb + r];
                const auto it = want.find({q.key >> 24, q.key & 0xFFFFFFu});
                if (it == want.end()) { fail(tx, ty, "record of no expected node", q.key >> 24, q.key & 0xFFFFFF); continue; }
                rec_of_node[it->second] = r;
            }
            if (rec_of_node.size() != want.size()) { fail(tx, ty, "records do not cover the expected nodes"); continue; }
            for (auto &kv : rec_of_node) {
                const RefNode &n = R.nodes[kv.first];
                const NodeRec &q = o.rec[nb + kv.second];
                if (o.aux[nb + kv.second] != 0) fail(tx, ty, "aux not zeroed");
                if (q.cnt != n.f_cnt) fail(tx, ty, "cnt", q.cnt, n.f_cnt);
                if ((q.nod & NODE_CNT) != n.f_nod) fail(tx, ty, "nod", q.nod & NODE_CNT, n.f_nod);
                const uint32_t flags = n.sides ? n.sides << 26 : NODE_CLOSED;
                if ((q.nod & ~NODE_CNT) != flags) fail(tx, ty, "flags", q.nod >> 24, flags >> 24);
                if ((int)q.x0 != n.fx0 || (int)q.x1 != n.fx1 || (int)q.y0 != n.fy0 || (int)q.y1 != n.fy1) fail(tx, ty, "box", q.x0 * 10000 + q.y0, n.fx0 * 10000 + n.fy0);
                int pa = n.parent;
                while (pa >= 0 && !R.nodes[pa].exported) pa = R.nodes[pa].parent;
                if (pa < 0) { if (q.par != NONE) fail(tx, ty, "par of a root", q.par); }
                else if (q.par == NONE || (q.par >> 24) != R.nodes[pa].level || (q.par & 0xFFFFFFu) != nb + rec_of_node[pa]) fail(tx, ty, "par", q.par, rec_of_node[pa]);
            }
            if (tx == 0 && ty == 0) {
                const uint32_t sn = o.ctr.start_node;
                if (R.start_node < 0) { if (sn != NONE) fail(tx, ty, "start node given, none expected", sn); }
                else if (sn != nb + rec_of_node[R.start_node]) fail(tx, ty, "start node", sn, nb + rec_of_node[R.start_node]);
            }
            // seam map
            auto expect = [&](int x, int y) -> uint16_t {
                const auto it = R.node_of_pixel.find({x, y});
                if (it == R.node_of_pixel.end()) return 0xFFFF;
                return (uint16_t)rec_of_node[it->second];
            };
            for (int x = tx * 64; x < std::min(P.w, tx * 64 + 64); ++x) {
                if (ty > 0 && o.seam[(size_t)((ty - 1) * 2 + 1) * P.w + x] != expect(x, ty * 32)) fail(tx, ty, "seam top", x, o.seam[(size_t)((ty - 1) * 2 + 1) * P.w + x]);
                if (ty + 1 < P.tiles_y && o.seam[(size_t)(ty * 2) * P.w + x] != expect(x, ty * 32 + 31)) fail(tx, ty, "seam bottom", x, o.seam[(size_t)(ty * 2) * P.w + x]);
            }
            for (int y = ty * 32; y < std::min(P.h, ty * 32 + 32); ++y) {
                if (tx > 0 && o.seam[voff + (size_t)((tx - 1) * 2 + 1) * P.h + y] != expect(tx * 64, y)) fail(tx, ty, "seam left", y, o.seam[voff + (size_t)((tx - 1) * 2 + 1) * P.h + y]);
                if (tx + 1 < P.tiles_x && o.seam[voff + (size_t)(tx * 2) * P.h + y] != expect(tx * 64 + 63, y)) fail(tx, ty, "seam right", y, o.seam[voff + (size_t)(tx * 2) * P.h + y]);
            }
        }
    if (o.ctr.n_nodes != total_recs) { printf("  [%s] n_nodes %u, tiles hold %u\n", what, o.ctr.n_nodes, total_recs); ++errors; }
    if (o.ctr.n_walls != walls) { printf("  [%s] n_walls %u, expected %u\n", what, o.ctr.n_walls, walls); ++errors; }
    return errors;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 60;
    if (argc > 4) {
        // a real plane (tools/sim_tile_dump.py writes quantised levels, one byte per pixel, 255 = wall): cost per tile
        const int W = atoi(argv[3]), H = atoi(argv[4]);
        Plane P;
        P.w = W; P.h = H; P.step = 8; P.hi = 32; P.min_area = 120; P.invert = 0; P.stride = (W + 63) / 64 * 64; P.tiles_x = (W + 63) / 64; P.tiles_y = (H + 31) / 32;
        P.pix.assign((size_t)P.stride * H + 64, 0);
        std::vector<uint8_t> lev((size_t)W * H);
        FILE *f = fopen(argv[2], "rb");
        if (!f || fread(lev.data(), 1, lev.size(), f) != lev.size()) { puts("cannot read the plane"); return 1; }
        fclose(f);
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) P.pix[(size_t)y * P.stride + x] = lev[(size_t)y * W + x] == 255 ? 255 : (uint8_t)(lev[(size_t)y * W + x] * 8);
        unsigned long long ops = 0; unsigned fbn = 0;
        const int e = check_plane(P, "file", &ops, &fbn);
        const double tiles = (double)P.tiles_x * P.tiles_y;
        printf("%s: %d errors, %.0f tiles, %u to the fall-back, %.0f vector ops per tile = %.0f per 512 pixels\n", argv[2], e, tiles, fbn, ops / tiles, ops / tiles / 4.0);
        HostWave::mark(0);
        const char *names[8] = {"load+quantise", "walls+presence+setup", "level masks", "seed", "flood", "node stats", "export node", "tile export"};
        printf("   growth rounds %.1f, vfills %.1f per pair; floods by rounds:", g_stat[0] / (tiles / 2), g_stat[1] / (tiles / 2));
        for (int i = 0; i < 16; ++i) printf(" %d:%.2f", i, g_stat[8 + i] / (tiles / 2));
        printf("\n");
        for (int i = 0; i < 8; ++i) printf("   %-22s %8.0f ops per tile, entered %.1f times per pair\n", names[i], g_phase_ops[i] / tiles, g_phase_n[i] / (tiles / 2));
        return e != 0;
    }
    int errors = 0, planes = 0;
    unsigned long long ops_all = 0, tiles_all = 0, fb_all = 0;
    const int sizes[][2] = {{64, 32}, {128, 32}, {128, 64}, {1, 1}, {3, 2}, {65, 33}, {200, 70}, {130, 97}, {64, 1}, {1, 40}, {191, 32}, {256, 96}, {63, 31}, {300, 45}};
    for (int r = 0; r < rounds; ++r)
        for (int kind = 0; kind < 16; ++kind) {
            const int *sz = sizes[(r * 16 + kind) % (int)(sizeof(sizes) / sizeof(sizes[0]))];
            const int  step = (r & 3) == 3 ? 16 : ((r % 7) == 5 ? 32 : 8);
            const int  min_area = (r % 5 == 0) ? 1 : ((r % 5 == 1) ? 20 : 120);
            const Plane P = make_plane(kind, sz[0], sz[1], step, min_area, (r & 1) != 0);
            char what[64];
            snprintf(what, sizeof what, "kind %d r %d min_area %d inv %d", kind, r, min_area, r & 1);
            unsigned long long ops = 0; unsigned fbn = 0;
            errors += check_plane(P, what, &ops, &fbn);
            ops_all += ops; tiles_all += (unsigned long long)P.tiles_x * P.tiles_y; fb_all += fbn;
            ++planes;
        }
    printf("tile2 model: %d planes, %llu tiles (%llu handed to the fall-back), %d errors; %.0f vector ops per tile on these\n", planes, tiles_all, fb_all, errors, (double)ops_all / (double)tiles_all);
    return errors != 0;
}
