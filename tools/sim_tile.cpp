// Lockstep SIMT model of k_tile_tree's connect rounds: counts wave iterations (proxy for VALU issue) under
// different edge schedules.  Usage: sim plane.lev W H variant
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
using namespace std;
static const uint32_t NONE = 0xFFFFFFFFu;
static const int TW = 64, TH = 32, TP = 2048, NTMAX = 256;
static int NWV = 4, NT = 256;        // waves per tile (SIM_WAVES=1|2|4: round 4's question -- what would ONE or TWO waves per tile cost?), lanes
struct Stats {
    double tiles = 0, edges[2] = {0, 0}, passes = 0, cas = 0, cas_lost = 0, iters[2] = {0, 0}, cost[2] = {0, 0}, pieces = 0, nodes = 0;
    double redundant2 = 0, lanepasses = 0, hops = 0;
};
struct Tile {
    uint16_t lev[TP];
    uint32_t par[TP];
};
struct Lane { bool active; uint32_t a, b, la, lb; uint32_t wa, wb; bool same; int hops_a, hops_b; };

static uint32_t find_(Tile &t, uint32_t &a, uint32_t la, int &hops)
{
    uint32_t wa = t.par[a];
    hops = 0;
    while ((wa >> 16) == la) {
        uint32_t nx = wa & 0xFFFF;
        uint32_t w2 = t.par[nx];
        if ((w2 >> 16) == la) t.par[a] = w2;
        a = nx; wa = w2; ++hops;
    }
    return wa;
}


// total order key of the grid edge: weight (max level), then class (implicit same-level H edge 0 < explicit H 1 < V 2), then position
static int VARIANT = 0;
static uint64_t treehash = 0;
struct EKey { uint32_t w, cls, pos; bool operator<(const EKey &o) const { return w != o.w ? w < o.w : cls != o.cls ? cls < o.cls : pos < o.pos; } };
static bool edge_exists(const Tile &t, int p, int q) { return t.lev[p] != 0xFFFF && t.lev[q] != 0xFFFF; }
static EKey hkey(const Tile &t, int p) { /* edge (p-1,p) */ const uint32_t a = t.lev[p - 1], b = t.lev[p]; return EKey{max(a, b), a == b ? 0u : 1u, (uint32_t)p}; }
static EKey vkey(const Tile &t, int p) { /* edge (p,p+TW) */ const uint32_t a = t.lev[p], b = t.lev[p + TW]; return EKey{max(a, b), 2u, (uint32_t)p}; }
// block with top-left pixel p (x < 63, y < 31): edges H(p+1), H(p+TW+1), V(p), V(p+1); all four must exist
static bool is_block_max(const Tile &t, int p, const EKey &k)
{
    if (!(edge_exists(t, p, p + 1) && edge_exists(t, p + TW, p + TW + 1))) return false;
    EKey e[4] = {hkey(t, p + 1), hkey(t, p + TW + 1), vkey(t, p), vkey(t, p + 1)};
    for (int i = 0; i < 4; ++i) if (k < e[i]) return false;
    return true;
}
static bool prune_h(const Tile &t, int p) { /* edge (p-1,p) */
    const int x = p % TW, y = p / TW; const EKey k = hkey(t, p);
    if (y > 0 && is_block_max(t, p - 1 - TW, k)) return true;
    if (y < TH - 1 && is_block_max(t, p - 1, k)) return true;
    return false;
}
static bool prune_v_closed(const Tile &t, int p) {
    const int x = p % TW; const uint32_t INF = 0x1FF;
    auto w = [&](int q) -> uint32_t { return (t.lev[q] == 0xFFFF || t.lev[q + TW] == 0xFFFF) ? INF : max<uint32_t>(t.lev[q], t.lev[q + TW]); };
    const uint32_t wx = w(p), wp = x > 0 ? w(p - 1) : INF, wn = x < TW - 1 ? w(p + 1) : INF;
    const bool keep = wx != INF && wx < wp && wx <= wn;
    return !keep;
}
static bool prune_v(const Tile &t, int p) { /* edge (p,p+TW) */
    if (VARIANT & 64) return prune_v_closed(t, p);
    const int x = p % TW; const EKey k = vkey(t, p);
    if (x > 0 && is_block_max(t, p - 1, k)) return true;
    if (x < TW - 1 && is_block_max(t, p, k)) return true;
    return false;
}

// variant bits: 1 = wave-dynamic schedule, 2 = combined single round, 4 = early-out for redundant cross-level edge, 8 = hand-out in list order,
// 16 = equal-level vertical edges first, 32 = block-maximum edges dropped (64: closed form for the vertical ones), 128 = the combined list sorted by
// edge weight (256: heaviest first).  The kernel as built: 1 + 2 + 8 + 32 + 64 = 107.  (128 / 256, text / noise plane: cost per wave 865 -> 902 / 921,
// 3488 -> 3516 / 3227: the order of the edges is not worth a sort.)


static void run_rounds(Tile &t, vector<uint16_t> *elist_round, int nrounds, const int *round_kind_of_list, Stats &st)
{
    for (int r = 0; r < nrounds; ++r) {
        vector<uint32_t> &dummy = *new vector<uint32_t>();
        (void)dummy;
        const vector<uint16_t> &el = elist_round[r];
        // entries: p | kind<<15?  we store kind separately: el_kind
        const uint32_t n = el.size();
        Lane L[NTMAX];
        uint32_t next[NTMAX], nend[NTMAX];
        uint32_t wcur[4], wend[4];
        for (int i = 0; i < NT; ++i) { L[i].active = false; next[i] = (uint32_t)((uint64_t)i * n / NT); nend[i] = (uint32_t)((uint64_t)(i + 1) * n / NT); }
        for (int w = 0; w < NWV; ++w) { wcur[w] = 0; wend[w] = nend[w * 64 + 63] - next[w * 64]; }
        bool done[4] = {false, false, false, false};
        int ndone = 0;
        while (ndone < NWV) {
            // phase 1: refill + finds for every wave
            bool any[4];
            int maxha[4], maxhb[4];
            for (int w = 0; w < NWV; ++w) {
                any[w] = false; maxha[w] = maxhb[w] = 0;
                if (done[w]) continue;
                const uint32_t base = next[w * 64] - 0; (void)base;
                for (int l = 0; l < 64; ++l) {
                    const int i = w * 64 + l;
                    Lane &ln = L[i];
                    if (!ln.active) {
                        uint32_t idx = NONE;
                        if (VARIANT & 1) {
                            // wave-dynamic: cursor over the wave's range, spread: c -> (c % 64) * ceil(m/64) + c / 64
                            const uint32_t w0 = (uint32_t)((uint64_t)(w * 64) * n / NT), w1 = (uint32_t)((uint64_t)(w * 64 + 64) * n / NT);
                            const uint32_t m = w1 - w0, per = (m + 63) / 64;
                            while (wcur[w] < per * 64) {
                                const uint32_t c = wcur[w]++;
                                const uint32_t e = (VARIANT & 8) ? c : (c % 64) * per + c / 64;
                                if (e < m) { idx = w0 + e; break; }
                            }
                        } else {
                            if (next[i] < nend[i]) idx = next[i]++;
                        }
                        if (idx != NONE) {
                            const uint32_t p = el[idx] & 0x7FFF; const int kind = el[idx] >> 15;
                            if (kind == 0) { ln.a = p - 1; ln.b = p; } else { ln.a = p; ln.b = p + TW; }
                            ln.la = t.lev[ln.a]; ln.lb = t.lev[ln.b]; ln.active = true;
                        }
                    }
                    if (ln.active) {
                        any[w] = true;
                        ln.wa = find_(t, ln.a, ln.la, ln.hops_a);
                        ln.wb = find_(t, ln.b, ln.lb, ln.hops_b);
                        maxha[w] = max(maxha[w], ln.hops_a); maxhb[w] = max(maxhb[w], ln.hops_b);
                        st.hops += ln.hops_a + ln.hops_b;
                    }
                }
            }
            // phase 2: CAS for every wave
            for (int w = 0; w < NWV; ++w) {
                if (done[w]) continue;
                if (!any[w]) { done[w] = true; ++ndone; continue; }
                st.iters[r] += 1;
                st.cost[r] += 45 + 8 * (maxha[w] + maxhb[w]);
                for (int l = 0; l < 64; ++l) {
                    Lane &ln = L[w * 64 + l];
                    if (!ln.active) continue;
                    st.lanepasses += 1;
                    uint32_t a = ln.a, b = ln.b, la = ln.la, lb = ln.lb, wa = ln.wa, wb = ln.wb;
                    const bool same = a == b;
                    if (la > lb || (la == lb && a < b)) { swap(a, b); swap(la, lb); wa = wb; }
                    bool link = !same && (la == lb || (wa >> 16) > lb);
                    if ((VARIANT & 4) && !same && !link && (wa >> 16) == lb && (wa & 0xFFFF) == b) {
                        // a's root already hangs under b's root: done
                        ln.active = false; st.redundant2 += 1; st.passes += 1; continue;
                    }
                    bool ok = true;
                    if (link) {
                        st.cas += 1;
                        if (t.par[a] == wa) t.par[a] = (lb << 16) | b; else { ok = false; st.cas_lost += 1; }
                    }
                    if (!same && ok) { ln.a = wa & 0xFFFF; ln.la = wa >> 16; ln.b = b; ln.lb = lb; }
                    else { ln.a = a; ln.la = la; ln.b = b; ln.lb = lb; }
                    st.passes += 1;
                    ln.active = !(same || (link && ok && wa == NONE));
                }
            }
        }
    }
}

int main(int argc, char **argv)
{
    const char *fn = argv[1];
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    VARIANT = argc > 4 ? atoi(argv[4]) : 0;
    if (getenv("SIM_WAVES")) { NWV = atoi(getenv("SIM_WAVES")); if (NWV != 1 && NWV != 2 && NWV != 4) NWV = 4; NT = 64 * NWV; }
    vector<uint8_t> img((size_t)W * H);
    FILE *f = fopen(fn, "rb"); if (!f || fread(img.data(), 1, img.size(), f) != img.size()) { perror("read"); return 1; } fclose(f);
    Stats st;
    const int tx_n = (W + TW - 1) / TW, ty_n = (H + TH - 1) / TH;
    for (int ty = 0; ty < ty_n; ++ty) for (int tx = 0; tx < tx_n; ++tx) {
        Tile t;
        for (int y = 0; y < TH; ++y) for (int x = 0; x < TW; ++x) {
            const int gx = tx * TW + x, gy = ty * TH + y;
            uint16_t l = 0xFFFF;
            if (gx < W && gy < H) { const uint8_t v = img[(size_t)gy * W + gx]; l = v == 255 ? 0xFFFF : v; }
            t.lev[y * TW + x] = l;
        }
        // prelink runs
        vector<uint16_t> el[2];
        uint32_t npieces = 0;
        for (int y = 0; y < TH; ++y) {
            uint32_t head = 0;
            for (int x = 0; x < TW; ++x) {
                const int p = y * TW + x;
                const uint16_t l = t.lev[p];
                const bool wall = l == 0xFFFF;
                const bool cont = !wall && x > 0 && t.lev[p - 1] == l;
                if (!cont) head = p;
                t.par[p] = cont ? (((uint32_t)l << 16) | head) : NONE;
                if (!wall && (!cont || (x & 7) == 0)) ++npieces;
            }
        }
        auto start = [&](int p) { const int x = p % TW; return t.lev[p] != 0xFFFF && (x == 0 || t.lev[p - 1] != t.lev[p]); };
        for (int p = 0; p < TP; ++p) {
            const int x = p % TW;
            if (start(p) && x > 0 && t.lev[p - 1] != 0xFFFF && !((VARIANT & 32) && prune_h(t, p))) el[0].push_back((uint16_t)p);
        }
        for (int p = 0; p < TP - TW; ++p) {
            if (t.lev[p] != 0xFFFF && t.lev[p + TW] != 0xFFFF && ((VARIANT & 32) ? !prune_v(t, p) : (start(p) || start(p + TW)))) el[1].push_back((uint16_t)(p | 0x8000));
        }
        st.tiles += 1; st.edges[0] += el[0].size(); st.edges[1] += el[1].size(); st.pieces += npieces;
        if (VARIANT & 16) {
            vector<uint16_t> c0, c1;
            size_t i0 = 0, i1 = 0;
            for (int tid = 0; tid < NTMAX; ++tid) {
                const int p0 = tid * 8;
                while (i0 < el[0].size() && (el[0][i0] & 0x7FFF) < p0 + 8) c1.push_back(el[0][i0++]);
                while (i1 < el[1].size() && (el[1][i1] & 0x7FFF) < p0 + 8) { const int p = el[1][i1] & 0x7FFF; (t.lev[p] == t.lev[p + TW] ? c0 : c1).push_back(el[1][i1]); ++i1; }
            }
            vector<uint16_t> lists[2] = {c0, c1};
            int kinds[2] = {0, 0};
            run_rounds(t, lists, 2, kinds, st);
        } else if (VARIANT & 2) {
            // one combined list, interleaved by lane: H then V of each lane (tid order)
            vector<uint16_t> c;
            size_t i0 = 0, i1 = 0;
            for (int tid = 0; tid < NTMAX; ++tid) {
                const int p0 = tid * 8;
                while (i0 < el[0].size() && (el[0][i0] & 0x7FFF) < p0 + 8) c.push_back(el[0][i0++]);
                while (i1 < el[1].size() && (el[1][i1] & 0x7FFF) < p0 + 8) c.push_back(el[1][i1++]);
            }
            if (VARIANT & 128) {      // the list sorted by the edge's weight max(level, level) (stable; 256: heaviest first)
                auto wt = [&](uint16_t e) -> uint32_t { const int p = e & 0x7FFF; return (e & 0x8000) ? max(t.lev[p], t.lev[p + TW]) : max(t.lev[p - 1], t.lev[p]); };
                stable_sort(c.begin(), c.end(), [&](uint16_t x, uint16_t y) { return (VARIANT & 256) ? wt(x) > wt(y) : wt(x) < wt(y); });
            }
            vector<uint16_t> lists[1] = {c};
            int kinds[1] = {0};
            run_rounds(t, lists, 1, kinds, st);
        } else {
            int kinds[2] = {0, 1};
            run_rounds(t, el, 2, kinds, st);
        }
        // count nodes
        uint32_t nodes = 0;
        for (int p = 0; p < TP; ++p) if (t.lev[p] != 0xFFFF && (t.par[p] == NONE || (t.par[p] >> 16) != t.lev[p])) ++nodes;
        st.nodes += nodes;
        {   // canonical tree hash
            auto rootof = [&](uint32_t p) { uint32_t l = t.lev[p]; for (;;) { uint32_t w = t.par[p]; if (w == NONE || (w >> 16) != l) return p; p = w & 0xFFFF; } };
            // canonical id of a node = min pixel over its members
            vector<uint32_t> canon(TP, NONE);
            for (int p = 0; p < TP; ++p) if (t.lev[p] != 0xFFFF) { uint32_t r = rootof(p); if (canon[r] == NONE || (uint32_t)p < canon[r]) canon[r] = min(canon[r], (uint32_t)p); }
            for (int p = 0; p < TP; ++p) if (t.lev[p] != 0xFFFF) {
                uint32_t r = rootof(p);
                if (r != (uint32_t)p) { treehash += (uint64_t)(p + 1) * 1000003ull * (canon[r] + 7); continue; }
                uint32_t w = t.par[r]; uint64_t pc = 0;
                if (w != NONE) { uint32_t q = rootof(w & 0xFFFF); pc = canon[q] + 13; }
                treehash += (uint64_t)(canon[r] + 1) * 2654435761ull * (pc + 1) + t.lev[r];
            }
        }
    }
    const double T = st.tiles;
    printf("%s var %d waves %d: tiles %.0f pieces/tile %.0f nodes/tile %.0f edges H %.0f V %.0f | passes/connect %.2f cas lost %.1f%% hops/pass %.2f | wave-iters/wave: r0 %.2f r1 %.2f | cost/wave r0 %.0f r1 %.0f total %.0f (per TILE %.0f) | lane eff %.1f%% red2 %.0f/tile\n",
           fn, VARIANT, NWV, T, st.pieces / T, st.nodes / T, st.edges[0] / T, st.edges[1] / T, st.passes / (st.edges[0] + st.edges[1]), 100.0 * st.cas_lost / max(1.0, st.cas), st.hops / st.passes,
           st.iters[0] / T / NWV, st.iters[1] / T / NWV, st.cost[0] / T / NWV, st.cost[1] / T / NWV, (st.cost[0] + st.cost[1]) / T / NWV, (st.cost[0] + st.cost[1]) / T,
           100.0 * st.lanepasses / ((st.iters[0] + st.iters[1]) * 64), st.redundant2 / T);
    printf("  treehash %016llx\n", (unsigned long long)treehash);
    return 0;
}
