// er_classify.inl -- part of er_kernels.hip (included there, inside namespace str_er; not a translation unit of its own): classify: ARAN(26), Mean-LBP histogram, the two cascades.
// ------------------------------------------------------------------------------------
// classify (src/ER.cpp:507-528): ARAN(26) -> Mean-LBP 24x24 -> 2x2x256 histogram ->
// strong cascade, then weak cascade if rejected.  One workgroup per candidate.
// ------------------------------------------------------------------------------------
constexpr int CLS_THREADS = 256;
constexpr int CLS_CHUNK = 1024;

struct ClsShared {
    uint32_t hist[1024];
    double   vals[CLS_CHUNK];
    double   acc;
    uint8_t  tile[26 * 26 + 4];
};

// make_LBP_hist (src/ER.cpp:789-816) + calc_LBP (:819-845) + OCR::ARAN (src/OCR.cpp:394-430)
__device__ void block_lbp_hist(ClsShared &sh, const uint8_t *__restrict__ pix, int stride, int inv, int bx, int by,
                               int bw, int bh, uint8_t *__restrict__ codes = nullptr)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += CLS_THREADS) sh.hist[i] = 0;
    for (int i = tid; i < 26 * 26; i += CLS_THREADS) sh.tile[i] = 0;
    __syncthreads();
    const double R1 = (bw > bh) ? (double)bh / bw : (double)bw / bh;
    const int    k = (int)(26.0 * sqrt(R1));   // (int)(L * pow(R1, 0.5))
    const int    dw = (bw > bh) ? 26 : k, dh = (bw > bh) ? k : 26;
    if (dw > 0 && dh > 0) {
        const int offy = (dw > dh) ? (26 - dh) / 2 : 0;
        const int offx = (dw > dh) ? 0 : (26 - dw) / 2;
        const ResizeGeom g = resize_geom(bw, bh, dw, dh);
        const uint8_t *roi = pix + (size_t)by * stride + bx;
        for (int i = tid; i < dw * dh; i += CLS_THREADS) {
            const int dy = i / dw, dx = i - dy * dw;
            sh.tile[(dy + offy) * 26 + dx + offx] = (uint8_t)resize_px(g, roi, stride, inv, dx, dy);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 24 * 24; idx += CLS_THREADS) {
        const int i = idx / 24, j = idx - i * 24;
        const int cpos = (i + 1) * 26 + (j + 1);
        // the reference indexes the 26-wide tile with a row stride of 24
        const int v0 = sh.tile[cpos - 25], v1 = sh.tile[cpos - 24], v2 = sh.tile[cpos - 23], v3 = sh.tile[cpos + 1];
        const int v4 = sh.tile[cpos + 25], v5 = sh.tile[cpos + 24], v6 = sh.tile[cpos + 23], v7 = sh.tile[cpos - 1];
        const int sum = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;   // v > sum/8.0  <=>  8v > sum
        const int code = (8 * v0 > sum) | ((8 * v1 > sum) << 1) | ((8 * v2 > sum) << 2) | ((8 * v3 > sum) << 3) |
                         ((8 * v4 > sum) << 4) | ((8 * v5 > sum) << 5) | ((8 * v6 > sum) << 6) | ((8 * v7 > sum) << 7);
        atomicAdd(&sh.hist[(i / 12) * 512 + (j / 12) * 256 + code], 1u);
        if (codes) codes[idx] = (uint8_t)code;          // the Mat calc_LBP returns (src/ER.cpp:819-845)
    }
    __syncthreads();
}

// CascadeBoost::predict (src/adaboost.cpp:507-542).  Stump outputs are produced by all
// lanes; the stage sum is formed by one lane in file order so it is bit-identical to the
// reference's sequential `score_stage += ...`.
__device__ double block_cascade(ClsShared &sh, const CascadeDev &c)
{
    const int tid = threadIdx.x;
    int       off = 0;
    double    score = 0;
    for (int s = 0; s < c.n_stages; ++s) {
        const int n = c.stage_n[s];
        if (tid == 0) sh.acc = 0;
        for (int base = 0; base < n; base += CLS_CHUNK) {
            const int m = min(CLS_CHUNK, n - base);
            for (int j = tid; j < m; j += CLS_THREADS) {
                const int st = off + base + j;
                double    v = 0;
                if (st < c.n_stumps) {
                    const double fv = (double)sh.hist[c.dim[st]], d = c.dir[st];
                    v = (fv * d < c.thr[st] * d) ? c.vp[st] : c.vn[st];
                }
                sh.vals[j] = v;
            }
            __syncthreads();
            if (tid == 0) {
                double a = sh.acc;
                const int mm = min(m, max(0, c.n_stumps - off - base));
                for (int j = 0; j < mm; ++j) a += sh.vals[j];
                sh.acc = a;
            }
            __syncthreads();
        }
        score = sh.acc;
        __syncthreads();
        if (score < (double)c.stage_thresh[s]) return -DBL_MAX;
        off += n;
    }
    return score;
}

// Batched classify: one workgroup takes 64 pooled ERs.
//   phase 1: each of the 16 waves builds the LBP histograms of 4 of them (wave per ER) and stores
//            them as 1028-byte rows of 8-bit counts in LDS (1028 = 1024 + 4: lane j reading
//            row j, column d hits bank (257 j + d/4) mod 32 -- conflict free);
//   phase 2: ONE wave scores all 64 with one ER per lane: every lane walks the stumps in file
//            order and adds in that order, so each stage sum is bit-identical to the
//            reference's sequential `score_stage +=` (src/adaboost.cpp:526-541), while the stump
//            parameters are wave-uniform (scalar loads).
constexpr int CLS_ROW = 1028;
constexpr int CLS_PER_BLOCK = 64;

constexpr int CLS64_WAVES = 16;
constexpr int CLS64_THREADS = CLS64_WAVES * 64;

struct Cls64Scratch {
    uint32_t hist[CLS64_WAVES][1024];
    uint8_t  tile[CLS64_WAVES][26 * 26 + 4];
};
struct Cls64Shared {
    uint8_t rows[CLS_PER_BLOCK * CLS_ROW];
    union {                              // phase 1 scratch, then the (A,B) tables of both cascades
        Cls64Scratch p1;
        double       ab[sizeof(Cls64Scratch) / sizeof(double)];
    } u;
    double stage_sum[CLS64_WAVES][64];   // phase 2: wave w leaves the sum of "its" stage for every ER here
};
constexpr int CLS_AB_CAP = (int)(sizeof(Cls64Scratch) / (2 * sizeof(double)));   // stumps that fit

__device__ __forceinline__ double readlane_f64(double v, int j)
{
    const unsigned long long u = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), j);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// One ER per lane.  The parameters of 64 stumps at a time are fetched with one coalesced vector
// load per field (lane i holds stump i) and broadcast with v_readlane; the lane's histogram bytes
// for 8 stumps are gathered from LDS together, so only the 8 adds are serial.
__device__ __forceinline__ double lane_cascade_generic(const CascadeDev &c, const uint8_t *row, bool valid)
{
    const int lane = threadIdx.x & 63;
    int    off = 0;
    bool   alive = valid;
    double score = 0;
    for (int s = 0; s < c.n_stages; ++s) {
        const int n = c.stage_n[s];
        double    acc = 0;
        const int m = min(n, max(0, c.n_stumps - off));
        for (int base = 0; base < m; base += 64) {
            const int  st = off + base + lane;
            const bool have = base + lane < m;
            StumpRec   r;
            r.dim = 0; r.mode = 0; r.thr = 0; r.vp = 0; r.vn = 0;
            if (have) r = c.rec[st];
            const double pr = (have && r.mode == 2) ? c.dir[st] : 1.0;
            const int    pd = r.dim | (r.mode << 16);
            const int    cnt = min(64, m - base);
            int j = 0;
            for (; j + 8 <= cnt; j += 8) {
                int    dj[8];
                double fv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { dj[u] = __builtin_amdgcn_readlane(pd, j + u); fv[u] = (double)row[dj[u] & 0xFFFF]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double t = readlane_f64(r.thr, j + u), vp = readlane_f64(r.vp, j + u), vn = readlane_f64(r.vn, j + u);
                    const int    mode = dj[u] >> 16;
                    bool         lt;
                    if (mode == 0) lt = fv[u] < t;
                    else if (mode == 1) lt = fv[u] > t;
                    else { const double d = readlane_f64(pr, j + u); lt = fv[u] * d < t * d; }
                    acc += lt ? vp : vn;
                }
            }
            for (; j < cnt; ++j) {
                const int    dw = __builtin_amdgcn_readlane(pd, j);
                const double fv = (double)row[dw & 0xFFFF];
                const double t = readlane_f64(r.thr, j), vp = readlane_f64(r.vp, j), vn = readlane_f64(r.vn, j);
                const int    mode = dw >> 16;
                bool         lt;
                if (mode == 0) lt = fv < t;
                else if (mode == 1) lt = fv > t;
                else { const double d = readlane_f64(pr, j); lt = fv * d < t * d; }
                acc += lt ? vp : vn;
            }
        }
        if (alive) {
            if (acc < (double)c.stage_thresh[s]) alive = false;
            else score = acc;
        }
        off += n;
        if (!__any(alive)) break;
    }
    return alive ? score : -DBL_MAX;
}

// Fast form for dir = +-1 cascades: histogram counts are integers, so `fv*dir < thr*dir` is the
// integer test h < T (T = ceil(thr), or floor(thr)+1 with the two outputs swapped for dir = -1).
// The (A,B) output pairs of all stumps sit in LDS (s_ab); the packed (dim, T) words of 64 stumps
// are fetched with one vector load and broadcast with v_readlane.  Adds stay in file order.
__device__ __forceinline__ double lane_cascade_fast(const CascadeDev &c, const uint8_t *row, const double *s_ab, bool valid)
{
    const int lane = threadIdx.x & 63;
    int    off = 0;
    bool   alive = valid;
    double score = 0;
    for (int s = 0; s < c.n_stages; ++s) {
        const int n = c.stage_n[s];
        double    acc = 0;
        const int m = min(n, max(0, c.n_stumps - off));
        for (int base = 0; base < m; base += 64) {
            const int pw = (base + lane < m) ? (int)c.w[off + base + lane] : 0;
            const int cnt = min(64, m - base);
            const double *ab = s_ab + 2 * (size_t)(off + base);
            int j = 0;
            for (; j + 8 <= cnt; j += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane(pw, j + u);
                    const uint32_t h = row[w & 1023u];
                    v[u] = ab[2 * (j + u) + (h < (w >> 10) ? 0 : 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; j < cnt; ++j) {
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane(pw, j);
                const uint32_t h = row[w & 1023u];
                acc += ab[2 * j + (h < (w >> 10) ? 0 : 1)];
            }
        }
        if (alive) {
            if (acc < (double)c.stage_thresh[s]) alive = false;
            else score = acc;
        }
        off += n;
        if (!__any(alive)) break;
    }
    return alive ? score : -DBL_MAX;
}

// The sum of ONE stage for the ER of every lane (same arithmetic as the stage loop above).  A cascade's stages do
// not feed each other -- stage s is "sum of its stumps >= thresh[s]" -- so the stages of both cascades can run on
// different waves at the same time; only the adds inside a stage are ordered.
__device__ __forceinline__ double lane_stage_fast(const CascadeDev &c, int s, const uint8_t *row, const double *s_ab)
{
    const int lane = threadIdx.x & 63;
    int       off = 0;
    for (int i = 0; i < s; ++i) off += c.stage_n[i];
    const int n = c.stage_n[s];
    const int m = min(n, max(0, c.n_stumps - off));
    double    acc = 0;
    // (round 4, measured in place with tools/dev_cls_trace.py: the stage sums were 55 % of a workgroup's 140 us -- 137 cycles per stump in the longest stage.
    // Two dependent waits went: the packed (dim, T) words of the NEXT 64 stumps are requested while this block is summed -- a trip to memory per block had
    // been waited for at its top --, and a stump's output pair (A, B) is read whole, at an address that does not depend on the histogram byte, and
    // picked afterwards: one trip to LDS per batch of 8 stumps instead of two.  Same adds in the same order.)
    int pw_next = (lane < m) ? (int)c.w[off + lane] : 0;
    for (int base = 0; base < m; base += 64) {
        const int pw = pw_next;
        if (base + 64 < m) pw_next = (base + 64 + lane < m) ? (int)c.w[off + base + 64 + lane] : 0;
        const int cnt = min(64, m - base);
        const double2 *ab = reinterpret_cast<const double2 *>(s_ab + 2 * (size_t)(off + base));
        int j = 0;
        if (cnt == 64) {
            // a full block: the reads of batch q + 1 are on their way while batch q is summed (the wave of a long stage is soon alone on its SIMD:
            // nobody else covers its trips to LDS)
            double2  pr[2][8];
            uint32_t hh[2][8], tt[2][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane(pw, u);
                hh[0][u] = row[w & 1023u]; tt[0][u] = w >> 10; pr[0][u] = ab[u];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q < 7) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane(pw, 8 * (q + 1) + u);
                        hh[(q + 1) & 1][u] = row[w & 1023u]; tt[(q + 1) & 1][u] = w >> 10; pr[(q + 1) & 1][u] = ab[8 * (q + 1) + u];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += hh[q & 1][u] < tt[q & 1][u] ? pr[q & 1][u].x : pr[q & 1][u].y;
            }
            j = 64;
        }
        for (; j + 8 <= cnt; j += 8) {
            double2  pr[8];
            uint32_t hh[8], tt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane(pw, j + u);
                hh[u] = row[w & 1023u];
                tt[u] = w >> 10;
                pr[u] = ab[j + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += hh[u] < tt[u] ? pr[u].x : pr[u].y;
        }
        for (; j < cnt; ++j) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane(pw, j);
            const uint32_t h = row[w & 1023u];
            const double2  p2 = ab[j];
            acc += h < (w >> 10) ? p2.x : p2.y;
        }
    }
    return acc;
}

// CascadeBoost::predict's stage loop (src/adaboost.cpp:507-542) over stage sums that are already there
__device__ __forceinline__ double cascade_from_stage_sums(const CascadeDev &c, const double *sums, int stride)
{
    double score = 0;
    for (int s = 0; s < c.n_stages; ++s) {
        const double acc = sums[(size_t)s * stride];
        if (acc < (double)c.stage_thresh[s]) return -DBL_MAX;
        score = acc;
    }
    return score;
}

// (list / n_list: only these candidates -- the planes whose pool the NMS tie pass changed)
// PER = candidates a workgroup takes at a time: 64 (a lane per candidate in the cascades), or 16 -- one per wave in the histogram phase instead of four
// in a row -- for calls with so few candidates (a frame or two: ~1000) that 64 a workgroup would leave most of the chip idle
template <int PER>
__global__ __launch_bounds__(CLS64_THREADS) void k_classify(BatchDev b, DetectParams prm, CascadeDev strong,
                                                          CascadeDev weak, int run_cascades, const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list)
{
    __shared__ Cls64Shared sh;
    const uint32_t total = list ? *n_list : *b.total_cands;
    const int      tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef STR_ER_WG_TRACE
#define CLS_MARK(i) do { if (tid == 0 && blockIdx.x % 8u == 0u && blockIdx.x / 8u < 128u) g_wg_trace[384 + blockIdx.x / 8u][(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CLS_MARK(i) do { } while (0)
#endif
    for (uint32_t c0 = blockIdx.x * PER; c0 < total; c0 += gridDim.x * PER) {
        CLS_MARK(0);
#ifdef STR_ER_PHASE_PROF
        const unsigned long long tp0 = wall_clock64();
#endif
        // ---- phase 1: histograms --------------------------------------------------------------
        // Every wave works on an ER of its own with scratch of its own (hist[wv], tile[wv], its row): between its steps it only has to wait for
        // ITSELF -- a wave's LDS operations are carried out in the order it issues them -- so the steps are separated by a wave-level fence, not
        // by a workgroup barrier: the 16 waves' ERs differ in size, and a barrier per step had every wave wait for the largest four times per ER.
        if (run_cascades) {
            for (int it = 0; it < PER / CLS64_WAVES; ++it) {
                const uint32_t cpos = c0 + it * CLS64_WAVES + wv;
                const bool     ok = cpos < total;
                const uint32_t cidx = ok && list ? list[cpos] : cpos;
                int bx = 0, by = 0, bw = 1, bh = 1, stride = 0, inv = 0;
                const uint8_t *pix = nullptr;
                if (ok) {
                    const int        pi = b.cand_plane[cidx];
                    const PlaneDesc &pd = b.planes[pi];
                    const uint32_t   slot = b.pool[pd.pool_base + (cidx - b.ctr[pi].cand_base)];
                    const size_t     ks = pd.kept_base + slot;
                    bx = b.ka.box[4 * ks]; by = b.ka.box[4 * ks + 1]; bw = b.ka.box[4 * ks + 2]; bh = b.ka.box[4 * ks + 3];
                    pix = pd.pix; stride = pd.stride; inv = pd.invert;
                }
                uint8_t  *tile = sh.u.p1.tile[wv];
                if (it == 0) CLS_MARK(5);
                uint32_t *rowq = reinterpret_cast<uint32_t *>(sh.rows + (size_t)(it * CLS64_WAVES + wv) * CLS_ROW);
                // The histogram is counted straight into the ER's packed row: a bin is a byte of it (a cell has 144 pixels, no count passes 255), so a pixel adds
                // 1 << 8 * (bin & 3) to the dword of its bin -- no carry leaves a byte.
                for (int k = 0; k < 4; ++k) rowq[lane + 64 * k] = 0;      // (a row starts on a dword, not on 16 bytes: CLS_ROW keeps the rows on different banks)
                for (int i = lane; i < (26 * 26 + 4) / 4; i += 64) reinterpret_cast<uint32_t *>(tile)[i] = 0;
                WAVE_SYNC();
                if (it == 0) CLS_MARK(6);
                if (ok) {
                    const double R1 = (bw > bh) ? (double)bh / bw : (double)bw / bh;
                    const int    k = (int)(26.0 * sqrt(R1));
                    const int    dw = (bw > bh) ? 26 : k, dh = (bw > bh) ? k : 26;
                    if (dw > 0 && dh > 0) {
                        const int offy = (dw > dh) ? (26 - dh) / 2 : 0;
                        const int offx = (dw > dh) ? 0 : (26 - dw) / 2;
                        const ResizeGeom g = resize_geom(bw, bh, dw, dh);
                        const uint8_t *roi = pix + (size_t)by * stride + bx;
                        if (__builtin_amdgcn_readfirstlane(g.mode) == 2) {
                            // The bilinear taps of the <= 26 x 26 tile are SEPARABLE: column dx fixes (sx, sx1, a0, a1), row dy fixes (y0, y1, b0, b1) -- cv::resize's own
                            // tables.  Lane dx < 32 makes the column entry, lane 32 + dy the row entry, ONCE per ER (resize_px's f64 / f32 arithmetic, unchanged);
                            // a pixel fetches its two entries by lane shuffle.  Round 4, measured in place (tools/dev_cls_trace.py): the resize was 14 k of an ER's
                            // 22 k cycles with every pixel redoing that arithmetic, eleven rounds per lane.
                            uint32_t tab0, tab1;
                            {
                                const bool isx = lane < 32;
                                const int  d = isx ? min(lane, dw - 1) : min(lane - 32, dh - 1);
                                float f = (float)((d + 0.5) * (isx ? g.scale_x : g.scale_y) - 0.5);
                                int   q = (int)floorf(f);
                                f -= (float)q;
                                if (isx) {
                                    if (q < 0) { f = 0.f; q = 0; }
                                    if (q >= g.sw - 1) { f = 0.f; q = g.sw - 1; }
                                }
                                const int c0 = __float2int_rn((1.f - f) * 2048.f), c1 = __float2int_rn(f * 2048.f);
                                const int p0 = isx ? q : min(max(q, 0), g.sh - 1);
                                const int p1 = isx ? ((q + 1 < g.sw) ? q + 1 : q) : min(max(q + 1, 0), g.sh - 1);
                                tab0 = (uint32_t)c1 | ((uint32_t)c0 << 16);
                                // columns: (sx, sx1); rows: the byte offset of row y0 from the box's corner, bit 31: y1 is the next row (not clamped onto y0)
                                tab1 = isx ? (uint32_t)p0 | ((uint32_t)p1 << 16) : (uint32_t)p0 * (uint32_t)stride | (p1 != p0 ? 0x80000000u : 0u);
                            }
                            const int      npx = dw * dh;
                            const uint32_t rcp_dw = (65536u + (uint32_t)dw - 1u) / (uint32_t)dw;
                            // the box's corner is the same for the whole wave: a scalar base, 32-bit lane offsets
                            typedef const __attribute__((address_space(1))) uint8_t *gbytes_t;
                            const uintptr_t roi_u = reinterpret_cast<uintptr_t>(roi);
                            const gbytes_t  groi = (gbytes_t)(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(roi_u >> 32)) << 32) |
                                                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)roi_u));
                            // This step is bound by instruction issue, 16 waves a CU (moving the box to LDS first, or 16-bit loads of tap pairs, made it slower): every
                            // product here fits 24 bits -- v_mul_u32_u24 is a full-rate instruction, the 32-bit multiply a quarter-rate one.
                            // Four rounds of taps in flight per lane: a round alone waits a full trip to memory (~1300 cycles measured), eleven in a row.
                            for (int i0 = 0; i0 < npx; i0 += 256) {
                                uint32_t xa[4], ya[4], t00[4], t01[4], t10[4], t11[4];
                                int      dst[4];
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const int      i = i0 + 64 * u + lane;
                                    const uint32_t ii = i < npx ? (uint32_t)i : 0u;
                                    const uint32_t dy = __umul24(ii, rcp_dw) >> 16, dx = ii - __umul24(dy, (uint32_t)dw);     // ii / dw: exact while ii * dw < 65536
                                    xa[u] = (uint32_t)__shfl((int)tab0, (int)dx);
                                    ya[u] = (uint32_t)__shfl((int)tab0, (int)(32u + dy));
                                    const uint32_t xb = (uint32_t)__shfl((int)tab1, (int)dx), yb = (uint32_t)__shfl((int)tab1, (int)(32u + dy));
                                    const uint32_t o0 = yb & 0x7FFFFFFFu, o1 = o0 + ((yb >> 31) ? (uint32_t)stride : 0u);
                                    const uint32_t sx = xb & 0xFFFFu, sx1 = xb >> 16;
                                    t00[u] = groi[o0 + sx], t01[u] = groi[o0 + sx1], t10[u] = groi[o1 + sx], t11[u] = groi[o1 + sx1];
                                    dst[u] = i < npx ? (int)(__umul24(dy + (uint32_t)offy, 26u) + dx + (uint32_t)offx) : -1;
                                }
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const uint32_t a0 = xa[u] >> 16, a1 = xa[u] & 0xFFFFu, b0 = ya[u] >> 16, b1 = ya[u] & 0xFFFFu;
                                    const uint32_t r0 = __umul24(t00[u] ^ (uint32_t)inv, a0) + __umul24(t01[u] ^ (uint32_t)inv, a1);        // <= 255 * 2049
                                    const uint32_t r1 = __umul24(t10[u] ^ (uint32_t)inv, a0) + __umul24(t11[u] ^ (uint32_t)inv, a1);
                                    const uint32_t v = ((__umul24(b0, r0 >> 4) >> 16) + (__umul24(b1, r1 >> 4) >> 16) + 2u) >> 2;           // (nothing is negative here)
                                    if (dst[u] >= 0) tile[dst[u]] = (uint8_t)min(v, 255u);
                                }
                            }
                        } else {
                            for (int i = lane; i < dw * dh; i += 64) {
                                const int dy = i / dw, dx = i - dy * dw;
                                tile[(dy + offy) * 26 + dx + offx] = (uint8_t)resize_px(g, roi, stride, inv, dx, dy);
                            }
                        }
                    }
                }
                WAVE_SYNC();
                if (it == 0) CLS_MARK(7);
                if (ok) {
                    for (int idx = lane; idx < 24 * 24; idx += 64) {
                        const int i = (int)(__umul24((uint32_t)idx, 2731u) >> 16), j = idx - (int)__umul24((uint32_t)i, 24u);      // idx / 24, exact below 576
                        const int cpos = (int)__umul24((uint32_t)i + 1u, 26u) + (j + 1);
                        const int v0 = tile[cpos - 25], v1 = tile[cpos - 24], v2 = tile[cpos - 23], v3 = tile[cpos + 1];
                        const int v4 = tile[cpos + 25], v5 = tile[cpos + 24], v6 = tile[cpos + 23], v7 = tile[cpos - 1];
                        const int sum = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
                        // bit k = (8 * v_k > sum) = the sign of sum - 8 * v_k, shifted in from the right, v7 first (v_alignbit: one instruction a bit)
                        uint32_t code = 0;
                        code = __builtin_amdgcn_alignbit(code, (uint32_t)(sum - 8 * v7), 31);
                        code = __builtin_amdgcn_alignbit(code, (uint32_t)(sum - 8 * v6), 31);
                        code = __builtin_amdgcn_alignbit(code, (uint32_t)(sum - 8 * v5), 31);
                        code = __builtin_amdgcn_alignbit(code, (uint32_t)(sum - 8 * v4), 31);
                        code = __builtin_amdgcn_alignbit(code, (uint32_t)(sum - 8 * v3), 31);
                        code = __builtin_amdgcn_alignbit(code, (uint32_t)(sum - 8 * v2), 31);
                        code = __builtin_amdgcn_alignbit(code, (uint32_t)(sum - 8 * v1), 31);
                        code = __builtin_amdgcn_alignbit(code, (uint32_t)(sum - 8 * v0), 31);
                        const uint32_t bin = (i >= 12 ? 512u : 0u) + (j >= 12 ? 256u : 0u) + code;
                        atomicAdd(&rowq[bin >> 2], 1u << (8u * (bin & 3u)));
                    }
                }
                WAVE_SYNC();
                if (it == 0) CLS_MARK(8);
                if (it == 0) CLS_MARK(9);
            }
            __syncthreads();            // every wave's rows are in place; the scratch (aliased by the tables below) is free
        }
        CLS_MARK(1);
#ifdef STR_ER_PHASE_PROF
        unsigned long long tp1 = wall_clock64();
        if (tid == 0) atomicAdd(&g_tile_phase[12], tp1 - tp0);
#endif
        // ---- phase 2: cascades, one ER per lane ---------------------------------------------------
        const bool fast = run_cascades && strong.all_unit && weak.all_unit && strong.n_stumps + weak.n_stumps <= CLS_AB_CAP;
        if (fast) {     // phase 1 is over (barrier above): reuse its scratch for the output tables
            for (int i = tid; i < 2 * strong.n_stumps; i += CLS64_THREADS) sh.u.ab[i] = strong.ab[i];
            for (int i = tid; i < 2 * weak.n_stumps; i += CLS64_THREADS) sh.u.ab[2 * strong.n_stumps + i] = weak.ab[i];
            __syncthreads();
        }
        CLS_MARK(2);
        // stage-parallel form: wave w < S + W sums stage w of the strong cascade or stage w - S of the weak one
        const bool par = fast && strong.n_stages + weak.n_stages <= CLS64_WAVES && strong.n_stages > 0 && weak.n_stages > 0;
        if (par) {
            if (wv < strong.n_stages + weak.n_stages) {
                const uint8_t *row = sh.rows + (size_t)lane * CLS_ROW;
                sh.stage_sum[wv][lane] = wv < strong.n_stages ? lane_stage_fast(strong, wv, row, sh.u.ab)
                                                              : lane_stage_fast(weak, wv - strong.n_stages, row, sh.u.ab + 2 * strong.n_stumps);
            }
            __syncthreads();
        }
        CLS_MARK(3);
        if (wv == 0) {
            const uint32_t cpos = c0 + lane;
            const bool     ok = cpos < total && lane < PER;
            const uint32_t cidx = ok && list ? list[cpos] : cpos;
            int    cls = 0;
            double ss = -DBL_MAX, sw = 0;
            if (run_cascades && par) {
                ss = cascade_from_stage_sums(strong, &sh.stage_sum[0][lane], 64);
                if (ss > -DBL_MAX) cls = 1;
                else {                              // the weak cascade only speaks for what the strong one rejected (src/ER.cpp:521-526)
                    sw = cascade_from_stage_sums(weak, &sh.stage_sum[strong.n_stages][lane], 64);
                    if (sw > -DBL_MAX) cls = 2;
                }
            } else if (run_cascades) {
                const uint8_t *row = sh.rows + (size_t)lane * CLS_ROW;
                ss = fast ? lane_cascade_fast(strong, row, sh.u.ab, ok) : lane_cascade_generic(strong, row, ok);
                const bool need_weak = ok && !(ss > -DBL_MAX);
                if (ss > -DBL_MAX) cls = 1;
                if (__any(need_weak)) {
                    const double w = fast ? lane_cascade_fast(weak, row, sh.u.ab + 2 * strong.n_stumps, need_weak)
                                          : lane_cascade_generic(weak, row, need_weak);
                    if (need_weak) { sw = w; if (sw > -DBL_MAX) cls = 2; }
                }
            }
            if (ok) {
                const int        pi = b.cand_plane[cidx];
                const PlaneDesc &pd = b.planes[pi];
                const uint32_t   slot = b.pool[pd.pool_base + (cidx - b.ctr[pi].cand_base)];
                const size_t     ks = pd.kept_base + slot;
                CandRec r;
                r.frame = pd.frame; r.ch = pd.ch; r.pyr = pd.pyr; r.level = b.ka.level[ks]; r.cls = (uint8_t)cls;
                r.x = b.ka.box[4 * ks]; r.y = b.ka.box[4 * ks + 1]; r.w = b.ka.box[4 * ks + 2]; r.h = b.ka.box[4 * ks + 3];
                r.area = b.ka.area[ks]; r.key = b.ka.key[ks]; r.node = (int32_t)slot; r.plane = (uint32_t)pi;
                r.score_strong = ss; r.score_weak = sw;
                b.cands[cidx] = r;
                if (cls == 1) atomicAdd(&b.ctr[pi].n_strong, 1u);
                if (cls == 2) atomicAdd(&b.ctr[pi].n_weak, 1u);
            }
        }
        CLS_MARK(4);
        __syncthreads();
    }
}

void launch_classify(hipStream_t s, const BatchDev &b, const DetectParams &p, CascadeDev strong, CascadeDev weak,
                     int run_cascades, const uint32_t *list, const uint32_t *n_list, bool few)
{
    static_assert(CLS_PER_BLOCK == 64 && CLS64_WAVES == 16, "the two builds: four candidates a wave, or one");
    if (few) hipLaunchKernelGGL(k_classify<16>, dim3(list ? 64 : 512), dim3(CLS64_THREADS), 0, s, b, p, strong, weak, run_cascades, list, n_list);
    else hipLaunchKernelGGL(k_classify<64>, dim3(list ? 64 : 1024), dim3(CLS64_THREADS), 0, s, b, p, strong, weak, run_cascades, list, n_list);
}

// Single-stage entry points (str_er_classify_boxes / str_er_lbp_hist): explicit boxes.
__global__ __launch_bounds__(CLS_THREADS) void k_lbp_boxes(const uint8_t *__restrict__ plane, int w, int h, int stride,
                                                           const int32_t *__restrict__ boxes, int n, double *hist,
                                                           uint8_t *tiles, uint8_t *codes, uint8_t *cls_out, double *s_strong,
                                                           double *s_weak, CascadeDev strong, CascadeDev weak,
                                                           int run_cascades)
{
    __shared__ ClsShared sh;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int bx = boxes[4 * i], by = boxes[4 * i + 1], bw = boxes[4 * i + 2], bh = boxes[4 * i + 3];
        block_lbp_hist(sh, plane, stride, 0, bx, by, bw, bh, codes ? codes + (size_t)i * 576 : nullptr);
        if (hist)
            for (int k = threadIdx.x; k < 1024; k += CLS_THREADS) hist[(size_t)i * 1024 + k] = (double)sh.hist[k];
        if (tiles)
            for (int k = threadIdx.x; k < 676; k += CLS_THREADS) tiles[(size_t)i * 676 + k] = sh.tile[k];
        if (run_cascades) {
            int    cls = 0;
            double ss = block_cascade(sh, strong), sw = 0;
            if (ss > -DBL_MAX) cls = 1;
            else {
                sw = block_cascade(sh, weak);
                if (sw > -DBL_MAX) cls = 2;
            }
            if (threadIdx.x == 0) { cls_out[i] = (uint8_t)cls; s_strong[i] = ss; s_weak[i] = sw; }
        }
        __syncthreads();
    }
}

void launch_lbp_boxes(hipStream_t s, const uint8_t *plane, int w, int h, int stride, const int32_t *boxes, int n,
                      double *hist, uint8_t *tiles, uint8_t *codes, uint8_t *cls, double *s_strong, double *s_weak, CascadeDev strong,
                      CascadeDev weak, int run_cascades)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_lbp_boxes, dim3(n < 2048 ? n : 2048), dim3(CLS_THREADS), 0, s, plane, w, h, stride, boxes, n,
                       hist, tiles, codes, cls, s_strong, s_weak, strong, weak, run_cascades);
}

// CascadeBoost::predict (src/adaboost.cpp:507-542) on caller-supplied feature vectors.
__global__ __launch_bounds__(CLS_THREADS) void k_cascade_fv(const double *__restrict__ fv, int n, double *out, CascadeDev c)
{
    __shared__ double s_fv[1024];
    __shared__ double s_vals[CLS_CHUNK];
    __shared__ double s_acc;
    const int tid = threadIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        for (int k = tid; k < 1024; k += CLS_THREADS) s_fv[k] = fv[(size_t)i * 1024 + k];
        __syncthreads();
        int    off = 0;
        double score = 0;
        bool   rejected = false;
        for (int s = 0; s < c.n_stages && !rejected; ++s) {
            const int nst = c.stage_n[s];
            if (tid == 0) s_acc = 0;
            for (int base = 0; base < nst; base += CLS_CHUNK) {
                const int m = min(CLS_CHUNK, nst - base);
                for (int j = tid; j < m; j += CLS_THREADS) {
                    const int st = off + base + j;
                    double    v = 0;
                    if (st < c.n_stumps) {
                        const double f = s_fv[c.dim[st]], d = c.dir[st];
                        v = (f * d < c.thr[st] * d) ? c.vp[st] : c.vn[st];
                    }
                    s_vals[j] = v;
                }
                __syncthreads();
                if (tid == 0) {
                    double a = s_acc;
                    const int mm = min(m, max(0, c.n_stumps - off - base));
                    for (int j = 0; j < mm; ++j) a += s_vals[j];
                    s_acc = a;
                }
                __syncthreads();
            }
            score = s_acc;
            __syncthreads();
            if (score < (double)c.stage_thresh[s]) rejected = true;
            off += nst;
        }
        if (tid == 0) out[i] = rejected ? -DBL_MAX : score;
        __syncthreads();
    }
}

void launch_cascade_fv(hipStream_t s, const double *fv, int n, double *out, CascadeDev c)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_cascade_fv, dim3(n < 2048 ? n : 2048), dim3(CLS_THREADS), 0, s, fv, n, out, c);
}
