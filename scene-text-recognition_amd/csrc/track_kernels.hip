// track_kernels.hip -- the first consumer of the classified candidates:
//
//   k_calc_color   calc_color            src/ER.cpp:1391-1419
//   k_er_track     ERFilter::er_track    src/ER.cpp:530-590
//
// calc_color is the only per-pixel work left after classify: Otsu over the ER's box on its own
// channel, then the mean of the three YCrCb bytes under the mask.  One workgroup per ER; sums are
// integers, so the result is the reference's f64 quotient exactly.  er_track is a closure: all_er
// starts as the strong ERs and grows by every weak ER that the rule at :575-587 ties to something
// already in it -- the SET does not depend on the visiting order, only all_er's order does (and
// er_grouping sorts it anyway), so the kernel runs it as a breadth-first frontier per image.
#include "track_kernels.h"

#include "er_device.h"
#include "ocr_device.h"

namespace str_er {

__device__ __forceinline__ uint32_t wsum32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// calc_color (src/ER.cpp:1391-1419) for n boxes whose Otsu thresholds are known (launch_box_thresholds): the mean of the three YCrCb bytes under
// the mask (255 - roi > threshold).  A wave per box; sums are integers, so the result is the reference's f64 quotient exactly.  The colour image
// is read from ITS row i / column j, not the box's (:1404-1405, kept).  Boxes above OCR_BIG_PX pixels are queued and summed by many workgroups
// (k_color_sums_big): a launch lasts as long as its longest wave, and the largest boxes of a batch have 50 times the pixels of the average one.
struct ColorJob {
    OcrSrc   src;        // the boxes (mask plane = the box's own channel)
    ColorSrc col;        // explicit boxes: the colour image; records: taken from the plane descriptor (Y, Cr, Cb planes of the box's level)
};

__device__ __forceinline__ ColorSrc color_of(const ColorJob &j, int bi)
{
    if (!j.src.recs) return j.col;
    const PlaneDesc &pd = j.src.planes[j.src.recs[j.src.list[bi]].plane];
    ColorSrc c;
    c.c0 = pd.pix - (size_t)(pd.ch % 3) * pd.color_pitch;          // Y, Cr, Cb planes of this pyramid level
    c.c1 = c.c0 + pd.color_pitch; c.c2 = c.c1 + pd.color_pitch;
    c.step = 1; c.stride = pd.stride;
    return c;
}

__device__ __forceinline__ void color_store(TrackRec *tr, unsigned long long c, unsigned long long t0, unsigned long long t1, unsigned long long t2)
{
    TrackRec r{};
    r.color1 = (double)t0 / (double)c;      // count == 0: 0.0 / 0 as in the reference
    r.color2 = (double)t1 / (double)c;
    r.color3 = (double)t2 / (double)c;
    *tr = r;
}

// rows [y_lo, y_hi) step y_step of box b: count and byte sums of the masked pixels, four pixels of a lane requested together -- the mask byte and the
// three colour bytes of each, at addresses clamped into the box, unconditionally (written as `if (in the box && masked) load` every load is a block
// of its own with a full wait behind it: sixteen memory round trips per pass instead of one)
__device__ __forceinline__ void color_rows(const OcrBox &b, const ColorSrc &col, int th, int y_lo, int y_hi, int y_step, int lane, uint32_t &cnt, uint32_t &a0,
                                           uint32_t &a1, uint32_t &a2)
{
    const GlobalBytes c0 = (GlobalBytes)col.c0, c1 = (GlobalBytes)col.c1, c2 = (GlobalBytes)col.c2;
    auto take = [&](const bool (&in)[4], const int (&xs)[4], const int (&ys)[4]) {
        uint32_t mk[4], v0[4], v1[4], v2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = (size_t)ys[u] * col.stride + (size_t)xs[u] * col.step;
            mk[u] = b.roi[(size_t)ys[u] * b.stride + xs[u]];
            v0[u] = c0[o]; v1[u] = c1[o]; v2[u] = c2[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool m = in[u] && (255 - (int)(mk[u] ^ (uint32_t)b.inv)) > th;
            cnt += m ? 1u : 0u; a0 += m ? v0[u] : 0u; a1 += m ? v1[u] : 0u; a2 += m ? v2[u] : 0u;
        }
    };
    if (b.bw <= 64) {
        const int rpp = 64 / b.bw, ry = lane / b.bw, x = min(lane - ry * b.bw, b.bw - 1);
        for (int y0 = y_lo; y0 < y_hi; y0 += 4 * rpp * y_step) {
            bool in[4];
            int  xs[4], ys[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int y = y0 + (u * rpp + ry) * y_step; in[u] = ry < rpp && y < y_hi; xs[u] = x; ys[u] = min(y, b.bh - 1); }
            take(in, xs, ys);
        }
    } else {
        for (int y = y_lo; y < y_hi; y += y_step)
            for (int x0 = 0; x0 < b.bw; x0 += 256) {
                bool in[4];
                int  xs[4], ys[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int x = x0 + 64 * u + lane; in[u] = x < b.bw; xs[u] = min(x, b.bw - 1); ys[u] = y; }
                take(in, xs, ys);
            }
    }
}

// tr index of box bi: the candidate it belongs to (records) or bi itself (explicit boxes)
__device__ __forceinline__ uint32_t color_slot(const ColorJob &j, int bi) { return j.src.recs ? j.src.list[bi] : (uint32_t)bi; }

__global__ __launch_bounds__(64 * OCR_WAVES) void k_color_sums(ColorJob job, int n, const int32_t *__restrict__ thresh, TrackRec *__restrict__ tr,
                                                              unsigned long long *__restrict__ sums, uint32_t *__restrict__ big)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int bi = blockIdx.x * OCR_WAVES + w; bi < n; bi += gridDim.x * OCR_WAVES) {
        const OcrBox b = ocr_box(job.src, bi);
        if (b.bw * b.bh > OCR_BIG_PX) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(big, 1u);
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            if (at < (uint32_t)OCR_BIG_CAP) {
                if (lane == 0) big[1 + at] = (uint32_t)bi;
                if (lane < 4) sums[4 * (size_t)bi + lane] = 0;
                continue;
            }
        }
        const ColorSrc col = color_of(job, bi);
        uint32_t cnt = 0, a0 = 0, a1 = 0, a2 = 0;           // (a lane's sums stay below 2^32: <= 255 x pixels / 64... a box has < 2^24 pixels)
        color_rows(b, col, thresh[bi], 0, b.bh, 1, lane, cnt, a0, a1, a2);
        unsigned long long c = cnt, t0 = a0, t1 = a1, t2 = a2;
        for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o); t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
        if (lane == 0) color_store(tr + color_slot(job, bi), c, t0, t1, t2);
    }
}

__global__ __launch_bounds__(64 * OCR_WAVES) void k_color_sums_big(ColorJob job, const int32_t *__restrict__ thresh, unsigned long long *__restrict__ sums,
                                                                  const uint32_t *__restrict__ big)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, part = blockIdx.x;
    const int nq = (int)min(big[0], (uint32_t)OCR_BIG_CAP);
    for (int qi = blockIdx.y; qi < nq; qi += gridDim.y) {
        const int    bi = (int)big[1 + qi];
        const OcrBox b = ocr_box(job.src, bi);
        const int    y_lo = (int)((long long)part * b.bh / OCR_BIG_PARTS), y_hi = (int)((long long)(part + 1) * b.bh / OCR_BIG_PARTS);
        if (y_lo + w >= y_hi) continue;
        const ColorSrc col = color_of(job, bi);
        uint32_t cnt = 0, a0 = 0, a1 = 0, a2 = 0;
        color_rows(b, col, thresh[bi], y_lo + w, y_hi, OCR_WAVES, lane, cnt, a0, a1, a2);
        unsigned long long c = cnt, t0 = a0, t1 = a1, t2 = a2;
        for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o); t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
        if (lane == 0) {
            atomicAdd(&sums[4 * (size_t)bi], c); atomicAdd(&sums[4 * (size_t)bi + 1], t0);
            atomicAdd(&sums[4 * (size_t)bi + 2], t1); atomicAdd(&sums[4 * (size_t)bi + 3], t2);
        }
    }
}

__global__ __launch_bounds__(256) void k_color_final_big(ColorJob job, const unsigned long long *__restrict__ sums, const uint32_t *__restrict__ big,
                                                         TrackRec *__restrict__ tr)
{
    const int nq = (int)min(big[0], (uint32_t)OCR_BIG_CAP);
    for (int qi = blockIdx.x * blockDim.x + threadIdx.x; qi < nq; qi += gridDim.x * blockDim.x) {
        const int bi = (int)big[1 + qi];
        color_store(tr + color_slot(job, bi), sums[4 * (size_t)bi], sums[4 * (size_t)bi + 1], sums[4 * (size_t)bi + 2], sums[4 * (size_t)bi + 3]);
    }
}

size_t calc_color_scratch_bytes(size_t n)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(n * 256 * 4) + up(4 * (size_t)(OCR_BIG_CAP + 1)) + up(n * 4) + up(n * 32) + up(4 * (size_t)(OCR_BIG_CAP + 1));
}

void launch_calc_color(hipStream_t s, const OcrSrc &src, const ColorSrc &col, int n, TrackRec *tr, uint8_t *scratch)
{
    if (n <= 0) return;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    uint32_t *hist = reinterpret_cast<uint32_t *>(scratch);
    uint32_t *big = reinterpret_cast<uint32_t *>(scratch + up((size_t)n * 256 * 4));
    int32_t  *thresh = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(big) + up(4 * (size_t)(OCR_BIG_CAP + 1)));
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(thresh) + up((size_t)n * 4));
    uint32_t *big2 = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(sums) + up((size_t)n * 32));
    launch_box_thresholds(s, src, n, hist, big, thresh);
    ColorJob job{src, col};
    (void)hipMemsetAsync(big2, 0, 4, s);
    const int n_cu = ocr_n_cu();
    const int wg = (n + OCR_WAVES - 1) / OCR_WAVES;
    hipLaunchKernelGGL(k_color_sums, dim3(wg < 8 * n_cu ? wg : 8 * n_cu), dim3(64 * OCR_WAVES), 0, s, job, n, (const int32_t *)thresh, tr, sums, big2);
    hipLaunchKernelGGL(k_color_sums_big, dim3(OCR_BIG_PARTS, 64), dim3(64 * OCR_WAVES), 0, s, job, (const int32_t *)thresh, sums, (const uint32_t *)big2);
    hipLaunchKernelGGL(k_color_final_big, dim3(16), dim3(256), 0, s, job, (const unsigned long long *)sums, (const uint32_t *)big2, tr);
}

__global__ void k_group_ranges(BatchDev b, int ppg, int n_groups, uint32_t *__restrict__ ranges)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int p0 = g * ppg, p1 = p0 + ppg - 1;
    ranges[2 * g] = b.ctr[p0].cand_base;
    ranges[2 * g + 1] = b.ctr[p1].cand_base + b.ctr[p1].n_pool;
}

void launch_group_ranges(hipStream_t s, const BatchDev &b, int planes_per_group, int n_groups, uint32_t *ranges)
{
    if (n_groups <= 0) return;
    hipLaunchKernelGGL(k_group_ranges, dim3((n_groups + 63) / 64), dim3(64), 0, s, b, planes_per_group, n_groups, ranges);
}

// the rule of src/ER.cpp:575-587 (USE_STROKE_WIDTH is not defined in the reference build)
__device__ __forceinline__ bool track_rule(const CandRec &s, const TrackRec &ts, const CandRec &w, const TrackRec &tw)
{
    const int sw = s.w, sh = s.h, ww = w.w, wh = w.h, sa = (int)s.area, wa = (int)w.area;
    return abs(ts.cx - tw.cx) + abs(ts.cy - tw.cy) < (max(sw, sh) << 1) &&
           abs(sh - wh) < min(sh, wh) &&
           abs(sw - ww) < ((sw + ww) >> 1) &&
           fabs(ts.color1 - tw.color1) < 25 &&
           fabs(ts.color2 - tw.color2) < 25 &&
           fabs(ts.color3 - tw.color3) < 25 &&
           abs(sa - wa) < min(sa, wa) * 3;
}

__global__ __launch_bounds__(256) void k_er_track(const CandRec *__restrict__ cands, TrackRec *tr, uint32_t *list,
                                                  const uint32_t *__restrict__ ranges)
{
    __shared__ uint32_t s_n;
    const uint32_t lo = ranges[2 * blockIdx.x], hi = ranges[2 * blockIdx.x + 1];
    const int      tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    __syncthreads();
    // it->center = Point(bound.x + bound.width / 2, bound.y + bound.height / 2) (:542, :552); strong ERs open all_er (:558-561)
    for (uint32_t i = lo + tid; i < hi; i += 256) {
        const CandRec &c = cands[i];
        tr[i].cx = (int)c.x + (int)c.w / 2;
        tr[i].cy = (int)c.y + (int)c.h / 2;
        const bool strong = c.cls == 1;
        tr[i].tracked = strong ? 1u : 0u;
        if (strong) list[lo + atomicAdd(&s_n, 1u)] = i;
    }
    __syncthreads();
    uint32_t qlo = 0, qhi = s_n;
    while (qhi > qlo) {
        for (uint32_t i = lo + tid; i < hi; i += 256) {
            const CandRec &w = cands[i];
            if (w.cls != 2 || tr[i].tracked) continue;
            const TrackRec tw = tr[i];
            for (uint32_t q = qlo; q < qhi; ++q) {
                const uint32_t si = list[lo + q];
                if (track_rule(cands[si], tr[si], w, tw)) {
                    tr[i].tracked = 1u;
                    list[lo + atomicAdd(&s_n, 1u)] = i;
                    break;
                }
            }
        }
        __syncthreads();
        qlo = qhi; qhi = s_n;
        __syncthreads();
    }
}

void launch_er_track(hipStream_t s, const CandRec *cands, TrackRec *tr, uint32_t *list, const uint32_t *ranges, int n_groups)
{
    if (n_groups <= 0) return;
    hipLaunchKernelGGL(k_er_track, dim3(n_groups), dim3(256), 0, s, cands, tr, list, ranges);
}

// ------------------------------------------------------------------------------------
// er_grouping (src/ER.cpp:612-692), the parts that are independent per pair of ERs: the sort by
// center.x (as a rank: position = number of ERs that come before), inner_suppression's flags
// (:893-922) and the pairwise rule (:631-644) as a list of pairs in the reference's visiting order.
// The greedy line assignment that consumes the pairs, and the per-line suppression / slope, are
// sequential by definition and run on the host (er_group.cpp).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan32(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(v, o);
        if (lane >= o) v += n;
    }
    return v;
}

// exclusive block scan of one value per lane (256 lanes); returns offset, *total = block sum
__device__ __forceinline__ uint32_t block_scan256(uint32_t v, uint32_t *s_w, uint32_t *total)
{
    const int      tid = threadIdx.x;
    const uint32_t incl = wave_incl_scan32(v);
    __syncthreads();
    if ((tid & 63) == 63) s_w[tid >> 6] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (i < (tid >> 6)) off += s_w[i]; tot += s_w[i]; }
    *total = tot;
    return off + incl - v;
}

struct GEr { int x, y, w, h, cx, cy, area; double c1, c2, c3; };
__device__ __forceinline__ GEr load_ger(const CandRec *cands, const TrackRec *tr, uint32_t ci)
{
    const CandRec &c = cands[ci];
    const TrackRec &t = tr[ci];
    GEr e;
    e.x = c.x; e.y = c.y; e.w = c.w; e.h = c.h; e.area = (int)c.area; e.cx = t.cx; e.cy = t.cy;
    e.c1 = t.color1; e.c2 = t.color2; e.c3 = t.color3;
    return e;
}

__global__ __launch_bounds__(256) void k_group_prepare(const CandRec *__restrict__ cands, const TrackRec *__restrict__ tr,
                                                       const uint32_t *__restrict__ ranges, int inner_sup, GroupBufs g)
{
    __shared__ uint32_t s_w[4];
    const uint32_t lo = ranges[2 * blockIdx.x], hi = ranges[2 * blockIdx.x + 1];
    const int      tid = threadIdx.x;
    uint32_t *A = g.tmp_a + lo, *B = g.tmp_b + lo, *S = g.sorted + lo;
    // 1. all_er in candidate order
    uint32_t n_t = 0;
    for (uint32_t i0 = lo; i0 < hi; i0 += 256) {
        const uint32_t i = i0 + tid;
        const uint32_t f = (i < hi && tr[i].tracked) ? 1u : 0u;
        uint32_t       tot;
        const uint32_t off = block_scan256(f, s_w, &tot);
        if (f) A[n_t + off] = i;
        n_t += tot;
    }
    __syncthreads();
    // 2. sort(all_er, center.x) (:614), stable: position = how many come before.  (inner_sup bit 1: the list comes sorted and
    //    overlap-suppressed from the host and keeps its order -- the reference does not sort again after overlap_suppression)
    if (inner_sup & 2) for (uint32_t k = tid; k < n_t; k += 256) B[k] = A[k];
    else
    for (uint32_t k = tid; k < n_t; k += 256) {
        const int ck = tr[A[k]].cx;
        uint32_t  rank = 0;
        for (uint32_t j = 0; j < n_t; ++j) {
            const int cj = tr[A[j]].cx;
            rank += (cj < ck || (cj == ck && j < k)) ? 1u : 0u;
        }
        B[rank] = A[k];
    }
    __syncthreads();
    // 3. inner_suppression (:893-922): j goes if some i contains it, shares its centre (within 0.2 of i's larger side) and
    //    has more than twice its box area
    uint32_t m = 0;
    for (uint32_t j0 = 0; j0 < n_t; j0 += 256) {
        const uint32_t j = j0 + tid;
        uint32_t       keep = 0;
        if (j < n_t) {
            keep = 1;
            if (inner_sup & 1) {
                const GEr b = load_ger(cands, tr, B[j]);
                for (uint32_t i = 0; i < n_t && keep; ++i) {
                    const GEr    a = load_ger(cands, tr, B[i]);
                    const double dx = a.cx - b.cx, dy = a.cy - b.cy;
                    if (sqrt(dx * dx + dy * dy) < 0.2 * max(a.w, a.h) && a.x <= b.x && a.y <= b.y && a.x + a.w >= b.x + b.w &&
                        a.y + a.h >= b.y + b.h && (double)(a.w * a.h) / (double)(b.w * b.h) > 2.0)
                        keep = 0;
                }
            }
        }
        uint32_t       tot;
        const uint32_t off = block_scan256(keep, s_w, &tot);
        if (keep) S[m + off] = B[j];
        m += tot;
    }
    if (tid == 0) g.n_sorted[blockIdx.x] = m;
}

// the rule of src/ER.cpp:631-644, a before b in the sorted list
__device__ __forceinline__ bool group_rule(const GEr &a, const GEr &b)
{
    return abs(a.cx - b.cx) < max(a.w, b.w) * 3.0 &&
           abs(a.cy - b.cy) < (a.h + b.h) * 0.25 &&
           abs(a.h - b.h) < min(a.h, b.h) &&
           abs(a.w - b.w) < min(a.h, b.h * 2) &&
           fabs(a.c1 - b.c1) < 25 && fabs(a.c2 - b.c2) < 25 && fabs(a.c3 - b.c3) < 25 &&
           abs(a.area - b.area) < min(a.area, b.area) * 4;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_group_pairs(const CandRec *__restrict__ cands, const TrackRec *__restrict__ tr,
                                                     const uint32_t *__restrict__ ranges, GroupBufs g, int n_groups)
{
    __shared__ uint32_t s_w[4];
    const uint32_t  lo = ranges[2 * blockIdx.x];
    const uint32_t  m = g.n_sorted[blockIdx.x];
    const uint32_t *S = g.sorted + lo;
    uint32_t       *rc = g.row_cnt + lo;
    const int       tid = threadIdx.x;
    if (FILL) {
        if (g.pair_off[n_groups] > g.pair_cap) return;            // the host grows the buffer and asks again
        // row offsets: exclusive prefix of the row counts, from the image's base
        uint32_t carry = g.pair_off[blockIdx.x];
        for (uint32_t i0 = 0; i0 < m; i0 += 256) {
            const uint32_t i = i0 + tid;
            const uint32_t v = i < m ? rc[i] : 0u;
            uint32_t       tot;
            const uint32_t off = block_scan256(v, s_w, &tot);
            __syncthreads();
            if (i < m) rc[i] = carry + off;
            carry += tot;
        }
        __syncthreads();
    }
    uint32_t mine = 0;
    for (uint32_t i = tid; i < m; i += 256) {
        const GEr a = load_ger(cands, tr, S[i]);
        uint32_t  n = 0, at = FILL ? rc[i] : 0u;
        for (uint32_t j = i + 1; j < m; ++j)
            if (group_rule(a, load_ger(cands, tr, S[j]))) {
                if (FILL) g.pairs[at++] = (i << 16) | j;
                ++n;
            }
        if (!FILL) rc[i] = n;
        mine += n;
    }
    if (!FILL) {
        uint32_t tot;
        (void)block_scan256(mine, s_w, &tot);
        if (tid == 0) g.pair_off[blockIdx.x] = tot;               // per-image count; k_pair_prefix turns it into offsets
    }
}

__global__ __launch_bounds__(1024) void k_pair_prefix(uint32_t *pair_off, int n_groups)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_groups; base += 1024) {
        const int      i = base + tid;
        const uint32_t v = i < n_groups ? pair_off[i] : 0u;
        const uint32_t incl = wave_incl_scan32(v);
        if ((tid & 63) == 63) s_w[tid >> 6] = incl;
        __syncthreads();
        uint32_t off = s_carry, tot = 0;
        for (int k = 0; k < 16; ++k) { if (k < (tid >> 6)) off += s_w[k]; tot += s_w[k]; }
        if (i < n_groups) pair_off[i] = off + incl - v;
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    if (tid == 0) pair_off[n_groups] = s_carry;
}

void launch_group_prepare(hipStream_t s, const CandRec *cands, const TrackRec *tr, const uint32_t *ranges, int n_groups, int inner_sup,
                          const GroupBufs &g)
{
    if (n_groups <= 0) return;
    hipLaunchKernelGGL(k_group_prepare, dim3(n_groups), dim3(256), 0, s, cands, tr, ranges, inner_sup, g);
}

void launch_group_pairs_count(hipStream_t s, const CandRec *cands, const TrackRec *tr, const uint32_t *ranges, int n_groups, const GroupBufs &g)
{
    if (n_groups <= 0) return;
    hipLaunchKernelGGL(k_group_pairs<false>, dim3(n_groups), dim3(256), 0, s, cands, tr, ranges, g, n_groups);
    hipLaunchKernelGGL(k_pair_prefix, dim3(1), dim3(1024), 0, s, g.pair_off, n_groups);
}

void launch_group_pairs_fill(hipStream_t s, const CandRec *cands, const TrackRec *tr, const uint32_t *ranges, int n_groups, const GroupBufs &g)
{
    if (n_groups <= 0) return;
    hipLaunchKernelGGL(k_group_pairs<true>, dim3(n_groups), dim3(256), 0, s, cands, tr, ranges, g, n_groups);
}

} // namespace str_er
