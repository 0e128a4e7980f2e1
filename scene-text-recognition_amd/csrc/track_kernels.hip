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

namespace str_er {

__device__ __forceinline__ uint32_t wsum32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void k_calc_color(const uint8_t *__restrict__ mask, int mstride, int inv, ColorSrc col,
                                                    const int32_t *__restrict__ boxes, int n_boxes, const CandRec *__restrict__ cands,
                                                    const uint32_t *__restrict__ total, const PlaneDesc *__restrict__ planes,
                                                    TrackRec *__restrict__ tr)
{
    __shared__ uint32_t s_hist[256];
    __shared__ int      s_th;
    __shared__ uint32_t s_part[4][4];
    const int tid = threadIdx.x;
    const int n = cands ? (int)*total : n_boxes;
    for (int bi = blockIdx.x; bi < n; bi += gridDim.x) {
        int bx, by, bw, bh;
        if (cands) {
            const CandRec &cd = cands[bi];
            if (cd.cls == 0) {
                if (tid == 0) { TrackRec z{}; tr[bi] = z; }
                continue;
            }
            const PlaneDesc &pd = planes[cd.plane];
            bx = cd.x; by = cd.y; bw = cd.w; bh = cd.h;
            mask = pd.pix; mstride = pd.stride; inv = pd.invert;
            col.c0 = pd.pix - (size_t)(pd.ch % 3) * pd.color_pitch;          // Y, Cr, Cb planes of this pyramid level
            col.c1 = col.c0 + pd.color_pitch; col.c2 = col.c1 + pd.color_pitch;
            col.step = 1; col.stride = pd.stride;
        } else {
            bx = boxes[4 * bi]; by = boxes[4 * bi + 1]; bw = boxes[4 * bi + 2]; bh = boxes[4 * bi + 3];
        }
        const uint8_t *roi = mask + (size_t)by * mstride + bx;
        s_hist[tid] = 0;
        __syncthreads();
        const int npx = bw * bh;
        for (int i = tid; i < npx; i += 256) {
            const int y = i / bw, x = i - y * bw;
            atomicAdd(&s_hist[255 - (roi[(size_t)y * mstride + x] ^ inv)], 1u);
        }
        __syncthreads();
        if (tid == 0) s_th = otsu_from_hist(s_hist, (double)bw * bh);          // threshold(255-img, ..., THRESH_OTSU), :1395
        __syncthreads();
        const int th = s_th;
        // masked sums; the colour image is read from ITS row i / column j, not the box's (:1404-1405, kept)
        uint32_t cnt = 0, a0 = 0, a1 = 0, a2 = 0;
        for (int i = tid; i < npx; i += 256) {
            const int y = i / bw, x = i - y * bw;
            if ((255 - (roi[(size_t)y * mstride + x] ^ inv)) > th) {
                const size_t o = (size_t)y * col.stride + (size_t)x * col.step;
                ++cnt; a0 += col.c0[o]; a1 += col.c1[o]; a2 += col.c2[o];
            }
        }
        cnt = wsum32(cnt); a0 = wsum32(a0); a1 = wsum32(a1); a2 = wsum32(a2);   // per-wave sums stay below 2^32 (<= 255 * pixels / 4)
        if ((tid & 63) == 0) { s_part[tid >> 6][0] = cnt; s_part[tid >> 6][1] = a0; s_part[tid >> 6][2] = a1; s_part[tid >> 6][3] = a2; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long c = 0, t0 = 0, t1 = 0, t2 = 0;
            for (int w = 0; w < 4; ++w) { c += s_part[w][0]; t0 += s_part[w][1]; t1 += s_part[w][2]; t2 += s_part[w][3]; }
            TrackRec r{};
            r.color1 = (double)t0 / (double)c;      // count == 0: 0.0 / 0 as in the reference
            r.color2 = (double)t1 / (double)c;
            r.color3 = (double)t2 / (double)c;
            tr[bi] = r;
        }
        __syncthreads();
    }
}

void launch_calc_color_batch(hipStream_t s, const BatchDev &b, TrackRec *tr)
{
    ColorSrc none{};
    hipLaunchKernelGGL(k_calc_color, dim3(2048), dim3(256), 0, s, (const uint8_t *)nullptr, 0, 0, none, (const int32_t *)nullptr, 0,
                       (const CandRec *)b.cands, (const uint32_t *)b.total_cands, (const PlaneDesc *)b.planes, tr);
}

void launch_calc_color_boxes(hipStream_t s, const uint8_t *mask, int mstride, ColorSrc col, const int32_t *boxes, int n, TrackRec *tr)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_calc_color, dim3(n < 2048 ? n : 2048), dim3(256), 0, s, mask, mstride, 0, col, boxes, n, (const CandRec *)nullptr,
                       (const uint32_t *)nullptr, (const PlaneDesc *)nullptr, tr);
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
