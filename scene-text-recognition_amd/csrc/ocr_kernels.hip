// ocr_kernels.hip -- the OCR scorer of config 3 for N boxes at once:
// OCR::chain_run (src/OCR.cpp:67-140) = chain-code features (extract_feature, :144-218) + svm_predict_probability
// (src/svm.cpp:2592-2629).  Launch chain of one call:
//
//   k_ocr_list      indices of the strong / weak candidates, in candidate order
//   k_ocr_hist      one wave per box: histogram of 255 - roi (boxes above 4096 px: k_ocr_hist_big)       (HBM: the ROI bytes, once)
//   k_ocr_otsu      one LANE per box: getThreshVal_Otsu_8u's sequential f64 scan -- 64 boxes per wave instead of one
//   k_ocr_features  one wave per box: ARAN(30) of the binarised (optionally rotated) ROI, direction bitmaps, 7x7 Gaussian,
//                   min-max normalisation, 2x2 decimation -> q[1800]; written as the svm's row of bf16 numerators + |x|^2
//   k_svm_kernel_q  K[n][i] = exp(-gamma |x_n - sv_i|^2) for rows of numerators: x.sv = (q.sv) / 255 as three
//                   v_mfma_f32_32x32x16_bf16 per 16 features (q exact in bf16, sv = three bf16 pieces exactly), exp in f64
//   k_svm_kernel    the same for f32 rows (vectors handed in as doubles): v_mfma_f32_32x32x2_f32
//   k_svm_couple    one wave per box: the k(k-1)/2 decision values (svm_predict_values, src/svm.cpp:2539-2566) in row passes
//                   of coalesced coefficient rows, Platt sigmoid (:1818-1826), Wu-Lin-Weng coupling (multiclass_probability,
//                   :1829-1890) with Q's rows in registers, arg max -- all in f64, pairwise values never leave LDS
//
// The OpenCV primitives involved (Otsu, findContours, GaussianBlur, normalize, resize) are restated from OpenCV 4.x and
// checked against the CPU restatement the tests use ("parity unpinned").
#include <hip/hip_runtime.h>

#include <type_traits>

#include <float.h>
#include <stdint.h>

#include "er_device.h"
#include "ocr_kernels.h"
#include "ocr_device.h"

namespace str_er {

typedef float float16v __attribute__((ext_vector_type(16)));

#ifndef SVM_Q_UNROLL
#define SVM_Q_UNROLL 1        // (66+ classes) support vectors per pass of the pair-indexed decision-value loops
#endif

// ---------------------------------------------------------------------------------------------------------
// boxes
// ---------------------------------------------------------------------------------------------------------
// indices of the strong / weak candidates of the batch, in candidate order: two launches of up to 256 workgroups, each over a contiguous range of the
// candidates -- count, then (base = the counts before mine) fill.  The class byte sits in a 48-byte record: one workgroup alone pulls 48 k cache lines
// through one compute unit's address pipeline for a batch (46 us); a thread asks for sixteen records' bytes at once.
constexpr int LIST_CPT = 16, LIST_THREADS = 256, LIST_CHUNK = LIST_CPT * LIST_THREADS;
template <bool FILL>
__global__ __launch_bounds__(LIST_THREADS) void k_ocr_list(const CandRec *__restrict__ cands, const uint32_t *__restrict__ total_cands,
                                                            uint32_t *__restrict__ hdr, uint32_t *__restrict__ list)
{
    __shared__ uint32_t s_w[LIST_THREADS / 64];
    __shared__ uint32_t s_carry;
    const int      tid = threadIdx.x, lane = tid & 63, G = gridDim.x, g = blockIdx.x;
    const uint32_t total = *total_cands;
    // my range: whole chunks, the same split in both launches
    const uint32_t chunks = (total + LIST_CHUNK - 1) / LIST_CHUNK, per = (chunks + G - 1) / G;
    const uint32_t lo = min((uint32_t)g * per * LIST_CHUNK, total), hi = min(lo + per * LIST_CHUNK, total);
    uint32_t *counts = hdr + 16;
    if (FILL) {
        uint32_t v = tid < g ? counts[tid] : 0u;           // (G <= 256 = the workgroup)
        for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
        if (lane == 0) s_w[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) s_carry = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    } else if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = lo; base < hi; base += LIST_CHUNK) {
        const uint32_t i0 = base + (uint32_t)LIST_CPT * (uint32_t)tid;
        uint32_t       mk = 0;
        uint8_t        cl[LIST_CPT];
#pragma unroll
        for (int u = 0; u < LIST_CPT; ++u) cl[u] = cands[min(i0 + u, total - 1)].cls;          // (unconditional: behind `i0 + u < hi &&` every load waits for the one before)
#pragma unroll
        for (int u = 0; u < LIST_CPT; ++u) mk |= (i0 + u < hi && cl[u] != 0) ? 1u << u : 0u;
        const uint32_t cnt = (uint32_t)__popc(mk);
        uint32_t       incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += t; }
        if (lane == 63) s_w[tid >> 6] = incl;
        __syncthreads();
        uint32_t off = s_carry + incl - cnt, tot = 0;
        for (int k = 0; k < LIST_THREADS / 64; ++k) { if (k < (tid >> 6)) off += s_w[k]; tot += s_w[k]; }
        if (FILL) for (uint32_t r = mk; r != 0; r &= r - 1u) list[off++] = i0 + (uint32_t)__builtin_ctz(r);
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    if (tid == 0) {
        if (!FILL) counts[g] = s_carry;
        else if (g == G - 1) hdr[0] = s_carry;            // (the last range's end = the number listed)
    }
}

// The same for the candidates NAMED in a list (from[0 .. *from_n): the candidates of the planes whose pools an NMS tie pass re-made, a few hundred at most):
// one workgroup, order of the list.
__global__ __launch_bounds__(256) void k_ocr_list_from(const CandRec *__restrict__ cands, const uint32_t *__restrict__ from, const uint32_t *__restrict__ from_n,
                                                      uint32_t *__restrict__ hdr, uint32_t *__restrict__ list)
{
    __shared__ uint32_t s_w[4];
    const int      tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t n = *from_n;
    uint32_t       carry = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + (uint32_t)tid;
        const uint32_t ci = i < n ? from[i] : 0u;
        const bool     on = i < n && cands[ci].cls != 0;
        const uint64_t bal = __builtin_amdgcn_ballot_w64(on);
        if (lane == 0) s_w[wv] = (uint32_t)__builtin_popcountll(bal);
        __syncthreads();
        uint32_t off = carry, tot = 0;
        for (int k = 0; k < 4; ++k) { if (k < wv) off += s_w[k]; tot += s_w[k]; }
        if (on) list[off + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull))] = ci;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) hdr[0] = carry;
}

// ---------------------------------------------------------------------------------------------------------
// Otsu, part 1: the histogram of 255 - roi.  One wave per box; eight interleaved sub-histograms (a binarisable
// ROI has two dominant grey values: with one copy most lanes of a wave would queue on the same LDS word).
// ---------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64 * OCR_WAVES) void k_ocr_hist(OcrSrc src, int n, uint32_t *__restrict__ hist, uint32_t *__restrict__ big)
{
    __shared__ uint32_t s_h[OCR_WAVES][256 * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t *h = s_h[w];
    n = ocr_count(src, n);
    for (int bi = blockIdx.x * OCR_WAVES + w; bi < n; bi += gridDim.x * OCR_WAVES) {
        const OcrBox b = ocr_box(src, bi);
        if (b.bw * b.bh > OCR_BIG_PX) {
            // queued for k_ocr_hist_big, which adds into the (zeroed) row; a full queue: the box is counted here after all
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(big, 1u);
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            if (at < (uint32_t)OCR_BIG_CAP) {
                if (lane == 0) big[1 + at] = (uint32_t)bi;
                for (int v = lane; v < 256; v += 64) hist[(size_t)bi * 256 + v] = 0;
                continue;
            }
        }
        for (int i = lane; i < 256 * 8 / 4; i += 64) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0, 0, 0, 0);
        const int sub = lane & 7;
        // four loads in flight per lane (a load per pass would pay a memory round trip per row of the box).  The addresses are clamped into the box and
        // the loads unconditional: as `ok ? load : 0` each load gets a block of its own with a full wait behind it
        auto bin = [&](int y, int x) -> uint32_t { return ((255u - ((uint32_t)b.roi[(size_t)y * b.stride + x] ^ (uint32_t)b.inv)) << 3) | (uint32_t)sub; };
        if (b.bw <= 64) {
            // several rows per pass: lane -> (row in the pass, column)
            const int rpp = 64 / b.bw, ry = lane / b.bw, x = lane - ry * b.bw;
            const bool on = ry < rpp;
            for (int y0 = 0; y0 < b.bh; y0 += 4 * rpp) {
                uint32_t e[4];
                bool     ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int y = y0 + u * rpp + ry; ok[u] = on && y < b.bh; e[u] = bin(min(y, b.bh - 1), min(x, b.bw - 1)); }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (ok[u]) atomicAdd(&h[e[u]], 1u);
            }
        } else {
            for (int y0 = 0; y0 < b.bh; y0 += 4)
                for (int x = lane; x < b.bw; x += 64) {
                    uint32_t e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) e[u] = bin(min(y0 + u, b.bh - 1), x);
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (y0 + u < b.bh) atomicAdd(&h[e[u]], 1u);
                }
        }
        for (int v = lane; v < 256; v += 64) {
            const uint4 a = *reinterpret_cast<const uint4 *>(h + 8 * v), c = *reinterpret_cast<const uint4 *>(h + 8 * v + 4);
            hist[(size_t)bi * 256 + v] = a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w;
        }
    }
}

// The queued boxes: workgroup (part, slot) counts rows [part bh / 32, (part + 1) bh / 32) of queue entry slot, slot + gridDim.y, ...; its four
// waves take every fourth row of the range, and the non-empty bins are added to the box's row of `hist` with device atomics.
__global__ __launch_bounds__(64 * OCR_WAVES) void k_ocr_hist_big(OcrSrc src, uint32_t *__restrict__ hist, const uint32_t *__restrict__ big)
{
    __shared__ uint32_t s_h[OCR_WAVES][256 * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, part = blockIdx.x;
    uint32_t *h = s_h[w];
    const int nq = (int)min(big[0], (uint32_t)OCR_BIG_CAP), sub = lane & 7;
    for (int qi = blockIdx.y; qi < nq; qi += gridDim.y) {
        const int    bi = (int)big[1 + qi];
        const OcrBox b = ocr_box(src, bi);
        const int    y_lo = (int)((long long)part * b.bh / OCR_BIG_PARTS), y_hi = (int)((long long)(part + 1) * b.bh / OCR_BIG_PARTS);
        if (y_lo + w >= y_hi) continue;
        for (int i = lane; i < 256 * 8 / 4; i += 64) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0, 0, 0, 0);
        for (int y = y_lo + w; y < y_hi; y += OCR_WAVES)
            for (int x0 = 0; x0 < b.bw; x0 += 256) {
                uint32_t e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {                                  // (clamped, unconditional loads: see k_ocr_hist)
                    const int x = min(x0 + 64 * u + lane, b.bw - 1);
                    e[u] = ((255u - ((uint32_t)b.roi[(size_t)y * b.stride + x] ^ (uint32_t)b.inv)) << 3) | (uint32_t)sub;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (x0 + 64 * u + lane < b.bw) atomicAdd(&h[e[u]], 1u);
            }
        for (int v = lane; v < 256; v += 64) {
            const uint4 a = *reinterpret_cast<const uint4 *>(h + 8 * v), c = *reinterpret_cast<const uint4 *>(h + 8 * v + 4);
            const uint32_t t = a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w;
            if (t) atomicAdd(&hist[(size_t)bi * 256 + v], t);
        }
    }
}

// Otsu, part 2: the scan is a chain of 256 dependent f64 steps (two divisions each) whose rounding must be the
// reference's: it cannot be spread over lanes, so every lane runs the scan of its own box.  The histograms of the
// wave's 64 boxes are transposed through LDS (row stride 257 words: conflict-free for both access patterns).
__global__ __launch_bounds__(256) void k_ocr_otsu(OcrSrc src, int n, const uint32_t *__restrict__ hist, int32_t *__restrict__ thresh)
{
    __shared__ uint32_t s_h[64 * 257];
    const int tid = threadIdx.x, b0 = blockIdx.x * 64;
    n = ocr_count(src, n);
    const int nb = min(64, n - b0);
    if (nb <= 0) return;
    // staging by all four waves, 16 loads of a lane in flight (one load per pass made this loop -- 256 memory round trips -- three quarters of the kernel)
    const uint32_t *hp = hist + (size_t)b0 * 256;
    for (int i0 = tid; i0 < nb * 256; i0 += 256 * 16) {
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = i0 + 256 * u; v[u] = i < nb * 256 ? hp[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = i0 + 256 * u; if (i < nb * 256) s_h[(i >> 8) * 257 + (i & 255)] = v[u]; }
    }
    __syncthreads();
    if (tid < nb) {
        const OcrBox b = ocr_box(src, b0 + tid);
        thresh[b0 + tid] = otsu_from_hist(s_h + tid * 257, (double)b.bw * b.bh);
    }
}

// ---------------------------------------------------------------------------------------------------------
// features
// ---------------------------------------------------------------------------------------------------------
// Source of ARAN(30): the Otsu-binarised ROI, tap = (255 - (p ^ inv)) > thresh ? 255 : 0 ...
struct BinSrc {
    GlobalBytes roi; int stride, inv, th;
    __device__ __forceinline__ int operator()(int x, int y) const { return (255 - (roi[(size_t)y * stride + x] ^ inv)) > th ? 255 : 0; }
};
// ... or that image seen through OCR::rotate_mat (src/OCR.cpp:282-352): canvas pixel (x, y) is rebuilt
// from its four binarised source taps with the reference's own f64 expression, in its order.
struct RotSrc {
    BinSrc b; int bw, bh; RotGeom r;
    __device__ __forceinline__ int operator()(int x, int y) const
    {
        const int i = y + r.min_y + r.ch, j = x + r.min_x;
        if (i >= r.max_y - r.ch || j >= r.max_x) return 0;                   // the loops are exclusive
        const double new_j = r.c * (double)j - r.s * (double)(i - r.ch) + (double)r.x0;
        const double new_i = r.s * (double)j + r.c * (double)(i - r.ch) + (double)r.y0;
        if (!(new_i > 0 && new_j > 0 && new_i < (double)(bh - 1) && new_j < (double)(bw - 1))) return 0;
        if (r.crop && !(i > r.min_y + r.ch && i < r.max_y - r.ch)) return 0;
        const int    sy = (int)new_i, sx = (int)new_j;
        const double fi = floor(new_i), fj = floor(new_j);
        if (new_i == fi && new_j == fj) return b(sx, sy);
        const double alpha = new_i - fi, beta = new_j - fj;
        const double A = (double)b(sx, sy), B = (double)b(sx + 1, sy), C = (double)b(sx, sy + 1), D = (double)b(sx + 1, sy + 1);
        const double v = (1 - alpha) * (1 - beta) * A + (1 - alpha) * beta * B + alpha * (1 - beta) * C + alpha * beta * D;
        return (int)(uint8_t)round(v);
    }
};

// cv::resize INTER_LINEAR 8UC1 over an arbitrary source (same arithmetic as resize_px, er_kernels.hip)
template <class Src>
__device__ __forceinline__ int resize_px_src(const ResizeGeom &g, const Src &src, int dx, int dy)
{
    if (g.mode == 0) return src(dx, dy);
    if (g.mode == 1) return (src(2 * dx, 2 * dy) + src(2 * dx + 1, 2 * dy) + src(2 * dx, 2 * dy + 1) + src(2 * dx + 1, 2 * dy + 1) + 2) >> 2;
    float fx = (float)((dx + 0.5) * g.scale_x - 0.5);
    int   sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= g.sw - 1) { fx = 0.f; sx = g.sw - 1; }
    const int a0 = __float2int_rn((1.f - fx) * 2048.f), a1 = __float2int_rn(fx * 2048.f);
    float fy = (float)((dy + 0.5) * g.scale_y - 0.5);
    int   sy = (int)floorf(fy);
    fy -= (float)sy;
    const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
    const int y0 = min(max(sy, 0), g.sh - 1), y1 = min(max(sy + 1, 0), g.sh - 1);
    const int sx1 = (sx + 1 < g.sw) ? sx + 1 : sx;
    const int r0 = src(sx, y0) * a0 + src(sx1, y0) * a1;
    const int r1 = src(sx, y1) * a0 + src(sx1, y1) * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    return min(max(v, 0), 255);
}

__device__ __forceinline__ int reflect101_30(int i) { return i < 0 ? -i : (i >= 30 ? 58 - i : i); }

// Direction bitmaps without following borders.  cv::findContours (RETR_LIST, CHAIN_APPROX_NONE) walks every outer and
// hole border of the 8-connected foreground once; extract_feature (src/OCR.cpp:144-197) then marks, for each contour
// point, the direction code of the step to the next point.  A step of the walk is a function of the state (pixel P,
// direction b of the previous border pixel): the next pixel is the first foreground neighbour counter-clockwise from
// b + 1.  That map is a bijection on the states whose neighbour b is foreground, so the states fall into disjoint
// cycles; the cycles the tracer follows are exactly those whose steps sweep over a 4-adjacent background pixel, and
// every step of such a cycle does (a step that sweeps over nothing, or over one diagonal background pixel only, lies
// on a 3- or 4-cycle around an inner corner that no border passes through).  Hence the set of (pixel, code) marks is a
// function of each pixel's 8-neighbour mask alone: bit c of chain_lut(nb) = "code c is marked at a foreground pixel
// whose neighbours are nb" (bit d of nb = neighbour d, d counter-clockwise from east).  Checked against the serial
// tracer of the CPU restatement on every mask and on random 30 x 30 images (tests/test_svm.py).
__device__ __forceinline__ uint32_t chain_lut(uint32_t nb)
{
    uint32_t out = 0;
    for (int b = 0; b < 8; ++b) {
        if (!((nb >> b) & 1u)) continue;
        bool swept4 = false;
        int  dn = b;
        for (int s = b + 1; s <= b + 8; ++s) {
            if ((nb >> (s & 7)) & 1u) { dn = s & 7; break; }
            if (!(s & 1)) swept4 = true;            // even directions are the 4-neighbours
        }
        // OCR::chain_code_direction (src/OCR.cpp:602-622) of the step east = 4, north-east = 3, ... : (4 - s) mod 8
        if (swept4) out |= 1u << ((4 - dn) & 7);
    }
    return out;
}

struct FeatWave {
    uint8_t  f[32 * 32];          // ARAN(30) image != 0, one pixel of border
    uint8_t  om[30 * 32];         // per pixel: the direction codes marked there (bit c)
    uint64_t rows[8 * 30];        // per code and row: the marks as bits 3 .. 32, with BORDER_REFLECT_101 columns at 0 .. 2 and 33 .. 35
    uint8_t  v[8 * 900];          // blurred maps
    uint32_t mn[8], mx[8];
    float    scale[8], shift[8];
};

template <bool ROT>
__device__ __forceinline__ void feat_aran(FeatWave &L, const OcrBox &b, int th, const RotGeom *rg, int lane)
{
    const BinSrc bsrc{b.roi, b.stride, b.inv, th};
    const int    sw = ROT ? rg->rw : b.bw, shh = ROT ? rg->rh : b.bh;
    const double R1 = (sw > shh) ? (double)shh / sw : (double)sw / shh;
    const int    k = (int)(30.0 * sqrt(R1));
    const int    dw = (sw > shh) ? 30 : k, dh = (sw > shh) ? k : 30;
    if (dw <= 0 || dh <= 0) return;
    const int offy = (dw > dh) ? (30 - dh) / 2 : 0, offx = (dw > dh) ? 0 : (30 - dw) / 2;
    const ResizeGeom g = resize_geom(sw, shh, dw, dh);
    const float inv_dw = 1.0f / (float)dw;
    for (int i = lane; i < dw * dh; i += 64) {
        const int dy = (int)(((float)i + 0.5f) * inv_dw), dx = i - dy * dw;      // exact for i < 900, dw <= 30
        int v;
        if (ROT) { const RotSrc rsrc{bsrc, b.bw, b.bh, *rg}; v = resize_px_src(g, rsrc, dx, dy); }
        else v = resize_px_src(g, bsrc, dx, dy);
        L.f[(dy + offy + 1) * 32 + dx + offx + 1] = v ? 1 : 0;
    }
}

__global__ __launch_bounds__(64 * OCR_WAVES) void k_ocr_features(OcrSrc src, int n, const int32_t *__restrict__ thresh, uint8_t *__restrict__ q_out,
                                                                uint16_t *__restrict__ xq, double *__restrict__ xnorm, int dq,
                                                                uint8_t *__restrict__ x8, int32_t *__restrict__ x8s, int dq8)
{
    __shared__ FeatWave s_w[OCR_WAVES];
    __shared__ uint8_t  s_lut[256];
    __shared__ uint16_t s_g7[128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    FeatWave &L = s_w[w];
    // tables: direction marks by neighbour mask; one row pass of the 7-tap Gaussian (8, 28, 56, 72, 56, 28, 8) over 7 mark bits (a mark is 255)
    s_lut[threadIdx.x] = (uint8_t)chain_lut(threadIdx.x);
    if (threadIdx.x < 128) {
        const int t = threadIdx.x;
        s_g7[t] = (uint16_t)(255 * (8 * ((t & 1) + ((t >> 6) & 1)) + 28 * (((t >> 1) & 1) + ((t >> 5) & 1)) + 56 * (((t >> 2) & 1) + ((t >> 4) & 1)) + 72 * ((t >> 3) & 1)));
    }
    __syncthreads();
    n = ocr_count(src, n);
    // (from here on the waves of the workgroup are independent: every wave owns its part of LDS; the wave is the synchronisation unit)
    for (int bi = blockIdx.x * OCR_WAVES + w; bi < n; bi += gridDim.x * OCR_WAVES) {
        const OcrBox b = ocr_box(src, bi);
        const int    th = thresh[bi];
        for (int i = lane; i < 32 * 32 / 4; i += 64) reinterpret_cast<uint32_t *>(L.f)[i] = 0;
        if (lane < 8) { L.mn[lane] = 255; L.mx[lane] = 0; }
        __builtin_amdgcn_wave_barrier();
        // ---- ARAN(30) of the binarised (and, for a slanted text line, rotated) ROI
        if (src.rot != nullptr && src.rot[bi].on != 0) feat_aran<true>(L, b, th, src.rot + bi, lane);
        else feat_aran<false>(L, b, th, nullptr, lane);
        __builtin_amdgcn_wave_barrier();
        // ---- direction marks of every pixel from its neighbour mask
        for (int i = lane; i < 30 * 32; i += 64) {
            const int y = i >> 5, x = i & 31;
            uint32_t  o = 0;
            if (x < 30) {
                const uint8_t *p = L.f + (y + 1) * 32 + x + 1;
                if (p[0]) {
                    const uint32_t nb = (uint32_t)p[1] | ((uint32_t)p[-31] << 1) | ((uint32_t)p[-32] << 2) | ((uint32_t)p[-33] << 3) | ((uint32_t)p[-1] << 4) |
                                        ((uint32_t)p[31] << 5) | ((uint32_t)p[32] << 6) | ((uint32_t)p[33] << 7);
                    o = s_lut[nb];
                }
            }
            L.om[i] = (uint8_t)o;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- per code and row: the 30 marks as a bit row, reflect-padded by 3 on both sides
        for (int i = lane; i < 8 * 30; i += 64) {
            const int c = i / 30, y = i - c * 30;
            const uint32_t *row = reinterpret_cast<const uint32_t *>(L.om + y * 32);
            uint32_t bits = 0;
#pragma unroll
            for (int d = 0; d < 8; ++d) bits |= ((((row[d] >> c) & 0x01010101u) * 0x00204081u >> 21) & 0xFu) << (4 * d);
            bits &= 0x3FFFFFFFu;
            uint64_t p = (uint64_t)bits << 3;
            p |= (uint64_t)((bits >> 3) & 1u) | (uint64_t)((bits >> 2) & 1u) << 1 | (uint64_t)((bits >> 1) & 1u) << 2;                 // x = -3, -2, -1 -> 3, 2, 1
            p |= (uint64_t)((bits >> 28) & 1u) << 33 | (uint64_t)((bits >> 27) & 1u) << 34 | (uint64_t)((bits >> 26) & 1u) << 35;       // x = 30, 31, 32 -> 28, 27, 26
            L.rows[i] = p;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- GaussianBlur 7x7 (BORDER_REFLECT_101), 8-bit fixed point: a lane owns a column of one map and slides down it
        for (int i = lane; i < 8 * 30; i += 64) {
            const int c = i / 30, x = i - c * 30;
            const uint64_t *rw = L.rows + c * 30;
            uint32_t h0, h1, h2, h3, h4, h5, h6;
            auto hrow = [&](int y) -> uint32_t { return s_g7[(uint32_t)(rw[reflect101_30(y)] >> x) & 127u]; };
            h0 = hrow(-3); h1 = hrow(-2); h2 = hrow(-1); h3 = hrow(0); h4 = hrow(1); h5 = hrow(2);
            uint32_t lo = 255, hi = 0;
            for (int y = 0; y < 30; ++y) {
                h6 = hrow(y + 3);
                const uint32_t s = 8 * (h0 + h6) + 28 * (h1 + h5) + 56 * (h2 + h4) + 72 * h3;
                const uint32_t v = min((s + (1u << 15)) >> 16, 255u);
                L.v[c * 900 + y * 30 + x] = (uint8_t)v;
                lo = min(lo, v); hi = max(hi, v);
                h0 = h1; h1 = h2; h2 = h3; h3 = h4; h4 = h5; h5 = h6;
            }
            atomicMin(&L.mn[c], lo);
            atomicMax(&L.mx[c], hi);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- normalize(0, 255, NORM_MINMAX) in float, then resize 30 -> 15 (exact 2x: area)
        if (lane < 8) {
            const int    mn = (int)L.mn[lane], mx = (int)L.mx[lane];
            const double scale = (mx - mn) > 0 ? 255.0 / (mx - mn) : 0.0, shift = 0.0 - mn * scale;
            L.scale[lane] = (float)scale; L.shift[lane] = (float)shift;
        }
        __builtin_amdgcn_wave_barrier();
        double nrm = 0;
        int    s1 = 0, s2 = 0;
        for (int i = lane; i < 1800; i += 64) {
            const int c = i / 225, r = i - c * 225, y = r / 15, x = r - y * 15;
            const uint8_t *m = L.v + c * 900 + (2 * y) * 30 + 2 * x;
            const float   sc = L.scale[c], sf = L.shift[c];
            auto nz = [&](uint8_t e) -> int { return min(max(__float2int_rn((float)e * sc + sf), 0), 255); };
            const int v = (nz(m[0]) + nz(m[1]) + nz(m[30]) + nz(m[31]) + 2) >> 2;
            if (q_out) q_out[(size_t)bi * 1800 + i] = (uint8_t)v;
            if (xq) {
                const double d = v / 255.0;                         // fv.value = ptr[p] / 255.0 (src/OCR.cpp:211)
                xq[(size_t)bi * dq + i] = (uint16_t)(__float_as_uint((float)v) >> 16);      // the numerator as bf16: 0 .. 255 are exact (k_svm_kernel_q)
                nrm += d * d;
            }
            if (x8) { x8[(size_t)bi * dq8 + i] = (uint8_t)(v ^ 0x80); s1 += v; s2 += v * v; }      // (the numerator itself, as a signed byte minus 128: k_svm_kernel_i8)
        }
        if (x8) {
            for (int i = 1800 + lane; i < dq8; i += 64) x8[(size_t)bi * dq8 + i] = 0x80;
            for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            if (lane == 0) { x8s[2 * (size_t)bi] = s1; x8s[2 * (size_t)bi + 1] = s2; }
        }
        if (xq) {
            for (int i = 1800 + lane; i < dq; i += 64) xq[(size_t)bi * dq + i] = 0;
            for (int o = 32; o > 0; o >>= 1) nrm += __shfl_xor(nrm, o);
            if (lane == 0) xnorm[bi] = nrm;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------
// libsvm inference
// ---------------------------------------------------------------------------------------------------------
// exp(x) for x <= 0 without the library's range cases: 2^n e^r, |r| <= ln 2 / 2, degree-11 Taylor (3e-17 relative); below e^-700 the result is 0
__device__ __forceinline__ double exp_neg(double x)
{
    x = fmax(x, -700.0);
    const double nn = __builtin_rint(x * 1.4426950408889634);
    double       r = __builtin_fma(nn, -6.93147180369123816490e-01, x);
    r = __builtin_fma(nn, -1.90821492927058770002e-10, r);
    double e = 1.0 / 39916800.0;
    e = __builtin_fma(e, r, 1.0 / 3628800.0); e = __builtin_fma(e, r, 1.0 / 362880.0); e = __builtin_fma(e, r, 1.0 / 40320.0);
    e = __builtin_fma(e, r, 1.0 / 5040.0); e = __builtin_fma(e, r, 1.0 / 720.0); e = __builtin_fma(e, r, 1.0 / 120.0);
    e = __builtin_fma(e, r, 1.0 / 24.0); e = __builtin_fma(e, r, 1.0 / 6.0); e = __builtin_fma(e, r, 0.5);
    e = __builtin_fma(e, r, 1.0); e = __builtin_fma(e, r, 1.0);
    return __builtin_amdgcn_ldexp(e, (int)nn);
}

// rows of a launch: the host's number, or -- a launch sized before the host knew it (OcrBuf::n_dev) -- the device's count, at most the number sized for
__device__ __forceinline__ int svm_count(const uint32_t *n_dev, int n) { return n_dev ? (int)min((uint32_t)n, *n_dev) : n; }

__global__ __launch_bounds__(256) void k_svm_prep(const double *__restrict__ x, int n, int dim, float *__restrict__ xf, int dpad,
                                                  double *__restrict__ xnorm)
{
    // one wave per vector
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n) return;
    double s = 0;
    for (int i = lane; i < dpad; i += 64) {
        const double d = i < dim ? x[(size_t)v * dim + i] : 0.0;
        xf[(size_t)v * dpad + i] = (float)d;
        s += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) xnorm[v] = s;
}

// K = exp(-gamma (|x|^2 + |sv|^2 - 2 x.sv)).  Workgroup tile 128 (vectors) x 64 (support vectors), four waves, each a 64 x 32
// sub-tile = two 32x32 accumulators sharing the B operand; K step 16, both operand tiles double-buffered in LDS as [k][row]
// (row stride padded to 132 / 68 words) so that an MFMA operand is one conflict-free ds_read_b32 per lane and a staging
// store is one ds_write_b32 per k -- the global side stays float4 along k.
constexpr int GM = 128, GN = 64, GK = 16;

__global__ __launch_bounds__(256) void k_svm_kernel(const float *__restrict__ xf, const double *__restrict__ xnorm, int n_rows,
                                                    const float *__restrict__ sv, const double *__restrict__ svnorm, int l_pad,
                                                    int dpad, double gamma, double *__restrict__ kv, const uint32_t *__restrict__ n_dev)
{
    __shared__ float As[2][GK][GM + 4];
    __shared__ float Bs[2][GK][GN + 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    n_rows = svm_count(n_dev, n_rows);
    if (m0 >= n_rows) return;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 32;
    // staging: A = 128 rows x 16 k = 512 float4 (2 per thread), B = 64 rows x 16 k = 256 float4 (1 per thread)
    const int arow0 = tid >> 2, arow1 = arow0 + 64, ak = (tid & 3) * 4;
    const float *pa0 = xf + (size_t)min(m0 + arow0, n_rows - 1) * dpad + ak;
    const float *pa1 = xf + (size_t)min(m0 + arow1, n_rows - 1) * dpad + ak;
    const float *pb = sv + (size_t)(n0 + arow0) * dpad + ak;
    float16v acc0 = {0}, acc1 = {0};
    float4 a0 = *reinterpret_cast<const float4 *>(pa0), a1 = *reinterpret_cast<const float4 *>(pa1), b0 = *reinterpret_cast<const float4 *>(pb);
    auto stage = [&](int buf) {
        As[buf][ak][arow0] = a0.x; As[buf][ak + 1][arow0] = a0.y; As[buf][ak + 2][arow0] = a0.z; As[buf][ak + 3][arow0] = a0.w;
        As[buf][ak][arow1] = a1.x; As[buf][ak + 1][arow1] = a1.y; As[buf][ak + 2][arow1] = a1.z; As[buf][ak + 3][arow1] = a1.w;
        Bs[buf][ak][arow0] = b0.x; Bs[buf][ak + 1][arow0] = b0.y; Bs[buf][ak + 2][arow0] = b0.z; Bs[buf][ak + 3][arow0] = b0.w;
    };
    stage(0);
    __syncthreads();
    const int nk = dpad / GK;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int ko = (kt + 1) * GK;
            a0 = *reinterpret_cast<const float4 *>(pa0 + ko); a1 = *reinterpret_cast<const float4 *>(pa1 + ko); b0 = *reinterpret_cast<const float4 *>(pb + ko);
        }
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const int   kr = kk + (lane >> 5), c = lane & 31;
            const float bv = Bs[cur][kr][wn + c];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(As[cur][kr][wm + c], bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(As[cur][kr][wm + 32 + c], bv, acc1, 0, 0, 0);
        }
        if (kt + 1 < nk) stage(cur ^ 1);
        __syncthreads();
    }
    // 32x32 accumulator layout: element i of lane L is row 8*(i/4) + 4*(L/32) + i%4, column L%32
    const int    col = n0 + wn + (lane & 31);
    const double sn = svnorm[col];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + wm + 32 * h + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            if (row < n_rows) {
                double d2 = xnorm[row] + sn - 2.0 * (double)(h ? acc1[i] : acc0[i]);
                d2 = d2 > 0 ? d2 : 0;
                kv[(size_t)row * l_pad + col] = exp_neg(-gamma * d2);
            }
        }
    }
}

// The same matrix for vectors that are 8-bit numerators over 255 (every vector k_ocr_features makes): x.sv = (q.sv) / 255 with q exact in bf16 and
// the f32 support vector split exactly into three bf16 pieces -- three v_mfma_f32_32x32x16_bf16 per 16 features (a 16-bit product is exact, the sum
// runs in f32 as before) instead of eight v_mfma_f32_32x32x2_f32 per accumulator: a fifth of the matrix-pipe time, half the bytes of the vectors.
// Workgroup tile 128 x 64, four waves of 64 x 32 (two accumulators sharing the B operands), K step 64; operands row-major in LDS, a row = 64 bf16
// + 8 of padding (144 bytes: the sixteen 16-byte reads of a quarter wave fall in disjoint banks); the next tile travels through registers.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int QM = 128, QN = 64, QK = 64, QS = QK + 8;

__global__ __launch_bounds__(256) void k_svm_kernel_q(const uint16_t *__restrict__ xq, const double *__restrict__ xnorm, int n_rows,
                                                      const uint16_t *__restrict__ svq, const double *__restrict__ svnorm, int l_pad, int dq,
                                                      double gamma, double *__restrict__ kv, const uint32_t *__restrict__ n_dev)
{
    __shared__ __attribute__((aligned(16))) uint16_t As[QM][QS];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[3][QN][QS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * QM, n0 = blockIdx.x * QN;
    n_rows = svm_count(n_dev, n_rows);
    if (m0 >= n_rows) return;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 32;
    // staging: a thread moves the 16-byte chunk `ch` of rows sr + 32 j: 4 of A, 2 of each B plane
    const int sr = tid >> 3, ch = (tid & 7) * 8;
    const uint16_t *pa[4], *pb[6];
#pragma unroll
    for (int j = 0; j < 4; ++j) pa[j] = xq + (size_t)min(m0 + sr + 32 * j, n_rows - 1) * dq + ch;
#pragma unroll
    for (int j = 0; j < 6; ++j) pb[j] = svq + ((size_t)(j >> 1) * l_pad + n0 + sr + 32 * (j & 1)) * dq + ch;
    // (the staging registers: native vectors in arrays indexed by unrolled loops of this scope -- as HIP's uint4 structs, or through a lambda's reference,
    // they end up in scratch memory)
    u32x4 ra[4], rb[6];
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const u32x4 *>(pa[j]);
#pragma unroll
    for (int j = 0; j < 6; ++j) rb[j] = *reinterpret_cast<const u32x4 *>(pb[j]);
    float16v acc0 = {0}, acc1 = {0};
    const int nk = dq / QK, fr = lane & 31, fk = 8 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4 *>(&As[sr + 32 * j][ch]) = ra[j];
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4 *>(&Bs[j >> 1][sr + 32 * (j & 1)][ch]) = rb[j];
        __syncthreads();
        if (kt + 1 < nk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const u32x4 *>(pa[j] + (kt + 1) * QK);
#pragma unroll
            for (int j = 0; j < 6; ++j) rb[j] = *reinterpret_cast<const u32x4 *>(pb[j] + (kt + 1) * QK);
        }
#pragma unroll
        for (int ks = 0; ks < QK; ks += 16) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(&As[wm + fr][ks + fk]), a1 = *reinterpret_cast<const bf16x8 *>(&As[wm + 32 + fr][ks + fk]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&Bs[pl][wn + fr][ks + fk]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc1, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // 32x32 accumulator layout: element i of lane L is row 8*(i/4) + 4*(L/32) + i%4, column L%32
    const int    col = n0 + wn + (lane & 31);
    const double sn = svnorm[col];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + wm + 32 * h + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            if (row < n_rows) {
                double d2 = xnorm[row] + sn - (2.0 / 255.0) * (double)(h ? acc1[i] : acc0[i]);
                d2 = d2 > 0 ? d2 : 0;
                kv[(size_t)row * l_pad + col] = exp_neg(-gamma * d2);
            }
        }
    }
}

// The same matrix when BOTH sides are 8-bit numerators over 255 -- every vector k_ocr_features makes, and the support vectors of a model trained on such
// vectors (the reference's: SvmDev::sv8): |x - sv|^2 = (sum a^2 + sum b^2 - 2 sum a b) / 255^2 with all three sums EXACT integers, sum a b from
// v_mfma_i32_32x32x32_i8 on a - 128, b - 128 (signed bytes; sum a b = sum (a - 128)(b - 128) + 128 (sum a + sum b) - 128^2 D, D = the padded row length).
// One matrix instruction per 32 features where the bf16 form takes six per 32: the kernel is then as long as its epilogue (exp, 8 bytes written per value).
// Same tiling as k_svm_kernel_q: workgroup 128 x 64, four waves of 64 x 32, K step 128 bytes, rows of 128 + 16 bytes in LDS, the next tile through registers.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int IK = 128, IS = IK + 16;
__global__ __launch_bounds__(256) void k_svm_kernel_i8(const uint8_t *__restrict__ x8, const int32_t *__restrict__ x8s, int n_rows, const uint8_t *__restrict__ sv8,
                                                       const int32_t *__restrict__ sv8s, int l_pad, int dq8, double gamma, double *__restrict__ kv,
                                                       const uint32_t *__restrict__ n_dev)
{
    __shared__ __attribute__((aligned(16))) uint8_t As[QM][IS];
    __shared__ __attribute__((aligned(16))) uint8_t Bs[QN][IS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * QM, n0 = blockIdx.x * QN;
    n_rows = svm_count(n_dev, n_rows);
    if (m0 >= n_rows) return;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 32;
    const int sr = tid >> 3, ch = (tid & 7) * 16;            // a thread moves the 16-byte chunk `ch` of rows sr + 32 j: 4 of A, 2 of B
    const uint8_t *pa[4], *pb[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) pa[j] = x8 + (size_t)min(m0 + sr + 32 * j, n_rows - 1) * dq8 + ch;
#pragma unroll
    for (int j = 0; j < 2; ++j) pb[j] = sv8 + (size_t)(n0 + sr + 32 * j) * dq8 + ch;
    u32x4 ra[4], rb[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const u32x4 *>(pa[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) rb[j] = *reinterpret_cast<const u32x4 *>(pb[j]);
    i32x16 acc0 = {0}, acc1 = {0};
    const int nk = dq8 / IK, fr = lane & 31, fk = 16 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4 *>(&As[sr + 32 * j][ch]) = ra[j];
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4 *>(&Bs[sr + 32 * j][ch]) = rb[j];
        __syncthreads();
        if (kt + 1 < nk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const u32x4 *>(pa[j] + (kt + 1) * IK);
#pragma unroll
            for (int j = 0; j < 2; ++j) rb[j] = *reinterpret_cast<const u32x4 *>(pb[j] + (kt + 1) * IK);
        }
#pragma unroll
        for (int ks = 0; ks < IK; ks += 32) {
            const i32x4 a0 = *reinterpret_cast<const i32x4 *>(&As[wm + fr][ks + fk]), a1 = *reinterpret_cast<const i32x4 *>(&As[wm + 32 + fr][ks + fk]);
            const i32x4 b = *reinterpret_cast<const i32x4 *>(&Bs[wn + fr][ks + fk]);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b, acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    // 32x32 accumulator layout: element i of lane L is row 8*(i/4) + 4*(L/32) + i%4, column L%32
    const int       col = n0 + wn + (lane & 31);
    const long long sb = sv8s[2 * (size_t)col], sb2 = sv8s[2 * (size_t)col + 1], base = 16384ll * dq8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = m0 + wm + 32 * h + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            if (row < n_rows) {
                const long long sa = x8s[2 * (size_t)row], sa2 = x8s[2 * (size_t)row + 1];
                const long long ab = (long long)(h ? acc1[i] : acc0[i]) + 128ll * (sa + sb) - base;
                const long long d2i = sa2 + sb2 - 2ll * ab;                      // sum (a - b)^2 >= 0, exact
                kv[(size_t)row * l_pad + col] = exp_neg(-gamma * ((double)d2i * (1.0 / 65025.0)));
            }
        }
    }
}

__device__ __forceinline__ double bcast(double v, int src)
{
    const unsigned long long u = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// (double)x that stays where it is written: the rows of Q are loop invariants of the sweeps, and hoisted out as doubles they no longer fit the registers
__device__ __forceinline__ double widen_here(float x)
{
    double d;
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x));
    return d;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// 1 / x for the coupling's step (x = a sum of probabilities, far from 0 and from the ends of the exponent range): the
// hardware estimate plus two Newton steps -- full f64 accuracy without the scaling and fix-up of a general division
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// Decision values, first half, for models with many support vectors a class (svm_uses_class_sums): per vector n, class c and slot j < k - 1
//   av[n][c][j] = sum over class c's support vectors q of sv_coef[j][q] K[n][q]
// (libsvm's sums of svm_predict_values, src/svm.cpp:2539-2566, split where it switches from class i's support vectors to class j's: the decision value of
// the pair i < j is av[i][j - 1] + av[j][i] - rho; slot j of class c is the other class j < c, or j + 1).  Per class that is a dense product -- kernel values
// [vectors x support vectors of c] times coefficients [support vectors of c x 64] -- and the one place of this path besides the kernel matrix where the matrix
// unit fits: v_mfma_f64_16x16x4_f64, a wave = 64 vectors x one class (16 accumulator tiles), operands straight from memory (a tile of kernel values is 16
// row pieces of 32 bytes, a tile of coefficients 4 x 128 bytes; every operand feeds four instructions).  2 l 64 flops a vector in all.
// (k_svm_couple walking the coefficient rows per vector did the same sums in 27 ms a batch of 16 k vectors on the 4299-vector model -- a dependent round
// trip per eight ranks, 2 MB of rows per vector; a first form with a lane per slot, eight vectors per wave and the kernel values by scalar loads took 1.1 ms:
// 8 bytes from memory per multiply-add, none of them reused.)
typedef double double4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_svm_decide(const double *__restrict__ kv, int l_pad, int n, SvmDev m, double *__restrict__ av, const uint32_t *__restrict__ n_dev)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int v0 = blockIdx.x * 64, c = blockIdx.y * 4 + wv;
    n = svm_count(n_dev, n);
    if (c >= m.k || v0 >= n) return;
    const int q0 = m.start[c], nq = m.nsv[c];
    const int r16 = lane & 15, kq = lane >> 4;
    double4v  acc[4][4];
#pragma unroll
    for (int bt = 0; bt < 4; ++bt)
#pragma unroll
        for (int st = 0; st < 4; ++st) acc[bt][st] = double4v{0.0, 0.0, 0.0, 0.0};
    // operand A: kernel value of vector v0 + 16 bt + (lane & 15), support vector q0 + step + (lane >> 4); B: coefficient of that support vector for slot
    // 16 st + (lane & 15), zero beyond the class (the fourth-rounded last step reads the next class's kernel values)
    const double *ka[4];
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) ka[bt] = kv + (size_t)min(v0 + 16 * bt + r16, n - 1) * l_pad + q0 + kq;
    const double *cb = m.coef_t + (size_t)(q0 + kq) * 64 + r16;
    // (the next step's eight operands are requested before this step's sixteen matrix instructions: with the accumulators in 128 registers there are only three
    // waves a SIMD to cover a trip to memory)
    double a[4], b[4];
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) a[bt] = ka[bt][0];
#pragma unroll
    for (int st = 0; st < 4; ++st) b[st] = cb[16 * st];
    for (int q = 0; q < nq; q += 4) {
        double     an[4], bn[4];
        const bool on = q + kq < nq;
        const int  qn = q + 4 < nq ? q + 4 : q;          // (the last step asks for its own operands again)
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) an[bt] = ka[bt][qn];
#pragma unroll
        for (int st = 0; st < 4; ++st) bn[st] = cb[(size_t)qn * 64 + 16 * st];
#pragma unroll
        for (int st = 0; st < 4; ++st) b[st] = on ? b[st] : 0.0;
#pragma unroll
        for (int bt = 0; bt < 4; ++bt)
#pragma unroll
            for (int st = 0; st < 4; ++st) acc[bt][st] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bt], b[st], acc[bt][st], 0, 0, 0);
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) a[bt] = an[bt];
#pragma unroll
        for (int st = 0; st < 4; ++st) b[st] = bn[st];
    }
    // result tile: element r of lane L is row (L >> 4) + 4 r, column L & 15
#pragma unroll
    for (int bt = 0; bt < 4; ++bt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = v0 + 16 * bt + kq + 4 * r;
            if (v < n) {
                double *dst = av + ((size_t)v * m.k + c) * 64 + r16;
#pragma unroll
                for (int st = 0; st < 4; ++st) dst[16 * st] = acc[bt][st][r];
            }
        }
}

// svm_predict_values + sigmoid_predict + multiclass_probability for one vector per wave.  Class-indexed vectors (p, Qp,
// Q's diagonal) live in registers: lane L holds class L (and L + 64 in the build for 66+ classes).  The pairwise table V (LDS, f32; row i =
// columns i .. k - 1 at rb(i) + j, rb(i) = i k - i (i + 1) / 2, the diagonal slot set to 1 so that the walks need no mask) holds the pairwise
// probabilities r_ij -- as f32: 8 KB instead of 16, which is what lets 16 waves share a compute unit's LDS.  An entry's rounding (6e-8
// relative) moves a probability by ~1e-8: Q is strongly diagonally dominant (Q_tt = sum of k - 1 squares, |Q_tj| <= 1/4).
//
// Decision values (svm_predict_values, src/svm.cpp:2539-2566: coef[j-1][q] K[q] over class i's support vectors, then coef[i][q] K[q] over class
// j's, then - rho).  k <= 65: row passes over SvmDev::coef_rows -- first class i per pass, second class j = lane + 1; see the comment in the
// kernel.  66+ classes: a lane per class pair, eight pairs of a lane at a time, branch-free.
//
// Coupling: the reference's Gauss-Seidel sweep divides p and Qp by (1 + diff) after every coordinate step (5 divisions
// per step, 2 of them on every lane).  Here the sweep runs on the unnormalised iterate -- p~ = p / sigma, B = Q p~,
// A = p~ Q p~, S = sum p~ = 1 / sigma -- for which a step is d = (A / S - B_t) / Q_tt; p~_t += d; B += d Q_t; A += d (d Q_tt + 2 B_t);
// S += d: one reciprocal on the critical path and one fused multiply-add per lane; the iterate is renormalised at the end
// of a sweep, where the reference's stopping test (max_t |Qp_t - pQp| < 0.005 / k, on the same quantities up to rounding)
// is evaluated.  The same fixed point, the same sweeps.  k <= 65: lane t keeps row t of Q in registers, a step reads no table.
// MODE 0: k <= 64.  MODE 1: k = 65 (the reference's 65 characters): the one class beyond the 64 lanes is carried as a wave-uniform value in
// every lane -- a second class register per lane would double the vector work of every step for one useful lane.  MODE 2: 66 <= k <= 125.
// MSV (MODE 0 / 1): 5 = at most five support vectors a class (the small stand-in model of rounds 1-5, tests/golden/make_svm_fixture.py: five samples a class;
// the reference's training set has 120, src/utils.cpp:1478-1541): class j's kernel values in registers, the next pass's loads issued ahead; 0 = any count up
// to eight ranks a class, in eights; -1 = the sums per (class, other class) come from k_svm_decide (models with more support vectors a class: the reference's
// shape -- svm_uses_class_sums).
// 4 waves per SIMD = the 16 waves per compute unit the LDS allows: 128 registers (a handful of values that live across a pass are parked in
// scratch, outside the loops)
// LDS of one box: QI[2 k], D[k], V[k (k + 1) / 2] floats, rounded up to 16 bytes
__host__ __device__ inline size_t svm_couple_lds_doubles(int k) { return ((size_t)3 * k + ((size_t)k * (k + 1) / 2 + 1) / 2 + 1) & ~(size_t)1; }
#ifndef SVM_COUPLE_WAVES
#define SVM_COUPLE_WAVES 4
#endif
#if SVM_COUPLE_WAVES > 0
#define SVM_COUPLE_OCC __attribute__((amdgpu_waves_per_eu(SVM_COUPLE_WAVES, SVM_COUPLE_WAVES)))
#else
#define SVM_COUPLE_OCC
#endif
// SVM_COUPLE_WPB boxes (waves) per workgroup, in step through the row passes (a barrier per pass): the passes stream the whole coefficient table
// (333 KB for the reference's shape) per box, far more than the 32 KB first-level cache keeps -- waves of one compute unit that read the same rows
// at the same time share the lines, waves that drift apart fetch them from the second level each on its own (6.4 GB a batch: the kernel's bound)
#ifndef SVM_COUPLE_WPB
#define SVM_COUPLE_WPB 1
#endif
constexpr int svm_couple_wpb(int mode) { return mode == 2 ? 1 : SVM_COUPLE_WPB; }      // (66+ classes: up to 34 KB a box, one box a workgroup)
template <int MODE, int MSV>
__global__ __launch_bounds__(64 * svm_couple_wpb(MODE)) SVM_COUPLE_OCC void k_svm_couple(const double *__restrict__ kv, int l_pad, int n, SvmDev m, double *__restrict__ dec_out,
                                                   double *__restrict__ prob, int32_t *__restrict__ label, double *__restrict__ pbest, const double *__restrict__ av,
                                                   const uint32_t *__restrict__ n_dev)
{
    constexpr bool TWO = MODE == 2, TAIL = MODE == 1;
    n = svm_count(n_dev, n);
    if ((int)blockIdx.x * svm_couple_wpb(MODE) >= n) return;          // (a launch sized for more boxes than the device counted)
    // LDS: QI[2 k] = {Q_tt, 1 / Q_tt} per class; D[k] = the sweep's steps; V[k (k + 1) / 2] pairwise table (f32)
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int k = TAIL ? 65 : m.k, np = k * (k - 1) / 2, nv = k * (k + 1) / 2, kc = m.kc, l = m.l;
    constexpr int WPB = svm_couple_wpb(MODE);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // (wave-uniform, and the compiler must know it: the box's rows are read with scalar loads)
    const int v_raw = blockIdx.x * WPB + wv, v = min(v_raw, n - 1);   // (a wave beyond the last box repeats it and stores nothing: it keeps the barriers)
    const bool live = v_raw < n;
    double *QI = lds_d + (size_t)wv * svm_couple_lds_doubles(k), *D = QI + 2 * k;
    float  *V = reinterpret_cast<float *>(D + k);
    const double *kr = kv + (size_t)v * l_pad;
    auto rb = [&](int i) -> int { return i * k - i * (i + 1) / 2; };         // row i = columns i .. k - 1 at rb(i) + j
    const int  t0 = lane, t1 = lane + 64;
    const bool on0 = MODE != 0 || t0 < k, on1 = TWO && t1 < k;         // (k >= 64: every lane has a class)
    const int  t0c = on0 ? t0 : 0;                                      // (an idle lane walks class 0's row and is masked where it matters)
    const int  rb0 = rb(t0c), rb1 = rb(t1);
    if (on0) V[rb0 + t0] = 1.f;                                         // r_tt := 1: its square (1 - r_tt)^2 and its product -(1 - r_tt) r_tt vanish without a mask
    if (on1) V[rb1 + t1] = 1.f;
    if (TAIL) V[rb(64) + 64] = 1.f;                                     // (every lane writes the same word)
    // ---- decision values, r_ij = sigmoid_predict(dec, A, B) clamped to [1e-7, 1 - 1e-7] (src/svm.cpp:2603-2611)
    const double min_prob = 1e-7;
    // sigmoid_predict (src/svm.cpp:1818-1826): exp(-f) / (1 + exp(-f)) for f >= 0, 1 / (1 + exp(f)) otherwise -- one exp of -|f| and one reciprocal serve
    // both.  exp on (-inf, 0] without the library's range cases: 2^n e^r, |r| <= ln 2 / 2, degree-11 Taylor (3e-17 relative), below e^-40 the result is
    // under the 1e-7 clamp whatever it is.  1 ulp here and in the reciprocal; the table entry is an f32.
    auto pair_prob = [min_prob](double fApB) -> double {
        const double x = fmax(-fabs(fApB), -40.0), nn = __builtin_rint(x * 1.4426950408889634);
        double       r = __builtin_fma(nn, -6.93147180369123816490e-01, x);
        r = __builtin_fma(nn, -1.90821492927058770002e-10, r);
        double e = 1.0 / 39916800.0;
        e = __builtin_fma(e, r, 1.0 / 3628800.0); e = __builtin_fma(e, r, 1.0 / 362880.0); e = __builtin_fma(e, r, 1.0 / 40320.0);
        e = __builtin_fma(e, r, 1.0 / 5040.0); e = __builtin_fma(e, r, 1.0 / 720.0); e = __builtin_fma(e, r, 1.0 / 120.0);
        e = __builtin_fma(e, r, 1.0 / 24.0); e = __builtin_fma(e, r, 1.0 / 6.0); e = __builtin_fma(e, r, 0.5);
        e = __builtin_fma(e, r, 1.0); e = __builtin_fma(e, r, 1.0);
        const double ex = __builtin_amdgcn_ldexp(e, (int)nn);
        double sg = (fApB >= 0 ? ex : 1.0) * rcp_nr(1.0 + ex);
        sg = sg > min_prob ? sg : min_prob;
        return sg < 1 - min_prob ? sg : 1 - min_prob;
    };
    if constexpr (!TWO) {
        // k <= 65: a lane is a second class j = lane + 1, a pass is a first class i (a row of the pair table): class i's kernel values are wave-uniform
        // (scalar loads), its coefficients one contiguous row per support vector (coef_rows[i][0], zero beyond nr_sv[i]); class j's kernel values are the
        // same for every row -- MSV registers, loaded once -- and its coefficients for first class i again one row per rank (coef_rows[i][1], zero beyond
        // nr_sv[j]).  Every load is a full 512-byte line set off a uniform base: the p-indexed walk below issued 640 scattered 8-byte loads per lane (20
        // cache lines an instruction) and was bound by the address pipeline, not by arithmetic.  The sums run in libsvm's order (adding 0 K is exact).
        // Rows i and k - 1 - i have 64 pairs between them (k = 65) and share the loads in flight, the sigmoid and the table write.
        constexpr int CH = MSV > 0 ? MSV : 8;
        const int     sj = m.start[min(lane + 1, k - 1)];
        double        K2[CH];
        if constexpr (MSV >= 0) {
#pragma unroll
            for (int u = 0; u < CH; ++u) K2[u] = kr[min(sj + u, l - 1)];      // (MSV = 0, more than 8 ranks: read again per row instead)
        }
        const double *tab = m.coef_rows + lane;
        auto pb = [&](int r) -> int { return r * (k - 1) - r * (r - 1) / 2 - r - 1; };      // pair (r, j) is libsvm's pair number pb(r) + j
        // pass a: rows a and b = k - 1 - a.  Lanes >= a: pair (a, lane + 1); lanes < a: pair (b, lane + b + 1), summed by lane + b
        auto pair_of = [&](int a, bool &ra, bool &on, int &row) -> int {
            const int b = k - 1 - a;
            ra = lane >= a; on = ra ? lane < k - 1 : b != a; row = ra ? a : b;
            return on ? (ra ? pb(a) + lane + 1 : pb(b) + lane + b + 1) : 0;
        };
        auto finish = [&](int a, double sa, double sb, double rho, double pA, double pB) {
            bool ra, on; int row;
            const int    p = pair_of(a, ra, on, row);
            const double so = __shfl(sb, (lane + k - 1 - a) & 63);
            if (on) {
                const double d = (ra ? sa : so) - rho;
                if (dec_out && live) dec_out[(size_t)v * np + p] = d;
                V[p + row + 1] = (float)pair_prob(d * pA + pB);           // (rb(row) + j: the rows have a diagonal slot)
            }
        };
        const int n_pass = (k - 1) / 2 + 1;
        if constexpr (MSV < 0) {
            // the sums per (class, other class) are there already (k_svm_decide): a pair i < j is av[i][j - 1] + av[j][i] - rho -- the first a lane's word of
            // a 512-byte row, the second one word of 64 different rows (66 such loads a lane for the whole table: the first-level cache holds the vector's 33 KB)
            const double *A = av + (size_t)v * k * 64;
            for (int a = 0; a < n_pass; ++a) {
                bool ra, on; int row;
                const int p = pair_of(a, ra, on, row);
                const int j = on ? (ra ? lane + 1 : lane + (k - 1 - a) + 1) : 1, i = on ? row : 0;       // (an idle lane reads pair (0, 1) and drops it)
                const double s1 = A[(size_t)i * 64 + (j - 1)], s2 = A[(size_t)j * 64 + i];
                const double rho = m.rho[p], pA = m.probA[p], pB = m.probB[p];
                if (on) {
                    const double d = (s1 + s2) - rho;
                    if (dec_out && live) dec_out[(size_t)v * np + p] = d;
                    V[p + row + 1] = (float)pair_prob(d * pA + pB);
                }
            }
        } else if constexpr (MSV > 0) {
            // a pass is one memory round trip (46 values a lane at MSV = 5): the next pass's loads are issued as soon as this one's sums are taken,
            // ahead of its sigmoid and table write
            struct Loads { double c[4][CH], kq[2][CH], rho, pA, pB; };
            auto issue = [&](int a, Loads &L) {
                const int b = k - 1 - a, sa = m.start[a], sb = m.start[b];
                const double *ta = tab + (size_t)(2 * a) * MSV * 64, *tb = tab + (size_t)(2 * b) * MSV * 64;
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    L.c[0][u] = ta[(size_t)u * 64]; L.c[1][u] = ta[(size_t)(MSV + u) * 64];
                    L.c[2][u] = tb[(size_t)u * 64]; L.c[3][u] = tb[(size_t)(MSV + u) * 64];
                    L.kq[0][u] = kr[min(sa + u, l - 1)]; L.kq[1][u] = kr[min(sb + u, l - 1)];
                }
                bool ra, on; int row;
                const int p = pair_of(a, ra, on, row);
                L.rho = m.rho[p]; L.pA = m.probA[p]; L.pB = m.probB[p];
            };
            Loads L;
            issue(0, L);
            for (int a = 0; a < n_pass; ++a) {
                double sa = 0, sb = 0;
#pragma unroll
                for (int u = 0; u < CH; ++u) sa = __builtin_fma(L.c[0][u], L.kq[0][u], sa);
#pragma unroll
                for (int u = 0; u < CH; ++u) sa = __builtin_fma(L.c[1][u], K2[u], sa);
#pragma unroll
                for (int u = 0; u < CH; ++u) sb = __builtin_fma(L.c[2][u], L.kq[1][u], sb);
#pragma unroll
                for (int u = 0; u < CH; ++u) sb = __builtin_fma(L.c[3][u], K2[u], sb);
                const double rho = L.rho, pA = L.pA, pB = L.pB;
                if (WPB > 1) __syncthreads();                             // (the workgroup's waves ask for the same rows together)
                if (a + 1 < n_pass) issue(a + 1, L);
                finish(a, sa, sb, rho, pA, pB);
            }
        } else {
            const int MP = m.mp;
            auto row_sum = [&](int i) -> double {
                const int     si = m.start[i];
                const double *t1 = tab + (size_t)(2 * i) * MP * 64, *t2 = t1 + (size_t)MP * 64;
                double        sum = 0;
                for (int m0 = 0; m0 < MP; m0 += CH) {
                    double c[CH], kq[CH];
#pragma unroll
                    for (int u = 0; u < CH; ++u) { c[u] = t1[(size_t)(m0 + u) * 64]; kq[u] = kr[min(si + m0 + u, l - 1)]; }
#pragma unroll
                    for (int u = 0; u < CH; ++u) sum = __builtin_fma(c[u], kq[u], sum);
                }
                for (int m0 = 0; m0 < MP; m0 += CH) {
                    double c[CH], kq[CH];
#pragma unroll
                    for (int u = 0; u < CH; ++u) { c[u] = t2[(size_t)(m0 + u) * 64]; kq[u] = kr[min(sj + m0 + u, l - 1)]; }
#pragma unroll
                    for (int u = 0; u < CH; ++u) sum = __builtin_fma(c[u], kq[u], sum);
                }
                return sum;
            };
            for (int a = 0; a < n_pass; ++a) {
                bool ra, on; int row;
                if (WPB > 1) __syncthreads();
                const int    p = pair_of(a, ra, on, row);
                const double rho = m.rho[p], pA = m.probA[p], pB = m.probB[p];
                const double sa = row_sum(a), sb = row_sum(k - 1 - a);
                finish(a, sa, sb, rho, pA, pB);
            }
        }
    } else {
        // PCN pairs of a lane at a time (pairs c0 + 64 u + lane).  Branch-free: a rank beyond the class reads the class's first support vector and
        // leaves the sum alone -- written as `if (q < na[u])` the compiler gives every pair its own block with a full wait on its two loads: 16
        // dependent memory round trips per rank instead of 16 loads in flight (and 24 more per pass for the class tables).
        auto pass = [&](int c0, auto pcn) {
            constexpr int PCN = decltype(pcn)::value;
            int    ci[PCN], cj[PCN], sa[PCN], na[PCN], sb[PCN], nb[PCN];
            double sum[PCN];
            int    qa = 0, qb = 0;
    #pragma unroll
            for (int u = 0; u < PCN; ++u) {
                const SvmPair pr = m.pairs[c0 + 64 * u + lane];               // (the table is padded to whole waves with pairs without support vectors)
                ci[u] = pr.ci; cj[u] = pr.cj;
                sa[u] = pr.sa; na[u] = pr.na;
                sb[u] = pr.sb; nb[u] = pr.nb;
                sum[u] = 0;
                qa = max(qa, na[u]); qb = max(qb, nb[u]);
            }
            for (int o = 32; o > 0; o >>= 1) { qa = max(qa, __shfl_xor(qa, o)); qb = max(qb, __shfl_xor(qb, o)); }
    #pragma unroll SVM_Q_UNROLL
            for (int q = 0; q < qa; ++q) {
                double c[PCN], kq[PCN];
    #pragma unroll
                for (int u = 0; u < PCN; ++u) {
                    const int r = sa[u] + (q < na[u] ? q : 0);
                    c[u] = m.coef_t[(size_t)r * kc + cj[u] - 1];
                    kq[u] = kr[r];
                }
    #pragma unroll
                for (int u = 0; u < PCN; ++u) { const double t = sum[u] + c[u] * kq[u]; sum[u] = q < na[u] ? t : sum[u]; }
            }
    #pragma unroll SVM_Q_UNROLL
            for (int q = 0; q < qb; ++q) {
                double c[PCN], kq[PCN];
    #pragma unroll
                for (int u = 0; u < PCN; ++u) {
                    const int r = sb[u] + (q < nb[u] ? q : 0);
                    c[u] = m.coef[(size_t)ci[u] * l + r];
                    kq[u] = kr[r];
                }
    #pragma unroll
                for (int u = 0; u < PCN; ++u) { const double t = sum[u] + c[u] * kq[u]; sum[u] = q < nb[u] ? t : sum[u]; }
            }
            double rho[PCN], pA[PCN], pB[PCN];
    #pragma unroll
            for (int u = 0; u < PCN; ++u) {
                const int p = min(c0 + 64 * u + lane, np - 1);
                rho[u] = m.rho[p]; pA[u] = m.probA[p]; pB[u] = m.probB[p];
            }
    #pragma unroll
            for (int u = 0; u < PCN; ++u) {
                const int p = c0 + 64 * u + lane;
                if (p < np) {
                    const double d = sum[u] - rho[u];
                    if (dec_out && live) dec_out[(size_t)v * np + p] = d;
                    V[p + ci[u] + 1] = (float)pair_prob(d * pA[u] + pB[u]);  // (rb(i) + j: the rows have a diagonal slot)
                }
            }
        };
        {
            constexpr int PC = 8;
            int c0 = 0;
            for (; c0 + 64 * PC <= np; c0 += 64 * PC) pass(c0, std::integral_constant<int, PC>{});
            for (; c0 < np; c0 += 64) pass(c0, std::integral_constant<int, 1>{});       // (k = 65: 2080 pairs = 4 passes of 512 + 32)
        }
    }
    __builtin_amdgcn_wave_barrier();
    const double ik = 1.0 / k;
    const int    max_iter = k > 100 ? k : 100;
    const double eps = 0.005 / k;
    double p0, p1 = 0, pT = 0;
    if constexpr (!TWO) {
        // ---- k <= 65: lane t keeps row t of Q in registers (f32; Q_tj = -r_jt r_tj, r_jt = V(j, t) for j < t and 1 - V(t, j) for j > t).  One
        // unrolled walk over j gives the row, Q_tt = sum_j r_jt^2 and B_t = sum_j Q_tj / k for p = 1 / k; the sweeps then read no table at all.
        constexpr int NR = TAIL ? 65 : 64;
        float  Qr[NR];
        double qd0 = 0, B0 = 0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            Qr[j] = 0.f;
            if (MODE != 0 || j < k) {
                const float  ea = V[rb(j) + t0c], eb = V[rb0 + j];
                const bool   lo = j < t0c;
                const double e = (double)(lo ? ea : eb), om = 1.0 - e, r = lo ? e : om, w = (-om) * e;
                qd0 = __builtin_fma(r, r, qd0);
                B0 = __builtin_fma(w, ik, B0);
                Qr[j] = (float)w;
            }
        }
        B0 = __builtin_fma(qd0, ik, B0);
        // the uniform class 64 (MODE 1): lane j holds its pair (j, 64); sums over the wave (the order of the additions differs from the reference's: 1 ulp)
        double qdT = 0, BT = 0;
        if (TAIL) {
            const double e = (double)V[rb0 + 64];
            qdT = wave_sum(e * e);
            BT = wave_sum(((-(1.0 - e)) * e) * ik) + qdT * ik;
            pT = ik;
        }
        if (on0) { QI[2 * t0] = qd0; QI[2 * t0 + 1] = 1.0 / qd0; }
        if (TAIL) { QI[128] = qdT; QI[129] = 1.0 / qdT; }              // (every lane writes the same words)
        __builtin_amdgcn_wave_barrier();
        p0 = on0 ? ik : 0.0;
        double A = wave_sum(p0 * B0) + pT * BT;
        for (int iter = 0; iter < max_iter; ++iter) {
            double err = on0 ? fabs(B0 - A) : 0.0;
            for (int o = 32; o > 0; o >>= 1) err = fmax(err, __shfl_xor(err, o));
            if (TAIL) err = fmax(err, fabs(BT - A));
            if (err < eps) break;
            double S = 1.0;
            // one coordinate step (the step of class t on its own B_t -- the diagonal -- is applied after the sweep: nothing reads B_t again before).
            // A, S, B and p follow the d that was applied, whatever it is: the hardware's 1 / S (24 bits) makes the step a coordinate step with a
            // relaxation factor 1 +- 1e-7 -- the same fixed point, the stopping test is evaluated on exact quantities.
            auto step = [&](double Bt, const double2 qi) -> double {
                const double d = __builtin_fma(__builtin_amdgcn_rcp(S), A, -Bt) * qi.y;
                A = __builtin_fma(d, __builtin_fma(d, qi.x, Bt + Bt), A);
                S += d;
                return d;
            };
#pragma unroll
            for (int t = 0; t < 64; ++t) {
                if (MODE != 0 || t < k) {                              // (no break: the loop must unroll for Qr to stay in registers)
                    const double d = step(bcast(B0, t), *reinterpret_cast<const double2 *>(QI + 2 * t));
                    D[t] = d;                                          // (every lane writes the same word)
                    B0 = __builtin_fma(d, widen_here(Qr[t]), B0);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const double dt = on0 ? D[t0] : 0.0;                       // this lane's class's step
            if (TAIL) {
                // class 64's B is read by its own step only: the 64 updates d_t Q(64, t) are one sum over the lanes
                BT += wave_sum(dt * widen_here(Qr[NR - 1]));
                const double d = step(BT, *reinterpret_cast<const double2 *>(QI + 128));
                B0 = __builtin_fma(d, widen_here(Qr[NR - 1]), B0);
                pT += d; BT = __builtin_fma(d, qdT, BT);
            }
            const double sg = rcp_nr(S);
            if (on0) { p0 += dt; B0 = __builtin_fma(dt, qd0, B0); }
            p0 *= sg; B0 *= sg; A *= sg * sg; pT *= sg; BT *= sg;
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        // ---- 66 <= k <= 125: two classes per lane, Q's off-diagonal entries replace the r_ij in the table
        // one walk over j for every class t of this lane: Q_tt = sum_{j != t} r_jt^2 and B_t = sum_j Q_tj / k for p = 1 / k, j ascending
        double qd0 = 0, qd1 = 0, B0 = 0, B1 = 0;
        for (int j = 0; j < k; ++j) {
            const int rbj = rb(j);
            {
                const double e = (double)V[j < t0 ? rbj + t0 : rb0 + j], om = 1.0 - e, r = j < t0 ? e : om;
                qd0 = __builtin_fma(r, r, qd0);
                B0 = __builtin_fma((-om) * e, ik, B0);
            }
            {
                const double e = (double)V[min(j < t1 ? rbj + t1 : rb1 + j, nv - 1)], om = 1.0 - e, r = j < t1 ? e : om;
                qd1 = on1 ? __builtin_fma(r, r, qd1) : 0.0;
                B1 = on1 ? __builtin_fma((-om) * e, ik, B1) : 0.0;
            }
        }
        B0 = __builtin_fma(qd0, ik, B0);
        B1 = on1 ? __builtin_fma(qd1, ik, B1) : 0.0;
        QI[2 * t0] = qd0; QI[2 * t0 + 1] = 1.0 / qd0;
        if (on1) { QI[2 * t1] = qd1; QI[2 * t1 + 1] = 1.0 / qd1; }
        __builtin_amdgcn_wave_barrier();
        for (int p = lane; p < nv; p += 64) { const double e = (double)V[p]; V[p] = (float)((-(1.0 - e)) * e); }      // (the diagonal slots become -0)
        __builtin_amdgcn_wave_barrier();
        p0 = ik; p1 = on1 ? ik : 0.0;
        double A = wave_sum(p0 * B0 + p1 * B1);
        for (int iter = 0; iter < max_iter; ++iter) {
            double err = fabs(B0 - A);
            if (on1) err = fmax(err, fabs(B1 - A));
            for (int o = 32; o > 0; o >>= 1) err = fmax(err, __shfl_xor(err, o));
            if (err < eps) break;
            double S = 1.0;
            auto step = [&](int t, double Bt) {
                const double2 qi = *reinterpret_cast<const double2 *>(QI + 2 * t);
                const double  d = (rcp_nr(S) * A - Bt) * qi.y;
                A = A + d * (d * qi.x + 2.0 * Bt);
                S += d;
                D[t] = d;                                              // (every lane writes the same word)
                const int rbt = rb(t);
                B0 = __builtin_fma(d, (double)V[t0 > t ? rbt + t0 : rb0 + t], B0);                    // (t = t0: the diagonal slot, -0)
                B1 = __builtin_fma(d, (double)V[min(t1 > t ? rbt + t1 : rb1 + t, nv - 1)], B1);       // (an idle second class: any entry, B1 is not read)
            };
            for (int t = 0; t < 64; ++t) step(t, bcast(B0, t));
            for (int t = 64; t < k; ++t) step(t, bcast(B1, t - 64));
            __builtin_amdgcn_wave_barrier();
            const double sg = rcp_nr(S);
            { const double d = D[t0]; p0 += d; B0 = __builtin_fma(d, qd0, B0); }
            if (on1) { const double d = D[t1]; p1 += d; B1 = __builtin_fma(d, qd1, B1); }
            p0 *= sg; p1 *= sg; B0 *= sg; B1 *= sg; A *= sg * sg;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (prob && live) {
        if (on0) prob[(size_t)v * k + t0] = p0;
        if (on1) prob[(size_t)v * k + t1] = p1;
        if (TAIL && lane == 0) prob[(size_t)v * k + 64] = pT;
    }
    // arg max with the reference's tie rule (first maximum, src/svm.cpp:2614-2617)
    double best = on0 ? p0 : -1.0;
    int    bi = t0;
    if (on1 && p1 > best) { best = p1; bi = t1; }
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int    oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (TAIL && pT > best) { best = pT; bi = 64; }
    // prob of the result = pv[label]: the reference indexes pv with the label VALUE (src/OCR.cpp:92-93), which is the arg max's entry only for a model whose
    // labels are 0 .. k - 1 in order; a label outside [0, k) makes the reference read outside pv -- the arg max's probability is returned then
    const int lab = m.label[bi];
    double    pf = best;
    if (lab >= 0 && lab < k) pf = lab < 64 ? bcast(p0, lab) : (TAIL ? pT : bcast(p1, (lab - 64) & 63));
    if (lane == 0 && live) { label[v] = lab; pbest[v] = pf; }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

OcrBuf ocr_layout(uint8_t *base, size_t n, const SvmDev *m, bool want_q, bool want_dec, bool want_prob)
{
    OcrBuf b{};
    size_t off = 0;
    auto take = [&](size_t bytes) -> uint8_t * { uint8_t *p = base ? base + off : nullptr; off = up256(off + bytes); return p; };
    auto skip = [](uint8_t *p, bool on) -> uint8_t * { return on ? p : nullptr; };
    const size_t n_pad = (n + 127) / 128 * 128;
    b.hist = reinterpret_cast<uint32_t *>(take(n * 256 * 4));
    b.thresh = reinterpret_cast<int32_t *>(take(n * 4));
    b.big = reinterpret_cast<uint32_t *>(take(4 * (size_t)(OCR_BIG_CAP + 1)));
    uint8_t *q = take(want_q ? n * 1800 : 0);
    b.q = base ? skip(q, want_q) : nullptr;
    if (m) {
        const size_t np = (size_t)m->k * (m->k - 1) / 2;
        b.xf = reinterpret_cast<float *>(take(n_pad * m->dpad * 4));
        b.xq = reinterpret_cast<uint16_t *>(take(n_pad * m->dq * 2));
        b.xnorm = reinterpret_cast<double *>(take(n_pad * 8));
        uint8_t *x8 = take(m->sv8 ? n_pad * (size_t)m->dq8 : 0), *x8s = take(m->sv8 ? n_pad * 8 : 0);
        b.x8 = base && m->sv8 ? x8 : nullptr; b.x8s = reinterpret_cast<int32_t *>(base && m->sv8 ? x8s : nullptr);
        b.kv = reinterpret_cast<double *>(take(n_pad * m->l_pad * 8));
        uint8_t *av = take(svm_uses_class_sums(*m) ? n_pad * (size_t)m->k * 64 * 8 : 0);
        b.av = reinterpret_cast<double *>(base ? skip(av, svm_uses_class_sums(*m)) : nullptr);
        uint8_t *d = take(want_dec ? n * np * 8 : 0), *p = take(want_prob ? n * m->k * 8 : 0);
        b.dec = reinterpret_cast<double *>(base ? skip(d, want_dec) : nullptr);
        b.prob = reinterpret_cast<double *>(base ? skip(p, want_prob) : nullptr);
        b.label = reinterpret_cast<int32_t *>(take(n * 4));
        b.pbest = reinterpret_cast<double *>(take(n * 8));
    }
    b.bytes = off;
    return b;
}

void launch_ocr_list(hipStream_t s, const BatchDev &b, uint32_t n_cands, uint32_t *hdr)
{
    if (!n_cands) { (void)hipMemsetAsync(hdr, 0, 4, s); return; }
    const uint32_t chunks = (n_cands + LIST_CHUNK - 1) / LIST_CHUNK, G = chunks < 256u ? chunks : 256u;
    hipLaunchKernelGGL(k_ocr_list<false>, dim3(G), dim3(LIST_THREADS), 0, s, (const CandRec *)b.cands, (const uint32_t *)b.total_cands, hdr, hdr + OCR_LIST_HDR);
    hipLaunchKernelGGL(k_ocr_list<true>, dim3(G), dim3(LIST_THREADS), 0, s, (const CandRec *)b.cands, (const uint32_t *)b.total_cands, hdr, hdr + OCR_LIST_HDR);
}

void launch_ocr_list_from(hipStream_t s, const BatchDev &b, const uint32_t *from, const uint32_t *from_n, uint32_t *hdr)
{
    hipLaunchKernelGGL(k_ocr_list_from, dim3(1), dim3(256), 0, s, (const CandRec *)b.cands, from, from_n, hdr, hdr + OCR_LIST_HDR);
}

int ocr_n_cu()
{
    // (initialised once, thread-safe; the compute-unit count of the first device used -- the library is built for one kind of GPU)
    static const int n_cu = [] { int dev = 0; hipDeviceProp_t prop{}; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256; }();
    return n_cu;
}

// grids of one round of resident workgroups (k_ocr_hist: 32 KB of LDS -> 5 per compute unit, k_ocr_features: 45 KB -> 3): a second, partly
// filled round would cost as much as a full one
void launch_box_thresholds(hipStream_t s, const OcrSrc &src, int n, uint32_t *hist, uint32_t *big, int32_t *thresh)
{
    if (n <= 0) return;
    const int wg = (n + OCR_WAVES - 1) / OCR_WAVES, n_cu = ocr_n_cu();
    (void)hipMemsetAsync(big, 0, 4, s);
    hipLaunchKernelGGL(k_ocr_hist, dim3(wg < 5 * n_cu ? wg : 5 * n_cu), dim3(64 * OCR_WAVES), 0, s, src, n, hist, big);
    hipLaunchKernelGGL(k_ocr_hist_big, dim3(OCR_BIG_PARTS, 64), dim3(64 * OCR_WAVES), 0, s, src, hist, (const uint32_t *)big);
    hipLaunchKernelGGL(k_ocr_otsu, dim3((n + 63) / 64), dim3(256), 0, s, src, n, (const uint32_t *)hist, thresh);
}

void launch_ocr_features(hipStream_t s, const OcrSrc &src, int n, const OcrBuf &buf, const SvmDev *m)
{
    if (n <= 0) return;
    const int wg = (n + OCR_WAVES - 1) / OCR_WAVES, n_cu = ocr_n_cu();
    launch_box_thresholds(s, src, n, buf.hist, buf.big, buf.thresh);
    hipLaunchKernelGGL(k_ocr_features, dim3(wg < 3 * n_cu ? wg : 3 * n_cu), dim3(64 * OCR_WAVES), 0, s, src, n, (const int32_t *)buf.thresh, buf.q,
                       m && !buf.x8 ? buf.xq : (uint16_t *)nullptr, m ? buf.xnorm : (double *)nullptr, m ? m->dq : 0, m ? buf.x8 : (uint8_t *)nullptr, m ? buf.x8s : (int32_t *)nullptr,
                       m ? m->dq8 : 0);
}

void launch_svm_prep(hipStream_t s, const double *x, int n, int dim, const OcrBuf &buf, const SvmDev &m)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_svm_prep, dim3((n + 3) / 4), dim3(256), 0, s, x, n, dim, buf.xf, m.dpad, buf.xnorm);
}

void launch_svm_kernel(hipStream_t s, int n, const OcrBuf &buf, const SvmDev &m, bool numerators)
{
    if (n <= 0) return;
    if (numerators && m.sv8 && buf.x8) {
        hipLaunchKernelGGL(k_svm_kernel_i8, dim3(m.l_pad / QN, (n + QM - 1) / QM), dim3(256), 0, s, (const uint8_t *)buf.x8, (const int32_t *)buf.x8s, n, m.sv8, m.sv8s, m.l_pad,
                           m.dq8, m.gamma, buf.kv, buf.n_dev);
        return;
    }
    if (numerators) {
        hipLaunchKernelGGL(k_svm_kernel_q, dim3(m.l_pad / QN, (n + QM - 1) / QM), dim3(256), 0, s, (const uint16_t *)buf.xq, (const double *)buf.xnorm, n, m.svq,
                           m.svnorm, m.l_pad, m.dq, m.gamma, buf.kv, buf.n_dev);
        return;
    }
    hipLaunchKernelGGL(k_svm_kernel, dim3(m.l_pad / GN, (n + GM - 1) / GM), dim3(256), 0, s, (const float *)buf.xf, (const double *)buf.xnorm, n, m.sv,
                       m.svnorm, m.l_pad, m.dpad, m.gamma, buf.kv, buf.n_dev);
}

void launch_svm_couple(hipStream_t s, int n, const OcrBuf &buf, const SvmDev &m)
{
    if (n <= 0) return;
    const int    wpb = svm_couple_wpb(m.k > 65 ? 2 : 0), wg = (n + wpb - 1) / wpb;
    const size_t lds = sizeof(double) * svm_couple_lds_doubles(m.k) * wpb;
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(wg), dim3(64 * wpb), lds, s, (const double *)buf.kv, m.l_pad, n, m, buf.dec, buf.prob, buf.label, buf.pbest, (const double *)buf.av, buf.n_dev); };
    if (svm_uses_class_sums(m) && buf.av) {
        hipLaunchKernelGGL(k_svm_decide, dim3((n + 63) / 64, (m.k + 3) / 4), dim3(256), 0, s, (const double *)buf.kv, m.l_pad, n, m, buf.av, buf.n_dev);
        if (m.k == 65) go(k_svm_couple<1, -1>); else go(k_svm_couple<0, -1>);
        return;
    }
    if (m.k > 65) go(k_svm_couple<2, 0>);
    else if (m.k == 65) { if (m.mp == 5) go(k_svm_couple<1, 5>); else go(k_svm_couple<1, 0>); }
    else { if (m.mp == 5) go(k_svm_couple<0, 5>); else go(k_svm_couple<0, 0>); }
}

void launch_svm_score(hipStream_t s, int n, const OcrBuf &buf, const SvmDev &m, bool numerators)
{
    launch_svm_kernel(s, n, buf, m, numerators);
    launch_svm_couple(s, n, buf, m);
}

} // namespace str_er
