// er_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the extremal-region
// hot path.  One kernel per function of the reference's per-plane loop
// (src/ER.cpp:50-60); citations are to /root/reference.
//
//   k_bgr_to_ycrcb   compute_channels                 src/ER.cpp:114-128
//   k_resize         cv::resize (pyramid + ARAN)      src/OCR.cpp:401
//   k_tile_tree      er_tree_extract inside a tile    src/ER.cpp:240-413
//   k_seam           ... across tile borders
//   k_resolve / k_accumulate / k_root / k_select / k_kept
//                    er_accumulate + er_merge + pruning   src/ER.cpp:131-191
//   k_nms            non_maximum_supression           src/ER.cpp:416-505
//   k_classify       classify -> make_LBP_hist -> calc_LBP -> ARAN -> CascadeBoost::predict
//                                                     src/ER.cpp:507-528, 789-845
//
// The component tree is NOT computed by the reference's sequential flood.  Each
// 64x32 tile builds its tree in LDS with a lock-free "connect" (merge of two sorted
// root paths, LDS compare-and-swap); tile trees are then joined along the seams with
// the same connect on global memory (agent-scope atomics).  The node set, levels,
// boxes and areas of a component tree do not depend on the order in which it is
// built, so the result equals the flood's (tests/ check this bit for bit).
//
// Compile with -ffp-contract=off: the resize coefficients must be computed with the
// same IEEE operations as the host statement of cv::resize.
#include <hip/hip_runtime.h>

#include <float.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "er_device.h"
#include "er_kernels.h"

namespace str_er {

// ------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------
#define LD_AGENT(p)      __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ST_AGENT(p, v)   __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LD_WG(p)         __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

__device__ __forceinline__ int find_plane_by_tile(const PlaneDesc *pl, int n, uint32_t tile)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pl[mid].tile_base <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ int find_plane_by_pair(const PlaneDesc *pl, int n, uint32_t pair)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pl[mid].pair_base <= pair) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ------------------------------------------------------------------------------------
// compute_channels (src/ER.cpp:114-128): OpenCV 8-bit BGR2YCrCb, yuv_shift = 14.
// One lane converts 4 pixels: 12 bytes in (three dwords), three dwords out.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void ycrcb_px(int B, int G, int R, int &Y, int &Cr, int &Cb)
{
    Y  = (1868 * B + 9617 * G + 4899 * R + 8192) >> 14;
    Cr = ((R - Y) * 11682 + (128 << 14) + 8192) >> 14;
    Cb = ((B - Y) * 9241 + (128 << 14) + 8192) >> 14;
    Y  = min(max(Y, 0), 255);
    Cr = min(max(Cr, 0), 255);
    Cb = min(max(Cb, 0), 255);
    // Toolchain hazard (ROCm 7.2 hipcc, gfx950): when two such clamped shifts are packed into
    // bytes, LLVM fuses them into v_ashr_pk_u8_i32 and then ORs further bytes into the result
    // assuming bits 31:16 are zero -- on MI355X they keep the old register contents, which
    // corrupted byte 2 of every packed Cr/Cb dword.  The empty asm makes each value opaque so
    // the fusion cannot happen.
    asm volatile("" : "+v"(Y));
    asm volatile("" : "+v"(Cr));
    asm volatile("" : "+v"(Cb));
}

__global__ __launch_bounds__(256) void k_bgr_to_ycrcb(const uint8_t *__restrict__ bgr, int w, int h,
                                                      int64_t stride, int64_t frame_pitch,
                                                      uint8_t *__restrict__ yp, uint8_t *__restrict__ crp,
                                                      uint8_t *__restrict__ cbp, int dstride,
                                                      int64_t dst_frame_pitch, int aligned)
{
    const int quad = blockIdx.x * blockDim.x + threadIdx.x; // 4-pixel group in the row
    const int y = blockIdx.y, f = blockIdx.z;
    const int x = quad * 4;
    if (x >= w) return;
    const uint8_t *src = bgr + (size_t)f * frame_pitch + (size_t)y * stride + (size_t)x * 3;
    const size_t   dof = (size_t)f * dst_frame_pitch + (size_t)y * dstride + x;
    if (aligned && x + 4 <= w) {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
        const uint32_t wd[3] = {s32[0], s32[1], s32[2]};
        // bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        int Y[4], Cr[4], Cb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i0 = 3 * k, i1 = 3 * k + 1, i2 = 3 * k + 2;
            const int B = (int)((wd[i0 >> 2] >> (8 * (i0 & 3))) & 255u);
            const int G = (int)((wd[i1 >> 2] >> (8 * (i1 & 3))) & 255u);
            const int R = (int)((wd[i2 >> 2] >> (8 * (i2 & 3))) & 255u);
            ycrcb_px(B, G, R, Y[k], Cr[k], Cb[k]);
        }
        *reinterpret_cast<uint32_t *>(yp + dof)  = Y[0] | (Y[1] << 8) | (Y[2] << 16) | (Y[3] << 24);
        *reinterpret_cast<uint32_t *>(crp + dof) = Cr[0] | (Cr[1] << 8) | (Cr[2] << 16) | (Cr[3] << 24);
        *reinterpret_cast<uint32_t *>(cbp + dof) = Cb[0] | (Cb[1] << 8) | (Cb[2] << 16) | (Cb[3] << 24);
    } else {
        for (int k = 0; k < 4 && x + k < w; ++k) {
            int Y, Cr, Cb;
            ycrcb_px(src[3 * k], src[3 * k + 1], src[3 * k + 2], Y, Cr, Cb);
            yp[dof + k] = (uint8_t)Y; crp[dof + k] = (uint8_t)Cr; cbp[dof + k] = (uint8_t)Cb;
        }
    }
}

void launch_bgr_to_ycrcb(hipStream_t s, const uint8_t *bgr, int w, int h, int64_t stride, int64_t frame_pitch,
                         int n_frames, uint8_t *y, uint8_t *cr, uint8_t *cb, int dstride, int64_t dst_frame_pitch)
{
    const int quads = (w + 3) / 4;
    const int aligned = ((reinterpret_cast<uintptr_t>(bgr) | (uintptr_t)stride | (uintptr_t)frame_pitch) % 4 == 0) &&
                        ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(cr) |
                          reinterpret_cast<uintptr_t>(cb) | (uintptr_t)dstride | (uintptr_t)dst_frame_pitch) % 4 == 0);
    dim3 grid((quads + 255) / 256, h, n_frames);
    hipLaunchKernelGGL(k_bgr_to_ycrcb, grid, dim3(256), 0, s, bgr, w, h, stride, frame_pitch, y, cr, cb, dstride,
                       dst_frame_pitch, aligned);
}

// NV12 ingest (build-defined, like the pyramid; SURVEY 8(f) row 3: "NV12 -> YCrCb directly, skipping BGR").  A decoder's frame is a
// full-resolution luma plane followed by one interleaved chroma plane at half resolution (Cb, Cr, Cb, Cr ...).  The three planes of
// the path are, by definition (oracle: ero_nv12_to_ycrcb):  Y = the luma byte;  Cr(x, y) = V(x / 2, y / 2);  Cb(x, y) = U(x / 2, y / 2)
// -- chroma replicated over its 2 x 2 block, no filter, no range conversion: the decoder's samples ARE the channel values.  Half the
// bytes of a BGR frame cross the host link.  One lane converts 4 pixels of a row: one luma dword, two chroma pairs.
__global__ __launch_bounds__(256) void k_nv12_to_ycrcb(const uint8_t *__restrict__ nv12, int w, int h, int64_t stride, int64_t frame_pitch,
                                                       uint8_t *__restrict__ yp, uint8_t *__restrict__ crp, uint8_t *__restrict__ cbp, int dstride,
                                                       int64_t dst_frame_pitch, int aligned)
{
    const int quad = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, f = blockIdx.z;
    const int x = quad * 4;
    if (x >= w) return;
    const uint8_t *ys = nv12 + (size_t)f * frame_pitch + (size_t)y * stride + x;
    const uint8_t *uv = nv12 + (size_t)f * frame_pitch + (size_t)h * stride + (size_t)(y >> 1) * stride + x;     // (x is even: pair x / 2 starts at byte x)
    const size_t   dof = (size_t)f * dst_frame_pitch + (size_t)y * dstride + x;
    if (aligned && x + 4 <= w) {
        const uint32_t yy = *reinterpret_cast<const uint32_t *>(ys), c = *reinterpret_cast<const uint32_t *>(uv);     // U0 V0 U1 V1
        const uint32_t u0 = c & 0xFFu, v0 = (c >> 8) & 0xFFu, u1 = (c >> 16) & 0xFFu, v1 = c >> 24;
        *reinterpret_cast<uint32_t *>(yp + dof) = yy;
        *reinterpret_cast<uint32_t *>(crp + dof) = v0 * 0x0101u | (v1 * 0x0101u) << 16;
        *reinterpret_cast<uint32_t *>(cbp + dof) = u0 * 0x0101u | (u1 * 0x0101u) << 16;
    } else {
        for (int k = 0; k < 4 && x + k < w; ++k) {
            yp[dof + k] = ys[k];
            cbp[dof + k] = uv[(k & ~1)];
            crp[dof + k] = uv[(k & ~1) + 1];
        }
    }
}

void launch_nv12_to_ycrcb(hipStream_t s, const uint8_t *nv12, int w, int h, int64_t stride, int64_t frame_pitch, int n_frames, uint8_t *y, uint8_t *cr,
                          uint8_t *cb, int dstride, int64_t dst_frame_pitch)
{
    const int quads = (w + 3) / 4;
    const int aligned = ((reinterpret_cast<uintptr_t>(nv12) | (uintptr_t)stride | (uintptr_t)frame_pitch) % 4 == 0) &&
                        ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(cr) | reinterpret_cast<uintptr_t>(cb) | (uintptr_t)dstride |
                          (uintptr_t)dst_frame_pitch) % 4 == 0);
    dim3 grid((quads + 255) / 256, h, n_frames);
    hipLaunchKernelGGL(k_nv12_to_ycrcb, grid, dim3(256), 0, s, nv12, w, h, stride, frame_pitch, y, cr, cb, dstride, dst_frame_pitch, aligned);
}

__global__ __launch_bounds__(256) void k_invert(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) dst[i] = (uint8_t)(255 - src[i]);
}

void launch_invert(hipStream_t s, const uint8_t *src, uint8_t *dst, size_t n)
{
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_invert, dim3(blocks ? blocks : 1), dim3(256), 0, s, src, dst, n);
}

// ------------------------------------------------------------------------------------
// cv::resize, INTER_LINEAR, 8UC1 (OpenCV 4.x semantics; see oracle/er_oracle.c for the
// statement this follows).  `inv` is xor-ed into every tap so an inverted channel is
// resized exactly like the materialised 255-x plane would be.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int resize_px(const ResizeGeom &g, const uint8_t *__restrict__ src, int sstride, int inv,
                                         int dx, int dy)
{
    if (g.mode == 0) return src[(size_t)dy * sstride + dx] ^ inv;
    if (g.mode == 1) {
        const uint8_t *r0 = src + (size_t)(2 * dy) * sstride + 2 * dx, *r1 = r0 + sstride;
        return ((r0[0] ^ inv) + (r0[1] ^ inv) + (r1[0] ^ inv) + (r1[1] ^ inv) + 2) >> 2;
    }
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
    const uint8_t *p0 = src + (size_t)y0 * sstride, *p1 = src + (size_t)y1 * sstride;
    const int r0 = (p0[sx] ^ inv) * a0 + (p0[sx1] ^ inv) * a1;
    const int r1 = (p1[sx] ^ inv) * a0 + (p1[sx1] ^ inv) * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    return min(max(v, 0), 255);
}

// Pyramid level: one wave per workgroup produces a 256 x 8 tile of the output.  Every lane owns 4 consecutive
// columns: their coefficients (the f64/f32 part of cv::resize's tables) are computed once and reused for the 8
// rows; one dword store per row.  The source window of the tile is first copied into LDS with coalesced dword
// loads -- byte gathers straight from global memory cost a texture-addresser pass per 4 lanes and bound the
// kernel -- and the taps are byte reads from LDS.  Windows that do not fit (large reductions, unaligned rows:
// only through str_er_resize_plane) take the taps from global memory.  The geometry is computed on the host.
constexpr int RESIZE_ROWS = 8;
constexpr int RS_WORDS = 96, RS_ROWS = 16;        // LDS window: 384 source bytes x 16 rows (a sqrt(2) step needs 364 x 14)

__device__ __forceinline__ int resize_sx(const ResizeGeom &g, int dx)
{
    const float fx = (float)((dx + 0.5) * g.scale_x - 0.5);
    return min(max((int)floorf(fx), 0), g.sw - 1);
}
__device__ __forceinline__ int resize_sy(const ResizeGeom &g, int dy)
{
    const float fy = (float)((dy + 0.5) * g.scale_y - 0.5);
    return (int)floorf(fy);
}

__global__ __launch_bounds__(64) void k_resize(const uint8_t *__restrict__ src, int sstride, int64_t splane_pitch,
                                               int64_t sframe_pitch, uint8_t *__restrict__ dst, int dstride,
                                               int64_t dplane_pitch, int64_t dframe_pitch, int planes_per_frame,
                                               ResizeGeom g)
{
    __shared__ uint32_t s_src[RS_ROWS * RS_WORDS + 2];      // (+2: a lane reads three dwords from its first tap on)
    const int tx0 = blockIdx.x * 256;
    const int dx0 = tx0 + (int)threadIdx.x * 4;
    const int dy0 = blockIdx.y * RESIZE_ROWS;
    const bool active = dx0 < g.dw;
    const int f = blockIdx.z / planes_per_frame, c = blockIdx.z % planes_per_frame;
    const uint8_t *s = src + (size_t)f * sframe_pitch + (size_t)c * splane_pitch;
    uint8_t       *d = dst + (size_t)f * dframe_pitch + (size_t)c * dplane_pitch;
    if (g.mode != 2) {      // copy / exact 2x2: no tables
        if (!active) return;
        for (int r = 0; r < RESIZE_ROWS && dy0 + r < g.dh; ++r)
            for (int k = 0; k < 4 && dx0 + k < g.dw; ++k)
                d[(size_t)(dy0 + r) * dstride + dx0 + k] = (uint8_t)resize_px(g, s, sstride, 0, dx0 + k, dy0 + r);
        return;
    }
    int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float fx = (float)((min(dx0 + k, g.dw - 1) + 0.5) * g.scale_x - 0.5);
        int   x = (int)floorf(fx);
        fx -= (float)x;
        if (x < 0) { fx = 0.f; x = 0; }
        if (x >= g.sw - 1) { fx = 0.f; x = g.sw - 1; }
        sx[k] = x; sx1[k] = (x + 1 < g.sw) ? x + 1 : x;
        a0[k] = __float2int_rn((1.f - fx) * 2048.f); a1[k] = __float2int_rn(fx * 2048.f);
    }
    // source window of the tile (uniform over the wave)
    const int x_lo = resize_sx(g, tx0) & ~3;
    const int x_last = resize_sx(g, min(tx0 + 255, g.dw - 1));
    const int x_hi = (x_last + 1 < g.sw) ? x_last + 1 : x_last;
    const int y_lo = min(max(resize_sy(g, dy0), 0), g.sh - 1);
    const int y_hi = min(max(resize_sy(g, min(dy0 + RESIZE_ROWS - 1, g.dh - 1)) + 1, 0), g.sh - 1);
    const int nwords = (x_hi - x_lo) / 4 + 1, nrows = y_hi - y_lo + 1;
    const bool staged = nwords <= RS_WORDS && nrows <= RS_ROWS && (sstride & 3) == 0 && (reinterpret_cast<uintptr_t>(s) & 3) == 0;
    if (staged) {
        // a lane fetches words lane and lane + 64 of every row: all loads of the window (up to 32 per lane) are issued before the first one is
        // waited for -- a loop of load / wait / write pays the memory latency once per round, and that, not arithmetic, was the kernel's time
        static_assert(RS_WORDS <= 128, "two words per lane and row");
        uint32_t v[RS_ROWS][2];
        const uint8_t *src0 = s + (size_t)y_lo * sstride + x_lo + 4 * (int)threadIdx.x;
#pragma unroll
        for (int r = 0; r < RS_ROWS; ++r) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                v[r][j] = 0;
                if (r < nrows && (int)threadIdx.x + 64 * j < nwords) v[r][j] = *reinterpret_cast<const uint32_t *>(src0 + (size_t)r * sstride + 256 * j);
            }
        }
#pragma unroll
        for (int r = 0; r < RS_ROWS; ++r) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (r < nrows && (int)threadIdx.x + 64 * j < nwords && (int)threadIdx.x + 64 * j < RS_WORDS) s_src[r * RS_WORDS + threadIdx.x + 64 * j] = v[r][j];
        }
        __syncthreads();
    }
    if (!active) return;
    const uint8_t *lds = reinterpret_cast<const uint8_t *>(s_src);
    const bool full = dx0 + 4 <= g.dw && (dstride & 3) == 0;
    if (staged && g.scale_x <= 1.5) {
        // The taps of the lane's 4 columns lie within 7 source bytes (reduction <= 1.5): per SOURCE row the lane reads the three dwords
        // that hold them, shifts them to its first tap (two v_alignbyte) and picks the 4 left and the 4 right taps with two byte
        // permutes whose selectors are fixed for the tile; the horizontal sums of a source row are kept for the next output row, which
        // mostly needs it again.  A third of the LDS reads of the form below (the byte reads bound this kernel: a byte read costs the
        // LDS what a dword read costs), same arithmetic, same result.
        const int      base = sx[0] & ~3, s0 = sx[0] - base;
        const uint32_t selL = (uint32_t)(sx[0] - sx[0]) | (uint32_t)(sx[1] - sx[0]) << 8 | (uint32_t)(sx[2] - sx[0]) << 16 | (uint32_t)(sx[3] - sx[0]) << 24;
        const uint32_t selR = (uint32_t)(sx1[0] - sx[0]) | (uint32_t)(sx1[1] - sx[0]) << 8 | (uint32_t)(sx1[2] - sx[0]) << 16 | (uint32_t)(sx1[3] - sx[0]) << 24;
        const uint32_t *col = s_src + (base - x_lo) / 4;
        auto hrow = [&](int y, int (&h)[4]) {           // horizontal pass of source row y for the lane's 4 columns
            const uint32_t *p = col + (y - y_lo) * RS_WORDS;
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)s0), hi = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)s0);
            const uint32_t L = __builtin_amdgcn_perm(hi, lo, selL), R = __builtin_amdgcn_perm(hi, lo, selR);
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = (int)(__umul24((L >> (8 * k)) & 0xFFu, (uint32_t)a0[k]) + __umul24((R >> (8 * k)) & 0xFFu, (uint32_t)a1[k]));
        };
        int ca = -1, cb = -1;               // source rows whose sums are in hA / hB (rows are >= 0)
        int hA[4] = {0, 0, 0, 0}, hB[4] = {0, 0, 0, 0};
#pragma unroll 1
        for (int r = 0; r < RESIZE_ROWS; ++r) {
            const int dy = dy0 + r;
            if (dy >= g.dh) break;
            float fy = (float)((dy + 0.5) * g.scale_y - 0.5);
            int   sy = (int)floorf(fy);
            fy -= (float)sy;
            const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
            const int ya = min(max(sy, 0), g.sh - 1), yb = min(max(sy + 1, 0), g.sh - 1);
            // (ya, yb are the same for every lane: uniform branches)
            if (ya == cb) {
#pragma unroll
                for (int k = 0; k < 4; ++k) hA[k] = hB[k];
                ca = cb;
            } else if (ya != ca) { hrow(ya, hA); ca = ya; }
            if (yb == ca) {
#pragma unroll
                for (int k = 0; k < 4; ++k) hB[k] = hA[k];
                cb = yb;
            } else if (yb != cb) { hrow(yb, hB); cb = yb; }
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = min((int)(((__umul24((uint32_t)b0, (uint32_t)hA[k] >> 4) >> 16) + (__umul24((uint32_t)b1, (uint32_t)hB[k] >> 4) >> 16) + 2u) >> 2), 255);
                v |= (uint32_t)o << (8 * k);
            }
            uint8_t *o = d + (size_t)dy * dstride + dx0;
            if (full) *reinterpret_cast<uint32_t *>(o) = v;
            else for (int k = 0; k < 4 && dx0 + k < g.dw; ++k) o[k] = (uint8_t)(v >> (8 * k));
        }
        return;
    }
#pragma unroll 4
    for (int r = 0; r < RESIZE_ROWS; ++r) {
        const int dy = min(dy0 + r, g.dh - 1);
        const bool live = dy0 + r < g.dh;
        float fy = (float)((dy + 0.5) * g.scale_y - 0.5);
        int   sy = (int)floorf(fy);
        fy -= (float)sy;
        const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
        const int ya = min(max(sy, 0), g.sh - 1), yb = min(max(sy + 1, 0), g.sh - 1);
        uint32_t v = 0;
        if (staged) {
            const uint8_t *p0 = lds + (ya - y_lo) * (RS_WORDS * 4) - x_lo, *p1 = lds + (yb - y_lo) * (RS_WORDS * 4) - x_lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t r0 = __umul24(p0[sx[k]], (uint32_t)a0[k]) + __umul24(p0[sx1[k]], (uint32_t)a1[k]);
                const uint32_t r1 = __umul24(p1[sx[k]], (uint32_t)a0[k]) + __umul24(p1[sx1[k]], (uint32_t)a1[k]);
                const int o = min((int)(((__umul24((uint32_t)b0, r0 >> 4) >> 16) + (__umul24((uint32_t)b1, r1 >> 4) >> 16) + 2u) >> 2), 255);
                v |= (uint32_t)o << (8 * k);
            }
        } else {
            const uint8_t *p0 = s + (size_t)ya * sstride, *p1 = s + (size_t)yb * sstride;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t r0 = __umul24(p0[sx[k]], (uint32_t)a0[k]) + __umul24(p0[sx1[k]], (uint32_t)a1[k]);
                const uint32_t r1 = __umul24(p1[sx[k]], (uint32_t)a0[k]) + __umul24(p1[sx1[k]], (uint32_t)a1[k]);
                const int o = min((int)(((__umul24((uint32_t)b0, r0 >> 4) >> 16) + (__umul24((uint32_t)b1, r1 >> 4) >> 16) + 2u) >> 2), 255);
                v |= (uint32_t)o << (8 * k);
            }
        }
        uint8_t *o = d + (size_t)dy * dstride + dx0;
        if (!live) continue;
        if (full) *reinterpret_cast<uint32_t *>(o) = v;
        else for (int k = 0; k < 4 && dx0 + k < g.dw; ++k) o[k] = (uint8_t)(v >> (8 * k));
    }
}

static ResizeGeom host_resize_geom(int sw, int sh, int dw, int dh)
{
    ResizeGeom g;
    g.sw = sw; g.sh = sh; g.dw = dw; g.dh = dh;
    g.scale_x = 1.0 / ((double)dw / sw);
    g.scale_y = 1.0 / ((double)dh / sh);
    if (dw == sw && dh == sh) { g.mode = 0; return g; }
    const int isx = (int)rint(g.scale_x), isy = (int)rint(g.scale_y);
    const bool fast = fabs(g.scale_x - isx) < DBL_EPSILON && fabs(g.scale_y - isy) < DBL_EPSILON;
    g.mode = (fast && isx == 2 && isy == 2) ? 1 : 2;
    return g;
}

void launch_resize(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, int64_t splane_pitch,
                   int64_t sframe_pitch, uint8_t *dst, int dw, int dh, int dstride, int64_t dplane_pitch,
                   int64_t dframe_pitch, int planes_per_frame, int n_frames)
{
    const int quads = (dw + 3) / 4;
    dim3 grid((quads + 63) / 64, (dh + RESIZE_ROWS - 1) / RESIZE_ROWS, planes_per_frame * n_frames);
    hipLaunchKernelGGL(k_resize, grid, dim3(64), 0, s, src, sstride, splane_pitch, sframe_pitch, dst, dstride,
                       dplane_pitch, dframe_pitch, planes_per_frame, host_resize_geom(sw, sh, dw, dh));
}

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
// (Measured, 32 frames text / noise, and not adopted.  Round 3: a static contiguous deal per lane -- round 2's form, 2.62 against 2.27 once the
// list was one; the k-th batch of 64 entries spread over the wave's whole share: no difference; the two ends carried as KEYS, (level << 16) |
// slot -- one compare orders them, the CAS value is the other key, a lane without an edge holds two equal keys and runs the pass as a
// no-op: 19 vector instructions per pass instead of 34 in the listing, but the compiler's loop has more branches: vector -0.8 %, scalar
// +6.7 %, branch +23 % per wave, 2.31 / 5.59 against 2.25 / 5.38 (masked idle lanes: 2.34 / 5.71) -- the kernel's time follows the TOTAL
// number of instructions its waves issue, of whatever kind; handing out only when 8 / 16 / 24 lanes are idle: no difference.
// Round 2: a workgroup-wide cursor that hands the next entries to whichever lanes are idle -- round 1's form -- 3.19 / 9.8 ms against
// 3.11 / 8.5: the ballots and the LDS atomic per refill cost more than the balance gains;
// lane i takes entries i, i + 256, ...: 3.28 / 10.2; both finds of a pass in one loop so that their loads are in flight together:
// 3.62 / 11.5, the loop runs as long as the longer chain with both halves' instructions; walking up a's chain to b's level in a loop of
// finds inside the pass: 3.67 / 12.3; the vertical round first: 3.21 / 10.8; three rounds with the vertical edges between equal levels first
// (plain unions while no node has a parent): 3.27 / 9.4 against 2.98 / 8.0; runs that start right below a pixel of their own level linked
// upwards in the load phase (a fifth of the vertical edges gone, but long same-level chains): 3.03 / 8.8 against 2.99 / 8.1; a lost CAS judged
// again on the spot with the word it returned instead of in the next pass: 3.22 / 9.9 against 2.98 / 8.05; every wave joining the horizontal
// and inner vertical edges of its own 8-row band by itself (wave-wide scans, no workgroup barriers, nobody else on its words), the three row
// pairs between bands in a workgroup-wide round afterwards: 3.22 / 8.4 against 2.99 / 8.07 -- the bands' edge counts differ; round 1's attempts -- fewer waves in the loop, two edges
// per lane in flight, a level-ordered form with a barrier per level, one combined round -- all lost as well.)
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
// (Measured and not adopted: one compacted list of all pieces of the tile, processed by all lanes evenly -- fewer instructions, but the
// two extra barriers and the dependent LDS reads of the list cost what they save; round 3: the parent word of a lane's NEXT piece fetched
// while the current one is walked / added up, in the flatten and the statistics loops -- 2.11 against 2.08 ms: the four instructions per
// round cost more than the latency they hide; a launch of resident workgroups that walk through the tiles -- the next tile's pixels requested
// a tile ahead, tiles handed out by a batch-wide cursor read two tiles ahead: 2.33 against 1.93 ms; 2 / 4 / 8 / 16 tiles per workgroup in a
// plain launch: 2.05 / 2.11 / 2.23 / 2.44 -- the loop costs ~150 instructions per wave and tile (descriptors, spills, a barrier), HBM latency
// was hidden by the seven other workgroups of the CU all along, and the hardware's dispatcher balances better than a cursor; a short cut for
// UNIFORM tiles (one level, no wall: a sixth of the chroma tiles of text-like and of natural frames), whose single record can be written down after
// the load phase: 1.899 against 1.903 ms -- such a tile was cheap already, and every other tile pays for the test; one barrier less between the
// ids and the statistics (zeroes earlier, the list's parent ids behind the statistics loop): no difference.)
// Per channel (32 frames, pyr3x8): luma 0.98 ms, Cr 0.50, Cb 0.50 -- a chroma tile has a sixth of a luma tile's nodes and costs half: what a
// tile costs is mostly what EVERY tile costs (load phase 22 % of a chroma tile, building the edge list, the reductions of the statistics pass).
// ------------------------------------------------------------------------------------
#define LEVK(k) ((((k) < 4 ? lev_lo : lev_hi) >> (8 * ((k) & 3))) & 0xFFu)
template <int FOLD_CAP>
__global__ __launch_bounds__(TILE_THREADS, (FOLD_CAP == FOLD_CAP_SPARSE ? 8 : 6)) void k_tile_tree(BatchDev b, DetectParams prm)
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
#ifdef STR_ER_PAD_LDS
        uint32_t pad[STR_ER_PAD_LDS / 4];     // developer aid: extra LDS per workgroup, to see what fewer resident workgroups per CU cost
#endif
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
#ifdef STR_ER_PAD_LDS
    if (b.n_tiles == 0xFFFFFFFFu) s_lds.pad[tid] = tid;
#endif
    const int       pi = b.tile_plane[blockIdx.x];
    const PlaneDesc pd = b.planes[pi];
    const uint32_t  tl = blockIdx.x - pd.tile_base;
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
        b.tile_nbase[blockIdx.x] = nbase;
        b.tile_nrec[blockIdx.x] = (uint16_t)total;
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

void launch_tile_tree(hipStream_t s, const BatchDev &b, const DetectParams &p, bool sparse)
{
    if (!b.n_tiles) return;
    if (sparse) hipLaunchKernelGGL(k_tile_tree<FOLD_CAP_SPARSE>, dim3(b.n_tiles), dim3(TILE_THREADS), 0, s, b, p);
    else        hipLaunchKernelGGL(k_tile_tree<FOLD_CAP_DENSE>, dim3(b.n_tiles), dim3(TILE_THREADS), 0, s, b, p);
}

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
__global__ __launch_bounds__(GROUP_THREADS) void k_group_merge(BatchDev b)
{
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
    const int       pi = b.group_plane[blockIdx.x];
    const PlaneDesc pd = b.planes[pi];
    const int       groups_x = (pd.tiles_x + GX - 1) / GX;
    const uint32_t  gl = blockIdx.x - pd.group_base;
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
    if (N == NONE || N == 0) return;
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
    if (tid == 0) b.group_done[blockIdx.x] = 1;
    GM_MARK(6);
}

// variant: LDS for 512 / 1024 / 2048 / 3072 records per group, 256 / 512 / 1024 lanes
void launch_group_merge(hipStream_t s, const BatchDev &b, int variant)
{
    if (!b.n_groups || b.group_x <= 0 || b.group_y <= 0 || b.group_x * b.group_y > GROUP_MAX_TILES) return;
    switch (variant) {
    case 0: hipLaunchKernelGGL((k_group_merge<512, 256>), dim3(b.n_groups), dim3(256), 0, s, b); break;
    case 1: hipLaunchKernelGGL((k_group_merge<1024, 256>), dim3(b.n_groups), dim3(256), 0, s, b); break;
    case 2: hipLaunchKernelGGL((k_group_merge<1024, 512>), dim3(b.n_groups), dim3(512), 0, s, b); break;
    case 3: hipLaunchKernelGGL((k_group_merge<2048, 512>), dim3(b.n_groups), dim3(512), 0, s, b); break;
    case 4: hipLaunchKernelGGL((k_group_merge<2048, 1024>), dim3(b.n_groups), dim3(1024), 0, s, b); break;
    case 5: hipLaunchKernelGGL((k_group_merge<3072, 1024>), dim3(b.n_groups), dim3(1024), 0, s, b); break;
    // 2528 records: 32 B each + the tile tables = 64 of the 1280-byte LDS granules, so TWO workgroups fit a CU
    case 6: hipLaunchKernelGGL((k_group_merge<2528, 1024>), dim3(b.n_groups), dim3(1024), 0, s, b); break;
    default: hipLaunchKernelGGL((k_group_merge<2528, 512>), dim3(b.n_groups), dim3(512), 0, s, b); break;
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
        // (a seam inside a group of tiles that k_group_merge has put together is no seam any more)
        const uint32_t GX = (uint32_t)b.group_x, GY = (uint32_t)b.group_y, groups_x = GX ? ((uint32_t)pd.tiles_x + GX - 1u) / GX : 0u;
        bool inner = false;
        if (i < pd.n_hpairs) {
            const uint32_t j = i / pd.w, x = i - j * pd.w;
            ta = j * pd.tiles_x + x / (uint32_t)TILE_W; tb = ta + pd.tiles_x;
            if (GX && (j + 1u) % GY != 0u) inner = b.group_done[pd.group_base + (j / GY) * groups_x + (x / (uint32_t)TILE_W) / GX] != 0;
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
            if (GX && (k + 1u) % GX != 0u) inner = b.group_done[pd.group_base + ((y / (uint32_t)TILE_H) / GY) * groups_x + k / GX] != 0;
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
    node_connect(b.na.rec + pd.node_base, s_pa[threadIdx.x], s_pb[threadIdx.x]);
}

void launch_seam(hipStream_t s, const BatchDev &b, bool xcd_affine)
{
    if (!b.n_seam_blocks) return;
    hipLaunchKernelGGL(k_seam, dim3((b.n_seam_blocks + 7u) / 8u * 8u), dim3(SEAM_BLOCK), 0, s, b, xcd_affine ? 1 : 0);
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
            uint32_t tortoise = i, hare = PAR_ID(w), power = 1, lam = 1;
            for (;;) {
                if (hare == tortoise) { ok = false; break; }
                const uint32_t wh = LD_AGENT(&rec[hare].par);
                if (wh == NONE || PAR_ID(wh) >= n) break;
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

// Nodes that were unified into another node of the same level hand their own statistics to the surviving level root;
// surviving nodes get a canonical parent (the parent node's level root).  Every node that will push its totals to a parent
// -- open, alive, not a tree root -- is counted in the parent's dependency counter (aux).
__global__ __launch_bounds__(256) void k_resolve(BatchDev b)
{
    const int       pi = b.nb_plane[blockIdx.x];
    const uint32_t  bi = blockIdx.x - b.planes[pi].nb_base, nbp = b.planes[pi].nb_count;     // this plane's workgroups: bi of nbp
    const uint32_t  n = plane_nodes(b, pi);
    NodeRec        *nr = b.na.rec + b.planes[pi].node_base;
    uint32_t       *aux = b.na.aux + b.planes[pi].node_base;
    uint32_t       *arr = b.na.arr + b.planes[pi].node_base;
    const int       lane = threadIdx.x & 63;
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
        // The pieces a seam cut a big node into all hand over to ONE survivor, and the children of a big node all count into ONE
        // parent: the lanes of a wave that share a target combine first (ballot + butterfly) -- one set of atomics per distinct
        // target and wave; without it these few hot words serialise the whole kernel.
        unsigned long long todo = __ballot(hand_to != NONE);
        while (todo) {
            const int      leader = __ffsll((long long)todo) - 1;
            const uint32_t lr = __shfl(hand_to, leader);
            const bool     mine = hand_to == lr;
            const unsigned long long m = __ballot(mine);
            if (__popcll(m) == 1) {
                if (mine) {
                    atomicAdd(&nr[lr].cnt, c);
                    if (nd) atomicAdd(&nr[lr].nod, nd);
                    atomicMin(&nr[lr].x0, bx0); atomicMin(&nr[lr].y0, by0); atomicMax(&nr[lr].x1, bx1); atomicMax(&nr[lr].y1, by1);
                    atomicMin(&nr[lr].key, ky);          // same level: the top byte is equal, the minimum is over the pixel index
                }
            } else {
                const uint32_t sc = wave_sum(mine ? c : 0u), sn = wave_sum(mine ? nd : 0u);
                const uint32_t mx0 = wave_min(mine ? bx0 : 0xFFFFFFFFu), my0 = wave_min(mine ? by0 : 0xFFFFFFFFu);
                const uint32_t mx1 = wave_max(mine ? bx1 : 0u), my1 = wave_max(mine ? by1 : 0u), mk = wave_min(mine ? ky : 0xFFFFFFFFu);
                if (lane == leader) {
                    atomicAdd(&nr[lr].cnt, sc);
                    if (sn) atomicAdd(&nr[lr].nod, sn);
                    atomicMin(&nr[lr].x0, mx0); atomicMin(&nr[lr].y0, my0); atomicMax(&nr[lr].x1, mx1); atomicMax(&nr[lr].y1, my1);
                    atomicMin(&nr[lr].key, mk);
                }
            }
            todo &= ~m;
        }
        todo = __ballot(push_to != NONE);
        while (todo) {
            const int      leader = __ffsll((long long)todo) - 1;
            const uint32_t lq = __shfl(push_to, leader);
            const unsigned long long m = __ballot(push_to == lq);
            if (lane == leader) atomicAdd(&aux[lq], (uint32_t)__popcll(m));
            todo &= ~m;
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
    const int       pi = b.nb_plane[blockIdx.x];
    const uint32_t  bi = blockIdx.x - b.planes[pi].nb_base, nbp = b.planes[pi].nb_count;
    const uint32_t  n = plane_nodes(b, pi);
    NodeRec        *nr = b.na.rec + b.planes[pi].node_base;
    const uint32_t *aux = b.na.aux + b.planes[pi].node_base;       // (constant here: plain loads)
    uint32_t       *arr = b.na.arr + b.planes[pi].node_base;
    const int       lane = threadIdx.x & 63;
    for (uint32_t x0 = bi * blockDim.x + (threadIdx.x & ~63u); x0 < n; x0 += nbp * blockDim.x) {
        const uint32_t x = x0 + lane;
        bool     act = false;
        uint32_t q = NONE, c = 0, nd = 0, bx0 = 0xFFFFFFFFu, by0 = 0xFFFFFFFFu, bx1 = 0, by1 = 0;
        if (x < n) {
            const uint32_t w = nr[x].par, f = nr[x].nod;          // parent and flags are final since k_resolve
            // (a node nobody pushes into is ready; the others are taken by their last child)
            act = w != NONE && !(f & (NODE_DEAD | NODE_CLOSED)) && aux[x] == 0u;
            if (act) {
                q = PAR_ID(w);
                node_totals(nr + x, c, nd, bx0, by0, bx1, by1);
                nd &= NODE_CNT;
            }
        }
        // first step: lanes of the wave that share a parent combine (ballot + butterfly): one set of atomics and one
        // decrement per distinct parent and wave -- the background node of a tile has hundreds of such children
        bool cont = false;
        unsigned long long todo = __ballot(act);
        while (todo) {
            const int      leader = __ffsll((long long)todo) - 1;
            const uint32_t lq = __shfl(q, leader);
            const bool     mine = act && q == lq;
            const unsigned long long m = __ballot(mine);
            const uint32_t k = (uint32_t)__popcll(m);
            if (k == 1) {
                if (mine) { const uint32_t ex = aux[lq]; cont = node_arrive(&arr[lq], ex, 1u, node_push(nr + lq, c, nd, bx0, by0, bx1, by1)); }
            } else {
                const uint32_t sc = wave_sum(mine ? c : 0u), sn = wave_sum(mine ? nd : 0u);
                const uint32_t mx0 = wave_min(mine ? bx0 : 0xFFFFFFFFu), my0 = wave_min(mine ? by0 : 0xFFFFFFFFu);
                const uint32_t mx1 = wave_max(mine ? bx1 : 0u), my1 = wave_max(mine ? by1 : 0u);
                if (lane == leader) { const uint32_t ex = aux[lq]; cont = node_arrive(&arr[lq], ex, k, node_push(nr + lq, sc, sn, mx0, my0, mx1, my1)); }
            }
            todo &= ~m;
        }
        // the lanes that now own a parent carry it upward
        uint32_t g = q;
        while (cont) {
            const uint32_t w = nr[g].par;                                  // (final since k_resolve, like the flags)
            uint32_t gc, gf, gx0, gy0, gx1, gy1;
            node_totals(nr + g, gc, gf, gx0, gy0, gx1, gy1);
            if (w == NONE || (gf & (NODE_DEAD | NODE_CLOSED))) break;       // a tree root (or a node that never pushes)
            const uint32_t p = PAR_ID(w);
            const uint32_t ex = aux[p];
            cont = node_arrive(&arr[p], ex, 1u, node_push(nr + p, gc, gf & NODE_CNT, gx0, gy0, gx1, gy1));
            g = p;
        }
    }
}

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
    if (in_lds && tid < 256) s_lcur[tid] = 0;
    if (in_lds) __syncthreads();
    for (uint32_t i = tid; i < K; i += NMS_THREADS) {
        kstart[i] = i; kncand[i] = 0; kbest[i] = ~0ull;
        if (in_lds) { s_npar[i] = b.ka.parent[kb + i]; s_nsa[i] = s_narea[i] = (int)kbox[4 * i + 2] * (int)kbox[4 * i + 3]; atomicAdd(&s_lcur[klev[i]], 1u); }
        atomicOr(&s_levels[klev[i] >> 5], 1u << (klev[i] & 31));
    }
    __syncthreads();
    if (in_lds) {
        if (tid < 64) {              // counts -> first positions: four levels a lane, a scan over the wave
            const uint32_t c0 = s_lcur[4 * tid], c1 = s_lcur[4 * tid + 1], c2 = s_lcur[4 * tid + 2], c3 = s_lcur[4 * tid + 3], tot = c0 + c1 + c2 + c3;
            uint32_t incl = tot;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if (tid >= d) incl += o; }
            const uint32_t e = incl - tot;
            s_lcur[4 * tid] = e; s_lcur[4 * tid + 1] = e + c0; s_lcur[4 * tid + 2] = e + c0 + c1; s_lcur[4 * tid + 3] = e + c0 + c1 + c2;
        }
        __syncthreads();
        for (uint32_t i = tid; i < K; i += NMS_THREADS) s_perm[atomicAdd(&s_lcur[klev[i]], 1u)] = (uint16_t)i;
        __syncthreads();
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
        for (uint32_t i = tid; i < K; i += NMS_THREADS) {
            if (klev[i] != t) continue;
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
    } else
    for (uint32_t X = tid; X < K; X += NMS_THREADS) {
        if (kstart[X] != X) continue;
        int      len = 1;
        uint32_t p = X;
        while (p != root && kstart[kpar[p]] == X) { p = (uint32_t)kpar[p]; ++len; }
        if (len < 1 + T) continue;
        uint32_t trail = X, lead = X;
        for (int i = 0; i < T; ++i) lead = (uint32_t)kpar[lead];
        uint32_t best = X;
        double   best_st = 0;
        int      best_a = 0;
        for (int i = 0; i < len - T; ++i) {
            const int a = barea(trail);
            const int bb = barea(lead);
            const double st = (double)a / (double)(bb - a);   // 0 denominator -> +inf, as in the reference
            if (i == 0 || st > best_st) { best = trail; best_st = st; best_a = a; }
            else if (st == best_st && a < best_a) { best = trail; best_a = a; }
            trail = (uint32_t)kpar[trail];
            lead = (uint32_t)kpar[lead];
        }
        const int    bw = kbox[4 * best + 2], bh = kbox[4 * best + 3];
        const double ar = (double)bw / (double)bh;
        const int    area = (int)b.ka.area[kb + best];
        if (ar < 2.0 && ar > 0.10 && area < prm.max_area && area > prm.min_area && bh < pd.h * 0.8 && bw < pd.w * 0.8) {
            const uint32_t slot = atomicAdd(&s_npool, 1u);
            if (slot < pd.pool_cap) b.pool_tmp[pb + slot] = best;
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

void launch_cand_reprefix(hipStream_t s, const BatchDev &b, const CandRec *from, uint32_t *redo, uint32_t *n_redo)
{
    hipLaunchKernelGGL(k_cand_reprefix, dim3(1), dim3(1024), 0, s, b, redo, n_redo);
    hipLaunchKernelGGL(k_cand_move, dim3(256), dim3(256), 0, s, b, from);
}

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
    for (uint32_t c0 = blockIdx.x * CLS_PER_BLOCK; c0 < total; c0 += gridDim.x * CLS_PER_BLOCK) {
        CLS_MARK(0);
#ifdef STR_ER_PHASE_PROF
        const unsigned long long tp0 = wall_clock64();
#endif
        // ---- phase 1: histograms --------------------------------------------------------------
        // Every wave works on an ER of its own with scratch of its own (hist[wv], tile[wv], its row): between its steps it only has to wait for
        // ITSELF -- a wave's LDS operations are carried out in the order it issues them -- so the steps are separated by a wave-level fence, not
        // by a workgroup barrier: the 16 waves' ERs differ in size, and a barrier per step had every wave wait for the largest four times per ER.
        if (run_cascades) {
            for (int it = 0; it < CLS_PER_BLOCK / CLS64_WAVES; ++it) {
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
            const bool     ok = cpos < total;
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
                     int run_cascades, const uint32_t *list, const uint32_t *n_list)
{
    hipLaunchKernelGGL(k_classify, dim3(list ? 64 : 1024), dim3(CLS64_THREADS), 0, s, b, p, strong, weak, run_cascades, list, n_list);
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

} // namespace str_er
