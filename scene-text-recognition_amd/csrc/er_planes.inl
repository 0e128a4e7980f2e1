// er_planes.inl -- part of er_kernels.hip (included there, inside namespace str_er; not a translation unit of its own): compute_channels, NV12 ingest, cv::resize (pyramid levels).
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
    // the rows' coefficients are the same for every lane: lane r works out those of row r (the f64 / f32 part of cv::resize's tables) once -- all 64 lanes,
    // before the lanes beyond the plane's last column leave -- and the row loop reads them into scalar registers: as every lane computing every row's they
    // were a sixth of the kernel's vector instructions
    int row_sy, row_b0, row_b1;
    {
        const int dy = dy0 + (int)(threadIdx.x & (RESIZE_ROWS - 1));
        float fy = (float)((dy + 0.5) * g.scale_y - 0.5);
        row_sy = (int)floorf(fy);
        fy -= (float)row_sy;
        row_b0 = __float2int_rn((1.f - fy) * 2048.f); row_b1 = __float2int_rn(fy * 2048.f);
    }
    static_assert((RESIZE_ROWS & (RESIZE_ROWS - 1)) == 0 && RESIZE_ROWS <= 64, "a lane per row of the tile");
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
    const uint8_t *lds = reinterpret_cast<const uint8_t *>(s_src);
    const bool full = dx0 + 4 <= g.dw && (dstride & 3) == 0;
    // (the lanes beyond the plane's last column stay for the form below -- their stores are masked, their taps clamped into the window: the row loop reads
    // the coefficients of row r from lane r, and a lane that had left could not be relied on to hold them)
    if (!(staged && g.scale_x <= 1.5) && !active) return;
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
            const int sy = __builtin_amdgcn_readlane(row_sy, r), b0 = __builtin_amdgcn_readlane(row_b0, r), b1 = __builtin_amdgcn_readlane(row_b1, r);
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
