// er_tile_tree2.inl -- part of er_kernels.hip (included there, inside namespace str_er): k_tile_tree2, the level-by-level / bit-mask form of the
// tile kernel for planes with few levels and few nodes per tile (the chroma planes), and the launchers that split a batch's tiles between it
// and k_tile_tree.  The algorithm is tile2_body.h (one source for the device and for the host check, tests/cpp/tile2_model_check.cpp); this
// file is its execution policy on gfx950: a "vector" is a per-lane scalar, a tile row's neighbour rows come by DPP wave shifts, the sums /
// ORs / minima over a tile's 32 rows are four fused DPP steps inside the 16-lane rows and one swizzle across them.
// Semantics as k_tile_tree: /root/reference/src/ER.cpp:131-191, 240-413.
// (tile2_body.h is included by er_kernels.hip, outside the namespace)

namespace t2 {

struct DevWave {
    typedef uint32_t u32;
    typedef uint64_t u64;
    typedef bool     mask;
    uint32_t *lds_rec;      // [2][REC_CAP][8]: the records of the wave's two tiles until they are written out
    uint8_t  *lds_idmap;    // [2][2048]: key pixel of an exported node -> its record

    static __device__ __forceinline__ void mark(int) {}
    static __device__ __forceinline__ void stat(int, int) {}
    __device__ __forceinline__ u32 lane() const { return threadIdx.x & 63u; }
    static __device__ __forceinline__ u32 bc(uint32_t s) { return s; }
    static __device__ __forceinline__ u64 bc64(uint64_t s) { return s; }
    static __device__ __forceinline__ mask all() { return true; }
    static __device__ __forceinline__ mask none() { return false; }
    static __device__ __forceinline__ bool any(mask m) { return __builtin_amdgcn_ballot_w64(m) != 0ull; }
    static __device__ __forceinline__ uint64_t ballot(mask m) { return __builtin_amdgcn_ballot_w64(m); }
    static __device__ __forceinline__ mask lanes_of(uint64_t bits) { return __builtin_amdgcn_inverse_ballot_w64(bits); }
    static __device__ __forceinline__ u32 sel(mask m, u32 a, u32 b) { return m ? a : b; }
    static __device__ __forceinline__ u64 sel64(mask m, u64 a, u64 b) { return m ? a : b; }
    static __device__ __forceinline__ u32 sel_half(mask isB, uint32_t vb, uint32_t va) { return isB ? vb : va; }
    static __device__ __forceinline__ u32 and_or(u32 a, uint32_t m, u32 c) { return (a & m) | c; }
    static __device__ __forceinline__ u64 and_or64(u64 a, u64 m, u64 c) { return (a & m) | c; }
    static __device__ __forceinline__ u32 lshl_or(u32 a, int s, u32 c) { return (a << s) | c; }
    static __device__ __forceinline__ u32 bfe(u32 a, int off, int wd) { return __builtin_amdgcn_ubfe(a, (uint32_t)off, (uint32_t)wd); }
    static __device__ __forceinline__ u64 bfi64(u64 x, u64 a, u64 b) { return (x & a) | (~x & b); }
    static __device__ __forceinline__ u64 brev64(u64 a) { return __builtin_bitreverse64(a); }
    static __device__ __forceinline__ u64 mk64(u32 lo, u32 hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }
    static __device__ __forceinline__ u32 lo(u64 a) { return (uint32_t)a; }
    static __device__ __forceinline__ u32 hi(u64 a) { return (uint32_t)(a >> 32); }
    static __device__ __forceinline__ u64 shl64(u64 a, u32 s) { return a << (s & 63u); }
    static __device__ __forceinline__ u32 popc64(u64 a) { return (uint32_t)__builtin_popcountll(a); }
    // (of a word that is not 0)
    static __device__ __forceinline__ u32 ffs64(u64 a) { return (uint32_t)(__ffsll((unsigned long long)a) - 1); }
    static __device__ __forceinline__ u32 fls64(u64 a) { return 63u - (uint32_t)__clzll((long long)a); }
    static __device__ __forceinline__ u32 ffs32(u32 a) { return (uint32_t)(__ffs((int)a) - 1); }
    static __device__ __forceinline__ u32 fls32(u32 a) { return 31u - (uint32_t)__clz((int)a); }
    static __device__ __forceinline__ u32 min_u(u32 a, uint32_t b) { return a < b ? a : b; }
    static __device__ __forceinline__ u32 max_i(u32 a, int b) { return (uint32_t)((int)a > b ? (int)a : b); }
    static __device__ __forceinline__ mask gt_i64(u32 a, int32_t b) { return (int64_t)a > (int64_t)b; }
    // acc | 1 << (byte & 31) for the four bytes of q: the shift count is taken from the byte by the instruction (SDWA)
    static __device__ __forceinline__ u32 onehot4_or(u32 q, u32 acc)
    {
        uint32_t a0, a1, a2, a3;
        const uint32_t one = 1u;
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(a0) : "v"(q), "v"(one));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a1) : "v"(q), "v"(one));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(a2) : "v"(q), "v"(one));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(a3) : "v"(q), "v"(one));
        return (acc | a0 | a1) | (a2 | a3);
    }
    // the row above / below: the value of lane - 1 / lane + 1 (DPP wave shift; a tile's first / last row gets the other tile's row or 0: the callers mask)
    static __device__ __forceinline__ u64 row_above(u64 a)
    {
        const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)a, 0x138, 0xF, 0xF, false);
        const uint32_t h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(a >> 32), 0x138, 0xF, 0xF, false);
        return (uint64_t)l | ((uint64_t)h << 32);
    }
    static __device__ __forceinline__ u64 row_below(u64 a)
    {
        const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)a, 0x130, 0xF, 0xF, false);
        const uint32_t h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(a >> 32), 0x130, 0xF, 0xF, false);
        return (uint64_t)l | ((uint64_t)h << 32);
    }
    // the value of the row k above / below inside the 16-lane DPP row, 0 where that leaves the row (row_shr / row_shl, bound_ctrl: 0)
    template <int CTRL> static __device__ __forceinline__ u64 dpp64(u64 a)
    {
        const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)a, CTRL, 0xF, 0xF, true);
        const uint32_t h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(a >> 32), CTRL, 0xF, 0xF, true);
        return (uint64_t)l | ((uint64_t)h << 32);
    }
    static __device__ __forceinline__ u64 rows_down(u64 a, int k) { return k == 1 ? dpp64<0x111>(a) : k == 2 ? dpp64<0x112>(a) : k == 4 ? dpp64<0x114>(a) : dpp64<0x118>(a); }
    static __device__ __forceinline__ u64 rows_up(u64 a, int k) { return k == 1 ? dpp64<0x101>(a) : k == 2 ? dpp64<0x102>(a) : k == 4 ? dpp64<0x104>(a) : dpp64<0x108>(a); }
    // the tile's row 15 as seen from its rows 16 .. 31 (row_bcast:15 writes lane 15 of every DPP row to the next row; other lanes: 0) / its row 16 as seen
    // from rows 0 .. 15 (no DPP form: two lane reads)
    static __device__ __forceinline__ u64 row15_of_upper(u64 a)
    {
        const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)a, 0x142, 0xA, 0xF, false);
        const uint32_t h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(a >> 32), 0x142, 0xA, 0xF, false);
        return (uint64_t)l | ((uint64_t)h << 32);
    }
    static __device__ __forceinline__ u64 row16_of_lower(u64 a)
    {
        const uint64_t a16 = read_lane64(a, 16), b16 = read_lane64(a, 48);
        return (threadIdx.x & 32u) ? b16 : a16;
    }
    // all-reduce over the 32 lanes of a tile: lane ^ 1, lane ^ 2, mirror in 8, mirror in 16 (fused DPP), lane ^ 16 (swizzle)
#define T2_HALF_RED(v, OPNAME, COMBINE)                                        \
    do {                                                                       \
        v = DPP_FUSED(OPNAME, "quad_perm:[1,0,3,2]", v);                       \
        v = DPP_FUSED(OPNAME, "quad_perm:[2,3,0,1]", v);                       \
        v = DPP_FUSED(OPNAME, "row_half_mirror", v);                           \
        v = DPP_FUSED(OPNAME, "row_mirror", v);                                \
        const uint32_t o_ = (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); \
        v = COMBINE(v, o_);                                                    \
    } while (0)
    static __device__ __forceinline__ u32 half_sum(u32 v) { T2_HALF_RED(v, "v_add_u32", OP_ADD); return v; }
    static __device__ __forceinline__ u32 half_or(u32 v) { T2_HALF_RED(v, "v_or_b32", OP_OR); return v; }
    static __device__ __forceinline__ u32 half_min(u32 v) { T2_HALF_RED(v, "v_min_u32", OP_MIN); return v; }
    static __device__ __forceinline__ uint32_t wave_or(u32 v)
    {
        v = half_or(v);
        return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
    }
    static __device__ __forceinline__ uint32_t read_lane(u32 v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
    static __device__ __forceinline__ uint64_t read_lane64(u64 v, int l)
    {
        return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32);
    }
    // the lane's 64 pixels; fast: four 16-byte loads (aligned rows, readable up to the tile's end); else byte by byte inside the image
    static __device__ __forceinline__ void load_row(const uint8_t *base, u32 off, mask rowvalid, u32 ncols, bool fast, u32 (&out)[16])
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) out[k] = 0u;
        if (fast) {
            if (rowvalid) {
                const uint4 *p = reinterpret_cast<const uint4 *>(base + off);
                const uint4  v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
                out[0] = v0.x; out[1] = v0.y; out[2] = v0.z; out[3] = v0.w; out[4] = v1.x; out[5] = v1.y; out[6] = v1.z; out[7] = v1.w;
                out[8] = v2.x; out[9] = v2.y; out[10] = v2.z; out[11] = v2.w; out[12] = v3.x; out[13] = v3.y; out[14] = v3.z; out[15] = v3.w;
            }
        } else if (rowvalid) {
            const uint8_t *p = base + off;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                uint32_t x = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((uint32_t)(4 * k + j) < ncols) x |= (uint32_t)p[4 * k + j] << (8 * j);
                out[k] = x;
            }
        }
    }
    __device__ __forceinline__ void rec_write(u32 half, u32 id, const u32 (&f)[8], mask m)
    {
        if (m) {
            uint4 *p = reinterpret_cast<uint4 *>(lds_rec + (half * (uint32_t)REC_CAP + id) * 8u);
            p[0] = make_uint4(f[0], f[1], f[2], f[3]);
            p[1] = make_uint4(f[4], f[5], f[6], f[7]);
        }
    }
    __device__ __forceinline__ void rec_read(u32 half, u32 id, u32 (&f)[8], mask m) const
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = 0u;
        if (m) {
            const uint4 *p = reinterpret_cast<const uint4 *>(lds_rec + (half * (uint32_t)REC_CAP + id) * 8u);
            const uint4  a = p[0], c = p[1];
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
        }
    }
    __device__ __forceinline__ void rec_set_par(u32 half, u32 id, u32 val, mask m) { if (m) lds_rec[(half * (uint32_t)REC_CAP + id) * 8u] = val; }
    __device__ __forceinline__ void idmap_write(u32 half, u32 pix, u32 id, mask m) { if (m) lds_idmap[half * 2048u + pix] = (uint8_t)id; }
    __device__ __forceinline__ u32 idmap_read(u32 half, u32 pix, mask m) const { return m ? (uint32_t)lds_idmap[half * 2048u + (pix & 2047u)] : 0u; }
    // one operation for the wave
    __device__ __forceinline__ uint32_t atomic_add(uint32_t *p, uint32_t v)
    {
        uint32_t r = 0;
        if ((threadIdx.x & 63u) == 0u) r = atomicAdd(p, v);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    }
    __device__ __forceinline__ void atomic_or(uint32_t *p, uint32_t v) { if ((threadIdx.x & 63u) == 0u) atomicOr(p, v); }
    template <class T> __device__ __forceinline__ void store_scalar(T *p, T v) { if ((threadIdx.x & 63u) == 0u) *p = v; }
    static __device__ __forceinline__ void store_u16(uint16_t *p, u32 idx, u32 val, mask m) { if (m) p[idx] = (uint16_t)val; }
    static __device__ __forceinline__ void store_u32(uint32_t *p, u32 idx, u32 val, mask m) { if (m) p[idx] = val; }
    static __device__ __forceinline__ void store_rec(NodeRec *p, u32 idx, const u32 (&f)[8], mask m)
    {
        if (m) {
            uint4 *d = reinterpret_cast<uint4 *>(p + idx);
            d[0] = make_uint4(f[0], f[1], f[2], f[3]);
            d[1] = make_uint4(f[4], f[5], f[6], f[7]);
        }
    }
};
static_assert(REC_CAP <= 256, "the key-pixel table holds record numbers in bytes");

} // namespace t2

// One wave per pair of tiles; 8 KB of LDS (the two tiles' records + key-pixel tables).
__global__ __launch_bounds__(64) void k_tile_tree2(BatchDev b, DetectParams prm, t2::Args a)
{
    __shared__ uint32_t s_rec[2 * t2::REC_CAP * 8] __attribute__((aligned(16)));
    __shared__ uint8_t  s_idmap[2 * 2048];
    t2::DevWave w{s_rec, s_idmap};
    t2::Body<t2::DevWave>::run(w, b, prm, a, blockIdx.x);
}

void launch_tile_tree2(hipStream_t s, const BatchDev &b, const DetectParams &p, const uint32_t *pairs, uint32_t n_pairs, uint32_t *fb_list, uint32_t *fb_count)
{
    if (!n_pairs) return;
    t2::Args a{pairs, n_pairs, fb_list, fb_count};
    hipLaunchKernelGGL(k_tile_tree2, dim3(n_pairs), dim3(64), 0, s, b, p, a);
}
