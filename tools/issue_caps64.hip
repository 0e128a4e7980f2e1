// Developer aid (GPU box): issue rate of the 64-bit / bit-manipulation instructions k_tile_tree2 is made of, 4 waves a SIMD (its occupancy), independent chains.
//   hipcc --offload-arch=gfx950 -O2 tools/issue_caps64.hip -o /tmp/ic64 && /tmp/ic64
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
enum { B_LSHLADD64, B_ADDCO, B_BFREV, B_BFI, B_AND, B_MOV64, B_LSHL64, B_BCNT, B_FFBL, B_DPPAND, N_BODY };
static const char *names[N_BODY] = {"v_lshl_add_u64 (64-bit add)", "v_add_co_u32 + v_addc_co_u32 (a 64-bit add as two)", "v_bfrev_b32", "v_bfi_b32", "v_and_b32", "v_mov_b64", "v_lshlrev_b64",
                                    "v_bcnt_u32_b32", "v_ffbl_b32", "v_and_b32 dpp row_shr:1"};
template <int BODY> __global__ __launch_bounds__(256) void k(uint64_t *out, int n)
{
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull, b = a ^ 0x1234567ull, c = a + 77, d = b + 99;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (BODY == B_LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %2\nv_lshl_add_u64 %1, %1, 0, %3\nv_lshl_add_u64 %2, %2, 0, %0\nv_lshl_add_u64 %3, %3, 0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            else if constexpr (BODY == B_ADDCO) { uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32), cl = (uint32_t)c, chh = (uint32_t)(c >> 32), dl = (uint32_t)d, dh = (uint32_t)(d >> 32);
                asm volatile("v_add_co_u32 %0, vcc, %0, %4\nv_addc_co_u32 %1, vcc, %1, %5, vcc\nv_add_co_u32 %2, vcc, %2, %6\nv_addc_co_u32 %3, vcc, %3, %7, vcc\n"
                             "v_add_co_u32 %4, vcc, %4, %0\nv_addc_co_u32 %5, vcc, %5, %1, vcc\nv_add_co_u32 %6, vcc, %6, %2\nv_addc_co_u32 %7, vcc, %7, %3, vcc"
                             : "+v"(al), "+v"(ah), "+v"(bl), "+v"(bh), "+v"(cl), "+v"(chh), "+v"(dl), "+v"(dh) :: "vcc");
                a = al | ((uint64_t)ah << 32); b = bl | ((uint64_t)bh << 32); c = cl | ((uint64_t)chh << 32); d = dl | ((uint64_t)dh << 32); }
            else if constexpr (BODY == B_BFREV) { uint32_t x = (uint32_t)a, y = (uint32_t)b, z = (uint32_t)c, w = (uint32_t)d; asm volatile("v_bfrev_b32 %0, %0\nv_bfrev_b32 %1, %1\nv_bfrev_b32 %2, %2\nv_bfrev_b32 %3, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); a = x; b = y; c = z; d = w; }
            else if constexpr (BODY == B_BFI) { uint32_t x = (uint32_t)a, y = (uint32_t)b, z = (uint32_t)c, w = (uint32_t)d; asm volatile("v_bfi_b32 %0, %1, %2, %0\nv_bfi_b32 %1, %2, %3, %1\nv_bfi_b32 %2, %3, %0, %2\nv_bfi_b32 %3, %0, %1, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); a = x; b = y; c = z; d = w; }
            else if constexpr (BODY == B_AND) { uint32_t x = (uint32_t)a, y = (uint32_t)b, z = (uint32_t)c, w = (uint32_t)d; asm volatile("v_and_b32 %0, %1, %0\nv_and_b32 %1, %2, %1\nv_and_b32 %2, %3, %2\nv_and_b32 %3, %0, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); a = x; b = y; c = z; d = w; }
            else if constexpr (BODY == B_MOV64) asm volatile("v_mov_b64 %0, %1\nv_mov_b64 %1, %2\nv_mov_b64 %2, %3\nv_mov_b64 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            else if constexpr (BODY == B_LSHL64) asm volatile("v_lshlrev_b64 %0, 1, %0\nv_lshlrev_b64 %1, 1, %1\nv_lshlrev_b64 %2, 1, %2\nv_lshlrev_b64 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            else if constexpr (BODY == B_BCNT) { uint32_t x = (uint32_t)a, y = (uint32_t)b, z = (uint32_t)c, w = (uint32_t)d; asm volatile("v_bcnt_u32_b32 %0, %1, %0\nv_bcnt_u32_b32 %1, %2, %1\nv_bcnt_u32_b32 %2, %3, %2\nv_bcnt_u32_b32 %3, %0, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); a = x; b = y; c = z; d = w; }
            else if constexpr (BODY == B_FFBL) { uint32_t x = (uint32_t)a, y = (uint32_t)b, z = (uint32_t)c, w = (uint32_t)d; asm volatile("v_ffbl_b32 %0, %1\nv_ffbl_b32 %1, %2\nv_ffbl_b32 %2, %3\nv_ffbl_b32 %3, %0" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); a = x; b = y; c = z; d = w; }
            else if constexpr (BODY == B_DPPAND) { uint32_t x = (uint32_t)a, y = (uint32_t)b, z = (uint32_t)c, w = (uint32_t)d; asm volatile("v_and_b32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_and_b32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_and_b32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_and_b32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); a = x; b = y; c = z; d = w; }
        }
    }
    if ((a ^ b ^ c ^ d) == 0x123456789ull) out[0] = a;
}
template <int BODY> static void run(uint64_t *d, int waves_per_simd)
{
    int dev; CK(hipGetDevice(&dev)); hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    const int n = 4000, blocks = p.multiProcessorCount * waves_per_simd;      // 256 lanes = 4 waves a block: one per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<BODY>, dim3(blocks), dim3(256), 0, 0, d, 10);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<BODY>, dim3(blocks), dim3(256), 0, 0, d, n); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_wave = (double)n * 8 * 4 * (BODY == B_ADDCO ? 1 : 1), cycles = ms * 1e-3 * (double)p.clockRate * 1e3;
    printf("%-52s %d waves/SIMD: %.3f instruction groups per cycle and SIMD (%s)\n", names[BODY], waves_per_simd, per_wave * waves_per_simd / cycles, BODY == B_ADDCO ? "a group = the pair" : "a group = one instruction");
}
int main()
{
    uint64_t *d; CK(hipMalloc(&d, 64));
    for (int w : {4, 8}) {
        run<B_LSHLADD64>(d, w); run<B_ADDCO>(d, w); run<B_BFREV>(d, w); run<B_BFI>(d, w); run<B_AND>(d, w); run<B_MOV64>(d, w); run<B_LSHL64>(d, w); run<B_BCNT>(d, w); run<B_FFBL>(d, w); run<B_DPPAND>(d, w);
    }
    return 0;
}
