// Developer aid (GPU box): what are the instruction issue caps of a gfx950 CU?
//
//   hipcc --offload-arch=gfx950 -O2 tools/issue_caps.hip -o gpurun_out/issue_caps && gpurun_out/issue_caps
//
// k_tile_tree's time follows the number of instructions its waves issue.  Whether that is a hardware cap (then only
// "fewer instructions of that kind" helps) or latency (then more overlap helps) depends on two numbers nobody had measured: how many
// vector instructions a SIMD issues per cycle for plain 32-bit integer work, and how many scalar / branch instructions a CU issues per
// cycle.  Every test is a loop of REPS blocks of UNROLL independent instructions (8 accumulators, so no dependency stalls at >= 2 waves),
// run with 1 / 2 / 4 / 8 waves per SIMD (256-thread workgroups, 1 / 2 / 4 / 8 per CU, every CU filled).  Per wave: s_memtime around the
// loop; reported: instructions per cycle per SIMD (vector) and per CU (scalar), from the mean wave duration and the number of waves that
// shared the unit (checked from HW_ID: every SIMD must hold exactly the intended number of waves).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REPS = 2000;

struct WaveRec { unsigned long long t0, t1; uint32_t hwid, xcc; };

#define V8(OP, PRE, POST)                                  \
    OP " %0, " PRE "%0" POST "\n" OP " %1, " PRE "%1" POST "\n" OP " %2, " PRE "%2" POST "\n" OP " %3, " PRE "%3" POST "\n" \
    OP " %4, " PRE "%4" POST "\n" OP " %5, " PRE "%5" POST "\n" OP " %6, " PRE "%6" POST "\n" OP " %7, " PRE "%7" POST "\n"
#define V32(OP, PRE, POST) V8(OP, PRE, POST) V8(OP, PRE, POST) V8(OP, PRE, POST) V8(OP, PRE, POST)
#define VREGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define SREGS "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)

// One kernel, the body picked by a template parameter (so that every body is its own straight-line loop).
enum Body {
    B_VADD, B_VAND, B_VFMA, B_VCNDMASK, B_VDPP, B_VCMP, B_VPERM, B_VMADU24, B_VMULLO, B_VLSHLOR, B_VBFE, B_VPKADD, B_VADD3,
    B_VOR, B_VXOR, B_VLSHL, B_VLSHR, B_VSUB, B_VMIN, B_VMAX, B_VMOV, B_VADD64, B_VCND64, B_VCNDVCC, B_VANDOR, B_VBFI, B_VALIGN, B_VFFBL, B_VBCNT, B_VMBCNT, B_VLSHLADD, B_VCMP64, B_VSDWA, B_VADDCO, B_VBPERM, B_VMED3, B_CND_A, B_CND_B, B_PAIR_VCC, B_PAIR_SGPR, B_PAIR_VCC64, B_CMP_2CND, B_CMP_4CND, B_CND_ADD, B_CND_MIX64, B_CMP_4CND64,
    B_SADD, B_SBRANCH_NT, B_SBRANCH_T, B_SMIX_VS, B_SMIX_VSB, B_VREADLANE, B_DSREAD, B_DSATOMIC, B_TILEMIX, B_N
};
static const char *body_name[B_N] = {
    "v_add_u32", "v_and_b32", "v_fma_f32", "v_cndmask_b32", "v_mov_b32 dpp row_shr:1", "v_cmp_lt_u32 -> vcc", "v_perm_b32", "v_mad_u32_u24",
    "v_mul_lo_u32", "v_lshl_or_b32", "v_bfe_u32", "v_pk_add_u16", "v_add3_u32",
    "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_u32", "v_min_u32", "v_max_u32", "v_mov_b32", "v_add_u32_e64 (sgpr operand)", "v_cndmask_b32_e64 (sgpr pair)", "v_cndmask_b32_e32 (vcc set before)", "v_and_or_b32", "v_bfi_b32", "v_alignbit_b32", "v_ffbl_b32", "v_bcnt_u32_b32", "v_mbcnt_lo_u32_b32", "v_lshl_add_u32", "v_cmp_lt_u32_e64 -> sgpr pair", "v_and_b32_sdwa (byte select)", "v_add_co_u32 (carry out)", "ds_bpermute_b32", "v_med3_u32", "v_cndmask_e32 vcc, distinct regs", "v_cndmask_e64 with vcc named", "pairs v_cmp_e32 vcc + v_cndmask_e32", "pairs v_cmp_e64 s[] + v_cndmask_e64 s[]", "pairs v_cmp_e64 vcc + v_cndmask_e64 vcc", "v_cmp_e32 + 2 v_cndmask_e32", "v_cmp_e32 + 4 v_cndmask_e32 (x6 + 2)", "v_cndmask_e32 alternating with v_add_u32", "v_cndmask_e32 alternating with v_cndmask_e64 s[]", "v_cmp_e32 + 4 v_cndmask_e64 vcc (x6 + 2)",
    "s_add_u32", "s_cbranch_scc0 (not taken) + s_cmp", "s_branch (taken)", "1 v_add : 1 s_add", "4 v_add : 2 s_add : 1 branch", "v_readlane_b32",
    "ds_read_b32 (8 in flight)", "ds_add_u32 (distinct addresses)", "tile mix 12 valu : 7 salu : 2 branch : 2 lds"};
// instructions of the counted kind per block of the loop body, {vector, scalar + branch, lds}
struct Mix { int v, s, l; };
static const Mix body_mix[B_N] = {
    {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0},
    {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {0, 0, 32}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {30, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0}, {32, 0, 0},
    {0, 32, 0}, {0, 32, 0}, {0, 32, 0}, {16, 16, 0}, {16, 12, 0}, {32, 0, 0}, {0, 0, 8}, {0, 0, 8}, {12, 9, 2}};

template <int BODY>
__global__ __launch_bounds__(256) void k_issue(WaveRec *out, int reps)
{
    __shared__ uint32_t lds[4096];
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    s0 = __builtin_amdgcn_readfirstlane(s0);
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    const uint32_t laddr = (threadIdx.x * 4u) & 16383u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if constexpr (BODY == B_VADD) asm volatile(V32("v_add_u32", "1, ", "") : VREGS);
        else if constexpr (BODY == B_VAND) asm volatile(V32("v_and_b32", "0x7fffffff, ", "") : VREGS);
        else if constexpr (BODY == B_VFMA) asm volatile(V32("v_fma_f32", "%0, %0, ", "") : VREGS);
        else if constexpr (BODY == B_VCNDMASK) asm volatile(V32("v_cndmask_b32", "%1, ", ", vcc") : VREGS :: "vcc");
        else if constexpr (BODY == B_VDPP)
            asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         : VREGS);
        else if constexpr (BODY == B_VCMP)
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %1, %2\nv_cmp_lt_u32 vcc, %2, %3\nv_cmp_lt_u32 vcc, %3, %4\n"
                         "v_cmp_lt_u32 vcc, %4, %5\nv_cmp_lt_u32 vcc, %5, %6\nv_cmp_lt_u32 vcc, %6, %7\nv_cmp_lt_u32 vcc, %7, %0\n"
                         "v_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %1, %2\nv_cmp_lt_u32 vcc, %2, %3\nv_cmp_lt_u32 vcc, %3, %4\n"
                         "v_cmp_lt_u32 vcc, %4, %5\nv_cmp_lt_u32 vcc, %5, %6\nv_cmp_lt_u32 vcc, %6, %7\nv_cmp_lt_u32 vcc, %7, %0\n"
                         "v_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %1, %2\nv_cmp_lt_u32 vcc, %2, %3\nv_cmp_lt_u32 vcc, %3, %4\n"
                         "v_cmp_lt_u32 vcc, %4, %5\nv_cmp_lt_u32 vcc, %5, %6\nv_cmp_lt_u32 vcc, %6, %7\nv_cmp_lt_u32 vcc, %7, %0\n"
                         "v_cmp_lt_u32 vcc, %0, %1\nv_cmp_lt_u32 vcc, %1, %2\nv_cmp_lt_u32 vcc, %2, %3\nv_cmp_lt_u32 vcc, %3, %4\n"
                         "v_cmp_lt_u32 vcc, %4, %5\nv_cmp_lt_u32 vcc, %5, %6\nv_cmp_lt_u32 vcc, %6, %7\nv_cmp_lt_u32 vcc, %7, %0\n"
                         : VREGS :: "vcc");
        else if constexpr (BODY == B_VPERM) asm volatile(V32("v_perm_b32", "%1, %2, ", "") : VREGS);
        else if constexpr (BODY == B_VMADU24) asm volatile(V32("v_mad_u32_u24", "3, %1, ", "") : VREGS);
        else if constexpr (BODY == B_VMULLO) asm volatile(V32("v_mul_lo_u32", "%1, ", "") : VREGS);
        else if constexpr (BODY == B_VLSHLOR) asm volatile(V32("v_lshl_or_b32", "%1, 1, ", "") : VREGS);
        else if constexpr (BODY == B_VBFE) asm volatile(V32("v_bfe_u32", "", ", 3, 8") : VREGS);
        else if constexpr (BODY == B_VPKADD) asm volatile(V32("v_pk_add_u16", "%1, ", "") : VREGS);
        else if constexpr (BODY == B_VADD3) asm volatile(V32("v_add3_u32", "%1, 1, ", "") : VREGS);
        else if constexpr (BODY == B_VOR) asm volatile(V32("v_or_b32", "0x10000, ", "") : VREGS);
        else if constexpr (BODY == B_VXOR) asm volatile(V32("v_xor_b32", "1, ", "") : VREGS);
        else if constexpr (BODY == B_VLSHL) asm volatile(V32("v_lshlrev_b32", "1, ", "") : VREGS);
        else if constexpr (BODY == B_VLSHR) asm volatile(V32("v_lshrrev_b32", "1, ", "") : VREGS);
        else if constexpr (BODY == B_VSUB) asm volatile(V32("v_sub_u32", "1, ", "") : VREGS);
        else if constexpr (BODY == B_VMIN) asm volatile(V32("v_min_u32", "%1, ", "") : VREGS);
        else if constexpr (BODY == B_VMAX) asm volatile(V32("v_max_u32", "%1, ", "") : VREGS);
        else if constexpr (BODY == B_VMOV)
            asm volatile("v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %0\n"
                         "v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %0\n"
                         "v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %0\n"
                         "v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %0\n" : VREGS);
        else if constexpr (BODY == B_VADD64) asm volatile(V32("v_add_u32_e64", "%8, ", "") : VREGS, SREGS);
        else if constexpr (BODY == B_VCND64) asm volatile(V32("v_cndmask_b32_e64", "%1, ", ", s[20:21]") : VREGS :: "s20", "s21");
        else if constexpr (BODY == B_VCNDVCC) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n" V32("v_cndmask_b32", "%1, ", ", vcc") : VREGS :: "vcc");
        else if constexpr (BODY == B_VANDOR) asm volatile(V32("v_and_or_b32", "%1, 63, ", "") : VREGS);
        else if constexpr (BODY == B_VBFI) asm volatile(V32("v_bfi_b32", "%1, %2, ", "") : VREGS);
        else if constexpr (BODY == B_VALIGN) asm volatile(V32("v_alignbit_b32", "%1, ", ", 8") : VREGS);
        else if constexpr (BODY == B_VFFBL) asm volatile(V32("v_ffbl_b32", "", "") : VREGS);
        else if constexpr (BODY == B_VBCNT) asm volatile(V32("v_bcnt_u32_b32", "%1, ", "") : VREGS);
        else if constexpr (BODY == B_VMBCNT) asm volatile(V32("v_mbcnt_lo_u32_b32", "-1, ", "") : VREGS);
        else if constexpr (BODY == B_VLSHLADD) asm volatile(V32("v_lshl_add_u32", "%1, 3, ", "") : VREGS);
        else if constexpr (BODY == B_VCMP64)
            asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\nv_cmp_lt_u32_e64 s[22:23], %1, %2\nv_cmp_lt_u32_e64 s[24:25], %2, %3\nv_cmp_lt_u32_e64 s[26:27], %3, %4\n"
                         "v_cmp_lt_u32_e64 s[20:21], %4, %5\nv_cmp_lt_u32_e64 s[22:23], %5, %6\nv_cmp_lt_u32_e64 s[24:25], %6, %7\nv_cmp_lt_u32_e64 s[26:27], %7, %0\n"
                         "v_cmp_lt_u32_e64 s[20:21], %0, %1\nv_cmp_lt_u32_e64 s[22:23], %1, %2\nv_cmp_lt_u32_e64 s[24:25], %2, %3\nv_cmp_lt_u32_e64 s[26:27], %3, %4\n"
                         "v_cmp_lt_u32_e64 s[20:21], %4, %5\nv_cmp_lt_u32_e64 s[22:23], %5, %6\nv_cmp_lt_u32_e64 s[24:25], %6, %7\nv_cmp_lt_u32_e64 s[26:27], %7, %0\n"
                         "v_cmp_lt_u32_e64 s[20:21], %0, %1\nv_cmp_lt_u32_e64 s[22:23], %1, %2\nv_cmp_lt_u32_e64 s[24:25], %2, %3\nv_cmp_lt_u32_e64 s[26:27], %3, %4\n"
                         "v_cmp_lt_u32_e64 s[20:21], %4, %5\nv_cmp_lt_u32_e64 s[22:23], %5, %6\nv_cmp_lt_u32_e64 s[24:25], %6, %7\nv_cmp_lt_u32_e64 s[26:27], %7, %0\n"
                         "v_cmp_lt_u32_e64 s[20:21], %0, %1\nv_cmp_lt_u32_e64 s[22:23], %1, %2\nv_cmp_lt_u32_e64 s[24:25], %2, %3\nv_cmp_lt_u32_e64 s[26:27], %3, %4\n"
                         "v_cmp_lt_u32_e64 s[20:21], %4, %5\nv_cmp_lt_u32_e64 s[22:23], %5, %6\nv_cmp_lt_u32_e64 s[24:25], %6, %7\nv_cmp_lt_u32_e64 s[26:27], %7, %0\n"
                         : VREGS :: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        else if constexpr (BODY == B_VSDWA) asm volatile(V32("v_and_b32_sdwa", "%1, ", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") : VREGS);
        else if constexpr (BODY == B_VADDCO) asm volatile(V32("v_add_co_u32", "vcc, 1, ", "") : VREGS :: "vcc");
        else if constexpr (BODY == B_VBPERM) {
            asm volatile(V32("ds_bpermute_b32", "%8, ", "") "s_waitcnt lgkmcnt(0)\n" : VREGS : "v"(laddr & 255u) : "memory");
        }
        else if constexpr (BODY == B_VMED3) asm volatile(V32("v_med3_u32", "%1, %2, ", "") : VREGS);
        else if constexpr (BODY == B_CND_A) {
#define CA(d, x, y) "v_cndmask_b32_e32 %" #d ", %" #x ", %" #y ", vcc\n"
#define CA8 CA(0, 1, 2) CA(3, 4, 5) CA(6, 7, 0) CA(1, 2, 3) CA(4, 5, 6) CA(7, 0, 1) CA(2, 3, 4) CA(5, 6, 7)
            asm volatile(CA8 CA8 CA8 CA8 : VREGS :: "vcc");
        } else if constexpr (BODY == B_CND_B) {
#define CB(d, x, y) "v_cndmask_b32_e64 %" #d ", %" #x ", %" #y ", vcc\n"
#define CB8 CB(0, 1, 2) CB(3, 4, 5) CB(6, 7, 0) CB(1, 2, 3) CB(4, 5, 6) CB(7, 0, 1) CB(2, 3, 4) CB(5, 6, 7)
            asm volatile(CB8 CB8 CB8 CB8 : VREGS :: "vcc");
        } else if constexpr (BODY == B_PAIR_VCC) {
#define PV(d, x, y) "v_cmp_lt_u32_e32 vcc, %" #x ", %" #y "\nv_cndmask_b32_e32 %" #d ", %" #x ", %" #y ", vcc\n"
#define PV8 PV(0, 1, 2) PV(3, 4, 5) PV(6, 7, 0) PV(1, 2, 3)
            asm volatile(PV8 PV8 PV8 PV8 : VREGS :: "vcc");
        } else if constexpr (BODY == B_PAIR_SGPR) {
#define PS(d, x, y, sr) "v_cmp_lt_u32_e64 " sr ", %" #x ", %" #y "\nv_cndmask_b32_e64 %" #d ", %" #x ", %" #y ", " sr "\n"
#define PS8 PS(0, 1, 2, "s[20:21]") PS(3, 4, 5, "s[22:23]") PS(6, 7, 0, "s[24:25]") PS(1, 2, 3, "s[26:27]")
            asm volatile(PS8 PS8 PS8 PS8 : VREGS :: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if constexpr (BODY == B_PAIR_VCC64) {
#define PW8 PS(0, 1, 2, "vcc") PS(3, 4, 5, "vcc") PS(6, 7, 0, "vcc") PS(1, 2, 3, "vcc")
            asm volatile(PW8 PW8 PW8 PW8 : VREGS :: "vcc");
        }
        else if constexpr (BODY == B_CMP_2CND) {
#define P2(d, e, x, y) "v_cmp_lt_u32_e32 vcc, %" #x ", %" #y "\nv_cndmask_b32_e32 %" #d ", %" #x ", %" #y ", vcc\nv_cndmask_b32_e32 %" #e ", %" #y ", %" #x ", vcc\n"
#define P2x5 P2(0, 1, 2, 3) P2(4, 5, 6, 7) P2(2, 3, 0, 1) P2(6, 7, 4, 5) P2(0, 1, 2, 3)
            asm volatile(P2x5 P2x5 : VREGS :: "vcc");
        } else if constexpr (BODY == B_CMP_4CND) {
#define P4(x, y) "v_cmp_lt_u32_e32 vcc, %" #x ", %" #y "\nv_cndmask_b32_e32 %0, %" #x ", %" #y ", vcc\nv_cndmask_b32_e32 %1, %" #y ", %" #x ", vcc\nv_cndmask_b32_e32 %2, %" #x ", %" #y ", vcc\nv_cndmask_b32_e32 %3, %" #y ", %" #x ", vcc\n"
            asm volatile(P4(4, 5) P4(6, 7) P4(4, 5) P4(6, 7) P4(4, 5) P4(6, 7) "v_add_u32 %4, 1, %4\nv_add_u32 %5, 1, %5\n" : VREGS :: "vcc");
        } else if constexpr (BODY == B_CND_ADD) {
#define CAD(d, x, y, z) "v_cndmask_b32_e32 %" #d ", %" #x ", %" #y ", vcc\nv_add_u32 %" #z ", 1, %" #z "\n"
#define CAD8 CAD(0, 1, 2, 3) CAD(4, 5, 6, 7) CAD(1, 2, 3, 0) CAD(5, 6, 7, 4)
            asm volatile(CAD8 CAD8 CAD8 CAD8 : VREGS :: "vcc");
        } else if constexpr (BODY == B_CND_MIX64) {
#define CMX(d, x, y, e) "v_cndmask_b32_e32 %" #d ", %" #x ", %" #y ", vcc\nv_cndmask_b32_e64 %" #e ", %" #x ", %" #y ", s[20:21]\n"
#define CMX8 CMX(0, 1, 2, 3) CMX(4, 5, 6, 7) CMX(1, 2, 3, 0) CMX(5, 6, 7, 4)
            asm volatile(CMX8 CMX8 CMX8 CMX8 : VREGS :: "vcc", "s20", "s21");
        } else if constexpr (BODY == B_CMP_4CND64) {
#define Q4(x, y) "v_cmp_lt_u32_e32 vcc, %" #x ", %" #y "\nv_cndmask_b32_e64 %0, %" #x ", %" #y ", vcc\nv_cndmask_b32_e64 %1, %" #y ", %" #x ", vcc\nv_cndmask_b32_e64 %2, %" #x ", %" #y ", vcc\nv_cndmask_b32_e64 %3, %" #y ", %" #x ", vcc\n"
            asm volatile(Q4(4, 5) Q4(6, 7) Q4(4, 5) Q4(6, 7) Q4(4, 5) Q4(6, 7) "v_add_u32 %4, 1, %4\nv_add_u32 %5, 1, %5\n" : VREGS :: "vcc");
        }
        else if constexpr (BODY == B_SADD) asm volatile(V32("s_add_u32", "", ", 1") : SREGS :: "scc");
        else if constexpr (BODY == B_SBRANCH_NT) {
            // 16 x (s_cmp + never-taken conditional branch)
#define NT2 "s_cmp_eq_u32 %0, %0\ns_cbranch_scc0 1f\n"
            asm volatile(NT2 NT2 NT2 NT2 NT2 NT2 NT2 NT2 NT2 NT2 NT2 NT2 NT2 NT2 NT2 NT2 "1:\n" : "+s"(s0) :: "scc");
        } else if constexpr (BODY == B_SBRANCH_T) {
#define T1(n) "s_branch " #n "f\n" #n ":\n"
            asm volatile(T1(1) T1(2) T1(3) T1(4) T1(5) T1(6) T1(7) T1(8) T1(9) T1(10) T1(11) T1(12) T1(13) T1(14) T1(15) T1(16) T1(17) T1(18) T1(19) T1(20) T1(21)
                             T1(22) T1(23) T1(24) T1(25) T1(26) T1(27) T1(28) T1(29) T1(30) T1(31) T1(32) ::: "memory");
        } else if constexpr (BODY == B_SMIX_VS) {
#define VS2(i, j) "v_add_u32 %" #i ", %" #i ", 1\ns_add_u32 %" #j ", %" #j ", 1\n"
            asm volatile(VS2(0, 8) VS2(1, 9) VS2(2, 10) VS2(3, 11) VS2(4, 12) VS2(5, 13) VS2(6, 14) VS2(7, 15)
                         VS2(0, 8) VS2(1, 9) VS2(2, 10) VS2(3, 11) VS2(4, 12) VS2(5, 13) VS2(6, 14) VS2(7, 15)
                         : VREGS, SREGS :: "scc");
        } else if constexpr (BODY == B_SMIX_VSB) {
            // 4 x (4 v_add, 2 s_add, 1 compare + never-taken branch): 16 vector, 8 + 4 scalar-unit instructions
#define VSB(i, j, k, l, m, n) "v_add_u32 %" #i ", %" #i ", 1\nv_add_u32 %" #j ", %" #j ", 1\ns_add_u32 %" #m ", %" #m ", 1\nv_add_u32 %" #k ", %" #k ", 1\n" \
                              "v_add_u32 %" #l ", %" #l ", 1\ns_cmp_eq_u32 %" #n ", 0\ns_cbranch_scc1 9f\n"
            asm volatile(VSB(0, 1, 2, 3, 8, 9) VSB(4, 5, 6, 7, 10, 11) VSB(0, 1, 2, 3, 12, 13) VSB(4, 5, 6, 7, 14, 15) "9:\n" : VREGS, SREGS :: "scc");
        } else if constexpr (BODY == B_VREADLANE) {
            asm volatile("v_readlane_b32 %8, %0, 3\nv_readlane_b32 %9, %1, 3\nv_readlane_b32 %10, %2, 3\nv_readlane_b32 %11, %3, 3\n"
                         "v_readlane_b32 %12, %4, 3\nv_readlane_b32 %13, %5, 3\nv_readlane_b32 %14, %6, 3\nv_readlane_b32 %15, %7, 3\n"
                         "v_readlane_b32 %8, %0, 3\nv_readlane_b32 %9, %1, 3\nv_readlane_b32 %10, %2, 3\nv_readlane_b32 %11, %3, 3\n"
                         "v_readlane_b32 %12, %4, 3\nv_readlane_b32 %13, %5, 3\nv_readlane_b32 %14, %6, 3\nv_readlane_b32 %15, %7, 3\n"
                         "v_readlane_b32 %8, %0, 3\nv_readlane_b32 %9, %1, 3\nv_readlane_b32 %10, %2, 3\nv_readlane_b32 %11, %3, 3\n"
                         "v_readlane_b32 %12, %4, 3\nv_readlane_b32 %13, %5, 3\nv_readlane_b32 %14, %6, 3\nv_readlane_b32 %15, %7, 3\n"
                         "v_readlane_b32 %8, %0, 3\nv_readlane_b32 %9, %1, 3\nv_readlane_b32 %10, %2, 3\nv_readlane_b32 %11, %3, 3\n"
                         "v_readlane_b32 %12, %4, 3\nv_readlane_b32 %13, %5, 3\nv_readlane_b32 %14, %6, 3\nv_readlane_b32 %15, %7, 3\n"
                         : VREGS, SREGS);
        } else if constexpr (BODY == B_DSREAD) {
            asm volatile("ds_read_b32 %0, %8\nds_read_b32 %1, %8 offset:256\nds_read_b32 %2, %8 offset:512\nds_read_b32 %3, %8 offset:768\n"
                         "ds_read_b32 %4, %8 offset:1024\nds_read_b32 %5, %8 offset:1280\nds_read_b32 %6, %8 offset:1536\nds_read_b32 %7, %8 offset:1792\n"
                         "s_waitcnt lgkmcnt(0)\n" : VREGS : "v"(laddr) : "memory");
        } else if constexpr (BODY == B_DSATOMIC) {
            asm volatile("ds_add_u32 %8, %0\nds_add_u32 %8, %1 offset:256\nds_add_u32 %8, %2 offset:512\nds_add_u32 %8, %3 offset:768\n"
                         "ds_add_u32 %8, %4 offset:1024\nds_add_u32 %8, %5 offset:1280\nds_add_u32 %8, %6 offset:1536\nds_add_u32 %8, %7 offset:1792\n"
                         "s_waitcnt lgkmcnt(0)\n" : VREGS : "v"(laddr) : "memory");
        } else if constexpr (BODY == B_TILEMIX) {
            // the tile kernel's per-wave mix (1369 vector : 774 scalar : 223 branch : 182 LDS), all independent: 12 v, 7 s, 2 (cmp + branch), 2 ds reads
            asm volatile("v_add_u32 %0, %0, 1\ns_add_u32 %8, %8, 1\nv_add_u32 %1, %1, 1\nds_read_b32 %6, %16\nv_add_u32 %2, %2, 1\ns_add_u32 %9, %9, 1\n"
                         "v_and_b32 %3, 0x7fffffff, %3\ns_add_u32 %10, %10, 1\nv_add_u32 %4, %4, 1\ns_cmp_eq_u32 %11, 0\ns_cbranch_scc1 9f\n"
                         "v_add_u32 %5, %5, 1\ns_add_u32 %12, %12, 1\nv_add_u32 %0, %0, 1\nds_read_b32 %7, %16 offset:256\nv_add_u32 %1, %1, 1\ns_add_u32 %13, %13, 1\n"
                         "v_and_b32 %2, 0x7fffffff, %2\ns_add_u32 %14, %14, 1\nv_add_u32 %3, %3, 1\ns_cmp_eq_u32 %15, 0\ns_cbranch_scc1 9f\n"
                         "v_add_u32 %4, %4, 1\nv_add_u32 %5, %5, 1\ns_waitcnt lgkmcnt(0)\n9:\n"
                         : VREGS, SREGS : "v"(laddr) : "scc", "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // (keep every accumulator alive)
    uint32_t keep = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
    if (keep == 0x12345678u) lds[0] = keep;
    if ((threadIdx.x & 63) == 0) {
        uint32_t hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[(blockIdx.x * 256 + threadIdx.x) >> 6] = WaveRec{t0, t1, hwid, xcc};
    }
    if (lds[threadIdx.x] == 0xFFFFFFFFu) out[0].hwid = keep;
}

template <int BODY>
static void run_one(WaveRec *d_out, int n_cu, double mhz)
{
    printf("%-44s", body_name[BODY]);
    for (int per_cu : {1, 2, 4, 8}) {
        const int blocks = n_cu * per_cu, waves = blocks * 4;
        std::vector<WaveRec> h(waves);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_issue<BODY>, dim3(blocks), dim3(256), 0, 0, d_out, 50);     // warm-up
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<BODY>, dim3(blocks), dim3(256), 0, 0, d_out, REPS);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d_out, sizeof(WaveRec) * waves, hipMemcpyDeviceToHost));
        double cyc = 0;
        std::map<uint32_t, int> per_simd;
        for (const WaveRec &w : h) {
            cyc += (double)(w.t1 - w.t0);
            // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (+ XCC id)
            per_simd[(w.xcc & 0xF) << 16 | (w.hwid & 0xFF30u)]++;
        }
        cyc /= waves;
        int lo = 1 << 30, hi = 0;
        for (auto &kv : per_simd) { lo = std::min(lo, kv.second); hi = std::max(hi, kv.second); }
        const Mix m = body_mix[BODY];
        // s_memtime counts at a constant 100 MHz on gfx9-family parts: convert to shader cycles with the kernel's wall time
        const double wall_cyc = ms * 1e-3 * mhz * 1e6;
        const double n = (double)REPS;
        // per SIMD: waves on it * vector instructions per wave / cycles; per CU: 4 * that for scalar
        printf(" | %d/SIMD(%d-%d)", per_cu, lo, hi);
        if (m.v) printf(" V %.3f", per_cu * m.v * n / wall_cyc);
        if (m.s) printf(" S %.3f", 4.0 * per_cu * (m.s + 3) * n / wall_cyc);       // (+ 3: the loop's own s_add / s_cmp / s_cbranch)
        if (m.l) printf(" L %.3f", 4.0 * per_cu * m.l * n / wall_cyc);
        printf(" t%.2f", cyc / wall_cyc);      // mean wave duration in s_memtime ticks over the wall time in clockRate cycles
    }
    printf("\n");
    fflush(stdout);
}

template <int B>
static void run_all(WaveRec *d, int n_cu, double mhz)
{
    if constexpr (B < B_N) { run_one<B>(d, n_cu, mhz); run_all<B + 1>(d, n_cu, mhz); }
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int    n_cu = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    printf("%s: %d CUs, clockRate %.0f MHz.  V = vector instructions per cycle per SIMD, S = scalar-unit (SALU + branch) instructions per cycle per CU,\n"
           "L = LDS instructions per cycle per CU; cycles = kernel wall time x clockRate (an upper bound of the real clock: the values are lower bounds);\n"
           "(lo-hi) = fewest / most waves seen on one SIMD.\n", p.name, n_cu, mhz);
    WaveRec *d;
    CK(hipMalloc(&d, sizeof(WaveRec) * n_cu * 8 * 4));
    run_all<0>(d, n_cu, mhz);
    return 0;
}
