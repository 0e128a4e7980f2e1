// Developer aid (GPU box): what does a global atomic cost by memory scope?   hipcc --offload-arch=gfx950 -O2 tools/atomic_scope.hip -o /tmp/as && /tmp/as
// The global passes of the component tree (k_seam / k_resolve / k_reduce) are bound by device-scope atomics -- the per-XCD L2s are not coherent, so an
// agent-scope atomic is carried out beyond the L2.  If everything that touches a plane's records ran on ONE XCD, workgroup-scope atomics (carried out
// in that XCD's L2) would do.  Measured here: (a) latency of a chain of dependent returning atomics, one lane; (b) throughput of independent atomics
// from every CU of one XCD (blocks b with b % 8 == 0 work, the others leave) on 4096 hot words / on distinct words.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SCOPE>
__global__ void k_chain(uint32_t *w, int n, unsigned long long *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t idx = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) idx = (__hip_atomic_fetch_add(&w[idx & 4095u], 1u, __ATOMIC_RELAXED, SCOPE) * 2654435761u) >> 20;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[0] = t1 - t0; out[1] = idx;
}

template <int SCOPE, bool RET>
__global__ __launch_bounds__(256) void k_tput(uint32_t *w, int n, uint32_t mask, uint32_t *sink)
{
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((xcc & 0xF) != 0) return;                        // one XCD only
    uint32_t x = blockIdx.x * 256 + threadIdx.x, acc = 0;
    for (int i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        if (RET) acc += __hip_atomic_fetch_add(&w[(x >> 8) & mask], 1u, __ATOMIC_RELAXED, SCOPE);
        else __hip_atomic_fetch_add(&w[(x >> 8) & mask], 1u, __ATOMIC_RELAXED, SCOPE);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int SCOPE>
static void run(const char *name, uint32_t *d, unsigned long long *dout, uint32_t *sink)
{
    unsigned long long h[2];
    hipLaunchKernelGGL(k_chain<SCOPE>, dim3(1), dim3(64), 0, 0, d, 2000, dout);
    hipLaunchKernelGGL(k_chain<SCOPE>, dim3(1), dim3(64), 0, 0, d, 20000, dout);
    CK(hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost));
    printf("%-10s chain of 20000 dependent returning atomics: %.0f ticks each (s_memtime, 100 MHz: x10 ns)\n", name, (double)h[0] / 20000);
    for (int ret = 0; ret < 2; ++ret)
        for (uint32_t mask : {4095u, (1u << 24) - 1u}) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int blocks = 2048, n = 2000;
            if (ret) hipLaunchKernelGGL((k_tput<SCOPE, true>), dim3(blocks), dim3(256), 0, 0, d, 10, mask, sink);
            else hipLaunchKernelGGL((k_tput<SCOPE, false>), dim3(blocks), dim3(256), 0, 0, d, 10, mask, sink);
            CK(hipEventRecord(e0));
            if (ret) hipLaunchKernelGGL((k_tput<SCOPE, true>), dim3(blocks), dim3(256), 0, 0, d, n, mask, sink);
            else hipLaunchKernelGGL((k_tput<SCOPE, false>), dim3(blocks), dim3(256), 0, 0, d, n, mask, sink);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-10s %s atomics from one XCD (256 of %d blocks work), %s words: %.2f G/s\n", name, ret ? "returning" : "fire-and-forget", blocks,
                   mask == 4095u ? "4096 hot" : "16 M", (double)(blocks / 8) * 256 * n / (ms * 1e6));
        }
}

int main()
{
    uint32_t *d, *sink; unsigned long long *dout;
    CK(hipMalloc(&d, sizeof(uint32_t) << 24)); CK(hipMemset(d, 0, sizeof(uint32_t) << 24));
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&dout, 64));
    run<__HIP_MEMORY_SCOPE_AGENT>("agent", d, dout, sink);
    run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup", d, dout, sink);
    run<__HIP_MEMORY_SCOPE_WAVEFRONT>("wavefront", d, dout, sink);
    return 0;
}
