// Developer aid (GPU box): what do device-scope atomics on ONE word cost when many waves of the whole chip issue them (the hot parent words of
// k_resolve / k_reduce: the background node of a plane)?   hipcc --offload-arch=gfx950 -O2 tools/atomic_hot.hip -o /tmp/ah && /tmp/ah
// One lane per wave issues `per` returning 64-bit adds + 4 returning 32-bit min / max on words picked from `nwords` hot records (32 bytes apart).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_hot(uint32_t *rec, uint32_t nwords, int per, int lanes, uint32_t *sink)
{
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if ((int)lane >= lanes) return;
    uint32_t acc = 0, x = wave * 2654435761u + lane * 40503u;
    for (int i = 0; i < per; ++i) {
        x = x * 1664525u + 1013904223u;
        uint32_t *r = rec + 8u * ((x >> 10) % nwords);
        const unsigned long long cn = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(r + 2), 1ull | (1ull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc |= (uint32_t)cn;
        acc |= __hip_atomic_fetch_min(r + 4, x & 1023u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc |= __hip_atomic_fetch_min(r + 5, x & 1023u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc |= __hip_atomic_fetch_max(r + 6, x & 1023u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc |= __hip_atomic_fetch_max(r + 7, x & 1023u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main()
{
    uint32_t *d, *sink;
    CK(hipMalloc(&d, 32u << 20)); CK(hipMemset(d, 0, 32u << 20)); CK(hipMalloc(&sink, 4));
    for (int lanes : {1, 64})
        for (uint32_t nwords : {1u, 4u, 64u, 4096u, 1u << 20})
            for (int blocks : {256, 4096}) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                const int per = 4;
                hipLaunchKernelGGL(k_hot, dim3(blocks), dim3(256), 0, 0, d, nwords, per, lanes, sink);
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_hot, dim3(blocks), dim3(256), 0, 0, d, nwords, per, lanes, sink);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double pushes = (double)blocks * 4 * lanes * per;
                printf("lanes/wave %2d, %7u hot records, %5d blocks: %8.0f pushes (5 atomics each) in %7.3f ms = %7.1f ns per push, %.2f G atomics/s\n", lanes, nwords, blocks, pushes, ms,
                       ms * 1e6 / pushes, pushes * 5 / (ms * 1e6));
            }
    return 0;
}
