// Where do the ~20 us go that k_grid_fwd_small spends whatever its batch (round 6)?  Times, for 2 and 256 workgroups of
// 1024 threads with 144 KiB of dynamic LDS: an empty kernel, the LDS staging as a dependent load -> store loop (the first
// form), the staging with all loads of a thread in flight, and the staging from a table every workgroup finds in its L2.
// Build + run: hipcc -O3 --offload-arch=gfx950 tools/probes/launch_probe.hip -o /tmp/launch_probe && /tmp/launch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

__global__ void __launch_bounds__(1024) k_empty(const uint4* src, uint32_t n4, uint32_t* out) {
    if (n4 == 0xFFFFFFFFu) out[0] = lds[threadIdx.x];
}
__global__ void __launch_bounds__(1024) k_stage_loop(const uint4* __restrict__ src, uint32_t n4, uint32_t* out) {
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (uint32_t e = threadIdx.x; e < n4; e += 1024) dst[e] = src[e];
    __syncthreads();
    if (lds[(threadIdx.x * 37u) % (n4 * 4u)] == 0x12345u) out[0] = 1;
}
__global__ void __launch_bounds__(1024) k_stage_flight(const uint4* __restrict__ src, uint32_t n4, uint32_t* out) {
    uint4* dst = reinterpret_cast<uint4*>(lds);
    uint4 st[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) st[k] = src[min(threadIdx.x + (uint32_t)k * 1024u, n4 - 1u)];
#pragma unroll
    for (int k = 0; k < 10; ++k) dst[min(threadIdx.x + (uint32_t)k * 1024u, n4 - 1u)] = st[k];
    __syncthreads();
    if (lds[(threadIdx.x * 37u) % (n4 * 4u)] == 0x12345u) out[0] = 1;
}
// the same loads without LDS: is it the loads or the stores?
__global__ void __launch_bounds__(1024) k_load_only(const uint4* __restrict__ src, uint32_t n4, uint32_t* out) {
    uint4 st[10];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 10; ++k) st[k] = src[min(threadIdx.x + (uint32_t)k * 1024u, n4 - 1u)];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc ^= st[k].x ^ st[k].y ^ st[k].z ^ st[k].w;
    if (acc == 0x12345u) out[0] = 1;
}
// small-LDS variant of the staging (16 KiB: several workgroups per CU possible)
__global__ void __launch_bounds__(1024) k_stage_small(const uint4* __restrict__ src, uint32_t n4, uint32_t* out) {
    uint4* dst = reinterpret_cast<uint4*>(lds);
    dst[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    if (lds[(threadIdx.x * 37u) % 4096u] == 0x12345u) out[0] = 1;
}

template <typename K>
static float time_kernel(K kern, int grid, size_t ldsb, const uint4* src, uint32_t n4, uint32_t* out, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), ldsb, 0, src, n4, out);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), ldsb, 0, src, n4, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    const uint32_t n4 = 9216;  // 144 KiB
    const size_t ldsb = (size_t)n4 * 16;
    uint4* src;
    uint32_t* out;
    CK(hipMalloc(&src, ldsb));
    CK(hipMemset(src, 1, ldsb));
    CK(hipMalloc(&out, 64));
    CK(hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    CK(hipFuncSetAttribute((const void*)k_stage_loop, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    CK(hipFuncSetAttribute((const void*)k_stage_flight, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    // a 256 MiB buffer to sweep between launches would model a cold L2; here the table stays hot (best case)
    for (int grid : {1, 2, 8, 64, 256, 512}) {
        printf("grid %4d: empty(144K LDS) %6.2f us | empty(0 LDS) %6.2f | stage loop %6.2f | stage in-flight %6.2f | loads only %6.2f | stage 16K %6.2f\n", grid,
               time_kernel(k_empty, grid, ldsb, src, n4, out, 200), time_kernel(k_empty, grid, 0, src, n4, out, 200),
               time_kernel(k_stage_loop, grid, ldsb, src, n4, out, 200), time_kernel(k_stage_flight, grid, ldsb, src, n4, out, 200),
               time_kernel(k_load_only, grid, 0, src, n4, out, 200), time_kernel(k_stage_small, grid, 16384, src, n4, out, 200));
    }
    return 0;
}
