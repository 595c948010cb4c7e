// Calibration probe: streaming-read bandwidth of ONE persistent 1024-thread workgroup per CU (the shape of the
// grid-backward accumulate kernels: 128 KiB of LDS accumulators => one workgroup per CU) as a function of bytes per
// lane and loads in flight.  Build: hipcc -O3 --offload-arch=gfx950 read_bw_probe.hip -o read_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename T, int U, int LDS_KB>
__global__ void __launch_bounds__(1024) k_read(const T* __restrict__ p, size_t n, unsigned* sink, size_t chunk) {
    extern __shared__ unsigned char lds[];
    if (LDS_KB && threadIdx.x == 0) lds[0] = 1;
    unsigned acc = 0;
    // each workgroup walks chunks of `chunk` elements, grid-strided (like work items)
    for (size_t c0 = (size_t)blockIdx.x * chunk; c0 < n; c0 += (size_t)gridDim.x * chunk) {
        const size_t end = c0 + chunk < n ? c0 + chunk : n;
        for (size_t i = c0 + threadIdx.x; i < end; i += (size_t)U * 1024) {
            T v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t j = i + (size_t)u * 1024;
                v[u] = p[j < end ? j : end - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned* w = reinterpret_cast<const unsigned*>(&v[u]);
                for (unsigned q = 0; q < sizeof(T) / 4; ++q) acc ^= w[q];
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename T, int U>
void run(const char* tag, const void* d, size_t bytes, unsigned* sink, int blocks, int threads_lds_kb, size_t chunk_bytes) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const size_t n = bytes / sizeof(T);
    auto k = k_read<T, U, 128>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, threads_lds_kb * 1024));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), threads_lds_kb * 1024, 0, (const T*)d, n, sink, chunk_bytes / sizeof(T));
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), threads_lds_kb * 1024, 0, (const T*)d, n, sink, chunk_bytes / sizeof(T));
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s bytes/lane %2zu unroll %2d lds %3d KiB chunk %7zu B blocks %4d: %7.1f us  %6.2f TB/s\n", tag, sizeof(T), U,
           threads_lds_kb, chunk_bytes, blocks, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

__global__ void k_fill(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}

int main() {
    const size_t bytes = (size_t)138 << 20;
    void* d;
    unsigned* sink;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc((void**)&sink, 64));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (uint4*)d, bytes / 16);
    CK(hipDeviceSynchronize());
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d, buffer %zu MB (re-read every launch: Infinity-Cache resident)\n", cus, bytes >> 20);
    for (int lds : {128, 64}) {  // 128 KiB => one workgroup per CU; 64 KiB => two can share a CU
        for (int blocks : {cus, 2 * cus}) {
            for (size_t chunk : {(size_t)196608, (size_t)262144}) {
                run<uint2, 6>("8B", d, bytes, sink, blocks, lds, chunk);
                run<uint2, 12>("8B", d, bytes, sink, blocks, lds, chunk);
                run<uint2, 24>("8B", d, bytes, sink, blocks, lds, chunk);
                run<uint4, 6>("16B", d, bytes, sink, blocks, lds, chunk);
                run<uint4, 12>("16B", d, bytes, sink, blocks, lds, chunk);
                run<uint4, 16>("16B", d, bytes, sink, blocks, lds, chunk);
            }
        }
    }
    return 0;
}
