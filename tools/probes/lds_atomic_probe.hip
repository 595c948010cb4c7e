// Microbenchmark: throughput of LDS atomic adds (f32 / u32 / u64 / pk_f16) with random addresses and
// a given fraction of active lanes.  hipcc --offload-arch=gfx950 -O3 lds_atomic_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdint.h>

constexpr int kEntries = 32768;  // dwords (128 KiB)

template <int MODE>
__global__ void __launch_bounds__(1024) k(const uint32_t* __restrict__ idx, int iters, int active_of_64, float* out) {
    extern __shared__ uint32_t lds[];
    for (int e = threadIdx.x; e < kEntries; e += 1024) lds[e] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = lane < active_of_64;
    uint32_t h = idx[blockIdx.x * 1024 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const uint32_t a = (h >> 8) % (kEntries / 2);
        if (on) {
            if (MODE == 0) atomicAdd((float*)&lds[2 * a], 1.0f);
            if (MODE == 1) atomicAdd(&lds[2 * a], 1u);
            if (MODE == 2) atomicAdd((unsigned long long*)&lds[2 * a], 1ull);
            if (MODE == 3) { atomicAdd((float*)&lds[2 * a], 1.0f); atomicAdd((float*)&lds[2 * a + 1], 2.0f); }
            if (MODE == 4) lds[2 * a] = h;  // plain store for reference
            if (MODE == 5) unsafeAtomicAdd((double*)&lds[2 * a], 1.0);
            if (MODE == 6) { unsafeAtomicAdd((float*)&lds[2 * a], 1.0f); }
        }
    }
    __syncthreads();
    float s = 0;
    for (int e = threadIdx.x; e < kEntries; e += 1024) s += (float)lds[e];
    if (s == 12345.f) out[0] = s;
}

template <int MODE>
void run(const char* name, const uint32_t* d_idx, float* d_out, int active) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kEntries * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 1024, kEntries * 4>>>(d_idx, 10, active, d_out);
    hipEventRecord(a);
    k<MODE><<<blocks, 1024, kEntries * 4>>>(d_idx, iters, active, d_out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)iters * 16;             // per CU (16 waves)
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-22s active %2d/64: %8.3f ms  -> %7.1f cycles per wave-instruction per CU, %6.2f lane-ops/clk/CU\n", name, active, ms,
           cyc / wave_instr, wave_instr * active * (MODE == 3 ? 2 : 1) / cyc);
}

int main() {
    uint32_t* d_idx; float* d_out;
    hipMalloc(&d_idx, 256 * 1024 * 4); hipMalloc(&d_out, 4);
    uint32_t* h = (uint32_t*)malloc(256 * 1024 * 4);
    for (int i = 0; i < 256 * 1024; ++i) h[i] = (uint32_t)rand() * 2654435761u + i;
    hipMemcpy(d_idx, h, 256 * 1024 * 4, hipMemcpyHostToDevice);
    for (int active : {64, 16, 4, 1}) {
        run<4>("ds_write_b32 (ref)", d_idx, d_out, active);
        run<0>("ds_add_f32", d_idx, d_out, active);
        run<1>("ds_add_u32", d_idx, d_out, active);
        run<2>("ds_add_u64", d_idx, d_out, active);
        run<3>("2x ds_add_f32 (pair)", d_idx, d_out, active);
        run<5>("ds_add_f64", d_idx, d_out, active);
        run<6>("ds_add_f32 (unsafe)", d_idx, d_out, active);
    }
    return 0;
}
