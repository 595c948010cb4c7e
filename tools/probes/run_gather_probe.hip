// Calibration probe for rocprofv3's FETCH_SIZE on the access pattern of k_tl_accumulate_p: every wave load reads 64
// 12-byte pair records = two runs of 32 consecutive records (384 bytes) that start at arbitrary record offsets inside two
// different 48 KiB tile regions, one 12-byte (dwordx3) load per lane.  The kernel reads EXACTLY n_wave_loads * 768 bytes
// of distinct data; FETCH_SIZE of the same launch, divided by that, is the factor tools/pmc_traffic.py applies to the
// record pass (gfx950 counts coalesced streaming reads at half their size -- MI355X_MICROARCH.md -- and this pattern is
// neither fully coalesced nor line-aligned).  Also prints a fully coalesced 12-byte-per-lane stream for comparison.
// Build: hipcc -O3 --offload-arch=gfx950 run_gather_probe.hip -o run_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr unsigned kRegionRecords = 4096;  // 48 KiB per (level, tile) region, as the scatter lays it out
constexpr unsigned kRun = 32;

// runs: one 384-byte run per (region, bin); a wave walks bins of its regions pairwise
__global__ void __launch_bounds__(512) k_run_gather(const unsigned* __restrict__ rec, unsigned n_regions, unsigned bins,
                                                     unsigned* sink) {
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned n_waves = (gridDim.x * blockDim.x) >> 6;
    unsigned acc = 0;
    // wave w takes bin (w % bins) of every region pair: like an accumulate item walking the tiles of its bin
    for (unsigned item = wave; item < bins * 64u; item += n_waves) {
        const unsigned bin = item % bins;
        for (unsigned r0 = (item / bins) * 2u; r0 + 1u < n_regions; r0 += 128u) {
            const unsigned region = r0 + (lane >> 5);
            // the bin's run starts where the previous bins' records end: pseudo-random but fixed, record-aligned
            const unsigned start = (bin * (kRegionRecords / bins) + ((region * 2654435761u) >> 27)) % (kRegionRecords - kRun);
            const unsigned* p = rec + 3u * ((size_t)region * kRegionRecords + start + (lane & 31u));
            acc ^= p[0] ^ p[1] ^ p[2];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void __launch_bounds__(512) k_stream12(const unsigned* __restrict__ rec, size_t n_records, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_records; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned* p = rec + 3u * i;
        acc ^= p[0] ^ p[1] ^ p[2];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void k_fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i * 2654435761u;
}

int main() {
    const unsigned n_regions = 384 * 11, bins = 64;  // the main grid's 11 hashed levels x 384 tiles
    const size_t words = (size_t)n_regions * kRegionRecords * 3;
    unsigned *d, *sink;
    CK(hipMalloc((void**)&d, words * 4));
    CK(hipMalloc((void**)&sink, 64));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, d, words);
    CK(hipDeviceSynchronize());
    // runs actually read: every (region, bin) once
    size_t wave_loads = 0;
    {
        const unsigned n_waves = 512u * 512u / 64u;
        for (unsigned wave = 0; wave < n_waves; ++wave)
            for (unsigned item = wave; item < bins * 64u; item += n_waves)
                for (unsigned r0 = (item / bins) * 2u; r0 + 1u < n_regions; r0 += 128u) ++wave_loads;
    }
    for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(k_run_gather, dim3(512), dim3(512), 0, 0, d, n_regions, bins, sink);
    CK(hipDeviceSynchronize());
    const size_t stream_records = (size_t)64 << 20;  // 768 MB would not fit the buffer: read what the buffer holds
    const size_t n_rec = words / 3 < stream_records ? words / 3 : stream_records;
    for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(k_stream12, dim3(2048), dim3(512), 0, 0, d, n_rec, sink);
    CK(hipDeviceSynchronize());
    printf("k_run_gather: %zu wave loads x 768 B = %.1f KB of records per launch (buffer %.1f MB)\n", wave_loads,
           wave_loads * 768.0 / 1024.0, words * 4.0 / 1e6);
    printf("k_stream12:   %zu records x 12 B = %.1f KB per launch\n", n_rec, n_rec * 12.0 / 1024.0);
    return 0;
}
