// Multi-resolution hash-grid encoding for gfx950: forward gather, parameter-gradient scatter and
// input-gradient kernels.  Replaces tiny-cuda-nn's kernel_grid / kernel_grid_backward /
// kernel_grid_backward_input (upstream grid.h is not vendored in
// /root/reference, the arithmetic is restated in oracle/grid.py + oracle/c/nvo_oracle.c).
//
// MI355X design notes
//  * One thread per (sample, level).  The 1-D grid is mapped so that workgroups that the
//    dispatcher deals to the same XCD (blockIdx % 8, observed round-robin -- speed only, never
//    correctness) work on the same pair of levels {xcd, xcd+8}: a hashed level's fp16 table is
//    2 MiB, so one coarse + one fine level stay resident in that XCD's private 4 MiB L2 instead
//    of the whole 24 MiB table thrashing every L2.
//  * Encoded features leave the kernel level-major ("SoA": [L][N] half2) so that a wave's store is
//    256 contiguous bytes; the fused MLP consumes that layout directly.  The tcnn-API path can ask
//    for sample-major ("AoS": [N][L] half2) instead.
//  * Interpolation accumulates in fp32 and rounds once to fp16 (tcnn accumulates in fp16).
//  * Parameter gradients accumulate in fp32 (tcnn: fp16 atomics for F=2).  Two kernels:
//      - k_grid_bwd_atomic: global float atomics, lane pairs adjacent on one corner (8 B) so a
//        wave instruction touches 32 distinct 64-B lines, not 64.
//      - k_grid_bwd_lds:    "slice owner" scatter -- each workgroup owns a slice of one level's
//        table in LDS (up to all 160 KiB of the CU), re-derives every sample's corner indices,
//        accumulates hits with LDS atomics (64-bit fixed point where slices are hit often: LDS
//        integer atomics run 14x the rate of LDS float atomics on gfx950) and writes the slice back with plain
//        coalesced stores.  Global atomics on random rows run at ~0.08 TB/s on MI355X
//        (MI355X_MICROARCH.md, Global float atomics); this path uses none.
#include "nvo_kernels.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

uint32_t nvo_grid_levels_init(NvoGridLevels* g, uint32_t n_levels, uint32_t n_features,
                              uint32_t log2_hashmap_size, uint32_t base_resolution,
                              float per_level_scale) {
    g->n_levels = n_levels;
    g->n_features = n_features;
    const float log2_pls = log2f(per_level_scale);
    uint32_t offset = 0;
    for (uint32_t i = 0; i < n_levels; ++i) {
        const float scale = exp2f((float)i * log2_pls) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2u;
        uint32_t params_in_level =
            powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
        params_in_level = (uint32_t)nvo_round_up(params_in_level, 8u);
        const uint32_t hash_cap = 1u << log2_hashmap_size;
        if (params_in_level > hash_cap) params_in_level = hash_cap;
        g->scale[i] = scale;
        g->resolution[i] = res;
        g->offset[i] = offset;
        g->hashed[i] = ((double)res * (double)res * (double)res > (double)params_in_level) ? 1u : 0u;
        offset += params_in_level;
    }
    for (uint32_t i = n_levels; i <= NVO_MAX_LEVELS; ++i) g->offset[i] = offset;
    for (uint32_t i = n_levels; i < NVO_MAX_LEVELS; ++i) {
        g->scale[i] = 0.f;
        g->resolution[i] = 0;
        g->hashed[i] = 0;
    }
    return offset;
}

// -DNVO_GRID_PHASE (debugging aid, never in the product build; NVO_EXTRA_CXXFLAGS of nerf_vo_amd/build.py,
// tools/grid_phase.py): shader-clock sums per phase of the record pass, the slice-owner items and the small-grid
// forward -- where DESIGN.md section 3.6's "what is this kernel waiting for" figures come from.  One thread per
// workgroup adds its phase durations with atomics (which disturb the timing of the instrumented launch itself: read the
// SHARES, not the totals).  Slots: [0..6] k_tl_accumulate_p (zero+L1, records, barrier, flush, end barrier, hashed items,
// dense items); [8..14] k_tl_scatter_p (loads+max, index+rank, barrier, bin scan, stage, copy-out, workgroups);
// [16..20] slice-owner dense items, [24..28] hashed items (zero, scan, barrier, flush, items); [32..37] k_grid_fwd_small
// (staging, loop, issue, LDS levels, consume, workgroups).
#ifdef NVO_GRID_PHASE
__device__ unsigned long long nvo_grid_phase_cycles[48];
#define GP_CLK(v) const unsigned long long v = __builtin_readcyclecounter()
#define GP_ADD(slot, d) atomicAdd(&nvo_grid_phase_cycles[slot], (unsigned long long)(d))
#else
#define GP_CLK(v) do { } while (0)
#define GP_ADD(slot, d) do { } while (0)
#endif

namespace {

constexpr int kGridBlock = 256;

// Two dwords at DWORD alignment (the x neighbours of a dense level start at any entry): loaded through a packed type, so
// that the 8-byte access is well defined at 4-byte alignment -- it still compiles to one global_load_dwordx2 (unaligned
// access mode is on for this target) -- instead of a uint2 reinterpretation that promises the compiler 8-byte alignment.
struct __attribute__((packed, aligned(4))) NvoU2A4 {
    uint32_t x, y;
};
__device__ __forceinline__ uint2 nvo_ld_u2_a4(const uint32_t* p) {
    const NvoU2A4 v = *reinterpret_cast<const NvoU2A4*>(p);
    return make_uint2(v.x, v.y);
}

// blockIdx -> (tile, level).  With n_levels a multiple of 8 the blocks that share blockIdx % 8
// (one XCD under round-robin placement) get levels {xcd, xcd + 8, ...}.
__device__ __forceinline__ void grid_block_map(uint32_t bid, uint32_t n_levels, uint32_t* tile,
                                               uint32_t* level) {
    if ((n_levels & 7u) == 0u) {
        const uint32_t xcd = bid & 7u;
        const uint32_t q = bid >> 3;
        const uint32_t per_xcd = n_levels >> 3;
        *level = xcd + 8u * (q % per_xcd);
        *tile = q / per_xcd;
    } else {
        *level = bid % n_levels;
        *tile = bid / n_levels;
    }
}

// XCD-BALANCED block map of the forward (round 3).  grid_block_map gives XCD x the levels {x, x + 8}: with 5 dense and 11
// hashed levels, XCDs 5-7 gather from TWO 2 MiB hashed tables while XCDs 0-4 have one hashed and one (cheap) dense level
// -- the busy three run at their L2-gather ceiling (2.9 TB/s each) and the others idle almost half the launch.  The plan
// below keeps what the pinning buys (an XCD's private 4 MiB L2 holds at most TWO hashed tables) and evens out the work:
// XCD x owns one hashed level completely, the remaining hashed levels are cut into tile ranges so that every XCD gets a
// piece of exactly ONE further table, and the dense levels' tiles fill the XCDs up to the same cost.  Placement of
// block b on XCD b % 8 is an observation, not a contract: it affects speed only.
constexpr int kPlanUnits = 8;
struct GridFwdPlan {
    uint32_t enabled;
    uint32_t n_units[8];
    uint32_t level[8][kPlanUnits];
    uint32_t tile0[8][kPlanUnits];
    uint32_t n_tiles[8][kPlanUnits];
};

__device__ __forceinline__ bool grid_plan_map(const GridFwdPlan& p, uint32_t bid, uint32_t* tile, uint32_t* level) {
    const uint32_t xcd = bid & 7u;
    uint32_t q = bid >> 3;
    for (uint32_t u = 0; u < p.n_units[xcd]; ++u) {
        if (q < p.n_tiles[xcd][u]) {
            *tile = p.tile0[xcd][u] + q;
            *level = p.level[xcd][u];
            return true;
        }
        q -= p.n_tiles[xcd][u];
    }
    return false;
}

// dL/d(encoded) pair of one (sample, level): fp16 (tcnn's precision), fp32, or bfloat16 (bf16 MLP mode)
struct Bf2 {
    uint32_t raw;
};
__device__ __forceinline__ float2 dy2f(__half2 v) { return __half22float2(v); }
__device__ __forceinline__ float2 dy2f(float2 v) { return v; }
__device__ __forceinline__ float2 dy2f(Bf2 v) {
    return make_float2(__uint_as_float(v.raw << 16), __uint_as_float(v.raw & 0xFFFF0000u));
}

struct Corner {
    uint32_t px, py, pz;  // cell base
    float wx, wy, wz;     // fractional position
};

__device__ __forceinline__ Corner grid_cell(float scale, float x, float y, float z) {
    // tcnn pos_fract: pos = fma(scale, x, 0.5); cell = floor(pos); frac = pos - cell.
    Corner c;
    float fx = fmaf(scale, x, 0.5f), fy = fmaf(scale, y, 0.5f), fz = fmaf(scale, z, 0.5f);
    float tx = floorf(fx), ty = floorf(fy), tz = floorf(fz);
    c.px = (uint32_t)(int)tx;
    c.py = (uint32_t)(int)ty;
    c.pz = (uint32_t)(int)tz;
    c.wx = fx - tx;
    c.wy = fy - ty;
    c.wz = fz - tz;
    return c;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// x      : [N][3] f32 (row-major), values expected in [0,1]
// table  : fp16 parameters, [entries][2]
// out    : SOA ? [L][N] half2 : [N][L] half2
// indices: optional debug/parity output, [L][N][8] uint32 (nullptr in production)
// dydx   : (DYDX) d(out)/d(cell coordinate) per level and axis, [L][3][N] half2 (feature pair) -- what tcnn's
//          forward stores when input gradients were requested (prepare_input_gradients): the backward w.r.t. the
//          input then is a coalesced stream instead of a second pass of 8 gathers per (sample, level).  Kept in
//          CELL units (the level's scale is applied by the consumer) so that it has the magnitude of a table
//          difference and fits fp16 like the table itself.
// SPT samples per thread (tile = SPT * kGridBlock samples, sample s of a thread is kGridBlock apart from sample s - 1:
// coalescing as before): the kernel is two dependent memory round trips per (sample, level) -- position, then the
// corner gathers -- and at full occupancy its time is (rounds of resident workgroups) x (that latency); with the gathers
// of SPT samples in flight per thread the rounds shrink by SPT.
template <bool SOA, bool DYDX, int SPT>
__global__ void __launch_bounds__(kGridBlock)
k_grid_fwd(NvoGridLevels g, uint32_t N, const float* __restrict__ x,
           const __half2* __restrict__ table, __half2* __restrict__ out,
           uint32_t* __restrict__ indices, __half2* __restrict__ dydx, int out_bf16, GridFwdPlan plan,
           const uint32_t* __restrict__ n_live) {
    uint32_t tile, level;
    if (plan.enabled) {
        if (!grid_plan_map(plan, blockIdx.x, &tile, &level)) return;
    } else {
        grid_block_map(blockIdx.x, g.n_levels, &tile, &level);
    }
    // (rows in use known on the device only: tiles past them leave at once; N stays the stride of the level-major output)
    if (n_live && tile * (kGridBlock * SPT) >= *n_live) return;
    const uint32_t i_first = tile * (kGridBlock * SPT) + threadIdx.x;
    if (i_first >= N) return;

    const uint32_t off = g.offset[level];
    const uint32_t size = g.offset[level + 1] - off;
    const uint32_t res = g.resolution[level];
    const uint32_t hashed = g.hashed[level];
    const __half2* __restrict__ tab = table + off;
    const float scale = g.scale[level];

    // every load is unconditional (a sample past the end reads sample N - 1 again and stores nothing): loads behind a
    // divergent branch would make the join wait for everything in flight
    uint32_t i[SPT];
    float px[SPT][3];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        i[s] = i_first + (uint32_t)s * kGridBlock;
        const uint32_t ic = min(i[s], N - 1u);
#pragma unroll
        for (int k = 0; k < 3; ++k) px[s][k] = x[3 * (size_t)ic + k];
    }
    Corner c[SPT];
    uint32_t idx[SPT][8];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        c[s] = grid_cell(scale, px[s][0], px[s][1], px[s][2]);
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            idx[s][k] = nvo_grid_index(hashed, size, res, c[s].px + (k & 1u), c[s].py + ((k >> 1) & 1u),
                                       c[s].pz + ((k >> 2) & 1u));
    }
    __half2 v[SPT][8];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        if (!hashed) {
            // dense level: the two x corners of a (y, z) pair are neighbours in memory -- one 8-byte load instead of two
            // 4-byte gathers (dword alignment suffices); a pair that straddles the table's wrap takes two loads
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                if (idx[s][2 * j + 1] == idx[s][2 * j] + 1u) {
                    const uint2 q = nvo_ld_u2_a4(reinterpret_cast<const uint32_t*>(tab) + idx[s][2 * j]);
                    v[s][2 * j] = __builtin_bit_cast(__half2, q.x);
                    v[s][2 * j + 1] = __builtin_bit_cast(__half2, q.y);
                } else {
                    v[s][2 * j] = tab[idx[s][2 * j]];
                    v[s][2 * j + 1] = tab[idx[s][2 * j + 1]];
                }
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) v[s][k] = tab[idx[s][k]];
        }
    }
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        if (i[s] >= N) continue;
        const Corner& cc = c[s];
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const float w = ((k & 1u) ? cc.wx : 1.f - cc.wx) * ((k & 2u) ? cc.wy : 1.f - cc.wy) *
                            ((k & 4u) ? cc.wz : 1.f - cc.wz);
            const float2 f = __half22float2(v[s][k]);
            r0 = fmaf(w, f.x, r0);
            r1 = fmaf(w, f.y, r1);
        }
        // fp32 interpolation, ONE rounding to the network's input format (fp16, or bfloat16 in the bf16 MLP mode)
        const uint32_t r = nvo_cvt16x2(r0, r1, out_bf16 != 0);
        uint32_t* __restrict__ o32 = reinterpret_cast<uint32_t*>(out);
        if (SOA) {
            o32[(size_t)level * N + i[s]] = r;
        } else {
            o32[(size_t)i[s] * g.n_levels + level] = r;
        }
        if constexpr (DYDX) {
            float2 f[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) f[k] = __half22float2(v[s][k]);
            const float wx0 = 1.f - cc.wx, wy0 = 1.f - cc.wy, wz0 = 1.f - cc.wz;
            // d/d(axis): (corner with the axis bit) - (corner without), weighted by the other two axes
            const float ax[4] = {wy0 * wz0, cc.wy * wz0, wy0 * cc.wz, cc.wy * cc.wz};
            const float ay[4] = {wx0 * wz0, cc.wx * wz0, wx0 * cc.wz, cc.wx * cc.wz};
            const float az[4] = {wx0 * wy0, cc.wx * wy0, wx0 * cc.wy, cc.wx * cc.wy};
            const float gx0 = ax[0] * (f[1].x - f[0].x) + ax[1] * (f[3].x - f[2].x) + ax[2] * (f[5].x - f[4].x) + ax[3] * (f[7].x - f[6].x);
            const float gx1 = ax[0] * (f[1].y - f[0].y) + ax[1] * (f[3].y - f[2].y) + ax[2] * (f[5].y - f[4].y) + ax[3] * (f[7].y - f[6].y);
            const float gy0 = ay[0] * (f[2].x - f[0].x) + ay[1] * (f[3].x - f[1].x) + ay[2] * (f[6].x - f[4].x) + ay[3] * (f[7].x - f[5].x);
            const float gy1 = ay[0] * (f[2].y - f[0].y) + ay[1] * (f[3].y - f[1].y) + ay[2] * (f[6].y - f[4].y) + ay[3] * (f[7].y - f[5].y);
            const float gz0 = az[0] * (f[4].x - f[0].x) + az[1] * (f[5].x - f[1].x) + az[2] * (f[6].x - f[2].x) + az[3] * (f[7].x - f[3].x);
            const float gz1 = az[0] * (f[4].y - f[0].y) + az[1] * (f[5].y - f[1].y) + az[2] * (f[6].y - f[2].y) + az[3] * (f[7].y - f[3].y);
            __half2* __restrict__ o = dydx + (size_t)level * 3 * N + i[s];
            o[0] = __floats2half2_rn(gx0, gx1);
            o[N] = __floats2half2_rn(gy0, gy1);
            o[2 * (size_t)N] = __floats2half2_rn(gz0, gz1);
        }
        if (indices) {
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) indices[((size_t)level * N + i[s]) * 8 + k] = idx[s][k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward over RUNS of four consecutive samples (level-major output; option "grid_fwd_runs")
// ------------------------------------------------------------------------------------------
// Samples arrive in ray order, and a TRAINED field concentrates a ray's samples at the surface: on every level whose
// cell is wider than their spacing, neighbours in memory fall into the same cell and repeat the gather instructions of
// the sample before.  A thread here walks four consecutive samples of one level and gathers only where the cell changes;
// positions are 3 x 16-byte loads per thread, the four results leave as one 16-byte store.  Pays where runs exist --
// full-image inference of a trained field: grid_fwd[L16] 301 -> 253 us per 32 768-ray chunk -- and costs 8 % where they
// do not (uniform samples of an untrained field: the 48-byte position stride, fewer waves per sample), so it is an
// option the inference path switches on (EXPERIMENTS.md 9.6b).  Same fp32 interpolation, same order, one rounding:
// bit-identical to k_grid_fwd.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_grid_fwd_runs(NvoGridLevels g, uint32_t N, const float* __restrict__ x, const __half2* __restrict__ table,
                __half2* __restrict__ out, int out_bf16, GridFwdPlan plan, const uint32_t* __restrict__ n_live) {
    constexpr int RUN = 4;
    uint32_t tile, level;
    if (plan.enabled) {
        if (!grid_plan_map(plan, blockIdx.x, &tile, &level)) return;
    } else {
        grid_block_map(blockIdx.x, g.n_levels, &tile, &level);
    }
    if (n_live && tile * (BLOCK * RUN) >= *n_live) return;
    const uint32_t i0 = tile * (BLOCK * RUN) + threadIdx.x * RUN;
    if (i0 >= N) return;  // (N is a multiple of RUN: a run is in range as a whole)

    const uint32_t off = g.offset[level];
    const uint32_t size = g.offset[level + 1] - off;
    const uint32_t res = g.resolution[level];
    const uint32_t hashed = g.hashed[level];
    const uint32_t* __restrict__ tab = reinterpret_cast<const uint32_t*>(table) + off;
    const float scale = g.scale[level];

    float p[RUN * 3];
    {
        const float4* __restrict__ xp = reinterpret_cast<const float4*>(x + 3 * (size_t)i0);
#pragma unroll
        for (int q = 0; q < RUN * 3 / 4; ++q) {
            const float4 f = xp[q];
            p[4 * q] = f.x; p[4 * q + 1] = f.y; p[4 * q + 2] = f.z; p[4 * q + 3] = f.w;
        }
    }
    Corner c[RUN];
    bool need[RUN];
#pragma unroll
    for (int s = 0; s < RUN; ++s) {
        c[s] = grid_cell(scale, p[3 * s], p[3 * s + 1], p[3 * s + 2]);
        need[s] = s == 0 || c[s].px != c[s - 1].px || c[s].py != c[s - 1].py || c[s].pz != c[s - 1].pz;
    }
    // every gather of the run is requested before anything is consumed
    uint32_t v[RUN][8];
#pragma unroll
    for (int s = 0; s < RUN; ++s) {
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) v[s][k] = 0u;
        if (!need[s]) continue;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t cy = c[s].py + (j & 1u), cz = c[s].pz + (j >> 1);
            const uint32_t a = nvo_grid_index(hashed, size, res, c[s].px, cy, cz);
            const uint32_t b = nvo_grid_index(hashed, size, res, c[s].px + 1u, cy, cz);
            if (!hashed && b == a + 1u) {
                // dense level: the two x corners are neighbours in memory (dword alignment suffices)
                const uint2 q = nvo_ld_u2_a4(tab + a);
                v[s][2 * j] = q.x;
                v[s][2 * j + 1] = q.y;
            } else {
                v[s][2 * j] = tab[a];
                v[s][2 * j + 1] = tab[b];
            }
        }
    }
    uint32_t cur[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t r[RUN];
#pragma unroll
    for (int s = 0; s < RUN; ++s) {
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) cur[k] = need[s] ? v[s][k] : cur[k];
        const Corner& cc = c[s];
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const float w = ((k & 1u) ? cc.wx : 1.f - cc.wx) * ((k & 2u) ? cc.wy : 1.f - cc.wy) *
                            ((k & 4u) ? cc.wz : 1.f - cc.wz);
            const float2 f = __half22float2(__builtin_bit_cast(__half2, cur[k]));
            r0 = fmaf(w, f.x, r0);
            r1 = fmaf(w, f.y, r1);
        }
        r[s] = nvo_cvt16x2(r0, r1, out_bf16 != 0);
    }
    uint32_t* __restrict__ o32 = reinterpret_cast<uint32_t*>(out);
    *reinterpret_cast<uint4*>(o32 + (size_t)level * N + i0) = make_uint4(r[0], r[1], r[2], r[3]);
}

// ------------------------------------------------------------------------------------------
// forward for SMALL grids (the proposal networks: 5 levels, 2^17-entry tables): coarse dense levels served from LDS
// ------------------------------------------------------------------------------------------
// k_grid_fwd on a proposal grid runs at the L1's tag-lookup rate, not at any bandwidth: 27 M line accesses per 1 M-sample
// launch, 90 % of them hits (profiles/r2_pmc_grid_fwd.txt) -- a random gather costs the vector L1 one tag look-up per
// DISTINCT LINE of the instruction whatever it hits (lanes in one line are one look-up: EXPERIMENTS.md 9.6b).  The
// two coarsest levels (16 KiB + 79 / 128 KiB) fit a CU's 160 KiB of LDS, whose banked gather costs ~8 cycles per wave
// instruction instead of 64.  One 1024-thread workgroup per CU stages them once and
// walks its share of the samples with a thread per SAMPLE (all levels: the position is loaded once instead of once per
// level, and a lane keeps 16-20 global gathers of the remaining levels in flight); on the hashed levels the two x
// corners of a (y, z) pair are ONE aligned 8-byte load whenever the cell's x index is even (then idx1 == idx0 ^ 1).
// L1 accesses per sample: 28 -> 16 (levels dense/dense/dense/hash/hash), 32 -> 18 (dense/dense/hash/hash/hash).
// Same fp32 interpolation, same order, one rounding: bit-identical to k_grid_fwd.
constexpr int kSmallBlock = 1024;

template <int NLDS, int NG, int SPT>
__global__ void __launch_bounds__(kSmallBlock)
k_grid_fwd_small(NvoGridLevels g, uint32_t N, const float* __restrict__ x, const __half2* __restrict__ table,
                 __half2* __restrict__ out, int out_bf16, uint32_t per_block) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_tab[];
    const uint32_t* __restrict__ tab32 = reinterpret_cast<const uint32_t*>(table);
    GP_CLK(gf0);
#ifdef NVO_GRID_PHASE
    unsigned long long gf_iss = 0, gf_lds = 0, gf_cons = 0;
#endif
    {   // stage the leading NLDS levels (contiguous from entry 0; level offsets are multiples of 8 entries)
        const uint32_t n4 = g.offset[NLDS] >> 2;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(table);
        uint4* dst = reinterpret_cast<uint4*>(lds_tab);
        for (uint32_t e = threadIdx.x; e < n4; e += kSmallBlock) dst[e] = src[e];
    }
    __syncthreads();
    GP_CLK(gf1);
    const uint32_t first = blockIdx.x * per_block;
    const uint32_t last = min(N, first + per_block);
    uint32_t* __restrict__ o32 = reinterpret_cast<uint32_t*>(out);
    // SPT samples per thread and pass (kSmallBlock apart: coalescing as with one): with one 1024-thread workgroup per CU
    // the launch is latency-bound (PMC: waves wait 59 % of their cycles) -- twice the gathers in flight per wave
    for (uint32_t i0 = first + threadIdx.x; i0 < last; i0 += kSmallBlock * SPT) {
        uint32_t is[SPT];
        float pos[SPT][3];
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            is[s] = i0 + (uint32_t)s * kSmallBlock;
            const uint32_t ic = min(is[s], last - 1u);  // (unconditional loads; a sample past the end stores nothing)
#pragma unroll
            for (int k = 0; k < 3; ++k) pos[s][k] = x[3 * (size_t)ic + k];
        }
        GP_CLK(gfa);
        // ---- global levels first: every gather of the samples is requested before anything is consumed
        Corner cg[SPT][NG];
        uint32_t vg[SPT][NG][8];
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const uint32_t level = NLDS + q;
                const uint32_t off = g.offset[level], size = g.offset[level + 1] - off, res = g.resolution[level];
                const uint32_t hashed = g.hashed[level];
                const uint32_t* __restrict__ tl = tab32 + off;
                cg[s][q] = grid_cell(g.scale[level], pos[s][0], pos[s][1], pos[s][2]);
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t cy = cg[s][q].py + (j & 1u), cz = cg[s][q].pz + (j >> 1);
                    const uint32_t i0c = nvo_grid_index(hashed, size, res, cg[s][q].px, cy, cz);
                    const uint32_t i1c = nvo_grid_index(hashed, size, res, cg[s][q].px + 1u, cy, cz);
                    if (hashed ? ((cg[s][q].px & 1u) == 0u) : false) {
                        // even cell x: idx1 == idx0 ^ 1 -- both corners sit in one aligned 8-byte pair
                        const uint2 pr = *reinterpret_cast<const uint2*>(tl + (i0c & ~1u));
                        vg[s][q][2 * j] = (i0c & 1u) ? pr.y : pr.x;
                        vg[s][q][2 * j + 1] = (i0c & 1u) ? pr.x : pr.y;
                    } else if (!hashed && i1c == i0c + 1u) {
                        // dense level: x neighbours are neighbours in memory (dword alignment suffices)
                        const uint2 pr = nvo_ld_u2_a4(tl + i0c);
                        vg[s][q][2 * j] = pr.x;
                        vg[s][q][2 * j + 1] = pr.y;
                    } else {
                        vg[s][q][2 * j] = tl[i0c];
                        vg[s][q][2 * j + 1] = tl[i1c];
                    }
                }
            }
        }
        GP_CLK(gfb);
        // ---- LDS levels while the gathers fly
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            if (is[s] >= last) continue;
#pragma unroll
            for (int l = 0; l < NLDS; ++l) {
                const uint32_t off = g.offset[l], size = g.offset[l + 1] - off, res = g.resolution[l];
                const uint32_t* tl = lds_tab + off;
                const Corner c = grid_cell(g.scale[l], pos[s][0], pos[s][1], pos[s][2]);
                float r0 = 0.f, r1 = 0.f;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const uint32_t idx = nvo_grid_index(0u, size, res, c.px + (k & 1u), c.py + ((k >> 1) & 1u), c.pz + ((k >> 2) & 1u));
                    const float w = ((k & 1u) ? c.wx : 1.f - c.wx) * ((k & 2u) ? c.wy : 1.f - c.wy) *
                                    ((k & 4u) ? c.wz : 1.f - c.wz);
                    const float2 f = __half22float2(__builtin_bit_cast(__half2, tl[idx]));
                    r0 = fmaf(w, f.x, r0);
                    r1 = fmaf(w, f.y, r1);
                }
                o32[(size_t)l * N + is[s]] = nvo_cvt16x2(r0, r1, out_bf16 != 0);
            }
        }
        GP_CLK(gfc);
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            if (is[s] >= last) continue;
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const Corner& c = cg[s][q];
                float r0 = 0.f, r1 = 0.f;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const float w = ((k & 1u) ? c.wx : 1.f - c.wx) * ((k & 2u) ? c.wy : 1.f - c.wy) *
                                    ((k & 4u) ? c.wz : 1.f - c.wz);
                    const float2 f = __half22float2(__builtin_bit_cast(__half2, vg[s][q][k]));
                    r0 = fmaf(w, f.x, r0);
                    r1 = fmaf(w, f.y, r1);
                }
                o32[(size_t)(NLDS + q) * N + is[s]] = nvo_cvt16x2(r0, r1, out_bf16 != 0);
            }
        }
#ifdef NVO_GRID_PHASE
        {
            GP_CLK(gfd);
            gf_iss += gfb - gfa;
            gf_lds += gfc - gfb;
            gf_cons += gfd - gfc;
        }
#endif
    }
#ifdef NVO_GRID_PHASE
    if (threadIdx.x == 0) {
        GP_CLK(gfe);
        GP_ADD(32, gf1 - gf0); GP_ADD(33, gfe - gf1); GP_ADD(34, gf_iss); GP_ADD(35, gf_lds); GP_ADD(36, gf_cons); GP_ADD(37, 1);
    }
#endif
}

// k_grid_fwd_small, SOFTWARE-PIPELINED (round 6).  A fit of the two training launches (1 M samples 33 us, 393 K samples
// 26 us) puts ~22 us of every launch into what does not scale with the samples: the LDS staging ran as ~10 dependent
// L2 round trips (one 16-byte load -> store per thread and iteration), and every pass of a workgroup exposed its own
// position load and gather latencies one after the other.  Here (a) all staging loads of a thread are in flight at once
// (<= kStageMax x 16 bytes in registers), (b) the first pass's positions and global-level gathers are requested BEFORE the
// staging loads are waited for, (c) pass p + 2's positions and pass p + 1's gathers are in flight while pass p is consumed
// (loads return in issue order, so each wait only covers what was issued before it).  Same arithmetic, same order, one
// rounding: bit-identical to k_grid_fwd_small.
constexpr int kStageMax = 10;  // 10 x 16 B x 1024 threads = 160 KiB >= the 152 KiB the launcher admits

template <int NLDS, int NG>
__global__ void __launch_bounds__(kSmallBlock)
k_grid_fwd_small_pipe(NvoGridLevels g, uint32_t N, const float* __restrict__ x, const __half2* __restrict__ table,
                      __half2* __restrict__ out, int out_bf16, uint32_t per_block) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_tab[];
    const uint32_t* __restrict__ tab32 = reinterpret_cast<const uint32_t*>(table);
    const uint32_t first = blockIdx.x * per_block;
    const uint32_t last = min(N, first + per_block);
    if (first >= last) return;  // (uniform)
    const uint32_t n_pass = (last - first + kSmallBlock - 1u) / kSmallBlock;
    uint32_t* __restrict__ o32 = reinterpret_cast<uint32_t*>(out);

    struct Pos { float v[3]; };
    auto load_pos = [&](uint32_t pass) {  // (unconditional; a slot past the end re-reads the last sample and stores nothing)
        const uint32_t ic = min(first + pass * kSmallBlock + threadIdx.x, last - 1u);
        Pos p;
#pragma unroll
        for (int k = 0; k < 3; ++k) p.v[k] = x[3 * (size_t)ic + k];
        return p;
    };
    // What a gather leaves in registers is consumed a pass later, and NOTHING is computed from a loaded value next to its
    // load: a select or convert behind a load inside a lane-divergent branch makes the compiler wait for that load right
    // there (the value must exist at the join), which serialised the twelve gathers of a sample into twelve L2 round trips
    // in the first form of this kernel.  Per (y, z) pair: `pr` = the aligned 8-byte pair that holds corner x (hashed levels;
    // on dense levels the 8 bytes at corner x), `ex` = corner x + 1 where `pr` does not hold it -- requested under a
    // branch that contains the load alone, so only the lanes that need it reach the L1.
    Corner cg[NG];
    uint2 pr[NG][4];
    uint32_t ex[NG][4];
    uint32_t sel[NG];  // bit j: corner x of pair j is pr.y (hashed, odd index); bit 4 + j: corner x + 1 comes from ex
    auto issue = [&](const Pos& p) {  // cells of the global levels + every gather of the sample
        // index arithmetic of ALL pairs first (the dense levels' modulo sits behind a branch), then nothing but loads: a
        // register written between two loads can collide with an outstanding load's destination and wait for it
        uint32_t a0[NG][4], a1[NG][4];
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const uint32_t level = NLDS + q;
            const uint32_t off = g.offset[level], size = g.offset[level + 1] - off, res = g.resolution[level];
            const uint32_t hashed = g.hashed[level];
            cg[q] = grid_cell(g.scale[level], p.v[0], p.v[1], p.v[2]);
            sel[q] = 0u;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t cy = cg[q].py + (j & 1u), cz = cg[q].pz + (j >> 1);
                const uint32_t i0c = nvo_grid_index(hashed, size, res, cg[q].px, cy, cz);
                const uint32_t i1c = nvo_grid_index(hashed, size, res, cg[q].px + 1u, cy, cz);
                // hashed: idx1 == idx0 ^ 1 when the cell's x is even -- both corners in one aligned pair; dense: x neighbours
                // are neighbours in memory unless the pair straddles the table's wrap (dword alignment suffices; the last
                // entry's pair would reach past the level: taken one entry lower and read from .y)
                const bool hi = hashed ? (i0c & 1u) != 0u : (i0c + 1u >= size);
                const bool extra = hashed ? (cg[q].px & 1u) != 0u : (i1c != i0c + 1u);
                a0[q][j] = off + (hashed ? (i0c & ~1u) : (hi ? i0c - 1u : i0c));
                a1[q][j] = off + i1c;
                sel[q] |= (hi ? 1u : 0u) << j | (extra ? 16u : 0u) << j;
            }
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                pr[q][j] = nvo_ld_u2_a4(tab32 + a0[q][j]);
                ex[q][j] = 0u;
                if ((sel[q] >> (4u + j)) & 1u) ex[q][j] = tab32[a1[q][j]];
            }
        }
    };
    auto corner = [&](int q, uint32_t k) -> uint32_t {  // raw half2 of corner k (bit 0: x, bit 1: y, bit 2: z)
        const uint32_t j = k >> 1;
        const bool hi = (sel[q] >> j) & 1u, extra = (sel[q] >> (4u + j)) & 1u;
        if ((k & 1u) == 0u) return hi ? pr[q][j].y : pr[q][j].x;
        return extra ? ex[q][j] : (hi ? pr[q][j].x : pr[q][j].y);
    };

    Pos p0 = load_pos(0u);
    {   // staging: every load of the thread requested before the first is stored
        const uint32_t n4 = g.offset[NLDS] >> 2;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(table);
        uint4* dst = reinterpret_cast<uint4*>(lds_tab);
        uint4 st[kStageMax];
#pragma unroll
        for (int k = 0; k < kStageMax; ++k) st[k] = src[min(threadIdx.x + (uint32_t)k * kSmallBlock, n4 - 1u)];
        issue(p0);  // (the first pass's gathers fly with the staging loads)
        // (branch-free: a slot past the end re-writes the last vector's own value -- a store under `if (e < n4)` lets the
        // compiler sink each load into its branch, which is the load -> wait -> store chain this form is there to avoid)
#pragma unroll
        for (int k = 0; k < kStageMax; ++k) dst[min(threadIdx.x + (uint32_t)k * kSmallBlock, n4 - 1u)] = st[k];
    }
    __syncthreads();
    for (uint32_t pass = 0; pass < n_pass; ++pass) {
        const uint32_t i = first + pass * kSmallBlock + threadIdx.x;
        // (a value carried into the next iteration while its load is still in flight would be waited for at the loop's
        // register copies together with everything requested after it: the next positions are requested here, have arrived
        // by the time the gathers have, and only then take p0's place)
        const Pos p1 = load_pos(min(pass + 1u, n_pass - 1u));
        const bool live = i < last;
        // ---- LDS levels while the gathers fly
        if (live) {
#pragma unroll
            for (int l = 0; l < NLDS; ++l) {
                const uint32_t off = g.offset[l], size = g.offset[l + 1] - off, res = g.resolution[l];
                const uint32_t* tl = lds_tab + off;
                const Corner c = grid_cell(g.scale[l], p0.v[0], p0.v[1], p0.v[2]);
                float r0 = 0.f, r1 = 0.f;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const uint32_t idx = nvo_grid_index(0u, size, res, c.px + (k & 1u), c.py + ((k >> 1) & 1u), c.pz + ((k >> 2) & 1u));
                    const float w = ((k & 1u) ? c.wx : 1.f - c.wx) * ((k & 2u) ? c.wy : 1.f - c.wy) *
                                    ((k & 4u) ? c.wz : 1.f - c.wz);
                    const float2 f = __half22float2(__builtin_bit_cast(__half2, tl[idx]));
                    r0 = fmaf(w, f.x, r0);
                    r1 = fmaf(w, f.y, r1);
                }
                o32[(size_t)l * N + i] = nvo_cvt16x2(r0, r1, out_bf16 != 0);
            }
        }
        // ---- consume this pass's gathers
        uint32_t rg[NG];
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const Corner& c = cg[q];
            float r0 = 0.f, r1 = 0.f;
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                const float w = ((k & 1u) ? c.wx : 1.f - c.wx) * ((k & 2u) ? c.wy : 1.f - c.wy) *
                                ((k & 4u) ? c.wz : 1.f - c.wz);
                const float2 f = __half22float2(__builtin_bit_cast(__half2, corner(q, k)));
                r0 = fmaf(w, f.x, r0);
                r1 = fmaf(w, f.y, r1);
            }
            rg[q] = nvo_cvt16x2(r0, r1, out_bf16 != 0);
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < NG; ++q) o32[(size_t)(NLDS + q) * N + i] = rg[q];
        }
        // ---- next pass's gathers
        if (pass + 1u < n_pass) issue(p1);  // (uniform)
        p0 = p1;
    }
}

// k_grid_fwd_small, INSTRUCTION-LEAN (round 6).  The phase clocks (profiles/r6_grid_phase_fwd_small.txt) and an A/B of
// the pipelined form above against the plain one (no gain with every gather in flight) say what bounds this kernel: not a
// memory system rate but the vector ALU -- a wave64 instruction occupies its SIMD for 4 cycles (16 for the quarter-rate
// 32-bit multiplies), four waves share a SIMD, and the first form spent ~900 issue slots per sample on 40 corners: every
// corner's index from scratch (v_mad_u64_u32 / v_mul_lo_u32: the compiler neither knows that cell coordinates fit 24
// bits nor that (py + 1) P = py P + P), 64-bit address arithmetic per load, both index rules compiled into every level
// and chosen at run time, both 16-bit output formats computed and selected.  Here:
//   * which global levels are hashed (HMASK) and the output format (BF) are template parameters;
//   * dense levels: ONE base index per sample from 24-bit multiplies (exact: coordinates <= resolution <= 2^12, checked
//     against the level's resolution first), corners = base + per-level constants; a sample whose cell touches the upper
//     domain faces (an index may wrap) or lies outside the level (positions outside [0, 1]) takes the generic rule,
//     corner by corner -- the same values, behind a branch that training batches never take;
//   * hashed levels: two 32-bit multiplies per sample and level, the 4 (y, z) combinations by xor, both x corners from
//     them; the aligned 8-byte pair where the cell's x is even (as before);
//   * 32-bit byte offsets against scalar base pointers (saddr loads / stores) instead of 64-bit pointer arithmetic;
//   * corner weights as (wx' wy') wz' exactly as the reference order has them, 12 multiplies per level;
//   * waves past the workgroup's last sample leave the pass loop, and the workgroups' shares are wave-granular (the
//     second proposal level's 393 216 samples are 1.5 passes per workgroup: the first form ran 2 on 192 of the 256 CUs);
//   * the software pipeline of k_grid_fwd_small_pipe (staging loads in flight, next positions / gathers requested early).
// Same fp32 interpolation, same order, one rounding: bit-identical to k_grid_fwd_small (tools/probes/fwd_small_ab.py).
template <bool BF>
__device__ __forceinline__ uint32_t lean_cvt16x2(float a, float b) {
    asm("" : "+v"(a));  // (as nvo_cvt16: fp32 first, then ONE rounding to 16 bits, whatever the surrounding code)
    asm("" : "+v"(b));
    if constexpr (BF)
        return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)b) << 16);
    else
        return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)b) << 16);
}

// weights of the 4 (y, z) pairs: [j][0] = x-even corner, [j][1] = x-odd corner; (wx' * wy') * wz' as k_grid_fwd.  Written on
// two-float vectors: the x-even / x-odd halves of every product are one v_pk_mul_f32 (6 instead of 12 multiplies a level).
typedef float nvo_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lean_weights(const Corner& c, float (&w)[4][2]) {
    const float wy0 = 1.f - c.wy, wz0 = 1.f - c.wz;
    const nvo_v2f wxp = {1.f - c.wx, c.wx};
    const nvo_v2f a0 = wxp * wy0, a1 = wxp * c.wy;
    const nvo_v2f w0 = a0 * wz0, w1 = a1 * wz0, w2 = a0 * c.wz, w3 = a1 * c.wz;
    w[0][0] = w0.x; w[0][1] = w0.y;
    w[1][0] = w1.x; w[1][1] = w1.y;
    w[2][0] = w2.x; w[2][1] = w2.y;
    w[3][0] = w3.x; w[3][1] = w3.y;
}

template <bool BF>
__device__ __forceinline__ uint32_t lean_interp(const float (&w)[4][2], const uint32_t (&even)[4], const uint32_t (&odd)[4]) {
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 fe = __half22float2(__builtin_bit_cast(__half2, even[j]));
        const float2 fo = __half22float2(__builtin_bit_cast(__half2, odd[j]));
        r0 = fmaf(w[j][0], fe.x, r0);
        r1 = fmaf(w[j][0], fe.y, r1);
        r0 = fmaf(w[j][1], fo.x, r0);
        r1 = fmaf(w[j][1], fo.y, r1);
    }
    return lean_cvt16x2<BF>(r0, r1);
}

template <int NLDS, int NG, uint32_t HMASK, bool BF>
__global__ void __launch_bounds__(kSmallBlock)
k_grid_fwd_small_lean(NvoGridLevels g, uint32_t N, const float* __restrict__ x, const __half2* __restrict__ table,
                      __half2* __restrict__ out, uint32_t per_block) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_tab[];
    const unsigned char* __restrict__ tab8 = reinterpret_cast<const unsigned char*>(table);
    const unsigned char* __restrict__ x8 = reinterpret_cast<const unsigned char*>(x);
    unsigned char* __restrict__ o8 = reinterpret_cast<unsigned char*>(out);
    const uint32_t first = blockIdx.x * per_block;
    const uint32_t last = min(N, first + per_block);
    if (first >= last) return;  // (uniform)
    GP_CLK(gl0);
    const uint32_t n_pass = (last - first + kSmallBlock - 1u) / kSmallBlock;
    const uint32_t wave_first = first + (threadIdx.x & ~63u);  // first sample of this wave in pass 0

    struct Pos { float v[3]; };
    auto load_pos = [&](uint32_t pass) {  // (unconditional; a slot past the end re-reads the last sample and stores nothing)
        const uint32_t ic = min(first + pass * kSmallBlock + threadIdx.x, last - 1u);
        const float* p = reinterpret_cast<const float*>(x8 + ic * 12u);  // (32-bit byte offset: N < 2^32 / 12, launcher)
        Pos r;
        r.v[0] = p[0]; r.v[1] = p[1]; r.v[2] = p[2];
        return r;
    };

    // Per global level and (y, z) pair j, what a gather leaves in registers (nothing is computed from a loaded value next to
    // its load, see k_grid_fwd_small_pipe):
    //   hashed level: with pe = px & ~1, the entries of x = pe and x = pe + 1 differ in index bit 0 only -- `pr` = that aligned
    //     8-byte pair (it holds corner px whatever its parity, and corner px + 1 when px is even), `ex` = corner px + 1 when px
    //     is odd.  Which half of the pair is x = pe: bit 0 of the (y, z) hash = (py ^ pz ^ j ^ (j >> 1)) & 1 (both primes are
    //     odd) -- TWO lane predicates per level (parity of px, parity of py ^ pz) decide every select of the level.
    //   dense level: `pr` = corner px and its memory neighbour px + 1 (the generic rule for a cell on the upper domain faces
    //     loads the two corners separately into the same registers).
    Corner cg[NG];
    uint2 pr[NG][4];
    uint32_t ex[NG][4];
    auto issue = [&](const Pos& p) {
        uint32_t a0[NG][4], a1[NG][4];  // BYTE offsets from the table's base
        bool split[NG];                 // dense level: this sample takes the generic rule
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const uint32_t level = NLDS + q;
            const uint32_t off = g.offset[level], size = g.offset[level + 1] - off, res = g.resolution[level];
            cg[q] = grid_cell(g.scale[level], p.v[0], p.v[1], p.v[2]);
            const Corner& c = cg[q];
            split[q] = false;
            if ((HMASK >> q) & 1u) {  // (a constant once the loop is unrolled)
                const uint32_t mask = size - 1u;
                const uint32_t hy0 = c.py * 2654435761u, hy1 = hy0 + 2654435761u;
                const uint32_t hz0 = c.pz * 805459861u, hz1 = hz0 + 805459861u;
                const uint32_t a[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    a0[q][j] = (off + ((c.px ^ a[j]) & mask & ~1u)) << 2;
                    a1[q][j] = (off + (((c.px + 1u) ^ a[j]) & mask)) << 2;
                }
            } else {
                const uint32_t res2 = res * res;  // (scalar)
                // fast rule: cell inside the level and no corner index reaches the table's end => index = base + constant
                const uint32_t base = c.px + __umul24(c.py, res) + __umul24(c.pz, res2);
                split[q] = !(max(max(c.px, c.py), c.pz) < res && base + res2 + res + 1u < size);
                const uint32_t dj[4] = {0u, res, res2, res2 + res};
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    uint32_t i0c = base + dj[j], i1c = i0c + 1u;
                    if (split[q]) {  // the generic rule, corner by corner (upper domain faces; positions outside [0, 1])
                        const uint32_t cy = c.py + (j & 1u), cz = c.pz + (j >> 1);
                        i0c = nvo_grid_index(0u, size, res, c.px, cy, cz);
                        i1c = nvo_grid_index(0u, size, res, c.px + 1u, cy, cz);
                    }
                    a0[q][j] = (off + i0c) << 2;
                    a1[q][j] = (off + i1c) << 2;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if ((HMASK >> q) & 1u) {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    pr[q][j] = *reinterpret_cast<const uint2*>(tab8 + a0[q][j]);
                    ex[q][j] = 0u;
                }
                if (cg[q].px & 1u) {  // (the branch holds the four loads alone)
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) ex[q][j] = *reinterpret_cast<const uint32_t*>(tab8 + a1[q][j]);
                }
            } else {
                if (!split[q]) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const NvoU2A4 v = *reinterpret_cast<const NvoU2A4*>(tab8 + a0[q][j]);
                        pr[q][j] = make_uint2(v.x, v.y);
                    }
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        pr[q][j].x = *reinterpret_cast<const uint32_t*>(tab8 + a0[q][j]);
                        pr[q][j].y = *reinterpret_cast<const uint32_t*>(tab8 + a1[q][j]);
                    }
                }
            }
        }
    };

#ifndef NVO_LEAN_STAGE
#define NVO_LEAN_STAGE 1
#endif
    Pos p0 = load_pos(0u);
    {   // staging
        const uint32_t n4 = g.offset[NLDS] >> 2;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(table);
        uint4* dst = reinterpret_cast<uint4*>(lds_tab);
#if NVO_LEAN_STAGE == 0
        for (uint32_t e = threadIdx.x; e < n4; e += kSmallBlock) dst[e] = src[e];
#elif NVO_LEAN_STAGE == 2
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 st[kStageMax / 2];
#pragma unroll
            for (int k = 0; k < kStageMax / 2; ++k) st[k] = src[min(threadIdx.x + (uint32_t)(h * (kStageMax / 2) + k) * kSmallBlock, n4 - 1u)];
#pragma unroll
            for (int k = 0; k < kStageMax / 2; ++k) dst[min(threadIdx.x + (uint32_t)(h * (kStageMax / 2) + k) * kSmallBlock, n4 - 1u)] = st[k];
        }
#else
        uint4 st[kStageMax];
#pragma unroll
        for (int k = 0; k < kStageMax; ++k) st[k] = src[min(threadIdx.x + (uint32_t)k * kSmallBlock, n4 - 1u)];
#pragma unroll
        for (int k = 0; k < kStageMax; ++k) dst[min(threadIdx.x + (uint32_t)k * kSmallBlock, n4 - 1u)] = st[k];
#endif
    }
#ifdef NVO_GRID_PHASE
    GP_CLK(gls);
    if (threadIdx.x == 0) GP_ADD(34, gls - gl0);  // (staging alone, before the first pass's index arithmetic)
#endif
    // The kernel's ONE barrier sits directly behind the staging stores and the first pass's gathers are requested behind
    // IT: the waves of a workgroup move through a gather-issue phase very unevenly (the address path serves 16 waves x 24
    // divergent gathers; phase clocks: the last wave reaches the end of that phase ~20 K cycles after the first), and
    // with the phase in front of the barrier every wave waited for the slowest one -- 25 K of a workgroup's 40 K cycles.
    __syncthreads();
    GP_CLK(gl1);
    issue(p0);
    for (uint32_t pass = 0; pass < n_pass; ++pass) {
        if (wave_first + pass * kSmallBlock >= last) break;  // (wave-uniform: nothing of this wave is left; no barrier below)
        const uint32_t i = first + pass * kSmallBlock + threadIdx.x;
        const Pos p1 = load_pos(min(pass + 1u, n_pass - 1u));
        const bool live = i < last;
        // ---- LDS levels while the gathers fly
#pragma unroll
        for (int l = 0; l < NLDS; ++l) {
            const uint32_t off = g.offset[l], size = g.offset[l + 1] - off, res = g.resolution[l];
            const uint32_t res2 = res * res;
            const Corner c = grid_cell(g.scale[l], p0.v[0], p0.v[1], p0.v[2]);
            const uint32_t base = c.px + __umul24(c.py, res) + __umul24(c.pz, res2);
            const bool fast = max(max(c.px, c.py), c.pz) < res && base + res2 + res + 1u < size;
            const uint32_t* tl = lds_tab + off;
            uint32_t even[4], odd[4];
            if (fast) {
                const uint32_t* b = tl + base;
                even[0] = b[0]; odd[0] = b[1];
                even[1] = b[res]; odd[1] = b[res + 1u];
                even[2] = b[res2]; odd[2] = b[res2 + 1u];
                even[3] = b[res2 + res]; odd[3] = b[res2 + res + 1u];
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t cy = c.py + (j & 1u), cz = c.pz + (j >> 1);
                    even[j] = tl[nvo_grid_index(0u, size, res, c.px, cy, cz)];
                    odd[j] = tl[nvo_grid_index(0u, size, res, c.px + 1u, cy, cz)];
                }
            }
            float w[4][2];
            lean_weights(c, w);
            const uint32_t r = lean_interp<BF>(w, even, odd);
            if (live) *reinterpret_cast<uint32_t*>(o8 + (((uint32_t)l * N + i) << 2)) = r;
        }
        // ---- consume this pass's gathers
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            uint32_t even[4], odd[4];
            if ((HMASK >> q) & 1u) {
                const bool px_odd = (cg[q].px & 1u) != 0u;
                const bool t = (((cg[q].py ^ cg[q].pz) & 1u) != 0u) != px_odd;  // corner px sits in pr.y for j = 0, 3 (pr.x for 1, 2)
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const bool hi = (j == 0u || j == 3u) ? t : !t;
                    even[j] = hi ? pr[q][j].y : pr[q][j].x;
                    odd[j] = px_odd ? ex[q][j] : (hi ? pr[q][j].x : pr[q][j].y);
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    even[j] = pr[q][j].x;
                    odd[j] = pr[q][j].y;
                }
            }
            float w[4][2];
            lean_weights(cg[q], w);
            const uint32_t r = lean_interp<BF>(w, even, odd);
            if (live) *reinterpret_cast<uint32_t*>(o8 + (((uint32_t)(NLDS + q) * N + i) << 2)) = r;
        }
        // ---- next pass's gathers
        if (wave_first + (pass + 1u) * kSmallBlock < last) issue(p1);  // (wave-uniform)
        p0 = p1;
    }
#ifdef NVO_GRID_PHASE
    if (threadIdx.x == 0) {  // (slots of k_grid_fwd_small: staging [+ the first pass's gather issue here], sample loop, workgroups)
        GP_CLK(gl2);
        GP_ADD(32, gl1 - gl0); GP_ADD(33, gl2 - gl1); GP_ADD(37, 1);
    }
#endif
}

// k_grid_fwd for the level-major production path, INSTRUCTION-LEAN (round 6): the blockIdx -> (tile, level) plan of
// k_grid_fwd (XCD-balanced), a thread per (sample, level), SPT samples per thread -- with what k_grid_fwd_small_lean
// does to the arithmetic: the level's kind is block-uniform and takes one of two straight-line paths; hashed levels
// compute TWO 32-bit multiplies per sample (the four (y, z) combinations by xor) and fetch the aligned 8-byte pair around
// corner px plus, for odd px only, corner px + 1 (two lane predicates steer every select) -- one L1 look-up instead of two
// for half the (y, z) pairs; dense levels one base index from 24-bit multiplies with the generic rule behind a branch;
// 32-bit byte offsets against scalar bases; packed weight products; the output format a template parameter.
// Same fp32 interpolation, same order, one rounding: bit-identical to k_grid_fwd.
template <int SPT, bool BF, bool PAIR>
__global__ void __launch_bounds__(kGridBlock)
k_grid_fwd_lean(NvoGridLevels g, uint32_t N, const float* __restrict__ x, const __half2* __restrict__ table,
                __half2* __restrict__ out, GridFwdPlan plan, const uint32_t* __restrict__ n_live) {
    uint32_t tile, level;
    if (plan.enabled) {
        if (!grid_plan_map(plan, blockIdx.x, &tile, &level)) return;
    } else {
        grid_block_map(blockIdx.x, g.n_levels, &tile, &level);
    }
    if (n_live && tile * (kGridBlock * SPT) >= *n_live) return;
    const uint32_t i_first = tile * (kGridBlock * SPT) + threadIdx.x;
    if (i_first >= N) return;
    const uint32_t off = g.offset[level];
    const uint32_t size = g.offset[level + 1] - off;
    const uint32_t res = g.resolution[level];
    const uint32_t hashed = g.hashed[level];
    const float scale = g.scale[level];
    const unsigned char* __restrict__ tab8 = reinterpret_cast<const unsigned char*>(table + off);  // (the level's base)
    const unsigned char* __restrict__ x8 = reinterpret_cast<const unsigned char*>(x);
    unsigned char* __restrict__ o8 = reinterpret_cast<unsigned char*>(out) + (size_t)level * N * 4u;

    // NOTHING below is conditional on a sample being in range: a slot past the end re-reads sample N - 1, computes that
    // sample's value and stores it to that sample's place (the same bits its owner stores).  A store under `if (i < N)`
    // lets the compiler sink the slot's index arithmetic AND its gathers into the branch -- behind the previous slot's
    // consumption -- which halves the gathers in flight (measured: 72 against 59 us on an untrained field's batch).
    uint32_t i[SPT];
    Corner c[SPT];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        i[s] = i_first + (uint32_t)s * kGridBlock;
        const float* p = reinterpret_cast<const float*>(x8 + min(i[s], N - 1u) * 12u);
        const float px = p[0], py = p[1], pz = p[2];
        c[s] = grid_cell(scale, px, py, pz);
    }
    uint2 pr[SPT][4];
    uint32_t ex[SPT][4];
    if (hashed) {  // (block-uniform)
        const uint32_t mask = size - 1u;
        uint32_t a0[SPT][4], a1[SPT][4];
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const uint32_t hy0 = c[s].py * 2654435761u, hy1 = hy0 + 2654435761u;
            const uint32_t hz0 = c[s].pz * 805459861u, hz1 = hz0 + 805459861u;
            const uint32_t a[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                a0[s][j] = ((c[s].px ^ a[j]) & mask & (PAIR ? ~1u : ~0u)) << 2;
                a1[s][j] = (((c[s].px + 1u) ^ a[j]) & mask) << 2;
            }
        }
        if constexpr (PAIR) {
            // the aligned 8-byte pair around corner px (+ corner px + 1 for odd px): fewer L1 look-ups, but an 8-byte gather
            // occupies the address path twice as long as a 4-byte one -- measured SLOWER where the gather binds (60 -> 66 us
            // on the spread samples of an untrained field, tools/probes/fwd_main_ab.py) and kept for the A/B only
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    pr[s][j] = *reinterpret_cast<const uint2*>(tab8 + a0[s][j]);
                    ex[s][j] = 0u;
                }
                if (c[s].px & 1u) {  // (the branch holds the four loads alone)
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) ex[s][j] = *reinterpret_cast<const uint32_t*>(tab8 + a1[s][j]);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    pr[s][j].x = *reinterpret_cast<const uint32_t*>(tab8 + a0[s][j]);
                    pr[s][j].y = *reinterpret_cast<const uint32_t*>(tab8 + a1[s][j]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            uint32_t even[4], odd[4];
            if constexpr (PAIR) {
                const bool px_odd = (c[s].px & 1u) != 0u;
                const bool t = (((c[s].py ^ c[s].pz) & 1u) != 0u) != px_odd;  // corner px sits in pr.y for j = 0, 3 (pr.x for 1, 2)
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const bool hi = (j == 0u || j == 3u) ? t : !t;
                    even[j] = hi ? pr[s][j].y : pr[s][j].x;
                    odd[j] = px_odd ? ex[s][j] : (hi ? pr[s][j].x : pr[s][j].y);
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    even[j] = pr[s][j].x;
                    odd[j] = pr[s][j].y;
                }
            }
            float w[4][2];
            lean_weights(c[s], w);
            *reinterpret_cast<uint32_t*>(o8 + (min(i[s], N - 1u) << 2)) = lean_interp<BF>(w, even, odd);
        }
    } else {
        const uint32_t res2 = res * res;
        uint32_t a0[SPT][4], a1[SPT][4];
        bool split[SPT];
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const uint32_t base = c[s].px + __umul24(c[s].py, res) + __umul24(c[s].pz, res2);
            // (24-bit multiplies: exact while coordinates and res^2 stay below 2^24 -- the launcher admits res <= 4096)
            split[s] = !(max(max(c[s].px, c[s].py), c[s].pz) < res && base + res2 + res + 1u < size);
            const uint32_t dj[4] = {0u, res, res2, res2 + res};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                uint32_t i0c = base + dj[j], i1c = i0c + 1u;
                if (split[s]) {  // the generic rule, corner by corner (upper domain faces; positions outside [0, 1])
                    const uint32_t cy = c[s].py + (j & 1u), cz = c[s].pz + (j >> 1);
                    i0c = nvo_grid_index(0u, size, res, c[s].px, cy, cz);
                    i1c = nvo_grid_index(0u, size, res, c[s].px + 1u, cy, cz);
                }
                a0[s][j] = i0c << 2;
                a1[s][j] = i1c << 2;
            }
        }
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            if (!split[s]) {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const NvoU2A4 v = *reinterpret_cast<const NvoU2A4*>(tab8 + a0[s][j]);
                    pr[s][j] = make_uint2(v.x, v.y);
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    pr[s][j].x = *reinterpret_cast<const uint32_t*>(tab8 + a0[s][j]);
                    pr[s][j].y = *reinterpret_cast<const uint32_t*>(tab8 + a1[s][j]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            uint32_t even[4], odd[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                even[j] = pr[s][j].x;
                odd[j] = pr[s][j].y;
            }
            float w[4][2];
            lean_weights(c[s], w);
            *reinterpret_cast<uint32_t*>(o8 + (min(i[s], N - 1u) << 2)) = lean_interp<BF>(w, even, odd);
        }
    }
}

// The same over RUNS of four consecutive samples per thread (see k_grid_fwd_runs): the first proposal level's samples are
// 256 per ray at uniform lin-disp spacing whatever the state of training -- 3.8 / 2.2 / 1.3 samples per cell on its three
// global levels -- so a thread that walks four neighbours gathers once per cell change.  Level by level (the values of
// four samples of ONE level are 32 registers; all three at once would spill at 1024 threads per workgroup), with an LDS
// level evaluated while each global level's gathers fly.  Bit-identical to k_grid_fwd_small.
template <int NLDS, int NG>
__global__ void __launch_bounds__(kSmallBlock)
k_grid_fwd_small_runs(NvoGridLevels g, uint32_t N, const float* __restrict__ x, const __half2* __restrict__ table,
                      __half2* __restrict__ out, int out_bf16, uint32_t per_block) {
    constexpr int RUN = 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_tab[];
    const uint32_t* __restrict__ tab32 = reinterpret_cast<const uint32_t*>(table);
    {
        const uint32_t n4 = g.offset[NLDS] >> 2;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(table);
        uint4* dst = reinterpret_cast<uint4*>(lds_tab);
        for (uint32_t e = threadIdx.x; e < n4; e += kSmallBlock) dst[e] = src[e];
    }
    __syncthreads();
    const uint32_t first = blockIdx.x * per_block;
    const uint32_t last = min(N, first + per_block);  // (both multiples of RUN)
    uint32_t* __restrict__ o32 = reinterpret_cast<uint32_t*>(out);
    auto lds_level = [&](int l, const float (&p)[RUN * 3], uint32_t i0) {
        const uint32_t off = g.offset[l], size = g.offset[l + 1] - off, res = g.resolution[l];
        const uint32_t* tl = lds_tab + off;
        uint32_t r[RUN];
#pragma unroll
        for (int s = 0; s < RUN; ++s) {
            const Corner c = grid_cell(g.scale[l], p[3 * s], p[3 * s + 1], p[3 * s + 2]);
            float r0 = 0.f, r1 = 0.f;
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t idx = nvo_grid_index(0u, size, res, c.px + (k & 1u), c.py + ((k >> 1) & 1u), c.pz + ((k >> 2) & 1u));
                const float w = ((k & 1u) ? c.wx : 1.f - c.wx) * ((k & 2u) ? c.wy : 1.f - c.wy) *
                                ((k & 4u) ? c.wz : 1.f - c.wz);
                const float2 f = __half22float2(__builtin_bit_cast(__half2, tl[idx]));
                r0 = fmaf(w, f.x, r0);
                r1 = fmaf(w, f.y, r1);
            }
            r[s] = nvo_cvt16x2(r0, r1, out_bf16 != 0);
        }
        *reinterpret_cast<uint4*>(o32 + (size_t)l * N + i0) = make_uint4(r[0], r[1], r[2], r[3]);
    };
    for (uint32_t i0 = first + threadIdx.x * RUN; i0 < last; i0 += kSmallBlock * RUN) {
        float p[RUN * 3];
        {
            const float4* __restrict__ xp = reinterpret_cast<const float4*>(x + 3 * (size_t)i0);
#pragma unroll
            for (int q = 0; q < RUN * 3 / 4; ++q) {
                const float4 f = xp[q];
                p[4 * q] = f.x; p[4 * q + 1] = f.y; p[4 * q + 2] = f.z; p[4 * q + 3] = f.w;
            }
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const uint32_t level = NLDS + q;
            const uint32_t off = g.offset[level], size = g.offset[level + 1] - off, res = g.resolution[level];
            const uint32_t hashed = g.hashed[level];
            const uint32_t* __restrict__ tl = tab32 + off;
            Corner c[RUN];
            bool need[RUN];
            uint32_t v[RUN][8];
#pragma unroll
            for (int s = 0; s < RUN; ++s) {
                c[s] = grid_cell(g.scale[level], p[3 * s], p[3 * s + 1], p[3 * s + 2]);
                need[s] = s == 0 || c[s].px != c[s - 1].px || c[s].py != c[s - 1].py || c[s].pz != c[s - 1].pz;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) v[s][k] = 0u;
                if (!need[s]) continue;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t cy = c[s].py + (j & 1u), cz = c[s].pz + (j >> 1);
                    const uint32_t a = nvo_grid_index(hashed, size, res, c[s].px, cy, cz);
                    const uint32_t b = nvo_grid_index(hashed, size, res, c[s].px + 1u, cy, cz);
                    if (hashed ? ((c[s].px & 1u) == 0u) : false) {
                        // even cell x: idx1 == idx0 ^ 1 -- both corners sit in one aligned 8-byte pair
                        const uint2 pr = *reinterpret_cast<const uint2*>(tl + (a & ~1u));
                        v[s][2 * j] = (a & 1u) ? pr.y : pr.x;
                        v[s][2 * j + 1] = (a & 1u) ? pr.x : pr.y;
                    } else if (!hashed && b == a + 1u) {
                        const uint2 pr = nvo_ld_u2_a4(tl + a);
                        v[s][2 * j] = pr.x;
                        v[s][2 * j + 1] = pr.y;
                    } else {
                        v[s][2 * j] = tl[a];
                        v[s][2 * j + 1] = tl[b];
                    }
                }
            }
            // an LDS level while the gathers fly
            if (q < NLDS) lds_level(q, p, i0);
            uint32_t cur[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            uint32_t r[RUN];
#pragma unroll
            for (int s = 0; s < RUN; ++s) {
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) cur[k] = need[s] ? v[s][k] : cur[k];
                float r0 = 0.f, r1 = 0.f;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const float w = ((k & 1u) ? c[s].wx : 1.f - c[s].wx) * ((k & 2u) ? c[s].wy : 1.f - c[s].wy) *
                                    ((k & 4u) ? c[s].wz : 1.f - c[s].wz);
                    const float2 f = __half22float2(__builtin_bit_cast(__half2, cur[k]));
                    r0 = fmaf(w, f.x, r0);
                    r1 = fmaf(w, f.y, r1);
                }
                r[s] = nvo_cvt16x2(r0, r1, out_bf16 != 0);
            }
            *reinterpret_cast<uint4*>(o32 + (size_t)level * N + i0) = make_uint4(r[0], r[1], r[2], r[3]);
        }
#pragma unroll
        for (int l = NG; l < NLDS; ++l) lds_level(l, p, i0);  // (more LDS levels than global ones: the rest)
    }
}

// k_grid_fwd_runs, INSTRUCTION-LEAN (round 6): the run-walking inference form of the main grid's forward (plan and tile
// shape of k_grid_fwd_runs: a thread walks four consecutive samples of ONE level, block-uniform) with the arithmetic of
// k_grid_fwd_lean.  Bit-identical to k_grid_fwd.
template <int BLOCK, bool BF>
__global__ void __launch_bounds__(BLOCK)
k_grid_fwd_runs_lean(NvoGridLevels g, uint32_t N, const float* __restrict__ x, const __half2* __restrict__ table,
                     __half2* __restrict__ out, GridFwdPlan plan, const uint32_t* __restrict__ n_live) {
    constexpr int RUN = 4;
    uint32_t tile, level;
    if (plan.enabled) {
        if (!grid_plan_map(plan, blockIdx.x, &tile, &level)) return;
    } else {
        grid_block_map(blockIdx.x, g.n_levels, &tile, &level);
    }
    if (n_live && tile * (BLOCK * RUN) >= *n_live) return;
    const uint32_t i0 = tile * (BLOCK * RUN) + threadIdx.x * RUN;
    if (i0 >= N) return;  // (N is a multiple of RUN: a run is in range as a whole)
    const uint32_t off = g.offset[level];
    const uint32_t size = g.offset[level + 1] - off;
    const uint32_t res = g.resolution[level];
    const uint32_t hashed = g.hashed[level];
    const float scale = g.scale[level];
    const unsigned char* __restrict__ tab8 = reinterpret_cast<const unsigned char*>(table + off);
    float p[RUN * 3];
    {
        const float4* __restrict__ xp = reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(x) + i0 * 12u);
#pragma unroll
        for (int q = 0; q < RUN * 3 / 4; ++q) {
            const float4 f = xp[q];
            p[4 * q] = f.x; p[4 * q + 1] = f.y; p[4 * q + 2] = f.z; p[4 * q + 3] = f.w;
        }
    }
    Corner c[RUN];
    bool need[RUN];
    uint2 pr[RUN][4];
    uint32_t ex[RUN][4];
#pragma unroll
    for (int s = 0; s < RUN; ++s) {
        c[s] = grid_cell(scale, p[3 * s], p[3 * s + 1], p[3 * s + 2]);
        need[s] = s == 0 || c[s].px != c[s - 1].px || c[s].py != c[s - 1].py || c[s].pz != c[s - 1].pz;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            pr[s][j] = make_uint2(0u, 0u);
            ex[s][j] = 0u;
        }
    }
    uint32_t r[RUN];
    uint32_t even[4] = {0u, 0u, 0u, 0u}, odd[4] = {0u, 0u, 0u, 0u};
    if (hashed) {  // (block-uniform)
        const uint32_t mask = size - 1u;
#pragma unroll
        for (int s = 0; s < RUN; ++s) {
            if (!need[s]) continue;  // (the branch holds index arithmetic and loads alone)
            const uint32_t hy0 = c[s].py * 2654435761u, hy1 = hy0 + 2654435761u;
            const uint32_t hz0 = c[s].pz * 805459861u, hz1 = hz0 + 805459861u;
            const uint32_t a[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                pr[s][j] = *reinterpret_cast<const uint2*>(tab8 + (((c[s].px ^ a[j]) & mask & ~1u) << 2));
            if (c[s].px & 1u) {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j)
                    ex[s][j] = *reinterpret_cast<const uint32_t*>(tab8 + ((((c[s].px + 1u) ^ a[j]) & mask) << 2));
            }
        }
#pragma unroll
        for (int s = 0; s < RUN; ++s) {
            const bool px_odd = (c[s].px & 1u) != 0u;
            const bool t = (((c[s].py ^ c[s].pz) & 1u) != 0u) != px_odd;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const bool hi = (j == 0u || j == 3u) ? t : !t;
                const uint32_t e = hi ? pr[s][j].y : pr[s][j].x;
                const uint32_t o = px_odd ? ex[s][j] : (hi ? pr[s][j].x : pr[s][j].y);
                even[j] = need[s] ? e : even[j];
                odd[j] = need[s] ? o : odd[j];
            }
            float w[4][2];
            lean_weights(c[s], w);
            r[s] = lean_interp<BF>(w, even, odd);
        }
    } else {
        const uint32_t res2 = res * res;
#pragma unroll
        for (int s = 0; s < RUN; ++s) {
            if (!need[s]) continue;
            const uint32_t base = c[s].px + __umul24(c[s].py, res) + __umul24(c[s].pz, res2);
            const bool split = !(max(max(c[s].px, c[s].py), c[s].pz) < res && base + res2 + res + 1u < size);
            const uint32_t dj[4] = {0u, res, res2, res2 + res};
            if (!split) {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) pr[s][j] = nvo_ld_u2_a4(reinterpret_cast<const uint32_t*>(tab8 + ((base + dj[j]) << 2)));
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t cy = c[s].py + (j & 1u), cz = c[s].pz + (j >> 1);
                    pr[s][j].x = *reinterpret_cast<const uint32_t*>(tab8 + (nvo_grid_index(0u, size, res, c[s].px, cy, cz) << 2));
                    pr[s][j].y = *reinterpret_cast<const uint32_t*>(tab8 + (nvo_grid_index(0u, size, res, c[s].px + 1u, cy, cz) << 2));
                }
            }
        }
#pragma unroll
        for (int s = 0; s < RUN; ++s) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                even[j] = need[s] ? pr[s][j].x : even[j];
                odd[j] = need[s] ? pr[s][j].y : odd[j];
            }
            float w[4][2];
            lean_weights(c[s], w);
            r[s] = lean_interp<BF>(w, even, odd);
        }
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(out) + (((size_t)level * N + i0) << 2)) = make_uint4(r[0], r[1], r[2], r[3]);
}

// k_grid_fwd_small_runs, INSTRUCTION-LEAN (round 6): the run-walking inference form of the small-grid forward (a thread
// takes four consecutive samples of a ray and gathers only where the cell changes) with the arithmetic of
// k_grid_fwd_small_lean -- level kinds and output format as template parameters, dense levels from ONE base index per
// sample (24-bit multiplies, the generic rule behind a branch), hashed levels from two 32-bit multiplies with the aligned
// 8-byte pair around corner px (+ corner px + 1 for odd px, two lane predicates steering every select), 32-bit byte
// offsets against scalar bases, packed weight products, and nothing computed from a loaded value next to its load.
// Level by level (the raw gathers of four samples of one level are 48 registers).  Bit-identical to k_grid_fwd_small.
template <int NLDS, int NG, uint32_t HMASK, bool BF>
__global__ void __launch_bounds__(kSmallBlock)
k_grid_fwd_small_runs_lean(NvoGridLevels g, uint32_t N, const float* __restrict__ x, const __half2* __restrict__ table,
                           __half2* __restrict__ out, uint32_t per_block) {
    constexpr int RUN = 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_tab[];
    const unsigned char* __restrict__ tab8 = reinterpret_cast<const unsigned char*>(table);
    unsigned char* __restrict__ o8 = reinterpret_cast<unsigned char*>(out);
    {   // staging: every load of the thread requested before the first is stored
        const uint32_t n4 = g.offset[NLDS] >> 2;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(table);
        uint4* dst = reinterpret_cast<uint4*>(lds_tab);
        uint4 st[kStageMax];
#pragma unroll
        for (int k = 0; k < kStageMax; ++k) st[k] = src[min(threadIdx.x + (uint32_t)k * kSmallBlock, n4 - 1u)];
#pragma unroll
        for (int k = 0; k < kStageMax; ++k) dst[min(threadIdx.x + (uint32_t)k * kSmallBlock, n4 - 1u)] = st[k];
    }
    __syncthreads();
    const uint32_t first = blockIdx.x * per_block;
    const uint32_t last = min(N, first + per_block);  // (both multiples of RUN)
    auto lds_level = [&](int l, const float (&p)[RUN * 3], uint32_t i0) {
        const uint32_t off = g.offset[l], size = g.offset[l + 1] - off, res = g.resolution[l];
        const uint32_t res2 = res * res;
        const uint32_t* tl = lds_tab + off;
        uint32_t r[RUN];
#pragma unroll
        for (int s = 0; s < RUN; ++s) {
            const Corner c = grid_cell(g.scale[l], p[3 * s], p[3 * s + 1], p[3 * s + 2]);
            const uint32_t base = c.px + __umul24(c.py, res) + __umul24(c.pz, res2);
            const bool fast = max(max(c.px, c.py), c.pz) < res && base + res2 + res + 1u < size;
            uint32_t even[4], odd[4];
            if (fast) {
                const uint32_t* b = tl + base;
                even[0] = b[0]; odd[0] = b[1];
                even[1] = b[res]; odd[1] = b[res + 1u];
                even[2] = b[res2]; odd[2] = b[res2 + 1u];
                even[3] = b[res2 + res]; odd[3] = b[res2 + res + 1u];
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t cy = c.py + (j & 1u), cz = c.pz + (j >> 1);
                    even[j] = tl[nvo_grid_index(0u, size, res, c.px, cy, cz)];
                    odd[j] = tl[nvo_grid_index(0u, size, res, c.px + 1u, cy, cz)];
                }
            }
            float w[4][2];
            lean_weights(c, w);
            r[s] = lean_interp<BF>(w, even, odd);
        }
        *reinterpret_cast<uint4*>(o8 + (((uint32_t)l * N + i0) << 2)) = make_uint4(r[0], r[1], r[2], r[3]);
    };
    for (uint32_t i0 = first + threadIdx.x * RUN; i0 < last; i0 += kSmallBlock * RUN) {
        float p[RUN * 3];
        {
            const float4* __restrict__ xp = reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(x) + i0 * 12u);
#pragma unroll
            for (int q = 0; q < RUN * 3 / 4; ++q) {
                const float4 f = xp[q];
                p[4 * q] = f.x; p[4 * q + 1] = f.y; p[4 * q + 2] = f.z; p[4 * q + 3] = f.w;
            }
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const uint32_t level = NLDS + q;
            const uint32_t off = g.offset[level], size = g.offset[level + 1] - off, res = g.resolution[level];
            Corner c[RUN];
            bool need[RUN];
            uint2 pr[RUN][4];
            uint32_t ex[RUN][4];
            bool split[RUN];
#pragma unroll
            for (int s = 0; s < RUN; ++s) {
                c[s] = grid_cell(g.scale[level], p[3 * s], p[3 * s + 1], p[3 * s + 2]);
                need[s] = s == 0 || c[s].px != c[s - 1].px || c[s].py != c[s - 1].py || c[s].pz != c[s - 1].pz;
                split[s] = false;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    pr[s][j] = make_uint2(0u, 0u);
                    ex[s][j] = 0u;
                }
            }
#pragma unroll
            for (int s = 0; s < RUN; ++s) {
                if (!need[s]) continue;  // (the branch holds index arithmetic and loads alone)
                if ((HMASK >> q) & 1u) {
                    const uint32_t mask = size - 1u;
                    const uint32_t hy0 = c[s].py * 2654435761u, hy1 = hy0 + 2654435761u;
                    const uint32_t hz0 = c[s].pz * 805459861u, hz1 = hz0 + 805459861u;
                    const uint32_t a[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j)
                        pr[s][j] = *reinterpret_cast<const uint2*>(tab8 + ((off + ((c[s].px ^ a[j]) & mask & ~1u)) << 2));
                    if (c[s].px & 1u) {
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j)
                            ex[s][j] = *reinterpret_cast<const uint32_t*>(tab8 + ((off + (((c[s].px + 1u) ^ a[j]) & mask)) << 2));
                    }
                } else {
                    const uint32_t res2 = res * res;
                    const uint32_t base = c[s].px + __umul24(c[s].py, res) + __umul24(c[s].pz, res2);
                    split[s] = !(max(max(c[s].px, c[s].py), c[s].pz) < res && base + res2 + res + 1u < size);
                    const uint32_t dj[4] = {0u, res, res2, res2 + res};
                    if (!split[s]) {
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j)
                            pr[s][j] = nvo_ld_u2_a4(reinterpret_cast<const uint32_t*>(tab8 + ((off + base + dj[j]) << 2)));
                    } else {
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) {
                            const uint32_t cy = c[s].py + (j & 1u), cz = c[s].pz + (j >> 1);
                            pr[s][j].x = *reinterpret_cast<const uint32_t*>(tab8 + ((off + nvo_grid_index(0u, size, res, c[s].px, cy, cz)) << 2));
                            pr[s][j].y = *reinterpret_cast<const uint32_t*>(tab8 + ((off + nvo_grid_index(0u, size, res, c[s].px + 1u, cy, cz)) << 2));
                        }
                    }
                }
            }
            // an LDS level while the gathers fly
            if (q < NLDS) lds_level(q, p, i0);
            uint32_t even[4] = {0u, 0u, 0u, 0u}, odd[4] = {0u, 0u, 0u, 0u};
            uint32_t r[RUN];
#pragma unroll
            for (int s = 0; s < RUN; ++s) {
                if ((HMASK >> q) & 1u) {
                    const bool px_odd = (c[s].px & 1u) != 0u;
                    const bool t = (((c[s].py ^ c[s].pz) & 1u) != 0u) != px_odd;
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const bool hi = (j == 0u || j == 3u) ? t : !t;
                        const uint32_t e = hi ? pr[s][j].y : pr[s][j].x;
                        const uint32_t o = px_odd ? ex[s][j] : (hi ? pr[s][j].x : pr[s][j].y);
                        even[j] = need[s] ? e : even[j];
                        odd[j] = need[s] ? o : odd[j];
                    }
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        even[j] = need[s] ? pr[s][j].x : even[j];
                        odd[j] = need[s] ? pr[s][j].y : odd[j];
                    }
                }
                float w[4][2];
                lean_weights(c[s], w);
                r[s] = lean_interp<BF>(w, even, odd);
            }
            *reinterpret_cast<uint4*>(o8 + (((uint32_t)level * N + i0) << 2)) = make_uint4(r[0], r[1], r[2], r[3]);
        }
#pragma unroll
        for (int l = NG; l < NLDS; ++l) lds_level(l, p, i0);  // (more LDS levels than global ones: the rest)
    }
}

// ------------------------------------------------------------------------------------------
// backward w.r.t. parameters, global-atomic form
// ------------------------------------------------------------------------------------------
// dy   : SOA ? [L][N] : [N][L] of (half2 | float2), already multiplied by the loss scale
// grad : fp32 [entries][2], pre-zeroed
template <bool SOA, typename DY2>
__global__ void __launch_bounds__(kGridBlock)
k_grid_bwd_atomic(NvoGridLevels g, uint32_t N, const float* __restrict__ x,
                  const DY2* __restrict__ dy, float* __restrict__ grad) {
    // lane pair (2j, 2j+1) shares one sample; even lane adds feature 0, odd lane feature 1,
    // so the two 4-byte adds of a corner sit in one 64-B line of one wave instruction.
    uint32_t tile, level;
    grid_block_map(blockIdx.x, g.n_levels, &tile, &level);
    const uint32_t t = threadIdx.x;
    const uint32_t i = tile * (kGridBlock / 2) + (t >> 1);
    if (i >= N) return;
    const uint32_t feat = t & 1u;

    const uint32_t off = g.offset[level];
    const uint32_t size = g.offset[level + 1] - off;
    const uint32_t res = g.resolution[level];
    const uint32_t hashed = g.hashed[level];
    float* __restrict__ gr = grad + 2 * (size_t)off;

    const Corner c = grid_cell(g.scale[level], x[3 * (size_t)i + 0], x[3 * (size_t)i + 1],
                               x[3 * (size_t)i + 2]);
    const DY2 d2 = SOA ? dy[(size_t)level * N + i] : dy[(size_t)i * g.n_levels + level];
    const float2 f = dy2f(d2);
    const float d = feat ? f.y : f.x;
    if (d == 0.f) return;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t idx = nvo_grid_index(hashed, size, res, c.px + (k & 1u),
                                            c.py + ((k >> 1) & 1u), c.pz + ((k >> 2) & 1u));
        const float w = ((k & 1u) ? c.wx : 1.f - c.wx) * ((k & 2u) ? c.wy : 1.f - c.wy) *
                        ((k & 4u) ? c.wz : 1.f - c.wz);
        atomicAdd(gr + 2 * (size_t)idx + feat, w * d);
    }
}

// ------------------------------------------------------------------------------------------
// backward w.r.t. parameters, LDS slice-owner form (no global atomics)
// ------------------------------------------------------------------------------------------
// Two accumulator kinds share one kernel; the host picks one PER LEVEL from the expected hit rate:
//  * FIXED  -- 64-bit fixed point, 8K-entry slices (128 KiB).  LDS integer atomics retire 4.6 lanes
//              per clock per CU on MI355X, LDS float atomics 0.33 (lane-serialised; measured with
//              tools/probes/lds_atomic_probe.hip), so every level whose slices are hit often (dense
//              levels, small hash tables) accumulates in integers.  Scale 2^26: resolution 1.5e-8 on
//              the loss-scaled gradient (finer than the fp16 subnormal step tcnn's fp16 accumulation
//              flushes at), range +-1.4e11.  Integer adds are associative, so a single-chunk item's
//              result is bitwise reproducible.
//  * FLOAT  -- fp32, 20K-entry slices (160 KiB = the whole LDS of a CU).  For 2^19-entry hashed
//              levels a slice receives ~1/26 of the lookups, the kernel is bound by re-deriving the
//              corner hashes for every (sample, slice) pair, and fewer/larger slices win.
constexpr uint32_t kSliceFixed = 8192;
constexpr uint32_t kSliceFloat = 20448;  // (160 KiB less 256 B: the item's work counter is static LDS)
constexpr uint32_t kLdsBwdBytes = 160 * 1024;
constexpr int kLdsBwdBlock = 1024;
// Hit queue of the slice-owner scan on HASHED levels (round 4): a (y, z) corner pair of a sample falls into the item's
// slice with probability 1 / (slices of the level) -- 1/8 for the proposal grids -- so nearly every wave ran the ~26
// instruction accumulate path 16 times per pass with an eighth of its lanes active.  Each wave now PUSHES its hits into a
// private ring in the LDS the 32-bit accumulators leave free (128 KiB of the CU's 160) and drains 64 entries at a time
// with every lane busy.  127 entries of 16 bytes per wave: 16 x 2032 B + 128 KiB = 160 KiB - 256 B.
constexpr uint32_t kHitCap = 127;
constexpr uint32_t kHitRingBytes = 16u * kHitCap * (kLdsBwdBlock / 64);
constexpr float kFixScale = 67108864.f;          // 2^26
constexpr float kFixInv = 1.0f / 67108864.f;

// float -> 2^26 fixed point through the double "magic number" trick: (double)v * 2^26 + 1.5 * 2^52
// leaves round-to-nearest(v * 2^26) in the low mantissa bits, so one cvt + one f64 fma + a 64-bit
// subtract replace the ~20-instruction software float->int64 conversion.  Valid for |v| < 2^25.
__device__ __forceinline__ unsigned long long to_fixed(float v) {
    const double magic = 6755399441055744.0;  // 1.5 * 2^52
    const double t = fma((double)v, (double)kFixScale, magic);
    return (unsigned long long)(__double_as_longlong(t) - __double_as_longlong(magic));
}

// Per-item scale context of the accumulators (only AccFixed32 uses it): s = fixed-point scale per feature,
// inv = its reciprocal.
struct AccScale {
    float s0, s1, inv0, inv1;
};

struct AccFixed {
    typedef unsigned long long T;
    static constexpr uint32_t kEntries = kSliceFixed;
    static __device__ __forceinline__ void add(T* acc, uint32_t rel, float v0, float v1, const AccScale&) {
        atomicAdd(&acc[2 * rel + 0], to_fixed(v0));
        atomicAdd(&acc[2 * rel + 1], to_fixed(v1));
    }
    // int64 -> float through double (cvt_f64_i32 + cvt_f64_u32 + one f64 fma + cvt_f32_f64: 4 instructions instead of
    // the ~12 of the software int64 -> float conversion; the double holds every |sum| < 2^53 -- 1.3e8 in gradient
    // units -- exactly, so the one rounding to float is the correctly rounded conversion)
    static __device__ __forceinline__ float get(const T* acc, uint32_t e, const AccScale&) {
        const unsigned long long v = acc[e];
        const double d = fma((double)(int)(uint32_t)(v >> 32), 4294967296.0, (double)(uint32_t)v);
        return (float)d * kFixInv;
    }
};
struct AccFloat {
    typedef float T;
    static constexpr uint32_t kEntries = kSliceFloat;
    static __device__ __forceinline__ void add(T* acc, uint32_t rel, float v0, float v1, const AccScale&) {
        atomicAdd(&acc[2 * rel + 0], v0);
        atomicAdd(&acc[2 * rel + 1], v1);
    }
    static __device__ __forceinline__ float get(const T* acc, uint32_t e, const AccScale&) { return acc[e]; }
};
// 32-bit fixed point with a DATA-DERIVED, overflow-proof scale: every contribution to an entry is w * dy with
// trilinear weights 0 <= w <= 1 that sum to 1 over a sample's corners, so |sum| <= L1_f = sum_i |dy_f(i)| of
// the level for any entry; scale_f = 2^29 / L1_f keeps every partial sum (rounding slack included) inside
// int32.  Resolution L1_f / 2^29 -- relative ~1e-5..1e-4 for a typical entry of a coarse level, i.e. finer
// than the fp16 accumulation of the reference (2^-11 per add) though coarser than the 64-bit form.  Half the
// LDS per entry => 16K-entry slices => half the slices of a level (half the redundant scans), a 2-instruction
// conversion instead of cvt_f64 + fma_f64 + 64-bit subtract, and 32-bit LDS atomics.  Integer adds are
// associative: results are bitwise reproducible for single-chunk items.
struct AccFixed32 {
    typedef int T;
    static constexpr uint32_t kEntries = 16384;
    static __device__ __forceinline__ void add(T* acc, uint32_t rel, float v0, float v1, const AccScale& sc) {
        atomicAdd(&acc[2 * rel + 0], __float2int_rn(v0 * sc.s0));
        atomicAdd(&acc[2 * rel + 1], __float2int_rn(v1 * sc.s1));
    }
    static __device__ __forceinline__ float get(const T* acc, uint32_t e, const AccScale& sc) {
        return (float)acc[e] * ((e & 1u) ? sc.inv1 : sc.inv0);
    }
};

// One workgroup per WORK ITEM = (level, slice, sample chunk).  A slice of a large hashed level is
// hit by a small share of all corner lookups and gets one item that scans every sample and writes
// its slice with plain stores.  A slice that is hit often (dense levels, small tables) is split into
// sample chunks of equal expected hit count (table built on the host from the hit share alone,
// independent of N); chunked items flush with row-contiguous float atomics (256-B shaped, the fast
// form) into a pre-zeroed range.
template <typename ACC, bool SOA, typename DY2>
__device__ __forceinline__ void grid_bwd_item(const NvoGridLevels& g, uint32_t N,
                                              const float* __restrict__ x, const DY2* __restrict__ dy,
                                              float* __restrict__ grad, uint32_t level, uint32_t first,
                                              uint32_t chunk, uint32_t n_chunks, void* lds_raw,
                                              const AccScale sc = AccScale{0.f, 0.f, 0.f, 0.f},
                                              const uint32_t* __restrict__ live = nullptr, bool merge = false,
                                              uint32_t slice_cap = ACC::kEntries, uint32_t* __restrict__ nf_flag = nullptr,
                                              const uint32_t* __restrict__ live_n = nullptr, uint32_t ring_off = 0u,
                                              const uint16_t* __restrict__ codes = nullptr, uint32_t list_pass = 4096u) {
    typename ACC::T* acc = reinterpret_cast<typename ACC::T*>(lds_raw);
    const uint32_t off = g.offset[level];
    const uint32_t size = g.offset[level + 1] - off;
    const uint32_t res = g.resolution[level];
    const uint32_t hashed = g.hashed[level];
    const float scale = g.scale[level];
    const uint32_t count = min(slice_cap, size - first);  // slice_cap <= ACC::kEntries
    // live != nullptr: *live_n = number of samples with a non-zero gradient, live[j] = their ids; the chunks then
    // partition that list.  (A list that holds most of the samples is not worth the indirection: identity scan.)
    const uint32_t n_live = live ? *live_n : N;
    const bool listed = live != nullptr && n_live < N - (N >> 2);
    const uint32_t n_scan = listed ? n_live : N;
    // A short list does not need all of a slice's chunks: an item costs ~10 us of zeroing / flushing / barriers
    // whatever it scans (measured: with 95 % of the samples dead the launch only got 15 % faster), so only as many
    // chunks stay active as have a full pass (list_pass = 4096 samples) to scan; the others leave at once.  The slice of
    // a chunked item is flushed with atomics into a pre-zeroed range, which any number of active chunks satisfies.
    // (list_pass = 1024, a stream layout's coarse levels behind k_live_rows: EVERY listed sample carries a gradient there
    // and it is their visits that cost, not the scan -- 31 K listed samples on 7 of a slice's 28 chunks took 25 us where the
    // full scan of 1.5 M mostly dead ones on all 28 takes 12.)
    uint32_t n_act = n_chunks;
    if (listed && n_chunks > 1u) {
        n_act = max(1u, min(n_chunks, n_scan / list_pass));
        if (chunk >= n_act) return;
    }

    // The item's sample range is handed out to the WAVES in blocks of 64 x 8 (or 64 x 4) consecutive samples from a
    // counter in LDS: with a fixed stride per wave the waves whose samples happen to hit this slice finish last and the
    // other fifteen wait at the barrier -- 20-25 % of an item's cycles by the phase clocks.
    __shared__ uint32_t s_next;
    GP_CLK(go0);
    if (threadIdx.x == 0) s_next = 0u;
    for (uint32_t e = threadIdx.x; e < 2 * count; e += kLdsBwdBlock) acc[e] = (typename ACC::T)0;
    __syncthreads();
    GP_CLK(go1);
    const uint32_t lane_id = threadIdx.x & 63u;
    auto wave_grab = [&](uint32_t n) -> uint32_t {
        uint32_t c = 0u;
        if (lane_id == 0u) c = atomicAdd(&s_next, n);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    };

    // (chunk boundaries on multiples of 8 samples: the run-merging scan loads 8 consecutive samples per lane)
    const uint32_t per_chunk = ((n_scan + n_act - 1) / n_act + 7u) & ~7u;
    const uint32_t begin = min(n_scan, chunk * per_chunk);
    const uint32_t end = min(n_scan, begin + per_chunk);
    // Samples are taken kUnroll at a time per thread with all of their loads issued up front: the
    // loop is otherwise one dependent L2 round trip per sample.
    constexpr uint32_t kUnroll = 4;
    const uint32_t mask = size - 1u;
    const uint32_t res2 = res * res;
    // (hashed levels) both x corners of a (y, z) pair share a slice when the slice size is a power of two
    // that exceeds every x coordinate
    const bool pair_bins = hashed && (ACC::kEntries & (ACC::kEntries - 1u)) == 0u && res + 1u < ACC::kEntries &&
                           (first & (ACC::kEntries - 1u)) == 0u && count == ACC::kEntries;
    // Integer accumulators cannot carry inf / NaN: a non-finite dy (fp16 overflow of the scaled loss gradient) is
    // remembered here and poisons the slice's first gradient entry after the flush, so that the optimiser's
    // non-finite check sees it exactly as it would with floating-point accumulation.
    bool bad = false;
    if (merge && !hashed) {
        // Run-merging scan of a DENSE level (option grid_bwd_runs).  Consecutive samples are neighbours on a ray and a
        // coarse cell holds a run of them: a lane takes 8 CONSECUTIVE samples (a wave 512, all 8 loads in flight at
        // once), sums the 8 x 2 corner contributions of the current cell in fp32 registers and goes to the LDS
        // accumulators once per run -- index arithmetic, slice tests, float -> fixed conversions and atomics per RUN
        // instead of per sample, and the lanes of one atomic instruction are 8 samples apart instead of adjacent
        // (the per-sample form had 2/3 of its LDS cycles in same-address conflicts).  A run whose 8 corners all
        // miss the slice costs the cell computation only.
        constexpr uint32_t kRun = 8;
        float a0[8], a1[8];
        uint32_t cx = 0, cy = 0, cz = 0, cbase = 0;
        bool open = false, hit = false;
        const uint32_t span = 1u + res + res2;  // largest corner offset from the cell's base index
        auto flush = [&]() {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                uint32_t i0 = cbase + ((j & 1u) ? res : 0u) + ((j & 2u) ? res2 : 0u);
                uint32_t i1 = i0 + 1u;
                if (i1 >= size) {
                    i0 %= size;
                    i1 %= size;
                }
                const uint32_t r0 = i0 - first, r1 = i1 - first;  // unsigned wrap -> huge when below the slice
                if (r0 < count) ACC::add(acc, r0, a0[2 * j], a1[2 * j], sc);
                if (r1 < count) ACC::add(acc, r1, a0[2 * j + 1], a1[2 * j + 1], sc);
            }
        };
        auto visit = [&](float px, float py, float pz, float2 d) {
            bad = bad || !(fabsf(d.x) < INFINITY) || !(fabsf(d.y) < INFINITY);
            if (d.x == 0.f && d.y == 0.f) return;
            const Corner c = grid_cell(scale, px, py, pz);
            if (!open || c.px != cx || c.py != cy || c.pz != cz) {
                if (open && hit) flush();
                cx = c.px;
                cy = c.py;
                cz = c.pz;
                open = true;
                cbase = c.px + c.py * res + c.pz * res2;
                // corners lie in [cbase, cbase + span] (or wrap, upper domain face only: treated as a hit)
                hit = cbase + span >= size || (cbase + span >= first && cbase < first + count);
                if (hit) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) a0[k] = a1[k] = 0.f;
                }
            }
            if (!hit) return;
            const float wx0 = 1.f - c.wx, wy0 = 1.f - c.wy, wz0 = 1.f - c.wz;
            const float wyz[4] = {wy0 * wz0, c.wy * wz0, wy0 * c.wz, c.wy * c.wz};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const float w0 = wx0 * wyz[j], w1 = c.wx * wyz[j];
                a0[2 * j] += w0 * d.x;
                a1[2 * j] += w0 * d.y;
                a0[2 * j + 1] += w1 * d.x;
                a1[2 * j + 1] += w1 * d.y;
            }
        };
        bool vec = false;
        if constexpr (SOA && sizeof(DY2) == 4) {
            vec = !listed && (N & 3u) == 0u && (((uintptr_t)dy) & 15u) == 0u && (((uintptr_t)x) & 15u) == 0u;
        }
        // (a list of samples that ALL carry a gradient -- list_pass < 4096, k_live_rows on a trained field, one sample per
        // ray: a lane takes ONE sample of a 64-sample block; list neighbours are not neighbours in space, there is nothing
        // to merge, and a short list then reaches every wave: 1.1 K samples per item are 17 blocks of 64 but only 3 of 512)
        const bool single = listed && list_pass < 4096u;
        const uint32_t kGrab = single ? 64u : 64u * kRun;
        const uint32_t kLaneRun = single ? 1u : kRun;
        for (uint32_t c0 = wave_grab(kGrab); begin + c0 < end; c0 = wave_grab(kGrab)) {
            const uint32_t b0 = begin + c0 + lane_id * kLaneRun;
            if (b0 >= end) continue;
            open = false;
            hit = false;
            bool done = false;
            if constexpr (SOA && sizeof(DY2) == 4) {
                if (vec && b0 + kRun <= end) {
                    const float4* __restrict__ xp = reinterpret_cast<const float4*>(x + 3 * (size_t)b0);
                    const uint4* __restrict__ dp = reinterpret_cast<const uint4*>(dy + (size_t)level * N + b0);
                    const float4 p0 = xp[0], p1 = xp[1], p2 = xp[2], p3 = xp[3], p4 = xp[4], p5 = xp[5];
                    const uint4 q0 = dp[0], q1 = dp[1];
                    visit(p0.x, p0.y, p0.z, dy2f(__builtin_bit_cast(DY2, q0.x)));
                    visit(p0.w, p1.x, p1.y, dy2f(__builtin_bit_cast(DY2, q0.y)));
                    visit(p1.z, p1.w, p2.x, dy2f(__builtin_bit_cast(DY2, q0.z)));
                    visit(p2.y, p2.z, p2.w, dy2f(__builtin_bit_cast(DY2, q0.w)));
                    visit(p3.x, p3.y, p3.z, dy2f(__builtin_bit_cast(DY2, q1.x)));
                    visit(p3.w, p4.x, p4.y, dy2f(__builtin_bit_cast(DY2, q1.y)));
                    visit(p4.z, p4.w, p5.x, dy2f(__builtin_bit_cast(DY2, q1.z)));
                    visit(p5.y, p5.z, p5.w, dy2f(__builtin_bit_cast(DY2, q1.w)));
                    done = true;
                }
            }
            if (!done && single) {
                const uint32_t id = live[b0];
                const DY2 dd = SOA ? dy[(size_t)level * N + id] : dy[(size_t)id * g.n_levels + level];
                visit(x[3 * (size_t)id + 0], x[3 * (size_t)id + 1], x[3 * (size_t)id + 2], dy2f(dd));
                done = true;
            }
            if (!done) {
                const uint32_t stop = min(end, b0 + kRun);
                for (uint32_t j = b0; j < stop; ++j) {
                    const uint32_t i = listed ? live[j] : j;
                    const DY2 d2 = SOA ? dy[(size_t)level * N + i] : dy[(size_t)i * g.n_levels + level];
                    visit(x[3 * (size_t)i + 0], x[3 * (size_t)i + 1], x[3 * (size_t)i + 2], dy2f(d2));
                }
            }
            if (open && hit) flush();
        }
    } else if (pair_bins && ring_off != 0u && codes != nullptr && sizeof(DY2) == 4 && ACC::kEntries == 16384u) {
      if constexpr (sizeof(DY2) == 4 && ACC::kEntries == 16384u) {  // (the launcher hands codes to this combination only)
        // HASHED level, SLICE CODES (round 6).  The scan below this branch derives every sample's cell, two 32-bit
        // multiplies and four hashes on EVERY one of the level's 8 slice visits to find the one (y, z) pair in eight that
        // lands in the slice: ~100 vector instructions per sample and visit, and the launch is bound by exactly that
        // (1 M samples x 16 hashed visits on ~250 CUs at one wave instruction per cycle and CU).  k_slice_codes derives them
        // ONCE per (sample, level) and leaves 12 bits -- the slice of each of the four (y, z) pairs; a visit here is a
        // 2-byte code + the 4-byte gradient, four field compares and one push of {sample | pair mask, gradient} for the
        // ~40 % of the samples that touch the slice at all; the cell, the hashes and the weights are computed when 64
        // entries are drained, with every lane busy.  Same products, integer accumulation: bit-identical gradients.
        const uint32_t my = (first >> 14) & 7u;
        constexpr uint32_t kCap = 2u * kHitCap;  // 8-byte entries in the wave's ring
        uint2* const ring = reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(lds_raw) + ring_off) +
                            (threadIdx.x >> 6) * kCap;
        uint32_t q_head = 0u, q_fill = 0u;
        auto q_drain = [&](uint32_t n) {  // the n <= 64 oldest entries
            if (lane_id < n) {
                uint32_t p = q_head + lane_id;
                if (p >= kCap) p -= kCap;
                const uint2 e = ring[p];
                const uint32_t i = e.x & 0x0FFFFFFFu, m = e.x >> 28;
                const float2 d = dy2f(__builtin_bit_cast(DY2, e.y));
                const float* xp = x + 3 * (size_t)i;
                const Corner c = grid_cell(scale, xp[0], xp[1], xp[2]);
                const float wx0 = 1.f - c.wx, wy0 = 1.f - c.wy, wz0 = 1.f - c.wz;
                const float wyz[4] = {wy0 * wz0, c.wy * wz0, wy0 * c.wz, c.wy * c.wz};
                const uint32_t hy0 = c.py * 2654435761u, hy1 = hy0 + 2654435761u;
                const uint32_t hz0 = c.pz * 805459861u, hz1 = hz0 + 805459861u;
                const uint32_t a[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    if ((m >> j) & 1u) {
                        const uint32_t lo = a[j] & mask & (ACC::kEntries - 1u);
                        const float u0 = wyz[j] * d.x, u1 = wyz[j] * d.y;
                        ACC::add(acc, lo ^ c.px, wx0 * u0, wx0 * u1, sc);
                        ACC::add(acc, lo ^ (c.px + 1u), c.wx * u0, c.wx * u1, sc);
                    }
                }
            }
            q_head += n;
            if (q_head >= kCap) q_head -= kCap;
            q_fill -= n;
        };
        for (uint32_t c0 = wave_grab(64u * kUnroll); begin + c0 < end; c0 = wave_grab(64u * kUnroll)) {
            const uint32_t i0 = begin + c0 + lane_id;
            uint32_t sid[kUnroll], code[kUnroll], dyr[kUnroll];
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; ++u) {
                const uint32_t j = i0 + u * 64u;
                sid[u] = j < end ? (listed ? live[j] : j) : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; ++u) {  // (unconditional loads at a valid index; slots past the end are masked below)
                code[u] = codes[sid[u]];
                const DY2 d2 = SOA ? dy[(size_t)level * N + sid[u]] : dy[(size_t)sid[u] * g.n_levels + level];
                dyr[u] = __builtin_bit_cast(uint32_t, d2);
            }
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; ++u) {
                const bool valid = i0 + u * 64u < end;
                const float2 d = dy2f(__builtin_bit_cast(DY2, dyr[u]));
                bad = bad || (valid && (!(fabsf(d.x) < INFINITY) || !(fabsf(d.y) < INFINITY)));
                uint32_t m = 0u;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) m |= (((code[u] >> (3u * j)) & 7u) == my ? 1u : 0u) << j;
                const bool hit = valid && m != 0u && (d.x != 0.f || d.y != 0.f);
                // (wave-uniform control flow around the ballot, as in the scan below)
                const unsigned long long bm = __ballot(hit);
                const uint32_t cnt = (uint32_t)__popcll(bm);
                if (hit) {
                    uint32_t p = q_head + q_fill + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                    if (p >= kCap) p -= kCap;
                    ring[p] = make_uint2(sid[u] | (m << 28), dyr[u]);
                }
                q_fill += cnt;  // (< 64 before, <= 64 more: never beyond the ring's 254 entries)
                if (q_fill >= 64u) q_drain(64u);
            }
        }
        while (q_fill) q_drain(min(q_fill, 64u));
      }
    } else if (pair_bins && ring_off != 0u) {
        // HASHED level with a hit queue (kHitCap).  Everything up to the slice test runs for all 64 lanes; the pairs that
        // fall into this slice (one in `slices of the level` on average) are pushed into the wave's ring -- 16 bytes:
        // in-slice offset | px << 14, w_y w_z dy_0, w_y w_z dy_1, w_x -- and whenever 64 are waiting the whole wave
        // splits them into their two x corners and adds them.  The ring is private to the wave (LDS operations of one
        // wave execute in order: no barrier), head and fill level are wave-uniform scalars, and every push sits in
        // wave-uniform control flow (a ballot under a lane-divergent branch would let the lanes disagree about them).
        uint4* const ring = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(lds_raw) + ring_off) +
                            (threadIdx.x >> 6) * kHitCap;
        uint32_t q_head = 0u, q_fill = 0u;
        auto q_drain = [&](uint32_t n) {  // the n <= 64 oldest entries
            if (lane_id < n) {
                uint32_t p = q_head + lane_id;
                if (p >= kHitCap) p -= kHitCap;
                const uint4 e = ring[p];
                const uint32_t lo = e.x & (ACC::kEntries - 1u), px = e.x >> 14;
                const float u0 = __uint_as_float(e.y), u1 = __uint_as_float(e.z), wx = __uint_as_float(e.w);
                const float wx0 = 1.f - wx;
                ACC::add(acc, lo ^ px, wx0 * u0, wx0 * u1, sc);
                ACC::add(acc, lo ^ (px + 1u), wx * u0, wx * u1, sc);
            }
            q_head += n;
            if (q_head >= kHitCap) q_head -= kHitCap;
            q_fill -= n;
        };
        auto q_push = [&](bool hit, uint32_t word, float u0, float u1, float wx) {
            const unsigned long long m = __ballot(hit);
            const uint32_t c = (uint32_t)__popcll(m);
            if (q_fill + c > kHitCap) {  // (cannot happen below 64 waiting entries unless > 63 lanes hit at once)
                while (q_fill) q_drain(min(q_fill, 64u));
            }
            if (hit) {
                uint32_t p = q_head + q_fill + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (p >= kHitCap) p -= kHitCap;
                ring[p] = make_uint4(word, __float_as_uint(u0), __float_as_uint(u1), __float_as_uint(wx));
            }
            q_fill += c;
            if (q_fill >= 64u) q_drain(64u);
        };
        for (uint32_t c0 = wave_grab(64u * kUnroll); begin + c0 < end; c0 = wave_grab(64u * kUnroll)) {
            const uint32_t i0 = begin + c0 + lane_id;
            float2 dv[kUnroll];
            float xv[kUnroll][3];
            uint32_t sid[kUnroll];
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; ++u) {
                const uint32_t j = i0 + u * 64u;
                sid[u] = j < end ? (listed ? live[j] : j) : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; ++u) {
                const uint32_t i = sid[u];
                dv[u] = make_float2(0.f, 0.f);
                xv[u][0] = xv[u][1] = xv[u][2] = 0.f;
                if (i0 + u * 64u < end) {
                    const DY2 d2 = SOA ? dy[(size_t)level * N + i] : dy[(size_t)i * g.n_levels + level];
                    dv[u] = dy2f(d2);
                    xv[u][0] = x[3 * (size_t)i + 0];
                    xv[u][1] = x[3 * (size_t)i + 1];
                    xv[u][2] = x[3 * (size_t)i + 2];
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < kUnroll; ++u) {
                const float2 d = dv[u];
                bad = bad || !(fabsf(d.x) < INFINITY) || !(fabsf(d.y) < INFINITY);
                const bool lv = d.x != 0.f || d.y != 0.f;  // (out-of-range slots carry d = 0)
                const Corner c = grid_cell(scale, xv[u][0], xv[u][1], xv[u][2]);
                const float wy0 = 1.f - c.wy, wz0 = 1.f - c.wz;
                const float wyz[4] = {wy0 * wz0, c.wy * wz0, wy0 * c.wz, c.wy * c.wz};
                const uint32_t hy0 = c.py * 2654435761u, hy1 = hy0 + 2654435761u;
                const uint32_t hz0 = c.pz * 805459861u, hz1 = hz0 + 805459861u;
                const uint32_t a[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const uint32_t h = a[j] & mask;
                    q_push(lv && (h & ~(ACC::kEntries - 1u)) == first, (h & (ACC::kEntries - 1u)) | (c.px << 14),
                           wyz[j] * d.x, wyz[j] * d.y, c.wx);
                }
            }
        }
        while (q_fill) q_drain(min(q_fill, 64u));
    } else
    for (uint32_t c0 = wave_grab(64u * kUnroll); begin + c0 < end; c0 = wave_grab(64u * kUnroll)) {
        const uint32_t i0 = begin + c0 + lane_id;
        float2 dv[kUnroll];
        float xv[kUnroll][3];
        uint32_t sid[kUnroll];
#pragma unroll
        for (uint32_t u = 0; u < kUnroll; ++u) {  // (one extra round trip per pass when the list is used)
            const uint32_t j = i0 + u * 64u;
            sid[u] = j < end ? (listed ? live[j] : j) : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < kUnroll; ++u) {
            const uint32_t i = sid[u];
            dv[u] = make_float2(0.f, 0.f);
            xv[u][0] = xv[u][1] = xv[u][2] = 0.f;
            if (i0 + u * 64u < end) {
                const DY2 d2 = SOA ? dy[(size_t)level * N + i] : dy[(size_t)i * g.n_levels + level];
                dv[u] = dy2f(d2);
                xv[u][0] = x[3 * (size_t)i + 0];
                xv[u][1] = x[3 * (size_t)i + 1];
                xv[u][2] = x[3 * (size_t)i + 2];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < kUnroll; ++u) {
            const float2 d = dv[u];
            bad = bad || !(fabsf(d.x) < INFINITY) || !(fabsf(d.y) < INFINITY);
            if (d.x == 0.f && d.y == 0.f) continue;
            const Corner c = grid_cell(scale, xv[u][0], xv[u][1], xv[u][2]);
            const float wx0 = 1.f - c.wx, wy0 = 1.f - c.wy, wz0 = 1.f - c.wz;
            const float wyz[4] = {wy0 * wz0, c.wy * wz0, wy0 * c.wz, c.wy * c.wz};
            if (hashed) {
                // two integer multiplies per sample; the 4 (y, z) corner pairs are xor combinations
                const uint32_t hy0 = c.py * 2654435761u, hy1 = hy0 + 2654435761u;
                const uint32_t hz0 = c.pz * 805459861u, hz1 = hz0 + 805459861u;
                const uint32_t a[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
                if (pair_bins) {
                    // power-of-two slices and px + 1 < slice size: the x coordinate only touches index bits
                    // below the slice bits, so both x corners of a (y, z) pair are in the SAME slice and one
                    // test on the (y, z) hash decides both
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint32_t h = a[j] & mask;
                        if ((h & ~(ACC::kEntries - 1u)) == first) {
                            const float wj = wyz[j];
                            const uint32_t lo = h & (ACC::kEntries - 1u);
                            ACC::add(acc, lo ^ c.px, wx0 * wj * d.x, wx0 * wj * d.y, sc);
                            ACC::add(acc, lo ^ (c.px + 1u), c.wx * wj * d.x, c.wx * wj * d.y, sc);
                        }
                    }
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint32_t r0 = ((c.px ^ a[j]) & mask) - first, r1 = (((c.px + 1u) ^ a[j]) & mask) - first;
                        const float wj = wyz[j];
                        if (r0 < count) ACC::add(acc, r0, wx0 * wj * d.x, wx0 * wj * d.y, sc);
                        if (r1 < count) ACC::add(acc, r1, c.wx * wj * d.x, c.wx * wj * d.y, sc);
                    }
                }
            } else {
                // dense stride index: one base + adds; the wrap (positions on the upper domain face only) sits
                // behind a branch that is almost never taken
                const uint32_t base = c.px + c.py * res + c.pz * res2;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    uint32_t i0 = base + ((j & 1u) ? res : 0u) + ((j & 2u) ? res2 : 0u);
                    uint32_t i1 = i0 + 1u;
                    if (i1 >= size) {
                        i0 %= size;
                        i1 %= size;
                    }
                    const uint32_t r0 = i0 - first, r1 = i1 - first;  // unsigned wrap -> huge when below the slice
                    const float wj = wyz[j];
                    if (r0 < count) ACC::add(acc, r0, wx0 * wj * d.x, wx0 * wj * d.y, sc);
                    if (r1 < count) ACC::add(acc, r1, c.wx * wj * d.x, c.wx * wj * d.y, sc);
                }
            }
        }
    }
    GP_CLK(go2);
    __syncthreads();
    GP_CLK(go3);
    float* __restrict__ gr = grad + 2 * ((size_t)off + first);
    if (n_chunks == 1) {
        for (uint32_t e = threadIdx.x; e < 2 * count; e += kLdsBwdBlock) gr[e] = ACC::get(acc, e, sc);
    } else {
        for (uint32_t e = threadIdx.x; e < 2 * count; e += kLdsBwdBlock) {
            const float v = ACC::get(acc, e, sc);
            if (v != 0.f) atomicAdd(gr + e, v);
        }
    }
    // (no LDS to spare for a block-wide vote: one atomic per affected wave, after every plain store has retired)
    __syncthreads();
#ifdef NVO_GRID_PHASE
    if (threadIdx.x == 0) {
        GP_CLK(go4);
        const int o = hashed ? 24 : 16;
        GP_ADD(o + 0, go1 - go0); GP_ADD(o + 1, go2 - go1); GP_ADD(o + 2, go3 - go2); GP_ADD(o + 3, go4 - go3); GP_ADD(o + 4, 1);
        if (level < 5u) { GP_ADD(38 + level, go2 - go1); GP_ADD(43 + level, 1); }  // scan cycles / items per level (L <= 5 grids)
    }
#endif
    if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0u) {
        atomicAdd(gr, __builtin_nanf(""));
        if (nf_flag) atomicOr(nf_flag, 1u);  // (the optimiser's overflow flag, raised at the source)
    }
}

// L1 norm of dy per (level, feature) as 2^8 fixed point in u64 (deterministic: fixed per-thread order, integer
// atomics between workgroups).  grid = (blocks, n_levels); half2 dy only.
template <bool SOA, typename DY2>
__global__ void __launch_bounds__(256)
k_dy_l1(NvoGridLevels g, uint32_t N, const DY2* __restrict__ dy, unsigned long long* __restrict__ l1, uint32_t level_mask) {
    __shared__ float red[2][4];
    // blockIdx.y counts the levels of level_mask (the levels that have slice-owner items: the stream form keeps only
    // its coarse levels there, and the norms of the others would be read for nothing)
    uint32_t level = 0;
    for (uint32_t seen = 0, l = 0; l < g.n_levels; ++l)
        if ((level_mask >> l) & 1u) {
            if (seen == blockIdx.y) level = l;
            ++seen;
        }
    float a = 0.f, b = 0.f;
    if (SOA && (N & 3u) == 0u && (((uintptr_t)dy) & 15u) == 0u) {
        // level-major layout: a level's pairs are one contiguous stream -- 16-byte loads (4 samples), two in flight
        const uint4* __restrict__ p = reinterpret_cast<const uint4*>(dy + (size_t)level * N);
        const uint32_t n4 = N >> 2, stride = gridDim.x * 256;
        auto acc4 = [&](uint4 q) {
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float2 d = dy2f(__builtin_bit_cast(DY2, w[k]));
                a += fabsf(d.x);
                b += fabsf(d.y);
            }
        };
        uint32_t i = blockIdx.x * 256 + threadIdx.x;
        for (; i + stride < n4; i += 2 * stride) {
            const uint4 q0 = p[i], q1 = p[i + stride];
            acc4(q0);
            acc4(q1);
        }
        if (i < n4) acc4(p[i]);
    } else {
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
            const float2 d = dy2f(SOA ? dy[(size_t)level * N + i] : dy[(size_t)i * g.n_levels + level]);
            a += fabsf(d.x);
            b += fabsf(d.y);
        }
    }
    a = nvo_wave_sum(a);
    b = nvo_wave_sum(b);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        // round UP to the 2^-8 grid: the sum must not under-estimate (it bounds every accumulator)
        atomicAdd(&l1[2 * level + threadIdx.x], (unsigned long long)ceilf(t * 256.f) + 1ull);
    }
}

// Slice codes of the hashed levels in `levels` (bit l): codes[row][i] = the 16384-entry slice of each of sample i's four
// (y, z) corner pairs on that level, 3 bits each (pair j in bits 3 j .. 3 j + 2); row = rank of the level among the coded
// ones.  A level qualifies when it is hashed, a power-of-two multiple of 16384 entries with at most 8 slices, and every x
// coordinate stays below the slice bits (the slice-owner items' `pair_bins` condition): the slice is then a function of
// the (y, z) hash alone.  One pass over the positions; what the slice owners otherwise re-derive on every visit.
__global__ void __launch_bounds__(256)
k_slice_codes(NvoGridLevels g, uint32_t N, const float* __restrict__ x, uint32_t levels, uint16_t* __restrict__ codes) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
    const float px = x[3 * (size_t)i], py = x[3 * (size_t)i + 1], pz = x[3 * (size_t)i + 2];
    uint32_t row = 0;
    for (uint32_t l = 0; l < g.n_levels; ++l) {
        if (!((levels >> l) & 1u)) continue;  // (uniform)
        const uint32_t mask = g.offset[l + 1] - g.offset[l] - 1u;
        const Corner c = grid_cell(g.scale[l], px, py, pz);
        const uint32_t hy0 = c.py * 2654435761u, hy1 = hy0 + 2654435761u;
        const uint32_t hz0 = c.pz * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t a[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        uint32_t code = 0u;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) code |= (((a[j] & mask) >> 14) & 7u) << (3u * j);
        codes[(size_t)row * N + i] = (uint16_t)code;
        ++row;
    }
}

template <bool SOA, typename DY2>
__global__ void __launch_bounds__(kLdsBwdBlock)
k_grid_bwd_lds(NvoGridLevels g, uint32_t N, const float* __restrict__ x,
               const DY2* __restrict__ dy, float* __restrict__ grad,
               const uint4* __restrict__ items, const unsigned long long* __restrict__ l1,
               const uint32_t* __restrict__ live, uint32_t* __restrict__ nf_flag, const uint32_t* __restrict__ live_n,
               uint32_t ring_off, const float* __restrict__ ext_l1, uint32_t ext_blocks, uint32_t ext_stride,
               const uint16_t* __restrict__ codes, uint32_t code_levels, uint32_t list_pass) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint4 item = items[blockIdx.x];  // {level, first entry, chunk, n_chunks | accumulator-kind flags}
    const uint32_t n_chunks = item.w & 0x1FFFFFFFu;
    const bool merge = (item.w >> 29) & 1u;
    const uint32_t level = item.x & 0xFFu, cap = item.x >> 8;  // cap: entries per slice of this level
    // 32-bit accumulators, L1 norms delivered by the fused-MLP backward that wrote dy (NvoGridSlices::ext_l1): the
    // workgroups' partial sums are added up in a fixed order (same scale in every item and every run)
    __shared__ float s_l1[kLdsBwdBlock / 64][2];
    float l1e[2] = {0.f, 0.f};
    if (ext_l1 && ((item.w >> 30) & 1u) && !(item.w >> 31)) {
        float a = 0.f, b = 0.f;
        for (uint32_t blk = threadIdx.x; blk < ext_blocks; blk += kLdsBwdBlock) {
            a += ext_l1[(size_t)blk * ext_stride + 2 * level];
            b += ext_l1[(size_t)blk * ext_stride + 2 * level + 1];
        }
        a = nvo_wave_sum(a);
        b = nvo_wave_sum(b);
        if ((threadIdx.x & 63u) == 0u) {
            s_l1[threadIdx.x >> 6][0] = a;
            s_l1[threadIdx.x >> 6][1] = b;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kLdsBwdBlock / 64; ++w) {
            l1e[0] += s_l1[w][0];
            l1e[1] += s_l1[w][1];
        }
        // upper bound of the exact sum (fp32 summation error of <= a few thousand terms) -- the 2^29 scale leaves a
        // factor of four of headroom inside int32 on top of this
        l1e[0] = l1e[0] * 1.001f + 1e-30f;
        l1e[1] = l1e[1] * 1.001f + 1e-30f;
    }
    if (item.w >> 31) {
        grid_bwd_item<AccFloat, SOA, DY2>(g, N, x, dy, grad, level, item.y, item.z, n_chunks, lds_raw,
                                          AccScale{0.f, 0.f, 0.f, 0.f}, live, merge, cap, nf_flag, live_n, 0u, nullptr, list_pass);
    } else if ((item.w >> 30) & 1u) {
        const float l1x = ext_l1 ? l1e[0] : (float)l1[2 * level] * (1.f / 256.f);
        const float l1y = ext_l1 ? l1e[1] : (float)l1[2 * level + 1] * (1.f / 256.f);
        AccScale sc;
        sc.s0 = l1x > 0.f ? 536870912.f / l1x : 0.f;  // 2^29 / L1
        sc.s1 = l1y > 0.f ? 536870912.f / l1y : 0.f;
        sc.inv0 = l1x * (1.f / 536870912.f);
        sc.inv1 = l1y * (1.f / 536870912.f);
        // (slice codes of this level, k_slice_codes: row = number of coded levels below it)
        const uint16_t* lc = (codes && ((code_levels >> level) & 1u))
                                 ? codes + (size_t)__builtin_popcount(code_levels & ((1u << level) - 1u)) * N : nullptr;
        grid_bwd_item<AccFixed32, SOA, DY2>(g, N, x, dy, grad, level, item.y, item.z, n_chunks, lds_raw, sc, live,
                                            merge, cap, nf_flag, live_n, ring_off, lc, list_pass);
    } else {
        grid_bwd_item<AccFixed, SOA, DY2>(g, N, x, dy, grad, level, item.y, item.z, n_chunks, lds_raw,
                                          AccScale{0.f, 0.f, 0.f, 0.f}, live, merge, cap, nf_flag, live_n, 0u, nullptr, list_pass);
    }
}

// List of the samples whose dL/dy is non-zero on any level (NvoGridSlices::compact_live): *count = their number (a word
// of the module's permanent state, zeroed by the launcher or -- external_zero -- by the step's single zero launch),
// out[k] = sample id.  Workgroup-aggregated append; the order of the workgroups is not
// deterministic.  For the per-sample scan that only moves samples between the chunks of an item (integer accumulation
// is order-free, chunked items combine with float atomics either way); with the run-merging scan (grid_bwd_runs) a lane
// sums 8 LIST-consecutive samples in fp32 before the conversion, so which samples share a run -- hence the rounding of
// the run sums -- follows the append order: compact_live + runs is not bitwise reproducible run to run (the
// deterministic mode therefore scans all samples).
template <bool SOA, typename DY2>
__global__ void __launch_bounds__(1024)
k_live_samples(NvoGridLevels g, uint32_t N, const DY2* __restrict__ dy, uint32_t* __restrict__ count,
               uint32_t* __restrict__ out, unsigned long long* __restrict__ l1, const uint32_t* __restrict__ ext_live,
               uint32_t ext_blocks, const uint16_t* __restrict__ ext_dout) {
    // ext_live (NvoGridSlices::ext_live): the fused-MLP backward that wrote dy counted the samples with a non-zero
    // dL/doutput per workgroup.  While at least 3/4 of them are live the items scan all samples anyway (grid_bwd_item:
    // `listed`): nobody needs the list, every workgroup leaves at once and the length word says "all N".
    if (ext_live) {
        __shared__ uint32_t s_cnt[16];
        uint32_t c = 0u;
        for (uint32_t b = threadIdx.x; b < ext_blocks; b += 1024u) c += ext_live[b];
        c = (uint32_t)nvo_wave_sum((float)c);  // (exact: < 2^24 per wave)
        if ((threadIdx.x & 63u) == 0u) s_cnt[threadIdx.x >> 6] = c;
        __syncthreads();
        uint32_t tot = 0u;
        for (int w = 0; w < 16; ++w) tot += s_cnt[w];
        if (tot >= N - (N >> 2)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(count, N);
            return;
        }
    }
    // l1 != nullptr (32-bit accumulators, <= 8 levels): the per-level L1 norms of dy that k_dy_l1 computes are summed
    // in the same pass -- both kernels read every dy once, and each cost ~14 us per 1 M-sample launch
    // one workgroup per 4096 consecutive samples, ONE global atomic per workgroup (a wave-level append put 16 K atomics
    // on one address for a 1 M-sample batch: the single counter serialised them -- +40 us per launch)
    constexpr uint32_t kPer = 4;
    __shared__ uint32_t wave_cnt[16];
    __shared__ uint32_t block_base;
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * (1024u * kPer);
    bool live[kPer];
    uint32_t mine = 0;
    constexpr uint32_t kL1Levels = 8;
    float la[kL1Levels], lb[kL1Levels];
#pragma unroll
    for (uint32_t l = 0; l < kL1Levels; ++l) la[l] = lb[l] = 0.f;
    __shared__ float l1_red[16][kL1Levels][2];
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t i = first + q * 1024u + threadIdx.x;
        live[q] = false;
        if (i < N) {
            if (ext_dout && !l1) {
                // (NvoGridSlices::ext_dout: dL/doutput of the network in front, one 16-bit value per sample -- dy = W^T dZ is
                // exactly zero on every level where it is zero, so the list built from it holds every sample the level-wise
                // test would list (and the few whose products all underflowed, which the scans skip): 2 bytes per sample
                // instead of 4 per sample and level)
                live[q] = (ext_dout[i] & 0x7FFFu) != 0u;
            } else if (l1) {
#pragma unroll
                for (uint32_t l = 0; l < kL1Levels; ++l) {
                    if (l < g.n_levels) {
                        const float2 d = dy2f(SOA ? dy[(size_t)l * N + i] : dy[(size_t)i * g.n_levels + l]);
                        live[q] = live[q] || d.x != 0.f || d.y != 0.f;
                        la[l] += fabsf(d.x);
                        lb[l] += fabsf(d.y);
                    }
                }
            } else {
                for (uint32_t l = 0; l < g.n_levels; ++l) {
                    const float2 d = dy2f(SOA ? dy[(size_t)l * N + i] : dy[(size_t)i * g.n_levels + l]);
                    live[q] = live[q] || d.x != 0.f || d.y != 0.f;  // (NaN != 0: non-finite gradients stay listed)
                }
            }
        }
        mine += (uint32_t)__popcll(__ballot(live[q]));  // (wave total of pass q, same in every lane)
    }
    if (l1) {
#pragma unroll
        for (uint32_t l = 0; l < kL1Levels; ++l) {
            if (l < g.n_levels) {  // (wave-uniform)
                const float a = nvo_wave_sum(la[l]), b = nvo_wave_sum(lb[l]);
                if (lane == 0u) {
                    l1_red[wib][l][0] = a;
                    l1_red[wib][l][1] = b;
                }
            }
        }
    }
    if (lane == 0u) wave_cnt[wib] = mine;
    __syncthreads();
    if (l1 && threadIdx.x < 2u * g.n_levels) {
        const uint32_t l = threadIdx.x >> 1, f = threadIdx.x & 1u;
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += l1_red[w][l][f];
        // round UP to the 2^-8 grid: the sum must not under-estimate (it bounds every accumulator) -- as k_dy_l1
        atomicAdd(&l1[2 * l + f], (unsigned long long)ceilf(t * 256.f) + 1ull);
    }
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < 16; ++w) {
            const uint32_t c = wave_cnt[w];
            wave_cnt[w] = tot;
            tot += c;
        }
        block_base = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    uint32_t pos = block_base + wave_cnt[wib];
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const unsigned long long m = __ballot(live[q]);
        if (live[q]) out[pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = first + q * 1024u + threadIdx.x;
        pos += (uint32_t)__popcll(m);
    }
}

// The same list from the dL/doutput ROWS of the network in front (16 16-bit values per sample) and the tile bytes of the
// kernel that wrote them (NvoGridSlices::ext_tile_live): a sample whose row is all zero has dL/d(encoded) = W^T dZ = 0 on
// every level, and a tile without its bits holds only such rows and is not read at all.  On a trained field one sample
// of a ray's 48 carries the weight: a third of the main level's tiles are live, 2 % of its samples, and the record
// scatter below spends its time on workgroups of 512 samples of which ten do anything.
__global__ void __launch_bounds__(1024)
k_live_rows(uint32_t N, const uint8_t* __restrict__ tile_live, uint32_t bits, const uint4* __restrict__ rows,
            const float* __restrict__ tile_count, uint32_t* __restrict__ count, uint32_t* __restrict__ out) {
    constexpr uint32_t kPer = 4;
    __shared__ uint32_t wave_cnt[16];
    __shared__ uint32_t block_base;
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    if (tile_count) {  // (uniform over the launch: every workgroup sums the same 64 words)
        const float c = nvo_wave_sum(tile_count[8u * lane]);
        if (c >= (float)((N >> 4) - (N >> 6))) {  // (3/4 of the tiles, as the consumers' rule for the list itself)
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(count, N);  // "all N": the consumers scan every sample
            return;
        }
    }
    const uint32_t first = blockIdx.x * (1024u * kPer);
    bool live[kPer];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const uint32_t i = first + q * 1024u + threadIdx.x;
        live[q] = false;
        if (i < N && (tile_live[i >> 4] & bits) != 0u) {
            const uint4 a = rows[2 * (size_t)i], b = rows[2 * (size_t)i + 1];
            live[q] = ((a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) & 0x7FFF7FFFu) != 0u;
        }
        mine += (uint32_t)__popcll(__ballot(live[q]));
    }
    if (lane == 0u) wave_cnt[wib] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < 16; ++w) {
            const uint32_t c = wave_cnt[w];
            wave_cnt[w] = tot;
            tot += c;
        }
        block_base = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    uint32_t pos = block_base + wave_cnt[wib];
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
        const unsigned long long m = __ballot(live[q]);
        if (live[q]) out[pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = first + q * 1024u + threadIdx.x;
        pos += (uint32_t)__popcll(m);
    }
}

// ------------------------------------------------------------------------------------------
// backward w.r.t. parameters, STREAMED form (mode 3): pair records sorted by bin inside each tile, streaming accumulate
// ------------------------------------------------------------------------------------------
// The slice-owner kernel above re-derives every sample's corner hashes once per slice of a level (26-64x redundancy on a
// 2^19-entry table).  The streamed form derives them once: a scatter pass over (tile, level) writes self-contained records,
// an accumulate pass streams the records of one bin into LDS and writes the slice with plain stores.  (History, removed in
// round 5: mode 2 -- count / scan / scatter of 4-byte (sample, corner) records and an accumulate pass bound by the
// dependent gathers behind every record; globally bin-sorted 8-byte records with count / scan passes; tile-local 8-byte
// records with 64-bit accumulators over 4096-entry bins.  What is left is the fastest of them.)
// Slice size the streamed / owner split is decided in (nvo_grid_stream_create: levels with few such bins stay slice-owner).
constexpr uint32_t kBinSlice = 4096;

template <bool SOA, typename DY2>
__device__ __forceinline__ bool load_dy_nonzero(const NvoGridLevels& g, const DY2* __restrict__ dy, uint32_t N,
                                                uint32_t level, uint32_t i, float2* out) {
    const DY2 d2 = SOA ? dy[(size_t)level * N + i] : dy[(size_t)i * g.n_levels + level];
    *out = dy2f(d2);
    return out->x != 0.f || out->y != 0.f;
}

__device__ __forceinline__ uint32_t hashed_corner(const Corner& c, uint32_t k, uint32_t mask) {
    return ((c.px + (k & 1u)) ^ ((c.py + ((k >> 1) & 1u)) * 2654435761u) ^ ((c.pz + ((k >> 2) & 1u)) * 805459861u)) & mask;
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int) { return nvo_wave_incl_scan(v); }  // DPP

// Work items of the accumulate pass = {bin, chunk | n_chunks << 16, streamed-level index | level << 8, slice}: the chunk
// walks tiles [chunk * per, (chunk + 1) * per) of its bin.  The header is self-contained (no bin-table hop), the NEXT
// item's header and segment words are requested while the current item streams, every wave owns a contiguous block of the
// item's tiles and reads its runs as one concatenated record stream, all of whose loads are in flight at once.
struct TlItem {
    uint32_t bin, chunk, n_chunks, lvl, level, slice, t0, t1;
};
__device__ __forceinline__ TlItem tl_decode(uint4 h, uint32_t n_tiles) {
    TlItem I;
    I.bin = h.x;
    I.chunk = h.y & 0xFFFFu;
    I.n_chunks = h.y >> 16;
    I.lvl = h.z & 0xFFu;
    I.level = h.z >> 8;
    I.slice = h.w;
    const uint32_t per = I.n_chunks ? (n_tiles + I.n_chunks - 1u) / I.n_chunks : 0u;  // (n_chunks == 0: padding item)
    I.t0 = min(n_tiles, I.chunk * per);
    I.t1 = min(n_tiles, I.t0 + per);
    return I;
}

// ---- tile-local layout, PACKED accumulators (NvoGridStream::acc_bits == 32; round 3) ---------------------------
// k_tl_accumulate spends an entry's two features in two 64-bit LDS atomics on 16 bytes of accumulator.  Here an entry is
// ONE 64-bit word holding two 32-bit fixed-point sums, lo = feature 0 and hi = feature 1, updated by ONE atomic add of
// X = a + b * 2^32 (two's complement: the low half's borrow rides into the high half and comes back out when the
// sums are separated: lo = (int32)X, hi = (X - lo) >> 32 -- exact while both sums stay inside int32).  Half the LDS
// atomics, half the zeroing and flushing, and 8 bytes per entry: a bin holds 8192 entries in the 64 KiB that 4096 needed,
// so a bin's runs are twice as long (~64 records = one full wave load) and there are half as many items.
// The fixed-point scale must be data-derived and overflow-proof.  |sum over an entry| <= L1(bin) = sum of |w * dy| over
// every record of the bin, and the SCATTER can deliver that bound for free: its rank atomic (one LDS atomic per x-corner
// pair that returns the pair's position inside its (tile, bin) run) becomes a 64-bit add whose upper fields sum the
// records' magnitudes, quantised UP against the tile's max |dy|:
//     hist64[bin] += count | q0 << 16 | q1 << 40,   q_f = ceil(|v_f| * 1023 / M_f)  (<= 1024; <= 4096 records per tile)
// The tile writes L1_f(tile, bin) <= S_f * M_f / 1023, rounded UP to bf16, next to its segment word; the accumulate item
// sums the words of its tiles (fixed order) and scales by 2^29 / L1.  Everything that feeds the scale is integer or
// fixed-order arithmetic, so single-chunk bins stay bitwise reproducible.
constexpr uint32_t kBinP = 8192;   // largest bin of the packed form (13-bit entry index in a record)
constexpr uint32_t kBinPSmall = 6176;  // (experiment, NVO_TL_BIN=6176: 85 bins per 2^19 table; measured slower)
constexpr int kTlBlockP = 512;
// wave loads of pair records per pass (a hashed item's 48 runs x ~32 pairs per wave = 24 loads).  14 is what 128 VGPRs
// hold (5 per window); 24 in one pass needs 175 and halves the occupancy (81 us), 256-thread workgroups with 32 windows
// measured 69.7 us against 52.2
constexpr uint32_t kTlWinP = 14;

template <uint32_t BIN>
__device__ __forceinline__ uint32_t st_bin_entries_p(const NvoGridLevels& g, uint32_t level, uint32_t slice) {
    const uint32_t size = g.offset[level + 1] - g.offset[level];
    return min(BIN, size - slice * BIN);
}

template <uint32_t BIN>
__global__ void __launch_bounds__(256)
k_st_zero_p(NvoGridLevels g, const uint32_t* __restrict__ bin_level, const uint32_t* __restrict__ bin_slice,
            const uint32_t* __restrict__ bin_chunks, float* __restrict__ grad) {
    if (bin_chunks[blockIdx.x] <= 1u) return;
    const uint32_t level = bin_level[blockIdx.x], slice = bin_slice[blockIdx.x];
    const uint32_t n = 2 * st_bin_entries_p<BIN>(g, level, slice);
    float* __restrict__ gr = grad + 2 * ((size_t)g.offset[level] + (size_t)slice * BIN);
    for (uint32_t e = threadIdx.x; e < n; e += 256) gr[e] = 0.f;
}

// maximum over the 64 lanes of a NON-NEGATIVE value, in every lane (DPP steps as nvo_wave_sum: a lane without a source
// reads 0, the identity for non-negative values)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = fmaxf(v, nvo_dpp_f32<0x128>(v));
    v = fmaxf(v, nvo_dpp_f32<0x124>(v));
    v = fmaxf(v, nvo_dpp_f32<0x122>(v));
    v = fmaxf(v, nvo_dpp_f32<0x121>(v));
    v = fmaxf(v, nvo_dpp_f32<0x142, 0xA>(v));
    v = fmaxf(v, nvo_dpp_f32<0x143, 0xC>(v));
    return nvo_wave_bcast(v, 63);
}

__device__ __forceinline__ uint32_t bf16_up(float v) {  // smallest bfloat16 >= v, v >= 0 and finite
    const uint32_t u = __float_as_uint(v) + 0xFFFFu;
    return u >> 16;
}

// PAIR records (12 bytes).  Both x corners of a (y, z) pair of a cell fall into the same bin (hashed levels: their
// indices differ by the xor mask px ^ (px + 1), which stays below the bin size; dense levels: they are neighbours), so
// one record carries the pair: u_f = w_y w_z dy_f (the two features, fp32 with the low 6 / 7 mantissa bits replaced by
// the first corner's 13-bit entry offset, as the per-corner records of k_tl_scatter), the x weight as a 16-bit fraction
// and a 4-bit code for the second corner's offset (0..12: rel ^ ((2 << code) - 1), 15: rel + 1, 14: no second corner).
// The accumulate item splits u into (1 - w) u and w u.  Three quarters of the record bytes and half the records of the
// per-corner form; a pair that does straddle two bins (a dense level's bin boundary, resolutions beyond the bin size)
// is written as two single-corner records.
constexpr uint32_t kPairSingle = 14u, kPairNext = 15u;
struct PairRec {
    uint32_t w0, w1, w2;
};
__device__ __forceinline__ PairRec pair_pack(uint32_t rel, float u0, float u1, uint32_t wq, uint32_t code) {
    const uint32_t a = __float_as_uint(u0) + 0x20u;  // round to nearest at bit 6
    const uint32_t b = __float_as_uint(u1) + 0x40u;  // ... at bit 7
    return PairRec{(a & ~0x3Fu) | (rel & 0x3Fu), (b & ~0x7Fu) | (rel >> 6), (wq & 0xFFFFu) | (code << 16)};
}

template <int TILE, bool SOA, typename DY2, uint32_t BIN, bool LISTED>
__global__ void __launch_bounds__(TILE)
k_tl_scatter_p(NvoGridLevels g, uint32_t N, const float* __restrict__ x, const DY2* __restrict__ dy,
               const uint32_t* __restrict__ st_levels, const uint32_t* __restrict__ bin_first,
               uint32_t* __restrict__ seg, uint32_t* __restrict__ segl1, uint32_t* __restrict__ records,
               uint32_t* __restrict__ nf_flag, const uint32_t* __restrict__ live_list, const uint32_t* __restrict__ live_n) {
    constexpr uint32_t kStBlock = TILE;
    // capacity: every pair split into two single-corner records.  (Staging only TILE * 4 records in LDS -- four workgroups
    // per CU instead of three -- with the overflow written straight to the region measured SLOWER: 42.5 -> 48.5 us.)
    constexpr uint32_t kStRecords = TILE * 8;
    constexpr uint32_t kWaves = TILE / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(lds_raw);  // [kStRecords][3]
    const uint32_t level = st_levels[blockIdx.y], tile = blockIdx.x, n_tiles = gridDim.x;
    const uint32_t bin0 = bin_first[blockIdx.y];
    const uint32_t n_slices = bin_first[blockIdx.y + 1] - bin0;
    const uint32_t size = g.offset[level + 1] - g.offset[level];
    const uint32_t res = g.resolution[level], hashed = g.hashed[level];
    unsigned long long* hist = reinterpret_cast<unsigned long long*>(stage + 3 * kStRecords);
    uint32_t* loff = reinterpret_cast<uint32_t*>(hist + n_slices);
    __shared__ uint32_t total_s;
    __shared__ float wmax[kWaves][2];
    // LISTED (live_list: k_live_rows / the network's backward / k_live_samples): the tiles are cut from the list of samples
    // that carry a gradient -- as the slice-owner items, only while it is shorter than 3/4 N; tiles past its end write
    // nothing and leave (k_tl_accumulate_p<.., true> skips their words).  A separate instantiation (see there).
    uint32_t i = tile * kStBlock + threadIdx.x;
    if constexpr (LISTED) {
        const uint32_t n_listed = *live_n;
        const bool listed = n_listed < N - (N >> 2);
        const uint32_t n_scan = listed ? n_listed : N;
        if (tile * kStBlock >= n_scan) return;  // (workgroup-uniform, before any barrier)
        i = i < n_scan ? (listed ? live_list[i] : i) : N;
    }
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    float2 d = make_float2(0.f, 0.f);
    float xs[3] = {0.f, 0.f, 0.f};
    bool live = false;
    GP_CLK(gq0);
    if (i < N) {  // dy and x in one round trip
        live = load_dy_nonzero<SOA, DY2>(g, dy, N, level, i, &d);
        xs[0] = x[3 * (size_t)i + 0];
        xs[1] = x[3 * (size_t)i + 1];
        xs[2] = x[3 * (size_t)i + 2];
    }
    for (uint32_t b = threadIdx.x; b < n_slices; b += kStBlock) hist[b] = 0ull;
    {   // tile maxima of |dy| per feature (non-finite values are flagged below and poison the level: treat them as 0 here)
        const bool fin = fabsf(d.x) < INFINITY && fabsf(d.y) < INFINITY;
        float m0 = fin ? fabsf(d.x) : 0.f, m1 = fin ? fabsf(d.y) : 0.f;
        m0 = wave_max_nonneg(m0);
        m1 = wave_max_nonneg(m1);
        if (lane == 0) {
            wmax[wib][0] = m0;
            wmax[wib][1] = m1;
        }
    }
    __syncthreads();
    GP_CLK(gq1);
    float M0 = 0.f, M1 = 0.f;
#pragma unroll
    for (uint32_t w = 0; w < kWaves; ++w) {
        M0 = fmaxf(M0, wmax[w][0]);
        M1 = fmaxf(M1, wmax[w][1]);
    }
    const float q0s = M0 > 0.f ? 1023.f / M0 : 0.f, q1s = M1 > 0.f ? 1023.f / M1 : 0.f;
    // per (y, z) pair j: one pair record (n_rec = 1) or two single-corner records (n_rec = 2)
    uint32_t bin_a[4], bin_b[4], slot_a[4], slot_b[4];
    PairRec rec_a[4], rec_b[4];
    bool split[4];
    const bool finite = fabsf(d.x) < INFINITY && fabsf(d.y) < INFINITY;
    if (live) {
        const Corner c = grid_cell(g.scale[level], xs[0], xs[1], xs[2]);
        const uint32_t wq = min(65535u, (uint32_t)__float2int_rn(c.wx * 65536.f));
        const float fb = (float)wq * (1.f / 65536.f), fa = 1.f - fb;
        // magnitude of a record in units of M / 1023, rounded UP (+1 covers the rounding of the product); <= 1024
        auto quant = [&](float v, float qs) -> unsigned long long {
            return finite ? (unsigned long long)min(1024u, (uint32_t)(fabsf(v) * qs) + 1u) : 0ull;
        };
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t cy = c.py + (j & 1u), cz = c.pz + (j >> 1);
            const uint32_t i0 = nvo_grid_index(hashed, size, res, c.px, cy, cz);
            const uint32_t i1 = nvo_grid_index(hashed, size, res, c.px + 1u, cy, cz);
            const float wyz = ((j & 1u) ? c.wy : 1.f - c.wy) * ((j & 2u) ? c.wz : 1.f - c.wz);
            const float u0 = wyz * d.x, u1 = wyz * d.y;
            const uint32_t b0 = i0 / BIN, b1 = i1 / BIN, r0 = i0 % BIN, r1 = i1 % BIN;
            bin_a[j] = b0;
            bin_b[j] = b1;
            // second corner's offset from the first: xor with a low mask (hashed levels) or + 1 (dense levels)
            const uint32_t xm = r0 ^ r1;
            uint32_t code = kPairSingle;
            if (b0 == b1) {
                if (r1 == r0 + 1u) code = kPairNext;
                else if ((xm & (xm + 1u)) == 0u && xm != 0u && xm <= 0x1FFFu) code = 31u - (uint32_t)__builtin_clz(xm);
            }
            split[j] = code == kPairSingle;
            if (!split[j]) {
                // rank inside (tile, bin) + magnitude sums: ONE LDS atomic per pair (|v_a| + |v_b| = |u|)
                rec_a[j] = pair_pack(r0, u0, u1, wq, code);
                slot_a[j] = (uint32_t)(atomicAdd(&hist[b0], 1ull | (quant(u0, q0s) << 16) | (quant(u1, q1s) << 40)) & 0xFFFFull);
            } else {
                const float a0 = fa * u0, a1 = fa * u1, c0 = fb * u0, c1 = fb * u1;
                rec_a[j] = pair_pack(r0, a0, a1, 0u, kPairSingle);
                rec_b[j] = pair_pack(r1, c0, c1, 0u, kPairSingle);
                slot_a[j] = (uint32_t)(atomicAdd(&hist[b0], 1ull | (quant(a0, q0s) << 16) | (quant(a1, q1s) << 40)) & 0xFFFFull);
                slot_b[j] = (uint32_t)(atomicAdd(&hist[b1], 1ull | (quant(c0, q0s) << 16) | (quant(c1, q1s) << 40)) & 0xFFFFull);
            }
        }
    }
    GP_CLK(gq2);
    const uint32_t bad_bit = __syncthreads_or(live && !finite) ? 0x80000000u : 0u;
    // (the accumulate pass poisons the chunk's gradient and raises the flag as well; raised HERE the verdict is final
    // before that pass starts, which lets it take the optimiser step of the entries it sums -- NvoGridAdam)
    if (bad_bit && nf_flag && threadIdx.x == 0) atomicOr(nf_flag, 1u);
    GP_CLK(gq3);
    if (threadIdx.x < 64) {  // wave 0: exclusive scan over the bins of this level
        uint32_t carry = 0;
        for (uint32_t b0 = 0; b0 < n_slices; b0 += 64) {
            const uint32_t b = b0 + lane;
            const unsigned long long h = b < n_slices ? hist[b] : 0ull;
            const uint32_t cnt = (uint32_t)(h & 0xFFFFull);
            const uint32_t incl = wave_incl_scan_u32(cnt, (int)lane);
            if (b < n_slices) {
                loff[b] = carry + incl - cnt;
                seg[(size_t)(bin0 + b) * n_tiles + tile] = (carry + incl - cnt) | (cnt << 16) | bad_bit;
                const float l0 = (float)((uint32_t)(h >> 16) & 0xFFFFFFu) * (M0 * (1.f / 1023.f));
                const float l1 = (float)(uint32_t)(h >> 40) * (M1 * (1.f / 1023.f));
                segl1[(size_t)(bin0 + b) * n_tiles + tile] = bf16_up(l0 * 1.0001f) | (bf16_up(l1 * 1.0001f) << 16);
            }
            carry += nvo_wave_bcast(incl, 63);
        }
        if (lane == 0) total_s = carry;
    }
    __syncthreads();
    GP_CLK(gq4);
    if (live) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            uint32_t* r = stage + 3u * (loff[bin_a[j]] + slot_a[j]);
            r[0] = rec_a[j].w0;
            r[1] = rec_a[j].w1;
            r[2] = rec_a[j].w2;
            if (split[j]) {
                uint32_t* r2 = stage + 3u * (loff[bin_b[j]] + slot_b[j]);
                r2[0] = rec_b[j].w0;
                r2[1] = rec_b[j].w1;
                r2[2] = rec_b[j].w2;
            }
        }
    }
    __syncthreads();
    GP_CLK(gq5);
    const uint32_t total = total_s;
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(records + 3u * ((size_t)blockIdx.y * n_tiles + tile) * kStRecords);
    const uint4* src = reinterpret_cast<const uint4*>(stage);
    for (uint32_t t = threadIdx.x; t < (3u * total + 3u) / 4u; t += kStBlock) dst[t] = src[t];
#ifdef NVO_GRID_PHASE
    if (threadIdx.x == 64) {
        GP_CLK(gq6);
        GP_ADD(8, gq1 - gq0); GP_ADD(9, gq2 - gq1); GP_ADD(10, gq3 - gq2); GP_ADD(11, gq4 - gq3); GP_ADD(12, gq5 - gq4);
        GP_ADD(13, gq6 - gq5); GP_ADD(14, 1);
    }
#endif
}

// (second bound: FOUR waves per SIMD, i.e. two of these workgroups per CU -- with the fused optimiser step the allocator
// otherwise takes 147 registers and only one workgroup fits: half the waves for a pass that is bound by waves x latency)
// LISTED: the scatter cut its tiles from a list of `*live_n` samples (k_tl_scatter_p<..., true>) and tiles past the list's end
// wrote nothing -- their words are stale and count as empty.  A separate instantiation: the pass runs at its scalar-register
// limit (24 v_writelane in the plain form), and the three values the bound needs, kept through the item loop, tripled
// the spills and cost every launch 12 us (86 -> 99 us alone) -- whether a list was in use or not.
template <uint32_t BIN, bool LISTED>
__global__ void __launch_bounds__(kTlBlockP, 4)
k_tl_accumulate_p(NvoGridLevels g, const uint4* __restrict__ items, uint32_t n_items, const uint32_t* __restrict__ seg,
                  const uint32_t* __restrict__ segl1, const uint32_t* __restrict__ records, uint32_t n_tiles,
                  uint32_t tile_records, float* __restrict__ grad, uint32_t* __restrict__ nf_flag, NvoGridAdam adam,
                  const uint32_t* __restrict__ live_n, uint32_t N) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds_raw);
    constexpr uint32_t kWaves = kTlBlockP / 64;
    // fused optimiser step (NvoGridAdam): the scalars of this launch, read once
    NvoAdamHyper ah{adam.lr, adam.beta1, adam.beta2, adam.eps, adam.bias1, adam.bias2_sqrt, adam.grad_scale, 0.f};
    bool adam_skip = false;
    float ema_keep = 0.f, ema_take = 0.f, ema_inv = 1.f;  // (k_ema_update_dev's factors, same expressions)
    if (adam.params && adam.ema) {
        const double d = (double)adam.ema_decay, t = (double)(*adam.ema_step_dev) + 1.0;
        ema_keep = (float)(d * (1.0 - pow(d, t - 1.0)));
        ema_inv = (float)(1.0 / (1.0 - pow(d, t)));
        ema_take = 1.0f - adam.ema_decay;
    }
    if (adam.params) {
        if (adam.hyper_dev) ah.lr = adam.hyper_dev[0];
        if (adam.bias_dev) {
            ah.bias1 = adam.bias_dev[0];
            ah.bias2_sqrt = adam.bias_dev[1];
        }
        if (adam.loss_scale_dev) ah.grad_scale = 1.0f / *adam.loss_scale_dev;
        adam_skip = adam.skip_flag && *adam.skip_flag != 0u;
    }
    __shared__ float wpart[kWaves][2];
    __shared__ uint32_t run_incl[kWaves][64], run_base[kWaves][64];  // per wave: the runs of its current tile block
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t it = blockIdx.x;
    if (it >= n_items) return;
    uint32_t n_used = n_tiles;
    if constexpr (LISTED) {
        const uint32_t n_listed = *live_n;  // (k_tl_scatter_p's rule for the list)
        if (n_listed < N - (N >> 2)) n_used = (n_listed + (tile_records >> 3) - 1u) / (tile_records >> 3);  // (tile = tile_records / 8 samples)
    }
    auto words_first = [&](const TlItem& I, uint32_t* l1w) -> uint32_t {
        const uint32_t n_span = I.t1 - I.t0;
        const uint32_t per_wave = (n_span + kWaves - 1u) / kWaves;
        const uint32_t first = min(n_span, wib * per_wave);
        const uint32_t n_mine = min(per_wave, n_span - first);
        const bool mine = lane < min(64u, n_mine) && (!LISTED || I.t0 + first + lane < n_used);
        const size_t o = (size_t)I.bin * n_tiles + I.t0 + first + lane;
        *l1w = mine ? segl1[o] : 0u;
        return mine ? seg[o] : 0u;
    };
    TlItem cur = tl_decode(items[it], n_tiles);
    uint32_t l1w = 0u;
    uint32_t segw = cur.n_chunks ? words_first(cur, &l1w) : 0u;
    for (;;) {
        const uint32_t it_next = it + gridDim.x;
        const bool has_next = it_next < n_items;
        const uint4 head_next = items[has_next ? it_next : it];  // in flight while the accumulators are zeroed
        if (cur.n_chunks == 0u) {  // padding of the balanced work list (n_chunks == 0): nothing to do in this round
            if (!has_next) break;
            it = it_next;
            cur = tl_decode(head_next, n_tiles);
            segw = cur.n_chunks ? words_first(cur, &l1w) : 0u;
            continue;
        }
        const uint32_t entries = st_bin_entries_p<BIN>(g, cur.level, cur.slice);
        float* __restrict__ gr = grad + 2 * ((size_t)g.offset[cur.level] + (size_t)cur.slice * BIN);
        GP_CLK(gp0);
        {
            uint4* z = reinterpret_cast<uint4*>(lds_raw);  // one uint4 = two entries
            for (uint32_t e = threadIdx.x; e < (entries + 1u) / 2u; e += kTlBlockP) z[e] = make_uint4(0u, 0u, 0u, 0u);
        }
        const uint32_t* __restrict__ rec_lvl = records + 3u * ((size_t)cur.lvl * n_tiles * tile_records);
        const uint32_t n_span = cur.t1 - cur.t0;
        const uint32_t per_wave = (n_span + kWaves - 1u) / kWaves;
        const uint32_t tile_first = cur.t0 + min(n_span, wib * per_wave);
        const uint32_t n_mine = min(per_wave, cur.t1 - tile_first);
        // ---- the bin's L1 bound: this wave's tiles (lane j: tile j; further blocks of 64 tiles in order), then the waves
        {
            float a0 = __uint_as_float(l1w << 16), a1 = __uint_as_float(l1w & 0xFFFF0000u);
            for (uint32_t j0 = 64u; j0 < n_mine; j0 += 64u) {
                const uint32_t w2 = (lane < min(64u, n_mine - j0) && (!LISTED || tile_first + j0 + lane < n_used)) ? segl1[(size_t)cur.bin * n_tiles + tile_first + j0 + lane] : 0u;
                a0 += __uint_as_float(w2 << 16);
                a1 += __uint_as_float(w2 & 0xFFFF0000u);
            }
            a0 = nvo_wave_sum(a0);
            a1 = nvo_wave_sum(a1);
            if (lane == 0) {
                wpart[wib][0] = a0;
                wpart[wib][1] = a1;
            }
        }
        __syncthreads();  // accumulators zeroed, L1 partials visible
        GP_CLK(gp1);
        float L0 = 0.f, L1 = 0.f;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) {
            L0 += wpart[w][0];
            L1 += wpart[w][1];
        }
        // |sum over an entry| <= L (a pair's two shares add up to |u|); rounding of each add <= 0.5: n <= 2^20 adds
        // leave 2^29 + 2^19 < 2^31
        const float s0 = L0 > 0.f ? 536870912.f / L0 : 0.f, s1 = L1 > 0.f ? 536870912.f / L1 : 0.f;
        const float inv0 = L0 * (1.f / 536870912.f), inv1 = L1 * (1.f / 536870912.f);
        bool bad = false;
        auto add = [&](const PairRec& r) {
            const uint32_t rel = (r.w0 & 0x3Fu) | ((r.w1 & 0x7Fu) << 6);
            const uint32_t code = (r.w2 >> 16) & 0xFu;
            const float u0 = __uint_as_float(r.w0 & ~0x3Fu) * s0, u1 = __uint_as_float(r.w1 & ~0x7Fu) * s1;
            const float fb = code == kPairSingle ? 0.f : (float)(r.w2 & 0xFFFFu) * (1.f / 65536.f), fa = 1.f - fb;
            {
                const long long X = (long long)__float2int_rn(u0 * fa) + ((long long)__float2int_rn(u1 * fa) << 32);
                atomicAdd(&acc[rel], (unsigned long long)X);
            }
            if (code != kPairSingle) {
                const uint32_t rel_b = code == kPairNext ? rel + 1u : rel ^ ((2u << code) - 1u);
                const long long X = (long long)__float2int_rn(u0 * fb) + ((long long)__float2int_rn(u1 * fb) << 32);
                atomicAdd(&acc[rel_b], (unsigned long long)X);
            }
        };
        TlItem nxt = cur;
        uint32_t segw_next = 0u, l1w_next = 0u;
        bool next_requested = false;
        for (uint32_t j0 = 0; j0 < n_mine; j0 += 64u) {
            const uint32_t n_here = min(64u, n_mine - j0);
            if (j0 > 0u) segw = (lane < n_here && (!LISTED || tile_first + j0 + lane < n_used)) ? seg[(size_t)cur.bin * n_tiles + tile_first + j0 + lane] : 0u;
            const uint32_t cnt = lane < n_here ? (segw >> 16) & 0x7FFFu : 0u;
            bad |= lane < n_here && (segw >> 31) != 0u;
            const uint32_t incl = wave_incl_scan_u32(cnt, (int)lane);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t base = (tile_first + j0 + lane) * tile_records + (segw & 0xFFFFu) - (incl - cnt);
            // record v of the concatenated runs belongs to the first run r with incl[r] > v: a 6-step binary search in
            // the wave's 64-entry table in LDS, all windows of a pass searched side by side (the dependent readlane
            // walk over run boundaries this replaces cost as many cycles as the loads and atomics together)
            run_incl[wib][lane] = incl;
            run_base[wib][lane] = base;
            for (uint32_t q0 = 0; q0 * 64u < total; q0 += kTlWinP) {
                PairRec rec[kTlWinP];
#pragma unroll
                for (uint32_t u = 0; u < kTlWinP; ++u) {
                    const uint32_t v = (q0 + u) * 64u + lane;
                    uint32_t r = 0u;
#pragma unroll
                    for (uint32_t step = 32u; step >= 1u; step >>= 1) r += run_incl[wib][r + step - 1u] <= v ? step : 0u;
                    const uint32_t my_base = run_base[wib][min(r, 63u)];
                    const uint32_t* __restrict__ rw = rec_lvl + 3u * (v < total ? my_base + v : 0u);  // one 12-byte load
                    rec[u] = PairRec{rw[0], rw[1], rw[2]};
                }
                if (!next_requested) {  // the next item's segment / L1 words ride behind this item's loads
                    next_requested = true;
                    if (has_next) {
                        nxt = tl_decode(head_next, n_tiles);
                        if (nxt.n_chunks) segw_next = words_first(nxt, &l1w_next);
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kTlWinP; ++u)
                    if ((q0 + u) * 64u + lane < total) add(rec[u]);
            }
        }
        if (!next_requested && has_next) {  // (a wave without records in this item)
            nxt = tl_decode(head_next, n_tiles);
            if (nxt.n_chunks) segw_next = words_first(nxt, &l1w_next);
        }
        GP_CLK(gp2);
        __syncthreads();
        GP_CLK(gp3);
        {
            // flush: two entries (four gradient scalars) per thread and step: one 16-byte LDS read, one 16-byte store
            const uint32_t n2 = entries >> 1;  // (entries of a bin are a multiple of 8)
            float4* __restrict__ gr4 = reinterpret_cast<float4*>(gr);
            const ulonglong2* acc2 = reinterpret_cast<const ulonglong2*>(acc);
            auto split = [&](unsigned long long w, float* f0, float* f1) {
                const long long X = (long long)w;
                const int lo = (int)X;
                const int hi = (int)((X - (long long)lo) >> 32);
                *f0 = (float)lo * inv0;
                *f1 = (float)hi * inv1;
            };
            if (cur.n_chunks == 1u && adam.params && g.hashed[cur.level]) {
                // the bin's gradient is complete right here: step its entries instead of storing it (skipped step: the
                // gradient is not needed either)
                if (!adam_skip) {
                    const size_t o4 = ((size_t)g.offset[cur.level] + (size_t)cur.slice * BIN) >> 1;  // in float4 units
                    float4* __restrict__ p4 = reinterpret_cast<float4*>(adam.params) + o4;
                    float4* __restrict__ m4 = reinterpret_cast<float4*>(adam.exp_avg) + o4;
                    float4* __restrict__ v4 = reinterpret_cast<float4*>(adam.exp_avg_sq) + o4;
                    uint2* __restrict__ h4 = reinterpret_cast<uint2*>(adam.params_half) + o4;
                    float4* __restrict__ e4 = adam.ema ? reinterpret_cast<float4*>(adam.ema) + o4 : nullptr;
                    uint2* __restrict__ eh4 = adam.ema_half ? reinterpret_cast<uint2*>(adam.ema_half) + o4 : nullptr;
                    // (four steps' loads -- 12 x 16 bytes per thread -- are requested before the first is used: the flush
                    // of a bin is a pure stream, and with two workgroups per CU only the thread itself hides its latency)
                    constexpr uint32_t kU = 4;
                    for (uint32_t e0 = threadIdx.x; e0 < n2; e0 += kU * kTlBlockP) {
                        float4 pv[kU], mv[kU], vv[kU];
#pragma unroll
                        for (uint32_t u = 0; u < kU; ++u) {
                            const uint32_t e = min(e0 + u * kTlBlockP, n2 - 1u);
                            pv[u] = p4[e];
                            mv[u] = m4[e];
                            vv[u] = v4[e];
                        }
#pragma unroll
                        for (uint32_t u = 0; u < kU; ++u) {
                            const uint32_t e = e0 + u * kTlBlockP;
                            if (e < n2) {
                                const ulonglong2 w = acc2[e];
                                float4 gv;
                                split(w.x, &gv.x, &gv.y);
                                split(w.y, &gv.z, &gv.w);
                                nvo_adam_one(pv[u].x, mv[u].x, vv[u].x, gv.x, ah);
                                nvo_adam_one(pv[u].y, mv[u].y, vv[u].y, gv.y, ah);
                                nvo_adam_one(pv[u].z, mv[u].z, vv[u].z, gv.z, ah);
                                nvo_adam_one(pv[u].w, mv[u].w, vv[u].w, gv.w, ah);
                                p4[e] = pv[u];
                                m4[e] = mv[u];
                                v4[e] = vv[u];
                                h4[e] = make_uint2(nvo_cvt16x2(pv[u].x, pv[u].y, false), nvo_cvt16x2(pv[u].z, pv[u].w, false));
                                if (e4) {  // the weight average of the entries just stepped (tcnn EmaOptimizer)
                                    float4 ev = e4[e];
                                    ev.x = (ev.x * ema_keep + pv[u].x * ema_take) * ema_inv;
                                    ev.y = (ev.y * ema_keep + pv[u].y * ema_take) * ema_inv;
                                    ev.z = (ev.z * ema_keep + pv[u].z * ema_take) * ema_inv;
                                    ev.w = (ev.w * ema_keep + pv[u].w * ema_take) * ema_inv;
                                    e4[e] = ev;
                                    if (eh4) eh4[e] = make_uint2(nvo_cvt16x2(ev.x, ev.y, false), nvo_cvt16x2(ev.z, ev.w, false));
                                }
                            }
                        }
                    }
                }
            } else if (cur.n_chunks == 1u) {
                for (uint32_t e = threadIdx.x; e < n2; e += kTlBlockP) {
                    const ulonglong2 w = acc2[e];
                    float4 v;
                    split(w.x, &v.x, &v.y);
                    split(w.y, &v.z, &v.w);
                    gr4[e] = v;
                }
            } else {
                for (uint32_t e = threadIdx.x; e < entries; e += kTlBlockP) {
                    float f0, f1;
                    split(acc[e], &f0, &f1);
                    if (f0 != 0.f) atomicAdd(gr + 2 * e, f0);
                    if (f1 != 0.f) atomicAdd(gr + 2 * e + 1, f1);
                }
            }
        }
        GP_CLK(gp4);
        __syncthreads();  // the next item zeroes the accumulators; the plain stores above have retired
#ifdef NVO_GRID_PHASE
        if (threadIdx.x == 0) {
            GP_CLK(gp5);
            GP_ADD(0, gp1 - gp0); GP_ADD(1, gp2 - gp1); GP_ADD(2, gp3 - gp2); GP_ADD(3, gp4 - gp3); GP_ADD(4, gp5 - gp4);
            GP_ADD(cur.n_chunks == 1u ? 5 : 6, 1);
        }
#endif
        if (__ballot(bad) != 0ull && lane == 0u) {  // poisoned chunk
            if (!(adam.params && g.hashed[cur.level])) atomicAdd(gr, __builtin_nanf(""));  // (fused: no gradient is kept)
            if (nf_flag) atomicOr(nf_flag, 1u);
        }
        if (!has_next) break;
        it = it_next;
        cur = nxt;
        segw = segw_next;
        l1w = l1w_next;
    }
}

// ------------------------------------------------------------------------------------------
// backward w.r.t. the input position (needed for analytic normals and pose gradients)
// ------------------------------------------------------------------------------------------
// One thread per (sample, level); per-level partials are combined with float atomics into
// dx[N][3] (16 adds per element, consecutive lanes -> consecutive 12-B rows).
// partial != nullptr: the per-level contribution is STORED to partial[level][i][3] (summed over the levels by
// k_sum_levels) instead of being added into dx with float atomics (16 x 3 atomics per sample).
template <bool SOA, typename DY2>
__global__ void __launch_bounds__(kGridBlock)
k_grid_bwd_input(NvoGridLevels g, uint32_t N, const float* __restrict__ x,
                 const __half2* __restrict__ table, const DY2* __restrict__ dy,
                 float* __restrict__ dx, float* __restrict__ partial) {
    uint32_t tile, level;
    grid_block_map(blockIdx.x, g.n_levels, &tile, &level);
    const uint32_t i = tile * kGridBlock + threadIdx.x;
    if (i >= N) return;
    float* __restrict__ pout = partial ? partial + 3 * ((size_t)level * N + i) : nullptr;
    const uint32_t off = g.offset[level];
    const uint32_t size = g.offset[level + 1] - off;
    const uint32_t res = g.resolution[level];
    const uint32_t hashed = g.hashed[level];
    const float scale = g.scale[level];
    const __half2* __restrict__ tab = table + off;

    const DY2 d2 = SOA ? dy[(size_t)level * N + i] : dy[(size_t)i * g.n_levels + level];
    const float2 d = dy2f(d2);
    if (d.x == 0.f && d.y == 0.f) {
        if (pout) pout[0] = pout[1] = pout[2] = 0.f;
        return;
    }

    const Corner c = grid_cell(scale, x[3 * (size_t)i + 0], x[3 * (size_t)i + 1],
                               x[3 * (size_t)i + 2]);
    float s[8];  // dL/dy . value at each corner
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t idx = nvo_grid_index(hashed, size, res, c.px + (k & 1u),
                                            c.py + ((k >> 1) & 1u), c.pz + ((k >> 2) & 1u));
        const float2 f = __half22float2(tab[idx]);
        s[k] = f.x * d.x + f.y * d.y;
    }
    const float wx0 = 1.f - c.wx, wy0 = 1.f - c.wy, wz0 = 1.f - c.wz;
    // d/dx: (corner with x-bit) - (corner without), weighted by the other two axes
    const float gx = scale * (wy0 * wz0 * (s[1] - s[0]) + c.wy * wz0 * (s[3] - s[2]) +
                              wy0 * c.wz * (s[5] - s[4]) + c.wy * c.wz * (s[7] - s[6]));
    const float gy = scale * (wx0 * wz0 * (s[2] - s[0]) + c.wx * wz0 * (s[3] - s[1]) +
                              wx0 * c.wz * (s[6] - s[4]) + c.wx * c.wz * (s[7] - s[5]));
    const float gz = scale * (wx0 * wy0 * (s[4] - s[0]) + c.wx * wy0 * (s[5] - s[1]) +
                              wx0 * c.wy * (s[6] - s[2]) + c.wx * c.wy * (s[7] - s[3]));
    if (pout) {
        pout[0] = gx;
        pout[1] = gy;
        pout[2] = gz;
        return;
    }
    atomicAdd(dx + 3 * (size_t)i + 0, gx);
    atomicAdd(dx + 3 * (size_t)i + 1, gy);
    atomicAdd(dx + 3 * (size_t)i + 2, gz);
}

// Input backward from the dy/dx the forward stored: one thread per sample streams L x (dy, 3 x dydx) half2 values,
// all coalesced.  dx[i][axis] (+)= sum_l scale_l * (dy_l . dydx_l,axis).
template <bool SOA, typename DY2>
__global__ void __launch_bounds__(256)
k_grid_bwd_input_dydx(NvoGridLevels g, uint32_t N, const __half2* __restrict__ dydx, const DY2* __restrict__ dy,
                      float* __restrict__ dx, int accumulate) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (uint32_t level = 0; level < g.n_levels; ++level) {
        const DY2 d2 = SOA ? dy[(size_t)level * N + i] : dy[(size_t)i * g.n_levels + level];
        const float2 d = dy2f(d2);
        const __half2* __restrict__ p = dydx + (size_t)level * 3 * N + i;
        const float2 gx = __half22float2(p[0]), gy = __half22float2(p[N]), gz = __half22float2(p[2 * (size_t)N]);
        const float sc = g.scale[level];
        a0 = fmaf(sc, d.x * gx.x + d.y * gx.y, a0);
        a1 = fmaf(sc, d.x * gy.x + d.y * gy.y, a1);
        a2 = fmaf(sc, d.x * gz.x + d.y * gz.y, a2);
    }
    float* __restrict__ o = dx + 3 * (size_t)i;
    if (accumulate) {
        a0 += o[0];
        a1 += o[1];
        a2 += o[2];
    }
    o[0] = a0;
    o[1] = a1;
    o[2] = a2;
}

// dx[e] (+)= sum over levels of partial[level][e], e over N * 3 floats
__global__ void __launch_bounds__(256)
k_sum_levels(uint32_t n_levels, size_t n, const float* __restrict__ partial, float* __restrict__ dx, int accumulate) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = accumulate ? dx[e] : 0.f;
    for (uint32_t l = 0; l < n_levels; ++l) s += partial[(size_t)l * n + e];
    dx[e] = s;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host launchers (C++ linkage, used by api.cpp and the fused pipeline)
// ---------------------------------------------------------------------------------------------
// dy_fmt (NVO_DY_*) -> element type of the dL/d(encoded) pairs
#define NVO_DY_DISPATCH(M, SOA_)                          \
    do {                                                  \
        if (dy_fmt == NVO_DY_FLOAT) M(SOA_, float2);      \
        else if (dy_fmt == NVO_DY_BF16) M(SOA_, Bf2);     \
        else M(SOA_, __half2);                            \
    } while (0)

// Builds the XCD-balanced plan (see GridFwdPlan); returns the blocks per XCD (grid = 8 x that), plan->enabled = 0 when
// the level set does not fit the scheme (fewer than 8 hashed levels, too many pieces).
static uint32_t grid_fwd_plan_build(const NvoGridLevels& g, uint32_t tiles, GridFwdPlan* plan) {
    static const float dense_cost = [] { const char* e = getenv("NVO_GRID_FWD_DENSE_COST"); return e ? (float)atof(e) : 0.4f; }();
    std::vector<uint32_t> hashed, dense;
    for (uint32_t l = 0; l < g.n_levels; ++l) (g.hashed[l] ? hashed : dense).push_back(l);
    if (hashed.size() < 8 || hashed.size() > 16) return 0;
    float load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto push = [&](uint32_t xcd, uint32_t level, uint32_t t0, uint32_t n, float cost) -> bool {
        if (n == 0) return true;
        uint32_t& k = plan->n_units[xcd];
        if (k >= (uint32_t)kPlanUnits) return false;
        plan->level[xcd][k] = level;
        plan->tile0[xcd][k] = t0;
        plan->n_tiles[xcd][k] = n;
        ++k;
        load[xcd] += cost * (float)n;
        return true;
    };
    // one whole hashed level per XCD (the finest first: they miss L1 most)
    for (uint32_t x = 0; x < 8; ++x)
        if (!push(x, hashed[hashed.size() - 1 - x], 0, tiles, 1.f)) return 0;
    // the remaining hashed levels in pieces: every XCD gets tiles of exactly ONE further table
    const uint32_t n_rest = (uint32_t)hashed.size() - 8;
    if (n_rest) {
        const uint32_t piece = nvo_div_up((uint64_t)n_rest * tiles, 8);
        uint32_t lv = 0, t0 = 0;
        for (uint32_t x = 0; x < 8 && lv < n_rest; ++x) {
            const uint32_t n = tiles - t0 < piece ? tiles - t0 : piece;
            if (!push(x, hashed[lv], t0, n, 1.f)) return 0;
            t0 += n;
            if (t0 >= tiles) {
                ++lv;
                t0 = 0;
            }
        }
        // (what the one-table-per-XCD rule left over goes to the XCD that already holds that table)
        while (lv < n_rest) {
            uint32_t owner = 7;
            for (uint32_t x = 0; x < 8; ++x)
                for (uint32_t u = 0; u < plan->n_units[x]; ++u)
                    if (plan->level[x][u] == hashed[lv]) owner = x;
            if (!push(owner, hashed[lv], t0, tiles - t0, 1.f)) return 0;
            ++lv;
            t0 = 0;
        }
    }
    // dense levels: tile ranges to the least loaded XCD until every XCD carries the same cost
    float total = 0.f;
    for (int x = 0; x < 8; ++x) total += load[x];
    total += dense_cost * (float)(dense.size() * tiles);
    const float target = total / 8.f;
    for (uint32_t l : dense) {
        uint32_t t0 = 0;
        while (t0 < tiles) {
            uint32_t best = 0;
            for (uint32_t x = 1; x < 8; ++x)
                if (load[x] < load[best]) best = x;
            float room = (target - load[best]) / dense_cost;
            uint32_t n = room < 8.f ? 8u : (uint32_t)room;
            if (n > tiles - t0) n = tiles - t0;
            if (plan->n_units[best] + 1 >= (uint32_t)kPlanUnits) n = tiles - t0;  // last free slot: take the rest
            if (!push(best, l, t0, n, dense_cost)) return 0;
            t0 += n;
        }
    }
    uint32_t max_blocks = 0;
    for (uint32_t x = 0; x < 8; ++x) {
        uint32_t b = 0;
        for (uint32_t u = 0; u < plan->n_units[x]; ++u) b += plan->n_tiles[x][u];
        if (b > max_blocks) max_blocks = b;
    }
    plan->enabled = 1;
    return max_blocks;
}

int nvo_grid_fwd_launch(const NvoGridLevels& g, hipStream_t stream, uint32_t N, const float* x,
                        const void* table_half, void* out_half, bool soa, uint32_t* indices, void* dydx_half,
                        bool out_bf16, const uint32_t* n_live, bool runs, int small_form) {
    if (N == 0) return NVO_OK;
    NVO_REQUIRE(g.n_features == 2, "grid: only n_features_per_level == 2 is supported (got %u)",
                g.n_features);
    NVO_PROF(stream, "grid_fwd[L%u]", g.n_levels);
    // small grids (the proposal networks): the two coarsest dense levels from LDS, a thread per sample (k_grid_fwd_small)
    // 0 off | 1 plain | 2 two samples per thread (measured slower: 35.0 vs 33.6 us) | 3 software-pipelined (round 6: no gain,
    // the kernel is bound by vector-instruction issue) | 4 instruction-lean + pipelined (round 6, default)
    static const int small_default = [] { const char* e = getenv("NVO_GRID_FWD_SMALL"); return e ? atoi(e) : 4; }();
    const int small_env = small_form >= 0 ? small_form : small_default;  // (module option grid_fwd_small_form: tests, A/B)
    if (small_env && soa && !indices && !dydx_half && !n_live && g.n_levels == 5 && !g.hashed[0] && !g.hashed[1] &&
        (size_t)g.offset[2] * 4 <= 152 * 1024 && (g.offset[2] & 3u) == 0u && (((uintptr_t)table_half) & 15u) == 0u) {
        static const uint32_t n_cus = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            return (uint32_t)(n > 0 ? n : 256);
        }();
        const size_t lds = (size_t)g.offset[2] * 4;
        static bool attr_set = false;
        if (!attr_set) {
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_grid_fwd_small<2, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              152 * 1024));
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_grid_fwd_small<2, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              152 * 1024));
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_grid_fwd_small_pipe<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              152 * 1024));
            attr_set = true;
        }
        if (runs && (N & 3u) == 0u && (((uintptr_t)x) & 15u) == 0u && (((uintptr_t)out_half) & 15u) == 0u) {
            // option "grid_fwd_runs" (inference: 8 M samples per chunk -- 175 -> 158 us; on a training batch of 0.4-1 M
            // samples a 1024-thread workgroup of four-sample runs has one pass or less to do and loses 8 us)
            static bool attr_runs = false;
            if (!attr_runs) {
                NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_grid_fwd_small_runs<2, 3>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
                attr_runs = true;
            }
            const uint32_t per_block = (uint32_t)nvo_round_up(nvo_div_up(N, n_cus), kSmallBlock * 4);
            const uint32_t hm = (g.hashed[2] ? 1u : 0u) | (g.hashed[3] ? 2u : 0u) | (g.hashed[4] ? 4u : 0u);
            if (small_env >= 4 && (hm == 6u || hm == 7u) && (uint64_t)N * 20u < (1ull << 32) && g.resolution[0] <= 4096u &&
                g.resolution[1] <= 4096u && g.resolution[2] <= 4096u) {
#define NVO_LAUNCH_RL(HM_, BF_)                                                                                           \
    do {                                                                                                                   \
        static bool attr_rl = false;                                                                                       \
        if (!attr_rl) {                                                                                                    \
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_grid_fwd_small_runs_lean<2, 3, HM_, BF_>,                     \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));                    \
            attr_rl = true;                                                                                                \
        }                                                                                                                  \
        NVO_LAUNCH((k_grid_fwd_small_runs_lean<2, 3, HM_, BF_>), dim3(nvo_div_up(N, per_block)), dim3(kSmallBlock), lds,   \
                   stream, g, N, x, (const __half2*)table_half, (__half2*)out_half, per_block);                            \
    } while (0)
                if (hm == 6u) { if (out_bf16) NVO_LAUNCH_RL(6u, true); else NVO_LAUNCH_RL(6u, false); }
                else { if (out_bf16) NVO_LAUNCH_RL(7u, true); else NVO_LAUNCH_RL(7u, false); }
#undef NVO_LAUNCH_RL
                NVO_CHECK_LAUNCH();
                return NVO_OK;
            }
            NVO_LAUNCH((k_grid_fwd_small_runs<2, 3>), dim3(nvo_div_up(N, per_block)), dim3(kSmallBlock), lds, stream, g, N, x,
                       (const __half2*)table_half, (__half2*)out_half, out_bf16 ? 1 : 0, per_block);
            NVO_CHECK_LAUNCH();
            return NVO_OK;
        }
        // one workgroup per CU, a whole number of passes each
        const uint32_t spt = small_env == 2 ? 2u : 1u;
        const uint32_t hmask = (g.hashed[2] ? 1u : 0u) | (g.hashed[3] ? 2u : 0u) | (g.hashed[4] ? 4u : 0u);
        if (small_env >= 4 && (hmask == 6u || hmask == 7u || hmask == 4u) && (uint64_t)N * 20u < (1ull << 32) &&
            g.resolution[0] <= 4096u && g.resolution[1] <= 4096u && g.resolution[2] <= 4096u) {
            // the instruction-lean form: wave-granular shares (a workgroup's last pass may be partly empty: its idle waves leave)
            const uint32_t per_lean = (uint32_t)nvo_round_up(nvo_div_up(N, n_cus), 64u);
            const dim3 gl(nvo_div_up(N, per_lean));
#define NVO_LAUNCH_LEAN(HM_, BF_)                                                                                          \
    do {                                                                                                                   \
        static bool attr_lean = false;                                                                                     \
        if (!attr_lean) {                                                                                                  \
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_grid_fwd_small_lean<2, 3, HM_, BF_>,                          \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));                    \
            attr_lean = true;                                                                                              \
        }                                                                                                                  \
        NVO_LAUNCH((k_grid_fwd_small_lean<2, 3, HM_, BF_>), gl, dim3(kSmallBlock), lds, stream, g, N, x,                   \
                   (const __half2*)table_half, (__half2*)out_half, per_lean);                                              \
    } while (0)
            if (hmask == 6u) { if (out_bf16) NVO_LAUNCH_LEAN(6u, true); else NVO_LAUNCH_LEAN(6u, false); }
            else if (hmask == 7u) { if (out_bf16) NVO_LAUNCH_LEAN(7u, true); else NVO_LAUNCH_LEAN(7u, false); }
            else { if (out_bf16) NVO_LAUNCH_LEAN(4u, true); else NVO_LAUNCH_LEAN(4u, false); }
#undef NVO_LAUNCH_LEAN
            NVO_CHECK_LAUNCH();
            return NVO_OK;
        }
        const uint32_t per_block = (uint32_t)nvo_round_up(nvo_div_up(N, n_cus), kSmallBlock * spt);
        if (small_env >= 3)
            NVO_LAUNCH((k_grid_fwd_small_pipe<2, 3>), dim3(nvo_div_up(N, per_block)), dim3(kSmallBlock), lds, stream, g, N, x,
                       (const __half2*)table_half, (__half2*)out_half, out_bf16 ? 1 : 0, per_block);
        else if (spt == 2)
            NVO_LAUNCH((k_grid_fwd_small<2, 3, 2>), dim3(nvo_div_up(N, per_block)), dim3(kSmallBlock), lds, stream, g, N, x,
                       (const __half2*)table_half, (__half2*)out_half, out_bf16 ? 1 : 0, per_block);
        else
            NVO_LAUNCH((k_grid_fwd_small<2, 3, 1>), dim3(nvo_div_up(N, per_block)), dim3(kSmallBlock), lds, stream, g, N, x,
                       (const __half2*)table_half, (__half2*)out_half, out_bf16 ? 1 : 0, per_block);
        NVO_CHECK_LAUNCH();
        return NVO_OK;
    }
    static const int spt_env = [] { const char* e = getenv("NVO_GRID_FWD_SPT"); return e ? atoi(e) : 2; }();
    // option "grid_fwd_runs": four consecutive samples per thread (k_grid_fwd_runs) wherever its 16-byte accesses line up
    runs = runs && soa && !indices && !dydx_half && (N & 3u) == 0u && (((uintptr_t)x) & 15u) == 0u &&
           (((uintptr_t)out_half) & 15u) == 0u;
    const int spt = runs ? 4 : (dydx_half || spt_env < 2) ? 1 : (spt_env >= 4 ? 4 : 2);  // samples per thread
    const uint32_t tiles = nvo_div_up(N, kGridBlock * spt);
    dim3 grid(tiles * g.n_levels), block(kGridBlock);
    GridFwdPlan plan;
    memset(&plan, 0, sizeof(plan));
    static const int balance_env = [] { const char* e = getenv("NVO_GRID_FWD_BALANCE"); return e ? atoi(e) : 1; }();
    if (balance_env && (g.n_levels & 7u) == 0u) {
        const uint32_t blocks_per_xcd = grid_fwd_plan_build(g, tiles, &plan);
        if (plan.enabled) grid = dim3(8u * blocks_per_xcd);
    }
#define NVO_LAUNCH_FWD_S(SOA_, DYDX_, SPT_)                                                                  \
    NVO_LAUNCH((k_grid_fwd<SOA_, DYDX_, SPT_>), grid, block, 0, stream, g, N, x, (const __half2*)table_half, \
               (__half2*)out_half, indices, (__half2*)dydx_half, out_bf16 ? 1 : 0, plan, n_live)
#define NVO_LAUNCH_FWD(SOA_, DYDX_)                                        \
    do {                                                                   \
        if (DYDX_ || spt == 1) NVO_LAUNCH_FWD_S(SOA_, DYDX_, 1);           \
        else if (spt == 2) NVO_LAUNCH_FWD_S(SOA_, false, 2);               \
        else NVO_LAUNCH_FWD_S(SOA_, false, 4);                             \
    } while (0)
    // the instruction-lean form of the level-major production path (form 0 = the first kernel, kept as the reference form)
    // NVO_GRID_FWD_LEAN=1 (A/B; default off): k_grid_fwd_lean for the main grid.  Measured (EXPERIMENTS 10.3): faster on
    // samples clustered within 2 % of the surface (35.5 -> 30.9 us per training batch, 243 -> 199 us per render chunk) but
    // SLOWER where the gather binds -- the spread samples of an untrained field: 59.5 -> 66 us -- and neutral inside the
    // mapping loop (windows at iterations 5000 / 7900: 0.398 / 0.385 vs 0.395 / 0.385 ms); the first kernel stays.
    static const bool lean_env = [] { const char* e = getenv("NVO_GRID_FWD_LEAN"); return e && atoi(e) != 0; }();
    bool lean = lean_env && small_env != 0 && !runs && soa && !indices && !dydx_half && (uint64_t)N * 12u < (1ull << 32);
    for (uint32_t l = 0; l < g.n_levels; ++l) lean = lean && g.resolution[l] <= 4096u;
    // the run-walking inference form has a lean counterpart as well (form 0 = the first kernel)
    static const bool lean_runs_env = [] { const char* e = getenv("NVO_GRID_FWD_RUNS_LEAN"); return !e || atoi(e) != 0; }();  // A/B
    bool lean_runs = lean_runs_env && small_env != 0 && runs && (uint64_t)N * 12u < (1ull << 32);
    for (uint32_t l = 0; l < g.n_levels; ++l) lean_runs = lean_runs && g.resolution[l] <= 4096u;
    if (lean) {
        static const bool pair_env = [] { const char* e = getenv("NVO_GRID_FWD_PAIR"); return !e || atoi(e) != 0; }();  // A/B (default on)
#define NVO_LAUNCH_LEANM(SPT_, BF_, PAIR_)                                                                   \
    NVO_LAUNCH((k_grid_fwd_lean<SPT_, BF_, PAIR_>), grid, block, 0, stream, g, N, x, (const __half2*)table_half, \
               (__half2*)out_half, plan, n_live)
#define NVO_LAUNCH_LEANS(SPT_)                                                                               \
    do {                                                                                                     \
        if (pair_env) { if (out_bf16) NVO_LAUNCH_LEANM(SPT_, true, true); else NVO_LAUNCH_LEANM(SPT_, false, true); } \
        else { if (out_bf16) NVO_LAUNCH_LEANM(SPT_, true, false); else NVO_LAUNCH_LEANM(SPT_, false, false); } \
    } while (0)
        if (spt == 1) NVO_LAUNCH_LEANS(1);
        else if (spt == 2) NVO_LAUNCH_LEANS(2);
        else NVO_LAUNCH_LEANS(4);
#undef NVO_LAUNCH_LEANS
#undef NVO_LAUNCH_LEANM
    } else if (runs && lean_runs) {
        if (out_bf16) NVO_LAUNCH((k_grid_fwd_runs_lean<kGridBlock, true>), grid, block, 0, stream, g, N, x, (const __half2*)table_half, (__half2*)out_half, plan, n_live);
        else NVO_LAUNCH((k_grid_fwd_runs_lean<kGridBlock, false>), grid, block, 0, stream, g, N, x, (const __half2*)table_half, (__half2*)out_half, plan, n_live);
    } else if (runs) {
        NVO_LAUNCH((k_grid_fwd_runs<kGridBlock>), grid, block, 0, stream, g, N, x, (const __half2*)table_half,
                   (__half2*)out_half, out_bf16 ? 1 : 0, plan, n_live);
    } else if (soa) {
        if (dydx_half) NVO_LAUNCH_FWD(true, true); else NVO_LAUNCH_FWD(true, false);
    } else {
        if (dydx_half) NVO_LAUNCH_FWD(false, true); else NVO_LAUNCH_FWD(false, false);
    }
#undef NVO_LAUNCH_FWD
#undef NVO_LAUNCH_FWD_S
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

// Slice tables for the LDS backward live in a small device buffer owned by the module.

int nvo_grid_slices_create(const NvoGridLevels& g, NvoGridSlices* s, uint32_t level_mask, uint32_t target,
                           bool env_items) {
    static_assert(2 * AccFixed32::kEntries * sizeof(int) <= kLdsBwdBytes, "32-bit slice does not fit the LDS");
    struct Item { uint32_t level, first, chunk, n_chunks; };
    // Per level: accumulator kind and slice size.  Large hashed tables (>= 2^18 entries: a 20K-entry
    // slice sees <= 8 % of the lookups) use fp32 / 20K-entry slices, everything else 64-bit fixed
    // point / 8K-entry slices.  NVO_GRID_BWD_ACC = "fixed" | "float" forces one kind (experiments).
    const char* force = getenv("NVO_GRID_BWD_ACC");
    const bool acc32 = s->acc_bits == 32;  // 32-bit fixed point with the L1-derived scale, every level
    auto float_mode = [&](uint32_t l) {
        if (acc32 || s->deterministic) return false;  // (LDS float atomics retire in no fixed order)
        if (force && !strcmp(force, "fixed")) return false;
        if (force && !strcmp(force, "float")) return true;
        return g.hashed[l] && (g.offset[l + 1] - g.offset[l]) >= (1u << 18);
    };
    // entries per slice of a level; fixed_cap (< 8192, dense levels only) shrinks the 64-bit fixed-point slices so that
    // the launch leaves LDS to a kernel running beside it (mode 3: the record scatter, see NvoGridStream::overlap)
    auto slice_entries = [&](uint32_t l) {
        if (acc32) return AccFixed32::kEntries;
        if (float_mode(l)) return kSliceFloat;
        return (s->fixed_cap && !g.hashed[l]) ? (s->fixed_cap < kSliceFixed ? s->fixed_cap : kSliceFixed) : kSliceFixed;
    };
    // pass 1: chunk counts from the hit share alone (unit = share of one float slice of a 2^19
    // table); pass 2: scale them so that the launch has enough (>= target) items to fill 256 CUs
    // for several rounds.
    if (const char* env = env_items ? getenv("NVO_GRID_BWD_ITEMS") : nullptr) target = (uint32_t)atoi(env);
    s->level_mask = level_mask;
    auto base_chunks = [&](uint32_t count, uint32_t size) {
        const double share = (double)count / (double)size * (524288.0 / (double)kSliceFloat);
        uint32_t n = (uint32_t)(share + 0.5);
        return n < 1 ? 1u : n;
    };
    if (const char* e = getenv("NVO_GRID_RUNS")) s->runs = atoi(e) != 0;  // A/B switch for measurements
    uint32_t base_total = 0;
    for (uint32_t l = 0; l < g.n_levels; ++l) {
        if (!((level_mask >> l) & 1u)) continue;
        const uint32_t size = g.offset[l + 1] - g.offset[l];
        const uint32_t se = slice_entries(l);
        for (uint32_t f = 0; f < size; f += se) base_total += base_chunks(size - f < se ? size - f : se, size);
    }
    const uint32_t factor = (base_total == 0 || base_total >= target) ? 1u : (target + base_total - 1) / base_total;
    // one-round rule (see NvoGridSlices::batch_hint)
    uint32_t even_chunks = 0, dense_chunks = 0, hashed_chunks = 0;
    if (s->runs && s->batch_hint) {
        uint32_t n_total = 0, n_dense = 0;
        for (uint32_t l = 0; l < g.n_levels; ++l) {
            if (!((level_mask >> l) & 1u)) continue;
            const uint32_t size = g.offset[l + 1] - g.offset[l];
            const uint32_t se = slice_entries(l);
            n_total += (size + se - 1) / se;
            if (!g.hashed[l]) n_dense += (size + se - 1) / se;
        }
        int dev = 0, n_cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (const char* e = getenv("NVO_GRID_ROUND_CUS")) n_cus = atoi(e);  // measurements
        if (n_total && n_total <= (uint32_t)n_cus) {
            const uint64_t pass = (uint64_t)kLdsBwdBlock * 8;
            const uint32_t per = (uint32_t)n_cus / n_total;
            const uint32_t k = (uint32_t)((s->batch_hint + pass * per - 1) / (pass * per));
            even_chunks = (uint32_t)((s->batch_hint + pass * k - 1) / (pass * k));
            dense_chunks = hashed_chunks = even_chunks;
            // option grid_bwd_dense_share (percent): how the one round is shared between dense and hashed slices.  With
            // most samples live (bf16 gradients, the reference's loss scale) a dense-level item takes 1.6 x as long as a
            // hashed-level one (phase clocks): 120 gives the dense slices 12 chunks and the hashed
            // ones 9 where both had 10; with few samples live it is the other way round and 100 (equal) stays.
            const uint32_t n_hashed = n_total - n_dense;
            if (s->dense_share_pct != 100u && n_dense && n_hashed) {
                uint32_t cd = (even_chunks * s->dense_share_pct + 50u) / 100u;
                if (cd < 1u) cd = 1u;
                while (cd > 1u && n_dense * cd + n_hashed > (uint32_t)n_cus) --cd;
                const uint32_t ch = ((uint32_t)n_cus - n_dense * cd) / n_hashed;
                if (ch >= 1u) {
                    dense_chunks = cd;
                    hashed_chunks = ch;
                }
            }
        }
    }
    if (const char* e = getenv("NVO_GRID_CHUNKS")) {  // measurements: "dense,hashed" chunks per slice
        unsigned d = 0, h = 0;
        if (sscanf(e, "%u,%u", &d, &h) == 2 && d && h) {
            even_chunks = d;
            dense_chunks = d;
            hashed_chunks = h;
        }
    }
    // Most expensive first: single-chunk items scan all N samples (long), chunked items scan
    // N / n_chunks samples with a high hit rate (short but atomic-heavy).
    std::vector<Item> single, chunked;
    for (int l = (int)g.n_levels - 1; l >= 0; --l) {
        if (!((level_mask >> l) & 1u)) continue;
        const uint32_t size = g.offset[l + 1] - g.offset[l];
        const bool fm = float_mode((uint32_t)l);
        const uint32_t se = slice_entries((uint32_t)l);
        for (uint32_t f = 0; f < size; f += se) {
            const uint32_t count = size - f < se ? size - f : se;
            uint32_t n_chunks = even_chunks ? (g.hashed[l] ? hashed_chunks : dense_chunks) : base_chunks(count, size) * factor;
            if (n_chunks > 1024) n_chunks = 1024;
            if (s->deterministic) n_chunks = 1;  // chunks of a slice meet in float atomics: one owner instead
            for (uint32_t c = 0; c < n_chunks; ++c)
                (n_chunks == 1 ? single : chunked).push_back(
                    Item{(uint32_t)l | (se << 8), f, c, n_chunks | (fm ? 0x80000000u : 0u) | (acc32 ? 0x40000000u : 0u) |
                                                (s->runs && !g.hashed[l] ? 0x20000000u : 0u)});
        }
    }
    std::vector<Item> all(single);
    all.insert(all.end(), chunked.begin(), chunked.end());
    s->n_slices = (uint32_t)all.size();
    s->lds_bytes = 0;
    for (const Item& it : all) {  // (every accumulator kind holds 2 values per entry)
        const uint32_t bytes = (it.level >> 8) * 2u * ((it.n_chunks >> 31) ? 4u : ((it.n_chunks >> 30) & 1u) ? 4u : 8u);
        if (bytes > s->lds_bytes) s->lds_bytes = bytes;
    }
    // contiguous run of entries owned by chunked items (needs zeroing before the atomic flush)
    s->zero_first = 0xFFFFFFFFu;
    s->zero_last = 0;
    for (const Item& it : chunked) {
        const uint32_t lo = g.offset[it.level & 0xFFu], hi = g.offset[(it.level & 0xFFu) + 1];
        if (lo < s->zero_first) s->zero_first = lo;
        if (hi > s->zero_last) s->zero_last = hi;
    }
    if (all.empty()) {
        s->d_level = s->d_first = nullptr;
        return NVO_OK;
    }
    // [items | length of the live-sample list (one u64 slot) | L1 norms (u64 [levels][2], used by the 32-bit accumulators)]
    NVO_CHECK_HIP(hipMalloc((void**)&s->d_level, sizeof(Item) * all.size() + sizeof(unsigned long long) * (1 + 2 * NVO_MAX_LEVELS)));
    s->d_live_n = reinterpret_cast<uint32_t*>(s->d_level + 4 * all.size());
    NVO_CHECK_HIP(hipMemset(s->d_live_n, 0, sizeof(unsigned long long)));
    s->d_l1 = reinterpret_cast<unsigned long long*>(s->d_live_n) + 1;
    s->d_first = nullptr;
    NVO_CHECK_HIP(hipMemcpy(s->d_level, all.data(), sizeof(Item) * all.size(), hipMemcpyHostToDevice));
    return NVO_OK;
}

void nvo_grid_slices_destroy(NvoGridSlices* s) {
    nvo_scratch_destroy(&s->live);
    nvo_scratch_destroy(&s->codes);
    if (s->d_level) (void)hipFree(s->d_level);
    s->d_level = s->d_first = nullptr;
    s->d_l1 = nullptr;
    s->d_live_n = nullptr;
    s->n_slices = 0;
}

void nvo_grid_slices_zero_ranges(const NvoGridLevels& g, const NvoGridSlices* s, float* grad, NvoZeroRanges* out) {
    if (s->zero_last > s->zero_first)
        out->push_back({grad + 2 * (size_t)s->zero_first, sizeof(float) * 2 * (size_t)(s->zero_last - s->zero_first)});
    // the live-list length and, behind it, the L1 norms of the 32-bit accumulators: one range
    if (s->d_live_n)
        out->push_back({s->d_live_n, sizeof(unsigned long long) * (1 + (s->acc_bits == 32 ? 2 * g.n_levels : 0))});
}

bool nvo_grid_stream_zero_ranges(const NvoGridLevels& g, const NvoGridStream* st, float* grad, NvoZeroRanges* out) {
    nvo_grid_slices_zero_ranges(g, &st->owner, grad, out);
    // streamed DENSE levels: their bins are split into tile ranges that combine with float atomics (k_st_zero)
    if (st->dense_chunks > 1)
        for (uint32_t l = 0; l < g.n_levels; ++l)
            if (((st->streamed_mask >> l) & 1u) && !g.hashed[l])
                out->push_back({grad + 2 * (size_t)g.offset[l], sizeof(float) * 2 * (size_t)(g.offset[l + 1] - g.offset[l])});
    return true;
}

// ---- mode 3 host side ---------------------------------------------------------------------------
int nvo_grid_stream_create(const NvoGridLevels& g, NvoGridStream* st) {
    // Levels with many kBinSlice-entry bins are streamed; levels with <= kStOwnerSlices slices (the coarse dense
    // ones: almost every lookup hits every slice, the redundancy of the slice-owner form is small and the
    // per-bin record lists would be long and conflict-heavy) keep slice-owner work items.
    std::vector<uint32_t> levels, first, bin_level, bin_slice;
    st->max_slices = 0;
    st->streamed_mask = 0;
    if (const char* e = getenv("NVO_GRID_STREAM_OVERLAP")) st->overlap = atoi(e) != 0;  // A/B switch for measurements
    if (const char* e = getenv("NVO_GRID_OWNER_SLICES")) st->owner_max_slices = (uint32_t)atoi(e);  // measurements
    // entries per bin: 8192 x 8 B (two 32-bit fixed-point sums in one 64-bit word) = 64 KiB
    // (6176-entry bins -- 85 per 2^19 table, 935 hashed items = two even rounds of 512 workgroups on paper -- measured
    // SLOWER: 139.0 / 146.1 us for the stage against 132.8 / 127.3 with 8192-entry bins (dealt / balanced work list):
    // the shorter runs cost more than the evener rounds save)
    uint32_t bin_p = kBinP;
    if (const char* e = getenv("NVO_TL_BIN")) bin_p = (uint32_t)atoi(e) == kBinPSmall ? kBinPSmall : kBinP;  // measurements
    const uint32_t bin_entries = bin_p;
    st->bin_entries = bin_entries;
    for (uint32_t l = 0; l < g.n_levels; ++l) {
        const uint32_t size = g.offset[l + 1] - g.offset[l];
        // (which levels stay slice-owner is decided in 4096-entry units, whatever the bin size)
        if ((size + kBinSlice - 1u) / kBinSlice <= st->owner_max_slices) continue;
        const uint32_t n_slices = (size + bin_entries - 1u) / bin_entries;
        levels.push_back(l);
        first.push_back((uint32_t)bin_level.size());
        if (n_slices > st->max_slices) st->max_slices = n_slices;
        for (uint32_t sl = 0; sl < n_slices; ++sl) {
            bin_level.push_back(l);
            bin_slice.push_back(sl);
        }
        st->streamed_mask |= 1u << l;
    }
    first.push_back((uint32_t)bin_level.size());
    st->n_levels = (uint32_t)levels.size();
    st->n_bins = (uint32_t)bin_level.size();
    const size_t nb = st->n_bins;
    if (nb) {
        const size_t words = levels.size() + first.size() + 2 * nb + nb /*chunks*/ + 4;
        NVO_CHECK_HIP(hipMalloc((void**)&st->d_meta, sizeof(uint32_t) * words));
        st->d_levels = st->d_meta;
        st->d_bin_first = st->d_levels + levels.size();
        st->d_bin_level = st->d_bin_first + first.size();
        st->d_bin_slice = st->d_bin_level + nb;
        st->d_bin_chunks = st->d_bin_slice + nb;
        NVO_CHECK_HIP(hipMemcpy(st->d_levels, levels.data(), 4 * levels.size(), hipMemcpyHostToDevice));
        NVO_CHECK_HIP(hipMemcpy(st->d_bin_first, first.data(), 4 * first.size(), hipMemcpyHostToDevice));
        NVO_CHECK_HIP(hipMemcpy(st->d_bin_level, bin_level.data(), 4 * nb, hipMemcpyHostToDevice));
        NVO_CHECK_HIP(hipMemcpy(st->d_bin_slice, bin_slice.data(), 4 * nb, hipMemcpyHostToDevice));
    }
    if (nb) {
        // static work list: hashed levels spread their records evenly over the bins (one item per bin); a streamed
        // DENSE level sees clustered samples, so its bins are split into tile ranges
        std::vector<uint32_t> items, chunks(nb, 1u);
        const bool spread = !(getenv("NVO_TL_ORDER") && atoi(getenv("NVO_TL_ORDER")) == 0);
        if (const char* e = getenv("NVO_TL_DENSE_CHUNKS")) st->dense_chunks = (uint32_t)atoi(e);
        if (st->deterministic) st->dense_chunks = 1;  // the tile ranges of a bin meet in float atomics: one item per bin
        for (uint32_t j = 0; j < levels.size(); ++j) {
            const uint32_t nc = g.hashed[levels[j]] ? 1u : st->dense_chunks;
            const uint32_t nbj = first[j + 1] - first[j];
            for (uint32_t q = 0; q < nbj; ++q) {
                // The persistent workgroups of ONE XCD take the items congruent to it modulo 8.  A bin's runs start
                // near bin * 512 B inside every 32 KiB tile region, so bins that are 8 apart would put all 32
                // workgroups of an XCD on the same few L2 channels: deal the bins so that an XCD gets CONSECUTIVE
                // bins (8 x 8 transpose of the order inside each group of 64).
                uint32_t bq = q;
                if (spread && nbj >= 64 && q < (nbj & ~63u)) bq = (q & ~63u) | ((q & 7u) << 3) | ((q >> 3) & 7u);
                const uint32_t b = first[j] + bq;
                chunks[b] = nc;
                for (uint32_t c = 0; c < nc; ++c) {  // self-contained header (k_tl_accumulate: tl_decode)
                    items.push_back(b);
                    items.push_back(c | (nc << 16));
                    items.push_back(j | (levels[j] << 8));
                    items.push_back(bin_slice[b]);
                }
            }
        }
        st->n_tl_slots = 0;
        if (!(getenv("NVO_TL_BALANCE") && atoi(getenv("NVO_TL_BALANCE")) == 0)) {
            // BALANCED work list for the persistent accumulate (2 workgroups per CU, workgroup w walks items w, w + slots,
            // ...).  Dealt round-robin, 704 hashed-level items + 208 cheap dense-level chunks gave some workgroups two
            // hashed items and a chunk, others one hashed item: the launch lasted as long as the former (132.8 -> 127.3
            // us for the stage with the list below: the chunks now go to the workgroups with one hashed item).  Longest-
            // processing-time-first over the slots (cost of an item ~ its expected records: 1 / bins of its level /
            // chunks), then laid out round by round with PADDING items (n_chunks = 0) where a slot has nothing left.
            int dev = 0, n_cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev);
            const uint32_t slots = 2u * (uint32_t)(n_cus > 0 ? n_cus : 256);
            const uint32_t n_it = (uint32_t)(items.size() / 4);
            std::vector<std::pair<float, uint32_t>> order;
            for (uint32_t i = 0; i < n_it; ++i) {
                const uint32_t j = items[4 * i + 2] & 0xFFu, nc = items[4 * i + 1] >> 16;
                order.push_back({1.f / (float)(first[j + 1] - first[j]) / (float)nc, i});
            }
            std::stable_sort(order.begin(), order.end(), [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first > b.first; });
            std::vector<float> load(slots, 0.f);
            std::vector<std::vector<uint32_t>> mine(slots);
            uint32_t cursor = 0;
            for (const auto& oc : order) {
                // least-loaded slot (linear scan: a few hundred slots, ~1200 items, once per module); ties go round-robin
                uint32_t best = cursor % slots;
                for (uint32_t q = 0; q < slots; ++q) {
                    const uint32_t sidx = (cursor + q) % slots;
                    if (load[sidx] < load[best]) best = sidx;
                }
                mine[best].push_back(oc.second);
                load[best] += oc.first;
                ++cursor;
            }
            size_t rounds = 0;
            for (const auto& m : mine) rounds = m.size() > rounds ? m.size() : rounds;
            NVO_REQUIRE(rounds <= 32, "grid_bwd_stream: %zu items per accumulate workgroup (k_tl_accumulate_p keeps 32 pending bins)", rounds);
            std::vector<uint32_t> laid(4 * rounds * slots, 0u);  // all-zero header = padding (n_chunks == 0)
            for (uint32_t w = 0; w < slots; ++w)
                for (size_t r = 0; r < mine[w].size(); ++r)
                    for (int k = 0; k < 4; ++k) laid[4 * (r * slots + w) + k] = items[4 * mine[w][r] + k];
            items.swap(laid);
            st->n_tl_slots = slots;
        }
        st->n_tl_items = (uint32_t)(items.size() / 4);
        NVO_CHECK_HIP(hipMalloc((void**)&st->d_tl_items, 4 * items.size()));
        NVO_CHECK_HIP(hipMemcpy(st->d_tl_items, items.data(), 4 * items.size(), hipMemcpyHostToDevice));
        NVO_CHECK_HIP(hipMemcpy(st->d_bin_chunks, chunks.data(), 4 * nb, hipMemcpyHostToDevice));
    }
    st->created = true;
    if (!st->aux) {
        NVO_CHECK_HIP(hipStreamCreateWithFlags(&st->aux, hipStreamNonBlocking));
        NVO_CHECK_HIP(hipEventCreateWithFlags(&st->ev_fork, hipEventDisableTiming));
        NVO_CHECK_HIP(hipEventCreateWithFlags(&st->ev_join, hipEventDisableTiming));
    }
    const uint32_t all = g.n_levels >= 32 ? 0xFFFFFFFFu : ((1u << g.n_levels) - 1u);
    // (a two-stage store + reduce form of the flush was measured slower)
    // measured optima for the coarse-only launch: 512 items (the atomic flush of a chunk costs as much as scanning ~2K
    // samples), 256 with the run-merging scan (cheaper scan, same flush)
    // 8000-entry slices: 125 KiB of LDS, which leaves room for a 512-sample scatter workgroup (32.5 KiB) on the same CU
    st->owner.fixed_cap = st->overlap ? 8000u : 0u;
    if (const char* e = getenv("NVO_GRID_OWNER_CAP")) st->owner.fixed_cap = (uint32_t)atoi(e);  // measurements
    if (const char* e = getenv("NVO_STREAM_OWNER_ACC_BITS")) st->owner.acc_bits = (uint32_t)atoi(e);  // measurements
    uint32_t owner_items = st->owner.runs ? 256 : 512;
    if (const char* e = getenv("NVO_GRID_OWNER_ITEMS")) owner_items = (uint32_t)atoi(e);  // measurements
    return nvo_grid_slices_create(g, &st->owner, all & ~st->streamed_mask, owner_items, false);
}

void nvo_grid_stream_destroy(NvoGridStream* st) {
    if (st->d_meta) (void)hipFree(st->d_meta);
    nvo_scratch_destroy(&st->work);
    if (st->d_tl_items) (void)hipFree(st->d_tl_items);
    st->d_tl_items = nullptr;
    st->n_tl_items = 0;
    st->d_meta = nullptr;
    st->n_bins = 0;
    st->created = false;
    if (st->aux) {
        (void)hipStreamSynchronize(st->aux);
        (void)hipStreamDestroy(st->aux);
        (void)hipEventDestroy(st->ev_fork);
        (void)hipEventDestroy(st->ev_join);
        st->aux = nullptr;
        st->ev_fork = st->ev_join = nullptr;
    }
    nvo_grid_slices_destroy(&st->owner);
}

void nvo_grid_stream_adam_range(const NvoGridLevels& g, const NvoGridStream* st, uint64_t* first, uint64_t* n) {
    *first = 0;
    *n = 0;
    if (!st->created) return;
    // the streamed hashed levels: one accumulate item per bin (nvo_grid_stream_create); they form the tail of the table.
    // (A streamed DENSE level sees clustered samples and its bins are split into tile ranges that meet in float atomics.
    // Round 6 tried both ways of stepping it inside the pass as well -- single items per bin, and the last chunk to check
    // in stepping the bin from the summed gradient -- so that the optimiser launch of the other parameters could run
    // beside the pass: EXPERIMENTS.md 10.9, both measured slower overall.)
    uint32_t lo = g.n_levels;
    for (uint32_t l = 0; l < g.n_levels; ++l)
        if (((st->streamed_mask >> l) & 1u) && g.hashed[l] && l < lo) lo = l;
    if (lo == g.n_levels) return;
    for (uint32_t l = lo; l < g.n_levels; ++l)
        if (!(((st->streamed_mask >> l) & 1u) && g.hashed[l])) return;  // (not a contiguous tail: leave it to the optimiser)
    *first = g.offset[lo];
    *n = (uint64_t)g.offset[g.n_levels] - g.offset[lo];
}

int nvo_grid_bwd_stream_launch(const NvoGridLevels& g, NvoGridStream* st, hipStream_t stream, uint32_t N,
                               const float* x, const void* dy, int dy_fmt, bool soa, float* grad) {
    NVO_REQUIRE(g.n_features == 2, "grid: only n_features_per_level == 2 is supported");
    NVO_REQUIRE((uint64_t)N * 8 * g.n_levels < 0xFFFFFFFFull, "grid_bwd_stream: too many records for 32-bit offsets");
    if (N == 0) return nvo_zero_async(grad, sizeof(float) * 2 * (size_t)g.offset[g.n_levels], stream);
    NVO_PROF(stream, "grid_bwd_stream[L%u]", g.n_levels);
    static const uint32_t n_cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return (uint32_t)(n > 0 ? n : 256);
    }();
    const uint32_t tile = st->tile;
    const uint32_t n_tiles = nvo_div_up(N, tile);
    // fork: with the per-kernel profiler on, everything stays on the caller's stream (its events live there)
    // ... and with the optimiser step armed (st->adam.params): k_tl_accumulate_p reads *adam.skip_flag once, so every
    // producer of that word -- the slice-owner items included -- must precede it in stream order; an owner item raising
    // the flag beside it would leave the group half stepped (GradScaler steps all of a group or nothing)
    const bool fork = st->overlap && st->aux && st->owner.n_slices && st->n_bins && !st->adam.params &&
                      !(nvo_prof_enabled() && nvo_prof_detail());
    // the samples that carry a gradient, listed from the network's dL/doutput rows (NvoGridSlices::ext_tile_live)
    const uint32_t* live_list = nullptr;
    const uint32_t* live_n = nullptr;
    if (st->owner.ext_list_given && st->owner.live.ptr && st->owner.d_live_n) {  // (listed by the network's backward)
        live_list = static_cast<const uint32_t*>(st->owner.live.ptr);
        live_n = st->owner.d_live_n;
    } else if (st->owner.ext_tile_live && st->owner.ext_rows && st->owner.d_live_n && !st->owner.deterministic && (N & 15u) == 0u) {
        NvoProfMute mute;
        if (int rc = nvo_scratch_reserve(&st->owner.live, sizeof(uint32_t) * ((size_t)N + 1), stream, "grid_bwd live list")) return rc;
        uint32_t* const d_live = static_cast<uint32_t*>(st->owner.live.ptr);
        if (!st->owner.external_zero)  // (otherwise cleared by the step's zero launch: nvo_grid_slices_zero_ranges)
            if (int rc = nvo_zero_async(st->owner.d_live_n, sizeof(uint32_t), stream)) return rc;
        NVO_LAUNCH(k_live_rows, dim3(nvo_div_up(N, 4096)), dim3(1024), 0, stream, N, st->owner.ext_tile_live, st->owner.ext_tile_bits,
                   static_cast<const uint4*>(st->owner.ext_rows), st->owner.ext_tile_count, st->owner.d_live_n, d_live);
        live_list = d_live;
        live_n = st->owner.d_live_n;
    }
    struct ListGuard {  // (the owner launch below reads ext_list; nobody else may)
        const NvoGridSlices* s;
        ~ListGuard() { s->ext_list = nullptr; }
    } list_guard{&st->owner};
    st->owner.ext_list = live_list;
    if (st->owner.n_slices) {  // coarse levels: slice-owner items (disjoint gradient ranges)
        NvoProfMute mute;
        if (fork) {
            NVO_CHECK_HIP(hipEventRecord(st->ev_fork, stream));
            NVO_CHECK_HIP(hipStreamWaitEvent(st->aux, st->ev_fork, 0));
        }
        if (int rc = nvo_grid_bwd_launch(g, &st->owner, fork ? st->aux : stream, N, x, dy, dy_fmt, soa, grad, 1))
            return rc;
        if (fork) NVO_CHECK_HIP(hipEventRecord(st->ev_join, st->aux));
    }
    if (st->n_bins == 0) return NVO_OK;
    // tile-local 12-byte pair records, two 32-bit fixed-point sums per 64-bit accumulator word, 8192-entry bins
    // (k_tl_scatter_p / k_tl_accumulate_p)
    const size_t tile_records = (size_t)tile * 8;
    const size_t rec_bytes_tl = nvo_round_up((size_t)st->n_levels * n_tiles * tile_records * 12, 256);
    const size_t seg_bytes = nvo_round_up((size_t)st->n_bins * n_tiles * sizeof(uint32_t), 256);
    const size_t need_tl = rec_bytes_tl + seg_bytes * 2;
    if (int rc = nvo_scratch_reserve(&st->work, need_tl, stream, "grid_bwd_stream records")) return rc;
    unsigned char* d_work = static_cast<unsigned char*>(st->work.ptr);
    uint2* records_tl = reinterpret_cast<uint2*>(d_work);
    uint32_t* seg = reinterpret_cast<uint32_t*>(d_work + rec_bytes_tl);
    const dim3 grid_tl(n_tiles, st->n_levels);
    NVO_REQUIRE(st->max_slices <= 4096, "grid_bwd_stream: level too large (%u bins)", st->max_slices);
    NVO_REQUIRE(tile == 512 || tile == 1024, "grid_bwd_stream: the packed accumulators take 512- or 1024-sample tiles");
    uint32_t* segl1 = reinterpret_cast<uint32_t*>(d_work + rec_bytes_tl + seg_bytes);
    const size_t lds_p = tile_records * 12 + (sizeof(unsigned long long) + sizeof(uint32_t)) * st->max_slices;
    const size_t lds_acc_p = sizeof(unsigned long long) * st->bin_entries;
    const uint32_t acc_grid = st->n_tl_slots ? st->n_tl_slots : (st->n_tl_items < 2 * n_cus ? st->n_tl_items : 2 * n_cus);
#define NVO_LAUNCH_TLP_L(SOA_, T_, BIN_, TILE_, LISTED_)                                                      \
    do {                                                                                                      \
        static bool attr_set = false;                                                                         \
        if (!attr_set) {                                                                                      \
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_tl_scatter_p<TILE_, SOA_, T_, BIN_, LISTED_>,    \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(TILE_ * 96 + 12 * 4096))); \
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_tl_accumulate_p<BIN_, LISTED_>,                  \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8 * BIN_)));  \
            attr_set = true;                                                                                  \
        }                                                                                                     \
        {                                                                                                     \
            NVO_PROF_SUB(stream, "tl_scatter[L%u]", g.n_levels);                                              \
            NVO_LAUNCH((k_tl_scatter_p<TILE_, SOA_, T_, BIN_, LISTED_>), grid_tl, dim3(TILE_), lds_p, stream, g, N, x, (const T_*)dy, \
                       st->d_levels, st->d_bin_first, seg, segl1, reinterpret_cast<uint32_t*>(records_tl),     \
                       st->owner.nf_flag, live_list, live_n);                                                 \
        }                                                                                                     \
        {                                                                                                     \
            NVO_PROF_SUB(stream, "tl_accumulate[L%u]", g.n_levels);                                           \
            if (!st->external_zero)                                                                           \
                NVO_LAUNCH(k_st_zero_p<BIN_>, dim3(st->n_bins), dim3(256), 0, stream, g, st->d_bin_level, st->d_bin_slice, \
                           st->d_bin_chunks, grad);                                                           \
            NVO_LAUNCH((k_tl_accumulate_p<BIN_, LISTED_>), dim3(acc_grid), dim3(kTlBlockP), lds_acc_p, stream, g, \
                       (const uint4*)st->d_tl_items, st->n_tl_items, seg, segl1,                              \
                       reinterpret_cast<const uint32_t*>(records_tl), n_tiles,                                \
                       (uint32_t)tile_records, grad, st->owner.nf_flag, st->adam, live_n, N);                 \
        }                                                                                                     \
    } while (0)
/* (the listed forms are separate instantiations: see k_tl_accumulate_p) */                                   
#define NVO_LAUNCH_TLP_B(SOA_, T_, BIN_, TILE_)                                                               \
    do {                                                                                                      \
        if (live_list) NVO_LAUNCH_TLP_L(SOA_, T_, BIN_, TILE_, true);                                         \
        else NVO_LAUNCH_TLP_L(SOA_, T_, BIN_, TILE_, false);                                                  \
    } while (0)
#define NVO_LAUNCH_TLP(SOA_, T_)                                                           \
    do {                                                                                   \
        if (st->bin_entries != kBinP) NVO_LAUNCH_TLP_B(SOA_, T_, kBinPSmall, 512);         \
        else if (tile == 1024) NVO_LAUNCH_TLP_B(SOA_, T_, kBinP, 1024);                    \
        else NVO_LAUNCH_TLP_B(SOA_, T_, kBinP, 512);                                       \
    } while (0)
    if (soa) NVO_DY_DISPATCH(NVO_LAUNCH_TLP, true); else NVO_DY_DISPATCH(NVO_LAUNCH_TLP, false);
#undef NVO_LAUNCH_TLP
#undef NVO_LAUNCH_TLP_B
#undef NVO_LAUNCH_TLP_L
    NVO_CHECK_LAUNCH();
    if (fork) NVO_CHECK_HIP(hipStreamWaitEvent(stream, st->ev_join, 0));  // join
    return NVO_OK;
}

// mode: 0 = global atomics, 1 = LDS slice owner.  dy_fmt selects float2 vs half2 input.
int nvo_grid_bwd_launch(const NvoGridLevels& g, const NvoGridSlices* slices, hipStream_t stream,
                        uint32_t N, const float* x, const void* dy, int dy_fmt, bool soa,
                        float* grad, int mode) {
    const size_t grad_bytes = sizeof(float) * 2 * (size_t)g.offset[g.n_levels];
    const bool owner_form = mode == 1 && slices && slices->n_slices;
    NVO_PROF(stream, "grid_bwd_%s[L%u]", owner_form ? "lds" : "atomic", g.n_levels);
    if (N == 0) {
        return nvo_zero_async(grad, grad_bytes, stream);
    }
    NVO_REQUIRE(g.n_features == 2, "grid: only n_features_per_level == 2 is supported");
    if (owner_form) {
        if (slices->zero_last > slices->zero_first && !slices->external_zero) {
            // chunked (atomically flushed) levels form one contiguous run of entries
            if (int rc = nvo_zero_async(grad + 2 * (size_t)slices->zero_first,
                                        sizeof(float) * 2 * (size_t)(slices->zero_last - slices->zero_first), stream))
                return rc;
        }
        const dim3 grid(slices->n_slices), block(kLdsBwdBlock);
        size_t lds = slices->lds_bytes;
        // hit queue of the hashed levels' items (kHitCap): the wave rings sit behind the 32-bit accumulators
        uint32_t ring_off = 0u;
        {
            static const bool hitq = [] { const char* e = getenv("NVO_GRID_HITQ"); return !e || atoi(e) != 0; }();
            bool any_hashed = false;
            for (uint32_t l = 0; l < g.n_levels; ++l) any_hashed = any_hashed || (((slices->level_mask >> l) & 1u) && g.hashed[l]);
            if (hitq && any_hashed && slices->acc_bits == 32 && lds == 2 * AccFixed32::kEntries * sizeof(int) &&
                lds + kHitRingBytes <= kLdsBwdBytes - 256) {
                ring_off = (uint32_t)lds;
                lds += kHitRingBytes;
            }
        }
        const uint32_t* live = slices->ext_list;  // (a stream layout's launcher has listed them already: k_live_rows)
        bool l1_fused = false;
        if (!live && slices->compact_live && !slices->deterministic) {  // (the list's append order would change the run sums)
            if (int rc = nvo_scratch_reserve(&slices->live, sizeof(uint32_t) * ((size_t)N + 1), stream, "grid_bwd live list"))
                return rc;
            uint32_t* const d_live = static_cast<uint32_t*>(slices->live.ptr);
            if (!slices->external_zero)  // (otherwise cleared by the step's zero launch: nvo_grid_slices_zero_ranges)
                if (int rc = nvo_zero_async(slices->d_live_n, sizeof(uint32_t), stream)) return rc;
            // (32-bit accumulators, <= 8 levels: the L1 norms of dy ride in the same pass)
            l1_fused = slices->acc_bits == 32 && g.n_levels <= 8 && dy_fmt != NVO_DY_FLOAT && !slices->ext_l1;
            if (l1_fused && !slices->external_zero)
                if (int rc = nvo_zero_async(slices->d_l1, sizeof(unsigned long long) * 2 * g.n_levels, stream)) return rc;
#define NVO_LAUNCH_LIVE(SOA_, T_)                                                                                    \
    NVO_LAUNCH((k_live_samples<SOA_, T_>), dim3(nvo_div_up(N, 4096)), dim3(1024), 0, stream, g, N, (const T_*)dy, \
               slices->d_live_n, d_live, l1_fused ? slices->d_l1 : nullptr, slices->ext_live, slices->ext_blocks, \
               (soa && !l1_fused) ? slices->ext_dout : nullptr)
            if (soa) { NVO_DY_DISPATCH(NVO_LAUNCH_LIVE, true); } else { NVO_DY_DISPATCH(NVO_LAUNCH_LIVE, false); }
#undef NVO_LAUNCH_LIVE
            live = d_live;
        }
        if (slices->acc_bits == 32 && !l1_fused && !slices->ext_l1) {
            NVO_REQUIRE(dy_fmt != NVO_DY_FLOAT, "grid: 32-bit accumulators need 16-bit dL/dy (set grid_acc_bits to 64)");
            if (!slices->external_zero)
                if (int rc = nvo_zero_async(slices->d_l1, sizeof(unsigned long long) * 2 * g.n_levels, stream)) return rc;
            uint32_t bx = nvo_div_up(N, 256 * 16);
            if (bx > 256) bx = 256;
            if (bx < 1) bx = 1;
            const uint32_t all_levels = g.n_levels >= 32 ? 0xFFFFFFFFu : ((1u << g.n_levels) - 1u);
            const uint32_t l1_mask = slices->level_mask & all_levels;
#define NVO_LAUNCH_L1(SOA_, T_) \
    NVO_LAUNCH((k_dy_l1<SOA_, T_>), dim3(bx, (uint32_t)__builtin_popcount(l1_mask)), dim3(256), 0, stream, g, N, (const T_*)dy, \
               slices->d_l1, l1_mask)
            if (soa) {
                if (dy_fmt == NVO_DY_BF16) NVO_LAUNCH_L1(true, Bf2); else NVO_LAUNCH_L1(true, __half2);
            } else {
                if (dy_fmt == NVO_DY_BF16) NVO_LAUNCH_L1(false, Bf2); else NVO_LAUNCH_L1(false, __half2);
            }
#undef NVO_LAUNCH_L1
        }
        // slice codes of the hashed levels (k_slice_codes; see the coded scan of grid_bwd_item): one pre-pass over the
        // positions instead of a cell + hash derivation on every slice visit
        const uint16_t* d_codes = nullptr;
        uint32_t code_levels = 0u;
        {
            // NVO_GRID_SLICE_CODES=1 (A/B; default OFF -- a measured negative, EXPERIMENTS 10.4: 201 -> 212 us on 1 M live
            // samples, 97 -> 108 us with 60 % dead: a drained entry re-derives cell, hashes and weights from an uncoalesced
            // position load, which costs what the 59 % of skipped visits save)
            static const bool codes_env = [] { const char* e = getenv("NVO_GRID_SLICE_CODES"); return e && atoi(e) != 0; }();
            if (codes_env && ring_off != 0u && dy_fmt != NVO_DY_FLOAT && N < (1u << 28)) {
                for (uint32_t l = 0; l < g.n_levels; ++l) {
                    const uint32_t size = g.offset[l + 1] - g.offset[l];
                    if (((slices->level_mask >> l) & 1u) && g.hashed[l] && (size & (size - 1u)) == 0u && size >= 16384u &&
                        size <= 8u * 16384u && g.resolution[l] + 1u < 16384u)
                        code_levels |= 1u << l;
                }
            }
            if (code_levels) {
                const size_t bytes = sizeof(uint16_t) * (size_t)__builtin_popcount(code_levels) * N;
                if (int rc = nvo_scratch_reserve(&slices->codes, bytes, stream, "grid_bwd slice codes")) return rc;
                uint16_t* dc = static_cast<uint16_t*>(slices->codes.ptr);
                NVO_LAUNCH(k_slice_codes, dim3(nvo_div_up(N, 256)), dim3(256), 0, stream, g, N, x, code_levels, dc);
                d_codes = dc;
            }
        }
#define NVO_LAUNCH_LDS(SOA_, T_)                                                              \
    do {                                                                                      \
        static bool attr_set = false; /* >64 KiB of dynamic LDS needs an explicit opt-in */   \
        if (!attr_set) {                                                                      \
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_grid_bwd_lds<SOA_, T_>,          \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                              (int)kLdsBwdBytes - 256)); /* (static: work counter) */ \
            attr_set = true;                                                                  \
        }                                                                                     \
        NVO_LAUNCH((k_grid_bwd_lds<SOA_, T_>), grid, block, lds, stream, g, N, x,     \
                           (const T_*)dy, grad, (const uint4*)slices->d_level, slices->d_l1, live, slices->nf_flag, slices->d_live_n, ring_off, \
                           slices->ext_l1, slices->ext_blocks, slices->ext_l1_stride, d_codes, code_levels,         \
                           slices->ext_list ? 1024u : 4096u);                                                   \
    } while (0)
        if (soa) {
            NVO_DY_DISPATCH(NVO_LAUNCH_LDS, true);
        } else {
            NVO_DY_DISPATCH(NVO_LAUNCH_LDS, false);
        }
#undef NVO_LAUNCH_LDS
        NVO_CHECK_LAUNCH();
        return NVO_OK;
    }
    if (int rc = nvo_zero_async(grad, grad_bytes, stream)) return rc;
    const uint32_t tiles = nvo_div_up(N, kGridBlock / 2);
    const dim3 grid(tiles * g.n_levels), block(kGridBlock);
#define NVO_LAUNCH_AT(SOA_, T_)                                                               \
    NVO_LAUNCH((k_grid_bwd_atomic<SOA_, T_>), grid, block, 0, stream, g, N, x,        \
                       (const T_*)dy, grad)
    if (soa) {
        NVO_DY_DISPATCH(NVO_LAUNCH_AT, true);
    } else {
        NVO_DY_DISPATCH(NVO_LAUNCH_AT, false);
    }
#undef NVO_LAUNCH_AT
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_grid_bwd_input_dydx_launch(const NvoGridLevels& g, hipStream_t stream, uint32_t N, const void* dydx_half,
                                   const void* dy, int dy_fmt, bool soa, float* dx, bool zero_dx) {
    if (N == 0) return NVO_OK;
    NVO_PROF(stream, "grid_bwd_input_dydx[L%u]", g.n_levels);
    const dim3 grid(nvo_div_up(N, 256)), block(256);
#define NVO_LAUNCH_IN(SOA_, T_)                                                                              \
    NVO_LAUNCH((k_grid_bwd_input_dydx<SOA_, T_>), grid, block, 0, stream, g, N, (const __half2*)dydx_half,   \
               (const T_*)dy, dx, zero_dx ? 0 : 1)
    if (soa) {
        NVO_DY_DISPATCH(NVO_LAUNCH_IN, true);
    } else {
        NVO_DY_DISPATCH(NVO_LAUNCH_IN, false);
    }
#undef NVO_LAUNCH_IN
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_grid_bwd_input_launch(const NvoGridLevels& g, hipStream_t stream, uint32_t N,
                              const float* x, const void* table_half, const void* dy,
                              int dy_fmt, bool soa, float* dx, bool zero_dx, NvoGridInputScratch* scratch) {
    if (N == 0) return NVO_OK;
    NVO_PROF(stream, "grid_bwd_input[L%u]", g.n_levels);
    float* partial = nullptr;
    if (scratch) {  // two-stage form: per-level partials + a sum kernel, no float atomics, no zeroing of dx
        const size_t need = (size_t)g.n_levels * N * 3;
        if (int rc = nvo_scratch_reserve(scratch, sizeof(float) * need, stream, "grid_bwd_input partials")) return rc;
        partial = static_cast<float*>(scratch->ptr);
    } else if (zero_dx) {
        if (int rc = nvo_zero_async(dx, sizeof(float) * 3 * (size_t)N, stream)) return rc;
    }
    const uint32_t tiles = nvo_div_up(N, kGridBlock);
    const dim3 grid(tiles * g.n_levels), block(kGridBlock);
#define NVO_LAUNCH_IN(SOA_, T_)                                                               \
    NVO_LAUNCH((k_grid_bwd_input<SOA_, T_>), grid, block, 0, stream, g, N, x,         \
                       (const __half2*)table_half, (const T_*)dy, dx, partial)
    if (soa) {
        NVO_DY_DISPATCH(NVO_LAUNCH_IN, true);
    } else {
        NVO_DY_DISPATCH(NVO_LAUNCH_IN, false);
    }
#undef NVO_LAUNCH_IN
    if (partial) {
        const size_t n = (size_t)N * 3;
        NVO_LAUNCH(k_sum_levels, dim3(nvo_div_up(n, 256)), dim3(256), 0, stream, g.n_levels, n, partial, dx,
                   zero_dx ? 0 : 1);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

#ifdef NVO_GRID_PHASE
// out48: the phase-clock slots described at the top of this file; reset != 0 clears them afterwards
extern "C" int nvo_debug_grid_phase(unsigned long long* out48, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(nvo_grid_phase_cycles), sizeof(unsigned long long) * 48) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[48] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nvo_grid_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
