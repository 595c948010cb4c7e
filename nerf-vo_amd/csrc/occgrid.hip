// Cascaded Morton-bitfield occupancy grid for gfx950: DDA ray marcher with deterministic packed
// output, density-grid EMA update, bitfield construction + cascade max-pool.
// Replaces instant-ngp's generate_training_samples_nerf / ema_grid_samples_nerf / grid_to_bitfield /
// bitfield_max_pool (SURVEY.md section 2.4 K13/K16 -- the `mapping_module: 'instant-ngp'` back-end,
// /root/reference/nerf_vo/mapping/instant_ngp.py:33-50,104-105) and nerfacc's traverse_grids (K17).
// CPU restatement: oracle/c/nvo_oracle.c (bit-exact: same IEEE operations in the same order; this
// file is compiled with -ffp-contract=off).
//
// MI355X notes
//  * The whole 3-cascade bitfield is 768 KiB: it lives in L2 (4 MiB per XCD) after the first touch;
//    one lane marches one ray (divergent trip counts), a wave = 64 neighbouring rays.
//  * Packing is deterministic: pass 1 counts the occupied steps per ray, a single-workgroup
//    wave-scan turns counts into offsets (ballot-free exclusive prefix sum with a carry across 64-lane
//    chunks), pass 2 re-marches and writes (ray, t, dt) at the ray's offset.  No global atomics, no
//    dependence on dispatch order; a ray whose samples would exceed the capacity gets count 0 in
//    the scan (instant-ngp drops such rays too).
#include "nvo_kernels.h"
#include "../../include/nerfvo_hip.h"

namespace {

constexpr int kG = 128;
constexpr uint32_t kCells = 128u * 128u * 128u;
constexpr uint32_t kMaxSteps = 1024u;

__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__device__ __forceinline__ float min_step() { return 1.7320508075688772f / 1024.0f; }
__device__ __forceinline__ float max_step() { return (1.7320508075688772f / 1024.0f) * 128.0f * 1024.0f / 128.0f; }
__device__ __forceinline__ float calc_dt(float t, float cone_angle) {
    float dt = t * cone_angle;
    if (dt < min_step()) dt = min_step();
    if (dt > max_step()) dt = max_step();
    return dt;
}
__device__ __forceinline__ int mip_from_pos(const float* p, int max_mip) {
    float m = fabsf(p[0] - 0.5f);
    if (fabsf(p[1] - 0.5f) > m) m = fabsf(p[1] - 0.5f);
    if (fabsf(p[2] - 0.5f) > m) m = fabsf(p[2] - 0.5f);
    int e;
    frexpf(m, &e);
    int mip = e + 1;
    if (mip < 0) mip = 0;
    if (mip > max_mip) mip = max_mip;
    return mip;
}
__device__ __forceinline__ int mip_from_dt(float dt, const float* p, int max_mip) {
    int mip = mip_from_pos(p, max_mip);
    dt *= 2.0f * (float)kG;
    if (dt < 1.0f) return mip;
    int e;
    frexpf(dt, &e);
    if (e > mip) mip = e;
    if (mip > max_mip) mip = max_mip;
    return mip;
}
__device__ __forceinline__ uint32_t cell_index(const float* p, int mip) {
    const float s = scalbnf(1.0f, -mip);
    int i[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = (p[k] - 0.5f) * s + 0.5f;
        i[k] = (int)(v * (float)kG);
        if (i[k] < 0 || i[k] >= kG) return 0xFFFFFFFFu;
    }
    return morton3d((uint32_t)i[0], (uint32_t)i[1], (uint32_t)i[2]);
}
__device__ __forceinline__ bool occupied(const float* p, const uint8_t* __restrict__ bitfield, int mip) {
    const uint32_t idx = cell_index(p, mip);
    if (idx == 0xFFFFFFFFu) return false;
    return (bitfield[idx / 8 + (size_t)mip * (kCells / 8)] >> (idx % 8)) & 1;
}
__device__ __forceinline__ float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
// instant-ngp advance_to_next_voxel, first half: the ray parameter at which the ray leaves the cell of cascade `mip`
// that holds p.  The march then steps along its FIXED progression t <- t + calc_dt(t) until t >= that target (at least
// one step), i.e. the sample positions of a ray do not depend on the occupancy, only which of them are visited.
__device__ __forceinline__ float voxel_exit_target(float t, const float* p, const float* d, const float* idir, int mip) {
    const float res = scalbnf((float)kG, -mip);
    float tmin = 3.0e38f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float q = res * (p[k] - 0.5f);
        const float tk = (floorf(q + 0.5f + 0.5f * sgn(d[k])) - q) * idir[k];
        if (tk < tmin) tmin = tk;
    }
    float dist = tmin / res;
    if (!(dist > 0.0f)) dist = 0.0f;
    return t + dist;
}
__device__ __forceinline__ float advance_to_next_voxel(float t, float cone_angle, const float* p, const float* d,
                                                       const float* idir, int mip) {
    const float t_target = voxel_exit_target(t, p, d, idir, mip);
    do {
        t += calc_dt(t, cone_angle);
    } while (t < t_target);
    return t;
}

// ONE march per ray: accepted samples (t, dt) go to a ray-major scratch [R][kMaxSteps] and their number to
// counts[r]; k_occ_compact then copies the runs to their scanned offsets.  (The first form marched twice -- count
// pass, scan, write pass -- and a march is a chain of dependent bitfield loads, ~0.36 us per step and 0.37 ms per
// launch for 4096 rays however little it writes.)
// Samples are staged in LDS (kStage per lane) and leave in bursts: global stores and the bitfield loads of the next
// step share one in-order counter (vmcnt), so a store per accepted sample made every following occupancy test wait for
// that store to reach memory (measured: 5.2 ms with per-sample stores against 0.37 ms without any).  With staging
// a wave stalls once per kStage accepted samples.
// Resumable like k_occ_march_wave (t_resume / max_new / t_next / run_offset, same meaning, same samples).  This form is
// what LARGE launches take (inference: >= 49 152 rays): a candidate costs one lane ~25 instructions here and one WAVE ~11
// there (every lane of a ray's wave evaluates the same fp32 recurrence), so with enough rays to fill the chip it is the
// cheaper one by far -- 816 000 rays marched to the end: 9.5 ms wave-per-ray whatever the launch size, ray-per-lane 13.5 ms
// in launches of 16 K rays, 3.6 ms at 65 K, 1.6 ms at 262 K (tools/probes/march_forms.py) -- while a training batch of 2-16 K
// rays leaves most SIMDs without a wave and is a chain of ~1000 dependent loads per ray.
__global__ void __launch_bounds__(256)
k_occ_march(uint32_t R, const float* __restrict__ origins, const float* __restrict__ directions,
            const uint8_t* __restrict__ bitfield, int n_levels, float cone_angle, float t_near,
            const float* __restrict__ jitter, uint32_t* __restrict__ counts, float2* __restrict__ scratch,
            const float* __restrict__ t_resume, uint32_t max_new, float* __restrict__ t_next,
            const uint32_t* __restrict__ R_dev, uint32_t run_offset) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (R_dev) R = min(R, *R_dev);
    // (no early return: the staging flush below is a wave-wide vote -- lanes without a ray simply have nothing to march)
    const bool has_ray = r < R;
    const uint32_t rc = has_ray ? r : 0u;
    const uint32_t budget = max_new < kMaxSteps ? max_new : kMaxSteps;
    const float resume = (t_resume && has_ray) ? t_resume[rc] : 0.f;
    const bool alive = has_ray && !(t_resume && !(resume >= 0.f));
    float next = -1.f;
    const int max_mip = n_levels - 1;
    const float half = 0.5f * (float)(1 << max_mip);
    const float lo = 0.5f - half, hi = 0.5f + half;
    float o[3], d[3], idir[3];
    float tmin = t_near, tmax = 3.0e38f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = origins[3 * (size_t)rc + k];
        d[k] = directions[3 * (size_t)rc + k];
        idir[k] = 1.0f / d[k];
        float t0 = (lo - o[k]) * idir[k], t1 = (hi - o[k]) * idir[k];
        if (t0 > t1) { const float tt = t0; t0 = t1; t1 = tt; }
        if (t0 > tmin) tmin = t0;
        if (t1 < tmax) tmax = t1;
    }
    constexpr uint32_t kStage = 16;
    __shared__ float2 stage[kStage][256];
    float2* __restrict__ run = scratch + (size_t)rc * kMaxSteps + run_offset;
    uint32_t staged = 0;
    auto flush = [&](uint32_t j_now) {
        const uint32_t first = j_now - staged;
#pragma unroll 4
        for (uint32_t q = 0; q < kStage; ++q)
            if (q < staged) run[first + q] = stage[q][threadIdx.x];
        staged = 0;
    };
    uint32_t j = 0;
    if (alive && tmax > tmin) {
        float t = t_resume ? resume : tmin + calc_dt(tmin, cone_angle) * (jitter ? jitter[rc] : 0.f);
        // (2^20 candidates: an exit every lane reaches even for a degenerate ray whose skip target is not finite -- the
        // bound of k_occ_march_wave's block loop)
        uint32_t guard = 1u << 20;
        while (guard) {
            float p[3];
            bool inside = true;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                p[k] = o[k] + d[k] * t;
                inside = inside && p[k] >= lo && p[k] <= hi;
            }
            if (!inside) break;  // left the box
            if (j >= budget) {   // sample budget of the ray (of this round) spent: this candidate is where it resumes
                next = t;
                break;
            }
            const float dt = calc_dt(t, cone_angle);
            const int mip = mip_from_dt(dt, p, max_mip);
            if (occupied(p, bitfield, mip)) {
                stage[staged][threadIdx.x] = make_float2(t, dt);
                ++staged;
                ++j;
                t += dt;
                --guard;
            } else {
                const float t_target = voxel_exit_target(t, p, d, idir, mip);
                do {  // (advance_to_next_voxel, with the guard)
                    t += calc_dt(t, cone_angle);
                    --guard;
                } while (t < t_target && guard);
            }
            // every lane still marching empties its stage when ANY of them is full: one burst, one stall
            if (__any(staged == kStage)) flush(j);
        }
    }
    if (staged) flush(j);
    if (has_ray) {
        counts[r] = j;
        if (t_next) t_next[r] = next;
    }
}

// ONE WAVE PER RAY (the default).  A ray's candidate positions are the fixed progression t_0, t_1 = t_0 + calc_dt(t_0), ...
// whatever the occupancy (see voxel_exit_target), so a wave tests 64 consecutive candidates at once -- one bitfield
// round trip per 64 candidates instead of one per candidate, and 4096 waves instead of 64 for a 4096-ray batch (the
// ray-per-lane form above keeps 960 of the chip's 1024 SIMDs idle and is a chain of ~1000 dependent loads) -- and then
// replays the sequential visiting order on wave-wide ballots: the next visited candidate is the first one at or past
// the current skip target; an occupied one is accepted and steps to its successor (a whole run of occupied candidates
// is accepted with one mask operation), an empty one raises the skip target to the exit of its cell.  The accepted
// samples leave through a popcount prefix over the ballot: coalesced stores in ray order, no atomics, and every
// (t, dt) is bit-identical to the sequential march (the progression itself is evaluated with the same fp32 recurrence).
__global__ void __launch_bounds__(256)
k_occ_march_wave(uint32_t R, const float* __restrict__ origins, const float* __restrict__ directions,
                 const uint8_t* __restrict__ bitfield, int n_levels, float cone_angle, float t_near,
                 const float* __restrict__ jitter, uint32_t* __restrict__ counts, float2* __restrict__ scratch,
                 const float* __restrict__ t_resume, uint32_t max_new, float* __restrict__ t_next,
                 const uint32_t* __restrict__ R_dev, uint32_t run_offset) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    if (R_dev) R = min(R, *R_dev);  // the launch covers the buffers' rows, the batch is the first *R_dev of them
    if (r >= R) return;
    // RESUMABLE form (inference in rounds): the ray takes up its progression at candidate t_resume[r] (a value an earlier
    // launch handed out through t_next: the recurrence depends on t alone, so the samples are the ones a single march
    // would have found), accepts at most max_new samples and reports the candidate it stopped in front of (-1: the ray
    // has left the scene box, or was not alive: t_resume[r] < 0)
    const uint32_t budget = max_new < kMaxSteps ? max_new : kMaxSteps;
    const float resume = t_resume ? t_resume[r] : 0.f;
    if (t_resume && !(resume >= 0.f)) {
        if (lane == 0u) {
            counts[r] = 0u;
            if (t_next) t_next[r] = -1.f;
        }
        return;
    }
    float next = -1.f;  // (uniform)
    const int max_mip = n_levels - 1;
    const float half = 0.5f * (float)(1 << max_mip);
    const float lo = 0.5f - half, hi = 0.5f + half;
    float o[3], d[3], idir[3];
    float tmin = t_near, tmax = 3.0e38f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = origins[3 * (size_t)r + k];
        d[k] = directions[3 * (size_t)r + k];
        idir[k] = 1.0f / d[k];
        float t0 = (lo - o[k]) * idir[k], t1 = (hi - o[k]) * idir[k];
        if (t0 > t1) { const float tt = t0; t0 = t1; t1 = tt; }
        if (t0 > tmin) tmin = t0;
        if (t1 < tmax) tmax = t1;
    }
    // (run_offset: a later round appends behind the samples the ray's earlier rounds left in its run)
    float2* __restrict__ run = scratch + (size_t)r * kMaxSteps + run_offset;
    uint32_t j = 0;  // accepted so far (uniform)
    if (tmax > tmin) {
        float t = t_resume ? resume : tmin + calc_dt(tmin, cone_angle) * (jitter ? jitter[r] : 0.f);  // candidate 0 of the current block
        float skip_to = -3.0e38f;  // (uniform) candidates before this ray parameter are stepped over
        bool done = false;
        // (16384 blocks = 2^20 candidates: an exit every wave reaches even for a degenerate ray whose skip target is
        // not finite -- a regular ray leaves a cascade-2 box after ~1000 candidates)
        for (uint32_t block = 0; !done && block < 16384u; ++block) {
            // lane k keeps the k-th value of the progression; every lane evaluates the same recurrence
            float my_t = t;
            for (uint32_t k = 0; k < 64u; ++k) {
                if (lane == k) my_t = t;
                t += calc_dt(t, cone_angle);
            }
            float p[3];
            bool inside = true;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                p[k] = o[k] + d[k] * my_t;
                inside = inside && p[k] >= lo && p[k] <= hi;
            }
            const float dt = calc_dt(my_t, cone_angle);
            const int mip = mip_from_dt(dt, p, max_mip);
            const bool occ = inside && occupied(p, bitfield, mip);
            const float exit_t = (inside && !occ) ? voxel_exit_target(my_t, p, d, idir, mip) : 0.f;
            const unsigned long long in_mask = __ballot(inside), occ_mask = __ballot(occ);
            unsigned long long accept = 0ull;
            uint32_t pos = 0;
            while (pos < 64u) {
                const unsigned long long cand = __ballot(my_t >= skip_to) & (~0ull << pos);
                if (cand == 0ull) break;  // the rest of the block is stepped over
                const uint32_t k = (uint32_t)__builtin_ctzll(cand);
                if (!((in_mask >> k) & 1ull)) {  // left the box
                    done = true;
                    break;
                }
                if (j >= budget) {  // sample budget of the ray (of this round) spent: candidate k is where it resumes
                    next = nvo_wave_bcast(my_t, (int)k);
                    done = true;
                    break;
                }
                if ((occ_mask >> k) & 1ull) {
                    // an occupied candidate steps to its successor: the whole run of occupied candidates is visited
                    const unsigned long long stop = ~occ_mask & (~0ull << k);
                    uint32_t u = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                    const uint32_t room = budget - j;
                    if (u - k > room) u = k + room;
                    accept |= (u >= 64u ? ~0ull : ((1ull << u) - 1ull)) & (~0ull << k);
                    j += u - k;
                    pos = u;
                    skip_to = -3.0e38f;
                } else {
                    skip_to = nvo_wave_bcast(exit_t, (int)k);
                    pos = k + 1u;
                }
            }
            if ((accept >> lane) & 1ull) {
                const uint32_t before = (uint32_t)__popcll(accept & ((1ull << lane) - 1ull));
                const uint32_t first = j - (uint32_t)__popcll(accept);
                run[first + before] = make_float2(my_t, dt);
            }
        }
    }
    if (lane == 0u) {
        counts[r] = j;
        if (t_next) t_next[r] = next;
    }
}

// one wave per ray: scratch run -> packed arrays at the scanned offset (rays dropped by the capacity clamp have
// counts[r] == 0)
__global__ void __launch_bounds__(256)
k_occ_compact(uint32_t R, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
              const float2* __restrict__ scratch, int32_t* __restrict__ ray_idx, float* __restrict__ t_out,
              float* __restrict__ dt_out, const uint32_t* __restrict__ R_dev, uint32_t run_offset) {
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (R_dev) R = min(R, *R_dev);
    if (r >= R) return;
    const uint32_t n = counts[r], base = offsets[r];
    const float2* __restrict__ run = scratch + (size_t)r * kMaxSteps + run_offset;
    for (uint32_t k = threadIdx.x & 63u; k < n; k += 64u) {
        const float2 v = run[k];
        ray_idx[base + k] = (int32_t)r;
        t_out[base + k] = v.x;
        dt_out[base + k] = v.y;
    }
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int) { return nvo_wave_incl_scan(v); }  // DPP

// Single-workgroup exclusive scan of counts_in[n] -> offsets[n], total in offsets[n]; entries that would
// push the running total beyond `capacity` are treated as 0 in counts_out (counts_out == counts_in: in place).
// totals (nullable): [0] = the total BEFORE the clamp, [1] = min(total, capacity) = the packed slots in use.
// LDS form (n <= kScanLds entries): the counts come in and the results leave with coalesced accesses; between them
// thread i owns the PER consecutive entries [i * PER, (i + 1) * PER) of the staged array -- one scan of the 1024 thread sums
// across the workgroup, no round structure.  (History: rounds of 1024 entries with three barriers each, then a thread's
// entries straight from global memory -- 64 different lines per wave instruction on ONE CU: 19-21 us for 13 K rays either
// way, three launches per step.)
constexpr uint32_t kScanLds = 16384;
template <int PER, bool LDS>
__global__ void __launch_bounds__(1024)
k_scan_counts(uint32_t n, const uint32_t* counts_in, uint32_t* counts_out, uint32_t* __restrict__ offsets,
              uint32_t capacity, uint32_t* __restrict__ totals, const uint32_t* __restrict__ n_dev) {
    extern __shared__ uint32_t stage[];  // LDS: [n] counts -> offsets, [n] counts_out behind them
    __shared__ uint32_t wave_tot[16];
    if (n_dev) n = min(n, *n_dev);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first = threadIdx.x * PER;
    uint32_t c[PER];
    if constexpr (LDS) {
        for (uint32_t i = threadIdx.x; i < n; i += 1024) stage[i] = counts_in[i];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) c[k] = (first + (uint32_t)k < n) ? stage[first + k] : 0u;
    } else {
#pragma unroll
        for (int k = 0; k < PER; ++k) c[k] = counts_in[min(first + (uint32_t)k, n ? n - 1u : 0u)];
    }
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (first + (uint32_t)k >= n) c[k] = 0u;
        sum += c[k];
    }
    const uint32_t incl = wave_incl_scan_u32(sum, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t prefix = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) prefix += wave_tot[w];
        total += wave_tot[w];
    }
    uint32_t run = prefix + incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = first + (uint32_t)k;
        if (i < n) {
            // capacity clamp: rays whose samples do not fit are dropped (count 0); offsets stay monotone
            const uint32_t kept = (run + c[k] > capacity) ? 0u : c[k];
            if constexpr (LDS) {
                stage[i] = run;
                stage[kScanLds + i] = kept;
            } else {
                offsets[i] = run;
                counts_out[i] = kept;
            }
            run += c[k];
        }
    }
    if constexpr (LDS) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            offsets[i] = stage[i];
            counts_out[i] = stage[kScanLds + i];
        }
    }
    if (threadIdx.x == 0) {
        offsets[n] = total;
        if (totals) {
            totals[0] = total;
            totals[1] = total < capacity ? total : capacity;
        }
    }
}

// ---- scan + compact (+ positions) in ONE launch ----------------------------------------------------------------
// nvo_occ_pack spends three launches on what is a few microseconds of work: a single-workgroup scan (10-17 us for
// 4-13 K rays: one CU, latency all the way), the copy of the runs, and -- in the training step of the occupancy-grid
// back-end -- the network input of every packed slot; three such sequences per step.  Here a workgroup owns 16 rays (64 left
// 47 workgroups for a 3 K-ray batch whose runs are hundreds of samples long):
// it scans their counts, PUBLISHES its total, reads the totals of the workgroups in front of it (a decoupled look-back
// without aggregation: at most 4095 eight-byte words, 16 per thread), and goes on to copy its rays' runs and write their positions.
// Cross-workgroup hand-off (MI355X_MICROARCH.md, "8-B agent atomics both sides"): a total travels as ONE 64-bit word
// {epoch + 1, total} written with an agent-scope atomic store and polled with agent-scope atomic loads -- no payload
// behind a flag, nothing that needs a release / acquire pair.  The epoch lives in the caller's state block and is advanced
// by the LAST workgroup to finish (a second counter), so the words of this launch are stale for the next one without
// anybody clearing them; launches that share a state block must be ordered (same stream).
// Forward progress: a workgroup only waits for workgroups with a LOWER index, which the dispatcher has started before it;
// (workgroups are dispatched in index order: whatever a resident workgroup waits for is resident or done).
struct OccPackState {
    unsigned long long sums[65536 / 16];
    uint32_t epoch, done;
};

template <uint32_t kPackRays, int kPackRaysLog2>
__global__ void __launch_bounds__(256)
k_occ_pack_fused(uint32_t R, const uint32_t* counts_in, uint32_t* counts_out, uint32_t* __restrict__ offsets,
                 uint32_t capacity, uint32_t* __restrict__ totals, const float2* __restrict__ scratch, uint32_t run_offset,
                 int32_t* __restrict__ ray_idx, float* __restrict__ t_out, float* __restrict__ dt_out,
                 const uint32_t* __restrict__ R_dev, OccPackState* __restrict__ st, const float* __restrict__ origins,
                 const float* __restrict__ directions, float aabb_lo, float aabb_inv_size, float* __restrict__ x01) {
    __shared__ uint32_t s_off[kPackRays], s_cnt[kPackRays], s_c[kPackRays], s_part[4];
    __shared__ float s_od[kPackRays][6];  // origin | direction of the workgroup's rays
    __shared__ uint32_t s_prefix;
    if (R_dev) R = min(R, *R_dev);
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t epoch = __hip_atomic_load(&st->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t tag = epoch + 1u;
    const uint32_t r0 = b * kPackRays;
    if (r0 >= R && b != 0u) {
        // past the batch (the launch covers the workspace's rows): nothing to scan, publish or copy -- only the check-in
        // that lets the last workgroup retire the epoch
        if (threadIdx.x == 0u) {
            const uint32_t before = __hip_atomic_fetch_add(&st->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before == gridDim.x - 1u) {
                __hip_atomic_store(&st->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&st->epoch, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    const uint32_t b_last = R ? (R - 1u) / kPackRays : 0u;  // the workgroup that holds the last ray in use
    // ---- this workgroup's 64 counts: exclusive scan in wave 0, total published
    uint32_t c = 0, excl = 0;
    if (wave == 0u) {
        const uint32_t r = r0 + lane;
        c = (lane < kPackRays && r < R) ? counts_in[r] : 0u;
        const uint32_t incl = nvo_wave_incl_scan(c);
        excl = incl - c;
        if (lane == 63u)
            __hip_atomic_store(&st->sums[b], ((unsigned long long)tag << 32) | incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (x01) {
        for (uint32_t i = threadIdx.x; i < kPackRays * 6u; i += 256u) {
            const uint32_t q = i / 6u, a = i - q * 6u, r = r0 + q;
            s_od[q][a] = r < R ? (a < 3u ? origins[3 * (size_t)r + a] : directions[3 * (size_t)r + a - 3u]) : 0.f;
        }
    }
    // ---- totals of the workgroups in front: every thread polls its share
    // (four words per thread requested at once -- they are all there when the workgroups of a launch start together --
    // then the stragglers one by one: a thread that polled its words one after the other spent a round trip on each)
    uint32_t part = 0;
    for (uint32_t j0 = threadIdx.x; j0 < b; j0 += 4u * 256u) {
        unsigned long long v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t j = j0 + (uint32_t)u * 256u;
            v[u] = j < b ? __hip_atomic_load(&st->sums[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t j = j0 + (uint32_t)u * 256u;
            while ((uint32_t)(v[u] >> 32) != tag) v[u] = __hip_atomic_load(&st->sums[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            part += (uint32_t)v[u];
        }
    }
    part = nvo_wave_bcast(nvo_wave_incl_scan(part), 63);
    if (lane == 0u) s_part[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0u) s_prefix = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    const uint32_t prefix = s_prefix;
    if (wave == 0u) {
        const uint32_t r = r0 + lane;
        const uint32_t off = prefix + excl;
        // capacity clamp: rays whose samples do not fit are dropped (count 0); offsets stay monotone
        const uint32_t kept = (off + c > capacity) ? 0u : c;
        if (lane < kPackRays) {
            s_off[lane] = off;
            s_cnt[lane] = kept;
            s_c[lane] = c;
            if (r < R) {
                offsets[r] = off;
                counts_out[r] = kept;
            }
        }
        if (lane == 63u && b == b_last) {  // (the workgroup of the last ray in use ends the scan)
            const uint32_t total = off + c;
            offsets[R] = total;
            if (totals) {
                totals[0] = total;
                totals[1] = total < capacity ? total : capacity;
            }
        }
    }
    __syncthreads();
    // ---- copy the runs and write the network input of every copied sample.  The workgroup's slots [prefix, prefix + its
    // total) are dealt to the threads one by one (a wave per ray left half the lanes idle on 20-32-sample runs and walked
    // its 16 rays one dependent load -> store after the other: 18.6 us per launch); a slot finds its ray by a 6-step
    // search of the 64 offsets in LDS
    const uint32_t blk_total = s_off[kPackRays - 1u] + s_c[kPackRays - 1u] - prefix;
    for (uint32_t p0 = threadIdx.x; p0 < blk_total; p0 += 4u * 256u) {
        uint32_t slot[4], q[4], k[4];
        float2 v[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t p = p0 + (uint32_t)u * 256u;
            slot[u] = prefix + p;
            uint32_t lo_ = 0u, hi_ = kPackRays;  // largest q with s_off[q] <= slot
#pragma unroll
            for (int it = 0; it < kPackRaysLog2; ++it) {
                const uint32_t mid = (lo_ + hi_) >> 1;
                if (s_off[mid] <= slot[u]) lo_ = mid; else hi_ = mid;
            }
            q[u] = lo_;
            k[u] = slot[u] - s_off[lo_];
            live[u] = p < blk_total && k[u] < s_cnt[lo_];  // (a dropped ray keeps its slot range, count 0)
            v[u] = make_float2(0.f, 0.f);
            if (live[u]) v[u] = scratch[(size_t)(r0 + q[u]) * kMaxSteps + run_offset + k[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!live[u]) continue;
            const uint32_t r = r0 + q[u];
            ray_idx[slot[u]] = (int32_t)r;
            t_out[slot[u]] = v[u].x;
            dt_out[slot[u]] = v[u].y;
            if (x01) {  // (k_ngp_positions' arithmetic: same values)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float w = s_od[q[u]][a] + s_od[q[u]][3 + a] * v[u].x;
                    x01[3 * (size_t)slot[u] + a] = fminf(fmaxf((w - aabb_lo) * aabb_inv_size, 0.f), 1.f);
                }
            }
        }
    }
    // ---- the last workgroup to get here retires the epoch (its words are stale for the next launch from then on)
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t before = __hip_atomic_fetch_add(&st->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == gridDim.x - 1u) {
            __hip_atomic_store(&st->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->epoch, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// instant-ngp ema_grid_samples_nerf: never-seen cells (negative) stay; others max(decay * old, new)
__global__ void __launch_bounds__(256)
k_occ_ema(uint64_t n, float* __restrict__ grid, const float* __restrict__ fresh, float decay) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float prev = grid[i];
    grid[i] = prev < 0.f ? prev : fmaxf(prev * decay, fresh[i]);
}

// sum of max(v, 0) over cascade 0 (fixed-point so the mean is order independent / reproducible)
__global__ void __launch_bounds__(256)
k_occ_sum(const float* __restrict__ grid, unsigned long long* __restrict__ acc) {
    // 2^-20 resolution fixed point: per-thread partial over a grid-stride range, wave reduction, one LDS add per wave and
    // ONE global add per workgroup (512 workgroups; it was one same-address atomic from each of 8192: 103 us)
    unsigned long long part = 0ull;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < kCells; i += gridDim.x * blockDim.x)
        part += (unsigned long long)(long long)llrintf(fmaxf(grid[i], 0.f) * 1048576.0f);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    __shared__ unsigned long long s;
    if (threadIdx.x == 0) s = 0ull;
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&s, part);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, s);
}

__global__ void __launch_bounds__(256)
k_occ_bitfield(const float* __restrict__ grid, int n_levels, float threshold,
               const unsigned long long* __restrict__ acc, uint8_t* __restrict__ bitfield) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (uint64_t)n_levels * (kCells / 8)) return;
    const float mean = (float)((double)(*acc) / 1048576.0 / (double)kCells);
    const float th = mean < threshold ? mean : threshold;
    uint8_t bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (grid[b * 8 + j] > th) bits |= (uint8_t)(1u << j);
    bitfield[b] = bits;
}

// coarse level `l` |= 2x2x2 max-pool of level l-1 over its inner half (one thread per coarse cell)
__global__ void __launch_bounds__(256)
k_occ_maxpool(int l, uint8_t* __restrict__ bitfield) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64u * 64u * 64u) return;
    const uint32_t x = i & 63u, y = (i >> 6) & 63u, z = i >> 12;
    const uint8_t* fine = bitfield + (size_t)(l - 1) * (kCells / 8);
    const uint32_t fi = morton3d(2 * x, 2 * y, 2 * z);  // multiple of 8: one byte = the 2x2x2 block
    if (fine[fi / 8]) {
        const uint32_t ci = morton3d(x + 32, y + 32, z + 32);
        // 8 threads may share a destination byte: set the bit atomically on the enclosing word
        uint32_t* word = reinterpret_cast<uint32_t*>(bitfield + (size_t)l * (kCells / 8)) + (ci / 32);
        atomicOr(word, 1u << (ci % 32));
    }
}

// Counter-based generator of the density-grid refresh samples: every value is a hash of (seed, stream, step, element), so
// the launch is stateless (the same generator as the pixel sampler of rays.hip).
__device__ __forceinline__ uint32_t occ_pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float occ_hash_uniform(uint32_t seed, uint32_t step, uint32_t stream, uint32_t i) {
    const uint32_t h = occ_pcg_hash(occ_pcg_hash(occ_pcg_hash(seed ^ (stream * 0x9E3779B9u)) + step) + i);
    return (float)(h >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

// Refresh samples of the density grid past the warm-up [UPSTREAM instant-ngp testbed_nerf.cu,
// generate_grid_samples_nerf_nonuniform]: sample i takes a cascade at random and the first of ten candidate cells
//   idx_j = ((i + step * n_total) * 56924617 + j * 19349663 + 96925573) mod 128^3        (32-bit wrap-around)
// whose grid value exceeds `thresh` (the tenth stays when none does), then a uniform point inside that cell.  Writes the
// point as the density network's input (position in the scene box, clamped to [0, 1]) and the cell it belongs to.
__global__ void __launch_bounds__(256)
k_occ_sample_cells(uint32_t n, uint32_t first, uint32_t n_total, uint32_t step, uint32_t seed, uint32_t stream_id,
                   int n_levels, const float* __restrict__ grid, float thresh, float aabb_lo, float aabb_hi,
                   float* __restrict__ x01, uint32_t* __restrict__ cell_idx) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = first + k;
    uint32_t level = (uint32_t)(occ_hash_uniform(seed, step, 4u * stream_id + 0u, i) * (float)n_levels);
    if (level >= (uint32_t)n_levels) level = (uint32_t)n_levels - 1u;
    const uint32_t base = (i + step * n_total) * 56924617u + 96925573u;
    uint32_t idx = 0;
    for (uint32_t j = 0; j < 10u; ++j) {
        idx = (base + j * 19349663u) & (kCells - 1u);
        if (grid[(size_t)level * kCells + idx] > thresh) break;
    }
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // inverse Morton
        uint32_t v = (idx >> a) & 0x49249249u;
        v = (v ^ (v >> 2)) & 0xC30C30C3u;
        v = (v ^ (v >> 4)) & 0x0F00F00Fu;
        v = (v ^ (v >> 8)) & 0xFF0000FFu;
        v = (v ^ (v >> 16)) & 0x0000FFFFu;
        c[a] = v;
    }
    const float scale = scalbnf(1.0f, (int)level);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float u = ((float)c[a] + occ_hash_uniform(seed, step, 4u * stream_id + 1u + (uint32_t)a, i)) / (float)kG;
        const float p = (u - 0.5f) * scale + 0.5f;
        float q = (p - aabb_lo) / (aabb_hi - aabb_lo);
        q = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
        x01[3 * (size_t)k + a] = q;
    }
    cell_idx[k] = level * kCells + idx;
}

// Cells no training camera sees are taken out of training [UPSTREAM instant-ngp mark_untrained_density_grid]: a cell of
// any cascade stays trainable iff one of its eight corners lies in front of some camera (cos of the angle to the viewing
// axis >= 1e-4) and projects strictly inside that camera's image.  Pinhole cameras in the engine's convention (OpenGL
// axes: pixel (px, py) looks along R ((px - cx) / fx, -(py - cy) / fy, -1), nvo_rays_given).  margin = 0 is upstream's
// rule; see the projection below for margin > 0.  grid: trainable cells that
// were marked come back as 0, cells that lose their last view become -1 (never occupied, never refreshed); all others keep
// their value.
__global__ void __launch_bounds__(256)
k_occ_mark_untrained(uint64_t n, float* __restrict__ grid, uint32_t n_images, const float* __restrict__ intrinsics,
                     const float* __restrict__ c2w, float W, float H, float margin) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t level = (uint32_t)(i / kCells), idx = (uint32_t)(i % kCells);
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // inverse Morton
        uint32_t v = (idx >> a) & 0x49249249u;
        v = (v ^ (v >> 2)) & 0xC30C30C3u;
        v = (v ^ (v >> 4)) & 0x0F00F00Fu;
        v = (v ^ (v >> 8)) & 0xFF0000FFu;
        v = (v ^ (v >> 16)) & 0x0000FFFFu;
        c[a] = v;
    }
    const float scale = scalbnf(1.0f, (int)level);
    const float size = scalbnf(1.0f / (float)kG, (int)level);
    float p0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) p0[a] = ((float)c[a] / (float)kG - 0.5f) * scale + 0.5f;
    bool seen = false;
    for (uint32_t j = 0; j < n_images && !seen; ++j) {
        const float* __restrict__ m = c2w + 12 * (size_t)j;
        const float fx = intrinsics[4 * j + 0], fy = intrinsics[4 * j + 1], cx = intrinsics[4 * j + 2], cy = intrinsics[4 * j + 3];
        for (uint32_t k = 0; k < 8u && !seen; ++k) {
            const float d[3] = {p0[0] + ((k & 1u) ? size : 0.f) - m[3], p0[1] + ((k & 2u) ? size : 0.f) - m[7],
                                p0[2] + ((k & 4u) ? size : 0.f) - m[11]};
            // camera frame: q = R^T d
            const float qx = m[0] * d[0] + m[4] * d[1] + m[8] * d[2];
            const float qy = m[1] * d[0] + m[5] * d[1] + m[9] * d[2];
            const float qz = m[2] * d[0] + m[6] * d[1] + m[10] * d[2];
            const float depth = -qz;
            const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            if (!(depth >= 1e-4f * len) || !(depth > 0.f)) continue;
            const float px = cx + fx * (qx / depth), py = cy - fy * (qy / depth);
            // margin > 0: the image grows by `margin` projected cell diagonals on every side, so that a cell the
            // frustum only clips (no corner inside it) keeps its view
            const float mx = margin * (fx * (size * 1.7320508f / depth)), my = margin * (fy * (size * 1.7320508f / depth));
            seen = px > -mx && py > -my && px < W + mx && py < H + my;
        }
    }
    const float g = grid[i];
    if ((g < 0.f) != !seen) grid[i] = seen ? 0.f : -1.f;
}

// cell centres of one cascade in Morton order, normalised frame; jitter [cells][3] in [0,1) or null
__global__ void __launch_bounds__(256)
k_occ_cell_positions(int level, const float* __restrict__ jitter, float* __restrict__ pos) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= kCells) return;
    // inverse Morton
    uint32_t c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint32_t v = (x >> k) & 0x49249249u;
        v = (v ^ (v >> 2)) & 0xC30C30C3u;
        v = (v ^ (v >> 4)) & 0x0F00F00Fu;
        v = (v ^ (v >> 8)) & 0xFF0000FFu;
        v = (v ^ (v >> 16)) & 0x0000FFFFu;
        c[k] = v;
    }
    const float scale = scalbnf(1.0f, level);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float u = ((float)c[k] + (jitter ? jitter[3 * (size_t)x + k] : 0.5f)) / (float)kG;
        pos[3 * (size_t)x + k] = (u - 0.5f) * scale + 0.5f;
    }
}

}  // namespace

extern "C" {

uint64_t nvo_occ_march_scratch_bytes(uint32_t R) { return (uint64_t)sizeof(float2) * R * kMaxSteps; }

int nvo_occ_march(nvo_stream_t stream, uint32_t R, const float* origins, const float* directions,
                  const uint8_t* bitfield, int n_levels, float cone_angle, float t_near, const float* jitter,
                  uint32_t capacity, uint32_t* counts, uint32_t* offsets, int32_t* ray_idx, float* t_out,
                  float* dt_out, void* scratch, uint64_t scratch_bytes) {
    return nvo_occ_march_resume(stream, R, origins, directions, bitfield, n_levels, cone_angle, t_near, jitter, capacity, counts,
                                offsets, ray_idx, t_out, dt_out, scratch, scratch_bytes, nullptr, kMaxSteps, nullptr);
}

int nvo_occ_march_resume(nvo_stream_t stream, uint32_t R, const float* origins, const float* directions,
                         const uint8_t* bitfield, int n_levels, float cone_angle, float t_near, const float* jitter,
                         uint32_t capacity, uint32_t* counts, uint32_t* offsets, int32_t* ray_idx, float* t_out,
                         float* dt_out, void* scratch, uint64_t scratch_bytes, const float* t_resume, uint32_t max_new,
                         float* t_next) {
    NVO_REQUIRE(R == 0 || (counts && offsets && ray_idx && t_out && dt_out), "occ_march: NULL argument");
    if (int rc = nvo_occ_march_runs(stream, R, origins, directions, bitfield, n_levels, cone_angle, t_near, jitter, counts,
                                    scratch, scratch_bytes, t_resume, max_new, t_next, nullptr, 0u))
        return rc;
    return nvo_occ_pack(stream, R, counts, capacity, counts, offsets, nullptr, scratch, scratch_bytes, ray_idx, t_out, dt_out,
                        nullptr, 0u);
}

int nvo_occ_march_runs(nvo_stream_t stream, uint32_t R, const float* origins, const float* directions,
                       const uint8_t* bitfield, int n_levels, float cone_angle, float t_near, const float* jitter,
                       uint32_t* counts, void* scratch, uint64_t scratch_bytes, const float* t_resume, uint32_t max_new,
                       float* t_next, const uint32_t* R_dev, uint32_t run_offset) {
    NVO_REQUIRE(n_levels >= 1 && n_levels <= 8, "occ_march: n_levels %d not in 1..8", n_levels);
    NVO_REQUIRE(max_new >= 1, "occ_march: max_new must be positive");
    NVO_REQUIRE((uint64_t)run_offset + (max_new < kMaxSteps ? max_new : kMaxSteps) <= kMaxSteps,
                "occ_march: run_offset %u + max_new %u pass the %u samples a run holds", run_offset, max_new, kMaxSteps);
    NVO_REQUIRE(R == 0 || (origins && directions && bitfield && counts), "occ_march: NULL argument");
    if (R == 0) return NVO_OK;
    // ray-major staging area of the single march: CALLER-owned (it used to be a process-global block that was freed
    // and re-allocated whenever a larger R arrived -- a captured graph would have kept the dangling pointer)
    NVO_REQUIRE(scratch && scratch_bytes >= nvo_occ_march_scratch_bytes(R),
                "occ_march: scratch of %llu bytes is too small for %u rays (nvo_occ_march_scratch_bytes: %llu)",
                (unsigned long long)scratch_bytes, R, (unsigned long long)nvo_occ_march_scratch_bytes(R));
    hipStream_t s = (hipStream_t)stream;
    float2* march_scratch = static_cast<float2*>(scratch);
    NVO_PROF(stream, "occ_march");
    // ray-per-lane for launches that fill the chip with rays (inference bundles), wave-per-ray for training batches;
    // NVO_OCC_MARCH_LANES = 0 | 1 forces one form (A/B, tests)
    static const int lanes_env = [] { const char* e = getenv("NVO_OCC_MARCH_LANES"); return e ? atoi(e) : -1; }();
    const bool ray_per_lane = lanes_env >= 0 ? lanes_env != 0 : R >= 49152u;
    if (ray_per_lane) {
        NVO_LAUNCH(k_occ_march, dim3(nvo_div_up(R, 256)), dim3(256), 0, s, R, origins, directions, bitfield,
                   n_levels, cone_angle, t_near, jitter, counts, march_scratch, t_resume, max_new, t_next, R_dev, run_offset);
    } else {
        NVO_LAUNCH(k_occ_march_wave, dim3(nvo_div_up(R, 4)), dim3(256), 0, s, R, origins, directions, bitfield,
                   n_levels, cone_angle, t_near, jitter, counts, march_scratch, t_resume, max_new, t_next, R_dev, run_offset);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_occ_pack(nvo_stream_t stream, uint32_t R, const uint32_t* counts_in, uint32_t capacity, uint32_t* counts_out,
                 uint32_t* offsets, uint32_t* totals, const void* scratch, uint64_t scratch_bytes, int32_t* ray_idx,
                 float* t_out, float* dt_out, const uint32_t* R_dev, uint32_t run_offset) {
    NVO_REQUIRE(run_offset < kMaxSteps, "occ_pack: run_offset %u outside a run", run_offset);
    NVO_REQUIRE(R == 0 || (counts_in && counts_out && offsets && scratch && ray_idx && t_out && dt_out), "occ_pack: NULL argument");
    NVO_REQUIRE(scratch_bytes >= nvo_occ_march_scratch_bytes(R), "occ_pack: scratch too small for %u rays", R);
    if (R == 0) return NVO_OK;
    hipStream_t s = (hipStream_t)stream;
    {
        NVO_PROF(stream, "occ_scan");
        NVO_REQUIRE(R <= 65536u, "occ_pack: at most 65536 rays per launch (got %u)", R);
        const size_t lds = 2 * kScanLds * sizeof(uint32_t);
        static bool attr_set = false;
        if (!attr_set) {
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_scan_counts<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            NVO_CHECK_HIP(hipFuncSetAttribute((const void*)k_scan_counts<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        if (R <= 4096u) NVO_LAUNCH((k_scan_counts<4, true>), dim3(1), dim3(1024), lds, s, R, counts_in, counts_out, offsets, capacity, totals, R_dev);
        else if (R <= kScanLds) NVO_LAUNCH((k_scan_counts<16, true>), dim3(1), dim3(1024), lds, s, R, counts_in, counts_out, offsets, capacity, totals, R_dev);
        else NVO_LAUNCH((k_scan_counts<64, false>), dim3(1), dim3(1024), 0, s, R, counts_in, counts_out, offsets, capacity, totals, R_dev);
        NVO_CHECK_LAUNCH();
    }
    {
        NVO_PROF(stream, "occ_compact");
        NVO_LAUNCH(k_occ_compact, dim3(nvo_div_up(R, 4)), dim3(256), 0, s, R, counts_out, offsets,
                   static_cast<const float2*>(scratch), ray_idx, t_out, dt_out, R_dev, run_offset);
        NVO_CHECK_LAUNCH();
    }
    return NVO_OK;
}

uint64_t nvo_occ_pack_state_bytes(void) { return sizeof(OccPackState); }

int nvo_occ_pack_fused(nvo_stream_t stream, uint32_t R, const uint32_t* counts_in, uint32_t capacity, uint32_t* counts_out,
                       uint32_t* offsets, uint32_t* totals, const void* scratch, uint64_t scratch_bytes, int32_t* ray_idx,
                       float* t_out, float* dt_out, const uint32_t* R_dev, uint32_t run_offset, void* state,
                       const float* origins, const float* directions, float aabb_lo, float aabb_hi, float* x01,
                       uint32_t rays_per_group) {
    NVO_REQUIRE(rays_per_group == 16u || rays_per_group == 64u, "occ_pack_fused: rays_per_group is 16 or 64 (got %u)", rays_per_group);
    NVO_REQUIRE(R == 0 || (counts_in && counts_out && offsets && scratch && ray_idx && t_out && dt_out && state),
                "occ_pack_fused: NULL argument");
    NVO_REQUIRE(scratch_bytes >= nvo_occ_march_scratch_bytes(R), "occ_pack_fused: scratch too small for %u rays", R);
    NVO_REQUIRE(run_offset < kMaxSteps, "occ_pack_fused: run_offset %u outside a run", run_offset);
    NVO_REQUIRE(nvo_div_up(R, rays_per_group) <= 4096u, "occ_pack_fused: at most %u rays per launch with %u rays per workgroup (got %u)",
                4096u * rays_per_group, rays_per_group, R);
    NVO_REQUIRE(!x01 || (origins && directions && aabb_hi > aabb_lo), "occ_pack_fused: positions need origins, directions and a box");
    NVO_REQUIRE((((uintptr_t)state) & 7u) == 0u, "occ_pack_fused: the state block must be 8-byte aligned");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "occ_pack");
    // 16 rays per workgroup while the batch is small (a 3 K-ray batch of early training has runs of hundreds of samples:
    // 64-ray workgroups would leave 47 of them to copy everything), 64 once it is large (856 workgroups polling each
    // other's 8-byte words hammer a few L2 channels: 19 us per launch against 13)
    if (rays_per_group == 16u)
        NVO_LAUNCH((k_occ_pack_fused<16, 4>), dim3(nvo_div_up(R, 16)), dim3(256), 0, (hipStream_t)stream, R, counts_in, counts_out,
                   offsets, capacity, totals, static_cast<const float2*>(scratch), run_offset, ray_idx, t_out, dt_out, R_dev,
                   static_cast<OccPackState*>(state), origins, directions, aabb_lo, x01 ? 1.0f / (aabb_hi - aabb_lo) : 0.f, x01);
    else
        NVO_LAUNCH((k_occ_pack_fused<64, 6>), dim3(nvo_div_up(R, 64)), dim3(256), 0, (hipStream_t)stream, R, counts_in, counts_out,
                   offsets, capacity, totals, static_cast<const float2*>(scratch), run_offset, ray_idx, t_out, dt_out, R_dev,
                   static_cast<OccPackState*>(state), origins, directions, aabb_lo, x01 ? 1.0f / (aabb_hi - aabb_lo) : 0.f, x01);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_occ_update(nvo_stream_t stream, int n_levels, float* grid, const float* fresh, float decay,
                   float threshold, uint8_t* bitfield, void* scratch8) {
    NVO_REQUIRE(n_levels >= 1 && n_levels <= 8 && grid && bitfield && scratch8, "occ_update: bad argument");
    hipStream_t s = (hipStream_t)stream;
    NVO_PROF(stream, "occ_update");
    const uint64_t n = (uint64_t)n_levels * kCells;
    if (fresh) {
        NVO_LAUNCH(k_occ_ema, dim3(nvo_div_up(n, 256)), dim3(256), 0, s, n, grid, fresh, decay);
        NVO_CHECK_LAUNCH();
    }
    if (int rc = nvo_zero_async(scratch8, 8, s)) return rc;
    NVO_LAUNCH(k_occ_sum, dim3(512), dim3(256), 0, s, grid, (unsigned long long*)scratch8);
    NVO_CHECK_LAUNCH();
    NVO_LAUNCH(k_occ_bitfield, dim3(nvo_div_up(n / 8, 256)), dim3(256), 0, s, grid, n_levels, threshold,
               (const unsigned long long*)scratch8, bitfield);
    NVO_CHECK_LAUNCH();
    for (int l = 1; l < n_levels; ++l) {
        NVO_LAUNCH(k_occ_maxpool, dim3(64 * 64 * 64 / 256), dim3(256), 0, s, l, bitfield);
        NVO_CHECK_LAUNCH();
    }
    return NVO_OK;
}

int nvo_occ_cell_positions(nvo_stream_t stream, int level, const float* jitter, float* positions) {
    NVO_REQUIRE(level >= 0 && level < 8 && positions, "occ_cell_positions: bad argument");
    NVO_PROF(stream, "occ_cell_positions");
    NVO_LAUNCH(k_occ_cell_positions, dim3(kCells / 256), dim3(256), 0, (hipStream_t)stream, level, jitter, positions);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_occ_sample_cells(nvo_stream_t stream, uint32_t n, uint32_t first, uint32_t n_total, uint32_t step, uint32_t seed,
                         uint32_t stream_id, int n_levels, const float* grid, float thresh, float aabb_lo, float aabb_hi,
                         float* x01, uint32_t* cell_idx) {
    NVO_REQUIRE(n_levels >= 1 && n_levels <= 8 && grid && x01 && cell_idx && aabb_hi > aabb_lo && (uint64_t)first + n <= n_total,
                "occ_sample_cells: bad argument");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "occ_sample_cells");
    NVO_LAUNCH(k_occ_sample_cells, dim3(nvo_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, first, n_total, step, seed,
               stream_id, n_levels, grid, thresh, aabb_lo, aabb_hi, x01, cell_idx);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_occ_mark_untrained(nvo_stream_t stream, int n_levels, float* grid, uint32_t n_images, const float* intrinsics,
                           const float* c2w, uint32_t H, uint32_t W, float margin) {
    NVO_REQUIRE(n_levels >= 1 && n_levels <= 8 && grid && (n_images == 0 || (intrinsics && c2w)) && H > 0 && W > 0 && margin >= 0.f,
                "occ_mark_untrained: bad argument");
    NVO_PROF(stream, "occ_mark_untrained");
    const uint64_t n = (uint64_t)n_levels * kCells;
    NVO_LAUNCH(k_occ_mark_untrained, dim3(nvo_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, grid, n_images, intrinsics,
               c2w, (float)W, (float)H, margin);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

}  // extern "C"
