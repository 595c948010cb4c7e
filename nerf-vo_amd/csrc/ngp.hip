// Occupancy-grid ("instant-ngp") training back-end for gfx950: per-sample positions for PACKED
// variable-length rays, front-to-back compositing fused with the rgb / depth L2 losses and their
// per-sample gradients, density -> optical-thickness conversion for the density-grid update.
// Replaces instant-ngp's compute_loss_kernel_train_nerf (+ the NeRF-SLAM fork's depth term) and the
// glue around generate_training_samples_nerf / update_density_grid_nerf (SURVEY.md section 2.4
// K13-K16; reference call sites /root/reference/nerf_vo/mapping/instant_ngp.py:33-50,87-105 --
// aabb_scale 4, L2 depth loss, extrinsics optimisation).  Upstream sources are not vendored; the
// arithmetic is restated in oracle/ngp.py.
//
// One WAVE per ray: a ray's samples are contiguous in the packed arrays (nvo_occ_march), lanes stride
// over them in chunks of 64; transmittance is a multiplicative wave scan with a carry across chunks.
#include "nvo_kernels.h"
#include "../../include/nerfvo_hip.h"

namespace {

// (DPP forms, nvo_common.h)
__device__ __forceinline__ float wave_sum(float v) { return nvo_wave_sum(v); }
__device__ __forceinline__ float wave_incl_scan(float v, int) { return nvo_wave_incl_scan(v); }

// x01 = (o + t d - aabb_lo) * aabb_inv_size for every packed slot; slots with ray_idx < 0 -> zeros
__global__ void __launch_bounds__(256)
k_ngp_positions(uint32_t capacity, const int32_t* __restrict__ ray_idx, const float* __restrict__ t,
                const float* __restrict__ origins, const float* __restrict__ directions, float aabb_lo,
                float aabb_inv_size, float* __restrict__ x01, const uint32_t* __restrict__ n_live) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= capacity) return;
    // (slots in use known on the device: the network launches behind this one stop at the tile that holds the last of
    // them, rows up to the next multiple of 4096 are kept finite for that tile)
    if (n_live && i >= ((*n_live + 4095u) & ~4095u)) return;
    const int32_t r = ray_idx[i];
    float p[3] = {0.f, 0.f, 0.f};
    if (r >= 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float w = origins[3 * (size_t)r + k] + directions[3 * (size_t)r + k] * t[i];
            p[k] = fminf(fmaxf((w - aabb_lo) * aabb_inv_size, 0.f), 1.f);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) x01[3 * (size_t)i + k] = p[k];
}

// Backward of k_ngp_positions for the extrinsics optimiser: dL/dx01 of the packed samples -> per-ray
// dL/dorigin and dL/ddirection (x01 = clamp((o + t d - lo) / size, 0, 1): the clamp passes the gradient
// strictly inside the box).  One wave per ray over its packed range [offsets[r], offsets[r] + counts[r]).
__global__ void __launch_bounds__(256)
k_ngp_positions_bwd(uint32_t R, uint32_t capacity, const int32_t* __restrict__ counts,
                    const int32_t* __restrict__ offsets, const float* __restrict__ t,
                    const float* __restrict__ origins, const float* __restrict__ directions, float aabb_lo,
                    float aabb_inv_size, const float* __restrict__ dx01, float* __restrict__ d_origin,
                    float* __restrict__ d_dir, const uint32_t* __restrict__ R_dev) {
    const int lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    if (R_dev && r >= *R_dev) {  // rows past the batch: zero gradients (the per-camera sum behind this covers all R rows)
        if (lane < 3) {
            d_origin[3 * (size_t)r + lane] = 0.f;
            d_dir[3 * (size_t)r + lane] = 0.f;
        }
        return;
    }
    const uint32_t off = (uint32_t)offsets[r];
    uint32_t n = (uint32_t)counts[r];
    if (off >= capacity) n = 0;
    else if (off + n > capacity) n = capacity - off;  // samples beyond the packed capacity were dropped
    const float o[3] = {origins[3 * (size_t)r], origins[3 * (size_t)r + 1], origins[3 * (size_t)r + 2]};
    const float d[3] = {directions[3 * (size_t)r], directions[3 * (size_t)r + 1], directions[3 * (size_t)r + 2]};
    float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
    for (uint32_t s = lane; s < n; s += 64) {
        const float ts = t[off + s];
        const float* gx = dx01 + 3 * (size_t)(off + s);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x = (o[k] + d[k] * ts - aabb_lo) * aabb_inv_size;
            const float g = (x > 0.f && x < 1.f) ? gx[k] * aabb_inv_size : 0.f;
            go[k] += g;
            gd[k] += g * ts;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        go[k] = nvo_wave_sum(go[k]);
        gd[k] = nvo_wave_sum(gd[k]);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d_origin[3 * (size_t)r + k] = go[k];
            d_dir[3 * (size_t)r + k] = gd[k];
        }
    }
}

}  // namespace

namespace {

// Optical depth of one sample, density x step, capped at 128: beyond it alpha = 1 - exp(-dd) is exactly 1 and the
// transmittance behind the sample exactly 0 in fp32 either way, but an uncapped huge value (a pre-activation of 30 is
// 1e13 x dt) wipes the smaller terms out of the prefix sum the transmittances are read from (T = exp(-(incl - dd)):
// accumulations above 1 were observed with pre-activations of +-70) and turns 0 x inf into NaN in the gradient.
__device__ __forceinline__ float ngp_optical_step(float sigma, float dt) { return fminf(sigma * dt, 128.f); }

__global__ void __launch_bounds__(256)
k_ngp_composite_loss(nvo_ngp_loss_args a, uint32_t ray_blocks) {
    if (blockIdx.x >= ray_blocks) {
        // (uniform) workgroups behind the rays' zero the gradient rows of packed slots that belong to no ray (beyond the last
        // offset / dropped rays) -- rows no ray writes: a launch of its own before
        const uint32_t i = (blockIdx.x - ray_blocks) * blockDim.x + threadIdx.x;
        if (i >= a.capacity || a.ray_idx[i] >= 0) return;
        _Float16* __restrict__ d_rgb = (_Float16*)a.d_rgb_out;
        for (uint32_t k = 0; k < a.d_rgb_stride; ++k) d_rgb[(size_t)i * a.d_rgb_stride + k] = (_Float16)0.f;
        a.d_density_pre[i] = 0.f;
        return;
    }
    const int lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= a.R) return;
    if (a.R_dev) {  // the batch is the first *R_dev rows; the losses are means over them (times the ranks)
        const uint32_t R = *a.R_dev;
        if (r >= R) return;
        a.inv_rays = 1.f / (float)(R * (a.world_size ? a.world_size : 1u));
    }
    // 0: the ray's samples reach the end of its march; 1: they stop where the transmittance fell below the training
    // threshold (nvo_ngp_count_alive); 2: dropped where the march was packed (no trace in the losses)
    const uint32_t state = a.ray_state ? a.ray_state[r] : 0u;
    const uint32_t n = a.counts[r];
    const uint32_t base = a.offsets[r];
    const _Float16* den = (const _Float16*)a.density_out;
    const _Float16* col = (const _Float16*)a.rgb_out;

    // ---- training, a ray of at most 64 samples (with the compacted batch: nearly every ray -- a ray keeps ~20 samples):
    // ONE pass with everything in registers.  Same operations in the same order as the chunked passes below (their carries
    // are exact zeros for a single chunk), so the values are theirs bit for bit; the chunked form loads every sample three
    // times and evaluates its four exponentials three times.
    if (a.d_rgb_out && n <= 64u) {
        const uint32_t j = (uint32_t)lane;
        const bool valid = j < n;
        const size_t s = base + j;
        float sigma = 0.f, dd = 0.f, tj = 0.f, dts = 0.f, rgb[3] = {0.f, 0.f, 0.f};
        if (valid) {
            sigma = __expf((float)den[s * a.density_stride]);
            dts = a.dt[s];
            dd = ngp_optical_step(sigma, dts);
            tj = a.t[s];
#pragma unroll
            for (int k = 0; k < 3; ++k) rgb[k] = 1.f / (1.f + __expf(-(float)col[s * a.rgb_stride + k]));
        }
        const float incl = wave_incl_scan(dd, lane) + 0.f;
        const float T = __expf(-(incl - dd));
        const float w = valid ? (1.f - __expf(-dd)) * T : 0.f;
        float pix[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pix[k] = 0.f + wave_sum(w * rgb[k]);
        const float depth = 0.f + wave_sum(w * tj);
        const float acc = 0.f + wave_sum(w);
        const float carry1 = nvo_wave_bcast(incl, 63);
        const float T_final = __expf(-carry1);
        float bg[3] = {0.f, 0.f, 0.f};
        if (a.background && state == 0u) {
#pragma unroll
            for (int k = 0; k < 3; ++k) bg[k] = a.background[3 * (size_t)r + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) pix[k] += T_final * bg[k];
        if (lane == 0) {
            if (a.out_rgb) {
#pragma unroll
                for (int k = 0; k < 3; ++k) a.out_rgb[3 * (size_t)r + k] = 0.f + pix[k];
            }
            if (a.out_depth) a.out_depth[r] = 0.f + depth;
            if (a.out_accumulation) a.out_accumulation[r] = 0.f + acc;
        }
        if (n == 0u && a.offsets[r + 1] != base) return;
        if (state == 2u) return;
        float g_pix[3], l_rgb = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float e = pix[k] - a.gt_rgb[3 * (size_t)r + k];
            l_rgb += e * e;
            g_pix[k] = 2.f * e * a.inv_rays * (1.f / 3.f) * a.rgb_mult;
        }
        l_rgb *= a.inv_rays * (1.f / 3.f) * a.rgb_mult;
        float g_depth = 0.f, l_depth = 0.f;
        if (a.gt_depth && a.depth_mult != 0.f) {
            const float z = a.gt_depth[r] * (a.directions_norm ? a.directions_norm[r] : 1.f);
            float wgt = 1.f;
            if (a.gt_depth_cov) {
                const float var = a.gt_depth_cov[r];
                wgt = (var > 0.f && var < __builtin_inff()) ? 1.f / var : 0.f;
            }
            if (z > 0.f) {
                const float e = depth - z;
                l_depth = e * e * wgt * a.inv_rays * a.depth_mult;
                g_depth = 2.f * e * wgt * a.inv_rays * a.depth_mult;
            }
        }
        if (lane == 0) {
            float* shard = a.losses + 8 * (r & 63u);
            atomicAdd(shard + 0, l_rgb);
            atomicAdd(shard + 1, l_depth);
        }
        float dot = g_depth * tj;
#pragma unroll
        for (int k = 0; k < 3; ++k) dot += g_pix[k] * rgb[k];
        const float q = valid ? w * dot : 0.f;
        const float total_q = 0.f + wave_sum(q);
        const float g_bg = T_final * (g_pix[0] * bg[0] + g_pix[1] * bg[1] + g_pix[2] * bg[2]);
        const float incl_q = wave_incl_scan(q, lane) + 0.f;
        if (valid) {
            const float suffix = total_q - incl_q;
            const bool dead = T < a.train_min_transmittance;
            const float dsigma = dts * (T * __expf(-dd) * dot - suffix - g_bg);
            a.d_density_pre[s] = dead ? 0.f : dsigma * fminf(sigma, 3.2690173e6f) * a.loss_scale;
            _Float16* d_rgb = (_Float16*)a.d_rgb_out;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                d_rgb[s * a.d_rgb_stride + k] = dead ? (_Float16)0.f : (_Float16)(w * g_pix[k] * rgb[k] * (1.f - rgb[k]) * a.loss_scale);
            for (uint32_t k = 3; k < a.d_rgb_stride; ++k) d_rgb[s * a.d_rgb_stride + k] = (_Float16)0.f;
        }
        return;
    }

    // ---- pass 1: composite front to back (T = prod (1 - alpha) through a log-space additive scan)
    // sum of density * dt of all previous samples (inference in rounds: what earlier rounds of this ray have gathered)
    float carry = (a.carry_in && !a.d_rgb_out) ? a.carry_in[r] : 0.f;
    float pix[3] = {0.f, 0.f, 0.f};
    float depth = 0.f, acc = 0.f;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t j = c0 + lane;
        float dd = 0.f, tj = 0.f, rgb[3] = {0.f, 0.f, 0.f};
        if (j < n) {
            const size_t s = base + j;
            const float sigma = __expf((float)den[s * a.density_stride]);
            dd = ngp_optical_step(sigma, a.dt[s]);
            tj = a.t[s];
#pragma unroll
            for (int k = 0; k < 3; ++k) rgb[k] = 1.f / (1.f + __expf(-(float)col[s * a.rgb_stride + k]));
        }
        const float incl = wave_incl_scan(dd, lane) + carry;
        const float T = __expf(-(incl - dd));
        const float w = (j < n) ? (1.f - __expf(-dd)) * T : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) pix[k] += wave_sum(w * rgb[k]);
        depth += wave_sum(w * tj);
        acc += wave_sum(w);
        carry = nvo_wave_bcast(incl, 63);
    }
    const float T_final = __expf(-carry);
    float bg[3] = {0.f, 0.f, 0.f};
    // (a ray that was cut at the threshold sees no background: what is left of it, < 1e-4, belongs to the samples behind
    // the cut [UPSTREAM compute_loss_kernel_train_nerf: `if (compacted_numsteps == numsteps) rgb_ray += T * background`])
    if (a.background && state == 0u) {
#pragma unroll
        for (int k = 0; k < 3; ++k) bg[k] = a.background[3 * (size_t)r + k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) pix[k] += T_final * bg[k];
    if (lane == 0) {
        const bool add = a.accumulate_outputs != 0u && !a.d_rgb_out;
        if (a.out_rgb) {
#pragma unroll
            for (int k = 0; k < 3; ++k) a.out_rgb[3 * (size_t)r + k] = (add ? a.out_rgb[3 * (size_t)r + k] : 0.f) + pix[k];
        }
        if (a.out_depth) a.out_depth[r] = (add ? a.out_depth[r] : 0.f) + depth;
        if (a.out_accumulation) a.out_accumulation[r] = (add ? a.out_accumulation[r] : 0.f) + acc;
        if (a.carry_out && !a.d_rgb_out) a.carry_out[r] = carry;
    }
    if (!a.d_rgb_out) return;
    // a ray the scan dropped at the packed capacity (count zeroed, its slot range in `offsets` kept) leaves no trace in
    // the losses [UPSTREAM generate_training_samples_nerf: an overflowing ray returns before it is counted]
    if (n == 0u && a.offsets[r + 1] != base) return;
    if (state == 2u) return;

    // ---- losses (means over the global ray count)
    float g_pix[3], l_rgb = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float e = pix[k] - a.gt_rgb[3 * (size_t)r + k];
        l_rgb += e * e;
        g_pix[k] = 2.f * e * a.inv_rays * (1.f / 3.f) * a.rgb_mult;
    }
    l_rgb *= a.inv_rays * (1.f / 3.f) * a.rgb_mult;
    float g_depth = 0.f, l_depth = 0.f;
    if (a.gt_depth && a.depth_mult != 0.f) {
        const float z = a.gt_depth[r] * (a.directions_norm ? a.directions_norm[r] : 1.f);
        // covariance-weighted residual (gt_depth_cov: variance of the target; 1 -> the plain L2 term, exactly)
        float wgt = 1.f;
        if (a.gt_depth_cov) {
            const float var = a.gt_depth_cov[r];
            wgt = (var > 0.f && var < __builtin_inff()) ? 1.f / var : 0.f;
        }
        if (z > 0.f) {
            const float e = depth - z;
            l_depth = e * e * wgt * a.inv_rays * a.depth_mult;
            g_depth = 2.f * e * wgt * a.inv_rays * a.depth_mult;
        }
    }
    if (lane == 0) {
        float* shard = a.losses + 8 * (r & 63u);
        atomicAdd(shard + 0, l_rgb);
        atomicAdd(shard + 1, l_depth);
    }

    // ---- pass 2: per-sample gradients.  With q_i = w_i (g_pix . rgb_i + g_depth t_i):
    //   dL/dsigma_i = dt_i [ T_i exp(-dd_i) (g.rgb_i + g_depth t_i) - sum_{j>i} q_j - T_final g.bg ]
    //   dL/drgb_i   = w_i g_pix
    float total_q = 0.f;
    carry = 0.f;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {  // total of q
        const uint32_t j = c0 + lane;
        float dd = 0.f, q = 0.f;
        if (j < n) {
            const size_t s = base + j;
            dd = ngp_optical_step(__expf((float)den[s * a.density_stride]), a.dt[s]);
        }
        const float incl = wave_incl_scan(dd, lane) + carry;
        if (j < n) {
            const size_t s = base + j;
            const float T = __expf(-(incl - dd));
            const float w = (1.f - __expf(-dd)) * T;
            float dot = g_depth * a.t[s];
#pragma unroll
            for (int k = 0; k < 3; ++k) dot += g_pix[k] / (1.f + __expf(-(float)col[s * a.rgb_stride + k]));
            q = w * dot;
        }
        total_q += wave_sum(q);
        carry = nvo_wave_bcast(incl, 63);
    }
    const float g_bg = T_final * (g_pix[0] * bg[0] + g_pix[1] * bg[1] + g_pix[2] * bg[2]);
    carry = 0.f;
    float carry_q = 0.f;
    _Float16* d_rgb = (_Float16*)a.d_rgb_out;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t j = c0 + lane;
        float dd = 0.f, q = 0.f, w = 0.f, T = 0.f, dot = 0.f, sigma = 0.f, rgb[3] = {0.f, 0.f, 0.f};
        if (j < n) {
            const size_t s = base + j;
            sigma = __expf((float)den[s * a.density_stride]);
            dd = ngp_optical_step(sigma, a.dt[s]);
        }
        const float incl = wave_incl_scan(dd, lane) + carry;
        if (j < n) {
            const size_t s = base + j;
            T = __expf(-(incl - dd));
            w = (1.f - __expf(-dd)) * T;
            dot = g_depth * a.t[s];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                rgb[k] = 1.f / (1.f + __expf(-(float)col[s * a.rgb_stride + k]));
                dot += g_pix[k] * rgb[k];
            }
            q = w * dot;
        }
        const float incl_q = wave_incl_scan(q, lane) + carry_q;
        if (j < n) {
            const size_t s = base + j;
            const float suffix = total_q - incl_q;
            // a sample the ray reaches with less than train_min_transmittance left trains nothing (upstream stops the ray
            // there): exact zeros, which the backwards behind this kernel skip
            const bool dead = T < a.train_min_transmittance;
            const float dsigma = a.dt[s] * (T * __expf(-dd) * dot - suffix - g_bg);
            // density = exp(x): dsigma/dx = sigma (clamped like tcnn's Exponential activation backward)
            a.d_density_pre[s] = dead ? 0.f : dsigma * fminf(sigma, 3.2690173e6f) * a.loss_scale;
#pragma unroll
            for (int k = 0; k < 3; ++k)  // rgb = sigmoid(y): drgb/dy = rgb (1 - rgb)
                d_rgb[s * a.d_rgb_stride + k] = dead ? (_Float16)0.f : (_Float16)(w * g_pix[k] * rgb[k] * (1.f - rgb[k]) * a.loss_scale);
            for (uint32_t k = 3; k < a.d_rgb_stride; ++k) d_rgb[s * a.d_rgb_stride + k] = (_Float16)0.f;
        }
        carry = nvo_wave_bcast(incl, 63);
        carry_q = nvo_wave_bcast(incl_q, 63);
    }
}

// Where each ray ends for TRAINING [UPSTREAM compute_loss_kernel_train_nerf: the loop over a ray's samples opens with
// `if (T < EPSILON) break`, EPSILON = 1e-4; the samples in front of that point are the ones compacted into the batch
// that is trained on]: kept[r] = index of the first sample the ray reaches with a transmittance below `min_t` (its
// sample count when there is none).  Same arithmetic as the compositing pass of k_ngp_composite_loss (capped optical
// step, wave scan with a carry), density pre-activations in COMPACT form (one 16-bit value per slot) or with a stride.
// state[r]: 0 = kept everything, 1 = cut, 2 = the ray was dropped where the march was packed (kept 0).
__global__ void __launch_bounds__(256)
k_ngp_count_alive(nvo_ngp_alive_args a) {
    const int lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t R = a.R;
    if (a.R_dev) R = min(R, *a.R_dev);
    if (r >= R) return;
    if (a.resume_in && !(a.resume_in[r] >= 0.f)) {  // not part of this round: the earlier rounds settled it
        if (lane == 0 && a.resume_out) a.resume_out[r] = -1.f;
        return;
    }
    const uint32_t n = a.counts[r];
    const uint32_t base = a.offsets[r];
    const bool dropped = n == 0u && a.offsets[r + 1] != base;
    const _Float16* den = (const _Float16*)a.density_out;
    float carry = a.carry_in ? a.carry_in[r] : 0.f;
    uint32_t kept = n;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t j = c0 + lane;
        float dd = 0.f;
        if (j < n) {
            const size_t s = base + j;
            dd = ngp_optical_step(__expf((float)den[s * a.density_stride]), a.dt[s]);
        }
        const float incl = wave_incl_scan(dd, lane) + carry;
        const float T = __expf(-(incl - dd));
        const unsigned long long below = __ballot(j < n && T < a.min_transmittance);
        if (below) {
            kept = c0 + (uint32_t)__builtin_ctzll(below);
            break;
        }
        carry = nvo_wave_bcast(incl, 63);
    }
    if (lane == 0) {
        const bool cut = kept < n;
        a.kept[r] = dropped ? 0u : a.kept_base + kept;
        a.state[r] = dropped ? 2u : (cut ? 1u : 0u);
        if (a.resume_out) {
            // goes on: every sample of this round kept, and the march stopped at its budget, not at the scene box.
            // (A ray whose transmittance falls below the threshold exactly BEHIND the round's last sample is found in the
            // next round, at its first sample: kept = kept_base' + 0, cut.)
            const bool more = !dropped && !cut && a.t_next && a.t_next[r] >= 0.f;
            a.resume_out[r] = more ? a.t_next[r] : -1.f;
            if (a.carry_out) a.carry_out[r] = carry;
        }
    }
}

// optical thickness of a density-grid cell sample: exp(pre) * sqrt(3)/1024 * 2^level
__global__ void __launch_bounds__(256)
k_ngp_thickness(uint32_t n, const _Float16* __restrict__ density_out, uint32_t stride, int level,
                float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = __expf((float)density_out[(size_t)i * stride]) * scalbnf(1.7320508075688772f / 1024.0f, level);
}

// the same for scattered refresh samples (nvo_occ_sample_cells): the cell keeps the LARGEST thickness any of its samples
// saw [UPSTREAM splat_grid_samples_nerf_max_nearest_neighbor] -- non-negative floats order like their bit patterns
__global__ void __launch_bounds__(256)
k_ngp_thickness_splat(uint32_t n, const _Float16* __restrict__ density_out, uint32_t stride,
                      const uint32_t* __restrict__ cell_idx, float* __restrict__ fresh) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t cell = cell_idx[i];
    const float v = __expf((float)density_out[(size_t)i * stride]) * scalbnf(1.7320508075688772f / 1024.0f, (int)(cell >> 21));
    if (!(v >= 0.0f)) return;  // (NaN)
    atomicMax(reinterpret_cast<unsigned int*>(fresh) + cell, __float_as_uint(v));
}

__global__ void __launch_bounds__(256)
k_fill_i32(uint32_t n, int32_t* __restrict__ p, int32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

extern "C" {

int nvo_ngp_positions(nvo_stream_t stream, uint32_t capacity, const int32_t* ray_idx, const float* t,
                      const float* origins, const float* directions, float aabb_lo, float aabb_hi, float* x01) {
    return nvo_ngp_positions_live(stream, capacity, ray_idx, t, origins, directions, aabb_lo, aabb_hi, x01, nullptr);
}

int nvo_ngp_positions_live(nvo_stream_t stream, uint32_t capacity, const int32_t* ray_idx, const float* t,
                           const float* origins, const float* directions, float aabb_lo, float aabb_hi, float* x01,
                           const uint32_t* n_live) {
    NVO_REQUIRE(capacity == 0 || (ray_idx && t && origins && directions && x01), "ngp_positions: NULL argument");
    NVO_REQUIRE(aabb_hi > aabb_lo, "ngp_positions: empty aabb");
    if (capacity == 0) return NVO_OK;
    NVO_PROF(stream, "ngp_positions");
    NVO_LAUNCH(k_ngp_positions, dim3(nvo_div_up(capacity, 256)), dim3(256), 0, (hipStream_t)stream, capacity, ray_idx,
               t, origins, directions, aabb_lo, 1.0f / (aabb_hi - aabb_lo), x01, n_live);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_ngp_count_alive(nvo_stream_t stream, const nvo_ngp_alive_args* args) {
    NVO_REQUIRE(args != nullptr, "ngp_count_alive: args is NULL");
    const nvo_ngp_alive_args a = *args;
    NVO_REQUIRE(a.R == 0 || (a.counts && a.offsets && a.dt && a.density_out && a.kept && a.state && a.density_stride >= 1),
                "ngp_count_alive: NULL argument");
    NVO_REQUIRE(a.min_transmittance >= 0.f && a.min_transmittance < 1.f, "ngp_count_alive: threshold outside [0, 1)");
    if (a.R == 0) return NVO_OK;
    NVO_PROF(stream, "ngp_count_alive");
    NVO_LAUNCH(k_ngp_count_alive, dim3(nvo_div_up(a.R, 4)), dim3(256), 0, (hipStream_t)stream, a);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_ngp_positions_bwd(nvo_stream_t stream, uint32_t R, uint32_t capacity, const int32_t* counts,
                          const int32_t* offsets, const float* t, const float* origins, const float* directions,
                          float aabb_lo, float aabb_hi, const float* dx01, float* d_origin, float* d_dir) {
    return nvo_ngp_positions_bwd_dev(stream, R, capacity, counts, offsets, t, origins, directions, aabb_lo, aabb_hi, dx01, d_origin,
                                     d_dir, nullptr);
}

int nvo_ngp_positions_bwd_dev(nvo_stream_t stream, uint32_t R, uint32_t capacity, const int32_t* counts,
                              const int32_t* offsets, const float* t, const float* origins, const float* directions,
                              float aabb_lo, float aabb_hi, const float* dx01, float* d_origin, float* d_dir,
                              const uint32_t* R_dev) {
    NVO_REQUIRE(R == 0 || (counts && offsets && t && origins && directions && dx01 && d_origin && d_dir),
                "ngp_positions_bwd: NULL argument");
    NVO_REQUIRE(aabb_hi > aabb_lo, "ngp_positions_bwd: empty box");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "ngp_positions_bwd");
    NVO_LAUNCH(k_ngp_positions_bwd, dim3(nvo_div_up(R, 4)), dim3(256), 0, (hipStream_t)stream, R, capacity, counts,
               offsets, t, origins, directions, aabb_lo, 1.0f / (aabb_hi - aabb_lo), dx01, d_origin, d_dir, R_dev);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_ngp_composite_loss(nvo_stream_t stream, const nvo_ngp_loss_args* args) {
    NVO_REQUIRE(args != nullptr, "ngp_composite_loss: args is NULL");
    const nvo_ngp_loss_args a = *args;
    NVO_REQUIRE(a.counts && a.offsets && a.t && a.dt && a.density_out && a.rgb_out, "ngp_composite_loss: NULL input");
    NVO_REQUIRE(!a.d_rgb_out || (a.d_density_pre && a.gt_rgb && a.losses && a.ray_idx && a.d_rgb_stride >= 3),
                "ngp_composite_loss: training mode needs d_density_pre, gt_rgb, losses, ray_idx");
    if (a.R == 0) return NVO_OK;
    hipStream_t s = (hipStream_t)stream;
    NVO_PROF(stream, "ngp_composite_loss");
    const uint32_t ray_blocks = nvo_div_up(a.R, 4);
    const uint32_t clear_blocks = a.d_rgb_out ? nvo_div_up(a.capacity, 256) : 0u;
    NVO_LAUNCH(k_ngp_composite_loss, dim3(ray_blocks + clear_blocks), dim3(256), 0, s, a, ray_blocks);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_ngp_thickness_splat(nvo_stream_t stream, uint32_t n, const void* density_out, uint32_t stride,
                            const uint32_t* cell_idx, float* fresh) {
    NVO_REQUIRE(n == 0 || (density_out && cell_idx && fresh && stride >= 1), "ngp_thickness_splat: bad argument");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "ngp_thickness_splat");
    NVO_LAUNCH(k_ngp_thickness_splat, dim3(nvo_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
               (const _Float16*)density_out, stride, cell_idx, fresh);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_ngp_thickness(nvo_stream_t stream, uint32_t n, const void* density_out, uint32_t stride, int level,
                      float* out) {
    NVO_REQUIRE(n == 0 || (density_out && out && stride >= 1), "ngp_thickness: bad argument");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "ngp_thickness");
    NVO_LAUNCH(k_ngp_thickness, dim3(nvo_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
               (const _Float16*)density_out, stride, level, out);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_fill_i32(nvo_stream_t stream, uint32_t n, int32_t* ptr, int32_t value) {
    NVO_REQUIRE(n == 0 || ptr, "fill_i32: NULL argument");
    if (n == 0) return NVO_OK;
    NVO_LAUNCH(k_fill_i32, dim3(nvo_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, ptr, value);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

}  // extern "C"
