// Per-ray kernels of the mapping training step for gfx950, ONE WAVE (64 lanes) PER RAY:
//   k_weights_pdf      density activation (trunc_exp) + volume-rendering weights + (optionally)
//                      histogram-CDF resampling of the next level's bins (PDFSampler)
//   k_main_render_loss compositing (rgb with last-sample background, accumulation, median and
//                      expected depth) fused with the rgb / distortion / depth losses and their
//                      gradients w.r.t. per-sample colour and density pre-activation
//   k_prop_loss        interlevel (mip-NeRF 360 outer-measure) + depth loss of one proposal level and
//                      the gradient w.r.t. that level's density pre-activation
// Prefix / suffix sums over a ray's samples are wavefront scans (shuffle network) with a carry
// across 64-sample chunks; per-ray scratch lives in the wave's private LDS slice.
// Replaces nerfstudio's torch-op chains RaySamples.get_weights, PDFSampler, RGBRenderer /
// AccumulationRenderer / DepthRenderer, interlevel_loss / distortion_loss / ds_nerf_depth_loss
// (reference hooks /root/reference/nerf_vo/mapping/
// nerfstudio_utils.py:337-350 and nerfstudio.py:71-82 for the loss multipliers).
// CPU restatement: oracle/rays.py.
#include "nvo_kernels.h"
#include "../../include/nerfvo_hip.h"

#include <float.h>

namespace {

constexpr int kRayBlock = 256;          // 4 waves = 4 rays per workgroup
constexpr int kRaysPerBlock = kRayBlock / 64;
constexpr int kMaxS = 256;              // samples per ray handled by the LDS scratch
constexpr float kLossEps = 1.0e-7f;     // nerfstudio losses.EPS

// (DPP forms, nvo_common.h: the shuffle forms were six dependent LDS round trips each)
__device__ __forceinline__ float wave_incl_scan(float v, int) { return nvo_wave_incl_scan(v); }
__device__ __forceinline__ float wave_sum(float v) { return nvo_wave_sum(v); }

__device__ __forceinline__ float nan_to_num(float v) {
    if (isnan(v)) return 0.f;
    if (isinf(v)) return v > 0.f ? FLT_MAX : -FLT_MAX;
    return v;
}

__device__ __forceinline__ float spacing_fn(float x) { return x < 1.f ? x * 0.5f : 1.f - 1.f / (2.f * x); }
__device__ __forceinline__ float spacing_fn_inv(float x) { return x < 0.5f ? 2.f * x : 1.f / (2.f - 2.f * x); }

// first index in a[0..n) with a[idx] > v   (torch.searchsorted(..., side="right"))
__device__ __forceinline__ int upper_bound(const float* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] > v) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Per-ray forward quantities from the density pre-activation; fills LDS arrays w[] and Tr[].
// sigma = selector * exp(pre + bias); w_i = (1 - exp(-delta_i sigma_i)) * exp(-sum_{j<i} delta_j sigma_j)
__device__ __forceinline__ void ray_weights(int lane, uint32_t S, const nvo_h16* __restrict__ pre, bool bf,
                                            uint32_t pre_stride, const float* __restrict__ x01,
                                            const float* __restrict__ tb, float bias,
                                            float* __restrict__ sigma_out, float* w, float* Tr) {
    float carry = 0.f;
    // One wave walks the ray in chunks of 64 samples with a carry between them: the NEXT chunk's inputs are requested
    // (unconditionally, at a clamped index: a load behind a divergent branch makes the join wait for everything in
    // flight) before the current chunk is scanned, so a 256-sample ray pays one memory round trip instead of four.
    struct In { float x0, pr, t0, t1; };
    auto fetch = [&](uint32_t base) {
        const uint32_t i = min(base + (uint32_t)lane, S - 1u);
        In v;
        v.x0 = x01[3 * (size_t)i];
        v.pr = nvo_ld16(pre + (size_t)i * pre_stride, bf);
        v.t0 = tb[i];
        v.t1 = tb[i + 1];
        return v;
    };
    In cur = fetch(0);
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t i = base + lane;
        In nxt = cur;
        if (base + 64 < S) nxt = fetch(base + 64);  // (wave-uniform; single-chunk rays request nothing twice)
        float dd = 0.f, sg = 0.f;
        if (i < S) {
            const bool sel = cur.x0 > 0.f;
            // (exponent capped at 60: sigma <= 1.1e26 -- alpha is 1 to the last bit long before that -- so that a run whose
            // density pre-activations have blown up cannot put inf - inf = NaN into the transmittance and, through it, a
            // NaN that no loss scale can back off from into every gradient of the step; trunc_exp's forward is exp(x))
            sg = sel ? __expf(fminf(cur.pr + bias, 60.f)) : 0.f;
            dd = (cur.t1 - cur.t0) * sg;
            if (sigma_out) sigma_out[i] = sg;
        }
        cur = nxt;
        const float incl = wave_incl_scan(dd, lane) + carry;
        const float T = __expf(-(incl - dd));
        if (i < S) {
            w[i] = nan_to_num((1.f - __expf(-dd)) * T);
            Tr[i] = T;
        }
        carry = nvo_wave_bcast(incl, 63);
    }
}

// dL/dw (in LDS array g[], overwritten) -> dL/dpre, written as fp16 * loss_scale
// dL/dsigma_k = delta_k [ g_k (T_k - w_k) - sum_{i>k} g_i w_i ],  dsigma/dpre = exp(clamp(pre+bias,-15,15))
// returns (per lane): a gradient of this ray does not survive the 16-bit format it is stored in
__device__ __forceinline__ bool ray_weights_bwd(int lane, uint32_t S, const nvo_h16* __restrict__ pre, bool bf,
                                                uint32_t pre_stride, const float* __restrict__ x01,
                                                const float* __restrict__ tb, float bias,
                                                const float* w, const float* Tr, const float* g,
                                                float loss_scale, nvo_h16* __restrict__ dpre,
                                                uint32_t dpre_stride, bool zero_row = false,
                                                unsigned long long* live = nullptr) {
    // (live, S <= 64: lanes whose STORED 16-bit gradient is not zero)
    bool overflow = false;
    const float fmt_max = bf ? 3.0e38f : 65504.0f;
    // total of g_i w_i, then inclusive prefix per chunk -> suffix (exclusive) = total - incl
    float total = 0.f;
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t i = base + lane;
        total += (i < S) ? g[i] * w[i] : 0.f;
    }
    total = wave_sum(total);
    float carry = 0.f;
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t i = base + lane;
        const float gw = (i < S) ? g[i] * w[i] : 0.f;
        const float incl = wave_incl_scan(gw, lane) + carry;
        if (i < S) {
            const bool sel = x01[3 * (size_t)i] > 0.f;
            float d = 0.f;
            if (sel) {
                const float x = nvo_ld16(pre + (size_t)i * pre_stride, bf) + bias;
                const float delta = tb[i + 1] - tb[i];
                const float dsig = delta * (g[i] * (Tr[i] - w[i]) - (total - incl));
                d = dsig * __expf(fminf(fmaxf(x, -15.f), 15.f));
            }
            overflow = overflow || !(fabsf(d * loss_scale) <= fmt_max);
            const nvo_h16 d16 = nvo_cvt16(d * loss_scale, bf);
            if (live) *live = __ballot((d16 & 0x7fffu) != 0u);
            if (zero_row && dpre_stride == 16) {
                // whole 32-byte row {d, 0 x 15} as two 16-byte stores (the MLP backward reads all 16 columns)
                uint4 lo = make_uint4(0u, 0u, 0u, 0u);
                lo.x = (uint32_t)d16;
                uint4* row = reinterpret_cast<uint4*>(dpre + (size_t)i * 16);
                row[0] = lo;
                row[1] = make_uint4(0u, 0u, 0u, 0u);
            } else {
                dpre[(size_t)i * dpre_stride] = d16;
            }
        }
        carry = nvo_wave_bcast(incl, 63);
    }
    return overflow;
}

// ------------------------------------------------------------------------------------------------
// weights (+ PDF resampling)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kRayBlock)
k_weights_pdf(nvo_weights_pdf_args a) {
    __shared__ float lds[kRaysPerBlock][3][kMaxS + 4];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * kRaysPerBlock + wib;
    if (r >= a.R) return;
    float* w = lds[wib][0];
    float* Tr = lds[wib][1];
    float* cdf = lds[wib][2];
    const uint32_t S = a.S;
    const size_t so = (size_t)r * S;
    const float* tb = a.tbins + (size_t)r * (S + 1);
    ray_weights(lane, S, (const nvo_h16*)a.pre + so * a.pre_stride, a.act_bf16 != 0, a.pre_stride, a.x01 + 3 * so, tb, a.density_bias,
                a.sigma ? a.sigma + so : nullptr, w, Tr);
    for (uint32_t i = lane; i < S; i += 64) a.weights[so + i] = w[i];
    if (a.S_out == 0) return;

    // ---- PDFSampler: annealed weights -> padded pdf -> clamped cdf (cdf[0] = 0)
    const float anneal = a.anneal_dev ? *a.anneal_dev : a.anneal;
    float part = 0.f;
    for (uint32_t i = lane; i < S; i += 64) {
        const float wa = (anneal == 1.0f ? w[i] : powf(w[i], anneal)) + a.histogram_padding;
        Tr[i] = wa;  // reuse as pdf numerator
        part += wa;
    }
    float sum = wave_sum(part);
    const float padding = fmaxf(1e-5f - sum, 0.f);
    sum += padding;
    const float pad_each = padding / (float)S;
    float carry = 0.f;
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t i = base + lane;
        const float p = (i < S) ? (Tr[i] + pad_each) / sum : 0.f;
        const float incl = wave_incl_scan(p, lane) + carry;
        if (i < S) cdf[i + 1] = fminf(1.f, incl);
        carry = nvo_wave_bcast(incl, 63);
    }
    if (lane == 0) cdf[0] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const float* sb = a.sbins + (size_t)r * (S + 1);
    const uint32_t nb = a.S_out + 1;
    const float inv_nb = 1.0f / (float)nb;
    const float shift = a.jitter ? a.jitter[r] * inv_nb : 0.5f * inv_nb;
    const float s_near = spacing_fn(a.near_plane), s_far = spacing_fn(a.far_plane);
    for (uint32_t j = lane; j < nb; j += 64) {
        const float u = (float)j * inv_nb + shift;
        const int inds = upper_bound(cdf, (int)S + 1, u);
        const int below = min(max(inds - 1, 0), (int)S), above = min(max(inds, 0), (int)S);
        const float c0 = cdf[below], c1 = cdf[above];
        float t = (u - c0) / (c1 - c0);
        if (isnan(t)) t = 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);
        const float b = sb[below] + t * (sb[above] - sb[below]);
        const float tv = spacing_fn_inv(b * s_far + (1.f - b) * s_near);
        a.sbins_out[(size_t)r * nb + j] = b;
        a.tbins_out[(size_t)r * nb + j] = tv;
        if (a.x01_out) w[j] = tv;  // the weights are in global memory already: reuse their LDS row
    }
    if (a.x01_out) {  // fused nvo_sample_positions of the resampled level
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float o[3] = {a.origins[3 * (size_t)r], a.origins[3 * (size_t)r + 1], a.origins[3 * (size_t)r + 2]};
        const float d[3] = {a.directions[3 * (size_t)r], a.directions[3 * (size_t)r + 1], a.directions[3 * (size_t)r + 2]};
        for (uint32_t j = lane; j < a.S_out; j += 64) {
            float p[3];
            nvo_contract_position01(o, d, (w[j] + w[j + 1]) * 0.5f, p);
            float* xo = a.x01_out + 3 * ((size_t)r * a.S_out + j);
            xo[0] = p[0]; xo[1] = p[1]; xo[2] = p[2];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// main level: render + losses + gradients
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kRayBlock)
k_main_render_loss(nvo_main_loss_args a) {
    __shared__ float lds[kRaysPerBlock][4][64 + 4];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x * kRaysPerBlock + wib;
    if (r >= a.R) return;
    if (a.loss_scale_dev) a.loss_scale = *a.loss_scale_dev;  // dynamic loss scale (GradScaler state on the device)
    float* w = lds[wib][0];
    float* Tr = lds[wib][1];
    float* g = lds[wib][2];
    float* ut = lds[wib][3];
    const uint32_t S = a.S;
    const size_t so = (size_t)r * S;
    const float* tb = a.tbins + (size_t)r * (S + 1);
    const float* sb = a.sbins + (size_t)r * (S + 1);
    const nvo_h16* pre = (const nvo_h16*)a.pre + so * a.pre_stride;
    const bool bf = a.act_bf16 != 0;
    const float* x01 = a.x01 + 3 * so;
    ray_weights(lane, S, pre, bf, a.pre_stride, x01, tb, a.density_bias, nullptr, w, Tr);
    const bool act = (uint32_t)lane < S;
    const float wi = act ? w[lane] : 0.f;
    if (a.weights && act) a.weights[so + lane] = wi;

    // ---- composite
    float c[3] = {0.f, 0.f, 0.f};
    if (act) {
        const nvo_h16* cp = (const nvo_h16*)a.rgb + (so + lane) * a.rgb_stride;
        c[0] = nvo_ld16(cp, bf); c[1] = nvo_ld16(cp + 1, bf); c[2] = nvo_ld16(cp + 2, bf);
    }
    const float acc = wave_sum(wi);
    float pix[3], clast[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        clast[k] = nvo_wave_bcast(c[k], (int)S - 1);
        pix[k] = wave_sum(wi * c[k]) + clast[k] * (1.f - acc);
    }
    const float mid = act ? 0.5f * (tb[lane] + tb[lane + 1]) : 0.f;
    // median depth: first sample whose cumulative weight reaches 0.5 (clamped to the last sample)
    const float cum = wave_incl_scan(wi, lane);
    const unsigned long long ballot = __ballot(act && cum >= 0.5f);
    const int med = ballot ? (int)__builtin_ctzll(ballot) : (int)S - 1;
    const float depth_med = nvo_wave_bcast(mid, med);
    if (lane == 0) {
        a.out_rgb[3 * (size_t)r + 0] = pix[0];
        a.out_rgb[3 * (size_t)r + 1] = pix[1];
        a.out_rgb[3 * (size_t)r + 2] = pix[2];
        a.out_depth[r] = depth_med;
        a.out_accumulation[r] = acc;
    }
    if (a.out_expected_depth) {
        const float num = wave_sum(wi * mid);
        if (lane == 0) a.out_expected_depth[r] = num / (acc + 1e-10f);  // clip to [min,max] steps: host
    }
    // ---- analytic normals: n_i = -normalize(d pre_i / d x01_i); rendered N = safe_normalize(sum w_i n_i);
    //      the "normals" output is NormalsShader(N) = (N + 1) / 2
    float nrm[3] = {0.f, 0.f, 0.f};
    float Nv[3] = {0.f, 0.f, 0.f}, Nhat[3] = {0.f, 0.f, 0.f};
    float rN = 0.f;
    const bool has_normals = a.dsigma_dx != nullptr;
    if (has_normals) {
        if (act) {
            const float* gp = a.dsigma_dx + 3 * (so + lane);
            const float gx = gp[0] * a.dsigma_inv_scale, gy = gp[1] * a.dsigma_inv_scale,
                        gz = gp[2] * a.dsigma_inv_scale;
            const float inv = -1.f / fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);  // F.normalize eps
            nrm[0] = gx * inv; nrm[1] = gy * inv; nrm[2] = gz * inv;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) Nv[k] = wave_sum(wi * nrm[k]);
        rN = sqrtf(Nv[0] * Nv[0] + Nv[1] * Nv[1] + Nv[2] * Nv[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) Nhat[k] = Nv[k] / (rN + 1e-10f);  // safe_normalize
        if (a.out_normals && lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) a.out_normals[3 * (size_t)r + k] = 0.5f * (Nhat[k] + 1.f);
        }
    }
    if (!a.dpre) return;

    // ---- losses
    float dpix[3];
    float l_rgb = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float e = pix[k] - a.gt_rgb[3 * (size_t)r + k];
        l_rgb += e * e;
        dpix[k] = a.rgb_mult * 2.f * e * a.inv_rays * (1.f / 3.f);
    }
    l_rgb *= a.inv_rays * (1.f / 3.f);
    // d/dw_i and d/dc_i from the rgb term
    float gw = 0.f;
    float dc[3] = {0.f, 0.f, 0.f};
    if (act) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gw += dpix[k] * (c[k] - clast[k]);
            dc[k] = dpix[k] * wi;
        }
        if ((uint32_t)lane == S - 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dc[k] += dpix[k] * (1.f - acc);
        }
    }
    // distortion (spacing domain): sum_i w_i sum_j w_j |ut_i-ut_j| + sum_i w_i^2 (s_{i+1}-s_i)/3
    float l_dist = 0.f;
    if (a.distortion_mult != 0.f) {
        const float uti = act ? 0.5f * (sb[lane] + sb[lane + 1]) : 0.f;
        if (act) ut[lane] = uti;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float inner = 0.f;
        if (act) {
            for (uint32_t j = 0; j < S; ++j) inner += w[j] * fabsf(uti - ut[j]);
        }
        const float ds = act ? (sb[lane + 1] - sb[lane]) : 0.f;
        l_dist = wave_sum(wi * inner + wi * wi * ds * (1.f / 3.f)) * a.inv_rays;
        if (act) gw += a.distortion_mult * a.inv_rays * (2.f * inner + 2.f * wi * ds * (1.f / 3.f));
    }
    // DS-NeRF depth loss on this level
    float l_depth = 0.f;
    if (a.depth_mult != 0.f && a.gt_depth) {
        const float z = a.gt_depth[r] * a.directions_norm[r];
        const float mask = z > 0.f ? 1.f : 0.f;
        float term = 0.f;
        if (act) {
            const float len = tb[lane + 1] - tb[lane];
            const float gss = __expf(-((mid - z) * (mid - z)) / (2.f * a.depth_sigma)) * len * mask;
            term = -__logf(wi + kLossEps) * gss;
            gw += a.depth_mult * a.depth_level_div * a.inv_rays * (-gss / (wi + kLossEps));
        }
        l_depth = wave_sum(term) * a.inv_rays * a.depth_level_div;
    }
    // monosdf normal loss on the shaded normals: L1 + (1 - cos) between the L2-normalised vectors
    float l_normal = 0.f;
    if (has_normals && a.gt_normal && a.normal_mult != 0.f) {
        float sh[3], q[3], p[3], dp[3];
        float rs = 0.f, rq = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sh[k] = 0.5f * (Nhat[k] + 1.f);
            q[k] = a.gt_normal[3 * (size_t)r + k];
            rs += sh[k] * sh[k];
            rq += q[k] * q[k];
        }
        rs = fmaxf(sqrtf(rs), 1e-12f);
        rq = fmaxf(sqrtf(rq), 1e-12f);
        float pdq = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = sh[k] / rs;
            q[k] /= rq;
            l_normal += fabsf(p[k] - q[k]);
            pdq += p[k] * q[k];
        }
        l_normal = (l_normal + 1.f - pdq) * a.inv_rays;
        // backward: p = sh / |sh|, sh = (Nhat + 1) / 2, Nhat = Nv / (|Nv| + 1e-10)
        float pdp = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float e = p[k] - q[k];
            dp[k] = a.normal_mult * a.inv_rays * ((e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) - q[k]);
            pdp += p[k] * dp[k];
        }
        float dN[3], NdN = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dN[k] = 0.5f * (dp[k] - p[k] * pdp) / rs;
            NdN += Nv[k] * dN[k];
        }
        const float den = rN + 1e-10f;
        const float radial = rN > 0.f ? NdN / (rN * den * den) : 0.f;
        if (act) {
#pragma unroll
            for (int k = 0; k < 3; ++k) gw += (dN[k] / den - Nv[k] * radial) * nrm[k];
        }
    }
    if (lane == 0) {
        // 64 shards of 8 floats: thousands of adds to ONE word serialise at the memory side
        // (~88 same-address atomics per us on MI355X); the host sums the shards.
        float* shard = a.losses + 8 * (r & 63u);
        atomicAdd(shard + 0, a.rgb_mult * l_rgb);
        atomicAdd(shard + 1, a.distortion_mult * l_dist);
        atomicAdd(shard + 2, a.depth_mult * l_depth);
        if (l_normal != 0.f) atomicAdd(shard + 6, a.normal_mult * l_normal);
    }
    if (act) g[lane] = gw;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned long long pre_live = 0ull;
    bool overflow = ray_weights_bwd(lane, S, pre, bf, a.pre_stride, x01, tb, a.density_bias, w, Tr, g, a.loss_scale,
                                    (nvo_h16*)a.dpre + so * a.dpre_stride, a.dpre_stride, false, &pre_live);
    if (act) {
        const float fmt_max = bf ? 3.0e38f : 65504.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) overflow = overflow || !(fabsf(dc[k] * a.loss_scale) <= fmt_max);
    }
    if (a.nonfinite_flag && __ballot(overflow) != 0ull && lane == 0) atomicOr(a.nonfinite_flag, 1u);
    bool rgb_nz = false;
    if (act && a.drgb_stride == 16) {
        // the colour MLP backward reads all 16 columns: one 32-byte row {dr, dg, db, 0 x 13} as two 16-byte stores
        uint4 lo = make_uint4(0u, 0u, 0u, 0u);
        lo.x = nvo_cvt16x2(dc[0] * a.loss_scale, dc[1] * a.loss_scale, bf);
        lo.y = (uint32_t)nvo_cvt16(dc[2] * a.loss_scale, bf);
        uint4* row = reinterpret_cast<uint4*>((nvo_h16*)a.drgb + (so + lane) * 16);
        row[0] = lo;
        row[1] = make_uint4(0u, 0u, 0u, 0u);
        rgb_nz = ((lo.x | lo.y) & 0x7fff7fffu) != 0u;
    } else if (act) {
        nvo_h16* dp = (nvo_h16*)a.drgb + (so + lane) * a.drgb_stride;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dp[k] = nvo_cvt16(dc[k] * a.loss_scale, bf);
            rgb_nz = rgb_nz || (dp[k] & 0x7fffu) != 0u;
        }
        for (uint32_t k = 3; k < a.drgb_stride; ++k) dp[k] = (nvo_h16)0;
    }
    if (a.tile_live) {  // (the launcher: S % 16 == 0, S <= 64) what the stored 16-bit gradients of each 16-sample tile hold
        const unsigned long long rgb_live = __ballot(rgb_nz);
        uint32_t byte = 0u;
        if (lane < (int)(S >> 4)) {
            const uint32_t sh = 16u * (uint32_t)lane;
            byte = (((rgb_live >> sh) & 0xffffull) != 0ull ? 1u : 0u) | (((pre_live >> sh) & 0xffffull) != 0ull ? 2u : 0u);
            a.tile_live[(so >> 4) + lane] = (uint8_t)byte;
        }
        // slot 7 of the loss shards: the number of tiles that carry any gradient (exact in fp32 up to 2^24) -- what tells the
        // hash grid's backward whether listing the live samples pays (k_live_rows)
        const uint32_t n_live_tiles = (uint32_t)__popcll(__ballot(byte != 0u));
        if (lane == 0 && n_live_tiles) atomicAdd(a.losses + 8 * (r & 63u) + 7, (float)n_live_tiles);
    }
}

// ------------------------------------------------------------------------------------------------
// proposal level: interlevel + depth loss and gradient
// ------------------------------------------------------------------------------------------------
// Workgroups [0, blocks0) serve a0, the rest a1: both proposal levels of a step in ONE launch (nvo_prop_loss_pair) --
// each level alone is a single round of 4096 waves that lasts as long as one ray's dependent chain (17-22 us), so two
// launches cost two chains.
__global__ void __launch_bounds__(kRayBlock)
k_prop_loss(nvo_prop_loss_args a0, nvo_prop_loss_args a1, uint32_t blocks0) {
    __shared__ float lds[kRaysPerBlock][5][kMaxS + 4];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const bool second = blockIdx.x >= blocks0;  // (uniform)
    nvo_prop_loss_args a = second ? a1 : a0;
    const uint32_t r = (second ? blockIdx.x - blocks0 : blockIdx.x) * kRaysPerBlock + wib;
    if (r >= a.R) return;
    if (a.loss_scale_dev) a.loss_scale = *a.loss_scale_dev;
    float* w = lds[wib][0];
    float* Tr = lds[wib][1];
    float* g = lds[wib][2];     // dL/dw of this level (built through a difference array)
    float* cy = lds[wib][3];    // exclusive cumsum of w: cy[0] = 0, cy[j+1] = sum_{<=j} w
    const uint32_t S = a.S, Sm = a.S_main;
    const size_t so = (size_t)r * S;
    const float* tb = a.tbins + (size_t)r * (S + 1);
    // the two binary searches per main interval walk this level's spacing bins: staged in LDS, each probe
    // is an LDS read instead of a dependent global load
    float* sb = lds[wib][4];
    {
        const float* sbg = a.sbins + (size_t)r * (S + 1);
        for (uint32_t i = lane; i < S + 1; i += 64) sb[i] = sbg[i];
    }
    const nvo_h16* pre = (const nvo_h16*)a.pre + so * a.pre_stride;
    const bool bf = a.act_bf16 != 0;
    const float* x01 = a.x01 + 3 * so;
    ray_weights(lane, S, pre, bf, a.pre_stride, x01, tb, a.density_bias, nullptr, w, Tr);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float carry = 0.f;
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t i = base + lane;
        const float v = (i < S) ? w[i] : 0.f;
        const float incl = wave_incl_scan(v, lane) + carry;
        if (i < S) cy[i + 1] = incl;
        carry = nvo_wave_bcast(incl, 63);
    }
    if (lane == 0) cy[0] = 0.f;
    for (uint32_t i = lane; i < S + 2; i += 64) g[i] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- interlevel: one main-level interval per lane
    float l_inter = 0.f;
    if ((uint32_t)lane < Sm) {
        const float* cm = a.sbins_main + (size_t)r * (Sm + 1);
        const float wm = a.weights_main[(size_t)r * Sm + lane];
        int lo = upper_bound(sb, (int)S, cm[lane]) - 1;          // over interval starts sb[0..S)
        lo = min(max(lo, 0), (int)S - 1);
        int hi = upper_bound(sb + 1, (int)S, cm[lane + 1]);      // over interval ends sb[1..S]
        hi = min(max(hi, 0), (int)S - 1);
        const float w_outer = cy[hi + 1] - cy[lo];
        const float diff = wm - w_outer;
        if (diff > 0.f) {
            l_inter = diff * diff / (wm + kLossEps);
            const float coef = a.interlevel_mult * a.inv_rays / (float)Sm;
            const float gi = -2.f * diff / (wm + kLossEps) * coef;  // d/d(w_outer)
            atomicAdd(&g[lo], gi);
            atomicAdd(&g[hi + 1], -gi);
        }
    }
    l_inter = wave_sum(l_inter) * a.inv_rays / (float)Sm;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // difference array -> dL/dw_j (inclusive prefix), plus the depth term
    float l_depth = 0.f;
    const bool use_depth = a.depth_mult != 0.f && a.gt_depth;
    const float z = use_depth ? a.gt_depth[r] * a.directions_norm[r] : 0.f;
    const float mask = z > 0.f ? 1.f : 0.f;
    carry = 0.f;
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t i = base + lane;
        const float v = (i < S) ? g[i] : 0.f;
        const float incl = wave_incl_scan(v, lane) + carry;
        carry = nvo_wave_bcast(incl, 63);
        float gi = incl;
        if (i < S && use_depth) {
            const float midp = 0.5f * (tb[i] + tb[i + 1]);
            const float len = tb[i + 1] - tb[i];
            const float gss = __expf(-((midp - z) * (midp - z)) / (2.f * a.depth_sigma)) * len * mask;
            l_depth += -__logf(w[i] + kLossEps) * gss;
            gi += a.depth_mult * a.depth_level_div * a.inv_rays * (-gss / (w[i] + kLossEps));
        }
        // all lanes finished reading g[i] of this chunk through the scan before it is overwritten
        if (i < S) g[i] = gi;
    }
    l_depth = wave_sum(l_depth) * a.inv_rays * a.depth_level_div;
    if (lane == 0) {
        float* shard = a.losses + 8 * (r & 63u);
        atomicAdd(shard + 0, a.interlevel_mult * l_inter);
        atomicAdd(shard + 1, a.depth_mult * l_depth);
    }
    // value-only call (dpre == NULL): nerfstudio evaluates the interlevel / proposal-level depth terms on EVERY step,
    // also on those where the proposal networks do not train -- the loss values without the gradient pass
    if (a.dpre == nullptr) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool overflow = ray_weights_bwd(lane, S, pre, bf, a.pre_stride, x01, tb, a.density_bias, w, Tr, g, a.loss_scale,
                                          (nvo_h16*)a.dpre + so * a.dpre_stride, a.dpre_stride, true);
    if (a.nonfinite_flag && __ballot(overflow) != 0ull && lane == 0) atomicOr(a.nonfinite_flag, 1u);
    if (a.dpre_stride != 16) {
        for (uint32_t i = lane; i < S; i += 64) {
            nvo_h16* dp = (nvo_h16*)a.dpre + (so + i) * a.dpre_stride;
            for (uint32_t k = 1; k < a.dpre_stride; ++k) dp[k] = (nvo_h16)0;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// exported entry points (group C of include/nerfvo_hip.h)
// ---------------------------------------------------------------------------------------------
extern "C" {

int nvo_weights_pdf(nvo_stream_t stream, const nvo_weights_pdf_args* args) {
    NVO_REQUIRE(args != nullptr, "weights_pdf: args is NULL");
    const nvo_weights_pdf_args a = *args;
    NVO_REQUIRE(a.S >= 1 && a.S <= (uint32_t)kMaxS, "weights_pdf: samples per ray %u not in 1..%d", a.S, kMaxS);
    NVO_REQUIRE(a.S_out <= (uint32_t)kMaxS, "weights_pdf: S_out %u > %d", a.S_out, kMaxS);
    NVO_REQUIRE(a.S_out == 0 || (a.sbins_out && a.tbins_out), "weights_pdf: output bins are NULL");
    NVO_REQUIRE(a.pre && a.x01 && a.tbins && a.weights && a.sbins, "weights_pdf: NULL input");
    NVO_REQUIRE(!a.x01_out || (a.origins && a.directions && a.S_out > 0), "weights_pdf: x01_out needs origins, directions and S_out");
    if (a.R == 0) return NVO_OK;
    NVO_PROF(stream, "weights_pdf[S%u]", a.S);
    NVO_LAUNCH(k_weights_pdf, dim3(nvo_div_up(a.R, kRaysPerBlock)), dim3(kRayBlock), 0,
                       (hipStream_t)stream, a);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_main_render_loss(nvo_stream_t stream, const nvo_main_loss_args* args) {
    NVO_REQUIRE(args != nullptr, "main_render_loss: args is NULL");
    const nvo_main_loss_args a = *args;
    NVO_REQUIRE(a.S >= 1 && a.S <= 64, "main_render_loss: samples per ray %u not in 1..64", a.S);
    NVO_REQUIRE(a.pre && a.rgb && a.x01 && a.sbins && a.tbins && a.out_rgb && a.out_depth &&
                a.out_accumulation, "main_render_loss: NULL input/output");
    NVO_REQUIRE(!a.dpre || (a.drgb && a.losses && a.gt_rgb && a.drgb_stride >= 3),
                "main_render_loss: training mode needs drgb, losses, gt_rgb");
    NVO_REQUIRE(!a.tile_live || (a.dpre && (a.S & 15u) == 0u), "main_render_loss: tile_live needs training mode and S %% 16 == 0 (S = %u)", a.S);
    if (a.R == 0) return NVO_OK;
    NVO_PROF(stream, "main_render_loss");
    NVO_LAUNCH(k_main_render_loss, dim3(nvo_div_up(a.R, kRaysPerBlock)), dim3(kRayBlock), 0,
                       (hipStream_t)stream, a);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_prop_loss(nvo_stream_t stream, const nvo_prop_loss_args* args) {
    NVO_REQUIRE(args != nullptr, "prop_loss: args is NULL");
    const nvo_prop_loss_args a = *args;
    NVO_REQUIRE(a.S >= 1 && a.S <= (uint32_t)kMaxS && a.S_main >= 1 && a.S_main <= 64,
                "prop_loss: S=%u (<=%d) S_main=%u (<=64)", a.S, kMaxS, a.S_main);
    NVO_REQUIRE(a.pre && a.x01 && a.sbins && a.tbins && a.sbins_main && a.weights_main && a.losses &&
                (a.dpre == nullptr || a.dpre_stride >= 1), "prop_loss: NULL input/output");
    if (a.R == 0) return NVO_OK;
    NVO_PROF(stream, "prop_loss[S%u]", a.S);
    const uint32_t blocks = nvo_div_up(a.R, kRaysPerBlock);
    NVO_LAUNCH(k_prop_loss, dim3(blocks), dim3(kRayBlock), 0, (hipStream_t)stream, a, a, blocks);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_prop_loss_pair(nvo_stream_t stream, const nvo_prop_loss_args* args0, const nvo_prop_loss_args* args1) {
    NVO_REQUIRE(args0 != nullptr && args1 != nullptr, "prop_loss_pair: args is NULL");
    for (const nvo_prop_loss_args* p : {args0, args1}) {
        const nvo_prop_loss_args& a = *p;
        NVO_REQUIRE(a.S >= 1 && a.S <= (uint32_t)kMaxS && a.S_main >= 1 && a.S_main <= 64,
                    "prop_loss_pair: S=%u (<=%d) S_main=%u (<=64)", a.S, kMaxS, a.S_main);
        NVO_REQUIRE(a.pre && a.x01 && a.sbins && a.tbins && a.sbins_main && a.weights_main && a.losses &&
                    (a.dpre == nullptr || a.dpre_stride >= 1), "prop_loss_pair: NULL input/output");
    }
    const uint32_t b0 = nvo_div_up(args0->R, kRaysPerBlock), b1 = nvo_div_up(args1->R, kRaysPerBlock);
    if (b0 + b1 == 0) return NVO_OK;
    NVO_PROF(stream, "prop_loss[S%u+S%u]", args0->S, args1->S);
    NVO_LAUNCH(k_prop_loss, dim3(b0 + b1), dim3(kRayBlock), 0, (hipStream_t)stream, *args0, *args1, b0);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

}  // extern "C"
