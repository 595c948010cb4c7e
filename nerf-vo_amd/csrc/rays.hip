// Ray-side kernels of the mapping training step for gfx950: pixel gather, pinhole ray generation
// with SE3 pose correction, piecewise lin-disparity bins, sample positions with L-inf scene
// contraction.  Replaces the torch-op chains nerfstudio runs per iteration (PixelSampler gather,
// RayGenerator / Cameras.generate_rays, CameraOptimizer.apply_to_raybundle,
// UniformLinDispPiecewiseSampler, Frustums.get_positions, SceneContraction) -- SURVEY.md section
// 2.4 K8/K12, reference call sites /root/reference/nerf_vo/mapping/nerfstudio_utils.py:286-300.
// CPU restatement: oracle/rays.py.
#include "nvo_kernels.h"
#include <string.h>
#include "../../include/nerfvo_hip.h"

namespace {

// ---- pinhole rays -----------------------------------------------------------------------------
__device__ __forceinline__ void rot_apply(const float* __restrict__ m /*[3][4]*/, float x, float y,
                                          float z, float* o) {
    o[0] = m[0] * x + m[1] * y + m[2] * z;
    o[1] = m[4] * x + m[5] * y + m[6] * z;
    o[2] = m[8] * x + m[9] * y + m[10] * z;
}

__device__ __forceinline__ float norm3(const float* v) {
    return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}

// One pinhole ray: pixel centre (px, py) of camera `cam` -> origin, unit direction, |direction before
// normalisation|, pixel area.  m: the camera's [3][4] pose rows (row stride 4 floats); c: its pose correction or nullptr.
__device__ __forceinline__ void raygen_one(float px, float py, const float* __restrict__ intr4,
                                           const float* __restrict__ m, const float* __restrict__ c, float* o,
                                           float* d0, float* dnorm, float* area) {
    const float fx = intr4[0], fy = intr4[1], cx = intr4[2], cy = intr4[3];
    float dx[3], dy[3];
    rot_apply(m, (px - cx) / fx, -(py - cy) / fy, -1.f, d0);
    rot_apply(m, (px + 1.f - cx) / fx, -(py - cy) / fy, -1.f, dx);
    rot_apply(m, (px - cx) / fx, -(py + 1.f - cy) / fy, -1.f, dy);
    const float n0 = norm3(d0), nx = norm3(dx), ny = norm3(dy);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d0[k] /= n0;
        const float ex = d0[k] - dx[k] / nx, ey = d0[k] - dy[k] / ny;
        a += ex * ex;
        b += ey * ey;
    }
    o[0] = m[3];
    o[1] = m[7];
    o[2] = m[11];
    if (c) {  // CameraOptimizer.apply_to_raybundle: o += t, d = R d
        float d1[3];
        rot_apply(c, d0[0], d0[1], d0[2], d1);
        d0[0] = d1[0]; d0[1] = d1[1]; d0[2] = d1[2];
        o[0] += c[3]; o[1] += c[7]; o[2] += c[11];
    }
    *dnorm = n0;
    *area = sqrtf(a) * sqrtf(b);
}

// ray_indices: [R][3] int64 (camera, y, x).  corrections: [F][3][4] or nullptr.
__global__ void __launch_bounds__(256)
k_raygen(uint32_t R, const int64_t* __restrict__ ray_indices, const float* __restrict__ intrinsics,
         const float* __restrict__ c2w, const float* __restrict__ corrections,
         float* __restrict__ origins, float* __restrict__ directions,
         float* __restrict__ directions_norm, float* __restrict__ pixel_area,
         int32_t* __restrict__ cam_idx) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int64_t cam = ray_indices[3 * (size_t)r + 0];
    const float py = (float)ray_indices[3 * (size_t)r + 1] + 0.5f;
    const float px = (float)ray_indices[3 * (size_t)r + 2] + 0.5f;
    float o[3], d0[3], n0, area;
    raygen_one(px, py, intrinsics + 4 * cam, c2w + 12 * cam, corrections ? corrections + 12 * cam : nullptr, o, d0, &n0,
               &area);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        origins[3 * (size_t)r + k] = o[k];
        directions[3 * (size_t)r + k] = d0[k];
    }
    directions_norm[r] = n0;
    if (pixel_area) pixel_area[r] = area;
    cam_idx[r] = (int32_t)cam;
}

// k_raygen + k_gather_pixels (colour, depth) + the direction-encoding input (d + 1) / 2 + its SH(4) encoding for GIVEN
// pixel indices, one thread per ray: the occupancy-grid back-end draws its pixels on the host side of pyngp.Testbed and
// spent five ~6 us launches on what is a microsecond of work.  Same values as the separate kernels, bit for bit.
__global__ void __launch_bounds__(256)
k_rays_given(uint32_t R, const int64_t* __restrict__ ray_indices, const float* __restrict__ intrinsics,
             const float* __restrict__ c2w, const float* __restrict__ corrections, uint32_t H, uint32_t W,
             const float* __restrict__ images, const float* __restrict__ depths, float* __restrict__ origins,
             float* __restrict__ directions, float* __restrict__ directions_norm, float* __restrict__ pixel_area,
             int32_t* __restrict__ cam_idx, float* __restrict__ gt_rgb, float* __restrict__ gt_depth,
             float* __restrict__ dirs01, nvo_h16* __restrict__ sh, const float* __restrict__ depths_cov,
             float* __restrict__ gt_depth_cov, const uint32_t* __restrict__ R_dev) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (R_dev) R = min(R, *R_dev);
    if (r >= R) return;
    const int64_t cam = ray_indices[3 * (size_t)r + 0], y = ray_indices[3 * (size_t)r + 1], x = ray_indices[3 * (size_t)r + 2];
    float o[3], d[3], n0, area;
    raygen_one((float)x + 0.5f, (float)y + 0.5f, intrinsics + 4 * cam, c2w + 12 * cam, corrections ? corrections + 12 * cam : nullptr,
               o, d, &n0, &area);
    const size_t pix = ((size_t)cam * H + (size_t)y) * W + (size_t)x;
    float d01[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        origins[3 * (size_t)r + k] = o[k];
        directions[3 * (size_t)r + k] = d[k];
        gt_rgb[3 * (size_t)r + k] = images[3 * pix + k];
        d01[k] = (d[k] + 1.f) * 0.5f;
        dirs01[3 * (size_t)r + k] = d01[k];
    }
    directions_norm[r] = n0;
    if (pixel_area) pixel_area[r] = area;
    cam_idx[r] = (int32_t)cam;
    if (depths) gt_depth[r] = depths[pix];
    if (depths_cov) gt_depth_cov[r] = depths_cov[pix];
    float c[16];
    nvo_sh4_eval(d01[0] * 2.f - 1.f, d01[1] * 2.f - 1.f, d01[2] * 2.f - 1.f, 4u, c);
#pragma unroll
    for (int k = 0; k < 16; ++k) sh[16 * (size_t)r + k] = nvo_cvt16(c[k], false);
}

// images: [F][H][W][Cn] float -> out [R][Cn]
__global__ void __launch_bounds__(256)
k_gather_pixels(uint32_t R, const int64_t* __restrict__ ray_indices, uint32_t H, uint32_t W,
                uint32_t Cn, const float* __restrict__ images, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * Cn) return;
    const uint32_t r = i / Cn, c = i - r * Cn;
    const int64_t cam = ray_indices[3 * (size_t)r + 0], y = ray_indices[3 * (size_t)r + 1],
                  x = ray_indices[3 * (size_t)r + 2];
    out[i] = images[(((size_t)cam * H + (size_t)y) * W + (size_t)x) * Cn + c];
}

// ---- spacing functions (UniformLinDispPiecewiseSampler) ----------------------------------------
__device__ __forceinline__ float spacing_fn(float x) { return x < 1.f ? x * 0.5f : 1.f - 1.f / (2.f * x); }
__device__ __forceinline__ float spacing_fn_inv(float x) { return x < 0.5f ? 2.f * x : 1.f / (2.f - 2.f * x); }

// sbins/tbins: [R][S+1].  jitter: [R] in [0,1) (single_jitter) or nullptr (eval: plain linspace).
__global__ void __launch_bounds__(256)
k_sample_lindisp(uint32_t R, uint32_t S, float near, float far, const float* __restrict__ jitter,
                 float* __restrict__ sbins, float* __restrict__ tbins) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * (S + 1)) return;
    const uint32_t r = i / (S + 1), j = i - r * (S + 1);
    const float step = 1.0f / (float)S;
    float b = (float)j * step;
    if (j == S) b = 1.0f;
    if (jitter) {
        const float prev = j > 0 ? (float)(j - 1) * step : 0.f;
        const float next = (j + 1 == S) ? 1.0f : (float)(j + 1) * step;
        const float lower = j == 0 ? b : (prev + b) * 0.5f;
        const float upper = j == S ? b : (b + next) * 0.5f;
        b = lower + (upper - lower) * jitter[r];
    }
    const float s_near = spacing_fn(near), s_far = spacing_fn(far);
    sbins[i] = b;
    tbins[i] = spacing_fn_inv(b * s_far + (1.f - b) * s_near);
}

// ---- sample positions -> contracted, normalised grid coordinates -------------------------------
// x01: [R*S][3]; rows whose selector is false are written as exact zeros (the selector is then
// recoverable as x01[.][0] > 0).  contraction: 0 = none (aabb normalisation), 1 = L-inf.
__global__ void __launch_bounds__(256)
k_sample_positions(uint32_t R, uint32_t S, const float* __restrict__ origins,
                   const float* __restrict__ directions, const float* __restrict__ tbins,
                   float* __restrict__ x01) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * S) return;
    const uint32_t r = i / S, s = i - r * S;
    const float mid = (tbins[(size_t)r * (S + 1) + s] + tbins[(size_t)r * (S + 1) + s + 1]) * 0.5f;
    const float o[3] = {origins[3 * (size_t)r], origins[3 * (size_t)r + 1], origins[3 * (size_t)r + 2]};
    const float d[3] = {directions[3 * (size_t)r], directions[3 * (size_t)r + 1], directions[3 * (size_t)r + 2]};
    float p[3];
    nvo_contract_position01(o, d, mid, p);
#pragma unroll
    for (int k = 0; k < 3; ++k) x01[3 * (size_t)i + k] = p[k];
}

// Fused first sampler level: lin-disp bins AND the sample positions of those bins in one launch
// (thread per (ray, sample); the thread of the last sample also writes the closing bin edge).
__device__ __forceinline__ float lindisp_bin(uint32_t j, uint32_t S, const float* jitter, uint32_t r) {
    const float step = 1.0f / (float)S;
    float b = (float)j * step;
    if (j == S) b = 1.0f;
    if (jitter) {
        const float prev = j > 0 ? (float)(j - 1) * step : 0.f;
        const float next = (j + 1 == S) ? 1.0f : (float)(j + 1) * step;
        const float lower = j == 0 ? b : (prev + b) * 0.5f;
        const float upper = j == S ? b : (b + next) * 0.5f;
        b = lower + (upper - lower) * jitter[r];
    }
    return b;
}

__global__ void __launch_bounds__(256)
k_lindisp_positions(uint32_t R, uint32_t S, float near, float far, const float* __restrict__ jitter,
                    const float* __restrict__ origins, const float* __restrict__ directions,
                    float* __restrict__ sbins, float* __restrict__ tbins, float* __restrict__ x01) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * S) return;
    const uint32_t r = i / S, j = i - r * S;
    const float s_near = spacing_fn(near), s_far = spacing_fn(far);
    const float b0 = lindisp_bin(j, S, jitter, r), b1 = lindisp_bin(j + 1, S, jitter, r);
    const float t0 = spacing_fn_inv(b0 * s_far + (1.f - b0) * s_near);
    const float t1 = spacing_fn_inv(b1 * s_far + (1.f - b1) * s_near);
    sbins[(size_t)r * (S + 1) + j] = b0;
    tbins[(size_t)r * (S + 1) + j] = t0;
    if (j + 1 == S) {
        sbins[(size_t)r * (S + 1) + S] = b1;
        tbins[(size_t)r * (S + 1) + S] = t1;
    }
    const float o[3] = {origins[3 * (size_t)r], origins[3 * (size_t)r + 1], origins[3 * (size_t)r + 2]};
    const float d[3] = {directions[3 * (size_t)r], directions[3 * (size_t)r + 1], directions[3 * (size_t)r + 2]};
    float p[3];
    nvo_contract_position01(o, d, (t0 + t1) * 0.5f, p);
#pragma unroll
    for (int k = 0; k < 3; ++k) x01[3 * (size_t)i + k] = p[k];
}

// Pixel sampler + per-ray sampler jitters of one step in one launch (PixelSampler: uniform (camera, y, x);
// the proposal sampler's single_jitter: one uniform per ray and level).  Counter-based generator: every
// value is a hash of (seed, step, element), so the kernel is stateless and replays correctly from a
// hipGraph -- the step counter is read from device memory.  Statistical quality only matters, not the
// stream: no parity test depends on RNG streams.
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float hash_uniform(uint32_t seed, uint32_t step, uint32_t stream, uint32_t i) {
    const uint32_t h = pcg_hash(pcg_hash(pcg_hash(seed ^ (stream * 0x9E3779B9u)) + step) + i);
    return (float)(h >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

__global__ void __launch_bounds__(256)
k_sample_pixels(uint32_t R, uint32_t seed, const float* __restrict__ step_dev, const float* __restrict__ extent,
                int64_t* __restrict__ ray_indices, float* __restrict__ jitter, uint32_t n_jitter) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const uint32_t step = (uint32_t)step_dev[0];
#pragma unroll
    for (uint32_t c = 0; c < 3; ++c) {
        const float e = extent[c];
        const float v = floorf(hash_uniform(seed, step, c, r) * e);
        ray_indices[3 * (size_t)r + c] = (int64_t)fminf(v, e - 1.f);
    }
    for (uint32_t j = 0; j < n_jitter; ++j) jitter[(size_t)j * R + r] = hash_uniform(seed, step, 3u + j, r);
}

// The whole per-ray prefix of a training step in ONE launch: pixel sampler + sampler jitters (k_sample_pixels), ray
// generation with pose correction (k_raygen), target gather + direction-encoding input (k_gather_targets), SH(4) of
// the direction (k_sh_fwd) and the first sampler level (k_lindisp_positions).  Thread per (ray, sample of level 0):
// every thread re-derives its ray (a hash and ~80 flops -- free next to the five launches it replaces, ~6 us each of
// dispatch + drain for 4096-ray kernels); the thread of sample 0 writes the per-ray outputs.  Same device functions
// as the separate kernels: the results are bit-identical (tests/test_engine_gpu.py::test_ray_head_matches_separate).
// Workgroups behind the ray blocks (zero.n != 0: nvo_ray_head_zero) clear the step's accumulate-into buffers: the first
// launch of a one-graph step does both, a 5 us launch less in front of the main field's forward.
__global__ void __launch_bounds__(256)
k_ray_head(nvo_ray_head_args a, NvoZeroPlan zero, uint32_t ray_blocks) {
    if (blockIdx.x >= ray_blocks) {  // (uniform)
        nvo_zero_plan_block(zero, blockIdx.x - ray_blocks);
        return;
    }
    // A workgroup's 256 samples belong to a handful of rays (ONE at 256 samples per ray): the first lanes derive them --
    // three hashes, ~30 loads of intrinsics / pose / correction, ~200 flops each -- and the workgroup takes them from LDS
    // (18.7 -> 17.4 us: most of the launch is its 25-31 MB of stores and the zero plan).  Same
    // device functions on the same inputs: the values do not change.
    constexpr uint32_t kHeadRays = 8;
    struct HeadRay {
        int32_t idx[3];
        float jit0, o[3], d[3], n0, area;
    };
    __shared__ HeadRay s_ray[kHeadRays];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = a.R * a.S;
    const uint32_t i_first = blockIdx.x * blockDim.x;
    const uint32_t i_last = min(total, i_first + blockDim.x) - 1u;  // (i_first < total: the grid covers exactly the samples)
    const uint32_t r_first = i_first / a.S, n_rays = i_last / a.S - r_first + 1u;
    const bool shared_rays = n_rays <= kHeadRays;  // (uniform)
    const uint32_t step = (uint32_t)a.step_dev[0];
    auto derive = [&](uint32_t r, HeadRay& h) {
        int64_t idx[3];
#pragma unroll
        for (uint32_t c = 0; c < 3; ++c) {
            const float e = a.extent_dev[c];
            const float v = floorf(hash_uniform(a.seed, step, c, r) * e);
            idx[c] = (int64_t)fminf(v, e - 1.f);
            h.idx[c] = (int32_t)idx[c];
        }
        h.jit0 = hash_uniform(a.seed, step, 3u, r);
        const int64_t cam = idx[0];
        raygen_one((float)idx[2] + 0.5f, (float)idx[1] + 0.5f, a.intrinsics + 4 * cam, a.c2w + (size_t)a.c2w_stride * cam,
                   a.corrections ? a.corrections + 12 * cam : nullptr, h.o, h.d, &h.n0, &h.area);
    };
    if (shared_rays) {
        if (threadIdx.x < n_rays) derive(r_first + threadIdx.x, s_ray[threadIdx.x]);
        __syncthreads();
    }
    if (i >= total) return;
    const uint32_t r = i / a.S, j = i - r * a.S;
    HeadRay hr;
    if (shared_rays) hr = s_ray[r - r_first]; else derive(r, hr);
    const int64_t idx[3] = {hr.idx[0], hr.idx[1], hr.idx[2]};
    const float jit0 = hr.jit0;
    const int64_t cam = idx[0];
    float o[3] = {hr.o[0], hr.o[1], hr.o[2]}, d[3] = {hr.d[0], hr.d[1], hr.d[2]};
    const float n0 = hr.n0, area = hr.area;
    // ---- first sampler level (k_lindisp_positions)
    const float s_near = spacing_fn(a.near_plane), s_far = spacing_fn(a.far_plane);
    const float b0 = lindisp_bin(j, a.S, &jit0, 0), b1 = lindisp_bin(j + 1, a.S, &jit0, 0);
    const float t0 = spacing_fn_inv(b0 * s_far + (1.f - b0) * s_near);
    const float t1 = spacing_fn_inv(b1 * s_far + (1.f - b1) * s_near);
    a.sbins[(size_t)r * (a.S + 1) + j] = b0;
    a.tbins[(size_t)r * (a.S + 1) + j] = t0;
    if (j + 1 == a.S) {
        a.sbins[(size_t)r * (a.S + 1) + a.S] = b1;
        a.tbins[(size_t)r * (a.S + 1) + a.S] = t1;
    }
    float p[3];
    nvo_contract_position01(o, d, (t0 + t1) * 0.5f, p);
#pragma unroll
    for (int k = 0; k < 3; ++k) a.x01[3 * (size_t)i + k] = p[k];
    if (j != 0) return;
    // ---- per-ray outputs
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a.ray_indices[3 * (size_t)r + k] = idx[k];
        a.origins[3 * (size_t)r + k] = o[k];
        a.directions[3 * (size_t)r + k] = d[k];
    }
    a.jitter[r] = jit0;
    for (uint32_t q = 1; q < a.n_jitter; ++q) a.jitter[(size_t)q * a.R + r] = hash_uniform(a.seed, step, 3u + q, r);
    a.directions_norm[r] = n0;
    if (a.pixel_area) a.pixel_area[r] = area;
    a.cam_idx[r] = (int32_t)cam;
    const size_t pix = ((size_t)cam * a.H + (size_t)idx[1]) * a.W + (size_t)idx[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.gt_rgb[3 * (size_t)r + k] = a.images[3 * pix + k];
    if (a.depths) a.gt_depth[r] = a.depths[pix];
    if (a.normals) {
#pragma unroll
        for (int k = 0; k < 3; ++k) a.gt_normal[3 * (size_t)r + k] = a.normals[3 * pix + k];
    }
    float d01[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d01[k] = (d[k] + 1.f) * 0.5f;
        a.dirs01[3 * (size_t)r + k] = d01[k];
    }
    if (a.sh) {  // k_sh_fwd on (d + 1) / 2, degree 4
        float c[16];
        nvo_sh4_eval(d01[0] * 2.f - 1.f, d01[1] * 2.f - 1.f, d01[2] * 2.f - 1.f, 4u, c);
        nvo_h16* __restrict__ sp = (nvo_h16*)a.sh + 16 * (size_t)r;
#pragma unroll
        for (int k = 0; k < 16; ++k) sp[k] = nvo_cvt16(c[k], a.sh_bf16 != 0);
    }
}

// Fused ray setup: ray generation + gather of the colour / depth / normal targets + the (d + 1) / 2
// direction-encoding input, one thread per ray.
__global__ void __launch_bounds__(256)
k_gather_targets(uint32_t R, const int64_t* __restrict__ ray_indices, uint32_t H, uint32_t W,
                 const float* __restrict__ images, const float* __restrict__ depths,
                 const float* __restrict__ normals, const float* __restrict__ directions,
                 float* __restrict__ gt_rgb, float* __restrict__ gt_depth, float* __restrict__ gt_normal,
                 float* __restrict__ dirs01) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int64_t cam = ray_indices[3 * (size_t)r + 0], y = ray_indices[3 * (size_t)r + 1],
                  x = ray_indices[3 * (size_t)r + 2];
    const size_t pix = ((size_t)cam * H + (size_t)y) * W + (size_t)x;
#pragma unroll
    for (int k = 0; k < 3; ++k) gt_rgb[3 * (size_t)r + k] = images[3 * pix + k];
    if (depths) gt_depth[r] = depths[pix];
    if (normals) {
#pragma unroll
        for (int k = 0; k < 3; ++k) gt_normal[3 * (size_t)r + k] = normals[3 * pix + k];
    }
    if (dirs01) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dirs01[3 * (size_t)r + k] = (directions[3 * (size_t)r + k] + 1.f) * 0.5f;
    }
}

// direction encoding input: (d + 1) / 2 per ray
__global__ void __launch_bounds__(256)
k_dirs01(uint32_t n, const float* __restrict__ d, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (d[i] + 1.f) * 0.5f;
}

// exp_map_SE3 of nerfstudio's camera optimizer (cameras/lie_groups.py): tangent [n][6] = (translation
// | rotation) -> [n][3][4], with the same small-angle Taylor branches (theta < 1e-2).
__global__ void __launch_bounds__(256)
k_se3_exp(uint32_t n, const float* __restrict__ tangent, float* __restrict__ out, int mode) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* t = tangent + 6 * (size_t)i;
    const float lx = t[0], ly = t[1], lz = t[2], ax = t[3], ay = t[4], az = t[5];
    if (mode == 1) {
        // exp_map_SO3xR3: R = I + sin(a)/a K + (1-cos a)/a^2 K^2 with a = sqrt(max(|w|^2, 1e-4)), t = lin
        const float ang = sqrtf(fmaxf(ax * ax + ay * ay + az * az, 1e-4f));
        const float f1 = sinf(ang) / ang, f2 = (1.f - cosf(ang)) / (ang * ang);
        float* o = out + 12 * (size_t)i;
        o[0] = 1.f + f2 * (-az * az - ay * ay); o[1] = -f1 * az + f2 * ax * ay;         o[2] = f1 * ay + f2 * ax * az;
        o[4] = f1 * az + f2 * ax * ay;          o[5] = 1.f + f2 * (-az * az - ax * ax); o[6] = -f1 * ax + f2 * ay * az;
        o[8] = -f1 * ay + f2 * ax * az;         o[9] = f1 * ax + f2 * ay * az;          o[10] = 1.f + f2 * (-ay * ay - ax * ax);
        o[3] = lx; o[7] = ly; o[11] = lz;
        return;
    }
    const float theta2 = ax * ax + ay * ay + az * az;
    const float theta = sqrtf(theta2);
    const bool nz = theta < 1e-2f;
    const float sine = sinf(theta);
    const float cosine = nz ? 8.f / (4.f + theta2) - 1.f : cosf(theta);
    const float sbt = nz ? 0.5f * cosine + 0.5f : sine / theta;
    const float omc = nz ? 0.5f * sbt : (1.f - cosine) / theta2;
    float* o = out + 12 * (size_t)i;
    o[0] = omc * ax * ax + cosine;   o[1] = omc * ax * ay - sbt * az;  o[2] = omc * ax * az + sbt * ay;
    o[4] = omc * ay * ax + sbt * az; o[5] = omc * ay * ay + cosine;    o[6] = omc * ay * az - sbt * ax;
    o[8] = omc * az * ax - sbt * ay; o[9] = omc * az * ay + sbt * ax;  o[10] = omc * az * az + cosine;
    const float sbt_t = nz ? 1.f - theta2 / 6.f : sbt;
    const float omc_t = nz ? 0.5f - theta2 / 24.f : omc;
    const float tms = nz ? 1.f / 6.f - theta2 / 120.f : (theta - sine) / (theta2 * theta);
    const float cx = ay * lz - az * ly, cy = az * lx - ax * lz, cz = ax * ly - ay * lx;  // ang x lin
    const float dot = ax * lx + ay * ly + az * lz;
    o[3] = sbt_t * lx + omc_t * cx + tms * ax * dot;
    o[7] = sbt_t * ly + omc_t * cy + tms * ay * dot;
    o[11] = sbt_t * lz + omc_t * cz + tms * az * dot;
}

}  // namespace

extern "C" {

int nvo_pose_exp_map(nvo_stream_t stream, uint32_t n, const float* tangent, float* out, int mode) {
    NVO_REQUIRE(n == 0 || (tangent && out), "pose_exp_map: NULL argument");
    NVO_REQUIRE(mode == 0 || mode == 1, "pose_exp_map: mode %d (0 = SE3, 1 = SO3xR3)", mode);
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "pose_exp_map");
    NVO_LAUNCH(k_se3_exp, dim3(nvo_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, tangent, out, mode);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_se3_exp_map(nvo_stream_t stream, uint32_t n, const float* tangent, float* out) {
    NVO_REQUIRE(n == 0 || (tangent && out), "se3_exp_map: NULL argument");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "se3_exp_map");
    NVO_LAUNCH(k_se3_exp, dim3(nvo_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, tangent, out, 0);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_raygen(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics,
               const float* c2w, const float* corrections, float* origins, float* directions,
               float* directions_norm, float* pixel_area, int32_t* cam_idx) {
    NVO_REQUIRE(R == 0 || (ray_indices && intrinsics && c2w && origins && directions &&
                           directions_norm && cam_idx), "raygen: NULL argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "raygen");
    NVO_LAUNCH(k_raygen, dim3(nvo_div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, R,
                       ray_indices, intrinsics, c2w, corrections, origins, directions, directions_norm,
                       pixel_area, cam_idx);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_rays_given(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics, const float* c2w,
                   const float* corrections, uint32_t H, uint32_t W, const float* images, const float* depths, float* origins,
                   float* directions, float* directions_norm, float* pixel_area, int32_t* cam_idx, float* gt_rgb,
                   float* gt_depth, float* dirs01, void* sh_half, const float* depths_cov, float* gt_depth_cov,
                   const uint32_t* R_dev) {
    NVO_REQUIRE(R == 0 || (ray_indices && intrinsics && c2w && images && origins && directions && directions_norm && cam_idx &&
                           gt_rgb && dirs01 && sh_half && (!depths || gt_depth) && (!depths_cov || gt_depth_cov)),
                "rays_given: NULL argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "rays_given");
    NVO_LAUNCH(k_rays_given, dim3(nvo_div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, R, ray_indices, intrinsics, c2w,
               corrections, H, W, images, depths, origins, directions, directions_norm, pixel_area, cam_idx, gt_rgb, gt_depth,
               dirs01, (nvo_h16*)sh_half, depths_cov, gt_depth_cov, R_dev);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_gather_pixels(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, uint32_t H,
                      uint32_t W, uint32_t Cn, const float* images, float* out) {
    NVO_REQUIRE(R == 0 || (ray_indices && images && out && Cn >= 1), "gather_pixels: NULL argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "gather_pixels");
    NVO_LAUNCH(k_gather_pixels, dim3(nvo_div_up((uint64_t)R * Cn, 256)), dim3(256), 0,
                       (hipStream_t)stream, R, ray_indices, H, W, Cn, images, out);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_sample_lindisp(nvo_stream_t stream, uint32_t R, uint32_t S, float near_plane, float far_plane,
                       const float* jitter, float* sbins, float* tbins) {
    NVO_REQUIRE(S >= 1 && (R == 0 || (sbins && tbins)), "sample_lindisp: bad argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "sample_lindisp");
    NVO_LAUNCH(k_sample_lindisp, dim3(nvo_div_up((uint64_t)R * (S + 1), 256)), dim3(256), 0,
                       (hipStream_t)stream, R, S, near_plane, far_plane, jitter, sbins, tbins);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_sample_positions(nvo_stream_t stream, uint32_t R, uint32_t S, const float* origins,
                         const float* directions, const float* tbins, float* x01) {
    NVO_REQUIRE(S >= 1 && (R == 0 || (origins && directions && tbins && x01)), "sample_positions: bad argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "sample_positions[S%u]", S);
    NVO_LAUNCH(k_sample_positions, dim3(nvo_div_up((uint64_t)R * S, 256)), dim3(256), 0,
                       (hipStream_t)stream, R, S, origins, directions, tbins, x01);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_sample_pixels(nvo_stream_t stream, uint32_t R, uint32_t seed, const float* step_dev, const float* extent_dev,
                      int64_t* ray_indices, float* jitter, uint32_t n_jitter) {
    NVO_REQUIRE(R == 0 || (step_dev && extent_dev && ray_indices && (n_jitter == 0 || jitter)), "sample_pixels: NULL argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "sample_pixels");
    NVO_LAUNCH(k_sample_pixels, dim3(nvo_div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, R, seed, step_dev,
               extent_dev, ray_indices, jitter, n_jitter);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_lindisp_positions(nvo_stream_t stream, uint32_t R, uint32_t S, float near_plane, float far_plane,
                          const float* jitter, const float* origins, const float* directions, float* sbins,
                          float* tbins, float* x01) {
    NVO_REQUIRE(S >= 1 && (R == 0 || (origins && directions && sbins && tbins && x01)), "lindisp_positions: bad argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "lindisp_positions[S%u]", S);
    NVO_LAUNCH(k_lindisp_positions, dim3(nvo_div_up((uint64_t)R * S, 256)), dim3(256), 0, (hipStream_t)stream, R, S,
               near_plane, far_plane, jitter, origins, directions, sbins, tbins, x01);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_gather_targets(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, uint32_t H, uint32_t W,
                       const float* images, const float* depths, const float* normals, const float* directions,
                       float* gt_rgb, float* gt_depth, float* gt_normal, float* dirs01) {
    NVO_REQUIRE(R == 0 || (ray_indices && images && gt_rgb), "gather_targets: NULL argument");
    NVO_REQUIRE(!depths || gt_depth, "gather_targets: gt_depth is NULL");
    NVO_REQUIRE(!normals || gt_normal, "gather_targets: gt_normal is NULL");
    NVO_REQUIRE(!dirs01 || directions, "gather_targets: directions is NULL");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "gather_targets");
    NVO_LAUNCH(k_gather_targets, dim3(nvo_div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, R, ray_indices, H, W,
               images, depths, normals, directions, gt_rgb, gt_depth, gt_normal, dirs01);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_dirs01(nvo_stream_t stream, uint32_t n, const float* d, float* out) {
    NVO_REQUIRE(n == 0 || (d && out), "dirs01: NULL argument");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "dirs01");
    NVO_LAUNCH(k_dirs01, dim3(nvo_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, d, out);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_sh_encode(nvo_stream_t stream, uint32_t R, uint32_t degree, const float* dirs01, void* out_half) {
    NVO_REQUIRE(R == 0 || (dirs01 && out_half), "sh_encode: NULL argument");
    return nvo_sh_fwd_launch((hipStream_t)stream, R, degree, dirs01, out_half, 16, 16);
}

static int ray_head_launch(nvo_stream_t stream, const nvo_ray_head_args& a, uint32_t n_ranges, void* const* ptrs,
                           const uint64_t* bytes) {
    NvoZeroPlan zero;
    memset(&zero, 0, sizeof(zero));
    int zero_blocks = 0;
    if (n_ranges) {
        zero_blocks = nvo_zero_plan_build(n_ranges, ptrs, bytes, &zero);
        if (zero_blocks < 0) return NVO_ERR_INVALID;
    }
    const uint32_t ray_blocks = (uint32_t)nvo_div_up((uint64_t)a.R * a.S, 256);
    if (ray_blocks + (uint32_t)zero_blocks == 0) return NVO_OK;
    NVO_PROF(stream, "ray_head[S%u]", a.S);
    NVO_LAUNCH(k_ray_head, dim3(ray_blocks + (uint32_t)zero_blocks), dim3(256), 0, (hipStream_t)stream, a, zero, ray_blocks);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_ray_head(nvo_stream_t stream, const nvo_ray_head_args* args) {
    NVO_REQUIRE(args != nullptr, "ray_head: args is NULL");
    const nvo_ray_head_args a = *args;
    NVO_REQUIRE(a.S >= 1 && a.n_jitter >= 1 && (a.c2w_stride == 12 || a.c2w_stride == 16), "ray_head: bad S / n_jitter / c2w_stride");
    NVO_REQUIRE(a.R == 0 || (a.step_dev && a.extent_dev && a.intrinsics && a.c2w && a.images && a.ray_indices &&
                             a.jitter && a.origins && a.directions && a.directions_norm && a.cam_idx && a.gt_rgb &&
                             a.dirs01 && a.sbins && a.tbins && a.x01), "ray_head: NULL argument");
    NVO_REQUIRE(!a.depths || a.gt_depth, "ray_head: depths without gt_depth");
    NVO_REQUIRE(!a.normals || a.gt_normal, "ray_head: normals without gt_normal");
    return ray_head_launch(stream, a, 0, nullptr, nullptr);
}

int nvo_ray_head_zero(nvo_stream_t stream, const nvo_ray_head_args* args, uint32_t n_ranges, void* const* ptrs,
                      const uint64_t* bytes) {
    NVO_REQUIRE(args != nullptr, "ray_head_zero: args is NULL");
    const nvo_ray_head_args a = *args;
    NVO_REQUIRE(a.S >= 1 && a.n_jitter >= 1 && (a.c2w_stride == 12 || a.c2w_stride == 16), "ray_head: bad S / n_jitter / c2w_stride");
    NVO_REQUIRE(a.R == 0 || (a.step_dev && a.extent_dev && a.intrinsics && a.c2w && a.images && a.ray_indices &&
                             a.jitter && a.origins && a.directions && a.directions_norm && a.cam_idx && a.gt_rgb &&
                             a.dirs01 && a.sbins && a.tbins && a.x01), "ray_head: NULL argument");
    NVO_REQUIRE(!a.depths || a.gt_depth, "ray_head: depths without gt_depth");
    NVO_REQUIRE(!a.normals || a.gt_normal, "ray_head: normals without gt_normal");
    return ray_head_launch(stream, a, n_ranges, ptrs, bytes);
}

int nvo_sh_encode_t(nvo_stream_t stream, uint32_t R, uint32_t degree, const float* dirs01, void* out, int out_bf16) {
    NVO_REQUIRE(R == 0 || (dirs01 && out), "sh_encode_t: NULL argument");
    return nvo_sh_fwd_launch((hipStream_t)stream, R, degree, dirs01, out, 16, 16, out_bf16 != 0);
}

}  // extern "C"
