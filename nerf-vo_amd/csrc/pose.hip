// Pose-gradient chain of the SE3 camera optimiser for gfx950: the
// reference trains with CameraOptimizerConfig(mode='SE3') (/root/reference/nerf_vo/mapping/
// nerfstudio.py:64) and exports the optimised poses (:208-216), so dL/d(pose_adjustment) must flow
//   dL/dx01 (k_grid_bwd_input)  ->  k_positions_bwd : selector, (x+2)/4, L-inf contraction Jacobian,
//                                   per-ray reduction -> dL/dorigin, dL/ddirection
//   dL/dSH   (colour-head bwd)  ->  nvo_sh_bwd_input_f32 -> dL/ddirection (x 1/2)
//   dL/dorigin, dL/ddirection   ->  k_pose_bwd      : o = t + t_corr, d = R_corr d_raw -> dL/dcorrection[cam]
//   dL/dcorrection [F][3][4]    ->  k_se3_exp_bwd   : forward-mode dual numbers through exp_map_SE3
//                                   (+ the camera_opt regulariser) -> dL/dpose_adjustment [F][6]
// In nerfstudio all of this is autograd through torch ops [UPSTREAM]; CPU restatement: oracle/rays.py
// (exp_map_se3, apply_pose_correction, contract_linf) differentiated by torch autograd.
#include "nvo_kernels.h"
#include "../../include/nerfvo_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
    return nvo_wave_sum(v);  // DPP form (nvo_common.h)
}

// One wave per ray: lanes stride over the ray's samples.
// dx01: [R*S][3] gradient w.r.t. the normalised grid coordinate (already carries the loss scale)
// d_origin / d_dir: [R][3], ACCUMULATED (caller zeroes once per step; called once per level)
__global__ void __launch_bounds__(256)
k_positions_bwd(uint32_t R, uint32_t S, const float* __restrict__ origins,
                const float* __restrict__ directions, const float* __restrict__ tbins,
                const float* __restrict__ dx01, float* __restrict__ d_origin, float* __restrict__ d_dir) {
    const int lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= R) return;
    const float o[3] = {origins[3 * (size_t)r], origins[3 * (size_t)r + 1], origins[3 * (size_t)r + 2]};
    const float d[3] = {directions[3 * (size_t)r], directions[3 * (size_t)r + 1], directions[3 * (size_t)r + 2]};
    const float* tb = tbins + (size_t)r * (S + 1);
    float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
    for (uint32_t s = lane; s < S; s += 64) {
        const float mid = 0.5f * (tb[s] + tb[s + 1]);
        float p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = o[k] + d[k] * mid;
        // forward: contraction, normalisation, selector (as k_sample_positions)
        const float a0 = fabsf(p[0]), a1 = fabsf(p[1]), a2 = fabsf(p[2]);
        const float mag = fmaxf(a0, fmaxf(a1, a2));
        const int m = (a0 >= a1 && a0 >= a2) ? 0 : (a1 >= a2 ? 1 : 2);
        const bool contracted = !(mag < 1.f);
        const float f = contracted ? (2.f - 1.f / mag) / mag : 1.f;
        bool sel = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x = (p[k] * f + 2.f) * 0.25f;
            sel = sel && (x > 0.f) && (x < 1.f);
        }
        if (!sel) continue;
        const float* gx = dx01 + 3 * ((size_t)r * S + s);
        float g[3] = {gx[0] * 0.25f, gx[1] * 0.25f, gx[2] * 0.25f};  // d x01 / d contracted = 1/4
        if (contracted) {
            // c = f(mag) p, f = 2/mag - 1/mag^2, mag = |p_m|:
            // dc_i/dp_j = f delta_ij + p_i f'(mag) sign(p_m) delta_jm
            const float fp = -2.f / (mag * mag) + 2.f / (mag * mag * mag);
            const float dot = g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
            const float sgn = p[m] >= 0.f ? 1.f : -1.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] *= f;
            g[m] += sgn * fp * dot;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            go[k] += g[k];
            gd[k] += g[k] * mid;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        go[k] = wave_sum(go[k]);
        gd[k] = wave_sum(gd[k]);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d_origin[3 * (size_t)r + k] += go[k];
            d_dir[3 * (size_t)r + k] += gd[k];
        }
    }
}

__device__ __forceinline__ void rot_apply(const float* __restrict__ m, float x, float y, float z, float* o) {
    o[0] = m[0] * x + m[1] * y + m[2] * z;
    o[1] = m[4] * x + m[5] * y + m[6] * z;
    o[2] = m[8] * x + m[9] * y + m[10] * z;
}

// d_corr: [F][3][4] accumulated with float atomics (pre-zeroed).  d_dir01 (nullable): gradient w.r.t.
// (d+1)/2 coming from the SH encoding, added as 0.5 * d_dir01.
__global__ void __launch_bounds__(256)
k_pose_bwd(uint32_t R, const int64_t* __restrict__ ray_indices, const float* __restrict__ intrinsics,
           const float* __restrict__ c2w, const float* __restrict__ d_origin,
           const float* __restrict__ d_dir, const float* __restrict__ d_dir01,
           float* __restrict__ d_corr, float* __restrict__ per_ray, uint32_t n_cameras) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int64_t cam = ray_indices[3 * (size_t)r + 0];
    if (n_cameras && (uint64_t)cam >= (uint64_t)n_cameras) {  // stale / out-of-range index: dropped before any read through it
        if (per_ray)
            for (int k = 0; k < 12; ++k) per_ray[12 * (size_t)r + k] = 0.f;
        return;
    }
    const float py = (float)ray_indices[3 * (size_t)r + 1] + 0.5f;
    const float px = (float)ray_indices[3 * (size_t)r + 2] + 0.5f;
    const float fx = intrinsics[4 * cam + 0], fy = intrinsics[4 * cam + 1];
    const float cx = intrinsics[4 * cam + 2], cy = intrinsics[4 * cam + 3];
    float d0[3];
    rot_apply(c2w + 12 * cam, (px - cx) / fx, -(py - cy) / fy, -1.f, d0);
    const float n0 = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
    float gd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d0[k] /= n0;
        gd[k] = d_dir[3 * (size_t)r + k] + (d_dir01 ? 0.5f * d_dir01[3 * (size_t)r + k] : 0.f);
    }
    if (per_ray) {  // deterministic form: stored per ray, summed per camera in a fixed order by nvo_reduce_by_camera
        float* g = per_ray + 12 * (size_t)r;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) g[4 * i + j] = gd[i] * d0[j];
            g[4 * i + 3] = d_origin[3 * (size_t)r + i];
        }
        return;
    }
    float* g = d_corr + 12 * cam;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) atomicAdd(g + 4 * i + j, gd[i] * d0[j]);  // d = R d_raw
        atomicAdd(g + 4 * i + 3, d_origin[3 * (size_t)r + i]);                 // o = t + t_corr
    }
}

// The same sum with a per-workgroup table in LDS: a camera owns 12 words, a batch of R rays over F cameras puts
// R / F rays on each of them, and as global float atomics those serialise in the L2 (7552 rays over 48 cameras:
// 157 adds per address, 60 us).  A workgroup of 256 - 1024 rays adds into its LDS copy of the table and flushes the non-zero
// words once: one global add per workgroup and address.  Same values, same (unordered) float summation as k_pose_bwd.
__global__ void __launch_bounds__(1024)
k_pose_bwd_lds(uint32_t R, uint32_t n_cameras, const int64_t* __restrict__ ray_indices, const float* __restrict__ intrinsics,
               const float* __restrict__ c2w, const float* __restrict__ d_origin, const float* __restrict__ d_dir,
               const float* __restrict__ d_dir01, float* __restrict__ d_corr) {
    extern __shared__ float pose_acc[];
    const uint32_t n_words = 12u * n_cameras;
    for (uint32_t e = threadIdx.x; e < n_words; e += blockDim.x) pose_acc[e] = 0.f;
    __syncthreads();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    // (a stale or out-of-range camera index is DROPPED before anything is read through it, not just before the adds)
    if (r < R && (uint64_t)ray_indices[3 * (size_t)r + 0] < (uint64_t)n_cameras) {
        const int64_t cam = ray_indices[3 * (size_t)r + 0];
        const float py = (float)ray_indices[3 * (size_t)r + 1] + 0.5f;
        const float px = (float)ray_indices[3 * (size_t)r + 2] + 0.5f;
        const float fx = intrinsics[4 * cam + 0], fy = intrinsics[4 * cam + 1];
        const float cx = intrinsics[4 * cam + 2], cy = intrinsics[4 * cam + 3];
        float d0[3];
        rot_apply(c2w + 12 * cam, (px - cx) / fx, -(py - cy) / fy, -1.f, d0);
        const float n0 = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
        float gd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d0[k] /= n0;
            gd[k] = d_dir[3 * (size_t)r + k] + (d_dir01 ? 0.5f * d_dir01[3 * (size_t)r + k] : 0.f);
        }
        {
            float* g = pose_acc + 12 * cam;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) atomicAdd(g + 4 * i + j, gd[i] * d0[j]);
                atomicAdd(g + 4 * i + 3, d_origin[3 * (size_t)r + i]);
            }
        }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < n_words; e += blockDim.x) {
        const float v = pose_acc[e];
        if (v != 0.f) atomicAdd(d_corr + e, v);
    }
}

// ---- forward-mode dual numbers (value + 6 partials) for the exp-map Jacobian ---------------------
struct Dual {
    float v;
    float d[6];
};
__device__ __forceinline__ Dual dconst(float c) {
    Dual r;
    r.v = c;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = 0.f;
    return r;
}
__device__ __forceinline__ Dual dvar(float c, int idx) {
    Dual r = dconst(c);
    r.d[idx] = 1.f;
    return r;
}
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) {
    Dual r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) {
    Dual r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) {
    Dual r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
    Dual r;
    r.v = a.v / b.v;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
__device__ __forceinline__ Dual dscale(const Dual& a, float c) {
    Dual r;
    r.v = a.v * c;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * c;
    return r;
}
__device__ __forceinline__ Dual dfun(const Dual& a, float value, float derivative) {
    Dual r;
    r.v = value;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * derivative;
    return r;
}

// d_tangent [F][6] (OVERWRITTEN): J^T d_corr + loss_scale * regulariser gradient
__global__ void __launch_bounds__(64)
k_se3_exp_bwd(uint32_t n, const float* __restrict__ tangent, const float* __restrict__ d_corr,
              float trans_penalty, float rot_penalty, float reg_scale, float* __restrict__ d_tangent,
              int mode, const float* __restrict__ reg_scale_dev) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (reg_scale_dev) reg_scale *= *reg_scale_dev;  // dynamic loss scale (device float)
    const float* t = tangent + 6 * (size_t)i;
    Dual l[3] = {dvar(t[0], 0), dvar(t[1], 1), dvar(t[2], 2)};
    Dual a[3] = {dvar(t[3], 3), dvar(t[4], 4), dvar(t[5], 5)};
    const Dual theta2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    const float th = sqrtf(theta2.v);
    Dual o[12];
    if (mode == 1) {  // exp_map_SO3xR3 (clamp(|w|^2, 1e-4): zero derivative below the clamp)
        const bool clamped = theta2.v < 1e-4f;
        const float ang_v = sqrtf(fmaxf(theta2.v, 1e-4f));
        const Dual ang = dfun(theta2, ang_v, clamped ? 0.f : 0.5f / ang_v);
        const Dual f1 = dfun(ang, sinf(ang_v), cosf(ang_v)) / ang;
        const Dual f2 = (dconst(1.f) - dfun(ang, cosf(ang_v), -sinf(ang_v))) / (ang * ang);
        const Dual one = dconst(1.f);
        o[0] = one - f2 * (a[2] * a[2] + a[1] * a[1]); o[1] = f2 * a[0] * a[1] - f1 * a[2];           o[2] = f1 * a[1] + f2 * a[0] * a[2];
        o[4] = f1 * a[2] + f2 * a[0] * a[1];           o[5] = one - f2 * (a[2] * a[2] + a[0] * a[0]); o[6] = f2 * a[1] * a[2] - f1 * a[0];
        o[8] = f2 * a[0] * a[2] - f1 * a[1];           o[9] = f1 * a[0] + f2 * a[1] * a[2];           o[10] = one - f2 * (a[1] * a[1] + a[0] * a[0]);
        o[3] = l[0]; o[7] = l[1]; o[11] = l[2];
    } else {
    const bool nz = th < 1e-2f;
    // theta = sqrt(theta2); its derivative is irrelevant in the small-angle branch (theta only enters
    // through theta2 there), and finite otherwise
    const Dual theta = dfun(theta2, th, nz ? 0.f : 0.5f / th);
    Dual sine = dfun(theta, sinf(th), cosf(th));
    Dual cosine, sbt, omc, sbt_t, omc_t, tms;
    if (nz) {
        cosine = dconst(8.f) / (dconst(4.f) + theta2) - dconst(1.f);
        sbt = dscale(cosine, 0.5f) + dconst(0.5f);
        omc = dscale(sbt, 0.5f);
        sbt_t = dconst(1.f) - dscale(theta2, 1.f / 6.f);
        omc_t = dconst(0.5f) - dscale(theta2, 1.f / 24.f);
        tms = dconst(1.f / 6.f) - dscale(theta2, 1.f / 120.f);
    } else {
        cosine = dfun(theta, cosf(th), -sinf(th));
        sbt = sine / theta;
        omc = (dconst(1.f) - cosine) / theta2;
        sbt_t = sbt;
        omc_t = omc;
        tms = (theta - sine) / (theta2 * theta);
    }
    o[0] = omc * a[0] * a[0] + cosine;      o[1] = omc * a[0] * a[1] - sbt * a[2];  o[2] = omc * a[0] * a[2] + sbt * a[1];
    o[4] = omc * a[1] * a[0] + sbt * a[2];  o[5] = omc * a[1] * a[1] + cosine;      o[6] = omc * a[1] * a[2] - sbt * a[0];
    o[8] = omc * a[2] * a[0] - sbt * a[1];  o[9] = omc * a[2] * a[1] + sbt * a[0];  o[10] = omc * a[2] * a[2] + cosine;
    const Dual cx = a[1] * l[2] - a[2] * l[1], cy = a[2] * l[0] - a[0] * l[2], cz = a[0] * l[1] - a[1] * l[0];
    const Dual dot = a[0] * l[0] + a[1] * l[1] + a[2] * l[2];
    o[3] = sbt_t * l[0] + omc_t * cx + tms * a[0] * dot;
    o[7] = sbt_t * l[1] + omc_t * cy + tms * a[1] * dot;
    o[11] = sbt_t * l[2] + omc_t * cz + tms * a[2] * dot;
    }
    const float* g = d_corr + 12 * (size_t)i;
    float out[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float gk = g[k];
#pragma unroll
        for (int j = 0; j < 6; ++j) out[j] += gk * o[k].d[j];
    }
    // camera_opt_regularizer = mean_i |trans_i| * trans_penalty + mean_i |rot_i| * rot_penalty
    // (norm's sub-gradient at 0 is 0, as in torch)
    const float tn = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    if (tn > 0.f && trans_penalty != 0.f)
        for (int j = 0; j < 3; ++j) out[j] += reg_scale * trans_penalty * t[j] / (tn * (float)n);
    if (th > 0.f && rot_penalty != 0.f)
        for (int j = 0; j < 3; ++j) out[3 + j] += reg_scale * rot_penalty * t[3 + j] / (th * (float)n);
#pragma unroll
    for (int j = 0; j < 6; ++j) d_tangent[6 * (size_t)i + j] = out[j];
}

// regulariser value (one block): sum_i |t_i| * pt / n + |r_i| * pr / n  -> atomically added to *loss
__global__ void __launch_bounds__(256)
k_pose_regularizer(uint32_t n, const float* __restrict__ tangent, float trans_penalty, float rot_penalty,
                   float* __restrict__ loss) {
    float acc = 0.f;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float* t = tangent + 6 * (size_t)i;
        acc += trans_penalty * sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) / (float)n +
               rot_penalty * sqrtf(t[3] * t[3] + t[4] * t[4] + t[5] * t[5]) / (float)n;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc);
}

}  // namespace

extern "C" {

int nvo_positions_bwd(nvo_stream_t stream, uint32_t R, uint32_t S, const float* origins,
                      const float* directions, const float* tbins, const float* dx01, float* d_origin,
                      float* d_dir) {
    NVO_REQUIRE(S >= 1 && (R == 0 || (origins && directions && tbins && dx01 && d_origin && d_dir)),
                "positions_bwd: bad argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "positions_bwd[S%u]", S);
    NVO_LAUNCH(k_positions_bwd, dim3(nvo_div_up(R, 4)), dim3(256), 0, (hipStream_t)stream, R, S, origins,
               directions, tbins, dx01, d_origin, d_dir);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

static int pose_bwd_plain(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics,
                          const float* c2w, const float* d_origin, const float* d_dir, const float* d_dir01,
                          float* d_corrections, uint32_t n_cameras) {
    NVO_REQUIRE(R == 0 || (ray_indices && intrinsics && c2w && d_origin && d_dir && d_corrections),
                "pose_bwd: NULL argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "pose_bwd");
    NVO_LAUNCH(k_pose_bwd, dim3(nvo_div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, R, ray_indices,
               intrinsics, c2w, d_origin, d_dir, d_dir01, d_corrections, (float*)nullptr, n_cameras);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_pose_bwd(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics,
                 const float* c2w, const float* d_origin, const float* d_dir, const float* d_dir01,
                 float* d_corrections) {
    return pose_bwd_plain(stream, R, ray_indices, intrinsics, c2w, d_origin, d_dir, d_dir01, d_corrections, 0u);
}

int nvo_pose_bwd_cams(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics,
                      const float* c2w, const float* d_origin, const float* d_dir, const float* d_dir01,
                      float* d_corrections, uint32_t n_cameras) {
    NVO_REQUIRE(R == 0 || (ray_indices && intrinsics && c2w && d_origin && d_dir && d_corrections),
                "pose_bwd_cams: NULL argument");
    if (R == 0) return NVO_OK;
    // The table pays when many rays meet on a camera (measured: 7552 rays / 48 cameras 60 -> 10 us; 4096 rays / 192
    // cameras 14.8 -> 22.4 us with 1024-ray workgroups, the zeroing and the flush of 2304 words per workgroup cost more than 21 adds per word
    // did): below 32 rays per camera, beyond 1024 cameras (48 KiB) or without a camera count: plain atomics.
    static const uint32_t min_per_cam = [] {
        const char* e = getenv("NVO_POSE_LDS_MIN_RAYS_PER_CAM");
        return e ? (uint32_t)atoi(e) : 32u;  // (2432 rays / 48 cameras = 50 per camera: 24.8 us plain, 9.5 us through the table)
    }();
    if (n_cameras == 0 || n_cameras > 1024 || R < min_per_cam * n_cameras)
        return pose_bwd_plain(stream, R, ray_indices, intrinsics, c2w, d_origin, d_dir, d_dir01, d_corrections, n_cameras);
    NVO_PROF(stream, "pose_bwd");
    static const uint32_t block = [] {
        const char* e = getenv("NVO_POSE_LDS_BLOCK");
        const uint32_t b = e ? (uint32_t)atoi(e) : 256u;  // measured 256 / 512 / 1024: 10.1 / 13.9 / 21.7 us
        return (b == 256u || b == 512u || b == 1024u) ? b : 256u;
    }();
    NVO_LAUNCH(k_pose_bwd_lds, dim3(nvo_div_up(R, block)), dim3(block), 12u * n_cameras * sizeof(float), (hipStream_t)stream, R,
               n_cameras, ray_indices, intrinsics, c2w, d_origin, d_dir, d_dir01, d_corrections);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_pose_bwd_det(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics,
                     const float* c2w, const float* d_origin, const float* d_dir, const float* d_dir01,
                     float* d_corrections, float* per_ray_scratch, uint32_t n_cameras) {
    NVO_REQUIRE(R == 0 || (ray_indices && intrinsics && c2w && d_origin && d_dir && d_corrections && per_ray_scratch),
                "pose_bwd_det: NULL argument");
    if (R == 0) return NVO_OK;
    NVO_PROF(stream, "pose_bwd");
    NVO_LAUNCH(k_pose_bwd, dim3(nvo_div_up(R, 256)), dim3(256), 0, (hipStream_t)stream, R, ray_indices,
               intrinsics, c2w, d_origin, d_dir, d_dir01, d_corrections, per_ray_scratch, n_cameras);
    NVO_CHECK_LAUNCH();
    return nvo_reduce_by_camera((hipStream_t)stream, R, 12, per_ray_scratch, 12, ray_indices, 1, n_cameras, d_corrections);
}

int nvo_se3_exp_map_bwd(nvo_stream_t stream, uint32_t n, const float* tangent, const float* d_corrections,
                        float trans_penalty, float rot_penalty, float reg_scale, float* d_tangent,
                        float* reg_loss, int mode) {
    return nvo_se3_exp_map_bwd_scaled(stream, n, tangent, d_corrections, trans_penalty, rot_penalty, reg_scale, d_tangent,
                                      reg_loss, mode, nullptr);
}

int nvo_se3_exp_map_bwd_scaled(nvo_stream_t stream, uint32_t n, const float* tangent, const float* d_corrections,
                               float trans_penalty, float rot_penalty, float reg_scale, float* d_tangent,
                               float* reg_loss, int mode, const float* reg_scale_dev) {
    NVO_REQUIRE(n == 0 || (tangent && d_corrections && d_tangent), "se3_exp_map_bwd: NULL argument");
    NVO_REQUIRE(mode == 0 || mode == 1, "se3_exp_map_bwd: mode %d (0 = SE3, 1 = SO3xR3)", mode);
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "se3_exp_map_bwd");
    NVO_LAUNCH(k_se3_exp_bwd, dim3(nvo_div_up(n, 64)), dim3(64), 0, (hipStream_t)stream, n, tangent,
               d_corrections, trans_penalty, rot_penalty, reg_scale, d_tangent, mode, reg_scale_dev);
    NVO_CHECK_LAUNCH();
    if (reg_loss) {
        NVO_LAUNCH(k_pose_regularizer, dim3(1), dim3(256), 0, (hipStream_t)stream, n, tangent, trans_penalty,
                   rot_penalty, reg_loss);
        NVO_CHECK_LAUNCH();
    }
    return NVO_OK;
}

}  // extern "C"
