/*
 * nerfvo_hip.h -- C-ABI of libnerfvo_hip.so, the MI355X (gfx950) implementation of NeRF-VO's mapping
 * hot path: hash-grid encoded radiance-field training step.
 *
 * Conventions
 *   - Every pointer named d_* / documented "device" is a raw HIP device pointer owned by the caller
 *     (PyTorch tensors on the Python side); the library only borrows it for the call.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises,
 *     nothing allocates after module creation (graph-capture safe).
 *   - Every function returns 0 on success or a non-zero NVO_ERR_* code; nvo_last_error() returns a
 *     thread-local message for the last failure.
 *   - Batch sizes must be multiples of 16 (tcnn's batch_size_granularity is 128; the Python layer
 *     pads to that).
 *
 * What each group replaces in the reference (jens-nau/NeRF-VO @ 2024-10-22):
 *   The reference reaches this arithmetic only through un-vendored submodules
 *   (/root/reference/.gitmodules:1-18): tiny-cuda-nn via nerfstudio's fields
 *   (call sites /root/reference/nerf_vo/mapping/nerfstudio.py:21-30,151 and
 *   /root/reference/nerf_vo/mapping/nerfstudio_utils.py:17-27,333-350) and pyngp
 *   (/root/reference/nerf_vo/mapping/instant_ngp.py:33-50,104-105).  Group A below is the FFI the
 *   tcnn torch binding (tinycudann/modules.py: _C.Module.fwd/bwd/...) binds; groups B-E are the
 *   kernels nerfstudio / nerfacc / instant-ngp run around it for one training iteration.
 */
#ifndef NERFVO_HIP_H
#define NERFVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVO_OK 0
#define NVO_ERR_INVALID 1
#define NVO_ERR_HIP 2
#define NVO_ERR_UNSUPPORTED 3

typedef struct nvo_module_s* nvo_module_t;
typedef void* nvo_stream_t; /* hipStream_t */

const char* nvo_last_error(void);
int nvo_version(void);
/* Per-launch device timing with HIP events recorded on the stream each kernel is enqueued on
 * (used by bench.py for the roofline numbers).  enable(1) clears previous records and starts
 * recording, enable(0) stops.  summary() waits for the recorded events and writes
 * "name,launches,total_ms\n" lines; returns the number of bytes the full text needs. */
int nvo_profile_enable(int on);
int64_t nvo_profile_summary(char* buf, uint64_t buf_size);

/* ------------------------------------------------------------------------------------------------
 * A. tiny-cuda-nn module boundary (tcnn bindings/torch: Module::fwd / bwd / initial_params /
 *    n_params / n_output_dims; SURVEY.md section 8b "inner boundary").
 *    encoding_json: {"otype":"HashGrid","n_levels":..,"n_features_per_level":2,
 *                    "log2_hashmap_size":..,"base_resolution":..,"per_level_scale":..}
 *                 | {"otype":"SphericalHarmonics","degree":1..4}
 *    network_json : {"otype":"FullyFusedMLP","activation":"ReLU"|"None"|"Sigmoid",
 *                    "output_activation":"None"|"Sigmoid"|"ReLU","n_neurons":16|64,
 *                    "n_hidden_layers":1..3}
 * ---------------------------------------------------------------------------------------------- */
int nvo_create_encoding(uint32_t n_input_dims, const char* encoding_json, nvo_module_t* out);
int nvo_create_network(uint32_t n_input_dims, uint32_t n_output_dims, const char* network_json,
                       nvo_module_t* out);
int nvo_create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims,
                                           const char* encoding_json, const char* network_json,
                                           nvo_module_t* out);
int nvo_destroy(nvo_module_t m);

uint32_t nvo_n_input_dims(nvo_module_t m);
uint32_t nvo_n_output_dims(nvo_module_t m);        /* logical width */
uint32_t nvo_padded_output_dims(nvo_module_t m);   /* width of the fp16 output rows fwd writes */
uint64_t nvo_n_params(nvo_module_t m);
/* Fills host_out[n_params] with the initial fp32 parameters (grid: U(-1e-4,1e-4); MLP: Xavier
 * uniform), PCG32 stream seeded with `seed` (tcnn default 1337). */
int nvo_initial_params(nvo_module_t m, uint64_t seed, float* host_out);
/* Bytes of caller-owned device scratch ("ctx") that one fwd/bwd pair of this batch size needs. */
uint64_t nvo_ctx_bytes(nvo_module_t m, uint32_t batch);
/* Integer options (tuning knobs: none changes results beyond summation order / the rounding of fixed-point
 * accumulators; which forms are bitwise reproducible run to run is said per option -- "deterministic" makes all of
 * them so).
 *   "grid_bwd_mode"   parameter-gradient kernel of a hash-grid encoding:
 *                       0 = global float atomics, 1 = LDS slice-owner scatter (module default),
 *                       3 = streamed scatter of self-contained tile-local records (no count / scan passes), coarse
 *                           levels slice-owner (what the engine selects for the main field)
 *                       (2 -- binned with gathers -- and the globally sorted record layout were removed in round 5:
 *                       nothing shipped them)
 *   "grid_stream_tile"          (mode 3) samples per scatter workgroup: 256 | 512 (default) | 1024
 *   "grid_stream_owner_slices"  (mode 3) levels with at most this many 4K-entry bins stay slice-owner (default 24)
 *   "grid_stream_overlap"       (mode 3) 1 = coarse-level launch on an auxiliary stream beside the record pipeline
 *   "grid_acc_bits"             accumulators of the slice-owner items: 64 (2^26 fixed point, default) | 32 (int32 with
 *                               a data-derived overflow-proof scale; needs 16-bit dL/dy)
 *   "prepare_input_gradients"   the forward also stores d(encoded)/d(position) (tcnn's prepare_input_gradients); set
 *                               before the ctx scratch is sized -- it changes nvo_ctx_bytes()
 *   "bf16"                      16-bit format of everything the network streams: 0 = fp16 (default), 1 = bfloat16
 *   "compact_output"            (networks with one output) only column 0 exists in memory: [batch] instead of [batch][16]
 *   "recompute_hidden"          the backward recomputes the hidden activations instead of reading stored ones
 *   "grid_compact_live"         (mode 1) the slice-owner items scan a list of the samples whose dL/dy is non-zero.  The
 *                               list is appended per workgroup in dispatch order: with "grid_bwd_runs" the fp32 sums of
 *                               a lane's 8 list-consecutive samples then depend on that order, so results are NOT
 *                               bitwise reproducible (without it only the chunk a sample falls in changes, which
 *                               integer accumulation does not see); "deterministic" switches the list off
 *   "grid_stream_acc_bits"      (mode 3) accumulators of the record pass: 32 only -- two 32-bit fixed-point sums in
 *                               ONE 64-bit word (one LDS atomic per record), 8192-entry bins, scale 2^29 / L1(bin) from
 *                               bounds the scatter delivers with its rank atomics; bitwise reproducible on hashed levels
 *                               (the 64-bit form with 4096-entry bins was removed in round 5)
 *   "deterministic"             bitwise reproducible gradients: every slice / bin has ONE owner work item (no sample
 *                               chunks meeting in float atomics), integer accumulators on every level, no live list;
 *                               networks sum their weight gradient over the workgroups in a fixed order.  Several
 *                               times slower; a debugging aid (EngineConfig.deterministic)
 *   "nonfinite_flag_ptr"        (hash-grid modes 1 and 3 / tile-local) device address of a uint32 the parameter backward
 *                               ORs with 1 when it meets a non-finite dL/dy (0 = off): GradScaler's found_inf raised
 *                               where the 16-bit gradient chain ends, instead of a scan of the gradient buffer
 *   "grid_bwd_runs"             (modes 1 and 3) slice-owner items of DENSE levels scan with run merging: a lane takes 8
 *                               consecutive samples and goes to the LDS accumulators once per run of samples that
 *                               share a cell (consecutive samples are neighbours on a ray); off = per-sample scan
 *   "grid_bwd_batch"            batch size the backward launches will see (0 = unknown): with "grid_bwd_runs" the
 *                               slices are chunked so that all items of a launch are resident at once
 *   "grid_bwd_dense_share"      (with "grid_bwd_batch") chunks of a DENSE slice relative to that even split, in percent
 *                               (default 100; 25..400): 120 when most samples carry a gradient (bf16 gradients, loss
 *                               scale 65536), where dense-level items are the slower kind
 *   "grid_fwd_runs"             the level-major forward (no input gradients requested) walks runs of four consecutive
 *                               samples per thread and gathers only where the cell changes: pays on ray-ordered samples
 *                               of a trained field (inference), costs ~8 % on uniform ones; bit-identical; default 0
 *   "grid_fwd_small_form"       form of the level-major forward: -1 = default; for 5-level grids whose two coarsest levels fit
 *                               the LDS: 0 the first thread-per-(sample, level) kernel, 1 coarse levels from LDS, 2 two
 *                               samples per thread, 3 software-pipelined, 4 instruction-lean + pipelined (the default);
 *                               other grids: 0 the first kernel, else the default.  Every form produces the same bits
 *                               (tests, A/B); with "grid_fwd_runs" the choice is between the first and the lean run forms
 *   "fuse_encoding"             (NetworkWithInputEncoding) the forward evaluates the hash grid inside the MLP kernel
 *   "external_zero"             1 = nvo_bwd does not clear what it accumulates into (MLP weight gradient, atomically
 *                               flushed grid ranges, scale scratch): the caller clears the ranges nvo_bwd_zero_ranges
 *                               lists -- with nvo_zero_ranges, ONE launch for all networks of a training step */
int nvo_set_option(nvo_module_t m, const char* key, int64_t value);
/* The device ranges nvo_bwd(m, ..., dL_dparams) clears before accumulating (see option "external_zero").  Returns
 * the number of ranges written to ptrs_out / bytes_out (<= capacity), or a negative error code. */
int nvo_bwd_zero_ranges(nvo_module_t m, float* dL_dparams, void** ptrs_out, uint64_t* bytes_out, uint32_t capacity);

/* input  : device float [batch][n_input_dims]
 * params : device fp16 [n_params]   (NetworkWithInputEncoding: network weights first, then grid)
 * output : device fp16 [batch][padded_output_dims]
 * ctx    : device scratch of nvo_ctx_bytes() bytes, or NULL for inference (nothing is saved)
 * batch  : a multiple of 16; 0 is a no-op (buffers may be NULL), and nvo_bwd of an empty batch writes zeros to
 *          dL_dparams */
int nvo_fwd(nvo_module_t m, nvo_stream_t stream, uint32_t batch, const float* input,
            const void* params, void* output, void* ctx);
/* dL_doutput: device fp16 [batch][padded_output_dims] (already multiplied by the loss scale)
 * dL_dinput : device float [batch][n_input_dims] or NULL
 * dL_dparams: device float [n_params] or NULL; overwritten (not accumulated into) */
int nvo_bwd(nvo_module_t m, nvo_stream_t stream, uint32_t batch, const float* input,
            const void* params, const void* output, const void* dL_doutput, void* ctx,
            float* dL_dinput, float* dL_dparams);
/* nvo_bwd with the two halves of a NetworkWithInputEncoding backward on two streams: after the network
 * backward, dL_dinput is produced in `stream` order (complete for later work on `stream` when the call returns)
 * while the parameter backward of the input encoding -- the long scatter -- is enqueued on `params_stream`,
 * forked from `stream` at that point.  The CALLER joins: whatever consumes dL_dparams must wait for
 * `params_stream`.  Lets the consumers of dL_dinput (pose / normal gradient chains) run beside the scatter, or (dL_dinput
 * may be NULL) the network backward of one module run beside another module's work that `params_stream` already holds.
 * params_stream == NULL or == stream, or a module without an input encoding: identical to nvo_bwd. */
/* Optimiser step inside the parameter backward.  The tile-local record pass of a hash grid (grid_bwd_mode 3, 32-bit
 * accumulators) holds the finished gradient of every entry of its streamed hashed levels in LDS when it flushes: with
 * nvo_set_fused_adam it applies torch.optim.Adam to those entries right there (same arithmetic as
 * nvo_adam_step_groups_scaled, bit for bit) instead of storing the gradient -- 8 bytes of HBM traffic per parameter less
 * (gradient write + the optimiser's read), and the optimiser launch shrinks to the parameters outside
 * [first_param, first_param + n_params) (nvo_fused_adam_range, relative to the module's first parameter).
 * Preconditions the caller guarantees: the group's overflow flag word is FINAL when the backward of this module runs
 * (every kernel that may raise it precedes it in stream order -- true for the producer flags of a nerfacto step), nothing
 * reads the table between this backward and the end of the step (no gather-form input gradient behind it), and
 * dL_dparams of that range is not consumed by anyone (it is left untouched).  The settings are read at LAUNCH time:
 * set, record / run the backward, then switch off (args = NULL) for launches that want the gradient.
 * The fused step applies NO weight decay (there is no such field): both engines decay MLP weights only, never a hash
 * table (instant-ngp's l2_reg, nvo_adam_group::weight_decay); a caller whose optimiser decays the grid range must keep
 * that range in its own nvo_adam_step_groups launch.  While a step is armed the slice-owner items of the coarse levels
 * run IN stream order in front of the accumulate pass (option grid_stream_overlap is ignored): they may raise the flag. */
typedef struct nvo_fused_adam_args {
    float* params;                 /* fp32 master weights: pointer to THIS MODULE's first parameter */
    void* params_half;             /* 16-bit working copy, same origin */
    float* exp_avg;
    float* exp_avg_sq;
    const float* hyper_dev;        /* device: [0] = learning rate (NULL: lr below) */
    const float* bias_dev;         /* device: {1 - beta1^t, sqrt(1 - beta2^t)} of the next applied step (nvo_opt_commit) */
    const float* loss_scale_dev;   /* device loss scale (NULL: grad_scale below = 1 / loss scale) */
    const uint32_t* skip_flag;     /* device: the group's overflow flag word; non-zero = no step */
    float lr, grad_scale, beta1, beta2, eps;
    uint32_t step;                 /* bias_dev == NULL: bias corrections of step `step` (counted from 1), computed on the host
                                      exactly as nvo_adam_step does */
    /* optional: tcnn EmaOptimizer folded in (nvo_ema_update_dev on the same entries, same arithmetic): the stepped
     * weights are averaged into ema / ema_half right away.  ema_step_dev is only READ (the caller's
     * nvo_ema_update_dev launch over the rest of the parameters advances it). */
    float* ema;                    /* fp32 average, same origin as params; NULL = no averaging */
    void* ema_half;                /* fp16 copy of the average (nullable) */
    float ema_decay;
    const uint32_t* ema_step_dev;  /* device: averages applied so far */
} nvo_fused_adam_args;
int nvo_fused_adam_range(nvo_module_t module, uint64_t* first_param, uint64_t* n_params);
int nvo_set_fused_adam(nvo_module_t module, const nvo_fused_adam_args* args);

int nvo_bwd_fork(nvo_module_t m, nvo_stream_t stream, nvo_stream_t params_stream, uint32_t batch,
                 const float* input, const void* params, const void* output, const void* dL_doutput, void* ctx,
                 float* dL_dinput, float* dL_dparams);

/* Parity/debug: the per-level table geometry and the 8 corner indices the encoder uses.
 * levels_out: host uint32 [n_levels][4] = {offset, size, resolution, hashed}; scales_out: host
 * float [n_levels].  indices_out: device uint32 [n_levels][batch][8]. */
int nvo_grid_describe(nvo_module_t m, uint32_t* levels_out, float* scales_out);
int nvo_grid_indices(nvo_module_t m, nvo_stream_t stream, uint32_t batch, const float* input,
                     uint32_t* indices_out);

/* ------------------------------------------------------------------------------------------------
 * B. Ray generation and sampling (nerfstudio PixelSampler gather, RayGenerator /
 *    Cameras.generate_rays, CameraOptimizer.apply_to_raybundle, UniformLinDispPiecewiseSampler,
 *    Frustums.get_positions + SceneContraction(L-inf); reference call sites
 *    /root/reference/nerf_vo/mapping/nerfstudio_utils.py:286-300,90-107).
 * ---------------------------------------------------------------------------------------------- */
/* ray_indices: device int64 [R][3] = (camera, y, x); intrinsics: device float [F][4] (fx,fy,cx,cy);
 * c2w: device float [F][3][4]; corrections: device float [F][3][4] (exp_map_SE3 of the pose
 * adjustment) or NULL.  Outputs: origins/directions [R][3], directions_norm [R], pixel_area [R]
 * (nullable), cam_idx int32 [R]. */
int nvo_raygen(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics,
               const float* c2w, const float* corrections, float* origins, float* directions,
               float* directions_norm, float* pixel_area, int32_t* cam_idx);
/* exp_map_SE3 of the camera optimizer (CameraOptimizerConfig(mode='SE3'),
 * /root/reference/nerf_vo/mapping/nerfstudio.py:64,209-212): tangent device float [n][6]
 * (translation | rotation) -> out device float [n][3][4] */
int nvo_se3_exp_map(nvo_stream_t stream, uint32_t n, const float* tangent, float* out);
/* mode 0 = exp_map_SE3 (as above), 1 = exp_map_SO3xR3 (nerfstudio's other camera-optimizer mode:
 * Rodrigues rotation with |w|^2 clamped at 1e-4, translation = the tangent's first three entries) */
int nvo_pose_exp_map(nvo_stream_t stream, uint32_t n, const float* tangent, float* out, int mode);
/* Pose-gradient chain of the SE3 camera optimiser (autograd through torch ops in nerfstudio [UPSTREAM];
 * reference use /root/reference/nerf_vo/mapping/nerfstudio.py:64,208-216).  All gradients carry the
 * loss scale of their inputs.
 *  positions_bwd : dx01 [R*S][3] (from nvo_bwd's dL_dinput) -> d_origin/d_dir [R][3], ACCUMULATED
 *                  (selector, (x+2)/4, L-inf contraction Jacobian, reduction over the ray's samples)
 *  sh_bwd_input_f32: dy float [R][16] (colour head d_sh) -> dd01 [R][3]
 *  pose_bwd      : d_origin, d_dir, d_dir01 (nullable, weighted 1/2) -> d_corrections [F][3][4],
 *                  atomically accumulated (caller zeroes)
 *  se3_exp_map_bwd: d_corrections -> d_tangent [n][6] (OVERWRITTEN) through exp_map_SE3, plus
 *                  reg_scale * d/dtangent of (mean|trans| * trans_penalty + mean|rot| * rot_penalty);
 *                  reg_loss (nullable): the regulariser value is atomically added to *reg_loss */
int nvo_positions_bwd(nvo_stream_t stream, uint32_t R, uint32_t S, const float* origins,
                      const float* directions, const float* tbins, const float* dx01, float* d_origin,
                      float* d_dir);
int nvo_sh_bwd_input_f32(nvo_stream_t stream, uint32_t N, uint32_t degree, const float* dirs01,
                         const float* dy, float* dd01);
int nvo_pose_bwd(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics,
                 const float* c2w, const float* d_origin, const float* d_dir, const float* d_dir01,
                 float* d_corrections);
/* same sum for a KNOWN camera count (camera index < n_cameras): when at least 64 rays meet on a camera on average, each
 * workgroup adds into its own camera table in LDS and flushes the non-zero words once -- one global float atomic per
 * workgroup and word instead of one per ray and word (n_cameras <= 1024; otherwise this is nvo_pose_bwd) */
int nvo_pose_bwd_cams(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics,
                      const float* c2w, const float* d_origin, const float* d_dir, const float* d_dir01,
                      float* d_corrections, uint32_t n_cameras);
/* deterministic form: per-ray contributions [R][12] go through per_ray_scratch and are summed per camera in a FIXED
 * order (n_cameras rows of d_corrections) instead of float atomics */
int nvo_pose_bwd_det(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics,
                     const float* c2w, const float* d_origin, const float* d_dir, const float* d_dir01,
                     float* d_corrections, float* per_ray_scratch, uint32_t n_cameras);
int nvo_se3_exp_map_bwd(nvo_stream_t stream, uint32_t n, const float* tangent, const float* d_corrections,
                        float trans_penalty, float rot_penalty, float reg_scale, float* d_tangent,
                        float* reg_loss, int mode);
/* reg_scale_dev (nullable): device float that multiplies reg_scale (the dynamic loss scale of the step) */
int nvo_se3_exp_map_bwd_scaled(nvo_stream_t stream, uint32_t n, const float* tangent, const float* d_corrections,
                               float trans_penalty, float rot_penalty, float reg_scale, float* d_tangent,
                               float* reg_loss, int mode, const float* reg_scale_dev);
/* images: device float [F][H][W][C] -> out [R][C] */
int nvo_gather_pixels(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, uint32_t H,
                      uint32_t W, uint32_t C, const float* images, float* out);
/* sbins/tbins: device float [R][S+1]; jitter: device float [R] in [0,1) or NULL (eval bins) */
int nvo_sample_lindisp(nvo_stream_t stream, uint32_t R, uint32_t S, float near_plane, float far_plane,
                       const float* jitter, float* sbins, float* tbins);

/* Pixel sampler + sampler jitters of one step (PixelSampler + single_jitter): ray_indices [R][3] int64 uniform in
 * [0, extent_dev[c]) (extent_dev = device float {active frames, H, W}), jitter [n_jitter][R] uniform in [0,1).
 * Counter-based: values are a hash of (seed, step_dev[0], element) -- stateless, hipGraph-replay safe. */
int nvo_sample_pixels(nvo_stream_t stream, uint32_t R, uint32_t seed, const float* step_dev, const float* extent_dev,
                      int64_t* ray_indices, float* jitter, uint32_t n_jitter);

/* Fused forms (fewer launches per step; identical arithmetic):
 *   nvo_lindisp_positions = nvo_sample_lindisp + nvo_sample_positions of those bins
 *   nvo_gather_targets    = nvo_gather_pixels of colour [F][H][W][3], depth [F][H][W] (nullable) and
 *                           normals [F][H][W][3] (nullable) + nvo_dirs01 (nullable) in one launch */
int nvo_lindisp_positions(nvo_stream_t stream, uint32_t R, uint32_t S, float near_plane, float far_plane,
                          const float* jitter, const float* origins, const float* directions, float* sbins,
                          float* tbins, float* x01);
int nvo_gather_targets(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, uint32_t H, uint32_t W,
                       const float* images, const float* depths, const float* normals, const float* directions,
                       float* gt_rgb, float* gt_depth, float* gt_normal, float* dirs01);
/* x01: device float [R*S][3] contracted + normalised sample positions; rows outside (0,1)^3 are
 * written as zeros (== nerfstudio's `positions * selector`) */
int nvo_sample_positions(nvo_stream_t stream, uint32_t R, uint32_t S, const float* origins,
                         const float* directions, const float* tbins, float* x01);
/* out[i] = (d[i] + 1) / 2 for n floats (direction encoding input) */
int nvo_dirs01(nvo_stream_t stream, uint32_t n, const float* d, float* out);
/* SH(degree) of (d+1)/2 for R rays -> device fp16 [R][16] (degree 4) */
int nvo_sh_encode(nvo_stream_t stream, uint32_t R, uint32_t degree, const float* dirs01, void* out_half);
/* The whole per-ray prefix of a training step in one launch: nvo_sample_pixels + nvo_raygen + nvo_gather_targets +
 * nvo_sh_encode_t (degree 4) + nvo_lindisp_positions, bit-identical to calling them one after the other
 * (/root/reference/nerf_vo/mapping/nerfstudio_utils.py:286-300: PixelSampler -> RayGenerator -> the model's first
 * sampler level).  c2w: [F] poses of c2w_stride floats each (12 = [3][4], 16 = the top rows of [4][4] matrices, read
 * in place -- no staging copy). */
typedef struct nvo_ray_head_args {
    uint32_t R, S;               /* rays, samples of the first sampler level */
    uint32_t seed, n_jitter;     /* counter-based sampler: see nvo_sample_pixels; jitter [n_jitter][R] */
    const float* step_dev;       /* device float: step counter */
    const float* extent_dev;     /* device float[3]: {active frames, H, W} */
    const float* intrinsics;     /* [F][4] */
    const float* c2w;
    uint32_t c2w_stride;
    const float* corrections;    /* [F][3][4] or NULL */
    uint32_t H, W;
    const float* images;         /* [F][H][W][3] */
    const float* depths;         /* [F][H][W] or NULL */
    const float* normals;        /* [F][H][W][3] or NULL */
    float near_plane, far_plane;
    int64_t* ray_indices;        /* [R][3] */
    float* jitter;
    float* origins;              /* [R][3] */
    float* directions;           /* [R][3] */
    float* directions_norm;      /* [R] */
    float* pixel_area;           /* [R] or NULL */
    int32_t* cam_idx;            /* [R] */
    float* gt_rgb;               /* [R][3] */
    float* gt_depth;             /* [R] (required with depths) */
    float* gt_normal;            /* [R][3] (required with normals) */
    float* dirs01;               /* [R][3] */
    void* sh;                    /* 16-bit [R][16] or NULL */
    int sh_bf16;
    float* sbins;                /* [R][S+1] */
    float* tbins;                /* [R][S+1] */
    float* x01;                  /* [R*S][3] */
} nvo_ray_head_args;
int nvo_ray_head(nvo_stream_t stream, const nvo_ray_head_args* args);
/* nvo_raygen + nvo_gather_pixels (colour, depth) + nvo_dirs01 + nvo_sh_encode (degree 4, fp16) for GIVEN pixel indices
 * in one launch (the occupancy-grid back-end: pyngp.Testbed.frame() draws its pixels itself); same values.
 * depths_cov [F][H][W] (nullable) -> gt_depth_cov [R]: the per-pixel depth variance update_training_images received
 * (/root/reference/nerf_vo/mapping/instant_ngp.py:77-86,93-94), gathered like the depth.
 * R_dev (nullable): device uint32, rays in use (the launch covers R rows, rows from *R_dev on are left alone). */
int nvo_rays_given(nvo_stream_t stream, uint32_t R, const int64_t* ray_indices, const float* intrinsics, const float* c2w,
                   const float* corrections, uint32_t H, uint32_t W, const float* images, const float* depths, float* origins,
                   float* directions, float* directions_norm, float* pixel_area, int32_t* cam_idx, float* gt_rgb,
                   float* gt_depth, float* dirs01, void* sh_half, const float* depths_cov, float* gt_depth_cov,
                   const uint32_t* R_dev);
/* the same launch with extra workgroups that clear up to 24 device ranges (as nvo_zero_ranges): the first launch of a
 * one-graph training step does both */
int nvo_ray_head_zero(nvo_stream_t stream, const nvo_ray_head_args* args, uint32_t n_ranges, void* const* ptrs,
                      const uint64_t* bytes);
/* same with the output format chosen: out_bf16 = 0 -> fp16, 1 -> bfloat16 */
int nvo_sh_encode_t(nvo_stream_t stream, uint32_t R, uint32_t degree, const float* dirs01, void* out, int out_bf16);

/* ------------------------------------------------------------------------------------------------
 * C. Per-ray volume rendering, resampling and losses, one wavefront per ray (nerfstudio
 *    RaySamples.get_weights, PDFSampler, RGB/Accumulation/Depth renderers, interlevel_loss,
 *    distortion_loss, ds_nerf_depth_loss; multipliers from
 *    /root/reference/nerf_vo/mapping/nerfstudio.py:71-82, hooks nerfstudio_utils.py:337-350).
 *    All *_half pointers are device fp16; gradients leave these kernels multiplied by loss_scale.
 * ---------------------------------------------------------------------------------------------- */
typedef struct nvo_weights_pdf_args {
    uint32_t R, S, S_out;        /* S <= 256; S_out == 0 -> weights only */
    const void* pre;             /* fp16 [R*S][pre_stride], density pre-activation in column 0 */
    uint32_t pre_stride;
    const float* x01;            /* [R*S][3] (selector = x01[.][0] > 0) */
    const float* sbins;          /* [R][S+1] spacing-domain bins */
    const float* tbins;          /* [R][S+1] euclidean bins */
    float density_bias;          /* sigma = exp(pre + density_bias) */
    float* sigma;                /* [R*S] or NULL */
    float* weights;              /* [R*S] */
    float anneal, histogram_padding, near_plane, far_plane;
    const float* jitter;         /* [R] or NULL */
    float* sbins_out;            /* [R][S_out+1] */
    float* tbins_out;
    const float* anneal_dev;     /* device float or NULL: overrides `anneal` (hipGraph replay) */
    /* optional fusion of nvo_sample_positions for the resampled level: all three non-NULL -> the contracted
     * positions of the S_out new samples are written too */
    const float* origins;        /* [R][3] */
    const float* directions;     /* [R][3] */
    float* x01_out;              /* [R*S_out][3] */
    int act_bf16;                /* 0: the 16-bit tensors above are fp16 (default), 1: bfloat16 (bf16 MLP mode) */
} nvo_weights_pdf_args;
int nvo_weights_pdf(nvo_stream_t stream, const nvo_weights_pdf_args* args);

typedef struct nvo_main_loss_args {
    uint32_t R, S;               /* S <= 64 */
    const void* pre;             /* fp16 [R*S][pre_stride] column 0 */
    uint32_t pre_stride;
    const void* rgb;             /* fp16 [R*S][rgb_stride] columns 0..2 */
    uint32_t rgb_stride;
    const float* x01;
    const float* sbins;
    const float* tbins;
    float density_bias;
    const float* gt_rgb;         /* [R][3] */
    const float* gt_depth;       /* [R] z-depth or NULL */
    const float* directions_norm;/* [R] */
    float rgb_mult, distortion_mult, depth_mult, depth_sigma;
    float inv_rays;              /* 1 / global ray count (mean reductions; multi-GPU aware) */
    float depth_level_div;       /* 1 / number of levels the depth loss averages over */
    float loss_scale;
    float* out_rgb;              /* [R][3] */
    float* out_depth;            /* [R] median depth */
    float* out_expected_depth;   /* [R] or NULL */
    float* out_accumulation;     /* [R] */
    float* weights;              /* [R*S] or NULL */
    float* losses;               /* [64][8] shards; slots 0..2 = rgb, distortion, depth: atomically accumulated,
                                    the caller zeroes the buffer and sums the 64 shards */
    void* dpre;                  /* fp16 [R*S][dpre_stride] column 0; NULL -> inference, no losses */
    uint32_t dpre_stride;
    void* drgb;                  /* fp16 [R*S][drgb_stride]: cols 0..2 gradient, others zeroed */
    uint32_t drgb_stride;
    /* analytic normals (nerfacto predict_normals + the reference's monosdf normal-loss hook,
     * /root/reference/nerf_vo/mapping/nerfstudio_utils.py:337-350).  All nullable / zero -> disabled. */
    const float* dsigma_dx;      /* [R*S][3] d(density pre-activation)/d(x01) from nvo_bwd(dL_dparams=NULL) with
                                    dL_doutput = dsigma_scale * e_0; per-sample normal = -normalize(.) and is a
                                    CONSTANT of the graph (torch.autograd.grad without create_graph) */
    float dsigma_inv_scale;      /* 1 / dsigma_scale, applied before the 1e-12 normalisation clamp */
    const float* gt_normal;      /* [R][3] target in the (n+1)/2 colour space the dataset hands out, or NULL */
    float normal_mult;           /* normal_loss_mult (5e-6 in the reference config) */
    float* out_normals;          /* [R][3] NormalsRenderer (safe-normalised) -> NormalsShader (n+1)/2, or NULL */
                                 /* loss slot 6 of the shard receives normal_mult * monosdf_normal_loss */
    int act_bf16;                /* 0: pre / rgb / dpre / drgb are fp16 (default), 1: bfloat16 (bf16 MLP mode) */
    const float* loss_scale_dev; /* nullable: device float that REPLACES loss_scale (dynamic loss scaling: the
                                    GradScaler state lives on the device so that a captured step stays valid) */
    uint32_t* nonfinite_flag;    /* nullable: OR-ed with 1 when a gradient stored to dpre / drgb overflows the 16-bit
                                    format (inf / NaN / > 65504 in fp16): the overflow check at its source */
    uint8_t* tile_live;          /* nullable (training, S % 16 == 0, S <= 64): [R*S/16] one byte per 16-sample tile --
                                    bit 0: some stored drgb value of the tile is non-zero, bit 1: some stored dpre value
                                    is.  The MLP backwards behind this kernel walk only the live tiles
                                    (nvo_color_args::tile_live, module option "bwd_tile_live_ptr"); slot 7 of the loss
                                    shards then receives the number of tiles with a non-zero byte */
} nvo_main_loss_args;
int nvo_main_render_loss(nvo_stream_t stream, const nvo_main_loss_args* args);

typedef struct nvo_prop_loss_args {
    uint32_t R, S, S_main;       /* S <= 256, S_main <= 64 */
    const void* pre;             /* fp16 [R*S][pre_stride] column 0 */
    uint32_t pre_stride;
    const float* x01;
    const float* sbins;
    const float* tbins;
    const float* sbins_main;     /* [R][S_main+1] */
    const float* weights_main;   /* [R*S_main] */
    float density_bias;
    const float* gt_depth;       /* nullable */
    const float* directions_norm;
    float interlevel_mult, depth_mult, depth_sigma;
    float inv_rays, depth_level_div, loss_scale;
    float* losses;               /* [64][8] shards (same buffer, base + 3): slots 0..1 = interlevel, depth */
    void* dpre;                  /* fp16 [R*S][dpre_stride]: column 0 gradient, others zeroed; NULL = loss VALUES only */
    uint32_t dpre_stride;
    int act_bf16;                /* 0: pre / dpre are fp16 (default), 1: bfloat16 (bf16 MLP mode) */
    const float* loss_scale_dev; /* nullable: device float that REPLACES loss_scale */
    uint32_t* nonfinite_flag;    /* nullable: as in nvo_main_loss_args */
} nvo_prop_loss_args;
int nvo_prop_loss(nvo_stream_t stream, const nvo_prop_loss_args* args);
/* both proposal levels of a step in ONE launch (the two calls are independent of each other; same results as two
 * nvo_prop_loss calls -- the loss terms are added to the same shards with float atomics either way) */
int nvo_prop_loss_pair(nvo_stream_t stream, const nvo_prop_loss_args* args0, const nvo_prop_loss_args* args1);

/* ------------------------------------------------------------------------------------------------
 * D. NerfactoField colour head (nerfstudio fields/nerfacto_field.py get_outputs: concat
 *    [SH(d) | geo features | appearance embedding] -> tcnn.Network 63->64->64->3 sigmoid).  The
 *    64-wide input row is assembled inside the MLP kernel, never written to HBM.
 * ---------------------------------------------------------------------------------------------- */
typedef struct nvo_color_args {
    uint32_t R, S;               /* R*S multiple of 16 */
    const void* sh;              /* fp16 [R][16] */
    const void* base_out;        /* fp16 [R*S][16]: col 0 density pre-activation, cols 1..15 geo */
    const void* embedding;       /* fp16 [F][32] */
    const int32_t* cam_idx;      /* [R] or NULL -> embedding row 0 for every ray (eval: mean row) */
    const void* weights;         /* fp16 colour MLP weights: [64][64], [64][64], [16][64] */
    void* rgb;                   /* fp16 [R*S][16], sigmoid rgb in cols 0..2 */
    void* hidden;                /* fp16 [2][R*S][64] or NULL: with NULL the forward stores no activations and the
                                    backward recomputes both hidden layers (bit-identical; 256 B/sample less traffic
                                    each way) */
    /* backward only */
    const void* drgb;            /* fp16 [R*S][16] (loss-scaled) */
    void* d_base_out;            /* fp16 [R*S][16]: cols 1..15 written */
    float* d_embedding;          /* [F][32] accumulated; nullable */
    float* d_sh;                 /* [R][16] accumulated; nullable */
    float* d_weights;            /* accumulated (caller zeroes) */
    int act_bf16;                /* 0: every "fp16" tensor above is fp16 (v_mfma_..._f16), 1: all of them are bfloat16
                                    and the network runs on v_mfma_f32_16x16x16_bf16 (BASELINE configs[4]) */
    /* deterministic mode (backward; det_scratch NULL = off): the weight gradient and the per-camera embedding / per-ray
     * SH-direction gradients are summed in FIXED orders through this caller-owned scratch instead of float atomics --
     * bitwise reproducible results, a few launches more.  Needs S % 16 == 0 and cam_idx. */
    void* det_scratch;           /* nvo_color_det_scratch_bytes(R, S) */
    uint64_t det_scratch_bytes;
    uint32_t n_cameras;          /* rows of d_embedding (deterministic mode only) */
    uint32_t* nonfinite_flag;    /* backward; nullable: OR-ed with 1 when a weight-gradient total of the head is not finite
                                    (an overflow inside its 16-bit chain; d_embedding / d_sh are non-finite only with it) */
    float* dw_replicas;          /* backward; nullable: n_dw_replicas zeroed copies [r][9216] of d_weights the workgroups spread */
    uint32_t n_dw_replicas;      /* their adds over (module option "dw_replicas"); fold with nvo_fold_replicas */
    const uint8_t* tile_live;    /* backward; nullable: nvo_main_loss_args::tile_live of the kernel that wrote drgb -- tiles
                                    without bit 0 are not evaluated, their d_base_out columns 1..15 are stored as zeros */
    const float* tile_live_count;/* nullable: the loss shards' slot 7 (64 floats, 8 apart; nvo_main_loss_args::tile_live) --
                                    while 3/4 of the tiles or more are live the kernel does not build its list */
} nvo_color_args;
int nvo_nerfacto_color_fwd(nvo_stream_t stream, const nvo_color_args* args);
int nvo_nerfacto_color_bwd(nvo_stream_t stream, const nvo_color_args* args);
uint64_t nvo_color_det_scratch_bytes(uint32_t R, uint32_t S);

/* ------------------------------------------------------------------------------------------------
 * F. Cascaded occupancy grid (instant-ngp testbed: generate_training_samples_nerf,
 *    ema_grid_samples_nerf, grid_to_bitfield, bitfield_max_pool -- the back-end behind
 *    `mapping_module: 'instant-ngp'`, /root/reference/nerf_vo/mapping/instant_ngp.py:33-50,104-105;
 *    nerfacc 0.5.2 traverse_grids / OccGridEstimator, /root/reference/requirements.txt:113).
 *    Normalised frame: cascade 0 covers [0,1]^3, cascade k covers [0.5 - 2^(k-1), 0.5 + 2^(k-1)]^3;
 *    128^3 cells per cascade in Morton order; bitfield = device uint8 [n_levels][128^3 / 8].
 * ---------------------------------------------------------------------------------------------- */
/* DDA march of R rays (unit directions) with deterministic packing: counts[R], offsets[R+1]
 * (exclusive scan, offsets[R] = total), then (ray_idx, t, dt)[capacity] written at the offsets.  Rays
 * whose samples would not fit get count 0 while offsets keeps their slot range: offsets[R] is the number of samples the
 * march FOUND (what the adaptive ray batch measures), and a ray with counts[r] == 0 < offsets[r+1] - offsets[r] is a
 * dropped one.  jitter: device float [R] in [0,1) or NULL.
 * Step size dt = clamp(t * cone_angle, sqrt(3)/1024, sqrt(3)/1024 * 1024).
 * scratch: CALLER-owned device staging area of at least nvo_occ_march_scratch_bytes(R) bytes (ray-major runs of the
 * single march; the native side allocates nothing, so a captured launch never holds a pointer it could lose). */
uint64_t nvo_occ_march_scratch_bytes(uint32_t R);
int nvo_occ_march(nvo_stream_t stream, uint32_t R, const float* origins, const float* directions,
                  const uint8_t* bitfield, int n_levels, float cone_angle, float t_near, const float* jitter,
                  uint32_t capacity, uint32_t* counts, uint32_t* offsets, int32_t* ray_idx, float* t_out,
                  float* dt_out, void* scratch, uint64_t scratch_bytes);
/* The same march in ROUNDS (inference with early termination: `render_min_transmittance`, which the reference sets to 1e-4,
 * /root/reference/evaluation/nerf_renderer.py:154): ray r takes up its progression at candidate t_resume[r] (NULL: a fresh
 * march from t_near with jitter; a negative entry: the ray is not alive, 0 samples), accepts at most max_new samples and
 * writes to t_next[r] (nullable) the candidate it stopped in front of, -1 once it has left the scene box.  The step
 * recurrence depends on t alone, so the rounds of a ray concatenate to exactly the samples of one uninterrupted march. */
int nvo_occ_march_resume(nvo_stream_t stream, uint32_t R, const float* origins, const float* directions,
                         const uint8_t* bitfield, int n_levels, float cone_angle, float t_near, const float* jitter,
                         uint32_t capacity, uint32_t* counts, uint32_t* offsets, int32_t* ray_idx, float* t_out,
                         float* dt_out, void* scratch, uint64_t scratch_bytes, const float* t_resume, uint32_t max_new,
                         float* t_next);
/* The two halves of nvo_occ_march_resume as calls of their own (the training step of the occupancy-grid back-end packs a
 * ray's run twice: everything the march found for the pass that finds where each ray ends, then the samples in front of
 * that point for the batch that is trained on).
 *  nvo_occ_march_runs : the march alone -- counts[r] and the ray-major runs in `scratch` ([R][1024] (t, dt) pairs).
 *  nvo_occ_pack       : exclusive scan of counts_in[R] with the capacity rule (a ray whose samples would pass `capacity`
 *                       gets counts_out 0 and keeps its slot range in offsets; counts_out may alias counts_in), then the
 *                       first counts_out[r] samples of every run are copied to (ray_idx, t, dt) at offsets[r].
 *                       totals (nullable): device uint32[2] = {sum of counts_in, min(that, capacity)}.
 * R_dev (nullable): device uint32, the number of rays in use -- the launch covers R rows, rows from *R_dev on are left
 * alone (a captured step is replayed while the adaptive ray batch moves).
 * run_offset: a march in rounds appends a later round behind the samples the earlier rounds left in the ray's run (the
 * march writes from run[run_offset] on, the pack reads from there): run_offset + max_new <= 1024. */
int nvo_occ_march_runs(nvo_stream_t stream, uint32_t R, const float* origins, const float* directions,
                       const uint8_t* bitfield, int n_levels, float cone_angle, float t_near, const float* jitter,
                       uint32_t* counts, void* scratch, uint64_t scratch_bytes, const float* t_resume, uint32_t max_new,
                       float* t_next, const uint32_t* R_dev, uint32_t run_offset);
int nvo_occ_pack(nvo_stream_t stream, uint32_t R, const uint32_t* counts_in, uint32_t capacity, uint32_t* counts_out,
                 uint32_t* offsets, uint32_t* totals, const void* scratch, uint64_t scratch_bytes, int32_t* ray_idx,
                 float* t_out, float* dt_out, const uint32_t* R_dev, uint32_t run_offset);
/* nvo_occ_pack in ONE launch (scan by decoupled look-back between 16- or 64-ray workgroups, copy, and -- x01 != NULL -- the
 * network input of every copied sample, nvo_ngp_positions' values: clamp((o + t d - aabb_lo) / (aabb_hi - aabb_lo), 0, 1)).
 * Same counts_out / offsets / totals / ray_idx / t / dt as nvo_occ_pack, bit for bit.  `state`: caller-owned device block of
 * nvo_occ_pack_state_bytes() bytes, 8-byte aligned, ZEROED once when it is allocated and never touched by the caller again
 * (it carries the launches' epoch); launches that share a block must be ordered on one stream. */
uint64_t nvo_occ_pack_state_bytes(void);
int nvo_occ_pack_fused(nvo_stream_t stream, uint32_t R, const uint32_t* counts_in, uint32_t capacity, uint32_t* counts_out,
                       uint32_t* offsets, uint32_t* totals, const void* scratch, uint64_t scratch_bytes, int32_t* ray_idx,
                       float* t_out, float* dt_out, const uint32_t* R_dev, uint32_t run_offset, void* state,
                       const float* origins, const float* directions, float aabb_lo, float aabb_hi, float* x01,
                       uint32_t rays_per_group /* 16 | 64: rays a workgroup owns (speed only) */);
/* grid: device float [n_levels][128^3]; fresh (nullable): same shape, the new optical thickness per
 * cell -> grid = grid < 0 ? grid : max(grid * decay, fresh); then bitfield = grid > min(threshold,
 * mean(max(grid[0], 0))) and every coarser cascade ORs in the 2x2x2 max-pool of the next finer one.
 * scratch8: 8 bytes of device scratch. */
int nvo_occ_update(nvo_stream_t stream, int n_levels, float* grid, const float* fresh, float decay,
                   float threshold, uint8_t* bitfield, void* scratch8);
/* centres (or jittered points, jitter device float [128^3][3]) of cascade `level` cells, Morton order */
int nvo_occ_cell_positions(nvo_stream_t stream, int level, const float* jitter, float* positions);
/* Cells no training camera sees are taken out of training (SURVEY.md section 2.4 K16 `mark_untrained_density_grid`
 * [UPSTREAM instant-ngp testbed_nerf.cu]; reached through pyngp's frame(), /root/reference/nerf_vo/mapping/instant_ngp.py:
 * 104-105, at step 0 and whenever training images were added): a cell of any cascade stays trainable iff one of its
 * eight corners lies in front of one of the first n_images cameras (cosine to the viewing axis >= 1e-4) and projects
 * strictly inside its H x W image -- grown on every side by `margin` projected cell diagonals (margin * f * sqrt(3) *
 * cell size / depth pixels; 0 = upstream's rule, which also excludes cells the frustum clips without containing one of
 * their corners: on small images that blanks the border rows of a view no other camera shares).  intrinsics [F][4] = (fx, fy, cx, cy) in pixels, c2w [F][3][4] (OpenGL axes, the
 * engine's normalised frame; the cameras nvo_rays_given takes).  A trainable cell that was marked becomes 0, a cell without
 * a view -1 (never occupied: nvo_occ_update keeps negative values, nvo_occ_sample_cells passes them over); others keep
 * their value. */
int nvo_occ_mark_untrained(nvo_stream_t stream, int n_levels, float* grid, uint32_t n_images, const float* intrinsics,
                           const float* c2w, uint32_t H, uint32_t W, float margin);
/* Refresh samples of the density grid PAST the first 256 steps ("all cells first 256 steps";
 * [UPSTREAM instant-ngp Testbed::update_density_grid_nerf / generate_grid_samples_nerf_nonuniform] -- the call the
 * reference reaches through pyngp's frame(), /root/reference/nerf_vo/mapping/instant_ngp.py:104-105): samples
 * first .. first + n - 1 of a pass of n_total.  Sample i takes a cascade uniformly at random and the first of ten
 * candidate cells ((i + step * n_total) * 56924617 + j * 19349663 + 96925573) mod 128^3 (32-bit wrap-around, j = 0..9)
 * whose grid value exceeds `thresh` (the tenth stays when none does; thresh < 0 takes every trained cell = the uniform
 * pass, thresh = the occupancy threshold = the pass over occupied cells), then a uniform point inside the cell.
 * x01 [n][3]: the point as the density network's input (position in the scene box [aabb_lo, aabb_hi]^3, clamped to
 * [0, 1]); cell_idx [n]: level * 128^3 + Morton index.  Random numbers are hashes of (seed, stream_id, step, i). */
int nvo_occ_sample_cells(nvo_stream_t stream, uint32_t n, uint32_t first, uint32_t n_total, uint32_t step, uint32_t seed,
                         uint32_t stream_id, int n_levels, const float* grid, float thresh, float aabb_lo, float aabb_hi,
                         float* x01, uint32_t* cell_idx);

/* Occupancy-grid ("instant-ngp") trainer pieces on PACKED samples (capacity slots, ray_idx < 0 = empty):
 *  ngp_positions      : x01 = clamp((o + t d - aabb_lo) / (aabb_hi - aabb_lo), 0, 1) per slot
 *  ngp_rgb_fwd / bwd  : rgb head 32 -> 64 -> 64 -> 3 on [density-net output 16 | SH16(ray)] (linear
 *                       output; the logistic is applied by the compositing kernel, as upstream)
 *  ngp_composite_loss : per ray (counts/offsets from nvo_occ_march) front-to-back compositing with
 *                       density = exp(pre), rgb = sigmoid(y), background blend; L2 rgb + L2 depth
 *                       losses; per-sample gradients (d_rgb_out fp16 rows, d_density_pre float); rays the march
 *                       dropped at the packed capacity are rendered (background) but add nothing to the losses
 *  ngp_thickness      : exp(pre) * sqrt(3)/1024 * 2^level for the density-grid update */
typedef struct nvo_ngp_rgb_args {
    uint32_t capacity;           /* multiple of 16 */
    const void* sh;              /* fp16 [R][16] */
    const void* density_out;     /* fp16 [capacity][16] */
    const int32_t* ray_idx;      /* [capacity] */
    const void* weights;         /* fp16 [64][32], [64][64], [16][64] */
    void* rgb_out;               /* fp16 [capacity][16] */
    void* hidden;                /* fp16 [2][capacity][64] or NULL: with NULL the forward stores no activations and the
                                    backward recomputes both hidden layers (bit-identical) */
    const void* d_rgb_out;       /* bwd: fp16 [capacity][16] */
    void* d_density_out;         /* bwd: fp16 [capacity][16] (all columns written) */
    const float* d_density_pre;  /* bwd: [capacity], added into column 0; nullable */
    float* d_weights;            /* bwd: accumulated */
    uint32_t* nonfinite_flag;    /* bwd; nullable: OR-ed with 1 when a weight-gradient total of the head is not finite */
    float* dw_replicas;          /* bwd; nullable: n_dw_replicas zeroed copies of d_weights (see nvo_color_args) */
    uint32_t n_dw_replicas;
} nvo_ngp_rgb_args;
int nvo_ngp_rgb_fwd(nvo_stream_t stream, const nvo_ngp_rgb_args* args);
int nvo_ngp_rgb_bwd(nvo_stream_t stream, const nvo_ngp_rgb_args* args);

typedef struct nvo_ngp_loss_args {
    uint32_t R, capacity;
    const uint32_t* counts;      /* [R] */
    const uint32_t* offsets;     /* [R+1] */
    const int32_t* ray_idx;      /* [capacity] (training only) */
    const float* t;              /* [capacity] */
    const float* dt;
    const void* density_out;     /* fp16 [capacity][density_stride], pre-activation in column 0 */
    uint32_t density_stride;
    const void* rgb_out;         /* fp16 [capacity][rgb_stride], pre-sigmoid rgb in columns 0..2 */
    uint32_t rgb_stride;
    const float* background;     /* [R][3] or NULL (black) */
    const float* gt_rgb;         /* [R][3] */
    const float* gt_depth;       /* [R] or NULL */
    const float* directions_norm;/* [R] or NULL */
    float rgb_mult, depth_mult, inv_rays, loss_scale;
    float* out_rgb;              /* [R][3] nullable */
    float* out_depth;            /* [R] nullable (expected ray distance) */
    float* out_accumulation;     /* [R] nullable */
    float* losses;               /* [64][8] shards: slot 0 rgb, slot 1 depth */
    void* d_rgb_out;             /* fp16 [capacity][d_rgb_stride]; NULL -> inference */
    uint32_t d_rgb_stride;
    float* d_density_pre;        /* [capacity] */
    /* inference in rounds (nvo_occ_march_resume): the optical depth the ray has gathered in earlier rounds (NULL: none),
     * where to leave the total (nullable), and whether out_rgb / out_depth / out_accumulation are ADDED to (a later round)
     * or overwritten.  Training ignores the three. */
    const float* carry_in;       /* [R] */
    float* carry_out;            /* [R] */
    uint32_t accumulate_outputs;
    /* training: samples a ray reaches with a transmittance below this get EXACTLY zero gradients (0 = off) -- upstream's
     * loss kernel stops a ray there and trains on the samples in front (`if (T < EPSILON) break`, EPSILON = 1e-4); the
     * zero rows then cost the fused-MLP and hash-grid backwards nothing (they skip zero-gradient samples) */
    float train_min_transmittance;
    /* [R] or NULL: variance of the ray's depth target (NeRF-SLAM's covariance-weighted depth term, spec by paper
     * [UPSTREAM]: L_D = ||D - D*||^2_Sigma = (D - D*)^2 / Sigma_D per pixel).  The depth residual of ray r is weighted by
     * 1 / gt_depth_cov[r]; a variance that is not a positive finite number switches the ray's depth term off.  NULL or
     * all ones: the plain L2 term, bit for bit.  The reference supplies it on every instant-ngp configuration
     * (/root/reference/nerf_vo/enhancement/enhancement_module.py:105-111, configs/nerf_slam_*.yaml: compute_covariances) */
    const float* gt_depth_cov;
    /* training on the samples in front of the point where a ray's transmittance falls below the threshold
     * (nvo_ngp_count_alive; counts / offsets then describe the COMPACTED batch): ray_state [R] (nullable) -- 1: the ray was
     * cut (no background term), 2: it was dropped where the march was packed (no trace in the losses). */
    const uint32_t* ray_state;
    /* R_dev (nullable): device uint32, rays in use; the launch covers R rows, inv_rays is then taken as
     * 1 / (*R_dev * max(world_size, 1)) */
    const uint32_t* R_dev;
    uint32_t world_size;
} nvo_ngp_loss_args;
/* Where each ray ends for training: kept[r] = index of the first of its packed samples reached with a transmittance
 * T = exp(-sum of min(exp(pre) dt, 128) over the samples in front) below min_transmittance (its count when none is),
 * state[r] = 0 (kept everything) | 1 (cut) | 2 (the ray was dropped where the march was packed: kept 0)
 * [UPSTREAM instant-ngp compute_loss_kernel_train_nerf `if (T < EPSILON) break`, EPSILON = 1e-4; the testbed the
 * reference drives through pyngp's frame(), /root/reference/nerf_vo/mapping/instant_ngp.py:104-105]. */
typedef struct nvo_ngp_alive_args {
    uint32_t R;
    const uint32_t* counts;      /* [R] packed samples per ray (0 for a dropped ray) */
    const uint32_t* offsets;     /* [R+1] */
    const float* dt;             /* [capacity] */
    const void* density_out;     /* fp16 [capacity][density_stride], pre-activation in column 0 (stride 1: compact) */
    uint32_t density_stride;
    float min_transmittance;
    uint32_t* kept;              /* [R] */
    uint32_t* state;             /* [R] */
    const uint32_t* R_dev;       /* nullable: device ray count */
    /* In ROUNDS (the march of nvo_occ_march_runs with max_new / t_next / t_resume): a round looks at the samples its march
     * added.  resume_in (nullable) [R]: rays with a negative entry are not part of this round (kept / state stay);
     * carry_in (nullable) [R]: optical depth the ray gathered in earlier rounds; kept_base: samples of the earlier rounds
     * (kept = kept_base + index inside this round).  A ray that is neither cut nor dropped and whose march stopped at
     * its sample budget (t_next[r] >= 0) goes on: resume_out[r] = t_next[r], carry_out[r] = its optical depth so far;
     * every other ray gets resume_out[r] = -1.  t_next / resume_out / carry_out NULL: a single round. */
    const float* resume_in;
    const float* carry_in;
    uint32_t kept_base;
    const float* t_next;
    float* resume_out;
    float* carry_out;
} nvo_ngp_alive_args;
int nvo_ngp_count_alive(nvo_stream_t stream, const nvo_ngp_alive_args* args);
/* nvo_ngp_positions with the number of slots in use on the device (n_live, nullable): slots from the next multiple of
 * 4096 on are left alone; nvo_ngp_positions_bwd with a device ray count (rows from *R_dev on receive zeros). */
int nvo_ngp_positions_live(nvo_stream_t stream, uint32_t capacity, const int32_t* ray_idx, const float* t,
                           const float* origins, const float* directions, float aabb_lo, float aabb_hi, float* x01,
                           const uint32_t* n_live);
int nvo_ngp_positions_bwd_dev(nvo_stream_t stream, uint32_t R, uint32_t capacity, const int32_t* counts,
                              const int32_t* offsets, const float* t, const float* origins, const float* directions,
                              float aabb_lo, float aabb_hi, const float* dx01, float* d_origin, float* d_dir,
                              const uint32_t* R_dev);
int nvo_ngp_positions(nvo_stream_t stream, uint32_t capacity, const int32_t* ray_idx, const float* t,
                      const float* origins, const float* directions, float aabb_lo, float aabb_hi, float* x01);
/* Extrinsics optimisation of the occupancy-grid back-end (`optimize_extrinsics = True`,
 * /root/reference/nerf_vo/mapping/instant_ngp.py:47): dL/dx01 [capacity][3] of the packed samples (from
 * nvo_bwd's dL_dinput) -> per-ray dL/dorigin, dL/ddirection [R][3] (overwritten), which nvo_pose_bwd /
 * nvo_se3_exp_map_bwd turn into the camera-offset gradient. */
int nvo_ngp_positions_bwd(nvo_stream_t stream, uint32_t R, uint32_t capacity, const int32_t* counts,
                          const int32_t* offsets, const float* t, const float* origins, const float* directions,
                          float aabb_lo, float aabb_hi, const float* dx01, float* d_origin, float* d_dir);
int nvo_ngp_composite_loss(nvo_stream_t stream, const nvo_ngp_loss_args* args);
int nvo_ngp_thickness(nvo_stream_t stream, uint32_t n, const void* density_out, uint32_t stride, int level,
                      float* out);
/* The same for scattered samples: fresh[cell_idx[i]] = max(fresh[cell_idx[i]], thickness_i) with the cascade taken from the
 * cell index (atomic; fresh zero-initialised by the caller) [UPSTREAM splat_grid_samples_nerf_max_nearest_neighbor]. */
int nvo_ngp_thickness_splat(nvo_stream_t stream, uint32_t n, const void* density_out, uint32_t stride,
                            const uint32_t* cell_idx, float* fresh);
int nvo_fill_i32(nvo_stream_t stream, uint32_t n, int32_t* ptr, int32_t value);

/* ------------------------------------------------------------------------------------------------
 * E. Optimiser (torch.optim.Adam as configured at /root/reference/nerf_vo/mapping/nerfstudio.py:84-100
 *    + GradScaler's skip-on-non-finite from mixed_precision=True, nerfstudio.py:59).
 * ---------------------------------------------------------------------------------------------- */
/* One fused pass over params[n]: updates exp_avg / exp_avg_sq / params and, if params_half != NULL,
 * the fp16 working copy.  grads are multiplied by grad_scale first (1/loss_scale).  step counts from 1.
 * skip_flag: device uint32 or NULL; a non-zero value makes the call a no-op. */
int nvo_adam_step(nvo_stream_t stream, uint64_t n, float* params, void* params_half,
                  const void* grads, int grads_are_half, float* exp_avg, float* exp_avg_sq, float lr,
                  float beta1, float beta2, float eps, uint32_t step, float grad_scale, float weight_decay,
                  const uint32_t* skip_flag, const float* hyper_dev);
/* The same update for up to 4 parameter groups of ONE flat buffer in one launch (the optimisers of
 * /root/reference/nerf_vo/mapping/nerfstudio.py:84-100 differ only in learning rate and step count).
 * offset / n are in elements of the flat buffers; hyper_dev: optional device float[3] = {lr, 1 - beta1^t,
 * sqrt(1 - beta2^t)} that overrides lr / step (graph replay). */
typedef struct nvo_adam_group {
    uint64_t offset, n;
    float lr;
    uint32_t step;
    const float* hyper_dev;
    /* bias_dev (nullable): device float[2] = {1 - beta1^t, sqrt(1 - beta2^t)} of the group's NEXT applied step, kept
     * by nvo_opt_commit next to its applied-step counter (`step` and hyper_dev[1..2] are then ignored): the counter
     * advances iff the group was not skipped -- torch.optim.Adam's state['step'] under GradScaler.step, which does not
     * count skipped steps. */
    const float* bias_dev;
    /* which word of skip_flags belongs to this group: the group's index in `groups` unless flag_slot_set != 0 (a step
     * that runs its groups in two launches keeps ONE flag word per group that way) */
    uint32_t flag_slot;
    uint32_t flag_slot_set;
    /* L2 weight decay of this group (folded into the gradient); used instead of the call's weight_decay when
     * weight_decay_set != 0 -- instant-ngp decays its MLP weights and leaves the hash table alone, in one launch */
    float weight_decay;
    uint32_t weight_decay_set;
} nvo_adam_group;
int nvo_adam_step_groups(nvo_stream_t stream, uint32_t n_groups, const nvo_adam_group* groups, float* params,
                         void* params_half, const void* grads, int grads_are_half, float* exp_avg, float* exp_avg_sq,
                         float beta1, float beta2, float eps, float grad_scale, float weight_decay,
                         const uint32_t* skip_flags);
/* skip_flags: device uint32 [n_groups] or NULL; group i is a no-op when skip_flags[i] != 0 -- GradScaler.step
 * decides per optimiser (nerfstudio's optimizer_scaler_step_all calls it once per parameter group).
 *
 * nvo_nonfinite_flag over up to 4 ranges (element offsets / sizes, host arrays) of one gradient buffer in one
 * launch: flags[i] (device uint32 [n_ranges], reset first) is raised iff range i holds an inf / NaN. */
int nvo_nonfinite_flag_ranges(nvo_stream_t stream, uint32_t n_ranges, const uint64_t* offsets, const uint64_t* sizes,
                              const void* grads, int grads_are_half, uint32_t* flags);
/* The same without resetting the flag words first (they were cleared earlier, e.g. by the step's nvo_zero_ranges):
 * one launch less in front of the optimiser. */
/* The same over up to 8 spans with an EXPLICIT flag word each (flags[slots[i]] |= 1; several spans may share a word):
 * the step whose overflow flags are raised at the source scans the small non-grid ranges of its groups -- fused-MLP
 * weight gradients, embedding, poses -- where every 16-bit overflow INSIDE the backward chain ends up (dW = dZ x H of
 * the layer that overflowed), see EngineConfig.producer_overflow_flags. */
int nvo_nonfinite_flag_spans_or(nvo_stream_t stream, uint32_t n_spans, const uint64_t* offsets, const uint64_t* sizes,
                                const uint32_t* slots, const void* grads, int grads_are_half, uint32_t* flags);
int nvo_nonfinite_flag_ranges_or(nvo_stream_t stream, uint32_t n_ranges, const uint64_t* offsets, const uint64_t* sizes,
                              const void* grads, int grads_are_half, uint32_t* flags);
/* grads: device float[n], or device fp16[n] when grads_are_half != 0 (the buffer a compressed
 * all-reduce leaves behind: no cast-back pass). */
/* hyper_dev (nullable): device float[3] = {lr, 1 - beta1^step, sqrt(1 - beta2^step)} overriding the
 * by-value arguments, so that a captured hipGraph of the step can be replayed with new values. */
/* dst[i] = host_values[i] for n <= 16 floats; the values travel as kernel arguments (no host buffer
 * has to stay alive), used to refresh per-step scalars ahead of a graph replay. */
int nvo_write_floats(nvo_stream_t stream, float* dst, uint32_t n, const float* host_values);
/* *flag = any(!isfinite(grads)) */
int nvo_nonfinite_flag(nvo_stream_t stream, uint64_t n, const void* grads, int grads_are_half, uint32_t* flag);
/* same check OR-ed into an already initialised flag (several disjoint gradient ranges, one flag) */
int nvo_nonfinite_flag_or(nvo_stream_t stream, uint64_t n, const void* grads, int grads_are_half, uint32_t* flag);
/* fp32 -> bfloat16 (round to nearest even; inf / NaN preserved): the compressed form of the gradient exchange.
 * Everywhere a `grads_are_half` argument appears, 0 = fp32, 1 = fp16, 2 = bfloat16. */
int nvo_cast_bf16(nvo_stream_t stream, uint64_t n, const float* src, void* dst_bf16);
int nvo_cast_half(nvo_stream_t stream, uint64_t n, const float* src, void* dst_half);
/* Sharded gradient exchange (multi-GPU: reduce-scatter -> Adam on this rank's 1/world slice -> all-gather of the 16-bit
 * working copy).  src[0, n) is cast to the wire format (1 = fp16, 2 = bf16) as `world` chunks of n / world elements,
 * each followed by `pad` FLAG slots: wire16[(i / per) * (per + pad) + i % per].  *flag (device uint32, OR-ed) is raised
 * when src holds an inf / NaN or a value the SUM over `world` ranks could not carry on the wire (fp16: |v| > 65504 / world
 * -- nothing scans the reduced shard, so a finite sum must follow from the local verdicts; bf16: fp32's range), and every
 * pad slot receives *flag ? 1 : 0 -- after the SUM reduce-scatter of the wire
 * buffer, rank r reads the number of ranks that overflowed from the pad of ITS chunk, so all ranks skip the group
 * together (GradScaler.step semantics) without a second collective: nvo_flag_from_wire ORs (slot != 0) into *flag.
 * n multiple of 4 * world; pad a positive multiple of 4; wire16 holds world * (n / world + pad) elements. */
int nvo_cast_shards(nvo_stream_t stream, uint64_t n, uint32_t world, uint32_t pad, const float* src, void* wire16,
                    int wire_fmt, uint32_t* flag);
int nvo_flag_from_wire(nvo_stream_t stream, const void* wire_slot16, uint32_t* flag);
/* Clears up to 24 device ranges (host arrays of pointers / byte counts, 4-byte granular) with one launch. */
int nvo_zero_ranges(nvo_stream_t stream, uint32_t n_ranges, void* const* ptrs, const uint64_t* bytes);
/* Exponential moving average of the weights, the "Ema" optimiser wrapper of instant-ngp's configs/nerf/base.json
 * (decay 0.95) that pyngp.Testbed trains with (/root/reference/nerf_vo/mapping/instant_ngp.py:45 loads that file):
 * ema = (ema * decay * (1 - decay^(step-1)) + params * (1 - decay)) / (1 - decay^step), step counting from 1; the
 * optional fp16 copy (ema_half) is what inference reads.  skip_flag as in nvo_adam_step. */
int nvo_ema_update(nvo_stream_t stream, uint64_t n, const float* params, float* ema, void* ema_half, float decay,
                   uint32_t step, const uint32_t* skip_flag);
/* The same with the step count on the device: *step_dev = averages applied so far; the call uses t = *step_dev + 1 and
 * advances the counter iff the step was not skipped (the debias factor never runs ahead of the average). */
/* nvo_ema_update_dev without the commit of the step counter: for callers that average the parameters in several
 * launches (the last one is nvo_ema_update_dev itself). */
int nvo_ema_update_dev_part(nvo_stream_t stream, uint64_t n, const float* params, float* ema, void* ema_half, float decay,
                            const uint32_t* step_dev, const uint32_t* skip_flag);
int nvo_ema_update_dev(nvo_stream_t stream, uint64_t n, const float* params, float* ema, void* ema_half, float decay,
                       uint32_t* step_dev, const uint32_t* skip_flag);
/* bf16 MLP mode (BASELINE configs[4]: "MFMA bf16 MLP + fp32 hash accumulate"): the 16-bit working copy of the flat
 * parameter buffer is bfloat16 inside up to 4 element ranges [bf16_lo[k], bf16_hi[k]) (the fused-MLP weights and the
 * appearance embedding; bounds multiples of 4) and fp16 elsewhere (the hash tables).  Host arrays.
 * nvo_adam_step_groups_mixed == nvo_adam_step_groups with that format for the copy it writes. */
int nvo_cast_working_copy(nvo_stream_t stream, uint64_t n, const float* src, void* dst16, uint32_t n_bf16_ranges,
                          const uint64_t* bf16_lo, const uint64_t* bf16_hi);
int nvo_adam_step_groups_mixed(nvo_stream_t stream, uint32_t n_groups, const nvo_adam_group* groups, float* params,
                               void* params_half, const void* grads, int grads_are_half, float* exp_avg,
                               float* exp_avg_sq, float beta1, float beta2, float eps, float grad_scale,
                               float weight_decay, const uint32_t* skip_flags, uint32_t n_bf16_ranges,
                               const uint64_t* bf16_lo, const uint64_t* bf16_hi);
/* The same with the loss scale read from the device: loss_scale_dev (nullable) = device float, grad_scale is then
 * 1 / *loss_scale_dev (dynamic loss scaling, the reference trains with mixed_precision=True i.e. torch's GradScaler:
 * /root/reference/nerf_vo/mapping/nerfstudio.py:59). */
int nvo_adam_step_groups_scaled(nvo_stream_t stream, uint32_t n_groups, const nvo_adam_group* groups, float* params,
                                void* params_half, const void* grads, int grads_are_half, float* exp_avg,
                                float* exp_avg_sq, float beta1, float beta2, float eps, float grad_scale,
                                float weight_decay, const uint32_t* skip_flags, uint32_t n_bf16_ranges,
                                const uint64_t* bf16_lo, const uint64_t* bf16_hi, const float* loss_scale_dev);
/* nvo_adam_step_groups_scaled with a TAIL in the same launch (the occupancy-grid trainer's step: four launches less):
 *  - the weight average of exactly the elements the launch steps (nvo_ema_update_dev's arithmetic on the value just
 *    written; a skipped group is not averaged), and
 *  - by the LAST workgroup of the grid, once every other workgroup has checked in (each has then read the scalars a
 *    commit changes), the step's commit: nvo_opt_commit (applied / scale / bias) and the average's counter (ema_commit
 *    != 0: *ema_step_dev += 1 iff skip_flags[ema_flag_slot] == 0, what nvo_ema_update_dev does behind its launch).
 * done_counter: a device word, zero before the first launch; the launch leaves it zero (graph replay safe).  Should the
 * check-in not arrive at exactly the grid size (a counter left dirty, a stall of ~0.5 s) the launch does NOT commit and
 * sets bit 31 of the word, for good: the host checks it where it reads the step's results (a non-zero word after a
 * synchronised launch = the optimiser state can no longer be trusted).  Each part is
 * optional (NULL pointers = not requested); tail == NULL is nvo_adam_step_groups_scaled.  Measured (EXPERIMENTS.md
 * 9.11): -6 us on the occupancy-grid step; the nerfacto step, whose average-free tail is one commit launch, gains
 * nothing from it and keeps nvo_opt_commit_table. */
typedef struct nvo_adam_tail {
    float* ema;                 /* fp32 average, same origin as params (NULL: no averaging) */
    void* ema_half;             /* fp16 copy of the average (nullable) */
    float ema_decay;
    uint32_t* ema_step_dev;     /* device: averages applied so far */
    uint32_t ema_flag_slot;     /* word of skip_flags that gates the counter */
    uint32_t ema_commit;        /* != 0: advance *ema_step_dev behind the launch */
    uint32_t* done_counter;
    uint32_t n_commit_groups, active_mask, scale_mask;   /* as nvo_opt_commit */
    uint32_t* applied;
    float* scale;
    uint32_t* growth_tracker;
    float growth_factor, backoff_factor;
    uint32_t growth_interval;
    float min_scale, max_scale;
    float* bias;
} nvo_adam_tail;
int nvo_adam_step_groups_tail(nvo_stream_t stream, uint32_t n_groups, const nvo_adam_group* groups, float* params,
                              void* params_half, const void* grads, int grads_are_half, float* exp_avg,
                              float* exp_avg_sq, float beta1, float beta2, float eps, float grad_scale,
                              float weight_decay, const uint32_t* skip_flags, uint32_t n_bf16_ranges,
                              const uint64_t* bf16_lo, const uint64_t* bf16_hi, const float* loss_scale_dev,
                              const nvo_adam_tail* tail);
/* What GradScaler.step / GradScaler.update leave behind, on the device (one tiny launch behind the optimiser launches
 * of a step, capturable): for every group i in active_mask (bit i), applied[i] += 1 iff skip_flags[i] == 0; and, when
 * scale != NULL, the loss scale backs off (x backoff_factor, not below min_scale) if ANY group of scale_mask was
 * skipped and grows (x growth_factor, not above max_scale) after growth_interval consecutive clean steps (torch
 * defaults: init 65536, growth 2, backoff 0.5, interval 2000).  (Two masks: a step that runs its optimisers in two
 * launches commits each launch's counters behind it and updates the scale once, from all of the step's groups.)
 * applied / skip_flags: device uint32 [n_groups] (nullable); scale: device float, growth_tracker: device uint32 (both
 * nullable together). */
int nvo_opt_commit(nvo_stream_t stream, uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* applied,
                   const uint32_t* skip_flags, float* scale, uint32_t* growth_tracker, float growth_factor,
                   float backoff_factor, uint32_t growth_interval, float min_scale, float max_scale, float* bias,
                   float beta1, float beta2);
/* bias (nullable): device float [n_groups][2]; a group whose counter advances to t gets {1 - beta1^(t+1),
 * sqrt(1 - beta2^(t+1))} -- what nvo_adam_group::bias_dev of its next step reads. */
/* nvo_opt_commit and nvo_write_floats(dst, n, host_values) in ONE launch: a graph-replayed step ends with its Adam
 * launch and the commit rides in the eager launch that writes the next step's scalars. */
int nvo_opt_commit_write(nvo_stream_t stream, uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* applied,
                         const uint32_t* skip_flags, float* scale, uint32_t* growth_tracker, float growth_factor,
                         float backoff_factor, uint32_t growth_interval, float min_scale, float max_scale, float* bias,
                         float beta1, float beta2, float* dst, uint32_t n, const float* host_values);
/* The same as a node INSIDE a captured step: the scalars come from a device table the host fills ahead -- row
 * (s % table_rows) holds the 16 scalars of step s -- and *next_step (device uint32) names the row to load and is
 * advanced by one.  Nothing of the launch depends on host values, so every replay of the graph commits and loads the
 * right row (replaces the eager nvo_opt_commit_write behind each replay: no launch latency behind the graph). */
/* Sums the copies a fused-MLP backward spread its weight-gradient adds over (module options "dw_replicas_ptr" /
 * "dw_replicas", nvo_color_args::dw_replicas) into the gradient buffers and clears the copies: for entry i,
 * dst[i][e] += sum_r replicas[i][r * n[i] + e], replicas[i][...] = 0.  One launch for up to 8 networks, behind their
 * backwards and in front of whatever consumes the gradient (exchange, optimiser). */
int nvo_fold_replicas(nvo_stream_t stream, uint32_t n_entries, float* const* replicas, const uint32_t* n_replicas,
                      const uint64_t* n, float* const* dst);
int nvo_opt_commit_table(nvo_stream_t stream, uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* applied,
                         const uint32_t* skip_flags, float* scale, uint32_t* growth_tracker, float growth_factor,
                         float backoff_factor, uint32_t growth_interval, float min_scale, float max_scale, float* bias,
                         float beta1, float beta2, float* dst, const float* table, uint32_t table_rows, uint32_t* next_step);

/* ------------------------------------------------------------------------------------------------
 * G. Keyframe depth alignment (the producer right before the mapping path; replaces the torch-op chain of
 *    /root/reference/nerf_vo/enhancement/enhancement_module.py:61-99 and dpvo_remove_outliers :131-146).
 *    patches: DPVO patches [K][M][3][P][P] = (x/4, y/4, inverse depth); noise [K][M] in [0,1) (the reference's
 *    torch.rand tie-breaker, drawn by the caller); frames_depth [K][H][W] monocular depth.
 *    out_depth = clip(frames_depth * scale_k + shift_k, 0, 5); may alias frames_depth.
 * ---------------------------------------------------------------------------------------------- */
typedef struct nvo_depth_align_args {
    uint32_t K, M, P, H, W;
    const float* patches;
    const float* noise;
    const float* frames_depth;
    float* out_depth;
    void* scratch;               /* nvo_depth_align_scratch_bytes(K, M) */
    float* scale_shift_out;      /* optional [K][2] */
} nvo_depth_align_args;
uint64_t nvo_depth_align_scratch_bytes(uint32_t K, uint32_t M);
int nvo_depth_align(nvo_stream_t stream, const nvo_depth_align_args* args);

#ifdef __cplusplus
}
#endif
#endif /* NERFVO_HIP_H */
