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
/* Integer options: "grid_bwd_mode" 0 = global float atomics, 1 = LDS slice-owner scatter. */
int nvo_set_option(nvo_module_t m, const char* key, int64_t value);

/* input  : device float [batch][n_input_dims]
 * params : device fp16 [n_params]   (NetworkWithInputEncoding: network weights first, then grid)
 * output : device fp16 [batch][padded_output_dims]
 * ctx    : device scratch of nvo_ctx_bytes() bytes, or NULL for inference (nothing is saved) */
int nvo_fwd(nvo_module_t m, nvo_stream_t stream, uint32_t batch, const float* input,
            const void* params, void* output, void* ctx);
/* dL_doutput: device fp16 [batch][padded_output_dims] (already multiplied by the loss scale)
 * dL_dinput : device float [batch][n_input_dims] or NULL
 * dL_dparams: device float [n_params] or NULL; overwritten (not accumulated into) */
int nvo_bwd(nvo_module_t m, nvo_stream_t stream, uint32_t batch, const float* input,
            const void* params, const void* output, const void* dL_doutput, void* ctx,
            float* dL_dinput, float* dL_dparams);

/* Parity/debug: the per-level table geometry and the 8 corner indices the encoder uses.
 * levels_out: host uint32 [n_levels][4] = {offset, size, resolution, hashed}; scales_out: host
 * float [n_levels].  indices_out: device uint32 [n_levels][batch][8]. */
int nvo_grid_describe(nvo_module_t m, uint32_t* levels_out, float* scales_out);
int nvo_grid_indices(nvo_module_t m, nvo_stream_t stream, uint32_t batch, const float* input,
                     uint32_t* indices_out);

#ifdef __cplusplus
}
#endif
#endif /* NERFVO_HIP_H */
