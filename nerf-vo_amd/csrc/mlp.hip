// Fully fused small-MLP forward / backward on CDNA4 matrix cores (gfx950).
// Replaces tiny-cuda-nn's kernel_mlp_fused / kernel_mlp_fused_backward + the CUTLASS split-K
// weight-gradient GEMMs (upstream fully_fused_mlp.cu is not vendored
// in /root/reference -- restated in oracle/mlp.py).
//
// Layout idea ("transposed chain", MI355X-first, no warp-shaped tiling):
//   Every layer is evaluated as  H_out^T[n][m] = W[n][k] * H_in^T[k][m]  on
//   v_mfma_f32_16x16x16_f16, with the SAMPLE index m on the MFMA column (lane & 15) and the
//   feature index in the lane's registers.  The C/D register map of that instruction
//   (row = 4*(lane>>4)+reg, col = lane&15) is exactly its own B-operand map
//   (k = 4*(lane>>4)+j, col = lane&15), so a layer's accumulator tile, once activated and packed to
//   fp16, IS the next layer's B operand: the whole MLP runs in registers, no LDS round trip, no
//   shuffles.  Weights (row-major [out][in] fp16, tcnn's order) are the A operand and are loaded
//   into registers once per wave; a wave then streams 16-sample tiles.
//   Backward runs the same chain with W^T as the A operand (dH_in^T = W^T * dZ^T) and forms
//   dW[n][k] = sum_m dZ[m][n] H[m][k] on the matrix cores too: both operands need the sample
//   index in registers, which is a 16x16 transpose of what the chain holds -- done with one
//   wave-private LDS tile and ds_read_b64_tr_b16 (hardware transposing read).
//   dW accumulates in fp32 registers across all of a wave's tiles and is flushed once with
//   row-contiguous float atomics.
//
// Numerics: fp16 operands, fp32 accumulate (tcnn accumulates in fp16), fp16 hidden activations.
// The kernel bodies live in mlp_impl.h and are compiled once per 16-bit element type: here for fp16 (tcnn's
// precision), in mlp_bf16.hip for bf16 (v_mfma_f32_16x16x16_bf16 -- BASELINE configs[4]); NvoMlpArgs::bf16 selects.
#define NVO_MLP_BF16 0
#include "mlp_impl.h"

#include "../../include/nerfvo_hip.h"

bool nvo_mlp_shape_supported(int in_pad, int width, int n_hidden, int out_pad) {
    return mlp_shape_supported_impl(in_pad, width, n_hidden, out_pad);
}

int nvo_mlp_fwd_launch(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a, hipStream_t stream) {
    return a.bf16 ? nvo_mlp_fwd_launch_bf16(in_pad, width, n_hidden, out_pad, a, stream)
                  : nvo_mlp_fwd_launch_f16(in_pad, width, n_hidden, out_pad, a, stream);
}

int nvo_mlp_bwd_launch(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a, hipStream_t stream) {
    return a.bf16 ? nvo_mlp_bwd_launch_bf16(in_pad, width, n_hidden, out_pad, a, stream)
                  : nvo_mlp_bwd_launch_f16(in_pad, width, n_hidden, out_pad, a, stream);
}

// ---------------------------------------------------------------------------------------------
// exported: NerfactoField colour head (group D of include/nerfvo_hip.h)
// ---------------------------------------------------------------------------------------------
static NvoMlpArgs color_args(const nvo_color_args& c) {
    NvoMlpArgs a;
    memset(&a, 0, sizeof(a));
    a.batch = c.R * c.S;
    a.n_in = 63;
    a.in_mode = NVO_IO_NERFACTO_COLOR;
    a.weights = (const _Float16*)c.weights;
    a.output = (_Float16*)c.rgb;
    a.hidden = (_Float16*)c.hidden;
    a.act = NVO_ACT_RELU;
    a.out_act = NVO_ACT_SIGMOID;
    a.samples_per_ray = c.S;
    a.sh = (const _Float16*)c.sh;
    a.base_out = (const _Float16*)c.base_out;
    a.embedding = (const _Float16*)c.embedding;
    a.cam_idx = c.cam_idx;
    a.bf16 = c.act_bf16 != 0;
    return a;
}

#ifdef NVO_MLP_PHASE
extern "C" int nvo_debug_mlp_phase(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(nvo_mlp_phase_cycles_f16), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif
extern "C" {

int nvo_nerfacto_color_fwd(nvo_stream_t stream, const nvo_color_args* args) {
    NVO_REQUIRE(args != nullptr, "color_fwd: args is NULL");
    const nvo_color_args c = *args;
    NVO_REQUIRE(c.S >= 1 && c.sh && c.base_out && c.embedding && c.weights && c.rgb, "color_fwd: NULL argument");
    const NvoMlpArgs a = color_args(c);
    return nvo_mlp_fwd_launch(64, 64, 2, 16, a, (hipStream_t)stream);
}

static NvoMlpArgs ngp_rgb_args(const nvo_ngp_rgb_args& c) {
    NvoMlpArgs a;
    memset(&a, 0, sizeof(a));
    a.batch = c.capacity;
    a.n_in = 32;
    a.in_mode = NVO_IO_NGP_RGB;
    a.weights = (const _Float16*)c.weights;
    a.output = (_Float16*)c.rgb_out;
    a.hidden = (_Float16*)c.hidden;
    a.act = NVO_ACT_RELU;
    a.out_act = NVO_ACT_NONE;  // the logistic lives in the compositing kernel, as in instant-ngp
    a.sh = (const _Float16*)c.sh;
    a.base_out = (const _Float16*)c.density_out;
    a.sample_ray = c.ray_idx;
    return a;
}

int nvo_ngp_rgb_fwd(nvo_stream_t stream, const nvo_ngp_rgb_args* args) {
    NVO_REQUIRE(args != nullptr, "ngp_rgb_fwd: args is NULL");
    const nvo_ngp_rgb_args c = *args;
    NVO_REQUIRE(c.sh && c.density_out && c.ray_idx && c.weights && c.rgb_out, "ngp_rgb_fwd: NULL argument");
    return nvo_mlp_fwd_launch(32, 64, 2, 16, ngp_rgb_args(c), (hipStream_t)stream);
}

int nvo_ngp_rgb_bwd(nvo_stream_t stream, const nvo_ngp_rgb_args* args) {
    NVO_REQUIRE(args != nullptr, "ngp_rgb_bwd: args is NULL");
    const nvo_ngp_rgb_args c = *args;
    NVO_REQUIRE(c.sh && c.density_out && c.ray_idx && c.weights && c.rgb_out && c.hidden && c.d_rgb_out &&
                c.d_density_out, "ngp_rgb_bwd: NULL argument");
    NvoMlpArgs a = ngp_rgb_args(c);
    a.doutput = (const _Float16*)c.d_rgb_out;
    a.dinput = c.d_density_out;
    a.din_mode = NVO_IO_NGP_RGB;
    a.d_base_out = (_Float16*)c.d_density_out;
    a.d_extra_col0 = c.d_density_pre;
    a.dweights = c.d_weights;
    return nvo_mlp_bwd_launch(32, 64, 2, 16, a, (hipStream_t)stream);
}

int nvo_nerfacto_color_bwd(nvo_stream_t stream, const nvo_color_args* args) {
    NVO_REQUIRE(args != nullptr, "color_bwd: args is NULL");
    const nvo_color_args c = *args;
    NVO_REQUIRE(c.S >= 1 && c.sh && c.base_out && c.embedding && c.weights && c.rgb && c.drgb &&
                c.d_base_out, "color_bwd: NULL argument");
    NvoMlpArgs a = color_args(c);
    a.doutput = (const _Float16*)c.drgb;
    a.dinput = c.d_base_out;
    a.din_mode = NVO_IO_NERFACTO_COLOR;
    a.d_base_out = (_Float16*)c.d_base_out;
    a.d_embedding = c.d_embedding;
    a.d_sh = c.d_sh;
    a.dweights = c.d_weights;
    a.recompute_hidden = c.hidden == nullptr;  // no stored activations: both hidden layers are recomputed
    return nvo_mlp_bwd_launch(64, 64, 2, 16, a, (hipStream_t)stream);
}

}  // extern "C"
