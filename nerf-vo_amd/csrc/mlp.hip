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

uint32_t nvo_mlp_bwd_blocks(int in_pad, int width, int n_hidden, uint32_t batch) {
    uint32_t blocks = nvo_div_up(batch >> 4, kWavesPerBlock);
    const uint32_t cap = env_blocks("NVO_MLP_BWD_BLOCKS", bwd_block_cap(in_pad, width, n_hidden));
    return blocks > cap ? cap : blocks;
}
bool nvo_mlp_bwd_lists_rows(int in_pad, int width, int n_hidden, uint32_t batch) {
    const char* e = getenv("NVO_MLP_ROLES");
    if (width != 64 || n_hidden != 1 || (e && atoi(e) == 0) || (batch & 15u)) return false;  // (the role-split form of launch_bwd_io_kernel)
    const uint32_t n_tiles = batch >> 4, blocks = nvo_mlp_bwd_blocks(in_pad, width, n_hidden, batch);
    if (!blocks) return false;
    const uint32_t n_own = nvo_div_up(n_tiles, blocks * 2u * kWavesPerBlock) * 2u * kWavesPerBlock;
    return n_own * 16u <= (uint32_t)kLiveRowCap;
}
uint64_t nvo_mlp_n_weights(int in_pad, int width, int n_hidden, int out_pad) {
    return (uint64_t)width * in_pad + (uint64_t)(n_hidden - 1) * width * width + (uint64_t)out_pad * width;
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
    a.nf_flag = c.nonfinite_flag;
    if (c.dw_replicas && c.n_dw_replicas && !c.det_scratch) {
        a.dw_replicas = c.dw_replicas;
        a.dw_n_replicas = c.n_dw_replicas;
    }
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
    a.nf_flag = c.nonfinite_flag;
    if (c.dw_replicas && c.n_dw_replicas) {
        a.dw_replicas = c.dw_replicas;
        a.dw_n_replicas = c.n_dw_replicas;
    }
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
    NVO_REQUIRE(c.sh && c.density_out && c.ray_idx && c.weights && c.rgb_out && c.d_rgb_out &&
                c.d_density_out, "ngp_rgb_bwd: NULL argument");
    NvoMlpArgs a = ngp_rgb_args(c);
    a.doutput = (const _Float16*)c.d_rgb_out;
    a.dinput = c.d_density_out;
    a.din_mode = NVO_IO_NGP_RGB;
    a.d_base_out = (_Float16*)c.d_density_out;
    a.d_extra_col0 = c.d_density_pre;
    a.dweights = c.d_weights;
    a.recompute_hidden = c.hidden == nullptr;  // no stored activations: both hidden layers are recomputed (as the colour head)
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
    if (c.tile_live && (c.S & 15u) == 0) {  // (bit 0: the tile's drgb rows are not all zero -- nvo_main_loss_args::tile_live)
        a.tile_live = c.tile_live;
        a.tile_live_bits = 1u;
        a.tile_live_count = c.tile_live_count;
    }
    if (c.det_scratch) {
        // deterministic mode: [dW block totals | per-tile embedding / SH sums | per-ray sums], all summed in fixed orders
        NVO_REQUIRE((c.S & 15u) == 0 && c.cam_idx && c.d_weights, "color_bwd: the deterministic form needs S %% 16 == 0, cam_idx, d_weights");
        NVO_REQUIRE(c.det_scratch_bytes >= nvo_color_det_scratch_bytes(c.R, c.S), "color_bwd: deterministic scratch too small");
        const uint32_t n_tiles = (c.R * c.S) >> 4;
        float* dw_partial = static_cast<float*>(c.det_scratch);
        float* tile_partial = dw_partial + (size_t)nvo_mlp_bwd_blocks(64, 64, 2, c.R * c.S) * nvo_mlp_n_weights(64, 64, 2, 16);
        float* per_ray = tile_partial + (size_t)n_tiles * 48;
        a.dw_partial = dw_partial;
        a.tile_partial = tile_partial;
        if (int rc = nvo_mlp_bwd_launch(64, 64, 2, 16, a, (hipStream_t)stream)) return rc;
        if (int rc = nvo_color_tiles_to_rays((hipStream_t)stream, c.R, c.S >> 4, tile_partial, per_ray, c.d_sh)) return rc;
        if (c.d_embedding)
            return nvo_reduce_by_camera((hipStream_t)stream, c.R, 32, per_ray, 48, c.cam_idx, 0, c.n_cameras, c.d_embedding);
        return NVO_OK;
    }
    return nvo_mlp_bwd_launch(64, 64, 2, 16, a, (hipStream_t)stream);
}

uint64_t nvo_color_det_scratch_bytes(uint32_t R, uint32_t S) {
    const uint64_t n = (uint64_t)R * S;
    return sizeof(float) * ((uint64_t)nvo_mlp_bwd_blocks(64, 64, 2, (uint32_t)n) * nvo_mlp_n_weights(64, 64, 2, 16) +
                            (n >> 4) * 48 + (uint64_t)R * 48);
}

}  // extern "C"
