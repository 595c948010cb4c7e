// Kernel bodies of the fully fused MLP, compiled once per 16-bit element type:
//   mlp.hip       #define NVO_MLP_BF16 0  ->  T = _Float16, v_mfma_f32_16x16x16_f16   (tcnn's precision)
//   mlp_bf16.hip  #define NVO_MLP_BF16 1  ->  T = __bf16,   v_mfma_f32_16x16x16_bf16  (BASELINE configs[4])
// Everything the network streams -- weights, inputs from the encoding, hidden activations, outputs and all of
// their gradients -- is T in memory; accumulation is fp32 in both.  See mlp.hip for the design notes.
#pragma once
#include <type_traits>
#include "nvo_kernels.h"

#include <stdlib.h>
#include <string.h>

#ifndef NVO_MLP_BF16
#error "define NVO_MLP_BF16 (0 | 1) before including mlp_impl.h"
#endif
#if NVO_MLP_BF16
#define NVO_MLP_NAME(x) x##_bf16
#define NVO_MLP_TAG ":bf16"
#else
#define NVO_MLP_NAME(x) x##_f16
#define NVO_MLP_TAG ""
#endif

namespace {

#if NVO_MLP_BF16
typedef __bf16 T;
#else
typedef _Float16 T;
#endif
typedef T T4 __attribute__((ext_vector_type(4)));
typedef T T2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef NvoMlpArgsT<T> Args;


constexpr int kMlpBlock = 256;
constexpr int kWavesPerBlock = kMlpBlock / 64;

// one MFMA 16x16x16 (K = 16: 4 operand elements per lane); fp16 and bf16 share the register layout
__device__ __forceinline__ f4 mfma16(T4 a, T4 b, f4 c) {
#if NVO_MLP_BF16
    typedef short s4 __attribute__((ext_vector_type(4)));
    s4 as, bs;
    __builtin_memcpy(&as, &a, sizeof(as));
    __builtin_memcpy(&bs, &b, sizeof(bs));
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as, bs, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
#endif
}

// gfx950's K = 32 form, operands given as two K = 16 fragments each: element j < 4 of lane group g pairs feature
// 16 (2t) + 4g + j of both operands, j >= 4 feature 16 (2t + 1) + 4g + (j - 4) -- the same 32 products as two chained
// 16x16x16 instructions in HALF the matrix-core time (the K = 16 form runs at the MI300 rate on this chip); the order of
// the fp32 additions inside differs, which forward and recomputing backward share.
#ifdef NVO_MLP_K16  // (A/B builds: NVO_EXTRA_CXXFLAGS=-DNVO_MLP_K16)
constexpr bool kUseK32 = false;
#else
constexpr bool kUseK32 = true;
#endif
__device__ __forceinline__ f4 mfma32(T4 a_lo, T4 a_hi, T4 b_lo, T4 b_hi, f4 c) {
    typedef T T8 __attribute__((ext_vector_type(8)));
    const T8 a = __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7);
    const T8 b = __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7);
#if NVO_MLP_BF16
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float act_fwd(int act, float v) {
    if (act == NVO_ACT_RELU) return fmaxf(v, 0.f);
    if (act == NVO_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// derivative expressed through the (saved, fp16-rounded) activation output
__device__ __forceinline__ float act_bwd_from_out(int act, float out) {
    if (act == NVO_ACT_RELU) return out > 0.f ? 1.f : 0.f;
    if (act == NVO_ACT_SIGMOID) return out * (1.f - out);
    return 1.f;
}

__device__ __forceinline__ T4 pack_act(int act, f4 v) {
    T4 r;
    r[0] = (T)act_fwd(act, v[0]);
    r[1] = (T)act_fwd(act, v[1]);
    r[2] = (T)act_fwd(act, v[2]);
    r[3] = (T)act_fwd(act, v[3]);
    return r;
}

// A-operand fragments of W[N_OUT][K_IN] (row-major): f[tn][tk] = W[16tn + (l&15)][16tk + 4g + j]
template <int N_OUT, int K_IN>
struct WFrag {
    T4 f[N_OUT / 16][K_IN / 16];
    __device__ __forceinline__ void load(const T* __restrict__ W, int lane) {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int tn = 0; tn < N_OUT / 16; ++tn)
#pragma unroll
            for (int tk = 0; tk < K_IN / 16; ++tk)
                f[tn][tk] = *reinterpret_cast<const T4*>(W + (size_t)(16 * tn + r) * K_IN + 16 * tk + 4 * g);
    }
    // the same fragments from the workgroup's row-major LDS copy of W (row stride K_IN + 4 halfs, see RowStage): the
    // direct form touches 16 rows x 32 bytes per load instruction -- 16 cache lines for 512 bytes -- and every wave of
    // the grid pays it for the whole weight set
    __device__ __forceinline__ void load_staged(const T* stage, int lane) {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int tn = 0; tn < N_OUT / 16; ++tn)
#pragma unroll
            for (int tk = 0; tk < K_IN / 16; ++tk)
                f[tn][tk] = *reinterpret_cast<const T4*>(stage + (16 * tn + r) * (K_IN + 4) + 16 * tk + 4 * g);
    }
};

// A-operand fragments of W^T: f[tk][tn] = W[16tn + 4g + j][16tk + (l&15)]   (strided gather, once)
template <int N_OUT, int K_IN>
struct WTFrag {
    T4 f[K_IN / 16][N_OUT / 16];
    __device__ __forceinline__ void load(const T* __restrict__ W, int lane) {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int tk = 0; tk < K_IN / 16; ++tk)
#pragma unroll
            for (int tn = 0; tn < N_OUT / 16; ++tn) {
                T4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = W[(size_t)(16 * tn + 4 * g + j) * K_IN + 16 * tk + r];
                f[tk][tn] = v;
            }
    }
    // Same fragments through LDS: the workgroup copies W row-major (coalesced 8-byte loads) into `stage`
    // (row stride K_IN + 4 halfs) and every wave takes its transposed fragments with ds_read_b64_tr_b16.
    // The direct form above costs 4 two-byte gathers per fragment with every lane on its own cache line
    // (144 load instructions x 64 lines per wave for the colour head) -- the dominant per-wave setup cost.
    // MUST be called by all threads of the workgroup (barriers); `stage` holds >= N_OUT * (K_IN + 4) halfs.
    __device__ __forceinline__ void load_lds(const T* __restrict__ W, T* stage, int lane) {
        constexpr int kStride = K_IN + 4;
        __syncthreads();  // previous users of the staging area are done
        for (int e = threadIdx.x; e < N_OUT * K_IN / 4; e += blockDim.x) {
            const int n = (4 * e) / K_IN, k = (4 * e) % K_IN;
            *reinterpret_cast<T4*>(stage + n * kStride + k) = *reinterpret_cast<const T4*>(W + (size_t)n * K_IN + k);
        }
        __syncthreads();
        read_staged(stage, lane);
    }
    // the fragment reads alone: `stage` holds W row-major with row stride K_IN + 4 (see RowStage)
    __device__ __forceinline__ void read_staged(const T* stage, int lane) {
        constexpr int kStride = K_IN + 4;
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
#pragma unroll
        for (int tk = 0; tk < K_IN / 16; ++tk)
#pragma unroll
            for (int tn = 0; tn < N_OUT / 16; ++tn) {
                const T* addr = stage + (16 * tn + 4 * g + q) * kStride + 16 * tk + 4 * pp;
                fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4_t*)addr);
                __builtin_memcpy(&f[tk][tn], &v, sizeof(T4));
            }
    }
};

// One matrix on its way global -> registers -> LDS staging (row stride K_IN + 4 halfs).  Split in two so that a kernel
// can have the loads of ALL its matrices in flight before the first LDS store waits for one of them.
template <int N_OUT, int K_IN>
struct RowStage {
    static constexpr int kT4 = N_OUT * K_IN / 4, kTrips = (kT4 + kMlpBlock - 1) / kMlpBlock, kHalfs = N_OUT * (K_IN + 4);
    T4 v[kTrips];
    __device__ __forceinline__ void issue(const T* __restrict__ W) {
#pragma unroll
        for (int i = 0; i < kTrips; ++i) {
            const int e = (int)threadIdx.x + i * kMlpBlock;
            v[i] = *reinterpret_cast<const T4*>(W + 4 * (size_t)(e < kT4 ? e : 0));
        }
    }
    __device__ __forceinline__ void store(T* stage) const {
#pragma unroll
        for (int i = 0; i < kTrips; ++i) {
            const int e = (int)threadIdx.x + i * kMlpBlock;
            if (e < kT4) *reinterpret_cast<T4*>(stage + ((4 * e) / K_IN) * (K_IN + 4) + (4 * e) % K_IN) = v[i];
        }
    }
};

// H_out^T tile = W * H_in^T
template <int N_OUT, int K_IN>
__device__ __forceinline__ void layer_mm(const WFrag<N_OUT, K_IN>& w, const T4 (&in)[K_IN / 16],
                                         f4 (&acc)[N_OUT / 16]) {
#pragma unroll
    for (int tn = 0; tn < N_OUT / 16; ++tn) {
        f4 c = {0.f, 0.f, 0.f, 0.f};
        if constexpr ((K_IN / 16) % 2 == 0 && kUseK32) {
#pragma unroll
            for (int tk = 0; tk < K_IN / 16; tk += 2) c = mfma32(w.f[tn][tk], w.f[tn][tk + 1], in[tk], in[tk + 1], c);
        } else {
#pragma unroll
            for (int tk = 0; tk < K_IN / 16; ++tk) c = mfma16(w.f[tn][tk], in[tk], c);
        }
        acc[tn] = c;
    }
}

// dH_in^T tile = W^T * dZ^T
template <int N_OUT, int K_IN>
__device__ __forceinline__ void layer_mm_t(const WTFrag<N_OUT, K_IN>& wt,
                                           const T4 (&dz)[N_OUT / 16], f4 (&acc)[K_IN / 16]) {
#pragma unroll
    for (int tk = 0; tk < K_IN / 16; ++tk) {
        f4 c = {0.f, 0.f, 0.f, 0.f};
        if constexpr ((N_OUT / 16) % 2 == 0 && kUseK32) {
#pragma unroll
            for (int tn = 0; tn < N_OUT / 16; tn += 2) c = mfma32(wt.f[tk][tn], wt.f[tk][tn + 1], dz[tn], dz[tn + 1], c);
        } else {
#pragma unroll
            for (int tn = 0; tn < N_OUT / 16; ++tn) c = mfma16(wt.f[tk][tn], dz[tn], c);
        }
        acc[tk] = c;
    }
}

// layer_mm with the A fragments read from a row-major LDS copy of W (row stride K_IN + 4 halfs, see RowStage) one output
// tile at a time: same fragments, same instruction sequence, same result as layer_mm on WFrag registers
template <int N_OUT, int K_IN>
__device__ __forceinline__ void layer_mm_lds(const T* stage, int lane, const T4 (&in)[K_IN / 16], f4 (&acc)[N_OUT / 16]) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < N_OUT / 16; ++tn) {
        T4 w[K_IN / 16];
#pragma unroll
        for (int tk = 0; tk < K_IN / 16; ++tk)
            w[tk] = *reinterpret_cast<const T4*>(stage + (16 * tn + r) * (K_IN + 4) + 16 * tk + 4 * g);
        f4 c = {0.f, 0.f, 0.f, 0.f};
        if constexpr ((K_IN / 16) % 2 == 0 && kUseK32) {
#pragma unroll
            for (int tk = 0; tk < K_IN / 16; tk += 2) c = mfma32(w[tk], w[tk + 1], in[tk], in[tk + 1], c);
        } else {
#pragma unroll
            for (int tk = 0; tk < K_IN / 16; ++tk) c = mfma16(w[tk], in[tk], c);
        }
        acc[tn] = c;
    }
}

// Load the B-operand fragments of one 16-sample input tile: x[tk][j] = in[m][16tk + 4g + j]
template <int IN_PAD, int IO>
__device__ __forceinline__ void load_input(const Args& a, uint32_t row, int g,
                                           T4 (&x)[IN_PAD / 16], uint32_t cam = 0u) {
    // cam: (NVO_IO_NERFACTO_COLOR only) appearance-embedding row of this sample's ray, loaded by the caller
    if constexpr (IO == NVO_IO_F32_ROWS) {
        const float* __restrict__ p = (const float*)a.input + (size_t)row * a.n_in;
#pragma unroll
        for (int tk = 0; tk < IN_PAD / 16; ++tk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t c = 16 * tk + 4 * g + j;
                x[tk][j] = c < a.n_in ? (T)p[c] : (T)1.0f;
            }
        }
    } else if constexpr (IO == NVO_IO_HALF2_SOA) {
        const T2* __restrict__ p = (const T2*)a.input;
        const uint32_t n_lv = a.n_in >> 1;
#pragma unroll
        for (int tk = 0; tk < IN_PAD / 16; ++tk) {
            const uint32_t lv = 8 * tk + 2 * g;
            T2 v0 = {(T)0.f, (T)0.f}, v1 = v0;
            if (lv < n_lv) v0 = p[(size_t)lv * a.batch + row];
            if (lv + 1 < n_lv) v1 = p[(size_t)(lv + 1) * a.batch + row];
            x[tk][0] = v0[0];
            x[tk][1] = v0[1];
            x[tk][2] = v1[0];
            x[tk][3] = v1[1];
        }
    } else if constexpr (IO == NVO_IO_NERFACTO_COLOR) {
        if constexpr (IN_PAD == 64) {
            // row = [SH16(ray) | geo15 = base_out[1..15] | embed32(cam) | 1]; features 16.. are the source rows
            // shifted by one half, fetched as aligned 8-byte pieces (v0 = src[4g..4g+3], v1 = src[4g+4..4g+7])
            const uint32_t ray = row / a.samples_per_ray;
            const T* __restrict__ sh = a.sh + (size_t)ray * 16;
            const T* __restrict__ bo = a.base_out + (size_t)row * 16;
            const T* __restrict__ em = a.embedding + (size_t)cam * 32;
            x[0] = *reinterpret_cast<const T4*>(sh + 4 * g);
            const T4 b0 = *reinterpret_cast<const T4*>(bo + 4 * g);
            // (only what is used is requested: a wider load whose upper part is dead hands the dead registers back to
            // the allocator while the load is still in flight, and their first reuse then waits for the whole prefetch)
            const T b1 = bo[g < 3 ? 4 * g + 4 : 12];  // g == 3: unused lane value
            const T4 e0 = *reinterpret_cast<const T4*>(em + 4 * g);
            const T e1 = em[4 * g + 4];
            const T4 e2 = *reinterpret_cast<const T4*>(em + 16 + 4 * g);
            const T e3 = em[g < 3 ? 20 + 4 * g : 28];
            const T em0 = em[0];
            x[1] = T4{b0[1], b0[2], b0[3], g < 3 ? b1 : em0};          // features 16 + 4g + j
            x[2] = T4{e0[1], e0[2], e0[3], e1};                        // embed 1 + 4g + j
            x[3] = T4{e2[1], e2[2], e2[3], g < 3 ? e3 : (T)1.0f};  // embed 17 + 4g + j | pad
        }
    } else if constexpr (IO == NVO_IO_NGP_RGB) {
        if constexpr (IN_PAD == 32) {
            const int32_t ray = a.sample_ray[row];
            x[0] = *reinterpret_cast<const T4*>(a.base_out + (size_t)row * 16 + 4 * g);
            T4 z = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
            x[1] = ray >= 0 ? *reinterpret_cast<const T4*>(a.sh + (size_t)ray * 16 + 4 * g) : z;
        }
    } else {
        const T* __restrict__ p = (const T*)a.input + (size_t)row * IN_PAD;
#pragma unroll
        for (int tk = 0; tk < IN_PAD / 16; ++tk)
            x[tk] = *reinterpret_cast<const T4*>(p + 16 * tk + 4 * g);
    }
}

// ---- NVO_IO_GRID_FUSED: the hash-grid gather of the two levels a lane feeds into each 16-feature tile --------------
// (lane group g of K-tile tk holds features 16 tk + 4 g .. + 3 = levels 8 tk + 2 g and 8 tk + 2 g + 1 of sample m.)
// Split in two so that the gathers of the NEXT tile are in flight while the current tile runs through the MFMA chain:
// grid_issue computes the corner indices and requests the 8 table entries per level; grid_finish interpolates in fp32
// in exactly the order of k_grid_fwd (bit-identical features), rounds once to E and optionally stores the pair.
template <int IN_PAD>
struct GridGather {
    uint32_t v[IN_PAD / 16][2][8];  // raw half2 table entries
    float w[IN_PAD / 16][2][3];     // fractional position per axis
    uint32_t row;
};
template <int IN_PAD>
__device__ __forceinline__ void grid_issue(const Args& a, uint32_t row, int g, GridGather<IN_PAD>& gg) {
    const NvoGridLevels& G = *a.grid;
    const float* __restrict__ xp = (const float*)a.input + 3 * (size_t)row;
    const float px = xp[0], py = xp[1], pz = xp[2];
    const uint32_t* __restrict__ table = (const uint32_t*)a.grid_table;
    gg.row = row;
#pragma unroll
    for (int tk = 0; tk < IN_PAD / 16; ++tk) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t lv = 8 * tk + 2 * g + h;
            if (lv < G.n_levels) {
                // tcnn pos_fract: pos = fma(scale, x, 0.5); cell = floor(pos); frac = pos - cell   (= grid.hip grid_cell)
                const float sc = G.scale[lv];
                const float fx = fmaf(sc, px, 0.5f), fy = fmaf(sc, py, 0.5f), fz = fmaf(sc, pz, 0.5f);
                const float tx = floorf(fx), ty = floorf(fy), tz = floorf(fz);
                const uint32_t cx = (uint32_t)(int)tx, cy = (uint32_t)(int)ty, cz = (uint32_t)(int)tz;
                gg.w[tk][h][0] = fx - tx;
                gg.w[tk][h][1] = fy - ty;
                gg.w[tk][h][2] = fz - tz;
                const uint32_t off = G.offset[lv], size = G.offset[lv + 1] - off;
                const uint32_t res = G.resolution[lv], hashed = G.hashed[lv];
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k)
                    gg.v[tk][h][k] = table[off + nvo_grid_index(hashed, size, res, cx + (k & 1u), cy + ((k >> 1) & 1u),
                                                               cz + ((k >> 2) & 1u))];
            }
        }
    }
}
template <int IN_PAD>
__device__ __forceinline__ void grid_finish(const Args& a, int g, const GridGather<IN_PAD>& gg, T4 (&x)[IN_PAD / 16]) {
    const uint32_t n_levels = a.grid->n_levels;
#pragma unroll
    for (int tk = 0; tk < IN_PAD / 16; ++tk) {
        T pair[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t lv = 8 * tk + 2 * g + h;
            pair[h][0] = pair[h][1] = (T)0.f;
            if (lv < n_levels) {
                const float wx = gg.w[tk][h][0], wy = gg.w[tk][h][1], wz = gg.w[tk][h][2];
                float r0 = 0.f, r1 = 0.f;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const float w = ((k & 1u) ? wx : 1.f - wx) * ((k & 2u) ? wy : 1.f - wy) * ((k & 4u) ? wz : 1.f - wz);
                    const float2 f = nvo_ld16x2(gg.v[tk][h][k], false);  // the tables are fp16 in both modes
                    r0 = fmaf(w, f.x, r0);
                    r1 = fmaf(w, f.y, r1);
                }
                const uint32_t packed = nvo_cvt16x2(r0, r1, NVO_MLP_BF16 != 0);  // ONE rounding, as k_grid_fwd
                pair[h][0] = __builtin_bit_cast(T, (nvo_h16)(packed & 0xFFFFu));
                pair[h][1] = __builtin_bit_cast(T, (nvo_h16)(packed >> 16));
                if (a.enc_out) reinterpret_cast<uint32_t*>(a.enc_out)[(size_t)lv * a.batch + gg.row] = packed;
            }
        }
        x[tk] = T4{pair[0][0], pair[0][1], pair[1][0], pair[1][1]};
    }
}

// sum over the 16 sample lanes (lane & 15) of one lane group
// Sum over the 16 lanes of a DPP row (= the 16 samples of a tile), every lane receives the total.  Row rotations are
// DPP modifiers of the add itself; __shfl_xor compiles to ds_bpermute_b32, an LDS round trip per step -- 64 dependent
// ones per tile in the colour head's epilogue, which with one wave per SIMD was 38 % of the tile loop (tools/mlp_phase.py).
template <int CTRL>
__device__ __forceinline__ float dpp_row_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float group16_sum(float v) {
    v += dpp_row_mov<0x128>(v);  // row_ror:8
    v += dpp_row_mov<0x124>(v);  // row_ror:4
    v += dpp_row_mov<0x122>(v);  // row_ror:2
    v += dpp_row_mov<0x121>(v);  // row_ror:1
    return v;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// COMPACT: only output column 0 exists in memory ([B] halfs instead of [B][OUT_PAD]): the density networks of
// the proposal sampler produce one number per sample, and a 32-byte row per sample costs 16x the traffic in
// this kernel, in the per-ray kernels that read it with a 32-byte stride, and again in the backward.
template <int IN_PAD, int WIDTH, int N_HIDDEN, int OUT_PAD, int IO, bool RELU, bool COMPACT>
__global__ void __launch_bounds__(kMlpBlock, 4)  // (four waves per SIMD: the colour head's 130 registers allowed three)
NVO_MLP_NAME(k_mlp_fwd)(Args a) {
    // hidden activation: compile-time ReLU (every network on the NeRF-VO path) or the run-time switch
    const int hidden_act = RELU ? (int)NVO_ACT_RELU : a.act;
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, g = lane >> 4;
    const uint32_t wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * kWavesPerBlock;
    const uint32_t n_tiles = (a.n_live ? min(a.batch, (*a.n_live + 15u) & ~15u) : a.batch) >> 4;

    WFrag<WIDTH, IN_PAD> w0;
    WFrag<WIDTH, WIDTH> wh[N_HIDDEN > 1 ? N_HIDDEN - 1 : 1];
    WFrag<OUT_PAD, WIDTH> wl;
    {
        // The workgroup copies every matrix row-major into LDS with coalesced 8-byte loads (all requested before the
        // first store waits), each wave takes its fragments from there.  Loading the fragments straight from global
        // memory made the prologue the cost of a wave (colour head, 18 KB of weights: 21 us at 256 workgroups, 47 us at
        // 2048), which capped the grid at two waves per SIMD -- too few to hide the latency of the tile inputs.
        constexpr int kH0 = RowStage<WIDTH, IN_PAD>::kHalfs, kHh = RowStage<WIDTH, WIDTH>::kHalfs;
        __shared__ __attribute__((aligned(16))) T stage[kH0 + (N_HIDDEN - 1) * kHh + RowStage<OUT_PAD, WIDTH>::kHalfs];
        const T* W = a.weights;
        RowStage<WIDTH, IN_PAD> s0;
        RowStage<WIDTH, WIDTH> sh[N_HIDDEN > 1 ? N_HIDDEN - 1 : 1];
        RowStage<OUT_PAD, WIDTH> sl;
        s0.issue(W);
#pragma unroll
        for (int l = 0; l < N_HIDDEN - 1; ++l) sh[l].issue(W + WIDTH * IN_PAD + l * WIDTH * WIDTH);
        sl.issue(W + WIDTH * IN_PAD + (N_HIDDEN - 1) * WIDTH * WIDTH);
        s0.store(stage);
#pragma unroll
        for (int l = 0; l < N_HIDDEN - 1; ++l) sh[l].store(stage + kH0 + l * kHh);
        sl.store(stage + kH0 + (N_HIDDEN - 1) * kHh);
        __syncthreads();
        w0.load_staged(stage, lane);
#pragma unroll
        for (int l = 0; l < N_HIDDEN - 1; ++l) wh[l].load_staged(stage + kH0 + l * kHh, lane);
        wl.load_staged(stage + kH0 + (N_HIDDEN - 1) * kHh, lane);
    }

    // the next tile's input row is requested before the current tile is computed (see NVO_MLP_NAME(k_mlp_bwd))
    // (colour head) the camera index heads a dependent chain (index -> embedding row): requested one tile before the
    // tile's other inputs, so that the address of the embedding row never waits for a round trip inside the loop
    auto load_cam = [&](uint32_t tile) -> uint32_t {
        if constexpr (IO == NVO_IO_NERFACTO_COLOR) {
            if (a.cam_idx) return (uint32_t)a.cam_idx[(tile * 16 + m) / a.samples_per_ray];
        }
        return 0u;
    };
    auto load_x = [&](uint32_t tile, T4 (&xo)[IN_PAD / 16], uint32_t cam) {
        load_input<IN_PAD, IO>(a, tile * 16 + m, g, xo, cam);
    };
    // (requesting TWO tiles ahead measured slower: colour head 17.5 vs 16.6 us, 16-wide 8.7 vs 8.3)
    uint32_t cam_nxt = 0;
    T4 x[IN_PAD / 16];
    if constexpr (IO == NVO_IO_GRID_FUSED) {
        if (wave < n_tiles) {
            GridGather<IN_PAD> g0;
            grid_issue<IN_PAD>(a, wave * 16 + m, g, g0);
            grid_finish<IN_PAD>(a, g, g0, x);
        }
    } else {
        if (wave < n_tiles) {
            load_x(wave, x, load_cam(wave));
            cam_nxt = load_cam(min(wave + n_waves, n_tiles - 1u));
        }
    }
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t row = tile * 16 + m;
        T4 xn[IN_PAD / 16];
        GridGather<IO == NVO_IO_GRID_FUSED ? IN_PAD : 16> gn;
        const bool has_next = tile + n_waves < n_tiles;
        if constexpr (IO == NVO_IO_GRID_FUSED) {
            // the next tile's 16 table gathers per lane are in flight while this tile runs through the MFMA chain
            if (has_next) grid_issue<IN_PAD>(a, (tile + n_waves) * 16 + m, g, gn);
        } else {
            load_x(min(tile + n_waves, n_tiles - 1u), xn, cam_nxt);
            cam_nxt = load_cam(min(tile + 2u * n_waves, n_tiles - 1u));
        }

        f4 acc[WIDTH / 16];
        T4 h[WIDTH / 16];
        layer_mm<WIDTH, IN_PAD>(w0, x, acc);
#pragma unroll
        for (int t = 0; t < WIDTH / 16; ++t) h[t] = pack_act(hidden_act, acc[t]);
        if (a.hidden) {
            T* hs = a.hidden + (size_t)row * WIDTH + 4 * g;
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) *reinterpret_cast<T4*>(hs + 16 * t) = h[t];
        }
#pragma unroll
        for (int l = 0; l < N_HIDDEN - 1; ++l) {
            layer_mm<WIDTH, WIDTH>(wh[l], h, acc);
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) h[t] = pack_act(hidden_act, acc[t]);
            if (a.hidden) {
                T* hs = a.hidden + ((size_t)(l + 1) * a.batch + row) * WIDTH + 4 * g;
#pragma unroll
                for (int t = 0; t < WIDTH / 16; ++t) *reinterpret_cast<T4*>(hs + 16 * t) = h[t];
            }
        }
        f4 o[OUT_PAD / 16];
        layer_mm<OUT_PAD, WIDTH>(wl, h, o);
        if constexpr (COMPACT) {
            const T4 v = pack_act(a.out_act, o[0]);
            if (g == 0) a.output[row] = v[0];  // 16 lanes -> 32 contiguous bytes per tile
        } else {
            T* op = a.output + (size_t)row * OUT_PAD + 4 * g;
#pragma unroll
            for (int t = 0; t < OUT_PAD / 16; ++t)
                *reinterpret_cast<T4*>(op + 16 * t) = pack_act(a.out_act, o[t]);
        }
        if constexpr (IO == NVO_IO_GRID_FUSED) {
            if (has_next) grid_finish<IN_PAD>(a, g, gn, x);
        } else {
#pragma unroll
            for (int t = 0; t < IN_PAD / 16; ++t) x[t] = xn[t];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// wave-private LDS tile: 16 samples x LD halfs.  Row stride LD+4 halfs (8-B pad) keeps the
// 8-B alignment ds_read_b64_tr_b16 needs and staggers rows across banks.
template <int LD>
struct LdsTile {
    static constexpr int kStride = LD + 4;
    T* base;
    // chain layout -> tile: lane (m,g) owns [m][16t + 4g .. +3]
    __device__ __forceinline__ void store(int m, int g, int t, T4 v) const {
        *reinterpret_cast<T4*>(base + m * kStride + 16 * t + 4 * g) = v;
    }
    // transposed fragment: element j = tile[4g + j][16t + (lane & 15)]
    __device__ __forceinline__ T4 load_tr(int lane, int t) const {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const T* addr = base + (4 * g + q) * kStride + 16 * t + 4 * p;
        fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) fp16x4_t*)addr);
        T4 r;
        __builtin_memcpy(&r, &v, sizeof(r));
        return r;
    }
};

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave complete in issue order; this only stops the compiler from
    // moving the transposed reads above the tile stores (and vice versa on the next tile).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int N_OUT, int K_IN>
struct DwAcc {
    f4 a[N_OUT / 16][K_IN / 16];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int tn = 0; tn < N_OUT / 16; ++tn)
#pragma unroll
            for (int tk = 0; tk < K_IN / 16; ++tk) a[tn][tk] = f4{0.f, 0.f, 0.f, 0.f};
    }
    // dW[n][k] += sum_m dZ[m][n] H[m][k]   A = dZ^T fragment, B = H fragment (both transposed reads)
    __device__ __forceinline__ void accumulate(const T4 (&zt)[N_OUT / 16], const T4 (&ht)[K_IN / 16]) {
#pragma unroll
        for (int tn = 0; tn < N_OUT / 16; ++tn)
#pragma unroll
            for (int tk = 0; tk < K_IN / 16; ++tk) a[tn][tk] = mfma16(zt[tn], ht[tk], a[tn][tk]);
    }
    // accumulator (lane, reg r) = dW[16tn + 4g + r][16tk + (lane&15)]
    __device__ __forceinline__ void flush(float* __restrict__ dW, int lane) const {
        const int c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int tn = 0; tn < N_OUT / 16; ++tn)
#pragma unroll
            for (int tk = 0; tk < K_IN / 16; ++tk)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    atomicAdd(dW + (size_t)(16 * tn + 4 * g + r) * K_IN + 16 * tk + c, a[tn][tk][r]);
    }
    // Workgroup-level flush: the block's waves sum their accumulators in LDS (same lane <-> element
    // map in every wave, so plain read-modify-write phases separated by barriers suffice), then all
    // threads add the block total to global memory in row order (256 contiguous bytes per wave
    // instruction -- the fast float-atomic shape -- and 4x fewer adds per address).
    // MUST be called by every wave of the block (contains __syncthreads()).
    __device__ __forceinline__ void flush_block(float* __restrict__ dW, float* red, int lane, int wib,
                                                float* __restrict__ partial = nullptr, uint32_t* nf_flag = nullptr,
                                                int tid = -1) const {
        if (tid < 0) tid = (int)threadIdx.x;  // (role kernels pass the index among the kMlpBlock flushing threads)
        const int c = lane & 15, g = lane >> 4;
        for (int w = 0; w < kWavesPerBlock; ++w) {
            if (wib == w) {
#pragma unroll
                for (int tn = 0; tn < N_OUT / 16; ++tn)
#pragma unroll
                    for (int tk = 0; tk < K_IN / 16; ++tk)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int e = (16 * tn + 4 * g + r) * K_IN + 16 * tk + c;
                            red[e] = (w == 0 ? 0.f : red[e]) + a[tn][tk][r];
                        }
            }
            __syncthreads();
        }
        // (every block adds to the same addresses in the same order; starting each block at its own offset was measured
        // SLOWER -- 57.5 vs 56.1 us for the colour head -- the L2 handles the convoy better than a spread)
        uint32_t chk = 0u;  // exponent bits all ones: inf or NaN
        if (partial) {  // deterministic mode: the block total is STORED; a second launch sums the blocks in order
            for (int e = tid; e < N_OUT * K_IN; e += kMlpBlock) {
                partial[e] = red[e];
                chk |= (uint32_t)((__float_as_uint(red[e]) & 0x7f800000u) == 0x7f800000u);
            }
        } else {
            for (int e = tid; e < N_OUT * K_IN; e += kMlpBlock) {
                atomicAdd(dW + e, red[e]);
                chk |= (uint32_t)((__float_as_uint(red[e]) & 0x7f800000u) == 0x7f800000u);
            }
        }
        if (nf_flag && __ballot(chk != 0u) != 0ull && lane == 0) atomicOr(nf_flag, 1u);
        __syncthreads();
    }
};

// -DNVO_MLP_PHASE (debugging aid, never in the product build; NVO_EXTRA_CXXFLAGS of nerf_vo_amd/build.py): wave 0 of
// the colour-head backward sums the shader cycles it spends in each phase of the tile loop (s_memtime) and leaves them
// in nvo_mlp_phase_cycles (read by tools/mlp_phase.py through nvo_debug_mlp_phase).
#ifdef NVO_MLP_PHASE
__device__ unsigned long long NVO_MLP_NAME(nvo_mlp_phase_cycles)[16];
#define NVO_PH_DECL unsigned long long ph_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t_ = __builtin_amdgcn_s_memtime()
#define NVO_PH_START const unsigned long long ph_start_ = __builtin_amdgcn_s_memtime()
#define NVO_PH(k)                                                     \
    do {                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
        ph_[k] += t_ - ph_t_;                                         \
        ph_t_ = t_;                                                   \
    } while (0)
#else
#define NVO_PH_DECL
#define NVO_PH_START
#define NVO_PH(k)
#endif

// LDS halfs of the backward kernel (kernel and launcher): two transposing tiles per wave, or -- roles -- one tile set
// {dZ_L | H_l | dZ_l | X} per chain wave (8 of them) + the row-major copies of the matrices, which stay
constexpr int kLiveListCap = 2048;  // (roles) entries of a workgroup's live-tile list (16-bit codes)
constexpr int kLiveRowCap = 8192;   // (roles, level-major input) sample ids a workgroup collects for Args::live_rows
constexpr int bwd_lds_halfs(int in_pad, int width, int n_hidden, int out_pad, bool roles, bool rows = false, bool lists = false) {
    const int maxw = width > in_pad ? (width > out_pad ? width : out_pad) : (in_pad > out_pad ? in_pad : out_pad);
    const int stage = width * (in_pad + 4) + (n_hidden - 1) * width * (width + 4) + out_pad * (width + 4);
    const int set = 16 * (out_pad + 4) + 2 * n_hidden * 16 * (width + 4) + 16 * (in_pad + 4);
    // (+ 32 hand-over / list words: full[8], free[8], live and dead counts, 12 per-wave counts; + the live-tile list)
    if (roles) return 2 * kWavesPerBlock * set + stage + 64 + (lists ? kLiveListCap + (rows ? 2 * kLiveRowCap : 0) : 0);
    const int tiles = kWavesPerBlock * 2 * 16 * (maxw + 4);
    return tiles > stage ? tiles : stage;
}

// RECOMP (single-hidden-layer ReLU networks): the hidden activation is not read back from memory but
// recomputed from the input row with the same MFMA sequence and the same fp16 rounding as the forward (so it
// is bit-identical) -- the forward then does not store it at all.  64..128 bytes per sample less traffic in
// each direction for one to eight extra MFMAs per 16-sample tile.
//
// ROLES (the 64-wide networks): 12 waves per workgroup.  Waves 0-7 ("chain") hold the transposed weight fragments and run
// everything that is a chain through the layers -- recomputation, dZ, dX, epilogue; waves 8-11 ("dW") hold the fp32
// weight-gradient accumulators and do nothing but transposed LDS reads + the dZ^T H products of the tiles the chain waves
// hand them.  The dW role is MODEL-parallel: wave d owns a quarter of every matrix (36 accumulator registers) and reads
// all eight tile sets, so the chain role (~150 registers) decides the register count: THREE waves per SIMD where the
// single-role kernel (280 registers) runs one -- and one chain wave per SIMD is what bounds that kernel (~700
// instructions per 16-sample tile, half of the cycles stalled on its own dependencies; EXPERIMENTS.md 8.11).
// Hand-over: ONE tile set {dZ_L | H_l | dZ_l | X} per chain wave in LDS, written as the tiles are produced behind
// barrier B (after the recomputation: the dW waves are done with the previous contents) and released by barrier A at
// the end of the step; the dW waves read while the chain waves load and recompute the next tile.
// LISTS (roles only): the live-tile list of a trained field, see "LIVE-TILE LIST" below -- a separate instantiation, picked by
// the launcher when Args::tile_live is given: with the list code compiled in, the plain tile loop spilled scalar registers
// and lost 10-15 % on every launch (colour head 37.9 -> 44.4 us, base network 19.8 -> 22.3 us), list or no list.
template <int IN_PAD, int WIDTH, int N_HIDDEN, int OUT_PAD, int IO, bool RELU, bool COMPACT, bool RECOMP, bool ROLES = false, bool LISTS = false>
__global__ void __launch_bounds__(ROLES ? 3 * kMlpBlock : kMlpBlock)
NVO_MLP_NAME(k_mlp_bwd)(Args a) {
    static_assert(!LISTS || ROLES, "lists: the role-split form only");
    static_assert(!RECOMP || RELU, "hidden recomputation: ReLU networks");
    static_assert(!ROLES || (RECOMP && !COMPACT), "roles: recomputing, non-compact networks");
    constexpr int kChainWaves = ROLES ? 2 * kWavesPerBlock : kWavesPerBlock;
    NVO_PH_START;
    const int hidden_act = RELU ? (int)NVO_ACT_RELU : a.act;
    constexpr int MAXW = (WIDTH > IN_PAD ? (WIDTH > OUT_PAD ? WIDTH : OUT_PAD)
                                         : (IN_PAD > OUT_PAD ? IN_PAD : OUT_PAD));
    // LDS: two 16-row tiles per wave (transposes); before the first tile the same bytes stage ALL weight matrices
    constexpr int kTileHalfs = 16 * (MAXW + 4);
    constexpr int kStageHalfs = RowStage<WIDTH, IN_PAD>::kHalfs + (N_HIDDEN - 1) * RowStage<WIDTH, WIDTH>::kHalfs +
                                RowStage<OUT_PAD, WIDTH>::kHalfs;
    // ROLES: per chain wave ONE set of tiles {dZ_L | H_0..H_{n-1} | dZ_0..dZ_{n-1} | X} (each with its own row stride),
    // and behind the sets the row-major copies of the matrices, which stay (the chain reads its forward fragments there)
    constexpr int kHTile = 16 * (WIDTH + 4), kOffH = 16 * (OUT_PAD + 4), kOffDZ = kOffH + N_HIDDEN * kHTile,
                  kOffX = kOffDZ + N_HIDDEN * kHTile, kSetHalfs = kOffX + 16 * (IN_PAD + 4);
    constexpr int kTilesHalfs = ROLES ? kChainWaves * kSetHalfs : kWavesPerBlock * 2 * kTileHalfs;
    constexpr bool kRowsIo = LISTS && IO == NVO_IO_HALF2_SOA;  // (the kernels that may list live rows: Args::live_rows)
    constexpr int kLdsHalfs = ROLES ? kTilesHalfs + kStageHalfs + 64 + (LISTS ? kLiveListCap + (kRowsIo ? 2 * kLiveRowCap : 0) : 0)
                                    : (kTilesHalfs > kStageHalfs ? kTilesHalfs : kStageHalfs);
    static_assert(kLdsHalfs == bwd_lds_halfs(IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, ROLES, kRowsIo, LISTS), "launcher and kernel disagree on the LDS size");
    __shared__ __attribute__((aligned(16))) T lds_static[ROLES ? 8 : kLdsHalfs];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];  // (ROLES: 112 KB, opted in by the launcher)
    T* const lds = ROLES ? reinterpret_cast<T*>(lds_dyn) : lds_static;
    // (roles) hand-over of a tile set.  Two hidden layers (kHandFlags): words in LDS -- full[c] = steps chain wave c has
    // completed writing, free[c] = reads of set c the dW waves have finished (4 per step), release / acquire at workgroup
    // scope, no barrier inside the tile loop: two workgroup barriers per step kept the eight chain waves in lockstep and
    // cost ~2500 of a chain wave's 8100 cycles per tile (colour head 52.3 -> 41.4 us).  One hidden layer: the two barriers
    // (short tiles, light dW role: 22.9 us with barriers, 27.3 with the words).
    constexpr bool kHandFlags = ROLES && N_HIDDEN >= 2;
    uint32_t* const hand_full = reinterpret_cast<uint32_t*>(lds + (ROLES ? kTilesHalfs + kStageHalfs : 0));
    uint32_t* const hand_free = hand_full + 8;
    if constexpr (ROLES) {
        if (threadIdx.x < 32) hand_full[threadIdx.x] = 0u;  // (visible behind the prologue's barriers)
    }
    // (roles) LIVE-TILE LIST (round 6).  On a trained field one sample of a ray's 48 carries the weight: 98 % of the main
    // field's dL/d(rgb) rows and two of a ray's three 16-sample tiles are exactly zero (tools/probes/dead_tiles.py), and
    // testing for that inside the tile loop costs a memory round trip per dead tile (the next tile's inputs are not there
    // yet; measured: colour head 36.9 -> 32.7 us only).  The kernel that WRITES dL/doutput knows: Args::tile_live, one
    // byte per tile.  The workgroup compacts the live ones of the tiles it owns (code = 8 step + chain wave, in order,
    // so that the deterministic mode keeps a fixed order) into live_list, every chain wave takes list entry 8 s + c in its
    // step s -- all eight stay busy, the hand-over is untouched, only the number of steps shrinks -- and the dead tiles'
    // dX is stored as zeros up front.
    uint32_t* const list_words = hand_full + 16;  // [0] per-wave counts of a round: live | dead << 16 (12 words)
    uint16_t* const live_list = reinterpret_cast<uint16_t*>(hand_full + 32);  // live codes from the front, dead ones from the back
    // (level-major input) Args::live_rows: ids of the samples with a non-zero dL/doutput row, collected here and appended to
    // the global list with ONE atomic per workgroup at the end; list_words[12] = their number, [13] = the reserved position
    uint32_t* const wg_rows = reinterpret_cast<uint32_t*>(live_list + kLiveListCap);

    const int lane = threadIdx.x & 63;
    const int m = lane & 15, g = lane >> 4;
    const int wib_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool is_dw = ROLES && wib_all >= kChainWaves;  // waves 8-11
    const int wib = is_dw ? wib_all - kChainWaves : wib_all;  // chain index 0..7 | dW index 0..3
    const uint32_t wave = blockIdx.x * kChainWaves + wib;     // (chain role) first tile
    const uint32_t n_waves = gridDim.x * kChainWaves;
    const uint32_t n_tiles = a.batch >> 4;
    const LdsTile<MAXW> tz{lds + (2 * wib) * kTileHalfs}, th{lds + (2 * wib + 1) * kTileHalfs};

    const bool need_dinput = a.dinput != nullptr;
    // (roles) the tiles this workgroup owns: code = 8 step + chain wave <-> tile blockIdx.x * 8 + (code & 7) + (code >> 3) * n_waves
    const uint32_t n_iter_all = ROLES ? (n_tiles + n_waves - 1u) / n_waves : 0u;
    const uint32_t n_own = n_iter_all * kChainWaves;
    constexpr bool kListIo = LISTS && (IO == NVO_IO_HALF2_SOA || (IO == NVO_IO_NERFACTO_COLOR && IN_PAD == 64));
    // (kernel-uniform; decided behind the prologue's round trip: Args::tile_live_count is requested with the matrices)
    const bool list_pre = kListIo && a.tile_live != nullptr && n_own <= (uint32_t)kLiveListCap;
    bool use_list = list_pre;
    float live_tiles_part = 0.f;
    if constexpr (kListIo) {
        if (list_pre && a.tile_live_count) live_tiles_part = a.tile_live_count[8 * lane];
    }
    auto own_tile = [&](uint32_t code) -> uint32_t { return blockIdx.x * kChainWaves + (code & (kChainWaves - 1u)) + (code >> 3) * n_waves; };
    auto tile_flag = [&](uint32_t code) -> uint32_t {  // 0 not a tile | 1 live | 2 dead
        const uint32_t t = own_tile(code);
        if (code >= n_own || t >= n_tiles) return 0u;
        return (a.tile_live[t] & a.tile_live_bits) != 0u ? 1u : 2u;
    };
    // (the bytes of the first round are requested with the matrices -- one round trip -- unless a count of the live tiles
    // is at hand: then they wait for the decision; requesting them early regardless measured the same)
    uint32_t flag0 = 0u;
    if constexpr (kListIo) {
        if (list_pre && !a.tile_live_count) flag0 = tile_flag(threadIdx.x);
    }

    // transposed weights for the dH chain (layer 0's only if dL/dinput is wanted)
    WTFrag<WIDTH, IN_PAD> wt0;
    WTFrag<WIDTH, WIDTH> wth[N_HIDDEN > 1 ? N_HIDDEN - 1 : 1];
    WTFrag<OUT_PAD, WIDTH> wtl;
    // Prologue in ONE memory round trip: the row-major copies of every matrix (for the transposed fragments), the
    // forward fragments of the recomputation and the first tile's camera index are all requested before anything
    // waits.  (It used to be a round trip per matrix -- copy, barrier, fragment reads, barrier -- then the forward
    // fragments, then camera index -> embedding row: ~18 K cycles of the colour head's 110 K, tools/mlp_phase.py.)
    WFrag<WIDTH, IN_PAD> w0f;  // forward weights (hidden recomputation only)
    WFrag<WIDTH, WIDTH> whf[N_HIDDEN > 1 ? N_HIDDEN - 1 : 1];
    uint32_t cam_first = 0;
    {
        const T* W = a.weights;
        RowStage<WIDTH, IN_PAD> s0;
        RowStage<WIDTH, WIDTH> sh[N_HIDDEN > 1 ? N_HIDDEN - 1 : 1];
        RowStage<OUT_PAD, WIDTH> sl;
        // (the recomputation's forward fragments come from the same row-major copies: W0 is staged for them alone when
        // no dL/dinput is wanted)
        const bool stage_w0 = need_dinput || RECOMP;
        const bool stager = !ROLES || threadIdx.x < kMlpBlock;  // (RowStage strides by kMlpBlock threads)
        T* stage = lds + (ROLES ? kTilesHalfs : 0);  // (single role: the wave tiles are idle until the first sample tile)
        T* stage_h = stage + RowStage<WIDTH, IN_PAD>::kHalfs;
        T* stage_l = stage_h + (N_HIDDEN - 1) * RowStage<WIDTH, WIDTH>::kHalfs;
        if (stager) {
            if (stage_w0) s0.issue(W);
#pragma unroll
            for (int l = 0; l < N_HIDDEN - 1; ++l) sh[l].issue(W + WIDTH * IN_PAD + l * WIDTH * WIDTH);
            sl.issue(W + WIDTH * IN_PAD + (N_HIDDEN - 1) * WIDTH * WIDTH);
        }
        if constexpr (IO == NVO_IO_NERFACTO_COLOR) {
            if (a.cam_idx && wave < n_tiles && !is_dw) cam_first = (uint32_t)a.cam_idx[(wave * 16 + m) / a.samples_per_ray];
        }
        if (stager) {
            if (stage_w0) s0.store(stage);
#pragma unroll
            for (int l = 0; l < N_HIDDEN - 1; ++l) sh[l].store(stage_h + l * RowStage<WIDTH, WIDTH>::kHalfs);
            sl.store(stage_l);
        }
        __syncthreads();
        if (!is_dw) {
            if (need_dinput) wt0.read_staged(stage, lane);
#pragma unroll
            for (int l = 0; l < N_HIDDEN - 1; ++l) wth[l].read_staged(stage_h + l * RowStage<WIDTH, WIDTH>::kHalfs, lane);
            wtl.read_staged(stage_l, lane);
            if constexpr (RECOMP && !ROLES) {  // (roles: the forward fragments are read from the copies for every tile)
                w0f.load_staged(stage, lane);
#pragma unroll
                for (int l = 0; l < N_HIDDEN - 1; ++l) whf[l].load_staged(stage_h + l * RowStage<WIDTH, WIDTH>::kHalfs, lane);
            }
        }
        __syncthreads();  // the staging bytes become the wave tiles
    }
    uint32_t n_live = 0u;  // (list) live tiles of this workgroup
    if constexpr (kListIo) {
        if (list_pre && a.tile_live_count) {  // while 3/4 of the tiles or more are live: no list, every tile in its turn
            float c = live_tiles_part;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
            use_list = c < (float)(n_tiles - (n_tiles >> 2));
            if (use_list) flag0 = tile_flag(threadIdx.x);
        }
        if constexpr (kRowsIo) {
            if (a.live_rows && !use_list && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.live_rows_n, a.batch);  // "all samples"
        }
        if (use_list) {
            uint32_t n_dead = 0u;
            const unsigned long long lt = (1ull << lane) - 1ull;
            for (uint32_t base = 0; base < n_own; base += 3 * kMlpBlock) {
                const uint32_t code = base + threadIdx.x;
                const uint32_t flag = base == 0u ? flag0 : tile_flag(code);
                const unsigned long long bl = __ballot(flag == 1u), bd = __ballot(flag == 2u);
                if (lane == 0) list_words[wib_all] = (uint32_t)__popcll(bl) | ((uint32_t)__popcll(bd) << 16);
                __syncthreads();
                uint32_t pl = n_live, pd = n_dead;
#pragma unroll
                for (int w = 0; w < 3 * kWavesPerBlock; ++w) {
                    const uint32_t c = list_words[w];
                    if (w < wib_all) {
                        pl += c & 0xffffu;
                        pd += c >> 16;
                    }
                    n_live += c & 0xffffu;
                    n_dead += c >> 16;
                }
                if (flag == 1u) live_list[pl + (uint32_t)__popcll(bl & lt)] = (uint16_t)code;
                if (flag == 2u) live_list[kLiveListCap - 1u - (pd + (uint32_t)__popcll(bd & lt))] = (uint16_t)code;
                __syncthreads();  // the list is complete up to here; the count words are free again
            }
            // (colour head) the camera index of this chain wave's first listed tile heads a dependent chain -- index ->
            // embedding row -> the tile's loads: requested here, it arrives while the zeros below are stored
            if constexpr (IO == NVO_IO_NERFACTO_COLOR) {
                if (a.cam_idx && !is_dw && n_live > (uint32_t)wib)
                    cam_first = (uint32_t)a.cam_idx[(own_tile(live_list[wib]) * 16 + m) / a.samples_per_ray];
            }
            // dX of the dead tiles: zeros (all twelve waves, a tile per wave and turn; stores only)
            if (need_dinput) {
                const T z = (T)0.f;
                for (uint32_t d = (uint32_t)wib_all; d < n_dead; d += 3u * kWavesPerBlock) {
                    const uint32_t tile = own_tile(live_list[kLiveListCap - 1u - d]);
                    const uint32_t row = tile * 16 + m;
                    if constexpr (IO == NVO_IO_HALF2_SOA) {
                        T2* __restrict__ p = (T2*)a.dinput;
                        const uint32_t n_lv = a.n_in >> 1;
#pragma unroll
                        for (int tk = 0; tk < IN_PAD / 16; ++tk) {
                            const uint32_t lv = 8 * tk + 2 * g;
                            if (lv < n_lv) p[(size_t)lv * a.batch + row] = T2{z, z};
                            if (lv + 1 < n_lv) p[(size_t)(lv + 1) * a.batch + row] = T2{z, z};
                        }
                    } else {
                        T* __restrict__ dbo = a.d_base_out + (size_t)row * 16;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (4 * g + j < 15) dbo[1 + 4 * g + j] = z;
                        if (a.tile_partial && lane < 48) a.tile_partial[(size_t)tile * 48 + lane] = 0.f;  // (deterministic mode)
                    }
                }
            }
        }
    }
    DwAcc<ROLES ? 16 : WIDTH, ROLES ? 16 : IN_PAD> dw0;  // (roles: the dW role declares its own, see there)
    DwAcc<ROLES ? 16 : WIDTH, ROLES ? 16 : WIDTH> dwh[N_HIDDEN > 1 ? N_HIDDEN - 1 : 1];
    DwAcc<ROLES ? 16 : OUT_PAD, ROLES ? 16 : WIDTH> dwl;
    // (level-major dinput) L1 norm of the 16-bit dL/dinput values this lane stores, per column 16 tk + 4 g + j, and the
    // wave's count of samples with a non-zero dL/doutput: Args::dx_l1_partial / dx_live_partial
    float l1a[IN_PAD / 16][4];
    uint32_t live_cnt = 0u;
#pragma unroll
    for (int tk = 0; tk < IN_PAD / 16; ++tk)
#pragma unroll
        for (int j = 0; j < 4; ++j) l1a[tk][j] = 0.f;
    if constexpr (!ROLES) {
        dw0.zero();
#pragma unroll
        for (int l = 0; l < N_HIDDEN - 1; ++l) dwh[l].zero();
        dwl.zero();
    }

    // Every global input of a tile (dL/dout, out, all hidden activations, the input row) is requested in one
    // go, and the NEXT tile's inputs are requested before the current tile is computed: with the dW
    // accumulators in registers a wave has its SIMD to itself (346 registers for the colour head), so memory
    // latency can only be hidden inside the wave.  Before this the tile loop paid three dependent HBM round
    // trips per tile (measured: 64 % of the wave cycles in s_waitcnt, 6.9 us per 16-sample tile).
    struct TileIn {
        T4 dzl[OUT_PAD / 16], out[OUT_PAD / 16];
        T4 hs[N_HIDDEN][WIDTH / 16];
        T4 x[IN_PAD / 16];
        uint32_t cam;  // (colour head) appearance-embedding row of this lane's sample
    };
    // output-activation derivative as arithmetic on wave-uniform coefficients (no per-element branches):
    // factor = 1 + c_sig * (o (1 - o) - 1) + c_relu * (step(o) - 1)
    const float c_sig = a.out_act == NVO_ACT_SIGMOID ? 1.f : 0.f, c_relu = a.out_act == NVO_ACT_RELU ? 1.f : 0.f;
    // The camera index heads a dependent chain (index -> embedding row).  With ONE wave per SIMD a dependent round trip
    // inside the tile loop stalls the whole SIMD (it used to: once in load_tile for the embedding address and once more
    // in the epilogue for the gradient's address -- half of the 55 % of wave cycles the counters showed in s_waitcnt),
    // so the index is requested a tile earlier than the tile's other inputs and carried along with them.
    auto load_cam = [&](uint32_t tile) -> uint32_t {
        if constexpr (IO == NVO_IO_NERFACTO_COLOR) {
            if (a.cam_idx) return (uint32_t)a.cam_idx[(tile * 16 + m) / a.samples_per_ray];
        }
        return 0u;
    };
    auto load_tile = [&](uint32_t tile, TileIn& t, uint32_t cam) {
        const uint32_t row = tile * 16 + m;
        t.cam = cam;
        if constexpr (COMPACT) {
            const T z = (T)0.f;
            const T dv = a.doutput[row], ov = a.output[row];  // every lane group reads, g == 0 keeps
#pragma unroll
            for (int i = 0; i < OUT_PAD / 16; ++i) {
                t.dzl[i] = T4{z, z, z, z};
                t.out[i] = T4{z, z, z, z};
            }
            if (g == 0) {
                t.dzl[0][0] = dv;
                t.out[0][0] = ov;
            }
        } else {
            const T* dp = a.doutput + (size_t)row * OUT_PAD + 4 * g;
            const T* op = a.output + (size_t)row * OUT_PAD + 4 * g;
#pragma unroll
            for (int i = 0; i < OUT_PAD / 16; ++i) {
                t.dzl[i] = *reinterpret_cast<const T4*>(dp + 16 * i);
                t.out[i] = *reinterpret_cast<const T4*>(op + 16 * i);
            }
        }
        if constexpr (!RECOMP) {
#pragma unroll
            for (int l = 0; l < N_HIDDEN; ++l) {
                const T* hp = a.hidden + ((size_t)l * a.batch + row) * WIDTH + 4 * g;
#pragma unroll
                for (int i = 0; i < WIDTH / 16; ++i) t.hs[l][i] = *reinterpret_cast<const T4*>(hp + 16 * i);
            }
        }
        load_input<IN_PAD, IO>(a, row, g, t.x, cam);
    };
    // step s of this (chain) wave: its s-th tile, clamped to a tile that exists (prefetches past the end)
    auto step_tile = [&](uint32_t st) -> uint32_t {
        if constexpr (kListIo) {
            if (use_list)
                return own_tile((uint32_t)__builtin_amdgcn_readfirstlane((int)live_list[min(st * kChainWaves + (uint32_t)wib, n_live - 1u)]));
        }
        return min(wave + st * n_waves, n_tiles - 1u);
    };
    const uint32_t my_steps = is_dw ? 0u
                              : use_list ? (n_live > (uint32_t)wib ? (n_live - (uint32_t)wib + kChainWaves - 1u) / kChainWaves : 0u)
                                         : (wave < n_tiles ? (n_tiles - wave + n_waves - 1u) / n_waves : 0u);
    TileIn cur;
    uint32_t cam_nxt = 0;
    uint32_t t_cur = 0u, t_nxt = 0u;
    if (my_steps) {
        t_cur = step_tile(0u);
        t_nxt = step_tile(1u);
        load_tile(t_cur, cur, cam_first);  // (list form: cam_first was re-requested for the first LISTED tile, see there)
        cam_nxt = load_cam(t_nxt);
    }
    // Nothing issued before the loop may still be pending when it starts: the compiler's waits for such loads (weight
    // fragments, the first tile) would sit INSIDE the loop as s_waitcnt vmcnt(N) with N counted along the entry path,
    // and in steady state such a count also drains the previous tile's stores and float atomics.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    NVO_PH_DECL;
#ifdef NVO_MLP_PHASE
    ph_[10] = ph_t_ - ph_start_;  // prologue: weight staging, fragment loads, first tile
#endif
    // ---- roles: step `it` of chain wave c handles tile blockIdx.x * 8 + c + it * n_waves; every wave of the workgroup
    // walks n_iter steps; hand-over per tile set through the words hand_full / hand_free (no barrier in the loop)
    const uint32_t n_iter = use_list ? (n_live + kChainWaves - 1u) / kChainWaves : n_iter_all;
    if (is_dw) {
        // MODEL-parallel: dW wave d owns rows 16 d .. 16 d + 15 of dW_0 and of every hidden dW, and column tile d of the
        // output layer's dW -- 36 accumulator registers instead of 144, so the CHAIN role decides the kernel's register
        // count -- and reads what it needs of ALL chain waves' tile sets (12 transposed fragments per tile).
        static_assert(!ROLES || (OUT_PAD == 16 && WIDTH == 64), "roles: 64-wide networks with a 16-wide (padded) output");
        f4 a0[IN_PAD / 16], ah[N_HIDDEN > 1 ? N_HIDDEN - 1 : 1][WIDTH / 16], al = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < IN_PAD / 16; ++t) a0[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l < N_HIDDEN - 1; ++l)
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) ah[l][t] = f4{0.f, 0.f, 0.f, 0.f};
        for (uint32_t it = 0; it < n_iter; ++it) {
            if constexpr (!kHandFlags) {
                __syncthreads();  // B: (this role is done with the previous sets)
                __syncthreads();  // A: the chain waves have written the sets of step `it`
            }
            for (int k = 0; k < kChainWaves; ++k) {
                const int c = (2 * wib + k) & (kChainWaves - 1);  // (every dW wave starts at another set)
                if (use_list ? it * kChainWaves + c >= n_live : blockIdx.x * kChainWaves + c + it * n_waves >= n_tiles) continue;  // (wave-uniform)
                if constexpr (kHandFlags) {
                    // (taking whichever set is ready first instead of this fixed order measured slower: 42.7 vs 41.4 us)
                    while (__hip_atomic_load(hand_full + c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= it)
                        __builtin_amdgcn_s_sleep(1);
                }
                T* const set = lds + (size_t)(c * kSetHalfs);
                // output layer: dW_l[n][16 d + k] += sum_m dZ_L[m][n] H_{n-1}[m][16 d + k]
                al = mfma16(LdsTile<OUT_PAD>{set}.load_tr(lane, 0), LdsTile<WIDTH>{set + kOffH + (N_HIDDEN - 1) * kHTile}.load_tr(lane, wib), al);
#pragma unroll
                for (int l = N_HIDDEN - 1; l >= 1; --l) {  // dW_h[l-1][16 d + r][k] += dZ_l^T H_{l-1}
                    const T4 zt = LdsTile<WIDTH>{set + kOffDZ + l * kHTile}.load_tr(lane, wib);
                    const LdsTile<WIDTH> thh{set + kOffH + (l - 1) * kHTile};
#pragma unroll
                    for (int t = 0; t < WIDTH / 16; ++t) ah[l - 1][t] = mfma16(zt, thh.load_tr(lane, t), ah[l - 1][t]);
                }
                {
                    const T4 zt = LdsTile<WIDTH>{set + kOffDZ}.load_tr(lane, wib);
                    const LdsTile<IN_PAD> tx{set + kOffX};
#pragma unroll
                    for (int t = 0; t < IN_PAD / 16; ++t) a0[t] = mfma16(zt, tx.load_tr(lane, t), a0[t]);
                }
                if constexpr (kHandFlags) {
                    if (lane == 0) __hip_atomic_fetch_add(hand_free + c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (a.dweights) {
            // every entry of dW is owned by ONE wave of the workgroup (no reduction across waves); accumulator (lane, r) =
            // dW[16 tn + 4 g + r][16 tk + (lane & 15)].  The entries still go through LDS so that the adds leave in ROW
            // order: 256 contiguous bytes per wave instruction is the fast float-atomic shape, 16 x 64-byte pieces
            // straight from the accumulator layout are not (the direct form cost as much as the whole tile loop).
            constexpr int kWeights = WIDTH * IN_PAD + (N_HIDDEN - 1) * WIDTH * WIDTH + OUT_PAD * WIDTH;
            float* red = reinterpret_cast<float*>(lds);  // (the tile sets are idle behind the barrier)
            static_assert(!ROLES || sizeof(float) * kWeights <= sizeof(T) * (size_t)kTilesHalfs, "dW staging does not fit the tile sets");
            __syncthreads();
#pragma unroll
            for (int t = 0; t < IN_PAD / 16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(16 * wib + 4 * g + r) * IN_PAD + 16 * t + m] = a0[t][r];
            float* rh = red + WIDTH * IN_PAD;
#pragma unroll
            for (int l = 0; l < N_HIDDEN - 1; ++l) {
#pragma unroll
                for (int t = 0; t < WIDTH / 16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rh[(16 * wib + 4 * g + r) * WIDTH + 16 * t + m] = ah[l][t][r];
                rh += WIDTH * WIDTH;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) rh[(4 * g + r) * WIDTH + 16 * wib + m] = al[r];
            __syncthreads();
            float* part = a.dw_partial ? a.dw_partial + (size_t)blockIdx.x * kWeights : nullptr;
            float* dW = a.dweights;
            if (a.dw_replicas) {  // (see NvoMlpArgsT::dw_replicas)
                const uint32_t r = blockIdx.x % (a.dw_n_replicas + 1u);
                if (r) dW = a.dw_replicas + (size_t)(r - 1u) * kWeights;
            }
            uint32_t chk = 0u;
            for (int e = (int)threadIdx.x - kChainWaves * 64; e < kWeights; e += kMlpBlock) {  // (the dW waves: threads 512..767)
                const float v = red[e];
                chk |= (uint32_t)((__float_as_uint(v) & 0x7f800000u) == 0x7f800000u);
                if (part) part[e] = v; else atomicAdd(dW + e, v);
            }
            if (a.nf_flag && __ballot(chk != 0u) != 0ull && lane == 0) atomicOr(a.nf_flag, 1u);
        }
    }
    uint32_t it = 0u;  // (roles: step counter of the chain role)
    T* const set = lds + (size_t)((ROLES ? wib : 0) * kSetHalfs);  // (roles) this chain wave's tile set
    // The tile loop, instantiated TWICE for the kernels that can walk a live-tile list: the list form carries the list state
    // (its length, the tile codes in LDS, the row buffer of Args::live_rows) through the loop, and with it in ONE loop the
    // plain form -- every tile live, the untrained field, every step of the headline measurement -- spilled scalar registers
    // (35 v_writelane / 53 v_readlane in the colour head: 37.9 -> 44.4 us; base network 19.8 -> 22.3 us).
    auto step_tile_c = [&](auto list_c, uint32_t st_) -> uint32_t {
        if constexpr (decltype(list_c)::value)
            return own_tile((uint32_t)__builtin_amdgcn_readfirstlane((int)live_list[min(st_ * kChainWaves + (uint32_t)wib, n_live - 1u)]));
        else
            return min(wave + st_ * n_waves, n_tiles - 1u);
    };
    auto chain_loop = [&](auto list_c) __attribute__((always_inline)) {
    constexpr bool kList = decltype(list_c)::value;
    // (the plain form keeps the loop it always had -- a tile index striding by the wave count: the list form's step
    // counter and carried tile indices cost the base network's plain kernel 1 us of its 19)
    for (uint32_t st = 0, tile_p = wave; kList ? st < my_steps : (tile_p < n_tiles && !is_dw); ++st, tile_p += n_waves) {
        const uint32_t tile = kList ? t_cur : tile_p;
        const uint32_t row = tile * 16 + m;
        NVO_PH(9);
        TileIn nxt;  // unconditional (clamped) so that no join forces the loads to complete here
        if constexpr (kList) {
            load_tile(t_nxt, nxt, cam_nxt);
            t_cur = t_nxt;
            t_nxt = step_tile_c(list_c, st + 2u);
            cam_nxt = load_cam(t_nxt);
        } else {
            load_tile(min(tile + n_waves, n_tiles - 1u), nxt, cam_nxt);
            cam_nxt = load_cam(min(tile + 2u * n_waves, n_tiles - 1u));
        }
        NVO_PH(0);
        if constexpr (kRowsIo) {
            if (kList && a.live_rows) {  // (kernel-uniform) list this tile's samples with a non-zero dL/doutput row
                bool nz = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) nz = nz || (float)cur.dzl[0][j] != 0.f;  // (NaN != 0: non-finite rows stay listed)
                const unsigned long long b = __ballot(nz);
                const uint32_t m16 = (uint32_t)((b | (b >> 16) | (b >> 32) | (b >> 48)) & 0xffffull);  // row m: any of its 4 lanes
                const uint32_t cnt = (uint32_t)__popc(m16);
                if (cnt) {
                    uint32_t base = 0u;
                    if (lane == 0) base = atomicAdd(&list_words[12], cnt);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (g == 0 && ((m16 >> m) & 1u)) wg_rows[base + (uint32_t)__popc(m16 & ((1u << m) - 1u))] = row;
                }
            }
        }
        if constexpr (COMPACT && IO == NVO_IO_HALF2_SOA) {
            // A tile whose 16 dL/dout values are all EXACTLY zero contributes nothing to any dW and its dX is zero: skip
            // the chain.  (The proposal networks of a nerfacto run: from a few hundred steps on 80-93 % of level 0's
            // tiles and 46-68 % of level 1's -- the interlevel loss is zero wherever the proposal weights stay under
            // the bound and fp16 flushes what is left below 6e-8 -- DESIGN.md section 7.1.)
            const unsigned long long live_m = __ballot((float)cur.dzl[0][0] != 0.f);  // (lanes g != 0 hold zeros)
            live_cnt += (uint32_t)__popcll(live_m);
            if (live_m == 0ull) {
                if (need_dinput) {
                    T2* __restrict__ p = (T2*)a.dinput;
                    const uint32_t n_lv = a.n_in >> 1;
                    const T z = (T)0.f;
#pragma unroll
                    for (int tk = 0; tk < IN_PAD / 16; ++tk) {
                        const uint32_t lv = 8 * tk + 2 * g;
                        if (lv < n_lv) p[(size_t)lv * a.batch + row] = T2{z, z};
                        if (lv + 1 < n_lv) p[(size_t)(lv + 1) * a.batch + row] = T2{z, z};
                    }
                }
                cur = nxt;
                continue;
            }
        }

        // ---- output layer: dZ_L = dL/dout * out_act'(out)
        T4 dzl[OUT_PAD / 16];
#pragma unroll
        for (int t = 0; t < OUT_PAD / 16; ++t) {
            T4 d = cur.dzl[t];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float o = (float)cur.out[t][j];
                const float f = 1.f + c_sig * (o * (1.f - o) - 1.f) + c_relu * ((o > 0.f ? 1.f : 0.f) - 1.f);
                d[j] = (T)((float)d[j] * f);
            }
            dzl[t] = d;
        }
        // last hidden activation H_{N_HIDDEN-1}
        T4 h[WIDTH / 16];
        if constexpr (RECOMP) {  // forward chain again: identical MFMA order and fp16 rounding
            f4 hacc[WIDTH / 16];
            const T* const stage_w = lds + (ROLES ? kTilesHalfs : 0);  // (roles) row-major W0 | W_h .. behind the tile sets
            if constexpr (ROLES) layer_mm_lds<WIDTH, IN_PAD>(stage_w, lane, cur.x, hacc);
            else layer_mm<WIDTH, IN_PAD>(w0f, cur.x, hacc);
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) cur.hs[0][t] = pack_act(NVO_ACT_RELU, hacc[t]);
#pragma unroll
            for (int l = 1; l < N_HIDDEN; ++l) {
                if constexpr (ROLES)
                    layer_mm_lds<WIDTH, WIDTH>(stage_w + RowStage<WIDTH, IN_PAD>::kHalfs + (l - 1) * RowStage<WIDTH, WIDTH>::kHalfs,
                                               lane, cur.hs[l - 1], hacc);
                else
                    layer_mm<WIDTH, WIDTH>(whf[l - 1], cur.hs[l - 1], hacc);
#pragma unroll
                for (int t = 0; t < WIDTH / 16; ++t) cur.hs[l][t] = pack_act(NVO_ACT_RELU, hacc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < WIDTH / 16; ++t) h[t] = cur.hs[N_HIDDEN - 1][t];
        NVO_PH(1);
        // dW_last += dZ_L^T H   (roles: all dW products belong to the dW waves; the tiles go to this wave's set as they are
        // produced, behind barrier B: the dW waves are done with the previous step's sets)
        if constexpr (ROLES) {
            // B: the four dW waves are done with what step it - 1 left in this wave's set
            if constexpr (kHandFlags) {
                while (__hip_atomic_load(hand_free + wib, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * it)
                    __builtin_amdgcn_s_sleep(1);
            } else {
                __syncthreads();
            }
            const LdsTile<OUT_PAD> tzl{set};
            const LdsTile<WIDTH> thl{set + kOffH + (N_HIDDEN - 1) * kHTile};
#pragma unroll
            for (int t = 0; t < OUT_PAD / 16; ++t) tzl.store(m, g, t, dzl[t]);
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) thl.store(m, g, t, h[t]);
            // (everything the recomputation produced goes now: the input tile and the lower hidden activations leave the
            // registers -- H_{l-1} is read back from this wave's own set when its dZ needs it)
            const LdsTile<IN_PAD> tx{set + kOffX};
#pragma unroll
            for (int t = 0; t < IN_PAD / 16; ++t) tx.store(m, g, t, cur.x[t]);
#pragma unroll
            for (int l = 0; l < N_HIDDEN - 1; ++l) {
                const LdsTile<WIDTH> thh{set + kOffH + l * kHTile};
#pragma unroll
                for (int t = 0; t < WIDTH / 16; ++t) thh.store(m, g, t, cur.hs[l][t]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < OUT_PAD / 16; ++t) tz.store(m, g, t, dzl[t]);
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) th.store(m, g, t, h[t]);
            wave_lds_sync();
            T4 zt[OUT_PAD / 16], ht[WIDTH / 16];
#pragma unroll
            for (int t = 0; t < OUT_PAD / 16; ++t) zt[t] = tz.load_tr(lane, t);
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) ht[t] = th.load_tr(lane, t);
            dwl.accumulate(zt, ht);
            wave_lds_sync();
        }
        NVO_PH(2);
        // dZ of the last hidden layer
        T4 dz[WIDTH / 16];
        {
            f4 acc[WIDTH / 16];
            layer_mm_t<OUT_PAD, WIDTH>(wtl, dzl, acc);
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dz[t][j] = (T)(acc[t][j] * act_bwd_from_out(hidden_act, (float)h[t][j]));
        }
        NVO_PH(3);
        // ---- hidden layers N_HIDDEN-1 .. 1 (weights wh[l-1] map H_{l-1} -> H_l)
#pragma unroll
        for (int l = N_HIDDEN - 1; l >= 1; --l) {
            T4 hp_[WIDTH / 16];  // H_{l-1}
            if constexpr (ROLES) {
                const LdsTile<WIDTH> tzz{set + kOffDZ + l * kHTile};
                const T* hrow = set + kOffH + (l - 1) * kHTile + m * (WIDTH + 4) + 4 * g;  // (this lane's own stores)
#pragma unroll
                for (int t = 0; t < WIDTH / 16; ++t) {
                    tzz.store(m, g, t, dz[t]);
                    hp_[t] = *reinterpret_cast<const T4*>(hrow + 16 * t);
                }
            } else {
#pragma unroll
                for (int t = 0; t < WIDTH / 16; ++t) hp_[t] = cur.hs[l - 1][t];
            }
            if constexpr (!ROLES) {
#pragma unroll
                for (int t = 0; t < WIDTH / 16; ++t) {
                    tz.store(m, g, t, dz[t]);
                    th.store(m, g, t, hp_[t]);
                }
                wave_lds_sync();
                {
                    T4 zt[WIDTH / 16], ht[WIDTH / 16];
#pragma unroll
                    for (int t = 0; t < WIDTH / 16; ++t) {
                        zt[t] = tz.load_tr(lane, t);
                        ht[t] = th.load_tr(lane, t);
                    }
                    dwh[l - 1].accumulate(zt, ht);
                }
                wave_lds_sync();
            }
            f4 acc[WIDTH / 16];
            layer_mm_t<WIDTH, WIDTH>(wth[l - 1], dz, acc);
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dz[t][j] = (T)(acc[t][j] * act_bwd_from_out(hidden_act, (float)hp_[t][j]));
        }
        NVO_PH(4);
        // ---- first layer: dW0 += dZ_0^T X, dX = W0^T dZ_0
        if constexpr (ROLES) {
            const LdsTile<WIDTH> tz0{set + kOffDZ};
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) tz0.store(m, g, t, dz[t]);
        } else {
            T4 x[IN_PAD / 16];
#pragma unroll
            for (int t = 0; t < IN_PAD / 16; ++t) x[t] = cur.x[t];
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) tz.store(m, g, t, dz[t]);
#pragma unroll
            for (int t = 0; t < IN_PAD / 16; ++t) th.store(m, g, t, x[t]);
            wave_lds_sync();
            T4 zt[WIDTH / 16], xt[IN_PAD / 16];
#pragma unroll
            for (int t = 0; t < WIDTH / 16; ++t) zt[t] = tz.load_tr(lane, t);
#pragma unroll
            for (int t = 0; t < IN_PAD / 16; ++t) xt[t] = th.load_tr(lane, t);
            dw0.accumulate(zt, xt);
            wave_lds_sync();
        }
        NVO_PH(5);
        if (need_dinput) {
            f4 acc[IN_PAD / 16];
            layer_mm_t<WIDTH, IN_PAD>(wt0, dz, acc);
            NVO_PH(6);
            if constexpr (IO == NVO_IO_F32_ROWS) {
                float* __restrict__ p = (float*)a.dinput + (size_t)row * a.n_in;
#pragma unroll
                for (int tk = 0; tk < IN_PAD / 16; ++tk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t c = 16 * tk + 4 * g + j;
                        if (c < a.n_in) p[c] = acc[tk][j];
                    }
            } else if constexpr (IO == NVO_IO_HALF2_SOA) {
                T2* __restrict__ p = (T2*)a.dinput;
                const uint32_t n_lv = a.n_in >> 1;
#pragma unroll
                for (int tk = 0; tk < IN_PAD / 16; ++tk) {
                    const uint32_t lv = 8 * tk + 2 * g;
                    const T2 q0 = T2{(T)acc[tk][0], (T)acc[tk][1]}, q1 = T2{(T)acc[tk][2], (T)acc[tk][3]};
                    if (lv < n_lv) p[(size_t)lv * a.batch + row] = q0;
                    if (lv + 1 < n_lv) p[(size_t)(lv + 1) * a.batch + row] = q1;
                    l1a[tk][0] += fabsf((float)q0[0]);  // (columns beyond n_in are never read)
                    l1a[tk][1] += fabsf((float)q0[1]);
                    l1a[tk][2] += fabsf((float)q1[0]);
                    l1a[tk][3] += fabsf((float)q1[1]);
                }
            } else if constexpr (IO == NVO_IO_NERFACTO_COLOR) {
                if constexpr (IN_PAD == 64) {
                    const uint32_t ray = row / a.samples_per_ray;
                    const uint32_t cam = cur.cam;
                    T* __restrict__ dbo = a.d_base_out + (size_t)row * 16;
                    // a 16-sample tile lies inside one ray when samples_per_ray % 16 == 0: reduce the
                    // per-ray quantities over the tile before touching memory
                    const bool tile_in_ray = (a.samples_per_ray & 15u) == 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * g + j < 15) dbo[1 + 4 * g + j] = (T)acc[1][j];
                    if (tile_in_ray) {
                        // After the row reduction every lane of a 16-lane row holds the sums; lane m < 4 keeps column
                        // j = m, so one atomic instruction covers 16 consecutive floats (one 64-byte request) where a
                        // per-j form issues four instructions with 4 lanes each.
                        // embedding columns: e0 <- feature 31, 1+q <- feature 32+q, 17+q <- feature 48+q
                        float e_lo = 0.f, e_hi = 0.f, s_sh = 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float lo = group16_sum(acc[2][j]);
                            const float hi = group16_sum(4 * g + j < 15 ? acc[3][j] : 0.f);
                            const float sh = group16_sum(acc[0][j]);
                            e_lo = m == j ? lo : e_lo;
                            e_hi = m == j ? hi : e_hi;
                            s_sh = m == j ? sh : s_sh;
                        }
                        const float e_0 = group16_sum(g == 3 ? acc[1][3] : 0.f);  // q == 15 only
                        if (m < 4 && a.tile_partial) {
                            // deterministic mode: the tile's sums are stored, nvo_color_tiles_to_rays /
                            // nvo_reduce_by_camera add them up in a fixed order
                            const int q = 4 * g + m;
                            float* tp = a.tile_partial + (size_t)tile * 48;
                            tp[1 + q] = e_lo;
                            if (q < 15) tp[17 + q] = e_hi;
                            if (q == 15) tp[0] = e_0;
                            tp[32 + q] = s_sh;
                        } else if (m < 4) {
                            const int q = 4 * g + m;
                            if (a.d_embedding) {
                                float* de = a.d_embedding + (size_t)cam * 32;
                                atomicAdd(de + 1 + q, e_lo);
                                if (q < 15) atomicAdd(de + 17 + q, e_hi);
                                if (q == 15) atomicAdd(de + 0, e_0);
                            }
                            if (a.d_sh) atomicAdd(a.d_sh + (size_t)ray * 16 + q, s_sh);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int q = 4 * g + j;
                            if (a.d_embedding) {
                                float* de = a.d_embedding + (size_t)cam * 32;
                                atomicAdd(de + 1 + q, acc[2][j]);
                                if (q < 15) atomicAdd(de + 17 + q, acc[3][j]);
                                if (q == 15) atomicAdd(de + 0, acc[1][j]);
                            }
                            if (a.d_sh) atomicAdd(a.d_sh + (size_t)ray * 16 + q, acc[0][j]);
                        }
                    }
                }
            } else if constexpr (IO == NVO_IO_NGP_RGB) {
                if constexpr (IN_PAD == 32) {
                    // d(density-net output): all 16 columns; column 0 also receives dL/d(density pre-activation)
                    T4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (T)acc[0][j];
                    if (g == 0 && a.d_extra_col0) v[0] = (T)(acc[0][0] + a.d_extra_col0[row]);
                    *reinterpret_cast<T4*>(a.d_base_out + (size_t)row * 16 + 4 * g) = v;
                }
            } else {
                T* __restrict__ p = (T*)a.dinput + (size_t)row * IN_PAD + 4 * g;
#pragma unroll
                for (int tk = 0; tk < IN_PAD / 16; ++tk) {
                    T4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (T)acc[tk][j];
                    *reinterpret_cast<T4*>(p + 16 * tk) = v;
                }
            }
        }
        NVO_PH(7);
        if constexpr (ROLES) {
            ++it;  // A: the set of this step is complete
            if constexpr (kHandFlags) {
                if (lane == 0) __hip_atomic_store(hand_full + wib, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __syncthreads();
            }
        }
        cur = nxt;
        NVO_PH(8);
    }
    };
    if constexpr (kListIo) {
        if (use_list) chain_loop(std::true_type{});
        else chain_loop(std::false_type{});
    } else {
        chain_loop(std::false_type{});
    }
    if constexpr (ROLES && !kHandFlags) {
        if (!is_dw)
            for (; it < n_iter; ++it) {  // (chain waves that ran out of tiles keep the workgroup's barrier count)
                __syncthreads();
                __syncthreads();
            }
    }
    // ---- flush weight gradients (block-reduced through the now idle LDS tiles)
    if constexpr (ROLES) {
        if (a.dweights && !is_dw) {  // (the two barriers of the dW role's flush)
            __syncthreads();
            __syncthreads();
        }
    }
    if (a.dweights && !is_dw && !ROLES) {
        __syncthreads();  // every wave is done with its LDS tiles
        float* red = reinterpret_cast<float*>(lds);
        constexpr int kLdsFloats = (int)(sizeof(T) * kLdsHalfs / sizeof(float));
        static_assert(WIDTH * IN_PAD <= kLdsFloats && WIDTH * WIDTH <= kLdsFloats && OUT_PAD * WIDTH <= kLdsFloats,
                      "dW block reduction does not fit the LDS tiles");
        constexpr int kWeights = WIDTH * IN_PAD + (N_HIDDEN - 1) * WIDTH * WIDTH + OUT_PAD * WIDTH;
        float* dW = a.dweights;
        if (a.dw_replicas) {  // (spread the workgroups' adds over several copies: see NvoMlpArgsT::dw_replicas)
            const uint32_t r = blockIdx.x % (a.dw_n_replicas + 1u);
            if (r) dW = a.dw_replicas + (size_t)(r - 1u) * kWeights;
        }
        float* part = a.dw_partial ? a.dw_partial + (size_t)blockIdx.x * kWeights : nullptr;
        if constexpr (!ROLES) {
            dw0.flush_block(dW, red, lane, wib, part, a.nf_flag);
            dW += WIDTH * IN_PAD;
            if (part) part += WIDTH * IN_PAD;
#pragma unroll
            for (int l = 0; l < N_HIDDEN - 1; ++l) {
                dwh[l].flush_block(dW, red, lane, wib, part, a.nf_flag);
                dW += WIDTH * WIDTH;
                if (part) part += WIDTH * WIDTH;
            }
            dwl.flush_block(dW, red, lane, wib, part, a.nf_flag);
        }
    }
    if constexpr (IO == NVO_IO_HALF2_SOA) {
        if (a.dx_l1_partial && need_dinput) {  // (kernel-uniform)
            __syncthreads();  // the LDS tiles / the dW reduction buffer are idle
            float* red = reinterpret_cast<float*>(lds);
#pragma unroll
            for (int tk = 0; tk < IN_PAD / 16; ++tk)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = group16_sum(l1a[tk][j]);  // over the 16 sample lanes of this lane group
                    if (m == 0 && !is_dw) red[wib * IN_PAD + 16 * tk + 4 * g + j] = v;
                }
            if (lane == 0 && !is_dw) reinterpret_cast<uint32_t*>(red)[kChainWaves * IN_PAD + wib] = live_cnt;
            __syncthreads();
            if ((int)threadIdx.x < IN_PAD) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < kChainWaves; ++w) t += red[w * IN_PAD + threadIdx.x];
                a.dx_l1_partial[(size_t)blockIdx.x * IN_PAD + threadIdx.x] = t;
            }
            if (threadIdx.x == 0 && a.dx_live_partial) {
                uint32_t c = 0u;
#pragma unroll
                for (int w = 0; w < kChainWaves; ++w) c += reinterpret_cast<uint32_t*>(red)[kChainWaves * IN_PAD + w];
                a.dx_live_partial[blockIdx.x] = c;
            }
        }
    }
    if constexpr (kRowsIo) {
        if (a.live_rows && use_list) {  // (kernel-uniform) this workgroup's rows -> the global list
            __syncthreads();
            const uint32_t total = list_words[12];
            if (threadIdx.x == 0) list_words[13] = total ? atomicAdd(a.live_rows_n, total) : 0u;
            __syncthreads();
            const uint32_t gbase = list_words[13];
            for (uint32_t i = threadIdx.x; i < total; i += 3 * kMlpBlock) a.live_rows[gbase + i] = wg_rows[i];
        }
    }
#ifdef NVO_MLP_PHASE
    NVO_PH(11);  // dW flush (block reduction through LDS + float atomics)
    if constexpr (IO == NVO_IO_NERFACTO_COLOR) {
        if (wave == 0 && lane == 0 && !is_dw)
            for (int k = 0; k < 12; ++k) NVO_MLP_NAME(nvo_mlp_phase_cycles)[k] = ph_[k];
    }
#endif
}

template <int IN_PAD, int WIDTH, int N_HIDDEN, int OUT_PAD, int IO>
int launch_fwd_io(const Args& a, hipStream_t stream, uint32_t max_blocks) {
    NVO_PROF(stream, "mlp_fwd[%d-%dx%d-%d]" NVO_MLP_TAG, IN_PAD, WIDTH, N_HIDDEN, OUT_PAD);
    const uint32_t n_tiles = a.batch >> 4;
    uint32_t blocks = nvo_div_up(n_tiles, kWavesPerBlock);
    if (blocks > max_blocks) blocks = max_blocks;
    if (a.compact_out) {
        if constexpr ((IO == NVO_IO_HALF2_SOA || IO == NVO_IO_GRID_FUSED) && OUT_PAD == 16) {
            if (a.act == NVO_ACT_RELU) {
                NVO_LAUNCH((NVO_MLP_NAME(k_mlp_fwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, true>), dim3(blocks), dim3(kMlpBlock), 0, stream, a);
                NVO_CHECK_LAUNCH();
                return NVO_OK;
            }
        }
        nvo_set_error("mlp: compact (column 0) output needs the level-major half2 input layout and ReLU");
        return NVO_ERR_UNSUPPORTED;
    }
    if (a.act == NVO_ACT_RELU) {
        NVO_LAUNCH((NVO_MLP_NAME(k_mlp_fwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, false>), dim3(blocks), dim3(kMlpBlock), 0, stream, a);
    } else {
        NVO_LAUNCH((NVO_MLP_NAME(k_mlp_fwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, false, false>), dim3(blocks), dim3(kMlpBlock), 0, stream, a);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

// The IO layout is a compile-time parameter of the kernels (run-time mode switches inside the tile loop cost
// branches and, worse, full `s_waitcnt vmcnt(0)` drains at every join, which defeats the input prefetch).
// Row-major fp32 / level-major half2 / row-major half exist for every shape, the two fused heads only for theirs.
template <int IN_PAD, int WIDTH, int N_HIDDEN, int OUT_PAD>
int launch_fwd(const Args& a, hipStream_t stream, uint32_t max_blocks) {
    switch (a.in_mode) {
        case NVO_IO_F32_ROWS: return launch_fwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_F32_ROWS>(a, stream, max_blocks);
        case NVO_IO_HALF2_SOA: return launch_fwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_HALF2_SOA>(a, stream, max_blocks);
        case NVO_IO_HALF_ROWS: return launch_fwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_HALF_ROWS>(a, stream, max_blocks);
        case NVO_IO_NERFACTO_COLOR:
            if constexpr (IN_PAD == 64 && WIDTH == 64 && N_HIDDEN == 2 && OUT_PAD == 16)
                return launch_fwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_NERFACTO_COLOR>(a, stream, max_blocks);
            break;
        case NVO_IO_NGP_RGB:
            if constexpr (IN_PAD == 32 && WIDTH == 64 && N_HIDDEN == 2 && OUT_PAD == 16)
                return launch_fwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_NGP_RGB>(a, stream, max_blocks);
            break;
        case NVO_IO_GRID_FUSED:
            if constexpr (IN_PAD <= 32 && N_HIDDEN == 1 && OUT_PAD == 16)
                return launch_fwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_GRID_FUSED>(a, stream, max_blocks);
            break;
    }
    nvo_set_error("mlp: IO mode %d is not available for shape %d-%dx%d-%d", a.in_mode, IN_PAD, WIDTH, N_HIDDEN, OUT_PAD);
    return NVO_ERR_UNSUPPORTED;
}

template <int IN_PAD, int WIDTH, int N_HIDDEN, int OUT_PAD, int IO>
int launch_bwd_io_kernel(const Args& a, hipStream_t stream, uint32_t blocks);

template <int IN_PAD, int WIDTH, int N_HIDDEN, int OUT_PAD, int IO>
int launch_bwd_io(const Args& a, hipStream_t stream, uint32_t max_blocks) {
    NVO_PROF(stream, "mlp_bwd[%d-%dx%d-%d]" NVO_MLP_TAG, IN_PAD, WIDTH, N_HIDDEN, OUT_PAD);
    const uint32_t n_tiles = a.batch >> 4;
    uint32_t blocks = nvo_div_up(n_tiles, kWavesPerBlock);
    if (blocks > max_blocks) blocks = max_blocks;
    if (int rc = launch_bwd_io_kernel<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO>(a, stream, blocks)) return rc;
    if (a.dw_partial && a.dweights) {  // deterministic mode: sum the workgroups' block totals in workgroup order
        constexpr uint64_t kWeights = (uint64_t)WIDTH * IN_PAD + (uint64_t)(N_HIDDEN - 1) * WIDTH * WIDTH + (uint64_t)OUT_PAD * WIDTH;
        return nvo_reduce_partials(stream, a.dw_partial, blocks, kWeights, a.dweights);
    }
    return NVO_OK;
}

template <int IN_PAD, int WIDTH, int N_HIDDEN, int OUT_PAD, int IO>
int launch_bwd_io_kernel(const Args& a, hipStream_t stream, uint32_t blocks) {
    if (a.compact_out || a.recompute_hidden) {
        // instantiated where the NeRF-VO path uses them: level-major input (networks behind a hash grid; compact
        // only there, recomputation for their single-hidden-layer shapes) and the fused colour head
        constexpr bool kSoa = IO == NVO_IO_HALF2_SOA && OUT_PAD == 16;
        constexpr bool kRecompOk = (kSoa && N_HIDDEN == 1) || IO == NVO_IO_NERFACTO_COLOR || IO == NVO_IO_NGP_RGB;
        if (a.act == NVO_ACT_RELU) {
            if constexpr (kRecompOk) {
                if (a.recompute_hidden && !a.compact_out) {
                    if constexpr (WIDTH == 64) {
                        // the 64-wide networks (base MLP, colour head, the NGP networks): chain / dW roles, see k_mlp_bwd --
                        // base network 30.3 -> 22.9 us, colour head 49.5 -> 41.4 us
                        static const bool roles = [] { const char* e = getenv("NVO_MLP_ROLES"); return !e || atoi(e) != 0; }();
                        if (roles) {
                            // (A/B and tests: NVO_MLP_SKIP_DEAD=0 walks every tile; read per launch -- a graph keeps what
                            // its capture saw -- so that one process can compare the two)
                            const char* const e_dead = getenv("NVO_MLP_SKIP_DEAD");
                            Args ar = a;
                            if (e_dead && atoi(e_dead) == 0) ar.tile_live = nullptr;
                            constexpr bool kCanList = IO == NVO_IO_HALF2_SOA || (IO == NVO_IO_NERFACTO_COLOR && IN_PAD == 64);
                            if constexpr (!kCanList) ar.tile_live = nullptr;
                            if (!ar.tile_live) ar.live_rows = nullptr;  // (the plain form lists nothing: the caller's fallback is k_live_rows)
                            if constexpr (IO == NVO_IO_HALF2_SOA) {
                                // (the caller asked nvo_mlp_bwd_lists_rows first; a workgroup's tiles must fit its row buffer)
                                const uint32_t n_tiles_l = a.batch >> 4;
                                const uint32_t n_own_l = nvo_div_up(n_tiles_l, blocks * 2u * kWavesPerBlock) * 2u * kWavesPerBlock;
                                if (ar.live_rows && n_own_l * 16u > (uint32_t)kLiveRowCap) {
                                    nvo_set_error("mlp: live_rows with %u tiles per workgroup (> %d rows)", n_own_l, kLiveRowCap);
                                    return NVO_ERR_UNSUPPORTED;
                                }
                            } else {
                                ar.live_rows = nullptr;
                            }
                            // (> 64 KiB of dynamic LDS needs an explicit opt-in, per instantiation)
                            if constexpr (kCanList) {
                                if (ar.tile_live) {
                                    constexpr size_t kBytesL = sizeof(T) * (size_t)bwd_lds_halfs(IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, true, IO == NVO_IO_HALF2_SOA, true);
                                    static bool attr_set_l = false;
                                    if (!attr_set_l) {
                                        NVO_CHECK_HIP(hipFuncSetAttribute(
                                            (const void*)NVO_MLP_NAME(k_mlp_bwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, false, true, true, true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBytesL));
                                        attr_set_l = true;
                                    }
                                    NVO_LAUNCH((NVO_MLP_NAME(k_mlp_bwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, false, true, true, true>),
                                               dim3(blocks), dim3(3 * kMlpBlock), kBytesL, stream, ar);
                                    NVO_CHECK_LAUNCH();
                                    return NVO_OK;
                                }
                            }
                            constexpr size_t kBytes = sizeof(T) * (size_t)bwd_lds_halfs(IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, true);
                            static bool attr_set = false;
                            if (!attr_set) {
                                NVO_CHECK_HIP(hipFuncSetAttribute(
                                    (const void*)NVO_MLP_NAME(k_mlp_bwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, false, true, true, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBytes));
                                attr_set = true;
                            }
                            NVO_LAUNCH((NVO_MLP_NAME(k_mlp_bwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, false, true, true, false>),
                                       dim3(blocks), dim3(3 * kMlpBlock), kBytes, stream, ar);
                            NVO_CHECK_LAUNCH();
                            return NVO_OK;
                        }
                    }
                    NVO_LAUNCH((NVO_MLP_NAME(k_mlp_bwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, false, true>), dim3(blocks), dim3(kMlpBlock), 0, stream, a);
                    NVO_CHECK_LAUNCH();
                    return NVO_OK;
                }
            }
            if constexpr (kSoa && N_HIDDEN == 1) {
                if (a.recompute_hidden && a.compact_out) {
                    NVO_LAUNCH((NVO_MLP_NAME(k_mlp_bwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, true, true>), dim3(blocks), dim3(kMlpBlock), 0, stream, a);
                    NVO_CHECK_LAUNCH();
                    return NVO_OK;
                }
            }
            if constexpr (kSoa) {
                if (a.compact_out && !a.recompute_hidden) {
                    NVO_LAUNCH((NVO_MLP_NAME(k_mlp_bwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, true, false>), dim3(blocks), dim3(kMlpBlock), 0, stream, a);
                    NVO_CHECK_LAUNCH();
                    return NVO_OK;
                }
            }
        }
        nvo_set_error("mlp: compact output / hidden recomputation are not available for this shape, layout or activation");
        return NVO_ERR_UNSUPPORTED;
    }
    if (a.act == NVO_ACT_RELU) {
        NVO_LAUNCH((NVO_MLP_NAME(k_mlp_bwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, true, false, false>), dim3(blocks), dim3(kMlpBlock), 0, stream, a);
    } else {
        NVO_LAUNCH((NVO_MLP_NAME(k_mlp_bwd)<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, IO, false, false, false>), dim3(blocks), dim3(kMlpBlock), 0, stream, a);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

template <int IN_PAD, int WIDTH, int N_HIDDEN, int OUT_PAD>
int launch_bwd(const Args& a, hipStream_t stream, uint32_t max_blocks) {
    if (a.dinput != nullptr && a.din_mode != a.in_mode) {
        nvo_set_error("mlp: dinput layout (%d) must equal the input layout (%d)", a.din_mode, a.in_mode);
        return NVO_ERR_UNSUPPORTED;
    }
    switch (a.in_mode) {
        case NVO_IO_F32_ROWS: return launch_bwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_F32_ROWS>(a, stream, max_blocks);
        case NVO_IO_HALF2_SOA: return launch_bwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_HALF2_SOA>(a, stream, max_blocks);
        case NVO_IO_HALF_ROWS: return launch_bwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_HALF_ROWS>(a, stream, max_blocks);
        case NVO_IO_NERFACTO_COLOR:
            if constexpr (IN_PAD == 64 && WIDTH == 64 && N_HIDDEN == 2 && OUT_PAD == 16)
                return launch_bwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_NERFACTO_COLOR>(a, stream, max_blocks);
            break;
        case NVO_IO_NGP_RGB:
            if constexpr (IN_PAD == 32 && WIDTH == 64 && N_HIDDEN == 2 && OUT_PAD == 16)
                return launch_bwd_io<IN_PAD, WIDTH, N_HIDDEN, OUT_PAD, NVO_IO_NGP_RGB>(a, stream, max_blocks);
            break;
    }
    nvo_set_error("mlp: IO mode %d is not available for shape %d-%dx%d-%d", a.in_mode, IN_PAD, WIDTH, N_HIDDEN, OUT_PAD);
    return NVO_ERR_UNSUPPORTED;
}


}  // namespace


// Supported shapes (every MLP on the NeRF-VO mapping path):
//   (in_pad, width, n_hidden, out_pad)
//   (32, 64, 1, 16) nerfacto base MLP      (64, 64, 2, 16) colour MLP
//   (32, 64, 3, 64) predicted-normals MLP  (16, 16, 1, 16) proposal density MLP
//   (16, 64, 1..2, 16), (32, 64, 2, 16), (64, 64, 1, 16) generic tcnn.Network uses
//   (16 | 32, 32, 1..2, 16) tcnn's n_neurons = 32 (not on the NeRF-VO path; the tcnn surface beyond nerfacto)
#define NVO_MLP_SHAPES(X) \
    X(32, 64, 1, 16)      \
    X(64, 64, 2, 16)      \
    X(32, 64, 3, 64)      \
    X(16, 16, 1, 16)      \
    X(16, 64, 1, 16)      \
    X(16, 64, 2, 16)      \
    X(32, 64, 2, 16)      \
    X(64, 64, 1, 16)      \
    X(16, 16, 2, 16)      \
    X(32, 16, 1, 16)      \
    X(16, 32, 1, 16)      \
    X(16, 32, 2, 16)      \
    X(32, 32, 1, 16)      \
    X(32, 32, 2, 16)

// workgroup caps (tuning knobs; the defaults are the measured optima on MI355X)
static uint32_t env_blocks(const char* name, uint32_t dflt) {
    const char* e = getenv(name);
    return e && atoi(e) > 0 ? (uint32_t)atoi(e) : dflt;
}

// Every workgroup stages the whole weight set through LDS and every wave takes its fragments into registers before
// its first tile, so the cap trades that setup (and, backward, the per-workgroup dW flush) against parallelism.
// Measured on MI355X (bench.py per-kernel table, N = 196 608 / 1 M rows, round 4, weights staged through LDS):
// colour head 64-64x2-16 forward 18.2 / 16.6 / 18.6 / 22.3 us at 256 / 512 / 1024 / 2048 workgroups (fragments straight
// from global memory: 21.4 / 22.4 / 30.2 / 46.7); base 32-64x1-16 14.6 / 10.5 / 9.7 / 9.9; the 16-wide proposal MLP
// 18.9 / 12.5 / 9.6 / 8.3.
static uint32_t fwd_block_cap(int in_pad, int width, int n_hidden) {
    const int weight_halfs = width * in_pad + (n_hidden - 1) * width * width;
    if (weight_halfs >= 8192) return 512;
    if (weight_halfs >= 2048) return 1024;
    return 2048;
}
static uint32_t bwd_block_cap(int in_pad, int width, int n_hidden) {
    // 16-wide proposal MLP: 38.8 / 27.7 / 24.0 / 27.9 us at 256 / 512 / 1024 / 2048 workgroups (with the dW copies; without
    // them every workgroup more cost 22 ns of serialised adds and 512 was the optimum)
    if (width * in_pad + (n_hidden - 1) * width * width < 2048) return 1024;
    // one workgroup per CU: with the tile inputs software-pipelined every shape is fastest at 256 (colour head
    // 74 us vs 107 us at 512; base 30.6 vs 37.7; proposal 49.5 vs 53.4) -- fewer dW flushes, and the dW
    // accumulators leave the wide shapes one wave per SIMD anyway
    return 256;
}

static bool mlp_shape_supported_impl(int in_pad, int width, int n_hidden, int out_pad) {
#define X(I, W, H, O) \
    if (in_pad == I && width == W && n_hidden == H && out_pad == O) return true;
    NVO_MLP_SHAPES(X)
#undef X
    return false;
}

int NVO_MLP_NAME(nvo_mlp_fwd_launch)(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a0,
                                     hipStream_t stream) {
    const Args& a = reinterpret_cast<const Args&>(a0);
    NVO_REQUIRE((a.batch & 15u) == 0, "mlp: batch (%u) must be a multiple of 16", a.batch);
    if (a.batch == 0) return NVO_OK;
#define X(I, W, H, O)                                                    \
    if (in_pad == I && width == W && n_hidden == H && out_pad == O)      \
        return launch_fwd<I, W, H, O>(a, stream, env_blocks("NVO_MLP_FWD_BLOCKS", fwd_block_cap(I, W, H)));
    NVO_MLP_SHAPES(X)
#undef X
    nvo_set_error("mlp: unsupported shape in_pad=%d width=%d n_hidden=%d out_pad=%d", in_pad, width,
                  n_hidden, out_pad);
    return NVO_ERR_UNSUPPORTED;
}

int NVO_MLP_NAME(nvo_mlp_bwd_launch)(int in_pad, int width, int n_hidden, int out_pad, const NvoMlpArgs& a0,
                                     hipStream_t stream) {
    const Args& a = reinterpret_cast<const Args&>(a0);
    NVO_REQUIRE((a.batch & 15u) == 0, "mlp: batch (%u) must be a multiple of 16", a.batch);
    if (a.batch == 0) return NVO_OK;
#define X(I, W, H, O)                                                    \
    if (in_pad == I && width == W && n_hidden == H && out_pad == O)      \
        return launch_bwd<I, W, H, O>(a, stream, env_blocks("NVO_MLP_BWD_BLOCKS", bwd_block_cap(I, W, H)));
    NVO_MLP_SHAPES(X)
#undef X
    nvo_set_error("mlp: unsupported shape in_pad=%d width=%d n_hidden=%d out_pad=%d", in_pad, width,
                  n_hidden, out_pad);
    return NVO_ERR_UNSUPPORTED;
}

