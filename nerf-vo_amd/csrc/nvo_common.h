// Shared declarations for the nerfvo HIP library (gfx950 / CDNA4 only).
//
// Everything in csrc/ is compiled with `hipcc --offload-arch=gfx950` into ONE shared library,
// libnerfvo_hip.so, whose exported surface is the C-ABI of include/nerfvo_hip.h.  No torch types
// cross that boundary: callers hand in raw device pointers + a hipStream_t.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stddef.h>

#define NVO_OK 0
#define NVO_ERR_INVALID 1
#define NVO_ERR_HIP 2
#define NVO_ERR_UNSUPPORTED 3

#define NVO_WAVE 64

extern "C" const char* nvo_last_error(void);
void nvo_set_error(const char* fmt, ...);

#define NVO_CHECK_HIP(expr)                                                                  \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            nvo_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return NVO_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

#define NVO_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            nvo_set_error(__VA_ARGS__);        \
            return NVO_ERR_INVALID;            \
        }                                      \
    } while (0)

#define NVO_CHECK_LAUNCH() NVO_CHECK_HIP(hipGetLastError())
// hipGetLastError() is sticky per thread and also latches errors raised by OTHER users of the runtime
// in this process (e.g. a benign device query inside PyTorch): clear it before every launch so that
// the check after the launch reports this launch only.
#define NVO_LAUNCH(...)                  \
    do {                                 \
        (void)hipGetLastError();         \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

// ---- optional per-launch HIP-event profiler (off by default; see nvo_profile_enable) ----------
bool nvo_prof_enabled();
void nvo_prof_begin(hipStream_t s, const char* fmt, ...);
void nvo_prof_end(hipStream_t s);
struct NvoProfScope {
    hipStream_t s;
    bool on;
    explicit NvoProfScope(hipStream_t s_, bool enable = true) : s(s_), on(enable && nvo_prof_enabled()) {}
    ~NvoProfScope() {
        if (on) nvo_prof_end(s);
    }
};
// Brackets everything the enclosing launcher enqueues with a pair of events on ITS stream.
#define NVO_PROF(stream, ...)                                   \
    NvoProfScope nvo_prof_scope__((hipStream_t)(stream));       \
    if (nvo_prof_scope__.on) nvo_prof_begin((hipStream_t)(stream), __VA_ARGS__)

// While an NvoProfMute is alive, NVO_PROF scopes of nested launchers are not recorded (their time is part
// of the enclosing scope) unless NVO_PROF_DETAIL=1.
void nvo_prof_mute(int delta);
struct NvoProfMute {
    NvoProfMute() { nvo_prof_mute(+1); }
    ~NvoProfMute() { nvo_prof_mute(-1); }
};

// Sub-scope inside a launcher that already has an NVO_PROF scope: recorded only with NVO_PROF_DETAIL=1 in
// the environment (the per-launcher totals would otherwise count the time twice).
bool nvo_prof_detail();
#define NVO_PROF_SUB(stream, ...)                                                      \
    NvoProfScope nvo_prof_sub__((hipStream_t)(stream), nvo_prof_detail());             \
    if (nvo_prof_sub__.on) nvo_prof_begin((hipStream_t)(stream), __VA_ARGS__)

// ---- graph-capture-safe growable device scratch -----------------------------------------------
// Module-owned scratch that grows with the largest batch seen.  A captured hipGraph addresses the block by POINTER, so
// once a launch that uses it has been captured the block must never be freed while the module lives: a larger batch
// later (an eager step or a render at another ray count between two replays) gets a NEW block and the old one is
// RETIRED -- kept allocated until the module is destroyed -- instead of hipFree'd.  Growing while a capture is in
// progress is refused (hipMalloc is not capturable): run one eager warm-up launch at that size first.
struct NvoScratch {
    void* ptr = nullptr;
    size_t bytes = 0;
    bool captured = false;     // a captured launch addresses `ptr`
    void* retired[8] = {};     // blocks captured graphs may still address (freed by nvo_scratch_destroy)
    uint32_t n_retired = 0;
};
// Makes s->ptr hold at least `need` bytes for a launch on `stream`; `what` names the buffer in error messages.
int nvo_scratch_reserve(NvoScratch* s, size_t need, hipStream_t stream, const char* what);
void nvo_scratch_destroy(NvoScratch* s);

static inline uint32_t nvo_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
static inline uint64_t nvo_round_up(uint64_t a, uint64_t b) { return ((a + b - 1) / b) * b; }

// ---------------------------------------------------------------------------------------------
// Multi-resolution grid level table (host-computed once, passed to kernels BY VALUE).
// Semantics follow tiny-cuda-nn's GridEncodingTemplated constructor and grid_scale /
// grid_resolution / grid_index helpers (upstream tcnn include/tiny-cuda-nn/encodings/grid.h;
// NOT vendored in /root/reference -- see SURVEY.md section 2.4 K1 and section 8c).
// ---------------------------------------------------------------------------------------------
#define NVO_MAX_LEVELS 32

struct NvoGridLevels {
    uint32_t n_levels;
    uint32_t n_features;                    // features per level (2 on every path NeRF-VO uses)
    uint32_t offset[NVO_MAX_LEVELS + 1];    // entry offset of each level (entries, not scalars)
    uint32_t resolution[NVO_MAX_LEVELS];    // vertices per axis
    float scale[NVO_MAX_LEVELS];            // pos = fma(scale, x, 0.5)
    uint32_t hashed[NVO_MAX_LEVELS];        // 1 -> spatial hash, 0 -> dense stride index
};

// Fills the table; returns total number of entries.  log2_hashmap_size / base_resolution /
// per_level_scale have tcnn's meaning.  All float math is done in fp32 with glibc exp2f/log2f/
// ceilf so that the oracle's C restatement (same libm) reproduces the table bit for bit.
uint32_t nvo_grid_levels_init(NvoGridLevels* g, uint32_t n_levels, uint32_t n_features,
                              uint32_t log2_hashmap_size, uint32_t base_resolution,
                              float per_level_scale);

// several device ranges cleared by one launch (a training step's accumulate-into buffers): blocks [first_block[k],
// first_block[k + 1]) serve range k.  Shared by k_zero_ranges and by the ray head's extra workgroups (nvo_ray_head_zero).
constexpr uint32_t kZeroMaxRanges = 24;
struct NvoZeroPlan {
    uint32_t n;
    uint32_t first_block[kZeroMaxRanges + 1];
    uint32_t* ptr[kZeroMaxRanges];
    uint64_t words[kZeroMaxRanges];
};
// host: fills `plan` from (ptrs, bytes); returns the number of 256-thread blocks it needs (0: nothing to clear), < 0 on a
// bad range (nvo_last_error is set)
int nvo_zero_plan_build(uint32_t n_ranges, void* const* ptrs, const uint64_t* bytes, NvoZeroPlan* plan);

#ifdef __HIPCC__
__device__ __forceinline__ void nvo_zero_plan_block(const NvoZeroPlan& r, uint32_t block) {  // blockDim.x == 256
    uint32_t k = 0;
    while (k + 1 < r.n && block >= r.first_block[k + 1]) ++k;
    const uint32_t nb = r.first_block[k + 1] - r.first_block[k];
    const uint64_t stride = (uint64_t)nb * 256u;
    uint32_t* __restrict__ p = r.ptr[k];
    const uint64_t n = r.words[k];
    const uint64_t tid = (uint64_t)(block - r.first_block[k]) * 256u + threadIdx.x;
    if ((((uintptr_t)p) & 15u) == 0u) {
        uint4* __restrict__ p4 = reinterpret_cast<uint4*>(p);
        for (uint64_t i = tid; i < n / 4; i += stride) p4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (uint64_t i = (n / 4) * 4 + tid; i < n; i += stride) p[i] = 0u;
    } else {
        for (uint64_t i = tid; i < n; i += stride) p[i] = 0u;
    }
}
// ---- 16-bit activation storage chosen at run time (wave-uniform flag): fp16 (tcnn's precision) or bfloat16
// (EngineConfig.mlp_dtype = "bf16", BASELINE configs[4]).  Used by the kernels around the fused MLP, where one
// select per element is noise; the MLP kernels themselves are compiled per type (mlp_impl.h).
typedef unsigned short nvo_h16;  // raw bits of a _Float16 or a __bf16
__device__ __forceinline__ float nvo_ld16(const nvo_h16* p, bool bf) {
    const nvo_h16 raw = *p;
    return bf ? __uint_as_float((uint32_t)raw << 16) : (float)__builtin_bit_cast(_Float16, raw);
}
__device__ __forceinline__ nvo_h16 nvo_cvt16(float v, bool bf) {  // round to nearest even in both formats
    // The value is made opaque first: when `v` is a visible fp32 product the backend otherwise selects
    // v_fma_mixlo_f16 (product kept exact, ONE rounding to fp16), and whether it can see the product depends on the
    // surrounding kernel -- fused and unfused launch sequences must round identically (fp32 first, then 16 bit).
    asm("" : "+v"(v));
    return bf ? __builtin_bit_cast(nvo_h16, (__bf16)v) : __builtin_bit_cast(nvo_h16, (_Float16)v);
}
__device__ __forceinline__ uint32_t nvo_cvt16x2(float a, float b, bool bf) {
    return (uint32_t)nvo_cvt16(a, bf) | ((uint32_t)nvo_cvt16(b, bf) << 16);
}

// torch.optim.Adam on one scalar (no AMSGrad, L2 weight decay folded into the gradient); shared by the optimiser launch
// (adam.hip) and the hash-grid backward that steps the entries it has just finished summing (grid.hip, NvoGridAdam) --
// ONE definition, so both produce the same bits (the library is built with -ffp-contract=off).
struct NvoAdamHyper {
    float lr, beta1, beta2, eps, bias1, bias2_sqrt, grad_scale, weight_decay;
};
__device__ __forceinline__ void nvo_adam_one(float& p, float& m, float& v, float g, const NvoAdamHyper& h) {
    float gi = g * h.grad_scale;
    if (h.weight_decay != 0.f) gi += h.weight_decay * p;
    m = h.beta1 * m + (1.f - h.beta1) * gi;
    v = h.beta2 * v + (1.f - h.beta2) * gi * gi;
    const float denom = sqrtf(v) / h.bias2_sqrt + h.eps;
    p -= (h.lr / h.bias1) * (m / denom);
}
__device__ __forceinline__ float2 nvo_ld16x2(uint32_t raw, bool bf) {
    if (bf) return make_float2(__uint_as_float(raw << 16), __uint_as_float(raw & 0xFFFF0000u));
    return make_float2((float)__builtin_bit_cast(_Float16, (nvo_h16)(raw & 0xFFFFu)),
                       (float)__builtin_bit_cast(_Float16, (nvo_h16)(raw >> 16)));
}

// ---- sample position -> contracted, normalised grid coordinate (nerfacto: Frustums.get_positions,
// SceneContraction(L-inf), (x + 2) / 4, selector mask -> masked positions are zeroed) ----------------------
__device__ __forceinline__ void nvo_contract_position01(const float o[3], const float d[3], float mid, float out[3]) {
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = o[k] + d[k] * mid;
    const float mag = fmaxf(fabsf(p[0]), fmaxf(fabsf(p[1]), fabsf(p[2])));
    if (!(mag < 1.f)) {
        const float f = (2.f - 1.f / mag) / mag;
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] *= f;
    }
    bool sel = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p[k] = (p[k] + 2.f) * 0.25f;
        sel = sel && (p[k] > 0.f) && (p[k] < 1.f);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = sel ? p[k] : 0.f;
}

// ---- real spherical harmonics up to degree 4 of a unit direction (tcnn SphericalHarmonics; sh.hip, rays.hip) ----
__device__ __forceinline__ void nvo_sh4_eval(float x, float y, float z, uint32_t degree, float* o) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    if (degree <= 1) return;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    if (degree <= 2) return;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    if (degree <= 3) return;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}


// ---- wave64 cross-lane primitives as DPP modifiers -------------------------------------------------
// __shfl / __shfl_up / __shfl_xor compile to ds_bpermute_b32: an LDS round trip (~100+ cycles) per step, six dependent
// ones per scan or reduction.  The per-ray kernels are one wave per ray and do little else between their scans, so the
// steps are written as DPP row shifts / row broadcasts (gfx9 family: row_shr:n inside a row of 16, row_bcast:15 / :31
// across rows), which ride on the add itself.  Identity 0: a lane without a source adds 0.
#if defined(__HIPCC__)
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float nvo_dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t nvo_dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ float nvo_wave_incl_scan(float v) {
    v += nvo_dpp_f32<0x111>(v);          // row_shr:1
    v += nvo_dpp_f32<0x112>(v);          // row_shr:2
    v += nvo_dpp_f32<0x114>(v);          // row_shr:4
    v += nvo_dpp_f32<0x118>(v);          // row_shr:8   -> inclusive scan inside every row of 16
    v += nvo_dpp_f32<0x142, 0xA>(v);     // row_bcast:15 into rows 1 and 3
    v += nvo_dpp_f32<0x143, 0xC>(v);     // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t nvo_wave_incl_scan(uint32_t v) {
    v += nvo_dpp_u32<0x111>(v);
    v += nvo_dpp_u32<0x112>(v);
    v += nvo_dpp_u32<0x114>(v);
    v += nvo_dpp_u32<0x118>(v);
    v += nvo_dpp_u32<0x142, 0xA>(v);
    v += nvo_dpp_u32<0x143, 0xC>(v);
    return v;
}
// value of one lane (wave-uniform index) in every lane
__device__ __forceinline__ float nvo_wave_bcast(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(src_lane)));
}
__device__ __forceinline__ uint32_t nvo_wave_bcast(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src_lane));
}
// sum over the 64 lanes, in every lane
__device__ __forceinline__ float nvo_wave_sum(float v) {
    v += nvo_dpp_f32<0x128>(v);          // row_ror:8
    v += nvo_dpp_f32<0x124>(v);          // row_ror:4
    v += nvo_dpp_f32<0x122>(v);          // row_ror:2
    v += nvo_dpp_f32<0x121>(v);          // row_ror:1   -> every lane holds its row's total
    v += nvo_dpp_f32<0x142, 0xA>(v);     // rows 1, 3 += rows 0, 2
    v += nvo_dpp_f32<0x143, 0xC>(v);     // rows 2, 3 += row 1 (= rows 0 + 1): lane 63 holds the total
    return nvo_wave_bcast(v, 63);
}
#endif

// ---- device helpers shared by grid kernels -------------------------------------------------
__device__ __forceinline__ uint32_t nvo_grid_index(uint32_t hashed, uint32_t hashmap_size,
                                                   uint32_t res, uint32_t px, uint32_t py,
                                                   uint32_t pz) {
    // tcnn grid_index<3, CoherentPrime>: dense stride index while stride <= hashmap_size,
    // spatial hash (primes 1, 2654435761, 805459861) when the level does not fit.
    uint32_t index;
    if (hashed) {
        index = px ^ (py * 2654435761u) ^ (pz * 805459861u);
        return index & (hashmap_size - 1u);  // hashed levels always have power-of-two size
    }
    // Dense levels: index < res^3 <= hashmap_size except for the +1 corners of positions at the upper
    // domain face (x == 1), so the modulo (a ~40-instruction software division on the GPU) sits behind
    // a branch that is almost never taken.
    index = px + py * res + pz * res * res;
    if (index >= hashmap_size) index %= hashmap_size;
    return index;
}
#endif
