// Fused Adam step over a flat fp32 parameter range for gfx950: reads grad / m / v / master weight,
// writes m / v / master weight and the fp16 working copy the kernels consume, in one pass
// (28 B + 2 B per parameter -- the HBM floor of the optimiser).
// Semantics = torch.optim.Adam (no AMSGrad, L2 weight decay folded into the gradient) as the
// reference configures it: AdamOptimizerConfig(lr=1e-2|1e-4, eps=1e-15)
// (/root/reference/nerf_vo/mapping/nerfstudio.py:84-100); the GradScaler "skip the step when a
// non-finite gradient was found" behaviour (mixed_precision=True, nerfstudio.py:59) is the
// optional skip flag.  CPU restatement: oracle/nerfacto.py::adam_reference.
#include "nvo_kernels.h"
#include "../../include/nerfvo_hip.h"

#include <string.h>

namespace {

__device__ __forceinline__ float load_grad(const float* g, uint64_t i) { return g[i]; }
__device__ __forceinline__ float load_grad(const _Float16* g, uint64_t i) { return (float)g[i]; }
// bfloat16 = the upper half of an fp32: same exponent range, 8 significant bits.  The compressed gradient exchange
// uses it rather than fp16 because Adam (eps 1e-15) turns ANY non-zero gradient into a full-size step, so the
// tiny gradients of rarely hit hash-grid entries must not flush to zero (fp16 underflows below 6e-8).
struct Bf16 { uint16_t bits; };
__device__ __forceinline__ float load_grad(const Bf16* g, uint64_t i) { return __uint_as_float((uint32_t)g[i].bits << 16); }
__device__ __forceinline__ uint16_t to_bf16(float x) {
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)((u >> 16) | ((u & 0xFFFFu) ? 0x40u : 0u));  // inf / NaN stay so
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);  // round to nearest even
}

typedef NvoAdamHyper AdamHyper;  // (nvo_common.h: shared with the hash-grid backward's fused step)

// Format of the 16-bit working copy the kernels read: fp16 everywhere except inside up to 4 element ranges
// [lo, hi) of the flat buffer, which are bfloat16 (bf16 MLP mode: the fused-MLP weights and the appearance
// embedding; the hash tables stay fp16 -- 11 significant bits for entries that are interpolated in fp32).
// Range bounds are multiples of 4 (checked on the host), so a 4-element vector never straddles a boundary.
constexpr uint32_t kMaxBf16Ranges = 4;
struct Copy16Fmt {
    uint32_t n;
    uint64_t lo[kMaxBf16Ranges], hi[kMaxBf16Ranges];
    __device__ __forceinline__ bool is_bf16(uint64_t i) const {
        bool r = false;
#pragma unroll
        for (uint32_t k = 0; k < kMaxBf16Ranges; ++k) r = r || (k < n && i >= lo[k] && i < hi[k]);
        return r;
    }
};

__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g, const AdamHyper& h) {
    nvo_adam_one(p, m, v, g, h);
}

// GT = float (local gradient), _Float16 or Bf16 (the 2-byte buffer a compressed all-reduce leaves behind).
// The body works on 4 consecutive parameters per thread through 16-byte accesses when the range is
// 16-byte aligned (HBM-bound kernel: 28 B + 2 B per parameter), scalar otherwise / for the tail.
// base: index of p[0] in the flat buffer (what Copy16Fmt's ranges refer to)
// Optional weight average of the stepped elements (tcnn EmaOptimizer; k_ema_update_dev's arithmetic on the value just
// written, term for term): ema / ema16 start at the range like p does; the 16-byte form needs them aligned like p / p16.
struct AdamEma {
    float* ema;
    nvo_h16* ema16;
    float keep, take, inv_debias;
};
template <typename GT>
__device__ __forceinline__ void adam_range(uint64_t n, float* __restrict__ p, nvo_h16* __restrict__ p16,
                                           const GT* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                           const AdamHyper& h, int vec4, uint32_t block_id, uint32_t n_blocks,
                                           const Copy16Fmt& fmt, uint64_t base, const AdamEma& ea = AdamEma{}) {
    const uint64_t stride = (uint64_t)n_blocks * blockDim.x;
    const uint64_t tid = (uint64_t)block_id * blockDim.x + threadIdx.x;
    uint64_t done = 0;
    if (vec4) {
        const uint64_t n4 = n >> 2;
        for (uint64_t q = tid; q < n4; q += stride) {
            float4 pv = reinterpret_cast<float4*>(p)[q];
            float4 mv = reinterpret_cast<float4*>(m)[q];
            float4 vv = reinterpret_cast<float4*>(v)[q];
            const uint64_t i = q << 2;
            adam_one(pv.x, mv.x, vv.x, load_grad(g, i + 0), h);
            adam_one(pv.y, mv.y, vv.y, load_grad(g, i + 1), h);
            adam_one(pv.z, mv.z, vv.z, load_grad(g, i + 2), h);
            adam_one(pv.w, mv.w, vv.w, load_grad(g, i + 3), h);
            reinterpret_cast<float4*>(p)[q] = pv;
            reinterpret_cast<float4*>(m)[q] = mv;
            reinterpret_cast<float4*>(v)[q] = vv;
            if (p16) {
                const bool bf = fmt.is_bf16(base + i);
                *reinterpret_cast<uint2*>(p16 + i) = make_uint2(nvo_cvt16x2(pv.x, pv.y, bf), nvo_cvt16x2(pv.z, pv.w, bf));
            }
            if (ea.ema) {
                float4 ev = reinterpret_cast<float4*>(ea.ema)[q];
                ev.x = (ev.x * ea.keep + pv.x * ea.take) * ea.inv_debias;
                ev.y = (ev.y * ea.keep + pv.y * ea.take) * ea.inv_debias;
                ev.z = (ev.z * ea.keep + pv.z * ea.take) * ea.inv_debias;
                ev.w = (ev.w * ea.keep + pv.w * ea.take) * ea.inv_debias;
                reinterpret_cast<float4*>(ea.ema)[q] = ev;
                if (ea.ema16)
                    *reinterpret_cast<uint2*>(ea.ema16 + i) =
                        make_uint2(nvo_cvt16x2(ev.x, ev.y, false), nvo_cvt16x2(ev.z, ev.w, false));
            }
        }
        done = n4 << 2;
    }
    for (uint64_t i = done + tid; i < n; i += stride) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_one(pi, mi, vi, load_grad(g, i), h);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
        if (p16) p16[i] = nvo_cvt16(pi, fmt.is_bf16(base + i));
        if (ea.ema) {
            const float e = (ea.ema[i] * ea.keep + pi * ea.take) * ea.inv_debias;
            ea.ema[i] = e;
            if (ea.ema16) ea.ema16[i] = nvo_cvt16(e, false);
        }
    }
}

template <typename GT>
__global__ void __launch_bounds__(256)
k_adam(uint64_t n, float* __restrict__ p, nvo_h16* __restrict__ p16, const GT* __restrict__ g,
       float* __restrict__ m, float* __restrict__ v, AdamHyper h, const uint32_t* __restrict__ skip_flag,
       const float* __restrict__ hyper_dev, int vec4) {
    if (skip_flag && *skip_flag) return;
    if (hyper_dev) {  // {lr, 1 - beta1^t, sqrt(1 - beta2^t)} kept in device memory (graph replay)
        h.lr = hyper_dev[0];
        h.bias1 = hyper_dev[1];
        h.bias2_sqrt = hyper_dev[2];
    }
    adam_range<GT>(n, p, p16, g, m, v, h, vec4, blockIdx.x, gridDim.x, Copy16Fmt{}, 0);
}

// Several parameter groups (own range, learning rate and step count each) of ONE flat buffer in one launch: the
// groups of a training step differ only in those scalars, and a launch per group costs ~8 us of dispatch for what
// may be a thousand parameters (the camera group).  Blocks [first_block[k], first_block[k + 1]) serve group k.
constexpr uint32_t kAdamMaxGroups = 4;
struct AdamGroups {
    uint32_t n_groups;
    uint32_t first_block[kAdamMaxGroups + 1];
    uint64_t offset[kAdamMaxGroups], n[kAdamMaxGroups];
    float lr[kAdamMaxGroups], bias1[kAdamMaxGroups], bias2_sqrt[kAdamMaxGroups];
    const float* hyper_dev[kAdamMaxGroups];
    const float* bias_dev[kAdamMaxGroups];  // {1 - beta1^t, sqrt(1 - beta2^t)} kept by nvo_opt_commit (nvo_adam_group::bias_dev)
    int vec4[kAdamMaxGroups];
    uint32_t slot[kAdamMaxGroups];  // index of the group's skip flag
    float wd[kAdamMaxGroups];       // L2 weight decay of the group
};

// What may ride behind the groups' step in the SAME launch (nvo_adam_tail): the weight average of the stepped elements
// and, by the last workgroup of the grid once the others have checked in, the commit of the step (nvo_opt_commit /
// k_ema_commit) -- every workgroup has read the scalars the commit changes before it checks in.
struct AdamTail {
    float* ema;
    nvo_h16* ema16;
    float decay;
    uint32_t* ema_step;
    uint32_t ema_slot, ema_commit;
    uint32_t* done;  // nullable: no commit
    uint32_t n_groups, active_mask, scale_mask;
    uint32_t* applied;
    float* scale;
    uint32_t* growth_tracker;
    float growth, backoff;
    uint32_t interval;
    float min_scale, max_scale;
    float* bias;
};

__device__ void opt_commit_thread(uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* __restrict__ applied,
                                  const uint32_t* __restrict__ skip_flags, float* __restrict__ scale,
                                  uint32_t* __restrict__ growth_tracker, float growth, float backoff, uint32_t interval,
                                  float min_scale, float max_scale, float* __restrict__ bias, float beta1, float beta2);

template <typename GT>
__global__ void __launch_bounds__(256)
k_adam_groups(AdamGroups gr, float* __restrict__ p, nvo_h16* __restrict__ p16, const GT* __restrict__ g,
              float* __restrict__ m, float* __restrict__ v, AdamHyper h, const uint32_t* __restrict__ skip_flags,
              Copy16Fmt fmt, const float* __restrict__ loss_scale_dev, AdamTail t) {
    uint32_t k = 0;
    while (k + 1 < gr.n_groups && blockIdx.x >= gr.first_block[k + 1]) ++k;
    __shared__ float fac[2];
    if (t.ema) {  // (uniform) the factors of k_ema_update_dev, from the counter as it stands
        if (threadIdx.x == 0) {
            const double d = (double)t.decay, st = (double)(*t.ema_step) + 1.0;
            fac[0] = (float)(d * (1.0 - pow(d, st - 1.0)));
            fac[1] = (float)(1.0 / (1.0 - pow(d, st)));
        }
        __syncthreads();
    }
    // GradScaler.step decides per optimiser: a group is skipped iff ITS gradients held a non-finite value
    if (!(skip_flags && skip_flags[gr.slot[k]])) {
        h.lr = gr.lr[k];
        h.bias1 = gr.bias1[k];
        h.bias2_sqrt = gr.bias2_sqrt[k];
        h.weight_decay = gr.wd[k];
        if (gr.hyper_dev[k]) {
            h.lr = gr.hyper_dev[k][0];
            h.bias1 = gr.hyper_dev[k][1];
            h.bias2_sqrt = gr.hyper_dev[k][2];
        }
        if (gr.bias_dev[k]) {
            // torch.optim.Adam under GradScaler.step: state['step'] counts the APPLIED steps only -- the counter and the
            // bias corrections of the NEXT applied step live on the device; the commit advances them behind this launch
            // (or as its last act: AdamTail) iff the group was not skipped
            h.bias1 = gr.bias_dev[k][0];
            h.bias2_sqrt = gr.bias_dev[k][1];
        }
        if (loss_scale_dev) h.grad_scale = 1.0f / *loss_scale_dev;  // dynamic loss scale (GradScaler state on the device)
        const uint64_t o = gr.offset[k];
        AdamEma ea{};
        if (t.ema) ea = AdamEma{t.ema + o, t.ema16 ? t.ema16 + o : nullptr, fac[0], 1.0f - t.decay, fac[1]};
        adam_range<GT>(gr.n[k], p + o, p16 ? p16 + o : nullptr, g + o, m + o, v + o, h, gr.vec4[k],
                       blockIdx.x - gr.first_block[k], gr.first_block[k + 1] - gr.first_block[k], fmt, o, ea);
    }
    if (!t.done) return;  // (uniform)
    // Check-in without a round trip: every workgroup adds one to the counter (fire and forget -- a RETURNING atomic per
    // workgroup on one address cost 75 ns each, +90 us on a 1200-workgroup launch), the LAST workgroup of the grid waits
    // for the others and commits (the final load is an acquire).  Workgroups are dispatched in index order, so everything it waits for is resident or
    // done: no deadlock.  What the ordering has to guarantee is only that every READ of the scalars the commit changes
    // (bias corrections, loss scale, learning rates, average counter) precedes it: those loads were consumed before the
    // workgroup's first store, long before its check-in -- so relaxed atomics at agent scope do (they act at the coherent
    // level; a release / __threadfence() per workgroup is an L2 write-back on this multi-XCD part, and nothing here needs
    // another workgroup's WRITES).
    __syncthreads();
    if (blockIdx.x != gridDim.x - 1u) {
        if (threadIdx.x == 0) __hip_atomic_fetch_add(t.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // The counter must arrive at EXACTLY gridDim.x - 1.  A counter someone left dirty (an aborted launch, a caller that
    // did not zero it) ends above that and a stall ends below: either way the wait runs out (~0.5 s) and the launch raises
    // a STICKY error instead of committing -- bit 31 of the counter, which the host checks wherever it reads the step's
    // results (nvo_adam_tail::done_counter) -- and neither commits nor resets: committing on a dirty counter could run
    // before every workgroup has read the scalars the commit changes, i.e. silent optimiser-state corruption.
    __shared__ uint32_t s_commit;
    if (threadIdx.x == 0) {
        uint32_t ok = 0u;
        for (uint32_t spin = 0; spin < (1u << 20); ++spin) {
            const uint32_t v = __hip_atomic_load(t.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (v & 0x80000000u) break;  // (an earlier launch already failed: stay failed)
            if (v == gridDim.x - 1u) {
                ok = 1u;
                break;
            }
            if (v > gridDim.x - 1u) break;  // (dirty: it can only grow)
            __builtin_amdgcn_s_sleep(16);
        }
        if (!ok) __hip_atomic_fetch_or(t.done, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_commit = ok;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_commit) {
        if (t.applied || t.scale)
            opt_commit_thread(t.n_groups, t.active_mask, t.scale_mask, t.applied, skip_flags, t.scale, t.growth_tracker, t.growth,
                              t.backoff, t.interval, t.min_scale, t.max_scale, t.bias, h.beta1, h.beta2);
        if (t.ema_commit && !(skip_flags && skip_flags[t.ema_slot] != 0u)) *t.ema_step += 1u;  // (k_ema_commit)
        *t.done = 0u;  // (the next launch / replay counts from zero again)
    }
}

// A pure streaming read: 16-byte loads, four of them in flight per thread (the scalar grid-stride form ran at
// 2.6 TB/s).  Non-finite <=> exponent field all ones; checked on the raw bits.
template <typename GT>
__device__ __forceinline__ bool nonfinite_range(uint64_t n, const GT* __restrict__ g, uint32_t block_id,
                                                uint32_t n_blocks) {
    constexpr uint32_t kPer = 16 / sizeof(GT);  // elements per 16-byte load
    const uint64_t n_vec = (((uintptr_t)g & 15u) == 0u) ? n / kPer : 0u;
    const uint4* __restrict__ gv = reinterpret_cast<const uint4*>(g);
    const uint64_t stride = (uint64_t)n_blocks * blockDim.x;
    const uint64_t tid = (uint64_t)block_id * blockDim.x + threadIdx.x;
    bool bad = false;
    auto check = [&](uint4 q) {
        if constexpr (sizeof(GT) == 4) {
            bad = bad || (q.x & 0x7F800000u) == 0x7F800000u || (q.y & 0x7F800000u) == 0x7F800000u ||
                  (q.z & 0x7F800000u) == 0x7F800000u || (q.w & 0x7F800000u) == 0x7F800000u;
        } else {
            constexpr uint32_t lo = sizeof(GT) == 2 && __is_same(GT, Bf16) ? 0x7F80u : 0x7C00u;  // exponent field
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                bad = bad || (w[k] & lo) == lo || (w[k] & (lo << 16)) == (lo << 16);
        }
    };
    uint64_t i = tid;
    for (; i + 3 * stride < n_vec; i += 4 * stride) {
        const uint4 a = gv[i], b = gv[i + stride], c = gv[i + 2 * stride], d = gv[i + 3 * stride];
        check(a);
        check(b);
        check(c);
        check(d);
    }
    for (; i < n_vec; i += stride) check(gv[i]);
    for (uint64_t j = n_vec * kPer + tid; j < n; j += stride) {  // unaligned base or tail
        const float x = load_grad(g, j);
        bad = bad || !(fabsf(x) <= 3.0e38f);
    }
    return bad;
}

template <typename GT>
__global__ void __launch_bounds__(256)
k_nonfinite_flag(uint64_t n, const GT* __restrict__ g, uint32_t* __restrict__ flag) {
    const bool bad = nonfinite_range<GT>(n, g, blockIdx.x, gridDim.x);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

struct FlagRanges {
    uint32_t n_ranges;
    uint32_t first_block[kAdamMaxGroups + 1];
    uint64_t offset[kAdamMaxGroups], n[kAdamMaxGroups];
    uint32_t slot[kAdamMaxGroups];  // index of the range in the caller's arrays (its flag word)
};

// several ranges of one gradient buffer in one launch, one flag word per range (reset by the launcher)
template <typename GT>
__global__ void __launch_bounds__(256)
k_nonfinite_flag_ranges(FlagRanges r, const GT* __restrict__ g, uint32_t* __restrict__ flags) {
    uint32_t k = 0;
    while (k + 1 < r.n_ranges && blockIdx.x >= r.first_block[k + 1]) ++k;
    const bool bad = nonfinite_range<GT>(r.n[k], g + r.offset[k], blockIdx.x - r.first_block[k],
                                         r.first_block[k + 1] - r.first_block[k]);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flags + r.slot[k], 1u);
}

// the same with an explicit flag word per span (several spans may raise the same word): the MLP-weight / embedding
// ranges of the parameter groups, scanned by the single-GPU step whose overflow flags are otherwise raised at the source
constexpr uint32_t kMaxSpans = 8;
struct FlagSpans {
    uint32_t n_spans;
    uint32_t first_block[kMaxSpans + 1];
    uint64_t offset[kMaxSpans], n[kMaxSpans];
    uint32_t slot[kMaxSpans];
};
template <typename GT>
__global__ void __launch_bounds__(256)
k_nonfinite_flag_spans(FlagSpans r, const GT* __restrict__ g, uint32_t* __restrict__ flags) {
    uint32_t k = 0;
    while (k + 1 < r.n_spans && blockIdx.x >= r.first_block[k + 1]) ++k;
    const bool bad = nonfinite_range<GT>(r.n[k], g + r.offset[k], blockIdx.x - r.first_block[k],
                                         r.first_block[k + 1] - r.first_block[k]);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flags + r.slot[k], 1u);
}

__global__ void __launch_bounds__(256)
k_cast_half(uint64_t n, const float* __restrict__ src, _Float16* __restrict__ dst) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool aligned = (((uintptr_t)src & 15u) | ((uintptr_t)dst & 7u)) == 0u;
    const uint64_t n_vec = aligned ? n / 4 : 0u;
    for (uint64_t i = tid; i < n_vec; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        typedef _Float16 h4v __attribute__((ext_vector_type(4)));
        reinterpret_cast<h4v*>(dst)[i] = h4v{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    }
    for (uint64_t i = n_vec * 4 + tid; i < n; i += stride) dst[i] = (_Float16)src[i];
}

// fp32 -> 16-bit working copy in the mixed format of Copy16Fmt
__global__ void __launch_bounds__(256)
k_cast_working_copy(uint64_t n, const float* __restrict__ src, nvo_h16* __restrict__ dst, Copy16Fmt fmt) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = nvo_cvt16(src[i], fmt.is_bf16(i));
}

__global__ void __launch_bounds__(256)
k_cast_bf16(uint64_t n, const float* __restrict__ src, uint16_t* __restrict__ dst) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool aligned = (((uintptr_t)src & 15u) | ((uintptr_t)dst & 7u)) == 0u;
    const uint64_t n_vec = aligned ? n / 4 : 0u;
    for (uint64_t i = tid; i < n_vec; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<uint2*>(dst)[i] = make_uint2((uint32_t)to_bf16(v.x) | ((uint32_t)to_bf16(v.y) << 16),
                                                      (uint32_t)to_bf16(v.z) | ((uint32_t)to_bf16(v.w) << 16));
    }
    for (uint64_t i = n_vec * 4 + tid; i < n; i += stride) dst[i] = to_bf16(src[i]);
}

// ---- sharded gradient exchange (reduce-scatter -> Adam on this rank's 1/W slice -> all-gather of the working copy) ----
// The fp32 gradient range [0, n) is cast to the 2-byte wire format as W chunks of `per` elements, each followed by `pad`
// FLAG slots: wire[(i / per) * (per + pad) + i % per] = cvt(src[i]).  A non-finite source value raises *flag (the rank's
// LOCAL overflow flag).  k_flag_to_wire then writes flag ? 1 : 0 into every chunk's pad slots, so that after the SUM
// reduce-scatter rank r finds the number of ranks that overflowed in the pad of ITS chunk -- every rank learns the same
// verdict without a second collective (GradScaler.step must skip the whole group everywhere or nowhere).
template <bool BF>
__global__ void __launch_bounds__(256)
k_cast_shards(uint64_t n, uint64_t per, uint32_t pad, const float* __restrict__ src, uint16_t* __restrict__ dst,
              uint32_t* __restrict__ flag) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    auto cvt = [](float x) -> uint16_t {
        if (BF) return to_bf16(x);
        return __builtin_bit_cast(uint16_t, (_Float16)x);
    };
    // per and pad are multiples of 4 (checked on the host): a 4-vector never straddles a chunk
    for (uint64_t q = tid; q < n / 4; q += stride) {
        const uint64_t i = q * 4;
        const float4 v = reinterpret_cast<const float4*>(src)[q];
        // (what does not survive the cast -- beyond 65504 on an fp16 wire -- counts as an overflow as well; and on that
        // wire the SUM of `world` finite addends must stay finite too, since nothing scans the reduced shard: a rank
        // whose own value exceeds 65504 / world raises the flag, which keeps the verdict global -- the flag slots are
        // summed with the gradient -- and the loss scale identical on all ranks.  bf16 has fp32's range.)
        const float lim = BF ? 3.0e38f : 65504.0f / (float)(n / per);
        bad = bad || !(fabsf(v.x) <= lim) || !(fabsf(v.y) <= lim) || !(fabsf(v.z) <= lim) || !(fabsf(v.w) <= lim);
        const uint64_t o = (i / per) * (per + pad) + (i % per);
        *reinterpret_cast<uint2*>(dst + o) = make_uint2((uint32_t)cvt(v.x) | ((uint32_t)cvt(v.y) << 16),
                                                        (uint32_t)cvt(v.z) | ((uint32_t)cvt(v.w) << 16));
    }
    if (flag && __ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

template <bool BF>
__global__ void k_flag_to_wire(const uint32_t* __restrict__ flag, uint16_t* __restrict__ wire, uint64_t per, uint32_t pad,
                               uint32_t world) {
    const uint16_t one = BF ? (uint16_t)0x3F80u : (uint16_t)0x3C00u;
    const uint16_t v = *flag ? one : (uint16_t)0;
    for (uint32_t t = threadIdx.x; t < world * pad; t += blockDim.x)
        wire[(uint64_t)(t / pad) * (per + pad) + per + (t % pad)] = v;
}

__global__ void k_flag_from_wire(const uint16_t* __restrict__ slot, uint32_t* __restrict__ flag) {
    if (threadIdx.x == 0 && (slot[0] & 0x7FFFu) != 0u) atomicOr(flag, 1u);  // any non-zero count (or a NaN) = overflow
}

// tcnn EmaOptimizer::step [UPSTREAM, restated]: the debiased exponential moving average of the weights,
//   ema_t = (ema_{t-1} * decay * (1 - decay^(t-1)) + w_t * (1 - decay)) / (1 - decay^t),
// kept in fp32 with a 16-bit copy for inference.  skip_flag (the optimiser's): non-zero = the step was skipped, the
// average keeps its value.
__global__ void __launch_bounds__(256)
k_ema_update(uint64_t n, const float* __restrict__ params, float* __restrict__ ema, _Float16* __restrict__ ema_half,
             float keep, float take, float inv_debias, const uint32_t* __restrict__ skip_flag) {
    if (skip_flag && skip_flag[0] != 0u) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float e = (ema[i] * keep + params[i] * take) * inv_debias;
        ema[i] = e;
        if (ema_half) ema_half[i] = (_Float16)e;
    }
}

// the same with the step count kept on the device: *step_dev = number of averages APPLIED so far; a skipped optimiser
// step leaves both the average and the counter alone (k_ema_commit), so the debias factor 1 / (1 - decay^t) never runs
// ahead of the average it normalises
__global__ void __launch_bounds__(256)
k_ema_update_dev(uint64_t n, const float* __restrict__ params, float* __restrict__ ema, _Float16* __restrict__ ema_half,
                 float decay, const uint32_t* __restrict__ step_dev, const uint32_t* __restrict__ skip_flag) {
    if (skip_flag && skip_flag[0] != 0u) return;
    __shared__ float fac[2];
    if (threadIdx.x == 0) {
        const double d = (double)decay, t = (double)(*step_dev) + 1.0;
        fac[0] = (float)(d * (1.0 - pow(d, t - 1.0)));
        fac[1] = (float)(1.0 / (1.0 - pow(d, t)));
    }
    __syncthreads();
    const float keep = fac[0], inv_debias = fac[1], take = 1.0f - decay;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // pure stream (8 B read + 6 B written per weight): 16-byte accesses when the buffers allow it
    uint64_t done = 0;
    if (((((uintptr_t)params) | ((uintptr_t)ema)) & 15u) == 0u && (((uintptr_t)ema_half) & 7u) == 0u) {
        const uint64_t n4 = n >> 2;
        for (uint64_t q = tid; q < n4; q += stride) {
            const float4 pv = reinterpret_cast<const float4*>(params)[q];
            float4 ev = reinterpret_cast<float4*>(ema)[q];
            ev.x = (ev.x * keep + pv.x * take) * inv_debias;
            ev.y = (ev.y * keep + pv.y * take) * inv_debias;
            ev.z = (ev.z * keep + pv.z * take) * inv_debias;
            ev.w = (ev.w * keep + pv.w * take) * inv_debias;
            reinterpret_cast<float4*>(ema)[q] = ev;
            if (ema_half) {
                *reinterpret_cast<uint2*>(ema_half + (q << 2)) =
                    make_uint2(nvo_cvt16x2(ev.x, ev.y, false), nvo_cvt16x2(ev.z, ev.w, false));
            }
        }
        done = n4 << 2;
    }
    for (uint64_t i = done + tid; i < n; i += stride) {
        const float e = (ema[i] * keep + params[i] * take) * inv_debias;
        ema[i] = e;
        if (ema_half) ema_half[i] = __builtin_bit_cast(_Float16, nvo_cvt16(e, false));  // (fp32 first, then fp16: as above)
    }
}
__global__ void k_ema_commit(uint32_t* __restrict__ step_dev, const uint32_t* __restrict__ skip_flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && !(skip_flag && skip_flag[0] != 0u)) *step_dev += 1u;
}

__global__ void __launch_bounds__(256)
k_zero_u32(uint32_t* __restrict__ p, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0u;
}

// GradScaler.update() + the optimisers' step counters, on the device (one thread): group i of `active_mask` advances its
// applied-step counter iff its skip flag is clear; the loss scale backs off when ANY active group saw a non-finite
// gradient and grows after `interval` clean steps (torch.cuda.amp.GradScaler: init 65536, x2 / 2000 steps, x0.5).
__device__ void opt_commit_thread(uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* __restrict__ applied,
                             const uint32_t* __restrict__ skip_flags, float* __restrict__ scale,
                             uint32_t* __restrict__ growth_tracker, float growth, float backoff, uint32_t interval,
                             float min_scale, float max_scale, float* __restrict__ bias, float beta1, float beta2) {
    bool any_bad = false;
    for (uint32_t i = 0; i < n_groups; ++i) {
        const bool bad = skip_flags && skip_flags[i] != 0u;
        if ((scale_mask >> i) & 1u) any_bad = any_bad || bad;
        if (((active_mask >> i) & 1u) && applied && !bad) {
            const uint32_t done = applied[i] + 1u;
            applied[i] = done;
            if (bias) {  // bias corrections of the group's NEXT applied step (double: 1 - 0.999^t loses digits in fp32)
                const double t = (double)done + 1.0;
                bias[2 * i + 0] = (float)(1.0 - pow((double)beta1, t));
                bias[2 * i + 1] = (float)sqrt(1.0 - pow((double)beta2, t));
            }
        }
    }
    if (scale) {
        float sc = *scale;
        uint32_t tr = *growth_tracker;
        if (any_bad) {
            sc = fmaxf(sc * backoff, min_scale);
            tr = 0u;
        } else if (++tr >= interval) {
            sc = fminf(sc * growth, max_scale);
            tr = 0u;
        }
        *scale = sc;
        *growth_tracker = tr;
    }
}

__global__ void k_opt_commit(uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* __restrict__ applied,
                             const uint32_t* __restrict__ skip_flags, float* __restrict__ scale,
                             uint32_t* __restrict__ growth_tracker, float growth, float backoff, uint32_t interval,
                             float min_scale, float max_scale, float* __restrict__ bias, float beta1, float beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    opt_commit_thread(n_groups, active_mask, scale_mask, applied, skip_flags, scale, growth_tracker, growth, backoff, interval,
                      min_scale, max_scale, bias, beta1, beta2);
}

// several device ranges cleared by ONE launch (a training step's accumulate-into buffers); plan: nvo_common.h
__global__ void __launch_bounds__(256)
k_zero_ranges(NvoZeroPlan r) {
    nvo_zero_plan_block(r, blockIdx.x);
}

// ---- deterministic reductions (EngineConfig.deterministic): fixed summation orders instead of float atomics ----
__global__ void __launch_bounds__(256)
k_reduce_partials(const float* __restrict__ partial, uint32_t n_blocks, uint64_t n, float* __restrict__ dst) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float acc = 0.f;
    for (uint32_t b = 0; b < n_blocks; ++b) acc += partial[(uint64_t)b * n + e];
    dst[e] += acc;
}

__global__ void __launch_bounds__(256)
k_reduce_by_camera(uint32_t R, uint32_t K, const float* __restrict__ rows, uint32_t row_stride, const void* __restrict__ cam,
                   int cam_i64x3, float* __restrict__ out) {
    // one workgroup per camera: 8 sub-sequences (rays r = s, s + 8, ...) of 32 lanes (columns), each summed in ray
    // order, then combined s = 0..7 -- every order is fixed, so the result does not depend on scheduling
    __shared__ float part[8][32];
    const uint32_t c = blockIdx.x, k = threadIdx.x & 31u, sub = threadIdx.x >> 5;
    float acc = 0.f;
    for (uint32_t r = sub; r < R; r += 8u) {
        const int64_t cr = cam_i64x3 ? reinterpret_cast<const int64_t*>(cam)[3 * (size_t)r]
                                     : (int64_t)reinterpret_cast<const int32_t*>(cam)[r];
        if (cr == (int64_t)c && k < K) acc += rows[(size_t)r * row_stride + k];
    }
    part[sub][k] = acc;
    __syncthreads();
    if (sub == 0 && k < K) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][k];
        out[(size_t)c * K + k] += t;
    }
}

__global__ void __launch_bounds__(256)
k_color_tiles_to_rays(uint32_t R, uint32_t tiles_per_ray, const float* __restrict__ tile_partial,
                      float* __restrict__ per_ray, float* __restrict__ d_sh) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * 48u) return;
    const uint32_t r = i / 48u, k = i % 48u;
    float acc = 0.f;
    for (uint32_t t = 0; t < tiles_per_ray; ++t) acc += tile_partial[((size_t)r * tiles_per_ray + t) * 48u + k];
    per_ray[i] = acc;
    if (d_sh && k >= 32u) d_sh[(size_t)r * 16u + (k - 32u)] += acc;
}

}  // namespace

int nvo_reduce_partials(hipStream_t stream, const float* partial, uint32_t n_blocks, uint64_t n, float* dst) {
    NVO_REQUIRE(partial && dst, "reduce_partials: NULL argument");
    if (n == 0 || n_blocks == 0) return NVO_OK;
    NVO_PROF(stream, "reduce_partials");
    NVO_LAUNCH(k_reduce_partials, dim3(nvo_div_up(n, 256)), dim3(256), 0, stream, partial, n_blocks, n, dst);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_reduce_by_camera(hipStream_t stream, uint32_t R, uint32_t K, const float* rows, uint32_t row_stride,
                         const void* cam, int cam_i64x3, uint32_t F, float* out) {
    NVO_REQUIRE(rows && cam && out && K >= 1 && K <= 32, "reduce_by_camera: bad argument (K <= 32)");
    if (R == 0 || F == 0) return NVO_OK;
    NVO_PROF(stream, "reduce_by_camera");
    NVO_LAUNCH(k_reduce_by_camera, dim3(F), dim3(256), 0, stream, R, K, rows, row_stride, cam, cam_i64x3, out);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_color_tiles_to_rays(hipStream_t stream, uint32_t R, uint32_t tiles_per_ray, const float* tile_partial,
                            float* per_ray, float* d_sh) {
    NVO_REQUIRE(tile_partial && per_ray && tiles_per_ray >= 1, "color_tiles_to_rays: bad argument");
    if (R == 0) return NVO_OK;
    NVO_LAUNCH(k_color_tiles_to_rays, dim3(nvo_div_up((uint64_t)R * 48, 256)), dim3(256), 0, stream, R, tiles_per_ray,
               tile_partial, per_ray, d_sh);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_zero_async(void* ptr, size_t bytes, hipStream_t stream) {
    NVO_REQUIRE((bytes & 3u) == 0 && ((uintptr_t)ptr & 3u) == 0, "zero_async: %zu bytes not 4-byte granular", bytes);
    if (bytes == 0) return NVO_OK;
    const uint64_t n = bytes / 4;
    uint32_t blocks = nvo_div_up(n, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    NVO_LAUNCH(k_zero_u32, dim3(blocks), dim3(256), 0, stream, (uint32_t*)ptr, n);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_zero_plan_build(uint32_t n_ranges, void* const* ptrs, const uint64_t* bytes, NvoZeroPlan* plan) {
    if (n_ranges > kZeroMaxRanges) {
        nvo_set_error("zero_ranges: at most %u ranges (got %u)", kZeroMaxRanges, n_ranges);
        return -1;
    }
    if (n_ranges != 0 && !(ptrs && bytes)) {
        nvo_set_error("zero_ranges: NULL argument");
        return -1;
    }
    NvoZeroPlan& r = *plan;
    memset(&r, 0, sizeof(r));
    uint32_t k = 0, blocks_total = 0;
    for (uint32_t i = 0; i < n_ranges; ++i) {
        if (bytes[i] == 0) continue;
        if (!(ptrs[i] && (bytes[i] & 3u) == 0 && ((uintptr_t)ptrs[i] & 3u) == 0)) {
            nvo_set_error("zero_ranges: range %u is not 4-byte granular", i);
            return -1;
        }
        r.ptr[k] = (uint32_t*)ptrs[i];
        r.words[k] = bytes[i] / 4;
        uint32_t b = nvo_div_up(r.words[k], (uint64_t)256 * 16);  // 16 dwords per thread and pass
        if (b < 1) b = 1;
        if (b > 512) b = 512;
        r.first_block[k] = blocks_total;
        blocks_total += b;
        ++k;
    }
    r.n = k;
    for (uint32_t i = k; i <= kZeroMaxRanges; ++i) r.first_block[i] = blocks_total;
    return (int)blocks_total;
}

extern "C" {

int nvo_adam_step(nvo_stream_t stream, uint64_t n, float* params, void* params_half,
                  const void* grads, int grads_are_half, float* exp_avg, float* exp_avg_sq, float lr,
                  float beta1, float beta2, float eps, uint32_t step, float grad_scale, float weight_decay,
                  const uint32_t* skip_flag, const float* hyper_dev) {
    NVO_REQUIRE(params && grads && exp_avg && exp_avg_sq, "adam_step: NULL argument");
    NVO_REQUIRE(step >= 1, "adam_step: step counts from 1");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "adam");
    AdamHyper h{lr, beta1, beta2, eps, 1.f - powf(beta1, (float)step), sqrtf(1.f - powf(beta2, (float)step)),
                grad_scale, weight_decay};
    const uintptr_t align = (uintptr_t)params | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq;
    const int vec4 = (align & 15u) == 0 && (!params_half || ((uintptr_t)params_half & 7u) == 0) &&
                     (((uintptr_t)grads & (grads_are_half ? 7u : 15u)) == 0);
    uint32_t blocks = nvo_div_up(n, 256 * 8);
    if (blocks > 4096) blocks = 4096;
    if (grads_are_half == 2) {
        NVO_LAUNCH(k_adam<Bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, params,
                   (nvo_h16*)params_half, (const Bf16*)grads, exp_avg, exp_avg_sq, h, skip_flag, hyper_dev, vec4);
    } else if (grads_are_half) {
        NVO_LAUNCH(k_adam<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, params,
                   (nvo_h16*)params_half, (const _Float16*)grads, exp_avg, exp_avg_sq, h, skip_flag, hyper_dev, vec4);
    } else {
        NVO_LAUNCH(k_adam<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, params,
                   (nvo_h16*)params_half, (const float*)grads, exp_avg, exp_avg_sq, h, skip_flag, hyper_dev, vec4);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

static int make_copy_fmt(uint32_t n_ranges, const uint64_t* lo, const uint64_t* hi, Copy16Fmt* fmt) {
    NVO_REQUIRE(n_ranges <= kMaxBf16Ranges && (n_ranges == 0 || (lo && hi)), "working copy: at most %u bf16 ranges",
                kMaxBf16Ranges);
    memset(fmt, 0, sizeof(*fmt));
    fmt->n = n_ranges;
    for (uint32_t k = 0; k < n_ranges; ++k) {
        NVO_REQUIRE((lo[k] & 3u) == 0 && (hi[k] & 3u) == 0 && lo[k] <= hi[k], "working copy: bf16 range %u is not 4-aligned", k);
        fmt->lo[k] = lo[k];
        fmt->hi[k] = hi[k];
    }
    return NVO_OK;
}

int nvo_adam_step_groups(nvo_stream_t stream, uint32_t n_groups, const nvo_adam_group* groups, float* params,
                         void* params_half, const void* grads, int grads_are_half, float* exp_avg, float* exp_avg_sq,
                         float beta1, float beta2, float eps, float grad_scale, float weight_decay,
                         const uint32_t* skip_flags) {
    return nvo_adam_step_groups_mixed(stream, n_groups, groups, params, params_half, grads, grads_are_half, exp_avg,
                                      exp_avg_sq, beta1, beta2, eps, grad_scale, weight_decay, skip_flags, 0, nullptr,
                                      nullptr);
}

int nvo_cast_working_copy(nvo_stream_t stream, uint64_t n, const float* src, void* dst16, uint32_t n_bf16_ranges,
                          const uint64_t* bf16_lo, const uint64_t* bf16_hi) {
    NVO_REQUIRE(src && dst16, "cast_working_copy: NULL argument");
    Copy16Fmt fmt;
    if (int rc = make_copy_fmt(n_bf16_ranges, bf16_lo, bf16_hi, &fmt)) return rc;
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "cast_half");
    uint32_t blocks = nvo_div_up(n, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    NVO_LAUNCH(k_cast_working_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, src, (nvo_h16*)dst16, fmt);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_adam_step_groups_mixed(nvo_stream_t stream, uint32_t n_groups, const nvo_adam_group* groups, float* params,
                               void* params_half, const void* grads, int grads_are_half, float* exp_avg,
                               float* exp_avg_sq, float beta1, float beta2, float eps, float grad_scale,
                               float weight_decay, const uint32_t* skip_flags, uint32_t n_bf16_ranges,
                               const uint64_t* bf16_lo, const uint64_t* bf16_hi) {
    return nvo_adam_step_groups_scaled(stream, n_groups, groups, params, params_half, grads, grads_are_half, exp_avg,
                                       exp_avg_sq, beta1, beta2, eps, grad_scale, weight_decay, skip_flags,
                                       n_bf16_ranges, bf16_lo, bf16_hi, nullptr);
}

int nvo_opt_commit(nvo_stream_t stream, uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* applied,
                   const uint32_t* skip_flags, float* scale, uint32_t* growth_tracker, float growth_factor,
                   float backoff_factor, uint32_t growth_interval, float min_scale, float max_scale, float* bias,
                   float beta1, float beta2) {
    NVO_REQUIRE(n_groups >= 1 && n_groups <= kAdamMaxGroups, "opt_commit: 1..%u groups (got %u)", kAdamMaxGroups, n_groups);
    NVO_REQUIRE(applied || scale, "opt_commit: nothing to update");
    NVO_REQUIRE(!scale || (growth_tracker && growth_interval >= 1 && growth_factor >= 1.f && backoff_factor > 0.f &&
                           backoff_factor <= 1.f && min_scale > 0.f && max_scale >= min_scale),
                "opt_commit: bad loss-scale schedule");
    NVO_LAUNCH(k_opt_commit, dim3(1), dim3(64), 0, (hipStream_t)stream, n_groups, active_mask, scale_mask, applied, skip_flags, scale,
               growth_tracker, growth_factor, backoff_factor, growth_interval, min_scale, max_scale, bias, beta1, beta2);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_adam_step_groups_scaled(nvo_stream_t stream, uint32_t n_groups, const nvo_adam_group* groups, float* params,
                                void* params_half, const void* grads, int grads_are_half, float* exp_avg,
                                float* exp_avg_sq, float beta1, float beta2, float eps, float grad_scale,
                                float weight_decay, const uint32_t* skip_flags, uint32_t n_bf16_ranges,
                                const uint64_t* bf16_lo, const uint64_t* bf16_hi, const float* loss_scale_dev) {
    return nvo_adam_step_groups_tail(stream, n_groups, groups, params, params_half, grads, grads_are_half, exp_avg, exp_avg_sq,
                                     beta1, beta2, eps, grad_scale, weight_decay, skip_flags, n_bf16_ranges, bf16_lo, bf16_hi,
                                     loss_scale_dev, nullptr);
}

int nvo_adam_step_groups_tail(nvo_stream_t stream, uint32_t n_groups, const nvo_adam_group* groups, float* params,
                              void* params_half, const void* grads, int grads_are_half, float* exp_avg,
                              float* exp_avg_sq, float beta1, float beta2, float eps, float grad_scale,
                              float weight_decay, const uint32_t* skip_flags, uint32_t n_bf16_ranges,
                              const uint64_t* bf16_lo, const uint64_t* bf16_hi, const float* loss_scale_dev,
                              const nvo_adam_tail* tail) {
    NVO_REQUIRE(params && grads && exp_avg && exp_avg_sq && groups, "adam_step_groups: NULL argument");
    AdamTail t{};
    if (tail) {
        NVO_REQUIRE(!tail->ema || (tail->ema_step_dev && tail->ema_decay >= 0.f && tail->ema_decay < 1.f &&
                                   tail->ema_flag_slot < kAdamMaxGroups),
                    "adam_step_groups_tail: weight average needs its step counter, 0 <= decay < 1");
        const bool commit = tail->applied || tail->scale || (tail->ema && tail->ema_commit);
        NVO_REQUIRE(!commit || tail->done_counter, "adam_step_groups_tail: the commit needs done_counter (a zeroed device word)");
        NVO_REQUIRE(!(tail->applied || tail->scale) || (tail->n_commit_groups >= 1 && tail->n_commit_groups <= kAdamMaxGroups),
                    "adam_step_groups_tail: 1..%u commit groups (got %u)", kAdamMaxGroups, tail->n_commit_groups);
        NVO_REQUIRE(!tail->scale || (tail->growth_tracker && tail->growth_interval >= 1 && tail->growth_factor >= 1.f &&
                                     tail->backoff_factor > 0.f && tail->backoff_factor <= 1.f && tail->min_scale > 0.f &&
                                     tail->max_scale >= tail->min_scale),
                    "adam_step_groups_tail: bad loss-scale schedule");
        t.ema = tail->ema;
        t.ema16 = (nvo_h16*)tail->ema_half;
        t.decay = tail->ema_decay;
        t.ema_step = tail->ema ? tail->ema_step_dev : nullptr;
        t.ema_slot = tail->ema_flag_slot;
        t.ema_commit = (tail->ema && tail->ema_commit) ? 1u : 0u;
        t.done = commit ? tail->done_counter : nullptr;
        t.n_groups = tail->n_commit_groups;
        t.active_mask = tail->active_mask;
        t.scale_mask = tail->scale_mask;
        t.applied = tail->applied;
        t.scale = tail->scale;
        t.growth_tracker = tail->growth_tracker;
        t.growth = tail->growth_factor;
        t.backoff = tail->backoff_factor;
        t.interval = tail->growth_interval;
        t.min_scale = tail->min_scale;
        t.max_scale = tail->max_scale;
        t.bias = tail->bias;
    }
    Copy16Fmt fmt;
    if (int rc = make_copy_fmt(n_bf16_ranges, bf16_lo, bf16_hi, &fmt)) return rc;
    NVO_REQUIRE(n_groups >= 1 && n_groups <= kAdamMaxGroups, "adam_step_groups: 1..%u groups (got %u)", kAdamMaxGroups,
                n_groups);
    NVO_PROF(stream, "adam");
    AdamGroups gr{};
    uint32_t k = 0, blocks_total = 0;
    const size_t gsz = grads_are_half ? 2 : 4;
    for (uint32_t i = 0; i < n_groups; ++i) {
        if (groups[i].n == 0) continue;
        NVO_REQUIRE(groups[i].step >= 1 || groups[i].bias_dev, "adam_step_groups: step counts from 1");
        NVO_REQUIRE(groups[i].flag_slot < kAdamMaxGroups, "adam_step_groups: flag_slot %u out of range", groups[i].flag_slot);
        const uint64_t o = groups[i].offset;
        gr.offset[k] = o;
        gr.n[k] = groups[i].n;
        gr.lr[k] = groups[i].lr;
        gr.bias1[k] = 1.f - powf(beta1, (float)(groups[i].step ? groups[i].step : 1u));
        gr.bias2_sqrt[k] = sqrtf(1.f - powf(beta2, (float)(groups[i].step ? groups[i].step : 1u)));
        gr.hyper_dev[k] = groups[i].hyper_dev;
        gr.bias_dev[k] = groups[i].bias_dev;
        // flag word of the group: its index in the caller's array unless the caller pins one (flag_slot + 1)
        gr.slot[k] = groups[i].flag_slot_set ? groups[i].flag_slot : i;
        gr.wd[k] = groups[i].weight_decay_set ? groups[i].weight_decay : weight_decay;
        const uintptr_t align = (uintptr_t)(params + o) | (uintptr_t)(exp_avg + o) | (uintptr_t)(exp_avg_sq + o);
        gr.vec4[k] = (align & 15u) == 0 && (!params_half || (((uintptr_t)params_half + 2 * o) & 7u) == 0) &&
                     ((((uintptr_t)grads + gsz * o) & (grads_are_half ? 7u : 15u)) == 0);
        uint32_t blocks = nvo_div_up(groups[i].n, 256 * 8);
        if (blocks > 4096) blocks = 4096;
        gr.first_block[k] = blocks_total;
        blocks_total += blocks;
        ++k;
    }
    NVO_REQUIRE(k > 0 || !(t.done || t.ema), "adam_step_groups_tail: a tail needs at least one non-empty group");
    if (k == 0) return NVO_OK;
    gr.n_groups = k;
    for (uint32_t i = k; i <= kAdamMaxGroups; ++i) gr.first_block[i] = blocks_total;
    if (t.ema) {
        // (the 16-byte form of a group reads the average like the parameters: same alignment or the scalar form)
        for (uint32_t i = 0; i < k; ++i) {
            const uintptr_t al = (uintptr_t)(t.ema + gr.offset[i]);
            if ((al & 15u) != 0 || (t.ema16 && (((uintptr_t)t.ema16 + 2 * gr.offset[i]) & 7u) != 0)) gr.vec4[i] = 0;
        }
    }
    AdamHyper h{0.f, beta1, beta2, eps, 1.f, 1.f, grad_scale, weight_decay};
    if (grads_are_half == 2) {
        NVO_LAUNCH(k_adam_groups<Bf16>, dim3(blocks_total), dim3(256), 0, (hipStream_t)stream, gr, params,
                   (nvo_h16*)params_half, (const Bf16*)grads, exp_avg, exp_avg_sq, h, skip_flags, fmt, loss_scale_dev, t);
    } else if (grads_are_half) {
        NVO_LAUNCH(k_adam_groups<_Float16>, dim3(blocks_total), dim3(256), 0, (hipStream_t)stream, gr, params,
                   (nvo_h16*)params_half, (const _Float16*)grads, exp_avg, exp_avg_sq, h, skip_flags, fmt, loss_scale_dev, t);
    } else {
        NVO_LAUNCH(k_adam_groups<float>, dim3(blocks_total), dim3(256), 0, (hipStream_t)stream, gr, params,
                   (nvo_h16*)params_half, (const float*)grads, exp_avg, exp_avg_sq, h, skip_flags, fmt, loss_scale_dev, t);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

static int nonfinite_ranges_launch(nvo_stream_t stream, uint32_t n_ranges, const uint64_t* offsets, const uint64_t* sizes,
                                   const void* grads, int grads_are_half, uint32_t* flags, bool reset);

int nvo_nonfinite_flag_ranges(nvo_stream_t stream, uint32_t n_ranges, const uint64_t* offsets, const uint64_t* sizes,
                              const void* grads, int grads_are_half, uint32_t* flags) {
    return nonfinite_ranges_launch(stream, n_ranges, offsets, sizes, grads, grads_are_half, flags, true);
}

int nvo_nonfinite_flag_ranges_or(nvo_stream_t stream, uint32_t n_ranges, const uint64_t* offsets, const uint64_t* sizes,
                                 const void* grads, int grads_are_half, uint32_t* flags) {
    return nonfinite_ranges_launch(stream, n_ranges, offsets, sizes, grads, grads_are_half, flags, false);
}

static int nonfinite_ranges_launch(nvo_stream_t stream, uint32_t n_ranges, const uint64_t* offsets, const uint64_t* sizes,
                                   const void* grads, int grads_are_half, uint32_t* flags, bool reset) {
    uint32_t* flag = flags;
    NVO_REQUIRE(grads && flag && offsets && sizes, "nonfinite_flag_ranges: NULL argument");
    NVO_REQUIRE(n_ranges >= 1 && n_ranges <= kAdamMaxGroups, "nonfinite_flag_ranges: 1..%u ranges (got %u)",
                kAdamMaxGroups, n_ranges);
    NVO_PROF(stream, "nonfinite_flag");
    if (reset)
        if (int rc = nvo_zero_async(flag, sizeof(uint32_t) * n_ranges, (hipStream_t)stream)) return rc;
    FlagRanges r{};
    uint32_t k = 0, blocks_total = 0;
    for (uint32_t i = 0; i < n_ranges; ++i) {
        if (sizes[i] == 0) continue;
        r.offset[k] = offsets[i];
        r.n[k] = sizes[i];
        r.slot[k] = i;
        uint32_t blocks = nvo_div_up(sizes[i], 256 * 8);
        if (blocks > 2048) blocks = 2048;
        r.first_block[k] = blocks_total;
        blocks_total += blocks;
        ++k;
    }
    if (k == 0) return NVO_OK;
    r.n_ranges = k;
    for (uint32_t i = k; i <= kAdamMaxGroups; ++i) r.first_block[i] = blocks_total;
    if (grads_are_half == 2) {
        NVO_LAUNCH(k_nonfinite_flag_ranges<Bf16>, dim3(blocks_total), dim3(256), 0, (hipStream_t)stream, r,
                   (const Bf16*)grads, flag);
    } else if (grads_are_half) {
        NVO_LAUNCH(k_nonfinite_flag_ranges<_Float16>, dim3(blocks_total), dim3(256), 0, (hipStream_t)stream, r,
                   (const _Float16*)grads, flag);
    } else {
        NVO_LAUNCH(k_nonfinite_flag_ranges<float>, dim3(blocks_total), dim3(256), 0, (hipStream_t)stream, r,
                   (const float*)grads, flag);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_nonfinite_flag_spans_or(nvo_stream_t stream, uint32_t n_spans, const uint64_t* offsets, const uint64_t* sizes,
                                const uint32_t* slots, const void* grads, int grads_are_half, uint32_t* flags) {
    NVO_REQUIRE(grads && flags && offsets && sizes && slots, "nonfinite_flag_spans: NULL argument");
    NVO_REQUIRE(n_spans >= 1 && n_spans <= kMaxSpans, "nonfinite_flag_spans: 1..%u spans (got %u)", kMaxSpans, n_spans);
    NVO_PROF(stream, "nonfinite_flag");
    FlagSpans r{};
    uint32_t k = 0, blocks_total = 0;
    for (uint32_t i = 0; i < n_spans; ++i) {
        if (sizes[i] == 0) continue;
        r.offset[k] = offsets[i];
        r.n[k] = sizes[i];
        r.slot[k] = slots[i];
        uint32_t blocks = nvo_div_up(sizes[i], 256 * 8);
        if (blocks > 2048) blocks = 2048;
        r.first_block[k] = blocks_total;
        blocks_total += blocks;
        ++k;
    }
    if (k == 0) return NVO_OK;
    r.n_spans = k;
    for (uint32_t i = k; i <= kMaxSpans; ++i) r.first_block[i] = blocks_total;
    if (grads_are_half == 2) {
        NVO_LAUNCH(k_nonfinite_flag_spans<Bf16>, dim3(blocks_total), dim3(256), 0, (hipStream_t)stream, r, (const Bf16*)grads, flags);
    } else if (grads_are_half) {
        NVO_LAUNCH(k_nonfinite_flag_spans<_Float16>, dim3(blocks_total), dim3(256), 0, (hipStream_t)stream, r, (const _Float16*)grads, flags);
    } else {
        NVO_LAUNCH(k_nonfinite_flag_spans<float>, dim3(blocks_total), dim3(256), 0, (hipStream_t)stream, r, (const float*)grads, flags);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

static int nonfinite_launch(nvo_stream_t stream, uint64_t n, const void* grads, int grads_are_half, uint32_t* flag,
                            bool reset);

int nvo_nonfinite_flag(nvo_stream_t stream, uint64_t n, const void* grads, int grads_are_half, uint32_t* flag) {
    return nonfinite_launch(stream, n, grads, grads_are_half, flag, true);
}

int nvo_nonfinite_flag_or(nvo_stream_t stream, uint64_t n, const void* grads, int grads_are_half, uint32_t* flag) {
    return nonfinite_launch(stream, n, grads, grads_are_half, flag, false);
}

static int nonfinite_launch(nvo_stream_t stream, uint64_t n, const void* grads, int grads_are_half, uint32_t* flag,
                            bool reset) {
    NVO_REQUIRE(grads && flag, "nonfinite_flag: NULL argument");
    NVO_PROF(stream, "nonfinite_flag");
    if (reset)
        if (int rc = nvo_zero_async(flag, sizeof(uint32_t), (hipStream_t)stream)) return rc;
    if (n == 0) return NVO_OK;
    uint32_t blocks = nvo_div_up(n, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (grads_are_half == 2) {
        NVO_LAUNCH(k_nonfinite_flag<Bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, (const Bf16*)grads, flag);
    } else if (grads_are_half) {
        NVO_LAUNCH(k_nonfinite_flag<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n,
                   (const _Float16*)grads, flag);
    } else {
        NVO_LAUNCH(k_nonfinite_flag<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, (const float*)grads, flag);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

struct NvoFloats16 { float v[16]; };
__global__ void k_write_floats(float* dst, uint32_t n, NvoFloats16 vals) {
    if (threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}

int nvo_write_floats(nvo_stream_t stream, float* dst, uint32_t n, const float* host_values) {
    NVO_REQUIRE(dst && host_values && n <= 16, "write_floats: bad argument (n <= 16)");
    if (n == 0) return NVO_OK;
    NvoFloats16 vals;
    for (uint32_t i = 0; i < 16; ++i) vals.v[i] = i < n ? host_values[i] : 0.f;
    NVO_LAUNCH(k_write_floats, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, n, vals);  // values travel by value
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

// GradScaler.update / step counters of the step that just ran AND the per-step scalars of the next one in ONE tiny
// launch: a graph-replayed step ends with its Adam launch, the commit rides in the eager launch that every step needs
// anyway for its scalars (one dependent 5-us launch less per step).
__global__ void k_opt_commit_write(uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* __restrict__ applied,
                                   const uint32_t* __restrict__ skip_flags, float* __restrict__ scale,
                                   uint32_t* __restrict__ growth_tracker, float growth, float backoff, uint32_t interval,
                                   float min_scale, float max_scale, float* __restrict__ bias, float beta1, float beta2,
                                   float* dst, uint32_t n, NvoFloats16 vals) {
    if (threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
    if (threadIdx.x == 63)
        opt_commit_thread(n_groups, active_mask, scale_mask, applied, skip_flags, scale, growth_tracker, growth, backoff, interval,
                          min_scale, max_scale, bias, beta1, beta2);
}

int nvo_opt_commit_write(nvo_stream_t stream, uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* applied,
                         const uint32_t* skip_flags, float* scale, uint32_t* growth_tracker, float growth_factor,
                         float backoff_factor, uint32_t growth_interval, float min_scale, float max_scale, float* bias,
                         float beta1, float beta2, float* dst, uint32_t n, const float* host_values) {
    NVO_REQUIRE(n_groups >= 1 && n_groups <= kAdamMaxGroups, "opt_commit_write: 1..%u groups (got %u)", kAdamMaxGroups, n_groups);
    NVO_REQUIRE(applied || scale, "opt_commit_write: nothing to update");
    NVO_REQUIRE(!scale || (growth_tracker && growth_interval >= 1 && growth_factor >= 1.f && backoff_factor > 0.f &&
                           backoff_factor <= 1.f && min_scale > 0.f && max_scale >= min_scale),
                "opt_commit_write: bad loss-scale schedule");
    NVO_REQUIRE(dst && host_values && n <= 16, "opt_commit_write: bad scalar block (n <= 16)");
    NvoFloats16 vals;
    for (uint32_t i = 0; i < 16; ++i) vals.v[i] = i < n ? host_values[i] : 0.f;
    NVO_LAUNCH(k_opt_commit_write, dim3(1), dim3(64), 0, (hipStream_t)stream, n_groups, active_mask, scale_mask, applied, skip_flags,
               scale, growth_tracker, growth_factor, backoff_factor, growth_interval, min_scale, max_scale, bias, beta1, beta2,
               dst, n, vals);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

// The same commit as a node INSIDE the captured step: the per-step scalars of the next step come from a device table the
// host fills ahead (row s % table_rows = the 16 scalars of step s) and a device step counter the kernel advances -- no
// eager launch behind the replay (measured: 8 us of launch latency between the graph's last kernel and the eager one,
// plus its 5 us, on every step).
__global__ void k_opt_commit_table(uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* __restrict__ applied,
                                   const uint32_t* __restrict__ skip_flags, float* __restrict__ scale,
                                   uint32_t* __restrict__ growth_tracker, float growth, float backoff, uint32_t interval,
                                   float min_scale, float max_scale, float* __restrict__ bias, float beta1, float beta2,
                                   float* dst, const float* __restrict__ table, uint32_t table_rows,
                                   uint32_t* __restrict__ next_step) {
    const uint32_t s = *next_step;
    if (threadIdx.x < 16) dst[threadIdx.x] = table[(size_t)(s % table_rows) * 16u + threadIdx.x];
    if (threadIdx.x == 63)
        opt_commit_thread(n_groups, active_mask, scale_mask, applied, skip_flags, scale, growth_tracker, growth, backoff, interval,
                          min_scale, max_scale, bias, beta1, beta2);
    __syncthreads();
    if (threadIdx.x == 0) *next_step = s + 1u;
}

int nvo_opt_commit_table(nvo_stream_t stream, uint32_t n_groups, uint32_t active_mask, uint32_t scale_mask, uint32_t* applied,
                         const uint32_t* skip_flags, float* scale, uint32_t* growth_tracker, float growth_factor,
                         float backoff_factor, uint32_t growth_interval, float min_scale, float max_scale, float* bias,
                         float beta1, float beta2, float* dst, const float* table, uint32_t table_rows, uint32_t* next_step) {
    NVO_REQUIRE(n_groups >= 1 && n_groups <= kAdamMaxGroups, "opt_commit_table: 1..%u groups (got %u)", kAdamMaxGroups, n_groups);
    NVO_REQUIRE(applied || scale, "opt_commit_table: nothing to update");
    NVO_REQUIRE(!scale || (growth_tracker && growth_interval >= 1 && growth_factor >= 1.f && backoff_factor > 0.f &&
                           backoff_factor <= 1.f && min_scale > 0.f && max_scale >= min_scale),
                "opt_commit_table: bad loss-scale schedule");
    NVO_REQUIRE(dst && table && next_step && table_rows >= 1, "opt_commit_table: scalar table missing");
    NVO_LAUNCH(k_opt_commit_table, dim3(1), dim3(64), 0, (hipStream_t)stream, n_groups, active_mask, scale_mask, applied, skip_flags,
               scale, growth_tracker, growth_factor, backoff_factor, growth_interval, min_scale, max_scale, bias, beta1, beta2,
               dst, table, table_rows, next_step);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

struct FoldEntries {
    uint32_t n_entries;
    float* rep[8];
    float* dst[8];
    uint32_t n_rep[8];
    uint64_t n[8], first[9];  // first[i]: index of entry i's first element in the launch's flat index space
};
__global__ void __launch_bounds__(256)
k_fold_replicas(FoldEntries f) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f.first[f.n_entries]) return;
    uint32_t k = 0;
    while (k + 1 < f.n_entries && i >= f.first[k + 1]) ++k;
    const uint64_t e = i - f.first[k];
    // (all copies requested before the first is used: one memory round trip; fixed summation order -- the copies
    // themselves were filled by float atomics)
    float acc = f.dst[k][e];
    for (uint32_t r0 = 0; r0 < f.n_rep[k]; r0 += 8u) {
        float v[8];
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) v[q] = r0 + q < f.n_rep[k] ? f.rep[k][(size_t)(r0 + q) * f.n[k] + e] : 0.f;
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) {
            acc += v[q];
            if (r0 + q < f.n_rep[k]) f.rep[k][(size_t)(r0 + q) * f.n[k] + e] = 0.f;
        }
    }
    f.dst[k][e] = acc;
}

int nvo_fold_replicas(nvo_stream_t stream, uint32_t n_entries, float* const* replicas, const uint32_t* n_replicas,
                      const uint64_t* n, float* const* dst) {
    NVO_REQUIRE(n_entries >= 1 && n_entries <= 8 && replicas && n_replicas && n && dst, "fold_replicas: 1..8 entries");
    FoldEntries f;
    memset(&f, 0, sizeof(f));
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_entries; ++i) {
        NVO_REQUIRE(replicas[i] && dst[i], "fold_replicas: NULL buffer");
        f.rep[f.n_entries] = replicas[i];
        f.dst[f.n_entries] = dst[i];
        f.n_rep[f.n_entries] = n_replicas[i];
        f.n[f.n_entries] = n[i];
        f.first[f.n_entries] = total;
        total += n[i];
        ++f.n_entries;
    }
    f.first[f.n_entries] = total;
    if (total == 0) return NVO_OK;
    NVO_PROF(stream, "fold_replicas");
    NVO_LAUNCH(k_fold_replicas, dim3((uint32_t)nvo_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, f);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_cast_bf16(nvo_stream_t stream, uint64_t n, const float* src, void* dst_bf16) {
    NVO_REQUIRE(src && dst_bf16, "cast_bf16: NULL argument");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "cast_bf16");
    uint32_t blocks = nvo_div_up(n, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    NVO_LAUNCH(k_cast_bf16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, src, (uint16_t*)dst_bf16);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_cast_shards(nvo_stream_t stream, uint64_t n, uint32_t world, uint32_t pad, const float* src, void* wire16,
                    int wire_fmt, uint32_t* flag) {
    NVO_REQUIRE(src && wire16 && flag, "cast_shards: NULL argument");
    NVO_REQUIRE(world >= 1 && n % ((uint64_t)world * 4) == 0 && pad % 4 == 0 && pad >= 4,
                "cast_shards: n (%llu) must be a multiple of 4 x world (%u), pad (%u) a positive multiple of 4",
                (unsigned long long)n, world, pad);
    NVO_REQUIRE(wire_fmt == 1 || wire_fmt == 2, "cast_shards: wire format 1 (fp16) or 2 (bf16)");
    NVO_REQUIRE((((uintptr_t)src & 15u) | ((uintptr_t)wire16 & 7u)) == 0, "cast_shards: unaligned buffers");
    if (n == 0) return NVO_OK;
    const uint64_t per = n / world;
    NVO_PROF(stream, "cast_shards");
    uint32_t blocks = nvo_div_up(n, 256 * 4 * 4);
    if (blocks > 2048) blocks = 2048;
    if (wire_fmt == 2) {
        NVO_LAUNCH(k_cast_shards<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, per, pad, src, (uint16_t*)wire16, flag);
        NVO_LAUNCH(k_flag_to_wire<true>, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, (uint16_t*)wire16, per, pad, world);
    } else {
        NVO_LAUNCH(k_cast_shards<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, per, pad, src, (uint16_t*)wire16, flag);
        NVO_LAUNCH(k_flag_to_wire<false>, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, (uint16_t*)wire16, per, pad, world);
    }
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_flag_from_wire(nvo_stream_t stream, const void* wire_slot16, uint32_t* flag) {
    NVO_REQUIRE(wire_slot16 && flag, "flag_from_wire: NULL argument");
    NVO_LAUNCH(k_flag_from_wire, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)wire_slot16, flag);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_zero_ranges(nvo_stream_t stream, uint32_t n_ranges, void* const* ptrs, const uint64_t* bytes) {
    NvoZeroPlan r;
    const int blocks_total = nvo_zero_plan_build(n_ranges, ptrs, bytes, &r);
    if (blocks_total < 0) return NVO_ERR_INVALID;
    if (blocks_total == 0) return NVO_OK;
    NVO_PROF(stream, "zero_ranges");
    NVO_LAUNCH(k_zero_ranges, dim3((uint32_t)blocks_total), dim3(256), 0, (hipStream_t)stream, r);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_ema_update(nvo_stream_t stream, uint64_t n, const float* params, float* ema, void* ema_half, float decay,
                   uint32_t step, const uint32_t* skip_flag) {
    NVO_REQUIRE(params && ema, "ema_update: NULL argument");
    NVO_REQUIRE(step >= 1 && decay >= 0.f && decay < 1.f, "ema_update: step counts from 1, 0 <= decay < 1");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "ema_update");
    const double d = (double)decay;
    const float keep = (float)(d * (1.0 - pow(d, (double)step - 1.0)));
    const float inv_debias = (float)(1.0 / (1.0 - pow(d, (double)step)));
    uint32_t blocks = nvo_div_up(n, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    NVO_LAUNCH(k_ema_update, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, params, ema, (_Float16*)ema_half, keep,
               1.0f - decay, inv_debias, skip_flag);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_ema_update_dev_part(nvo_stream_t stream, uint64_t n, const float* params, float* ema, void* ema_half, float decay,
                            const uint32_t* step_dev, const uint32_t* skip_flag) {
    NVO_REQUIRE(params && ema && step_dev, "ema_update_dev_part: NULL argument");
    NVO_REQUIRE(decay >= 0.f && decay < 1.f, "ema_update_dev_part: 0 <= decay < 1");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "ema_update");
    uint32_t blocks = nvo_div_up(n, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    NVO_LAUNCH(k_ema_update_dev, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, params, ema, (_Float16*)ema_half, decay,
               step_dev, skip_flag);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_ema_update_dev(nvo_stream_t stream, uint64_t n, const float* params, float* ema, void* ema_half, float decay,
                       uint32_t* step_dev, const uint32_t* skip_flag) {
    NVO_REQUIRE(params && ema && step_dev, "ema_update_dev: NULL argument");
    NVO_REQUIRE(decay >= 0.f && decay < 1.f, "ema_update_dev: 0 <= decay < 1");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "ema_update");
    uint32_t blocks = nvo_div_up(n, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    NVO_LAUNCH(k_ema_update_dev, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, params, ema, (_Float16*)ema_half, decay,
               (const uint32_t*)step_dev, skip_flag);
    NVO_LAUNCH(k_ema_commit, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, skip_flag);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_cast_half(nvo_stream_t stream, uint64_t n, const float* src, void* dst_half) {
    NVO_REQUIRE(src && dst_half, "cast_half: NULL argument");
    if (n == 0) return NVO_OK;
    NVO_PROF(stream, "cast_half");
    uint32_t blocks = nvo_div_up(n, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    NVO_LAUNCH(k_cast_half, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, src,
                       (_Float16*)dst_half);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

}  // extern "C"
