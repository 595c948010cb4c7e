// Keyframe depth alignment on the GPU: the step right before the mapping hot
// path.  Monocular depth maps are brought to the metric scale of DPVO's sparse patches by a per-frame
// scale / shift after a quantile-based outlier removal of the patches.  Mirrors, operation for operation,
// /root/reference/nerf_vo/enhancement/enhancement_module.py:61-99 (alignment) and :131-146
// (dpvo_remove_outliers, including its global boolean-mask compaction + reshape and the `except` fallback);
// the reference does this with ~25 torch ops and a host-visible exception path per keyframe batch.
// CPU restatement (pinned against the reference's own outputs): oracle/enhancement.py.
//
//   k_patch_flags   (one workgroup per frame)  centre inverse depth + tie-breaking noise, 1/12 and 11/12
//                                              quantiles with torch.quantile's linear interpolation, keep flags
//   k_patch_compact (one workgroup)            global exclusive scan of the flags (row-major, like boolean-mask
//                                              indexing), survivor count check, fallback mean
//   k_frame_affine  (one workgroup per frame)  gathers the mono depth under every surviving patch centre,
//                                              mean / unbiased std of both samples, frame mean -> scale, shift
//   k_apply_affine  (elementwise)              clip(depth * scale + shift, 0, 5)
#include "nvo_kernels.h"
#include "../../include/nerfvo_hip.h"

#include <math.h>

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kMaxPatches = 1024;  // patches per frame held in LDS

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = nvo_wave_sum(v);  // DPP form (nvo_common.h)
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < kBlock / 64; ++w) s += red[w];
    return s;
}

// torch.quantile(..., interpolation='linear') on a sorted array: rank = q * (n - 1) in fp32, lerp with
// torch.lerp's two-sided formula
__device__ __forceinline__ float quantile_sorted(const float* a, uint32_t n, float q) {
    const float rank = q * (float)(n - 1);
    const float lo_f = floorf(rank);
    const uint32_t lo = (uint32_t)lo_f, hi = min(lo + 1u, n - 1u);
    const float w = rank - lo_f;
    const float s = a[lo], e = a[hi];
    return w < 0.5f ? s + w * (e - s) : e - (e - s) * (1.f - w);
}

__global__ void __launch_bounds__(kBlock)
k_patch_flags(uint32_t M, uint32_t P, const float* __restrict__ patches, const float* __restrict__ noise,
              uint32_t* __restrict__ keep) {
    __shared__ float v[kMaxPatches], sorted[kMaxPatches];
    const uint32_t k = blockIdx.x;
    const uint32_t pp = P * P, centre = (P / 2) * P + (P / 2);
    for (uint32_t m = threadIdx.x; m < M; m += kBlock)
        v[m] = patches[(((size_t)k * M + m) * 3 + 2) * pp + centre] + noise[(size_t)k * M + m] * 1e-4f;
    __syncthreads();
    // rank sort (M <= 1024: M^2 comparisons are nothing); ties cannot matter for the flags
    for (uint32_t m = threadIdx.x; m < M; m += kBlock) {
        const float x = v[m];
        uint32_t r = 0;
        for (uint32_t j = 0; j < M; ++j) r += (v[j] < x || (v[j] == x && j < m)) ? 1u : 0u;
        sorted[r] = x;
    }
    __syncthreads();
    const float q_lo = quantile_sorted(sorted, M, (float)(1.0 / 12.0));
    const float q_hi = quantile_sorted(sorted, M, (float)(11.0 / 12.0));
    for (uint32_t m = threadIdx.x; m < M; m += kBlock)
        keep[(size_t)k * M + m] = (v[m] < q_lo || v[m] > q_hi) ? 0u : 1u;
}

// header[0] = rows per frame after the removal (M' = int(M*5/6), or M in the fallback), header[1] = fallback
// flag, header[2] = bits of the global mean of the noisy patch tensor (fallback replacement value)
__global__ void __launch_bounds__(kBlock)
k_patch_compact(uint32_t K, uint32_t M, uint32_t P, const float* __restrict__ patches,
                const float* __restrict__ noise, const uint32_t* __restrict__ keep, uint32_t* __restrict__ src,
                uint32_t* __restrict__ header) {
    __shared__ uint32_t part[kBlock];
    __shared__ float red[kBlock / 64];
    const uint32_t n = K * M;
    const uint32_t per = (n + kBlock - 1) / kBlock;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint32_t c = 0;
    for (uint32_t i = lo; i < hi; ++i) c += keep[i];
    part[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < kBlock; ++t) {
            const uint32_t x = part[t];
            part[t] = run;
            run += x;
        }
        const uint32_t want = (uint32_t)((double)M * 5.0 / 6.0);
        const bool ok = run == K * want;
        header[0] = ok ? want : M;
        header[1] = ok ? 0u : 1u;
    }
    __syncthreads();
    const bool fallback = header[1] != 0u;
    if (!fallback) {
        uint32_t run = part[threadIdx.x];
        for (uint32_t i = lo; i < hi; ++i)
            if (keep[i]) src[run++] = i;
    } else {
        for (uint32_t i = lo; i < hi; ++i) src[i] = i;
    }
    // global mean of the (noisy) patch tensor: the fallback replaces every element < 1e-3 with it
    const uint32_t pp = P * P;
    const size_t total = (size_t)n * 3 * pp;
    float s = 0.f;
    for (size_t e = threadIdx.x; e < total; e += kBlock) {
        const size_t patch = e / (3 * pp);
        const uint32_t ch = (uint32_t)((e / pp) % 3);
        s += patches[e] + (ch == 2 ? noise[patch] * 1e-4f : 0.f);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) header[2] = __float_as_uint(s / (float)total);
}

__global__ void __launch_bounds__(kBlock)
k_frame_affine(uint32_t M, uint32_t P, uint32_t H, uint32_t W, const float* __restrict__ patches,
               const float* __restrict__ noise, const float* __restrict__ frames_depth,
               const uint32_t* __restrict__ src, const uint32_t* __restrict__ header,
               float* __restrict__ scale_shift) {
    __shared__ float red[kBlock / 64];
    __shared__ float sparse_s[kMaxPatches], at_s[kMaxPatches];
    const uint32_t k = blockIdx.x;
    const uint32_t Mp = header[0];
    const bool fallback = header[1] != 0u;
    const float repl = __uint_as_float(header[2]);
    const uint32_t pp = P * P, centre = (P / 2) * P + (P / 2);
    const float* frame = frames_depth + (size_t)k * H * W;
    for (uint32_t j = threadIdx.x; j < Mp; j += kBlock) {
        const uint32_t i = src[(size_t)k * Mp + j];  // source patch (frame-major index into [K*M])
        const float* p = patches + (size_t)i * 3 * pp;
        float px = p[centre], py = p[pp + centre], inv = p[2 * pp + centre] + noise[i] * 1e-4f;
        if (fallback) {
            if (px < 1e-3f) px = repl;
            if (py < 1e-3f) py = repl;
            if (inv < 1e-3f) inv = repl;
        }
        const float x = px * 4.f, y = py * 4.f;
        const float d = fminf(fmaxf(1.f / inv, 0.f), 5.f);
        // .long(): truncation toward zero; clamped so that a patch on the image border cannot fault
        const uint32_t xi = (uint32_t)min(max((int)x, 0), (int)W - 1), yi = (uint32_t)min(max((int)y, 0), (int)H - 1);
        sparse_s[j] = d;
        at_s[j] = frame[(size_t)yi * W + xi];
    }
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
    for (uint32_t j = threadIdx.x; j < Mp; j += kBlock) {
        s0 += sparse_s[j];
        s1 += at_s[j];
    }
    const float mean_sparse = block_sum(s0, red) / (float)Mp;
    const float mean_at = block_sum(s1, red) / (float)Mp;
    float v0 = 0.f, v1 = 0.f;
    for (uint32_t j = threadIdx.x; j < Mp; j += kBlock) {
        const float a = sparse_s[j] - mean_sparse, b = at_s[j] - mean_at;
        v0 += a * a;
        v1 += b * b;
    }
    const float std_sparse = sqrtf(block_sum(v0, red) / (float)(Mp - 1));  // torch.std: unbiased
    const float std_at = sqrtf(block_sum(v1, red) / (float)(Mp - 1));
    float fs = 0.f;
    for (uint32_t e = threadIdx.x; e < H * W; e += kBlock) fs += frame[e];
    const float mean_frame = block_sum(fs, red) / (float)(H * W);
    if (threadIdx.x == 0) {
        const float scale = std_sparse / std_at;
        scale_shift[2 * k + 0] = scale;
        scale_shift[2 * k + 1] = mean_frame * (mean_sparse / mean_at - scale);
    }
}

__global__ void __launch_bounds__(kBlock)
k_apply_affine(uint32_t HW, const float* __restrict__ frames_depth, const float* __restrict__ scale_shift,
               float* __restrict__ out) {
    const uint32_t k = blockIdx.y;
    const float scale = scale_shift[2 * k], shift = scale_shift[2 * k + 1];
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < HW; e += gridDim.x * kBlock) {
        const size_t i = (size_t)k * HW + e;
        out[i] = fminf(fmaxf(frames_depth[i] * scale + shift, 0.f), 5.f);
    }
}

}  // namespace

extern "C" {

uint64_t nvo_depth_align_scratch_bytes(uint32_t K, uint32_t M) {
    // keep flags [K*M] | source indices [K*M] | header [4] | scale/shift [2K]
    return sizeof(uint32_t) * (2ull * K * M + 4) + sizeof(float) * 2ull * K;
}

int nvo_depth_align(nvo_stream_t stream, const nvo_depth_align_args* args) {
    NVO_REQUIRE(args != nullptr, "depth_align: args is NULL");
    const nvo_depth_align_args a = *args;
    NVO_REQUIRE(a.K >= 1 && a.M >= 6 && a.M <= kMaxPatches, "depth_align: patches per frame %u not in 6..%u", a.M, kMaxPatches);
    NVO_REQUIRE(a.P >= 1 && (a.P & 1u), "depth_align: patch size %u must be odd", a.P);
    NVO_REQUIRE(a.H >= 1 && a.W >= 1 && a.patches && a.noise && a.frames_depth && a.out_depth && a.scratch,
                "depth_align: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    uint32_t* keep = (uint32_t*)a.scratch;
    uint32_t* src = keep + (size_t)a.K * a.M;
    uint32_t* header = src + (size_t)a.K * a.M;
    float* scale_shift = (float*)(header + 4);
    NVO_PROF(stream, "depth_align");
    NVO_LAUNCH(k_patch_flags, dim3(a.K), dim3(kBlock), 0, s, a.M, a.P, a.patches, a.noise, keep);
    NVO_LAUNCH(k_patch_compact, dim3(1), dim3(kBlock), 0, s, a.K, a.M, a.P, a.patches, a.noise, keep, src, header);
    NVO_LAUNCH(k_frame_affine, dim3(a.K), dim3(kBlock), 0, s, a.M, a.P, a.H, a.W, a.patches, a.noise, a.frames_depth, src,
               header, scale_shift);
    const uint32_t hw = a.H * a.W;
    uint32_t bx = nvo_div_up(hw, kBlock * 4);
    if (bx > 1024) bx = 1024;
    NVO_LAUNCH(k_apply_affine, dim3(bx, a.K), dim3(kBlock), 0, s, hw, a.frames_depth, scale_shift, a.out_depth);
    NVO_CHECK_LAUNCH();
    if (a.scale_shift_out)
        NVO_CHECK_HIP(hipMemcpyAsync(a.scale_shift_out, scale_shift, sizeof(float) * 2 * a.K, hipMemcpyDeviceToDevice, s));
    return NVO_OK;
}

}  // extern "C"
