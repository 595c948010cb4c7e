// Real spherical-harmonics direction encoding, degree <= 4 (16 coefficients), forward + input
// gradient.  Replaces tiny-cuda-nn's kernel_sh / kernel_sh_backward (SURVEY.md section 2.4 K6;
// upstream spherical_harmonics.h is not vendored -- restated in oracle/sh.py).  Input follows the
// tcnn convention: d01 = (d + 1) / 2 in [0,1]^3, the kernel maps it back with 2*d01 - 1.
#include "nvo_kernels.h"

namespace {

// out: [N][out_stride] fp16 (bf = 0) or bfloat16 (bf = 1); writes n_coeff coefficients then pads up to out_width with 1.0
__global__ void __launch_bounds__(256)
k_sh_fwd(uint32_t N, uint32_t degree, const float* __restrict__ d01, nvo_h16* __restrict__ out,
         uint32_t out_stride, uint32_t out_width, int bf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float x = d01[3 * (size_t)i + 0] * 2.f - 1.f;
    const float y = d01[3 * (size_t)i + 1] * 2.f - 1.f;
    const float z = d01[3 * (size_t)i + 2] * 2.f - 1.f;
    float o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = 0.f;
    nvo_sh4_eval(x, y, z, degree, o);
    const uint32_t n_coeff = degree * degree;
    nvo_h16* __restrict__ p = out + (size_t)i * out_stride;
    for (uint32_t k = 0; k < out_width; ++k) p[k] = nvo_cvt16(k < n_coeff ? o[k] : 1.0f, bf != 0);
}

// dL/dd01 = 2 * sum_k dL/dy_k * dy_k/d(x,y,z)
__device__ __forceinline__ float sh_to_float(__half v) { return __half2float(v); }
__device__ __forceinline__ float sh_to_float(float v) { return v; }

template <typename DY>
__global__ void __launch_bounds__(256)
k_sh_bwd_input(uint32_t N, uint32_t degree, const float* __restrict__ d01,
               const DY* __restrict__ dy, uint32_t dy_stride, float* __restrict__ dd01) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float x = d01[3 * (size_t)i + 0] * 2.f - 1.f;
    const float y = d01[3 * (size_t)i + 1] * 2.f - 1.f;
    const float z = d01[3 * (size_t)i + 2] * 2.f - 1.f;
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    float g[16];
    const uint32_t n_coeff = degree * degree;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) g[k] = k < n_coeff ? sh_to_float(dy[(size_t)i * dy_stride + k]) : 0.f;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    // degree 2
    gy += -0.48860251190291987f * g[1];
    gz += 0.48860251190291987f * g[2];
    gx += -0.48860251190291987f * g[3];
    // degree 3
    gx += 1.0925484305920792f * y * g[4];
    gy += 1.0925484305920792f * x * g[4];
    gy += -1.0925484305920792f * z * g[5];
    gz += -1.0925484305920792f * y * g[5];
    gz += 2.f * 0.94617469575755997f * z * g[6];
    gx += -1.0925484305920792f * z * g[7];
    gz += -1.0925484305920792f * x * g[7];
    gx += 2.f * 0.54627421529603959f * x * g[8];
    gy += -2.f * 0.54627421529603959f * y * g[8];
    // degree 4
    gx += 0.59004358992664352f * (-6.f * x * y) * g[9];
    gy += 0.59004358992664352f * (-3.f * x2 + 3.f * y2) * g[9];
    gx += 2.8906114426405538f * y * z * g[10];
    gy += 2.8906114426405538f * x * z * g[10];
    gz += 2.8906114426405538f * x * y * g[10];
    gy += 0.45704579946446572f * (1.f - 5.f * z2) * g[11];
    gz += 0.45704579946446572f * (-10.f * y * z) * g[11];
    gz += 0.3731763325901154f * (15.f * z2 - 3.f) * g[12];
    gx += 0.45704579946446572f * (1.f - 5.f * z2) * g[13];
    gz += 0.45704579946446572f * (-10.f * x * z) * g[13];
    gx += 1.4453057213202769f * (2.f * x * z) * g[14];
    gy += 1.4453057213202769f * (-2.f * y * z) * g[14];
    gz += 1.4453057213202769f * (x2 - y2) * g[14];
    gx += 0.59004358992664352f * (-3.f * x2 + 3.f * y2) * g[15];
    gy += 0.59004358992664352f * (6.f * x * y) * g[15];
    dd01[3 * (size_t)i + 0] = 2.f * gx;
    dd01[3 * (size_t)i + 1] = 2.f * gy;
    dd01[3 * (size_t)i + 2] = 2.f * gz;
}

}  // namespace

int nvo_sh_fwd_launch(hipStream_t stream, uint32_t N, uint32_t degree, const float* d01,
                      void* out_half, uint32_t out_stride, uint32_t out_width, bool out_bf16) {
    NVO_REQUIRE(degree >= 1 && degree <= 4, "SphericalHarmonics: degree %u not in 1..4", degree);
    if (N == 0) return NVO_OK;
    NVO_PROF(stream, "sh_fwd");
    NVO_LAUNCH(k_sh_fwd, dim3(nvo_div_up(N, 256)), dim3(256), 0, stream, N, degree, d01,
                       (nvo_h16*)out_half, out_stride, out_width, out_bf16 ? 1 : 0);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

int nvo_sh_bwd_input_launch(hipStream_t stream, uint32_t N, uint32_t degree, const float* d01,
                            const void* dy_half, uint32_t dy_stride, float* dd01) {
    NVO_REQUIRE(degree >= 1 && degree <= 4, "SphericalHarmonics: degree %u not in 1..4", degree);
    if (N == 0) return NVO_OK;
    NVO_PROF(stream, "sh_bwd_input");
    NVO_LAUNCH(k_sh_bwd_input<__half>, dim3(nvo_div_up(N, 256)), dim3(256), 0, stream, N, degree,
                       d01, (const __half*)dy_half, dy_stride, dd01);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}

// dy: float [N][16] (the colour head's per-ray SH gradient) -> dd01 [N][3]
extern "C" int nvo_sh_bwd_input_f32(void* stream, uint32_t N, uint32_t degree, const float* d01,
                                    const float* dy, float* dd01) {
    NVO_REQUIRE(degree >= 1 && degree <= 4, "sh_bwd_input_f32: degree %u not in 1..4", degree);
    NVO_REQUIRE(N == 0 || (d01 && dy && dd01), "sh_bwd_input_f32: NULL argument");
    if (N == 0) return NVO_OK;
    NVO_PROF(stream, "sh_bwd_input");
    NVO_LAUNCH(k_sh_bwd_input<float>, dim3(nvo_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N, degree,
               d01, dy, 16u, dd01);
    NVO_CHECK_LAUNCH();
    return NVO_OK;
}
