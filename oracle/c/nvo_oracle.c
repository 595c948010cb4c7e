/*
 * CPU oracle, integer / index paths (TEST INFRASTRUCTURE -- see oracle/__init__.py; parity unpinned).
 *
 * Restates, in scalar C with glibc libm, the index arithmetic of tiny-cuda-nn's multiresolution hash
 * grid (upstream include/tiny-cuda-nn/encodings/grid.h: GridEncodingTemplated ctor, grid_scale,
 * grid_resolution, pos_fract, grid_index, coherent_prime_hash), which the reference reaches through
 * nerfstudio's NerfactoField / HashMLPDensityField (call site
 * /root/reference/nerf_vo/mapping/nerfstudio.py:151 -> trainer.train_iteration).  The upstream
 * source is not vendored in /root/reference; this is written from the published algorithm
 * (Mueller et al. 2022, "Instant Neural Graphics Primitives", section 3 + appendix A).
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC -> oracle/libnvo_oracle.so)
 */
#include <math.h>
#include <stdint.h>

/* levels_out[l] = {offset, size, resolution, hashed}; returns total entries */
uint32_t nvo_oracle_level_table(uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution,
                                float per_level_scale, uint32_t* levels_out, float* scales_out) {
    const float log2_pls = log2f(per_level_scale);
    uint32_t offset = 0;
    for (uint32_t l = 0; l < n_levels; ++l) {
        const float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2u;
        uint32_t n = powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
        n = (n + 7u) / 8u * 8u; /* rows stay 16-byte aligned */
        if (n > (1u << log2_hashmap_size)) n = 1u << log2_hashmap_size;
        /* grid_index's dense loop runs while stride <= size; the hash is used iff size < final stride */
        uint32_t stride = 1;
        for (int d = 0; d < 3 && stride <= n; ++d) stride *= res;
        levels_out[4 * l + 0] = offset;
        levels_out[4 * l + 1] = n;
        levels_out[4 * l + 2] = res;
        levels_out[4 * l + 3] = n < stride ? 1u : 0u;
        scales_out[l] = scale;
        offset += n;
    }
    return offset;
}

static uint32_t grid_index(uint32_t size, uint32_t res, const uint32_t p[3]) {
    uint32_t stride = 1, index = 0;
    for (int d = 0; d < 3 && stride <= size; ++d) {
        index += p[d] * stride;
        stride *= res;
    }
    if (size < stride) {
        static const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
        index = 0;
        for (int d = 0; d < 3; ++d) index ^= p[d] * primes[d];
    }
    return index % size;
}

/* x: [n][3] float; indices_out: [n_levels][n][8] (level-relative entry index);
 * weights_out (nullable): [n_levels][n][8] trilinear weights in fp32 */
void nvo_oracle_grid_indices(uint32_t n_levels, const uint32_t* levels, const float* scales, uint32_t n,
                             const float* x, uint32_t* indices_out, float* weights_out) {
    for (uint32_t l = 0; l < n_levels; ++l) {
        const uint32_t size = levels[4 * l + 1], res = levels[4 * l + 2];
        const float scale = scales[l];
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t cell[3];
            float frac[3];
            for (int d = 0; d < 3; ++d) {
                const float pos = fmaf(scale, x[3 * i + d], 0.5f);
                const float fl = floorf(pos);
                cell[d] = (uint32_t)(int)fl;
                frac[d] = pos - fl;
            }
            for (uint32_t k = 0; k < 8; ++k) {
                uint32_t p[3];
                float w = 1.0f;
                for (int d = 0; d < 3; ++d) {
                    if (k & (1u << d)) {
                        p[d] = cell[d] + 1u;
                        w *= frac[d];
                    } else {
                        p[d] = cell[d];
                        w *= 1.0f - frac[d];
                    }
                }
                indices_out[((uint64_t)l * n + i) * 8 + k] = grid_index(size, res, p);
                if (weights_out) weights_out[((uint64_t)l * n + i) * 8 + k] = w;
            }
        }
    }
}
