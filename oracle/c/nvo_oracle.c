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

/* ===============================================================================================
 * Occupancy-grid ray marching (TEST INFRASTRUCTURE; parity unpinned).
 *
 * Restates, in scalar C, the cascaded 128^3 Morton-bitfield occupancy grid and DDA marcher of
 * instant-ngp's NeRF testbed (upstream include/neural-graphics-primitives/nerf_device.cuh:
 * morton3D, calc_dt, mip_from_pos, mip_from_dt, cascaded_grid_idx_at, density_grid_occupied_at,
 * distance_to_next_voxel, advance_to_next_voxel; src/testbed_nerf.cu: generate_training_samples_nerf,
 * grid_to_bitfield, bitfield_max_pool) -- the back-end the reference drives through pyngp
 * (/root/reference/nerf_vo/mapping/instant_ngp.py:33-50,104-105; aabb_scale 4 -> 3 cascades, :37-42).
 * The upstream sources are not vendored; written from the published algorithm (Mueller et al. 2022,
 * section 5 + appendix E).  Everything happens in the normalised frame where cascade 0 is [0,1]^3.
 * ============================================================================================= */
#include <stddef.h>

#define NVO_OCC_GRID 128
#define NVO_OCC_CELLS (128u * 128u * 128u)
#define NVO_OCC_MAX_STEPS 1024u

static uint32_t occ_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
uint32_t nvo_oracle_morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return occ_expand_bits(x) | (occ_expand_bits(y) << 1) | (occ_expand_bits(z) << 2);
}
static float occ_min_step(void) { return 1.7320508075688772f / 1024.0f; }
static float occ_max_step(void) { return (1.7320508075688772f / 1024.0f) * 128.0f * 1024.0f / 128.0f; }
static float occ_calc_dt(float t, float cone_angle) {
    float dt = t * cone_angle;
    if (dt < occ_min_step()) dt = occ_min_step();
    if (dt > occ_max_step()) dt = occ_max_step();
    return dt;
}
static int occ_mip_from_pos(const float p[3], int max_mip) {
    float m = fabsf(p[0] - 0.5f);
    if (fabsf(p[1] - 0.5f) > m) m = fabsf(p[1] - 0.5f);
    if (fabsf(p[2] - 0.5f) > m) m = fabsf(p[2] - 0.5f);
    int e;
    frexpf(m, &e);
    int mip = e + 1;
    if (mip < 0) mip = 0;
    if (mip > max_mip) mip = max_mip;
    return mip;
}
static int occ_mip_from_dt(float dt, const float p[3], int max_mip) {
    int mip = occ_mip_from_pos(p, max_mip);
    dt *= 2.0f * (float)NVO_OCC_GRID;
    if (dt < 1.0f) return mip;
    int e;
    frexpf(dt, &e);
    if (e > mip) mip = e;
    if (mip > max_mip) mip = max_mip;
    return mip;
}
static uint32_t occ_cell_index(const float p[3], int mip) {
    const float s = scalbnf(1.0f, -mip);
    int i[3];
    for (int k = 0; k < 3; ++k) {
        float v = (p[k] - 0.5f) * s + 0.5f;
        i[k] = (int)(v * (float)NVO_OCC_GRID); /* truncation toward zero, as upstream */
        if (i[k] < 0 || i[k] >= NVO_OCC_GRID) return 0xFFFFFFFFu;
    }
    return nvo_oracle_morton3d((uint32_t)i[0], (uint32_t)i[1], (uint32_t)i[2]);
}
static int occ_occupied(const float p[3], const uint8_t* bitfield, int mip) {
    const uint32_t idx = occ_cell_index(p, mip);
    if (idx == 0xFFFFFFFFu) return 0;
    return (bitfield[idx / 8 + (size_t)mip * (NVO_OCC_CELLS / 8)] >> (idx % 8)) & 1;
}
static float occ_sign(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
static float occ_advance_to_next_voxel(float t, float cone_angle, const float p[3], const float d[3],
                                       const float idir[3], int mip) {
    const float res = scalbnf((float)NVO_OCC_GRID, -mip);
    float tmin = 3.0e38f;
    for (int k = 0; k < 3; ++k) {
        const float q = res * (p[k] - 0.5f);
        const float tk = (floorf(q + 0.5f + 0.5f * occ_sign(d[k])) - q) * idir[k];
        if (tk < tmin) tmin = tk;
    }
    float dist = tmin / res;
    if (!(dist > 0.0f)) dist = 0.0f;
    const float t_target = t + dist;
    do {
        t += occ_calc_dt(t, cone_angle);
    } while (t < t_target);
    return t;
}

/* March one ray (normalised frame, unit direction).  aabb = [lo, hi]^3 of the outermost cascade.
 * Writes up to max_out (t, dt) pairs; returns the number of occupied steps found (<= 1024). */
uint32_t nvo_oracle_occ_march_ray(const float o[3], const float d[3], const uint8_t* bitfield, int n_levels,
                                  float cone_angle, float t_near, float jitter, float* t_out, float* dt_out,
                                  uint32_t max_out) {
    const int max_mip = n_levels - 1;
    const float half = 0.5f * (float)(1 << max_mip);
    const float lo = 0.5f - half, hi = 0.5f + half;
    float idir[3], tmin = t_near, tmax = 3.0e38f;
    for (int k = 0; k < 3; ++k) {
        idir[k] = 1.0f / d[k];
        float t0 = (lo - o[k]) * idir[k], t1 = (hi - o[k]) * idir[k];
        if (t0 > t1) { float tt = t0; t0 = t1; t1 = tt; }
        if (t0 > tmin) tmin = t0;
        if (t1 < tmax) tmax = t1;
    }
    if (!(tmax > tmin)) return 0;
    float t = tmin + occ_calc_dt(tmin, cone_angle) * jitter;
    uint32_t j = 0;
    for (;;) {
        float p[3];
        int inside = 1;
        for (int k = 0; k < 3; ++k) {
            p[k] = o[k] + d[k] * t;
            inside = inside && p[k] >= lo && p[k] <= hi;
        }
        if (!inside || j >= NVO_OCC_MAX_STEPS) break;
        const float dt = occ_calc_dt(t, cone_angle);
        const int mip = occ_mip_from_dt(dt, p, max_mip);
        if (occ_occupied(p, bitfield, mip)) {
            if (j < max_out) { t_out[j] = t; dt_out[j] = dt; }
            ++j;
            t += dt;
        } else {
            t = occ_advance_to_next_voxel(t, cone_angle, p, d, idir, mip);
        }
    }
    return j;
}

/* grid: float [n_levels][128^3] (Morton order) -> bitfield bytes [n_levels][128^3 / 8].
 * bit = grid > min(threshold, mean of max(v,0) over cascade 0), then every coarser level ORs in
 * the 2x2x2 max-pool of the next finer level over its inner half (instant-ngp grid_to_bitfield +
 * bitfield_max_pool). */
void nvo_oracle_occ_bitfield(const float* grid, int n_levels, float threshold, uint8_t* bitfield) {
    /* mean of max(v, 0) over ALL cells of cascade 0 (upstream density_grid_mean), accumulated in 2^-20
     * fixed point so that the result does not depend on summation order (the HIP kernel reduces
     * with integer atomics) */
    unsigned long long acc = 0;
    for (uint32_t i = 0; i < NVO_OCC_CELLS; ++i)
        acc += (unsigned long long)llrintf((grid[i] > 0.0f ? grid[i] : 0.0f) * 1048576.0f);
    float mean = (float)((double)acc / 1048576.0 / (double)NVO_OCC_CELLS);
    const float th = mean < threshold ? mean : threshold;
    for (int l = 0; l < n_levels; ++l)
        for (uint32_t b = 0; b < NVO_OCC_CELLS / 8; ++b) {
            uint8_t bits = 0;
            for (int j = 0; j < 8; ++j)
                if (grid[(size_t)l * NVO_OCC_CELLS + b * 8 + j] > th) bits |= (uint8_t)(1u << j);
            bitfield[(size_t)l * (NVO_OCC_CELLS / 8) + b] = bits;
        }
    for (int l = 1; l < n_levels; ++l) {
        const uint8_t* fine = bitfield + (size_t)(l - 1) * (NVO_OCC_CELLS / 8);
        uint8_t* coarse = bitfield + (size_t)l * (NVO_OCC_CELLS / 8);
        /* a byte of the fine level holds the 8 Morton-consecutive cells = one 2x2x2 block */
        for (uint32_t x = 0; x < 64; ++x)
            for (uint32_t y = 0; y < 64; ++y)
                for (uint32_t z = 0; z < 64; ++z) {
                    const uint32_t fine_idx = nvo_oracle_morton3d(2 * x, 2 * y, 2 * z);
                    if (fine[fine_idx / 8]) {
                        const uint32_t ci = nvo_oracle_morton3d(x + 32, y + 32, z + 32);
                        coarse[ci / 8] |= (uint8_t)(1u << (ci % 8));
                    }
                }
    }
}
