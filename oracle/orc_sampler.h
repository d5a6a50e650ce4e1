// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Parity unpinned.
// Restates src/pathtracer/sampler/sobol.rs, sampler/mod.rs:9-91 (CoreSampler with no arrays) and
// src/pathtracer/lowdiscrepancy.rs.  Tables: data/sobol_tables.bin (numbers extracted from
// sobolmatrices.rs by tools/extract_sobol_tables.py).
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "orc_math.h"

namespace orc {

struct SobolTables {
    uint32_t num_dims = 0, matrix_size = 0, stride = 0;
    std::vector<uint32_t> matrices;      // [1024*52]   sobolmatrices.rs:7
    std::vector<uint64_t> vdc, vdc_inv;  // [25*52], [26*52] zero padded (53463, 54155)
    bool load(const char *path) {
        FILE *f = std::fopen(path, "rb");
        if (!f) return false;
        char magic[8];
        uint32_t hdr[6];
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTRSSOB1", 8) == 0 && std::fread(hdr, 4, 6, f) == 6;
        if (ok) {
            num_dims = hdr[0]; matrix_size = hdr[1]; stride = hdr[4];
            matrices.resize((size_t)num_dims * matrix_size);
            uint32_t lens[25 + 26 + 1];
            vdc.resize((size_t)hdr[2] * stride);
            vdc_inv.resize((size_t)hdr[3] * stride);
            ok = std::fread(matrices.data(), 4, matrices.size(), f) == matrices.size() && std::fread(lens, 4, 52, f) == 52 &&
                 std::fread(vdc.data(), 8, vdc.size(), f) == vdc.size() && std::fread(vdc_inv.data(), 8, vdc_inv.size(), f) == vdc_inv.size();
        }
        std::fclose(f);
        return ok;
    }
};

// lowdiscrepancy.rs:9-39
inline uint64_t sobol_interval_to_index(const SobolTables &T, uint32_t m, uint64_t frame, int32_t px, int32_t py) {
    if (m == 0) return 0;
    const uint32_t m2 = m << 1;
    uint64_t index = frame << m2;
    uint64_t delta = 0;
    for (uint32_t c = 0; frame != 0; frame >>= 1, c++)
        if (frame & 1) delta ^= T.vdc[(size_t)(m - 1) * T.stride + c];
    uint64_t b = ((uint64_t)((uint32_t)px << m) | (uint64_t)(int64_t)py) ^ delta; // (p.x as u32) << m, p.y as u64
    for (uint32_t c = 0; b != 0; b >>= 1, c++)
        if (b & 1) index ^= T.vdc_inv[(size_t)(m - 1) * T.stride + c];
    return index;
}

// lowdiscrepancy.rs:42-57 (a: i64, arithmetic shift; indices are non-negative)
inline float sobol_sample(const SobolTables &T, int64_t a, size_t dimension, uint64_t scramble) {
    uint32_t v = (uint32_t)scramble;
    for (size_t i = dimension * T.matrix_size; a != 0; a >>= 1, i++)
        if (a & 1) v ^= T.matrices[i];
    return fmin_rs(ONE_MINUS_EPSILON, (float)v * 0x1p-32f);
}

struct Bounds2i { int32_t min_x, min_y, max_x, max_y; };

// sobol.rs:13-193; ARRAY_START_DIM = 5 and no sample arrays are ever requested, so
// array_end_dim == 5 and get_2d at dimension 4 jumps to 5 (Q2).
struct SobolSampler {
    const SobolTables *T = nullptr;
    size_t samples_per_pixel = 1;
    Bounds2i sample_bounds{};
    int32_t resolution = 1;
    uint32_t log_2_resolution = 0;
    // state
    int32_t cur_x = 0, cur_y = 0;
    size_t current_pixel_sample_index = 0;
    size_t dimension = 0;
    int64_t interval_sample_index = 0;
    uint64_t current_scramble_index = 0;
    static constexpr size_t ARRAY_START_DIM = 5;
    size_t array_end_dim = 0;

    // SobolSamplerBuilder::new, sobol.rs:35-60
    void configure(const SobolTables *tables, size_t spp, Bounds2i sb) {
        T = tables;
        samples_per_pixel = (size_t)round_up_pow2_i64((int64_t)spp);
        sample_bounds = sb;
        int32_t dx = sb.max_x - sb.min_x, dy = sb.max_y - sb.min_y;
        resolution = round_up_pow2_i32(dx > dy ? dx : dy);
        log_2_resolution = log2_int((uint64_t)resolution);
    }
    int64_t get_index_for_sample(uint64_t sample_num) const { // 169-175
        return (int64_t)sobol_interval_to_index(*T, log_2_resolution, sample_num, cur_x - sample_bounds.min_x, cur_y - sample_bounds.min_y);
    }
    void start_pixel(int32_t x, int32_t y) { // 81-114
        cur_x = x; cur_y = y; current_pixel_sample_index = 0;
        current_scramble_index = cantor_pairing((uint64_t)(int64_t)(x + HALF_MAX_I_32), (uint64_t)(int64_t)(y + HALF_MAX_I_32));
        dimension = 0;
        interval_sample_index = get_index_for_sample(0);
        array_end_dim = ARRAY_START_DIM;
    }
    bool start_next_sample() { // 122-127 + CoreSampler::start_next_sample mod.rs:76-81
        dimension = 0;
        interval_sample_index = get_index_for_sample((uint64_t)(current_pixel_sample_index + 1));
        current_pixel_sample_index += 1;
        return current_pixel_sample_index < samples_per_pixel;
    }
    float sample_dimension(int64_t index, size_t dim) const { // 177-193
        if (dim > 1024) throw std::runtime_error("sobol sampler can only sample up to 1024 dimensions.");
        float s = sobol_sample(*T, index, dim, current_scramble_index);
        if (dim == 0 || dim == 1) {
            int32_t pmin = dim == 0 ? sample_bounds.min_x : sample_bounds.min_y;
            int32_t cur = dim == 0 ? cur_x : cur_y;
            s = s * (float)resolution + (float)pmin;
            s = clamp_rs(s - (float)cur, 0.0f, ONE_MINUS_EPSILON);
        }
        return s;
    }
    float get_1d() { // 129-137
        if (dimension >= ARRAY_START_DIM && dimension < array_end_dim) dimension = array_end_dim;
        float s = sample_dimension(interval_sample_index, dimension);
        dimension += 1;
        return s;
    }
    Vec2 get_2d() { // 139-151
        if (dimension + 1 >= ARRAY_START_DIM && dimension < array_end_dim) dimension = array_end_dim;
        Vec2 s;
        s.x = sample_dimension(interval_sample_index, dimension);
        s.y = sample_dimension(interval_sample_index, dimension + 1);
        dimension += 2;
        return s;
    }
    Vec2 get_camera_sample(int32_t px, int32_t py) { // 116-120: p_raster as f32 + get_2d
        Vec2 u = get_2d();
        Vec2 p; p.x = (float)px + u.x; p.y = (float)py + u.y;
        return p;
    }
};

} // namespace orc
