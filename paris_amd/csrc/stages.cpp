// Backend-neutral stage wrappers and geometry (host code), restated for the HIP backend.
//
// The reference derives every scalar the kernels consume in five small host files and caches them in
// function-local statics. They are part of the numeric contract, so they are restated here
// one rounding at a time, without the statics:
//   src/geometry.cpp:36-130     -> paris_hip_calculate_volume_geometry, paris_hip_apply_roi
//   src/weighting.cpp:32-45     -> paris_hip_stage_weight
//   src/filtering.cpp:32-45     -> paris_hip_filter_size, paris_hip_stage_filter
//   src/backprojection.cpp:37-69-> paris_hip_stage_backproject
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "paris_hip_internal.h"

extern "C" int paris_hip_calculate_volume_geometry(const paris_detector_geometry* det_geo, paris_volume_geometry* out)
{
    if(det_geo == nullptr || out == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    // src/geometry.cpp:36-67
    const float n_row = static_cast<float>(det_geo->n_row);
    const float l_px_row = det_geo->l_px_row;
    const float delta_s = std::abs(det_geo->delta_s * l_px_row); // offset is measured in pixels
    const float n_col = static_cast<float>(det_geo->n_col);
    const float l_px_col = det_geo->l_px_col;
    const float delta_t = std::abs(det_geo->delta_t * l_px_col);
    const float d_so = std::abs(det_geo->d_so);
    const float d_sd = std::abs(det_geo->d_od) + d_so;

    const float alpha = std::atan((((n_row * l_px_row) / 2.f) + delta_s) / d_sd); // :54
    const float r = d_so * std::sin(alpha);                                        // :55

    out->l_vx_x = r / ((((n_row * l_px_row) / 2.f) + delta_s) / l_px_row); // :57
    out->l_vx_y = out->l_vx_x;
    out->dim_x = static_cast<uint32_t>((2.f * r) / out->l_vx_x); // :60
    out->dim_y = out->dim_x;
    out->l_vx_z = out->l_vx_x;
    out->dim_z = static_cast<uint32_t>(((n_col * l_px_col / 2.f) + delta_t) * (d_so / d_sd) * (2.f / out->l_vx_z)); // :65
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_apply_roi(const paris_volume_geometry* vol_geo, const paris_region_of_interest* roi,
                                   paris_volume_geometry* out)
{
    if(vol_geo == nullptr || roi == nullptr || out == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    // src/geometry.cpp:86-130: an invalid or oversized ROI leaves the geometry unchanged (reference: warning)
    *out = *vol_geo;
    if(roi->x1 < roi->x2 && roi->y1 < roi->y2 && roi->z1 < roi->z2)
    {
        uint32_t dim_x = roi->x2 - roi->x1;
        uint32_t dim_y = roi->y2 - roi->y1;
        uint32_t dim_z = roi->z2 - roi->z1;
        if(roi->x1 == 0) ++dim_x; // :102-107
        if(roi->y1 == 0) ++dim_y;
        if(roi->z1 == 0) ++dim_z;
        if(dim_x <= vol_geo->dim_x && dim_y <= vol_geo->dim_y && dim_z <= vol_geo->dim_z)
        {
            out->dim_x = dim_x;
            out->dim_y = dim_y;
            out->dim_z = dim_z;
        }
    }
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_stage_weight(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                      const paris_detector_geometry* det_geo)
{
    if(det_geo == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    // src/weighting.cpp:37-42
    const float n_row_f = static_cast<float>(det_geo->n_row);
    const float n_col_f = static_cast<float>(det_geo->n_col);
    const float h_min = (det_geo->delta_s * det_geo->l_px_row) - ((n_row_f * det_geo->l_px_row) / 2);
    const float v_min = (det_geo->delta_t * det_geo->l_px_col) - ((n_col_f * det_geo->l_px_col) / 2);
    const float d_sd = std::abs(det_geo->d_so) + std::abs(det_geo->d_od);
    return paris_hip_weight(ctx, d_p, pitch, dim_x, dim_y, h_min, v_min, d_sd, det_geo->l_px_row, det_geo->l_px_col);
}

extern "C" int paris_hip_stage_weight_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                           uint32_t row_first, uint32_t row_count, const paris_detector_geometry* det_geo)
{
    if(det_geo == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const float n_row_f = static_cast<float>(det_geo->n_row);
    const float n_col_f = static_cast<float>(det_geo->n_col);
    const float h_min = (det_geo->delta_s * det_geo->l_px_row) - ((n_row_f * det_geo->l_px_row) / 2);
    const float v_min = (det_geo->delta_t * det_geo->l_px_col) - ((n_col_f * det_geo->l_px_col) / 2);
    const float d_sd = std::abs(det_geo->d_so) + std::abs(det_geo->d_od);
    return paris_hip_weight_rows(ctx, d_p, pitch, dim_x, dim_y, row_first, row_count, h_min, v_min, d_sd, det_geo->l_px_row,
                                 det_geo->l_px_col);
}

extern "C" uint32_t paris_hip_filter_size(uint32_t n_row)
{
    // src/filtering.cpp:37
    return static_cast<uint32_t>(2 * std::pow(2.f, std::ceil(std::log2(n_row))));
}

extern "C" int paris_hip_stage_filter_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                           uint32_t row_first, uint32_t row_count, const paris_detector_geometry* det_geo)
{
    if(ctx == nullptr || det_geo == nullptr || d_p == nullptr || row_first > dim_y || row_count > dim_y - row_first)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    // src/filtering.cpp:37-44; K is built once per ctx and (filter_size, tau), where the reference keeps a
    // thread_local static (:42)
    const uint32_t filter_size = paris_hip_filter_size(det_geo->n_row);
    const float tau = det_geo->l_px_row;
    if(ctx->stage_k == nullptr || ctx->stage_k_size != filter_size || ctx->stage_k_tau != tau || ctx->stage_k_window != ctx->stage_window)
    {
        if(ctx->stage_k != nullptr)
        {
            if(int rc = paris_hip_free(ctx, ctx->stage_k))
                return rc;
            ctx->stage_k = nullptr;
        }
        if(int rc = paris_hip_make_filter_windowed(ctx, filter_size, tau, ctx->stage_window, &ctx->stage_k))
            return rc;
        ctx->stage_k_size = filter_size;
        ctx->stage_k_tau = tau;
        ctx->stage_k_window = ctx->stage_window;
        // once per detector, here rather than inside the first backprojection: the exhaustive check of the fast division by
        // the pixel pitches blocks for ~2 ms on a private stream; paris_hip_backproject then finds it cached and stays asynchronous
        if(int rc = paris_hip_prevalidate_fast_division(ctx, det_geo->l_px_row, det_geo->l_px_col))
            return rc;
        if(int rc = paris_hip_prevalidate_lean_division(ctx, det_geo->d_so, det_geo->d_od))
            return rc;
    }
    float* rows = reinterpret_cast<float*>(reinterpret_cast<char*>(d_p) + static_cast<size_t>(row_first) * pitch);
    return paris_hip_apply_filter(ctx, rows, pitch, dim_x, row_count, ctx->stage_k, filter_size, row_count);
}

// Extension: paris::weight + paris::filter of a row band in ONE launch (paris_hip_weight_filter_rows), optionally storing the
// result as IEEE half into d_half (BASELINE config 5) instead of fp32 in place. Filter lengths below 1024 (detectors narrower
// than 257 pixels) run the two stages (and the conversion) one after the other: same result.
extern "C" int paris_hip_stage_weight_filter_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                                  uint32_t row_first, uint32_t row_count, const paris_detector_geometry* det_geo,
                                                  uint16_t* d_half, size_t half_pitch)
{
    if(ctx == nullptr || det_geo == nullptr || d_p == nullptr || row_first > dim_y || row_count > dim_y - row_first)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const uint32_t filter_size = paris_hip_filter_size(det_geo->n_row);
    if(filter_size >= 1024u && ctx->filter_variant == 0)
    {
        // K of the stage wrapper, built once per ctx and (filter_size, tau): the filter call below with no rows does just that
        if(int rc = paris_hip_stage_filter_rows(ctx, d_p, pitch, dim_x, dim_y, row_first, 0u, det_geo))
            return rc;
        const float n_row_f = static_cast<float>(det_geo->n_row);
        const float n_col_f = static_cast<float>(det_geo->n_col);
        const float h_min = (det_geo->delta_s * det_geo->l_px_row) - ((n_row_f * det_geo->l_px_row) / 2); // src/weighting.cpp:37-42
        const float v_min = (det_geo->delta_t * det_geo->l_px_col) - ((n_col_f * det_geo->l_px_col) / 2);
        const float d_sd = std::abs(det_geo->d_so) + std::abs(det_geo->d_od);
        return paris_hip_weight_filter_rows(ctx, d_p, pitch, dim_x, dim_y, row_first, row_count, h_min, v_min, d_sd, det_geo->l_px_row,
                                            det_geo->l_px_col, ctx->stage_k, filter_size, d_half, half_pitch);
    }
    if(int rc = paris_hip_stage_weight_rows(ctx, d_p, pitch, dim_x, dim_y, row_first, row_count, det_geo))
        return rc;
    if(int rc = paris_hip_stage_filter_rows(ctx, d_p, pitch, dim_x, dim_y, row_first, row_count, det_geo))
        return rc;
    if(d_half == nullptr || row_count == 0)
        return PARIS_HIP_SUCCESS;
    return paris_hip_convert_projection_f16(ctx, reinterpret_cast<float*>(reinterpret_cast<char*>(d_p) + static_cast<size_t>(row_first) * pitch), pitch,
                                            reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(d_half) + static_cast<size_t>(row_first) * half_pitch),
                                            half_pitch, dim_x, row_count);
}

// Extension: paris_hip_stage_weight_filter_rows for a group of n_frames projections frame_stride bytes apart, in one launch
// (paris_hip_weight_filter_batch); narrow detectors (filter length < 1024) run frame by frame: same result.
extern "C" int paris_hip_stage_weight_filter_batch(paris_hip_ctx* ctx, float* d_p, size_t pitch, size_t frame_stride, uint32_t n_frames,
                                                   uint32_t dim_x, uint32_t dim_y, uint32_t row_first, uint32_t row_count,
                                                   const paris_detector_geometry* det_geo, uint16_t* d_half, size_t half_pitch,
                                                   size_t half_frame_stride)
{
    if(ctx == nullptr || det_geo == nullptr || d_p == nullptr || row_first > dim_y || row_count > dim_y - row_first)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const uint32_t filter_size = paris_hip_filter_size(det_geo->n_row);
    if(filter_size >= 1024u && ctx->filter_variant == 0)
    {
        if(int rc = paris_hip_stage_filter_rows(ctx, d_p, pitch, dim_x, dim_y, row_first, 0u, det_geo)) // builds / finds the cached K
            return rc;
        const float n_row_f = static_cast<float>(det_geo->n_row);
        const float n_col_f = static_cast<float>(det_geo->n_col);
        const float h_min = (det_geo->delta_s * det_geo->l_px_row) - ((n_row_f * det_geo->l_px_row) / 2); // src/weighting.cpp:37-42
        const float v_min = (det_geo->delta_t * det_geo->l_px_col) - ((n_col_f * det_geo->l_px_col) / 2);
        const float d_sd = std::abs(det_geo->d_so) + std::abs(det_geo->d_od);
        return paris_hip_weight_filter_batch(ctx, d_p, pitch, frame_stride, n_frames, dim_x, dim_y, row_first, row_count, h_min, v_min, d_sd,
                                             det_geo->l_px_row, det_geo->l_px_col, ctx->stage_k, filter_size, d_half, half_pitch, half_frame_stride);
    }
    for(uint32_t f = 0; f < n_frames; ++f)
        if(int rc = paris_hip_stage_weight_filter_rows(ctx, reinterpret_cast<float*>(reinterpret_cast<char*>(d_p) + f * frame_stride), pitch, dim_x, dim_y,
                                                       row_first, row_count, det_geo,
                                                       d_half ? reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(d_half) + f * half_frame_stride) : nullptr,
                                                       half_pitch))
            return rc;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_stage_filter(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                      const paris_detector_geometry* det_geo)
{
    // n_col is what the reference passes (src/filtering.cpp:44); apply_filter requires it to equal the row count
    if(det_geo != nullptr && det_geo->n_col != dim_y)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    return paris_hip_stage_filter_rows(ctx, d_p, pitch, dim_x, dim_y, 0u, dim_y, det_geo);
}

extern "C" int paris_hip_stage_angle(const paris_detector_geometry* det_geo, uint32_t idx, int enable_angles, float phi,
                                     float* sin_phi, float* cos_phi)
{
    if(det_geo == nullptr || sin_phi == nullptr || cos_phi == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    // src/backprojection.cpp:52-63
    float a = 0.f;
    if(enable_angles)
        a = phi;
    else
        a = static_cast<float>(idx) * det_geo->delta_phi;
    a *= static_cast<float>(M_PI) / 180.f;
    *sin_phi = std::sin(a);
    *cos_phi = std::cos(a);
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_stage_backproject(paris_hip_ctx* ctx, const float* d_p, size_t p_pitch, uint32_t p_dim_x,
                                           uint32_t p_dim_y, uint32_t p_idx, float p_phi, float* d_v,
                                           uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                                           const paris_detector_geometry* det_geo,
                                           const paris_volume_geometry* vol_geo, int enable_angles, int enable_roi,
                                           const paris_region_of_interest* roi)
{
    if(det_geo == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    // src/backprojection.cpp:49-50
    const float delta_s = det_geo->delta_s * det_geo->l_px_row;
    const float delta_t = det_geo->delta_t * det_geo->l_px_col;
    float sin_phi = 0.f, cos_phi = 0.f;
    if(int rc = paris_hip_stage_angle(det_geo, p_idx, enable_angles, p_phi, &sin_phi, &cos_phi))
        return rc;
    return paris_hip_backproject(ctx, d_p, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo,
                                 vol_geo, enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t);
}

// f4 (no reference counterpart): the detector rows a slab can read.
// With the coordinates of src/openmp/backprojection.cpp:116-133, v = (z*f - v_min)/l_px_col - 1/2 where
// f = d_sd/(s + d_so) and |s| <= S = hypot(max|x|, max|y|) over the slab's columns for every angle. The extremes of
// z*f over z in [z_lo, z_hi] and f in [f(d_so+S), f(d_so-S)] bound v; a tap reads rows floor(v) and floor(v)+1.
// Evaluated in double with a slack of two rows against the kernel's fp32 rounding.
extern "C" int paris_hip_slab_row_band(const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo,
                                       uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                                       int enable_roi, const paris_region_of_interest* roi, uint32_t* row_first,
                                       uint32_t* row_count)
{
    if(det_geo == nullptr || vol_geo == nullptr || row_first == nullptr || row_count == nullptr || (enable_roi && roi == nullptr))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const uint32_t n_col = det_geo->n_col;
    *row_first = 0;
    *row_count = n_col; // whenever the bound below does not hold: the whole detector
    if(v_dim_x == 0 || v_dim_y == 0 || v_dim_z == 0 || n_col == 0)
    {
        *row_count = 0;
        return PARIS_HIP_SUCCESS;
    }
    const auto centred = [](double coord, double dim, double size) { return -(dim * size / 2.0) + size / 2.0 + coord * size; };
    const double rx = enable_roi ? roi->x1 : 0u, ry = enable_roi ? roi->y1 : 0u, rz = enable_roi ? roi->z1 : 0u;
    const double lx = vol_geo->l_vx_x, ly = vol_geo->l_vx_y, lz = vol_geo->l_vx_z;
    const double x_a = centred(rx, vol_geo->dim_x, lx), x_b = centred(rx + (v_dim_x - 1.0), vol_geo->dim_x, lx);
    const double y_a = centred(ry, vol_geo->dim_y, ly), y_b = centred(ry + (v_dim_y - 1.0), vol_geo->dim_y, ly);
    const double z_a = centred(rz + v_offset, vol_geo->dim_z, lz);
    const double z_b = centred(rz + v_offset + (v_dim_z - 1.0), vol_geo->dim_z, lz);
    const double l_c = det_geo->l_px_col;
    const double d_so = det_geo->d_so; // raw and signed, as the backprojection uses it
    const double d_sd = std::abs(static_cast<double>(det_geo->d_so)) + std::abs(static_cast<double>(det_geo->d_od));
    const double x_max = std::max(std::abs(x_a), std::abs(x_b)), y_max = std::max(std::abs(y_a), std::abs(y_b));
    const double S = std::hypot(x_max, y_max) * (1.0 + 1e-6) + 1e-3 * std::max(std::abs(lx), std::abs(ly));
    const double den_lo = d_so - S, den_hi = d_so + S;
    if(!(std::isfinite(S) && std::isfinite(d_sd) && std::isfinite(z_a) && std::isfinite(z_b) && std::isfinite(l_c)) || l_c == 0.0
       || !(den_lo > 0.0 || den_hi < 0.0)) // the source distance may vanish inside the slab's circle: no bound
        return PARIS_HIP_SUCCESS;
    const double f_1 = d_sd / den_lo, f_2 = d_sd / den_hi;
    const double v_min_mm = -(static_cast<double>(n_col) * l_c / 2.0) - static_cast<double>(det_geo->delta_t) * l_c; // :45-50
    double lo = 0.0, hi = 0.0;
    bool first = true;
    for(const double z : {z_a, z_b})
        for(const double f : {f_1, f_2})
        {
            const double v = (z * f - v_min_mm) / l_c - 0.5;
            lo = first ? v : std::min(lo, v);
            hi = first ? v : std::max(hi, v);
            first = false;
        }
    if(!(std::isfinite(lo) && std::isfinite(hi)))
        return PARIS_HIP_SUCCESS;
    const double r_lo = std::floor(lo) - 2.0, r_hi = std::floor(hi) + 1.0 + 2.0; // inclusive rows
    if(r_hi < 0.0 || r_lo > static_cast<double>(n_col) - 1.0)
    {
        *row_count = 0; // the slab projects outside the detector for every angle
        return PARIS_HIP_SUCCESS;
    }
    uint32_t a = r_lo < 0.0 ? 0u : static_cast<uint32_t>(r_lo);
    uint32_t b = r_hi > static_cast<double>(n_col) - 1.0 ? n_col - 1u : static_cast<uint32_t>(r_hi);
    // the filter transforms rows in pairs (2t, 2t+1) as one complex signal: whole pairs keep the band's rounding
    // identical to a full-projection filter
    a &= ~1u;
    b = std::min(b | 1u, n_col - 1u);
    *row_first = a;
    *row_count = b - a + 1u;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_filter_window(paris_hip_ctx* ctx, int window)
{
    if(ctx == nullptr || (window != PARIS_HIP_WINDOW_RAMP && window != PARIS_HIP_WINDOW_SHEPP_LOGAN))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->stage_window = window; // the cached K is rebuilt by the next paris_hip_stage_filter
    return PARIS_HIP_SUCCESS;
}
