// Backend-neutral stage wrappers and geometry (host code), restated for the HIP backend.
//
// The reference derives every scalar the kernels consume in five small host files and caches them in
// function-local statics. They are part of the numeric contract, so they are restated here
// one rounding at a time, without the statics:
//   src/geometry.cpp:36-130     -> paris_hip_calculate_volume_geometry, paris_hip_apply_roi
//   src/weighting.cpp:32-45     -> paris_hip_stage_weight
//   src/filtering.cpp:32-45     -> paris_hip_filter_size, paris_hip_stage_filter
//   src/backprojection.cpp:37-69-> paris_hip_stage_backproject
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

extern "C" uint32_t paris_hip_filter_size(uint32_t n_row)
{
    // src/filtering.cpp:37
    return static_cast<uint32_t>(2 * std::pow(2.f, std::ceil(std::log2(n_row))));
}

extern "C" int paris_hip_stage_filter(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                      const paris_detector_geometry* det_geo)
{
    if(ctx == nullptr || det_geo == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    // src/filtering.cpp:37-44; K is built once per ctx and (filter_size, tau), where the reference keeps a
    // thread_local static (:42)
    const uint32_t filter_size = paris_hip_filter_size(det_geo->n_row);
    const float tau = det_geo->l_px_row;
    if(ctx->stage_k == nullptr || ctx->stage_k_size != filter_size || ctx->stage_k_tau != tau)
    {
        if(ctx->stage_k != nullptr)
        {
            if(int rc = paris_hip_free(ctx, ctx->stage_k))
                return rc;
            ctx->stage_k = nullptr;
        }
        if(int rc = paris_hip_make_filter(ctx, filter_size, tau, &ctx->stage_k))
            return rc;
        ctx->stage_k_size = filter_size;
        ctx->stage_k_tau = tau;
    }
    return paris_hip_apply_filter(ctx, d_p, pitch, dim_x, dim_y, ctx->stage_k, filter_size, det_geo->n_col);
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
