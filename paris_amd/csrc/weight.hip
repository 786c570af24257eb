// Cosine pre-weighting for gfx950.
//
// Replaces paris::openmp::weight (src/openmp/weighting.cpp:32-57) / paris::cuda::weight
// (src/cuda/weighting.cu:35-73) behind paris_hip_weight. The CUDA backend uses rsqrtf; this kernel uses
// the OpenMP backend's IEEE sqrt + divide so the result is bit-identical to the parity target.
// HBM-bound: 8 B per pixel (one read, one write).
#include "paris_hip_internal.h"

namespace
{
    __global__ void __launch_bounds__(256)
        weight_kernel(float* p, uint32_t pitch_f, uint32_t dim_x, uint32_t row_first, uint32_t row_end, float h_min, float v_min,
                      float d_sd, float l_px_row, float l_px_col)
    {
        const uint32_t s = blockIdx.x * 256u + threadIdx.x;
        if(s >= dim_x)
            return;
        const float s_f = static_cast<float>(s);
        const float h_s = (l_px_row / 2) + s_f * l_px_row + h_min; // src/openmp/weighting.cpp:48
        const float hh = h_s * h_s;
        const float dd = d_sd * d_sd;
        for(uint32_t t = row_first + blockIdx.y; t < row_end; t += gridDim.y)
        {
            const float t_f = static_cast<float>(t);
            const float v_t = (l_px_col / 2) + t_f * l_px_col + v_min; // :49
            const float w_st = d_sd / sqrtf(dd + hh + v_t * v_t);      // :52
            float* px = p + static_cast<size_t>(t) * pitch_f + s;
            *px *= w_st; // :54
        }
    }
}

static int weight_rows_now(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t row_first, uint32_t row_count,
                           float h_min, float v_min, float d_sd, float l_px_row, float l_px_col)
{
    const dim3 grid((dim_x + 255u) / 256u, row_count < 65535u ? row_count : 65535u);
    hipLaunchKernelGGL(weight_kernel, grid, dim3(256), 0, ctx->stream, d_p, static_cast<uint32_t>(pitch / sizeof(float)),
                       dim_x, row_first, row_first + row_count, h_min, v_min, d_sd, l_px_row, l_px_col);
    return paris_hip_note_projection_use(ctx, reinterpret_cast<const char*>(d_p) + static_cast<size_t>(row_first) * pitch, pitch * row_count);
}

int paris_hip_flush_pending_weight(paris_hip_ctx* ctx)
{
    if(ctx == nullptr || !ctx->pending_weight.active)
        return PARIS_HIP_SUCCESS;
    const auto w = ctx->pending_weight;
    ctx->pending_weight.active = false;
    ctx->pending_weight.filter = false;
    PARIS_HIP_TRY(hipSetDevice(ctx->device));
    if(w.filter) // weighting and row filter were both held back (filter deferral): the one launch apply_filter would have made
    {
        float* rows = reinterpret_cast<float*>(reinterpret_cast<char*>(w.d_p) + static_cast<size_t>(w.row_first) * w.pitch);
        if(int rc = paris_hip_fused_filter_launch(ctx, rows, static_cast<uint32_t>(w.pitch / sizeof(float)), w.dim_x, w.row_count, w.row_first, true,
                                                  w.h_min, w.v_min, w.d_sd, w.l_px_row, w.l_px_col, w.d_kp, w.plan, w.filter_size, nullptr, 0u))
            return rc;
        return paris_hip_note_projection_use(ctx, rows, w.pitch * w.row_count);
    }
    if(int rc = weight_rows_now(ctx, w.d_p, w.pitch, w.dim_x, w.row_first, w.row_count, w.h_min, w.v_min, w.d_sd, w.l_px_row, w.l_px_col))
        return rc;
    return static_cast<int>(hipGetLastError());
}

extern "C" int paris_hip_weight_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                     uint32_t row_first, uint32_t row_count, float h_min, float v_min, float d_sd,
                                     float l_px_row, float l_px_col)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(int rc = paris_hip_flush_pending_weight(ctx)) // an earlier weighting nobody filtered
        return rc;
    if(d_p == nullptr || pitch < static_cast<size_t>(dim_x) * sizeof(float) || pitch % sizeof(float) != 0
       || row_first > dim_y || row_count > dim_y - row_first)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(dim_x == 0 || row_count == 0)
        return paris_hip_finish(ctx);
    // (deferral by reference: a buffer the pending group reads must not be weighted again before that group has run)
    if(int rc = paris_hip_projection_guard(ctx, d_p, pitch * dim_y, ctx->stream, true))
        return rc;
    if(ctx->stage_fusion != 0 && !(ctx->flags & PARIS_HIP_CTX_SYNCHRONOUS))
    {
        // held back: the filter call that follows weights in its load (one launch, 8 instead of 16 bytes of traffic per pixel)
        auto& w = ctx->pending_weight;
        w.active = true;
        w.filter = false;
        w.d_p = d_p;
        w.pitch = pitch;
        w.dim_x = dim_x;
        w.dim_y = dim_y;
        w.row_first = row_first;
        w.row_count = row_count;
        w.h_min = h_min;
        w.v_min = v_min;
        w.d_sd = d_sd;
        w.l_px_row = l_px_row;
        w.l_px_col = l_px_col;
        return PARIS_HIP_SUCCESS;
    }
    if(int rc = weight_rows_now(ctx, d_p, pitch, dim_x, row_first, row_count, h_min, v_min, d_sd, l_px_row, l_px_col))
        return rc;
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_weight(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                float h_min, float v_min, float d_sd, float l_px_row, float l_px_col)
{
    return paris_hip_weight_rows(ctx, d_p, pitch, dim_x, dim_y, 0u, dim_y, h_min, v_min, d_sd, l_px_row, l_px_col);
}

extern "C" int paris_hip_set_filter_deferral(paris_hip_ctx* ctx, int enable)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = paris_hip_flush_deferred(ctx)) // (runs a held-back weighting / filter as well)
        return rc;
    ctx->filter_deferral = enable == 2 ? 2 : (enable ? 1 : 0);
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_stage_fusion(paris_hip_ctx* ctx, int enable)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
    ctx->stage_fusion = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

// PARIS_HIP_CTX_WARM: a query of one kernel of this translation unit makes the runtime load its code object now
void paris_hip_warm_weight()
{
    hipFuncAttributes a{};
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&weight_kernel));
}
