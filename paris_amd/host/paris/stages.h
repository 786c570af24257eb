// Backend-neutral stage functions of the hot loop (src/main.cpp:100-104) for the HIP backend:
// load (src/loader.cpp:28-33), make_volume (src/make_volume.cpp:30-37), weight (src/weighting.cpp:32-45),
// filter (src/filtering.cpp:32-45), backproject (src/backprojection.cpp:37-69).
//
// The reference caches every derived constant in function-local statics (one geometry per process, SURVEY.md
// Q1/Q2); these versions derive them on every call through the paris_hip_stage_* entry points, so a process may
// reconstruct any number of geometries, slabs and ROIs.
#ifndef PARIS_AMD_HOST_STAGES_H_
#define PARIS_AMD_HOST_STAGES_H_

#include "hip/backend.h"

namespace paris
{
    namespace backend = hip;

    inline auto load(const backend::projection_host_type& p) -> backend::projection_device_type
    {
        auto d_p = backend::make_projection_device(p.dim_x, p.dim_y);
        backend::copy_h2d(p, d_p);
        return d_p;
    }

    inline auto make_volume(const subvolume_geometry& subvol_geo, bool last) -> backend::volume_device_type
    {
        // the last slab also takes the slices that did not divide evenly
        const auto dim_z = subvol_geo.dim_z + (last ? subvol_geo.remainder : 0u);
        return backend::make_volume_device(subvol_geo.dim_x, subvol_geo.dim_y, dim_z);
    }

    inline auto weight(backend::projection_device_type& p, const detector_geometry& det_geo) -> void
    {
        backend::detail::runtime_check(paris_hip_stage_weight(backend::current_ctx(), p.buf.get(), p.buf.pitch(), p.dim_x,
                                                              p.dim_y, &det_geo), "weight()");
    }

    inline auto filter(backend::projection_device_type& p, const detector_geometry& det_geo) -> void
    {
        backend::detail::runtime_check(paris_hip_stage_filter(backend::current_ctx(), p.buf.get(), p.buf.pitch(), p.dim_x,
                                                              p.dim_y, &det_geo), "filter()");
    }

    inline auto backproject(const backend::projection_device_type& p, backend::volume_device_type& v,
                            std::uint32_t v_offset, const detector_geometry& det_geo, const volume_geometry& vol_geo,
                            bool enable_angles, bool enable_roi, const region_of_interest& roi) -> void
    {
        backend::detail::runtime_check(
            paris_hip_stage_backproject(backend::current_ctx(), p.buf.get(), p.buf.pitch(), p.dim_x, p.dim_y, p.idx, p.phi,
                                        v.buf.get(), v.dim_x, v.dim_y, v.dim_z, v_offset, &det_geo, &vol_geo,
                                        enable_angles ? 1 : 0, enable_roi ? 1 : 0, &roi),
            "backproject()");
    }
}

#endif
