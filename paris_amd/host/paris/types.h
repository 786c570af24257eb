// Value types of the PARIS backend surface, expressed over the C ABI structs of include/paris_hip.h.
//
// The reference declares these in five headers (src/geometry.h:30-69, src/projection.h:31-46, src/volume.h:31-45,
// src/region_of_interest.h:30-38, src/subvolume_information.h:30-34). The geometry / ROI / subvolume blocks are
// plain aliases of the C structs, so they are field-for-field identical by construction; projection and volume
// are the owning wrappers a backend instantiates with its buffer types.
#ifndef PARIS_AMD_HOST_TYPES_H_
#define PARIS_AMD_HOST_TYPES_H_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>

#include "paris_hip.h"

namespace paris
{
    using detector_geometry = ::paris_detector_geometry;
    using volume_geometry = ::paris_volume_geometry;
    using subvolume_geometry = ::paris_subvolume_geometry;
    using region_of_interest = ::paris_region_of_interest;
    using subvolume_info = ::paris_subvolume_info;

    // one detector frame: dim_x = pixels per row (fastest), dim_y = rows; idx / phi identify the view
    template <typename Buffer, typename Meta>
    struct projection
    {
        Buffer buf{};
        std::uint32_t dim_x = 0, dim_y = 0;
        std::uint32_t idx = 0;
        float phi = 0.f;
        Meta meta{};

        projection() noexcept = default;
        projection(Buffer b, std::uint32_t x, std::uint32_t y, std::uint32_t i, float ph, Meta m) noexcept
        : buf(std::move(b)), dim_x(x), dim_y(y), idx(i), phi(ph), meta(std::move(m)) {}
    };

    // a (sub)volume, x fastest; off = first global slice (see SURVEY.md Q4 for how the reference misuses it)
    template <typename Buffer>
    struct volume
    {
        Buffer buf{};
        std::uint32_t dim_x = 0, dim_y = 0, dim_z = 0;
        std::uint32_t off = 0;

        volume() noexcept = default;
        volume(Buffer b, std::uint32_t x, std::uint32_t y, std::uint32_t z, std::uint32_t o) noexcept
        : buf(std::move(b)), dim_x(x), dim_y(y), dim_z(z), off(o) {}
    };

    // src/exception.h:31-41
    struct stage_construction_error : std::runtime_error { using std::runtime_error::runtime_error; };
    struct stage_runtime_error : std::runtime_error { using std::runtime_error::runtime_error; };

    // src/geometry.cpp:71-130
    inline auto calculate_volume_geometry(const detector_geometry& det_geo) -> volume_geometry
    {
        volume_geometry v{};
        if(int rc = paris_hip_calculate_volume_geometry(&det_geo, &v))
            throw stage_construction_error{std::string{"calculate_volume_geometry(): "} + paris_hip_strerror(rc)};
        return v;
    }

    inline auto apply_roi(const volume_geometry& vol_geo, std::uint32_t x1, std::uint32_t x2, std::uint32_t y1,
                          std::uint32_t y2, std::uint32_t z1, std::uint32_t z2) -> volume_geometry
    {
        const region_of_interest roi{x1, x2, y1, y2, z1, z2};
        volume_geometry v{};
        if(int rc = paris_hip_apply_roi(&vol_geo, &roi, &v))
            throw stage_construction_error{std::string{"apply_roi(): "} + paris_hip_strerror(rc)};
        return v;
    }
}

#endif
