// namespace paris::hip -- the MI355X backend behind the PARIS backend surface.
//
// Same types and the same 14 free functions as src/openmp/backend.h:42-89 / src/cuda/backend.h:50-103 (cleanest
// statement: src/generic/backend.h:54-88), implemented over the C ABI of include/paris_hip.h. A maintainer adds
//     #elif defined(PARIS_ENABLE_HIP)
//     #include "hip/backend.h"          // + `namespace backend = hip;` in the alias block
// to src/backend.h:26-47.
//
// Per-device state: the reference binds a host thread to a device with set_device() and keeps everything else
// in thread_local statics. Here set_device() creates (once per thread and device) the paris_hip_ctx that holds
// that state; every other function uses the calling thread's current ctx. Calls enqueue on the ctx's stream and only
// the copies to the host wait (PARIS_HIP_SYNCHRONOUS_CALLS=1 restores the reference's sync-per-call). A non-zero C status becomes stage_runtime_error (stage_construction_error for allocation and
// subvolume planning), after which the reference's main() aborts (src/main.cpp:181-192).
#ifndef PARIS_AMD_HOST_HIP_BACKEND_H_
#define PARIS_AMD_HOST_HIP_BACKEND_H_

#ifndef PARIS_HIP_BACKPROJECT_DEFERRAL
#define PARIS_HIP_BACKPROJECT_DEFERRAL 48
#endif
// 1 (default): weight() is held back and rides along in the load of the apply_filter() call that follows (one launch per
// weight / filter pair of src/main.cpp:102-103; paris_hip_set_stage_fusion). 0: one launch per call.
#ifndef PARIS_HIP_STAGE_FUSION
#define PARIS_HIP_STAGE_FUSION 1
#endif
// 2 (default): the apply_filter() of that pair is held back too when nobody can tell -- the projection is a make_projection_device
// buffer that a deferred backproject() takes by reference (PARIS_HIP_BACKPROJECT_REFERENCES): weighting + filter then run IN PLACE, one
// launch for the whole group right before its fused backprojection, and every other call that touches the buffer runs them first
// (paris_hip_set_filter_deferral(2)). 1: held back whenever a backprojection follows (a snapshotted projection's own buffer then
// keeps its unfiltered pixels). 0: the filter runs when it is called. Round 4 left this off: the ring's snapshots made it visible
// in the caller's buffer, and the loop was bound elsewhere; with references it is invisible and removes a launch per projection
// from the host's path (profiles/r05_demo_paris_hip_mirror.txt).
#ifndef PARIS_HIP_FILTER_DEFERRAL
#define PARIS_HIP_FILTER_DEFERRAL 2
#endif
// 1: every backend call returns after its work has finished, like the reference's backends (stream sync before return:
// src/cuda/weighting.cu:72, filtering.cu:260, backprojection.cu:236). 0 (default): calls enqueue and return; only the
// copies to the host wait. The call sequence of src/main.cpp:98-105 observes results through copy_d2h alone, and the
// library hands a released host / device projection buffer out again only after the work that used it has finished, so
// the loop's allocate / fill / load / free per projection stays correct while host and GPU work overlap.
#ifndef PARIS_HIP_SYNCHRONOUS_CALLS
#define PARIS_HIP_SYNCHRONOUS_CALLS 0
#endif
// 1 (default): a full group's fused backprojection runs on the ctx's second stream and copy_h2d(projection) on its upload stream, so
// that the next group's uploads, filters and snapshots -- which PARIS's loop keeps issuing, src/main.cpp:98-105 -- pass the running
// launch instead of queueing behind it (paris_hip_set_backproject_overlap, paris_hip_upload_projection). The library orders every
// observer of the volume behind the launches; results are bit-identical. 0: one stream for everything.
#ifndef PARIS_HIP_BACKPROJECT_OVERLAP
#define PARIS_HIP_BACKPROJECT_OVERLAP 1
#endif

// 1 (default): a deferred backproject() of a make_projection_device buffer takes no snapshot -- the group's fused launch reads the
// buffer itself, the buffer PARIS's loop frees right after the call (src/main.cpp:98-105, src/loader.cpp:28-33) goes back to the
// pool behind that launch, and anything else that touches it first launches the group (paris_hip_set_backproject_references). 0:
// every call snapshots its projection into the library's ring.
#ifndef PARIS_HIP_BACKPROJECT_REFERENCES
#define PARIS_HIP_BACKPROJECT_REFERENCES 1
#endif

// Which stream copy_h2d(projection) runs on. 1: the ctx's upload stream (paris_hip_upload_projection; the compute stream is ordered
// behind every copy by an event pair, ~2.3 us of runtime calls per frame): what the loop needs when kernels of its own -- a filter
// per projection, snapshot copies -- share the compute stream with the copies. 0: the compute stream itself. Default: 0 when the
// projections are taken by reference and filtered in place a group at a time (PARIS_HIP_BACKPROJECT_REFERENCES with
// PARIS_HIP_FILTER_DEFERRAL 2, which needs PARIS_HIP_STAGE_FUSION): the loop then enqueues NOTHING but its copies on the compute stream -- the group's filter and fused
// launch run on the second stream, ordered behind the copies by the one event each launch records anyway -- so a stream of their own
// buys the copies nothing and costs the event pair; 1 otherwise.
#ifndef PARIS_HIP_UPLOAD_ON_ITS_OWN_STREAM
#define PARIS_HIP_UPLOAD_ON_ITS_OWN_STREAM                                                                                        \
    (PARIS_HIP_BACKPROJECT_OVERLAP                                                                                               \
     && !(PARIS_HIP_BACKPROJECT_REFERENCES && PARIS_HIP_FILTER_DEFERRAL == 2 && PARIS_HIP_STAGE_FUSION && PARIS_HIP_BACKPROJECT_DEFERRAL > 1))
#endif

#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#if defined(PARIS_HIP_INSIDE_PARIS)
// Installed as src/hip/backend.h of a PARIS checkout: PARIS's own value types are used and handed to the C ABI by
// pointer (they are field-for-field the C structs: checked below).
#include "../exception.h"
#include "../geometry.h"
#include "../projection.h"
#include "../region_of_interest.h"
#include "../subvolume_information.h"
#include "../volume.h"
#include "paris_hip.h"
#else
#include "../types.h"
#endif

namespace paris
{
    namespace hip
    {
        using device_handle = int;

        namespace detail
        {
            // PARIS value types -> C ABI structs. With ../types.h they are the same types; inside PARIS they are
            // layout-identical PODs.
            template <typename To, typename From>
            inline auto as_c(const From& v) noexcept -> const To*
            {
                static_assert(sizeof(To) == sizeof(From) && alignof(To) == alignof(From), "PARIS type and C ABI struct differ");
                return reinterpret_cast<const To*>(&v);
            }
            static_assert(offsetof(detector_geometry, delta_phi) == offsetof(paris_detector_geometry, delta_phi), "detector_geometry");
            static_assert(offsetof(volume_geometry, l_vx_z) == offsetof(paris_volume_geometry, l_vx_z), "volume_geometry");
            static_assert(offsetof(region_of_interest, z2) == offsetof(paris_region_of_interest, z2), "region_of_interest");
            static_assert(offsetof(subvolume_info, num) == offsetof(paris_subvolume_info, num), "subvolume_info");

            struct ctx_deleter { void operator()(paris_hip_ctx* c) const noexcept { paris_hip_ctx_destroy(c); } };

            struct thread_state
            {
                std::map<device_handle, std::unique_ptr<paris_hip_ctx, ctx_deleter>> per_device;
                paris_hip_ctx* current = nullptr;
            };

            inline auto state() -> thread_state&
            {
                thread_local thread_state s;
                return s;
            }

            inline void runtime_check(int rc, const char* what)
            {
                if(rc != PARIS_HIP_SUCCESS)
                    throw stage_runtime_error{std::string{what} + " failed: " + paris_hip_strerror(rc)};
            }

            inline void construction_check(int rc, const char* what)
            {
                if(rc != PARIS_HIP_SUCCESS)
                    throw stage_construction_error{std::string{what} + " failed: " + paris_hip_strerror(rc)};
            }
        }

        // ---- device management (src/cuda/device.cpp:31-47) --------------------------------------------------
        inline auto get_devices() -> std::vector<device_handle>
        {
            int n = 0;
            detail::construction_check(paris_hip_device_count(&n), "get_devices()");
            auto v = std::vector<device_handle>{};
            for(int d = 0; d < n; ++d)
                v.push_back(d);
            return v;
        }

        inline auto set_device(device_handle& d) -> int
        {
            auto& s = detail::state();
            auto it = s.per_device.find(d);
            if(it == s.per_device.end())
            {
                paris_hip_ctx* c = nullptr;
                detail::construction_check(paris_hip_ctx_create(d, nullptr, (PARIS_HIP_SYNCHRONOUS_CALLS ? PARIS_HIP_CTX_SYNCHRONOUS : PARIS_HIP_CTX_DEFAULT) | PARIS_HIP_CTX_WARM, &c),
                                           "set_device()");
                // PARIS calls backproject() once per projection (src/main.cpp:98-105). The library snapshots each call's
                // projection and adds PARIS_HIP_BACKPROJECT_DEFERRAL of them with one fused launch -- bit-identical, the
                // slab is read and written once per group instead of once per projection -- and flushes before anything
                // observes the volume (copy_d2h, free): see paris_hip_set_backproject_deferral. 1 = one launch per call.
                detail::construction_check(paris_hip_set_backproject_deferral(c, PARIS_HIP_BACKPROJECT_DEFERRAL), "set_device()");
                detail::construction_check(paris_hip_set_stage_fusion(c, PARIS_HIP_STAGE_FUSION), "set_device()");
                detail::construction_check(paris_hip_set_filter_deferral(c, PARIS_HIP_FILTER_DEFERRAL), "set_device()");
                detail::construction_check(paris_hip_set_backproject_overlap(c, PARIS_HIP_BACKPROJECT_OVERLAP), "set_device()");
                detail::construction_check(paris_hip_set_backproject_references(c, PARIS_HIP_BACKPROJECT_REFERENCES && !PARIS_HIP_SYNCHRONOUS_CALLS),
                                           "set_device()");
                // validators of the hand-expanded IEEE sequences run beside the first projections instead of in front of them
                detail::construction_check(paris_hip_set_async_validation(c, !PARIS_HIP_SYNCHRONOUS_CALLS), "set_device()");
                it = s.per_device.emplace(d, std::unique_ptr<paris_hip_ctx, detail::ctx_deleter>{c}).first;
            }
            s.current = it->second.get();
            return 0;
        }

        inline auto current_ctx() -> paris_hip_ctx*
        {
            auto& s = detail::state();
            if(s.current == nullptr)
            {
                auto d = device_handle{0}; // the reference's single-device path never calls set_device explicitly
                set_device(d);
            }
            return s.current;
        }

        // ---- buffers ------------------------------------------------------------------------------------------
        struct device_free { void operator()(float* p) const noexcept { paris_hip_free(current_ctx(), p); } };
        struct host_free { void operator()(float* p) const noexcept { paris_hip_free_host(current_ctx(), p); } };

        // device projection buffer with a row pitch in bytes, like cuda::pitched_device_ptr
        class pitched_buffer
        {
        public:
            pitched_buffer() noexcept = default;
            pitched_buffer(float* p, std::size_t pitch_bytes) noexcept : ptr_{p}, pitch_{pitch_bytes} {}
            auto get() const noexcept -> float* { return ptr_.get(); }
            auto pitch() const noexcept -> std::size_t { return pitch_; }
            explicit operator bool() const noexcept { return static_cast<bool>(ptr_); }
        private:
            std::unique_ptr<float, device_free> ptr_{};
            std::size_t pitch_ = 0;
        };

        using projection_host_buffer_type = std::unique_ptr<float, host_free>;
        using projection_device_buffer_type = pitched_buffer;
        using volume_host_buffer_type = std::unique_ptr<float, host_free>;
        using volume_device_buffer_type = std::unique_ptr<float, device_free>;
        using filter_buffer_type = std::unique_ptr<float, device_free>;

        struct metadata {};

        using projection_host_type = projection<projection_host_buffer_type, metadata>;
        using projection_device_type = projection<projection_device_buffer_type, metadata>;
        using volume_host_type = volume<volume_host_buffer_type>;
        using volume_device_type = volume<volume_device_buffer_type>;

        // ---- memory (src/cuda/memory.cpp:33-102) ------------------------------------------------------------------
        inline auto make_projection_host(std::uint32_t dim_x, std::uint32_t dim_y) -> projection_host_type
        {
            void* p = nullptr;
            detail::construction_check(paris_hip_malloc_host(current_ctx(), std::size_t{dim_x} * dim_y * sizeof(float), &p),
                                       "make_projection_host()");
            return {projection_host_buffer_type{static_cast<float*>(p)}, dim_x, dim_y, 0, 0.f, metadata{}};
        }

        inline auto make_projection_device(std::uint32_t dim_x, std::uint32_t dim_y) -> projection_device_type
        {
            float* p = nullptr;
            std::size_t pitch = 0;
            detail::construction_check(paris_hip_malloc_projection(current_ctx(), dim_x, dim_y, &p, &pitch),
                                       "make_projection_device()");
            return {pitched_buffer{p, pitch}, dim_x, dim_y, 0, 0.f, metadata{}};
        }

        inline auto make_volume_host(std::uint32_t dim_x, std::uint32_t dim_y, std::uint32_t dim_z) -> volume_host_type
        {
            const auto n = std::size_t{dim_x} * dim_y * dim_z;
            void* p = nullptr;
            detail::construction_check(paris_hip_malloc_host(current_ctx(), n * sizeof(float), &p), "make_volume_host()");
            auto f = static_cast<float*>(p);
            for(std::size_t i = 0; i < n; ++i)
                f[i] = 0.f; // zero-filled like src/openmp/memory.cpp:46-47
            return {volume_host_buffer_type{f}, dim_x, dim_y, dim_z, 0};
        }

        inline auto make_volume_device(std::uint32_t dim_x, std::uint32_t dim_y, std::uint32_t dim_z) -> volume_device_type
        {
            float* p = nullptr;
            detail::construction_check(paris_hip_malloc_volume(current_ctx(), dim_x, dim_y, dim_z, &p), "make_volume_device()");
            return {volume_device_buffer_type{p}, dim_x, dim_y, dim_z, 0};
        }

        inline auto copy_h2d(const projection_host_type& h_p, projection_device_type& d_p) -> void
        {
            // on the upload stream when the loop's own kernels share the compute stream: the transfer does not wait behind them, and
            // the pinned source is released when this copy is done (the compute stream is ordered behind the copy by the library);
            // on the compute stream when nothing else is enqueued there (PARIS_HIP_UPLOAD_ON_ITS_OWN_STREAM above)
            detail::runtime_check((PARIS_HIP_UPLOAD_ON_ITS_OWN_STREAM && !PARIS_HIP_SYNCHRONOUS_CALLS ? paris_hip_upload_projection : paris_hip_memcpy_projection_h2d)(
                                      current_ctx(), d_p.buf.get(), d_p.buf.pitch(), h_p.buf.get(), std::size_t{h_p.dim_x} * sizeof(float), h_p.dim_x, h_p.dim_y),
                                  "copy_h2d(projection)");
            d_p.idx = h_p.idx; // src/openmp/memory.cpp:60-62
            d_p.phi = h_p.phi;
            d_p.meta = h_p.meta;
        }

        inline auto copy_d2h(const projection_device_type& d_p, projection_host_type& h_p) -> void
        {
            detail::runtime_check(paris_hip_memcpy_projection_d2h(current_ctx(), h_p.buf.get(), std::size_t{h_p.dim_x} * sizeof(float),
                                                                  d_p.buf.get(), d_p.buf.pitch(), d_p.dim_x, d_p.dim_y),
                                  "copy_d2h(projection)");
            detail::runtime_check(paris_hip_ctx_synchronize(current_ctx()), "copy_d2h(projection)"); // the host reads h_p next
            h_p.idx = d_p.idx;
            h_p.phi = d_p.phi;
            h_p.meta = d_p.meta;
        }

        inline auto copy_h2d(const volume_host_type& h_v, volume_device_type& d_v) -> void
        {
            detail::runtime_check(paris_hip_memcpy_volume_h2d(current_ctx(), d_v.buf.get(), h_v.buf.get(), h_v.dim_x, h_v.dim_y, h_v.dim_z),
                                  "copy_h2d(volume)");
            d_v.off = h_v.off; // src/openmp/memory.cpp:73
        }

        inline auto copy_d2h(const volume_device_type& d_v, volume_host_type& h_v) -> void
        {
            detail::runtime_check(paris_hip_memcpy_volume_d2h(current_ctx(), h_v.buf.get(), d_v.buf.get(), d_v.dim_x, d_v.dim_y, d_v.dim_z),
                                  "copy_d2h(volume)");
            detail::runtime_check(paris_hip_ctx_synchronize(current_ctx()), "copy_d2h(volume)"); // the host reads h_v next
            h_v.off = d_v.off;
        }

        // Extension (no reference counterpart): runs what is deferred and waits for everything enqueued on this thread's device.
        inline auto synchronize() -> void
        {
            detail::runtime_check(paris_hip_ctx_synchronize(current_ctx()), "synchronize()");
        }

        // ---- subvolume planning (src/cuda/subvolume_information.cpp:63-118) ---------------------------------------
        inline auto make_subvolume_information(const volume_geometry& vol_geo, const detector_geometry& det_geo) -> subvolume_info
        {
            paris_subvolume_info c_info{};
            // beside the slab a device also holds this backend's projection buffers: the rotation of make_projection_device buffers
            // and the pending group (or the snapshot ring) -- they do not shrink when the slab count doubles
            std::size_t reserve = 0;
            detail::construction_check(paris_hip_projection_reserve_bytes(current_ctx(), det_geo.n_row, det_geo.n_col, &reserve),
                                       "make_subvolume_information()");
            detail::construction_check(paris_hip_make_subvolume_information_reserving(detail::as_c<paris_volume_geometry>(vol_geo),
                                                                                      detail::as_c<paris_detector_geometry>(det_geo), 0, reserve, &c_info),
                                       "make_subvolume_information()");
            subvolume_info info{};
            info.geo.dim_x = c_info.geo.dim_x;
            info.geo.dim_y = c_info.geo.dim_y;
            info.geo.dim_z = c_info.geo.dim_z;
            info.geo.remainder = c_info.geo.remainder;
            info.num = c_info.num;
            return info;
        }

        // ---- the numeric stages ---------------------------------------------------------------------------------------
        inline auto weight(projection_device_type& p, float h_min, float v_min, float d_sd, float l_px_row, float l_px_col) -> void
        {
            detail::runtime_check(paris_hip_weight(current_ctx(), p.buf.get(), p.buf.pitch(), p.dim_x, p.dim_y, h_min, v_min, d_sd,
                                                   l_px_row, l_px_col), "weight()");
        }

        inline auto make_filter(std::uint32_t size, float tau) -> filter_buffer_type
        {
            float* k = nullptr;
            detail::construction_check(paris_hip_make_filter(current_ctx(), size, tau, &k), "make_filter()");
            return filter_buffer_type{k};
        }

        inline auto apply_filter(projection_device_type& p, const filter_buffer_type& k, std::uint32_t filter_size,
                                 std::uint32_t n_col) -> void
        {
            detail::runtime_check(paris_hip_apply_filter(current_ctx(), p.buf.get(), p.buf.pitch(), p.dim_x, p.dim_y, k.get(),
                                                         filter_size, n_col), "apply_filter()");
        }

        inline auto backproject(const projection_device_type& p, volume_device_type& v, std::uint32_t v_offset,
                                const detector_geometry& det_geo, const volume_geometry& vol_geo, bool enable_roi,
                                const region_of_interest& roi, float sin, float cos, float delta_s, float delta_t) -> void
        {
            detail::runtime_check(paris_hip_backproject(current_ctx(), p.buf.get(), p.buf.pitch(), p.dim_x, p.dim_y, v.buf.get(),
                                                        v.dim_x, v.dim_y, v.dim_z, v_offset, detail::as_c<paris_detector_geometry>(det_geo),
                                                        detail::as_c<paris_volume_geometry>(vol_geo), enable_roi ? 1 : 0,
                                                        detail::as_c<paris_region_of_interest>(roi), sin, cos, delta_s, delta_t),
                                  "backproject()");
        }
    }
}

#endif
