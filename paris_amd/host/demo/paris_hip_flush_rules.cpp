// The deferred plugin boundary at full size, through the C++ mirror.
//
// paris::hip defers backproject() calls (PARIS_HIP_BACKPROJECT_DEFERRAL per fused launch) and holds weight() back for the filter that follows
// (stage fusion). Both are invisible only if every way PARIS can observe device data first runs what is pending
// (include/paris_hip.h, paris_hip_set_backproject_deferral / paris_hip_set_stage_fusion). This program drives the call
// sequence of src/main.cpp:98-105 on the geometry of BASELINE config 3 (2048^2 detector, the 2048^3 grid) into two
// z-slabs of the volume twice -- once with one launch per call (deferral 1, fusion off: the reference's behaviour call by
// call), once with the mirror's defaults -- and interleaves, in the second run, everything that must flush:
//   copy_d2h of a slab in the middle of a group of deferred calls, backproject() calls alternating between two slabs,
//   make/free of an unrelated volume and of projection buffers while calls are pending, a weight() whose projection is
//   read back before any filter, synchronize().
// Every read-back of the second run must equal the first run's bit for bit. Exit code 0 and "flush rules ok" on success.
//
// usage: paris_hip_flush_rules [slices per slab = 64] [projections = 40]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "paris/stages.h"

namespace
{
    namespace be = paris::backend;

    void lcg_fill(float* p, std::size_t n, std::uint32_t idx)
    {
        std::uint32_t s = 12345u + idx;
        for(std::size_t i = 0; i < n; ++i)
        {
            s = s * 1664525u + 1013904223u;
            p[i] = static_cast<float>(s >> 8) / 16777216.f - 0.5f;
        }
    }

    struct snapshot
    {
        std::vector<float> a_mid, a_end, b_end, weighted;
    };

    auto to_vector(const be::volume_device_type& d_v) -> std::vector<float>
    {
        auto h = be::make_volume_host(d_v.dim_x, d_v.dim_y, d_v.dim_z);
        be::copy_d2h(d_v, h); // flushes what is pending for d_v
        const auto n = static_cast<std::size_t>(d_v.dim_x) * d_v.dim_y * d_v.dim_z;
        return std::vector<float>(h.buf.get(), h.buf.get() + n);
    }

    // interleave = false: the plain loop. true: the same arithmetic with the flush triggers in between
    auto run(const paris::detector_geometry& det, const paris::volume_geometry& vol_geo, std::uint32_t slab_z, std::uint32_t n_proj,
             const std::vector<std::vector<float>>& frames, bool interleave) -> snapshot
    {
        auto snap = snapshot{};
        const auto off_a = vol_geo.dim_z / 2u - slab_z, off_b = vol_geo.dim_z - slab_z; // a central slab and the top one
        auto v_a = be::make_volume_device(vol_geo.dim_x, vol_geo.dim_y, slab_z);
        auto v_b = be::make_volume_device(vol_geo.dim_x, vol_geo.dim_y, slab_z);
        const auto roi = paris::region_of_interest{};
        for(std::uint32_t i = 0; i < n_proj; ++i)
        {
            auto h_p = be::make_projection_host(det.n_row, det.n_col);
            std::memcpy(h_p.buf.get(), frames[i % frames.size()].data(), static_cast<std::size_t>(det.n_row) * det.n_col * sizeof(float));
            h_p.idx = i * 36u; // 9 degrees apart: the group of 16 pending calls spans 144 degrees
            auto d_p = paris::load(h_p);             // src/main.cpp:101
            paris::weight(d_p, det);                 // :102 (held back under stage fusion)
            if(i == 3u)
            {
                // read the weighted projection back before any filter call: the held-back weighting must have run
                auto h_w = be::make_projection_host(det.n_row, det.n_col);
                be::copy_d2h(d_p, h_w);
                snap.weighted.assign(h_w.buf.get(), h_w.buf.get() + static_cast<std::size_t>(det.n_row) * det.n_col);
            }
            paris::filter(d_p, det);                 // :103
            // projections 0..12 go into slab A (13 calls pending when A is read at i == 12), 13..23 into slab B (11 pending when
            // the alternation starts), then the calls alternate: every switch must flush the other slab's pending group
            const bool into_b = (i > 12u && i < 24u) || (i >= 24u && (i % 2u) == 1u);
            paris::backproject(d_p, into_b ? v_b : v_a, into_b ? off_b : off_a, det, vol_geo, false, false, roi); // :104
            if(interleave)
            {
                if(i == 9u)
                {
                    // an unrelated volume comes and goes while calls for v_a and v_b are pending
                    auto other = be::make_volume_device(256, 256, 16);
                    (void)other;
                }
                if(i == 20u)
                    be::synchronize();
            }
            if(i == 12u) // in the middle of a group of deferred calls (both runs read here; only the deferred one has work pending)
                snap.a_mid = to_vector(v_a);
            // d_p and h_p are released here, as in PARIS's loop: the pools recycle them while launches are still queued
        }
        snap.a_end = to_vector(v_a);
        snap.b_end = to_vector(v_b);
        return snap;
    }

    auto same(const std::vector<float>& x, const std::vector<float>& y, const char* what) -> bool
    {
        if(x.size() != y.size() || std::memcmp(x.data(), y.data(), x.size() * sizeof(float)) != 0)
        {
            std::size_t first = 0;
            while(first < x.size() && first < y.size() && std::memcmp(&x[first], &y[first], sizeof(float)) == 0)
                ++first;
            std::fprintf(stderr, "MISMATCH in %s (sizes %zu / %zu, first difference at %zu)\n", what, x.size(), y.size(), first);
            return false;
        }
        double sum = 0;
        for(float v : x)
            sum += v;
        std::printf("%s: %zu values equal bit for bit (sum %.9g)\n", what, x.size(), sum);
        return true;
    }
}

int main(int argc, char** argv)
{
    try
    {
        const auto slab_z = static_cast<std::uint32_t>(argc > 1 ? std::atoi(argv[1]) : 64);
        const auto n_proj = static_cast<std::uint32_t>(argc > 2 ? std::atoi(argv[2]) : 40);
        // BASELINE config 3 (bench.py): 2048^2 detector, 0.2 mm pixels, d_so = d_od = 500 mm, the natural 2048^3 grid
        auto det = paris::detector_geometry{};
        det.n_row = 2048; det.n_col = 2048; det.l_px_row = 0.2f; det.l_px_col = 0.2f;
        det.delta_s = 0.f; det.delta_t = 0.f; det.d_so = 500.f; det.d_od = 500.f; det.delta_phi = 0.25f;
        auto vol_geo = paris::calculate_volume_geometry(det);
        vol_geo.dim_x = vol_geo.dim_y = vol_geo.dim_z = 2048; // bench.py's grid (the natural one is 2048 +- rounding)

        auto devices = be::get_devices();
        if(devices.empty())
            throw paris::stage_construction_error{"no HIP device"};
        be::set_device(devices[0]);
        auto* ctx = be::current_ctx();

        auto frames = std::vector<std::vector<float>>(4, std::vector<float>(static_cast<std::size_t>(det.n_row) * det.n_col));
        for(std::uint32_t f = 0; f < frames.size(); ++f)
            lcg_fill(frames[f].data(), frames[f].size(), f);

        // run 1: one launch per call, nothing held back
        be::detail::runtime_check(paris_hip_set_backproject_deferral(ctx, 1), "deferral");
        be::detail::runtime_check(paris_hip_set_stage_fusion(ctx, 0), "fusion");
        const auto plain = run(det, vol_geo, slab_z, n_proj, frames, false);
        // run 2: the mirror's defaults, with the flush triggers interleaved
        be::detail::runtime_check(paris_hip_set_backproject_deferral(ctx, PARIS_HIP_BACKPROJECT_DEFERRAL), "deferral");
        be::detail::runtime_check(paris_hip_set_stage_fusion(ctx, PARIS_HIP_STAGE_FUSION), "fusion");
        const auto deferred = run(det, vol_geo, slab_z, n_proj, frames, true);

        // run 3: the same with the fused launches on the ctx's second stream (paris_hip_set_backproject_overlap; off by default):
        // every observer must also join the two streams
        be::detail::runtime_check(paris_hip_set_backproject_overlap(ctx, 1), "overlap");
        const auto overlapped = run(det, vol_geo, slab_z, n_proj, frames, true);
        be::detail::runtime_check(paris_hip_set_backproject_overlap(ctx, 0), "overlap");

        bool ok = same(plain.weighted, deferred.weighted, "weighted projection read before its filter");
        ok = same(plain.a_mid, overlapped.a_mid, "slab A read in the middle of a deferred group, launches on the second stream") && ok;
        ok = same(plain.a_end, overlapped.a_end, "slab A at the end, launches on the second stream") && ok;
        ok = same(plain.b_end, overlapped.b_end, "slab B at the end, launches on the second stream") && ok;
        ok = same(plain.a_mid, deferred.a_mid, "slab A read in the middle of a deferred group") && ok;
        ok = same(plain.a_end, deferred.a_end, "slab A at the end") && ok;
        ok = same(plain.b_end, deferred.b_end, "slab B at the end (calls alternating between the slabs)") && ok;
        if(!ok)
            return 1;
        std::printf("flush rules ok: 2048^2 detector, 2048^3 grid, two slabs of %u slices, %u projections, deferral %d, stage fusion %d\n",
                    slab_z, n_proj, PARIS_HIP_BACKPROJECT_DEFERRAL, PARIS_HIP_STAGE_FUSION);
        return 0;
    }
    catch(const std::exception& e)
    {
        std::fprintf(stderr, "paris_hip_flush_rules: %s\n", e.what());
        return 1;
    }
}
