// A PARIS-style driver over the C++ backend mirror: the per-device loop of src/main.cpp:79-109 with the file
// stages replaced by raw float32 files, so tests can feed known projections and compare the volume.
//
// usage: paris_hip_demo <n_row> <n_col> <l_px_row> <l_px_col> <delta_s> <delta_t> <d_so> <d_od> <delta_phi>
//                       <n_proj> <in.raw | lcg> <out.raw> [--no-weight] [--no-filter]
//                       [--slabs N] [--roi x1 x2 y1 y2 z1 z2] [--vol dx dy dz l_vx] [--cycle K] [--no-out] [--order N] [--json]
// in.raw holds n_proj frames of n_col x n_row float32; "lcg" generates the SURVEY.md 8c noise frames.
// --cycle K: only K distinct lcg frames are held in host memory and projection i is frame i mod K (throughput runs over a whole
// circle of large frames: 1440 frames of 2048^2 would be 23 GiB); --no-out: the volume is neither read back nor written to out.raw.
// --json: one more line, the same figures as a JSON object (bench.py's paris_loop leg reads it).
// out.raw receives the whole (ROI) volume, slabs written at their slice offsets (fixing SURVEY.md Q4).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "paris/stages.h"

namespace
{
    void lcg_fill(float* p, std::size_t n, std::uint32_t idx)
    {
        std::uint32_t s = 12345u + idx;
        for(std::size_t i = 0; i < n; ++i)
        {
            s = s * 1664525u + 1013904223u;
            p[i] = static_cast<float>(s >> 8) / 16777216.f;
        }
    }
}

int main(int argc, char** argv)
{
    if(argc < 13)
    {
        std::fprintf(stderr, "usage: see the header of paris_hip_demo.cpp\n");
        return 2;
    }
    try
    {
        auto det = paris::detector_geometry{};
        det.n_row = static_cast<std::uint32_t>(std::atoi(argv[1]));
        det.n_col = static_cast<std::uint32_t>(std::atoi(argv[2]));
        det.l_px_row = std::strtof(argv[3], nullptr);
        det.l_px_col = std::strtof(argv[4], nullptr);
        det.delta_s = std::strtof(argv[5], nullptr);
        det.delta_t = std::strtof(argv[6], nullptr);
        det.d_so = std::strtof(argv[7], nullptr);
        det.d_od = std::strtof(argv[8], nullptr);
        det.delta_phi = std::strtof(argv[9], nullptr);
        const auto n_proj = static_cast<std::uint32_t>(std::atoi(argv[10]));
        const auto in_path = std::string{argv[11]};
        const auto out_path = std::string{argv[12]};

        bool do_weight = true, do_filter = true, enable_roi = false, write_out = true, json = false;
        int slabs = 1;
        std::uint32_t cycle = 0;
        int order = -1; // --order N: workgroup -> tile order of the backprojection kernels (A/B; -1 = the library's choice)
        auto roi = paris::region_of_interest{};
        auto vol_geo = paris::calculate_volume_geometry(det);
        for(int a = 13; a < argc; ++a)
        {
            if(!std::strcmp(argv[a], "--no-weight")) do_weight = false;
            else if(!std::strcmp(argv[a], "--no-filter")) do_filter = false;
            else if(!std::strcmp(argv[a], "--slabs") && a + 1 < argc) slabs = std::atoi(argv[++a]);
            else if(!std::strcmp(argv[a], "--cycle") && a + 1 < argc) cycle = static_cast<std::uint32_t>(std::atoi(argv[++a]));
            else if(!std::strcmp(argv[a], "--no-out")) write_out = false;
            else if(!std::strcmp(argv[a], "--json")) json = true;
            else if(!std::strcmp(argv[a], "--order") && a + 1 < argc) order = std::atoi(argv[++a]);
            else if(!std::strcmp(argv[a], "--roi") && a + 6 < argc)
            {
                enable_roi = true;
                roi.x1 = std::atoi(argv[a + 1]); roi.x2 = std::atoi(argv[a + 2]);
                roi.y1 = std::atoi(argv[a + 3]); roi.y2 = std::atoi(argv[a + 4]);
                roi.z1 = std::atoi(argv[a + 5]); roi.z2 = std::atoi(argv[a + 6]);
                a += 6;
            }
            else if(!std::strcmp(argv[a], "--vol") && a + 4 < argc)
            {
                vol_geo.dim_x = std::atoi(argv[a + 1]); vol_geo.dim_y = std::atoi(argv[a + 2]); vol_geo.dim_z = std::atoi(argv[a + 3]);
                vol_geo.l_vx_x = vol_geo.l_vx_y = vol_geo.l_vx_z = std::strtof(argv[a + 4], nullptr);
                a += 4;
            }
            else
            {
                std::fprintf(stderr, "unknown argument %s\n", argv[a]);
                return 2;
            }
        }
        auto roi_geo = vol_geo;
        if(enable_roi)
            roi_geo = paris::apply_roi(vol_geo, roi.x1, roi.x2, roi.y1, roi.y2, roi.z1, roi.z2); // src/main.cpp:125-130

        auto devices = paris::backend::get_devices();
        if(devices.empty())
            throw paris::stage_construction_error{"no HIP device"};
        paris::backend::set_device(devices[0]); // src/main.cpp:87
        if(order >= 0)
            paris::backend::detail::runtime_check(paris_hip_set_backproject_order(paris::backend::current_ctx(), order, -1), "--order");

        // fixed slab count (the memory-driven count is backend::make_subvolume_information)
        auto info = paris::subvolume_info{};
        info.num = slabs < 1 ? 1 : slabs;
        info.geo = {roi_geo.dim_x, roi_geo.dim_y, roi_geo.dim_z / static_cast<std::uint32_t>(info.num),
                    roi_geo.dim_z % static_cast<std::uint32_t>(info.num)};

        // all frames in host memory (the reference re-reads them per task: src/main.cpp:93)
        const auto frame = std::size_t{det.n_row} * det.n_col;
        const auto n_held = (in_path == "lcg" && cycle != 0 && cycle < n_proj) ? cycle : n_proj;
        auto frames = std::vector<float>(frame * n_held);
        if(in_path == "lcg")
            for(std::uint32_t i = 0; i < n_held; ++i)
                lcg_fill(frames.data() + frame * i, frame, i);
        else
        {
            std::FILE* f = std::fopen(in_path.c_str(), "rb");
            if(f == nullptr || std::fread(frames.data(), sizeof(float), frames.size(), f) != frames.size())
                throw paris::stage_runtime_error{"cannot read " + in_path};
            std::fclose(f);
        }

        std::FILE* out = write_out ? std::fopen(out_path.c_str(), "wb") : nullptr;
        if(write_out && out == nullptr)
            throw paris::stage_runtime_error{"cannot open " + out_path};

        using clock = std::chrono::steady_clock;
        double split_s[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // make_projection_host, frame fill, load, weight, filter, backproject, free device, free host
        double fill_s = 0.0, tail_s = 0.0; // of loop_s: the host's own frame fill (memcpy into the pinned buffer), the wait for the GPU after the last call
        double loop_s = 0.0; // the per-projection loops only (volume allocation, read-back and file output are not the hot path)
        for(int id = 0; id < info.num; ++id) // one task per slab: src/task.cpp:38-48, src/main.cpp:89-108
        {
            const bool last = (info.num - id) <= 1;
            auto v = paris::make_volume(info.geo, last);
            const auto offset = static_cast<std::uint32_t>(id) * info.geo.dim_z;
            const auto t_start = std::chrono::steady_clock::now();
            for(std::uint32_t i = 0; i < n_proj; ++i)
            {
                // every call of one iteration of src/main.cpp:98-105 between two clock reads (~20 ns each)
                const auto c0 = clock::now();
                auto p = paris::backend::make_projection_host(det.n_row, det.n_col);
                const auto c1 = clock::now();
                std::memcpy(p.buf.get(), frames.data() + frame * (i % n_held), frame * sizeof(float));
                const auto c2 = clock::now();
                p.idx = i;
                auto d_p = paris::load(p);
                const auto c3 = clock::now();
                if(do_weight) paris::weight(d_p, det);
                const auto c4 = clock::now();
                if(do_filter) paris::filter(d_p, det);
                const auto c5 = clock::now();
                paris::backproject(d_p, v, offset, det, vol_geo, false, enable_roi, roi);
                const auto c6 = clock::now();
                d_p.buf = paris::backend::projection_device_buffer_type{}; // (what the end of the reference's loop body does)
                const auto c7 = clock::now();
                p.buf.reset();
                const auto c8 = clock::now();
                const clock::time_point c[9] = {c0, c1, c2, c3, c4, c5, c6, c7, c8};
                for(int k = 0; k < 8; ++k)
                    split_s[k] += std::chrono::duration<double>(c[k + 1] - c[k]).count();
            }
            fill_s = split_s[1];
            const auto t_tail = std::chrono::steady_clock::now();
            paris::backend::synchronize(); // the timed region ends when the GPU has finished, not when the last call returned
            tail_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tail).count();
            loop_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            if(out == nullptr) // --no-out: a throughput run, the volume is neither read back nor written
                continue;
            auto h_v = paris::backend::make_volume_host(v.dim_x, v.dim_y, v.dim_z);
            paris::backend::copy_d2h(v, h_v);
            const auto n = std::size_t{v.dim_x} * v.dim_y * v.dim_z;
            std::fseek(out, static_cast<long>(std::size_t{v.dim_x} * v.dim_y * offset * sizeof(float)), SEEK_SET);
            if(std::fwrite(h_v.buf.get(), sizeof(float), n, out) != n)
                throw paris::stage_runtime_error{"short write"};
        }
        if(out != nullptr)
            std::fclose(out);
        std::printf("ok %u %u %u\n", roi_geo.dim_x, roi_geo.dim_y, roi_geo.dim_z);
        std::printf("projection loops %.3f s: %.1f GVoxel-updates/s through paris::load / weight / filter / backproject (deferral depth %d, %s)\n", loop_s,
                    static_cast<double>(roi_geo.dim_x) * roi_geo.dim_y * roi_geo.dim_z * n_proj / loop_s / 1e9, PARIS_HIP_BACKPROJECT_DEFERRAL,
                    PARIS_HIP_BACKPROJECT_OVERLAP ? (PARIS_HIP_UPLOAD_ON_ITS_OWN_STREAM ? "fused launches on the second stream, uploads on the upload stream"
                                                                                         : "fused launches on the second stream, uploads on the compute stream")
                                                  : "one stream");
        std::printf("of which: host frame fill (memcpy into the pinned buffer) %.3f s, backend calls %.3f s, final wait for the GPU %.3f s\n", fill_s,
                    loop_s - fill_s - tail_s, tail_s);
        {
            const double per = 1e6 / (static_cast<double>(n_proj) * info.num);
            std::printf("per projection [us]: make_projection_host %.2f, frame fill %.2f, load (make_projection_device + copy_h2d) %.2f, weight %.2f, filter %.2f, "
                        "backproject %.2f, free device buffer %.2f, free host buffer %.2f\n", split_s[0] * per, split_s[1] * per, split_s[2] * per,
                        split_s[3] * per, split_s[4] * per, split_s[5] * per, split_s[6] * per, split_s[7] * per);
        }
        if(json)
        {
            const double per = 1e6 / (static_cast<double>(n_proj) * info.num);
            // what the device holds when the loops are over (the last slab's volume has been released by now): the library's parked
            // projection buffers and tables -- bounded, however many projections went through
            std::size_t mem_free = 0, mem_total = 0;
            (void)paris_hip_device_memory(devices[0], &mem_free, &mem_total);
            std::printf("{\"volume\": [%u, %u, %u], \"projections\": %u, \"frame\": [%u, %u], \"slabs\": %d, \"seconds\": %.6f, \"value\": %.3f, "
                        "\"unit\": \"GVoxel-updates/s\", \"host_fill_seconds\": %.6f, \"backend_call_seconds\": %.6f, \"final_wait_seconds\": %.6f, "
                        "\"deferral\": %d, \"streams\": \"%s\", \"filter_deferral\": %d, \"by_reference\": %d, \"device_bytes_in_use_after_the_loops\": %zu, \"us_per_projection\": {\"make_projection_host\": %.3f, "
                        "\"frame_fill\": %.3f, \"load\": %.3f, \"weight\": %.3f, \"filter\": %.3f, \"backproject\": %.3f, \"free_device\": %.3f, "
                        "\"free_host\": %.3f}}\n", roi_geo.dim_x, roi_geo.dim_y, roi_geo.dim_z, n_proj, det.n_row, det.n_col, info.num, loop_s,
                        static_cast<double>(roi_geo.dim_x) * roi_geo.dim_y * roi_geo.dim_z * n_proj / loop_s / 1e9, fill_s, loop_s - fill_s - tail_s, tail_s,
                        PARIS_HIP_BACKPROJECT_DEFERRAL, PARIS_HIP_BACKPROJECT_OVERLAP ? (PARIS_HIP_UPLOAD_ON_ITS_OWN_STREAM ? "compute + second (fused launches) + upload" : "compute (copies) + second (filter, fused launches)") : "one",
                        PARIS_HIP_FILTER_DEFERRAL, PARIS_HIP_BACKPROJECT_REFERENCES, mem_total - mem_free, split_s[0] * per, split_s[1] * per, split_s[2] * per, split_s[3] * per, split_s[4] * per,
                        split_s[5] * per, split_s[6] * per, split_s[7] * per);
        }
        return 0;
    }
    catch(const paris::stage_construction_error& e)
    {
        std::fprintf(stderr, "pipeline construction failed: %s\n", e.what());
        return 1;
    }
    catch(const paris::stage_runtime_error& e)
    {
        std::fprintf(stderr, "pipeline execution failed: %s\n", e.what());
        return 1;
    }
}
