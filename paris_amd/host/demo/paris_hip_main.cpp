// paris.hip -- command-line reconstruction with the MI355X backend: HIS projections in, one DDBVF volume out.
// Options carry the reference's names (src/program_options.cpp:46-78); parsing is deliberately minimal (the
// reference's Boost.Program_options front end is out of scope).
//
//   paris.hip --geometry geo.ini --input <dir> --output <dir> [--name vol] [--angles file] [--quality q]
//             [--roi --roi-x1 a --roi-x2 b --roi-y1 c --roi-y2 d --roi-z1 e --roi-z2 f]
//             [--slabs n] [--devices n] [--f16] [--no-row-band] [--batch n] [--drain-chunk-kib n] [--share-frames 0|1] [--one-volume] [--pipeline-slabs n] [--no-read-ahead]
//             [--window ramp|shepp-logan]
// geo.ini: key=value lines for n_row n_col l_px_row l_px_col delta_s delta_t d_so d_od delta_phi (:83-91).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <string>

#include "paris/reconstruct.h"

namespace
{
    auto parse_geometry(const std::string& path) -> paris::detector_geometry
    {
        auto file = std::ifstream{path};
        if(!file)
            throw paris::stage_construction_error{"cannot open geometry file " + path};
        auto kv = std::map<std::string, std::string>{};
        auto line = std::string{};
        while(std::getline(file, line))
        {
            const auto hash = line.find('#');
            if(hash != std::string::npos)
                line.erase(hash);
            const auto eq = line.find('=');
            if(eq == std::string::npos)
                continue;
            auto trim = [](std::string s) {
                const auto b = s.find_first_not_of(" \t\r");
                const auto e = s.find_last_not_of(" \t\r");
                return b == std::string::npos ? std::string{} : s.substr(b, e - b + 1);
            };
            kv[trim(line.substr(0, eq))] = trim(line.substr(eq + 1));
        }
        auto need = [&](const char* k) -> const std::string& {
            auto it = kv.find(k);
            if(it == kv.end())
                throw paris::stage_construction_error{std::string{"the option '"} + k + "' is required but missing"};
            return it->second;
        };
        auto g = paris::detector_geometry{};
        g.n_row = static_cast<std::uint32_t>(std::stoul(need("n_row")));
        g.n_col = static_cast<std::uint32_t>(std::stoul(need("n_col")));
        g.l_px_row = std::stof(need("l_px_row"));
        g.l_px_col = std::stof(need("l_px_col"));
        g.delta_s = std::stof(need("delta_s"));
        g.delta_t = std::stof(need("delta_t"));
        g.d_so = std::stof(need("d_so"));
        g.d_od = std::stof(need("d_od"));
        g.delta_phi = std::stof(need("delta_phi"));
        return g;
    }
}

int main(int argc, char** argv)
{
    try
    {
        auto po = paris::program_options{};
        auto geometry = std::string{};
        for(int a = 1; a < argc; ++a)
        {
            const auto k = std::string{argv[a]};
            auto val = [&]() -> std::string {
                if(a + 1 >= argc)
                    throw paris::stage_construction_error{"missing value for " + k};
                return argv[++a];
            };
            if(k == "--help" || k == "-h")
            {
                std::printf("paris.hip --geometry geo.ini --input <dir of .his files> --output <dir> [--name vol] [--angles file] [--quality q]\n"
                            "          [--roi --roi-x1 a --roi-x2 b --roi-y1 c --roi-y2 d --roi-z1 e --roi-z2 f]\n"
                            "          [--slabs n] [--devices n] [--f16] [--window ramp|shepp-logan] [--batch n] [--no-row-band] [--share-frames 0|1] [--one-volume] [--pipeline-slabs n] [--no-read-ahead]\n"
                            "          [--drain-chunk-kib n]\n"
                            "geo.ini: key=value lines for n_row n_col l_px_row l_px_col delta_s delta_t d_so d_od delta_phi\n"
                            "Reconstructs the HIS projections of <dir> (sorted by path) into <output>/<name>.ddbvf on all MI355X of the node.\n");
                return 0;
            }
            if(k == "--geometry") geometry = val();
            else if(k == "--input") po.input_path = val();
            else if(k == "--output") po.output_path = val();
            else if(k == "--name") po.prefix = val();
            else if(k == "--angles") { po.angle_path = val(); po.enable_angles = true; }
            else if(k == "--quality") po.quality = static_cast<std::uint16_t>(std::stoul(val()));
            else if(k == "--roi") po.enable_roi = true;
            else if(k == "--roi-x1") po.roi.x1 = static_cast<std::uint32_t>(std::stoul(val()));
            else if(k == "--roi-x2") po.roi.x2 = static_cast<std::uint32_t>(std::stoul(val()));
            else if(k == "--roi-y1") po.roi.y1 = static_cast<std::uint32_t>(std::stoul(val()));
            else if(k == "--roi-y2") po.roi.y2 = static_cast<std::uint32_t>(std::stoul(val()));
            else if(k == "--roi-z1") po.roi.z1 = static_cast<std::uint32_t>(std::stoul(val()));
            else if(k == "--roi-z2") po.roi.z2 = static_cast<std::uint32_t>(std::stoul(val()));
            else if(k == "--slabs") po.slabs = std::stoi(val());
            else if(k == "--devices") po.devices = std::stoi(val());
            else if(k == "--f16") po.f16 = true;
            else if(k == "--no-row-band") po.row_band = false;
            else if(k == "--window")
            {
                const auto w = val();
                if(w == "ramp") po.window = PARIS_HIP_WINDOW_RAMP;
                else if(w == "shepp-logan") po.window = PARIS_HIP_WINDOW_SHEPP_LOGAN;
                else throw paris::stage_construction_error{"unknown filter window " + w};
            }
            else if(k == "--batch") po.batch = std::stoi(val());
            else if(k == "--share-frames") po.share_frames = std::stoi(val());
            else if(k == "--one-volume") po.two_volumes = false;
            else if(k == "--no-read-ahead") po.read_ahead = false;
            else if(k == "--pipeline-slabs") po.pipeline_slabs = std::stoi(val());
            else if(k == "--drain-chunk-kib") po.drain_chunk_bytes = static_cast<std::size_t>(std::stoull(val())) << 10;
            else throw paris::stage_construction_error{"unknown option " + k};
        }
        if(geometry.empty())
            throw paris::stage_construction_error{"the option '--geometry' is required but missing"};
        po.det_geo = parse_geometry(geometry);
        if(po.input_path.empty() || po.output_path.empty()) // src/program_options.cpp:117-122: both or neither
        {
            const auto v = paris::calculate_volume_geometry(po.det_geo);
            std::printf("Volume dimensions [vx]: %u x %u x %u, voxel size %.6g mm (no --input/--output: nothing to do)\n", v.dim_x, v.dim_y,
                        v.dim_z, v.l_vx_x);
            return 0;
        }
        po.enable_io = true;
        const auto r = paris::run(po);
        std::printf("volume %u x %u x %u (%d slab%s) -> %s in %.3f s\n", r.roi_geo.dim_x, r.roi_geo.dim_y, r.roi_geo.dim_z, r.info.num,
                    r.info.num == 1 ? "" : "s", r.output_file.c_str(), r.wall_s);
        if(r.devices.size() > 1 && r.shared_source)
            std::printf("shared frame source: %llu frames read from the files for %llu frame requests of the device threads\n",
                        static_cast<unsigned long long>(r.frames_read), static_cast<unsigned long long>(r.frames_requested));
        else if(r.devices.size() > 1)
            std::printf("frame source: one stream per device thread, each reads its slab's detector rows (%.2f detectors' worth per pass)\n",
                        r.rows_per_pass);
        for(const auto& s : r.skipped)
            std::printf("  skipped invalid file %s\n", s.c_str());
        for(const auto& d : r.devices)
        {
            std::printf("device %d: %u task(s), %u projections, %.0f detector rows per projection; host: setup %.3f s, source %.3f s%s, enqueue %.3f s, "
                        "waiting for the drain thread %.3f s (%s; drain thread: D2H %.3f s, save %.3f s)\n", d.device, d.tasks, d.projections,
                        d.tasks ? static_cast<double>(d.band_rows) / d.tasks : 0.0, d.setup_s, d.source_s,
                        po.read_ahead ? (" on the feed thread (the device thread waited " + std::to_string(d.source_wait_s).substr(0, 5) + " s for it)").c_str() : "",
                        d.enqueue_s, d.drain_wait_s, d.two_volumes ? "two volume buffers" : "one volume buffer", d.drain_s, d.save_s);
            for(const auto& s : d.skipped)
                std::printf("  skipped invalid file %s\n", s.c_str());
        }
        return 0;
    }
    catch(const paris::stage_construction_error& e)
    {
        std::fprintf(stderr, "main(): Pipeline construction failed: %s\nAborting.\n", e.what());
        return 1;
    }
    catch(const paris::stage_runtime_error& e)
    {
        std::fprintf(stderr, "main(): Pipeline execution failed: %s\nAborting.\n", e.what());
        return 1;
    }
    catch(const std::exception& e)
    {
        std::fprintf(stderr, "main(): %s\nAborting.\n", e.what());
        return 1;
    }
}
