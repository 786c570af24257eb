// C entry points over the host-only I/O code (HIS, DDBVF, directory listing, angle files) so that the CPU test suite
// can exercise it through ctypes. No GPU code in here; built as paris_amd/lib/libparis_io.so.
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "paris/ddbvf.h"
#include "paris/his.h"
#include "paris/source.h"

extern "C" {

// loads every frame of a HIS file; *data receives n_frames*dim_x*dim_y floats (free with paris_io_free).
// returns 0, or 1 when the file cannot be opened. A non-HIS file gives n_frames = 0.
int paris_io_his_load(const char* path, uint32_t* n_frames, uint32_t* dim_x, uint32_t* dim_y, float** data)
{
    try
    {
        const auto frames = paris::his::load(path);
        *n_frames = static_cast<uint32_t>(frames.size());
        *dim_x = frames.empty() ? 0 : frames[0].dim_x;
        *dim_y = frames.empty() ? 0 : frames[0].dim_y;
        *data = nullptr;
        if(!frames.empty())
        {
            const auto n = static_cast<size_t>(*dim_x) * *dim_y;
            *data = static_cast<float*>(std::malloc(n * frames.size() * sizeof(float)));
            for(size_t i = 0; i < frames.size(); ++i)
                std::memcpy(*data + n * i, frames[i].pixels.data(), n * sizeof(float));
        }
        return 0;
    }
    catch(const std::exception&) { return 1; }
}

void paris_io_free(void* p) { std::free(p); }

int paris_io_his_save(const char* path, const float* frames, uint16_t n_frames, uint16_t dim_x, uint16_t dim_y, uint16_t number_type,
                      uint16_t image_header_size)
{
    try { paris::his::save(path, frames, n_frames, dim_x, dim_y, static_cast<paris::his::number_type>(number_type), image_header_size); return 0; }
    catch(const std::exception&) { return 1; }
}

// creates <path>.ddbvf and writes one slab; `create` != 0 truncates and writes the header first
int paris_io_ddbvf_write(const char* path, int create, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, const float* voxels,
                         uint32_t dim_z_slab, uint32_t first)
{
    try
    {
        paris::ddbvf::handle_type h;
        if(create)
            h = paris::ddbvf::create(path, dim_x, dim_y, dim_z);
        else
        {
            h.reset(new paris::ddbvf::handle);
            h->dim_x = dim_x; h->dim_y = dim_y; h->dim_z = dim_z; h->offset = 8;
            h->file = std::fopen((std::string{path} + ".ddbvf").c_str(), "r+b");
            if(!h->file)
                return 1;
        }
        paris::ddbvf::write(h, voxels, dim_x, dim_y, dim_z_slab, first);
        return 0;
    }
    catch(const std::runtime_error&) { return 2; }
    catch(const std::exception&) { return 1; }
}

// angle file -> malloc'd float array
int paris_io_read_angles(const char* path, uint32_t* n, float** out)
{
    try
    {
        const auto a = paris::read_angles(path);
        *n = static_cast<uint32_t>(a.size());
        *out = static_cast<float*>(std::malloc((a.size() + 1) * sizeof(float)));
        std::memcpy(*out, a.data(), a.size() * sizeof(float));
        return 0;
    }
    catch(const std::exception&) { return 1; }
}

// drains a source; returns per-frame idx / phi and the number of skipped files. idx_out/phi_out hold up to cap entries.
int paris_io_source_scan(const char* dir, int enable_angles, const char* angle_file, uint16_t quality, uint32_t cap, uint32_t* n_frames,
                         uint32_t* idx_out, float* phi_out, float* first_pixel_out, uint32_t* n_skipped)
{
    try
    {
        paris::source src{dir, enable_angles != 0, angle_file ? angle_file : "", quality};
        uint32_t n = 0;
        while(!src.drained())
        {
            const auto p = src.load_next();
            if(!p.valid())
                break;
            if(n < cap)
            {
                idx_out[n] = p.idx;
                phi_out[n] = p.phi;
                first_pixel_out[n] = p.pixels.empty() ? 0.f : p.pixels[0];
            }
            ++n;
        }
        *n_frames = n;
        *n_skipped = static_cast<uint32_t>(src.skipped_files().size());
        return 0;
    }
    catch(const std::exception&) { return 1; }
}
// the same scan through frame_stream: frame i is written into data + i * dim_x * dim_y (caller-filled with a sentinel),
// only rows [row_first, row_first + row_count). Frames of another size end the scan with return code 3.
int paris_io_stream_scan(const char* dir, int enable_angles, const char* angle_file, uint16_t quality, uint32_t dim_x, uint32_t dim_y,
                         uint32_t row_first, uint32_t row_count, uint32_t cap, uint32_t* n_frames, uint32_t* idx_out, float* phi_out,
                         float* data, uint32_t* n_skipped)
{
    try
    {
        paris::frame_stream src{dir, enable_angles != 0, angle_file ? angle_file : "", quality};
        auto scratch = std::vector<float>(static_cast<size_t>(dim_x) * dim_y);
        uint32_t n = 0;
        for(;;)
        {
            float* dst = n < cap ? data + static_cast<size_t>(n) * dim_x * dim_y : scratch.data();
            const auto info = src.next(dst, dim_x, dim_y, row_first, row_count);
            if(!info.valid())
                break;
            if(info.dim_x != dim_x || info.dim_y != dim_y)
                return 3;
            if(n < cap)
            {
                idx_out[n] = info.idx;
                phi_out[n] = info.phi;
            }
            ++n;
        }
        *n_frames = n;
        *n_skipped = static_cast<uint32_t>(src.skipped_files().size());
        return 0;
    }
    catch(const std::exception&) { return 1; }
}

// shared_frames under `n_threads` consumers: thread i reads rows [row_first[i], row_first[i] + row_count[i]) of every frame into
// data + i * cap * dim_x * dim_y (caller-filled with a sentinel); thread i sleeps delay_us[i] microseconds per frame, so a
// slow consumer falls out of the ring (capacity) and takes its own fallback stream. counters: produced, served, reread.
int paris_io_shared_scan(const char* dir, int enable_angles, const char* angle_file, uint16_t quality, uint32_t dim_x, uint32_t dim_y,
                         uint32_t n_threads, const uint32_t* row_first, const uint32_t* row_count, const uint32_t* delay_us, uint32_t capacity,
                         uint32_t cap, uint32_t* n_frames, uint32_t* idx_out, float* phi_out, float* data, uint64_t* counters)
{
    try
    {
        paris::shared_frames shared{dir, enable_angles != 0, angle_file ? angle_file : "", quality, dim_x, dim_y, capacity};
        auto failed = std::vector<int>(n_threads, 0);
        auto workers = std::vector<std::thread>{};
        const auto frame = static_cast<size_t>(dim_x) * dim_y;
        for(uint32_t i = 0; i < n_threads; ++i)
            workers.emplace_back([&, i] {
                try
                {
                    auto cur = paris::shared_frames::cursor{};
                    auto scratch = std::vector<float>(frame);
                    uint32_t n = 0;
                    for(;;)
                    {
                        float* dst = n < cap ? data + (static_cast<size_t>(i) * cap + n) * frame : scratch.data();
                        const auto info = shared.next(cur, dst, dim_x, dim_y, row_first[i], row_count[i]);
                        if(!info.valid())
                            break;
                        if(n < cap)
                        {
                            idx_out[static_cast<size_t>(i) * cap + n] = info.idx;
                            phi_out[static_cast<size_t>(i) * cap + n] = info.phi;
                        }
                        ++n;
                        if(delay_us[i])
                            std::this_thread::sleep_for(std::chrono::microseconds(delay_us[i]));
                    }
                    n_frames[i] = n;
                }
                catch(const std::exception&) { failed[i] = 1; }
            });
        for(auto& w : workers)
            w.join();
        for(int f : failed)
            if(f)
                return 1;
        const auto st = shared.stats();
        counters[0] = st.produced;
        counters[1] = st.served;
        counters[2] = st.reread;
        return 0;
    }
    catch(const std::exception&) { return 1; }
}
}
