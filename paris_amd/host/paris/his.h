// HIS projection files: reader with the reference reader's exact acceptance rules, and a writer for synthetic sets.
//
// Reader semantics follow src/his.cpp:105-198: 68-byte little-endian file header read field by field, frames of
// (brx-ulx+1) x (bry-uly+1) pixels in one of five number types converted to float, and -- quirk Q14 -- the
// `image_header_size` bytes are skipped before EVERY frame. A file that is not HIS (wrong id / header size /
// number type) yields no frames (the caller logs and skips it: src/source.cpp:96-100); an unopenable file throws.
#ifndef PARIS_AMD_HOST_HIS_H_
#define PARIS_AMD_HOST_HIS_H_

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <system_error>
#include <vector>

namespace paris
{
    namespace his
    {
        constexpr std::uint16_t file_id = 0x7000;       // src/his.cpp:50
        constexpr std::uint16_t file_header_size = 68;  // :46

        enum number_type : std::uint16_t                // :70-78
        {
            type_uchar = 2,
            type_ushort = 4,
            type_dword = 32,
            type_double = 64,
            type_float = 128
        };

        struct header // :52-68 (on disk the fields are packed, little endian)
        {
            std::uint16_t file_type = file_id;
            std::uint16_t header_size = file_header_size;
            std::uint16_t header_version = 100;
            std::uint32_t file_size = 0;
            std::uint16_t image_header_size = 0;
            std::uint16_t ulx = 0, uly = 0, brx = 0, bry = 0;
            std::uint16_t frame_number = 0;
            std::uint16_t correction = 0;
            double integration_time = 0.0;
            std::uint16_t number_type = type_float;
        };

        struct frame
        {
            std::vector<float> pixels; // dim_x fastest
            std::uint32_t dim_x = 0, dim_y = 0;
        };

        namespace detail
        {
            struct file_closer { void operator()(std::FILE* f) const noexcept { if(f) std::fclose(f); } };
            using file_ptr = std::unique_ptr<std::FILE, file_closer>;

            template <typename T>
            bool get(std::FILE* f, T& v) { return std::fread(&v, sizeof(T), 1, f) == 1; }

            template <typename T>
            void put(std::FILE* f, const T& v) { if(std::fwrite(&v, sizeof(T), 1, f) != 1) throw std::system_error{errno, std::generic_category()}; }

            template <typename T>
            bool read_as_float(std::FILE* f, float* dst, std::size_t n)
            {
                auto tmp = std::vector<T>(n);
                const bool ok = std::fread(tmp.data(), sizeof(T), n, f) == n; // a short read leaves the tail as is
                for(std::size_t i = 0; i < n; ++i)
                    dst[i] = static_cast<float>(tmp[i]); // std::copy's implicit conversion: src/his.cpp:99
                return ok;
            }

            inline bool read_header(std::FILE* f, header& h)
            {
                std::uint8_t rest[34];
                return get(f, h.file_type) && get(f, h.header_size) && get(f, h.header_version) && get(f, h.file_size)
                       && get(f, h.image_header_size) && get(f, h.ulx) && get(f, h.uly) && get(f, h.brx) && get(f, h.bry)
                       && get(f, h.frame_number) && get(f, h.correction) && get(f, h.integration_time)
                       && get(f, h.number_type) && std::fread(rest, 1, sizeof(rest), f) == sizeof(rest);
            }
        }

        // src/his.cpp:105-198. `out_header` (optional) receives the parsed file header.
        inline auto load(const std::string& path, header* out_header = nullptr) -> std::vector<frame>
        {
            auto frames = std::vector<frame>{};
            auto f = detail::file_ptr{std::fopen(path.c_str(), "rb")};
            if(!f)
                throw std::system_error{errno, std::generic_category(), "his::load(): cannot open " + path};

            auto h = header{};
            h.file_type = 0;
            h.header_size = 0;
            detail::read_header(f.get(), h); // a short header leaves the id / size checks to fail below
            if(out_header)
                *out_header = h;
            if(h.file_type != file_id)            // :130-134
                return frames;
            if(h.header_size != file_header_size) // :135-139
                return frames;
            if(h.number_type == static_cast<std::uint16_t>(-1)) // :140-144
                return frames;

            const auto width = static_cast<std::uint32_t>(h.brx) - static_cast<std::uint32_t>(h.ulx) + 1u;  // :146-151
            const auto height = static_cast<std::uint32_t>(h.bry) - static_cast<std::uint32_t>(h.uly) + 1u;
            const auto n = static_cast<std::size_t>(static_cast<std::uint16_t>(width)) * static_cast<std::uint16_t>(height);
            for(std::uint32_t i = 0; i < h.frame_number; ++i)
            {
                if(h.image_header_size) // skipped before every frame: :155-159 (Q14)
                {
                    auto skip = std::vector<std::uint8_t>(h.image_header_size);
                    if(std::fread(skip.data(), 1, skip.size(), f.get()) != skip.size())
                        return frames; // truncated file
                }
                auto fr = frame{};
                fr.dim_x = width;
                fr.dim_y = height;
                fr.pixels.assign(static_cast<std::size_t>(width) * height, 0.f);
                switch(h.number_type) // :166-191
                {
                    case type_uchar: detail::read_as_float<std::uint8_t>(f.get(), fr.pixels.data(), n); break;
                    case type_ushort: detail::read_as_float<std::uint16_t>(f.get(), fr.pixels.data(), n); break;
                    case type_dword: detail::read_as_float<std::uint32_t>(f.get(), fr.pixels.data(), n); break;
                    case type_double: detail::read_as_float<double>(f.get(), fr.pixels.data(), n); break;
                    case type_float: detail::read_as_float<float>(f.get(), fr.pixels.data(), n); break;
                    default: return frames; // unsupported type: what was read so far (:188-190)
                }
                frames.push_back(std::move(fr));
            }
            return frames;
        }

        // Writer for synthetic projection sets (no reference counterpart; produces what load() accepts).
        // `frames` holds n_frames * dim_x * dim_y floats; values are converted to `type` with a plain cast.
        inline void save(const std::string& path, const float* frames, std::uint16_t n_frames, std::uint16_t dim_x,
                         std::uint16_t dim_y, number_type type = type_float, std::uint16_t image_header_size = 0)
        {
            auto f = detail::file_ptr{std::fopen(path.c_str(), "wb")};
            if(!f)
                throw std::system_error{errno, std::generic_category(), "his::save(): cannot open " + path};
            std::size_t px = 4;
            switch(type)
            {
                case type_uchar: px = 1; break;
                case type_ushort: px = 2; break;
                case type_dword: px = 4; break;
                case type_double: px = 8; break;
                case type_float: px = 4; break;
            }
            const auto n = static_cast<std::size_t>(dim_x) * dim_y;
            auto h = header{};
            h.image_header_size = image_header_size;
            h.ulx = 1; h.uly = 1; h.brx = dim_x; h.bry = dim_y; // 1-based inclusive rectangle
            h.frame_number = n_frames;
            h.number_type = type;
            h.file_size = static_cast<std::uint32_t>(file_header_size + n_frames * (image_header_size + n * px));
            using detail::put;
            put(f.get(), h.file_type); put(f.get(), h.header_size); put(f.get(), h.header_version); put(f.get(), h.file_size);
            put(f.get(), h.image_header_size); put(f.get(), h.ulx); put(f.get(), h.uly); put(f.get(), h.brx); put(f.get(), h.bry);
            put(f.get(), h.frame_number); put(f.get(), h.correction); put(f.get(), h.integration_time); put(f.get(), h.number_type);
            const std::uint8_t rest[34] = {};
            put(f.get(), rest);
            const auto img_hdr = std::vector<std::uint8_t>(image_header_size, 0xAB);
            for(std::uint16_t i = 0; i < n_frames; ++i)
            {
                if(image_header_size && std::fwrite(img_hdr.data(), 1, img_hdr.size(), f.get()) != img_hdr.size())
                    throw std::system_error{errno, std::generic_category()};
                const float* src = frames + n * i;
                for(std::size_t j = 0; j < n; ++j)
                {
                    switch(type)
                    {
                        case type_uchar: put(f.get(), static_cast<std::uint8_t>(src[j])); break;
                        case type_ushort: put(f.get(), static_cast<std::uint16_t>(src[j])); break;
                        case type_dword: put(f.get(), static_cast<std::uint32_t>(src[j])); break;
                        case type_double: put(f.get(), static_cast<double>(src[j])); break;
                        case type_float: put(f.get(), src[j]); break;
                    }
                }
            }
        }
    }
}

#endif
