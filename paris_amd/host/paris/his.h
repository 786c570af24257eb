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

            inline bool read_header(std::FILE* f, header& h)
            {
                std::uint8_t rest[34];
                return get(f, h.file_type) && get(f, h.header_size) && get(f, h.header_version) && get(f, h.file_size)
                       && get(f, h.image_header_size) && get(f, h.ulx) && get(f, h.uly) && get(f, h.brx) && get(f, h.bry)
                       && get(f, h.frame_number) && get(f, h.correction) && get(f, h.integration_time)
                       && get(f, h.number_type) && std::fread(rest, 1, sizeof(rest), f) == sizeof(rest);
            }
        }

        // Frame-at-a-time reader with the acceptance rules of src/his.cpp:105-198. load() below is built on it; the
        // pipelined driver uses it directly to convert a frame (or only a band of its rows) straight into its pinned
        // upload buffer, without the intermediate copy the reference's reader makes.
        class reader
        {
        public:
            explicit reader(const std::string& path) : f_{std::fopen(path.c_str(), "rb")}
            {
                if(!f_)
                    throw std::system_error{errno, std::generic_category(), "his::load(): cannot open " + path};
                h_.file_type = 0;
                h_.header_size = 0;
                detail::read_header(f_.get(), h_); // a short header leaves the id / size checks to fail below
                accepted_ = h_.file_type == file_id                              // :130-134
                            && h_.header_size == file_header_size                // :135-139
                            && h_.number_type != static_cast<std::uint16_t>(-1); // :140-144
                const auto width = static_cast<std::uint32_t>(h_.brx) - static_cast<std::uint32_t>(h_.ulx) + 1u; // :146-151
                const auto height = static_cast<std::uint32_t>(h_.bry) - static_cast<std::uint32_t>(h_.uly) + 1u;
                dim_x_ = width;
                dim_y_ = height;
                // the reference reads (u16)width * (u16)height elements into a width * height buffer
                n_ = static_cast<std::size_t>(static_cast<std::uint16_t>(width)) * static_cast<std::uint16_t>(height);
                switch(h_.number_type) // :166-191
                {
                    case type_uchar: px_ = 1; break;
                    case type_ushort: px_ = 2; break;
                    case type_dword: px_ = 4; break;
                    case type_double: px_ = 8; break;
                    case type_float: px_ = 4; break;
                    default: px_ = 0; break; // unsupported type: no (further) frames (:188-190)
                }
            }

            auto head() const noexcept -> const header& { return h_; }
            auto dim_x() const noexcept -> std::uint32_t { return dim_x_; }
            auto dim_y() const noexcept -> std::uint32_t { return dim_y_; }

            // Moves to the next frame; false when the file holds no further readable frame.
            bool advance()
            {
                if(in_frame_)
                    skip_payload();
                if(!accepted_ || px_ == 0 || next_ >= h_.frame_number)
                    return false;
                if(h_.image_header_size) // skipped before every frame: :155-159 (Q14)
                {
                    skip_.resize(h_.image_header_size);
                    if(std::fread(skip_.data(), 1, skip_.size(), f_.get()) != skip_.size())
                    {
                        accepted_ = false; // truncated file
                        return false;
                    }
                }
                ++next_;
                in_frame_ = true;
                return true;
            }

            // Converts rows [row_first, row_first + row_count) of the current frame into dst (dim_x * dim_y floats, row
            // stride dim_x); other rows of dst are left untouched. Elements the file no longer holds read as 0.
            void read_rows(float* dst, std::uint32_t row_first, std::uint32_t row_count)
            {
                if(!in_frame_)
                    return;
                const auto total = static_cast<std::size_t>(dim_x_) * dim_y_;
                auto a = static_cast<std::size_t>(row_first) * dim_x_;
                auto b = a + static_cast<std::size_t>(row_count) * dim_x_;
                a = a < total ? a : total;
                b = b < total ? b : total;
                const auto stored_b = b < n_ ? b : n_; // elements beyond n_ are not in the file: they stay 0
                const auto stored_a = a < stored_b ? a : stored_b;
                const auto count = stored_b - stored_a;
                raw_.resize(count * px_);
                std::size_t got = 0;
                if(count && std::fseek(f_.get(), static_cast<long>(stored_a * px_), SEEK_CUR) == 0)
                    got = std::fread(raw_.data(), px_, count, f_.get());
                consumed_ = stored_a * px_ + got * px_;
                if(got < count)
                    std::memset(raw_.data() + got * px_, 0, (count - got) * px_); // a short read leaves zeros (:99)
                float* out = dst + stored_a;
                switch(h_.number_type)
                {
                    case type_uchar: convert<std::uint8_t>(out, count); break;
                    case type_ushort: convert<std::uint16_t>(out, count); break;
                    case type_dword: convert<std::uint32_t>(out, count); break;
                    case type_double: convert<double>(out, count); break;
                    default: convert<float>(out, count); break;
                }
                for(auto i = stored_b; i < b; ++i)
                    dst[i] = 0.f;
                skip_payload();
            }

        private:
            template <typename T>
            void convert(float* out, std::size_t count) const
            {
                const auto* src = raw_.data();
                for(std::size_t i = 0; i < count; ++i)
                {
                    T v;
                    std::memcpy(&v, src + i * sizeof(T), sizeof(T));
                    out[i] = static_cast<float>(v); // std::copy's implicit conversion: src/his.cpp:99
                }
            }

            void skip_payload() // leaves the file position behind the current frame's pixels
            {
                const auto bytes = n_ * px_;
                if(consumed_ < bytes)
                    std::fseek(f_.get(), static_cast<long>(bytes - consumed_), SEEK_CUR); // past the end is fine: the next read fails
                consumed_ = 0;
                in_frame_ = false;
            }

            detail::file_ptr f_;
            header h_{};
            bool accepted_ = false, in_frame_ = false;
            std::uint32_t dim_x_ = 0, dim_y_ = 0, next_ = 0;
            std::size_t n_ = 0, px_ = 0, consumed_ = 0;
            std::vector<std::uint8_t> skip_, raw_;
        };

        // src/his.cpp:105-198. `out_header` (optional) receives the parsed file header.
        inline auto load(const std::string& path, header* out_header = nullptr) -> std::vector<frame>
        {
            auto frames = std::vector<frame>{};
            auto r = reader{path};
            if(out_header)
                *out_header = r.head();
            while(r.advance())
            {
                auto fr = frame{};
                fr.dim_x = r.dim_x();
                fr.dim_y = r.dim_y();
                fr.pixels.assign(static_cast<std::size_t>(fr.dim_x) * fr.dim_y, 0.f);
                r.read_rows(fr.pixels.data(), 0, fr.dim_y);
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
