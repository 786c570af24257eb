// DDBVF volume files: the layout written by src/ddbvf.cpp:72-153.
//
//   bytes  0..3   u32 0xEFDDDAFA
//          4..7   i32 0x0010            (the reference writes an `int` here: quirk Q13)
//          8..23  u32 dim_x, dim_y, dim_z, offset (= 8)
//         24..31  zero
//         32..    float32 voxels, x fastest; slice `first` starts at 32 + dim_x*dim_y*first*4 (64-bit here; the
//                 reference's 32-bit product overflows past 4 GiB, Q3)
#ifndef PARIS_AMD_HOST_DDBVF_H_
#define PARIS_AMD_HOST_DDBVF_H_

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <system_error>

#include <unistd.h>

namespace paris
{
    namespace ddbvf
    {
        constexpr std::uint32_t id = 0xEFDDDAFAu;  // src/ddbvf.cpp:45
        constexpr std::int32_t version = 0x0010;   // :46
        constexpr long first_pos = 32;              // :58

        struct handle
        {
            std::uint32_t dim_x = 0, dim_y = 0, dim_z = 0, offset = 0;
            std::FILE* file = nullptr;
            ~handle() { if(file) std::fclose(file); }
        };
        using handle_type = std::unique_ptr<handle>;

        // src/ddbvf.cpp:72-99: creates `path`.ddbvf and writes the 32-byte header
        inline auto create(const std::string& path, std::uint32_t dim_x, std::uint32_t dim_y, std::uint32_t dim_z) -> handle_type
        {
            auto h = handle_type{new handle};
            h->dim_x = dim_x;
            h->dim_y = dim_y;
            h->dim_z = dim_z;
            h->offset = static_cast<std::uint32_t>(first_pos - 4 - 4 - 16); // :80
            const auto full = path + ".ddbvf";
            h->file = std::fopen(full.c_str(), "w+b");
            if(!h->file)
                throw std::system_error{errno, std::generic_category(), "ddbvf::create(): " + full};
            const std::uint32_t dims[4] = {dim_x, dim_y, dim_z, h->offset};
            const char zeros[8] = {};
            if(std::fwrite(&id, 4, 1, h->file) != 1 || std::fwrite(&version, 4, 1, h->file) != 1
               || std::fwrite(dims, 4, 4, h->file) != 4 || std::fwrite(zeros, 1, h->offset, h->file) != h->offset)
                throw std::system_error{errno, std::generic_category(), "ddbvf::create(): header"};
            std::fflush(h->file);
            return h;
        }

        // src/ddbvf.cpp:125-153: writes a slab of dim_z_slab slices starting at global slice `first`
        inline void write(handle_type& h, const float* voxels, std::uint32_t dim_x, std::uint32_t dim_y,
                          std::uint32_t dim_z_slab, std::uint32_t first)
        {
            if(!h || voxels == nullptr) // :127-128
                return;
            if(first >= h->dim_z)       // :131-132
                throw std::runtime_error{"ddbvf::write(): Starting position out of bounds"};
            if(dim_x != h->dim_x || dim_y != h->dim_y || dim_z_slab > h->dim_z) // :134-135
                throw std::runtime_error{"ddbvf::write(): Attempting to save volume to file with wrong dimensions"};
            if(dim_z_slab > h->dim_z - first) // the reference checks the two above only: a slab must also end inside the volume
                throw std::runtime_error{"ddbvf::write(): Slab extends past the end of the volume"};
            const auto slice = static_cast<std::uint64_t>(dim_x) * dim_y * sizeof(float);
            const auto pos = static_cast<std::uint64_t>(first_pos) + slice * first;
            // positioned writes on the descriptor: no shared stream position, so slabs of different device threads can be
            // written concurrently (the reference's fseek + fwrite needs the mutex of src/sink.cpp:79-80)
            const int fd = fileno(h->file);
            const auto* bytes = reinterpret_cast<const char*>(voxels);
            auto left = static_cast<std::uint64_t>(dim_x) * dim_y * dim_z_slab * sizeof(float);
            auto at = pos;
            while(left != 0)
            {
                const auto chunk = left < (std::uint64_t{1} << 30) ? static_cast<std::size_t>(left) : (std::size_t{1} << 30);
                const auto done = ::pwrite(fd, bytes, chunk, static_cast<off_t>(at));
                if(done < 0)
                {
                    if(errno == EINTR)
                        continue;
                    throw std::system_error{errno, std::generic_category(), "ddbvf::write()"};
                }
                if(done == 0) // no progress and no error (a full device can do this): never spin on it
                    throw std::runtime_error{"ddbvf::write(): pwrite made no progress"};
                bytes += done;
                at += static_cast<std::uint64_t>(done);
                left -= static_cast<std::uint64_t>(done);
            }
        }
    }
}

#endif
