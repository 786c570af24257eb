// Volume sink: one DDBVF file that every device thread writes its slabs into (src/sink.cpp:39-94).
// Repair of quirk Q4: the slab's first slice is passed explicitly (the reference writes every slab at volume::off,
// which nothing ever sets, so all slabs land on slice 0).
#ifndef PARIS_AMD_HOST_SINK_H_
#define PARIS_AMD_HOST_SINK_H_

#include <string>

#include <sys/stat.h>

#include "ddbvf.h"
#include "types.h"

namespace paris
{
    // src/filesystem.cpp:69-90 (mkdir -p)
    inline auto create_directory(const std::string& path) -> bool
    {
        struct stat st{};
        if(::stat(path.c_str(), &st) == 0)
        {
            if(S_ISDIR(st.st_mode))
                return true;
            throw std::runtime_error{path + " exists but is not a directory."};
        }
        for(std::size_t pos = 1; pos <= path.size(); ++pos)
            if(pos == path.size() || path[pos] == '/')
            {
                const auto sub = path.substr(0, pos);
                if(::mkdir(sub.c_str(), 0777) != 0 && errno != EEXIST)
                    return false;
            }
        return true;
    }

    class sink
    {
    public:
        sink(const std::string& path, const std::string& prefix, const volume_geometry& vol_geo)
        : path_{path}, vol_geo_(vol_geo)
        {
            try
            {
                if(path_.empty() || path_.back() != '/')
                    path_ += '/';
                path_ += prefix;
                if(!create_directory(path))
                    throw stage_construction_error{"sink::sink() failed to create output directory at " + path};
                handle_ = ddbvf::create(path_, vol_geo_.dim_x, vol_geo_.dim_y, vol_geo_.dim_z);
            }
            catch(const std::system_error& se) { throw stage_runtime_error{std::string{"sink::sink() failed: "} + se.what()}; }
            catch(const std::runtime_error& re) { throw stage_construction_error{std::string{"sink::sink() failed: "} + re.what()}; }
        }

        auto file_path() const -> std::string { return path_ + ".ddbvf"; }

        // host voxels of one slab (or a chunk of it) starting at global slice `first`. The reference serialises this with a
        // static mutex (src/sink.cpp:79-81) because its writer moves a shared stream position; ddbvf::write uses positioned
        // writes, so device threads write their disjoint slice ranges concurrently.
        auto save(const float* voxels, std::uint32_t dim_x, std::uint32_t dim_y, std::uint32_t dim_z, std::uint32_t first) -> void
        {
            try { ddbvf::write(handle_, voxels, dim_x, dim_y, dim_z, first); }
            catch(const std::system_error& se) { throw stage_runtime_error{std::string{"sink::save() failed: "} + se.what()}; }
            catch(const std::runtime_error& re) { throw stage_runtime_error{std::string{"sink::save() failed: "} + re.what()}; }
        }

    private:
        std::string path_;
        ddbvf::handle_type handle_;
        volume_geometry vol_geo_;
    };
}

#endif
