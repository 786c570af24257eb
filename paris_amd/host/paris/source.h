// Projection source: a directory of HIS files read in sorted path order, an optional angle file, a quality stride.
//
// Follows src/source.cpp:37-134 and src/filesystem.cpp:37-67 with two repairs: the frame counter belongs to the
// source object (the reference's `thread_local static i` keeps counting across tasks, quirk Q5), and running out
// of files while looking for a valid one ends the stream instead of indexing an empty vector.
#ifndef PARIS_AMD_HOST_SOURCE_H_
#define PARIS_AMD_HOST_SOURCE_H_

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <iterator>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <dirent.h>
#include <limits.h>
#include <sys/stat.h>

#include "his.h"

namespace paris
{
    // src/filesystem.cpp:37-67: canonical paths of every directory entry, sorted
    inline auto read_directory(const std::string& path) -> std::vector<std::string>
    {
        struct stat st{};
        if(::stat(path.c_str(), &st) != 0)
            throw std::runtime_error{path + " does not exist."};
        if(S_ISREG(st.st_mode))
            throw std::runtime_error{path + " is not a directory."};
        if(!S_ISDIR(st.st_mode))
            throw std::runtime_error{path + " exists but is neither a regular file nor a directory."};
        auto ret = std::vector<std::string>{};
        DIR* d = ::opendir(path.c_str());
        if(d == nullptr)
            throw std::runtime_error{path + " could not be read"};
        while(auto* e = ::readdir(d))
        {
            const auto name = std::string{e->d_name};
            if(name == "." || name == "..")
                continue;
            char resolved[PATH_MAX];
            const auto full = path + "/" + name;
            if(::realpath(full.c_str(), resolved) != nullptr)
                ret.emplace_back(resolved);
        }
        ::closedir(d);
        std::sort(ret.begin(), ret.end());
        return ret;
    }

    // src/source.cpp:39-72. Values are whitespace separated; a ',' on the first line selects the decimal comma.
    // The reference's `while(!eof){file >> a; push_back(a);}` appends one extra 0 when the file does not end right
    // after the last digit (quirk Q15); that is reproduced. An unopenable file yields no angles (defaults are used).
    inline auto read_angles(const std::string& path) -> std::vector<float>
    {
        auto angles = std::vector<float>{};
        auto file = std::ifstream{path.c_str()};
        if(!file.is_open())
            return angles;
        auto text = std::string{std::istreambuf_iterator<char>{file}, std::istreambuf_iterator<char>{}};
        const auto first_line = text.substr(0, text.find('\n'));
        if(first_line.find(',') != std::string::npos)
            std::replace(text.begin(), text.end(), ',', '.');
        std::size_t pos = 0;
        while(true)
        {
            while(pos < text.size() && std::isspace(static_cast<unsigned char>(text[pos])))
                ++pos;
            if(pos >= text.size())
                break;
            const char* begin = text.c_str() + pos;
            char* end = nullptr;
            const float a = std::strtof(begin, &end);
            if(end == begin)
                throw std::runtime_error{"read_angles(): malformed value in " + path};
            angles.push_back(a);
            pos += static_cast<std::size_t>(end - begin);
        }
        if(!text.empty() && std::isspace(static_cast<unsigned char>(text.back())))
            angles.push_back(0.f); // the failed final extraction of the reference's loop
        return angles;
    }

    // One frame as the source hands it out (host memory; the driver copies it into its pinned upload buffer)
    struct host_frame
    {
        std::vector<float> pixels;
        std::uint32_t dim_x = 0, dim_y = 0;
        std::uint32_t idx = 0;
        float phi = 0.f;
        bool valid() const noexcept { return dim_x != 0 && dim_y != 0; }
    };

    class source
    {
    public:
        source(const std::string& proj_dir, bool enable_angles = false, const std::string& angle_file = "",
               std::uint16_t quality = 1)
        : enable_angles_{enable_angles}, quality_{quality == 0 ? std::uint16_t{1} : quality}
        {
            paths_ = read_directory(proj_dir); // src/source.cpp:79
            if(enable_angles_)
                angles_ = read_angles(angle_file); // :83-84
            refill();
        }

        auto drained() const noexcept -> bool { return queue_.empty(); }
        auto skipped_files() const noexcept -> const std::vector<std::string>& { return skipped_; }

        // src/source.cpp:88-130
        auto load_next() -> host_frame
        {
            if(queue_.empty())
                return host_frame{};
            auto p = std::move(queue_.front());
            queue_.pop_front();
            if(queue_.empty())
                refill();
            return p;
        }

    private:
        void refill()
        {
            while(queue_.empty() && next_path_ < paths_.size())
            {
                const auto& path = paths_[next_path_++];
                auto frames = std::vector<his::frame>{};
                try { frames = his::load(path); }
                catch(const std::system_error&) { frames.clear(); }
                if(frames.empty())
                {
                    skipped_.push_back(path); // "Skipping invalid file": :96-100
                    continue;
                }
                for(auto& f : frames)
                {
                    if(counter_ % quality_ == 0u) // :105-113: the stride keeps the original index
                    {
                        auto p = host_frame{};
                        p.pixels = std::move(f.pixels);
                        p.dim_x = f.dim_x;
                        p.dim_y = f.dim_y;
                        p.idx = counter_;
                        if(enable_angles_ && !angles_.empty())
                            p.phi = angles_.at(counter_);
                        queue_.push_back(std::move(p));
                    }
                    ++counter_;
                }
            }
        }

        std::vector<std::string> paths_;
        std::size_t next_path_ = 0;
        std::deque<host_frame> queue_;
        std::vector<std::string> skipped_;
        bool enable_angles_;
        std::vector<float> angles_;
        std::uint16_t quality_;
        std::uint32_t counter_ = 0;
    };
    // Metadata of a frame that frame_stream::next() wrote into caller memory
    struct frame_info
    {
        std::uint32_t dim_x = 0, dim_y = 0;
        std::uint32_t idx = 0;
        float phi = 0.f;
        bool valid() const noexcept { return dim_x != 0 && dim_y != 0; }
    };

    // The same frames in the same order, with the same indices, stride and skipped files as `source`, but one at a time
    // and converted straight into memory the caller owns (the driver's pinned upload slot): no per-file frame queue, no
    // extra copy, and only the detector rows the caller asks for are read and converted (f4).
    // Frames dropped by the quality stride are seeked over, not converted.
    class frame_stream
    {
    public:
        frame_stream(const std::string& proj_dir, bool enable_angles = false, const std::string& angle_file = "",
                     std::uint16_t quality = 1)
        : enable_angles_{enable_angles}, quality_{quality == 0 ? std::uint16_t{1} : quality}
        {
            paths_ = read_directory(proj_dir); // src/source.cpp:79
            if(enable_angles_)
                angles_ = read_angles(angle_file); // :83-84
        }

        auto skipped_files() const noexcept -> const std::vector<std::string>& { return skipped_; }

        // Writes rows [row_first, row_first + row_count) of the next kept frame into dst (dim_x * dim_y floats, row
        // stride dim_x) and returns its metadata; !valid() when the directory is exhausted. A frame whose size is not
        // dim_x x dim_y is reported with its own size and nothing is written.
        auto next(float* dst, std::uint32_t dim_x, std::uint32_t dim_y, std::uint32_t row_first, std::uint32_t row_count) -> frame_info
        {
            for(;;)
            {
                if(!reader_)
                {
                    if(next_path_ >= paths_.size())
                        return frame_info{};
                    const auto& path = paths_[next_path_++];
                    try { reader_.reset(new his::reader{path}); }
                    catch(const std::system_error&) { reader_.reset(); }
                    frames_of_file_ = 0;
                    if(!reader_)
                    {
                        skipped_.push_back(path); // "Skipping invalid file": :96-100
                        continue;
                    }
                }
                if(!reader_->advance())
                {
                    if(frames_of_file_ == 0)
                        skipped_.push_back(paths_[next_path_ - 1]);
                    reader_.reset();
                    continue;
                }
                ++frames_of_file_;
                const auto counter = counter_++;
                if(counter % quality_ != 0u) // :105-113: the stride keeps the original index
                    continue;                // advance() seeks over the unread frame
                auto info = frame_info{};
                info.dim_x = reader_->dim_x();
                info.dim_y = reader_->dim_y();
                info.idx = counter;
                if(enable_angles_ && !angles_.empty())
                    info.phi = angles_.at(counter);
                if(info.dim_x == dim_x && info.dim_y == dim_y)
                    reader_->read_rows(dst, row_first, row_count);
                return info;
            }
        }

    private:
        std::vector<std::string> paths_;
        std::size_t next_path_ = 0;
        std::unique_ptr<his::reader> reader_;
        std::uint32_t frames_of_file_ = 0;
        std::vector<std::string> skipped_;
        bool enable_angles_;
        std::vector<float> angles_;
        std::uint16_t quality_;
        std::uint32_t counter_ = 0;
    };
}

#endif
