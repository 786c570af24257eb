// Projection source: a directory of HIS files read in sorted path order, an optional angle file, a quality stride.
//
// Follows src/source.cpp:37-134 and src/filesystem.cpp:37-67 with two repairs: the frame counter belongs to the
// source object (the reference's `thread_local static i` keeps counting across tasks, quirk Q5), and running out
// of files while looking for a valid one ends the stream instead of indexing an empty vector.
#ifndef PARIS_AMD_HOST_SOURCE_H_
#define PARIS_AMD_HOST_SOURCE_H_

#include <algorithm>
#include <cctype>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <fstream>
#include <iterator>
#include <memory>
#include <mutex>
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

    // Read-once frame source shared by the device threads of one run. Every device needs every projection for its slab;
    // the reference lets each device thread read the whole set again (src/main.cpp:93: a source per task). Here every kept
    // frame is read and converted once, into a buffer of a small ring, by whichever thread gets to it first; the other
    // threads copy the rows they need (their slab's detector band) from that buffer. Frames are claimed in order, one
    // thread each, and several are in production at a time: a thread whose own frame is still being read by another one
    // does not sleep, it claims and reads the next unclaimed frame ahead (up to half a ring ahead of its own position), so
    // N device threads convert N frames side by side instead of queueing behind one reader. Every producing thread reads
    // through a stream of its own (the cursor's), stepping over the frames other threads produce without converting them.
    // Correctness never depends on the ring: a consumer whose frame has already been recycled (it lags more than `capacity`
    // frames behind, or it starts a later task) falls back to a frame_stream of its own, exactly the single-device behaviour.
    // Frames, order, indices, angles and skipped files are frame_stream's.
    class shared_frames
    {
    public:
        struct counters
        {
            std::uint64_t produced = 0; // frames read from the files by this object (each at most once)
            std::uint64_t served = 0;   // next() calls answered from the ring
            std::uint64_t reread = 0;   // next() calls answered by a consumer's own fallback stream
        };

        // per consumer and task: where it is in the sequence, the stream it produces shared frames through, and its
        // fallback stream once it has needed one
        class cursor
        {
            friend class shared_frames;
            std::uint64_t next_ = 0;
            std::unique_ptr<frame_stream> own_;
            std::uint64_t own_pos_ = 0;
            std::unique_ptr<frame_stream> prod_;
            std::uint64_t prod_pos_ = 0;
        public:
            auto skipped_files() const -> std::vector<std::string> { return own_ ? own_->skipped_files() : std::vector<std::string>{}; }
        };

        shared_frames(const std::string& proj_dir, bool enable_angles, const std::string& angle_file, std::uint16_t quality,
                      std::uint32_t dim_x, std::uint32_t dim_y, std::size_t capacity = 32)
        : dir_{proj_dir}, enable_angles_{enable_angles}, angle_file_{angle_file}, quality_{quality}, dim_x_{dim_x}, dim_y_{dim_y},
          ring_(capacity == 0 ? 1 : capacity)
        {}

        // frame_stream::next for the consumer behind `c`
        auto next(cursor& c, float* dst, std::uint32_t dim_x, std::uint32_t dim_y, std::uint32_t row_first, std::uint32_t row_count) -> frame_info
        {
            if(dim_x != dim_x_ || dim_y != dim_y_)
                throw std::runtime_error{"shared_frames::next(): frame size differs from the one the cache was built for"};
            const auto k = c.next_++;
            if(c.own_) // once behind, stay on the private stream: it is already positioned
                return from_own(c, k, dst, row_first, row_count);
            const std::uint64_t ahead = ring_.size() / 2u; // how far beyond its own frame a waiting thread may produce
            std::shared_ptr<const entry> e;
            {
                std::unique_lock<std::mutex> lock{m_};
                for(;;)
                {
                    if(failed_ && k >= fail_at_)
                        std::rethrow_exception(failed_); // reading frame fail_at_ failed: every consumer that gets there reports it
                    if(ended_ && k >= end_)
                        return frame_info{};
                    const auto& slot = ring_[k % ring_.size()];
                    if(slot && slot->ordinal == k)
                    {
                        e = slot;
                        ++stats_.served;
                        break;
                    }
                    if(slot && slot->ordinal > k)
                        break; // recycled: e stays empty
                    // frame k is not there yet. Unclaimed frames are claimed in order; if k itself is already somebody's, this
                    // thread reads the next unclaimed one (within `ahead`) rather than sleep
                    if(claimed_ <= k || (claimed_ <= k + ahead && !(ended_ && claimed_ >= end_) && !(failed_ && claimed_ >= fail_at_)))
                    {
                        const auto j = claimed_++;
                        lock.unlock();
                        std::shared_ptr<const entry> fresh;
                        std::exception_ptr error;
                        try { fresh = produce(c, j); }
                        catch(...) { error = std::current_exception(); }
                        lock.lock();
                        if(error)
                        {
                            // an I/O or allocation error while reading frame j (an angle file shorter than the frame set:
                            // std::out_of_range; bad_alloc): consumers parked in cv_.wait() for it must not wait for ever
                            if(!failed_ || j < fail_at_)
                            {
                                failed_ = error;
                                fail_at_ = j;
                            }
                        }
                        else if(fresh)
                        {
                            auto& target = ring_[j % ring_.size()];
                            if(!target || target->ordinal < j) // (a straggler never replaces a later frame)
                                target = fresh;
                            ++stats_.produced;
                        }
                        else if(!ended_ || j < end_)
                        {
                            ended_ = true; // the stream ends before frame j
                            end_ = j;
                            skipped_ = c.prod_->skipped_files(); // this stream has seen every file
                        }
                        cv_.notify_all();
                        continue;
                    }
                    cv_.wait(lock); // frame k is being read by another thread and there is nothing left to help with
                }
            }
            if(!e)
                return from_own(c, k, dst, row_first, row_count);
            if(e->info.dim_x == dim_x && e->info.dim_y == dim_y && row_count != 0)
                std::memcpy(dst + static_cast<std::size_t>(row_first) * dim_x, e->pixels.data() + static_cast<std::size_t>(row_first) * dim_x,
                            static_cast<std::size_t>(row_count) * dim_x * sizeof(float));
            return e->info;
        }

        // the files the shared streams skipped (complete once a consumer has reached the end of the set)
        auto skipped_files() const -> std::vector<std::string>
        {
            std::lock_guard<std::mutex> lock{m_};
            return skipped_;
        }

        auto stats() const -> counters
        {
            std::lock_guard<std::mutex> lock{m_};
            return stats_;
        }

    private:
        struct entry
        {
            std::uint64_t ordinal = 0;
            frame_info info{};
            std::vector<float> pixels; // the whole frame (consumers want different bands)
        };

        // reads kept frame j through the cursor's producing stream (frames are claimed in rising order, so the stream only ever
        // moves forward); nullptr when the set ends before frame j. Runs without the lock, several threads at a time.
        auto produce(cursor& c, std::uint64_t j) -> std::shared_ptr<const entry>
        {
            if(!c.prod_)
            {
                c.prod_.reset(new frame_stream{dir_, enable_angles_, angle_file_, quality_});
                c.prod_pos_ = 0;
            }
            for(; c.prod_pos_ < j; ++c.prod_pos_) // frames other threads produce are stepped over without conversion
                if(!c.prod_->next(nullptr, dim_x_, dim_y_, 0, 0).valid())
                    return nullptr;
            // frame buffers are recycled: the deleter of an entry hands its pixel vector back (a fresh 16 MiB vector per
            // frame would cost a page-faulting zero fill each time)
            auto* raw = new entry;
            {
                std::lock_guard<std::mutex> lock{spare_m_};
                if(!spare_.empty())
                {
                    raw->pixels = std::move(spare_.back());
                    spare_.pop_back();
                }
            }
            auto e = std::shared_ptr<entry>{raw, [this](entry* d) {
                {
                    std::lock_guard<std::mutex> lock{spare_m_};
                    if(spare_.size() < ring_.size())
                        spare_.push_back(std::move(d->pixels));
                }
                delete d;
            }};
            e->ordinal = j;
            e->pixels.resize(static_cast<std::size_t>(dim_x_) * dim_y_);
            e->info = c.prod_->next(e->pixels.data(), dim_x_, dim_y_, 0, dim_y_);
            ++c.prod_pos_;
            if(!e->info.valid())
                return nullptr;
            return e;
        }

        auto from_own(cursor& c, std::uint64_t k, float* dst, std::uint32_t row_first, std::uint32_t row_count) -> frame_info
        {
            if(!c.own_)
            {
                c.own_.reset(new frame_stream{dir_, enable_angles_, angle_file_, quality_});
                c.own_pos_ = 0;
            }
            for(; c.own_pos_ < k; ++c.own_pos_) // seek: kept frames before k are stepped over without conversion
                if(!c.own_->next(dst, dim_x_, dim_y_, 0, 0).valid())
                    return frame_info{};
            ++c.own_pos_;
            const auto info = c.own_->next(dst, dim_x_, dim_y_, row_first, row_count);
            if(info.valid())
            {
                std::lock_guard<std::mutex> lock{m_};
                ++stats_.reread;
            }
            return info;
        }

        std::string dir_;
        bool enable_angles_;
        std::string angle_file_;
        std::uint16_t quality_;
        std::uint32_t dim_x_, dim_y_;
        mutable std::mutex m_;
        std::condition_variable cv_;
        std::mutex spare_m_;
        std::vector<std::vector<float>> spare_; // declared before ring_: the entries' deleters use it while ring_ is destroyed
        std::vector<std::shared_ptr<const entry>> ring_;
        std::uint64_t claimed_ = 0;             // frames 0 .. claimed_ - 1 have a producer (or are done)
        std::uint64_t end_ = 0, fail_at_ = 0;
        bool ended_ = false;
        std::exception_ptr failed_; // set under m_ when a produce() threw, with the frame it was reading
        std::vector<std::string> skipped_;
        counters stats_;
    };
}

#endif
