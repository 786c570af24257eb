// The reconstruction driver: tasks, per-device loop, device fan-out -- src/task.{h,cpp}, src/main.cpp:79-109,132-169.
//
// What is kept: one task per z-slab (src/task.cpp:38-48), a shared task queue drained by one host thread per device
// (src/main.cpp:157-167), the stage order load -> weight -> filter -> backproject per projection (:98-105), one
// sink shared by all threads.
// What is rebuilt for the GPU:
//   - the per-projection chain is enqueued asynchronously on the device's stream; the host thread meanwhile reads and
//     converts the next HIS frame into a pinned upload slot (slot = pinned host + device frame); the upload runs on a
//     second stream and the compute stream waits for it by event, so file I/O, PCIe upload and GPU work all overlap;
//   - slots form two groups of `batch` frames; a full group is backprojected by one fused launch and guarded by a
//     stream fence, the host fills the other group meanwhile;
//   - per slab only the detector rows it can read are converted, uploaded, weighted and filtered (8f-4,
//     paris_hip_slab_row_band); the slab reaches the file through two pinned chunks (D2H of one overlapping the write
//     of the other) instead of a pinned copy of the whole slab, on a drain thread with a ctx of its own: a device that has
//     another slab to do works on it in a second volume buffer while the finished one is copied down and written;
//   - geometry constants are derived per call and the slab offset is passed per task (Q1, Q2), the source restarts
//     its frame index per task (Q5), slabs are written at their own slice offset (Q4);
//   - slab planning is 64-bit and memory driven (paris_hip_make_subvolume_information) with an optional fixed count.
#ifndef PARIS_AMD_HOST_RECONSTRUCT_H_
#define PARIS_AMD_HOST_RECONSTRUCT_H_

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "sink.h"
#include "source.h"
#include "types.h"

namespace paris
{
    // src/program_options.h:34-50 (the parts the hot path needs)
    struct program_options
    {
        detector_geometry det_geo{};
        bool enable_io = false;
        std::string input_path, output_path, prefix = "vol";
        bool enable_roi = false;
        region_of_interest roi{};
        bool enable_angles = false;
        std::string angle_path;
        std::uint16_t quality = 1;
        int slabs = 0;     // 0: memory driven (reference behaviour); > 0: fixed slab count
        int devices = 0;   // 0: all
        int slots = 4;     // pinned upload buffers per device
        bool f16 = false;  // store filtered projections as IEEE half before backprojection (BASELINE config 5)
        int batch = 32;    // projections per fused backprojection launch, up to 64 (1: one launch per projection, as the reference)
        // pinned staging per buffer for the volume's way to the file: small, because pinning costs more than it saves (two
        // 256 MiB buffers added 0.4 s to a 1.3 s reconstruction; 16 MiB chunks still run the copy at full rate)
        std::size_t drain_chunk_bytes = std::size_t{16} << 20;
        int window = PARIS_HIP_WINDOW_RAMP; // filter window (extension: PARIS_HIP_WINDOW_SHEPP_LOGAN; the reference has the ramp only)
        bool row_band = true; // f4: per slab, upload / weight / filter only the detector rows the slab can read
        // several devices: 1 = the device threads of a pass share one read-once frame source (source.h: shared_frames), 0 = each reads
        // the files through a stream of its own (only its slab's detector rows when row_band is on), -1 = decided by how many
        // detector rows the slabs of a pass need between them (run())
        int share_frames = -1;
        bool read_ahead = true;  // frames are read and converted by a feed thread per device, ahead of the device thread
        bool two_volumes = true; // a device with several slabs to do keeps two volume buffers when memory allows (drain thread, below)
        // memory-driven split that gives every device one slab of at least 1 GiB: cut each into this many (0: leave it), so that all
        // but the last of a device's slabs go to the file while the next one is reconstructed
        int pipeline_slabs = 4;
    };

    // src/task.h:33-57
    struct task
    {
        std::uint32_t id, num;
        std::string input_path;
        detector_geometry det_geo;
        volume_geometry vol_geo;
        subvolume_geometry subvol_geo;
        bool enable_roi;
        region_of_interest roi;
        bool enable_angles;
        std::string angle_path;
        std::uint16_t quality;
    };

    // src/task.cpp:33-51
    inline auto make_tasks(const program_options& po, const volume_geometry& vol_geo, const subvolume_info& info) -> std::queue<task>
    {
        auto q = std::queue<task>{};
        for(int i = 0; i < info.num; ++i)
            q.push(task{static_cast<std::uint32_t>(i), static_cast<std::uint32_t>(info.num), po.input_path, po.det_geo, vol_geo,
                        info.geo, po.enable_roi, po.roi, po.enable_angles, po.angle_path, po.quality});
        return q;
    }

    // the part of GLADOS' task_queue the driver uses: thread-safe pop / empty
    class task_queue
    {
    public:
        explicit task_queue(std::queue<task> q) : q_{std::move(q)} {}
        auto pop(task& out) -> bool
        {
            std::lock_guard<std::mutex> lock{m_};
            if(q_.empty())
                return false;
            out = q_.front();
            q_.pop();
            return true;
        }
    private:
        std::queue<task> q_;
        std::mutex m_;
    };

    // One read-once frame source (source.h: shared_frames) per pass of the task queue over the devices: tasks
    // [p * n_dev, (p + 1) * n_dev) are popped at about the same time by the n_dev device threads and all need every frame,
    // so the first thread to reach a frame reads it and the others copy their detector band from memory. A thread that
    // arrives after the others have finished the pass gets a fresh source (= its own read of the files, as before).
    class frame_pool
    {
    public:
        frame_pool(int n_dev, std::uint32_t n_row, std::uint32_t n_col) : n_dev_{n_dev < 1 ? 1 : n_dev}, n_row_{n_row}, n_col_{n_col} {}

        auto get(const struct task& t) -> std::shared_ptr<shared_frames>;

        struct totals
        {
            std::uint64_t produced = 0, served = 0, reread = 0;
        };
        auto stats() -> totals
        {
            std::lock_guard<std::mutex> lock{m_};
            auto sum = done_;
            for(auto& kv : live_)
                if(auto s = kv.second.lock())
                {
                    const auto c = s->stats();
                    sum.produced += c.produced;
                    sum.served += c.served;
                    sum.reread += c.reread;
                }
            return sum;
        }
        auto skipped_files() -> std::vector<std::string>
        {
            std::lock_guard<std::mutex> lock{m_};
            return skipped_;
        }
        // called by the last user of a pass's source (shared_ptr deleter)
        void retire(shared_frames* s)
        {
            {
                std::lock_guard<std::mutex> lock{m_};
                const auto c = s->stats();
                done_.produced += c.produced;
                done_.served += c.served;
                done_.reread += c.reread;
                if(skipped_.empty())
                    skipped_ = s->skipped_files();
            }
            delete s;
        }

    private:
        int n_dev_;
        std::uint32_t n_row_, n_col_;
        std::mutex m_;
        std::map<std::uint32_t, std::weak_ptr<shared_frames>> live_;
        totals done_;
        std::vector<std::string> skipped_;
    };

    struct device_report
    {
        int device = 0;
        std::uint32_t tasks = 0, projections = 0;
        std::uint64_t band_rows = 0; // sum over tasks of the detector rows processed per projection (f4)
        double source_s = 0, enqueue_s = 0, drain_s = 0, save_s = 0;
        double setup_s = 0;       // ctx, pinned slots, device frames: before the first task is popped
        double source_wait_s = 0; // the device thread waiting for the feed thread's next frame (read_ahead)
        double drain_wait_s = 0; // the device thread waiting for the drain thread (a volume buffer to come free; the end of the run)
        bool two_volumes = false; // a second slab buffer was in use: slab k went to the file while slab k + 1 was reconstructed
        std::vector<std::string> skipped;
    };

    inline auto frame_pool::get(const task& t) -> std::shared_ptr<shared_frames>
    {
        const auto pass = t.id / static_cast<std::uint32_t>(n_dev_);
        std::lock_guard<std::mutex> lock{m_};
        auto& slot = live_[pass];
        if(auto s = slot.lock())
            return s;
        // ring depth: how far the device threads may drift apart before the slower one re-reads; at most ~1 GiB of frames
        const auto frame_bytes = static_cast<std::size_t>(n_row_) * n_col_ * sizeof(float);
        const auto capacity = std::max<std::size_t>(4, std::min<std::size_t>(32, (std::size_t{1} << 30) / std::max<std::size_t>(1, frame_bytes)));
        auto s = std::shared_ptr<shared_frames>{new shared_frames{t.input_path, t.enable_angles, t.angle_path, t.quality, n_row_, n_col_, capacity},
                                                [this](shared_frames* p) { retire(p); }};
        slot = s;
        return s;
    }

    namespace detail
    {
        inline void rt(int rc, const char* what)
        {
            if(rc != PARIS_HIP_SUCCESS)
                throw stage_runtime_error{std::string{what} + " failed: " + paris_hip_strerror(rc)};
        }
        using clock = std::chrono::steady_clock;
        inline double since(clock::time_point t) { return std::chrono::duration<double>(clock::now() - t).count(); }

        // frames per group and groups, as reconstruct() lays its slots out
        inline auto slot_shape(const program_options& po) -> std::pair<std::uint32_t, std::uint32_t>
        {
            const std::uint32_t batch = po.batch < 1 ? 1u : static_cast<std::uint32_t>(po.batch > 64 ? 64 : po.batch);
            const std::uint32_t groups = batch == 1u ? static_cast<std::uint32_t>(po.slots < 1 ? 1 : po.slots) : 2u;
            return {batch, groups};
        }

        // device memory reconstruct() keeps per device beside the slab: every slot's fp32 frame (rows padded to 256 B), its
        // half copy with --f16, and two drain chunks' worth of slack for the runtime's own staging
        inline auto driver_bytes(const program_options& po) -> std::size_t
        {
            const auto shape = slot_shape(po);
            const auto slots = static_cast<std::size_t>(shape.first) * shape.second;
            const auto row = (static_cast<std::size_t>(po.det_geo.n_row) * sizeof(float) + 255u) / 256u * 256u;
            const auto half_row = (static_cast<std::size_t>(po.det_geo.n_row) * 2u + 255u) / 256u * 256u;
            return slots * po.det_geo.n_col * (row + (po.f16 ? half_row : 0u)) + 2u * po.drain_chunk_bytes;
        }

        // Large detectors: halve the frames per group until the driver's buffers take at most a quarter of the smallest
        // device's free memory (8192^2 frames: 64 slots would be 16 GiB of device frames plus as much pinned host memory)
        inline auto fit_batch(program_options po, int n_dev) -> program_options
        {
            std::size_t smallest = ~std::size_t{0};
            for(int d = 0; d < n_dev; ++d)
            {
                std::size_t free_b = 0, total_b = 0;
                rt(paris_hip_device_memory(d, &free_b, &total_b), "device memory");
                smallest = std::min(smallest, free_b);
            }
            while(po.batch > 1 && driver_bytes(po) > smallest / 4u)
                po.batch /= 2;
            return po;
        }
    }

    namespace detail
    {
        // A finished slab's way to the file (src/sink.cpp:72-82), on a thread and a ctx of its own: waits for the fence the device
        // thread recorded behind the slab's last kernel, copies the slab down in chunks of whole slices through two pinned buffers
        // (D2H of chunk k + 1 on the drain ctx's stream overlaps the file write of chunk k) and writes them at the slab's slice
        // offset. The device thread meanwhile reconstructs its next slab in another volume buffer; it only waits when it wants a
        // buffer back (wait) or is done (finish). An error in here is rethrown in the device thread by the next wait / finish.
        class volume_drain
        {
        public:
            struct job
            {
                const float* d_v = nullptr;
                std::uint32_t dim_x = 0, dim_y = 0, dim_z = 0, first = 0; // `first`: global slice of the slab's slice 0
                paris_hip_fence* ready = nullptr;                          // recorded by the device thread on its own ctx
                int buffer = 0;                                            // which of the device thread's volume buffers d_v is
            };

            volume_drain(int device, sink& out, std::size_t chunk_bytes) : device_{device}, out_(out), chunk_bytes_{chunk_bytes}
            {
                thread_ = std::thread{[this] { work(); }};
            }
            volume_drain(const volume_drain&) = delete;
            auto operator=(const volume_drain&) -> volume_drain& = delete;
            ~volume_drain()
            {
                {
                    std::lock_guard<std::mutex> lock{m_};
                    quit_ = true;
                }
                cv_.notify_all();
                if(thread_.joinable())
                    thread_.join();
            }

            void submit(const job& j)
            {
                {
                    std::lock_guard<std::mutex> lock{m_};
                    if(error_)
                        std::rethrow_exception(error_);
                    jobs_.push_back(j);
                    ++busy_[j.buffer & 1];
                }
                cv_.notify_all();
            }
            // until no submitted slab lives in volume buffer `buffer` any more
            void wait(int buffer)
            {
                std::unique_lock<std::mutex> lock{m_};
                cv_.wait(lock, [&] { return error_ || busy_[buffer & 1] == 0; });
                if(error_)
                    std::rethrow_exception(error_);
            }
            void finish()
            {
                wait(0);
                wait(1);
            }
            auto drain_seconds() -> double { std::lock_guard<std::mutex> lock{m_}; return drain_s_; }
            auto save_seconds() -> double { std::lock_guard<std::mutex> lock{m_}; return save_s_; }

        private:
            void work()
            {
                paris_hip_ctx* ctx = nullptr;
                float* h_stage[2] = {nullptr, nullptr};
                paris_hip_fence* stage_fence[2] = {nullptr, nullptr};
                std::size_t stage_floats = 0;
                try
                {
                    rt(paris_hip_ctx_create(device_, nullptr, PARIS_HIP_CTX_DEFAULT, &ctx), "set_device()");
                    for(;;)
                    {
                        job j;
                        {
                            std::unique_lock<std::mutex> lock{m_};
                            cv_.wait(lock, [&] { return quit_ || !jobs_.empty(); });
                            if(jobs_.empty())
                                break;
                            j = jobs_.front();
                            jobs_.pop_front();
                        }
                        double drain_s = 0, save_s = 0;
                        auto t0 = clock::now();
                        rt(paris_hip_fence_wait(ctx, j.ready), "fence wait"); // the slab's last kernel has run
                        // the host side of the slab is two pinned chunks of whole slices (the reference allocates the whole slab:
                        // src/sink.cpp:76); pinning gigabytes costs more than the reconstruction of a small data set
                        const auto slice_floats = static_cast<std::size_t>(j.dim_x) * j.dim_y;
                        const auto voxels = slice_floats * j.dim_z;
                        const auto want_floats = std::max(slice_floats, std::min(voxels, chunk_bytes_ / sizeof(float)) / slice_floats * slice_floats);
                        if(want_floats > stage_floats)
                        {
                            for(int s = 0; s < 2; ++s)
                            {
                                rt(paris_hip_free_host(ctx, h_stage[s]), "free");
                                h_stage[s] = nullptr;
                                void* p = nullptr;
                                rt(paris_hip_malloc_host(ctx, want_floats * sizeof(float), &p), "make_volume_host()");
                                h_stage[s] = static_cast<float*>(p);
                                if(stage_fence[s] == nullptr)
                                    rt(paris_hip_fence_create(ctx, &stage_fence[s]), "fence");
                            }
                            stage_floats = want_floats;
                        }
                        const auto chunk_z = static_cast<std::uint32_t>(stage_floats / slice_floats);
                        const auto n_chunks = (j.dim_z + chunk_z - 1u) / chunk_z;
                        const auto copy_down = [&](std::uint32_t c) {
                            const auto z0 = c * chunk_z;
                            const auto nz = std::min(chunk_z, j.dim_z - z0);
                            rt(paris_hip_memcpy_volume_d2h(ctx, h_stage[c % 2u], j.d_v + static_cast<std::size_t>(z0) * slice_floats, j.dim_x, j.dim_y, nz),
                               "copy_d2h()");
                            rt(paris_hip_fence_record(ctx, stage_fence[c % 2u]), "fence record");
                        };
                        copy_down(0);
                        drain_s += since(t0);
                        for(std::uint32_t c = 0; c < n_chunks; ++c)
                        {
                            t0 = clock::now();
                            if(c + 1u < n_chunks)
                                copy_down(c + 1u); // its buffer was written to the file in the previous iteration
                            rt(paris_hip_fence_wait(ctx, stage_fence[c % 2u]), "fence wait");
                            drain_s += since(t0);
                            t0 = clock::now();
                            const auto z0 = c * chunk_z;
                            out_.save(h_stage[c % 2u], j.dim_x, j.dim_y, std::min(chunk_z, j.dim_z - z0), j.first + z0); // :107
                            save_s += since(t0);
                        }
                        {
                            std::lock_guard<std::mutex> lock{m_};
                            --busy_[j.buffer & 1];
                            drain_s_ += drain_s;
                            save_s_ += save_s;
                        }
                        cv_.notify_all();
                    }
                }
                catch(...)
                {
                    std::lock_guard<std::mutex> lock{m_};
                    error_ = std::current_exception();
                    jobs_.clear();
                    cv_.notify_all();
                }
                for(int s = 0; s < 2; ++s)
                {
                    paris_hip_fence_destroy(ctx, stage_fence[s]);
                    paris_hip_free_host(ctx, h_stage[s]);
                }
                paris_hip_ctx_destroy(ctx);
            }

            int device_;
            sink& out_;
            std::size_t chunk_bytes_;
            std::mutex m_;
            std::condition_variable cv_;
            std::deque<job> jobs_;
            int busy_[2] = {0, 0};
            bool quit_ = false;
            std::exception_ptr error_;
            double drain_s_ = 0, save_s_ = 0;
            std::thread thread_; // last: started when everything above exists
        };
    }

    namespace detail
    {
        // The frames of one task, read and converted ahead of the device thread on a thread of their own, straight into the pinned
        // upload slots (src/main.cpp:100 `load`, taken off the device thread: converting a frame costs about as much host time as
        // enqueueing its upload, filter and share of a launch, and the two now overlap). The slots are filled in order, group after
        // group; before it overwrites a group's slots the reader waits until the device thread has enqueued the launch that last used
        // them (flushed) and for that launch's fence (paris_hip_fence_wait touches no ctx state: any thread may call it). The device
        // thread takes the frames in the same order. An error while reading is rethrown by take().
        class frame_feed
        {
        public:
            frame_feed(paris_hip_ctx* ctx, std::function<frame_info(float*)> next, const std::vector<float*>& slots, std::uint32_t batch,
                       const std::vector<paris_hip_fence*>& fences)
            : ctx_{ctx}, next_{std::move(next)}, slots_(slots), batch_{batch}, fences_(fences)
            {
                thread_ = std::thread{[this] { work(); }};
            }
            frame_feed(const frame_feed&) = delete;
            auto operator=(const frame_feed&) -> frame_feed& = delete;
            ~frame_feed()
            {
                {
                    std::lock_guard<std::mutex> lock{m_};
                    quit_ = true;
                }
                cv_.notify_all();
                if(thread_.joinable())
                    thread_.join();
            }

            // the next frame's metadata; its pixels are in slot (taken so far) % slots. !valid(): the set is exhausted
            auto take() -> frame_info
            {
                std::unique_lock<std::mutex> lock{m_};
                cv_.wait(lock, [&] { return error_ || !ready_.empty(); });
                if(ready_.empty())
                    std::rethrow_exception(error_);
                const auto info = ready_.front();
                ready_.pop_front();
                return info;
            }
            // the device thread has enqueued (and fenced) the launch over one more group of slots
            void flushed()
            {
                {
                    std::lock_guard<std::mutex> lock{m_};
                    ++flushed_;
                }
                cv_.notify_all();
            }
            auto read_seconds() -> double { std::lock_guard<std::mutex> lock{m_}; return read_s_; }

        private:
            void work()
            {
                try
                {
                    const auto groups = static_cast<std::uint64_t>(fences_.size());
                    for(std::uint64_t round = 0;; ++round)
                    {
                        {
                            std::unique_lock<std::mutex> lock{m_};
                            cv_.wait(lock, [&] { return quit_ || round < groups || flushed_ + groups > round; });
                            if(quit_)
                                return;
                        }
                        const auto g = static_cast<std::size_t>(round % groups);
                        rt(paris_hip_fence_wait(ctx_, fences_[g]), "fence wait"); // whatever last used this group's slots is done
                        for(std::uint32_t i = 0; i < batch_; ++i)
                        {
                            const auto t0 = clock::now();
                            const auto info = next_(slots_[g * batch_ + i]);
                            const auto dt = since(t0);
                            {
                                std::lock_guard<std::mutex> lock{m_};
                                read_s_ += dt;
                                ready_.push_back(info);
                                if(quit_)
                                    return;
                            }
                            cv_.notify_all();
                            if(!info.valid())
                                return;
                        }
                    }
                }
                catch(...)
                {
                    std::lock_guard<std::mutex> lock{m_};
                    error_ = std::current_exception();
                    cv_.notify_all();
                }
            }

            paris_hip_ctx* ctx_;
            std::function<frame_info(float*)> next_;
            std::vector<float*> slots_;
            std::uint32_t batch_;
            std::vector<paris_hip_fence*> fences_;
            std::mutex m_;
            std::condition_variable cv_;
            std::deque<frame_info> ready_;
            std::uint64_t flushed_ = 0;
            bool quit_ = false;
            std::exception_ptr error_;
            double read_s_ = 0;
            std::thread thread_; // last: started when everything above exists
        };
    }

    // src/main.cpp:79-109, pipelined
    inline auto reconstruct(task_queue& queue, int device, sink& out, const program_options& po, frame_pool* pool = nullptr) -> device_report
    {
        using namespace detail;
        auto rep = device_report{};
        rep.device = device;
        const auto t_setup = clock::now();
        paris_hip_ctx* ctx = nullptr;
        rt(paris_hip_ctx_create(device, nullptr, PARIS_HIP_CTX_DEFAULT, &ctx), "set_device()"); // :87

        // Projections travel in groups: a group of `batch` frames is converted, uploaded, weighted and filtered one by one,
        // then backprojected with ONE fused launch (paris_hip_backproject_batch[_f16]: bit-identical to the sequence, the slab
        // is read and written once per group). While the GPU works on one group the host fills the next. batch = 1 is the
        // reference's one launch per projection with `slots` single-frame groups.
        const std::uint32_t batch = slot_shape(po).first, groups = slot_shape(po).second;
        rt(paris_hip_set_filter_window(ctx, po.window), "filter window");
        const int slots = static_cast<int>(batch * groups);
        const auto n_row = po.det_geo.n_row, n_col = po.det_geo.n_col;
        const auto frame_bytes = static_cast<std::size_t>(n_row) * n_col * sizeof(float);
        auto h_buf = std::vector<float*>(slots, nullptr);
        auto d_buf = std::vector<float*>(slots, nullptr);
        auto fence = std::vector<paris_hip_fence*>(groups, nullptr);
        float* d_all = nullptr; // the device frames of all slots, one under the other: slot s starts at row s * n_col
        std::size_t d_pitch = 0, h16_pitch = 0, h16_stride = 0;
        std::uint16_t* d_half = nullptr;
        // Two volume buffers when the device has room for them: the slab just finished is copied down and written by the drain
        // thread while the next one is reconstructed in the other buffer. The second buffer is allocated when (and if) this device
        // thread pops a second task and a slab's worth of memory is still free.
        float* d_vol[2] = {nullptr, nullptr};
        std::size_t v_cap[2] = {0, 0};
        paris_hip_fence* slab_done[2] = {nullptr, nullptr};
        std::unique_ptr<volume_drain> drain;
        auto cleanup = [&] {
            drain.reset(); // joins the drain thread: nothing reads the volumes or the fences after this
            for(auto f : fence)
                paris_hip_fence_destroy(ctx, f);
            for(int s = 0; s < slots; ++s)
                paris_hip_free_host(ctx, h_buf[s]);
            paris_hip_free(ctx, d_all);
            paris_hip_free(ctx, d_half);
            for(int s = 0; s < 2; ++s)
            {
                paris_hip_free(ctx, d_vol[s]);
                paris_hip_fence_destroy(ctx, slab_done[s]);
            }
            paris_hip_ctx_destroy(ctx);
        };
        try
        {
            if(static_cast<std::uint64_t>(n_col) * static_cast<std::uint64_t>(slots) > 0xffffffffull)
                throw stage_construction_error{"too many projection slots for this detector"};
            rt(paris_hip_malloc_projection(ctx, n_row, n_col * static_cast<std::uint32_t>(slots), &d_all, &d_pitch), "make_projection_device()");
            const auto d_stride = d_pitch * n_col; // bytes from one slot's frame to the next
            for(int s = 0; s < slots; ++s)
            {
                void* p = nullptr;
                rt(paris_hip_malloc_host(ctx, frame_bytes, &p), "make_projection_host()");
                h_buf[s] = static_cast<float*>(p);
                d_buf[s] = reinterpret_cast<float*>(reinterpret_cast<char*>(d_all) + d_stride * static_cast<std::size_t>(s));
            }
            for(auto& f : fence)
                rt(paris_hip_fence_create(ctx, &f), "fence");
            if(po.f16)
            {
                float* raw = nullptr;
                h16_pitch = (static_cast<std::size_t>(n_row) * 2 + 255) / 256 * 256;
                std::size_t got = 0;
                // the half copies of all slots, one under the other like d_all
                rt(paris_hip_malloc_projection(ctx, static_cast<std::uint32_t>(h16_pitch / 4), n_col * static_cast<std::uint32_t>(slots), &raw, &got),
                   "half projection");
                d_half = reinterpret_cast<std::uint16_t*>(raw);
                h16_pitch = got;
                h16_stride = h16_pitch * n_col;
            }

            for(auto& f : slab_done)
                rt(paris_hip_fence_create(ctx, &f), "fence");
            drain.reset(new volume_drain{device, out, po.drain_chunk_bytes});
            rep.setup_s = since(t_setup);
            task t{};
            bool two = false, decided = false;
            while(queue.pop(t)) // :89-91
            {
                const bool last = (t.num - t.id) <= 1;                                    // :92
                const auto dim_z = t.subvol_geo.dim_z + (last ? t.subvol_geo.remainder : 0u); // make_volume: src/make_volume.cpp:32-34
                const auto offset = t.id * t.subvol_geo.dim_z;                             // :96
                const auto voxels = static_cast<std::size_t>(t.subvol_geo.dim_x) * t.subvol_geo.dim_y * dim_z;
                if(rep.tasks == 1u && !decided) // a second slab for this device: is there room to keep the first while it is written?
                {
                    std::size_t free_b = 0, total_b = 0;
                    rt(paris_hip_device_memory(device, &free_b, &total_b), "device memory");
                    // a slab the size of the largest of the run (the last one carries the remainder) plus slack for the runtime
                    const auto need = static_cast<std::size_t>(t.subvol_geo.dim_x) * t.subvol_geo.dim_y * (t.subvol_geo.dim_z + t.subvol_geo.remainder) * sizeof(float);
                    two = po.two_volumes && free_b > need + need / 8u + (std::size_t{256} << 20);
                    decided = true;
                    rep.two_volumes = two;
                }
                const int b = two ? static_cast<int>(rep.tasks & 1u) : 0;
                auto t0 = clock::now();
                drain->wait(b); // the slab that lived in this buffer is in the file
                rep.drain_wait_s += since(t0);
                float*& d_v = d_vol[b];
                if(voxels > v_cap[b]) // slabs of one run have (almost) the same size: allocate once, re-zero per task
                {
                    rt(paris_hip_free(ctx, d_v), "free");
                    d_v = nullptr;
                    v_cap[b] = 0;
                    rt(paris_hip_malloc_volume(ctx, t.subvol_geo.dim_x, t.subvol_geo.dim_y, dim_z, &d_v), "make_volume()");
                    v_cap[b] = voxels;
                }
                else
                    rt(paris_hip_memset_volume(ctx, d_v, t.subvol_geo.dim_x, t.subvol_geo.dim_y, dim_z), "make_volume()");

                std::uint32_t band_first = 0, band_count = n_col;
                if(po.row_band)
                    rt(paris_hip_slab_row_band(&t.det_geo, &t.vol_geo, t.subvol_geo.dim_x, t.subvol_geo.dim_y, dim_z, offset, t.enable_roi,
                                               &t.roi, &band_first, &band_count), "slab_row_band()");
                rep.band_rows += band_count;
                const auto row_bytes = static_cast<std::size_t>(n_row) * sizeof(float);

                t0 = clock::now();
                // :93 (index restarts per task). Several devices: the frames come from the pass's read-once source
                auto shared = pool != nullptr ? pool->get(t) : std::shared_ptr<shared_frames>{};
                auto cur = shared_frames::cursor{};
                auto own = std::unique_ptr<frame_stream>{};
                if(!shared)
                    own.reset(new frame_stream{t.input_path, t.enable_angles, t.angle_path, t.quality});
                rep.source_s += since(t0);
                std::uint32_t group = 0, filled = 0; // frames of the current group already enqueued
                auto feed = std::unique_ptr<frame_feed>{};
                auto sines = std::vector<float>(batch), cosines = std::vector<float>(batch);
                const float delta_s = t.det_geo.delta_s * t.det_geo.l_px_row, delta_t = t.det_geo.delta_t * t.det_geo.l_px_col; // src/backprojection.cpp:49-50
                // frames of up to 1024 x 1024 are weighted and filtered group by group (one launch for up to `batch` frames when the
                // group is flushed) instead of frame by frame behind each upload: their launches are mostly latency
                const bool filter_by_group = batch > 1u && static_cast<std::uint64_t>(n_row) * n_col <= (1ull << 20);
                const auto flush = [&] { // one fused launch for the frames of the current group, then on to the other group
                    if(filled == 0)
                        return;
                    const auto t1 = clock::now();
                    if(filter_by_group && band_count != 0)
                        rt(paris_hip_stage_weight_filter_batch(ctx, d_buf[group * batch], d_pitch, d_stride, filled, n_row, n_col, band_first, band_count,
                                                               &t.det_geo,
                                                               po.f16 ? reinterpret_cast<std::uint16_t*>(reinterpret_cast<char*>(d_half) + h16_stride * group * batch) : nullptr,
                                                               h16_pitch, h16_stride), "weight() + filter()");
                    if(po.f16)
                        rt(paris_hip_backproject_batch_f16(ctx, reinterpret_cast<const std::uint16_t*>(reinterpret_cast<const char*>(d_half) + h16_stride * group * batch),
                                                           h16_pitch, h16_stride, filled, n_row, n_col, d_v, t.subvol_geo.dim_x, t.subvol_geo.dim_y, dim_z,
                                                           offset, &t.det_geo, &t.vol_geo, t.enable_roi, &t.roi, sines.data(), cosines.data(), delta_s,
                                                           delta_t), "backproject()");
                    else
                        rt(paris_hip_backproject_batch(ctx, d_buf[group * batch], d_pitch, d_stride, filled, n_row, n_col, d_v, t.subvol_geo.dim_x,
                                                       t.subvol_geo.dim_y, dim_z, offset, &t.det_geo, &t.vol_geo, t.enable_roi, &t.roi, sines.data(),
                                                       cosines.data(), delta_s, delta_t), "backproject()"); // :104
                    rt(paris_hip_fence_record(ctx, fence[group]), "fence record"); // the group's slots are free once this has run
                    rep.enqueue_s += since(t1);
                    group = (group + 1u) % groups;
                    filled = 0;
                    if(feed)
                        feed->flushed();
                };
                // f4: only the detector rows this slab can read are converted, uploaded, weighted and filtered; the
                // buffers keep their full size, rows outside the band are never read for a voxel of the slab
                const auto next_frame = [&](float* dst) {
                    return shared ? shared->next(cur, dst, n_row, n_col, band_first, band_count)
                                  : own->next(dst, n_row, n_col, band_first, band_count); // :100, straight into pinned memory
                };
                if(po.read_ahead)
                    feed.reset(new frame_feed{ctx, next_frame, h_buf, batch, fence});
                for(;;) // :98
                {
                    const int slot = static_cast<int>(group * batch + filled);
                    auto p = frame_info{};
                    if(feed)
                    {
                        t0 = clock::now();
                        p = feed->take(); // read by the feed thread, which also waited for the slot to be free
                        rep.source_wait_s += since(t0);
                    }
                    else
                    {
                        t0 = clock::now();
                        if(filled == 0)
                            rt(paris_hip_fence_wait(ctx, fence[group]), "fence wait"); // everything that last used this group's slots is done
                        rep.enqueue_s += since(t0);
                        t0 = clock::now();
                        p = next_frame(h_buf[slot]);
                        rep.source_s += since(t0);
                    }
                    if(!p.valid())
                        break;
                    if(p.dim_x != n_row || p.dim_y != n_col)
                        throw stage_runtime_error{"projection size does not match the detector geometry"};
                    t0 = clock::now();
                    const auto band_off = static_cast<std::size_t>(band_first) * n_row;
                    auto* d_band = reinterpret_cast<float*>(reinterpret_cast<char*>(d_buf[slot]) + static_cast<std::size_t>(band_first) * d_pitch);
                    if(band_count != 0)
                    {
                        // :101 -- on the upload stream, overlapping the kernels of the previous projections
                        rt(paris_hip_upload_projection(ctx, d_band, d_pitch, h_buf[slot] + band_off, row_bytes, n_row, band_count), "load()");
                        // :102-103 in one launch: the weight rides along in the row filter's load; with --f16 (BASELINE config 5) the
                        // filtered band is stored as IEEE half straight into the slot's half frame
                        // (small frames: the whole group by one launch when it is flushed -- a launch per 512^2 frame is mostly latency)
                        auto* half_frame = po.f16 ? reinterpret_cast<std::uint16_t*>(reinterpret_cast<char*>(d_half) + h16_stride * static_cast<std::size_t>(slot)) : nullptr;
                        if(!filter_by_group)
                            rt(paris_hip_stage_weight_filter_rows(ctx, d_buf[slot], d_pitch, n_row, n_col, band_first, band_count, &t.det_geo, half_frame,
                                                                  h16_pitch), "weight() + filter()");
                    }
                    rt(paris_hip_stage_angle(&t.det_geo, p.idx, t.enable_angles, p.phi, &sines[filled], &cosines[filled]), "angle"); // src/backprojection.cpp:52-63
                    rep.enqueue_s += since(t0);
                    if(++filled == batch)
                        flush();
                    ++rep.projections;
                }
                flush(); // the last, possibly partial group
                if(feed)
                {
                    rep.source_s += feed->read_seconds();
                    feed.reset(); // joins the feed thread: the source objects are this thread's again
                }
                for(const auto& s : (shared ? cur.skipped_files() : own->skipped_files())) // the shared source's list: run_report
                    rep.skipped.push_back(s);

                // src/sink.cpp:76-82: the slab is complete behind this fence; the drain thread takes it from there
                rt(paris_hip_fence_record(ctx, slab_done[b]), "fence record");
                auto j = volume_drain::job{};
                j.d_v = d_v;
                j.dim_x = t.subvol_geo.dim_x;
                j.dim_y = t.subvol_geo.dim_y;
                j.dim_z = dim_z;
                j.first = offset;
                j.ready = slab_done[b];
                j.buffer = b;
                drain->submit(j);
                ++rep.tasks;
            }
            const auto t_end = clock::now();
            drain->finish();
            rep.drain_wait_s += since(t_end);
            rep.drain_s = drain->drain_seconds();
            rep.save_s = drain->save_seconds();
        }
        catch(...)
        {
            cleanup();
            throw;
        }
        cleanup();
        return rep;
    }

    struct run_report
    {
        volume_geometry vol_geo{}, roi_geo{};
        subvolume_info info{};
        std::vector<device_report> devices;
        std::uint64_t frames_read = 0, frames_requested = 0; // several devices: frames converted from the files / frames handed to device threads
        bool shared_source = false;   // several devices: the read-once shared frame source was used
        double rows_per_pass = 0.0;   // several devices: detector rows the slabs of one pass need between them, in detectors
        std::vector<std::string> skipped;                  // several devices: invalid files skipped by the shared source
        int batch = 0; // frames per fused launch actually used (program_options::batch, halved until the slots fit the devices)
        double wall_s = 0;
        std::string output_file;
    };

    // src/main.cpp:120-178
    inline auto run(const program_options& requested) -> run_report
    {
        auto po = requested; // batch may shrink to fit the devices' memory (detail::fit_batch)
        auto r = run_report{};
        const auto start = detail::clock::now();
        r.vol_geo = calculate_volume_geometry(po.det_geo); // :122
        r.roi_geo = r.vol_geo;
        if(po.enable_roi)                                  // :124-130
            r.roi_geo = apply_roi(r.vol_geo, po.roi.x1, po.roi.x2, po.roi.y1, po.roi.y2, po.roi.z1, po.roi.z2);

        int n_dev = 0;
        detail::rt(paris_hip_device_count(&n_dev), "get_devices()"); // :145
        if(n_dev == 0)
            throw stage_construction_error{"no HIP device"};
        if(po.devices > 0 && po.devices < n_dev)
            n_dev = po.devices;
        po = detail::fit_batch(po, n_dev);
        r.batch = po.batch;

        if(po.slabs > 0) // fixed split (src/cuda/subvolume_information.cpp:112-116 with a given count)
        {
            const auto num = static_cast<std::uint32_t>(po.slabs) > r.roi_geo.dim_z ? r.roi_geo.dim_z : static_cast<std::uint32_t>(po.slabs);
            r.info.num = static_cast<int>(num);
            r.info.geo = {r.roi_geo.dim_x, r.roi_geo.dim_y, r.roi_geo.dim_z / num, r.roi_geo.dim_z % num};
        }
        else // :137, with the driver's own per-device buffers charged next to the slab
        {
            detail::rt(paris_hip_make_subvolume_information_reserving(&r.roi_geo, &po.det_geo, n_dev, detail::driver_bytes(po), &r.info),
                       "make_subvolume_information()");
            // The volume fits (one slab per device): nothing of it could reach the file before the last projection is in. Thinner
            // slabs, several per device, let the drain thread write slab k while slab k + 1 is reconstructed (two volume buffers:
            // they fit where the one big slab did); each slab only reads its own detector rows, so the extra passes over the
            // projections cost little (row band, f4).
            const auto per = static_cast<std::uint32_t>(po.pipeline_slabs > 0 ? po.pipeline_slabs : 0);
            const auto slab_bytes = static_cast<std::uint64_t>(r.info.geo.dim_x) * r.info.geo.dim_y * r.info.geo.dim_z * sizeof(float);
            if(po.two_volumes && po.row_band && per > 1u && r.info.num == n_dev && slab_bytes >= (std::uint64_t{1} << 30)
               && r.info.geo.dim_z >= 32u * per)
            {
                const auto num = static_cast<std::uint32_t>(n_dev) * per;
                r.info.num = static_cast<int>(num);
                r.info.geo = {r.roi_geo.dim_x, r.roi_geo.dim_y, r.roi_geo.dim_z / num, r.roi_geo.dim_z % num};
            }
        }

        task_queue queue{make_tasks(po, r.vol_geo, r.info)}; // :140-141
        sink out{po.output_path, po.prefix, r.roi_geo};           // :154
        r.output_file = out.file_path();

        if(n_dev > 1) // :157-167
        {
            // Where the frames come from. Every device thread needs every projection, but with the row band (f4) only the detector
            // rows its slab can read: z-slabs of one pass over the queue need little more than one detector's worth of rows between
            // them, and a stream per thread that reads and converts just its band does less work per frame than one whole-frame
            // conversion plus a band copy per thread, with nothing shared to wait for (8 bands of a 2048^2 frame: 0.67 against
            // 1.8 ms per frame on 8 cores, tools/shared_source_bench.py). Bands that overlap widely (no row band, thick cones, few
            // rows: the pass would read the files more than twice over) are read once and shared instead.
            const auto first_pass = std::min<std::uint32_t>(static_cast<std::uint32_t>(n_dev), static_cast<std::uint32_t>(r.info.num));
            if(po.row_band && po.det_geo.n_col != 0)
            {
                std::uint64_t rows = 0;
                for(std::uint32_t i = 0; i < first_pass; ++i)
                {
                    const bool last = i + 1u == static_cast<std::uint32_t>(r.info.num);
                    std::uint32_t band_first = 0, band_count = po.det_geo.n_col;
                    detail::rt(paris_hip_slab_row_band(&po.det_geo, &r.vol_geo, r.info.geo.dim_x, r.info.geo.dim_y,
                                                       r.info.geo.dim_z + (last ? r.info.geo.remainder : 0u), i * r.info.geo.dim_z, po.enable_roi,
                                                       &po.roi, &band_first, &band_count), "slab_row_band()");
                    rows += band_count;
                }
                r.rows_per_pass = static_cast<double>(rows) / po.det_geo.n_col;
            }
            else
                r.rows_per_pass = static_cast<double>(first_pass);
            r.shared_source = po.share_frames < 0 ? r.rows_per_pass > 2.0 : po.share_frames != 0;
            frame_pool pool{n_dev, po.det_geo.n_row, po.det_geo.n_col}; // every HIS frame is read once per pass, not once per device
            auto* shared = r.shared_source ? &pool : nullptr;
            auto futures = std::vector<std::future<device_report>>{};
            for(int d = 0; d < n_dev; ++d)
                futures.emplace_back(std::async(std::launch::async, [&queue, &out, &po, shared, d] { return reconstruct(queue, d, out, po, shared); }));
            for(auto& f : futures)
                r.devices.push_back(f.get());
            if(r.shared_source)
            {
                const auto st = pool.stats();
                r.frames_read = st.produced + st.reread;
                r.frames_requested = st.served + st.reread;
                r.skipped = pool.skipped_files();
            }
        }
        else
            r.devices.push_back(reconstruct(queue, 0, out, po)); // :169
        r.wall_s = detail::since(start);
        return r;
    }
}

#endif
