// Internal declarations shared by the translation units of libparis_hip.so (gfx950 only).
#ifndef PARIS_HIP_INTERNAL_H_
#define PARIS_HIP_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <array>
#include <cstdint>
#include <map>
#include <vector>

#include "paris_hip.h"

// Every fp32 operation of the numeric path must round once, in the reference's order (DESIGN.md,
// "Numerics"): no FMA contraction anywhere in this library. The Makefile passes -ffp-contract=off as well.
#pragma clang fp contract(off)

#define PARIS_HIP_TRY(expr)                                   \
    do                                                        \
    {                                                         \
        hipError_t paris_hip_err__ = (expr);                  \
        if(paris_hip_err__ != hipSuccess)                     \
            return static_cast<int>(paris_hip_err__);         \
    } while(0)

struct paris_hip_fft_plan
{
    float2* d_twiddle = nullptr; // exp(-2 pi i k / n), k < n/2
    // fused weight + filter kernel (filter_fused.hip), n >= 1024: inter-pass twiddle tables W_B^(o q), rounded from double
    float2* d_tab_first = nullptr;                  // first / last pass, B = n
    float2* d_tab_mid[3] = {nullptr, nullptr, nullptr}; // middle passes with B = 256, 4096, 65536
};

// a filter K made by paris_hip_make_filter*: its length and the copy of K in the order the fused kernel's middle pass reads it
struct paris_hip_filter_info
{
    uint32_t size = 0;
    float* d_kp = nullptr;
};

struct paris_hip_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    unsigned flags = 0;
    // ring of HIP event pairs recorded on the ctx stream around backprojection launches (bench.py's
    // roofline figure); capacity 1 by default, i.e. "the last launch"
    std::vector<hipEvent_t> bp_start, bp_stop;
    uint64_t bp_launches = 0; // launches recorded since the ring was (re)armed
    int bp_variant = 0;
    int bp_vx = 0, bp_unroll = 0; // 0 = automatic
    uint32_t bp_tz = 0, bp_lds_bytes = 0;
    int bp_order = -1; // -1 = default mapping
    int bp_slice_nw = 0, bp_slice_rpl = 0; // slice kernel shape, 0 = default
    int bp_nt = -1;    // volume stream policy: -1 automatic by slab size, 0 plain, 1 nontemporal, 2 nontemporal + write-through stores
    int bp_stage_vec4 = 1; // stage the detector box 4 pixels per lane when the projection's alignment allows
    int bp_fastdiv = 1; // use the validated multiply+2 FMA division by the pixel pitch when it is exact
    int bp_skip_invalid = 1; // waves none of whose columns a projection's rays reach leave their tile untouched (clean volumes only)
    // Volumes known to hold no -0: allocated (zero-filled) or zero-filled again by this library and written by nothing but
    // backprojections since (base address -> bytes). A host upload into one removes it; memory the library did not allocate is
    // never listed.
    std::map<uintptr_t, size_t> clean_volumes;
    // every volume paris_hip_malloc_volume handed out and paris_hip_free has not taken back (base address -> bytes): the only
    // memory paris_hip_ctx_destroy will run still-deferred projections into (a foreign address may have changed hands)
    std::map<uintptr_t, size_t> volume_allocs;
    int bp_lean_div = 1; // share one reciprocal between the two per-column divisions by s + d_so when the operands are in range
    std::map<uint32_t, bool> fastdiv_exact; // divisor bits -> exhaustive check result
    std::map<std::array<uint32_t, 4>, bool> lean_checks; // (kind, operand bits...) -> device validation result (validate.hip)
    int lean_validate = 1; // 0: trust the range tests alone (A/B and tests of the validators themselves)
    bool filter_lds_attr_set = false;
    bool filter_r16_attr_set[5] = {false, false, false, false, false}; // LOG2N 10..14
    int filter_variant = 0; // 0: radix-16 register passes for N >= 1024, 1: radix-2 kernel for every N
    std::map<uint32_t, paris_hip_fft_plan> plans; // keyed by FFT length
    std::map<const float*, paris_hip_filter_info> filters; // K buffers handed out by paris_hip_make_filter*, until paris_hip_free
    // Stage fusion (paris_hip_set_stage_fusion): a paris_hip_weight[_rows] call is held back; the paris_hip_apply_filter call
    // that follows on the same rows runs ONE kernel that weights in its load (filter_fused.hip). Anything else that touches a
    // projection or completes work first runs the held-back weighting as its own kernel (paris_hip_flush_pending_weight).
    int stage_fusion = 0;
    struct pending_weight_t
    {
        bool active = false;
        float* d_p = nullptr;
        size_t pitch = 0;
        uint32_t dim_x = 0, dim_y = 0, row_first = 0, row_count = 0;
        float h_min = 0.f, v_min = 0.f, d_sd = 0.f, l_px_row = 0.f, l_px_col = 0.f;
        // filter deferral (below): the row filter that followed on the same rows is held back too
        bool filter = false;
        const float* d_kp = nullptr;
        const paris_hip_fft_plan* plan = nullptr;
        uint32_t filter_size = 0;
    } pending_weight;
    // Filter deferral (paris_hip_set_filter_deferral; needs stage fusion and a backprojection deferral depth > 1): the
    // paris_hip_apply_filter call that would run the one weight + filter launch is held back as well. If the next call backprojects
    // those rows' projection, the UNFILTERED frame is snapshotted into the deferral ring and weighting + filter run on the ring, one
    // launch for the whole group, right before the fused backprojection; any other call runs the held-back launch first, in place.
    int filter_deferral = 0;
    std::vector<pending_weight_t> defer_wf; // per ring slot of the current group: the weight + filter it still needs
    // dedicated upload stream + ring of events ordering the compute stream behind each upload (paris_hip_upload_projection)
    hipStream_t upload_stream = nullptr;
    std::vector<hipEvent_t> upload_events;
    uint64_t uploads = 0;
    // device buffers that have been the destination of paris_hip_upload_projection: the event marks the last library call
    // enqueued on the compute stream that reads or writes the buffer, so the next upload into it (slot reuse) waits for
    // exactly that work and nothing else. Buffers never uploaded into this way are not tracked (no cost for other callers).
    struct upload_target
    {
        hipEvent_t last_use = nullptr;
        bool used = false;
        size_t bytes = 0; // extent of the buffer as uploaded (pitch x rows): stage calls on a row band pass interior pointers
    };
    std::map<const void*, upload_target> upload_targets;
    // K cached by paris_hip_stage_filter (reference: thread_local static in src/filtering.cpp:42)
    float* stage_k = nullptr;
    uint32_t stage_k_size = 0;
    float stage_k_tau = 0.f;
    int stage_window = 0, stage_k_window = 0; // requested window / window of the cached K
    // Projection-sized buffers are recycled instead of returned to the runtime (the reference's CUDA backend pools its device
    // projection buffers too, src/cuda/memory.cpp:42-44): PARIS allocates and frees one host and one device buffer per
    // projection (src/loader.cpp:28-33), and hipMalloc / hipFree / hipHostMalloc cost more than the kernels of a small frame.
    // A released buffer carries an event recorded behind its LAST USER, on the stream that ran it: for a pinned host buffer the
    // H2D copy that read it (the upload stream when the frame went through paris_hip_upload_projection -- nothing queued on the
    // compute stream holds it back), for a device buffer the compute stream at release time (its last reader -- the filter, the
    // deferral ring's snapshot copy -- was enqueued there just before). A buffer nothing used is free at once (no event). It is
    // handed out again only after that event has completed, so the new owner may touch it from the host or from any stream. While
    // the oldest released buffer is still busy and fewer than pool_capacity(bytes) are parked, a fresh one is allocated instead of
    // waiting: an asynchronous caller ends up rotating through a few buffers and never stalls.
    struct pooled_buffer
    {
        void* ptr;
        hipEvent_t released; // nullptr: free at once
        uint64_t group = 0;  // the fused launch (group_seq) that reads it last, by reference (0: none): free once that group is done
    };
    struct proj_alloc
    {
        size_t bytes = 0;
        bool touched = false; // a library call has read or written it since it was handed out
        uint32_t held = 0;    // slots of the PENDING deferred group that refer to this buffer itself (no snapshot was taken)
        uint64_t group = 0;   // the last LAUNCHED group that reads it by reference
    };
    struct host_alloc
    {
        size_t bytes = 0;
        unsigned used = 0; // streams with copies from / into it since it was handed out (USED_*)
    };
    static constexpr unsigned USED_COMPUTE = 1u, USED_UPLOAD = 2u;
    std::map<void*, proj_alloc> proj_allocs;           // live buffers of paris_hip_malloc_projection
    std::multimap<size_t, pooled_buffer> proj_pool;    // released ones, by size, oldest first (at most pool_capacity(size) per size)
    std::map<void*, host_alloc> host_allocs;           // live pinned buffers of paris_hip_malloc_host up to POOL_HOST_BYTES
    std::multimap<size_t, pooled_buffer> host_pool;
    static constexpr size_t POOL_HOST_BYTES = size_t{64} << 20; // a 4096 x 4096 frame
    // How far the host may run ahead of the device in PINNED buffers of one size: 8 at least, up to 64 MiB worth, 16 at most (a caller
    // that allocates and frees a buffer per projection blocks in paris_hip_malloc_host once that many are parked and busy). A pinned
    // buffer is busy only until its own H2D copy is done, so a few suffice, and pinning costs ~0.25 ms per MiB (64 buffers of a
    // 1024^2 frame: 60 ms of a 500 ms job, measured).
    static constexpr size_t pool_capacity(size_t bytes)
    {
        const size_t by_bytes = bytes ? (size_t{64} << 20) / bytes : 16u;
        return by_bytes < 8u ? 8u : (by_bytes > 16u ? 16u : by_bytes);
    }
    // DEVICE projection buffers stay busy until the compute stream has run their filter and snapshot -- beside a running fused
    // launch that takes milliseconds per frame (the filter's waves do not fit next to the fused kernel's) -- so the rotation holds
    // a whole group of deferred projections and a few more, as far as 2 GiB go: a host that supplies frames faster than that trickle
    // keeps filling while the launch runs, and the backlog drains in a millisecond once the launch ends (2048^2 frames into a
    // 256-slice slab, filter in place: the loop waited 0.38 ms per frame on its 8 buffers). Device allocations cost ~0.1 ms each.
    // Deferral by reference (defer_refs): the pending group's buffers are not in the pool at all, the group in flight and the one
    // queued behind it are parked and busy -- two groups and a few more, as far as 4 GiB go.
    size_t device_pool_capacity(size_t bytes) const
    {
        const size_t want = (defer_refs != 0 && defer_depth > 1u) ? 2u * defer_depth + 8u : 56u;
        const size_t by_bytes = bytes ? (size_t{defer_refs != 0 ? 4u : 2u} << 30) / bytes : want;
        return by_bytes < 8u ? 8u : (by_bytes > want ? want : by_bytes);
    }
    // what the pools of ALL sizes may keep parked per ctx. Beyond it idle buffers (their last user has finished) are
    // returned to the runtime, other sizes than the one being parked first; on hipErrorOutOfMemory every pool is drained and the
    // allocation tried once more (paris_hip_drain_device_pool).
    static constexpr size_t PARKED_DEVICE_LIMIT = size_t{6} << 30;
    size_t parked_device_bytes = 0;
    // validators (backproject.hip: fast division; validate.hip) never run on the caller's stream -- it may be capturing, or hold
    // queued work the caller does not want to wait for -- but on a small stream of the ctx's own (round 4 borrowed the upload
    // stream: a check could then wait behind uploads that wait for a fused launch), with 8-byte mismatch counters;
    // made on first use (or by PARIS_HIP_CTX_WARM) and kept: creating and destroying a stream per check cost more than the checks
    hipStream_t aux_stream = nullptr;
    unsigned long long* aux_counter = nullptr; // AUX_SLOTS mismatch counters on the device (slot 0: the blocking checks)
    unsigned long long* aux_result = nullptr;  // their values read back, pinned host memory
    static constexpr int AUX_SLOTS = 8;
    // Asynchronous validation (paris_hip_set_async_validation; off in the bare library, on in paris::hip): a validator whose answer
    // is not cached yet is launched and NOT waited for -- the kernels use the compiler's IEEE forms (same bits, a few instructions
    // more per column) until a later call finds the check finished. The first call of a reconstruction no longer blocks for the
    // 2.3 ms exhaustive check of the division by the pixel pitch: for 360 small frames that was a tenth of the whole job.
    int async_validate = 0;
    struct pending_check
    {
        std::array<uint32_t, 4> key;
        int slot;
        hipEvent_t done;
    };
    std::vector<pending_check> pending_checks;
    std::vector<hipEvent_t> spare_events; // timing-disabled events ready for reuse (pool releases, upload targets)
    // deferred backprojection (paris_hip_set_backproject_deferral): projections copied at call time into a device ring and
    // added by one fused launch per `defer_depth` calls; the key_* fields are the arguments the pending calls share
    // Deferral BY REFERENCE (paris_hip_set_backproject_references; off in the bare library, on in paris::hip): a call whose projection
    // is a whole buffer of paris_hip_malloc_projection puts the buffer itself into the group -- no snapshot copy, no ring. The
    // library then owns what happens to that buffer until the group has run: paris_hip_free() of it only marks it (it is parked
    // behind the group's launch, ONE event per group), and any other library call that reads or writes it launches the group first
    // (paris_hip_projection_guard). With filter deferral the held-back weight + filter of such a projection runs IN PLACE, in the
    // group's one filter launch -- the buffer holds what the caller asked for whenever anything looks at it. Work the caller enqueues
    // on the ctx stream by itself is invisible to the library: callers that write a projection buffer with kernels of their own
    // keep the snapshots (the default).
    int defer_refs = 0;
    uint32_t held_count = 0;                              // slots of the pending group that are references
    std::vector<const void*> defer_ptr;                   // per slot of the pending group: where the projection lives
    std::vector<std::pair<void*, size_t>> defer_zombies;  // buffers the caller freed while the pending group refers to them
    bool defer_uses_ring = false;                         // the pending group has snapshot slots in ring half defer_half
    // fused launches in sequence: group g's completion is event group_events[g % GROUP_EVENTS] (recorded behind the launch, on the
    // stream that ran it; launches complete in sequence order -- every launch is ordered behind the one before it). A slot that
    // has been recorded again by a later group answers for the earlier one too.
    static constexpr uint32_t GROUP_EVENTS = 8;
    hipEvent_t group_events[GROUP_EVENTS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint64_t group_seq = 0, group_done_seq = 0;
    uint32_t defer_depth = 1; // 1 = immediate
    uint32_t defer_count = 0; // projections pending in the ring
    // The first groups of a reconstruction are launched early -- after 8, 16, 32 ... calls, until the depth is reached -- so that
    // the device starts on the volume while the caller is still feeding the first full group (48 frames of 2048^2 take a host loop
    // ~60 ms to supply). Reset when a call with other arguments starts a new sequence. Same additions in the same order.
    uint32_t defer_ramp = 8;
    float* defer_ring = nullptr;
    size_t defer_pitch = 0;
    uint32_t defer_dim_x = 0, defer_dim_y = 0, defer_slots = 0;
    bool defer_f16 = false; // the ring holds IEEE half pixels (calls through paris_hip_backproject_f16)
    float* key_v = nullptr;
    bool key_valid = false; // the key_* fields describe the group launched last (or pending): a matching call continues it
    uint32_t key_dims[4] = {0, 0, 0, 0}; // v_dim_x, v_dim_y, v_dim_z, v_offset
    paris_detector_geometry key_det{};
    paris_volume_geometry key_vol{};
    int key_enable_roi = 0;
    paris_region_of_interest key_roi{};
    float key_delta_s = 0.f, key_delta_t = 0.f;
    std::vector<float> defer_sin, defer_cos;
    // The fused launch of a full ring runs on a stream of its own (bp_stream), ordered behind the group's snapshot copies by an
    // event, so that the copies / weightings / filters of the NEXT group, which the caller keeps enqueuing on `stream`, run
    // beside it instead of behind it. Round 3 measured a loss and kept it off -- but its first call of every group still joined
    // the streams, so nothing overlapped; since round 4 a call that continues the reconstruction of the group launched
    // last (key_valid) starts the next group without a join, and PARIS's per-projection loop through paris::hip -- whose uploads,
    // filters, snapshots and buffer releases otherwise queue behind the running launch -- gains 19 ... 100 %
    // (profiles/r04_demo_paris_hip_mirror.txt). Off by default in the bare library, on in paris::hip. The ring has two halves,
    // written alternately; a half is written again only after the launch that read it has finished (bp_half_done). Every entry
    // point that observes a volume or completes work joins: `stream` is made to wait for the last fused launch
    // (paris_hip_flush_deferred); a group that runs on `stream` itself (a single projection, a synchronous ctx) joins first.
    int bp_overlap = 0;               // knob (paris_hip_set_backproject_overlap), off by default; ignored under PARIS_HIP_CTX_SYNCHRONOUS
    hipStream_t bp_stream = nullptr;
    hipEvent_t bp_ring_ready = nullptr;            // recorded on `stream`: the group's snapshots are in the ring
    hipEvent_t bp_half_done[2] = {nullptr, nullptr}; // recorded on bp_stream behind the launch that read ring half h
    bool bp_half_busy[2] = {false, false};         // bp_half_done[h] is pending: wait for it before writing half h again
    bool bp_inflight = false;                      // fused work on bp_stream that `stream` has not been ordered behind yet
    uint32_t bp_last_half = 0;
    uint32_t defer_half = 0;                       // ring half the pending group is being written to
    // two-pass backprojection (variant 5): factor / h / u planes of the slab's (x, y) plane, rewritten per projection
    float* colstate = nullptr;
    size_t colstate_floats = 0;
    // device copies of the per-projection sin/cos for the batched launch
    float* d_sincos = nullptr;
    uint32_t d_sincos_cap = 0;
};

// finishes a call: propagates launch errors, honours PARIS_HIP_CTX_SYNCHRONOUS
inline int paris_hip_finish(paris_hip_ctx* ctx)
{
    PARIS_HIP_TRY(hipGetLastError());
    if(ctx->flags & PARIS_HIP_CTX_SYNCHRONOUS)
        PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PARIS_HIP_SUCCESS;
}

// a pooled projection buffer that contains d_p is no longer "nothing has used it since it was handed out"
inline void paris_hip_mark_touched(paris_hip_ctx* ctx, const void* d_p)
{
    if(ctx->proj_allocs.empty())
        return;
    auto a = ctx->proj_allocs.upper_bound(const_cast<void*>(d_p));
    if(a == ctx->proj_allocs.begin())
        return;
    --a;
    if(static_cast<const char*>(d_p) < static_cast<const char*>(a->first) + a->second.bytes)
        a->second.touched = true;
}

// called by every stage entry point after it has enqueued work that reads or writes `bytes` bytes of projection memory
// starting at d_p (a whole frame, or the rows of a band): every registered upload destination that overlaps the range gets
// its last-use event re-recorded. Uploads and stage calls may address different row ranges of one frame buffer.
inline int paris_hip_note_projection_use(paris_hip_ctx* ctx, const void* d_p, size_t bytes)
{
    const char* lo = static_cast<const char*>(d_p);
    const char* hi = lo + (bytes ? bytes : 1u);
    paris_hip_mark_touched(ctx, d_p);
    if(ctx->upload_targets.empty())
        return PARIS_HIP_SUCCESS;
    // Targets may overlap (a driver uploads to interior band pointers that change per task), so ANY earlier-starting target can
    // still reach into the range: all targets that start before hi are tested (a handful of slots per ctx).
    for(auto it = ctx->upload_targets.begin(); it != ctx->upload_targets.end() && static_cast<const char*>(it->first) < hi; ++it)
    {
        if(static_cast<const char*>(it->first) + it->second.bytes <= lo)
            continue;
        PARIS_HIP_TRY(hipEventRecord(it->second.last_use, ctx->stream));
        it->second.used = true;
    }
    return PARIS_HIP_SUCCESS;
}

void paris_hip_forget_upload_target(paris_hip_ctx* ctx, const void* d_p);

// capi.hip: has fused launch `group` (a paris_hip_ctx::group_seq value) finished? wait = true: blocks until it has
int paris_hip_group_done(paris_hip_ctx* ctx, uint64_t group, bool wait, bool* done);
// capi.hip: every parked device buffer goes back to the runtime (after the streams have drained): the answer to hipErrorOutOfMemory
int paris_hip_drain_device_pool(paris_hip_ctx* ctx);
// capi.hip: the pending group has been launched as `group` (0: dropped) -- buffers freed meanwhile are parked behind it, the others
// remember it as their last reader
int paris_hip_release_group_references(paris_hip_ctx* ctx, uint64_t group);
// backproject.hip: a library call is about to read (writer == nullptr) or write (on stream `writer`) [d_p, d_p + bytes) of a
// projection buffer. A buffer the pending deferred group refers to: the group is launched first. A write into a buffer a launched
// group may still be reading: `writer` is made to wait for that group.
int paris_hip_projection_guard_slow(paris_hip_ctx* ctx, const void* d_p, size_t bytes, hipStream_t writer, bool writes);
inline int paris_hip_projection_guard(paris_hip_ctx* ctx, const void* d_p, size_t bytes, hipStream_t writer, bool writes)
{
    if(ctx == nullptr || (ctx->held_count == 0u && ctx->group_seq == ctx->group_done_seq))
        return PARIS_HIP_SUCCESS; // nothing pending by reference, no launch outstanding
    return paris_hip_projection_guard_slow(ctx, d_p, bytes, writer, writes);
}

// capi.hip: the ctx's lazily made pieces (PARIS_HIP_CTX_WARM makes them all at once)
int paris_hip_ensure_aux(paris_hip_ctx* ctx);           // aux_stream + aux_counter
// capi.hip: runs one validator -- enqueue(stream, counter) launches a kernel that adds its mismatches to *counter -- on the ctx's
// auxiliary stream. *known = true: *ok is the answer (no mismatch). *known = false (asynchronous validation only): the check is
// running, ask again later with the same key; the caller uses the unvalidated-safe form meanwhile and caches nothing.
int paris_hip_run_check(paris_hip_ctx* ctx, const std::array<uint32_t, 4>& key, void (*enqueue)(hipStream_t, unsigned long long*, const void*),
                        const void* arg, bool* ok, bool* known);
int paris_hip_ensure_upload_stream(paris_hip_ctx* ctx); // upload_stream + its event ring
int paris_hip_ensure_bp_stream(paris_hip_ctx* ctx);     // bp_stream + its events
// backproject.hip / filter.hip / filter_fused.hip / weight.hip / validate.hip: one cheap query per translation unit that makes the
// runtime load its code object now rather than at the first launch
void paris_hip_warm_backproject();
void paris_hip_warm_backproject_fused();
void paris_hip_warm_filter();
void paris_hip_warm_filter_fused();
void paris_hip_warm_weight();
void paris_hip_warm_validate();

// capi.hip: timing-disabled events, recycled through ctx->spare_events
int paris_hip_take_event(paris_hip_ctx* ctx, hipEvent_t* out);
void paris_hip_give_event(paris_hip_ctx* ctx, hipEvent_t e);
// capi.hip: a copy on `stream_bit`'s stream (paris_hip_ctx::USED_*) reads or writes pinned host memory at h_ptr
void paris_hip_note_host_use(paris_hip_ctx* ctx, const void* h_ptr, unsigned stream_bit);

inline int paris_hip_bind(paris_hip_ctx* ctx)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    PARIS_HIP_TRY(hipSetDevice(ctx->device));
    return PARIS_HIP_SUCCESS;
}

int paris_hip_get_plan(paris_hip_ctx* ctx, uint32_t n, paris_hip_fft_plan** out);

// weight.hip: runs a held-back weighting (stage fusion) as its own kernel; no-op when none is pending
int paris_hip_flush_pending_weight(paris_hip_ctx* ctx);

// filter_fused.hip
bool paris_hip_volume_is_clean(const paris_hip_ctx* ctx, const void* d_v, size_t bytes); // capi.hip
int paris_hip_fused_filter_tables(paris_hip_ctx* ctx, uint32_t n, paris_hip_fft_plan* plan);
int paris_hip_fused_filter_permute_k(paris_hip_ctx* ctx, const float* d_k, uint32_t n, float** d_kp);
int paris_hip_fused_filter_launch(paris_hip_ctx* ctx, float* d_rows, uint32_t pitch_f, uint32_t dim_x, uint32_t n_rows, uint32_t row_first,
                                  bool weight, float h_min, float v_min, float d_sd, float l_px_row, float l_px_col, const float* d_kp,
                                  const paris_hip_fft_plan* plan, uint32_t filter_size, uint16_t* d_half, uint32_t half_pitch,
                                  uint32_t n_frames = 1u, size_t frame_stride_f = 0u, size_t half_frame_stride = 0u,
                                  float* const* frame_rows = nullptr); // frame_rows: the first band row of each frame instead of d_rows + f * stride

// backproject.hip: runs the projections pending in the deferral ring (no-op when there are none). Called by every entry
// point that observes or changes a volume, completes work, or changes how backprojection runs.
int paris_hip_flush_deferred(paris_hip_ctx* ctx);
// the same without the join: the pending projections are launched (on bp_stream when overlapping) and `stream` is NOT made to
// wait for them -- for the deferring call itself when its ring is full
int paris_hip_launch_deferred(paris_hip_ctx* ctx);

// validate.hip: device validators of the hand-expanded IEEE sequences of ieee_lean.h, cached per process, device and operand range
// (*ok = false: the kernels use the compiler's IEEE forms). lean division: both quotients d_sd / den and d_so / den for every fp32
// den in [0.09, 1.92] x d_so. lean weighting: sqrt and d_sd / sqrt for every fp32 radicand in [dd, q_max] (widened a little).
int paris_hip_lean_division_check(paris_hip_ctx* ctx, float d_sd, float d_so, bool* ok);
int paris_hip_lean_weighting_check(paris_hip_ctx* ctx, float d_sd, double dd, double q_max, bool* ok);

// backproject.hip: exhaustive validation of the fast division by the detector's pixel pitches, ahead of the first
// backprojection (cached per process and device; a no-op once known or when the fast division is switched off)
int paris_hip_prevalidate_fast_division(paris_hip_ctx* ctx, float l_px_row, float l_px_col);
int paris_hip_prevalidate_lean_division(paris_hip_ctx* ctx, float d_so, float d_od);

#endif
