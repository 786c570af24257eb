// Context, memory and device management entry points of libparis_hip.so (include/paris_hip.h).
//
// Replaces the reference's per-backend device/memory/stream files: src/cuda/device.cpp, src/cuda/stream.cpp,
// src/cuda/memory.cpp, src/cuda/subvolume_information.cpp (and their trivial OpenMP counterparts
// src/openmp/memory.cpp, src/openmp/subvolume_information.cpp). All implicit thread_local state of the
// reference lives in paris_hip_ctx.
#include "paris_hip_internal.h"
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <new>

namespace
{
    int physical_device_count(int* count)
    {
        int n = 0;
        const hipError_t err = hipGetDeviceCount(&n);
        if(err == hipErrorNoDevice)
        {
            *count = 0;
            return PARIS_HIP_SUCCESS;
        }
        PARIS_HIP_TRY(err);
        *count = n;
        return PARIS_HIP_SUCCESS;
    }

    // Test hook: PARIS_HIP_VIRTUAL_DEVICES=k reports k device handles that map round-robin onto the physical GPUs, so the
    // one-host-thread-per-device driver (several ctxs, a shared task queue and sink) can be exercised on a one-GPU box.
    int virtual_devices()
    {
        static const int k = [] { const char* e = std::getenv("PARIS_HIP_VIRTUAL_DEVICES"); return e ? std::atoi(e) : 0; }();
        return k;
    }
}

extern "C" int paris_hip_device_count(int* count)
{
    if(count == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = physical_device_count(count))
        return rc;
    if(*count > 0 && virtual_devices() > *count)
        *count = virtual_devices();
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_ctx_create(int device, void* stream, unsigned flags, paris_hip_ctx** out)
{
    if(out == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if(int rc = paris_hip_device_count(&n))
        return rc;
    if(n == 0)
        return PARIS_HIP_ERROR_NO_DEVICE;
    if(device < 0 || device >= n)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    int physical = 0;
    if(int rc = physical_device_count(&physical))
        return rc;
    device %= physical; // identity unless PARIS_HIP_VIRTUAL_DEVICES is set
    PARIS_HIP_TRY(hipSetDevice(device)); // set_device: src/cuda/device.cpp:40-47

    paris_hip_ctx* ctx = new(std::nothrow) paris_hip_ctx;
    if(ctx == nullptr)
        return static_cast<int>(hipErrorOutOfMemory);
    ctx->device = device;
    ctx->flags = flags;
    if(stream != nullptr || (flags & PARIS_HIP_CTX_LEGACY_STREAM))
    {
        ctx->stream = static_cast<hipStream_t>(stream); // NULL = the legacy default stream
        ctx->owns_stream = false;
    }
    else
    {
        const hipError_t err = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if(err != hipSuccess)
        {
            delete ctx;
            return static_cast<int>(err);
        }
        ctx->owns_stream = true;
    }
    if(int rc = paris_hip_backproject_timing_arm(ctx, 1))
    {
        paris_hip_ctx_destroy(ctx);
        return rc;
    }
    if(flags & PARIS_HIP_CTX_WARM)
    {
        int rc = paris_hip_ensure_aux(ctx);
        if(rc == PARIS_HIP_SUCCESS)
            rc = paris_hip_ensure_upload_stream(ctx);
        if(rc == PARIS_HIP_SUCCESS && !(flags & PARIS_HIP_CTX_SYNCHRONOUS))
            rc = paris_hip_ensure_bp_stream(ctx);
        if(rc == PARIS_HIP_SUCCESS)
        {
            paris_hip_warm_backproject();
            paris_hip_warm_backproject_fused();
            paris_hip_warm_filter();
            paris_hip_warm_filter_fused();
            paris_hip_warm_weight();
            paris_hip_warm_validate();
            (void)hipGetLastError();
            // the runtime sets up its staging path on the first blocking host-to-device copy (~9 ms), its DMA queue on the first
            // asynchronous one from pinned memory (~7 ms): both paid here, on the counter's 8 bytes
            const unsigned long long zero = 0ull;
            hipError_t err = hipMemcpy(ctx->aux_counter, &zero, sizeof(zero), hipMemcpyHostToDevice);
            void* pinned = nullptr;
            if(err == hipSuccess)
                err = hipHostMalloc(&pinned, 4096, hipHostMallocDefault);
            if(err == hipSuccess)
            {
                std::memset(pinned, 0, 4096);
                err = hipMemcpy2DAsync(ctx->aux_counter, sizeof(zero), pinned, sizeof(zero), sizeof(zero), 1, hipMemcpyHostToDevice, ctx->upload_stream);
                if(err == hipSuccess)
                    err = hipStreamSynchronize(ctx->upload_stream);
                (void)hipHostFree(pinned);
            }
            rc = static_cast<int>(err);
        }
        if(rc != PARIS_HIP_SUCCESS)
        {
            paris_hip_ctx_destroy(ctx);
            return rc;
        }
    }
    *out = ctx;
    return PARIS_HIP_SUCCESS;
}

int paris_hip_ensure_aux(paris_hip_ctx* ctx)
{
    // a stream of the validators' own (library-owned, never captured by the caller, nothing else ever queued on it): one stream
    // create per ctx, absorbed by PARIS_HIP_CTX_WARM
    if(ctx->aux_stream == nullptr)
        PARIS_HIP_TRY(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    if(ctx->aux_counter == nullptr)
        PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->aux_counter), paris_hip_ctx::AUX_SLOTS * sizeof(unsigned long long)));
    if(ctx->aux_result == nullptr)
        PARIS_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->aux_result), paris_hip_ctx::AUX_SLOTS * sizeof(unsigned long long), hipHostMallocDefault));
    return PARIS_HIP_SUCCESS;
}

int paris_hip_run_check(paris_hip_ctx* ctx, const std::array<uint32_t, 4>& key, void (*enqueue)(hipStream_t, unsigned long long*, const void*),
                        const void* arg, bool* ok, bool* known)
{
    *ok = false;
    *known = false;
    if(int rc = paris_hip_ensure_aux(ctx))
        return rc;
    hipStream_t s = ctx->aux_stream;
    // is this very check running already (asynchronous validation)?
    for(auto it = ctx->pending_checks.begin(); it != ctx->pending_checks.end(); ++it)
    {
        if(it->key != key)
            continue;
        const hipError_t state = ctx->async_validate != 0 ? hipEventQuery(it->done) : hipEventSynchronize(it->done);
        if(state == hipErrorNotReady)
        {
            (void)hipGetLastError();
            return PARIS_HIP_SUCCESS; // still running: not known yet
        }
        PARIS_HIP_TRY(state);
        *ok = ctx->aux_result[it->slot] == 0ull;
        *known = true;
        paris_hip_give_event(ctx, it->done);
        ctx->pending_checks.erase(it);
        return PARIS_HIP_SUCCESS;
    }
    int slot = 0;
    if(ctx->async_validate != 0)
    {
        bool used[paris_hip_ctx::AUX_SLOTS] = {true}; // (slot 0 serves the blocking checks)
        for(const auto& p : ctx->pending_checks)
            used[p.slot] = true;
        for(int i = 1; i < paris_hip_ctx::AUX_SLOTS && slot == 0; ++i)
            if(!used[i])
                slot = i;
    }
    ctx->aux_result[slot] = ~0ull;
    PARIS_HIP_TRY(hipMemsetAsync(ctx->aux_counter + slot, 0, sizeof(unsigned long long), s));
    enqueue(s, ctx->aux_counter + slot, arg);
    PARIS_HIP_TRY(hipGetLastError());
    PARIS_HIP_TRY(hipMemcpyAsync(ctx->aux_result + slot, ctx->aux_counter + slot, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    if(slot == 0) // blocking (the default), or every asynchronous slot taken
    {
        PARIS_HIP_TRY(hipStreamSynchronize(s));
        *ok = ctx->aux_result[0] == 0ull;
        *known = true;
        return PARIS_HIP_SUCCESS;
    }
    paris_hip_ctx::pending_check p{key, slot, nullptr};
    if(int rc = paris_hip_take_event(ctx, &p.done))
        return rc;
    const hipError_t err = hipEventRecord(p.done, s);
    if(err != hipSuccess)
    {
        paris_hip_give_event(ctx, p.done);
        (void)hipStreamSynchronize(s);
        return static_cast<int>(err);
    }
    ctx->pending_checks.push_back(p);
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_async_validation(paris_hip_ctx* ctx, int enable)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->async_validate = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

int paris_hip_ensure_upload_stream(paris_hip_ctx* ctx)
{
    if(ctx->upload_stream != nullptr)
        return PARIS_HIP_SUCCESS;
    PARIS_HIP_TRY(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    for(int i = 0; i < 16; ++i)
    {
        hipEvent_t e = nullptr;
        PARIS_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->upload_events.push_back(e);
    }
    return PARIS_HIP_SUCCESS;
}

int paris_hip_ensure_bp_stream(paris_hip_ctx* ctx)
{
    if(ctx->bp_stream != nullptr)
        return PARIS_HIP_SUCCESS;
    PARIS_HIP_TRY(hipStreamCreateWithFlags(&ctx->bp_stream, hipStreamNonBlocking));
    PARIS_HIP_TRY(hipEventCreateWithFlags(&ctx->bp_ring_ready, hipEventDisableTiming));
    for(hipEvent_t& e : ctx->bp_half_done)
        PARIS_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return PARIS_HIP_SUCCESS;
}

static void destroy_events(paris_hip_ctx* ctx)
{
    for(hipEvent_t e : ctx->bp_start)
        (void)hipEventDestroy(e);
    for(hipEvent_t e : ctx->bp_stop)
        (void)hipEventDestroy(e);
    ctx->bp_start.clear();
    ctx->bp_stop.clear();
}

extern "C" int paris_hip_backproject_timing_arm(paris_hip_ctx* ctx, uint32_t capacity)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(capacity > 65536u)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if(capacity == 0) // no events around the launches at all: what a caller capturing the ctx stream into a hipGraph wants
    {
        destroy_events(ctx);
        ctx->bp_launches = 0;
        return PARIS_HIP_SUCCESS;
    }
    if(ctx->bp_start.size() != capacity)
    {
        destroy_events(ctx);
        for(uint32_t i = 0; i < capacity; ++i)
        {
            hipEvent_t a = nullptr, b = nullptr;
            PARIS_HIP_TRY(hipEventCreate(&a));
            ctx->bp_start.push_back(a);
            PARIS_HIP_TRY(hipEventCreate(&b));
            ctx->bp_stop.push_back(b);
        }
    }
    ctx->bp_launches = 0;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_backproject_timing_collect(paris_hip_ctx* ctx, float* ms, uint32_t max_n, uint32_t* n_out)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(ms == nullptr || n_out == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint64_t cap = ctx->bp_start.size();
    if(cap == 0)
    {
        *n_out = 0;
        return PARIS_HIP_SUCCESS;
    }
    const uint64_t have = ctx->bp_launches < cap ? ctx->bp_launches : cap;
    const uint64_t first = ctx->bp_launches - have; // oldest launch still in the ring
    uint32_t n = 0;
    for(uint64_t i = first; i < ctx->bp_launches && n < max_n; ++i, ++n)
        PARIS_HIP_TRY(hipEventElapsedTime(&ms[n], ctx->bp_start[i % cap], ctx->bp_stop[i % cap]));
    *n_out = n;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_ctx_destroy(paris_hip_ctx* ctx)
{
    if(ctx == nullptr)
        return PARIS_HIP_SUCCESS;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream); // NULL: the legacy default stream
    if(ctx->upload_stream != nullptr)
        (void)hipStreamSynchronize(ctx->upload_stream);
    ctx->pending_weight.active = false; // a weighting nobody filtered or read: dropped with the ctx
    if(ctx->defer_count != 0)
    {
        // Projections still deferred belong to key_v. Every library entry point that reads or frees a volume has flushed them
        // already, so they are pending only if the caller touched the volume some other way (own kernel, torch tensor, plain
        // hipFree). They are run only into a volume THIS ctx allocated and has not freed (paris_hip_malloc_volume's bookkeeping):
        // for any other address "is a live device allocation" does not say whose -- a wrapped tensor released to a caching
        // allocator may already back something else -- so those are dropped (callers of wrapped volumes call paris_hip_flush
        // before they let go of the memory; paris_amd.backend.Backend.close does). First thing in destroy: the launch needs
        // the ctx's tables, ring and stream intact.
        bool ours = false;
        {
            const uintptr_t a = reinterpret_cast<uintptr_t>(ctx->key_v);
            const size_t bytes = static_cast<size_t>(ctx->key_dims[0]) * ctx->key_dims[1] * ctx->key_dims[2] * sizeof(float);
            auto it = ctx->volume_allocs.upper_bound(a);
            if(it != ctx->volume_allocs.begin())
            {
                --it;
                ours = a >= it->first && a + bytes <= it->first + it->second;
            }
        }
        if(ours)
        {
            (void)paris_hip_flush_deferred(ctx);
            (void)hipStreamSynchronize(ctx->stream);
        }
        ctx->defer_count = 0;
    }
    if(ctx->bp_stream != nullptr)
    {
        (void)hipStreamSynchronize(ctx->bp_stream); // fused launches the caller never joined (it never looked at the volume again)
        if(ctx->bp_ring_ready != nullptr)
            (void)hipEventDestroy(ctx->bp_ring_ready);
        for(hipEvent_t e : ctx->bp_half_done)
            if(e != nullptr)
                (void)hipEventDestroy(e);
        (void)hipStreamDestroy(ctx->bp_stream);
        ctx->bp_stream = nullptr;
    }
    if(ctx->aux_stream != nullptr)
    {
        (void)hipStreamSynchronize(ctx->aux_stream); // a validator still running (asynchronous validation) writes the counters below
        (void)hipStreamDestroy(ctx->aux_stream);
        ctx->aux_stream = nullptr;
    }
    for(auto& p : ctx->pending_checks)
        (void)hipEventDestroy(p.done);
    if(ctx->aux_counter != nullptr)
        (void)hipFree(ctx->aux_counter);
    if(ctx->aux_result != nullptr)
        (void)hipHostFree(ctx->aux_result);
    for(auto& kv : ctx->plans)
    {
        (void)hipFree(kv.second.d_twiddle);
        if(kv.second.d_tab_first != nullptr)
            (void)hipFree(kv.second.d_tab_first);
        for(float2* t : kv.second.d_tab_mid)
            if(t != nullptr)
                (void)hipFree(t);
    }
    for(auto& kv : ctx->filters)
        if(kv.second.d_kp != nullptr)
            (void)hipFree(kv.second.d_kp);
    if(ctx->d_sincos != nullptr)
        (void)hipFree(ctx->d_sincos);
    if(ctx->colstate != nullptr)
        (void)hipFree(ctx->colstate);
    for(auto& kv : ctx->proj_pool)
    {
        if(kv.second.released != nullptr)
            (void)hipEventDestroy(kv.second.released);
        (void)hipFree(kv.second.ptr);
    }
    for(auto& z : ctx->defer_zombies) // freed by the caller while a group that was never run referred to them
        (void)hipFree(z.first);
    for(hipEvent_t e : ctx->group_events)
        if(e != nullptr)
            (void)hipEventDestroy(e);
    for(auto& kv : ctx->host_pool)
    {
        if(kv.second.released != nullptr)
            (void)hipEventDestroy(kv.second.released);
        (void)hipHostFree(kv.second.ptr);
    }
    for(auto& kv : ctx->upload_targets)
        (void)hipEventDestroy(kv.second.last_use);
    for(hipEvent_t e : ctx->spare_events)
        (void)hipEventDestroy(e);
    if(ctx->defer_ring != nullptr)
        (void)hipFree(ctx->defer_ring);
    if(ctx->stage_k != nullptr)
        (void)hipFree(ctx->stage_k);
    if(ctx->upload_stream != nullptr)
    {
        for(hipEvent_t e : ctx->upload_events)
            (void)hipEventDestroy(e);
        (void)hipStreamDestroy(ctx->upload_stream);
    }
    for(hipEvent_t e : ctx->bp_start)
        (void)hipEventDestroy(e);
    for(hipEvent_t e : ctx->bp_stop)
        (void)hipEventDestroy(e);
    if(ctx->owns_stream && ctx->stream != nullptr)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_ctx_synchronize(paris_hip_ctx* ctx)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PARIS_HIP_SUCCESS;
}

extern "C" void* paris_hip_ctx_stream(paris_hip_ctx* ctx)
{
    return ctx ? static_cast<void*>(ctx->stream) : nullptr;
}

// ---- fences: let a pipelined host loop reuse pinned buffers safely -----------------------------------------

struct paris_hip_fence
{
    hipEvent_t event = nullptr;
    bool recorded = false;
};

extern "C" int paris_hip_fence_create(paris_hip_ctx* ctx, paris_hip_fence** out)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(out == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    paris_hip_fence* f = new(std::nothrow) paris_hip_fence;
    if(f == nullptr)
        return static_cast<int>(hipErrorOutOfMemory);
    const hipError_t err = hipEventCreateWithFlags(&f->event, hipEventDisableTiming);
    if(err != hipSuccess)
    {
        delete f;
        return static_cast<int>(err);
    }
    *out = f;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_fence_record(paris_hip_ctx* ctx, paris_hip_fence* fence)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(fence == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    PARIS_HIP_TRY(hipEventRecord(fence->event, ctx->stream));
    fence->recorded = true;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_fence_wait(paris_hip_ctx* ctx, paris_hip_fence* fence)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(fence == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(fence->recorded)
        PARIS_HIP_TRY(hipEventSynchronize(fence->event));
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_fence_destroy(paris_hip_ctx* ctx, paris_hip_fence* fence)
{
    if(fence == nullptr)
        return PARIS_HIP_SUCCESS;
    if(ctx != nullptr)
        (void)hipSetDevice(ctx->device);
    (void)hipEventDestroy(fence->event);
    delete fence;
    return PARIS_HIP_SUCCESS;
}

// ---- memory ------------------------------------------------------------------------------------------

int paris_hip_take_event(paris_hip_ctx* ctx, hipEvent_t* out)
{
    if(!ctx->spare_events.empty())
    {
        *out = ctx->spare_events.back();
        ctx->spare_events.pop_back();
        return PARIS_HIP_SUCCESS;
    }
    PARIS_HIP_TRY(hipEventCreateWithFlags(out, hipEventDisableTiming));
    return PARIS_HIP_SUCCESS;
}

void paris_hip_give_event(paris_hip_ctx* ctx, hipEvent_t e)
{
    if(e == nullptr)
        return;
    if(ctx->spare_events.size() < 256u)
        ctx->spare_events.push_back(e);
    else
        (void)hipEventDestroy(e);
}

void paris_hip_note_host_use(paris_hip_ctx* ctx, const void* h_ptr, unsigned stream_bit)
{
    if(ctx->host_allocs.empty())
        return;
    auto a = ctx->host_allocs.upper_bound(const_cast<void*>(h_ptr));
    if(a == ctx->host_allocs.begin())
        return;
    --a;
    if(static_cast<const char*>(h_ptr) < static_cast<const char*>(a->first) + a->second.bytes)
        a->second.used |= stream_bit;
}

namespace
{
    using pool_t = std::multimap<size_t, paris_hip_ctx::pooled_buffer>;

    // takes the oldest released buffer of `bytes` if it is free to use (or if that size's share of the pool is full: then waits for it)
    int pool_take(paris_hip_ctx* ctx, pool_t& pool, size_t bytes, size_t capacity, void** out)
    {
        *out = nullptr;
        auto it = pool.lower_bound(bytes);
        if(it == pool.end() || it->first != bytes)
            return PARIS_HIP_SUCCESS;
        // let the caller allocate another one while the rotation is below its capacity. Buffers the caller has freed but the pending
        // deferred group still refers to (deferral by reference) are part of the rotation: with them counted, a host that runs ahead
        // of the device waits HERE for the oldest launch's buffers -- two groups in flight behind the one being filled
        size_t in_rotation = pool.count(bytes);
        if(&pool == &ctx->proj_pool)
            for(const auto& z : ctx->defer_zombies)
                in_rotation += z.second == bytes ? 1u : 0u;
        const bool may_grow = in_rotation < capacity;
        if(it->second.group != 0u)
        {
            bool done = false;
            if(int rc = paris_hip_group_done(ctx, it->second.group, false, &done))
                return rc;
            if(!done)
            {
                if(may_grow)
                    return PARIS_HIP_SUCCESS;
                if(int rc = paris_hip_group_done(ctx, it->second.group, true, &done))
                    return rc;
            }
        }
        if(it->second.released != nullptr)
        {
            const hipError_t state = hipEventQuery(it->second.released);
            if(state == hipErrorNotReady)
            {
                (void)hipGetLastError();
                if(may_grow)
                    return PARIS_HIP_SUCCESS;
                PARIS_HIP_TRY(hipEventSynchronize(it->second.released));
            }
            else
                PARIS_HIP_TRY(state);
            paris_hip_give_event(ctx, it->second.released);
        }
        *out = it->second.ptr;
        if(&pool == &ctx->proj_pool)
            ctx->parked_device_bytes -= std::min(ctx->parked_device_bytes, bytes);
        pool.erase(it);
        return PARIS_HIP_SUCCESS;
    }

    // is this parked buffer's last user done? (never blocks; an error counts as busy)
    bool pool_entry_idle(paris_hip_ctx* ctx, const paris_hip_ctx::pooled_buffer& b)
    {
        if(b.group != 0u)
        {
            bool done = false;
            if(paris_hip_group_done(ctx, b.group, false, &done) != PARIS_HIP_SUCCESS || !done)
                return false;
        }
        if(b.released != nullptr && hipEventQuery(b.released) != hipSuccess)
        {
            (void)hipGetLastError();
            return false;
        }
        return true;
    }

    // the device pool's parked bytes are bounded over ALL sizes. Idle buffers go back to the runtime until the pool is
    // under the limit again -- sizes other than `keep` first (a driver that moved on to another frame or band size never asks for
    // the old one again), oldest first inside a size.
    void pool_trim(paris_hip_ctx* ctx, size_t keep)
    {
        for(int pass = 0; pass < 2 && ctx->parked_device_bytes > paris_hip_ctx::PARKED_DEVICE_LIMIT; ++pass)
            for(auto it = ctx->proj_pool.begin(); it != ctx->proj_pool.end() && ctx->parked_device_bytes > paris_hip_ctx::PARKED_DEVICE_LIMIT;)
            {
                if((pass == 0 && it->first == keep) || !pool_entry_idle(ctx, it->second))
                {
                    ++it;
                    continue;
                }
                paris_hip_give_event(ctx, it->second.released);
                (void)hipFree(it->second.ptr);
                ctx->parked_device_bytes -= std::min(ctx->parked_device_bytes, it->first);
                it = ctx->proj_pool.erase(it);
            }
    }

    // parks a buffer behind its last user: an event recorded on `last` now (everything enqueued there so far), a fused launch that
    // reads it by reference (group != 0), or nothing at all when nothing used the buffer (last == nullptr and !used); *parked = false
    // when that size's share of the pool is full
    int pool_park(paris_hip_ctx* ctx, pool_t& pool, size_t bytes, size_t capacity, void* ptr, bool used, hipStream_t last, bool* parked,
                  uint64_t group = 0u)
    {
        *parked = false;
        if(pool.count(bytes) >= capacity)
            return PARIS_HIP_SUCCESS;
        hipEvent_t e = nullptr;
        if(used)
        {
            if(int rc = paris_hip_take_event(ctx, &e))
                return rc;
            const hipError_t err = hipEventRecord(e, last);
            if(err != hipSuccess)
            {
                paris_hip_give_event(ctx, e);
                return static_cast<int>(err);
            }
        }
        // free-at-once buffers go to the front of their size class (taken first), busy ones behind the older busy ones
        if(e == nullptr && group == 0u)
            pool.emplace_hint(pool.lower_bound(bytes), bytes, paris_hip_ctx::pooled_buffer{ptr, e, 0u});
        else
            pool.emplace(bytes, paris_hip_ctx::pooled_buffer{ptr, e, group});
        *parked = true;
        if(&pool == &ctx->proj_pool)
        {
            ctx->parked_device_bytes += bytes;
            if(ctx->parked_device_bytes > paris_hip_ctx::PARKED_DEVICE_LIMIT)
                pool_trim(ctx, bytes);
        }
        return PARIS_HIP_SUCCESS;
    }
}

int paris_hip_group_done(paris_hip_ctx* ctx, uint64_t group, bool wait, bool* done)
{
    *done = false;
    while(ctx->group_done_seq < group)
    {
        const uint64_t next = ctx->group_done_seq + 1u;
        hipEvent_t e = ctx->group_events[next % paris_hip_ctx::GROUP_EVENTS];
        if(e == nullptr) // (cannot happen: a group number is handed out with its event recorded)
            return PARIS_HIP_ERROR_INVALID_ARGUMENT;
        const hipError_t state = wait ? hipEventSynchronize(e) : hipEventQuery(e);
        if(state == hipErrorNotReady)
        {
            (void)hipGetLastError();
            return PARIS_HIP_SUCCESS;
        }
        PARIS_HIP_TRY(state);
        ctx->group_done_seq = next;
    }
    *done = true;
    return PARIS_HIP_SUCCESS;
}

int paris_hip_drain_device_pool(paris_hip_ctx* ctx)
{
    if(ctx->proj_pool.empty())
        return PARIS_HIP_SUCCESS;
    // every user of a parked buffer was enqueued on one of the ctx's streams
    PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if(ctx->bp_stream != nullptr)
        PARIS_HIP_TRY(hipStreamSynchronize(ctx->bp_stream));
    if(ctx->upload_stream != nullptr)
        PARIS_HIP_TRY(hipStreamSynchronize(ctx->upload_stream));
    for(auto& kv : ctx->proj_pool)
    {
        paris_hip_give_event(ctx, kv.second.released);
        (void)hipFree(kv.second.ptr);
    }
    ctx->proj_pool.clear();
    ctx->parked_device_bytes = 0;
    return PARIS_HIP_SUCCESS;
}

int paris_hip_release_group_references(paris_hip_ctx* ctx, uint64_t group)
{
    int rc = PARIS_HIP_SUCCESS;
    for(auto& z : ctx->defer_zombies)
    {
        bool parked = false;
        int prc = PARIS_HIP_SUCCESS;
        if(group != 0u) // (always room: what the rotation may hold was settled when the buffers were handed out, pool_take)
            prc = pool_park(ctx, ctx->proj_pool, z.second, ~size_t{0}, z.first, false, nullptr, &parked, group);
        if(!parked)
        {
            // no room in the pool (or the group was dropped): wait for whatever may still read the buffer, then give it back
            if(group != 0u)
            {
                bool done = false;
                (void)paris_hip_group_done(ctx, group, true, &done);
            }
            else
                (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(z.first);
        }
        if(prc != PARIS_HIP_SUCCESS && rc == PARIS_HIP_SUCCESS)
            rc = prc;
    }
    ctx->defer_zombies.clear();
    if(ctx->held_count != 0u)
        for(auto& kv : ctx->proj_allocs)
            if(kv.second.held != 0u)
            {
                kv.second.held = 0u;
                if(group != 0u)
                    kv.second.group = group;
            }
    ctx->held_count = 0u;
    return rc;
}

extern "C" int paris_hip_malloc_projection(paris_hip_ctx* ctx, uint32_t dim_x, uint32_t dim_y, float** d_ptr,
                                           size_t* pitch)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_ptr == nullptr || pitch == nullptr || dim_x == 0 || dim_y == 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    // rows padded to 256 B: every row starts on a cache-line pair and float4 staging stays aligned
    const size_t row = (static_cast<size_t>(dim_x) * sizeof(float) + 255u) & ~static_cast<size_t>(255u);
    const size_t bytes = row * dim_y;
    void* p = nullptr;
    if(int rc = pool_take(ctx, ctx->proj_pool, bytes, ctx->device_pool_capacity(bytes), &p))
        return rc;
    if(p == nullptr)
    {
        hipError_t err = hipMalloc(&p, bytes);
        if(err == hipErrorOutOfMemory)
        {
            // buffers parked in the pools (other sizes, a finished job's rotation) are memory too
            (void)hipGetLastError();
            if(int rc = paris_hip_drain_device_pool(ctx))
                return rc;
            err = hipMalloc(&p, bytes);
        }
        PARIS_HIP_TRY(err);
    }
    ctx->proj_allocs[p] = paris_hip_ctx::proj_alloc{bytes, false, 0u, 0u};
    *d_ptr = static_cast<float*>(p);
    *pitch = row;
    return PARIS_HIP_SUCCESS;
}

namespace
{
    void erase_overlapping(std::map<uintptr_t, size_t>& m, uintptr_t a, size_t bytes)
    {
        for(auto it = m.begin(); it != m.end();)
            it = (a < it->first + it->second && it->first < a + bytes) ? m.erase(it) : std::next(it);
    }

    // [p, p + bytes) no longer belongs to a volume of this ctx
    void forget_volume_range(paris_hip_ctx* ctx, const void* p, size_t bytes)
    {
        erase_overlapping(ctx->clean_volumes, reinterpret_cast<uintptr_t>(p), bytes);
        erase_overlapping(ctx->volume_allocs, reinterpret_cast<uintptr_t>(p), bytes);
    }
}

extern "C" int paris_hip_memset_volume(paris_hip_ctx* ctx, float* d_ptr, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_ptr == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const size_t bytes = static_cast<size_t>(dim_x) * dim_y * dim_z * sizeof(float);
    PARIS_HIP_TRY(hipMemsetAsync(d_ptr, 0, bytes, ctx->stream));
    // zeros written into a listed volume leave it clean; a volume of this ctx that was taken off the list (host upload,
    // paris_hip_volume_mark_dirty) is listed again when the fill covers it whole: every voxel is +0 once more
    auto mine = ctx->volume_allocs.find(reinterpret_cast<uintptr_t>(d_ptr));
    if(mine != ctx->volume_allocs.end() && mine->second == bytes)
        ctx->clean_volumes[mine->first] = bytes;
    return paris_hip_finish(ctx);
}

// Extension (include/paris_hip.h): the caller wrote [d_ptr, d_ptr + bytes) itself (own kernel, a torch tensor over the same
// memory) and the values may include -0 -- volumes that overlap the range always take every addition from now on
extern "C" int paris_hip_volume_mark_dirty(paris_hip_ctx* ctx, const void* d_ptr, size_t bytes)
{
    if(int rc = paris_hip_flush_deferred(ctx)) // pending projections were enqueued under the old promise: same result either way, but keep the order simple
        return rc;
    if(ctx == nullptr || d_ptr == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    erase_overlapping(ctx->clean_volumes, reinterpret_cast<uintptr_t>(d_ptr), bytes ? bytes : 1u);
    return PARIS_HIP_SUCCESS;
}

// Extension: the caller vouches that [d_ptr, d_ptr + bytes) holds no -0 right now (freshly zero-filled, or written by nothing but
// backprojections since): backprojections into it may skip the tiles no ray reaches. Also for memory the library did not allocate.
extern "C" int paris_hip_volume_mark_clean(paris_hip_ctx* ctx, const void* d_ptr, size_t bytes)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(ctx == nullptr || d_ptr == nullptr || bytes == 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_ptr);
    if(paris_hip_volume_is_clean(ctx, d_ptr, bytes))
        return PARIS_HIP_SUCCESS;
    erase_overlapping(ctx->clean_volumes, a, bytes); // entries are kept disjoint: the lookup tests one of them
    ctx->clean_volumes[a] = bytes;
    return PARIS_HIP_SUCCESS;
}

namespace
{
    // counts the words of w[0, n) that are the bit pattern of -0.0f: the 16-byte aligned body as uint4 loads (read once, not
    // kept in the caches), the few words before and after it one by one
    __global__ void __launch_bounds__(256) count_negative_zero_kernel(const uint32_t* __restrict__ w, size_t n, unsigned long long* __restrict__ out)
    {
        const size_t head = min(n, static_cast<size_t>(((16u - static_cast<uint32_t>(reinterpret_cast<uintptr_t>(w) & 15u)) & 15u) / 4u));
        const size_t n4 = (n - head) / 4u;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* __restrict__ body = reinterpret_cast<const u32x4*>(w + head);
        uint32_t c = 0;
        for(size_t i = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256u)
        {
            const u32x4 q = __builtin_nontemporal_load(body + i);
            c += (q.x == 0x80000000u) + (q.y == 0x80000000u) + (q.z == 0x80000000u) + (q.w == 0x80000000u);
        }
        if(blockIdx.x == 0)
        {
            const size_t tail0 = head + 4u * n4;
            if(threadIdx.x < head)
                c += w[threadIdx.x] == 0x80000000u;
            if(tail0 + threadIdx.x < n)
                c += w[tail0 + threadIdx.x] == 0x80000000u;
        }
        if(c != 0)
            atomicAdd(out, static_cast<unsigned long long>(c));
    }
}

// Extension (include/paris_hip.h): looks for -0 in [d_ptr, d_ptr + bytes) once, on the device, and lists the range as clean when
// there is none -- the way to let memory of unknown history (a tensor of the caller's) skip the tiles no ray reaches
extern "C" int paris_hip_volume_scan_clean(paris_hip_ctx* ctx, const void* d_ptr, size_t bytes, uint64_t* negative_zeros)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_ptr == nullptr || bytes == 0 || bytes % sizeof(float) != 0 || reinterpret_cast<uintptr_t>(d_ptr) % sizeof(float) != 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    hipPointerAttribute_t attr{};
    if(hipPointerGetAttributes(&attr, d_ptr) != hipSuccess || attr.type != hipMemoryTypeDevice)
    {
        (void)hipGetLastError();
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    }
    unsigned long long* d_count = nullptr;
    PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_count), sizeof(unsigned long long)));
    const size_t n = bytes / sizeof(float);
    const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>(16384u, std::max<size_t>(1u, (n / 4u + 255u) / 256u)));
    unsigned long long count = 0;
    hipError_t err = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream);
    if(err == hipSuccess)
    {
        hipLaunchKernelGGL(count_negative_zero_kernel, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(d_ptr), n, d_count);
        err = hipGetLastError();
    }
    if(err == hipSuccess)
        err = hipMemcpyAsync(&count, d_count, sizeof(count), hipMemcpyDeviceToHost, ctx->stream);
    if(err == hipSuccess)
        err = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_count);
    PARIS_HIP_TRY(err);
    if(negative_zeros != nullptr)
        *negative_zeros = count;
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_ptr);
    if(count != 0)
    {
        erase_overlapping(ctx->clean_volumes, a, bytes); // whatever was promised about the range before, it holds a -0 now
        return PARIS_HIP_SUCCESS;
    }
    if(!paris_hip_volume_is_clean(ctx, d_ptr, bytes))
    {
        erase_overlapping(ctx->clean_volumes, a, bytes);
        ctx->clean_volumes[a] = bytes;
    }
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_malloc_volume(paris_hip_ctx* ctx, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, float** d_ptr)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_ptr == nullptr || dim_x == 0 || dim_y == 0 || dim_z == 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const size_t bytes = static_cast<size_t>(dim_x) * dim_y * dim_z * sizeof(float);
    void* p = nullptr;
    {
        hipError_t err = hipMalloc(&p, bytes);
        if(err == hipErrorOutOfMemory) // projection buffers parked in the pool are memory too
        {
            (void)hipGetLastError();
            if(int rc = paris_hip_drain_device_pool(ctx))
                return rc;
            err = hipMalloc(&p, bytes);
        }
        PARIS_HIP_TRY(err);
    }
    // make_volume_* zero-fills: src/openmp/memory.cpp:46-47, src/cuda/memory.cpp:55-56
    const hipError_t err = hipMemsetAsync(p, 0, bytes, ctx->stream);
    if(err != hipSuccess)
    {
        (void)hipFree(p);
        return static_cast<int>(err);
    }
    *d_ptr = static_cast<float*>(p);
    // an entry that overlaps the new allocation is stale (its memory was released behind the library's back and the address
    // range has been handed out again): it must neither vouch for the new volume nor shadow it
    forget_volume_range(ctx, p, bytes);
    ctx->volume_allocs[reinterpret_cast<uintptr_t>(p)] = bytes;
    ctx->clean_volumes[reinterpret_cast<uintptr_t>(p)] = bytes;
    return paris_hip_finish(ctx);
}

// [d_v, d_v + bytes) lies inside a volume this library allocated zero-filled and nothing but backprojections wrote since
bool paris_hip_volume_is_clean(const paris_hip_ctx* ctx, const void* d_v, size_t bytes)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_v);
    auto it = ctx->clean_volumes.upper_bound(a);
    if(it == ctx->clean_volumes.begin())
        return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
}

namespace
{
    // a write from outside the backprojection kernels into [p, p + bytes): volumes it touches are no longer known to be free of -0
    void volume_written(paris_hip_ctx* ctx, const void* p, size_t bytes)
    {
        erase_overlapping(ctx->clean_volumes, reinterpret_cast<uintptr_t>(p), bytes);
    }
}

extern "C" int paris_hip_free(paris_hip_ctx* ctx, void* d_ptr)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_ptr == nullptr)
        return PARIS_HIP_SUCCESS;
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
    auto filt = ctx->filters.find(static_cast<const float*>(d_ptr));
    if(filt != ctx->filters.end())
    {
        // Filter deferral: slots of the pending group may still hold a weight + filter that reads this K's permuted copy, and the
        // group filter of a launched group may still run on the second stream: the group runs first, and the compute
        // stream is ordered behind the second one before the host waits for it
        bool referenced = ctx->bp_inflight;
        for(uint32_t i = 0; i < ctx->defer_count && i < ctx->defer_wf.size() && !referenced; ++i)
            referenced = ctx->defer_wf[i].active && ctx->defer_wf[i].d_kp == filt->second.d_kp;
        if(referenced)
            if(int rc = paris_hip_flush_deferred(ctx))
                return rc;
        PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream)); // a filter launch may still read the permuted copy
        if(filt->second.d_kp != nullptr)
            PARIS_HIP_TRY(hipFree(filt->second.d_kp));
        ctx->filters.erase(filt);
    }
    auto proj = ctx->proj_allocs.find(d_ptr);
    if(proj != ctx->proj_allocs.end())
    {
        // a projection buffer: deferred backprojections hold their own snapshots, nothing pending refers to it. Its last user
        // was enqueued on the compute stream (every upload into it has been followed by a wait of the compute stream for it)
        const size_t bytes = proj->second.bytes;
        const uint32_t held = proj->second.held;
        uint64_t group = proj->second.group;
        ctx->proj_allocs.erase(proj);
        paris_hip_forget_upload_target(ctx, d_ptr);
        if(held != 0u)
        {
            // the pending deferred group refers to this very buffer (deferral by reference): it is parked when that group has been
            // launched, behind the launch (paris_hip_release_group_references) -- no event, no copy, nothing enqueued now
            ctx->defer_zombies.emplace_back(d_ptr, bytes);
            return PARIS_HIP_SUCCESS;
        }
        if(group != 0u)
        {
            bool done = false;
            if(int rc = paris_hip_group_done(ctx, group, false, &done))
                return rc;
            if(done)
                group = 0u;
        }
        bool parked = false;
        // (an event always: work the caller enqueued on the ctx stream itself may use the buffer too)
        if(int rc = pool_park(ctx, ctx->proj_pool, bytes, ctx->device_pool_capacity(bytes), d_ptr, true, ctx->stream, &parked, group))
            return rc;
        if(parked)
            return PARIS_HIP_SUCCESS;
        if(group != 0u)
        {
            bool done = false;
            if(int rc = paris_hip_group_done(ctx, group, true, &done))
                return rc;
        }
        PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if(ctx->upload_stream != nullptr)
            PARIS_HIP_TRY(hipStreamSynchronize(ctx->upload_stream));
        PARIS_HIP_TRY(hipFree(d_ptr));
        return PARIS_HIP_SUCCESS;
    }
    if(ctx->defer_count == 0 && ctx->bp_inflight)
        if(int rc = paris_hip_flush_deferred(ctx)) // nothing pending, but a fused launch may still run on the second stream: join
            return rc;
    if(ctx->defer_count != 0)
    {
        // pending projections write into key_v: run them first if that volume lives in the allocation being freed
        void* base = nullptr;
        size_t size = 0;
        const bool known = hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, d_ptr) == hipSuccess;
        const char* kv = reinterpret_cast<const char*>(ctx->key_v);
        if(!known || (kv >= static_cast<const char*>(base) && kv < static_cast<const char*>(base) + size) || ctx->bp_inflight)
            if(int rc = paris_hip_flush_deferred(ctx))
                return rc;
        (void)hipGetLastError();
    }
    {
        // nothing of the allocation may stay listed as a volume (clean or not): by range, whatever sub-ranges were marked
        auto mine = ctx->volume_allocs.find(reinterpret_cast<uintptr_t>(d_ptr));
        size_t size = mine != ctx->volume_allocs.end() ? mine->second : 0u;
        if(size == 0u)
        {
            void* base = nullptr;
            if(hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, d_ptr) != hipSuccess)
                size = 1u;
            (void)hipGetLastError();
        }
        forget_volume_range(ctx, d_ptr, size);
    }
    paris_hip_forget_upload_target(ctx, d_ptr);
    PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    PARIS_HIP_TRY(hipFree(d_ptr));
    return PARIS_HIP_SUCCESS;
}

void paris_hip_forget_upload_target(paris_hip_ctx* ctx, const void* d_p)
{
    auto it = ctx->upload_targets.find(d_p);
    if(it != ctx->upload_targets.end())
    {
        paris_hip_give_event(ctx, it->second.last_use);
        ctx->upload_targets.erase(it);
    }
}

extern "C" int paris_hip_malloc_host(paris_hip_ctx* ctx, size_t bytes, void** h_ptr)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(h_ptr == nullptr || bytes == 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = pool_take(ctx, ctx->host_pool, bytes, paris_hip_ctx::pool_capacity(bytes), h_ptr))
        return rc;
    if(*h_ptr == nullptr)
        PARIS_HIP_TRY(hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    if(bytes <= paris_hip_ctx::POOL_HOST_BYTES)
        ctx->host_allocs[*h_ptr] = paris_hip_ctx::host_alloc{bytes, 0u};
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_free_host(paris_hip_ctx* ctx, void* h_ptr)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(h_ptr == nullptr)
        return PARIS_HIP_SUCCESS;
    auto live = ctx->host_allocs.find(h_ptr);
    if(live != ctx->host_allocs.end())
    {
        const size_t bytes = live->second.bytes;
        const unsigned used = live->second.used;
        ctx->host_allocs.erase(live);
        // Copies from / into it may still be in flight; the event goes behind them on the stream that ran them. A frame that went
        // up through the upload stream alone is released when ITS copy is done, whatever the compute stream still has queued
        // (the fused launch of the previous group, say). Used from both streams: the compute stream has been made to wait for
        // every upload (paris_hip_upload_projection), so its tail covers them.
        hipStream_t last = (used == paris_hip_ctx::USED_UPLOAD && ctx->upload_stream != nullptr) ? ctx->upload_stream : ctx->stream;
        bool parked = false;
        if(int rc = pool_park(ctx, ctx->host_pool, bytes, paris_hip_ctx::pool_capacity(bytes), h_ptr, used != 0u, last, &parked))
            return rc;
        if(parked)
            return PARIS_HIP_SUCCESS;
    }
    PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if(ctx->upload_stream != nullptr)
        PARIS_HIP_TRY(hipStreamSynchronize(ctx->upload_stream));
    PARIS_HIP_TRY(hipHostFree(h_ptr));
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_memcpy_projection_h2d(paris_hip_ctx* ctx, float* d_dst, size_t d_pitch, const float* h_src,
                                               size_t h_pitch, uint32_t dim_x, uint32_t dim_y)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
    if(d_dst == nullptr || h_src == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = paris_hip_projection_guard(ctx, d_dst, d_pitch * dim_y, ctx->stream, true))
        return rc;
    // rows as far apart on both sides as they are long: one linear copy (the 2-D form costs the runtime more per call)
    if(d_pitch == h_pitch && d_pitch == static_cast<size_t>(dim_x) * sizeof(float))
        PARIS_HIP_TRY(hipMemcpyAsync(d_dst, h_src, d_pitch * dim_y, hipMemcpyHostToDevice, ctx->stream));
    else
        PARIS_HIP_TRY(hipMemcpy2DAsync(d_dst, d_pitch, h_src, h_pitch, static_cast<size_t>(dim_x) * sizeof(float), dim_y,
                                       hipMemcpyHostToDevice, ctx->stream));
    paris_hip_note_host_use(ctx, h_src, paris_hip_ctx::USED_COMPUTE);
    if(int rc = paris_hip_note_projection_use(ctx, d_dst, d_pitch * dim_y)) // a later upload into the buffer must not overtake this copy
        return rc;
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_upload_projection(paris_hip_ctx* ctx, float* d_dst, size_t d_pitch, const float* h_src, size_t h_pitch,
                                           uint32_t dim_x, uint32_t dim_y)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
    if(d_dst == nullptr || h_src == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
#ifdef PARIS_HIP_EXPERIMENTS
    static const bool serial = [] { const char* e = std::getenv("PARIS_HIP_UPLOAD_STREAM"); return e != nullptr && e[0] == '0'; }();
    if(serial) // diagnostic: PARIS_HIP_UPLOAD_STREAM=0 keeps the copy on the compute stream (A/B of the overlap)
        return paris_hip_memcpy_projection_h2d(ctx, d_dst, d_pitch, h_src, h_pitch, dim_x, dim_y);
#endif
    if(int rc = paris_hip_ensure_upload_stream(ctx))
        return rc;
    if(int rc = paris_hip_projection_guard(ctx, d_dst, d_pitch * dim_y, ctx->upload_stream, true))
        return rc;
    // Write-after-read on slot reuse: kernels already queued on the compute stream may still read d_dst. The upload waits for
    // the LAST library call that touched this very buffer (paris_hip_note_projection_use records it), not for everything queued:
    // work on other buffers keeps overlapping the transfer. The first upload into a buffer registers it.
    auto target = ctx->upload_targets.find(d_dst);
    bool fresh = false;
    if(target == ctx->upload_targets.end())
    {
        // First upload into this buffer: nothing has been recorded for it yet, but work already queued on the compute stream
        // may read or write it (a frame put there with paris_hip_memcpy_projection_h2d and still being filtered, say) -- the
        // upload waits for everything queued so far, once per buffer. Not so for a buffer of paris_hip_malloc_projection that no
        // library call has touched since it was handed out: the pool gives out only buffers whose last user has finished, so the
        // transfer starts at once, whatever the compute stream is still busy with, and the buffer is not even registered --
        // PARIS's loop takes a fresh buffer per projection and frees it after one use (src/loader.cpp:28-33); should the caller
        // upload into it a second time after all, that upload finds it touched and unregistered and takes the conservative path.
        auto mine = ctx->proj_allocs.find(d_dst);
        fresh = mine != ctx->proj_allocs.end() && !mine->second.touched;
        if(!fresh)
        {
            paris_hip_ctx::upload_target t;
            if(int rc = paris_hip_take_event(ctx, &t.last_use))
                return rc;
            PARIS_HIP_TRY(hipEventRecord(t.last_use, ctx->stream));
            t.used = true;
            target = ctx->upload_targets.emplace(d_dst, t).first;
        }
    }
    if(!fresh)
    {
        if(target->second.used)
            PARIS_HIP_TRY(hipStreamWaitEvent(ctx->upload_stream, target->second.last_use, 0));
        target->second.bytes = std::max(target->second.bytes, d_pitch * dim_y);
    }
    hipEvent_t done = ctx->upload_events[ctx->uploads++ % ctx->upload_events.size()];
    // rows as far apart on both sides as they are long: one linear copy (the 2-D form costs the runtime more per call)
    if(d_pitch == h_pitch && d_pitch == static_cast<size_t>(dim_x) * sizeof(float))
        PARIS_HIP_TRY(hipMemcpyAsync(d_dst, h_src, d_pitch * dim_y, hipMemcpyHostToDevice, ctx->upload_stream));
    else
        PARIS_HIP_TRY(hipMemcpy2DAsync(d_dst, d_pitch, h_src, h_pitch, static_cast<size_t>(dim_x) * sizeof(float), dim_y,
                                       hipMemcpyHostToDevice, ctx->upload_stream));
    paris_hip_note_host_use(ctx, h_src, paris_hip_ctx::USED_UPLOAD);
    paris_hip_mark_touched(ctx, d_dst);
    PARIS_HIP_TRY(hipEventRecord(done, ctx->upload_stream));
    PARIS_HIP_TRY(hipStreamWaitEvent(ctx->stream, done, 0)); // kernels enqueued from now on see the uploaded frame
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_memcpy_projection_d2h(paris_hip_ctx* ctx, float* h_dst, size_t h_pitch, const float* d_src,
                                               size_t d_pitch, uint32_t dim_x, uint32_t dim_y)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
    if(h_dst == nullptr || d_src == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = paris_hip_projection_guard(ctx, d_src, d_pitch * dim_y, nullptr, false)) // (a held-back filter of the pending group runs first)
        return rc;
    PARIS_HIP_TRY(hipMemcpy2DAsync(h_dst, h_pitch, d_src, d_pitch, static_cast<size_t>(dim_x) * sizeof(float), dim_y,
                                   hipMemcpyDeviceToHost, ctx->stream));
    paris_hip_note_host_use(ctx, h_dst, paris_hip_ctx::USED_COMPUTE);
    if(int rc = paris_hip_note_projection_use(ctx, d_src, d_pitch * dim_y))
        return rc;
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_memcpy_volume_h2d(paris_hip_ctx* ctx, float* d_dst, const float* h_src, uint32_t dim_x,
                                           uint32_t dim_y, uint32_t dim_z)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_dst == nullptr || h_src == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const size_t bytes = static_cast<size_t>(dim_x) * dim_y * dim_z * sizeof(float);
    volume_written(ctx, d_dst, bytes); // the host's data may hold -0
    PARIS_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    paris_hip_note_host_use(ctx, h_src, paris_hip_ctx::USED_COMPUTE);
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_memcpy_volume_d2h(paris_hip_ctx* ctx, float* h_dst, const float* d_src, uint32_t dim_x,
                                           uint32_t dim_y, uint32_t dim_z)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(h_dst == nullptr || d_src == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const size_t bytes = static_cast<size_t>(dim_x) * dim_y * dim_z * sizeof(float);
    PARIS_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    paris_hip_note_host_use(ctx, h_dst, paris_hip_ctx::USED_COMPUTE);
    return paris_hip_finish(ctx);
}

// ---- subvolume split ---------------------------------------------------------------------------------

// src/cuda/subvolume_information.cpp:63-118: start with one slab per device and double the slab count until
// (volume + 10 projections) / devices fits the free memory of every device. 64-bit sizes; the
// reference's trial allocation is replaced by the free-memory test alone with a 5 % safety margin.
extern "C" int paris_hip_make_subvolume_information_reserving(const paris_volume_geometry* vol_geo,
                                                              const paris_detector_geometry* det_geo, int n_devices,
                                                              size_t reserve_bytes, paris_subvolume_info* out)
{
    if(vol_geo == nullptr || det_geo == nullptr || out == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    int devices = 0;
    if(int rc = paris_hip_device_count(&devices))
        return rc;
    if(devices == 0)
        return PARIS_HIP_ERROR_NO_DEVICE;
    if(n_devices > 0)
        devices = std::min(devices, n_devices);

    const size_t vol = static_cast<size_t>(vol_geo->dim_x) * vol_geo->dim_y * vol_geo->dim_z * sizeof(float); // :53
    const size_t proj = static_cast<size_t>(det_geo->n_row) * det_geo->n_col * sizeof(float);                // :54
    // :73-77 charges every device (volume + 10 projections) / devices. A driver that keeps more than that beside the slab
    // (upload slots, half-precision copies, the deferral ring) passes its per-device total as reserve_bytes: unlike the
    // slab it does not shrink when the slab count doubles.
    const size_t vol_dev = vol / static_cast<size_t>(devices);
    const size_t fixed = std::max(10u * proj / static_cast<size_t>(devices), reserve_bytes);
    uint32_t vols_needed = static_cast<uint32_t>(devices);                                                     // :79

    int current = 0;
    (void)hipGetDevice(&current);
    int physical = 0;
    if(int rc = physical_device_count(&physical))
        return rc;
    for(int d = 0; d < devices; ++d)
    {
        PARIS_HIP_TRY(hipSetDevice(d % physical));
        size_t mem_free = 0, mem_total = 0;
        PARIS_HIP_TRY(hipMemGetInfo(&mem_free, &mem_total));
        mem_free -= mem_free / 20u;
        if(reserve_bytes != 0 && reserve_bytes >= mem_free)
        {
            (void)hipSetDevice(current);
            return static_cast<int>(hipErrorOutOfMemory); // the driver's own buffers do not fit: it must shrink them first
        }
        while(vol_dev / (vols_needed / static_cast<uint32_t>(devices)) + fixed >= mem_free && vols_needed < vol_geo->dim_z) // :91-95
            vols_needed *= 2;
    }
    (void)hipSetDevice(current);
    if(vols_needed > vol_geo->dim_z)
        vols_needed = vol_geo->dim_z ? vol_geo->dim_z : 1u;

    out->geo.dim_x = vol_geo->dim_x; // :112-116
    out->geo.dim_y = vol_geo->dim_y;
    out->geo.dim_z = vol_geo->dim_z / vols_needed;
    out->geo.remainder = vol_geo->dim_z % vols_needed;
    out->num = static_cast<int>(vols_needed);
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_make_subvolume_information(const paris_volume_geometry* vol_geo,
                                                    const paris_detector_geometry* det_geo, int n_devices,
                                                    paris_subvolume_info* out)
{
    return paris_hip_make_subvolume_information_reserving(vol_geo, det_geo, n_devices, 0, out);
}

extern "C" int paris_hip_device_memory(int device, size_t* free_bytes, size_t* total_bytes)
{
    if(free_bytes == nullptr || total_bytes == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    int physical = 0;
    if(int rc = physical_device_count(&physical))
        return rc;
    if(physical == 0)
        return PARIS_HIP_ERROR_NO_DEVICE;
    if(device < 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    int current = 0;
    (void)hipGetDevice(&current);
    PARIS_HIP_TRY(hipSetDevice(device % physical));
    const hipError_t err = hipMemGetInfo(free_bytes, total_bytes);
    (void)hipSetDevice(current);
    PARIS_HIP_TRY(err);
    return PARIS_HIP_SUCCESS;
}

// ---- diagnostics ---------------------------------------------------------------------------------------

extern "C" const char* paris_hip_strerror(int status)
{
    switch(status)
    {
        case PARIS_HIP_SUCCESS: return "success";
        case PARIS_HIP_ERROR_INVALID_ARGUMENT: return "paris_hip: invalid argument";
        case PARIS_HIP_ERROR_NO_DEVICE: return "paris_hip: no HIP device";
        case PARIS_HIP_ERROR_UNSUPPORTED: return "paris_hip: unsupported size";
        default: return hipGetErrorString(static_cast<hipError_t>(status));
    }
}

extern "C" const char* paris_hip_version(void)
{
    return "0.1.0";
}
