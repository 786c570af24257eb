// Device validators of the hand-expanded IEEE sequences of ieee_lean.h.
//
// The backprojection's per-column divisions d_sd / (s + d_so) and d_so / (s + d_so) (src/openmp/backprojection.cpp:125,139) and the
// weighting's d_sd / sqrt(d_sd^2 + h^2 + v^2) (src/openmp/weighting.cpp:52) are correctly rounded IEEE operations in the reference.
// The kernels evaluate them with shorter sequences that have the bits of the compiler's expansion "for ordinary operands". That is a
// property of this compiler and of the operand range, so it is CHECKED, on the device, for every fp32 operand a launch can produce,
// before a kernel is allowed to use the short form -- the way backproject.hip's fastdiv_validate_kernel proves the division by the
// pixel pitch. The ranges are small (a launch's denominators span ~4.4 binades = 3.7e7 floats; a detector's radicands usually a
// fraction of one binade), so a check costs microseconds of GPU time; results are cached per process, device and range.
#include "ieee_lean.h"
#include "paris_hip_internal.h"

#include <cmath>
#include <cstring>
#include <mutex>

namespace
{
    // every fp32 den with bits in [first, first + count): both lean quotients against the compiler's IEEE divisions
    __global__ void __launch_bounds__(256) lean_div_validate_kernel(uint32_t first, uint64_t count, float d_sd, float d_so,
                                                                    unsigned long long* mismatches)
    {
        unsigned int bad = 0;
        for(uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; i < count; i += static_cast<uint64_t>(gridDim.x) * 256u)
        {
            const float den = __uint_as_float(first + static_cast<uint32_t>(i));
            const float r = paris_lean::refined_rcp(den);
            const float got_sd = paris_lean::div_with_rcp(d_sd, den, r);
            const float got_so = paris_lean::div_with_rcp(d_so, den, r);
            const float want_sd = d_sd / den;
            const float want_so = d_so / den;
            bad += (__float_as_uint(got_sd) != __float_as_uint(want_sd)) + (__float_as_uint(got_so) != __float_as_uint(want_so));
        }
        if(bad)
            atomicAdd(mismatches, static_cast<unsigned long long>(bad));
    }

    // every fp32 radicand q with bits in [first, first + count): the lean sqrt against sqrtf, and the lean division of d_sd by the
    // (correct) root against the compiler's IEEE division -- the two steps of `weighted` in filter_fused.hip
    __global__ void __launch_bounds__(256) lean_weight_validate_kernel(uint32_t first, uint64_t count, float d_sd, unsigned long long* mismatches)
    {
        unsigned int bad = 0;
        for(uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; i < count; i += static_cast<uint64_t>(gridDim.x) * 256u)
        {
            const float q = __uint_as_float(first + static_cast<uint32_t>(i));
            const float want_root = sqrtf(q);
            const float got_root = paris_lean::sqrt_rn_safe_range(q);
            const float want_w = d_sd / want_root;
            const float got_w = paris_lean::div_rn_safe_range(d_sd, want_root);
            bad += (__float_as_uint(got_root) != __float_as_uint(want_root)) + (__float_as_uint(got_w) != __float_as_uint(want_w));
        }
        if(bad)
            atomicAdd(mismatches, static_cast<unsigned long long>(bad));
    }

    std::mutex lean_mutex;
    std::map<std::pair<int, std::array<uint32_t, 4>>, bool> lean_cache; // (device, key) -> validation result, process-wide

    // one validator launch (capi.hip: paris_hip_run_check puts it on the ctx's auxiliary stream -- never the caller's, which may be
    // capturing or hold queued work the caller does not want to wait for -- and reads the mismatch count back)
    struct lean_launch
    {
        int kind; // 0: division, 1: weighting
        uint32_t first;
        uint64_t count;
        float a, b;
    };

    void lean_enqueue(hipStream_t s, unsigned long long* counter, const void* arg)
    {
        const lean_launch& l = *static_cast<const lean_launch*>(arg);
        const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>(4096u, std::max<uint64_t>(1u, (l.count + 255u) / 256u)));
        if(l.kind == 0)
            hipLaunchKernelGGL(lean_div_validate_kernel, dim3(blocks), dim3(256), 0, s, l.first, l.count, l.a, l.b, counter);
        else
            hipLaunchKernelGGL(lean_weight_validate_kernel, dim3(blocks), dim3(256), 0, s, l.first, l.count, l.a, counter);
    }

    uint32_t bits_of(float x)
    {
        uint32_t u;
        std::memcpy(&u, &x, sizeof(u));
        return u;
    }

    // looks the key up in the ctx's own map (no lock on the hot path), then in the process-wide one, then asks the device
    int cached_check(paris_hip_ctx* ctx, const std::array<uint32_t, 4>& key, int kind, uint32_t first, uint64_t count, float a, float b, bool* ok)
    {
        auto it = ctx->lean_checks.find(key);
        if(it == ctx->lean_checks.end())
        {
            std::lock_guard<std::mutex> lock(lean_mutex);
            const auto pkey = std::make_pair(ctx->device, key);
            auto pit = lean_cache.find(pkey);
            if(pit == lean_cache.end())
            {
                bool exact = false, known = false;
                if(int rc = paris_hip_bind(ctx))
                    return rc;
                const lean_launch l{kind, first, count, a, b};
                if(int rc = paris_hip_run_check(ctx, key, lean_enqueue, &l, &exact, &known))
                    return rc;
                if(!known) // asynchronous validation: still running -- the compiler's IEEE forms serve meanwhile (*ok stays false)
                {
                    *ok = false;
                    return PARIS_HIP_SUCCESS;
                }
                pit = lean_cache.emplace(pkey, exact).first;
            }
            it = ctx->lean_checks.emplace(key, pit->second).first;
        }
        *ok = it->second;
        return PARIS_HIP_SUCCESS;
    }
}

// Both per-column quotients for every fp32 denominator in [0.09, 1.92] x d_so: fill_params enables lean_div only when every column of
// the launch has |s| <= 0.9 d_so (evaluated in double), i.e. den = s + d_so in [0.1, 1.9] x d_so; the margin covers the fp32 rounding
// of s and of the sum many times over.
int paris_hip_lean_division_check(paris_hip_ctx* ctx, float d_sd, float d_so, bool* ok)
{
    *ok = false;
    if(!(d_so > 0.f) || !(d_sd > 0.f) || !std::isfinite(d_so) || !std::isfinite(d_sd))
        return PARIS_HIP_SUCCESS;
    if(ctx->lean_validate == 0)
    {
        *ok = true;
        return PARIS_HIP_SUCCESS;
    }
    const float lo = 0.09f * d_so, hi = 1.92f * d_so;
    if(!(lo > 0.f) || !std::isfinite(hi) || !std::isnormal(lo))
        return PARIS_HIP_SUCCESS;
    const uint32_t first = bits_of(lo);
    const uint64_t count = static_cast<uint64_t>(bits_of(hi)) - first + 1u; // positive floats are ordered like their bits
    return cached_check(ctx, {0u, bits_of(d_sd), bits_of(d_so), 0u}, 0, first, count, d_sd, d_so, ok);
}

// sqrt and d_sd / sqrt for every fp32 radicand between d_sd^2 and the largest one of the rows being weighted (dd, q_max: the host's
// double-precision bounds; widened by 2^-10 either side against the kernel's fp32 roundings, and q_max rounded up to a sixteenth of
// a binade so that row bands of one detector share a few cache entries). Ranges of more than 2^30 floats are not validated: no lean.
int paris_hip_lean_weighting_check(paris_hip_ctx* ctx, float d_sd, double dd, double q_max, bool* ok)
{
    *ok = false;
    if(!(d_sd > 0.f) || !(dd > 0.0) || !(q_max >= dd) || !std::isfinite(q_max))
        return PARIS_HIP_SUCCESS;
    if(ctx->lean_validate == 0)
    {
        *ok = true;
        return PARIS_HIP_SUCCESS;
    }
    const float lo = static_cast<float>(dd * (1.0 - 1.0 / 1024.0)), hi = static_cast<float>(q_max * (1.0 + 1.0 / 1024.0));
    if(!std::isnormal(lo) || !std::isfinite(hi))
        return PARIS_HIP_SUCCESS;
    const uint32_t first = bits_of(lo) & ~((1u << 19) - 1u);          // down to a sixteenth of a binade
    const uint32_t last = bits_of(hi) | ((1u << 19) - 1u);            // up to one
    if(last >= 0x7f800000u)
        return PARIS_HIP_SUCCESS;
    const uint64_t count = static_cast<uint64_t>(last) - first + 1u;
    if(count > (1ull << 30))
        return PARIS_HIP_SUCCESS;
    return cached_check(ctx, {1u, bits_of(d_sd), first, last}, 1, first, count, d_sd, 0.f, ok);
}

extern "C" int paris_hip_lean_division_is_exact(paris_hip_ctx* ctx, float d_sd, float d_so, int* exact)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(exact == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    bool ok = false;
    const int saved = ctx->lean_validate, saved_async = ctx->async_validate;
    ctx->lean_validate = 1; // the question is what the device says, whatever the tuning switches: wait for it
    ctx->async_validate = 0;
    const int rc = paris_hip_lean_division_check(ctx, d_sd, d_so, &ok);
    ctx->lean_validate = saved;
    ctx->async_validate = saved_async;
    *exact = ok ? 1 : 0;
    return rc;
}

extern "C" int paris_hip_lean_weighting_is_exact(paris_hip_ctx* ctx, float d_sd, float q_lo, float q_hi, int* exact)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(exact == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    bool ok = false;
    const int saved = ctx->lean_validate, saved_async = ctx->async_validate;
    ctx->lean_validate = 1;
    ctx->async_validate = 0;
    const int rc = paris_hip_lean_weighting_check(ctx, d_sd, static_cast<double>(q_lo), static_cast<double>(q_hi), &ok);
    ctx->lean_validate = saved;
    ctx->async_validate = saved_async;
    *exact = ok ? 1 : 0;
    return rc;
}

// Test hook: 0 = the kernels trust the host's range tests alone (as in rounds 2-3), 1 (default) = the device validation decides.
extern "C" int paris_hip_set_lean_validation(paris_hip_ctx* ctx, int enable)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    ctx->lean_validate = enable ? 1 : 0;
    ctx->lean_checks.clear();
    return PARIS_HIP_SUCCESS;
}

// PARIS_HIP_CTX_WARM: a query of one kernel of this translation unit makes the runtime load its code object now
void paris_hip_warm_validate()
{
    hipFuncAttributes a{};
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&lean_div_validate_kernel));
}
