// Ramp row filter for gfx950: filter generation and zero-padded FFT filtering of detector rows.
//
// Replaces paris::openmp::make_filter / apply_filter (src/openmp/filtering.cpp:139-219) and
// paris::cuda::make_filter / apply_filter (src/cuda/filtering.cu:172-261) behind paris_hip_make_filter /
// paris_hip_apply_filter. The reference runs >= 6 full passes over a padded copy of the projection (expand,
// r2c, multiply, c2r, shrink, normalise) through FFTW / cuFFT; here one workgroup loads two detector rows
// straight into LDS, transforms, scales and inverse-transforms them there, and writes the first dim_x samples
// back: 8 B of HBM traffic per pixel.
//
// Algorithm (profiles/HISTORY.md 4.2):
//   - two real rows a, b are packed as z = a + i b. The filter is a real, even multiplier K, so
//     IFFT(K * FFT(z)) = filt(a) + i filt(b): no split/merge of the spectra is needed;
//   - forward transform = radix-2 decimation in frequency (natural order in, bit-reversed order out),
//     the multiply indexes K by the bit-reversed position, the inverse = radix-2 decimation in time
//     (bit-reversed in, natural out): no reordering pass at all;
//   - twiddles exp(-2 pi i k / N) are rounded from double on the host, once per ctx and FFT length.
// Results agree with the OpenMP backend to FFT rounding (its FFT is FFTW3f, a third-party library).
#include "paris_hip_internal.h"

#include <cmath>
#include <vector>

namespace
{
    constexpr uint32_t MIN_N = 8;
    constexpr uint32_t MAX_N = 16384;

    __device__ __forceinline__ float2 cmul(float2 a, float2 b)
    {
        return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    }

    // in-LDS forward transform: natural order in, bit-reversed order out
    __device__ __forceinline__ void fft_dif(float2* x, const float2* __restrict__ tw, uint32_t log2n, uint32_t tid,
                                            uint32_t nthreads)
    {
        const uint32_t nb = 1u << (log2n - 1); // butterflies per stage
        for(uint32_t lh = log2n - 1;; --lh)
        {
            const uint32_t half = 1u << lh;
            for(uint32_t bf = tid; bf < nb; bf += nthreads)
            {
                const uint32_t pos = bf & (half - 1u);
                const uint32_t i = ((bf >> lh) << (lh + 1)) + pos;
                const uint32_t j = i + half;
                const float2 w = tw[pos << (log2n - 1 - lh)];
                const float2 a = x[i];
                const float2 c = x[j];
                x[i] = make_float2(a.x + c.x, a.y + c.y);
                x[j] = cmul(make_float2(a.x - c.x, a.y - c.y), w);
            }
            __syncthreads();
            if(lh == 0)
                break;
        }
    }

    // in-LDS unnormalised inverse transform: bit-reversed order in, natural order out
    __device__ __forceinline__ void fft_dit_inverse(float2* x, const float2* __restrict__ tw, uint32_t log2n,
                                                    uint32_t tid, uint32_t nthreads)
    {
        const uint32_t nb = 1u << (log2n - 1);
        for(uint32_t lh = 0; lh < log2n; ++lh)
        {
            const uint32_t half = 1u << lh;
            for(uint32_t bf = tid; bf < nb; bf += nthreads)
            {
                const uint32_t pos = bf & (half - 1u);
                const uint32_t i = ((bf >> lh) << (lh + 1)) + pos;
                const uint32_t j = i + half;
                float2 w = tw[pos << (log2n - 1 - lh)];
                w.y = -w.y;
                const float2 a = x[i];
                const float2 c = cmul(x[j], w);
                x[i] = make_float2(a.x + c.x, a.y + c.y);
                x[j] = make_float2(a.x - c.x, a.y - c.y);
            }
            __syncthreads();
        }
    }

    __device__ __forceinline__ uint32_t bit_reverse(uint32_t p, uint32_t log2n)
    {
        return __brev(p) >> (32u - log2n);
    }

    // src/openmp/filtering.cpp:52-73 + :139-165 -- one workgroup
    __global__ void make_filter_kernel(float* __restrict__ k, const float2* __restrict__ tw, uint32_t log2n, float tau, int window)
    {
        extern __shared__ __attribute__((aligned(16))) float2 fx[];
        const uint32_t n = 1u << log2n;
        const uint32_t tid = threadIdx.x;
        const uint32_t nthreads = blockDim.x;
        const int32_t j0 = -(static_cast<int32_t>(n) - 2) / 2; // :55
        const float pi_f = static_cast<float>(M_PI);
        for(uint32_t x = tid; x < n; x += nthreads)
        {
            const int32_t j = j0 + static_cast<int32_t>(x);
            float r;
            if(j == 0)
                r = (1.f / 8.f) * (1.f / (tau * tau)); // :64
            else if(j % 2 == 0)
                r = 0.f; // :68
            else
                r = -(1.f / (2.f * static_cast<float>(j * j) * (pi_f * pi_f) * (tau * tau))); // :70
            fx[x] = make_float2(r, 0.f);
        }
        __syncthreads();
        fft_dif(fx, tw, log2n, tid, nthreads);
        for(uint32_t p = tid; p < n; p += nthreads)
        {
            const uint32_t f = bit_reverse(p, log2n);
            if(f <= n / 2)
            {
                const float2 v = fx[p];
                float kf = tau * fabsf(sqrtf(v.x * v.x + v.y * v.y)); // :157
                if(window == PARIS_HIP_WINDOW_SHEPP_LOGAN && f != 0u)
                {
                    // extension (the reference implements the ramp only): Shepp-Logan window
                    // sinc(pi f / N): 1 at DC, 2/pi at the Nyquist bin f = N/2
                    const float x = pi_f * static_cast<float>(f) / static_cast<float>(n);
                    kf *= sinf(x) / x;
                }
                k[f] = kf;
            }
        }
    }

    // src/openmp/filtering.cpp:167-219 for rows 2*blockIdx.x and 2*blockIdx.x + 1
    __global__ void apply_filter_kernel(float* __restrict__ p, uint32_t pitch_f, uint32_t dim_x, uint32_t dim_y,
                                        const float* __restrict__ k, const float2* __restrict__ tw, uint32_t log2n)
    {
        extern __shared__ __attribute__((aligned(16))) float2 fx[];
        const uint32_t n = 1u << log2n;
        const uint32_t tid = threadIdx.x;
        const uint32_t nthreads = blockDim.x;
        const uint32_t row_a = 2u * blockIdx.x;
        const uint32_t row_b = row_a + 1u;
        const bool has_b = row_b < dim_y;
        float* pa = p + static_cast<size_t>(row_a) * pitch_f;
        float* pb = p + static_cast<size_t>(row_b) * pitch_f;

        // expand :75-90 (zero padding), two rows packed as re / im
        for(uint32_t s = tid; s < n; s += nthreads)
        {
            float a = 0.f, b = 0.f;
            if(s < dim_x)
            {
                a = pa[s];
                if(has_b)
                    b = pb[s];
            }
            fx[s] = make_float2(a, b);
        }
        __syncthreads();

        fft_dif(fx, tw, log2n, tid, nthreads); // forward :208

        // do_filtering :92-105 -- K is real and even: K[n - f] = K[f]
        for(uint32_t pos = tid; pos < n; pos += nthreads)
        {
            const uint32_t f = bit_reverse(pos, log2n);
            const float kv = k[f <= n / 2 ? f : n - f];
            float2 v = fx[pos];
            v.x *= kv;
            v.y *= kv;
            fx[pos] = v;
        }
        __syncthreads();

        fft_dit_inverse(fx, tw, log2n, tid, nthreads); // inverse :214

        // shrink :107-118 + normalize :120-131 (n is a power of two: the division is an exact scaling)
        const float n_f = static_cast<float>(n);
        for(uint32_t s = tid; s < dim_x; s += nthreads)
        {
            const float2 v = fx[s];
            pa[s] = v.x / n_f;
            if(has_b)
                pb[s] = v.y / n_f;
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // Radix-16 row filter for N = 2^LOG2N >= 1024: N/16 threads, each holding 16 complex values in registers and doing
    // four radix-2 stages per pass, so a 4096-point transform pair crosses LDS 4 times instead of 24 (and the first
    // pass reads the detector rows straight from global memory, the last writes them back):
    //   forward  : first pass (LOG2N - 4*(NPASS-1) stages, strided by N/2^R), NPASS-2 middle passes (4 stages)
    //   fused    : last 4 forward stages + multiply by K[bitrev] + first 4 inverse stages on 16 contiguous values
    //   inverse  : NPASS-2 middle passes, last pass (the remaining stages) with the 1/N scaling and the store
    // Same transform pair as fft_dif / fft_dit_inverse above (decimation in frequency, then in time, no reordering).
    // A stage's twiddle is base_t * (a 16th root of unity): base_t comes from the table (one load per stage and
    // thread), the roots are compile-time constants.
    // ------------------------------------------------------------------------------------------------------------
#ifdef PARIS_HIP_EXPERIMENTS // the first radix-16 kernel (twiddles formed per stage): superseded by filter_fused.hip; cross-checks and tools only
#include "experiments/filter_r16.inc"
#endif

    inline bool is_pow2(uint32_t v) { return v != 0 && (v & (v - 1)) == 0; }

    inline uint32_t ilog2(uint32_t v)
    {
        uint32_t l = 0;
        while((1u << l) < v)
            ++l;
        return l;
    }

    inline uint32_t threads_for(uint32_t n)
    {
        uint32_t t = n / 8;
        if(t < 64)
            t = 64;
        if(t > 1024)
            t = 1024;
        return t;
    }

    int ensure_lds_limit(paris_hip_ctx* ctx)
    {
        // N = 16384 needs 128 KiB of dynamic LDS (gfx950: 160 KiB per CU); the attribute is per device
        if(!ctx->filter_lds_attr_set)
        {
            PARIS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(apply_filter_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, MAX_N * sizeof(float2)));
            PARIS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(make_filter_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, MAX_N * sizeof(float2)));
            ctx->filter_lds_attr_set = true;
        }
        return PARIS_HIP_SUCCESS;
    }
}

int paris_hip_get_plan(paris_hip_ctx* ctx, uint32_t n, paris_hip_fft_plan** out)
{
    auto it = ctx->plans.find(n);
    if(it == ctx->plans.end())
    {
        std::vector<float2> tw(n / 2);
        for(uint32_t i = 0; i < n / 2; ++i)
        {
            const double a = -2.0 * M_PI * static_cast<double>(i) / static_cast<double>(n);
            tw[i] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
        }
        paris_hip_fft_plan plan;
        PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&plan.d_twiddle), tw.size() * sizeof(float2)));
        // one-off, tiny: a synchronous copy keeps the pageable staging vector alive long enough
        PARIS_HIP_TRY(hipMemcpy(plan.d_twiddle, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
        if(int rc = paris_hip_fused_filter_tables(ctx, n, &plan)) // n >= 1024: the fused kernel's inter-pass twiddles
        {
            // all or nothing: a plan with some of its tables missing must never reach a kernel
            (void)hipFree(plan.d_twiddle);
            if(plan.d_tab_first != nullptr)
                (void)hipFree(plan.d_tab_first);
            for(float2* t : plan.d_tab_mid)
                if(t != nullptr)
                    (void)hipFree(t);
            return rc;
        }
        it = ctx->plans.emplace(n, plan).first;
    }
    *out = &it->second;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_make_filter(paris_hip_ctx* ctx, uint32_t size, float tau, float** d_k)
{
    return paris_hip_make_filter_windowed(ctx, size, tau, PARIS_HIP_WINDOW_RAMP, d_k);
}

extern "C" int paris_hip_make_filter_windowed(paris_hip_ctx* ctx, uint32_t size, float tau, int window, float** d_k)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_k == nullptr || !is_pow2(size) || size < MIN_N || size > MAX_N
       || (window != PARIS_HIP_WINDOW_RAMP && window != PARIS_HIP_WINDOW_SHEPP_LOGAN))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = ensure_lds_limit(ctx))
        return rc;
    paris_hip_fft_plan* plan = nullptr;
    if(int rc = paris_hip_get_plan(ctx, size, &plan))
        return rc;
    float* k = nullptr;
    PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&k), (size / 2 + 1) * sizeof(float)));
    hipLaunchKernelGGL(make_filter_kernel, dim3(1), dim3(threads_for(size)), size * sizeof(float2), ctx->stream, k,
                       plan->d_twiddle, ilog2(size), tau, window);
    paris_hip_filter_info info;
    info.size = size;
    if(size >= 1024u) // the fused kernel reads K in the order its middle pass holds the frequencies
        if(int rc = paris_hip_fused_filter_permute_k(ctx, k, size, &info.d_kp))
        {
            (void)hipFree(k);
            return rc;
        }
    ctx->filters[k] = info;
    *d_k = k;
    return paris_hip_finish(ctx);
}

// which kernel serves a filter call: the table-twiddle kernel of filter_fused.hip needs a K made by paris_hip_make_filter*
// on this ctx (its permuted copy) and a length >= 1024; filter_variant 1 / 2 force the radix-2 / the first radix-16 kernel
static const paris_hip_filter_info* fused_filter_of(paris_hip_ctx* ctx, const float* d_k, uint32_t filter_size)
{
    if(ctx->filter_variant != 0 || filter_size < 1024u)
        return nullptr;
    auto it = ctx->filters.find(d_k);
    if(it == ctx->filters.end() || it->second.size != filter_size || it->second.d_kp == nullptr)
        return nullptr;
    return &it->second;
}

extern "C" int paris_hip_apply_filter(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                      const float* d_k, uint32_t filter_size, uint32_t n_col)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_p == nullptr || d_k == nullptr || !is_pow2(filter_size) || filter_size < MIN_N || filter_size > MAX_N
       || dim_x > filter_size || n_col != dim_y || pitch < static_cast<size_t>(dim_x) * sizeof(float) || pitch % sizeof(float) != 0)
    {
        (void)paris_hip_flush_pending_weight(ctx);
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    }
    if(dim_x == 0 || dim_y == 0)
    {
        if(int rc = paris_hip_flush_pending_weight(ctx))
            return rc;
        return paris_hip_finish(ctx);
    }
    if(int rc = ensure_lds_limit(ctx))
        return rc;
    if(int rc = paris_hip_projection_guard(ctx, d_p, pitch * dim_y, ctx->stream, true))
        return rc;
    paris_hip_fft_plan* plan = nullptr;
    if(int rc = paris_hip_get_plan(ctx, filter_size, &plan))
        return rc;
    const uint32_t pitch_f = static_cast<uint32_t>(pitch / sizeof(float));
    const uint32_t log2n = ilog2(filter_size);
    if(const paris_hip_filter_info* info = fused_filter_of(ctx, d_k, filter_size))
    {
        // a weighting held back for exactly these rows (stage fusion) rides along in the load
        auto& w = ctx->pending_weight;
        if(w.active && w.filter) // a filter is held back already (filter deferral): it comes first
            if(int rc = paris_hip_flush_pending_weight(ctx))
                return rc;
        const bool fuse = w.active && w.pitch == pitch && w.dim_x == dim_x && w.row_count == dim_y
                          && reinterpret_cast<char*>(w.d_p) + static_cast<size_t>(w.row_first) * w.pitch == reinterpret_cast<char*>(d_p);
        bool hold = fuse && ctx->filter_deferral != 0 && ctx->defer_depth > 1u && !(ctx->flags & PARIS_HIP_CTX_SYNCHRONOUS);
        if(hold && ctx->filter_deferral == 2)
        {
            // only where nobody can tell: the whole buffer is one of paris_hip_malloc_projection's with the pitch a deferred
            // backprojection takes BY REFERENCE -- the filter then runs in place, in the group's launch, and every other touch of the
            // buffer runs it first (paris_hip_projection_guard)
            auto mine = ctx->proj_allocs.find(d_p);
            hold = ctx->defer_refs != 0 && w.row_first == 0u && mine != ctx->proj_allocs.end()
                   && pitch == ((static_cast<size_t>(dim_x) * sizeof(float) + 255u) & ~static_cast<size_t>(255u))
                   && mine->second.bytes >= pitch * static_cast<size_t>(dim_y);
        }
        if(hold)
        {
            // filter deferral: held back with the weighting. A backprojection of this projection takes both along into its ring slot
            // (backproject.hip: defer_backproject); anything else runs them first, in place (paris_hip_flush_pending_weight)
            w.filter = true;
            w.d_kp = info->d_kp;
            w.plan = plan;
            w.filter_size = filter_size;
            return PARIS_HIP_SUCCESS;
        }
        if(fuse)
            w.active = false;
        else if(int rc = paris_hip_flush_pending_weight(ctx))
            return rc;
        if(int rc = paris_hip_fused_filter_launch(ctx, d_p, pitch_f, dim_x, dim_y, fuse ? w.row_first : 0u, fuse, w.h_min, w.v_min, w.d_sd,
                                                  w.l_px_row, w.l_px_col, info->d_kp, plan, filter_size, nullptr, 0u))
            return rc;
        if(int rc = paris_hip_note_projection_use(ctx, d_p, pitch * dim_y))
            return rc;
        return paris_hip_finish(ctx);
    }
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
#ifdef PARIS_HIP_EXPERIMENTS
    if(ctx->filter_variant != 1 && log2n >= 10u)
    {
        int rc = PARIS_HIP_SUCCESS;
        switch(log2n)
        {
            case 10: rc = launch_r16<10>(ctx, d_p, pitch_f, dim_x, dim_y, d_k, plan->d_twiddle); break;
            case 11: rc = launch_r16<11>(ctx, d_p, pitch_f, dim_x, dim_y, d_k, plan->d_twiddle); break;
            case 12: rc = launch_r16<12>(ctx, d_p, pitch_f, dim_x, dim_y, d_k, plan->d_twiddle); break;
            case 13: rc = launch_r16<13>(ctx, d_p, pitch_f, dim_x, dim_y, d_k, plan->d_twiddle); break;
            default: rc = launch_r16<14>(ctx, d_p, pitch_f, dim_x, dim_y, d_k, plan->d_twiddle); break;
        }
        if(rc != PARIS_HIP_SUCCESS)
            return rc;
        if(int rc2 = paris_hip_note_projection_use(ctx, d_p, pitch * dim_y))
            return rc2;
        return paris_hip_finish(ctx);
    }
#endif // (product build: a K this ctx did not make, or a length below 1024, runs the radix-2 kernel)
    hipLaunchKernelGGL(apply_filter_kernel, dim3((dim_y + 1u) / 2u), dim3(threads_for(filter_size)),
                       filter_size * sizeof(float2), ctx->stream, d_p, pitch_f, dim_x, dim_y, d_k, plan->d_twiddle, log2n);
    if(int rc = paris_hip_note_projection_use(ctx, d_p, pitch * dim_y))
        return rc;
    return paris_hip_finish(ctx);
}

// Extension: cosine weighting and row filter of rows [row_first, row_first + row_count) in ONE launch (filter_fused.hip);
// d_half != NULL: the filtered rows are stored as IEEE half into d_half (same row numbering, half_pitch bytes per row)
// and the fp32 rows are left as they were. Needs a K of paris_hip_make_filter* and filter_size >= 1024
// (PARIS_HIP_ERROR_UNSUPPORTED otherwise: the caller then runs the two stages).
extern "C" int paris_hip_weight_filter_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y, uint32_t row_first,
                                            uint32_t row_count, float h_min, float v_min, float d_sd, float l_px_row, float l_px_col,
                                            const float* d_k, uint32_t filter_size, uint16_t* d_half, size_t half_pitch)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
    if(d_p == nullptr || d_k == nullptr || !is_pow2(filter_size) || filter_size < MIN_N || filter_size > MAX_N || dim_x > filter_size
       || pitch < static_cast<size_t>(dim_x) * sizeof(float) || pitch % sizeof(float) != 0 || row_first > dim_y || row_count > dim_y - row_first
       || (d_half != nullptr && (half_pitch < static_cast<size_t>(dim_x) * sizeof(uint16_t) || half_pitch % sizeof(uint16_t) != 0)))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(dim_x == 0 || row_count == 0)
        return paris_hip_finish(ctx);
    const paris_hip_filter_info* info = fused_filter_of(ctx, d_k, filter_size);
    if(info == nullptr)
        return PARIS_HIP_ERROR_UNSUPPORTED;
    paris_hip_fft_plan* plan = nullptr;
    if(int rc = paris_hip_get_plan(ctx, filter_size, &plan))
        return rc;
    float* rows = reinterpret_cast<float*>(reinterpret_cast<char*>(d_p) + static_cast<size_t>(row_first) * pitch);
    uint16_t* half_rows = d_half ? reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(d_half) + static_cast<size_t>(row_first) * half_pitch) : nullptr;
    if(int rc = paris_hip_projection_guard(ctx, rows, pitch * row_count, ctx->stream, d_half == nullptr))
        return rc;
    if(half_rows != nullptr)
        if(int rc = paris_hip_projection_guard(ctx, half_rows, half_pitch * row_count, ctx->stream, true))
            return rc;
    if(int rc = paris_hip_fused_filter_launch(ctx, rows, static_cast<uint32_t>(pitch / sizeof(float)), dim_x, row_count, row_first, true, h_min, v_min,
                                              d_sd, l_px_row, l_px_col, info->d_kp, plan, filter_size, half_rows,
                                              static_cast<uint32_t>(half_pitch / sizeof(uint16_t))))
        return rc;
    if(int rc = paris_hip_note_projection_use(ctx, rows, pitch * row_count))
        return rc;
    if(half_rows != nullptr) // (the half-precision destination is written by this launch too)
        if(int rc = paris_hip_note_projection_use(ctx, half_rows, half_pitch * row_count))
            return rc;
    return paris_hip_finish(ctx);
}

// Extension: paris_hip_weight_filter_rows for n_frames projections in ONE launch (grid.y = frame): frame f lives frame_stride bytes
// behind frame f - 1 (d_half likewise, half_frame_stride bytes apart). What a driver that holds a group of uploaded frames wants
// before its fused backprojection: for small detectors a launch per frame is mostly launch latency (512^2: 256 workgroups of one
// wave, 6.9 us each; sixteen frames in one launch cost little more than one). Bit-identical to the per-frame calls (the same
// workgroup per row pair). filter_size < 1024 or a foreign K: the frames are run one by one through the separate stages' entry.
extern "C" int paris_hip_weight_filter_batch(paris_hip_ctx* ctx, float* d_p, size_t pitch, size_t frame_stride, uint32_t n_frames, uint32_t dim_x,
                                             uint32_t dim_y, uint32_t row_first, uint32_t row_count, float h_min, float v_min, float d_sd,
                                             float l_px_row, float l_px_col, const float* d_k, uint32_t filter_size, uint16_t* d_half,
                                             size_t half_pitch, size_t half_frame_stride)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
    if(d_p == nullptr || d_k == nullptr || !is_pow2(filter_size) || filter_size < MIN_N || filter_size > MAX_N || dim_x > filter_size
       || pitch < static_cast<size_t>(dim_x) * sizeof(float) || pitch % sizeof(float) != 0 || row_first > dim_y || row_count > dim_y - row_first
       || (d_half != nullptr && (half_pitch < static_cast<size_t>(dim_x) * sizeof(uint16_t) || half_pitch % sizeof(uint16_t) != 0))
       || n_frames > 65535u)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(n_frames > 1u && (frame_stride % sizeof(float) != 0 || frame_stride < pitch * static_cast<size_t>(dim_y)
                         || (d_half != nullptr && (half_frame_stride % sizeof(uint16_t) != 0 || half_frame_stride < half_pitch * static_cast<size_t>(dim_y)))))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT; // frames must not overlap
    if(dim_x == 0 || row_count == 0 || n_frames == 0)
        return paris_hip_finish(ctx);
    const paris_hip_filter_info* info = fused_filter_of(ctx, d_k, filter_size);
    if(info == nullptr)
        return PARIS_HIP_ERROR_UNSUPPORTED;
    paris_hip_fft_plan* plan = nullptr;
    if(int rc = paris_hip_get_plan(ctx, filter_size, &plan))
        return rc;
    float* rows = reinterpret_cast<float*>(reinterpret_cast<char*>(d_p) + static_cast<size_t>(row_first) * pitch);
    uint16_t* half_rows = d_half ? reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(d_half) + static_cast<size_t>(row_first) * half_pitch) : nullptr;
    if(int rc = paris_hip_projection_guard(ctx, rows, pitch * row_count, ctx->stream, d_half == nullptr))
        return rc;
    if(half_rows != nullptr)
        if(int rc = paris_hip_projection_guard(ctx, half_rows, half_pitch * row_count, ctx->stream, true))
            return rc;
    if(int rc = paris_hip_fused_filter_launch(ctx, rows, static_cast<uint32_t>(pitch / sizeof(float)), dim_x, row_count, row_first, true, h_min, v_min,
                                              d_sd, l_px_row, l_px_col, info->d_kp, plan, filter_size, half_rows,
                                              static_cast<uint32_t>(half_pitch / sizeof(uint16_t)), n_frames, frame_stride / sizeof(float),
                                              half_frame_stride / sizeof(uint16_t)))
        return rc;
    for(uint32_t f = 0; f < n_frames && (!ctx->upload_targets.empty() || !ctx->proj_allocs.empty()); ++f)
    {
        if(int rc = paris_hip_note_projection_use(ctx, reinterpret_cast<char*>(rows) + f * frame_stride, pitch * row_count))
            return rc;
        if(half_rows != nullptr)
            if(int rc = paris_hip_note_projection_use(ctx, reinterpret_cast<char*>(half_rows) + f * half_frame_stride, half_pitch * row_count))
                return rc;
    }
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_set_filter_variant(paris_hip_ctx* ctx, int variant)
{
    if(ctx == nullptr || variant < 0 || variant > 2)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
#ifndef PARIS_HIP_EXPERIMENTS
    if(variant == 2) // the first radix-16 kernel lives in the experiments build only
        return PARIS_HIP_ERROR_UNSUPPORTED;
#endif
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
    ctx->filter_variant = variant;
    return PARIS_HIP_SUCCESS;
}

// PARIS_HIP_CTX_WARM: a query of one kernel of this translation unit makes the runtime load its code object now
void paris_hip_warm_filter()
{
    hipFuncAttributes a{};
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&make_filter_kernel));
}
