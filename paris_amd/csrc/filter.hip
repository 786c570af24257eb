// Ramp row filter for gfx950: filter generation and zero-padded FFT filtering of detector rows.
//
// Replaces paris::openmp::make_filter / apply_filter (src/openmp/filtering.cpp:139-219) and
// paris::cuda::make_filter / apply_filter (src/cuda/filtering.cu:172-261) behind paris_hip_make_filter /
// paris_hip_apply_filter. The reference runs >= 6 full passes over a padded copy of the projection (expand,
// r2c, multiply, c2r, shrink, normalise) through FFTW / cuFFT; here one workgroup loads two detector rows
// straight into LDS, transforms, scales and inverse-transforms them there, and writes the first dim_x samples
// back: 8 B of HBM traffic per pixel.
//
// Algorithm (DESIGN.md "Filter kernel"):
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
    __global__ void make_filter_kernel(float* __restrict__ k, const float2* __restrict__ tw, uint32_t log2n, float tau)
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
                k[f] = tau * fabsf(sqrtf(v.x * v.x + v.y * v.y)); // :157
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
        it = ctx->plans.emplace(n, plan).first;
    }
    *out = &it->second;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_make_filter(paris_hip_ctx* ctx, uint32_t size, float tau, float** d_k)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_k == nullptr || !is_pow2(size) || size < MIN_N || size > MAX_N)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = ensure_lds_limit(ctx))
        return rc;
    paris_hip_fft_plan* plan = nullptr;
    if(int rc = paris_hip_get_plan(ctx, size, &plan))
        return rc;
    float* k = nullptr;
    PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&k), (size / 2 + 1) * sizeof(float)));
    hipLaunchKernelGGL(make_filter_kernel, dim3(1), dim3(threads_for(size)), size * sizeof(float2), ctx->stream, k,
                       plan->d_twiddle, ilog2(size), tau);
    *d_k = k;
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_apply_filter(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                      const float* d_k, uint32_t filter_size, uint32_t n_col)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_p == nullptr || d_k == nullptr || !is_pow2(filter_size) || filter_size < MIN_N || filter_size > MAX_N)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(dim_x > filter_size || n_col != dim_y || pitch < static_cast<size_t>(dim_x) * sizeof(float)
       || pitch % sizeof(float) != 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(dim_x == 0 || dim_y == 0)
        return paris_hip_finish(ctx);
    if(int rc = ensure_lds_limit(ctx))
        return rc;
    paris_hip_fft_plan* plan = nullptr;
    if(int rc = paris_hip_get_plan(ctx, filter_size, &plan))
        return rc;
    hipLaunchKernelGGL(apply_filter_kernel, dim3((dim_y + 1u) / 2u), dim3(threads_for(filter_size)),
                       filter_size * sizeof(float2), ctx->stream, d_p, static_cast<uint32_t>(pitch / sizeof(float)),
                       dim_x, dim_y, d_k, plan->d_twiddle, ilog2(filter_size));
    return paris_hip_finish(ctx);
}
