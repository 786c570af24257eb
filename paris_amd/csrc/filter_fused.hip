// Fused cosine weighting + ramp row filter for gfx950: ONE launch per projection instead of weight_kernel + apply_filter.
//
// Replaces paris::openmp::weight (src/openmp/weighting.cpp:32-57) followed by paris::openmp::apply_filter
// (src/openmp/filtering.cpp:167-219) -- in the CUDA backend src/cuda/weighting.cu:35-73 and src/cuda/filtering.cu:195-261 --
// behind paris_hip_weight + paris_hip_apply_filter (stage fusion, capi.hip) and paris_hip_weight_filter_rows.
//
// Data movement is the radix-16 row kernel's (filter.hip): one workgroup of N/16 threads per pair of detector rows packed as
// re / im, N = filter length, first pass straight from global memory, 16 complex values per thread in registers, passes
// exchanged through LDS, forward = decimation in frequency, the multiply by K, inverse = decimation in time, last pass scales
// by 1/N and stores. What is new:
//   * the weight d_sd / sqrt(d_sd^2 + h_s^2 + v_t^2) is applied to each pixel as it is loaded: the very operations of
//     weight_kernel (weight.hip) in the same order, rounded once each (no contraction), so the weighted value entering the
//     transform has the bits the unfused path would have written to memory and read back;
//   * every pass is a true radix-R butterfly: an R-point DFT whose twiddles are compile-time 16th roots of unity (1, -i and
//     (1 +- i)/sqrt 2 cost nothing or two multiplies), followed by ONE multiplication per output with the inter-pass twiddle
//     W_B^(o q), read from a per-length table computed in double precision on the host -- 266 instead of 512 floating-point
//     operations per 16 points and pass; fp32 contraction is allowed inside the transform (fused multiply-adds round less,
//     and the filter is held to a tolerance against the CPU restatement, not to bit equality: the reference FFT is FFTW3f);
//   * zero padding is exploited: with dim_x <= N/2 the upper half of the first pass's inputs is zero (no adds in its first
//     stage) and the upper half of the last pass's outputs is never stored (no subtractions in its last stage);
//   * K is read in the order the fused middle pass holds the frequencies (one 16-float run per thread) from a permuted
//     copy made once per filter, instead of 16 scattered loads per thread;
//   * optionally the last pass stores IEEE half pixels into a second buffer (BASELINE config 5) instead of fp32 in place.
#if (defined(PARIS_FILTER_TIMING_NO_COMPUTE) || defined(PARIS_FILTER_TIMING_NO_MEMORY)) && !defined(PARIS_HIP_EXPERIMENTS)
#error "the timing-only switches compute WRONG results: they exist in the experiments build only (make EXPERIMENTS=1)"
#endif
#include "paris_hip_internal.h"
#include "ieee_lean.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace
{
    // ---- weighting: src/openmp/weighting.cpp:48-54, exactly as weight_kernel evaluates it -------------------------------
    struct WeightParams
    {
        float h_min, v_min, d_sd, l_px_row, l_px_col;
        uint32_t lean; // every radicand and quotient of this projection is far from the fp32 range limits (decided on the host)
    };

    __device__ __forceinline__ float column_term(const WeightParams& w, uint32_t s) // dd + hh of detector column s
    {
#pragma clang fp contract(off)
        const float s_f = static_cast<float>(s);
        const float h_s = (w.l_px_row / 2) + s_f * w.l_px_row + w.h_min; // :48
        const float hh = h_s * h_s;
        const float dd = w.d_sd * w.d_sd;
        return dd + hh;
    }

    __device__ __forceinline__ float row_term(const WeightParams& w, uint32_t t) // v_t^2 of detector row t
    {
#pragma clang fp contract(off)
        const float t_f = static_cast<float>(t);
        const float v_t = (w.l_px_col / 2) + t_f * w.l_px_col + w.v_min; // :49
        return v_t * v_t;
    }

    // Correctly rounded sqrt and division as hipcc expands them, minus the parts that only matter at the edges of the fp32 range
    // (ieee_lean.h). The host enables them (WeightParams::lean) only after validate.hip has compared both with sqrtf and `/` for
    // every fp32 radicand between d_sd^2 and the largest one of the detector (paris_hip_lean_weighting_check).
    using paris_lean::div_rn_safe_range;
    using paris_lean::sqrt_rn_safe_range;

    __device__ __forceinline__ float weighted(const WeightParams& w, float px, float dd_hh, float vv)
    {
#pragma clang fp contract(off)
        const float q = dd_hh + vv; // :52 (dd + hh + v_t * v_t, left to right)
        const float w_st = w.lean ? div_rn_safe_range(w.d_sd, sqrt_rn_safe_range(q)) : w.d_sd / sqrtf(q);
        return px * w_st;           // :54
    }

    // ---- butterflies with compile-time twiddles ---------------------------------------------------------------------------
    // a * exp(-2 pi i M / 16) (CONJ: the conjugate root), M in [0, 8)
    template <int M, bool CONJ>
    __device__ __forceinline__ float2 mul_root16(float2 a)
    {
#pragma clang fp contract(fast)
        constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
        static_assert(M >= 0 && M < 8, "root index");
        if constexpr(M == 0)
            return a;
        else if constexpr(M == 4)
            return CONJ ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); // * (-i) / * (+i)
        else if constexpr(M == 2)
            return CONJ ? make_float2((a.x - a.y) * H, (a.x + a.y) * H) : make_float2((a.x + a.y) * H, (a.y - a.x) * H);
        else if constexpr(M == 6)
            return CONJ ? make_float2(-(a.x + a.y) * H, (a.x - a.y) * H) : make_float2((a.y - a.x) * H, -(a.x + a.y) * H);
        else
        {
            // root = c - i s (forward), c + i s (conjugate)
            constexpr float c = M == 1 ? C1 : (M == 3 ? S1 : (M == 5 ? -S1 : -C1));
            constexpr float s = (M == 1 || M == 7) ? S1 : C1;
            if(CONJ)
                return make_float2(a.x * c - a.y * s, a.y * c + a.x * s);
            return make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
        }
    }

    __device__ __forceinline__ float2 cmul_fast(float2 a, float2 b)
    {
#pragma clang fp contract(fast)
        return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    }

    __device__ __forceinline__ float2 cmul_conj_fast(float2 a, float2 b) // a * conj(b)
    {
#pragma clang fp contract(fast)
        return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
    }

    constexpr int brev(int i, int bits)
    {
        int r = 0;
        for(int b = 0; b < bits; ++b)
            r |= ((i >> b) & 1) << (bits - 1 - b);
        return r;
    }

    // one stage of the decimation-in-frequency network on E = 2^LOGR values: groups of Et = E >> T, pairs (j, j + Et/2)
    template <int LOGR, int T, bool UPPER_ZERO, int J = 0>
    __device__ __forceinline__ void dif_stage(float2 (&v)[1 << LOGR])
    {
        constexpr int E = 1 << LOGR, Et = E >> T, half = Et / 2;
        if constexpr(J < E)
        {
            if constexpr((J % Et) < half)
            {
                constexpr int M = (J % Et) * (16 / Et); // W_Et^(j') as a 16th root
                const float2 lo = v[J];
                if constexpr(UPPER_ZERO)
                {
                    v[J + half] = mul_root16<M, false>(lo); // the upper input is zero: lo + 0, (lo - 0) * w
                }
                else
                {
                    const float2 hi = v[J + half];
                    v[J] = make_float2(lo.x + hi.x, lo.y + hi.y);
                    v[J + half] = mul_root16<M, false>(make_float2(lo.x - hi.x, lo.y - hi.y));
                }
            }
            dif_stage<LOGR, T, UPPER_ZERO, J + 1>(v);
        }
    }

    // R-point forward DFT, R = 2^LOGR <= 16: natural order in, X_q in v[brev(q)]. HALF: inputs v[R/2 ..] are zero.
    template <int LOGR, bool HALF, int T = 0>
    __device__ __forceinline__ void dft_forward(float2 (&v)[1 << LOGR])
    {
        if constexpr(T < LOGR)
        {
            dif_stage<LOGR, T, HALF && T == 0>(v);
            dft_forward<LOGR, HALF, T + 1>(v);
        }
    }

    // one stage of the decimation-in-time inverse network: pairs (j, j + 2^T), conjugate roots
    template <int LOGR, int T, bool LOWER_ONLY, int J = 0>
    __device__ __forceinline__ void dit_stage(float2 (&v)[1 << LOGR])
    {
        constexpr int E = 1 << LOGR, span = 1 << T;
        if constexpr(J < E)
        {
            if constexpr((J % (2 * span)) < span)
            {
                constexpr int M = (J % span) * (16 / (2 * span));
                const float2 lo = v[J];
                const float2 c = mul_root16<M, true>(v[J + span]);
                v[J] = make_float2(lo.x + c.x, lo.y + c.y);
                if constexpr(!LOWER_ONLY)
                    v[J + span] = make_float2(lo.x - c.x, lo.y - c.y);
            }
            dit_stage<LOGR, T, LOWER_ONLY, J + 1>(v);
        }
    }

    // R-point unnormalised inverse DFT: Y_q in v[brev(q)] in, natural order out. HALF: only outputs v[0 .. R/2) are needed.
    template <int LOGR, bool HALF, int T = 0>
    __device__ __forceinline__ void dft_inverse(float2 (&v)[1 << LOGR])
    {
        if constexpr(T < LOGR)
        {
            dit_stage<LOGR, T, HALF && T == LOGR - 1>(v);
            dft_inverse<LOGR, HALF, T + 1>(v);
        }
    }

    __device__ __forceinline__ uint32_t lds_pad(uint32_t i) { return i + (i >> 4); } // one spare slot per 16: conflict-free passes

    constexpr uint32_t FRAME_TAB_MAX = 64u;

    struct FusedFilterArgs
    {
        float* p;            // first row of the band
        uint32_t pitch_f;    // floats per row
        uint32_t dim_x;      // pixels per row
        uint32_t n_rows;     // rows of the band
        uint32_t row_first;  // detector row index of p's first row (weighting)
        WeightParams w;
        const float* kp;         // K in the order of the fused pass: kp[tid * 16 + i]
        const float2* tab_first; // [(q - 1) * (N / EF) + o] = W_N^(o q), q = 1 .. EF - 1
        const float2* tab_mid[3]; // middle pass with block B = 16^(k + 2): [(q - 1) * (B / 16) + o] = W_B^(o q), q = 1 .. 15
        _Float16* half_out;      // F16OUT: destination of the filtered rows (first row of the band)
        uint32_t half_pitch;     // halves per row
        // several frames per launch (grid.y = frame): frame f's band starts frame_stride floats (half_frame_stride halves) behind
        // frame f - 1's; the detector row indices, hence the weights, are the same for every frame
        size_t frame_stride, half_frame_stride;
        // ... or, frames that do not lie one stride apart (the deferral's projections by reference: the callers' own buffers), the
        // first band row of each frame (n_tab != 0: at most FRAME_TAB_MAX frames per launch, fp32 in place)
        uint32_t n_tab;
        float* frame_tab[FRAME_TAB_MAX];
    };

    // HALF: dim_x <= N/2 (always true for the reference's filter length 2 * 2^ceil(log2 n_row)); WEIGHT: apply the cosine
    // weight in the load; F16OUT: store halves into half_out instead of fp32 in place
    template <int LOG2N, bool HALF, bool WEIGHT, bool F16OUT>
    __global__ void __launch_bounds__((1 << LOG2N) / 16, 4) filter_rows_kernel(const FusedFilterArgs a) // 4 waves per SIMD: at most 128 VGPRs
    {
        constexpr uint32_t N = 1u << LOG2N;
        constexpr int NPASS = (LOG2N + 3) / 4;
        constexpr int RF = LOG2N - 4 * (NPASS - 1); // log2 of the first / last pass's radix
        constexpr uint32_t EF = 1u << RF;
        constexpr uint32_t T = N / 16u;             // threads
        constexpr uint32_t NO = N / EF;             // offsets o of the first pass = 16^(NPASS - 1)
        extern __shared__ __attribute__((aligned(16))) float2 fx[];
        // the last middle pass's twiddles (block 256: 15 x 16 values, the same for every thread with the same tid % 16) live in
        // LDS behind the data: read at LDS latency inside the passes instead of L2 latency
        float2* ltab = fx + (N + N / 16u);
        constexpr uint32_t LTAB = 15u * 16u;

        const uint32_t tid = threadIdx.x;
        const uint32_t row_a = 2u * blockIdx.x;
        const uint32_t row_b = row_a + 1u;
        const bool has_b = row_b < a.n_rows;
        float* frame = a.n_tab != 0u ? a.frame_tab[blockIdx.y] : a.p + static_cast<size_t>(blockIdx.y) * a.frame_stride;
        _Float16* half_frame = F16OUT ? a.half_out + static_cast<size_t>(blockIdx.y) * a.half_frame_stride : nullptr;
        float* pa = frame + static_cast<size_t>(row_a) * a.pitch_f;
        float* pb = frame + static_cast<size_t>(row_b) * a.pitch_f;
        float vv_a = 0.f, vv_b = 0.f;
        if(WEIGHT)
        {
            vv_a = row_term(a.w, a.row_first + row_a);
            vv_b = row_term(a.w, a.row_first + row_b);
        }

        if(NPASS >= 3)
            for(uint32_t i = tid; i < LTAB; i += T)
                ltab[i] = a.tab_mid[0][i]; // visible after the first pass's barrier
        // this thread's 16 filter values, requested now: after a backprojection has streamed the volume through the caches they
        // come from HBM, and the fused pass would otherwise wait for them
        float4 kreg[4];
        {
            const float4* kp4 = reinterpret_cast<const float4*>(a.kp + tid * 16u);
#pragma unroll
            for(uint32_t i4 = 0; i4 < 4u; ++i4)
                kreg[i4] = kp4[i4];
        }

        // ---- forward, first pass: radix EF on {o + j * NO}; 16 / EF groups per thread; global -> registers -> LDS
#pragma unroll
        for(uint32_t g = 0; g < 16u / EF; ++g)
        {
            const uint32_t o = tid + g * T; // < NO
            float2 v[EF];
            // this group's inter-pass twiddles, requested before the pixels so that they are there when the butterflies are done
            float2 tw[EF > 1u ? EF - 1u : 1u];
#pragma unroll
            for(uint32_t q = 1; q < EF; ++q)
                tw[q - 1u] = a.tab_first[(q - 1u) * NO + o];
            // all loads of the group first, without branches (a pixel beyond the row reads column 0 and is discarded), so that
            // they are in flight together; the weights are computed while they travel
            constexpr uint32_t JN = HALF ? (EF / 2u > 0u ? EF / 2u : 1u) : EF;
            float xs[JN], ys[JN];
            const float* pb_safe = has_b ? pb : pa;
#pragma unroll
            for(uint32_t j = 0; j < JN; ++j)
            {
                const uint32_t idx = o + j * NO;
                const uint32_t at = idx < a.dim_x ? idx : 0u;
#ifdef PARIS_FILTER_TIMING_NO_MEMORY // wrong results: prices the arithmetic alone (tools/README.md)
                xs[j] = static_cast<float>(at) * 1e-3f;
                ys[j] = static_cast<float>(at + blockIdx.x) * 1e-3f;
                (void)pb_safe;
#else
                xs[j] = pa[at];
                ys[j] = pb_safe[at];
#endif
            }
#pragma unroll
            for(uint32_t j = 0; j < EF; ++j)
            {
                float x = 0.f, y = 0.f; // expand (src/openmp/filtering.cpp:75-90): zero padding beyond the detector row
                if(j < JN)
                {
                    const uint32_t idx = o + j * NO;
                    x = xs[j < JN ? j : 0u];
                    y = ys[j < JN ? j : 0u];
                    if(WEIGHT)
                    {
                        const float ch = column_term(a.w, idx);
                        x = weighted(a.w, x, ch, vv_a);
                        y = weighted(a.w, y, ch, vv_b);
                    }
                    const bool inside = idx < a.dim_x;
                    x = inside ? x : 0.f;
                    y = (inside && has_b) ? y : 0.f;
                }
                v[j] = make_float2(x, y);
            }
#ifdef PARIS_FILTER_TIMING_NO_COMPUTE // wrong results: prices the loads (with the weighting) and the stores alone
#pragma unroll
            for(uint32_t j = 0; j < JN; ++j)
            {
                const uint32_t idx = o + j * NO;
                if(idx < a.dim_x)
                {
                    pa[idx] = v[j].x;
                    if(has_b)
                        pb[idx] = v[j].y;
                }
            }
            continue;
#endif
            dft_forward<RF, HALF && (RF > 0)>(v);
#pragma unroll
            for(uint32_t i = 0; i < EF; ++i)
            {
                const uint32_t q = static_cast<uint32_t>(brev(static_cast<int>(i), RF));
                float2 x = v[i];
                if(q != 0u)
                    x = cmul_fast(x, tw[q - 1u]);
                fx[lds_pad(q * NO + o)] = x;
            }
        }
#ifdef PARIS_FILTER_TIMING_NO_COMPUTE
        return;
#endif
        __syncthreads();

        // ---- forward middle passes: radix 16 on blocks of B = 16^(NPASS - pass)
#pragma unroll
        for(int pass = 1; pass < NPASS - 1; ++pass)
        {
            const uint32_t B = NO >> (4 * (pass - 1));
            const uint32_t sub = B / 16u;
            const uint32_t blk = tid / sub, o = tid % sub;
            const float2* tab = a.tab_mid[NPASS - 2 - pass]; // B = 256 -> 0, 4096 -> 1, 65536 -> 2
            const bool in_lds = pass == NPASS - 2;           // the block-256 pass
            float2 v[16];
#pragma unroll
            for(uint32_t j = 0; j < 16u; ++j)
                v[j] = fx[lds_pad(blk * B + o + j * sub)];
            dft_forward<4, false>(v);
#pragma unroll
            for(uint32_t i = 0; i < 16u; ++i)
            {
                const uint32_t q = static_cast<uint32_t>(brev(static_cast<int>(i), 4));
                float2 x = v[i];
                if(q != 0u)
                    x = cmul_fast(x, in_lds ? ltab[(q - 1u) * 16u + o] : tab[(q - 1u) * sub + o]);
                fx[lds_pad(blk * B + q * sub + o)] = x; // the thread's own 16 slots: no barrier between its reads and writes
            }
            __syncthreads();
        }

        // ---- fused pass on 16 contiguous values: last forward radix-16, multiply by K, first inverse radix-16
        {
            float2 v[16];
#pragma unroll
            for(uint32_t j = 0; j < 16u; ++j)
                v[j] = fx[lds_pad(tid * 16u + j)];
            dft_forward<4, false>(v);
#pragma unroll
            for(uint32_t i4 = 0; i4 < 4u; ++i4)
            {
                const float4 k = kreg[i4]; // do_filtering (src/openmp/filtering.cpp:92-105): both components times the real K
                v[4u * i4 + 0u].x *= k.x; v[4u * i4 + 0u].y *= k.x;
                v[4u * i4 + 1u].x *= k.y; v[4u * i4 + 1u].y *= k.y;
                v[4u * i4 + 2u].x *= k.z; v[4u * i4 + 2u].y *= k.z;
                v[4u * i4 + 3u].x *= k.w; v[4u * i4 + 3u].y *= k.w;
            }
            dft_inverse<4, false>(v);
#pragma unroll
            for(uint32_t j = 0; j < 16u; ++j)
                fx[lds_pad(tid * 16u + j)] = v[j];
        }
        __syncthreads();

        // ---- inverse middle passes, smallest block first
#pragma unroll
        for(int pass = NPASS - 2; pass >= 1; --pass)
        {
            const uint32_t B = NO >> (4 * (pass - 1));
            const uint32_t sub = B / 16u;
            const uint32_t blk = tid / sub, o = tid % sub;
            const float2* tab = a.tab_mid[NPASS - 2 - pass];
            const bool in_lds = pass == NPASS - 2;
            float2 v[16];
#pragma unroll
            for(uint32_t i = 0; i < 16u; ++i)
            {
                const uint32_t q = static_cast<uint32_t>(brev(static_cast<int>(i), 4));
                float2 x = fx[lds_pad(blk * B + q * sub + o)];
                if(q != 0u)
                    x = cmul_conj_fast(x, in_lds ? ltab[(q - 1u) * 16u + o] : tab[(q - 1u) * sub + o]);
                v[i] = x;
            }
            dft_inverse<4, false>(v);
#pragma unroll
            for(uint32_t j = 0; j < 16u; ++j)
                fx[lds_pad(blk * B + o + j * sub)] = v[j];
            if(pass > 1)
                __syncthreads();
        }
        // the last pass's twiddles (first group) are requested before the barrier that precedes it: their L2 latency overlaps
        // the wait for the other waves
        float2 tw_last[EF > 1u ? EF - 1u : 1u];
#pragma unroll
        for(uint32_t q = 1; q < EF; ++q)
            tw_last[q - 1u] = a.tab_first[(q - 1u) * NO + tid];
        if(NPASS >= 3)
            __syncthreads();

        // ---- inverse last pass: radix EF, shrink (:107-118) + normalize (:120-131; N is a power of two: exact) + store
        constexpr float inv_n = 1.f / static_cast<float>(N);
#pragma unroll
        for(uint32_t g = 0; g < 16u / EF; ++g)
        {
            const uint32_t o = tid + g * T;
            float2 v[EF];
#pragma unroll
            for(uint32_t i = 0; i < EF; ++i)
            {
                const uint32_t q = static_cast<uint32_t>(brev(static_cast<int>(i), RF));
                float2 x = fx[lds_pad(q * NO + o)];
                if(q != 0u)
                    x = cmul_conj_fast(x, g == 0u ? tw_last[q - 1u] : a.tab_first[(q - 1u) * NO + o]);
                v[i] = x;
            }
            dft_inverse<RF, HALF && (RF > 0)>(v);
#pragma unroll
            for(uint32_t j = 0; j < EF; ++j)
            {
                if(HALF && j >= EF / 2u)
                    continue;
                const uint32_t idx = o + j * NO;
                if(idx < a.dim_x)
                {
                    if(F16OUT)
                    {
                        half_frame[static_cast<size_t>(row_a) * a.half_pitch + idx] = static_cast<_Float16>(v[j].x * inv_n);
                        if(has_b)
                            half_frame[static_cast<size_t>(row_b) * a.half_pitch + idx] = static_cast<_Float16>(v[j].y * inv_n);
                    }
                    else
                    {
#ifdef PARIS_FILTER_TIMING_NO_MEMORY
                        if(v[j].x == 123456.789f) // never: keeps the arithmetic alive without the stores
#endif
                        pa[idx] = v[j].x * inv_n;
                        if(has_b)
#ifdef PARIS_FILTER_TIMING_NO_MEMORY
                            if(v[j].y == 123456.789f)
#endif
                            pb[idx] = v[j].y * inv_n;
                    }
                }
            }
        }
    }

    // kp[p] = K[fold(f(p))]: the frequency the fused pass holds in register i of thread tid (p = tid * 16 + i)
    __global__ void __launch_bounds__(256) permute_k_kernel(const float* __restrict__ k, float* __restrict__ kp, uint32_t log2n)
    {
        const uint32_t n = 1u << log2n;
        const uint32_t p = blockIdx.x * 256u + threadIdx.x;
        if(p >= n)
            return;
        const uint32_t npass = (log2n + 3u) / 4u;
        const uint32_t rf = log2n - 4u * (npass - 1u);
        // digits of the position, most significant first: the first pass's output index q1 (radix 2^rf), then one radix-16
        // digit per middle pass, then the register index i whose content is X_(brev4 i)
        uint32_t f = 0, stride = 1u, rest = p, block = n;
        {
            const uint32_t sub = block >> rf;
            f += (rest / sub) * stride;
            stride <<= rf;
            rest %= sub;
            block = sub;
        }
        while(block > 16u)
        {
            const uint32_t sub = block / 16u;
            f += (rest / sub) * stride;
            stride *= 16u;
            rest %= sub;
            block = sub;
        }
        f += (__brev(rest) >> 28) * stride;
        kp[p] = k[f <= n / 2u ? f : n - f]; // K is real and even
    }

    template <int LOG2N, bool HALF, bool WEIGHT, bool F16OUT>
    int launch(paris_hip_ctx* ctx, const FusedFilterArgs& a, uint32_t n_frames)
    {
        constexpr uint32_t N = 1u << LOG2N;
        constexpr uint32_t lds_bytes = (N + N / 16u + 15u * 16u) * sizeof(float2); // data + the block-256 twiddles
        if(lds_bytes > 64u * 1024u)
            PARIS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(filter_rows_kernel<LOG2N, HALF, WEIGHT, F16OUT>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        hipLaunchKernelGGL((filter_rows_kernel<LOG2N, HALF, WEIGHT, F16OUT>), dim3((a.n_rows + 1u) / 2u, n_frames), dim3(N / 16u), lds_bytes,
                           ctx->stream, a);
        return PARIS_HIP_SUCCESS;
    }

    template <int LOG2N>
    int launch_flags(paris_hip_ctx* ctx, const FusedFilterArgs& a, bool half_ok, bool weight, bool f16out, uint32_t n)
    {
        if(f16out)
        {
            if(half_ok)
                return weight ? launch<LOG2N, true, true, true>(ctx, a, n) : launch<LOG2N, true, false, true>(ctx, a, n);
            return weight ? launch<LOG2N, false, true, true>(ctx, a, n) : launch<LOG2N, false, false, true>(ctx, a, n);
        }
        if(half_ok)
            return weight ? launch<LOG2N, true, true, false>(ctx, a, n) : launch<LOG2N, true, false, false>(ctx, a, n);
        return weight ? launch<LOG2N, false, true, false>(ctx, a, n) : launch<LOG2N, false, false, false>(ctx, a, n);
    }

    uint32_t ilog2(uint32_t v)
    {
        uint32_t l = 0;
        while((1u << l) < v)
            ++l;
        return l;
    }

    // W_B^(o q) tables of one pass, q = 1 .. R - 1, o < B / R, rounded from double
    int make_table(uint32_t B, uint32_t R, float2** out)
    {
        const uint32_t sub = B / R;
        std::vector<float2> h(static_cast<size_t>(R - 1u) * sub);
        for(uint32_t q = 1; q < R; ++q)
            for(uint32_t o = 0; o < sub; ++o)
            {
                const double ang = -2.0 * M_PI * static_cast<double>((static_cast<uint64_t>(o) * q) % B) / static_cast<double>(B);
                h[static_cast<size_t>(q - 1u) * sub + o] = make_float2(static_cast<float>(std::cos(ang)), static_cast<float>(std::sin(ang)));
            }
        PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(out), h.size() * sizeof(float2)));
        const hipError_t err = hipMemcpy(*out, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice); // one-off, small
        if(err != hipSuccess)
        {
            (void)hipFree(*out);
            *out = nullptr;
            return static_cast<int>(err);
        }
        return PARIS_HIP_SUCCESS;
    }
}

int paris_hip_fused_filter_tables(paris_hip_ctx* ctx, uint32_t n, paris_hip_fft_plan* plan)
{
    (void)ctx;
    if(plan->d_tab_first != nullptr || n < 1024u)
        return PARIS_HIP_SUCCESS;
    const uint32_t log2n = ilog2(n);
    const uint32_t npass = (log2n + 3u) / 4u;
    const uint32_t ef = 1u << (log2n - 4u * (npass - 1u));
    if(ef > 1u)
    {
        if(int rc = make_table(n, ef, &plan->d_tab_first))
            return rc;
    }
    else // radix 1 cannot happen for n >= 1024 (log2n = 4 m + 1 gives ef = 2); kept for completeness
        PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&plan->d_tab_first), sizeof(float2)));
    uint32_t B = 256u;
    for(uint32_t k = 0; k + 2u < npass && k < 3u; ++k, B *= 16u)
        if(int rc = make_table(B, 16u, &plan->d_tab_mid[k]))
            return rc;
    return PARIS_HIP_SUCCESS;
}

int paris_hip_fused_filter_permute_k(paris_hip_ctx* ctx, const float* d_k, uint32_t n, float** d_kp)
{
    PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(d_kp), static_cast<size_t>(n) * sizeof(float)));
    hipLaunchKernelGGL(permute_k_kernel, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, d_k, *d_kp, ilog2(n));
    const hipError_t err = hipGetLastError();
    if(err != hipSuccess)
    {
        (void)hipFree(*d_kp);
        *d_kp = nullptr;
        return static_cast<int>(err);
    }
    return PARIS_HIP_SUCCESS;
}

int paris_hip_fused_filter_launch(paris_hip_ctx* ctx, float* d_rows, uint32_t pitch_f, uint32_t dim_x, uint32_t n_rows, uint32_t row_first,
                                  bool weight, float h_min, float v_min, float d_sd, float l_px_row, float l_px_col, const float* d_kp,
                                  const paris_hip_fft_plan* plan, uint32_t filter_size, uint16_t* d_half, uint32_t half_pitch, uint32_t n_frames,
                                  size_t frame_stride_f, size_t half_frame_stride, float* const* frame_rows)
{
    if(n_frames == 0u || n_frames > 65535u)
        return n_frames == 0u ? PARIS_HIP_SUCCESS : PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(frame_rows != nullptr && (n_frames > FRAME_TAB_MAX || d_half != nullptr))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    FusedFilterArgs a{};
    a.frame_stride = frame_stride_f;
    a.half_frame_stride = half_frame_stride;
    a.p = d_rows;
    if(frame_rows != nullptr)
    {
        a.n_tab = n_frames;
        for(uint32_t f = 0; f < FRAME_TAB_MAX; ++f)
            a.frame_tab[f] = frame_rows[f < n_frames ? f : 0u];
        a.p = frame_rows[0];
    }
    a.pitch_f = pitch_f;
    a.dim_x = dim_x;
    a.n_rows = n_rows;
    a.row_first = row_first;
    a.w = WeightParams{h_min, v_min, d_sd, l_px_row, l_px_col, 0u};
    if(weight)
    {
        // the largest radicand d_sd^2 + h^2 + v^2 over the detector, in double: the lean sqrt / divide need it (and d_sd^2, the
        // smallest) well inside the fp32 range
        const double h0 = 0.5 * l_px_row + h_min, h1 = 0.5 * l_px_row + (static_cast<double>(dim_x) - 1.0) * l_px_row + h_min;
        const double rows_all = static_cast<double>(row_first) + n_rows;
        const double v0 = 0.5 * l_px_col + v_min, v1 = 0.5 * l_px_col + (rows_all - 1.0) * l_px_col + v_min;
        const double dd = static_cast<double>(d_sd) * d_sd;
        const double q_max = dd + std::max(h0 * h0, h1 * h1) + std::max(v0 * v0, v1 * v1);
        const double lo = std::ldexp(1.0, -60), hi = std::ldexp(1.0, 60);
        a.w.lean = 0u;
        if(dd >= lo && q_max <= hi && std::isfinite(q_max) && d_sd > 0.f)
        {
            // ... and the device has compared the lean forms with the compiler's sqrtf and `/` for every fp32 radicand of the range
            // (cached per range: once per detector / row band)
            bool exact = false;
            if(int rc = paris_hip_lean_weighting_check(ctx, d_sd, dd, q_max, &exact))
                return rc;
            a.w.lean = exact ? 1u : 0u;
        }
    }
    a.kp = d_kp;
    a.tab_first = plan->d_tab_first;
    for(int k = 0; k < 3; ++k)
        a.tab_mid[k] = plan->d_tab_mid[k];
    a.half_out = reinterpret_cast<_Float16*>(d_half);
    a.half_pitch = half_pitch;
    const bool half_ok = dim_x <= filter_size / 2u;
    const bool f16out = d_half != nullptr;
    switch(ilog2(filter_size))
    {
        case 10: return launch_flags<10>(ctx, a, half_ok, weight, f16out, n_frames);
        case 11: return launch_flags<11>(ctx, a, half_ok, weight, f16out, n_frames);
        case 12: return launch_flags<12>(ctx, a, half_ok, weight, f16out, n_frames);
        case 13: return launch_flags<13>(ctx, a, half_ok, weight, f16out, n_frames);
        case 14: return launch_flags<14>(ctx, a, half_ok, weight, f16out, n_frames);
        default: return PARIS_HIP_ERROR_UNSUPPORTED;
    }
}

// PARIS_HIP_CTX_WARM: a query of one kernel of this translation unit makes the runtime load its code object now
void paris_hip_warm_filter_fused()
{
    hipFuncAttributes a{};
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&permute_k_kernel));
}
