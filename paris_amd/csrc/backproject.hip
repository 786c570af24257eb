// Voxel-driven cone-beam backprojection for gfx950 (MI355X).
//
// Replaces paris::openmp::backproject (src/openmp/backprojection.cpp:86-199) and paris::cuda::backproject
// (src/cuda/backprojection.cu:64-243) behind paris_hip_backproject (include/paris_hip.h).
//
// Numerics: every fp32 operation is the reference's, in the reference's order, rounded once (IEEE divide,
// no FMA contraction), so the volume is bit-identical to the OpenMP backend's.
//
// Mapping (DESIGN.md "Backprojection kernel"):
//   - a 256-thread workgroup owns a tile of 64 (x) x TY (y) voxel columns and walks TZ slices in z;
//   - lanes of a wave cover x contiguously (VX voxels per lane, 16 B / 8 B / 4 B accesses), so every
//     volume load/store instruction touches whole 256-byte runs of the x-fastest volume;
//   - the detector footprint of the tile (a bounding box computed from the tile's corner rays) is staged
//     into LDS once per tile; the four bilinear taps are LDS reads. A tap that falls outside the staged
//     box (never for sane geometries; possible when the box is larger than the LDS budget) is fetched
//     from global memory instead, so the result does not depend on the box being right;
//   - per (x,y) column s, t, factor, h, u and the x-interpolation weights are z-invariant and kept in
//     registers; per z step only v is recomputed (src/openmp/backprojection.cpp:130-133).
#include "paris_hip_internal.h"

#include <cmath>
#include <cstring>

namespace
{
    struct BpParams
    {
        const float* proj;
        float* vol;
        uint32_t p_pitch; // floats per detector row
        uint32_t p_dim_x, p_dim_y;
        uint32_t v_dim_x, v_dim_y, v_dim_z;
        uint32_t k_off, l_off, m_off; // roi.x1, roi.y1, roi.z1 + v_offset
        float x_base, y_base, z_base; // -(dim_full * l_vx/2) + l_vx/2
        float l_vx_x, l_vx_y, l_vx_z;
        float sin_phi, cos_phi;
        float d_so, d_sd;
        float min_h, min_v; // -(p_dim * l_px/2) - delta
        float l_px_x, l_px_y;
        float rcp_l_px_y; // RN(1 / l_px_y), used only by the validated fast division
        float p_dim_x_f, p_dim_y_f;
        uint32_t lds_floats;
        uint32_t tz; // slices per tile
        uint32_t ntx, nty, ntz; // tiles per axis
        uint32_t order;         // workgroup -> tile mapping, see tile_of_block
    };

    struct ColConst
    {
        float factor, h, u;
    };

    // src/openmp/backprojection.cpp:116-129,139 for one (x,y) column; K, L are global voxel indices
    __device__ __forceinline__ ColConst column_constants(const BpParams& g, uint32_t K, uint32_t L)
    {
        const float x_k = g.x_base + static_cast<float>(K) * g.l_vx_x; // :39-43
        const float y_l = g.y_base + static_cast<float>(L) * g.l_vx_y;
        const float s = x_k * g.cos_phi + y_l * g.sin_phi;  // :121
        const float t = -x_k * g.sin_phi + y_l * g.cos_phi; // :122
        const float den = s + g.d_so;
        ColConst c;
        c.factor = g.d_sd / den;                                      // :125
        c.h = ((t * c.factor) - g.min_h) / g.l_px_x - (1.f / 2.f);   // :45-50
        c.u = -(g.d_so / den);                                        // :139
        return c;
    }

    // x / c for a divisor c that is constant over the launch, with r = RN(1 / c): one multiply and two FMAs
    // (Markstein's correction step) instead of the ~10-instruction IEEE sequence. Only used after
    // fastdiv_validate_kernel has checked, for THIS c and EVERY fp32 x, that the result has the bits of x / c.
    __device__ __forceinline__ float div_by_constant(float x, float c, float r)
    {
        const float q = x * r;
        const float e = __builtin_fmaf(-q, c, x); // exact remainder
        return __builtin_fmaf(e, r, q);
    }

    // v detector coordinate of slice z_m for a column with magnification `factor` (:130-133, :45-50)
    template <bool FD>
    __device__ __forceinline__ float v_coordinate(const BpParams& g, float z_m, float factor)
    {
        const float b = (z_m * factor) - g.min_v;
        const float q = FD ? div_by_constant(b, g.l_px_y, g.rcp_l_px_y) : b / g.l_px_y;
        return q - (1.f / 2.f);
    }

    // Exhaustive check of div_by_constant for one divisor: all 2^32 bit patterns of x. The quotient is consumed
    // only as v = q - 0.5f (v_coordinate), so a pattern passes when v has the same bits either way, or both are NaN,
    // or both lie beyond +-2^24 / are non-finite (no detector has 2^24 rows: such a coordinate fails the validity
    // test either way and the contribution is the same exact 0). In practice the two forms differ only where the
    // remainder underflows (|x| < 2^-103: v = -0.5 both ways) or the product overflows (|x| > 2^124).
    __global__ void __launch_bounds__(256) fastdiv_validate_kernel(float c, float r, unsigned long long* mismatches)
    {
        const uint32_t t = blockIdx.x * 256u + threadIdx.x; // 2^24 threads x 256 patterns
        unsigned int bad = 0;
        for(uint32_t i = 0; i < 256u; ++i)
        {
            const uint32_t bits = (i << 24) | t;
            const float x = __uint_as_float(bits);
            const float want = x / c - (1.f / 2.f);
            const float got = div_by_constant(x, c, r) - (1.f / 2.f);
            const bool same = __float_as_uint(want) == __float_as_uint(got);
            const bool both_nan = (want != want) && (got != got);
            const bool want_far = !(fabsf(want) <= 16777216.f); // beyond 2^24, inf or NaN
            const bool got_far = !(fabsf(got) <= 16777216.f);
            if(!(same || both_nan || (want_far && got_far)))
                ++bad;
        }
        if(bad)
            atomicAdd(mismatches, static_cast<unsigned long long>(bad));
    }

    __device__ __forceinline__ int to_int_clamped(float x)
    {
        x = fminf(fmaxf(x, -1.0e9f), 1.0e9f); // NaN -> -1e9
        return static_cast<int>(floorf(x));
    }

    template <int VX> struct vec_of;
    template <> struct vec_of<1> { using type = float; };
    template <> struct vec_of<2> { using type = float2; };
    template <> struct vec_of<4> { using type = float4; };

    template <int VX> __device__ __forceinline__ float& elem(typename vec_of<VX>::type& v, int j);
    template <> __device__ __forceinline__ float& elem<1>(float& v, int) { return v; }
    template <> __device__ __forceinline__ float& elem<2>(float2& v, int j) { return j == 0 ? v.x : v.y; }
    template <> __device__ __forceinline__ float& elem<4>(float4& v, int j)
    {
        return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
    }

    // ext-vector twins of float/float2/float4 for the nontemporal builtins
    template <int VX> struct ext_of;
    template <> struct ext_of<1> { typedef float type; };
    template <> struct ext_of<2> { typedef float type __attribute__((ext_vector_type(2))); };
    template <> struct ext_of<4> { typedef float type __attribute__((ext_vector_type(4))); };

    // Volume voxels are touched exactly once per launch: with NT the loads/stores carry the nontemporal hint so
    // the stream does not displace the projection from L2 / Infinity Cache (measured +5..10 % on z-walks).
    template <int VX, bool NT> __device__ __forceinline__ typename vec_of<VX>::type load_voxels(const float* p)
    {
        using ext_t = typename ext_of<VX>::type;
        using vec_t = typename vec_of<VX>::type;
        ext_t e = NT ? __builtin_nontemporal_load(reinterpret_cast<const ext_t*>(p)) : *reinterpret_cast<const ext_t*>(p);
        return *reinterpret_cast<vec_t*>(&e);
    }
    template <int VX, bool NT> __device__ __forceinline__ void store_voxels(float* p, typename vec_of<VX>::type v)
    {
        using ext_t = typename ext_of<VX>::type;
        const ext_t e = *reinterpret_cast<ext_t*>(&v);
        if(NT)
            __builtin_nontemporal_store(e, reinterpret_cast<ext_t*>(p));
        else
            *reinterpret_cast<ext_t*>(p) = e;
    }

    // Workgroup -> tile mapping. Blocks are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8), and
    // which tiles run concurrently decides the DRAM locality of the volume stream (tools/membench5.hip):
    //   0: x tiles fastest, then y, then z (XCD k keeps hitting the same x columns: slowest)
    //   1: z tiles fastest, then x, then y
    //   5: XCD k sweeps its own contiguous eighth of the (x, y, z) tile sequence
    __device__ __forceinline__ bool tile_of_block(const BpParams& g, uint32_t b, uint32_t& bx, uint32_t& by, uint32_t& bz)
    {
        const uint32_t total = g.ntx * g.nty * g.ntz;
        if(g.order == 1u)
        {
            if(b >= total)
                return false;
            bz = b % g.ntz;
            b /= g.ntz;
            bx = b % g.ntx;
            by = b / g.ntx;
            return true;
        }
        if(g.order == 5u)
        {
            const uint32_t per = (total + 7u) / 8u;
            b = (b % 8u) * per + b / 8u;
        }
        if(b >= total)
            return false;
        bx = b % g.ntx;
        b /= g.ntx;
        by = b % g.nty;
        bz = b / g.nty;
        return true;
    }

    // --------------------------------------------------------------------------------------------
    // Tile kernel. 1-D grid over ntx * nty * ntz tiles of 64 x TY x g.tz voxels, TY = 4*VX.
    // --------------------------------------------------------------------------------------------
    template <int VX, int UNROLL, bool NT, bool FD>
    __global__ void __launch_bounds__(256) bp_tile_kernel(const BpParams g)
    {
        extern __shared__ __attribute__((aligned(16))) float lds[];

        constexpr uint32_t XL = 64u / VX; // lanes along x per wave
        constexpr uint32_t RW = VX;       // volume rows per wave
        constexpr uint32_t TY = 4u * RW;

        const uint32_t tid = threadIdx.x;
        const uint32_t lane = tid & 63u;
        const uint32_t wave = tid >> 6;

        uint32_t bx, by, bz;
        if(!tile_of_block(g, blockIdx.x, bx, by, bz)) // uniform: whole workgroup leaves before the barrier
            return;
        const uint32_t k0 = bx * 64u;
        const uint32_t l0 = by * TY;
        const uint32_t m0 = bz * g.tz;
        const uint32_t k1 = min(k0 + 63u, g.v_dim_x - 1u);
        const uint32_t l1 = min(l0 + TY - 1u, g.v_dim_y - 1u);
        const uint32_t m1 = min(m0 + g.tz - 1u, g.v_dim_z - 1u);

        // ---- detector bounding box of the tile (uniform) ------------------------------------
        // h is a projective function of (x,y) and v of (z, factor): extremes sit on tile corners.
        float hmin = INFINITY, hmax = -INFINITY, fmin = INFINITY, fmax = -INFINITY;
#pragma unroll
        for(int ci = 0; ci < 4; ++ci)
        {
            const ColConst c = column_constants(g, g.k_off + ((ci & 1) ? k1 : k0), g.l_off + ((ci & 2) ? l1 : l0));
            hmin = fminf(hmin, c.h);
            hmax = fmaxf(hmax, c.h);
            fmin = fminf(fmin, c.factor);
            fmax = fmaxf(fmax, c.factor);
        }
        const float z_lo = g.z_base + static_cast<float>(g.m_off + m0) * g.l_vx_z;
        const float z_hi = g.z_base + static_cast<float>(g.m_off + m1) * g.l_vx_z;
        const float v00 = v_coordinate<false>(g, z_lo, fmin), v01 = v_coordinate<false>(g, z_lo, fmax);
        const float v10 = v_coordinate<false>(g, z_hi, fmin), v11 = v_coordinate<false>(g, z_hi, fmax);
        const float vmin = fminf(fminf(v00, v01), fminf(v10, v11));
        const float vmax = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));

        const int bx0 = max(to_int_clamped(hmin) - 1, 0);
        const int bx1 = min(to_int_clamped(hmax) + 2, static_cast<int>(g.p_dim_x) - 1);
        const int by0 = max(to_int_clamped(vmin) - 1, 0);
        const int by1 = min(to_int_clamped(vmax) + 2, static_cast<int>(g.p_dim_y) - 1);
        int bw = bx1 - bx0 + 1;
        int bh = by1 - by0 + 1;
        if(bw < 2 || bh < 2)
        {
            bw = 0;
            bh = 0;
        }
        int stride = bw | 1;
        int bhs = min(bh, static_cast<int>(g.lds_floats) / stride); // rows that fit the LDS budget
        if(bhs < 2)
        {
            // nothing useful fits: stage nothing, every valid tap takes the global path
            bw = 0;
            bhs = 0;
            stride = 1;
        }
        const int bw_m2 = bw - 2;
        const int bhs_m2 = bhs - 2;

        // ---- stage the box: one wave per detector row, lanes along the row ---------------------
        for(int r = static_cast<int>(wave); r < bhs; r += 4)
        {
            const float* src = g.proj + static_cast<size_t>(by0 + r) * g.p_pitch + bx0;
            float* dst = lds + r * stride;
            for(int c = static_cast<int>(lane); c < bw; c += 64)
                dst[c] = src[c];
        }
        __syncthreads();

        // ---- per-lane columns --------------------------------------------------------------------
        const uint32_t xq = lane % XL;
        const uint32_t yy = lane / XL;
        const uint32_t k = k0 + xq * VX;
        const uint32_t l = l0 + wave * RW + yy;
        if(k >= g.v_dim_x || l >= g.v_dim_y)
            return;

        float factor[VX], u[VX], wx1[VX], wx2[VX], ymax[VX];
        int xoff[VX], x1i[VX];
        bool colin[VX];
#pragma unroll
        for(int j = 0; j < VX; ++j)
        {
            const ColConst c = column_constants(g, g.k_off + k + j, g.l_off + l);
            const float x1 = floorf(c.h); // :55-58
            const float x2 = x1 + 1.f;
            const bool x_valid = (x1 >= 0.f) && (x2 < g.p_dim_x_f); // :65-66
            factor[j] = c.factor;
            u[j] = c.u;
            // :77-78 divide by (x2 - x1), which is exactly 1.f whenever x_valid (|x1| < 2^24): the divisions
            // are the identity and are dropped; for an invalid column the weights are never used
            wx2[j] = x2 - c.h;
            wx1[j] = c.h - x1;
            ymax[j] = x_valid ? g.p_dim_y_f : -INFINITY; // folds the x validity into the y2 test
            x1i[j] = static_cast<int>(x1);
            const int rel = x1i[j] - bx0;
            colin[j] = x_valid && rel >= 0 && rel <= bw_m2;
            xoff[j] = min(max(rel, 0), max(bw_m2, 0));
        }

        using vec_t = typename vec_of<VX>::type;
        const size_t slice = static_cast<size_t>(g.v_dim_x) * g.v_dim_y;
        float* vp = g.vol + (static_cast<size_t>(m0) * g.v_dim_y + l) * g.v_dim_x + k;
        const uint32_t mcount = m1 - m0 + 1u;

        auto update = [&](vec_t& acc, uint32_t m_local) {
            const float z_m = g.z_base + static_cast<float>(g.m_off + m0 + m_local) * g.l_vx_z; // :118
#pragma unroll
            for(int j = 0; j < VX; ++j)
            {
                const float v = v_coordinate<FD>(g, z_m, factor[j]);
                const float y1 = floorf(v);
                const float y2 = y1 + 1.f;
                const bool valid = (y1 >= 0.f) && (y2 < ymax[j]); // :67-68 (+ x validity)
                const int y1i = static_cast<int>(y1);
                const int rrel = y1i - by0;
                const bool inbox = colin[j] && rrel >= 0 && rrel <= bhs_m2;
                const int rc = min(max(rrel, 0), max(bhs_m2, 0));
                const int base = rc * stride + xoff[j];
                float q11 = lds[base];
                float q21 = lds[base + 1];
                float q12 = lds[base + stride];
                float q22 = lds[base + stride + 1];
                if(valid && !inbox)
                {
                    // tap outside the staged box: read the detector directly (valid => in bounds). volatile keeps
                    // the compiler from merging these loads with the LDS reads into flat loads of a selected pointer
                    const volatile float* pr = g.proj + static_cast<size_t>(y1i) * g.p_pitch + x1i[j];
                    q11 = pr[0];
                    q21 = pr[1];
                    q12 = pr[g.p_pitch];
                    q22 = pr[g.p_pitch + 1];
                }
                const float interp_y1 = wx2[j] * q11 + wx1[j] * q21; // :77
                const float interp_y2 = wx2[j] * q12 + wx1[j] * q22; // :78
                // :80 divides by (y2 - y1) == 1.f exactly whenever valid -- dropped, as above
                float det = (y2 - v) * interp_y1 + (v - y1) * interp_y2;
                det = valid ? det : 0.f; // :71
                elem<VX>(acc, j) += 0.5f * det * u[j] * u[j];                                    // :140
            }
        };

        uint32_t mm = 0;
        for(; mm + UNROLL <= mcount; mm += UNROLL)
        {
            vec_t acc[UNROLL];
#pragma unroll
            for(int i = 0; i < UNROLL; ++i)
                acc[i] = load_voxels<VX, NT>(vp + (mm + i) * slice);
#pragma unroll
            for(int i = 0; i < UNROLL; ++i)
                update(acc[i], mm + i);
#pragma unroll
            for(int i = 0; i < UNROLL; ++i)
                store_voxels<VX, NT>(vp + (mm + i) * slice, acc[i]);
        }
        for(; mm < mcount; ++mm)
        {
            vec_t acc = load_voxels<VX, NT>(vp + mm * slice);
            update(acc, mm);
            store_voxels<VX, NT>(vp + mm * slice, acc);
        }
    }

    // --------------------------------------------------------------------------------------------
    // Cross-check kernel (variant 1): one thread per voxel, taps straight from global memory, the
    // reference's loop body verbatim in structure. Slow; used by tests to validate the tile kernel.
    // --------------------------------------------------------------------------------------------
    __global__ void __launch_bounds__(256) bp_gather_kernel(const BpParams g)
    {
        const uint32_t k = blockIdx.x * 256u + threadIdx.x;
        const uint32_t l = blockIdx.y;
        const uint32_t m = blockIdx.z;
        if(k >= g.v_dim_x)
            return;
        const ColConst c = column_constants(g, g.k_off + k, g.l_off + l);
        const float z_m = g.z_base + static_cast<float>(g.m_off + m) * g.l_vx_z;
        const float x = c.h;
        const float y = v_coordinate<false>(g, z_m, c.factor);
        const float x1 = floorf(x), x2 = x1 + 1.f, y1 = floorf(y), y2 = y1 + 1.f;
        float interp = 0.f;
        if(x1 >= 0.f && x2 < g.p_dim_x_f && y1 >= 0.f && y2 < g.p_dim_y_f)
        {
            const float* pr = g.proj + static_cast<size_t>(static_cast<uint32_t>(y1)) * g.p_pitch
                              + static_cast<uint32_t>(x1);
            const float q11 = pr[0], q21 = pr[1], q12 = pr[g.p_pitch], q22 = pr[g.p_pitch + 1];
            const float interp_y1 = (x2 - x) / (x2 - x1) * q11 + (x - x1) / (x2 - x1) * q21;
            const float interp_y2 = (x2 - x) / (x2 - x1) * q12 + (x - x1) / (x2 - x1) * q22;
            interp = (y2 - y) / (y2 - y1) * interp_y1 + (y - y1) / (y2 - y1) * interp_y2;
        }
        float* out = g.vol + (static_cast<size_t>(m) * g.v_dim_y + l) * g.v_dim_x + k;
        *out += 0.5f * interp * c.u * c.u;
    }

    // host-side constants, computed in fp32 exactly as the reference computes them per voxel
    inline float centered_base(uint32_t dim, float size)
    {
        const float size2 = size / 2.f;
        return -(static_cast<float>(dim) * size2) + size2; // src/openmp/backprojection.cpp:41-42
    }

    inline float detector_min(uint32_t dim, float size, float offset)
    {
        const float size2 = size / 2.f;
        return -(static_cast<float>(dim) * size2) - offset; // src/openmp/backprojection.cpp:47-48
    }

    // defaults from tools/tune_bp.py on MI355X (2048^2 x 256 slab): 5.5 TB/s of volume traffic
    constexpr uint32_t TZ_DEFAULT = 16;
    constexpr uint32_t LDS_BYTES_DEFAULT = 24u * 1024u;
    constexpr uint32_t LDS_BYTES_MAX = 64u * 1024u;

    template <int VX, int UNROLL, bool NT, bool FD>
    void launch_tile(BpParams& g, hipStream_t stream)
    {
        constexpr uint32_t TY = 4u * VX;
        g.ntx = (g.v_dim_x + 63u) / 64u;
        g.nty = (g.v_dim_y + TY - 1u) / TY;
        g.ntz = (g.v_dim_z + g.tz - 1u) / g.tz;
        uint32_t blocks = g.ntx * g.nty * g.ntz;
        if(g.order == 5u)
            blocks = ((blocks + 7u) / 8u) * 8u;
        hipLaunchKernelGGL((bp_tile_kernel<VX, UNROLL, NT, FD>), dim3(blocks), dim3(256), g.lds_floats * sizeof(float), stream, g);
    }

    template <int VX, bool NT, bool FD>
    void launch_tile_unroll(BpParams& g, int unroll, hipStream_t stream)
    {
        switch(unroll)
        {
            case 1: launch_tile<VX, 1, NT, FD>(g, stream); break;
            case 2: launch_tile<VX, 2, NT, FD>(g, stream); break;
            default: launch_tile<VX, 4, NT, FD>(g, stream); break;
        }
    }

    template <int VX>
    void launch_tile_flags(BpParams& g, int unroll, bool nt, bool fd, hipStream_t stream)
    {
        if(nt && fd)
            launch_tile_unroll<VX, true, true>(g, unroll, stream);
        else if(nt)
            launch_tile_unroll<VX, true, false>(g, unroll, stream);
        else if(fd)
            launch_tile_unroll<VX, false, true>(g, unroll, stream);
        else
            launch_tile_unroll<VX, false, false>(g, unroll, stream);
    }

    // Is div_by_constant exact for this divisor? Checked once per ctx and divisor on the GPU (about 2 ms).
    int fastdiv_is_exact(paris_hip_ctx* ctx, float c, bool* ok)
    {
        *ok = false;
        if(!(c > 0.f) || !(c < INFINITY))
            return PARIS_HIP_SUCCESS;
        uint32_t key;
        static_assert(sizeof(key) == sizeof(c), "fp32");
        std::memcpy(&key, &c, sizeof(key));
        auto it = ctx->fastdiv_exact.find(key);
        if(it == ctx->fastdiv_exact.end())
        {
            unsigned long long* d_bad = nullptr;
            PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_bad), sizeof(*d_bad)));
            PARIS_HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(*d_bad), ctx->stream));
            hipLaunchKernelGGL(fastdiv_validate_kernel, dim3(1u << 16), dim3(256), 0, ctx->stream, c, 1.f / c, d_bad);
            unsigned long long bad = ~0ull;
            PARIS_HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
            PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
            PARIS_HIP_TRY(hipFree(d_bad));
            it = ctx->fastdiv_exact.emplace(key, bad == 0ull).first;
        }
        *ok = it->second;
        return PARIS_HIP_SUCCESS;
    }
}

extern "C" int paris_hip_backproject(paris_hip_ctx* ctx, const float* d_p, size_t p_pitch, uint32_t p_dim_x,
                                     uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y,
                                     uint32_t v_dim_z, uint32_t v_offset, const paris_detector_geometry* det_geo,
                                     const paris_volume_geometry* vol_geo, int enable_roi,
                                     const paris_region_of_interest* roi, float sin_phi, float cos_phi,
                                     float delta_s, float delta_t)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_p == nullptr || d_v == nullptr || det_geo == nullptr || vol_geo == nullptr || (enable_roi && roi == nullptr))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(p_dim_x == 0 || p_dim_y == 0 || p_pitch < static_cast<size_t>(p_dim_x) * sizeof(float) || p_pitch % sizeof(float) != 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(v_dim_x == 0 || v_dim_y == 0 || v_dim_z == 0)
        return paris_hip_finish(ctx);
    const uint32_t tz = ctx->bp_tz ? ctx->bp_tz : TZ_DEFAULT;
    {
        // the 1-D grid must hold every tile (narrowest tile: 64 x 4 x tz)
        const uint64_t tiles = static_cast<uint64_t>((v_dim_x + 63u) / 64u) * ((v_dim_y + 3u) / 4u) * ((v_dim_z + tz - 1u) / tz);
        if(tiles > 0x7fffff00ull)
            return PARIS_HIP_ERROR_UNSUPPORTED;
    }

    BpParams g{};
    g.proj = d_p;
    g.vol = d_v;
    g.p_pitch = static_cast<uint32_t>(p_pitch / sizeof(float));
    g.p_dim_x = p_dim_x;
    g.p_dim_y = p_dim_y;
    g.v_dim_x = v_dim_x;
    g.v_dim_y = v_dim_y;
    g.v_dim_z = v_dim_z;
    g.k_off = enable_roi ? roi->x1 : 0u; // src/openmp/backprojection.cpp:105-110
    g.l_off = enable_roi ? roi->y1 : 0u;
    g.m_off = (enable_roi ? roi->z1 : 0u) + v_offset; // :109,:113
    g.x_base = centered_base(vol_geo->dim_x, vol_geo->l_vx_x);
    g.y_base = centered_base(vol_geo->dim_y, vol_geo->l_vx_y);
    g.z_base = centered_base(vol_geo->dim_z, vol_geo->l_vx_z);
    g.l_vx_x = vol_geo->l_vx_x;
    g.l_vx_y = vol_geo->l_vx_y;
    g.l_vx_z = vol_geo->l_vx_z;
    g.sin_phi = sin_phi;
    g.cos_phi = cos_phi;
    g.d_so = det_geo->d_so;                                        // :176 (signed)
    g.d_sd = std::fabs(det_geo->d_so) + std::fabs(det_geo->d_od); // :177
    g.l_px_x = det_geo->l_px_row;                                  // :170-171
    g.l_px_y = det_geo->l_px_col;
    g.min_h = detector_min(p_dim_x, g.l_px_x, delta_s);
    g.min_v = detector_min(p_dim_y, g.l_px_y, delta_t);
    g.rcp_l_px_y = 1.f / g.l_px_y;
    bool fd = false;
    if(ctx->bp_fastdiv != 0)
        if(int rc = fastdiv_is_exact(ctx, g.l_px_y, &fd))
            return rc;
    g.p_dim_x_f = static_cast<float>(p_dim_x);
    g.p_dim_y_f = static_cast<float>(p_dim_y);
    g.lds_floats = (ctx->bp_lds_bytes ? ctx->bp_lds_bytes : LDS_BYTES_DEFAULT) / sizeof(float);
    g.tz = tz;
    g.order = ctx->bp_order >= 0 ? static_cast<uint32_t>(ctx->bp_order) : 5u;

    const size_t ev = static_cast<size_t>(ctx->bp_launches % ctx->bp_start.size());
    PARIS_HIP_TRY(hipEventRecord(ctx->bp_start[ev], ctx->stream));
    if(ctx->bp_variant == 1)
    {
        const dim3 grid((v_dim_x + 255u) / 256u, v_dim_y, v_dim_z);
        if(v_dim_y > 65535u || v_dim_z > 65535u)
            return PARIS_HIP_ERROR_UNSUPPORTED;
        hipLaunchKernelGGL(bp_gather_kernel, grid, dim3(256), 0, ctx->stream, g);
    }
    else
    {
        // widest per-lane access the volume's alignment allows, unless the tuning knob asks for less
        const uintptr_t addr = reinterpret_cast<uintptr_t>(d_v);
        int vx = 1;
        if(v_dim_x % 4u == 0 && addr % 16u == 0)
            vx = 4;
        else if(v_dim_x % 2u == 0 && addr % 8u == 0)
            vx = 2;
        if(ctx->bp_vx && ctx->bp_vx < vx)
            vx = ctx->bp_vx;
        const int unroll = ctx->bp_unroll ? ctx->bp_unroll : 2;
        const bool nt = ctx->bp_nt != 0;
        if(vx == 4)
            launch_tile_flags<4>(g, unroll, nt, fd, ctx->stream);
        else if(vx == 2)
            launch_tile_flags<2>(g, unroll, nt, fd, ctx->stream);
        else
            launch_tile_flags<1>(g, unroll, nt, fd, ctx->stream);
    }
    PARIS_HIP_TRY(hipEventRecord(ctx->bp_stop[ev], ctx->stream));
    ++ctx->bp_launches;
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_backproject_batch(paris_hip_ctx* ctx, const float* d_p, size_t p_pitch,
                                           size_t p_stride_bytes, uint32_t n_proj, uint32_t p_dim_x,
                                           uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y,
                                           uint32_t v_dim_z, uint32_t v_offset,
                                           const paris_detector_geometry* det_geo,
                                           const paris_volume_geometry* vol_geo, int enable_roi,
                                           const paris_region_of_interest* roi, const float* sin_phi,
                                           const float* cos_phi, float delta_s, float delta_t)
{
    // Round 1: the batched entry point is the sequence of single-projection launches it is defined to
    // equal (include/paris_hip.h). A fused multi-projection kernel replaces this loop later.
    if(sin_phi == nullptr || cos_phi == nullptr || p_stride_bytes % sizeof(float) != 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const unsigned saved = ctx ? ctx->flags : 0u;
    if(ctx)
        ctx->flags &= ~PARIS_HIP_CTX_SYNCHRONOUS;
    int rc = PARIS_HIP_SUCCESS;
    for(uint32_t i = 0; i < n_proj && rc == PARIS_HIP_SUCCESS; ++i)
    {
        const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(d_p) + i * p_stride_bytes);
        rc = paris_hip_backproject(ctx, p, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset,
                                   det_geo, vol_geo, enable_roi, roi, sin_phi[i], cos_phi[i], delta_s, delta_t);
    }
    if(ctx)
        ctx->flags = saved;
    if(rc != PARIS_HIP_SUCCESS)
        return rc;
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_last_backproject_ms(paris_hip_ctx* ctx, float* ms)
{
    if(ctx == nullptr || ms == nullptr || ctx->bp_launches == 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    const size_t ev = static_cast<size_t>((ctx->bp_launches - 1) % ctx->bp_start.size());
    PARIS_HIP_TRY(hipEventSynchronize(ctx->bp_stop[ev]));
    PARIS_HIP_TRY(hipEventElapsedTime(ms, ctx->bp_start[ev], ctx->bp_stop[ev]));
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_variant(paris_hip_ctx* ctx, int variant)
{
    if(ctx == nullptr || variant < 0 || variant > 1)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_variant = variant;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_order(paris_hip_ctx* ctx, int order, int nontemporal)
{
    if(ctx == nullptr || !(order == -1 || order == 0 || order == 1 || order == 5) || nontemporal < -1 || nontemporal > 1)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_order = order;
    ctx->bp_nt = nontemporal < 0 ? 1 : nontemporal;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_fast_division_is_exact(paris_hip_ctx* ctx, float divisor, int* exact)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(exact == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    bool ok = false;
    if(int rc = fastdiv_is_exact(ctx, divisor, &ok))
        return rc;
    *exact = ok ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_fast_division(paris_hip_ctx* ctx, int enable)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_fastdiv = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_tuning(paris_hip_ctx* ctx, int vx, int unroll, int tz, int lds_bytes)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(!(vx == 0 || vx == 1 || vx == 2 || vx == 4) || !(unroll == 0 || unroll == 1 || unroll == 2 || unroll == 4))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(tz < 0 || tz > 4096 || lds_bytes < 0 || lds_bytes > static_cast<int>(LDS_BYTES_MAX)
       || (lds_bytes != 0 && lds_bytes < 1024))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_vx = vx;
    ctx->bp_unroll = unroll;
    ctx->bp_tz = static_cast<uint32_t>(tz);
    ctx->bp_lds_bytes = static_cast<uint32_t>(lds_bytes);
    return PARIS_HIP_SUCCESS;
}
