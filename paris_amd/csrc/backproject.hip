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

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

namespace
{
    struct BpParams
    {
        const void* proj;     // fp32 pixels, or IEEE half pixels when proj_f16 != 0
        uint32_t proj_f16;
        float* vol;
        uint32_t p_pitch; // pixels per detector row
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
        float rcp_l_px_x;
        float p_dim_x_f, p_dim_y_f;
        uint32_t lds_floats;
        uint32_t tz; // slices per tile
        uint32_t ntx, nty, ntz; // tiles per axis
        uint32_t order;         // workgroup -> tile mapping, see tile_of_block
        uint32_t store_sc1;     // nontemporal stores also carry sc1 (write-through)
        uint32_t stage_vec4;    // detector rows may be staged 4 pixels at a time (base and pitch aligned)
    };

    struct ColConst
    {
        float factor, h, u;
    };

    // x / c for a divisor c that is constant over the launch, with r = RN(1 / c): one multiply and two FMAs
    // (Markstein's correction step) instead of the ~10-instruction IEEE sequence. Only used after
    // fastdiv_validate_kernel has checked, for THIS c and EVERY fp32 x, that the result has the bits of x / c.
    __device__ __forceinline__ float div_by_constant(float x, float c, float r)
    {
        const float q = x * r;
        const float e = __builtin_fmaf(-q, c, x); // exact remainder
        return __builtin_fmaf(e, r, q);
    }

    // src/openmp/backprojection.cpp:116-129,139 for one (x,y) column; K, L are global voxel indices. FD: the division
    // by the horizontal pixel pitch uses the validated multiply + 2 FMA form (h = q - 0.5 has the shape the exhaustive
    // check covers; an h beyond +-2^24 makes the column invalid either way).
    template <bool FD>
    __device__ __forceinline__ ColConst column_constants(const BpParams& g, uint32_t K, uint32_t L)
    {
        const float x_k = g.x_base + static_cast<float>(K) * g.l_vx_x; // :39-43
        const float y_l = g.y_base + static_cast<float>(L) * g.l_vx_y;
        const float s = x_k * g.cos_phi + y_l * g.sin_phi;  // :121
        const float t = -x_k * g.sin_phi + y_l * g.cos_phi; // :122
        const float den = s + g.d_so;
        ColConst c;
        c.factor = g.d_sd / den; // :125
        const float b = (t * c.factor) - g.min_h;
        const float q = FD ? div_by_constant(b, g.l_px_x, g.rcp_l_px_x) : b / g.l_px_x;
        c.h = q - (1.f / 2.f);     // :45-50
        c.u = -(g.d_so / den);     // :139
        return c;
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

    // 1-D grid size for the tile mapping in g.order (tile_of_block rejects the padding blocks)
    inline uint32_t grid_blocks(const BpParams& g)
    {
        const uint32_t total = g.ntx * g.nty * g.ntz;
        if(g.order == 5u)
            return ((total + 7u) / 8u) * 8u;
        if(g.order == 8u)
            return 8u * ((g.nty + 7u) / 8u) * g.ntx * g.ntz;
        return total;
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
    template <int VX, bool NT> __device__ __forceinline__ void store_voxels(float* p, typename vec_of<VX>::type v, bool sc1 = true)
    {
        using ext_t = typename ext_of<VX>::type;
        const ext_t e = *reinterpret_cast<ext_t*>(&v);
        if(NT && VX == 4 && sc1)
        {
            // Write-through + nontemporal ("sc1 nt") is the fastest policy for this once-written stream
            // (tools/membench7.hip: +3 % over "nt" alone). No builtin emits that pair for a plain global store, so the
            // store is inline asm; hipcc neither counts it (nothing waits on a store) nor pads it: the trailing
            // s_nop 1 keeps the next instruction from overwriting the four data registers before they are read.
            asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(e) : "memory");
        }
        else if(NT)
            __builtin_nontemporal_store(e, reinterpret_cast<ext_t*>(p));
        else
            *reinterpret_cast<ext_t*>(p) = e;
    }

    // Workgroup -> tile mapping. Blocks are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8), and
    // which tiles run concurrently decides the DRAM locality of the volume stream (tools/membench5.hip):
    //   0: x tiles fastest, then y, then z (XCD k keeps hitting the same x columns: slowest)
    //   1: z tiles fastest, then x, then y
    //   5: XCD k sweeps its own contiguous eighth of the (x, y, z) tile sequence
    //   8: XCD k owns a band of y tiles; x fastest, then z, then y inside the band (fastest: tools/membench5.hip)
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
        if(g.order == 8u)
        {
            // XCD k owns the contiguous band k of y tiles for the whole slab; inside the band x runs fastest, then the
            // z tile, then y: the tiles in flight on one XCD share their detector rows and sit in few DRAM pages
            const uint32_t band = (g.nty + 7u) / 8u;
            const uint32_t xcd = b % 8u;
            uint32_t r = b / 8u;
            bx = r % g.ntx;
            r /= g.ntx;
            bz = r % g.ntz;
            const uint32_t yb = r / g.ntz;
            by = xcd * band + yb;
            return yb < band && by < g.nty;
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
    // Pieces shared by the two LDS-staged kernels
    // --------------------------------------------------------------------------------------------

    // the staged detector box of one tile (workgroup-uniform)
    struct Box
    {
        int bx0, by0; // first staged detector column / row
        int bw;       // staged columns (0: nothing staged)
        int bhs;      // staged rows
        int stride;   // LDS row stride in floats (odd)
    };

    // Detector bounding box of the voxel tile [k0,k1] x [l0,l1] x [m0,m1]. h is a projective function of (x,y) and
    // v of (z, factor), so the extremes sit on tile corners; the four corners are evaluated by four lanes in
    // parallel and min/max-reduced with two butterfly shuffles (every wave computes the same box). The box is
    // widened by the taps' reach plus one pixel of rounding slack, clipped to the detector and cut to the LDS
    // budget; a tap that still falls outside is served from global memory, so this only has to be right for speed.
    __device__ __forceinline__ Box tile_box(const BpParams& g, uint32_t k0, uint32_t k1, uint32_t l0, uint32_t l1,
                                            uint32_t m0, uint32_t m1, uint32_t lane, uint32_t box_floats)
    {
        const uint32_t ci = lane & 3u;
        const ColConst c = column_constants<false>(g, g.k_off + ((ci & 1u) ? k1 : k0), g.l_off + ((ci & 2u) ? l1 : l0));
        float hmin = c.h, hmax = c.h, fmin = c.factor, fmax = c.factor;
#pragma unroll
        for(int m = 1; m <= 2; m <<= 1)
        {
            hmin = fminf(hmin, __shfl_xor(hmin, m));
            hmax = fmaxf(hmax, __shfl_xor(hmax, m));
            fmin = fminf(fmin, __shfl_xor(fmin, m));
            fmax = fmaxf(fmax, __shfl_xor(fmax, m));
        }
        const float z_c = g.z_base + static_cast<float>(g.m_off + ((ci & 1u) ? m1 : m0)) * g.l_vx_z;
        const float v_c = v_coordinate<false>(g, z_c, (ci & 2u) ? fmax : fmin);
        float vmin = v_c, vmax = v_c;
#pragma unroll
        for(int m = 1; m <= 2; m <<= 1)
        {
            vmin = fminf(vmin, __shfl_xor(vmin, m));
            vmax = fmaxf(vmax, __shfl_xor(vmax, m));
        }
        // identical in every lane by construction; readfirstlane moves them to scalar registers
        const int ihmin = __builtin_amdgcn_readfirstlane(to_int_clamped(hmin));
        const int ihmax = __builtin_amdgcn_readfirstlane(to_int_clamped(hmax));
        const int ivmin = __builtin_amdgcn_readfirstlane(to_int_clamped(vmin));
        const int ivmax = __builtin_amdgcn_readfirstlane(to_int_clamped(vmax));

        Box b;
        b.bx0 = max(ihmin - 1, 0);
        b.by0 = max(ivmin - 1, 0);
        const int bx1 = min(ihmax + 2, static_cast<int>(g.p_dim_x) - 1);
        const int by1 = min(ivmax + 2, static_cast<int>(g.p_dim_y) - 1);
        b.bw = bx1 - b.bx0 + 1;
        int bh = by1 - b.by0 + 1;
        if(b.bw < 2 || bh < 2)
        {
            b.bw = 0;
            bh = 0;
        }
        if(g.stage_vec4 && b.bw > 0)
        {
            // 4-pixel staging: start on a multiple of 4 and cover whole groups of 4. Columns past p_dim_x (< pitch,
            // which is a multiple of 4 here) are readable padding that no valid tap addresses.
            const int end = min((bx1 + 4) & ~3, static_cast<int>(g.p_pitch));
            b.bx0 &= ~3;
            b.bw = end - b.bx0;
            b.stride = b.bw + 4; // multiple of 4: 16-byte aligned rows for ds_write_b128, not a power of two
        }
        else
            b.stride = b.bw | 1;
        b.bhs = min(bh, static_cast<int>(box_floats) / b.stride); // rows that fit the LDS budget
        if(b.bhs < 2)
        {
            // nothing useful fits: stage nothing, every valid tap takes the global path
            b.bw = 0;
            b.bhs = 0;
            b.stride = 1;
        }
        return b;
    }

    // one detector pixel from global memory as fp32 (half pixels widen exactly); idx in pixels
    __device__ __forceinline__ float read_pixel(const BpParams& g, size_t idx)
    {
        if(g.proj_f16)
            return static_cast<float>(static_cast<const volatile _Float16*>(g.proj)[idx]);
        return static_cast<const volatile float*>(g.proj)[idx];
    }

    // global -> LDS. Aligned projections (stage_vec4) move 4 pixels per lane and instruction (16 B fp32 / 8 B half),
    // several short rows per wave-instruction; otherwise one wave per detector row, one pixel per lane.
    __device__ __forceinline__ void stage_box(const BpParams& g, const Box& b, float* lds_box, uint32_t wave,
                                              uint32_t n_waves, uint32_t lane)
    {
        if(g.stage_vec4)
        {
            const uint32_t n4 = static_cast<uint32_t>(b.bw) >> 2; // groups of 4 per row
            if(n4 == 0u)
                return;
            const uint32_t rows_per_pass = n4 <= 64u ? 64u / n4 : 1u; // rows one wave-instruction covers
            const uint32_t lr = n4 <= 64u ? lane / n4 : 0u;
            const uint32_t lc = n4 <= 64u ? lane - lr * n4 : lane;
            if(lr >= rows_per_pass)
                return; // lanes beyond the last whole row of the pass idle
            for(uint32_t r = wave * rows_per_pass + lr; r < static_cast<uint32_t>(b.bhs); r += n_waves * rows_per_pass)
            {
                const size_t row = static_cast<size_t>(b.by0 + static_cast<int>(r)) * g.p_pitch + static_cast<size_t>(b.bx0);
                float* dst = lds_box + r * static_cast<uint32_t>(b.stride);
                for(uint32_t c4 = lc; c4 < n4; c4 += 64u)
                {
                    float4 v;
                    if(g.proj_f16)
                    {
                        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                        const half4 h = *reinterpret_cast<const half4*>(static_cast<const _Float16*>(g.proj) + row + 4u * c4);
                        v = make_float4(static_cast<float>(h.x), static_cast<float>(h.y), static_cast<float>(h.z), static_cast<float>(h.w));
                    }
                    else
                        v = *reinterpret_cast<const float4*>(static_cast<const float*>(g.proj) + row + 4u * c4);
                    *reinterpret_cast<float4*>(dst + 4u * c4) = v;
                }
            }
            return;
        }
        if(g.proj_f16)
        {
            // fp16 projection storage (BASELINE config 5): widened to fp32 here, all arithmetic stays fp32
            for(int r = static_cast<int>(wave); r < b.bhs; r += static_cast<int>(n_waves))
            {
                const _Float16* src = static_cast<const _Float16*>(g.proj) + static_cast<size_t>(b.by0 + r) * g.p_pitch + b.bx0;
                float* dst = lds_box + r * b.stride;
                for(int c = static_cast<int>(lane); c < b.bw; c += 64)
                    dst[c] = static_cast<float>(src[c]);
            }
            return;
        }
        for(int r = static_cast<int>(wave); r < b.bhs; r += static_cast<int>(n_waves))
        {
            const float* src = static_cast<const float*>(g.proj) + static_cast<size_t>(b.by0 + r) * g.p_pitch + b.bx0;
            float* dst = lds_box + r * b.stride;
            for(int c = static_cast<int>(lane); c < b.bw; c += 64)
                dst[c] = src[c];
        }
    }

    // z-invariant state of one (x,y) voxel column
    struct Column
    {
        float factor, u;
        float wx1, wx2; // x interpolation weights
        float ymax;     // p_dim_y, or -inf when the column's x taps are outside the detector
        int xoff;       // LDS column of the left tap, or -1 when it is not inside the staged box
        int x1i;        // detector column of the left tap (global-memory path)
        bool fast;      // every valid tap of this column, over the tile's whole z range, lies inside the staged box
    };

    // z_first / z_last: centred z of the tile's first and last slice. The per-slice coordinate v is a monotone
    // function of the slice index for a fixed column (every step of its evaluation -- multiply by factor, subtract,
    // divide by the pitch, subtract 0.5, each rounded to nearest -- is monotone), so the rows touched over the tile
    // lie between the rows touched at its two end slices: `fast` is decided from those two evaluations alone, with
    // the very expression the slice loop uses, and needs no error bound.
    template <bool FD>
    __device__ __forceinline__ Column make_column(const BpParams& g, const Box& b, uint32_t K, uint32_t L, float z_first,
                                                  float z_last)
    {
        const ColConst c = column_constants<FD>(g, K, L);
        const float x1 = floorf(c.h); // :55-58
        const float x2 = x1 + 1.f;
        const bool x_valid = (x1 >= 0.f) && (x2 < g.p_dim_x_f); // :65-66
        Column col;
        col.factor = c.factor;
        col.u = c.u;
        // :77-78 divide by (x2 - x1), which is exactly 1.f whenever x_valid (|x1| < 2^24): the divisions are the
        // identity and are dropped; for an invalid column the weights are never used
        col.wx2 = x2 - c.h;
        col.wx1 = c.h - x1;
        col.ymax = x_valid ? g.p_dim_y_f : -INFINITY; // folds the x validity into the y2 test
        col.x1i = static_cast<int>(x1);
        const int rel = col.x1i - b.bx0;
        col.xoff = (x_valid && rel >= 0 && rel <= b.bw - 2) ? rel : -1;

        const float va = v_coordinate<FD>(g, z_first, c.factor);
        const float vb = v_coordinate<FD>(g, z_last, c.factor);
        const bool ordered = (va == va) && (vb == vb); // no NaN
        const int r_lo = max(static_cast<int>(floorf(fminf(va, vb))), 0);                              // valid taps start at row 0
        const int r_hi = min(static_cast<int>(floorf(fmaxf(va, vb))), static_cast<int>(g.p_dim_y) - 2); // and end at dim_y - 2
        const bool rows_inside = (r_lo > r_hi) || (r_lo >= b.by0 && r_hi <= b.by0 + b.bhs - 2);
        const bool finite_factor = (c.factor - c.factor) == 0.f; // with a finite factor v is never NaN (it may overflow to inf)
        col.fast = !x_valid || (ordered && finite_factor && col.xoff >= 0 && rows_inside);
        return col;
    }

    // one voxel-update: src/openmp/backprojection.cpp:130-140 for slice coordinate z_m of column col
    // FAST: the column was proven to stay inside the staged box (Column::fast), so the global-memory path and its
    // branch are compiled out and the body is straight-line code the scheduler can overlap across voxels.
    template <bool FD, bool FAST>
    __device__ __forceinline__ float voxel_contribution(const BpParams& g, const Box& b, const float* lds_box, float z_m,
                                                        const Column& col)
    {
        const float v = v_coordinate<FD>(g, z_m, col.factor);
        const float y1 = floorf(v);
        const float y2 = y1 + 1.f;
        const int y1i = static_cast<int>(y1);
        const int rrel = y1i - b.by0;
        const int bhs_m2 = b.bhs - 2;
        bool valid;
        if(FAST)
        {
            // For a `fast` column "valid" (:65-68) is the same as "row and row + 1 inside the staged box": valid taps are
            // inside by construction of `fast`, and the box is clipped to the detector, so a row pair inside it is a
            // valid pair (floor(v) as int equals the float exactly below 2^24; a non-finite v saturates the conversion
            // far outside; `fast` excludes a non-finite factor, the only source of NaN). One unsigned compare, with the
            // column's x validity folded into the limit.
            const unsigned rowlim = col.ymax > 0.f ? static_cast<unsigned>(max(b.bhs - 1, 0)) : 0u;
            valid = static_cast<unsigned>(rrel) < rowlim;
        }
        else
            valid = (y1 >= 0.f) && (y2 < col.ymax); // :67-68 (+ x validity)
        // clamp of the row to the staged box: one median instruction (the compiler's min/max pair cannot know 0 <= hi)
        int rc;
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(rc) : "v"(rrel), "v"(max(bhs_m2, 0)));
        // LDS byte address of the upper left tap = row * stride4 + (box base + 4 * column): the second term is z-invariant
        // (hoisted with the column state), so a tap pair costs one 24-bit multiply-add (rc < 2^12, stride4 < 2^16;
        // v_mul_lo_u32 is quarter rate) and the row below one add. Integer addresses keep the compiler from adding the
        // (zero) link-time base of the dynamic LDS array to every access.
        using lds_cptr = const __attribute__((address_space(3))) float*;
        const int stride4 = b.stride << 2;
        const uint32_t xaddr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_box)) + (static_cast<uint32_t>(max(col.xoff, 0)) << 2);
        const uint32_t a1 = static_cast<uint32_t>(__mul24(rc, stride4)) + xaddr;
        const uint32_t a2 = a1 + static_cast<uint32_t>(stride4);
        lds_cptr r1 = reinterpret_cast<lds_cptr>(a1);
        lds_cptr r2 = reinterpret_cast<lds_cptr>(a2);
        float q11 = r1[0];
        float q21 = r1[1];
        float q12 = r2[0];
        float q22 = r2[1];
        if(!FAST)
        {
            const bool inbox = col.xoff >= 0 && rrel >= 0 && rrel <= bhs_m2;
            if(valid && !inbox)
            {
                // tap outside the staged box: read the detector directly (valid => in bounds). The volatile reads in
                // read_pixel keep the compiler from merging these loads with the LDS reads into flat loads
                const size_t at = static_cast<size_t>(y1i) * g.p_pitch + col.x1i;
                q11 = read_pixel(g, at);
                q21 = read_pixel(g, at + 1);
                q12 = read_pixel(g, at + g.p_pitch);
                q22 = read_pixel(g, at + g.p_pitch + 1);
            }
        }
        const float interp_y1 = col.wx2 * q11 + col.wx1 * q21; // :77
        const float interp_y2 = col.wx2 * q12 + col.wx1 * q22; // :78
        // :80 divides by (y2 - y1) == 1.f exactly whenever valid -- dropped, as above.
        // For a valid tap (v >= 0) both v - y1 and y2 - v are exact: below 1 they are v and RN(1 - v), from 1 up multiples
        // of ulp(v) >= 2^-23 inside [0, 1]. So y2 - v == 1 - (v - y1) bit for bit, and the fast path (which does not need y2
        // for its validity test) saves the addition; for an invalid tap the value is discarded below.
        const float wy1 = v - y1;
        const float wy2 = FAST ? 1.f - wy1 : y2 - v;
        float det = wy2 * interp_y1 + wy1 * interp_y2;
        det = valid ? det : 0.f;         // :71
        return 0.5f * det * col.u * col.u; // :140
    }

    // --------------------------------------------------------------------------------------------
    // Tile kernel (z-walk in time). 1-D grid over ntx * nty * ntz tiles of 64 x TY x g.tz voxels, TY = 4*VX.
    // Serves every alignment (VX = 1, 2, 4).
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

        const Box box = tile_box(g, k0, k1, l0, l1, m0, m1, lane, g.lds_floats);
        stage_box(g, box, lds, wave, 4u, lane);

        // ---- per-lane columns --------------------------------------------------------------------
        const uint32_t xq = lane % XL;
        const uint32_t yy = lane / XL;
        const uint32_t k = k0 + xq * VX;
        const uint32_t l = l0 + wave * RW + yy;
        const bool active = k < g.v_dim_x && l < g.v_dim_y;

        const float z_first = g.z_base + static_cast<float>(g.m_off + m0) * g.l_vx_z;
        const float z_last = g.z_base + static_cast<float>(g.m_off + m1) * g.l_vx_z;
        Column col[VX];
        bool all_fast = true;
#pragma unroll
        for(int j = 0; j < VX; ++j)
        {
            col[j] = make_column<FD>(g, box, g.k_off + min(k + j, k1), g.l_off + min(l, l1), z_first, z_last);
            all_fast = all_fast && col[j].fast;
        }

        // (keeping the box loads in flight across this setup was measured slower: profiles/r01_ab_full_volume_6.jsonl)
        __syncthreads();
        if(!active)
            return;

        using vec_t = typename vec_of<VX>::type;
        const size_t slice = static_cast<size_t>(g.v_dim_x) * g.v_dim_y;
        float* vp = g.vol + (static_cast<size_t>(m0) * g.v_dim_y + l) * g.v_dim_x + k;
        const uint32_t mcount = m1 - m0 + 1u;

        auto walk = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            auto update = [&](vec_t& acc, uint32_t m_local) {
                const float z_m = g.z_base + static_cast<float>(g.m_off + m0 + m_local) * g.l_vx_z; // :118
#pragma unroll
                for(int j = 0; j < VX; ++j)
                    elem<VX>(acc, j) += voxel_contribution<FD, FAST>(g, box, lds, z_m, col[j]);
            };
            if constexpr(UNROLL == 3)
            {
                // one slice at a time, with the next slice's load issued before the current one is updated and stored:
                // two loads in flight per lane for 4 extra registers instead of a second unrolled body
                vec_t cur = load_voxels<VX, NT>(vp);
                for(uint32_t mm = 0; mm < mcount; ++mm)
                {
                    vec_t nxt = cur;
                    if(mm + 1u < mcount)
                        nxt = load_voxels<VX, NT>(vp + (mm + 1u) * slice);
                    update(cur, mm);
                    store_voxels<VX, NT>(vp + mm * slice, cur, g.store_sc1 != 0u);
                    cur = nxt;
                }
            }
            else
            {
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
                        store_voxels<VX, NT>(vp + (mm + i) * slice, acc[i], g.store_sc1 != 0u);
                }
                for(; mm < mcount; ++mm)
                {
                    vec_t acc = load_voxels<VX, NT>(vp + mm * slice);
                    update(acc, mm);
                    store_voxels<VX, NT>(vp + mm * slice, acc, g.store_sc1 != 0u);
                }
            }
        };
        if(all_fast)
            walk(std::true_type{});
        else
            walk(std::false_type{}); // some tap of this lane may leave the staged box: per-voxel check + global path
    }

    // --------------------------------------------------------------------------------------------
    // Slice kernel (z across waves). A workgroup of NW waves owns a tile of 64 x (4*RPL) columns x NW slices; wave w
    // updates slice w, so a thread issues its RPL 16-byte loads once, updates 4*RPL voxels and stores: the volume
    // stream has the shape of the fastest plain sweep (tools/membench6.hip: 6.0 TB/s against 5.5 for the z-walk).
    // The z-invariant column state that the tile kernel keeps in registers is computed once per workgroup (one
    // thread per column) and shared through LDS as structure-of-arrays, read back as float4 per 4 columns.
    // Needs dim_x % 4 == 0 and a 16-byte aligned volume. LDS: 8 * 64 * 4*RPL words of column state + the box.
    // --------------------------------------------------------------------------------------------
    constexpr uint32_t SLICE_STATE_ARRAYS = 8u;

    template <int NW, int RPL, bool NT, bool FD>
    __global__ void __launch_bounds__(NW * 64, 8) bp_slice_kernel(const BpParams g)
    {
        extern __shared__ __attribute__((aligned(16))) float lds[];
        constexpr uint32_t TY = 4u * RPL;
        constexpr uint32_t NCOL = 64u * TY;
        float* c_factor = lds;
        float* c_u = lds + NCOL;
        float* c_wx1 = lds + 2u * NCOL;
        float* c_wx2 = lds + 3u * NCOL;
        float* c_ymax = lds + 4u * NCOL;
        int* c_xoff = reinterpret_cast<int*>(lds + 5u * NCOL);
        int* c_x1i = reinterpret_cast<int*>(lds + 6u * NCOL);
        int* c_fast = reinterpret_cast<int*>(lds + 7u * NCOL);
        float* lds_box = lds + SLICE_STATE_ARRAYS * NCOL;

        const uint32_t tid = threadIdx.x;
        const uint32_t lane = tid & 63u;
        const uint32_t wave = tid >> 6;

        uint32_t bx, by, bz;
        if(!tile_of_block(g, blockIdx.x, bx, by, bz))
            return;
        const uint32_t k0 = bx * 64u;
        const uint32_t l0 = by * TY;
        const uint32_t m0 = bz * NW;
        const uint32_t k1 = min(k0 + 63u, g.v_dim_x - 1u);
        const uint32_t l1 = min(l0 + TY - 1u, g.v_dim_y - 1u);
        const uint32_t m1 = min(m0 + NW - 1u, g.v_dim_z - 1u);

        const Box box = tile_box(g, k0, k1, l0, l1, m0, m1, lane, g.lds_floats - SLICE_STATE_ARRAYS * NCOL);
        stage_box(g, box, lds_box, wave, NW, lane);

        // column state: thread t computes columns t, t + NW*64, ...; column c = cy * 64 + cx
        const float z_first = g.z_base + static_cast<float>(g.m_off + m0) * g.l_vx_z;
        const float z_last = g.z_base + static_cast<float>(g.m_off + m1) * g.l_vx_z;
        for(uint32_t c = tid; c < NCOL; c += NW * 64u)
        {
            const uint32_t cx = c & 63u, cy = c >> 6;
            const Column col = make_column<FD>(g, box, g.k_off + min(k0 + cx, k1), g.l_off + min(l0 + cy, l1), z_first, z_last);
            c_factor[c] = col.factor;
            c_u[c] = col.u;
            c_wx1[c] = col.wx1;
            c_wx2[c] = col.wx2;
            c_ymax[c] = col.ymax;
            c_xoff[c] = col.xoff;
            c_x1i[c] = col.x1i;
            c_fast[c] = col.fast ? 1 : 0;
        }
        __syncthreads();

        const uint32_t m = m0 + wave;
        if(m > m1)
            return;
        const uint32_t xq = lane & 15u, yy = lane >> 4;
        const uint32_t k = k0 + xq * 4u;
        if(k >= g.v_dim_x)
            return;
        const float z_m = g.z_base + static_cast<float>(g.m_off + m) * g.l_vx_z; // :118
        float* vp = g.vol + (static_cast<size_t>(m) * g.v_dim_y + l0 + yy) * g.v_dim_x + k;
        const size_t row4 = static_cast<size_t>(4u) * g.v_dim_x;

        float4 acc[RPL];
        bool all_fast = true;
#pragma unroll
        for(int r = 0; r < RPL; ++r)
            if(l0 + yy + 4u * r < g.v_dim_y)
            {
                acc[r] = load_voxels<4, NT>(vp + r * row4);
                const int4 f = *reinterpret_cast<const int4*>(c_fast + (yy + 4u * r) * 64u + xq * 4u);
                all_fast = all_fast && ((f.x & f.y & f.z & f.w) != 0);
            }
        auto add = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
            for(int r = 0; r < RPL; ++r)
            {
                if(l0 + yy + 4u * r >= g.v_dim_y)
                    continue;
                const uint32_t c = (yy + 4u * r) * 64u + xq * 4u;
                const float4 f4 = *reinterpret_cast<const float4*>(c_factor + c);
                const float4 u4 = *reinterpret_cast<const float4*>(c_u + c);
                const float4 a4 = *reinterpret_cast<const float4*>(c_wx1 + c);
                const float4 b4 = *reinterpret_cast<const float4*>(c_wx2 + c);
                const float4 y4 = *reinterpret_cast<const float4*>(c_ymax + c);
                const int4 o4 = *reinterpret_cast<const int4*>(c_xoff + c);
                const int4 i4 = *reinterpret_cast<const int4*>(c_x1i + c);
                acc[r].x += voxel_contribution<FD, FAST>(g, box, lds_box, z_m, Column{f4.x, u4.x, a4.x, b4.x, y4.x, o4.x, i4.x, FAST});
                acc[r].y += voxel_contribution<FD, FAST>(g, box, lds_box, z_m, Column{f4.y, u4.y, a4.y, b4.y, y4.y, o4.y, i4.y, FAST});
                acc[r].z += voxel_contribution<FD, FAST>(g, box, lds_box, z_m, Column{f4.z, u4.z, a4.z, b4.z, y4.z, o4.z, i4.z, FAST});
                acc[r].w += voxel_contribution<FD, FAST>(g, box, lds_box, z_m, Column{f4.w, u4.w, a4.w, b4.w, y4.w, o4.w, i4.w, FAST});
            }
        };
        if(all_fast)
            add(std::true_type{});
        else
            add(std::false_type{});
#pragma unroll
        for(int r = 0; r < RPL; ++r)
            if(l0 + yy + 4u * r < g.v_dim_y)
                store_voxels<4, NT>(vp + r * row4, acc[r]);
    }

    // --------------------------------------------------------------------------------------------
    // Fused kernel (extension: paris_hip_backproject_batch). One launch adds n_proj projections: a lane keeps its
    // 4 x TZ voxels in registers, and for every projection in turn the workgroup stages that projection's box,
    // rebuilds the column state and adds the TZ contributions -- in projection order, so every voxel sees exactly
    // the additions, in exactly the order, of n_proj single launches (bit-identical), while the volume is read
    // and written once per batch: 8 / n_proj bytes per voxel-update. With the HBM term gone the kernel is bound by
    // vector ALU issue (about 40 instructions per voxel-update plus the per-projection column setup).
    // --------------------------------------------------------------------------------------------
    constexpr int FUSED_MAX = 32;

    struct FusedParams
    {
        BpParams g;           // proj = first projection; sin/cos overwritten per projection
        uint32_t n_proj;
        uint32_t proj_stride; // pixels between consecutive projections
        float sin_phi[FUSED_MAX];
        float cos_phi[FUSED_MAX];
    };

    // (Double-buffering the box with the next projection's loads kept in flight was measured slower: the kernel is
    // bound by vector ALU issue, not by staging latency, and the extra live registers cost occupancy.)
    template <int TZ, bool NT, bool FD>
    __global__ void __launch_bounds__(256) bp_fused_kernel(const FusedParams fp)
    {
        extern __shared__ __attribute__((aligned(16))) float lds[];
        BpParams g = fp.g;

        const uint32_t tid = threadIdx.x;
        const uint32_t lane = tid & 63u;
        const uint32_t wave = tid >> 6;

        uint32_t bx, by, bz;
        if(!tile_of_block(g, blockIdx.x, bx, by, bz))
            return;
        const uint32_t k0 = bx * 64u;
        const uint32_t l0 = by * 16u;
        const uint32_t m0 = bz * TZ;
        const uint32_t k1 = min(k0 + 63u, g.v_dim_x - 1u);
        const uint32_t l1 = min(l0 + 15u, g.v_dim_y - 1u);
        const uint32_t m1 = min(m0 + TZ - 1u, g.v_dim_z - 1u);
        const uint32_t mcount = m1 - m0 + 1u;

        const uint32_t xq = lane & 15u, yy = lane >> 4;
        const uint32_t k = k0 + xq * 4u;
        const uint32_t l = l0 + wave * 4u + yy;
        const bool active = k < g.v_dim_x && l < g.v_dim_y; // inactive lanes still take part in the barriers

        const size_t slice = static_cast<size_t>(g.v_dim_x) * g.v_dim_y;
        float* vp = g.vol + (static_cast<size_t>(m0) * g.v_dim_y + l) * g.v_dim_x + k;
        float4 acc[TZ];
#pragma unroll
        for(int z = 0; z < TZ; ++z)
            if(active && static_cast<uint32_t>(z) < mcount)
                acc[z] = load_voxels<4, NT>(vp + z * slice);

        const float z_first = g.z_base + static_cast<float>(g.m_off + m0) * g.l_vx_z;
        const float z_last = g.z_base + static_cast<float>(g.m_off + m1) * g.l_vx_z;
        const char* base = static_cast<const char*>(fp.g.proj);
        const size_t px = g.proj_f16 ? 2u : 4u;
        for(uint32_t p = 0; p < fp.n_proj; ++p)
        {
            g.sin_phi = fp.sin_phi[p];
            g.cos_phi = fp.cos_phi[p];
            g.proj = base + static_cast<size_t>(p) * fp.proj_stride * px;
            const Box box = tile_box(g, k0, k1, l0, l1, m0, m1, lane, g.lds_floats);
            __syncthreads(); // the previous projection's taps are done with the LDS box
            stage_box(g, box, lds, wave, 4u, lane);
            __syncthreads();
            if(active)
            {
                Column col[4];
                bool all_fast = true;
#pragma unroll
                for(int j = 0; j < 4; ++j)
                {
                    col[j] = make_column<FD>(g, box, g.k_off + k + j, g.l_off + l, z_first, z_last);
                    all_fast = all_fast && col[j].fast;
                }
                auto add_projection = [&](auto fast_tag, auto full_tag) {
                    constexpr bool FAST = decltype(fast_tag)::value;
                    constexpr bool FULL = decltype(full_tag)::value; // whole tile: no per-slice test, one straight block
#pragma unroll
                    for(int z = 0; z < TZ; ++z)
                    {
                        if(FULL || static_cast<uint32_t>(z) < mcount) // uniform; no break, so acc stays in registers
                        {
                            const float z_m = g.z_base + static_cast<float>(g.m_off + m0 + z) * g.l_vx_z; // :118
                            acc[z].x += voxel_contribution<FD, FAST>(g, box, lds, z_m, col[0]);
                            acc[z].y += voxel_contribution<FD, FAST>(g, box, lds, z_m, col[1]);
                            acc[z].z += voxel_contribution<FD, FAST>(g, box, lds, z_m, col[2]);
                            acc[z].w += voxel_contribution<FD, FAST>(g, box, lds, z_m, col[3]);
                        }
                    }
                };
                if(all_fast && mcount == TZ)
                    add_projection(std::true_type{}, std::true_type{});
                else if(all_fast)
                    add_projection(std::true_type{}, std::false_type{});
                else
                    add_projection(std::false_type{}, std::false_type{});
            }
        }
#pragma unroll
        for(int z = 0; z < TZ; ++z)
            if(active && static_cast<uint32_t>(z) < mcount)
                store_voxels<4, NT>(vp + z * slice, acc[z]);
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
        const ColConst c = column_constants<false>(g, g.k_off + k, g.l_off + l);
        const float z_m = g.z_base + static_cast<float>(g.m_off + m) * g.l_vx_z;
        const float x = c.h;
        const float y = v_coordinate<false>(g, z_m, c.factor);
        const float x1 = floorf(x), x2 = x1 + 1.f, y1 = floorf(y), y2 = y1 + 1.f;
        float interp = 0.f;
        if(x1 >= 0.f && x2 < g.p_dim_x_f && y1 >= 0.f && y2 < g.p_dim_y_f)
        {
            const size_t at = static_cast<size_t>(static_cast<uint32_t>(y1)) * g.p_pitch + static_cast<uint32_t>(x1);
            const float q11 = read_pixel(g, at), q21 = read_pixel(g, at + 1);
            const float q12 = read_pixel(g, at + g.p_pitch), q22 = read_pixel(g, at + g.p_pitch + 1);
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
        uint32_t blocks = grid_blocks(g);
        hipLaunchKernelGGL((bp_tile_kernel<VX, UNROLL, NT, FD>), dim3(blocks), dim3(256), g.lds_floats * sizeof(float), stream, g);
    }

    template <int VX, bool NT, bool FD>
    void launch_tile_unroll(BpParams& g, int unroll, hipStream_t stream)
    {
        switch(unroll)
        {
            case 1: launch_tile<VX, 1, NT, FD>(g, stream); break;
            case 2: launch_tile<VX, 2, NT, FD>(g, stream); break;
            case 3: launch_tile<VX, 3, NT, FD>(g, stream); break; // 1 + prefetch of the next slice
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

    // ---- slice kernel launchers ------------------------------------------------------------------------------
    template <int NW, int RPL, bool NT, bool FD>
    int launch_slice(BpParams& g, uint32_t box_bytes, hipStream_t stream)
    {
        constexpr uint32_t TY = 4u * RPL;
        constexpr uint32_t state_floats = SLICE_STATE_ARRAYS * 64u * TY;
        g.tz = NW;
        g.lds_floats = state_floats + box_bytes / sizeof(float);
        g.ntx = (g.v_dim_x + 63u) / 64u;
        g.nty = (g.v_dim_y + TY - 1u) / TY;
        g.ntz = (g.v_dim_z + NW - 1u) / NW;
        uint32_t blocks = grid_blocks(g);
        const uint32_t lds_bytes = g.lds_floats * sizeof(float);
        if(lds_bytes > 64u * 1024u) // beyond the default dynamic-LDS limit (only with a raised box budget): per launch, cheap
            PARIS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(bp_slice_kernel<NW, RPL, NT, FD>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((bp_slice_kernel<NW, RPL, NT, FD>), dim3(blocks), dim3(NW * 64), lds_bytes, stream, g);
        return PARIS_HIP_SUCCESS;
    }

    template <int NW, int RPL>
    int launch_slice_flags(BpParams& g, uint32_t box_bytes, bool nt, bool fd, hipStream_t stream)
    {
        if(nt && fd)
            return launch_slice<NW, RPL, true, true>(g, box_bytes, stream);
        if(nt)
            return launch_slice<NW, RPL, true, false>(g, box_bytes, stream);
        if(fd)
            return launch_slice<NW, RPL, false, true>(g, box_bytes, stream);
        return launch_slice<NW, RPL, false, false>(g, box_bytes, stream);
    }

    // shape = (waves = slices per tile, row groups per lane); supported: 16x4 16x2 8x4 8x2 8x1
    int launch_slice_shape(BpParams& g, int nw, int rpl, uint32_t box_bytes, bool nt, bool fd, hipStream_t stream)
    {
        if(nw == 16 && rpl == 4) return launch_slice_flags<16, 4>(g, box_bytes, nt, fd, stream);
        if(nw == 16 && rpl == 2) return launch_slice_flags<16, 2>(g, box_bytes, nt, fd, stream);
        if(nw == 8 && rpl == 4) return launch_slice_flags<8, 4>(g, box_bytes, nt, fd, stream);
        if(nw == 8 && rpl == 2) return launch_slice_flags<8, 2>(g, box_bytes, nt, fd, stream);
        if(nw == 8 && rpl == 1) return launch_slice_flags<8, 1>(g, box_bytes, nt, fd, stream);
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    }

    template <int TZ, bool NT, bool FD>
    void launch_fused(FusedParams& fp, hipStream_t stream)
    {
        BpParams& g = fp.g;
        g.tz = TZ;
        g.ntx = (g.v_dim_x + 63u) / 64u;
        g.nty = (g.v_dim_y + 15u) / 16u;
        g.ntz = (g.v_dim_z + TZ - 1u) / TZ;
        uint32_t blocks = grid_blocks(g);
        hipLaunchKernelGGL((bp_fused_kernel<TZ, NT, FD>), dim3(blocks), dim3(256), g.lds_floats * sizeof(float), stream, fp);
    }

    template <int TZ>
    void launch_fused_flags(FusedParams& fp, bool nt, bool fd, hipStream_t stream)
    {
        if(nt && fd)
            launch_fused<TZ, true, true>(fp, stream);
        else if(nt)
            launch_fused<TZ, true, false>(fp, stream);
        else if(fd)
            launch_fused<TZ, false, true>(fp, stream);
        else
            launch_fused<TZ, false, false>(fp, stream);
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

// validates the arguments of one backprojection and derives the kernel parameters; *skip = true for an empty volume
static int fill_params(paris_hip_ctx* ctx, const void* d_p, bool f16, size_t p_pitch, uint32_t p_dim_x, uint32_t p_dim_y,
                       float* d_v, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                       const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo, int enable_roi,
                       const paris_region_of_interest* roi, float sin_phi, float cos_phi, float delta_s, float delta_t,
                       BpParams& g, bool& fd, bool& skip)
{
    const size_t px = f16 ? sizeof(uint16_t) : sizeof(float);
    skip = false;
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_p == nullptr || d_v == nullptr || det_geo == nullptr || vol_geo == nullptr || (enable_roi && roi == nullptr))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(p_dim_x == 0 || p_dim_y == 0 || p_pitch < static_cast<size_t>(p_dim_x) * px || p_pitch % px != 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(v_dim_x == 0 || v_dim_y == 0 || v_dim_z == 0)
    {
        skip = true;
        return PARIS_HIP_SUCCESS;
    }
    const uint32_t tz = ctx->bp_tz ? ctx->bp_tz : TZ_DEFAULT;
    {
        // the 1-D grid must hold every tile (narrowest tile: 64 x 4 x tz)
        const uint64_t tiles = static_cast<uint64_t>((v_dim_x + 63u) / 64u) * ((v_dim_y + 3u) / 4u) * ((v_dim_z + std::min(tz, 8u) - 1u) / std::min(tz, 8u));
        if(tiles > 0x7fffff00ull)
            return PARIS_HIP_ERROR_UNSUPPORTED;
    }

    g = BpParams{};
    g.proj = d_p;
    g.proj_f16 = f16 ? 1u : 0u;
    g.vol = d_v;
    g.p_pitch = static_cast<uint32_t>(p_pitch / px);
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
    g.rcp_l_px_x = 1.f / g.l_px_x;
    fd = false;
    if(ctx->bp_fastdiv != 0)
    {
        bool fd_x = false; // both pitches must pass (usually the same divisor: one cached check)
        if(int rc = fastdiv_is_exact(ctx, g.l_px_y, &fd))
            return rc;
        if(int rc = fastdiv_is_exact(ctx, g.l_px_x, &fd_x))
            return rc;
        fd = fd && fd_x;
    }
    g.p_dim_x_f = static_cast<float>(p_dim_x);
    g.p_dim_y_f = static_cast<float>(p_dim_y);
    g.lds_floats = (ctx->bp_lds_bytes ? ctx->bp_lds_bytes : LDS_BYTES_DEFAULT) / sizeof(float);
    g.tz = tz;
    // default mapping: a contiguous run of tiles per XCD (best or within 0.4 % of the best on 1024^3 ... 2048^3 and slabs;
    // the y-band mapping 8 ties at 2048 rows and loses 12 % at 1024: tools/ab_bp.py, tools/tune_bp.py)
    g.order = ctx->bp_order >= 0 ? static_cast<uint32_t>(ctx->bp_order) : 5u;
    g.store_sc1 = ctx->bp_nt == 2 ? 1u : 0u;
    // 4-pixel staging needs every detector row to start 16-byte (half: 8-byte) aligned
    g.stage_vec4 = (ctx->bp_stage_vec4 != 0 && g.p_pitch % 4u == 0 && reinterpret_cast<uintptr_t>(d_p) % (4u * px) == 0)
                       ? 1u : 0u;
    return PARIS_HIP_SUCCESS;
}

// widest per-lane access the volume's alignment allows
static int lane_width(const float* d_v, uint32_t v_dim_x)
{
    const uintptr_t addr = reinterpret_cast<uintptr_t>(d_v);
    if(v_dim_x % 4u == 0 && addr % 16u == 0)
        return 4;
    if(v_dim_x % 2u == 0 && addr % 8u == 0)
        return 2;
    return 1;
}

static int backproject_impl(paris_hip_ctx* ctx, const void* d_p, bool f16, size_t p_pitch, uint32_t p_dim_x,
                            uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z,
                            uint32_t v_offset, const paris_detector_geometry* det_geo,
                            const paris_volume_geometry* vol_geo, int enable_roi, const paris_region_of_interest* roi,
                            float sin_phi, float cos_phi, float delta_s, float delta_t)
{
    BpParams g;
    bool fd = false, skip = false;
    if(int rc = fill_params(ctx, d_p, f16, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo, vol_geo,
                            enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t, g, fd, skip))
        return rc;
    if(skip)
        return paris_hip_finish(ctx);

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
        int vx = lane_width(d_v, v_dim_x); // unless the tuning knob asks for less
        if(ctx->bp_vx && ctx->bp_vx < vx)
            vx = ctx->bp_vx;
        const int unroll = ctx->bp_unroll ? ctx->bp_unroll : 2; // interleaved A/B on 2048^3 (tools/ab_bp.py, profiles/)
        const bool nt = ctx->bp_nt != 0;
        const bool want_slice = ctx->bp_variant == 3; // measured slower than the tile kernel so far: opt-in only
        if(want_slice && vx == 4)
        {
            const int nw = ctx->bp_slice_nw ? ctx->bp_slice_nw : 16;
            const int rpl = ctx->bp_slice_rpl ? ctx->bp_slice_rpl : 4;
            const uint32_t box_bytes = ctx->bp_lds_bytes ? ctx->bp_lds_bytes : LDS_BYTES_DEFAULT;
            if(int rc = launch_slice_shape(g, nw, rpl, box_bytes, nt, fd, ctx->stream))
                return rc;
        }
        else if(vx == 4)
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

extern "C" int paris_hip_backproject(paris_hip_ctx* ctx, const float* d_p, size_t p_pitch, uint32_t p_dim_x,
                                     uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y,
                                     uint32_t v_dim_z, uint32_t v_offset, const paris_detector_geometry* det_geo,
                                     const paris_volume_geometry* vol_geo, int enable_roi,
                                     const paris_region_of_interest* roi, float sin_phi, float cos_phi,
                                     float delta_s, float delta_t)
{
    return backproject_impl(ctx, d_p, false, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo,
                            vol_geo, enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t);
}

extern "C" int paris_hip_backproject_f16(paris_hip_ctx* ctx, const uint16_t* d_p, size_t p_pitch, uint32_t p_dim_x,
                                         uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y,
                                         uint32_t v_dim_z, uint32_t v_offset, const paris_detector_geometry* det_geo,
                                         const paris_volume_geometry* vol_geo, int enable_roi,
                                         const paris_region_of_interest* roi, float sin_phi, float cos_phi,
                                         float delta_s, float delta_t)
{
    return backproject_impl(ctx, d_p, true, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo,
                            vol_geo, enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t);
}

namespace
{
    // fp32 -> IEEE half, round to nearest even (v_cvt_f16_f32), one pixel per thread
    __global__ void __launch_bounds__(256) to_half_kernel(const float* __restrict__ src, uint32_t src_pitch,
                                                           _Float16* __restrict__ dst, uint32_t dst_pitch, uint32_t dim_x,
                                                           uint32_t dim_y)
    {
        const uint32_t s = blockIdx.x * 256u + threadIdx.x;
        if(s >= dim_x)
            return;
        for(uint32_t t = blockIdx.y; t < dim_y; t += gridDim.y)
            dst[static_cast<size_t>(t) * dst_pitch + s] = static_cast<_Float16>(src[static_cast<size_t>(t) * src_pitch + s]);
    }
}

extern "C" int paris_hip_convert_projection_f16(paris_hip_ctx* ctx, const float* d_src, size_t src_pitch,
                                                uint16_t* d_dst, size_t dst_pitch, uint32_t dim_x, uint32_t dim_y)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(d_src == nullptr || d_dst == nullptr || src_pitch % sizeof(float) != 0 || dst_pitch % sizeof(uint16_t) != 0
       || src_pitch < static_cast<size_t>(dim_x) * sizeof(float) || dst_pitch < static_cast<size_t>(dim_x) * sizeof(uint16_t))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(dim_x == 0 || dim_y == 0)
        return paris_hip_finish(ctx);
    const dim3 grid((dim_x + 255u) / 256u, dim_y < 65535u ? dim_y : 65535u);
    hipLaunchKernelGGL(to_half_kernel, grid, dim3(256), 0, ctx->stream, d_src, static_cast<uint32_t>(src_pitch / sizeof(float)),
                       reinterpret_cast<_Float16*>(d_dst), static_cast<uint32_t>(dst_pitch / sizeof(uint16_t)), dim_x, dim_y);
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
    if(ctx == nullptr || sin_phi == nullptr || cos_phi == nullptr || p_stride_bytes % sizeof(float) != 0)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(n_proj == 0)
        return paris_hip_finish(ctx);

    // The fused kernel needs 16-byte lanes; any other volume, and the cross-check variants, take the sequence of
    // single-projection launches that the fused kernel is defined to equal.
    // A single projection is the tile kernel's case (memory bound, nothing to fuse).
    const bool fused_ok = n_proj > 1 && (ctx->bp_variant == 0 || ctx->bp_variant == 4) && d_v != nullptr
                          && lane_width(d_v, v_dim_x) == 4 && (ctx->bp_vx == 0 || ctx->bp_vx == 4);
    if(!fused_ok)
    {
        const unsigned saved = ctx->flags;
        ctx->flags &= ~PARIS_HIP_CTX_SYNCHRONOUS;
        int rc = PARIS_HIP_SUCCESS;
        for(uint32_t i = 0; i < n_proj && rc == PARIS_HIP_SUCCESS; ++i)
        {
            const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(d_p) + i * p_stride_bytes);
            rc = paris_hip_backproject(ctx, p, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo, vol_geo,
                                       enable_roi, roi, sin_phi[i], cos_phi[i], delta_s, delta_t);
        }
        ctx->flags = saved;
        if(rc != PARIS_HIP_SUCCESS)
            return rc;
        return paris_hip_finish(ctx);
    }

    const bool nt = ctx->bp_nt != 0;
    const bool tz16 = ctx->bp_tz != 8u; // 16 slices per tile unless 8 is asked for (tools/tune_bp.py --fused)
    for(uint32_t first = 0; first < n_proj; first += FUSED_MAX)
    {
        const uint32_t n = std::min<uint32_t>(FUSED_MAX, n_proj - first);
        FusedParams fp;
        bool fd = false, skip = false;
        const char* p0 = reinterpret_cast<const char*>(d_p) + static_cast<size_t>(first) * p_stride_bytes;
        if(int rc = fill_params(ctx, p0, false, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo,
                                vol_geo, enable_roi, roi, sin_phi[first], cos_phi[first], delta_s, delta_t, fp.g, fd, skip))
            return rc;
        if(skip)
            break;
        fp.n_proj = n;
        fp.proj_stride = static_cast<uint32_t>(p_stride_bytes / sizeof(float));
        for(uint32_t i = 0; i < n; ++i)
        {
            fp.sin_phi[i] = sin_phi[first + i];
            fp.cos_phi[i] = cos_phi[first + i];
        }
        const size_t ev = static_cast<size_t>(ctx->bp_launches % ctx->bp_start.size());
        PARIS_HIP_TRY(hipEventRecord(ctx->bp_start[ev], ctx->stream));
        if(tz16)
            launch_fused_flags<16>(fp, nt, fd, ctx->stream);
        else
            launch_fused_flags<8>(fp, nt, fd, ctx->stream);
        PARIS_HIP_TRY(hipEventRecord(ctx->bp_stop[ev], ctx->stream));
        ++ctx->bp_launches;
    }
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
    if(ctx == nullptr || variant < 0 || variant > 4)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_variant = variant;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_order(paris_hip_ctx* ctx, int order, int nontemporal)
{
    if(ctx == nullptr || !(order == -1 || order == 0 || order == 1 || order == 5 || order == 8) || nontemporal < -1 || nontemporal > 2)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_order = order;
    ctx->bp_nt = nontemporal < 0 ? 2 : nontemporal;
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

extern "C" int paris_hip_set_backproject_vector_staging(paris_hip_ctx* ctx, int enable)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_stage_vec4 = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_fast_division(paris_hip_ctx* ctx, int enable)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_fastdiv = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_slice_shape(paris_hip_ctx* ctx, int waves, int row_groups)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const bool ok = (waves == 0 && row_groups == 0) || (waves == 16 && (row_groups == 4 || row_groups == 2))
                    || (waves == 8 && (row_groups == 4 || row_groups == 2 || row_groups == 1));
    if(!ok)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_slice_nw = waves;
    ctx->bp_slice_rpl = row_groups;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_tuning(paris_hip_ctx* ctx, int vx, int unroll, int tz, int lds_bytes)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(!(vx == 0 || vx == 1 || vx == 2 || vx == 4) || !(unroll == 0 || unroll == 1 || unroll == 2 || unroll == 3 || unroll == 4))
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
