// Device-side building blocks shared by the backprojection kernels (backproject.hip: tile / slice / gather kernels and
// the host entry points; backproject_fused.hip: the fused multi-projection kernel, a translation unit of its own because
// it is compiled with -fno-slp-vectorize, see the Makefile). Everything here has internal linkage.
#ifndef PARIS_HIP_BP_DEVICE_H_
#define PARIS_HIP_BP_DEVICE_H_

#include "paris_hip_internal.h"
#include "ieee_lean.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

// 1: the fast path takes floor(v) with v_cvt_flr_i32_f32 and the row weight with v_fract_f32 (one instruction less per
// voxel-update; both exhaustively equal to the floorf forms, tools/flr_probe.hip). Worth +1 % in the ALU-bound fused kernel,
// which sets it; the memory-bound tile kernel measured 0.3 % slower with the inline asm in its loop and keeps the plain form.
#ifndef PARIS_BP_SINGLE_INSTRUCTION_FLOOR
#define PARIS_BP_SINGLE_INSTRUCTION_FLOOR 0
#endif

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
        uint32_t zchunk;        // order 12: z tiles per chunk
        uint32_t lean_div;      // the two divisions by s + d_so may share one reciprocal (operands far from the fp32 range limits)
        const float* colstate;  // two-pass variant: factor, h, u planes of v_dim_x * v_dim_y floats each (NULL: computed in the kernel)
        uint32_t store_sc1;     // nontemporal stores also carry sc1 (write-through)
        uint32_t stage_vec4;    // detector rows may be staged 4 pixels at a time (base and pitch aligned)
        // The volume holds no -0 (zero-filled by the library and written by backprojections only since: a sum of floats is -0 only
        // if both terms are): adding +0 then changes nothing, and waves whose columns all have Column::none skip the tile
        uint32_t skip_invalid;
        // Nesting inside the dealt orders 14 .. 17. 0: x tiles fastest, then the z tile inside the chunk, then the XCD's y tiles, then
        // the chunk (round 3's first form). 2: x fastest, then the y tiles of one dealt group, then the z tile, then the XCD's next
        // group -- the single-projection kernel's default: the resident set of an XCD is 32 adjacent rows of every x tile and two z
        // tiles instead of 16 rows and four (2048^3, four interleaved pairs on two devices: +0.2 ... +0.3 % of the HBM fraction). 1: the
        // y tiles of one dealt group fastest, then x, then the XCD's next group (order 18: y fastest, then x) -- the fused kernel,
        // whose co-resident tiles should share their detector boxes (backproject.hip: batch_impl)
        uint32_t yfast;
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
        float so_over_den;
        if(g.lean_div)
        {
            // d_sd / den and d_so / den share their denominator: one refined reciprocal serves both numerators, 13 instead of 22
            // instructions per column (ieee_lean.h). The host sets lean_div only after validate.hip has compared both quotients with
            // the compiler's IEEE divisions for every fp32 denominator this launch can produce (fill_params).
            const float r = paris_lean::refined_rcp(den);
            c.factor = paris_lean::div_with_rcp(g.d_sd, den, r);
            so_over_den = paris_lean::div_with_rcp(g.d_so, den, r);
        }
        else
        {
            c.factor = g.d_sd / den; // :125
            so_over_den = g.d_so / den;
        }
        const float b = (t * c.factor) - g.min_h;
        const float q = FD ? div_by_constant(b, g.l_px_x, g.rcp_l_px_x) : b / g.l_px_x;
        c.h = q - (1.f / 2.f);     // :45-50
        c.u = -so_over_den;        // :139
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

    // orders 14 .. 17: y tiles are dealt to the XCDs in groups of 1, 2, 4, 8; an XCD's share of the y tiles, whole groups
    __host__ __device__ inline uint32_t dealt_band(uint32_t nty, uint32_t order)
    {
        const uint32_t grp = 1u << (order - 14u);
        return grp * ((nty + 8u * grp - 1u) / (8u * grp));
    }

    // order 18, planes left over after the whole rounds of eight: their y tiles are dealt to the XCDs in groups of 8 for the fused
    // kernel (y tiles of a group fastest: co-resident tiles share their detector boxes) and in pairs for the single-projection one
    __host__ __device__ inline uint32_t order18_tail_group_log2(const BpParams& g)
    {
        return g.yfast == 1u ? 3u : 1u;
    }

    // 1-D grid size for the tile mapping in g.order (tile_of_block rejects the padding blocks)
    // (64-bit: the padded count of a band order can exceed the tile count by the band and chunk rounding; fill_params has checked
    // that the worst case over kernels and orders fits a 1-D grid)
    inline uint64_t grid_blocks(const BpParams& g)
    {
        const uint64_t total = static_cast<uint64_t>(g.ntx) * g.nty * g.ntz;
        if(g.order == 5u)
            return ((total + 7u) / 8u) * 8u;
#ifdef PARIS_HIP_EXPERIMENTS
        if(g.order == 8u || g.order == 9u)
            return 8ull * ((g.nty + 7u) / 8u) * g.ntx * g.ntz;
        if(g.order == 12u)
            return 8ull * ((g.nty + 7u) / 8u) * g.ntx * g.zchunk * ((g.ntz + g.zchunk - 1u) / g.zchunk);
#endif
        if(g.order == 18u) // z tiles dealt: 8 XCDs x whole planes x whole rounds, then the planes left over with their y tiles dealt
            return 8ull * (static_cast<uint64_t>(g.ntx) * g.nty * (g.ntz / 8u)
                           + static_cast<uint64_t>(g.ntz % 8u) * dealt_band(g.nty, 14u + order18_tail_group_log2(g)) * g.ntx);
        if(g.order >= 14u && g.order <= 17u)
            return 8ull * dealt_band(g.nty, g.order) * g.ntx * g.zchunk * ((g.ntz + g.zchunk - 1u) / g.zchunk);
        return total;
    }

    // Order 18 deals whole planes of tiles to the XCDs, one z tile each in turn; the ntz % 8 planes left over are shared by all XCDs
    // (tile_of_block). Volumes of fewer than 8 z tiles have nothing to deal: order 5. Launchers call this after they know their ntz.
    inline void settle_order(BpParams& g)
    {
        if(g.order == 18u && g.ntz < 8u)
            g.order = 5u;
    }

    // z tiles per chunk of the chunked orders, never more z tiles than the volume has (a shallow slab would otherwise launch up
    // to 32 times as many workgroups as it has tiles, all but a few leaving at once: ADVICE r02). Order 12 (a y BAND per XCD)
    // wants chunks of 256 slices: the detector band of a chunk then stays in the XCD's L2. The dealt orders 14 .. 17 (all XCDs
    // on adjacent y tiles) want shallow chunks: 2048^3 with order 15, same device, single-projection kernel 0.7673 / 0.7706 /
    // 0.7745 / 0.7726 of the HBM peak for chunks of 256 / 128 / 64 / 32 slices, fused kernel 1699 / 1719 / 1726 / 1740 GVox/s
    // (profiles/r03_ab_tile_order.txt): `deep` slices for them (64 for the tile kernel, 32 = one tile for the fused kernel).
    inline uint32_t chunk_tiles(uint32_t order, uint32_t tz, uint32_t ntz, uint32_t deep)
    {
        const uint32_t slices = order >= 14u ? deep : 256u;
        return std::min(std::max(1u, slices / std::max(1u, tz)), std::max(1u, ntz));
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
    //   8: XCD k owns a band of y tiles; x fastest, then z, then y inside the band
    //   9: XCD k owns a band of y tiles; x fastest, then y, then z inside the band
    //  12: order 8 chunk by chunk of 256 slices (deep volumes)
    //  18: z tiles dealt round-robin to the XCDs (XCD k owns z tiles k, k + 8, ...), x fastest, then y
    //  14: order 12 with y tiles dealt round-robin to the XCDs (XCD k owns y tiles k, k + 8, ...); 15, 16, 17: dealt in groups of 2, 4, 8
    __device__ __forceinline__ bool tile_of_block(const BpParams& g, uint32_t b, uint32_t& bx, uint32_t& by, uint32_t& bz)
    {
        const uint32_t total = g.ntx * g.nty * g.ntz;
#ifdef PARIS_HIP_EXPERIMENTS // orders 1, 8 (and 9, 12 below): the mappings that lost; the product build has 5 and 14 .. 18
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
#endif
        if(g.order == 18u)
        {
            // z tiles dealt to the XCDs: XCD k owns z tiles k, k + 8, ...; x runs fastest, then y (whole planes of tiles). The default
            // for planes up to 1024^2 since round 3: order 5 gives every XCD a contiguous eighth of the depth there, so the slices
            // above / below the cone -- skipped -- all belong to the first and the last XCD, which then finish early. 1024^3, three
            // interleaved rounds on one device: 0.7458 -> 0.7529 of the HBM peak, fused kernel 1799 -> 1825 GVox/s (groups of 2 or
            // 4 z tiles: slower; profiles/r03_ab_zdeal.txt).
            // A z tile count that does not divide among the eight (natural volumes: 1029 slices = 33 tiles of 32) leaves ntz % 8 planes
            // over. Dealt as planes they would give some XCDs a plane more than the others (33 tiles: 5 against 4); run by order 5
            // instead -- rounds 1-3 -- every XCD sweeps a contiguous eighth of the depth and the fused kernel loses 13 % (1024 x 1024 x
            // 1029 against 1024^3, profiles/r04_ab_order18_tail.txt). So the whole rounds of eight planes are dealt as before and the
            // planes left over are shared by all XCDs, their y tiles dealt in groups (the dealt orders' mapping, below).
            const uint32_t xcd = b % 8u;
            uint32_t r = b / 8u;
            const uint32_t full = g.ntz / 8u, plane = g.ntx * g.nty;
            if(r >= full * plane)
            {
                r -= full * plane;
                const uint32_t lg = order18_tail_group_log2(g), grp = 1u << lg;
                const uint32_t band = dealt_band(g.nty, 14u + lg);
                uint32_t yb;
                if(g.yfast == 1u)
                {
                    const uint32_t yl = r % grp;
                    r /= grp;
                    bx = r % g.ntx;
                    r /= g.ntx;
                    const uint32_t ngrp = band / grp;
                    yb = (r % ngrp) * grp + yl;
                    r /= ngrp;
                }
                else
                {
                    bx = r % g.ntx;
                    r /= g.ntx;
                    yb = r % band;
                    r /= band;
                }
                bz = full * 8u + r;
                by = (yb / grp) * (8u * grp) + xcd * grp + yb % grp;
                return bz < g.ntz && by < g.nty;
            }
            if(g.yfast == 1u) // (fused kernel: y fastest, see the dealt orders below)
            {
                by = r % g.nty;
                r /= g.nty;
                bx = r % g.ntx;
                bz = (r / g.ntx) * 8u + xcd;
            }
            else
            {
                bx = r % g.ntx;
                r /= g.ntx;
                by = r % g.nty;
                bz = (r / g.nty) * 8u + xcd;
            }
            return bz < g.ntz;
        }
        if(g.order >= 14u && g.order <= 17u)
        {
            // order 12 with the y tiles DEALT to the XCDs instead of banded. Order 14: XCD k owns y tiles k, k + 8, k + 16, ...;
            // order 15 deals pairs (XCD k owns y tiles 2k, 2k + 1, 2k + 16, 2k + 17, ...), 16 and 17 groups of 4 and 8. x runs
            // fastest, then the z tile inside the chunk, then the XCD's next y tile, then the chunk. Every XCD sees an even sample
            // of the plane, so tiles no ray reaches (the grid's corners on the source side, whole y bands at some angles) thin out
            // every XCD's share alike -- with banded orders the XCDs of the central bands, which have nothing to skip, finish last
            // and the launch waits for them (2048^3, same device: 0.748 -> 0.763-0.766 of the HBM peak for the single-projection
            // kernel, 1.63 -> 1.70-1.72 TVox/s for the fused one, profiles/r03_ab_tile_order.txt). The eight XCDs work on adjacent
            // y tiles at any time: a compact window. Pairs keep the banded order's rate where nothing is skipped (0.719 against
            // 0.711 for single tiles) at the same rate with the skip.
            const uint32_t band = dealt_band(g.nty, g.order);
            const uint32_t grp = 1u << (g.order - 14u);
            const uint32_t zchunk = g.zchunk;
            const uint32_t xcd = b % 8u;
            uint32_t r = b / 8u;
            uint32_t zl, yb;
            if(g.yfast == 2u) // x fastest, then the y tiles of a group, then the z tile inside the chunk, then the XCD's next group, then the chunk
            {
                bx = r % g.ntx;
                r /= g.ntx;
                const uint32_t yl = r % grp;
                r /= grp;
                zl = r % zchunk;
                r /= zchunk;
                const uint32_t ngrp = band / grp;
                yb = (r % ngrp) * grp + yl;
                bz = (r / ngrp) * zchunk + zl;
            }
            else if(g.yfast == 1u) // the y tiles of a group fastest, then x, then the XCD's next group, then the z tile inside the chunk, then the chunk
            {
                const uint32_t yl = r % grp;
                r /= grp;
                bx = r % g.ntx;
                r /= g.ntx;
                const uint32_t ngrp = band / grp;
                yb = (r % ngrp) * grp + yl;
                r /= ngrp;
                zl = r % zchunk;
                bz = (r / zchunk) * zchunk + zl;
            }
            else
            {
                bx = r % g.ntx;
                r /= g.ntx;
                zl = r % zchunk;
                r /= zchunk;
                yb = r % band;
                bz = (r / band) * zchunk + zl;
            }
            by = (yb / grp) * (8u * grp) + xcd * grp + yb % grp;
            return bz < g.ntz && by < g.nty;
        }
#ifdef PARIS_HIP_EXPERIMENTS
        if(g.order == 12u)
        {
            // order 8 applied to one chunk of zchunk z tiles after the other: XCD k owns the band k of y tiles; x runs fastest,
            // then the z tile inside the chunk, then y inside the band, then the chunk. A chunk of 256 slices reads a detector
            // band that stays in the XCD's L2 (the whole depth of a 2048^3 volume does not: every XCD would read every detector
            // row for each of its y tiles), so the deep volume streams like eight 256-slice slabs.
            const uint32_t band = (g.nty + 7u) / 8u;
            const uint32_t zchunk = g.zchunk;
            const uint32_t xcd = b % 8u;
            uint32_t r = b / 8u;
            bx = r % g.ntx;
            r /= g.ntx;
            const uint32_t zl = r % zchunk;
            r /= zchunk;
            const uint32_t yb = r % band;
            bz = (r / band) * zchunk + zl;
            by = xcd * band + yb;
            return bz < g.ntz && by < g.nty;
        }
        if(g.order == 9u)
        {
            // XCD k owns the contiguous band k of y tiles; inside the band x runs fastest, then y, then the z tile: all eight
            // XCDs work on the same slices at any time (one eighth of each slice each), so the chip touches tz slices at once
            // instead of 8 x tz
            const uint32_t band = (g.nty + 7u) / 8u;
            const uint32_t xcd = b % 8u;
            uint32_t r = b / 8u;
            bx = r % g.ntx;
            r /= g.ntx;
            const uint32_t yb = r % band;
            bz = r / band;
            by = xcd * band + yb;
            return bz < g.ntz && by < g.nty;
        }
#endif
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
        // 4-pixel staging of rows of at most 64 groups: rows one wave-instruction covers (64 / groups) and the multiplier that turns
        // lane / groups into a multiply and a shift (ceil(2^16 / groups): exact for lane < 64); 0 for wider rows
        int rpp, magic;
        // row bounds of the box as floats for make_column's tests (workgroup-uniform): a column's valid taps are inside the box iff
        // v_min >= row_lo and v_max < row_hi, where the bound is infinite on a side on which the box reaches the detector's edge
        float row_lo, row_hi;
        float first_row, end_row; // by0 and by0 + bhs - 1 as floats: every tap (valid or not) inside iff first_row <= v < end_row
    };

    // Detector bounding box of the voxel tile [k0,k1] x [l0,l1] x [m0,m1]. h is a projective function of (x,y) and
    // v of (z, factor), so the extremes sit on tile corners; the four corners are evaluated by four lanes in
    // parallel and min/max-reduced with two butterfly shuffles (every wave computes the same box). The box is
    // widened by the taps' reach plus one pixel of rounding slack, clipped to the detector and cut to the LDS
    // budget; a tap that still falls outside is served from global memory, so this only has to be right for speed.
    // prefer_stride (floats, a multiple of 4 and an odd one; 0: none): the stride to use whenever the box is no wider and all its
    // rows still fit -- a kernel that knows the stride at compile time addresses both rows of a tap from one register
    // (ds_read2_b32 with the row stride in its second offset, voxel_contribution<..., CS>).
    // the four extents of the box as integers, identical in the four lanes of a group of lanes 4i .. 4i + 3 (which share sin_phi and
    // cos_phi); different groups may evaluate different projections
    struct BoxExtents
    {
        int hmin, hmax, vmin, vmax;
    };

    __device__ __forceinline__ BoxExtents tile_box_extents(const BpParams& g, float sin_phi, float cos_phi, uint32_t k0, uint32_t k1,
                                                           uint32_t l0, uint32_t l1, uint32_t m0, uint32_t m1, uint32_t lane)
    {
        const uint32_t ci = lane & 3u;
        // The box only has to be right for speed (a tap outside it takes the global path, and whether a column may skip the
        // per-tap check is decided from the exact coordinates in make_column), so the corner rays are evaluated with the
        // hardware reciprocal instead of three IEEE divisions: ~2e-7 relative, far inside the one pixel of slack below.
        const float x_k = g.x_base + static_cast<float>(g.k_off + ((ci & 1u) ? k1 : k0)) * g.l_vx_x;
        const float y_l = g.y_base + static_cast<float>(g.l_off + ((ci & 2u) ? l1 : l0)) * g.l_vx_y;
        const float s = x_k * cos_phi + y_l * sin_phi;
        const float t = -x_k * sin_phi + y_l * cos_phi;
        const float factor = g.d_sd * __builtin_amdgcn_rcpf(s + g.d_so);
        const float h = ((t * factor) - g.min_h) * g.rcp_l_px_x - (1.f / 2.f);
        float hmin = h, hmax = h, fmin = factor, fmax = factor;
#pragma unroll
        for(int m = 1; m <= 2; m <<= 1)
        {
            hmin = fminf(hmin, __shfl_xor(hmin, m));
            hmax = fmaxf(hmax, __shfl_xor(hmax, m));
            fmin = fminf(fmin, __shfl_xor(fmin, m));
            fmax = fmaxf(fmax, __shfl_xor(fmax, m));
        }
        const float z_c = g.z_base + static_cast<float>(g.m_off + ((ci & 1u) ? m1 : m0)) * g.l_vx_z;
        const float v_c = ((z_c * ((ci & 2u) ? fmax : fmin)) - g.min_v) * g.rcp_l_px_y - (1.f / 2.f);
        float vmin = v_c, vmax = v_c;
#pragma unroll
        for(int m = 1; m <= 2; m <<= 1)
        {
            vmin = fminf(vmin, __shfl_xor(vmin, m));
            vmax = fmaxf(vmax, __shfl_xor(vmax, m));
        }
        BoxExtents e;
        e.hmin = to_int_clamped(hmin);
        e.hmax = to_int_clamped(hmax);
        e.vmin = to_int_clamped(vmin);
        e.vmax = to_int_clamped(vmax);
        return e;
    }

    // the extents widened by the taps' reach plus one pixel of rounding slack, clipped to the detector and cut to the LDS budget
    // (integer arithmetic only: on scalar registers when the extents are uniform, per lane otherwise)
    __device__ __forceinline__ Box box_of_extents(const BpParams& g, int ihmin, int ihmax, int ivmin, int ivmax, uint32_t box_floats,
                                                  int prefer_stride)
    {
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
            // multiple of 4 (16-byte aligned rows for ds_write_b128) and an ODD multiple: rows r, r + 1, ... r + 7 then start in
            // eight different groups of four banks, so lanes of one wave that tap different detector rows (the wave spans several
            // volume rows) do not pile onto the same banks as they do with a stride of 0, 8 or 16 mod 32
            b.stride = b.bw + 4;
            if(((b.stride >> 2) & 1) == 0)
                b.stride += 4;
            if(prefer_stride != 0 && b.stride <= prefer_stride && bh <= static_cast<int>(box_floats) / prefer_stride)
                b.stride = prefer_stride;
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
        b.first_row = static_cast<float>(b.by0);
        b.end_row = static_cast<float>(b.by0 + b.bhs - 1);
        b.row_lo = b.by0 == 0 ? -INFINITY : b.first_row;
        b.row_hi = (b.by0 + b.bhs >= static_cast<int>(g.p_dim_y)) ? INFINITY : b.end_row;
        const int n4 = b.bw >> 2;
        b.rpp = (n4 >= 1 && n4 <= 64) ? 64 / n4 : 0;
        b.magic = (n4 >= 1 && n4 <= 64) ? (65536 + n4 - 1) / n4 : 0;
        return b;
    }

    __device__ __forceinline__ Box tile_box(const BpParams& g, uint32_t k0, uint32_t k1, uint32_t l0, uint32_t l1,
                                            uint32_t m0, uint32_t m1, uint32_t lane, uint32_t box_floats, int prefer_stride = 0)
    {
        const BoxExtents e = tile_box_extents(g, g.sin_phi, g.cos_phi, k0, k1, l0, l1, m0, m1, lane);
        // identical in every lane by construction; readfirstlane moves them to scalar registers
        return box_of_extents(g, __builtin_amdgcn_readfirstlane(e.hmin), __builtin_amdgcn_readfirstlane(e.hmax),
                              __builtin_amdgcn_readfirstlane(e.vmin), __builtin_amdgcn_readfirstlane(e.vmax), box_floats, prefer_stride);
    }

    // The boxes of up to 16 projections at once (fused kernel): lanes 4i .. 4i + 3 evaluate the four corner rays of projection
    // first + i, lane 4i writes that box to tab[first + i] (BOX_WORDS words each). The per-projection cost of the box then is one
    // LDS read and nine v_readlane instead of the whole evaluation repeated by every lane for every projection.
    constexpr int BOX_WORDS = 12;

    __device__ __forceinline__ void tile_boxes_to_lds(const BpParams& g, const float* sin_tab, const float* cos_tab, uint32_t first,
                                                      uint32_t n_proj, uint32_t k0, uint32_t k1, uint32_t l0, uint32_t l1, uint32_t m0,
                                                      uint32_t m1, uint32_t lane, uint32_t box_floats, int prefer_stride, int* tab)
    {
        // sin_tab / cos_tab are kernel arguments: indexed uniformly and handed to the owning lanes by a select
        float sin_phi = 0.f, cos_phi = 1.f;
#pragma unroll
        for(uint32_t i = 0; i < 16u; ++i)
        {
            const uint32_t p = min(first + i, n_proj - 1u);
            const bool mine = (lane >> 2) == i;
            sin_phi = mine ? sin_tab[p] : sin_phi;
            cos_phi = mine ? cos_tab[p] : cos_phi;
        }
        const BoxExtents e = tile_box_extents(g, sin_phi, cos_phi, k0, k1, l0, l1, m0, m1, lane);
        const Box b = box_of_extents(g, e.hmin, e.hmax, e.vmin, e.vmax, box_floats, prefer_stride);
        const uint32_t p = first + (lane >> 2);
        if((lane & 3u) == 0u && p < n_proj)
        {
            int* w = tab + p * BOX_WORDS;
            w[0] = b.bx0;
            w[1] = b.by0;
            w[2] = b.bw;
            w[3] = b.bhs;
            w[4] = b.stride;
            w[5] = __float_as_int(b.row_lo);
            w[6] = __float_as_int(b.row_hi);
            w[7] = __float_as_int(b.first_row);
            w[8] = __float_as_int(b.end_row);
            w[9] = b.rpp;
            w[10] = b.magic;
        }
    }

    // box p of the table, into scalar registers: lanes 0 .. 8 read one word each, nine v_readlane spread them
    __device__ __forceinline__ Box box_from_lds(const int* tab, uint32_t p, uint32_t lane)
    {
        const int w = tab[p * BOX_WORDS + min(lane, static_cast<uint32_t>(BOX_WORDS - 1))];
        Box b;
        b.bx0 = __builtin_amdgcn_readlane(w, 0);
        b.by0 = __builtin_amdgcn_readlane(w, 1);
        b.bw = __builtin_amdgcn_readlane(w, 2);
        b.bhs = __builtin_amdgcn_readlane(w, 3);
        b.stride = __builtin_amdgcn_readlane(w, 4);
        b.row_lo = __int_as_float(__builtin_amdgcn_readlane(w, 5));
        b.row_hi = __int_as_float(__builtin_amdgcn_readlane(w, 6));
        b.first_row = __int_as_float(__builtin_amdgcn_readlane(w, 7));
        b.end_row = __int_as_float(__builtin_amdgcn_readlane(w, 8));
        b.rpp = __builtin_amdgcn_readlane(w, 9);
        b.magic = __builtin_amdgcn_readlane(w, 10);
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
            const uint32_t rows_per_pass = n4 <= 64u ? static_cast<uint32_t>(b.rpp) : 1u; // rows one wave-instruction covers
            const uint32_t lr = n4 <= 64u ? (lane * static_cast<uint32_t>(b.magic)) >> 16 : 0u; // lane / n4
            const uint32_t lc = n4 <= 64u ? lane - lr * n4 : lane;
            if(lr >= rows_per_pass)
                return; // lanes beyond the last whole row of the pass idle
            if(n4 <= 64u)
            {
                // one group per lane and row: both addresses advance by a uniform step, two rows in flight per lane
                const uint32_t step = n_waves * rows_per_pass;
                uint32_t r = wave * rows_per_pass + lr;
                const size_t first = (static_cast<size_t>(b.by0) + r) * g.p_pitch + static_cast<size_t>(b.bx0) + 4u * lc;
                float* dst = lds_box + r * static_cast<uint32_t>(b.stride) + 4u * lc;
                const size_t src_step = static_cast<size_t>(step) * g.p_pitch;
                const uint32_t dst_step = step * static_cast<uint32_t>(b.stride);
                const uint32_t rows = static_cast<uint32_t>(b.bhs);
                if(g.proj_f16)
                {
                    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                    auto widen = [](half4 h) { return make_float4(static_cast<float>(h.x), static_cast<float>(h.y), static_cast<float>(h.z), static_cast<float>(h.w)); };
                    const _Float16* src = static_cast<const _Float16*>(g.proj) + first;
                    for(; r + step < rows; r += 2u * step)
                    {
                        const half4 h0 = *reinterpret_cast<const half4*>(src);
                        const half4 h1 = *reinterpret_cast<const half4*>(src + src_step);
                        *reinterpret_cast<float4*>(dst) = widen(h0);
                        *reinterpret_cast<float4*>(dst + dst_step) = widen(h1);
                        src += 2u * src_step;
                        dst += 2u * dst_step;
                    }
                    if(r < rows)
                        *reinterpret_cast<float4*>(dst) = widen(*reinterpret_cast<const half4*>(src));
                    return;
                }
                const float* src = static_cast<const float*>(g.proj) + first;
                for(; r + step < rows; r += 2u * step)
                {
                    const float4 v0 = *reinterpret_cast<const float4*>(src);
                    const float4 v1 = *reinterpret_cast<const float4*>(src + src_step);
                    *reinterpret_cast<float4*>(dst) = v0;
                    *reinterpret_cast<float4*>(dst + dst_step) = v1;
                    src += 2u * src_step;
                    dst += 2u * dst_step;
                }
                if(r < rows)
                    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
                return;
            }
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
        bool inside;    // stronger: EVERY tap of this column over the tile's z range is valid and inside the staged box
        bool none;      // NO tap of this column over the tile's z range is valid and u is finite: every contribution is exactly +0
    };

    // z_first / z_last: centred z of the tile's first and last slice. The per-slice coordinate v is a monotone
    // function of the slice index for a fixed column (every step of its evaluation -- multiply by factor, subtract,
    // divide by the pitch, subtract 0.5, each rounded to nearest -- is monotone), so the rows touched over the tile
    // lie between the rows touched at its two end slices: `fast` is decided from those two evaluations alone, with
    // the very expression the slice loop uses, and needs no error bound.
    template <bool FD>
    __device__ __forceinline__ Column make_column_from(const BpParams& g, const Box& b, const ColConst& c, float z_first, float z_last);

    template <bool FD>
    __device__ __forceinline__ Column make_column(const BpParams& g, const Box& b, uint32_t K, uint32_t L, float z_first,
                                                  float z_last)
    {
        return make_column_from<FD>(g, b, column_constants<FD>(g, K, L), z_first, z_last);
    }

    // the same from column constants computed elsewhere (two-pass variant: bp_column_state_kernel wrote them for every column
    // of the plane, with column_constants<FD> itself)
    template <bool FD>
    __device__ __forceinline__ Column make_column_from(const BpParams& g, const Box& b, const ColConst& c, float z_first, float z_last)
    {
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
        // Valid taps have their upper row in [0, dim_y - 2]. With r_lo = max(floor(v_min), 0) and r_hi = min(floor(v_max), dim_y - 2) the
        // column's valid rows are inside the box iff r_lo > r_hi (none) or by0 <= r_lo and r_hi <= by0 + bhs - 2. In floats, for
        // integers n: floor(v) >= n <=> v >= n and floor(v) <= n <=> v < n + 1, so
        //   none      <=> v_max < 0 or v_min >= dim_y - 1
        //   r_lo >= by0           <=> v_min >= by0 or by0 == 0            (= v_min >= row_lo)
        //   r_hi <= by0 + bhs - 2 <=> v_max < by0 + bhs - 1 or the box reaches the last row (= v_max < row_hi)
        // (infinite coordinates compare like the saturating conversions did; NaN is excluded by `ordered`)
        const float v_min = fminf(va, vb), v_max = fmaxf(va, vb);
        const bool rows_inside = (v_max < 0.f) || (v_min >= g.p_dim_y_f - 1.f) || (v_min >= b.row_lo && v_max < b.row_hi);
        const bool finite_factor = (c.factor - c.factor) == 0.f; // with a finite factor v is never NaN (it may overflow to inf)
        col.fast = !x_valid || (ordered && finite_factor && col.xoff >= 0 && rows_inside);
        // All taps valid: the unclamped rows of both end slices lie in [by0, by0 + bhs - 2] (the box is clipped to the detector, so
        // a row pair inside it is a valid pair). floor(v) >= by0 <=> v >= by0 and floor(v) <= by0 + bhs - 2 <=> v < by0 + bhs - 1
        // for the integers by0, bhs; a NaN fails both comparisons. By the monotonicity argument above every slice in between is
        // inside too: such a column needs neither the per-voxel validity test nor the row clamp nor the final select.
        col.inside = x_valid && finite_factor && col.xoff >= 0 && b.bhs >= 2 && v_min >= b.first_row && v_max < b.end_row;
        // No valid tap at all -- the column's x taps are off the detector, or (by the same monotonicity) its rows are above or below
        // it over the whole tile: :71 then sets det = 0 and :140 adds 0.5 * 0 * u * u, which is +0 for every finite u. A wave of such
        // columns has nothing to add (and, where the volume is known to hold no -0, nothing to load or store: BpParams::skip_invalid).
        const bool finite_u = (c.u - c.u) == 0.f;
        col.none = finite_u && (!x_valid || (ordered && ((v_max < 0.f) || (v_min >= g.p_dim_y_f - 1.f))));
        return col;
    }

    // one voxel-update: src/openmp/backprojection.cpp:130-140 for slice coordinate z_m of column col
    // FAST: the column was proven to stay inside the staged box (Column::fast), so the global-memory path and its
    // branch are compiled out and the body is straight-line code the scheduler can overlap across voxels.
    // ALLVALID (implies FAST): the column was proven to have every tap valid and inside the box (Column::inside)
    // CS: the LDS row stride in floats when the kernel fixed it at compile time (b.stride == CS), else 0
    template <bool FD, bool FAST, bool ALLVALID = false, int CS = 0>
    __device__ __forceinline__ float voxel_contribution(const BpParams& g, const Box& b, const float* lds_box, float z_m,
                                                        const Column& col)
    {
        static_assert(!ALLVALID || FAST, "ALLVALID is a refinement of FAST");
        static_assert(CS >= 0 && CS < 255, "both rows of a tap must be within reach of one ds_read2_b32");
        const float v = v_coordinate<FD>(g, z_m, col.factor);
        float y1 = 0.f, y2 = 0.f;
        int y1i;
        if(FAST && PARIS_BP_SINGLE_INSTRUCTION_FLOOR)
        {
            // floor(v) as an integer in one instruction; its float form is not needed on this path (the row weights come from
            // v_fract below). tools/flr_probe.hip checks both single-instruction forms against floorf for every fp32 value.
            asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(y1i) : "v"(v));
        }
        else
        {
            y1 = floorf(v);
            y2 = y1 + 1.f;
            y1i = static_cast<int>(y1);
        }
        const int rrel = y1i - b.by0;
        const int bhs_m2 = b.bhs - 2;
        bool valid = true;
        if(ALLVALID)
        {
        }
        else if(FAST)
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
        int rc = rrel; // ALLVALID: 0 <= rrel <= bhs - 2 by construction
        if(!ALLVALID)
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(rc) : "v"(rrel), "v"(max(bhs_m2, 0)));
        // LDS byte address of the upper left tap = row * stride4 + (box base + 4 * column): the second term is z-invariant
        // (hoisted with the column state), so a tap pair costs one 24-bit multiply-add (rc < 2^12, stride4 < 2^16;
        // v_mul_lo_u32 is quarter rate) and the row below one add. Integer addresses keep the compiler from adding the
        // (zero) link-time base of the dynamic LDS array to every access.
        using lds_cptr = const __attribute__((address_space(3))) float*;
        const int stride4 = (CS != 0 ? CS : b.stride) << 2;
        const uint32_t xaddr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_box)) + (static_cast<uint32_t>(max(col.xoff, 0)) << 2);
        // ALLVALID: no clamp, so the box's first row folds into the z-invariant term as well (wrapping 32-bit arithmetic; the
        // kernel takes this path only for detectors of fewer than 2^23 rows, the reach of the 24-bit multiply)
        uint32_t a1;
        if(ALLVALID)
        {
            uint32_t xrel = xaddr - static_cast<uint32_t>(__mul24(b.by0, stride4));
            asm("" : "+v"(xrel)); // opaque: keeps the compiler from refactoring into (y1i - by0) * stride4, one more instruction per tap
            a1 = static_cast<uint32_t>(__mul24(y1i, stride4)) + xrel;
        }
        else
            a1 = static_cast<uint32_t>(__mul24(rc, stride4)) + xaddr;
        lds_cptr r1 = reinterpret_cast<lds_cptr>(a1);
        lds_cptr r2 = CS != 0 ? r1 + CS : reinterpret_cast<lds_cptr>(a1 + static_cast<uint32_t>(stride4));
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
        // v_fract_f32(v) == v - floor(v) bit for bit for 0 <= v < 2^24, i.e. for every valid tap (exhaustive: tools/flr_probe.hip)
        const float wy1 = (FAST && PARIS_BP_SINGLE_INSTRUCTION_FLOOR) ? __builtin_amdgcn_fractf(v) : v - y1;
        const float wy2 = FAST ? 1.f - wy1 : y2 - v;
        float det = wy2 * interp_y1 + wy1 * interp_y2;
        if(!ALLVALID)
            det = valid ? det : 0.f;     // :71
        return 0.5f * det * col.u * col.u; // :140
    }

    // voxel_contribution<FD, true, ALLVALID, CS> in two halves, for a slice loop that issues the LDS reads of the next slice before it
    // finishes the current one (the fused kernel: a wave otherwise sits out one LDS round trip per voxel-update). Same
    // operations on the same values in the same order per voxel; only the interleaving across voxels differs.
    struct Tap
    {
        float q11, q21, q12, q22; // the four detector pixels
        float wy1;                // row weight of the lower pair
        bool valid;
    };

    template <bool FD, bool ALLVALID, int CS>
    __device__ __forceinline__ Tap fetch_tap(const BpParams& g, const Box& b, const float* lds_box, float z_m, const Column& col)
    {
        Tap t;
        const float v = v_coordinate<FD>(g, z_m, col.factor);
        int y1i;
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(y1i) : "v"(v));
        t.wy1 = __builtin_amdgcn_fractf(v);
        using lds_cptr = const __attribute__((address_space(3))) float*;
        const int stride4 = (CS != 0 ? CS : b.stride) << 2;
        const uint32_t xaddr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_box)) + (static_cast<uint32_t>(max(col.xoff, 0)) << 2);
        uint32_t a1;
        if(ALLVALID)
        {
            t.valid = true;
            uint32_t xrel = xaddr - static_cast<uint32_t>(__mul24(b.by0, stride4));
            asm("" : "+v"(xrel));
            a1 = static_cast<uint32_t>(__mul24(y1i, stride4)) + xrel;
        }
        else
        {
            const int rrel = y1i - b.by0;
            const unsigned rowlim = col.ymax > 0.f ? static_cast<unsigned>(max(b.bhs - 1, 0)) : 0u;
            t.valid = static_cast<unsigned>(rrel) < rowlim;
            int rc;
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(rc) : "v"(rrel), "v"(max(b.bhs - 2, 0)));
            a1 = static_cast<uint32_t>(__mul24(rc, stride4)) + xaddr;
        }
        lds_cptr r1 = reinterpret_cast<lds_cptr>(a1);
        lds_cptr r2 = CS != 0 ? r1 + CS : reinterpret_cast<lds_cptr>(a1 + static_cast<uint32_t>(stride4));
        t.q11 = r1[0];
        t.q21 = r1[1];
        t.q12 = r2[0];
        t.q22 = r2[1];
        return t;
    }

    template <bool ALLVALID>
    __device__ __forceinline__ float finish_tap(const Column& col, const Tap& t)
    {
        const float interp_y1 = col.wx2 * t.q11 + col.wx1 * t.q21; // :77
        const float interp_y2 = col.wx2 * t.q12 + col.wx1 * t.q22; // :78
        const float wy2 = 1.f - t.wy1;
        float det = wy2 * interp_y1 + t.wy1 * interp_y2;
        if(!ALLVALID)
            det = t.valid ? det : 0.f;     // :71
        return 0.5f * det * col.u * col.u; // :140
    }

    // --------------------------------------------------------------------------------------------
    // Fused kernel (extension: paris_hip_backproject_batch). One launch adds n_proj projections: a lane keeps its
    // 4 x TZ voxels in registers, and for every projection in turn the workgroup stages that projection's box,
    // rebuilds the column state and adds the TZ contributions -- in projection order, so every voxel sees exactly
    // the additions, in exactly the order, of n_proj single launches (bit-identical), while the volume is read
    // and written once per batch: 8 / n_proj bytes per voxel-update. With the HBM term gone the kernel is bound by
    // vector ALU issue (about 40 instructions per voxel-update plus the per-projection column setup).
    // --------------------------------------------------------------------------------------------
    constexpr int FUSED_MAX = 64;

    struct FusedParams
    {
        BpParams g;           // proj = first projection; sin/cos overwritten per projection
        uint32_t n_proj;
        float sin_phi[FUSED_MAX];
        float cos_phi[FUSED_MAX];
        // where each projection of the launch lives: a stack one stride apart (paris_hip_backproject_batch, the deferral ring's
        // snapshots) or the callers' own buffers (deferral by reference: no snapshot was taken)
        const void* proj_tab[FUSED_MAX];
    };
}

// backproject_fused.hip: launches bp_fused_kernel<tz, nt, fd> for *fused_params (a FusedParams; passed untyped because the
// type has internal linkage in each translation unit -- both see the one definition above). vx (voxels per lane along x) is 2 or 4, tz 8 or 16.
void paris_hip_bp_launch_fused(const void* fused_params, int vx, int tz, bool nt, bool fd, hipStream_t stream);

#endif
