// Voxel-driven cone-beam backprojection for gfx950 (MI355X).
//
// Replaces paris::openmp::backproject (src/openmp/backprojection.cpp:86-199) and paris::cuda::backproject
// (src/cuda/backprojection.cu:64-243) behind paris_hip_backproject (include/paris_hip.h).
//
// Numerics: every fp32 operation is the reference's, in the reference's order, rounded once (IEEE divide,
// no FMA contraction), so the volume is bit-identical to the OpenMP backend's.
//
// Mapping (the measurements behind it: profiles/HISTORY.md 4.1):
//   - a 256-thread workgroup owns a tile of 64 (x) x TY (y) voxel columns and walks TZ slices in z;
//   - lanes of a wave cover x contiguously (VX voxels per lane, 16 B / 8 B / 4 B accesses), so every
//     volume load/store instruction touches whole 256-byte runs of the x-fastest volume;
//   - the detector footprint of the tile (a bounding box computed from the tile's corner rays) is staged
//     into LDS once per tile; the four bilinear taps are LDS reads. A tap that falls outside the staged
//     box (never for sane geometries; possible when the box is larger than the LDS budget) is fetched
//     from global memory instead, so the result does not depend on the box being right;
//   - per (x,y) column s, t, factor, h, u and the x-interpolation weights are z-invariant and kept in
//     registers; per z step only v is recomputed (src/openmp/backprojection.cpp:130-133).
#include "bp_device.h"

#include <mutex>
#include <utility>

namespace
{
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

    // --------------------------------------------------------------------------------------------
    // Tile kernel (z-walk in time). 1-D grid over ntx * nty * ntz tiles of 64 x TY x g.tz voxels, TY = 4*VX.
    // Serves every alignment (VX = 1, 2, 4).
    // --------------------------------------------------------------------------------------------
#ifndef PARIS_TILE_MIN_WAVES
#define PARIS_TILE_MIN_WAVES 1
#endif
    template <int VX, int UNROLL, bool NT, bool FD, bool PRE = false>
    __global__ void __launch_bounds__(256, PARIS_TILE_MIN_WAVES) bp_tile_kernel(const BpParams g)
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

        // ---- per-lane columns --------------------------------------------------------------------
        const uint32_t xq = lane % XL;
        const uint32_t yy = lane / XL;
        const uint32_t k = k0 + xq * VX;
        const uint32_t l = l0 + wave * RW + yy;
        const bool active = k < g.v_dim_x && l < g.v_dim_y;

        // two-pass variant: this lane's column constants were written by bp_column_state_kernel; requested before the box so
        // that both travel together (VX == 4, dim_x % 4 == 0: one 16-byte load per plane)
        float4 pre_f = make_float4(0.f, 0.f, 0.f, 0.f), pre_h = pre_f, pre_u = pre_f;
        if(PRE && VX == 4)
        {
            const size_t plane = static_cast<size_t>(g.v_dim_x) * g.v_dim_y;
            const size_t at = static_cast<size_t>(min(l, l1)) * g.v_dim_x + min(k, g.v_dim_x - 4u);
            pre_f = *reinterpret_cast<const float4*>(g.colstate + at);
            pre_h = *reinterpret_cast<const float4*>(g.colstate + plane + at);
            pre_u = *reinterpret_cast<const float4*>(g.colstate + 2u * plane + at);
        }

        const Box box = tile_box(g, k0, k1, l0, l1, m0, m1, lane, g.lds_floats);
        stage_box(g, box, lds, wave, 4u, lane);

        const float z_first = g.z_base + static_cast<float>(g.m_off + m0) * g.l_vx_z;
        const float z_last = g.z_base + static_cast<float>(g.m_off + m1) * g.l_vx_z;
        Column col[VX];
        bool all_fast = true, all_none = true;
#pragma unroll
        for(int j = 0; j < VX; ++j)
        {
            if(PRE && VX == 4)
            {
                ColConst c;
                c.factor = elem<4>(pre_f, j);
                c.h = elem<4>(pre_h, j);
                c.u = elem<4>(pre_u, j);
                col[j] = make_column_from<FD>(g, box, c, z_first, z_last);
            }
            else
                col[j] = make_column<FD>(g, box, g.k_off + min(k + j, k1), g.l_off + min(l, l1), z_first, z_last);
            all_fast = all_fast && col[j].fast;
            all_none = all_none && col[j].none;
        }

        // (keeping the box loads in flight across this setup was measured slower: profiles/r01_ab_full_volume_6.jsonl)
        __syncthreads();
        if(!active)
            return;
        // no ray of this projection reaches any column of the wave (the corners of the grid outside the field of view, slices above /
        // below the cone on the source side): every contribution is +0 and the volume holds no -0, so the tile is left as it is --
        // about a tenth of a 2048^3 launch's traffic at the natural grid
        if(g.skip_invalid != 0u && __all(all_none ? 1 : 0) != 0)
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

#ifdef PARIS_HIP_EXPERIMENTS // kernels that were measured and lost (the two-pass variant's pass A, the slice kernel): experiments build only
#include "experiments/bp_kernels.inc"
#endif
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
        settle_order(g);
        g.zchunk = chunk_tiles(g.order, g.tz, g.ntz, 64u);
        const uint32_t blocks = static_cast<uint32_t>(grid_blocks(g));
        hipLaunchKernelGGL((bp_tile_kernel<VX, UNROLL, NT, FD>), dim3(blocks), dim3(256), g.lds_floats * sizeof(float), stream, g);
    }

    template <int VX, bool NT, bool FD>
    void launch_tile_unroll(BpParams& g, int unroll, hipStream_t stream)
    {
        switch(unroll) // the bodies the launcher picks by itself are 1 and 2; 3 and 4 exist in the experiments build (tuning)
        {
            case 1: launch_tile<VX, 1, NT, FD>(g, stream); break;
#ifdef PARIS_HIP_EXPERIMENTS
            case 3: launch_tile<VX, 3, NT, FD>(g, stream); break; // 1 + prefetch of the next slice
            case 4: launch_tile<VX, 4, NT, FD>(g, stream); break;
#endif
            default: launch_tile<VX, 2, NT, FD>(g, stream); break;
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

#ifdef PARIS_HIP_EXPERIMENTS // ... and their launchers
#include "experiments/bp_launchers.inc"
#endif
    // Is div_by_constant exact for this divisor? Checked once per process, device and divisor on the GPU (about 2 ms): the
    // result is a property of the divisor, so it is cached process-wide (every ctx of every host thread shares it) and the
    // check runs on a private blocking-free stream of its own -- never on the caller's stream, which may be capturing or may
    // hold queued work the caller does not want to wait for. paris_hip_stage_filter pre-validates the detector's pixel pitches
    // (they are known from det_geo long before the first backprojection), so the first paris_hip_backproject of a PARIS loop
    // finds the answer cached and stays asynchronous.
    std::mutex fastdiv_mutex;
    std::map<std::pair<int, uint32_t>, bool> fastdiv_cache; // (device, divisor bits) -> exhaustive check result

    void fastdiv_enqueue(hipStream_t s, unsigned long long* counter, const void* arg)
    {
        const float c = *static_cast<const float*>(arg);
        hipLaunchKernelGGL(fastdiv_validate_kernel, dim3(1u << 16), dim3(256), 0, s, c, 1.f / c, counter);
    }

    int fastdiv_is_exact(paris_hip_ctx* ctx, float c, bool* ok)
    {
        *ok = false;
        if(!(c > 0.f) || !(c < INFINITY))
            return PARIS_HIP_SUCCESS;
        uint32_t key;
        static_assert(sizeof(key) == sizeof(c), "fp32");
        std::memcpy(&key, &c, sizeof(key));
        auto it = ctx->fastdiv_exact.find(key); // per-ctx copy: no lock on the hot path
        if(it == ctx->fastdiv_exact.end())
        {
            std::lock_guard<std::mutex> lock(fastdiv_mutex);
            const auto pkey = std::make_pair(ctx->device, key);
            auto pit = fastdiv_cache.find(pkey);
            if(pit == fastdiv_cache.end())
            {
                bool exact = false, known = false;
                if(int rc = paris_hip_run_check(ctx, {2u, key, 0u, 0u}, fastdiv_enqueue, &c, &exact, &known))
                    return rc;
                if(!known) // asynchronous validation: the check is running; the IEEE division serves until it has answered
                    return PARIS_HIP_SUCCESS;
                pit = fastdiv_cache.emplace(pkey, exact).first;
            }
            it = ctx->fastdiv_exact.emplace(key, pit->second).first;
        }
        *ok = it->second;
        return PARIS_HIP_SUCCESS;
    }
}

// stages.cpp (paris_hip_stage_filter): validates the fast division for the detector's pixel pitches ahead of the first
// backprojection, so that call does not block
int paris_hip_prevalidate_fast_division(paris_hip_ctx* ctx, float l_px_row, float l_px_col)
{
    if(ctx == nullptr || ctx->bp_fastdiv == 0)
        return PARIS_HIP_SUCCESS;
    bool ok = false;
    if(int rc = fastdiv_is_exact(ctx, l_px_row, &ok))
        return rc;
    return fastdiv_is_exact(ctx, l_px_col, &ok);
}

// the same for the shared-reciprocal form of the two per-column divisions (validate.hip), whose operands d_sd, d_so the detector
// geometry fixes as well
int paris_hip_prevalidate_lean_division(paris_hip_ctx* ctx, float d_so, float d_od)
{
    if(ctx == nullptr || ctx->bp_lean_div == 0)
        return PARIS_HIP_SUCCESS;
    bool ok = false;
    return paris_hip_lean_division_check(ctx, std::fabs(d_so) + std::fabs(d_od), d_so, &ok);
}

// Cache policy of the volume stream: 0 plain, 1 nontemporal, 2 nontemporal loads + write-through nontemporal stores. Automatic
// (ctx->bp_nt < 0): a slab that (mostly) stays in the 256 MiB Infinity Cache between launches must not be pushed out to HBM by
// nontemporal / write-through accesses -- plain accesses are faster up to about 400 MiB (256^3: 27.3 against 32.8 us per launch,
// 384^3 = 216 MiB: 71.8 against 92.0, 448^3 = 343 MiB: 128 against 142; 512^3 = 512 MiB: 221 against 212:
// profiles/r02_cache_policy_by_size.txt); anything larger is touched once per launch and streams best with policy 2.
static int volume_stream_policy(const paris_hip_ctx* ctx, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z)
{
    if(ctx->bp_nt >= 0)
        return ctx->bp_nt;
    const uint64_t bytes = 4ull * v_dim_x * v_dim_y * v_dim_z;
    return bytes <= (384ull << 20) ? 0 : 2;
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
    if(int rc = paris_hip_flush_pending_weight(ctx)) // a weighting that no filter call picked up
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
    // Tile depth and workgroup -> tile order by volume shape (tools/tune_bp.py and interleaved A/B of bench.py steps,
    // profiles/r02_tune_bp_*.jsonl, profiles/r02_ab_tile_order.txt; bare access patterns: tools/membench.hip,
    // profiles/r02_membench.txt). The volume stream wants a compact, sliding set of concurrently touched addresses: with a y band
    // per XCD swept x -> z -> y (order 8) the set shrinks with the tile depth (bare: 6.03 / 6.25 / 6.35 / 6.5 TB/s at 16 / 8 / 4 / 2
    // slices, against 6.67 for a linear sweep and 5.9 for the XCD-contiguous order 5 at 16), while the column setup a tile pays
    // grows as 1 / depth and the detector rows an XCD reads grow with the depth it sweeps. Planes beyond 1024^2: rounds 1-2 ran order
    // 12 (order 8 in chunks of 256 slices, so the detector band of a chunk stays in the XCD's L2); since round 3 the y tiles are
    // DEALT to the XCDs in pairs (order 15, bp_device.h: tile_of_block) -- once tiles no ray reaches are skipped, a band per XCD
    // leaves the XCDs of the central bands with all the work: 0.748 -> 0.7745 of the HBM peak at 2048^3, 0.744 -> 0.756 on the
    // 256-slice slab, 0.734 -> 0.739 at config 5 where nothing is skipped (profiles/r03_ab_tile_order.txt) -- with 8-slice tiles for slabs up to 512
    // slices (2048 x 2048 x 256: 1.447 against 1.491 ms inside bench.py's step, +3 %), with 16-slice tiles for deeper volumes
    // (2048^3: +0.2 %; 8-slice tiles are +1.1 % on one device and -0.4 % on another there; measured again after the tile prologue
    // lost its integer divisions: +0.65 % on one device, -1.2 % on the next, profiles/r02_ab_tile_depth_full_volume.txt). Planes up to 1024^2: order 5 with 8-slice
    // tiles (+2.5 %; the band orders lose 3 %), up to 512^2 with 16-slice tiles.
    const uint64_t plane = static_cast<uint64_t>(v_dim_x) * v_dim_y;
    // (round 4: the 8-slice tiles of large planes end at 256 slices, not 512 -- with one slice in flight at four workgroups per CU the
    // 16-slice tiles win above that: 2048 x 2048 x 512 0.743-0.759 -> 0.759-0.763, x 352 +2 %, x 320 +0.6 %, x 288 +0.5 %; at 256 the two
    // are within +-0.3 % of each other by device and the slab keeps its 8-slice tiles; profiles/r04_ab_tile_occupancy.txt section 6)
    const uint32_t tz_auto = plane > (1ull << 20) ? (v_dim_z <= 256u ? 8u : TZ_DEFAULT) : (plane > (1ull << 18) ? 8u : TZ_DEFAULT);
    const uint32_t tz = ctx->bp_tz ? ctx->bp_tz : tz_auto;
    {
        // the 1-D grid must hold every tile of the narrowest, shallowest tiling any kernel uses (64 x 4 x min(tz, 8)), INCLUDING
        // the padding blocks of the band orders: y tiles rounded up to 8 bands, z tiles to whole chunks
        const uint32_t tz_min = std::min(tz, 8u);
        const uint64_t ntx = (v_dim_x + 63u) / 64u, nty = (v_dim_y + 3u) / 4u, ntz = (v_dim_z + tz_min - 1u) / tz_min;
        const uint64_t zchunk = chunk_tiles(12u, tz_min, static_cast<uint32_t>(ntz), 64u); // the deepest chunk pads most
        const uint64_t padded = std::max<uint64_t>(8ull * (8u * ((nty + 63u) / 64u)) * ntx * zchunk * ((ntz + zchunk - 1u) / zchunk), // widest deal: groups of 8
                                                   8ull * ntx * nty * (ntz / 8u) + 8ull * 7u * 8u * ((nty + 63u) / 64u) * ntx);       // order 18: whole rounds + at most 7 shared planes
        if(padded > 0x7fffff00ull)
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
    {
        // lean_div (bp_device.h: column_constants): every denominator s + d_so of this launch lies in [0.1, 1.9] x d_so -- the slab's
        // (x, y) extent stays within 0.9 d_so of the axis, |s| <= hypot(max |x|, max |y|) -- and d_so, d_sd are ordinary numbers
        const double x0 = g.x_base + static_cast<double>(g.k_off) * g.l_vx_x, x1 = g.x_base + (static_cast<double>(g.k_off) + v_dim_x - 1.0) * g.l_vx_x;
        const double y0 = g.y_base + static_cast<double>(g.l_off) * g.l_vx_y, y1 = g.y_base + (static_cast<double>(g.l_off) + v_dim_y - 1.0) * g.l_vx_y;
        const double reach = std::hypot(std::max(std::fabs(x0), std::fabs(x1)), std::max(std::fabs(y0), std::fabs(y1)));
        const double lo = std::ldexp(1.0, -40), hi = std::ldexp(1.0, 40);
        const bool ordinary = g.d_so >= lo && g.d_so <= hi && g.d_sd >= lo && g.d_sd <= hi;
        g.lean_div = (ctx->bp_lean_div != 0 && ordinary && std::isfinite(reach) && reach <= 0.9 * g.d_so) ? 1u : 0u;
        if(g.lean_div != 0u)
        {
            // ... and the device has compared both lean quotients with the compiler's IEEE divisions for every fp32 denominator of
            // that range (validate.hip; cached per d_sd, d_so -- paris_hip_stage_filter asks ahead of the first backprojection)
            bool exact = false;
            if(int rc = paris_hip_lean_division_check(ctx, g.d_sd, g.d_so, &exact))
                return rc;
            g.lean_div = exact ? 1u : 0u;
        }
    }
    g.p_dim_x_f = static_cast<float>(p_dim_x);
    g.p_dim_y_f = static_cast<float>(p_dim_y);
    g.skip_invalid = (ctx->bp_skip_invalid != 0 && d_v != nullptr
                      && paris_hip_volume_is_clean(ctx, d_v, static_cast<size_t>(v_dim_x) * v_dim_y * v_dim_z * sizeof(float))) ? 1u : 0u;
    g.lds_floats = (ctx->bp_lds_bytes ? ctx->bp_lds_bytes : LDS_BYTES_DEFAULT) / sizeof(float);
    g.tz = tz;
    // default mapping: beyond 1024^2 planes the y tiles are dealt to the XCDs in pairs, in shallow z chunks (order 15: round 3;
    // rounds 1-2 ran a y band per XCD there, order 12); below, z tiles dealt to the XCDs (order 18; volumes of fewer than 8 z tiles,
    // and rounds 1-2: a contiguous run of tiles per XCD, order 5)
    {
        // nesting inside the dealt orders (BpParams::yfast): deep volumes run a group's y tiles before the next z tile (2048^3: 0.7754 ->
        // 0.7774, 0.7757 -> 0.7784; the 2048^3 ROI of config 5: 0.7387 -> 0.7396, 0.7384 -> 0.7400), a 256-slice slab keeps the z tile
        // first (0.7584 -> 0.7565, 0.7572 -> 0.7540 with the other nesting); same device, interleaved, profiles/r03_ab_tile_order.txt
#ifdef PARIS_HIP_EXPERIMENTS
        static const int tile_nest = std::getenv("PARIS_TILE_NEST") ? std::atoi(std::getenv("PARIS_TILE_NEST")) : -1; // A/B switch
#else
        constexpr int tile_nest = -1;
#endif
        g.yfast = tile_nest >= 0 ? static_cast<uint32_t>(tile_nest) : (v_dim_z > 256u ? 2u : 0u); // (round 4: with the 16-slice tiles, from 257 slices: +0.15 % at 512, +0.25 % at 1024)
    }
    // (y tiles fastest is for the fused kernel's box sharing; the volume stream of this kernel loses with it: 2048^3 0.772 -> 0.662, 1024^3 0.743 -> 0.567)
    g.order = ctx->bp_order >= 0 ? static_cast<uint32_t>(ctx->bp_order) : (plane > (1ull << 20) ? 15u : 18u); // (18 falls back to 5 for volumes of fewer than 8 z tiles: settle_order)
    g.store_sc1 = volume_stream_policy(ctx, v_dim_x, v_dim_y, v_dim_z) == 2 ? 1u : 0u;
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

    const bool timed = !ctx->bp_start.empty(); // paris_hip_backproject_timing_arm(ctx, 0): no events (stream capture)
    const size_t ev = timed ? static_cast<size_t>(ctx->bp_launches % ctx->bp_start.size()) : 0u;
    if(timed)
        PARIS_HIP_TRY(hipEventRecord(ctx->bp_start[ev], ctx->stream));
    if(ctx->bp_variant == 4 && !f16) // the fused kernel with a single projection (measurement: all slices of a tile in flight)
    {
        FusedParams fp{};
        fp.g = g;
        fp.n_proj = 1;
        fp.proj_tab[0] = d_p;
        fp.sin_phi[0] = sin_phi;
        fp.cos_phi[0] = cos_phi;
        const int width = lane_width(d_v, v_dim_x);
        const int vx = ctx->bp_vx == 4 && width == 4 ? 4 : (ctx->bp_vx == 1 || width < 2 ? 1 : 2);
        paris_hip_bp_launch_fused(&fp, vx, ctx->bp_tz == 8u ? 8 : (ctx->bp_tz == 32u ? 32 : 16), volume_stream_policy(ctx, v_dim_x, v_dim_y, v_dim_z) != 0, fd, ctx->stream);
    }
    else
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
        // A small volume cut into 64 x 16 x tz tiles is ONE generation of workgroups or less (256^3: 1024 = 4 per CU), and the
        // launch is then a latency chain -- every workgroup computes its box and columns, stages, and only then streams. Tiles of
        // 64 x 8 columns (two voxels per lane, 8-byte accesses) double the workgroups, so that one's prologue runs under
        // another's stream: 28.5 -> 26.4 us at 256^3 (profiles/r03_ab_c1_tile.txt; the volume lives in the Infinity Cache there,
        // the narrower accesses cost nothing)
        if(ctx->bp_vx == 0 && vx == 4 && ctx->bp_variant != 3 && ctx->bp_variant != 5)
        {
            const uint64_t tiles = static_cast<uint64_t>((v_dim_x + 63u) / 64u) * ((v_dim_y + 15u) / 16u) * ((v_dim_z + g.tz - 1u) / g.tz);
            if(tiles < 2048u)
                vx = 2;
        }
        // slices in flight per lane: 2 (interleaved A/B on 2048^3: tools/ab_bp.py, profiles/); with the 8-slice tiles of
        // 1024^2 planes one slice plus the prefetch of the next is as fast and leaves more registers
        // Deep volumes on planes beyond 1024^2 (16-slice tiles): ONE slice in flight per lane and FOUR workgroups per CU -- the
        // registers of the one-slice body would admit five, so the launch asks for 40 KiB of LDS per workgroup (160 / 40 = 4; the
        // box budget grows with it). Round 4, same device: 2048^3 0.7841-0.7850 -> 0.7919-0.7931 of the HBM peak, the 2048^3 ROI of
        // config 5 0.7424 -> 0.7640; with five workgroups 0.780, with three 0.705, two slices in flight at four (rounds 1-3) 0.785,
        // a prefetched next slice 0.763. This stream runs fastest with FEW requests in flight -- as many as hide the latency and no
        // more: what sets its rate is the order in which the tiles' requests reach DRAM (profiles/r04_ab_tile_occupancy.txt,
        // r04_ab_tile_prefetch.txt). 8-slice tiles (slabs up to 256 slices, planes up to 1024^2) keep one slice at five: 0.765
        // against 0.711 at four.
        const bool deep_big = g.tz == 16u && ctx->bp_tz == 0u && static_cast<uint64_t>(v_dim_x) * v_dim_y > (1ull << 20);
        if(deep_big && ctx->bp_unroll == 0 && ctx->bp_lds_bytes == 0u && vx == 4)
            g.lds_floats = 40u * 1024u / sizeof(float);
        const int unroll = ctx->bp_unroll ? ctx->bp_unroll : ((g.tz == 8u && ctx->bp_tz == 0u) || (deep_big && ctx->bp_lds_bytes == 0u && vx == 4) ? 1 : 2);
        const bool nt = volume_stream_policy(ctx, v_dim_x, v_dim_y, v_dim_z) != 0;
#ifdef PARIS_HIP_EXPERIMENTS
        const bool want_slice = ctx->bp_variant == 3; // measured slower than the tile kernel so far: opt-in only
        if(want_slice && vx == 4)
        {
            const int nw = ctx->bp_slice_nw ? ctx->bp_slice_nw : 16;
            const int rpl = ctx->bp_slice_rpl ? ctx->bp_slice_rpl : 4;
            const uint32_t box_bytes = ctx->bp_lds_bytes ? ctx->bp_lds_bytes : LDS_BYTES_DEFAULT;
            if(int rc = launch_slice_shape(g, nw, rpl, box_bytes, nt, fd, ctx->stream))
                return rc;
        }
        else if(vx == 4 && ctx->bp_variant == 5 && v_dim_y <= 65535u)
        {
            const size_t need = 3u * static_cast<size_t>(v_dim_x) * v_dim_y;
            if(ctx->colstate_floats < need)
            {
                if(ctx->colstate != nullptr)
                {
                    PARIS_HIP_TRY(hipStreamSynchronize(ctx->stream));
                    PARIS_HIP_TRY(hipFree(ctx->colstate));
                    ctx->colstate = nullptr;
                }
                PARIS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->colstate), need * sizeof(float)));
                ctx->colstate_floats = need;
            }
            g.colstate = ctx->colstate;
            if(unroll == 1)
                fd ? launch_tile_pre<1, true>(g, ctx->stream) : launch_tile_pre<1, false>(g, ctx->stream);
            else
                fd ? launch_tile_pre<2, true>(g, ctx->stream) : launch_tile_pre<2, false>(g, ctx->stream);
        }
        else
#endif // PARIS_HIP_EXPERIMENTS
        if(vx == 4)
            launch_tile_flags<4>(g, unroll, nt, fd, ctx->stream);
        else if(vx == 2)
            launch_tile_flags<2>(g, unroll, nt, fd, ctx->stream);
        else
            launch_tile_flags<1>(g, unroll, nt, fd, ctx->stream);
    }
    if(timed)
        PARIS_HIP_TRY(hipEventRecord(ctx->bp_stop[ev], ctx->stream));
    ++ctx->bp_launches;
    if(int rc = paris_hip_note_projection_use(ctx, d_p, p_pitch * p_dim_y))
        return rc;
    return paris_hip_finish(ctx);
}

// ---- deferred backprojection -------------------------------------------------------------------------------------
// With a deferral depth n > 1 a paris_hip_backproject call snapshots its projection into a device ring (stream-ordered
// copy, so the caller may reuse its buffer as usual) and returns; n pending calls -- or fewer when a call with other
// volume / geometry arguments arrives or when the volume is observed -- are added by ONE fused launch in call order,
// which is bit-identical to n single launches and moves 8/n instead of 8 bytes per voxel-update.
static int batch_impl(paris_hip_ctx* ctx, const void* d_p, bool f16, size_t p_pitch, size_t p_stride_bytes, uint32_t n_proj, uint32_t p_dim_x,
                      uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                      const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo, int enable_roi,
                      const paris_region_of_interest* roi, const float* sin_phi, const float* cos_phi, float delta_s, float delta_t,
                      const void* const* ptrs = nullptr);

int paris_hip_launch_deferred(paris_hip_ctx* ctx)
{
    if(ctx != nullptr && ctx->pending_weight.active) // every observer of device state also sees a held-back weighting done
        if(int rc = paris_hip_flush_pending_weight(ctx))
            return rc;
    if(ctx == nullptr || ctx->defer_count == 0)
        return PARIS_HIP_SUCCESS;
    const uint32_t n = ctx->defer_count;
    const uint32_t depth = ctx->defer_depth;
    const uint32_t half = ctx->defer_half;
    const bool has_refs = ctx->held_count != 0u || !ctx->defer_zombies.empty();
    // slot i of the group lives at defer_ptr[i]: a snapshot in this half of the ring, or the caller's own buffer (by reference)
    const void* const* where = ctx->defer_ptr.data();
    // beside the caller's next calls, on the ctx's second stream -- unless the caller wants every call complete on return, or is
    // capturing `stream` into a graph (a fork it does not know of would leave the capture unjoined)
    bool overlap = ctx->bp_overlap != 0 && n > 1 && !(ctx->flags & PARIS_HIP_CTX_SYNCHRONOUS);
    if(overlap)
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if(hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
            overlap = false;
        (void)hipGetLastError();
    }
    hipStream_t caller_stream = ctx->stream;
    if(!overlap && ctx->bp_inflight)
    {
        // this group runs on the caller's stream (a single projection, a synchronous ctx, a capture) while earlier groups may still
        // run on the second one: they add to the same volume and come first
        PARIS_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->bp_half_done[ctx->bp_last_half], 0));
        ctx->bp_inflight = false;
        ctx->bp_half_busy[0] = ctx->bp_half_busy[1] = false;
    }
    if(overlap)
    {
        if(int rc = paris_hip_ensure_bp_stream(ctx))
            return rc;
        // everything enqueued on the caller's stream so far -- the snapshots, and whatever touched the volume before them --
        // comes first
        PARIS_HIP_TRY(hipEventRecord(ctx->bp_ring_ready, caller_stream));
        PARIS_HIP_TRY(hipStreamWaitEvent(ctx->bp_stream, ctx->bp_ring_ready, 0));
        ctx->stream = ctx->bp_stream;
    }
    // slots that took a held-back weighting + filter along (filter deferral): one launch for the group when every slot asks for the
    // same rows with the same constants, a launch per slot otherwise; on the stream the fused launch follows on
    {
        uint32_t flagged = 0;
        bool uniform = true;
        for(uint32_t i = 0; i < n && i < ctx->defer_wf.size(); ++i)
        {
            const auto& w = ctx->defer_wf[i];
            if(!w.active)
                continue;
            ++flagged;
            const auto& w0 = ctx->defer_wf[0];
            uniform = uniform && w0.active && w.dim_x == w0.dim_x && w.row_first == w0.row_first && w.row_count == w0.row_count
                      && std::memcmp(&w.h_min, &w0.h_min, 5u * sizeof(float)) == 0 && w.d_kp == w0.d_kp && w.plan == w0.plan
                      && w.filter_size == w0.filter_size;
        }
        const uint32_t pitch_f = static_cast<uint32_t>(ctx->defer_pitch / sizeof(float));
        const auto rows_of = [&](uint32_t i, uint32_t row_first) {
            return reinterpret_cast<float*>(static_cast<char*>(const_cast<void*>(where[i])) + static_cast<size_t>(row_first) * ctx->defer_pitch);
        };
        int wrc = PARIS_HIP_SUCCESS;
        if(flagged == n && uniform && n > 0)
        {
            // one launch for the group: the frames' band rows by table (snapshots in the ring and buffers by reference alike)
            const auto& w = ctx->defer_wf[0];
            float* tab[FUSED_MAX];
            for(uint32_t i = 0; i < n; ++i)
                tab[i] = rows_of(i, w.row_first);
            wrc = paris_hip_fused_filter_launch(ctx, tab[0], pitch_f, w.dim_x, w.row_count, w.row_first, true, w.h_min, w.v_min, w.d_sd, w.l_px_row,
                                                w.l_px_col, w.d_kp, w.plan, w.filter_size, nullptr, 0u, n, 0u, 0u, tab);
        }
        else
            for(uint32_t i = 0; i < n && i < ctx->defer_wf.size() && wrc == PARIS_HIP_SUCCESS; ++i)
            {
                const auto& w = ctx->defer_wf[i];
                if(!w.active)
                    continue;
                wrc = paris_hip_fused_filter_launch(ctx, rows_of(i, w.row_first), pitch_f, w.dim_x, w.row_count, w.row_first, true, w.h_min, w.v_min, w.d_sd,
                                                    w.l_px_row, w.l_px_col, w.d_kp, w.plan, w.filter_size, nullptr, 0u);
            }
        for(auto& w : ctx->defer_wf)
            w.active = false;
        if(wrc != PARIS_HIP_SUCCESS)
        {
            // the group's projections are unfiltered and stay so: the group is dropped rather than added to the volume as it is by
            // the next flush; the caller sees the error
            ctx->defer_count = 0;
            ctx->defer_uses_ring = false;
            (void)hipStreamSynchronize(ctx->stream); // filter launches that did go out may still run in buffers that are given back below
            ctx->stream = caller_stream;
            (void)paris_hip_release_group_references(ctx, 0u);
            return wrc;
        }
    }
    ctx->defer_count = 0; // before the launch: batch_impl may fall back to single launches, which must not be deferred again
    ctx->defer_depth = 1;
    // (held_count is parked while the launch is made: the projections it reads are the group's own, nothing to guard against)
    const uint32_t held = ctx->held_count;
    ctx->held_count = 0u;
    int rc = batch_impl(ctx, where[0], ctx->defer_f16, ctx->defer_pitch, ctx->defer_pitch * ctx->defer_dim_y, n, ctx->defer_dim_x,
                        ctx->defer_dim_y, ctx->key_v, ctx->key_dims[0], ctx->key_dims[1], ctx->key_dims[2], ctx->key_dims[3],
                        &ctx->key_det, &ctx->key_vol, ctx->key_enable_roi, &ctx->key_roi, ctx->defer_sin.data(),
                        ctx->defer_cos.data(), ctx->key_delta_s, ctx->key_delta_t, where);
    ctx->held_count = held;
    hipStream_t ran_on = ctx->stream;
    ctx->stream = caller_stream;
    ctx->defer_depth = depth;
    if(has_refs)
    {
        // ONE event for every buffer the group read by reference: recorded behind the launch on the stream that ran it
        uint64_t group = 0u;
        if(rc == PARIS_HIP_SUCCESS)
        {
            group = ctx->group_seq + 1u;
            hipEvent_t& e = ctx->group_events[group % paris_hip_ctx::GROUP_EVENTS];
            hipError_t err = e != nullptr ? hipSuccess : hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if(err == hipSuccess)
                err = hipEventRecord(e, ran_on);
            if(err != hipSuccess)
            {
                // no event to wait for later: wait now (the buffers are then free at once)
                (void)hipStreamSynchronize(ran_on);
                group = 0u;
                rc = static_cast<int>(err);
            }
            else
                ctx->group_seq = group;
        }
        else
            (void)hipStreamSynchronize(ran_on); // whatever part of the group was launched has read its buffers
        const int rrc = paris_hip_release_group_references(ctx, group);
        if(rc == PARIS_HIP_SUCCESS)
            rc = rrc;
    }
    if(overlap)
    {
        PARIS_HIP_TRY(hipEventRecord(ctx->bp_half_done[half], ctx->bp_stream));
        if(ctx->defer_uses_ring) // (a group of references leaves its half of the ring alone)
            ctx->bp_half_busy[half] = true;
        ctx->bp_inflight = true;
        ctx->bp_last_half = half;
    }
    if(ctx->defer_uses_ring)
        ctx->defer_half = half ^ 1u;
    ctx->defer_uses_ring = false;
    return rc;
}

int paris_hip_flush_deferred(paris_hip_ctx* ctx)
{
    if(int rc = paris_hip_launch_deferred(ctx))
        return rc;
    if(ctx != nullptr && ctx->bp_inflight)
    {
        // join: what the caller enqueues on its stream from here on comes after every fused launch (launches on bp_stream are
        // in order, so the last one's event covers all)
        PARIS_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->bp_half_done[ctx->bp_last_half], 0));
        ctx->bp_inflight = false;
        ctx->bp_half_busy[0] = ctx->bp_half_busy[1] = false; // both halves' launches precede that event
    }
    return PARIS_HIP_SUCCESS;
}

// how many calls the first group of a sequence waits for (then twice as many, ... until the depth)
static uint32_t first_ramp()
{
#ifdef PARIS_HIP_EXPERIMENTS
    static const uint32_t v = [] { const char* e = std::getenv("PARIS_DEFER_RAMP"); const int n = e ? std::atoi(e) : 0; return n > 0 ? static_cast<uint32_t>(n) : 8u; }(); // A/B switch
    return v;
#else
    return 8u;
#endif
}

static int defer_backproject(paris_hip_ctx* ctx, const void* d_p, bool f16, size_t p_pitch, uint32_t p_dim_x, uint32_t p_dim_y, float* d_v,
                             uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                             const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo, int enable_roi,
                             const paris_region_of_interest* roi, float sin_phi, float cos_phi, float delta_s, float delta_t)
{
    // Filter deferral: weighting and row filter of exactly this projection are still held back (paris_hip_apply_filter). They move
    // into the ring slot with the unfiltered snapshot and run there, with the rest of the group, before the fused launch.
    paris_hip_ctx::pending_weight_t taken{};
    {
        auto& w = ctx->pending_weight;
        if(w.active && w.filter && !f16 && w.d_p == d_p && w.pitch == p_pitch && w.dim_x == p_dim_x && w.dim_y == p_dim_y)
        {
            taken = w;
            w.active = false;
            w.filter = false;
        }
    }
    const auto give_back = [&](int rc) { // the call fails or has nothing to add: the filter was still asked for
        if(taken.active)
        {
            ctx->pending_weight = taken;
            (void)paris_hip_flush_pending_weight(ctx);
        }
        return rc;
    };
    // the checks of an immediate call, so that a bad argument is reported by the call that made it
    BpParams g;
    bool fd = false, skip = false;
    if(int rc = fill_params(ctx, d_p, f16, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo, vol_geo,
                            enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t, g, fd, skip))
        return give_back(rc);
    if(skip)
        return give_back(paris_hip_finish(ctx));
    const size_t px = f16 ? sizeof(uint16_t) : sizeof(float);
    const paris_region_of_interest no_roi{};
    const paris_region_of_interest& r = enable_roi ? *roi : no_roi;
    const uint32_t dims[4] = {v_dim_x, v_dim_y, v_dim_z, v_offset};
    // the arguments the pending calls share -- or, with nothing pending, those of the group launched last (key_valid): a call that
    // continues the same reconstruction starts the next group without joining the second stream, so that a whole group's copies
    // and filters run beside the previous group's fused launch (the join used to come with every group's first call)
    const bool same = (ctx->defer_count != 0 || ctx->key_valid) && ctx->key_v == d_v && std::memcmp(ctx->key_dims, dims, sizeof(dims)) == 0
                      && std::memcmp(&ctx->key_det, det_geo, sizeof(*det_geo)) == 0 && std::memcmp(&ctx->key_vol, vol_geo, sizeof(*vol_geo)) == 0
                      && ctx->key_enable_roi == (enable_roi ? 1 : 0) && std::memcmp(&ctx->key_roi, &r, sizeof(r)) == 0
                      && std::memcmp(&ctx->key_delta_s, &delta_s, sizeof(float)) == 0 && std::memcmp(&ctx->key_delta_t, &delta_t, sizeof(float)) == 0
                      && ctx->defer_dim_x == p_dim_x && ctx->defer_dim_y == p_dim_y && ctx->defer_f16 == f16
                      && ctx->defer_ptr.size() >= ctx->defer_depth;
    if(!same)
    {
        if(int rc = paris_hip_flush_deferred(ctx))
            return give_back(rc);
        ctx->key_valid = false;
        if(ctx->defer_ring != nullptr && (ctx->defer_dim_x != p_dim_x || ctx->defer_dim_y != p_dim_y || ctx->defer_slots < ctx->defer_depth || ctx->defer_f16 != f16))
        {
            // the ring is of another shape: a new one is made when a snapshot needs it
            hipError_t err = hipStreamSynchronize(ctx->stream); // a launch may still read the old ring (the flush above joined the second stream)
            if(err == hipSuccess)
                err = hipFree(ctx->defer_ring);
            if(err != hipSuccess)
                return give_back(static_cast<int>(err));
            ctx->defer_ring = nullptr;
            ctx->defer_slots = 0;
        }
        ctx->defer_pitch = (static_cast<size_t>(p_dim_x) * px + 255u) / 256u * 256u;
        ctx->defer_f16 = f16;
        ctx->defer_dim_x = p_dim_x;
        ctx->defer_dim_y = p_dim_y;
        ctx->key_v = d_v;
        std::memcpy(ctx->key_dims, dims, sizeof(dims));
        ctx->key_det = *det_geo;
        ctx->key_vol = *vol_geo;
        ctx->key_enable_roi = enable_roi ? 1 : 0;
        ctx->key_roi = r;
        ctx->key_delta_s = delta_s;
        ctx->key_delta_t = delta_t;
        ctx->key_valid = true;
        ctx->defer_ramp = first_ramp(); // a new sequence: its first groups are launched early
        ctx->defer_sin.assign(ctx->defer_depth, 0.f);
        ctx->defer_cos.assign(ctx->defer_depth, 0.f);
        ctx->defer_ptr.assign(ctx->defer_depth, nullptr);
        ctx->defer_wf.assign(ctx->defer_depth, paris_hip_ctx::pending_weight_t{});
    }
    // By reference (paris_hip_set_backproject_references): the projection is a whole buffer of paris_hip_malloc_projection with the
    // ring's row pitch -- the group reads the buffer itself. Until the group has been launched the library answers for the buffer:
    // paris_hip_free() marks it, every other call that touches it launches the group first (paris_hip_projection_guard).
    paris_hip_ctx::proj_alloc* mine = nullptr;
    bool by_reference = ctx->defer_refs != 0 && p_pitch == ctx->defer_pitch && !(ctx->flags & PARIS_HIP_CTX_SYNCHRONOUS);
    if(by_reference && !ctx->owns_stream)
    {
        // a caller's stream may be being captured into a graph: the group's completion event could then never be queried, and a
        // replay would read buffers long recycled -- such calls keep their snapshots
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if(hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
            by_reference = false;
        (void)hipGetLastError();
    }
    if(by_reference)
    {
        auto a = ctx->proj_allocs.find(const_cast<void*>(d_p));
        if(a != ctx->proj_allocs.end() && a->second.bytes >= p_pitch * static_cast<size_t>(p_dim_y))
            mine = &a->second;
    }
    const void* lives_at = d_p;
    if(mine != nullptr)
    {
        ++mine->held;
        ++ctx->held_count;
        mine->touched = true;
    }
    else
    {
        if(ctx->defer_ring == nullptr)
        {
            hipError_t err = hipMalloc(reinterpret_cast<void**>(&ctx->defer_ring), 2u * ctx->defer_pitch * p_dim_y * ctx->defer_depth); // two halves
            if(err == hipErrorOutOfMemory)
            {
                (void)hipGetLastError();
                if(int rc = paris_hip_drain_device_pool(ctx))
                    return give_back(rc);
                err = hipMalloc(reinterpret_cast<void**>(&ctx->defer_ring), 2u * ctx->defer_pitch * p_dim_y * ctx->defer_depth);
            }
            if(err != hipSuccess)
            {
                ctx->defer_ring = nullptr;
                return give_back(static_cast<int>(err));
            }
            ctx->defer_slots = ctx->defer_depth;
        }
        const uint32_t half = ctx->defer_half;
        if(!ctx->defer_uses_ring && ctx->bp_half_busy[half])
        {
            // the fused launch that read this half of the ring (two ring groups ago) must be done before the half is written again
            const hipError_t err = hipStreamWaitEvent(ctx->stream, ctx->bp_half_done[half], 0);
            if(err != hipSuccess)
                return give_back(static_cast<int>(err));
            ctx->bp_half_busy[half] = false;
        }
        ctx->defer_uses_ring = true;
        char* slot = reinterpret_cast<char*>(ctx->defer_ring) + ctx->defer_pitch * p_dim_y * (static_cast<size_t>(half) * ctx->defer_slots + ctx->defer_count);
        hipError_t err = hipSuccess;
        if(p_pitch == ctx->defer_pitch) // rows as far apart as the ring's: one linear copy (the padding travels along)
            err = hipMemcpyAsync(slot, d_p, p_pitch * (p_dim_y - 1u) + static_cast<size_t>(p_dim_x) * px, hipMemcpyDeviceToDevice, ctx->stream);
        else
            err = hipMemcpy2DAsync(slot, ctx->defer_pitch, d_p, p_pitch, static_cast<size_t>(p_dim_x) * px, p_dim_y, hipMemcpyDeviceToDevice, ctx->stream);
        if(err != hipSuccess)
            return give_back(static_cast<int>(err));
        if(int rc = paris_hip_note_projection_use(ctx, d_p, p_pitch * p_dim_y)) // the snapshot copy is the last reader of the caller's buffer
            return give_back(rc);
        lives_at = slot;
    }
    if(ctx->defer_ptr.size() < ctx->defer_depth)
        ctx->defer_ptr.resize(ctx->defer_depth, nullptr);
    ctx->defer_ptr[ctx->defer_count] = lives_at;
    ctx->defer_sin[ctx->defer_count] = sin_phi;
    ctx->defer_cos[ctx->defer_count] = cos_phi;
    if(ctx->defer_wf.size() < ctx->defer_depth)
        ctx->defer_wf.resize(ctx->defer_depth);
    ctx->defer_wf[ctx->defer_count] = taken; // (inactive: the snapshot is filtered already)
    if(++ctx->defer_count >= std::min(ctx->defer_depth, ctx->defer_ramp))
    {
        ctx->defer_ramp = ctx->defer_ramp >= ctx->defer_depth ? ctx->defer_depth : ctx->defer_ramp * 2u; // 8, 16, 32 ... then the depth
        if(int rc = paris_hip_launch_deferred(ctx)) // no join: the caller's next calls run beside the launch
            return rc;
        if(ctx->flags & PARIS_HIP_CTX_SYNCHRONOUS)
            return paris_hip_flush_deferred(ctx);
    }
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_set_backproject_deferral(paris_hip_ctx* ctx, uint32_t depth)
{
    if(ctx == nullptr || depth == 0 || depth > 64)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    ctx->defer_depth = depth;
    // a sequence continued after this call starts over: its tables are sized for the new depth and its first groups are launched
    // early again
    ctx->key_valid = false;
    ctx->defer_ramp = first_ramp();
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_references(paris_hip_ctx* ctx, int enable)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    ctx->defer_refs = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

// What this ctx may keep allocated for projections of dim_x x dim_y pixels beside the caller's volume, with its present deferral
// settings: the buffer rotation of paris_hip_malloc_projection (parked buffers), the pending group's buffers or the ring of
// snapshots. A driver passes it to paris_hip_make_subvolume_information_reserving.
extern "C" int paris_hip_projection_reserve_bytes(paris_hip_ctx* ctx, uint32_t dim_x, uint32_t dim_y, size_t* bytes)
{
    if(ctx == nullptr || bytes == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const size_t frame = ((static_cast<size_t>(dim_x) * sizeof(float) + 255u) & ~static_cast<size_t>(255u)) * dim_y;
    size_t total = std::min(ctx->device_pool_capacity(frame) * frame, paris_hip_ctx::PARKED_DEVICE_LIMIT + frame);
    if(ctx->defer_depth > 1u)
        total += (ctx->defer_refs != 0 ? 1u : 2u) * static_cast<size_t>(ctx->defer_depth) * frame; // the pending group by reference, or the ring's two halves
    *bytes = total;
    return PARIS_HIP_SUCCESS;
}

int paris_hip_projection_guard_slow(paris_hip_ctx* ctx, const void* d_p, size_t bytes, hipStream_t writer, bool writes)
{
    (void)bytes; // (a call's range lies inside one allocation: the one that holds its first byte)
    if(d_p == nullptr || ctx->proj_allocs.empty())
        return PARIS_HIP_SUCCESS;
    auto a = ctx->proj_allocs.upper_bound(const_cast<void*>(d_p));
    if(a == ctx->proj_allocs.begin())
        return PARIS_HIP_SUCCESS;
    --a;
    if(static_cast<const char*>(d_p) >= static_cast<const char*>(a->first) + a->second.bytes)
        return PARIS_HIP_SUCCESS;
    if(a->second.held != 0u)
    {
        // the pending group reads this buffer itself (and may still have its weight + filter to run in it): the group goes first
        if(int rc = paris_hip_launch_deferred(ctx))
            return rc;
    }
    (void)writes;
    // a launched group may still be reading the buffer -- or, with filter deferral, still be filtering it in place: a write waits for
    // that launch, and so does a read (it must see the filtered pixels)
    if(a->second.group != 0u)
    {
        bool done = false;
        if(int rc = paris_hip_group_done(ctx, a->second.group, false, &done))
            return rc;
        if(done)
            a->second.group = 0u;
        else // (the event of that launch, or of a later one recorded in the same slot)
            PARIS_HIP_TRY(hipStreamWaitEvent(writer != nullptr ? writer : ctx->stream, ctx->group_events[a->second.group % paris_hip_ctx::GROUP_EVENTS], 0));
    }
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_overlap(paris_hip_ctx* ctx, int enable)
{
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    ctx->bp_overlap = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

void paris_hip_warm_backproject()
{
    hipFuncAttributes a{};
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&fastdiv_validate_kernel));
}

extern "C" int paris_hip_pending_backprojections(paris_hip_ctx* ctx, uint32_t* count, void** d_v)
{
    if(ctx == nullptr || count == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    *count = ctx->defer_count;
    if(d_v != nullptr)
        *d_v = ctx->defer_count != 0 ? ctx->key_v : nullptr;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_flush(paris_hip_ctx* ctx)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    return paris_hip_flush_deferred(ctx);
}

extern "C" int paris_hip_backproject(paris_hip_ctx* ctx, const float* d_p, size_t p_pitch, uint32_t p_dim_x,
                                     uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z,
                                     uint32_t v_offset, const paris_detector_geometry* det_geo,
                                     const paris_volume_geometry* vol_geo, int enable_roi,
                                     const paris_region_of_interest* roi, float sin_phi, float cos_phi, float delta_s,
                                     float delta_t)
{
    if(ctx != nullptr && ctx->defer_depth > 1 && (ctx->bp_variant == 0 || ctx->bp_variant == 4))
        return defer_backproject(ctx, d_p, false, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo, vol_geo,
                                 enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t);
    return backproject_impl(ctx, d_p, false, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo, vol_geo,
                            enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t);
}

extern "C" int paris_hip_backproject_f16(paris_hip_ctx* ctx, const uint16_t* d_p, size_t p_pitch, uint32_t p_dim_x,
                                         uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y,
                                         uint32_t v_dim_z, uint32_t v_offset, const paris_detector_geometry* det_geo,
                                         const paris_volume_geometry* vol_geo, int enable_roi,
                                         const paris_region_of_interest* roi, float sin_phi, float cos_phi,
                                         float delta_s, float delta_t)
{
    if(ctx != nullptr && ctx->defer_depth > 1 && (ctx->bp_variant == 0 || ctx->bp_variant == 4))
        return defer_backproject(ctx, d_p, true, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo, vol_geo,
                                 enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t);
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
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
    if(int rc = paris_hip_flush_pending_weight(ctx))
        return rc;
    if(d_src == nullptr || d_dst == nullptr || src_pitch % sizeof(float) != 0 || dst_pitch % sizeof(uint16_t) != 0
       || src_pitch < static_cast<size_t>(dim_x) * sizeof(float) || dst_pitch < static_cast<size_t>(dim_x) * sizeof(uint16_t))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(dim_x == 0 || dim_y == 0)
        return paris_hip_finish(ctx);
    if(int rc = paris_hip_projection_guard(ctx, d_src, src_pitch * dim_y, nullptr, false))
        return rc;
    if(int rc = paris_hip_projection_guard(ctx, d_dst, dst_pitch * dim_y, ctx->stream, true))
        return rc;
    const dim3 grid((dim_x + 255u) / 256u, dim_y < 65535u ? dim_y : 65535u);
    hipLaunchKernelGGL(to_half_kernel, grid, dim3(256), 0, ctx->stream, d_src, static_cast<uint32_t>(src_pitch / sizeof(float)),
                       reinterpret_cast<_Float16*>(d_dst), static_cast<uint32_t>(dst_pitch / sizeof(uint16_t)), dim_x, dim_y);
    if(int rc = paris_hip_note_projection_use(ctx, d_src, src_pitch * dim_y))
        return rc;
    if(int rc = paris_hip_note_projection_use(ctx, d_dst, dst_pitch * dim_y)) // (the destination is written by this launch)
        return rc;
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
    if(int rc = paris_hip_flush_deferred(ctx)) // keeps the call order of deferred and explicit batches
        return rc;
    return batch_impl(ctx, d_p, false, p_pitch, p_stride_bytes, n_proj, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo,
                      vol_geo, enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t);
}

extern "C" int paris_hip_backproject_batch_f16(paris_hip_ctx* ctx, const uint16_t* d_p, size_t p_pitch, size_t p_stride_bytes,
                                               uint32_t n_proj, uint32_t p_dim_x, uint32_t p_dim_y, float* d_v, uint32_t v_dim_x,
                                               uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                                               const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo,
                                               int enable_roi, const paris_region_of_interest* roi, const float* sin_phi,
                                               const float* cos_phi, float delta_s, float delta_t)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    return batch_impl(ctx, d_p, true, p_pitch, p_stride_bytes, n_proj, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo,
                      vol_geo, enable_roi, roi, sin_phi, cos_phi, delta_s, delta_t);
}

static int batch_impl(paris_hip_ctx* ctx, const void* d_p, bool f16, size_t p_pitch, size_t p_stride_bytes, uint32_t n_proj, uint32_t p_dim_x,
                      uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                      const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo, int enable_roi,
                      const paris_region_of_interest* roi, const float* sin_phi, const float* cos_phi, float delta_s, float delta_t,
                      const void* const* ptrs)
{
    const size_t px = f16 ? sizeof(uint16_t) : sizeof(float);
    if(ctx == nullptr || sin_phi == nullptr || cos_phi == nullptr || (ptrs == nullptr && p_stride_bytes % px != 0))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(n_proj == 0)
        return paris_hip_finish(ctx);
    // projection i of the batch lives at base + i * stride (consecutive projections must not overlap), or -- the deferral's
    // references and snapshots -- wherever ptrs[i] says
    if(ptrs == nullptr && n_proj > 1 && p_stride_bytes < p_pitch * static_cast<size_t>(p_dim_y))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const auto where = [&](uint32_t i) { return ptrs != nullptr ? static_cast<const char*>(ptrs[i]) : static_cast<const char*>(d_p) + static_cast<size_t>(i) * p_stride_bytes; };

    // The cross-check variants take the sequence of single-projection launches that the fused kernel is defined to
    // equal, and so does a single projection: that is the tile kernel's case (memory bound, nothing to fuse).
    // Shape of the fused kernel. Default: one voxel per lane along x, 32 slices deep -- a column's per-projection setup (its two
    // divisions, the detector column and weights, the tests that let its taps skip the per-voxel checks) is paid once per 32
    // voxel-updates: +5 % over two columns x 16 slices on the 2048^3 grid, +9 % at 1024^3 (profiles/r02_tune_fused_shapes.txt);
    // volumes of at most 16 slices keep the 16-slice tile. The tuning knob can ask for 2 or 4 voxels per lane (rows 8- / 16-byte
    // aligned) and for 8, 16 or 32 slices (32 only with one voxel per lane, 8 only with 2 or 4).
    const int width = d_v != nullptr ? lane_width(d_v, v_dim_x) : 1;
    // ... provided the detector box of a 64 x 4 x 32 tile fits the LDS budget at this geometry's magnification (coarse grids on
    // fine detectors do not: 256^3 from 512^2 projections needs 76 rows of 144 floats for 32 slices, most taps would take the
    // global path and 2 x 16 is 24 % faster there, profiles/r02_ab_fused_steps.txt)
    bool deep_box_fits = true;
    {
        const double d_so = std::fabs(static_cast<double>(det_geo->d_so)), d_sd = d_so + std::fabs(static_cast<double>(det_geo->d_od));
        const double mag = d_so > 0.0 ? d_sd / d_so : 1.0; // at the rotation axis: most tiles are near it
        const double rows = 32.0 * vol_geo->l_vx_z * mag / det_geo->l_px_col + 4.0 * std::max(vol_geo->l_vx_x, vol_geo->l_vx_y) * mag / det_geo->l_px_col + 8.0;
        const double cols = 64.0 * std::max(vol_geo->l_vx_x, vol_geo->l_vx_y) * mag / det_geo->l_px_row + 16.0;
        const double budget = ctx->bp_lds_bytes != 0u ? ctx->bp_lds_bytes : 32.0 * 1024.0;
        deep_box_fits = std::isfinite(rows) && std::isfinite(cols) && rows * std::max(132.0, cols) * sizeof(float) <= budget;
    }
    int fused_vx = ((v_dim_z <= 16u || !deep_box_fits) && width >= 2) ? 2 : 1;
    if(ctx->bp_vx == 4 && width == 4)
        fused_vx = 4;
    else if(ctx->bp_vx == 2 && width >= 2)
        fused_vx = 2;
    else if(ctx->bp_vx == 1)
        fused_vx = 1;
    int fused_tz = fused_vx == 1 ? ((v_dim_z <= 16u || !deep_box_fits) ? 16 : 32) : 16;
    if(ctx->bp_tz == 8u && fused_vx != 1)
        fused_tz = 8;
    else if(ctx->bp_tz == 16u)
        fused_tz = 16;
    else if(ctx->bp_tz == 32u && fused_vx == 1)
        fused_tz = 32;
    const bool fused_ok = n_proj > 1 && (ctx->bp_variant == 0 || ctx->bp_variant == 4) && d_v != nullptr;
    if(!fused_ok)
    {
        const unsigned saved = ctx->flags;
        ctx->flags &= ~PARIS_HIP_CTX_SYNCHRONOUS;
        int rc = PARIS_HIP_SUCCESS;
        for(uint32_t i = 0; i < n_proj && rc == PARIS_HIP_SUCCESS; ++i)
        {
            const char* p = where(i);
            rc = backproject_impl(ctx, p, f16, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo, vol_geo,
                                  enable_roi, roi, sin_phi[i], cos_phi[i], delta_s, delta_t);
        }
        ctx->flags = saved;
        if(rc != PARIS_HIP_SUCCESS)
            return rc;
        return paris_hip_finish(ctx);
    }

    const bool nt = volume_stream_policy(ctx, v_dim_x, v_dim_y, v_dim_z) != 0;
    for(uint32_t first = 0; first < n_proj; first += FUSED_MAX)
    {
        const uint32_t n = std::min<uint32_t>(FUSED_MAX, n_proj - first);
        FusedParams fp{};
        bool fd = false, skip = false;
        const char* p0 = where(first);
        if(int rc = fill_params(ctx, p0, f16, p_pitch, p_dim_x, p_dim_y, d_v, v_dim_x, v_dim_y, v_dim_z, v_offset, det_geo,
                                vol_geo, enable_roi, roi, sin_phi[first], cos_phi[first], delta_s, delta_t, fp.g, fd, skip))
            return rc;
        if(skip)
            break;
        fp.n_proj = n;
        // The default dealt order hands an XCD y tiles in groups of 32 rows (two 16-row tiles of the single-projection kernel). The fused
        // kernel's tiles are 4 * fused_vx rows: the same 32 rows are 8 / 4 / 2 of them (orders 17 / 16 / 15). Its tiles stage a detector
        // box per projection, and tiles 4 rows apart share nine tenths of it: with pairs of 4-row tiles an XCD's L2 saw every box row
        // about half as often as it was fetched (358 GB per 48-projection launch at 2048^3 against 69 GB of algorithmic bytes)
        if(ctx->bp_order < 0 && fp.g.order == 15u)
            fp.g.order = fused_vx == 1 ? 17u : (fused_vx == 2 ? 16u : 15u);
        // ... and an XCD runs the eight (four, two) y tiles of one dealt group fastest, then x, then its next group: the 128 workgroups
        // resident on an XCD are 32 adjacent rows of 16 (32, 64) x tiles instead of 4 y tiles of 32 x tiles. A tile stages one detector
        // box per projection; where x runs across the detector (projections near 90 and 270 degrees) every x tile has a box of its own
        // and only tiles that differ in y -- the depth there -- share one, and tiles far apart in depth see it shifted and scaled. Per
        // 48-projection launch at 2048^3 the L2 misses fell from 378 ... 485 GB to 78 ... 91 GB at those angles (53 against 95 GB near 0
        // and 180 degrees), the mean over the circle from 319 to 74 GB against 35 GB of compulsory reads, and the launch got 4 % faster
        // (profiles/r03_ab_fused_traffic.txt). Wider groups cut the misses further (64 rows: 57 GB, 128: 48) but unbalance the XCDs'
        // shares of skipped tiles: -1 % and -3 %.
#ifdef PARIS_HIP_EXPERIMENTS
        static const bool x_fast = std::getenv("PARIS_FUSED_XFAST") != nullptr; // A/B switch: the single-projection kernel's nesting
#else
        constexpr bool x_fast = false;
#endif
        fp.g.yfast = x_fast ? 0u : 1u;
        if(ctx->bp_lds_bytes == 0u) // the fused kernel runs 4 workgroups per CU: a 32 KiB box budget (tools/tune_bp.py --fused: +1.7 %)
            fp.g.lds_floats = 32u * 1024u / sizeof(float);
        for(uint32_t i = 0; i < n; ++i)
        {
            fp.sin_phi[i] = sin_phi[first + i];
            fp.cos_phi[i] = cos_phi[first + i];
            fp.proj_tab[i] = where(first + i);
            // fill_params decided the 4-pixel staging from the first projection's address: every later base must be as aligned
            if(reinterpret_cast<uintptr_t>(fp.proj_tab[i]) % (4u * px) != 0)
                fp.g.stage_vec4 = 0u;
        }
        for(uint32_t i = n; i < FUSED_MAX; ++i)
            fp.proj_tab[i] = fp.proj_tab[0];
        const bool timed = !ctx->bp_start.empty();
        const size_t ev = timed ? static_cast<size_t>(ctx->bp_launches % ctx->bp_start.size()) : 0u;
        if(timed)
            PARIS_HIP_TRY(hipEventRecord(ctx->bp_start[ev], ctx->stream));
        paris_hip_bp_launch_fused(&fp, fused_vx, fused_tz, nt, fd, ctx->stream);
        if(timed)
            PARIS_HIP_TRY(hipEventRecord(ctx->bp_stop[ev], ctx->stream));
        ++ctx->bp_launches;
        for(uint32_t i = 0; i < n && !ctx->upload_targets.empty(); ++i)
            if(int rc = paris_hip_note_projection_use(ctx, where(first + i), p_pitch * p_dim_y))
                return rc;
    }
    return paris_hip_finish(ctx);
}

extern "C" int paris_hip_last_backproject_ms(paris_hip_ctx* ctx, float* ms)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(ctx == nullptr || ms == nullptr || ctx->bp_launches == 0 || ctx->bp_start.empty())
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
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(ctx == nullptr || variant < 0 || variant > 5)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
#ifndef PARIS_HIP_EXPERIMENTS
    if(variant == 3 || variant == 5) // the slice kernel and the two-pass variant lost and live in the experiments build only
        return PARIS_HIP_ERROR_UNSUPPORTED;
#endif
    ctx->bp_variant = variant;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_order(paris_hip_ctx* ctx, int order, int nontemporal)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(ctx == nullptr || !(order == -1 || order == 0 || order == 1 || order == 5 || order == 8 || order == 9 || order == 12 || (order >= 14 && order <= 18)) || nontemporal < -1 || nontemporal > 2)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
#ifndef PARIS_HIP_EXPERIMENTS
    if(order == 0 || order == 1 || order == 8 || order == 9 || order == 12) // the orders that lost: experiments build only
        return PARIS_HIP_ERROR_UNSUPPORTED;
#endif
    ctx->bp_order = order;
    ctx->bp_nt = nontemporal; // < 0: automatic (by slab size)
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_fast_division_is_exact(paris_hip_ctx* ctx, float divisor, int* exact)
{
    if(int rc = paris_hip_bind(ctx))
        return rc;
    if(exact == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    bool ok = false;
    const int saved = ctx->async_validate;
    ctx->async_validate = 0; // the question is what the device says: wait for it
    const int rc = fastdiv_is_exact(ctx, divisor, &ok);
    ctx->async_validate = saved;
    if(rc != PARIS_HIP_SUCCESS)
        return rc;
    *exact = ok ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_vector_staging(paris_hip_ctx* ctx, int enable)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_stage_vec4 = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_skip_invalid(paris_hip_ctx* ctx, int enable)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_skip_invalid = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_fast_division(paris_hip_ctx* ctx, int enable)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    ctx->bp_lean_div = enable ? 1 : 0; // the shared-reciprocal form of the two per-column divisions follows the same switch
    ctx->bp_fastdiv = enable ? 1 : 0;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_slice_shape(paris_hip_ctx* ctx, int waves, int row_groups)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    const bool ok = (waves == 0 && row_groups == 0) || (waves == 16 && (row_groups == 4 || row_groups == 2))
                    || (waves == 8 && (row_groups == 4 || row_groups == 2 || row_groups == 1));
    if(!ok)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
#ifndef PARIS_HIP_EXPERIMENTS
    if(waves != 0) // there is no slice kernel in this build
        return PARIS_HIP_ERROR_UNSUPPORTED;
#endif
    ctx->bp_slice_nw = waves;
    ctx->bp_slice_rpl = row_groups;
    return PARIS_HIP_SUCCESS;
}

extern "C" int paris_hip_set_backproject_tuning(paris_hip_ctx* ctx, int vx, int unroll, int tz, int lds_bytes)
{
    if(int rc = paris_hip_flush_deferred(ctx))
        return rc;
    if(ctx == nullptr)
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(!(vx == 0 || vx == 1 || vx == 2 || vx == 4) || !(unroll == 0 || unroll == 1 || unroll == 2 || unroll == 3 || unroll == 4))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
    if(tz < 0 || tz > 4096 || lds_bytes < 0 || lds_bytes > static_cast<int>(LDS_BYTES_MAX)
       || (lds_bytes != 0 && lds_bytes < 1024))
        return PARIS_HIP_ERROR_INVALID_ARGUMENT;
#ifndef PARIS_HIP_EXPERIMENTS
    if(unroll > 2) // the launcher picks 1 or 2 by itself; the other bodies are compiled into the experiments build only
        return PARIS_HIP_ERROR_UNSUPPORTED;
#endif
    ctx->bp_vx = vx;
    ctx->bp_unroll = unroll;
    ctx->bp_tz = static_cast<uint32_t>(tz);
    ctx->bp_lds_bytes = static_cast<uint32_t>(lds_bytes);
    return PARIS_HIP_SUCCESS;
}

// 1: this library was built with make EXPERIMENTS=1 -- the kernels, tile orders and switches that lost their A/B runs are compiled in
// (tools/, the variant tests); 0: the product build
extern "C" int paris_hip_has_experiments(void)
{
#ifdef PARIS_HIP_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
