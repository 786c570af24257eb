// Fused multi-projection backprojection kernel (paris_hip_backproject_batch) for gfx950.
//
// A translation unit of its own: its hot loop is bound by LDS / ALU latency at three waves per SIMD, and clang's SLP
// vectoriser turns its fp32 multiplies and adds into v_pk_mul_f32 / v_pk_add_f32, which on gfx950 issue at half the
// rate of the plain instructions (tools/pkbench.hip) and need v_mov packing on top: -fno-slp-vectorize (Makefile) is
// worth +12 % here, while the memory-bound tile kernel in backproject.hip is indifferent to it.
#define PARIS_BP_SINGLE_INSTRUCTION_FLOOR 1
#include "bp_device.h"

namespace
{
    // (Double-buffering the box with the next projection's loads kept in flight was measured slower: the kernel is
    // bound by vector ALU issue, not by staging latency, and the extra live registers cost occupancy.)
    // Waves per SIMD the fused kernel is compiled for. Its loop is bound by LDS / ALU latency until about four waves are
    // resident, so the register budget decides: 4 voxels per lane x 16 slices need 196 VGPRs unconstrained (2 waves,
    // 0.93 TVox/s on the 2048^3 volume), 168 with a few cold-path spills (3 waves, 1.16); 2 voxels per lane run at 4 waves
    // without spills (1.30; a fifth wave adds nothing); 1 voxel per lane fits 7 but pays more staging per voxel (1.16). The base value below is the
    // 4-voxel / 16-slice case; narrower lanes and 8-slice tiles add to it.
#ifndef PARIS_FUSED_PIPELINE
#define PARIS_FUSED_PIPELINE 1
#endif
#ifndef PARIS_FUSED_PARTS
#define PARIS_FUSED_PARTS 4
#endif
#ifndef PARIS_FUSED_THREE_MODES
#define PARIS_FUSED_THREE_MODES 1
#endif
#ifndef PARIS_FUSED_WAVES
#define PARIS_FUSED_WAVES 3
#endif
    // LDS row stride the kernel prefers (floats; 4 x 33: rows 16-byte aligned, consecutive rows start 4 banks apart): boxes up to
    // 128 pixels wide -- the 64-column tile at magnifications up to about 2 -- take it, wider ones keep their own stride
    constexpr int FIXED_STRIDE = 132;

    template <int VX, int TZ, bool NT, bool FD>
    __global__ void __launch_bounds__(256, PARIS_FUSED_WAVES + (VX == 2 ? 1 : VX == 1 ? (TZ == 32 ? 1 : 4) : 0) + (TZ == 8 ? 1 : 0)) bp_fused_kernel(const FusedParams fp)
    {
        extern __shared__ __attribute__((aligned(16))) float lds[];
        BpParams g = fp.g;
        using vec = typename vec_of<VX>::type;
        constexpr uint32_t XL = 64u / VX; // lanes along x per wave
        constexpr uint32_t RW = VX;       // volume rows per wave
        constexpr uint32_t TY = 4u * RW;

        const uint32_t tid = threadIdx.x;
        const uint32_t lane = tid & 63u;
        const uint32_t wave = tid >> 6;

        uint32_t bx, by, bz;
        if(!tile_of_block(g, blockIdx.x, bx, by, bz))
            return;
        const uint32_t k0 = bx * 64u;
        const uint32_t l0 = by * TY;
        const uint32_t m0 = bz * TZ;
        const uint32_t k1 = min(k0 + 63u, g.v_dim_x - 1u);
        const uint32_t l1 = min(l0 + TY - 1u, g.v_dim_y - 1u);
        const uint32_t m1 = min(m0 + TZ - 1u, g.v_dim_z - 1u);
        const uint32_t mcount = m1 - m0 + 1u;

        const uint32_t xq = lane % XL, yy = lane / XL;
        const uint32_t k = k0 + xq * VX;
        const uint32_t l = l0 + wave * RW + yy;
        const bool active = k < g.v_dim_x && l < g.v_dim_y; // inactive lanes still take part in the barriers

        const size_t slice = static_cast<size_t>(g.v_dim_x) * g.v_dim_y;
        float* vp = g.vol + (static_cast<size_t>(m0) * g.v_dim_y + l) * g.v_dim_x + k;
        vec acc[TZ];
#pragma unroll
        for(int z = 0; z < TZ; ++z)
        {
#pragma unroll
            for(int j = 0; j < VX; ++j)
                elem<VX>(acc[z], j) = 0.f; // lanes and slices beyond the volume: summed like the others, never stored
            if(active && static_cast<uint32_t>(z) < mcount)
                acc[z] = load_voxels<VX, NT>(vp + z * slice);
        }

        const float z_first = g.z_base + static_cast<float>(g.m_off + m0) * g.l_vx_z;
        const float z_last = g.z_base + static_cast<float>(g.m_off + m1) * g.l_vx_z;
        // (Keeping the loop constants of the per-voxel chain in VGPRs -- a plain fp32 multiply / add issues 1.6x faster in
        // isolation when no source is a scalar register, tools/pkbench.hip -- was measured: no gain in this instruction mix.)
        const char* base = static_cast<const char*>(fp.g.proj);
        const size_t px = g.proj_f16 ? 2u : 4u;
        // the detector boxes of all projections of this tile, 16 per wave at once (waves 0 and 1: FUSED_MAX = 32)
        __shared__ int box_tab[FUSED_MAX * BOX_WORDS];
        if(wave * 16u < fp.n_proj && wave < 2u)
            tile_boxes_to_lds(g, fp.sin_phi, fp.cos_phi, wave * 16u, fp.n_proj, k0, k1, l0, l1, m0, m1, lane, g.lds_floats, FIXED_STRIDE, box_tab);
        for(uint32_t p = 0; p < fp.n_proj; ++p)
        {
            g.sin_phi = fp.sin_phi[p];
            g.cos_phi = fp.cos_phi[p];
            g.proj = base + static_cast<size_t>(p) * fp.proj_stride * px;
#ifdef PARIS_TIMING_ONLY_NO_BARRIERS // wrong results: prices the two barriers per projection
            if(p == 0u)
#endif
            __syncthreads(); // the previous projection's taps are done with the LDS box (p == 0: the table of boxes is written)
            const Box box = box_from_lds(box_tab, p, lane);
#ifdef PARIS_TIMING_ONLY_STAGE_ONCE // wrong results: prices the per-projection staging (tools/README.md)
            if(p == 0u)
#endif
            stage_box(g, box, lds, wave, 4u, lane);
#ifdef PARIS_TIMING_ONLY_NO_BARRIERS
            if(p == 0u)
#endif
            __syncthreads();
            // ONE code path adds a projection, and every lane takes it -- the lanes beyond the volume's edge with the columns of the
            // last lane inside (their sums are never stored). With several paths (or a per-lane bypass) updating the TZ x VX sums the
            // register allocator keeps an "in" and an "out" set of them, copies one onto the other at the end of every projection and
            // spills around them: -8 % (timing-only build with the all-valid path alone, profiles/r02_ab_fused_steps.txt). What
            // differs between interior tiles, boundary tiles and taps that left the staged box is confined to how a tap is FETCHED
            // (wave-uniform `mode`, below); the sums see one addition per slice in one place.
            {
                Column col[VX];
                bool all_fast = true, all_inside = true, all_none = true;
#pragma unroll
                for(int j = 0; j < VX; ++j)
                {
                    col[j] = make_column<FD>(g, box, g.k_off + min(k + static_cast<uint32_t>(j), k1), g.l_off + min(l, l1), z_first, z_last);
                    all_fast = all_fast && col[j].fast;
                    all_inside = all_inside && col[j].inside;
                    all_none = all_none && col[j].none;
                }
                // 0: every tap of the wave valid and inside a box with the compile-time stride (interior tiles: most of the field of
                //    view) -- no validity test, no clamp, both rows from one address register;
                // 1: every lane's valid taps inside the box (boundary tiles, wide boxes, a last tile of fewer than TZ slices);
                // 2: some tap may leave the box: per-tap check and the global-memory path;
                // 3: no ray of this projection reaches any column of the wave: every contribution is +0, and the volume holds no -0
                //    (BpParams::skip_invalid): nothing to add. About a tenth of the (wave, projection) pairs of a 2048^3 launch.
                const bool wave_inside = mcount == TZ && g.p_dim_y < (1u << 23) && box.stride == FIXED_STRIDE && __all(all_inside ? 1 : 0) != 0;
#ifdef PARIS_TIMING_ONLY_FORCE_MODE // wrong results: every wave through one mode's code
                const int mode = (void(wave_inside), PARIS_TIMING_ONLY_FORCE_MODE);
#else
                const int mode = (g.skip_invalid != 0u && __all(all_none ? 1 : 0) != 0) ? 3 : wave_inside ? 0 : (__all(all_fast ? 1 : 0) != 0 ? 1 : 2);
#endif
                constexpr int AHEAD = PARIS_FUSED_PIPELINE > 0 ? PARIS_FUSED_PIPELINE : 1;
                // The tile's slices in PARTS runs: each run's contributions are computed by the mode's own straight-line, software-
                // pipelined code (the LDS reads of the next AHEAD slices in flight while a slice is finished) into temporaries, and
                // added to the sums after the branches have joined -- the sums have ONE update site per run.
                constexpr int PARTS = PARIS_FUSED_PARTS <= TZ / 4 ? PARIS_FUSED_PARTS : 1;
                constexpr int ZN = TZ / PARTS;
                auto z_of = [&](int z) { return g.z_base + static_cast<float>(g.m_off + m0 + static_cast<uint32_t>(z)) * g.l_vx_z; }; // :118
                // (the TZ slice coordinates stay in registers: reading them back from LDS per slice to make room for a deeper pipeline
                // cost 5 %, profiles/r02_ab_fused_steps.txt)
                auto run = [&](auto part_tag) {
                    constexpr int Z0 = decltype(part_tag)::value * ZN;
                    float c[ZN][VX];
                    auto contributions = [&](auto mode_tag) {
                        constexpr int MODE = decltype(mode_tag)::value;
                        if constexpr(MODE == 2)
                        {
                            // the rare general path: one slice after the other
#pragma unroll
                            for(int z = 0; z < ZN; ++z)
#pragma unroll
                                for(int j = 0; j < VX; ++j)
                                {
                                    c[z][j] = finish_tap<false>(col[j], fetch_tap_general<FD>(g, box, lds, z_of(Z0 + z), col[j], static_cast<uint32_t>(Z0 + z) < mcount));
                                    asm volatile("" : "+v"(c[z][j])); // anchor: keeps the compiler from gathering all fetches first and holding every tap
                                }
                            return;
                        }
                        Tap ring[AHEAD + 1][VX];
                        auto fetch = [&](int z) {
                            const float z_m = z_of(Z0 + z);
                            const bool live = static_cast<uint32_t>(Z0 + z) < mcount;
#pragma unroll
                            for(int j = 0; j < VX; ++j)
                            {
                                Tap t;
                                if constexpr(MODE == 0)
                                    t = fetch_tap<FD, true, FIXED_STRIDE>(g, box, lds, z_m, col[j]);
                                else if constexpr(MODE == 1)
                                {
                                    t = fetch_tap<FD, false, 0>(g, box, lds, z_m, col[j]);
                                    t.valid = t.valid && live;
                                }
                                else
                                    t = fetch_tap_general<FD>(g, box, lds, z_m, col[j], live);
                                ring[z % (AHEAD + 1)][j] = t;
                            }
                        };
#pragma unroll
                        for(int z = 0; z < AHEAD && z < ZN; ++z)
                            fetch(z);
#pragma unroll
                        for(int z = 0; z < ZN; ++z)
                        {
                            if(z + AHEAD < ZN)
                                fetch(z + AHEAD);
#pragma unroll
                            for(int j = 0; j < VX; ++j)
                                c[z][j] = finish_tap<MODE == 0>(col[j], ring[z % (AHEAD + 1)][j]);
                        }
                    };
                    if(mode == 3)
                    {
#pragma unroll
                        for(int z = 0; z < ZN; ++z)
#pragma unroll
                            for(int j = 0; j < VX; ++j)
                                c[z][j] = 0.f;
                    }
                    else if(mode == 0)
                        contributions(std::integral_constant<int, 0>{});
#if PARIS_FUSED_THREE_MODES
                    else if(mode == 1)
                        contributions(std::integral_constant<int, 1>{});
#endif
                    else
                        contributions(std::integral_constant<int, 2>{});
#pragma unroll
                    for(int z = 0; z < ZN; ++z)
#pragma unroll
                        for(int j = 0; j < VX; ++j)
                            elem<VX>(acc[Z0 + z], j) += c[z][j];
                };
                run(std::integral_constant<int, 0>{});
                if constexpr(PARTS > 1)
                    run(std::integral_constant<int, 1>{});
                if constexpr(PARTS > 2)
                {
                    run(std::integral_constant<int, 2>{});
                    run(std::integral_constant<int, 3>{});
                }
            }
        }
#pragma unroll
        for(int z = 0; z < TZ; ++z)
            if(active && static_cast<uint32_t>(z) < mcount)
                store_voxels<VX, NT>(vp + z * slice, acc[z], g.store_sc1 != 0u);
    }

    template <int VX, int TZ, bool NT, bool FD>
    void launch_fused(FusedParams& fp, hipStream_t stream)
    {
        BpParams& g = fp.g;
        g.tz = TZ;
        g.ntx = (g.v_dim_x + 63u) / 64u;
        g.nty = (g.v_dim_y + 4u * VX - 1u) / (4u * VX);
        g.ntz = (g.v_dim_z + TZ - 1u) / TZ;
        g.zchunk = 256u / TZ; // order 12: chunks of 256 slices
        uint32_t blocks = grid_blocks(g);
        hipLaunchKernelGGL((bp_fused_kernel<VX, TZ, NT, FD>), dim3(blocks), dim3(256), g.lds_floats * sizeof(float), stream, fp);
    }

    template <int VX, int TZ>
    void launch_fused_flags(FusedParams& fp, bool nt, bool fd, hipStream_t stream)
    {
        if(nt && fd)
            launch_fused<VX, TZ, true, true>(fp, stream);
        else if(nt)
            launch_fused<VX, TZ, true, false>(fp, stream);
        else if(fd)
            launch_fused<VX, TZ, false, true>(fp, stream);
        else
            launch_fused<VX, TZ, false, false>(fp, stream);
    }
}

void paris_hip_bp_launch_fused(const void* fused_params, int vx, int tz, bool nt, bool fd, hipStream_t stream)
{
    FusedParams fp = *static_cast<const FusedParams*>(fused_params);
    if(vx == 1 && tz == 32) // one column per lane, 32 slices deep: half the column setups per voxel-update of <2, 16>
        launch_fused_flags<1, 32>(fp, nt, fd, stream);
    else if(vx == 1) // tz 8 is not built for this width
        launch_fused_flags<1, 16>(fp, nt, fd, stream);
    else if(vx == 2 && tz == 8)
        launch_fused_flags<2, 8>(fp, nt, fd, stream);
    else if(vx == 2)
        launch_fused_flags<2, 16>(fp, nt, fd, stream);
    else if(tz == 8)
        launch_fused_flags<4, 8>(fp, nt, fd, stream);
    else
        launch_fused_flags<4, 16>(fp, nt, fd, stream);
}
