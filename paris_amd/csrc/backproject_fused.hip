// Fused multi-projection backprojection kernel (paris_hip_backproject_batch) for gfx950.
//
// A translation unit of its own: its hot loop is bound by LDS / ALU latency at three waves per SIMD, and clang's SLP
// vectoriser turns its fp32 multiplies and adds into v_pk_mul_f32 / v_pk_add_f32, which on gfx950 issue at half the
// rate of the plain instructions (tools/pkbench.hip) and need v_mov packing on top: -fno-slp-vectorize (Makefile) is
// worth +12 % here, while the memory-bound tile kernel in backproject.hip is indifferent to it.
#define PARIS_BP_SINGLE_INSTRUCTION_FLOOR 1
#if (defined(PARIS_TIMING_ONLY_NO_BARRIERS) || defined(PARIS_TIMING_ONLY_STAGE_ONCE)) && !defined(PARIS_HIP_EXPERIMENTS)
#error "the timing-only switches compute WRONG results: they exist in the experiments build only (make EXPERIMENTS=1)"
#endif
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
        vec acc[TZ] = {};
#pragma unroll
        for(int z = 0; z < TZ; ++z)
            if(active && static_cast<uint32_t>(z) < mcount)
                acc[z] = load_voxels<VX, NT>(vp + z * slice);

        const float z_first = g.z_base + static_cast<float>(g.m_off + m0) * g.l_vx_z;
        const float z_last = g.z_base + static_cast<float>(g.m_off + m1) * g.l_vx_z;
        // (Keeping the loop constants of the per-voxel chain in VGPRs -- a plain fp32 multiply / add issues 1.6x faster in
        // isolation when no source is a scalar register, tools/pkbench.hip -- was measured: no gain in this instruction mix.)
        // the detector boxes of all projections of this tile, 16 per wave at once (four waves: FUSED_MAX = 64)
        __shared__ int box_tab[FUSED_MAX * BOX_WORDS];
        if(wave * 16u < fp.n_proj)
            tile_boxes_to_lds(g, fp.sin_phi, fp.cos_phi, wave * 16u, fp.n_proj, k0, k1, l0, l1, m0, m1, lane, g.lds_floats, FIXED_STRIDE, box_tab);
        for(uint32_t p = 0; p < fp.n_proj; ++p)
        {
            g.sin_phi = fp.sin_phi[p];
            g.cos_phi = fp.cos_phi[p];
            g.proj = fp.proj_tab[p];
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
            // EVERY lane takes part: a lane beyond the volume's edge works on the nearest column inside it and its sums are never
            // stored. A per-lane `if(active)` around the body would be a divergent region, in which the compiler must keep the old
            // sums alive for the lanes that sit it out; with only wave-uniform branches left (and the uniform regions not
            // structurized: Makefile) the TZ x VX sums are added in place -- 106 VGPRs and no scratch against 128 and moves.
            {
                Column col[VX];
                bool all_fast = true, all_inside = true, all_none = true;
#pragma unroll
                for(int j = 0; j < VX; ++j)
                {
                    col[j] = make_column<FD>(g, box, g.k_off + min(k + j, k1), g.l_off + min(l, l1), z_first, z_last);
                    all_fast = all_fast && col[j].fast;
                    all_inside = all_inside && col[j].inside;
                    all_none = all_none && col[j].none;
                }
                auto add_projection = [&](auto fast_tag, auto full_tag, auto inside_tag, auto stride_tag) {
                    constexpr bool FAST = decltype(fast_tag)::value;
                    constexpr bool FULL = decltype(full_tag)::value; // whole tile: no per-slice test, one straight block
                    constexpr bool INSIDE = decltype(inside_tag)::value; // every tap valid: no validity test, clamp or select
                    constexpr int CS = decltype(stride_tag)::value;      // the box has the compile-time row stride
                    if constexpr(FAST && FULL && PARIS_FUSED_PIPELINE > 0)
                    {
                        // software pipeline over the slices: the LDS reads of the next PARIS_FUSED_PIPELINE slices are in flight while
                        // slice z is finished
                        constexpr int AHEAD = PARIS_FUSED_PIPELINE;
                        Tap ring[AHEAD + 1][VX];
                        auto fetch = [&](int z) {
                            // (the TZ slice coordinates stay in registers: reading them back from LDS per slice to make room for a deeper
                            // pipeline cost 5 %, profiles/r02_ab_fused_steps.txt)
                            const float z_m = g.z_base + static_cast<float>(g.m_off + m0 + static_cast<uint32_t>(z)) * g.l_vx_z; // :118
#pragma unroll
                            for(int j = 0; j < VX; ++j)
                                ring[z % (AHEAD + 1)][j] = fetch_tap<FD, INSIDE, CS>(g, box, lds, z_m, col[j]);
                        };
#pragma unroll
                        for(int z = 0; z < AHEAD && z < TZ; ++z)
                            fetch(z);
#pragma unroll
                        for(int z = 0; z < TZ; ++z)
                        {
                            if(z + AHEAD < TZ)
                                fetch(z + AHEAD);
#pragma unroll
                            for(int j = 0; j < VX; ++j)
                                elem<VX>(acc[z], j) += finish_tap<INSIDE>(col[j], ring[z % (AHEAD + 1)][j]);
                        }
                        return;
                    }
#pragma unroll
                    for(int z = 0; z < TZ; ++z)
                    {
                        if(FULL || static_cast<uint32_t>(z) < mcount) // uniform; no break, so acc stays in registers
                        {
                            const float z_m = g.z_base + static_cast<float>(g.m_off + m0 + z) * g.l_vx_z; // :118
#pragma unroll
                            for(int j = 0; j < VX; ++j)
                                elem<VX>(acc[z], j) += voxel_contribution<FD, FAST, INSIDE, CS>(g, box, lds, z_m, col[j]);
                        }
                    }
                };
                // Wave-uniform choice of the all-valid path (a wave with one boundary lane takes the fast path for all its lanes:
                // a per-lane branch would run both bodies). Interior tiles -- most of the field of view -- take it.
                const bool wave_inside = mcount == TZ && g.p_dim_y < (1u << 23) && __all(all_inside ? 1 : 0) != 0;
                const bool fixed = box.stride == FIXED_STRIDE; // workgroup-uniform
                using no_stride = std::integral_constant<int, 0>;
                using the_stride = std::integral_constant<int, FIXED_STRIDE>;
                if(g.skip_invalid != 0u && __all(all_none ? 1 : 0) != 0)
                {
                    // no ray of this projection reaches any column of the wave: every contribution is +0 and the volume holds no -0
                    // (BpParams::skip_invalid) -- nothing to add. About a tenth of the (wave, projection) pairs of a 2048^3 launch.
                }
                else if(wave_inside && fixed)
                    add_projection(std::true_type{}, std::true_type{}, std::true_type{}, the_stride{});
                else if(wave_inside)
                    add_projection(std::true_type{}, std::true_type{}, std::true_type{}, no_stride{});
                else if(all_fast && mcount == TZ && fixed)
                    add_projection(std::true_type{}, std::true_type{}, std::false_type{}, the_stride{});
                else if(all_fast && mcount == TZ)
                    add_projection(std::true_type{}, std::true_type{}, std::false_type{}, no_stride{});
                else if(all_fast)
                    add_projection(std::true_type{}, std::false_type{}, std::false_type{}, no_stride{});
                else
                    add_projection(std::false_type{}, std::false_type{}, std::false_type{}, no_stride{});
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
        settle_order(g);
        g.zchunk = chunk_tiles(g.order, TZ, g.ntz, 32u);
        const uint32_t blocks = static_cast<uint32_t>(grid_blocks(g));
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

// PARIS_HIP_CTX_WARM: a query of one kernel of this translation unit makes the runtime load its code object now
void paris_hip_warm_backproject_fused()
{
    hipFuncAttributes a{};
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>((&bp_fused_kernel<1, 32, true, true>)));
}
