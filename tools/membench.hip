// One microbenchmark for the access pattern of the backprojection's volume stream (in-place update: every voxel is loaded,
// changed and stored once per launch). It replaces the exploration files membench.hip ... membench9.hip of round 1 (their
// output is kept in profiles/r01_membench.txt) and round 2's persistent-sweep experiment (profiles/r02_membench_persist.txt):
// every family of those files is a set of rows here, selected by a substring filter on the row name.
//
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
//   tools/membench [filter] [dz = 256] [dx = 2048] [dy = 2048]        e.g.  tools/membench persist   tools/membench "tile 64x16"
//
// Families (volume dx x dy x dz floats, x fastest; 256-thread workgroups unless noted):
//   linear      one 16-byte load + store per thread, addresses in dispatch order: the ceiling of a sweep with no z reuse
//   tile        the kernel's pattern: a workgroup owns a tile of TX x TY columns and walks TZ slices with IF slices in flight;
//               TX x TY = 64 x 16 (16 lanes along x, four rows per wave) or 256 x 4 (64 lanes along x); workgroup -> tile orders
//               0 (x fastest, then y, z), 1 (z fastest), 5 (XCD k sweeps its own contiguous eighth), 8 (XCD k owns a y band: x, z, y),
//               9 (XCD k owns a y band: x, y, z -- every XCD at the same slices);
//               cache policy plain / nt (nontemporal loads and stores) / ntsc1 (nontemporal loads, write-through nt stores)
//   zwave       a workgroup of NW waves owns NW slices of a 64 x 16 tile, one slice per wave (the slice kernel's pattern)
//   persist     a grid of exactly the resident workgroups; each owns its columns for the whole depth (column state would be
//               computed once per column), XCD k owns a band of rows per pass; optional pacing: after every S slices the
//               workgroups of an XCD wait (bounded) until all of them have finished chunk c - LAG (relaxed counter, no data
//               handed over, so no fences)
// Every row adds 1 to every voxel exactly once per launch (checked on samples at the end).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v4f __attribute__((ext_vector_type(4)));

enum Policy { PLAIN = 0, NT = 1, NTSC1 = 2 };

template <int POL> __device__ __forceinline__ v4f ld(const float* p)
{
    if(POL == PLAIN)
        return *reinterpret_cast<const v4f*>(p);
    return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
}
template <int POL> __device__ __forceinline__ void st(float* p, v4f a)
{
    if(POL == PLAIN)
        *reinterpret_cast<v4f*>(p) = a;
    else if(POL == NT)
        __builtin_nontemporal_store(a, reinterpret_cast<v4f*>(p));
    else
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(a) : "memory");
}

template <int POL> __global__ void __launch_bounds__(256) linear(float* vol)
{
    float* p = vol + ((size_t)blockIdx.x * 256u + threadIdx.x) * 4u;
    v4f a = ld<POL>(p); a += 1.f; st<POL>(p, a);
}

// workgroup -> tile (the mappings of bp_device.h: tile_of_block)
__device__ __forceinline__ void tile_of(uint32_t b, uint32_t order, uint32_t ntx, uint32_t nty, uint32_t ntz, uint32_t& bx, uint32_t& by, uint32_t& bz)
{
    if(order == 1u) { bz = b % ntz; b /= ntz; bx = b % ntx; by = b / ntx; return; }
    if(order == 8u)
    {
        const uint32_t band = nty / 8u, xcd = b % 8u;
        uint32_t r = b / 8u;
        bx = r % ntx; r /= ntx; bz = r % ntz; by = xcd * band + r / ntz;
        return;
    }
    if(order == 10u) // XCD k owns a y band; inside it the z tile runs fastest, then x, then y
    {
        const uint32_t band = nty / 8u, xcd = b % 8u;
        uint32_t r = b / 8u;
        bz = r % ntz; r /= ntz; bx = r % ntx; by = xcd * band + r / ntx;
        return;
    }
    if(order == 11u) // XCD k owns a y band; x fastest, then two z tiles, then y, then the remaining z: a compromise of 8 and 9
    {
        const uint32_t band = nty / 8u, xcd = b % 8u;
        uint32_t r = b / 8u;
        bx = r % ntx; r /= ntx;
        const uint32_t zl = r % 4u; r /= 4u;
        by = xcd * band + r % band; bz = (r / band) * 4u + zl;
        return;
    }
    if(order == 9u) // XCD k owns a y band; inside it x fastest, then y, then z: all XCDs work on the same slices at a time
    {
        const uint32_t band = nty / 8u, xcd = b % 8u;
        uint32_t r = b / 8u;
        bx = r % ntx; r /= ntx; by = xcd * band + r % band; bz = r / band;
        return;
    }
    if(order >= 140u) // the dealt orders of round 3: order = 100 * kernel order (14 .. 17) + z tiles per chunk; y tiles dealt to the XCDs in groups
    {
        const uint32_t grp = 1u << (order / 100u - 14u), zchunk = order % 100u ? order % 100u : ntz;
        const uint32_t band = nty / 8u, xcd = b % 8u;
        uint32_t r = b / 8u;
        bx = r % ntx; r /= ntx;
        const uint32_t zl = r % zchunk; r /= zchunk;
        const uint32_t yb = r % band;
        bz = (r / band) * zchunk + zl;
        by = (yb / grp) * (8u * grp) + xcd * grp + yb % grp;
        return;
    }
    if(order == 5u) { const uint32_t per = (ntx * nty * ntz) / 8u; b = (b % 8u) * per + b / 8u; }
    bx = b % ntx; b /= ntx; by = b % nty; bz = b / nty;
}

// XL lanes along x per wave: tile (4 XL) x (256 / XL); IF slices in flight
template <int XL, int IF, int POL>
__global__ void __launch_bounds__(256) tile(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t tz, uint32_t order)
{
    constexpr uint32_t TX = XL * 4u, TY = 256u / XL;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t bx, by, bz;
    tile_of(blockIdx.x, order, dx / TX, dy / TY, dz / tz, bx, by, bz);
    const uint32_t k = bx * TX + (lane % XL) * 4u;
    const uint32_t l = by * TY + wave * (64u / XL) + lane / XL;
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)bz * tz * dy + l) * dx + k;
    for(uint32_t m = 0; m + IF <= tz; m += IF, vp += IF * slice) // tz is a multiple of IF (main() only builds such rows)
    {
        v4f q[IF];
#pragma unroll
        for(int i = 0; i < IF; ++i)
            q[i] = ld<POL>(vp + (size_t)i * slice);
#pragma unroll
        for(int i = 0; i < IF; ++i)
            q[i] += 1.f;
#pragma unroll
        for(int i = 0; i < IF; ++i)
            st<POL>(vp + (size_t)i * slice, q[i]);
    }
}

// tall tiles: 64 x (16 RG) columns, every lane owns RG row groups (16 rows apart) of each slice; IF slices in flight
template <int RG, int IF, int POL>
__global__ void __launch_bounds__(256) tile_tall(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t tz, uint32_t order)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t bx, by, bz;
    tile_of(blockIdx.x, order, dx / 64u, dy / (16u * RG), dz / tz, bx, by, bz);
    const uint32_t k = bx * 64u + (lane & 15u) * 4u;
    const uint32_t l = by * 16u * RG + wave * 4u + (lane >> 4);
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)bz * tz * dy + l) * dx + k;
    for(uint32_t m = 0; m + IF <= tz; m += IF, vp += IF * slice)
    {
        v4f q[IF][RG];
#pragma unroll
        for(int i = 0; i < IF; ++i)
#pragma unroll
            for(int r = 0; r < RG; ++r)
                q[i][r] = ld<POL>(vp + (size_t)i * slice + (size_t)r * 16u * dx);
#pragma unroll
        for(int i = 0; i < IF; ++i)
#pragma unroll
            for(int r = 0; r < RG; ++r)
            {
                q[i][r] += 1.f;
                st<POL>(vp + (size_t)i * slice + (size_t)r * 16u * dx, q[i][r]);
            }
    }
}

// NW waves, one slice each, of a 64 x 16 tile (4 row groups of 4 rows per wave)
template <int NW, int POL>
__global__ void __launch_bounds__(NW * 64) zwave(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t order)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t bx, by, bz;
    tile_of(blockIdx.x, order, dx / 64u, dy / 16u, dz / NW, bx, by, bz);
    float* vp = vol + ((size_t)(bz * NW + wave) * dy + by * 16u + (lane >> 4)) * dx + bx * 64u + (lane & 15u) * 4u;
    v4f q[4];
#pragma unroll
    for(int r = 0; r < 4; ++r)
        q[r] = ld<POL>(vp + (size_t)r * 4u * dx);
#pragma unroll
    for(int r = 0; r < 4; ++r)
    {
        q[r] += 1.f;
        st<POL>(vp + (size_t)r * 4u * dx, q[r]);
    }
}

template <int XL, int IF, int S, int LAG, int SLEEP>
__global__ void __launch_bounds__(256) persist(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t wpx, unsigned* counters,
                                               unsigned epoch_base)
{
    constexpr uint32_t TX = XL * 4u, TY = 256u / XL;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t xcd = blockIdx.x % 8u, r = blockIdx.x / 8u;
    const uint32_t ntx = dx / TX;
    const uint32_t bx = r % ntx, byl = r / ntx;
    const uint32_t band = (wpx / ntx) * TY; // rows per XCD band
    const uint32_t passes = dy / (8u * band);
    const uint32_t k = bx * TX + (lane % XL) * 4u;
    const uint32_t lrow = byl * TY + wave * (64u / XL) + lane / XL;
    const size_t slice = (size_t)dx * dy;
    unsigned* cnt = counters + xcd * 64u;
    unsigned chunk = 0;
    for(uint32_t pass = 0; pass < passes; ++pass)
    {
        float* vp = vol + (size_t)((pass * 8u + xcd) * band + lrow) * dx + k;
        v4f q[IF];
#pragma unroll
        for(int i = 0; i < IF; ++i)
            q[i] = ld<NTSC1>(vp + (size_t)i * slice);
        for(uint32_t m = 0; m < dz; m += IF)
        {
#pragma unroll
            for(int i = 0; i < IF; ++i)
            {
                v4f a = q[i];
                if(m + IF + i < dz)
                    q[i] = ld<NTSC1>(vp + (size_t)(m + IF + i) * slice);
                a += 1.f;
                st<NTSC1>(vp + (size_t)(m + i) * slice, a);
            }
            if(S > 0 && ((m + IF) % S) == 0)
            {
                ++chunk;
                if(threadIdx.x == 0)
                {
                    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if(chunk > LAG)
                    {
                        const unsigned want = epoch_base + (chunk - LAG) * wpx;
                        for(int spin = 0; spin < (1 << 14); ++spin) // bounded: pacing only, never correctness
                        {
                            if((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0)
                                break;
                            __builtin_amdgcn_s_sleep(SLEEP);
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
}

struct Row { std::string name; std::function<void()> f; std::vector<float> ms; };

int main(int argc, char** argv)
{
    const char* filter = argc > 1 ? argv[1] : "";
    const uint32_t dz = argc > 2 ? atoi(argv[2]) : 256, dx = argc > 3 ? atoi(argv[3]) : 2048, dy = argc > 4 ? atoi(argv[4]) : 2048;
    const size_t n = (size_t)dx * dy * dz;
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const double gb = 2.0 * n * 4 / 1e9;
    std::vector<Row> rows;
    auto add = [&](const std::string& name, std::function<void()> f) { if(strstr(name.c_str(), filter)) rows.push_back({name, f, {}}); };

    add("linear plain", [=] { linear<PLAIN><<<(unsigned)(n / 1024), 256>>>(a); });
    add("linear nt", [=] { linear<NT><<<(unsigned)(n / 1024), 256>>>(a); });
    add("linear ntsc1 (ceiling)", [=] { linear<NTSC1><<<(unsigned)(n / 1024), 256>>>(a); });
    for(uint32_t tz : {2u, 4u, 8u, 16u, 32u, 64u})
        for(uint32_t order : {0u, 1u, 5u, 8u, 9u, 10u, 11u})
        {
            if(dz % tz || tz % 2u || (order != 5u && order != 8u && tz != 8u && tz != 16u))
                continue; // all orders at the two depths the kernel uses, the default order at every depth
            const unsigned g16 = (dx / 64) * (dy / 16) * (dz / tz), g64 = (dx / 256) * (dy / 4) * (dz / tz);
            char nm[128];
            snprintf(nm, sizeof nm, "tile 64x16x%u order %u if2 ntsc1%s", tz, order, tz == 16 && order == 5 ? " (the kernel today)" : "");
            add(nm, [=] { tile<16, 2, NTSC1><<<g16, 256>>>(a, dx, dy, dz, tz, order); });
            snprintf(nm, sizeof nm, "tile 256x4x%u order %u if2 ntsc1", tz, order);
            add(nm, [=] { tile<64, 2, NTSC1><<<g64, 256>>>(a, dx, dy, dz, tz, order); });
        }
    for(uint32_t order : {1400u, 1404u, 1500u, 1504u, 1502u, 1501u, 1508u, 1604u})
        for(uint32_t tz : {16u, 8u})
        {
            const unsigned g16 = (dx / 64) * (dy / 16) * (dz / tz);
            char nm[128];
            snprintf(nm, sizeof nm, "dealt tile 64x16x%u order %u (groups of %u y tiles, %u z tiles per chunk) if2 ntsc1", tz, order / 100u, 1u << (order / 100u - 14u), order % 100u);
            add(nm, [=] { tile<16, 2, NTSC1><<<g16, 256>>>(a, dx, dy, dz, tz, order); });
        }
    {
        const unsigned g = (dx / 64) * (dy / 16) * (dz / 16);
        add("tile 64x16x16 order 5 if1 ntsc1", [=] { tile<16, 1, NTSC1><<<g, 256>>>(a, dx, dy, dz, 16, 5); });
        add("tile 64x16x16 order 5 if4 ntsc1", [=] { tile<16, 4, NTSC1><<<g, 256>>>(a, dx, dy, dz, 16, 5); });
        add("tile 64x16x16 order 5 if2 plain", [=] { tile<16, 2, PLAIN><<<g, 256>>>(a, dx, dy, dz, 16, 5); });
        add("tile 64x16x16 order 5 if2 nt", [=] { tile<16, 2, NT><<<g, 256>>>(a, dx, dy, dz, 16, 5); });
    }
    for(uint32_t order : {5u, 8u})
        for(uint32_t tz : {4u, 8u, 16u})
        {
            char nm[128];
            snprintf(nm, sizeof nm, "tall tile 64x32x%u order %u if2 ntsc1", tz, order);
            add(nm, [=] { tile_tall<2, 2, NTSC1><<<(dx / 64) * (dy / 32) * (dz / tz), 256>>>(a, dx, dy, dz, tz, order); });
            snprintf(nm, sizeof nm, "tall tile 64x32x%u order %u if1 ntsc1", tz, order);
            add(nm, [=] { tile_tall<2, 1, NTSC1><<<(dx / 64) * (dy / 32) * (dz / tz), 256>>>(a, dx, dy, dz, tz, order); });
            snprintf(nm, sizeof nm, "tall tile 64x64x%u order %u if1 ntsc1", tz, order);
            add(nm, [=] { tile_tall<4, 1, NTSC1><<<(dx / 64) * (dy / 64) * (dz / tz), 256>>>(a, dx, dy, dz, tz, order); });
        }
    add("zwave 8 waves order 5 ntsc1", [=] { zwave<8, NTSC1><<<(dx / 64) * (dy / 16) * (dz / 8), 512>>>(a, dx, dy, dz, 5); });
    add("zwave 16 waves order 5 ntsc1", [=] { zwave<16, NTSC1><<<(dx / 64) * (dy / 16) * (dz / 16), 1024>>>(a, dx, dy, dz, 5); });

    unsigned* cnt; CK(hipMalloc(&cnt, 64 * 8 * 64 * 4)); CK(hipMemset(cnt, 0, 64 * 8 * 64 * 4));
    int cblock = 0;
#define PERSIST(XL, IF, S, LAG, SLEEP, WPX) { unsigned* c = cnt + (cblock++) * 8 * 64; unsigned* ep = new unsigned(0);                                 \
        const unsigned chunks = S ? (dy / (8u * ((WPX / (dx / (XL * 4u))) * (256u / XL)))) * (dz / (S ? S : 1)) : 0u;                                   \
        add("persist " #XL " lanes along x, if" #IF ", " #WPX " workgroups per XCD, " + std::string(S ? "paced every " #S " slices lag " #LAG " sleep " #SLEEP : "no pacing"), \
            [=] { persist<XL, IF, S, LAG, SLEEP><<<8 * WPX, 256>>>(a, dx, dy, dz, WPX, c, *ep); *ep += chunks * WPX; }); }
    PERSIST(16, 2, 0, 1, 2, 256); PERSIST(16, 4, 0, 1, 2, 256); PERSIST(64, 2, 0, 1, 2, 256); PERSIST(64, 4, 0, 1, 2, 128); PERSIST(16, 8, 0, 1, 2, 64);
    PERSIST(16, 2, 4, 1, 2, 256); PERSIST(16, 2, 16, 1, 2, 256); PERSIST(16, 2, 32, 2, 32, 256); PERSIST(64, 2, 32, 2, 32, 256);
    PERSIST(64, 4, 32, 4, 64, 128); PERSIST(16, 2, 64, 2, 64, 256);

    if(rows.empty()) { printf("no row matches '%s'\n", filter); return 1; }
    for(auto& r : rows) r.f();
    CK(hipDeviceSynchronize());
    const int rounds = 7;
    for(int round = 0; round < rounds; ++round)
        for(auto& r : rows)
        {
            CK(hipEventRecord(ea)); r.f(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
            float ms; CK(hipEventElapsedTime(&ms, ea, eb)); r.ms.push_back(ms);
        }
    printf("volume %u x %u x %u, %.2f GB moved per launch (load + store)\n", dx, dy, dz, gb);
    for(auto& r : rows)
    {
        std::sort(r.ms.begin(), r.ms.end());
        printf("%-82s median %.3f ms (min %.3f)  %5.0f GB/s\n", r.name.c_str(), r.ms[rounds / 2], r.ms[0], gb / r.ms[rounds / 2] * 1e3);
    }
    std::vector<float> h(4096);
    const float want = (float)((rounds + 1) * rows.size());
    for(size_t at : {(size_t)0, n / 2, n - 4096, n / 3})
    {
        CK(hipMemcpy(h.data(), a + at, h.size() * 4, hipMemcpyDeviceToHost));
        for(float x : h) if(x != want) { printf("MISMATCH %f != %f at %zu\n", x, want, at); return 1; }
    }
    printf("sample check ok (%g)\n", want);
    return 0;
}
