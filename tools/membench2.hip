// Access-pattern study for the in-place volume update: which tile shape / order / cache policy streams fastest.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ v4f ld(const float* p)
{
    if(NT) return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return *reinterpret_cast<const v4f*>(p);
}
template <bool NT> __device__ __forceinline__ void st(float* p, v4f v)
{
    if(NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
    else *reinterpret_cast<v4f*>(p) = v;
}

// XL lanes (x4 floats) per row per wave; wave covers 64/XL rows; WG = 4 waves; tile = (4*XL) x (256/XL) columns, tz slices.
// ZFAST: blockIdx.x enumerates z tiles fastest (instead of x tiles).
template <int XL, int UNROLL, bool NT, bool ZFAST>
__global__ void __launch_bounds__(256) rmw_tile(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t tz)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t RW = 64 / XL;
    uint32_t bx, by, bz;
    const uint32_t ntx = dx / (4 * XL), nty = dy / (4 * RW), ntz = dz / tz;
    uint32_t b = blockIdx.x;
    if(ZFAST) { bz = b % ntz; b /= ntz; bx = b % ntx; by = b / ntx; }
    else { bx = b % ntx; b /= ntx; by = b % nty; bz = b / nty; }
    const uint32_t k = bx * 4 * XL + (lane % XL) * 4u;
    const uint32_t l = by * 4 * RW + wave * RW + lane / XL;
    const uint32_t m0 = bz * tz;
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)m0 * dy + l) * dx + k;
    for(uint32_t mm = 0; mm < tz; mm += UNROLL)
    {
        v4f acc[UNROLL];
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) acc[i] = ld<NT>(vp + (mm + i) * slice);
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) acc[i] += 1.f;
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) st<NT>(vp + (mm + i) * slice, acc[i]);
    }
}

// linear: each WG handles CH consecutive float4 chunks of 256 (4 KB each), contiguous
template <int CH, bool NT>
__global__ void __launch_bounds__(256) rmw_lin(float* vol, size_t n4)
{
    size_t base = ((size_t)blockIdx.x * CH) * 256 + threadIdx.x;
    v4f acc[CH];
#pragma unroll
    for(int i = 0; i < CH; ++i) acc[i] = ld<NT>(vol + (base + (size_t)i * 256) * 4);
#pragma unroll
    for(int i = 0; i < CH; ++i) acc[i] += 1.f;
#pragma unroll
    for(int i = 0; i < CH; ++i) st<NT>(vol + (base + (size_t)i * 256) * 4, acc[i]);
}

hipEvent_t ea, eb;
template <class F> void run(const char* name, double gb, F f)
{
    f(); CK(hipDeviceSynchronize());
    float sum = 0, best = 1e30f;
    for(int r = 0; r < 5; ++r)
    {
        CK(hipEventRecord(ea)); f(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb)); sum += ms; best = ms < best ? ms : best;
    }
    printf("%-44s avg %.3f ms min %.3f ms -> %.0f GB/s\n", name, sum / 5, best, 2 * gb / (sum / 5) * 1e3);
}

#define TILE(XL, UN, NT, ZF, TZ) { char nm[128]; snprintf(nm, 128, "tile XL%d un%d nt%d zfast%d tz%d", XL, UN, NT, ZF, TZ); \
    unsigned nb = (dx / (4 * XL)) * (dy / (256 / XL)) * (dz / TZ); \
    run(nm, gb, [&] { rmw_tile<XL, UN, NT, ZF><<<nb, 256>>>(a, dx, dy, dz, TZ); }); }

int main()
{
    const uint32_t dx = 2048, dy = 2048, dz = 256;
    const size_t n = (size_t)dx * dy * dz;
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const double gb = n * 4 / 1e9;
    run("lin CH1", gb, [&] { rmw_lin<1, false><<<(unsigned)(n / 4 / 256), 256>>>(a, n / 4); });
    run("lin CH1 nt", gb, [&] { rmw_lin<1, true><<<(unsigned)(n / 4 / 256), 256>>>(a, n / 4); });
    run("lin CH4", gb, [&] { rmw_lin<4, false><<<(unsigned)(n / 4 / 256 / 4), 256>>>(a, n / 4); });
    run("lin CH4 nt", gb, [&] { rmw_lin<4, true><<<(unsigned)(n / 4 / 256 / 4), 256>>>(a, n / 4); });
    run("lin CH8", gb, [&] { rmw_lin<8, false><<<(unsigned)(n / 4 / 256 / 8), 256>>>(a, n / 4); });
    TILE(16, 1, false, false, 32) TILE(16, 1, true, false, 32) TILE(16, 4, false, false, 32) TILE(16, 4, true, false, 32)
    TILE(32, 1, false, false, 32) TILE(32, 4, false, false, 32) TILE(32, 4, true, false, 32)
    TILE(64, 1, false, false, 32) TILE(64, 4, false, false, 32) TILE(64, 4, true, false, 32)
    TILE(16, 4, false, true, 32) TILE(64, 4, false, true, 32) TILE(16, 1, false, true, 8)
    TILE(16, 1, false, false, 1) TILE(64, 1, false, false, 1) TILE(16, 1, true, false, 1) TILE(64, 1, true, false, 1)
    TILE(16, 2, false, false, 2) TILE(64, 2, false, false, 2) TILE(64, 4, false, false, 4) TILE(16, 4, false, false, 4) TILE(16, 4, false, false, 8)
    return 0;
}
