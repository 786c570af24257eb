// Sweep of z-walk patterns usable by the backprojection kernel: piece width (XL), unroll, tile depth, block order,
// cache policy (nt), and a non-power-of-two volume to expose DRAM bank aliasing of the 16 MiB slice stride.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ v4f ld(const float* p)
{ if(NT) return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return *reinterpret_cast<const v4f*>(p); }
template <bool NT> __device__ __forceinline__ void st(float* p, v4f v)
{ if(NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); else *reinterpret_cast<v4f*>(p) = v; }

// order 0: x tiles fastest, then y, then z;  1: z tiles fastest, then x, then y;  2: x fastest, then z, then y
template <int XL, int UNROLL, bool NT>
__global__ void __launch_bounds__(256) tile(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t tz, int order)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t RW = 64 / XL;
    const uint32_t ntx = dx / (4 * XL), nty = dy / (4 * RW), ntz = dz / tz;
    uint32_t b = blockIdx.x, bx, by, bz;
    if(order == 0) { bx = b % ntx; b /= ntx; by = b % nty; bz = b / nty; }
    else if(order == 1) { bz = b % ntz; b /= ntz; bx = b % ntx; by = b / ntx; }
    else if(order == 2) { bx = b % ntx; b /= ntx; bz = b % ntz; by = b / ntz; }
    else if(order == 3) { bx = b % ntx; b /= ntx; by = b % nty; bz = b / nty; bx = (bx + by) % ntx; } // x fastest, skewed by y
    else if(order == 4) { by = b % nty; b /= nty; bx = b % ntx; bz = b / ntx; }                      // y fastest
    else if(order == 5) { const uint32_t per = (ntx * nty * ntz) / 8u; const uint32_t t = (b % 8u) * per + b / 8u; // one band per XCD
                          bx = t % ntx; by = (t / ntx) % nty; bz = t / (ntx * nty); }
    else if(order == 6) { bx = b % ntx; b /= ntx; by = b % nty; bz = b / nty; bx = (bx + by + bz) % ntx; by = (by + 3u * bz) % nty; } // 6: skew x by y+z
    else if(order == 7) { // z layer major; inside a layer XCD k owns the contiguous y band k, x fastest
        const uint32_t layer = ntx * nty; bz = b / layer; uint32_t r = b % layer; const uint32_t xcd = r % 8u; r /= 8u;
        const uint32_t band = nty / 8u; bx = r % ntx; by = xcd * band + r / ntx; }
    else { // 8: y band per XCD over the whole slab: XCD k owns y band k for all z; inside: x fastest, then z tile, then y
        const uint32_t xcd = b % 8u; uint32_t r = b / 8u; const uint32_t band = nty / 8u;
        bx = r % ntx; r /= ntx; bz = r % ntz; by = xcd * band + r / ntz; }
    const uint32_t k = bx * 4 * XL + (lane % XL) * 4u;
    const uint32_t l = by * 4 * RW + wave * RW + lane / XL;
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)bz * tz * dy + l) * dx + k;
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

hipEvent_t ea, eb;
template <class F> float run(F f)
{
    f(); CK(hipDeviceSynchronize());
    float sum = 0;
    for(int r = 0; r < 4; ++r)
    {
        CK(hipEventRecord(ea)); f(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb)); sum += ms;
    }
    return sum / 4;
}

template <int XL, int UN, bool NT>
void sweep(float* a, uint32_t dx, uint32_t dy, uint32_t dz)
{
    const double gb = 2.0 * dx * dy * dz * 4 / 1e9;
    for(int order : {5, 7, 8})
        for(uint32_t tz : {8u, 16u})
        {
            if(tz % UN) continue;
            const unsigned nb = (dx / (4 * XL)) * (dy / (256 / XL)) * (dz / tz);
            const float ms = run([&] { tile<XL, UN, NT><<<nb, 256>>>(a, dx, dy, dz, tz, order); });
            printf("vol %ux%ux%u XL%d un%d nt%d order%d tz%-2u  %.3f ms  %.0f GB/s\n", dx, dy, dz, XL, UN, (int)NT, order, tz, ms, gb / ms * 1e3);
        }
}

int main()
{
    const size_t n = (size_t)2112 * 2048 * 256 + 1024; // enough for both shapes
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    sweep<16, 2, true>(a, 2048, 2048, 256);
    sweep<16, 1, true>(a, 2048, 2048, 256);
    sweep<16, 2, true>(a, 2048, 2048, 256);
    sweep<16, 1, true>(a, 2048, 2048, 256);
    return 0;
}
