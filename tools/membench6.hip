// "z across waves": a workgroup of NW waves updates a 64 x RY x NW tile, wave w owning slice w, so every thread issues
// exactly one load and one store (the structure of the fastest linear sweep) while the (x,y) columns are shared by NW
// slices. Compared against the z-walk-in-time tile of bp_tile_kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ v4f ld(const float* p)
{ if(NT) return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return *reinterpret_cast<const v4f*>(p); }
template <bool NT> __device__ __forceinline__ void st(float* p, v4f v)
{ if(NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); else *reinterpret_cast<v4f*>(p) = v; }

// XL lanes per row (x4 floats), wave covers 64/XL rows of ONE slice; NW waves = NW consecutive slices.
// order 0: x,y,z  1: z fastest  5: XCD bands
template <int XL, int NW, bool NT>
__global__ void __launch_bounds__(NW * 64) zwave(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, int order)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t RW = 64 / XL;
    const uint32_t ntx = dx / (4 * XL), nty = dy / RW, ntz = dz / NW;
    const uint32_t total = ntx * nty * ntz;
    uint32_t b = blockIdx.x, bx, by, bz;
    if(order == 1) { bz = b % ntz; b /= ntz; bx = b % ntx; by = b / ntx; }
    else
    {
        if(order == 5) { const uint32_t per = total / 8u; b = (b % 8u) * per + b / 8u; }
        bx = b % ntx; b /= ntx; by = b % nty; bz = b / nty;
    }
    const uint32_t k = bx * 4 * XL + (lane % XL) * 4u;
    const uint32_t l = by * RW + lane / XL;
    const uint32_t m = bz * NW + wave;
    float* p = vol + ((size_t)m * dy + l) * dx + k;
    v4f v = ld<NT>(p);
    v += 1.f;
    st<NT>(p, v);
}

// zwave with RPL row groups per lane: tile 64 x (4*RPL) x NW; BATCH: all RPL loads first, then the stores
template <int NW, int RPL, bool BATCH>
__global__ void __launch_bounds__(NW * 64) zwave_rows(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, int order)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t ntx = dx / 64, nty = dy / (4 * RPL), ntz = dz / NW;
    const uint32_t total = ntx * nty * ntz;
    uint32_t b = blockIdx.x, bx, by, bz;
    if(order == 1) { bz = b % ntz; b /= ntz; bx = b % ntx; by = b / ntx; }
    else
    {
        if(order == 5) { const uint32_t per = total / 8u; b = (b % 8u) * per + b / 8u; }
        bx = b % ntx; b /= ntx; by = b % nty; bz = b / nty;
    }
    const uint32_t k = bx * 64 + (lane & 15u) * 4u;
    const uint32_t l = by * 4 * RPL + (lane >> 4);
    const uint32_t m = bz * NW + wave;
    float* p = vol + ((size_t)m * dy + l) * dx + k;
    if(BATCH)
    {
        v4f v[RPL];
#pragma unroll
        for(int r = 0; r < RPL; ++r) v[r] = ld<true>(p + (size_t)r * 4 * dx);
#pragma unroll
        for(int r = 0; r < RPL; ++r) v[r] += 1.f;
#pragma unroll
        for(int r = 0; r < RPL; ++r) st<true>(p + (size_t)r * 4 * dx, v[r]);
    }
    else
    {
#pragma unroll
        for(int r = 0; r < RPL; ++r) { v4f v = ld<true>(p + (size_t)r * 4 * dx); v += 1.f; st<true>(p + (size_t)r * 4 * dx, v); }
    }
}

#include <algorithm>
#include <functional>
#include <string>
#include <vector>
hipEvent_t ea, eb;
struct Variant { std::string name; std::function<void()> f; std::vector<float> ms; };

int main()
{
    const uint32_t dx = 2048, dy = 2048, dz = 256;
    const size_t n = (size_t)dx * dy * dz;
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const double gb = 2.0 * dx * dy * dz * 4 / 1e9;
    std::vector<Variant> vs;
#define ZW(XL, NW, NT, ORD) vs.push_back({"zwave XL" #XL " NW" #NW " nt" #NT " order" #ORD, [=] { zwave<XL, NW, NT><<<(dx / (4 * XL)) * (dy / (64 / XL)) * (dz / NW), NW * 64>>>(a, dx, dy, dz, ORD); }, {}})
#define ZR(NW, RPL, BATCH, ORD) vs.push_back({"zwave_rows NW" #NW " RPL" #RPL " batch" #BATCH " order" #ORD, [=] { zwave_rows<NW, RPL, BATCH><<<(dx / 64) * (dy / (4 * RPL)) * (dz / NW), NW * 64>>>(a, dx, dy, dz, ORD); }, {}})
    ZW(16, 8, true, 5); ZW(16, 8, true, 1); ZW(16, 16, true, 5); ZW(64, 8, true, 5); ZW(32, 8, true, 5);
    ZR(8, 1, false, 5); ZR(8, 1, false, 1);
    ZR(8, 2, false, 5); ZR(8, 2, true, 5); ZR(8, 4, false, 5); ZR(8, 4, true, 5); ZR(8, 4, true, 1);
    ZR(16, 2, true, 5); ZR(16, 4, true, 5); ZR(16, 4, false, 5); ZR(4, 4, true, 5); ZR(4, 4, true, 1);
    for(auto& v : vs) { v.f(); }
    CK(hipDeviceSynchronize());
    for(int round = 0; round < 5; ++round)
        for(auto& v : vs)
        {
            CK(hipEventRecord(ea)); v.f(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
            float ms; CK(hipEventElapsedTime(&ms, ea, eb)); v.ms.push_back(ms);
        }
    for(auto& v : vs)
    {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-44s median %.3f ms (min %.3f max %.3f)  %.0f GB/s\n", v.name.c_str(), v.ms[2], v.ms[0], v.ms[4], gb / v.ms[2] * 1e3);
    }
    return 0;
}
