// Does a taller workgroup (more waves = more rows of the same slice updated together) help the z-walk stream?
// Tile = 64 x (4*NW) x TZ, NW waves per workgroup, XCD-banded order, nontemporal loads, sc1 nt stores, 2 slices in flight.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NW, int TZ>
__global__ void __launch_bounds__(NW * 64) tile(float* vol, uint32_t dx, uint32_t dy, uint32_t dz)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t ntx = dx / 64, nty = dy / (4 * NW), ntz = dz / TZ;
    const uint32_t per = (ntx * nty * ntz) / 8u;
    uint32_t b = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    const uint32_t bx = b % ntx; b /= ntx;
    const uint32_t by = b % nty; const uint32_t bz = b / nty;
    const uint32_t k = bx * 64 + (lane & 15u) * 4u;
    const uint32_t l = by * 4 * NW + wave * 4 + (lane >> 4);
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)bz * TZ * dy + l) * dx + k;
    for(uint32_t mm = 0; mm < TZ; mm += 2)
    {
        v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(vp + mm * slice));
        v4f c = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(vp + (mm + 1) * slice));
        a += 1.f; c += 1.f;
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(vp + mm * slice), "v"(a) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(vp + (mm + 1) * slice), "v"(c) : "memory");
    }
}

hipEvent_t ea, eb;
struct Variant { std::string name; std::function<void()> f; std::vector<float> ms; };

int main()
{
    const uint32_t dx = 2048, dy = 2048, dz = 256;
    const size_t n = (size_t)dx * dy * dz;
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const double gb = 2.0 * n * 4 / 1e9;
    std::vector<Variant> vs;
#define ADD(NW, TZ) vs.push_back({"NW" #NW " tz" #TZ, [=] { tile<NW, TZ><<<(dx / 64) * (dy / (4 * NW)) * (dz / TZ), NW * 64>>>(a, dx, dy, dz); }, {}})
    ADD(4, 16); ADD(8, 16); ADD(16, 16); ADD(4, 8); ADD(8, 8); ADD(16, 8); ADD(2, 16); ADD(8, 32); ADD(16, 4); ADD(16, 2);
    for(auto& v : vs) v.f();
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
        printf("%-12s median %.3f ms  %.0f GB/s\n", v.name.c_str(), v.ms[2], gb / v.ms[2] * 1e3);
    }
    return 0;
}
