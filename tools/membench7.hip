// Cache-policy bits on the volume stream of the z-walk tile pattern (64 x 16 x 16 tiles, 2 slices in flight, XCD-banded
// tile order): every combination of {plain, nt, sc1, sc0 sc1, sc1 nt, sc0 sc1 nt} on loads and stores.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int P> __device__ __forceinline__ v4f ld(const float* p)
{
    v4f v;
    if(P == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if(P == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if(P == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if(P == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if(P == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if(P == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int P> __device__ __forceinline__ void st(float* p, v4f v)
{
    if(P == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if(P == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if(P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if(P == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if(P == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if(P == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

template <int PL, int PS>
__global__ void __launch_bounds__(256) tile(float* vol, uint32_t dx, uint32_t dy, uint32_t dz)
{
    constexpr uint32_t TZ = 16;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t ntx = dx / 64, nty = dy / 16, ntz = dz / TZ;
    const uint32_t per = (ntx * nty * ntz) / 8u;
    uint32_t b = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    const uint32_t bx = b % ntx; b /= ntx;
    const uint32_t by = b % nty; const uint32_t bz = b / nty;
    const uint32_t k = bx * 64 + (lane & 15u) * 4u;
    const uint32_t l = by * 16 + wave * 4 + (lane >> 4);
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)bz * TZ * dy + l) * dx + k;
    for(uint32_t mm = 0; mm < TZ; mm += 2)
    {
        v4f a = ld<PL>(vp + mm * slice), c = ld<PL>(vp + (mm + 1) * slice);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a += 1.f; c += 1.f;
        st<PS>(vp + mm * slice, a); st<PS>(vp + (mm + 1) * slice, c);
    }
}

hipEvent_t ea, eb;
struct Variant { std::string name; std::function<void()> f; std::vector<float> ms; };
const char* NAMES[6] = {"plain", "nt", "sc1", "sc0sc1", "sc1nt", "sc0sc1nt"};

template <int PL, int PS> void add(std::vector<Variant>& vs, float* a, uint32_t dx, uint32_t dy, uint32_t dz)
{
    vs.push_back({std::string("load ") + NAMES[PL] + " store " + NAMES[PS], [=] { tile<PL, PS><<<(dx / 64) * (dy / 16) * (dz / 16), 256>>>(a, dx, dy, dz); }, {}});
}
template <int PL> void add_row(std::vector<Variant>& vs, float* a, uint32_t dx, uint32_t dy, uint32_t dz)
{
    add<PL, 0>(vs, a, dx, dy, dz); add<PL, 1>(vs, a, dx, dy, dz); add<PL, 2>(vs, a, dx, dy, dz);
    add<PL, 3>(vs, a, dx, dy, dz); add<PL, 4>(vs, a, dx, dy, dz); add<PL, 5>(vs, a, dx, dy, dz);
}

int main()
{
    const uint32_t dx = 2048, dy = 2048, dz = 256;
    const size_t n = (size_t)dx * dy * dz;
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const double gb = 2.0 * n * 4 / 1e9;
    std::vector<Variant> vs;
    add_row<0>(vs, a, dx, dy, dz); add_row<1>(vs, a, dx, dy, dz); add_row<2>(vs, a, dx, dy, dz);
    add_row<3>(vs, a, dx, dy, dz); add_row<4>(vs, a, dx, dy, dz); add_row<5>(vs, a, dx, dy, dz);
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
        printf("%-36s median %.3f ms  %.0f GB/s\n", v.name.c_str(), v.ms[2], gb / v.ms[2] * 1e3);
    }
    return 0;
}
