// Cache-policy bits of the volume stream with compiler-tracked buffer instructions (no inline asm): the z-walk tile
// pattern of the backprojection kernel (64 x 16 x 16 tile, 4 waves, XCD-banded order, 2 slices in flight), loads and stores
// through raw buffer builtins with every combination of sc0 / nt / sc1 (aux bits 1 / 2 / 16). The resource covers one
// tile's z range, so the 32-bit offsets stay small whatever the volume size.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int LD, int ST, int TZ>
__global__ void __launch_bounds__(256) tile(float* vol, uint32_t dx, uint32_t dy, uint32_t dz)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t ntx = dx / 64, nty = dy / 16, ntz = dz / TZ;
    const uint32_t per = (ntx * nty * ntz) / 8u;
    uint32_t b = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    const uint32_t bx = b % ntx; b /= ntx;
    const uint32_t by = b % nty; const uint32_t bz = b / nty;
    const uint32_t k = bx * 64 + (lane & 15u) * 4u;
    const uint32_t l = by * 16 + wave * 4 + (lane >> 4);
    const uint32_t slice_bytes = dx * dy * 4u;                       // < 2^32 / TZ for the sizes used here
    float* base = vol + (size_t)bz * TZ * dy * dx;                   // uniform per workgroup
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, slice_bytes * TZ, 0x00020000);
    const uint32_t off = (l * dx + k) * 4u;
    for(uint32_t mm = 0; mm < TZ; mm += 2)
    {
        v4f a = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, mm * slice_bytes, LD));
        v4f c = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, (mm + 1) * slice_bytes, LD));
        a += 1.f; c += 1.f;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, a), r, off, mm * slice_bytes, ST);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, c), r, off, (mm + 1) * slice_bytes, ST);
    }
}

// the kernel's current instructions, for reference on the same box
template <int TZ>
__global__ void __launch_bounds__(256) tile_global(float* vol, uint32_t dx, uint32_t dy, uint32_t dz)
{
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
    if(TZ == 1)
    {
        v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(vp));
        a += 1.f;
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(vp), "v"(a) : "memory");
        return;
    }
    for(uint32_t mm = 0; mm + 1 < TZ; mm += 2)
    {
        v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(vp + mm * slice));
        v4f c = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(vp + (mm + 1) * slice));
        a += 1.f; c += 1.f;
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(vp + mm * slice), "v"(a) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(vp + (mm + 1) * slice), "v"(c) : "memory");
    }
}

// in-flight accesses of a lane within ONE slice: R rows (l, l + 16, ...) of a 64 x 16R x TZ tile per z step
template <int R, int TZ>
__global__ void __launch_bounds__(256) tile_rows(float* vol, uint32_t dx, uint32_t dy, uint32_t dz)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t ntx = dx / 64, nty = dy / (16 * R), ntz = dz / TZ;
    const uint32_t per = (ntx * nty * ntz) / 8u;
    uint32_t b = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    const uint32_t bx = b % ntx; b /= ntx;
    const uint32_t by = b % nty; const uint32_t bz = b / nty;
    const uint32_t k = bx * 64 + (lane & 15u) * 4u;
    const uint32_t l = by * 16 * R + wave * 4 + (lane >> 4);
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)bz * TZ * dy + l) * dx + k;
    for(uint32_t mm = 0; mm < TZ; ++mm)
    {
        v4f a[R];
#pragma unroll
        for(int r = 0; r < R; ++r)
            a[r] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(vp + mm * slice + (size_t)r * 16 * dx));
#pragma unroll
        for(int r = 0; r < R; ++r)
        {
            a[r] += 1.f;
            asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(vp + mm * slice + (size_t)r * 16 * dx), "v"(a[r]) : "memory");
        }
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
    const unsigned grid = (dx / 64) * (dy / 16) * (dz / 16);
    std::vector<Variant> vs;
    vs.push_back({"global nt load / sc1 nt store (kernel today)", [=] { tile_global<16><<<grid, 256>>>(a, dx, dy, dz); }, {}});
#define ADDTZ(TZ) vs.push_back({"global nt / sc1 nt, tile depth " #TZ, [=] { tile_global<TZ><<<(dx / 64) * (dy / 16) * (dz / TZ), 256>>>(a, dx, dy, dz); }, {}})
    ADDTZ(1); ADDTZ(2); ADDTZ(4); ADDTZ(8); ADDTZ(32); ADDTZ(64);
#define ADDR(R, TZ) vs.push_back({"rows in flight " #R ", tile depth " #TZ, [=] { tile_rows<R, TZ><<<(dx / 64) * (dy / (16 * R)) * (dz / TZ), 256>>>(a, dx, dy, dz); }, {}})
    ADDR(1, 16); ADDR(2, 16); ADDR(4, 16); ADDR(2, 8); ADDR(4, 8); ADDR(2, 32); ADDR(4, 4); ADDR(8, 4); ADDR(2, 1); ADDR(4, 1);
#define ADD(LD, ST) vs.push_back({"buffer load aux " #LD " store aux " #ST, [=] { tile<LD, ST, 16><<<grid, 256>>>(a, dx, dy, dz); }, {}})
    ADD(2, 18); ADD(0, 18); ADD(1, 18); ADD(3, 18); ADD(16, 18); ADD(17, 18); ADD(18, 18); ADD(19, 18);
    ADD(2, 2); ADD(2, 16); ADD(2, 19); ADD(2, 17); ADD(2, 3); ADD(2, 0); ADD(18, 19); ADD(19, 19); ADD(0, 0);
    for(auto& v : vs) v.f();
    CK(hipDeviceSynchronize());
    for(int round = 0; round < 7; ++round)
        for(auto& v : vs)
        {
            CK(hipEventRecord(ea)); v.f(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
            float ms; CK(hipEventElapsedTime(&ms, ea, eb)); v.ms.push_back(ms);
        }
    for(auto& v : vs)
    {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-48s median %.3f ms  %.0f GB/s\n", v.name.c_str(), v.ms[3], gb / v.ms[3] * 1e3);
    }
    // every variant adds 1 to every voxel exactly once per launch: check the sum of launches on a sample
    std::vector<float> h(4096);
    CK(hipMemcpy(h.data(), a + n / 2, h.size() * 4, hipMemcpyDeviceToHost));
    const float want = 8.f * vs.size();
    for(float x : h) if(x != want) { printf("MISMATCH %f != %f\n", x, want); return 1; }
    printf("sample check ok (%g)\n", want);
    return 0;
}
