// Why does a z-walk (one workgroup touching slices 16 MiB apart) stream slower than a linear sweep? Experiments on
// slice stride, workgroup size and per-workgroup page count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v4f __attribute__((ext_vector_type(4)));

// generic tile: WG of NW waves; wave covers XL lanes*4 floats in x, 64/XL rows; tile rows = NW*64/XL; tz slices
template <int XL, int NW, int UNROLL>
__global__ void __launch_bounds__(NW * 64) rmw_tile(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t tz)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t RW = 64 / XL;
    const uint32_t ntx = dx / (4 * XL), nty = dy / (NW * RW);
    uint32_t b = blockIdx.x;
    const uint32_t bx = b % ntx; b /= ntx;
    const uint32_t by = b % nty; const uint32_t bz = b / nty;
    const uint32_t k = bx * 4 * XL + (lane % XL) * 4u;
    const uint32_t l = by * NW * RW + wave * RW + lane / XL;
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)bz * tz * dy + l) * dx + k;
    for(uint32_t mm = 0; mm < tz; mm += UNROLL)
    {
        v4f acc[UNROLL];
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) acc[i] = *reinterpret_cast<const v4f*>(vp + (mm + i) * slice);
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) acc[i] += 1.f;
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) *reinterpret_cast<v4f*>(vp + (mm + i) * slice) = acc[i];
    }
}

// The buffer is P planes of stride_f floats. A workgroup updates one 4 KiB chunk in each of CH consecutive planes.
// Block order: wpg consecutive blocks sweep wpg consecutive chunks of one plane group, then the next plane group.
template <int CH>
__global__ void __launch_bounds__(256) rmw_strided(float* vol, size_t stride_f, uint32_t groups, uint32_t wpg)
{
    const uint32_t b = blockIdx.x;
    const uint32_t c_lo = b % wpg;
    const uint32_t t = b / wpg;
    const uint32_t g = t % groups;
    const uint32_t c = (t / groups) * wpg + c_lo; // chunk inside the plane, < stride_f / 1024
    float* p = vol + (size_t)g * CH * stride_f + (size_t)c * 1024 + threadIdx.x * 4;
    v4f acc[CH];
#pragma unroll
    for(int i = 0; i < CH; ++i) acc[i] = *reinterpret_cast<const v4f*>(p + i * stride_f);
#pragma unroll
    for(int i = 0; i < CH; ++i) acc[i] += 1.f;
#pragma unroll
    for(int i = 0; i < CH; ++i) *reinterpret_cast<v4f*>(p + i * stride_f) = acc[i];
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
    printf("%-52s avg %.3f ms min %.3f ms -> %.0f GB/s\n", name, sum / 5, best, 2 * gb / (sum / 5) * 1e3);
}
#define TILE(XL, NW, UN, DX, DY, DZ, TZ) { char nm[128]; snprintf(nm, 128, "tile XL%d NW%d un%d vol %dx%dx%d tz%d", XL, NW, UN, DX, DY, DZ, TZ); \
    unsigned nb = ((DX) / (4 * XL)) * ((DY) / (NW * 64 / XL)) * ((DZ) / (TZ)); \
    run(nm, gb, [&] { rmw_tile<XL, NW, UN><<<nb, NW * 64>>>(a, DX, DY, DZ, TZ); }); }

int main()
{
    const size_t n = (size_t)2048 * 2048 * 256;
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const double gb = n * 4 / 1e9;
    // E1/E2: slice stride 128 KiB, 512 KiB, 2 MiB, 16 MiB with the same tile kernel
    TILE(16, 4, 4, 2048, 16, 32768, 32)
    TILE(16, 4, 4, 2048, 64, 8192, 32)
    TILE(16, 4, 4, 2048, 256, 2048, 32)
    TILE(16, 4, 4, 2048, 2048, 256, 32)
    TILE(16, 4, 1, 2048, 16, 32768, 32)
    TILE(16, 4, 1, 2048, 2048, 256, 32)
    // E3: bigger workgroups (16 waves): 64x64 columns, and 256x16
    TILE(16, 16, 4, 2048, 2048, 256, 32)
    TILE(16, 16, 1, 2048, 2048, 256, 32)
    TILE(64, 16, 4, 2048, 2048, 256, 32)
    TILE(64, 16, 1, 2048, 2048, 256, 32)
    TILE(64, 16, 1, 2048, 2048, 256, 8)
    TILE(64, 8, 1, 2048, 2048, 256, 8)
    // E5: linear 4 KiB chunks, CH planes 16 MiB apart; wpg = how many consecutive chunks of a plane group run together
    {
        const size_t stride_f = (size_t)4 << 20;           // 16 MiB planes
        const uint32_t planes = (uint32_t)(n / stride_f);  // 256
        const uint32_t cpp = (uint32_t)(stride_f / 1024);  // 4096 chunks per plane
        for(uint32_t wpg : {1u, 32u, 4096u})
        {
            char nm[128];
            snprintf(nm, 128, "strided CH2 16MiB planes, %u chunks/group-visit", wpg);
            run(nm, gb, [&] { rmw_strided<2><<<planes / 2 * cpp, 256>>>(a, stride_f, planes / 2, wpg); });
            snprintf(nm, 128, "strided CH4 16MiB planes, %u chunks/group-visit", wpg);
            run(nm, gb, [&] { rmw_strided<4><<<planes / 4 * cpp, 256>>>(a, stride_f, planes / 4, wpg); });
            snprintf(nm, 128, "strided CH8 16MiB planes, %u chunks/group-visit", wpg);
            run(nm, gb, [&] { rmw_strided<8><<<planes / 8 * cpp, 256>>>(a, stride_f, planes / 8, wpg); });
        }
    }
    return 0;
}
