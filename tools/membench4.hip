// Isolating what makes the multi-slice update slower than the linear one: number of concurrent address streams
// (with ONE load + ONE store per thread), read-only vs write-only z-walks, and occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v4f __attribute__((ext_vector_type(4)));

// n4 float4 elements, split into S equal regions; block b works on region b % S, chunk b / S. One load, one store.
__global__ void __launch_bounds__(256) rmw_streams(float* vol, size_t n4, uint32_t S)
{
    const size_t region = blockIdx.x % S, chunk = blockIdx.x / S;
    const size_t i = region * (n4 / S) + chunk * 256 + threadIdx.x; // < n4 because gridDim.x == n4/256 and S | n4/256
    v4f v = *reinterpret_cast<const v4f*>(vol + i * 4);
    v += 1.f;
    *reinterpret_cast<v4f*>(vol + i * 4) = v;
}

// MODE 0: rmw, 1: read only (sum kept alive), 2: write only
template <int XL, int UNROLL, int MODE>
__global__ void __launch_bounds__(256) tile(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t tz, float* sink)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t RW = 64 / XL;
    const uint32_t ntx = dx / (4 * XL), nty = dy / (4 * RW);
    uint32_t b = blockIdx.x;
    const uint32_t bx = b % ntx; b /= ntx;
    const uint32_t by = b % nty; const uint32_t bz = b / nty;
    const uint32_t k = bx * 4 * XL + (lane % XL) * 4u;
    const uint32_t l = by * 4 * RW + wave * RW + lane / XL;
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)bz * tz * dy + l) * dx + k;
    v4f tot = 0.f;
    for(uint32_t mm = 0; mm < tz; mm += UNROLL)
    {
        v4f acc[UNROLL];
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) acc[i] = MODE == 2 ? v4f(1.f) : *reinterpret_cast<const v4f*>(vp + (mm + i) * slice);
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) acc[i] += 1.f;
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) { if(MODE == 1) tot += acc[i]; else *reinterpret_cast<v4f*>(vp + (mm + i) * slice) = acc[i]; }
    }
    if(MODE == 1 && tot.x == 123.456f) sink[0] = tot.y;
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
    printf("%-52s avg %.3f ms min %.3f ms -> %.0f GB/s\n", name, sum / 5, best, gb / (sum / 5) * 1e3);
}

int main()
{
    const uint32_t dx = 2048, dy = 2048, dz = 256;
    const size_t n = (size_t)dx * dy * dz;
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    float* sink; CK(hipMalloc(&sink, 64));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const double gb = n * 4 / 1e9;
    const size_t n4 = n / 4;
    for(uint32_t S : {1u, 2u, 4u, 8u, 32u, 256u})
    {
        char nm[128]; snprintf(nm, 128, "1 load + 1 store per thread, %u concurrent streams", S);
        run(nm, 2 * gb, [&] { rmw_streams<<<(unsigned)(n4 / 256), 256>>>(a, n4, S); });
    }
    const unsigned nb32 = (dx / 64) * (dy / 16) * (dz / 32), nb1 = (dx / 64) * (dy / 16) * dz;
    run("tile XL16 tz32 un4 rmw", 2 * gb, [&] { tile<16, 4, 0><<<nb32, 256>>>(a, dx, dy, dz, 32, sink); });
    run("tile XL16 tz32 un4 read-only", gb, [&] { tile<16, 4, 1><<<nb32, 256>>>(a, dx, dy, dz, 32, sink); });
    run("tile XL16 tz32 un4 write-only", gb, [&] { tile<16, 4, 2><<<nb32, 256>>>(a, dx, dy, dz, 32, sink); });
    run("tile XL16 tz1 rmw", 2 * gb, [&] { tile<16, 1, 0><<<nb1, 256>>>(a, dx, dy, dz, 1, sink); });
    run("tile XL16 tz1 read-only", gb, [&] { tile<16, 1, 1><<<nb1, 256>>>(a, dx, dy, dz, 1, sink); });
    run("tile XL16 tz1 write-only", gb, [&] { tile<16, 1, 2><<<nb1, 256>>>(a, dx, dy, dz, 1, sink); });
    const unsigned nb64_1 = (dx / 256) * (dy / 4) * dz, nb64_32 = (dx / 256) * (dy / 4) * (dz / 32);
    run("tile XL64 tz1 rmw", 2 * gb, [&] { tile<64, 1, 0><<<nb64_1, 256>>>(a, dx, dy, dz, 1, sink); });
    run("tile XL64 tz1 read-only", gb, [&] { tile<64, 1, 1><<<nb64_1, 256>>>(a, dx, dy, dz, 1, sink); });
    run("tile XL64 tz1 write-only", gb, [&] { tile<64, 1, 2><<<nb64_1, 256>>>(a, dx, dy, dz, 1, sink); });
    run("tile XL64 tz32 un4 read-only", gb, [&] { tile<64, 4, 1><<<nb64_32, 256>>>(a, dx, dy, dz, 32, sink); });
    run("tile XL64 tz32 un4 write-only", gb, [&] { tile<64, 4, 2><<<nb64_32, 256>>>(a, dx, dy, dz, 32, sink); });
    return 0;
}
