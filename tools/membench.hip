// Streaming ceilings for the backprojection's access pattern on MI355X: float4 copy, in-place read-modify-write in
// linear order, and in-place read-modify-write in the tile order of bp_tile_kernel (64 x 16 columns x TZ slices per
// workgroup, UNROLL slices in flight per lane). Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)

__global__ void __launch_bounds__(256) copy_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
    for(size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
        b[i] = a[i];
}

__global__ void __launch_bounds__(256) rmw_k(float4* a, size_t n)
{
    for(size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
    {
        float4 v = a[i];
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        a[i] = v;
    }
}

// one float4 per thread, no grid-stride loop
__global__ void __launch_bounds__(256) rmw_flat_k(float4* a, size_t n)
{
    size_t i = blockIdx.x * 256ull + threadIdx.x;
    if(i < n)
    {
        float4 v = a[i];
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        a[i] = v;
    }
}

template <int UNROLL>
__global__ void __launch_bounds__(256) rmw_tile_k(float* vol, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t tz)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t k = blockIdx.x * 64u + (lane & 15u) * 4u;
    const uint32_t l = blockIdx.y * 16u + wave * 4u + (lane >> 4);
    const uint32_t m0 = blockIdx.z * tz;
    if(k >= dx || l >= dy) return;
    const size_t slice = (size_t)dx * dy;
    float* vp = vol + ((size_t)m0 * dy + l) * dx + k;
    for(uint32_t mm = 0; mm < tz; mm += UNROLL)
    {
        float4 acc[UNROLL];
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) acc[i] = *reinterpret_cast<const float4*>(vp + (mm + i) * slice);
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) { acc[i].x += 1.f; acc[i].y += 1.f; acc[i].z += 1.f; acc[i].w += 1.f; }
#pragma unroll
        for(int i = 0; i < UNROLL; ++i) *reinterpret_cast<float4*>(vp + (mm + i) * slice) = acc[i];
    }
}

template <class F>
float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for(int r = 0; r < reps; ++r)
    {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best; sum += ms;
    }
    printf("  avg %.3f ms  min %.3f ms", sum / reps, best);
    return sum / reps;
}

int main()
{
    const uint32_t dx = 2048, dy = 2048, dz = 256;
    const size_t n = (size_t)dx * dy * dz; // floats: 4 GiB
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    const double gb = n * 4 / 1e9;
    for(int grid : {2048, 8192, 65536})
    {
        printf("copy float4 grid %d:", grid);
        float ms = time_ms([&] { copy_k<<<grid, 256>>>((const float4*)a, (float4*)b, n / 4); }, 5);
        printf("  -> %.0f GB/s (R+W)\n", 2 * gb / ms * 1e3);
        printf("rmw  float4 grid %d:", grid);
        ms = time_ms([&] { rmw_k<<<grid, 256>>>((float4*)a, n / 4); }, 5);
        printf("  -> %.0f GB/s (R+W)\n", 2 * gb / ms * 1e3);
    }
    {
        printf("rmw flat (1 float4/thread):");
        float ms = time_ms([&] { rmw_flat_k<<<(unsigned)(n / 4 / 256), 256>>>((float4*)a, n / 4); }, 5);
        printf("  -> %.0f GB/s (R+W)\n", 2 * gb / ms * 1e3);
    }
    for(uint32_t tz : {8u, 16u, 32u, 64u})
    {
        dim3 grid(dx / 64, dy / 16, dz / tz);
        printf("rmw tile tz %u unroll 1:", tz);
        float ms = time_ms([&] { rmw_tile_k<1><<<grid, 256>>>(a, dx, dy, dz, tz); }, 5);
        printf("  -> %.0f GB/s\n", 2 * gb / ms * 1e3);
        printf("rmw tile tz %u unroll 2:", tz);
        ms = time_ms([&] { rmw_tile_k<2><<<grid, 256>>>(a, dx, dy, dz, tz); }, 5);
        printf("  -> %.0f GB/s\n", 2 * gb / ms * 1e3);
        printf("rmw tile tz %u unroll 4:", tz);
        ms = time_ms([&] { rmw_tile_k<4><<<grid, 256>>>(a, dx, dy, dz, tz); }, 5);
        printf("  -> %.0f GB/s\n", 2 * gb / ms * 1e3);
    }
    return 0;
}
