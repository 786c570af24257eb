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

template <int XL, int NW, bool NT> void sweep(float* a, uint32_t dx, uint32_t dy, uint32_t dz)
{
    const double gb = 2.0 * dx * dy * dz * 4 / 1e9;
    for(int order : {0, 1, 5})
    {
        const unsigned nb = (dx / (4 * XL)) * (dy / (64 / XL)) * (dz / NW);
        const float ms = run([&] { zwave<XL, NW, NT><<<nb, NW * 64>>>(a, dx, dy, dz, order); });
        printf("zwave XL%d rows/wave %d NW%d nt%d order%d  %.3f ms  %.0f GB/s\n", XL, 64 / XL, NW, (int)NT, order, ms, gb / ms * 1e3);
    }
}

int main()
{
    const uint32_t dx = 2048, dy = 2048, dz = 256;
    const size_t n = (size_t)dx * dy * dz;
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    sweep<16, 16, false>(a, dx, dy, dz); sweep<16, 16, true>(a, dx, dy, dz);
    sweep<16, 8, false>(a, dx, dy, dz);  sweep<16, 8, true>(a, dx, dy, dz);
    sweep<16, 4, false>(a, dx, dy, dz);  sweep<16, 4, true>(a, dx, dy, dz);
    sweep<64, 16, false>(a, dx, dy, dz); sweep<64, 16, true>(a, dx, dy, dz);
    sweep<64, 8, true>(a, dx, dy, dz);   sweep<32, 16, true>(a, dx, dy, dz);
    sweep<32, 8, true>(a, dx, dy, dz);
    return 0;
}
