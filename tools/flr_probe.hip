// Exhaustive check, all 2^32 fp32 bit patterns, of the two single-instruction forms the fast path would use:
//   v_cvt_flr_i32_f32(v) == (int)floorf(v)        for finite |v| < 2^31
//   v_fract_f32(v)       == v - floorf(v)         for 0 <= v < 2^24  (the valid taps)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)

__global__ void probe(unsigned long long* bad_flr, unsigned long long* bad_fract, unsigned long long* checked)
{
    const unsigned tid = blockIdx.x * 256u + threadIdx.x; // 2^24 threads x 256 patterns
    unsigned long long bf = 0, bfr = 0, n = 0;
    for(unsigned j = 0; j < 256u; ++j)
    {
        const unsigned bits = tid * 256u + j;
        const float v = __uint_as_float(bits);
        if(v == v && fabsf(v) < 2147483648.f)
        {
            int i;
            asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(v));
            if(i != static_cast<int>(floorf(v)))
                ++bf;
            ++n;
        }
        if(v >= 0.f && v < 16777216.f)
        {
            const float a = __builtin_amdgcn_fractf(v);
            const float b = v - floorf(v);
            if(__float_as_uint(a) != __float_as_uint(b))
                ++bfr;
        }
    }
    if(bf) atomicAdd(bad_flr, bf);
    if(bfr) atomicAdd(bad_fract, bfr);
    atomicAdd(checked, n);
}

int main()
{
    unsigned long long* d;
    CK(hipMalloc(&d, 24));
    CK(hipMemset(d, 0, 24));
    probe<<<1u << 16, 256>>>(d, d + 1, d + 2);
    unsigned long long h[3];
    CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
    printf("v_cvt_flr_i32_f32 vs (int)floorf: %llu mismatches over %llu finite |v| < 2^31\n", h[0], h[2]);
    printf("v_fract_f32 vs v - floorf(v): %llu mismatches over all 0 <= v < 2^24\n", h[1]);
    return (h[0] || h[1]) ? 1 : 0;
}
