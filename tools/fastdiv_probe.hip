// Where does multiply + 2 FMA division by a constant differ from IEEE division? Prints a histogram by exponent.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#pragma clang fp contract(off)
__device__ __forceinline__ float div_by_constant(float x, float c, float r)
{
    const float q = x * r;
    const float e = __builtin_fmaf(-q, c, x);
    return __builtin_fmaf(e, r, q);
}
__global__ void probe(float c, float r, unsigned long long* hist, unsigned int* examples, unsigned int* n_ex)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    for(uint32_t i = 0; i < 256u; ++i)
    {
        const uint32_t bits = (i << 24) | t;
        const float x = __uint_as_float(bits);
        const float want = x / c, got = div_by_constant(x, c, r);
        if(__float_as_uint(want) != __float_as_uint(got) && !((want != want) && (got != got)))
        {
            atomicAdd(&hist[(bits >> 23) & 0xff], 1ull);
            unsigned int k = atomicAdd(n_ex, 1u);
            if(k < 16) examples[k] = bits;
        }
    }
}
int main(int argc, char** argv)
{
    for(int a = 1; a < argc; ++a)
    {
        float c = strtof(argv[a], nullptr);
        unsigned long long* h; unsigned int *ex, *n;
        hipMalloc(&h, 256 * 8); hipMalloc(&ex, 64); hipMalloc(&n, 4);
        hipMemset(h, 0, 256 * 8); hipMemset(n, 0, 4);
        probe<<<1 << 16, 256>>>(c, 1.f / c, h, ex, n);
        unsigned long long hh[256]; unsigned int e[16], nn;
        hipMemcpy(hh, h, sizeof(hh), hipMemcpyDeviceToHost); hipMemcpy(e, ex, 64, hipMemcpyDeviceToHost); hipMemcpy(&nn, n, 4, hipMemcpyDeviceToHost);
        printf("c = %.9g (r = %.9g): %u mismatches; by biased exponent of x:", c, 1.f / c, nn);
        for(int i = 0; i < 256; ++i) if(hh[i]) printf(" [%d]=%llu", i, hh[i]);
        printf("\n");
        for(unsigned k = 0; k < nn && k < 6; ++k) { float x; memcpy(&x, &e[k], 4); printf("   x = %.9g (0x%08x): x/c = %.9g\n", x, e[k], x / c); }
    }
    return 0;
}
