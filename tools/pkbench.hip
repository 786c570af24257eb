// Issue rate of packed fp32 (v_pk_mul_f32 / v_pk_fma_f32) vs plain v_mul_f32 / v_fma_f32 on gfx950: 8 independent
// accumulator chains per lane, 4096 iterations, enough waves to fill every SIMD. Prints lane-operations per second.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, int iters)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const v2f aa = {a, a};
    for(int i = 0; i < iters; ++i)
    {
        if(MODE == 0)
        {
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        }
        else if(MODE == 1)
        {
            asm volatile("v_pk_mul_f32 %0, %8, %0\n v_pk_mul_f32 %1, %8, %1\n v_pk_mul_f32 %2, %8, %2\n v_pk_mul_f32 %3, %8, %3\n"
                         "v_pk_mul_f32 %4, %8, %4\n v_pk_mul_f32 %5, %8, %5\n v_pk_mul_f32 %6, %8, %6\n v_pk_mul_f32 %7, %8, %7"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(aa));
        }
        else if(MODE == 2)
        {
            asm volatile("v_fma_f32 %0, %8, %0, %8\n v_fma_f32 %1, %8, %1, %8\n v_fma_f32 %2, %8, %2, %8\n v_fma_f32 %3, %8, %3, %8\n"
                         "v_fma_f32 %4, %8, %4, %8\n v_fma_f32 %5, %8, %5, %8\n v_fma_f32 %6, %8, %6, %8\n v_fma_f32 %7, %8, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        }
        else
        {
            asm volatile("v_pk_fma_f32 %0, %8, %0, %8\n v_pk_fma_f32 %1, %8, %1, %8\n v_pk_fma_f32 %2, %8, %2, %8\n v_pk_fma_f32 %3, %8, %3, %8\n"
                         "v_pk_fma_f32 %4, %8, %4, %8\n v_pk_fma_f32 %5, %8, %5, %8\n v_pk_fma_f32 %6, %8, %6, %8\n v_pk_fma_f32 %7, %8, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(aa));
        }
    }
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y
              + p6.x + p6.y + p7.x + p7.y;
    if(s == 12345.678f)
        out[0] = s;
}

template <int MODE>
void run(const char* name, float* d, int lanes_per_instr)
{
    const int iters = 4096, blocks = 256 * 8 * 4;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<blocks, 256>>>(d, 0.999f, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k<MODE><<<blocks, 256>>>(d, 0.999f, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double instr = (double)blocks * 256 * iters * 8; // per-lane instructions
    printf("%-14s %.3f ms  %.2f T lane-instr/s  %.2f T lane-ops/s\n", name, ms, instr / ms / 1e9, instr * lanes_per_instr / ms / 1e9);
}

int main()
{
    float* d; CK(hipMalloc(&d, 4));
    run<0>("v_mul_f32", d, 1);
    run<1>("v_pk_mul_f32", d, 2);
    run<2>("v_fma_f32", d, 1);
    run<3>("v_pk_fma_f32", d, 2);
    return 0;
}
