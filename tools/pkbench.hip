// Issue rates of the vector instructions the backprojection loop is made of, on gfx950: 8 independent chains per lane,
// 4096 iterations, enough waves to fill every SIMD. Prints lane-instructions per second (packed ones also lane-operations).
// Findings (profiles/r01_pkbench.txt): the packed fp32 multiply/add issue at half the rate of the plain instructions, so
// SLP-vectorised fp32 code gains nothing and pays v_mov packing; VOP3-encoded instructions issue slower than VOP2 ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while(0)
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(INSTR) INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)
#define OUTS "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
#define POUTS "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, int iters)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const v2f aa = {a, a};
    for(int i = 0; i < iters; ++i)
    {
#define I(n) "v_mul_f32_e32 %" #n ", %8, %" #n "\n"
        if(MODE == 0) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_pk_mul_f32 %" #n ", %8, %" #n "\n"
        if(MODE == 1) asm volatile(REP8(I) : POUTS : "v"(aa));
#undef I
#define I(n) "v_fma_f32 %" #n ", %8, %" #n ", %8\n"
        if(MODE == 2) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_pk_fma_f32 %" #n ", %8, %" #n ", %8\n"
        if(MODE == 3) asm volatile(REP8(I) : POUTS : "v"(aa));
#undef I
#define I(n) "v_mul_f32_e64 %" #n ", %8, %" #n "\n"
        if(MODE == 4) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_add_f32_e32 %" #n ", %8, %" #n "\n"
        if(MODE == 5) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_pk_add_f32 %" #n ", %8, %" #n "\n"
        if(MODE == 6) asm volatile(REP8(I) : POUTS : "v"(aa));
#undef I
#define I(n) "v_fmac_f32_e32 %" #n ", %8, %8\n"
        if(MODE == 7) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_mad_i32_i24 %" #n ", %8, %" #n ", %8\n"
        if(MODE == 8) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_med3_i32 %" #n ", %8, %" #n ", 0\n"
        if(MODE == 9) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_floor_f32_e32 %" #n ", %" #n "\n"
        if(MODE == 10) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_cvt_i32_f32_e32 %" #n ", %" #n "\n"
        if(MODE == 11) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_cndmask_b32_e32 %" #n ", %8, %" #n ", vcc\n"
        if(MODE == 12) asm volatile(REP8(I) : OUTS : "v"(a) : "vcc");
#undef I
#define I(n) "v_cndmask_b32_e64 %" #n ", %8, %" #n ", s[10:11]\n"
        if(MODE == 13) asm volatile(REP8(I) : OUTS : "v"(a) : "s10", "s11");
#undef I
#define I(n) "v_add_u32_e32 %" #n ", %8, %" #n "\n"
        if(MODE == 14) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_mov_b32_e32 %" #n ", %8\n"
        if(MODE == 15) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_mul_f32_e32 %" #n ", s20, %" #n "\n"
        if(MODE == 16) asm volatile(REP8(I) : OUTS : "v"(a) : "s20");
#undef I
#define I(n) "v_mul_f32_e32 %" #n ", 0.5, %" #n "\n"
        if(MODE == 17) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_mul_f32_e32 %" #n ", 0x3f7fbe77, %" #n "\n"
        if(MODE == 18) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_add_f32_e32 %" #n ", s20, %" #n "\n"
        if(MODE == 19) asm volatile(REP8(I) : OUTS : "v"(a) : "s20");
#undef I
#define I(n) "v_fma_f32 %" #n ", s20, %" #n ", %8\n"
        if(MODE == 20) asm volatile(REP8(I) : OUTS : "v"(a) : "s20");
#undef I
#define I(n) "v_cndmask_b32_e64 %" #n ", %8, %" #n ", vcc\n"
        if(MODE == 21) asm volatile(REP8(I) : OUTS : "v"(a) : "vcc");
#undef I
#define I(n) "v_cmp_gt_u32_e64 s[10:11], %8, %" #n "\n"
        if(MODE == 22) asm volatile(REP8(I) : OUTS : "v"(a) : "s10", "s11");
#undef I
#define I(n) "v_cmp_gt_u32_e32 vcc, %8, %" #n "\n"
        if(MODE == 23) asm volatile(REP8(I) : OUTS : "v"(a) : "vcc");
#undef I
#define I(n) "v_sub_f32_e32 %" #n ", %8, %" #n "\n"
        if(MODE == 24) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
#define I(n) "v_mul_f32_e32 %" #n ", %" #n ", %" #n "\n"
        if(MODE == 25) asm volatile(REP8(I) : OUTS : "v"(a));
#undef I
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
    printf("%-22s %.3f ms  %6.2f T lane-instr/s  %6.2f T lane-ops/s\n", name, ms, instr / ms / 1e9, instr * lanes_per_instr / ms / 1e9);
}

int main()
{
    float* d; CK(hipMalloc(&d, 4));
    run<0>("v_mul_f32_e32", d, 1);
    run<4>("v_mul_f32_e64", d, 1);
    run<16>("v_mul_f32_e32 sgpr", d, 1);
    run<5>("v_add_f32_e32", d, 1);
    run<1>("v_pk_mul_f32", d, 2);
    run<6>("v_pk_add_f32", d, 2);
    run<7>("v_fmac_f32_e32", d, 1);
    run<2>("v_fma_f32", d, 1);
    run<3>("v_pk_fma_f32", d, 2);
    run<8>("v_mad_i32_i24", d, 1);
    run<9>("v_med3_i32", d, 1);
    run<10>("v_floor_f32_e32", d, 1);
    run<11>("v_cvt_i32_f32_e32", d, 1);
    run<12>("v_cndmask_b32_e32 vcc", d, 1);
    run<13>("v_cndmask_b32_e64 sgpr", d, 1);
    run<14>("v_add_u32_e32", d, 1);
    run<15>("v_mov_b32_e32", d, 1);
    run<17>("v_mul_f32 inline 0.5", d, 1);
    run<18>("v_mul_f32 literal", d, 1);
    run<19>("v_add_f32_e32 sgpr", d, 1);
    run<20>("v_fma_f32 sgpr", d, 1);
    run<21>("v_cndmask_b32_e64 vcc", d, 1);
    run<22>("v_cmp_gt_u32_e64 sgpr", d, 1);
    run<23>("v_cmp_gt_u32_e32 vcc", d, 1);
    run<24>("v_sub_f32_e32", d, 1);
    run<25>("v_mul_f32 x*x", d, 1);
    return 0;
}
