// Host cost of the HIP runtime calls PARIS's loop makes per projection through paris::hip (what is left of the library's 12-13 us):
// each call timed alone, 2000 times, on otherwise idle streams.   hipcc --offload-arch=gfx950 -O2 tools/api_cost.hip -o tools/api_cost
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void tiny(float* p) { if(p != nullptr && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

template <typename F>
static double us_per_call(int n, F&& f)
{
    const auto t0 = std::chrono::steady_clock::now();
    for(int i = 0; i < n; ++i)
        f(i);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main()
{
    const size_t frame = 1u << 20;
    hipStream_t a, b;
    (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    void *h = nullptr, *d = nullptr;
    (void)hipHostMalloc(&h, frame * 16, hipHostMallocDefault);
    (void)hipMalloc(&d, frame * 16);
    std::memset(h, 1, frame * 16);
    hipEvent_t ev[64];
    for(auto& e : ev)
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int n = 2000;
    (void)hipDeviceSynchronize();
    std::printf("hipSetDevice                                   %6.2f us\n", us_per_call(n, [&](int) { (void)hipSetDevice(0); }));
    std::printf("hipGetLastError                                %6.2f us\n", us_per_call(n, [&](int) { (void)hipGetLastError(); }));
    for(size_t bytes : {size_t{1} << 20, size_t{4} << 20})
    {
        const double t = us_per_call(n, [&](int i) { (void)hipMemcpyAsync(static_cast<char*>(d) + (i % 4) * bytes, static_cast<char*>(h) + (i % 4) * bytes, bytes, hipMemcpyHostToDevice, a); });
        (void)hipStreamSynchronize(a);
        std::printf("hipMemcpyAsync H2D %zu MiB pinned (enqueue)        %6.2f us (stream kept busy: the copies queue up)\n", bytes >> 20, t);
    }
    {
        // paced: one copy every 50 us, like the loop's own rhythm (the queue is empty when the next one arrives)
        double sum = 0;
        for(int i = 0; i < 500; ++i)
        {
            const auto t0 = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync(d, h, frame, hipMemcpyHostToDevice, a);
            const auto t1 = std::chrono::steady_clock::now();
            sum += std::chrono::duration<double, std::micro>(t1 - t0).count();
            while(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 50.0) {}
        }
        std::printf("hipMemcpyAsync H2D 1 MiB pinned, one per 50 us   %6.2f us\n", sum / 500);
    }
    (void)hipStreamSynchronize(a);
    std::printf("hipEventRecord                                 %6.2f us\n", us_per_call(n, [&](int i) { (void)hipEventRecord(ev[i % 64], a); }));
    std::printf("hipStreamWaitEvent                             %6.2f us\n", us_per_call(n, [&](int i) { (void)hipStreamWaitEvent(b, ev[i % 64], 0); }));
    (void)hipDeviceSynchronize();
    std::printf("hipEventQuery (complete)                       %6.2f us\n", us_per_call(n, [&](int i) { (void)hipEventQuery(ev[i % 64]); }));
    std::printf("kernel launch (empty grid of 1)                %6.2f us\n", us_per_call(n, [&](int) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, static_cast<float*>(nullptr)); }));
    (void)hipDeviceSynchronize();
    {
        double sum = 0;
        for(int i = 0; i < 500; ++i)
        {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, static_cast<float*>(nullptr));
            const auto t1 = std::chrono::steady_clock::now();
            sum += std::chrono::duration<double, std::micro>(t1 - t0).count();
            while(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 50.0) {}
        }
        std::printf("kernel launch, one per 50 us                     %6.2f us\n", sum / 500);
    }
    {
        const double t = us_per_call(n, [&](int i) { std::memcpy(static_cast<char*>(h) + (i % 16) * frame, static_cast<char*>(h) + ((i + 8) % 16) * frame, frame); });
        std::printf("memcpy of 1 MiB between pinned buffers (the loop's own frame fill) %6.2f us = %.1f GB/s\n", t, frame / t / 1e3);
    }
    (void)hipDeviceSynchronize();
    return 0;
}
