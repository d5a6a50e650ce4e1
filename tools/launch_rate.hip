// tools/launch_rate.hip -- what a kernel launch costs on this box: host time per hipLaunchKernelGGL and GPU time per dependent kernel of a
// stream, for an empty kernel without arguments and with a 1 KB argument block; 1 stream and 4 streams; from 1 host thread and from 4.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/launch_rate tools/launch_rate.hip -lpthread && tools/bin/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct Big { unsigned long long w[128]; };
__global__ void k_empty() {}
__global__ void k_big(Big b, unsigned *out) { if (b.w[5] == 12345ull && out) *out = 1; }
__global__ __launch_bounds__(256) void k_lds(Big b, unsigned *out) { __shared__ unsigned l[6656]; l[threadIdx.x] = (unsigned)b.w[1]; __syncthreads(); if (b.w[5] == 12345ull && out) *out = l[(threadIdx.x + 1) & 255]; }
__global__ __launch_bounds__(256) void k_scratch(Big b, unsigned *out) { // a private array indexed at run time: lives in scratch memory
    volatile unsigned a[64];
    for (int i = 0; i < 64; ++i) a[i] = (unsigned)b.w[i] + threadIdx.x;
    if (b.w[5] == 12345ull && out) *out = a[b.w[6] & 63];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int N = 2000;
    hipStream_t s[4];
    for (auto &x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    Big b{}; 
    for (int rep = 0; rep < 2; ++rep) {
        for (int big = 0; big < 2; ++big) {
            (void)hipDeviceSynchronize();
            double t0 = now();
            for (int i = 0; i < N; ++i) { if (big) hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, s[0], b, (unsigned *)nullptr); else hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, s[0]); }
            double t1 = now();
            (void)hipDeviceSynchronize();
            double t2 = now();
            if (rep) printf("1 stream, %s: host %.2f us per launch, until the stream drained %.2f us per kernel\n", big ? "1 KB of arguments" : "no arguments", (t1 - t0) * 1e6 / N, (t2 - t0) * 1e6 / N);
        }
        for (int kind = 0; kind < 2; ++kind) {
            (void)hipDeviceSynchronize();
            double t0 = now();
            for (int i = 0; i < N; ++i) { if (kind) hipLaunchKernelGGL(k_scratch, dim3(512), dim3(256), 0, s[0], b, (unsigned *)nullptr); else hipLaunchKernelGGL(k_lds, dim3(512), dim3(256), 0, s[0], b, (unsigned *)nullptr); }
            double t1 = now();
            (void)hipDeviceSynchronize();
            double t2 = now();
            if (rep) printf("1 stream, 1 KB of arguments, %s: host %.2f us per launch, drained %.2f us per kernel\n", kind ? "256 B of scratch per thread" : "26 KB of LDS", (t1 - t0) * 1e6 / N, (t2 - t0) * 1e6 / N);
        }
        {
            (void)hipDeviceSynchronize();
            double t0 = now();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, s[i & 3], b, (unsigned *)nullptr);
            double t1 = now();
            (void)hipDeviceSynchronize();
            double t2 = now();
            if (rep) printf("4 streams round-robin from one thread: host %.2f us per launch, drained %.2f us per kernel\n", (t1 - t0) * 1e6 / N, (t2 - t0) * 1e6 / N);
        }
        {
            (void)hipDeviceSynchronize();
            double t0 = now();
            std::vector<std::thread> th;
            for (int k = 0; k < 4; ++k) th.emplace_back([&, k]() { for (int i = 0; i < N / 4; ++i) hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, s[k], b, (unsigned *)nullptr); });
            for (auto &t : th) t.join();
            double t1 = now();
            (void)hipDeviceSynchronize();
            double t2 = now();
            if (rep) printf("4 streams, one thread each: host %.2f us per launch (wall / all launches), drained %.2f us per kernel\n", (t1 - t0) * 1e6 / N, (t2 - t0) * 1e6 / N);
        }
    }
    return 0;
}
