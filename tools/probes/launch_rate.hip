// Launch-rate probe (round 3): T host threads, each with its own stream, each doing M x { launch a 64-workgroup kernel that runs
// ~15 us; wait for the stream } — the skeleton of a one-block host-pointer call.  Calls per second for T = 1, 2, 4, 8, 16, 32:
// what ANY one-block-per-call API can reach on this runtime, whatever the library does inside the call.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/launch_rate.hip -o tools/probes/_bin/launch_rate -lpthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <chrono>
#include <atomic>
__global__ void spin_kernel(float* out, int cycles)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)cycles) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0 && out) out[blockIdx.x] = 1.0f;
}
int main()
{
    float* d; (void)hipMalloc(&d, 1 << 20);
    const int M = 2000;
    for (int kernelsPerCall : { 1, 2 })
        for (int T : { 1, 2, 4, 8, 16, 32 }) {
            std::vector<hipStream_t> st(T);
            for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            std::atomic<int> ready{0}; std::atomic<bool> go{false};
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t]() {
                    for (int i = 0; i < 20; i++) { hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(128), 0, st[t], d + t * 64, 30000); (void)hipStreamSynchronize(st[t]); }
                    ready++;
                    while (!go.load()) std::this_thread::yield();
                    for (int i = 0; i < M; i++) {
                        for (int k = 0; k < kernelsPerCall; k++) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(128), 0, st[t], d + t * 64, 30000 / kernelsPerCall);
                        (void)hipStreamSynchronize(st[t]);
                    }
                });
            while (ready.load() < T) std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            go = true;
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("kernels per call %d, threads %2d: %9.0f calls/s (%.1f us per call per thread), x%.2f of one thread's rate\n", kernelsPerCall, T, T * M / dt, 1e6 * dt / M, 0.0);
            for (auto& s : st) (void)hipStreamDestroy(s);
        }
    return 0;
}
