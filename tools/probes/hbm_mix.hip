// HBM stream rates by read : write mix (one float4 per lane and stream per iteration, grid-stride; 4 in flight per lane measured slower, buffers far larger than the 256 MB Infinity Cache):
// what a kernel that writes twice what it reads (afstft_analysis: 512 B in, 1064 B out per channel-hop) can expect at best.
//   hipcc -O3 --offload-arch=gfx950 hbm_mix.hip -o _bin/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R, int W, bool NT = false> __global__ void k(const float4* __restrict__ in, float4* __restrict__ out, size_t n, float4* sink)
{
    constexpr int U = 1;                                  // float4 per lane per stream in flight
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++) { const float4 t = in[i + u * stride + r * n]; v[u].x += t.x; v[u].y += t.y; v[u].z += t.z; v[u].w += t.w; }
#pragma unroll
        for (int w = 0; w < W; w++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (NT) { typedef float v4f __attribute__((ext_vector_type(4))); v4f t = { v[u].x, v[u].y, v[u].z, v[u].w }; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(&out[i + u * stride + w * n])); }
                else out[i + u * stride + w * n] = v[u];
            }
        if (W == 0) for (int u = 0; u < U; u++) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (W == 0 && acc.x == 1.2345f) *sink = acc;
}
template <int R, int W, bool NT = false> void run(const float4* in, float4* out, size_t n, float4* sink, const char* name)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<R, W, NT>), dim3(256 * 16), dim3(256), 0, 0, in, out, n, sink);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)(R + W) * n * 16;
    printf("%-28s %6.3f ms  %7.1f GB/s  (%.2f of 8 TB/s)\n", name, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0);
}
int main()
{
    const size_t n = (size_t)1 << 26;                 // 2^26 float4 = 1 GiB per stream
    float4 *in, *out, *sink;
    if (hipMalloc(&in, 2 * n * 16) != hipSuccess || hipMalloc(&out, 2 * n * 16) != hipSuccess || hipMalloc(&sink, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(in, 0, 2 * n * 16); (void)hipMemset(out, 0, 2 * n * 16);
    run<1, 0>(in, out, n, sink, "read only");
    run<2, 0>(in, out, n, sink, "read only, two streams");
    run<0, 1>(in, out, n, sink, "write only");
    run<0, 2>(in, out, n, sink, "write only, two streams");
    run<1, 1>(in, out, n, sink, "copy (1 read : 1 write)");
    run<1, 2>(in, out, n, sink, "1 read : 2 writes");
    run<2, 1>(in, out, n, sink, "2 reads : 1 write");
    run<0, 1, true>(in, out, n, sink, "write only, nt stores");
    run<1, 1, true>(in, out, n, sink, "copy, nt stores");
    run<1, 2, true>(in, out, n, sink, "1 read : 2 writes, nt stores");
    return 0;
}
