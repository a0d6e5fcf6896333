// Co-run probe (round 3): does a kernel on a second stream slow a vector-bound kernel chip-wide, and by what — the clock?
// Kernel V: independent v_fma_f32 chains (3 waves per SIMD, 6 workgroups of 128 threads per CU); every 64th workgroup stamps
// s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop -> in-kernel clock.  Beside it, on a second stream:
// nothing / an fp32 MFMA loop / a bf16 MFMA loop / a streaming copy, on P workgroups of 128 threads (one per CU for P = 256).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(128, 3) void kV(float* out, unsigned long long* st, int iters)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    const float b = out[0] + 1.0001f, c = out[1] + 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = fmaf(a[i], b, c);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((blockIdx.x & 63) == 0 && threadIdx.x == 0) { st[2 * (blockIdx.x >> 6)] = t1 - t0; st[2 * (blockIdx.x >> 6) + 1] = r1 - r0; }
}
template <int KIND> __global__ __launch_bounds__(128) void kM(float* out, const float4* src, float4* dst, int iters, size_t n4)
{
    if (KIND == 0) {            // fp32 MFMA, 4 independent accumulators, back to back
        f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        const float a = threadIdx.x * 0.01f, b = 1.0f + threadIdx.x * 1e-4f;
        for (int it = 0; it < iters; it++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
        out[(size_t)blockIdx.x * 128 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == 1) {     // bf16 MFMA
        f16v c0 = {0}, c1 = {0};
        bf8 a, b;
        for (int i = 0; i < 8; i++) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(1.0f + i); }
        for (int it = 0; it < iters; it++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
        out[(size_t)blockIdx.x * 128 + threadIdx.x] = c0[0] + c1[1];
    } else {                    // streaming copy, 16 bytes per lane
        for (int it = 0; it < iters; it++)
            for (size_t i = (size_t)blockIdx.x * 128 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 128) dst[i] = src[i];
    }
}
int main(int argc, char** argv)
{
    float *d, *dm; (void)hipMalloc(&d, 64 << 20); (void)hipMemset(d, 0, 64 << 20); (void)hipMalloc(&dm, 64 << 20);
    const size_t n4 = (size_t)(1u << 30) / 16;
    float4 *src, *dst; (void)hipMalloc(&src, n4 * 16); (void)hipMalloc(&dst, n4 * 16); (void)hipMemset(src, 0, n4 * 16);
    unsigned long long* st; (void)hipMalloc(&st, 4096);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e0, e1, f0, f1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    const int itersV = 6000;
    const char* names[] = { "nothing", "fp32 MFMA 32x32x2", "bf16 MFMA 32x32x16", "streaming copy 1 GiB" };
    for (int kind = -1; kind < 3; kind++)
        for (int P : { 64, 256, 512 }) {
            if (kind < 0 && P != 256) continue;
            for (int rep = 0; rep < 2; rep++) {
                hipDeviceSynchronize();
                if (kind >= 0) hipEventRecord(f0, s2);
                if (kind == 0) hipLaunchKernelGGL(kM<0>, dim3(P), dim3(128), 0, s2, dm, src, dst, 400000, n4);
                if (kind == 1) hipLaunchKernelGGL(kM<1>, dim3(P), dim3(128), 0, s2, dm, src, dst, 1600000, n4);
                if (kind == 2) hipLaunchKernelGGL(kM<2>, dim3(P), dim3(128), 0, s2, dm, src, dst, 40, n4);
                if (kind >= 0) hipEventRecord(f1, s2);
                hipEventRecord(e0, s1);
                hipLaunchKernelGGL(kV, dim3(256 * 6 * 4), dim3(128), 0, s1, d, st, itersV);
                hipEventRecord(e1, s1);
                hipDeviceSynchronize();
            }
            float ms, msM = 0; hipEventElapsedTime(&ms, e0, e1); if (kind >= 0) hipEventElapsedTime(&msM, f0, f1);
            std::vector<unsigned long long> h(2 * 96); hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> clk; for (int i = 0; i < 96; i++) if (h[2 * i + 1]) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
            std::sort(clk.begin(), clk.end());
            printf("beside: %-22s P=%3d | vector kernel %.3f ms, in-kernel clock median %.2f GHz (min %.2f max %.2f) | side kernel %.3f ms\n",
                   kind < 0 ? names[0] : names[kind + 1], kind < 0 ? 0 : P, ms, clk[clk.size() / 2], clk.front(), clk.back(), msM);
        }
    return 0;
}
