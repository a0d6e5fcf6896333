// VALU issue-rate probe: independent v_fma_f32 chains, W waves per SIMD (block = 256*W threads on every CU), mixes with v_add / v_mul / v_mov / ds.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, int MIX> __global__ void k(float* out, int iters)
{
    float a[ILP];
    for (int i = 0; i < ILP; i++) a[i] = threadIdx.x * 0.001f + i;
    const float b = out[0] + 1.0001f, c = out[1] + 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (MIX == 0) a[i] = fmaf(a[i], b, c);
                else if (MIX == 1) a[i] = (r & 1) ? a[i] + c : a[i] * b;          // add / mul alternating
                else a[i] = (r & 1) ? fmaf(a[i], b, c) : a[i] - c;
            }
    }
    float s = 0; for (int i = 0; i < ILP; i++) s += a[i];
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float* d; (void)hipMalloc(&d, 64 << 20); (void)hipMemset(d, 0, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int W = 1; W <= 4; W++)
        for (int mix = 0; mix < 3; mix++) {
            dim3 grid(256 * 4), block(64 * W);        // 4 blocks per CU: one per SIMD (hopefully), W waves each
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mix == 0) hipLaunchKernelGGL((k<8, 0>), grid, block, 0, 0, d, iters);
                else if (mix == 1) hipLaunchKernelGGL((k<8, 1>), grid, block, 0, 0, d, iters);
                else hipLaunchKernelGGL((k<8, 2>), grid, block, 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)iters * 16 * 8 * W * 1024;      // wave-instructions total (1024 blocks x W waves)
            printf("waves/SIMD %d mix %d: %.3f ms, %.2f G wave-instr/s total, %.3f instr/ns/SIMD (1024 SIMDs)\n", W, mix, ms, instr / ms / 1e6, instr / ms / 1e6 / 1024);
        }
    return 0;
}
