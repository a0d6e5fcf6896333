// Issue-rate probe: how much does scalar / LDS / dependent work in a wave's stream slow its VALU stream at 3 waves per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
// MODE 0: 8 independent FMA chains.  1: ONE dependent chain.  2: 8 chains + 1 SALU per 4 VALU.  3: + 1 SALU per 2 VALU.  4: + 1 SALU per VALU.
// 5: 8 chains, operands rotate over 3 different registers (register-bank pressure).  6: 8 chains + 1 ds_read_b32 per 8 VALU (result unused until the end).
template <int MODE> __global__ void k(float* out, int iters)
{
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    float b = out[0] + 1.0001f, c = out[1] + 0.5f, d = out[2] + 0.25f, e = out[3] + 0.75f;
    unsigned sx = 0; float acc = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) a[0] = fmaf(a[0], b, c);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (MODE == 5) a[i] = fmaf(a[i], (i & 1) ? b : d, (i & 2) ? c : e);
                    else a[i] = fmaf(a[i], b, c);
                    if (MODE == 2 && (i & 3) == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
                    if (MODE == 3 && (i & 1) == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
                    if (MODE == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
                }
                if (MODE == 6) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((threadIdx.x & 63) * 4)); acc += 0.f; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc = v; }
            }
        }
    }
    float s = acc + (float)sx; for (int i = 0; i < 8; i++) s += a[i];
    out[4 + blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float* d; (void)hipMalloc(&d, 64 << 20); (void)hipMemset(d, 0, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000;
    const char* names[] = { "8 FMA chains", "1 dependent chain", "+1 SALU / 4 VALU", "+1 SALU / 2 VALU", "+1 SALU / 1 VALU", "rotating operands", "+1 ds_read+wait / 8 VALU" };
    for (int W = 1; W <= 4; W += 2)
        for (int mode = 0; mode < 7; mode++) {
            dim3 grid(1024), block(64 * W);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, grid, block, 0, 0, d, iters); break;
                default: hipLaunchKernelGGL(k<6>, grid, block, 0, 0, d, iters); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double valu = (double)iters * 16 * 8 * W * 1024;
            printf("waves/SIMD %d  %-26s %.3f ms  %.3f VALU/ns/SIMD\n", W, names[mode], ms, valu / ms / 1e6 / 1024);
        }
    return 0;
}
