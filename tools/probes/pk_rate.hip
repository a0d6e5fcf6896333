// Packed-fp32 issue-rate probe: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (two fp32 results per lane per instruction) against v_fma_f32,
// W waves per SIMD on every CU.  Answers: does a packed instruction issue at the scalar-fp32 rate (2x the flops) on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int ILP, int MIX> __global__ void k(float* out, int iters)
{
    v2f a[ILP];
    for (int i = 0; i < ILP; i++) a[i] = v2f{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
    const v2f b = v2f{out[0] + 1.0001f, out[1] + 0.9999f}, c = v2f{out[1] + 0.5f, out[0] - 0.5f};
    float s1[ILP * 2];
    for (int i = 0; i < ILP * 2; i++) s1[i] = threadIdx.x * 0.003f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (MIX == 0) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }                      // v_pk_fma_f32
                else if (MIX == 1) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }     // v_pk_add_f32
                else if (MIX == 2) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }     // v_pk_mul_f32
                else if (MIX == 3) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a[i]) : "v"(a[(i + 1) % ILP])); }  // a + i*b
                else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s1[2 * i]) : "v"(b.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s1[2 * i + 1]) : "v"(b.y), "v"(c.y)); }   // 2 x v_fma_f32
            }
    }
    float s = 0; for (int i = 0; i < ILP; i++) s += a[i].x + a[i].y + s1[2 * i] + s1[2 * i + 1];
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float* d; (void)hipMalloc(&d, 64 << 20); (void)hipMemset(d, 0, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[5] = {"v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_add_f32 op_sel/neg (a + i b)", "2 x v_fma_f32"};
    for (int W = 1; W <= 4; W++)
        for (int mix = 0; mix < 5; mix++) {
            dim3 grid(256 * 4), block(64 * W);
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, 0);
                if (mix == 0) hipLaunchKernelGGL((k<8, 0>), grid, block, 0, 0, d, iters);
                else if (mix == 1) hipLaunchKernelGGL((k<8, 1>), grid, block, 0, 0, d, iters);
                else if (mix == 2) hipLaunchKernelGGL((k<8, 2>), grid, block, 0, 0, d, iters);
                else if (mix == 3) hipLaunchKernelGGL((k<8, 3>), grid, block, 0, 0, d, iters);
                else hipLaunchKernelGGL((k<8, 4>), grid, block, 0, 0, d, iters);
                hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double instr = (double)grid.x * W * iters * 16 * 8 * (mix == 4 ? 2 : 1);
            printf("waves/SIMD %d  %-36s %.3f ms  %.3f instr/ns/SIMD  %.1f TFLOP/s-equivalent\n", W, names[mix], ms, instr / ms / 1e6 / 1024,
                   instr * 64 * (mix == 0 ? 4 : mix == 4 ? 2 : 2) / ms / 1e9);
        }
    return 0;
}
