// Infinity Cache probe (round 3): does data a kernel has just WRITTEN get served from the 256 MB Infinity Cache when the next kernel
// reads it?  write kernel (16 B per lane, streaming) over S bytes, then read kernel over the same S bytes; read time and rate by S.
// Variants: plain stores / nt stores; the read after a write of the SAME buffer vs after a write of ANOTHER buffer of the same size.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mall_probe.hip -o tools/probes/_bin/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT> __global__ void wr(float4* p, size_t n4, float v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = make_float4(v, v + 1, v + 2, (float)i);
        typedef float v4 __attribute__((ext_vector_type(4)));
        if (NT) { v4 w; w.x = x.x; w.y = x.y; w.z = x.z; w.w = x.w; __builtin_nontemporal_store(w, reinterpret_cast<v4*>(p + i)); } else p[i] = x;
    }
}
__global__ void rd(const float4* p, size_t n4, float* out)
{
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 x = p[i]; acc += x.x + x.y + x.z + x.w; }
    if (acc == 1.2345f) out[0] = acc;
}
int main()
{
    const size_t maxB = (size_t)2 << 30;
    float4 *a, *b; float* o;
    (void)hipMalloc(&a, maxB); (void)hipMalloc(&b, maxB); (void)hipMalloc(&o, 64);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const dim3 grid(256 * 8), block(256);
    printf("%10s | %28s | %28s | %28s\n", "S (MB)", "write S, read the SAME S", "nt-write S, read the SAME S", "write ANOTHER S, read S (cold)");
    for (size_t mb : { 16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048 }) {
        const size_t n4 = mb * (1 << 20) / 16;
        float res[3][2];
        for (int var = 0; var < 3; var++) {
            float tw = 0, tr = 0;
            for (int rep = 0; rep < 4; rep++) {
                hipLaunchKernelGGL(wr<0>, grid, block, 0, 0, a, n4, 1.0f);        /* make `a` exist with known content */
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (var == 0) hipLaunchKernelGGL(wr<0>, grid, block, 0, 0, a, n4, (float)rep);
                if (var == 1) hipLaunchKernelGGL(wr<1>, grid, block, 0, 0, a, n4, (float)rep);
                if (var == 2) hipLaunchKernelGGL(wr<0>, grid, block, 0, 0, b, n4, (float)rep);
                hipEventRecord(e1);
                hipLaunchKernelGGL(rd, grid, block, 0, 0, a, n4, o);
                hipEventRecord(e2); hipEventSynchronize(e2);
                hipEventElapsedTime(&tw, e0, e1); hipEventElapsedTime(&tr, e1, e2);
            }
            res[var][0] = mb / 1024.0f / (tw * 1e-3f) / 1000.0f; res[var][1] = mb / 1024.0f / (tr * 1e-3f) / 1000.0f;      /* TB/s */
        }
        printf("%10zu | write %5.2f TB/s read %5.2f TB/s | write %5.2f TB/s read %5.2f TB/s | write %5.2f TB/s read %5.2f TB/s\n", mb,
               res[0][0], res[0][1], res[1][0], res[1][1], res[2][0], res[2][1]);
    }
    return 0;
}
