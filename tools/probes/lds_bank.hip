// LDS banking probe for 8-byte accesses: which lane->address patterns conflict for ds_read_b64 / ds_read2_b64 /
// ds_write_b64 / ds_write2_b64 on gfx950.  One kernel launch per (instruction, pattern, slot stride); read the counters
// with rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS.  Kernel arguments are encoded in the grid so
// that the dispatch list can be decoded: gridDim.x = 256 + id.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 256

// pattern: 0 = rows (lane j of group g: g*S + 2j), 1 = transpose write (g*S + 16p + 2(j^(p&7))), 2 = second read (g*S + 16j + 2(i^j)),
//          3 = rows within a 32-lane contiguous run (g4 = lane>>5: g4*S + 2*(lane&31)), 4 = "column": lane = slot (16) x item(4): (lane&15)*S + 2*(lane>>4)
__device__ __forceinline__ unsigned addr_of(int pat, int lane, int S, int it)
{
    const int g = lane >> 3, j = lane & 7;
    unsigned a;
    switch (pat) {
    case 0: a = g * S + 2 * j; break;
    case 1: a = g * S + 16 * (it & 15) + 2 * (j ^ (it & 7)); break;
    case 2: a = g * S + 16 * j + 2 * ((it & 7) ^ j); break;
    case 3: a = (lane >> 5) * S + 2 * (lane & 31); break;
    default: a = (lane & 15) * S + 2 * (lane >> 4); break;
    }
    return a * 4u;
}

template <int INS> __global__ __launch_bounds__(64) void k(float* out, int pat, int S)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8 * 1024; i += 64) lds[i] = (float)i;
    __syncthreads();
    float acc = 0.f;
    for (int r = 0; r < REP; r++) {
        const unsigned a = addr_of(pat, lane, S, r);
        if (INS == 0) { f2 v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc += v.x + v.y; }
        if (INS == 1) { f4 v; asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:8\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc += v.x + v.w; }
        if (INS == 2) { f2 v = { acc, 1.f }; asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v) : "memory"); }
        if (INS == 3) { f2 v = { acc, 1.f }; asm volatile("ds_write2_b64 %0, %1, %1 offset0:0 offset1:8\n s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v) : "memory"); }
        if (INS == 4) { f4 v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a & ~15u) : "memory"); acc += v.x + v.w; }
        if (INS == 5) { float v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); acc += v; }
    }
    out[blockIdx.x * 64 + lane] = acc + lds[lane];
}

int main()
{
    float* d; (void)hipMalloc(&d, 4096 * 64 * 4);
    const int strides[] = { 16, 256, 258, 260, 264, 272, 288, 320 };
    int id = 0;
    for (int ins = 0; ins < 6; ins++)
        for (int pat = 0; pat < 5; pat++)
            for (int s = 0; s < 8; s++) {
                const int S = strides[s];
                if (pat == 4 && S == 16) { id++; continue; }
                dim3 grid(256 + id);
                const size_t sh = 8 * 1024 * 4 + 1024;
                switch (ins) {
                case 0: hipLaunchKernelGGL(k<0>, grid, dim3(64), sh, 0, d, pat, S); break;
                case 1: hipLaunchKernelGGL(k<1>, grid, dim3(64), sh, 0, d, pat, S); break;
                case 2: hipLaunchKernelGGL(k<2>, grid, dim3(64), sh, 0, d, pat, S); break;
                case 3: hipLaunchKernelGGL(k<3>, grid, dim3(64), sh, 0, d, pat, S); break;
                case 4: hipLaunchKernelGGL(k<4>, grid, dim3(64), sh, 0, d, pat, S); break;
                default: hipLaunchKernelGGL(k<5>, grid, dim3(64), sh, 0, d, pat, S); break;
                }
                id++;
            }
    (void)hipDeviceSynchronize();
    printf("launched %d\n", id);
    return 0;
}
