// LDS bank-conflict probe: each kernel repeats ONE phase of the filterbank kernels on a ring of LDS slots, so that
// rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS attributes conflicts to phases.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -I spatial_audio_framework_amd/csrc tools/probes/lds_probe.hip -o gpurun_out/lds_probe
#include "afstft_device.h"
#include <cstdio>
using namespace saf;
#define NSLOT 20
#define REP 64

__global__ __launch_bounds__(128) void k_fold(float* out, int swz)
{
    __shared__ __attribute__((aligned(16))) float s_ring[NSLOT * SLOT];
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int pos = (t + r) % NSLOT;
            float* slot = s_ring + pos * SLOT;
            const int fa = swz ? 2 * ((tid >> 1) ^ SLOT_SG(pos)) + (tid & 1) : tid;
            slot[fa] = (float)(r + t); slot[128 + fa] = (float)(r - t);
        }
        lds_barrier();
        acc += s_ring[tid];
    }
    out[blockIdx.x * 128 + tid] = acc;
}
__global__ __launch_bounds__(128) void k_ola(float* out, int swz)
{
    __shared__ __attribute__((aligned(16))) float s_ring[NSLOT * SLOT];
    const int tid = threadIdx.x;
    for (int i = tid; i < NSLOT * SLOT; i += 128) s_ring[i] = (float)i;
    __syncthreads();
    float acc = 0.f;
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int pos = (t + r) % NSLOT;
            const float* slot = s_ring + pos * SLOT;
            const int fa = swz ? 2 * ((tid >> 1) ^ SLOT_SG(pos)) + (tid & 1) : tid;
            acc += slot[fa] * 0.5f + slot[128 + fa];
        }
    }
    out[blockIdx.x * 128 + tid] = acc;
}
template <bool INV> __global__ __launch_bounds__(128) void k_fft(float* out, const float2* tw, int swz, int twmode)
{
    __shared__ __attribute__((aligned(16))) float s_ring[NSLOT * SLOT];
    __shared__ float2 s_twJ[128];
    const int tid = threadIdx.x;
    for (int i = tid; i < NSLOT * SLOT; i += 128) s_ring[i] = (float)(i & 255) * 0.01f;
    if (twmode) load_twiddles_pj(s_twJ, tw, tid); else s_twJ[tid] = tw[tid];
    __syncthreads();
    const int ff = tid >> 3, fj = tid & 7;
    for (int r = 0; r < REP; r++) {
        const int pos = (ff + r) % NSLOT;
        if (twmode) fft128_slot<INV>(s_ring + pos * SLOT, fj, TwCol{ s_twJ + fj }, swz ? SLOT_SG(pos) : 0);
        else fft128_slot<INV>(s_ring + pos * SLOT, fj, s_twJ + fj * 16, swz ? SLOT_SG(pos) : 0);
        lds_barrier();
    }
    out[blockIdx.x * 128 + tid] = s_ring[tid];
}
// the bin-pair phase of the equaliser kernel: lane = bin pair of one slot, consecutive 8-byte accesses
__global__ __launch_bounds__(128) void k_bins_rows(float* out)
{
    __shared__ __attribute__((aligned(16))) float s_ring[NSLOT * SLOT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < NSLOT * SLOT; i += 128) s_ring[i] = (float)(i & 255) * 0.01f;
    __syncthreads();
    const int k = lane + 1;
    for (int r = 0; r < REP; r++) {
#pragma unroll 2
        for (int i = 0; i < 8; i++) {
            float* slot = s_ring + ((wv + 2 * i + r) % NSLOT) * SLOT;
            const float2 Zk = *reinterpret_cast<const float2*>(slot + 2 * k);
            const float2 Zm = *reinterpret_cast<const float2*>(slot + 2 * (128 - k));
            *reinterpret_cast<float2*>(slot + 2 * k) = make_float2(Zk.x + Zm.x, Zk.y - Zm.y);
            if (k != 64) *reinterpret_cast<float2*>(slot + 2 * (128 - k)) = make_float2(Zk.x - Zm.x, Zk.y + Zm.y);
        }
        lds_barrier();
    }
    out[blockIdx.x * 128 + tid] = s_ring[tid];
}
// the split phase of the analysis kernel: 16 lanes = the same bin of 16 consecutive slots ("column" access), swizzled
__global__ __launch_bounds__(128) void k_bins_cols(float* out, int swz)
{
    __shared__ __attribute__((aligned(16))) float s_ring[22 * SLOT];
    const int tid = threadIdx.x;
    for (int i = tid; i < 22 * SLOT; i += 128) s_ring[i] = (float)(i & 255) * 0.01f;
    __syncthreads();
    const int st = tid & 15, sr = tid >> 4;
    float acc = 0.f;
    for (int r = 0; r < REP; r++) {
        int pos = (r * 16 + 6 + st) % 22;
        const int sg = swz ? SLOT_SG(pos) : 0;
        const float* slot = s_ring + pos * SLOT;
#pragma unroll
        for (int ii = 0; ii < 8; ii++) {
            const int k = sr + 8 * ii;
            const float2 Zk = *reinterpret_cast<const float2*>(slot + 2 * (k ^ sg));
            const float2 Zm = *reinterpret_cast<const float2*>(slot + 2 * (((128 - k) & 127) ^ sg));
            acc += Zk.x * Zm.y + Zk.y - Zm.x;
        }
    }
    out[blockIdx.x * 128 + tid] = acc;
}

int main()
{
    float* d; (void)hipMalloc(&d, 1024 * 128 * 4);
    float2 htw[128];
    for (int j = 0; j < 8; j++) for (int q = 0; q < 16; q++) { const double a = -2.0 * 3.14159265358979323846 * (double)(j * q) / 128.0; htw[j * 16 + q] = make_float2((float)cos(a), (float)sin(a)); }
    float2* tw; (void)hipMalloc(&tw, sizeof(htw)); (void)hipMemcpy(tw, htw, sizeof(htw), hipMemcpyHostToDevice);
    for (int it = 0; it < 2; it++) {
        hipLaunchKernelGGL(k_fold, dim3(1024), dim3(128), 0, 0, d, 0);
        hipLaunchKernelGGL(k_fold, dim3(1024), dim3(128), 0, 0, d, 1);
        hipLaunchKernelGGL(k_ola, dim3(1024), dim3(128), 0, 0, d, 0);
        hipLaunchKernelGGL(k_ola, dim3(1024), dim3(128), 0, 0, d, 1);
        hipLaunchKernelGGL(k_fft<false>, dim3(1024), dim3(128), 0, 0, d, tw, 0, 1);
        hipLaunchKernelGGL(k_fft<false>, dim3(1024), dim3(128), 0, 0, d, tw, 1, 0);
        hipLaunchKernelGGL(k_fft<true>, dim3(1024), dim3(128), 0, 0, d, tw, 0, 1);
        hipLaunchKernelGGL(k_bins_rows, dim3(1024), dim3(128), 0, 0, d);
        hipLaunchKernelGGL(k_bins_cols, dim3(1024), dim3(128), 0, 0, d, 0);
        hipLaunchKernelGGL(k_bins_cols, dim3(1024), dim3(128), 0, 0, d, 1);
    }
    (void)hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
