/*
 * gemm_kernels.hip — band-batched real GEMM on the fp32 matrix cores (gfx950).
 *
 * Replaces the per-band `cblas_cgemm` + `cblas_sscal` of the reference decode
 * step (examples/src/ambi_dec/ambi_dec.c:518-540; the same shape recurs in
 * panner.c:266-274).  The decoder matrices are real, so instead of a complex
 * GEMM the interleaved re/im time slots are treated as 2*H real columns:
 *
 *     Y_b [64 x 2H] = A_{mat(b)} [64 x 64] * X_b [64 x 2H]      for every (instance, band b)
 *
 * A already contains the M_norm scale and is zero-padded to 64 x 64, so lower
 * per-band orders and fewer loudspeakers need no special cases.
 *
 * fp32-input MFMA (v_mfma_f32_32x32x2_f32) is an exact fp32 FMA chain, so the
 * 1e-5 parity budget is untouched; bf16/fp16 MFMA would not meet it.
 */
#include "saf_hip_common.h"

namespace saf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs { BandGemmLaunch g; };

/* One wave = one 32x32 output tile, K = 64 in 32 MFMA steps.
 * A operand of 32x32x2: lane l holds A[i = l&31][k = l>>5]; B: B[k = l>>5][j = l&31];
 * C/D: reg r, lane l -> row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31. */
__global__ __launch_bounds__(256) void band_gemm_kernel(GemmArgs a)
{
    const BandGemmLaunch& g = a.g;
    const int band = blockIdx.y, inst = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rowTile = wave & 1;
    const int col0 = (blockIdx.x * 2 + (wave >> 1)) * 32;
    if (col0 >= g.N) return;
    const int mat = g.band2mat[inst * g.nBands + band];
    const float* A = g.Afrag + (long long)inst * g.a_inst + (long long)(mat * 2 + rowTile) * 32 * 64;
    const float* X = g.X + (long long)inst * g.x_inst + (long long)band * g.x_band;
    const int col = col0 + (lane & 31);
    const bool cvalid = col < g.N;
    const int kh = lane >> 5;

    float av[32], bv[32];
#pragma unroll
    for (int s = 0; s < 32; s++) av[s] = A[s * 64 + lane];
#pragma unroll
    for (int s = 0; s < 32; s++) bv[s] = cvalid ? X[(long long)(2 * s + kh) * g.x_row + col] : 0.0f;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 32; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);

    if (cvalid) {
        float* Y = g.Y + (long long)inst * g.y_inst + (long long)band * g.y_band;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = rowTile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            Y[(long long)row * g.y_row + col] = acc[r];
        }
    }
}

void launch_band_gemm(const BandGemmLaunch& g)
{
    if (g.N <= 0 || g.nInst <= 0) return;
    GemmArgs a;
    a.g = g;
    dim3 grid((g.N + 63) / 64, g.nBands, g.nInst);
    KernelTimer kt("band_gemm");
    hipLaunchKernelGGL(band_gemm_kernel, grid, dim3(256), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

void pack_A(const float* A, float* Afrag)
{
    for (int rt = 0; rt < 2; rt++)
        for (int s = 0; s < 32; s++)
            for (int l = 0; l < 64; l++)
                Afrag[(rt * 32 + s) * 64 + l] = A[(rt * 32 + (l & 31)) * 64 + 2 * s + (l >> 5)];
}

}  // namespace saf
