/*
 * gemm_kernels.hip — the dense real GEMMs of the block path on the fp32 matrix cores (gfx950).
 *
 * band_gemm_kernel replaces the per-band `cblas_cgemm` + `cblas_sscal` of the reference decode
 * step (examples/src/ambi_dec/ambi_dec.c:518-540; the same shape recurs in panner.c:266-274).
 * The decoder matrices are real, so instead of a complex GEMM the interleaved re/im time slots
 * are treated as 2*H real columns:
 *
 *     Y_b [64 x 2H] = A_{mat(b)} [64 x 64] * X_b [64 x 2H]      for every (instance, band b)
 *
 * A already contains the M_norm scale and is zero-padded to 64 x 64, so lower per-band orders
 * and fewer loudspeakers need no special cases.
 *
 * enc_gemm_kernel replaces the `cblas_sgemm` pair, the cross-fade, the 1/sqrt(nSources) scale and
 * the output convention conversions of ambi_enc_process (examples/src/ambi_enc/ambi_enc.c:138-190).
 *
 * Both use the 32 x 128 wave tile of mfma_tile.h: every global access is a 16-byte load/store of
 * a 512-byte contiguous row segment.
 */
#include "saf_hip_common.h"
#include "mfma_tile.h"

namespace saf {

/* ========================================================================== */
/*                               band GEMM                                    */
/* ========================================================================== */

struct GemmArgs { BandGemmLaunch g; };

/* Workgroup = 2 waves = the two 32-row halves of one 64 x 128 output tile of one (instance, band). */
__global__ __launch_bounds__(128) void band_gemm_kernel(GemmArgs a)
{
    const BandGemmLaunch& g = a.g;
    const int band = blockIdx.y, inst = blockIdx.z;
    const int lane = threadIdx.x & 63, rt = threadIdx.x >> 6;
    const int col = blockIdx.x * 128 + 4 * (lane & 31);
    const int nValid = g.N - col;                       /* floats of this lane's 4 that are inside the matrix */
    const int kh = lane >> 5;
    const int mat = g.band2mat[inst * g.nBands + band];
    const float* A = g.Afrag + (long long)inst * g.a_inst + (long long)(mat * 2 + rt) * 32 * 64 + lane;
    const float* X = g.X + (long long)inst * g.x_inst + (long long)band * g.x_band + (long long)kh * g.x_row + col;

    float av[32];
#pragma unroll
    for (int s = 0; s < 32; s++) av[s] = A[s * 64];

    Tile128 t;
    tile_zero(t);
    /* K = 64 as 32 k-pair steps, streamed in 4 groups of 8 rows so loads of the next group fly under the MFMAs */
    float4 b[2][8];
#pragma unroll
    for (int i = 0; i < 8; i++) b[0][i] = load4_bounded(X + (long long)(2 * i) * g.x_row, nValid);
#pragma unroll
    for (int grp = 0; grp < 4; grp++) {
        if (grp < 3) {
#pragma unroll
            for (int i = 0; i < 8; i++) b[(grp + 1) & 1][i] = load4_bounded(X + (long long)(2 * (8 * (grp + 1) + i)) * g.x_row, nValid);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) tile_step(t, av[8 * grp + i], b[grp & 1][i]);
    }

    if (nValid > 0) {
        float* Y = g.Y + (long long)inst * g.y_inst + (long long)band * g.y_band + col;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = rt * 32 + tile_row(r, lane);
            store4_bounded(Y + (long long)row * g.y_row, make_float4(t.c[0][r], t.c[1][r], t.c[2][r], t.c[3][r]), nValid);
        }
    }
}

void launch_band_gemm(const BandGemmLaunch& g)
{
    if (g.N <= 0 || g.nInst <= 0) return;
    if ((g.x_row | g.y_row | g.x_band | g.y_band | g.x_inst | g.y_inst) & 3) SAF_FATAL("band gemm: strides must be multiples of 4 floats");
    if ((((uintptr_t)g.X) | ((uintptr_t)g.Y)) & 15) SAF_FATAL("band gemm: operands must be 16-byte aligned");
    GemmArgs a;
    a.g = g;
    dim3 grid((g.N + 127) / 128, g.nBands, g.nInst);
    KernelTimer kt("band_gemm");
    hipLaunchKernelGGL(band_gemm_kernel, grid, dim3(128), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

/* A [64][64] row-major -> MFMA fragment order [rt][s][lane]: lane l of step s holds A[rt*32 + (l&31)][2s + (l>>5)] */
void pack_A(const float* A, float* Afrag)
{
    for (int rt = 0; rt < 2; rt++)
        for (int s = 0; s < 32; s++)
            for (int l = 0; l < 64; l++)
                Afrag[(rt * 32 + s) * 64 + l] = A[(rt * 32 + (l & 31)) * 64 + 2 * s + (l >> 5)];
}

__global__ void pack_A_kernel(const float* A, float* Afrag, int nMat)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nMat * 4096) return;
    const int m = idx >> 12, e = idx & 4095;
    const int rt = e >> 11, s = (e >> 6) & 31, l = e & 63;
    Afrag[idx] = A[(long long)m * 4096 + (rt * 32 + (l & 31)) * 64 + 2 * s + (l >> 5)];
}

void launch_pack_A(const float* d_A, float* d_Afrag, int nMat)
{
    if (nMat <= 0) return;
    hipLaunchKernelGGL(pack_A_kernel, dim3((nMat * 4096 + 255) / 256), dim3(256), 0, stream(), d_A, d_Afrag, nMat);
    HIP_CHECK(hipGetLastError());
}

/* ========================================================================== */
/*                          ambi_enc encode GEMM                              */
/* ========================================================================== */

struct EncArgs { EncLaunch e; int frameBase; };

/* grid (column tiles of 128 samples, nFrames + 1, nInst); block row nFrames only saves the last
 * input frame (after gains) as the next call's "previous frame" (ambi_enc.c:165). */
template <bool CANMIX>
__global__ __launch_bounds__(128) void enc_gemm_kernel(EncArgs a)
{
#pragma clang fp contract(off)      /* the reference scales / fades in separate BLAS + veclib calls: keep the roundings */
    const EncLaunch& e = a.e;
    const int frame = a.frameBase + blockIdx.y, inst = blockIdx.z;
    const int lane = threadIdx.x & 63, rt = threadIdx.x >> 6;
    const int nSrc = e.nSrc[inst];
    const float* gains = e.gains + inst * SAF_MAXCH;

    if (frame == e.nFrames) {
        /* prev_inputFrameTD <- gained input of the last frame; rows beyond the present sources are zero */
        const float* src = e.in + (long long)inst * e.in_inst + (long long)(e.nFrames - 1) * e.in_frame;
        float* dst = e.prev_wr + (long long)inst * SAF_MAXCH * e.F;
        const int c4 = blockIdx.x * 32 + (threadIdx.x & 31);        /* float4 column of this thread */
        if (c4 * 4 < e.F)
            for (int row = threadIdx.x >> 5; row < SAF_MAXCH; row += 4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < nSrc) {
                    v = load4_bounded(src + (long long)row * e.in_ch + c4 * 4, e.F - c4 * 4);
                    const float gq = gains[row];
                    v.x *= gq; v.y *= gq; v.z *= gq; v.w *= gq;
                }
                store4_bounded(dst + (long long)row * e.F + c4 * 4, v, e.F - c4 * 4);
            }
        return;
    }

    const int col = blockIdx.x * 128 + 4 * (lane & 31);
    const int nValid = e.F - col;
    const int kh = lane >> 5;
    const bool fromState = frame == 0;
    const float* X = fromState ? e.prev_rd + (long long)inst * SAF_MAXCH * e.F
                               : e.in + (long long)inst * e.in_inst + (long long)(frame - 1) * e.in_frame;
    const long long xrow = fromState ? e.F : e.in_ch;
    const bool mix = CANMIX && fromState && e.mix != nullptr && e.mix[inst] != 0;
    const float* A = e.Afrag + ((long long)inst * 2 * 2 + rt) * 32 * 64 + lane;          /* Y */
    const float* Ap = A + 2 * 32 * 64;                                                    /* prev_Y */

    Tile128 t, tp;
    tile_zero(t);
    if (mix) tile_zero(tp);
#pragma unroll 8
    for (int s = 0; s < 32; s++) {
        const int k = 2 * s + kh;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < nSrc) {
            b = load4_bounded(X + (long long)k * xrow + col, nValid);
            if (!fromState) { const float gq = gains[k]; b.x *= gq; b.y *= gq; b.z *= gq; b.w *= gq; }
        }
        tile_step(t, A[s * 64], b);
        if (mix) tile_step(tp, Ap[s * 64], b);
    }

    if (nValid <= 0) return;
    const float post = e.postScale[inst];
    const float* rowScale = e.rowScale + inst * SAF_MAXCH;
    const int* rowMap = e.rowMap + inst * SAF_MAXCH;
    float fin[4], fout[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {                       /* interpolators of ambi_enc_init (ambi_enc.c:76-79) */
        fin[c] = (float)(col + c + 1) * 1.0f / (float)e.F;
        fout[c] = 1.0f - fin[c];
    }
    float* O = e.out + (long long)inst * e.out_inst + (long long)frame * e.out_frame + col;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = rt * 32 + tile_row(r, lane);
        const int orow = rowMap[row];
        if (orow < 0 || orow >= e.nOut) continue;
        float v[4] = { t.c[0][r], t.c[1][r], t.c[2][r], t.c[3][r] };
        if (mix) {
            const float w[4] = { tp.c[0][r], tp.c[1][r], tp.c[2][r], tp.c[3][r] };
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = fin[c] * v[c] + fout[c] * w[c];
        }
        const float rs = rowScale[row];
#pragma unroll
        for (int c = 0; c < 4; c++) { v[c] = v[c] * post; v[c] = v[c] * rs; }
        store4_bounded(O + (long long)orow * e.out_ch, make_float4(v[0], v[1], v[2], v[3]), nValid);
    }
}

void launch_enc_gemm(const EncLaunch& e)
{
    if (e.nFrames <= 0 || e.nInst <= 0) return;
    if ((e.in_inst | e.in_frame | e.in_ch | e.out_inst | e.out_frame | e.out_ch | e.F) & 3) SAF_FATAL("ambi_enc: strides and block size must be multiples of 4 floats");
    if ((((uintptr_t)e.in) | ((uintptr_t)e.out)) & 15) SAF_FATAL("ambi_enc: sample buffers must be 16-byte aligned");
    EncArgs a;
    a.e = e;
    KernelTimer kt("sh_encode");
    /* only block 0 of a call can cross-fade (direction changes arrive between calls): it alone runs the
     * variant that carries a second accumulator tile */
    const int first = e.mix ? 1 : 0;
    if (first) {
        a.frameBase = 0;
        hipLaunchKernelGGL(enc_gemm_kernel<true>, dim3((e.F + 127) / 128, 1, e.nInst), dim3(128), 0, stream(), a);
    }
    a.frameBase = first;
    hipLaunchKernelGGL(enc_gemm_kernel<false>, dim3((e.F + 127) / 128, e.nFrames + 1 - first, e.nInst), dim3(128), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
