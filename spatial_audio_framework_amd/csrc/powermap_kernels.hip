/*
 * powermap_kernels.hip — covariance update and PWD activity map of the powermap operator for gfx950.
 *
 *   cov_update_kernel  per band  Cx <- a*Cx + (1-a) * X X^H  over the T time slots of a frame (powermap.c:258-267:
 *                      cblas_cgemm NoTrans/ConjTrans + sscal + saxpy), frame after frame inside one launch so the
 *                      4.4 MB of covariance matrices cross HBM once per call.  One workgroup per band, a 16 x 16 thread
 *                      grid of 4 x 4 register blocks; the frame's [nSH x T] spectra tile is staged in LDS.
 *   cgrp_kernel        C_grp = sum_band 1e3*EQ_b * Cx_b (top-left block of the band's order), bands in ascending order
 *                      (powermap.c:281-289).  Only Re(C_grp) is kept: the PWD map is y^T C y with a REAL steering
 *                      vector, so Im(C) cannot reach the real part.
 *   pwd_kernel         pmap[d] = sum_i Y[i][d] * (sum_j Re C[i][j] * Y[j][d])   (generatePWDmap, saf_sh.c:1544-1584)
 *                      followed by the temporal smoothing with the previous map (powermap.c:345-347).
 */
#include "saf_hip_common.h"

namespace saf {

struct CovArgs { CovLaunch l; };

__global__ __launch_bounds__(256) void cov_update_kernel(CovArgs a)
{
    __shared__ float2 s_x[64][17];                /* [ch][t], T <= 16, padded */
    const CovLaunch& l = a.l;
    const int band = blockIdx.x;
    const int tid = threadIdx.x;
    const int bi = (tid >> 4) * 4, bj = (tid & 15) * 4;      /* this thread's 4 x 4 block */
    const int nSH = l.nSH, T = l.T;
    float2* C = l.Cx + (long long)band * 64 * 64;
    float2 c[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) c[u][v] = (bi + u < nSH && bj + v < nSH) ? C[(bi + u) * 64 + bj + v] : make_float2(0.f, 0.f);
    const float2* X = l.X + (long long)band * l.x_band;
    const float al = l.alpha, be = 1.0f - l.alpha;
    for (int f = 0; f < l.nFrames; f++) {
        __syncthreads();
        for (int idx = tid; idx < 64 * T; idx += 256) {
            const int ch = idx / T, t = idx - ch * T;
            s_x[ch][t] = ch < nSH ? X[(long long)ch * l.x_ch + f * T + t] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        float2 n[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) n[u][v] = make_float2(0.f, 0.f);
        for (int t = 0; t < T; t++) {
            float2 xi[4], xj[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { xi[u] = s_x[bi + u][t]; xj[u] = s_x[bj + u][t]; }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) {        /* x_i * conj(x_j) */
                    n[u][v].x = fmaf(xi[u].x, xj[v].x, n[u][v].x); n[u][v].x = fmaf(xi[u].y, xj[v].y, n[u][v].x);
                    n[u][v].y = fmaf(xi[u].y, xj[v].x, n[u][v].y); n[u][v].y = fmaf(-xi[u].x, xj[v].y, n[u][v].y);
                }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                c[u][v].x = c[u][v].x * al; c[u][v].y = c[u][v].y * al;                     /* cblas_sscal */
                c[u][v].x = fmaf(be, n[u][v].x, c[u][v].x); c[u][v].y = fmaf(be, n[u][v].y, c[u][v].y);   /* cblas_saxpy */
            }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) if (bi + u < nSH && bj + v < nSH) C[(bi + u) * 64 + bj + v] = c[u][v];
}

void launch_cov_update(const CovLaunch& l)
{
    if (l.nFrames <= 0) return;
    if (l.T > 16) SAF_FATAL("powermap: more than 16 time slots per frame are not supported (frame size <= 2048)");
    CovArgs a; a.l = l;
    KernelTimer kt("cov_update");
    hipLaunchKernelGGL(cov_update_kernel, dim3(SAF_NBANDS), dim3(256), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

struct PwdArgs { PwdLaunch l; };

__global__ __launch_bounds__(256) void cgrp_kernel(PwdArgs a)
{
    const PwdLaunch& l = a.l;
    const int e = blockIdx.x * 256 + threadIdx.x;          /* all 64 x 64 entries: outside the nM x nM block the sum is empty */
    const int i = e >> 6, j = e & 63;
    float acc = 0.0f;
    for (int band = 0; band < l.nBands; band++) {
        const int ns = l.bandNSH[band];
        if (i < ns && j < ns) acc += l.Cx[(long long)band * 64 * 64 + i * 64 + j].x * l.bandScale[band];   /* crmulf then ccaddf (powermap.c:288) */
    }
    l.Cg[i * 64 + j] = acc;
}

__global__ __launch_bounds__(256) void pwd_kernel(PwdArgs a)
{
    __shared__ float s_C[64 * 64];
    const PwdLaunch& l = a.l;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) s_C[e] = l.Cg[e];
    __syncthreads();
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= l.G) return;
    float y[64];
#pragma unroll
    for (int i = 0; i < 64; i++) y[i] = i < l.nM ? l.Ygrid[(long long)i * l.G + d] : 0.0f;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 64; i++) {                         /* rows/columns beyond nM are zero on both sides */
        float cy = 0.0f;
#pragma unroll
        for (int j = 0; j < 64; j++) cy = fmaf(s_C[i * 64 + j], y[j], cy);
        acc = fmaf(y[i], cy, acc);
    }
    const float v = (1.0f - l.avg) * acc + l.avg * l.prev_pmap[d];
    l.pmap[d] = v;
    l.prev_pmap[d] = v;
}

void launch_pwd_map(const PwdLaunch& l)
{
    PwdArgs a; a.l = l;
    KernelTimer kt("pwd_map");
    hipLaunchKernelGGL(cgrp_kernel, dim3(16), dim3(256), 0, stream(), a);
    hipLaunchKernelGGL(pwd_kernel, dim3((l.G + 255) / 256), dim3(256), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
