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
    /* this thread's elements of a frame's [64 x T] tile (T <= 16: at most 4 per thread); loads are unconditional (rows
     * beyond nSH re-read row nSH-1 and are zeroed) and the NEXT frame's are issued before the current frame is multiplied */
    int tch[4], tt[4]; bool ton[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int idx = tid + 256 * q;
        ton[q] = idx < 64 * T;
        tch[q] = ton[q] ? idx / T : 0; tt[q] = ton[q] ? idx - tch[q] * T : 0;
    }
    float2 pre[4];
    auto fetch = [&](int f) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int chc = tch[q] < nSH ? tch[q] : nSH - 1;
            pre[q] = X[(long long)chc * l.x_ch + f * T + tt[q]];
        }
    };
    fetch(0);
    for (int f = 0; f < l.nFrames; f++) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (ton[q]) s_x[tch[q]][tt[q]] = tch[q] < nSH ? pre[q] : make_float2(0.f, 0.f);
        __syncthreads();
        fetch(f + 1 < l.nFrames ? f + 1 : f);
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

/* grid (64 rows); 256 threads = 64 columns x 4 band groups.  Every load is unconditional (a load under a branch gets its
 * own wait: the first version made 133 dependent round trips, 60 us); a band's contribution is masked by a select.  Each
 * group sums its contiguous band range in ascending order and the four partial sums are added in group order. */
__global__ __launch_bounds__(256) void cgrp_kernel(PwdArgs a)
{
    __shared__ float s_p[4][64];
    const PwdLaunch& l = a.l;
    const int i = blockIdx.x, j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int per = (l.nBands + 3) / 4;
    const int b0 = g * per, b1 = b0 + per < l.nBands ? b0 + per : l.nBands;
    float acc = 0.0f;
    const float2* Ce = l.Cx + i * 64 + j;
#pragma unroll 17
    for (int band = b0; band < b1; band++) {
        const int ns = l.bandNSH[band];
        const float v = Ce[(long long)band * 64 * 64].x * l.bandScale[band];      /* crmulf then ccaddf (powermap.c:288) */
        acc += (i < ns && j < ns) ? v : 0.0f;
    }
    s_p[g][j] = acc;
    __syncthreads();
    if (g == 0) l.Cg[i * 64 + j] = ((s_p[0][j] + s_p[1][j]) + s_p[2][j]) + s_p[3][j];
}

/* 256 threads = 64 directions x 4 row groups: thread (d, g) forms sum_{i in 16 g .. 16 g + 15} y_i (C y)_i; the four
 * partial sums meet through LDS in group order. */
__global__ __launch_bounds__(256) void pwd_kernel(PwdArgs a)
{
    __shared__ float s_C[64 * 64];
    __shared__ float s_part[4][64];
    const PwdLaunch& l = a.l;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) s_C[e] = l.Cg[e];
    const int dl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl;
    const int dc = d < l.G ? d : l.G - 1;
    float y[64];
#pragma unroll
    for (int i = 0; i < 64; i++) { const float v = l.Ygrid[(long long)(i < l.nM ? i : 0) * l.G + dc]; y[i] = i < l.nM ? v : 0.0f; }
    float yrow[16];                                        /* y_i of this thread's rows (y[] is indexed statically only) */
#pragma unroll
    for (int ii = 0; ii < 16; ii++) { const int i = g * 16 + ii; const float v = l.Ygrid[(long long)(i < l.nM ? i : 0) * l.G + dc]; yrow[ii] = i < l.nM ? v : 0.0f; }
    __syncthreads();
    float acc = 0.0f;
#pragma unroll
    for (int ii = 0; ii < 16; ii++) {                      /* rows/columns beyond nM are zero on both sides */
        const int i = g * 16 + ii;
        float cy = 0.0f;
#pragma unroll
        for (int j = 0; j < 64; j++) cy = fmaf(s_C[i * 64 + j], y[j], cy);
        acc = fmaf(yrow[ii], cy, acc);
    }
    s_part[g][dl] = acc;
    __syncthreads();
    if (g != 0 || d >= l.G) return;
    const float tot = ((s_part[0][dl] + s_part[1][dl]) + s_part[2][dl]) + s_part[3][dl];
    const float v = (1.0f - l.avg) * tot + l.avg * l.prev_pmap[d];
    l.pmap[d] = v;
    l.prev_pmap[d] = v;
}

void launch_pwd_map(const PwdLaunch& l)
{
    PwdArgs a; a.l = l;
    KernelTimer kt("pwd_map");
    hipLaunchKernelGGL(cgrp_kernel, dim3(64), dim3(256), 0, stream(), a);
    hipLaunchKernelGGL(pwd_kernel, dim3((l.G + 63) / 64), dim3(256), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
