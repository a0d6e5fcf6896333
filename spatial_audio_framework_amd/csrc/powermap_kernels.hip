/*
 * powermap_kernels.hip — covariance update and PWD activity map of the powermap operator for gfx950.
 *
 *   cov_update_kernel  per band  Cx <- a*Cx + (1-a) * X X^H  over the T time slots of a frame (powermap.c:258-267:
 *                      cblas_cgemm NoTrans/ConjTrans + sscal + saxpy), frame after frame inside one launch so the
 *                      4.4 MB of covariance matrices cross HBM once per call.  One workgroup per band and 32 x 32 quadrant, a
 *                      16 x 16 thread grid of 2 x 2 register blocks; up to 128 time slots of spectra staged in LDS per round.
 *   cgrp_kernel        C_grp = sum_band 1e3*EQ_b * Cx_b (top-left block of the band's order), bands in ascending order
 *                      (powermap.c:281-289).  Only Re(C_grp) is kept: the PWD map is y^T C y with a REAL steering
 *                      vector, so Im(C) cannot reach the real part.
 *   pwd_kernel         pmap[d] = sum_i Y[i][d] * (sum_j Re C[i][j] * Y[j][d])   (generatePWDmap, saf_sh.c:1544-1584)
 *                      followed by the temporal smoothing with the previous map (powermap.c:345-347).
 */
#include "saf_hip_common.h"

namespace saf {

struct CovArgs { CovLaunch l; };

/* grid (band, quadrant): a workgroup owns one 32 x 32 quadrant of the band's covariance matrix, a 16 x 16 thread grid of
 * 2 x 2 register blocks (rows i, i + 16 / columns j, j + 16 of the quadrant: consecutive lanes read consecutive LDS rows whose
 * stride is odd in 8-byte words, conflict-free).  Rounds of up to COV_SLOTS time slots: the rows of the quadrant's two channel
 * groups are staged in LDS by ONE wave of loads, then the frames are multiplied out of LDS one after the other with the
 * reference's update per frame (same arithmetic per matrix element as one workgroup per band: 4 x the workgroups, a quarter of
 * the arithmetic each; one workgroup per band with one frame per round was 32 us, latency- and issue-bound on half the chip). */
#define COV_SLOTS 128
__global__ __launch_bounds__(256) void cov_update_kernel(CovArgs a)
{
    extern __shared__ float2 s_x[];                          /* [64][ld]: rows 0..31 = the quadrant's i channels, 32..63 = its j channels */
    const CovLaunch& l = a.l;
    const int band = blockIdx.x;
    const int qi = (blockIdx.y >> 1) * 32, qj = (blockIdx.y & 1) * 32;
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;                  /* rows ti, ti + 16 and columns tj, tj + 16 of the quadrant */
    const int nSH = l.nSH, T = l.T;
    if (qi >= nSH || qj >= nSH) return;                      /* (uniform) nothing of this quadrant exists */
    const int fpr = COV_SLOTS / T;                            /* frames per round */
    const int ld = fpr * T + 1;
    float2* C = l.Cx + (long long)band * 64 * 64;
    float2 c[2][2];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const int i = qi + ti + 16 * u, j = qj + tj + 16 * v;
            c[u][v] = (i < nSH && j < nSH) ? C[i * 64 + j] : make_float2(0.f, 0.f);
        }
    const float2* X = l.X + (long long)band * l.x_band;
    const float al = l.alpha, be = 1.0f - l.alpha;
    for (int f0 = 0; f0 < l.nFrames; f0 += fpr) {
        const int nf = min(fpr, l.nFrames - f0), K = nf * T;  /* slots of this round */
        __syncthreads();
        /* unconditional loads (rows beyond nSH re-read row nSH-1 and are zeroed, slots beyond K re-read slot K-1 and are not
         * used), all issued before the first LDS write */
        float2 pre[COV_SLOTS * 64 / 256];
#pragma unroll
        for (int q = 0; q < COV_SLOTS * 64 / 256; q++) {
            const int idx = tid + 256 * q, r = idx / COV_SLOTS, t = idx - r * COV_SLOTS;
            const int ch = r < 32 ? qi + r : qj + r - 32;
            pre[q] = X[(long long)(ch < nSH ? ch : nSH - 1) * l.x_ch + f0 * T + (t < K ? t : K - 1)];
        }
#pragma unroll
        for (int q = 0; q < COV_SLOTS * 64 / 256; q++) {
            const int idx = tid + 256 * q, r = idx / COV_SLOTS, t = idx - r * COV_SLOTS;
            const int ch = r < 32 ? qi + r : qj + r - 32;
            if (t < K) s_x[r * ld + t] = ch < nSH ? pre[q] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        for (int f = 0; f < nf; f++) {
            float2 n[2][2];
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int v = 0; v < 2; v++) n[u][v] = make_float2(0.f, 0.f);
#pragma unroll 8
            for (int t = f * T; t < f * T + T; t++) {
                float2 xi[2], xj[2];
#pragma unroll
                for (int u = 0; u < 2; u++) { xi[u] = s_x[(ti + 16 * u) * ld + t]; xj[u] = s_x[(32 + tj + 16 * u) * ld + t]; }
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int v = 0; v < 2; v++) {        /* x_i * conj(x_j) */
                        n[u][v].x = fmaf(xi[u].x, xj[v].x, n[u][v].x); n[u][v].x = fmaf(xi[u].y, xj[v].y, n[u][v].x);
                        n[u][v].y = fmaf(xi[u].y, xj[v].x, n[u][v].y); n[u][v].y = fmaf(-xi[u].x, xj[v].y, n[u][v].y);
                    }
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int v = 0; v < 2; v++) {
                    c[u][v].x = c[u][v].x * al; c[u][v].y = c[u][v].y * al;                     /* cblas_sscal */
                    c[u][v].x = fmaf(be, n[u][v].x, c[u][v].x); c[u][v].y = fmaf(be, n[u][v].y, c[u][v].y);   /* cblas_saxpy */
                }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const int i = qi + ti + 16 * u, j = qj + tj + 16 * v;
            if (i < nSH && j < nSH) C[i * 64 + j] = c[u][v];
        }
}

void launch_cov_update(const CovLaunch& l)
{
    if (l.nFrames <= 0) return;
    if (l.T > 16) SAF_FATAL("powermap: more than 16 time slots per frame are not supported (frame size <= 2048)");
    CovArgs a; a.l = l;
    const size_t lds = sizeof(float2) * 64 * ((size_t)(COV_SLOTS / l.T) * l.T + 1);
    /* 66 KB of dynamic LDS: above the 64 KB a kernel gets without asking (set once; thread-safe static initialisation) */
    static const int attr_set = []() { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cov_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (COV_SLOTS + 1) * (int)sizeof(float2))); return 1; }();
    (void)attr_set;
    KernelTimer kt("cov_update");
    hipLaunchKernelGGL(cov_update_kernel, dim3(SAF_NBANDS, 4), dim3(256), lds, stream(), a);
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
