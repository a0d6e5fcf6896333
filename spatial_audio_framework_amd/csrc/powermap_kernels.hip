/*
 * powermap_kernels.hip — covariance update and PWD activity map of the powermap operator for gfx950.
 *
 *   cov_update_kernel  per band  Cx <- a*Cx + (1-a) * X X^H  over the T time slots of a frame (powermap.c:258-267:
 *                      cblas_cgemm NoTrans/ConjTrans + sscal + saxpy), frame after frame inside one launch so the
 *                      4.4 MB of covariance matrices cross HBM once per call.  fp32 MFMA: one workgroup per (band, instance),
 *                      four waves = the four 32 x 32 quadrants, 64 time slots of spectra staged in LDS per round.
 *   cgrp_kernel        C_grp = sum_band 1e3*EQ_b * Cx_b (top-left block of the band's order), bands in ascending order
 *                      (powermap.c:281-289).  Only Re(C_grp) is kept: the PWD map is y^T C y with a REAL steering
 *                      vector, so Im(C) cannot reach the real part.
 *   pwd_kernel         pmap[d] = sum_i Y[i][d] * (sum_j Re C[i][j] * Y[j][d])   (generatePWDmap, saf_sh.c:1544-1584)
 *                      followed by the temporal smoothing with the previous map (powermap.c:345-347).
 */
#include "saf_hip_common.h"

namespace saf {

struct CovArgs { CovLaunch l; };

/* The covariance update on the fp32 matrix cores.  X X^H of a frame (64 channels x T time slots, complex) as two REAL products over
 * K = 2 T: with P[i][2t] = Re x_i[t], P[i][2t+1] = Im x_i[t] and Q[i][2t] = Im x_i[t], Q[i][2t+1] = -Re x_i[t]
 *     Re (X X^H) = P P^T,      Im (X X^H) = Q P^T
 * so one time slot is ONE k-pair step of v_mfma_f32_32x32x2_f32 for each of the two: the A operand of lane l is component
 * (l >> 5) of x_row(l & 31)[t] (for Q: the other component, sign flipped for Re), the B operand component (l >> 5) of
 * x_col(l & 31)[t].  grid (band, instance), 4 waves = the four 32 x 32 quadrants of the 64 x 64 matrix, their running
 * covariance in 2 x 16 accumulator registers per lane for the whole launch; the frames of a call are multiplied out of LDS one
 * after the other with the reference's update per frame (cblas_sscal by alpha, cblas_saxpy of (1 - alpha) X X^H: powermap.c:262-266),
 * so the 4.4 MB of covariances of an instance cross HBM once per call.  Rounds of COV_SLOTS time slots are staged in LDS by one
 * wave of coalesced loads (row stride odd in 8-byte words: the operand reads of a step are conflict-free).
 * Against round 2's vector kernel (16 x 16 threads x 2 x 2 register blocks): 558 MFLOP per 16 frames in 26 us = 21 TFLOP/s. */
#define COV_SLOTS 64
typedef float cov_f16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void cov_update_kernel(CovArgs a)
{
    __shared__ float2 s_x[64 * (COV_SLOTS + 1)];             /* [channel][slot], 33 KB */
    const CovLaunch& l = a.l;
    const int band = blockIdx.x, inst = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qi = (wv >> 1) * 32, qj = (wv & 1) * 32;      /* this wave's quadrant */
    const int kh = lane >> 5, lr = lane & 31;
    const int nSH = l.nSH, T = l.T;
    const bool quadOn = qi < nSH && qj < nSH;                /* (uniform per wave) */
    const int fpr = COV_SLOTS / T;                            /* frames per round */
    const int ld = COV_SLOTS + 1;
    float2* C = l.Cx + (long long)inst * l.cx_inst + (long long)band * 64 * 64;
    cov_f16 cr, ci;
#pragma unroll
    for (int r = 0; r < 16; r++) {                            /* register r of this lane: row (r & 3) + 8 (r >> 2) + 4 kh, column lr */
        const int i = qi + (r & 3) + 8 * (r >> 2) + 4 * kh, j = qj + lr;
        const float2 v = (quadOn && i < nSH && j < nSH) ? C[i * 64 + j] : make_float2(0.f, 0.f);
        cr[r] = v.x; ci[r] = v.y;
    }
    const float2* X = l.X + (long long)inst * l.x_inst + (long long)band * l.x_band;
    const float al = l.alphaInst ? l.alphaInst[inst] : l.alpha, be = 1.0f - al;
    const float2* xi = s_x + (qi + lr) * ld;
    const float2* xj = s_x + (qj + lr) * ld;
    for (int f0 = 0; f0 < l.nFrames; f0 += fpr) {
        const int nf = min(fpr, l.nFrames - f0), K = nf * T;  /* slots of this round */
        __syncthreads();
        /* unconditional loads (rows beyond nSH re-read row nSH-1 and are zeroed, slots beyond K re-read slot K-1 and are not
         * used), all issued before the first LDS write */
        float2 pre[COV_SLOTS * 64 / 256];
#pragma unroll
        for (int q = 0; q < COV_SLOTS * 64 / 256; q++) {
            const int idx = tid + 256 * q, r = idx / COV_SLOTS, t = idx - r * COV_SLOTS;
            pre[q] = X[(long long)(r < nSH ? r : nSH - 1) * l.x_ch + f0 * T + (t < K ? t : K - 1)];
        }
#pragma unroll
        for (int q = 0; q < COV_SLOTS * 64 / 256; q++) {
            const int idx = tid + 256 * q, r = idx / COV_SLOTS, t = idx - r * COV_SLOTS;
            s_x[r * ld + t] = r < nSH ? pre[q] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        if (quadOn)
            for (int f = 0; f < nf; f++) {
                cov_f16 nr, ni;
#pragma unroll
                for (int r = 0; r < 16; r++) { nr[r] = 0.0f; ni[r] = 0.0f; }
#pragma unroll 8
                for (int t = f * T; t < f * T + T; t++) {
                    const float2 vi = xi[t], vj = xj[t];
                    const float b = kh ? vj.y : vj.x;
                    nr = __builtin_amdgcn_mfma_f32_32x32x2f32(kh ? vi.y : vi.x, b, nr, 0, 0, 0);
                    ni = __builtin_amdgcn_mfma_f32_32x32x2f32(kh ? -vi.x : vi.y, b, ni, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    cr[r] = cr[r] * al; ci[r] = ci[r] * al;                                   /* cblas_sscal */
                    cr[r] = fmaf(be, nr[r], cr[r]); ci[r] = fmaf(be, ni[r], ci[r]);          /* cblas_saxpy */
                }
            }
    }
    if (!quadOn) return;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = qi + (r & 3) + 8 * (r >> 2) + 4 * kh, j = qj + lr;
        if (i < nSH && j < nSH) C[i * 64 + j] = make_float2(cr[r], ci[r]);
    }
}

void launch_cov_update(const CovLaunch& l)
{
    if (l.nFrames <= 0 || l.nInst <= 0) return;
    if (l.T > 16 || COV_SLOTS % l.T != 0) SAF_FATAL("powermap: the time slots per frame must divide 64 (frame sizes 128 ... 2048 in powers of two)");
    CovArgs a; a.l = l;
    KernelTimer kt("cov_update");
    hipLaunchKernelGGL(cov_update_kernel, dim3(SAF_NBANDS, l.nInst), dim3(256), 0, stream(), a);
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
    const int inst = blockIdx.y;
    if (l.mapOrder && l.mapOrder[inst] == 0) return;          /* (uniform) no map asked for this instance */
    const int i = blockIdx.x, j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int per = (l.nBands + 3) / 4;
    const int b0 = g * per, b1 = b0 + per < l.nBands ? b0 + per : l.nBands;
    float acc = 0.0f;
    const float2* Ce = l.Cx + (long long)inst * l.cx_inst + i * 64 + j;
    const int* bandNSH = l.bandNSH + (l.nInst > 1 ? inst * l.nBands : 0);
    const float* bandScale = l.bandScale + (l.nInst > 1 ? inst * l.nBands : 0);
#pragma unroll 17
    for (int band = b0; band < b1; band++) {
        const int ns = bandNSH[band];
        const float v = Ce[(long long)band * 64 * 64].x * bandScale[band];      /* crmulf then ccaddf (powermap.c:288) */
        acc += (i < ns && j < ns) ? v : 0.0f;
    }
    s_p[g][j] = acc;
    __syncthreads();
    if (g == 0) l.Cg[(long long)inst * 64 * 64 + i * 64 + j] = ((s_p[0][j] + s_p[1][j]) + s_p[2][j]) + s_p[3][j];
}

/* 256 threads = 64 directions x 4 row groups: thread (d, g) forms sum_{i in 16 g .. 16 g + 15} y_i (C y)_i; the four
 * partial sums meet through LDS in group order. */
__global__ __launch_bounds__(256) void pwd_kernel(PwdArgs a)
{
    __shared__ float s_C[64 * 64];
    __shared__ float s_part[4][64];
    const PwdLaunch& l = a.l;
    const int inst = blockIdx.y;
    const int mo = l.mapOrder ? l.mapOrder[inst] : -1;
    if (mo == 0) return;                                   /* (uniform) no map asked for this instance */
    const int nM = mo > 0 ? (mo + 1) * (mo + 1) : l.nM;
    const float* Ygrid = mo > 0 ? l.YgridByOrder[mo - 1] : l.Ygrid;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) s_C[e] = l.Cg[(long long)inst * 64 * 64 + e];
    const int dl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl;
    const int dc = d < l.G ? d : l.G - 1;
    float y[64];
#pragma unroll
    for (int i = 0; i < 64; i++) { const float v = Ygrid[(long long)(i < nM ? i : 0) * l.G + dc]; y[i] = i < nM ? v : 0.0f; }
    float yrow[16];                                        /* y_i of this thread's rows (y[] is indexed statically only) */
#pragma unroll
    for (int ii = 0; ii < 16; ii++) { const int i = g * 16 + ii; const float v = Ygrid[(long long)(i < nM ? i : 0) * l.G + dc]; yrow[ii] = i < nM ? v : 0.0f; }
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
    const float avg = l.avgInst ? l.avgInst[inst] : l.avg;
    const long long o = (long long)inst * l.G + d;
    const float v = (1.0f - avg) * tot + avg * l.prev_pmap[o];
    l.pmap[o] = v;
    l.prev_pmap[o] = v;
}

void launch_pwd_map(const PwdLaunch& l)
{
    PwdArgs a; a.l = l;
    KernelTimer kt("pwd_map");
    const int nI = l.nInst > 0 ? l.nInst : 1;
    hipLaunchKernelGGL(cgrp_kernel, dim3(64, nI), dim3(256), 0, stream(), a);
    hipLaunchKernelGGL(pwd_kernel, dim3((l.G + 63) / 64, nI), dim3(256), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
