/*
 * binaural_kernels.hip — the per-block kernels of the binauraliser (examples/src/binauraliser) for gfx950.
 *
 *   hrtf_interp_kernel   binauraliser_interpHRTFs (binauraliser_internal.c:46-123): nearest point of the 2 x 5 degree
 *                        VBAP grid over the HRIR directions -> 3 gains + 3 HRIR indices -> either the complex
 *                        triangular interpolation (INTERP_TRI) or magnitude + ITD interpolation with the interaural
 *                        phase re-introduced below 1.5 kHz (INTERP_TRI_PS).  One workgroup per source that moved.
 *   dvf_scale_kernel     binauraliser_nf (binauraliser_nf.c:299-338): response of each near source's two first-order DVF
 *                        shelves at the 133 band centres, multiplied onto the interpolated HRTF (the reference's product
 *                        (magnitude + i phase) * hrtf, kept as it is); far sources pass through.
 *   dec_rotate_kernel    ambi_bin (ambi_bin.c:437-456): the scene rotation folded into the binaural decoder, per band a
 *                        [2 x nSH] complex by [nSH x nSH] real product written in the band MAC's operand layout.
 *   binaural_mac_kernel  the band MAC of binauraliser_process (binauraliser.c:252-268): per band a [2 x nSrc] x
 *                        [nSrc x T] complex product.  Spectra rows [src][hop] are read exactly once; the source sum is
 *                        split over the thread groups of a workgroup and folded through LDS.
 */
#include "saf_hip_common.h"

namespace saf {

__device__ __forceinline__ float dev_matlab_fmodf(float x, float y) { const float t = fmodf(x, y); return t >= 0 ? t : t + y; }

struct InterpArgs { HrtfInterpLaunch l; };

__global__ __launch_bounds__(192) void hrtf_interp_kernel(InterpArgs a)
{
    const HrtfInterpLaunch& l = a.l;
    const int src = blockIdx.y * l.srcStride + blockIdx.x, band = threadIdx.x;
    if (!l.recalc[src] || band >= SAF_NBANDS) return;
    const float azi = l.srcDirs[src * 2], elev = l.srcDirs[src * 2 + 1];
    const float aziRes = (float)l.aziRes, elevRes = (float)l.elevRes;
    const int N_azi = (int)(360.0f / aziRes + 0.5f) + 1;
    const int aziIndex = (int)(dev_matlab_fmodf(azi + 180.0f, 360.0f) / aziRes + 0.5f);
    const int elevIndex = (int)((elev + 90.0f) / elevRes + 0.5f);
    const int idx3d = elevIndex * N_azi + aziIndex;
    float w[3]; int id[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { w[i] = l.gtComp[idx3d * 3 + i]; id[i] = l.gtIdx[idx3d * 3 + i]; }
    float2* out = l.hrtf_interp + ((long long)src * SAF_NBANDS + band) * 2;
    if (l.mode == 1) {                                                  /* INTERP_TRI (binauraliser.h:58-61) */
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float2* h = l.hrtf_fb + ((long long)band * 2 + e) * l.N;
            float re = 0.0f, im = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; i++) { const float2 v = h[id[i]]; re = fmaf(v.x, w[i], re); im = fmaf(v.y, w[i], im); }
            out[e] = make_float2(re, im);
        }
    } else {                                                            /* INTERP_TRI_PS */
        float itd = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; i++) itd = fmaf(w[i], l.itds[id[i]], itd);
        float mag[2] = { 0.0f, 0.0f };
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
            for (int i = 0; i < 3; i++) mag[e] = fmaf(w[i], l.hrtf_mag[((long long)band * 2 + e) * l.N + id[i]], mag[e]);
        const float f = l.freq[band];
        const float ipd = f < 1.5e3f ? (dev_matlab_fmodf(2.0f * SAF_PI * f * itd + SAF_PI, 2.0f * SAF_PI) - SAF_PI) / 2.0f : 0.0f;
        float s, c;
        sincosf(ipd, &s, &c);
        out[0] = make_float2(c * mag[0], s * mag[0]);
        out[1] = make_float2(c * mag[1], -s * mag[1]);
    }
}

void launch_hrtf_interp(const HrtfInterpLaunch& l)
{
    if (l.nSrc <= 0) return;
    InterpArgs a; a.l = l;
    KernelTimer kt("hrtf_interp");
    hipLaunchKernelGGL(hrtf_interp_kernel, dim3(l.nSrc, l.nInst > 0 ? l.nInst : 1), dim3(192), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

struct DvfArgs { DvfScaleLaunch l; };

/* grid (nSrc, nInst), one thread per band */
__global__ __launch_bounds__(192) void dvf_scale_kernel(DvfArgs a)
{
    const DvfScaleLaunch& l = a.l;
    const int src = blockIdx.y * l.srcStride + blockIdx.x, band = threadIdx.x;
    if (band >= SAF_NBANDS) return;
    const float2* hin = l.hrtf_interp + ((long long)src * SAF_NBANDS + band) * 2;
    float2* hout = l.hrtf_nf + ((long long)src * SAF_NBANDS + band) * 2;
    const float w = l.freq[band] * (float)(-2.0 * SAF_PI / l.fs);      /* z^-1 = e^{-jw} */
    float sx, cx;
    sincosf(w, &sx, &cx);
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const float4 k = *reinterpret_cast<const float4*>(l.coef + ((long long)src * 2 + e) * 4);    /* b0, b1, a1, near */
        const float2 h = hin[e];
        if (k.w == 0.0f) { hout[e] = h; continue; }
        const float nr = k.x + k.y * cx, ni = k.y * sx, dr = 1.0f + k.z * cx, di = k.z * sx;
        const double inv = 1.0 / (double)(dr * dr + di * di + 2.23e-7f);
        const float mag = (float)sqrt((double)(nr * nr + ni * ni) * inv);
        const float hr = (float)((double)(nr * dr + ni * di) * inv), hi = (float)((double)(ni * dr - nr * di) * inv);
        const float ph = (float)atan2((double)hi, (double)hr);
        hout[e] = make_float2(mag * h.x - ph * h.y, mag * h.y + ph * h.x);
    }
}

void launch_dvf_scale(const DvfScaleLaunch& l)
{
    if (l.nSrc <= 0) return;
    DvfArgs a; a.l = l;
    KernelTimer kt("dvf_scale");
    hipLaunchKernelGGL(dvf_scale_kernel, dim3(l.nSrc, l.nInst > 0 ? l.nInst : 1), dim3(192), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

struct DecRotArgs { DecRotLaunch l; };

/* grid 133 (band), 128 threads = 2 ears x 64 output columns; M_rot and the band's two decoder rows in LDS */
__global__ __launch_bounds__(128) void dec_rotate_kernel(DecRotArgs a)
{
    __shared__ float s_R[64 * 64];
    __shared__ float2 s_M[2][64];
    const DecRotLaunch& l = a.l;
    const int band = blockIdx.x, tid = threadIdx.x, e = tid >> 6, j = tid & 63, nSH = l.nSH;
    s_M[e][j] = l.Mdec[((long long)band * 2 + e) * 64 + j];
    if (l.Mrot) for (int i = tid; i < nSH * nSH; i += 128) s_R[i] = l.Mrot[i];
    __syncthreads();
    float2 acc = make_float2(0.0f, 0.0f);
    if (j < nSH) {
        if (l.Mrot) {
            for (int k = 0; k < nSH; k++) {
                const float r = s_R[k * nSH + j];
                const float2 m = s_M[e][k];
                acc.x = fmaf(m.x, r, acc.x); acc.y = fmaf(m.y, r, acc.y);
            }
        } else acc = s_M[e][j];
    }
    l.out[((long long)j * SAF_NBANDS + band) * 2 + e] = acc;
}

void launch_dec_rotate(const DecRotLaunch& l)
{
    if (l.nSH < 1 || l.nSH > 64) SAF_FATAL("dec_rotate: nSH out of range");
    DecRotArgs a; a.l = l;
    KernelTimer kt("dec_rotate");
    hipLaunchKernelGGL(dec_rotate_kernel, dim3(SAF_NBANDS), dim3(128), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

struct MacArgs2 { BinMacLaunch l; int TT, logTT; };

/* grid (ceil(H / TT), 133, nInst); 256 threads = TT hops x (256 / TT) source groups */
__global__ __launch_bounds__(256) void binaural_mac_kernel(MacArgs2 a)
{
    __shared__ float2 s_red[256][2];
    const BinMacLaunch& l = a.l;
    const int band = blockIdx.y;
    const int tid = threadIdx.x;
    const int t = tid & (a.TT - 1), sg = tid >> a.logTT, nG = 256 >> a.logTT;
    const int hop = blockIdx.x * a.TT + t;
    float2 accL = make_float2(0.f, 0.f), accR = make_float2(0.f, 0.f);
    {
        /* unconditional loads (hops beyond H re-read the last one and are never stored), 4 sources in flight per thread */
        const int hopc = hop < l.H ? hop : l.H - 1;
        const float2* X = l.X + (long long)blockIdx.z * l.x_inst + (long long)band * l.x_band + hopc;
        const float2* hh = l.h + (long long)blockIdx.z * l.h_inst + (long long)band * 2;
#pragma unroll 4
        for (int src = sg; src < l.nSrc; src += nG) {
            const float2 x = X[(long long)src * l.x_ch];
            const float4 hv = *reinterpret_cast<const float4*>(hh + (long long)src * SAF_NBANDS * 2);     /* left, right */
            accL.x = fmaf(hv.x, x.x, accL.x); accL.x = fmaf(-hv.y, x.y, accL.x);
            accL.y = fmaf(hv.x, x.y, accL.y); accL.y = fmaf(hv.y, x.x, accL.y);
            accR.x = fmaf(hv.z, x.x, accR.x); accR.x = fmaf(-hv.w, x.y, accR.x);
            accR.y = fmaf(hv.z, x.y, accR.y); accR.y = fmaf(hv.w, x.x, accR.y);
        }
    }
    s_red[tid][0] = accL; s_red[tid][1] = accR;
    __syncthreads();
    for (int stride = nG >> 1; stride >= 1; stride >>= 1) {
        if (sg < stride) {
            const int o = tid + (stride << a.logTT);
#pragma unroll
            for (int e = 0; e < 2; e++) { s_red[tid][e].x += s_red[o][e].x; s_red[tid][e].y += s_red[o][e].y; }
        }
        __syncthreads();
    }
    if (sg == 0 && hop < l.H) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float2 v = s_red[tid][e];
            l.Y[(long long)blockIdx.z * l.y_inst + (long long)band * l.y_band + (long long)e * l.y_ch + hop] = make_float2(v.x * l.scale, v.y * l.scale);     /* cblas_sscal 1/sqrt(nSources) */
        }
    }
}

/* grid (133, nInst); 128 threads = (SH channel, ear).  Runs only when a decoder matrix, the band -> matrix map or an
 * interpolated HRTF changed. */
struct FoldArgs { BinFoldLaunch l; };
__global__ __launch_bounds__(128) void binaural_fold_kernel(FoldArgs a)
{
    const BinFoldLaunch& l = a.l;
    const int band = blockIdx.x, inst = blockIdx.y;
    const int sh = threadIdx.x >> 1, ear = threadIdx.x & 1;
    const float* A = l.A + ((long long)inst * l.nMat + l.band2mat[inst * SAF_NBANDS + band]) * 64 * 64;
    const float2* h = l.h + (long long)inst * SAF_MAXCH * SAF_NBANDS * 2;
    float re = 0.0f, im = 0.0f;
    for (int ls = 0; ls < l.nLS; ls++) {
        const float2 hv = h[((long long)ls * SAF_NBANDS + band) * 2 + ear];
        const float m = A[ls * 64 + sh];
        re = fmaf(hv.x, m, re); im = fmaf(hv.y, m, im);
    }
    l.HM[(long long)inst * SAF_MAXCH * SAF_NBANDS * 2 + ((long long)sh * SAF_NBANDS + band) * 2 + ear] = make_float2(re, im);
}
void launch_binaural_fold(const BinFoldLaunch& l)
{
    if (l.nInst <= 0) return;
    FoldArgs a; a.l = l;
    KernelTimer kt("binaural_fold");
    hipLaunchKernelGGL(binaural_fold_kernel, dim3(SAF_NBANDS, l.nInst), dim3(128), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

void launch_binaural_mac(const BinMacLaunch& l)
{
    if (l.H <= 0 || l.nSrc <= 0) return;
    MacArgs2 a; a.l = l;
    int TT = 1, lg = 0;
    while (TT < l.H && TT < 64) { TT <<= 1; lg++; }
    a.TT = TT; a.logTT = lg;
    KernelTimer kt("binaural_mac");
    hipLaunchKernelGGL(binaural_mac_kernel, dim3((l.H + TT - 1) / TT, SAF_NBANDS, l.nInst > 0 ? l.nInst : 1), dim3(256), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
