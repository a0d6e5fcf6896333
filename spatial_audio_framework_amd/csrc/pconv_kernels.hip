/*
 * pconv_kernels.hip — FFT-domain matrix convolution kernels for gfx950.
 *
 * Replaces the block loop of saf_matrixConv_apply
 * (framework/modules/saf_utilities/saf_utility_matrixConv.c:165-236) and the filter
 * transforms of saf_matrixConv_create (:49-130):
 *
 *   pconv_rfft_fwd   zero-padded real FFT of every input block / filter partition (saf_rfft_forward,
 *                    saf_utility_fft.c:690-726: unscaled, N/2+1 bins)
 *   pconv_mac        Y[o][bin] = sum_p sum_i H[o][p][i][bin] * X[t-p][i][bin]          (:219, "the bulk of the CPU work")
 *   pconv_irfft      ONE inverse transform per output (saf_rfft_backward, :728-753: scaled 1/N; Im of DC and
 *                    Nyquist ignored) instead of the reference's nPartitions*nInputs transforms summed in the
 *                    time domain (:220-227) — the sum commutes with the transform
 *   pconv_ola        out[t] = sum_k z[t-k][k*hop : (k+1)*hop]                          (:225-233, :196-203)
 *
 * A real FFT of size N is computed as a complex FFT of M = N/2 points held in LDS (in-place radix-2
 * decimation in time, input scattered in bit-reversed order) followed by the usual split.  N is a power
 * of two >= 2*hop; any N >= hop + partitionLength - 1 gives the same linear convolution as the
 * reference's N = 2*hop / N = numOvrlpAddBlocks*hop, so arbitrary hop sizes are supported.
 */
#include "saf_hip_common.h"

namespace saf {

__device__ __forceinline__ float2 pc_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

/* in-place complex FFT of M = 1 << logM points in LDS; s holds the input in bit-reversed order.
 * tw[k] = exp(-2 pi i k / (2M)), k < M.  inverse: conjugate twiddles (unscaled). */
__device__ void lds_fft(float2* s, int M, int logM, const float2* __restrict__ tw, bool inverse)
{
    for (int st = 0; st < logM; st++) {
        const int half = 1 << st;
        __syncthreads();
        for (int b = threadIdx.x; b < (M >> 1); b += blockDim.x) {
            const int j = b & (half - 1);
            const int i0 = ((b >> st) << (st + 1)) + j;
            const int i1 = i0 + half;
            float2 w = tw[j * (M >> st)];
            if (inverse) w.y = -w.y;
            const float2 a = s[i0];
            const float2 t = pc_cmul(s[i1], w);
            s[i0] = make_float2(a.x + t.x, a.y + t.y);
            s[i1] = make_float2(a.x - t.x, a.y - t.y);
        }
    }
    __syncthreads();
}

struct FwdArgs {
    const float* src; long long s0, s1, s2;       /* element (x, y, z) of the grid starts at src + x*s0 + y*s1 + z*s2 */
    int nValid;                                   /* samples taken from src, the rest of the N-frame is zero */
    int yValidStep, yValidTotal;                  /* if yValidStep > 0: nValid = clamp(yValidTotal - y*yValidStep, 0, nValid) (last filter partition) */
    float2* dst; long long d0, d1, d2;
    int ringLen, ringHead;                        /* y is a time index: its slot is (ringHead + y) % ringLen (ringLen 0: plain y) */
    const float2* tw;
    int M, logM;
};

__global__ __launch_bounds__(256) void pconv_rfft_fwd_kernel(FwdArgs a)
{
    extern __shared__ float2 s_fft[];
    const int M = a.M, logM = a.logM;
    const float* src = a.src + (long long)blockIdx.x * a.s0 + (long long)blockIdx.y * a.s1 + (long long)blockIdx.z * a.s2;
    int nValid = a.nValid;
    if (a.yValidStep > 0) { int r = a.yValidTotal - (int)blockIdx.y * a.yValidStep; nValid = r < 0 ? 0 : (r < nValid ? r : nValid); }
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const float re = 2 * m < nValid ? src[2 * m] : 0.0f;
        const float im = 2 * m + 1 < nValid ? src[2 * m + 1] : 0.0f;
        s_fft[__brev((unsigned)m) >> (32 - logM)] = make_float2(re, im);
    }
    lds_fft(s_fft, M, logM, a.tw, false);
    const int y = a.ringLen ? (a.ringHead + (int)blockIdx.y) % a.ringLen : (int)blockIdx.y;
    float2* dst = a.dst + (long long)blockIdx.x * a.d0 + (long long)y * a.d1 + (long long)blockIdx.z * a.d2;
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
        const float2 Zk = s_fft[k & (M - 1)], Zm = s_fft[(M - k) & (M - 1)];
        const float2 W = k < M ? a.tw[k] : make_float2(-1.0f, 0.0f);
        const float2 e = make_float2(Zk.x + Zm.x, Zk.y - Zm.y);
        const float2 d = make_float2(Zk.x - Zm.x, Zk.y + Zm.y);
        const float2 t = pc_cmul(W, d);
        float2 X = make_float2(0.5f * (e.x + t.y), 0.5f * (e.y - t.x));
        if (k == 0 || k == M) X.y = 0.0f;
        dst[k] = X;
    }
}

/* -------------------------------------------------------------------------- */

#define MAC_TB 4      /* consecutive blocks that share one pass over the filter spectra */

struct MacArgs {
    const float2* Hf;      /* [nOut][nTerms = nFB*nIn][nBinsP] */
    const float2* Xr;      /* [ringLen][nIn][nBinsP] */
    float2* P;             /* partial sums [T][nOut][kSplit][nBinsP] */
    int nIn, nFB, nOut, nBins, nBinsP, kSplit, termsPerSplit;
    int ringLen, ringHead; /* slot of block t of this call = (ringHead + t) % ringLen */
    int T;
    int nX, diag;          /* channels per ring slot; diag = 1: output o convolves input channel o only (saf_multiConv), nIn = 1 */
};

__global__ __launch_bounds__(256) void pconv_mac_kernel(MacArgs a)
{
    __shared__ float2 s_red[MAC_TB][4][64];
    const int b = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int bin = blockIdx.x * 64 + b, o = blockIdx.y, ks = blockIdx.z;
    const bool live = bin < a.nBins;
    const int nTerms = a.nFB * a.nIn;
    const int k0 = ks * a.termsPerSplit;
    const int k1 = k0 + a.termsPerSplit < nTerms ? k0 + a.termsPerSplit : nTerms;
    const float2* H = a.Hf + (long long)o * nTerms * a.nBinsP + bin;
    for (int t0 = 0; t0 < a.T; t0 += MAC_TB) {
        float2 acc[MAC_TB];
#pragma unroll
        for (int u = 0; u < MAC_TB; u++) acc[u] = make_float2(0.f, 0.f);
        if (live)
            for (int k = k0 + q; k < k1; k += 4) {
                const int p = k / a.nIn, i = k - p * a.nIn;
                const float2 h = H[(long long)k * a.nBinsP];
#pragma unroll
                for (int u = 0; u < MAC_TB; u++) {
                    if (t0 + u >= a.T) break;
                    int slot = (a.ringHead + t0 + u - p) % a.ringLen; if (slot < 0) slot += a.ringLen;
                    const float2 x = a.Xr[((long long)slot * a.nX + i + o * a.diag) * a.nBinsP + bin];
                    acc[u].x = fmaf(h.x, x.x, acc[u].x); acc[u].x = fmaf(-h.y, x.y, acc[u].x);
                    acc[u].y = fmaf(h.x, x.y, acc[u].y); acc[u].y = fmaf(h.y, x.x, acc[u].y);
                }
            }
#pragma unroll
        for (int u = 0; u < MAC_TB; u++) s_red[u][q][b] = acc[u];
        __syncthreads();
        if (q == 0 && live) {
#pragma unroll
            for (int u = 0; u < MAC_TB; u++) {
                if (t0 + u >= a.T) break;
                float2 r = s_red[u][0][b];
#pragma unroll
                for (int qq = 1; qq < 4; qq++) { r.x += s_red[u][qq][b].x; r.y += s_red[u][qq][b].y; }
                a.P[(((long long)(t0 + u) * a.nOut + o) * a.kSplit + ks) * a.nBinsP + bin] = r;
            }
        }
        __syncthreads();
    }
}

/* -------------------------------------------------------------------------- */

struct InvArgs {
    const float2* P;       /* [T][nOut][kSplit][nBinsP] */
    float* zs;             /* [zRing][nOut][N] */
    const float2* tw;
    int nOut, kSplit, nBinsP, M, logM;
    int zRing, zHead;      /* slot of block t = (zHead + t) % zRing */
};

/* grid (nOut, T) */
__global__ __launch_bounds__(256) void pconv_irfft_kernel(InvArgs a)
{
    extern __shared__ float2 s_fft[];
    const int M = a.M, logM = a.logM, o = blockIdx.x, t = blockIdx.y;
    const float2* P = a.P + ((long long)t * a.nOut + o) * a.kSplit * a.nBinsP;
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        float2 Xk = make_float2(0.f, 0.f), Xm = make_float2(0.f, 0.f);
        for (int s = 0; s < a.kSplit; s++) {
            const float2 u = P[(long long)s * a.nBinsP + k], v = P[(long long)s * a.nBinsP + (M - k)];
            Xk.x += u.x; Xk.y += u.y; Xm.x += v.x; Xm.y += v.y;
        }
        if (k == 0) { Xk.y = 0.0f; Xm.y = 0.0f; }                   /* C2R ignores Im of DC and Nyquist (kiss_fftr.c:125-161) */
        const float2 E = make_float2(0.5f * (Xk.x + Xm.x), 0.5f * (Xk.y - Xm.y));
        const float2 D = make_float2(0.5f * (Xk.x - Xm.x), 0.5f * (Xk.y + Xm.y));
        float2 W = a.tw[k]; W.y = -W.y;                               /* e^{+2 pi i k / N} */
        const float2 O = pc_cmul(D, W);
        s_fft[__brev((unsigned)k) >> (32 - logM)] = make_float2(E.x - O.y, E.y + O.x);   /* E + i O */
    }
    lds_fft(s_fft, M, logM, a.tw, true);
    const int slot = (a.zHead + t) % a.zRing;
    float2* z = reinterpret_cast<float2*>(a.zs + ((long long)slot * a.nOut + o) * (2 * M));
    const float sc = 1.0f / (float)M;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const float2 v = s_fft[m];
        z[m] = make_float2(v.x * sc, v.y * sc);
    }
}

/* -------------------------------------------------------------------------- */

struct OlaArgs {
    const float* zs; float* out;
    long long out_ch, out_blk;
    int nOut, N, hop, nOB, zRing, zHead;
};

/* grid (ceil(hop/256), nOut, T): out[t][o][n] = sum_{k = nOB-1 .. 0} z[t-k][o][k*hop + n]  (oldest block first) */
__global__ __launch_bounds__(256) void pconv_ola_kernel(OlaArgs a)
{
    const int n = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, t = blockIdx.z;
    if (n >= a.hop) return;
    float acc = 0.0f;
    for (int k = a.nOB - 1; k >= 0; k--) {
        const int idx = k * a.hop + n;
        if (idx >= a.N) continue;
        int slot = (a.zHead + t - k) % a.zRing; if (slot < 0) slot += a.zRing;
        acc += a.zs[((long long)slot * a.nOut + o) * a.N + idx];
    }
    a.out[(long long)t * a.out_blk + (long long)o * a.out_ch + n] = acc;
}

/* -------------------------------------------------------------------------- */
/*  time-varying convolver (saf_TVConv_apply, saf_utility_matrixConv.c:554-620) */
/* -------------------------------------------------------------------------- */

struct TvMacArgs {
    const float2* Hf;      /* [nIRs][nOut][nFB][nBinsP] */
    const float2* Xr;      /* [ringLen][nBinsP] (one input channel) */
    float2* P;             /* [T][nOut*3][nBinsP] */
    const int* irSel;      /* [T][3]: IR of block t, of t-1, of t-2 */
    int nFB, nOut, nBins, nBinsP, ringLen, ringHead;
};

/* grid (ceil(nBins/64), nOut*3, T); 256 threads = 64 bins x 4 partition groups */
__global__ __launch_bounds__(256) void tvconv_mac_kernel(TvMacArgs a)
{
    __shared__ float2 s_red[4][64];
    const int b = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int bin = blockIdx.x * 64 + b, ov = blockIdx.y, t = blockIdx.z;
    const int o = ov / 3, v = ov - 3 * o;
    const int ir = a.irSel[t * 3 + v];
    float2 acc = make_float2(0.f, 0.f);
    if (bin < a.nBins) {
        const float2* H = a.Hf + ((long long)ir * a.nOut + o) * a.nFB * a.nBinsP + bin;
        for (int p = q; p < a.nFB; p += 4) {
            int slot = (a.ringHead + t - p) % a.ringLen; if (slot < 0) slot += a.ringLen;
            const float2 h = H[(long long)p * a.nBinsP], x = a.Xr[(long long)slot * a.nBinsP + bin];
            acc.x = fmaf(h.x, x.x, acc.x); acc.x = fmaf(-h.y, x.y, acc.x);
            acc.y = fmaf(h.x, x.y, acc.y); acc.y = fmaf(h.y, x.x, acc.y);
        }
    }
    s_red[q][b] = acc;
    __syncthreads();
    if (q == 0 && bin < a.nBins) {
        float2 r = s_red[0][b];
#pragma unroll
        for (int qq = 1; qq < 4; qq++) { r.x += s_red[qq][b].x; r.y += s_red[qq][b].y; }
        a.P[((long long)t * a.nOut * 3 + ov) * a.nBinsP + bin] = r;
    }
}

struct TvMixArgs {
    const float* zs;       /* [zRing][nOut*3][N] */
    float* out; long long out_ch, out_blk;
    int nOut, N, hop, zRing, zHead;
};

/* grid (ceil(hop/256), nOut, T):  out = (z_last[n] + tail of the previous block's z_cur) * fadeIn
 *                                     + (z_last2[n] + tail of the previous block's z_last) * fadeOut      (:600-611) */
__global__ __launch_bounds__(256) void tvconv_mix_kernel(TvMixArgs a)
{
    const int n = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, t = blockIdx.z;
    if (n >= a.hop) return;
    const int sl = (a.zHead + t) % a.zRing;
    int sp = (a.zHead + t - 1) % a.zRing; if (sp < 0) sp += a.zRing;
    const float* zc = a.zs + ((long long)sl * a.nOut * 3 + o * 3) * a.N;
    const float* zp = a.zs + ((long long)sp * a.nOut * 3 + o * 3) * a.N;
    const float out1 = __fadd_rn(zc[(long long)1 * a.N + n], zp[(long long)0 * a.N + a.hop + n]);
    const float out2 = __fadd_rn(zc[(long long)2 * a.N + n], zp[(long long)1 * a.N + a.hop + n]);
    const float fi = (float)n / (float)(a.hop - 1), fo = (float)(a.hop - 1 - n) / (float)(a.hop - 1);
    a.out[(long long)t * a.out_blk + (long long)o * a.out_ch + n] = __fadd_rn(__fmul_rn(out1, fi), __fmul_rn(out2, fo));
}

/* -------------------------------------------------------------------------- */
/*                                launchers                                   */
/* -------------------------------------------------------------------------- */

void pconv_twiddles(int N, DevBuf<float2>& tw)
{
    const int M = N / 2;
    std::vector<float2> h(M);
    for (int k = 0; k < M; k++) {
        const double ang = -2.0 * SAF_PId * (double)k / (double)N;
        h[k] = make_float2((float)cos(ang), (float)sin(ang));
    }
    tw.alloc(M, false);
    HIP_CHECK(hipMemcpy(tw.p, h.data(), sizeof(float2) * M, hipMemcpyHostToDevice));
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }

void pconv_launch_fwd(const PconvFwd& f)
{
    FwdArgs a;
    a.src = f.src; a.s0 = f.s0; a.s1 = f.s1; a.s2 = f.s2; a.nValid = f.nValid; a.yValidStep = f.yValidStep; a.yValidTotal = f.yValidTotal;
    a.dst = f.dst; a.d0 = f.d0; a.d1 = f.d1; a.d2 = f.d2; a.ringLen = f.ringLen; a.ringHead = f.ringHead;
    a.tw = f.tw; a.M = f.N / 2; a.logM = ilog2(a.M);
    if ((1 << a.logM) != a.M || a.M < 2 || a.M > 16384) SAF_FATAL("matrixConv: FFT size %d unsupported", f.N);
    const size_t lds = sizeof(float2) * a.M;
    static bool raised = false;
    if (!raised) {
        HIP_CHECK(hipFuncSetAttribute((const void*)pconv_rfft_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        HIP_CHECK(hipFuncSetAttribute((const void*)pconv_irfft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        raised = true;
    }
    KernelTimer kt("pconv_fft");
    hipLaunchKernelGGL(pconv_rfft_fwd_kernel, dim3(f.g0, f.g1, f.g2), dim3(256), lds, stream(), a);
    HIP_CHECK(hipGetLastError());
}

void pconv_launch_apply(const PconvApply& p)
{
    const int M = p.N / 2, logM = ilog2(M);
    {
        MacArgs a;
        a.Hf = p.Hf; a.Xr = p.Xr; a.P = p.P; a.nIn = p.nIn; a.nFB = p.nFB; a.nOut = p.nOut; a.nBins = M + 1; a.nBinsP = p.nBinsP;
        a.kSplit = p.kSplit; a.termsPerSplit = p.termsPerSplit; a.ringLen = p.xRing; a.ringHead = p.xHead; a.T = p.T;
        a.diag = p.diag ? 1 : 0; a.nX = p.diag ? p.nOut : p.nIn;
        KernelTimer kt("pconv_mac");
        hipLaunchKernelGGL(pconv_mac_kernel, dim3((M + 1 + 63) / 64, p.nOut, p.kSplit), dim3(256), 0, stream(), a);
        HIP_CHECK(hipGetLastError());
    }
    {
        InvArgs a;
        a.P = p.P; a.zs = p.zs; a.tw = p.tw; a.nOut = p.nOut; a.kSplit = p.kSplit; a.nBinsP = p.nBinsP; a.M = M; a.logM = logM;
        a.zRing = p.zRing; a.zHead = p.zHead;
        KernelTimer kt("pconv_ifft");
        hipLaunchKernelGGL(pconv_irfft_kernel, dim3(p.nOut, p.T), dim3(256), sizeof(float2) * M, stream(), a);
        HIP_CHECK(hipGetLastError());
    }
    {
        OlaArgs a;
        a.zs = p.zs; a.out = p.out; a.out_ch = p.out_ch; a.out_blk = p.out_blk; a.nOut = p.nOut; a.N = p.N; a.hop = p.hop; a.nOB = p.nOB;
        a.zRing = p.zRing; a.zHead = p.zHead;
        hipLaunchKernelGGL(pconv_ola_kernel, dim3((p.hop + 255) / 256, p.nOut, p.T), dim3(256), 0, stream(), a);
        HIP_CHECK(hipGetLastError());
    }
}

void tvconv_launch_apply(const TvApply& p)
{
    const int M = p.N / 2, logM = ilog2(M);
    {
        TvMacArgs a;
        a.Hf = p.Hf; a.Xr = p.Xr; a.P = p.P; a.irSel = p.irSel; a.nFB = p.nFB; a.nOut = p.nOut; a.nBins = M + 1; a.nBinsP = p.nBinsP;
        a.ringLen = p.xRing; a.ringHead = p.xHead;
        KernelTimer kt("tvconv_mac");
        hipLaunchKernelGGL(tvconv_mac_kernel, dim3((M + 1 + 63) / 64, p.nOut * 3, p.T), dim3(256), 0, stream(), a);
        HIP_CHECK(hipGetLastError());
    }
    {
        InvArgs a;
        a.P = p.P; a.zs = p.zs; a.tw = p.tw; a.nOut = p.nOut * 3; a.kSplit = 1; a.nBinsP = p.nBinsP; a.M = M; a.logM = logM;
        a.zRing = p.zRing; a.zHead = p.zHead;
        KernelTimer kt("pconv_ifft");
        hipLaunchKernelGGL(pconv_irfft_kernel, dim3(p.nOut * 3, p.T), dim3(256), sizeof(float2) * M, stream(), a);
        HIP_CHECK(hipGetLastError());
    }
    {
        TvMixArgs a;
        a.zs = p.zs; a.out = p.out; a.out_ch = p.out_ch; a.out_blk = p.out_blk; a.nOut = p.nOut; a.N = p.N; a.hop = p.hop; a.zRing = p.zRing; a.zHead = p.zHead;
        hipLaunchKernelGGL(tvconv_mix_kernel, dim3((p.hop + 255) / 256, p.nOut, p.T), dim3(256), 0, stream(), a);
        HIP_CHECK(hipGetLastError());
    }
}

}  // namespace saf
