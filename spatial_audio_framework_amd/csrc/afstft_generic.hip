/*
 * afstft_generic.hip — afSTFT analysis / synthesis for the hop sizes no operator of the path uses (64 and 256).
 *
 * afSTFT_create accepts hop sizes 64, 128 and 256 in hybrid mode (framework/resources/afSTFT/afSTFTlib.c:158-159); every
 * operator fixes 128, which is what afstft_kernels.hip is specialised for.  These kernels restate the same algorithm
 * (afSTFT_internal.c:237-653) with the hop size as a template parameter and no tuning: one workgroup of HOP threads per
 * (channel, instance) walks through the hops of a call in order; the 2*HOP-point real FFT is a plain radix-2 complex FFT
 * in LDS.  State layouts are those of afstft_kernels.hip with HOP in place of 128: the last 15 input hops per analysis
 * channel, the last 9 synthesised frames (2*HOP samples) per synthesis channel.
 *
 *   fold       f[(k&1)*HOP + n] = sum_k x[(t-9+k)*HOP + n] * w[k*HOP + n]               (afSTFT_internal.c:276-301)
 *   analysis   S_t = rFFT_{2 HOP}(f); hybrid: all bands delayed 3 hops, bins 1..4 split with the half-band FIR (:523-623)
 *   synthesis  B = merged bands, g_t = irFFT(B) / (2 HOP), out_t[n] = sum_k wS[k*HOP + n] * g_{t-k}[(k&1)*HOP + n]   (:335-453)
 */
#include "saf_hip_common.h"

namespace saf {

#define GCOEFF1 0.031273141818515176604f   /* afSTFT_internal.h:74 */
#define GCOEFF2 0.28127313041521179171f    /* afSTFT_internal.h:75 */

__device__ __forceinline__ int g_brev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

/* in-place radix-2 decimation-in-time FFT of N = 2*HOP complex points held bit-reversed in (re, im); HOP threads, one butterfly
 * each per stage; tw[j] = exp(-2 pi i j / N), j < HOP; INV conjugates the twiddles (unscaled) */
template <int HOP, bool INV> __device__ __forceinline__ void g_fft(float* re, float* im, const float2* tw, int tid)
{
    constexpr int N = 2 * HOP;
    for (int len = 2; len <= N; len <<= 1) {
        const int half = len >> 1;
        const int grp = tid / half, pos = tid - grp * half;
        const int i0 = grp * len + pos, i1 = i0 + half;
        float2 w = tw[pos * (N / len)];
        if (INV) w.y = -w.y;
        const float ur = re[i0], ui = im[i0], xr = re[i1], xi = im[i1];
        const float vr = xr * w.x - xi * w.y, vi = xr * w.y + xi * w.x;
        re[i0] = ur + vr; im[i0] = ui + vi; re[i1] = ur - vr; im[i1] = ui - vi;
        __syncthreads();
    }
}

struct GenAnaArgs { AnaLaunch a; const float* win; };

template <int HOP>
__global__ __launch_bounds__(HOP) void afstft_analysis_generic_kernel(GenAnaArgs g)
{
    constexpr int N = 2 * HOP, BITS = HOP == 64 ? 7 : 9;
    __shared__ float s_re[N], s_im[N];
    __shared__ float2 s_tw[HOP];
    __shared__ float2 s_del[4][HOP + 1];       /* spectra of the last 4 hops (3-hop delay of the hybrid mode) */
    __shared__ float2 s_low[8][4];             /* bins 1..4 of the last 8 hops (half-band FIR) */
    const AnaLaunch& a = g.a;
    const int n = threadIdx.x, ch = blockIdx.x, inst = blockIdx.y;
    const int T = a.hopsPerFrame;
    { float sn, cs; sincospif(-2.0f * (float)n / (float)N, &sn, &cs); s_tw[n] = make_float2(cs, sn); }
    const int srcch = a.ch_map ? a.ch_map[inst * (a.tab_stride ? a.tab_stride : a.nCh) + ch] : ch;
    const bool valid = srcch >= 0 && srcch < a.nChIn;
    const float scale = valid ? (a.ch_scale ? a.ch_scale[inst * (a.tab_stride ? a.tab_stride : a.nCh) + ch] : 1.0f) : 0.0f;
    const float* in = a.in + (long long)inst * a.in_inst + (long long)(valid ? srcch : 0) * a.in_ch + n;
    const float* hist = a.hist_rd + ((long long)inst * a.nCh + ch) * SAF_ANA_HIST * HOP + n;
    float w[10], xw[10];
#pragma unroll
    for (int k = 0; k < 10; k++) w[k] = g.win[k * HOP + n];
    auto sample = [&](int h) -> float {        /* hop h of [history | input] */
        if (h < 0) return hist[(SAF_ANA_HIST + h) * HOP];
        const int fr = h / T, sb = h - fr * T;
        return in[(long long)fr * a.in_frame + sb * HOP] * scale;
    };
    /* warm-up from hop -6 (hybrid: the FIR reaches 6 hops back); xw[k] = x[t - 9 + k] */
    const int t0 = a.hybrid ? -6 : 0;
#pragma unroll
    for (int k = 0; k < 9; k++) xw[k + 1] = sample(t0 - 9 + k);
    __syncthreads();
    float2* out = a.out + (long long)inst * a.out_inst + (long long)ch * a.out_ch;
    for (int t = t0; t < a.H; t++) {
#pragma unroll
        for (int k = 0; k < 9; k++) xw[k] = xw[k + 1];
        xw[9] = sample(t);
        float fe = 0.0f, fo = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; i++) { fe = fmaf(xw[2 * i], w[2 * i], fe); fo = fmaf(xw[2 * i + 1], w[2 * i + 1], fo); }
        s_re[g_brev(n, BITS)] = fe; s_im[g_brev(n, BITS)] = 0.0f;
        s_re[g_brev(n + HOP, BITS)] = fo; s_im[g_brev(n + HOP, BITS)] = 0.0f;
        __syncthreads();
        g_fft<HOP, false>(s_re, s_im, s_tw, n);
        if (!a.hybrid) {
            if (t >= 0) {
                out[(long long)n * a.out_band + t] = make_float2(s_re[n], n == 0 ? 0.0f : s_im[n]);
                if (n == 0) out[(long long)HOP * a.out_band + t] = make_float2(s_re[HOP], 0.0f);
            }
            __syncthreads();
            continue;
        }
        const int slot = (t + 8) & 3;
        s_del[slot][n] = make_float2(s_re[n], n == 0 ? 0.0f : s_im[n]);
        if (n == 0) s_del[slot][HOP] = make_float2(s_re[HOP], 0.0f);
        if (n >= 1 && n <= 4) s_low[(t + 8) & 7][n - 1] = make_float2(s_re[n], s_im[n]);
        __syncthreads();
        if (t >= 0) {
            const float2* D = s_del[(t - 3 + 8) & 3];                     /* all bands are delayed 3 hops */
            if (n == 0) {
                out[t] = D[0];
                out[(long long)(HOP + 4) * a.out_band + t] = D[HOP];      /* bin HOP -> band HOP + 4 */
            } else if (n <= 4) {
                const float2 S0 = s_low[(t + 8) & 7][n - 1], S2 = s_low[(t - 2 + 8) & 7][n - 1];
                const float2 S4 = s_low[(t - 4 + 8) & 7][n - 1], S6 = s_low[(t - 6 + 8) & 7][n - 1];
                float gr, gi;
                gr = -GCOEFF1 * S0.y;          gi = GCOEFF1 * S0.x;
                gr -= GCOEFF2 * S2.y;          gi += GCOEFF2 * S2.x;
                gr += GCOEFF2 * S4.y;          gi -= GCOEFF2 * S4.x;
                gr += GCOEFF1 * S6.y;          gi -= GCOEFF1 * S6.x;
                const float dr = D[n].x * 0.5f, di = D[n].y * 0.5f;
                const float sgn = (n & 1) ? -1.0f : 1.0f;                  /* afSTFT_internal.c:606-619 */
                out[(long long)(2 * n - 1) * a.out_band + t] = make_float2(dr + sgn * gr, di + sgn * gi);
                out[(long long)(2 * n) * a.out_band + t] = make_float2(dr - sgn * gr, di - sgn * gi);
            } else
                out[(long long)(n + 4) * a.out_band + t] = D[n];
        }
        __syncthreads();
    }
    if (a.hist_wr) {
        float* dst = a.hist_wr + ((long long)inst * a.nCh + ch) * SAF_ANA_HIST * HOP + n;
        for (int row = 0; row < SAF_ANA_HIST; row++) dst[row * HOP] = sample(a.H - SAF_ANA_HIST + row);
    }
}

struct GenSynArgs { SynLaunch s; const float* win; };

template <int HOP>
__global__ __launch_bounds__(HOP) void afstft_synthesis_generic_kernel(GenSynArgs g)
{
    constexpr int N = 2 * HOP, BITS = HOP == 64 ? 7 : 9;
    __shared__ float s_re[N], s_im[N];
    __shared__ float2 s_tw[HOP];
    __shared__ float s_fr[10][N];              /* ring of synthesised frames: frame of hop t in slot t % 10 */
    const SynLaunch& s = g.s;
    const int n = threadIdx.x, ch = blockIdx.x, inst = blockIdx.y;
    const int T = s.hopsPerFrame;
    { float sn, cs; sincospif(-2.0f * (float)n / (float)N, &sn, &cs); s_tw[n] = make_float2(cs, sn); }
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; k++) w[k] = g.win[k * HOP + n];
    const float* hist = s.hist_rd + ((long long)inst * s.nCh + ch) * SAF_SYN_HIST * N;
    for (int i = 0; i < 9; i++) {               /* frames of hops -9 .. -1 */
        const int slot = (i + 1) % 10;          /* hop -9 + i  ->  slot (hop + 10) % 10 */
        s_fr[slot][n] = hist[i * N + n]; s_fr[slot][HOP + n] = hist[i * N + HOP + n];
    }
    __syncthreads();
    const float2* in = s.in + (long long)inst * s.in_inst + (long long)ch * s.in_ch;
    float* out = s.out + (long long)inst * s.out_inst + (long long)ch * s.out_ch + n;
    auto band = [&](int b, int t) { return in[(long long)b * s.in_band + t]; };
    for (int t = 0; t < s.H; t++) {
        /* thread k builds bin k (thread 0 also bin HOP) and its mirror: afHybridInverse (afSTFT_internal.c:625-653) */
        float2 B;
        if (!s.hybrid) B = band(n, t);
        else if (n == 0) B = band(0, t);
        else if (n <= 4) { const float2 u = band(2 * n - 1, t), v = band(2 * n, t); B = make_float2(u.x + v.x, u.y + v.y); }
        else B = band(n + 4, t);
        if (s.lowDelay && (n & 1)) { B.x = -B.x; B.y = -B.y; }             /* circular half-frame shift (afSTFT_internal.c:366-369) */
        if (n == 0) {
            const float2 Bn = band(s.hybrid ? HOP + 4 : HOP, t);
            s_re[0] = B.x; s_im[0] = 0.0f;                                   /* imaginary parts of DC and Nyquist are ignored */
            s_re[g_brev(HOP, BITS)] = Bn.x; s_im[g_brev(HOP, BITS)] = 0.0f;
        } else {
            s_re[g_brev(n, BITS)] = B.x; s_im[g_brev(n, BITS)] = B.y;
            s_re[g_brev(N - n, BITS)] = B.x; s_im[g_brev(N - n, BITS)] = -B.y;
        }
        __syncthreads();
        g_fft<HOP, true>(s_re, s_im, s_tw, n);
        const int slot = t % 10;
        const float sc = 1.0f / (float)N;                                    /* saf_rfft_backward scales by 1/N (saf_utility_fft.c:751) */
        s_fr[slot][n] = s_re[n] * sc; s_fr[slot][HOP + n] = s_re[HOP + n] * sc;
        /* 10-segment overlap-add, oldest frame first (afSTFT_internal.c:396-444); only the thread's own sample position */
        float acc = 0.0f;
#pragma unroll
        for (int k = 9; k >= 0; k--) acc = fmaf(w[k], s_fr[(t - k + 20) % 10][(k & 1) * HOP + n], acc);
        const int fr = t / T, sb = t - fr * T;
        out[(long long)fr * s.out_frame + sb * HOP] = acc;
        __syncthreads();
    }
    if (s.hist_wr) {
        float* h = s.hist_wr + ((long long)inst * s.nCh + ch) * SAF_SYN_HIST * N;
        for (int i = 0; i < 9; i++) {
            const int hop = s.H - 9 + i;          /* frames of the last 9 hops (from the old history when the call was shorter) */
            const int slot = ((hop % 10) + 10) % 10;
            h[i * N + n] = s_fr[slot][n]; h[i * N + HOP + n] = s_fr[slot][HOP + n];
        }
    }
}

/* window of hop size `hop`: every (1024 / hop)-th tap of the 10240-tap prototype, reversed, times eq; the low-delay
 * synthesis window is not reversed (afSTFT_internal.c:122-145) */
static float* g_gen_win[2][2][2] = {};
static const float* dev_window_hop(int hop, int lowDelay, int synthesis)
{
    float*& d = g_gen_win[hop == 64 ? 0 : 1][lowDelay ? 1 : 0][synthesis ? 1 : 0];
    if (d) return d;
    const float* p = table_required(lowDelay ? "afSTFT_protoFilter1024LD" : "afSTFT_protoFilter1024", 10240);
    const float eq = lowDelay ? 2.0f / sqrtf(4.544559956f) : 2.0f / sqrtf(5.487604141f);
    const int ds = 1024 / hop, L = 10 * hop;
    std::vector<float> w(L);
    for (int k = 0; k < L; k++) {
        const float v = p[k * ds] * eq;
        if (lowDelay && synthesis) w[k] = v; else w[L - k - 1] = v;
    }
    HIP_CHECK(hipMalloc((void**)&d, L * sizeof(float)));
    HIP_CHECK(hipMemcpy(d, w.data(), L * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

void launch_analysis_generic(const AnaLaunch& a)
{
    if (a.H <= 0 || a.nCh <= 0 || a.nInst <= 0) return;
    if (a.hop != 64 && a.hop != 256) SAF_FATAL("afSTFT: hop size %d is not supported (64, 128, 256)", a.hop);
    GenAnaArgs g; g.a = a; g.win = dev_window_hop(a.hop, a.lowDelay, 0);
    KernelTimer kt("afstft_analysis_generic");
    if (a.hop == 64) hipLaunchKernelGGL(afstft_analysis_generic_kernel<64>, dim3(a.nCh, a.nInst), dim3(64), 0, stream(), g);
    else             hipLaunchKernelGGL(afstft_analysis_generic_kernel<256>, dim3(a.nCh, a.nInst), dim3(256), 0, stream(), g);
    HIP_CHECK(hipGetLastError());
}

void launch_synthesis_generic(const SynLaunch& s)
{
    if (s.H <= 0 || s.nCh <= 0 || s.nInst <= 0) return;
    if (s.hop != 64 && s.hop != 256) SAF_FATAL("afSTFT: hop size %d is not supported (64, 128, 256)", s.hop);
    GenSynArgs g; g.s = s; g.win = dev_window_hop(s.hop, s.lowDelay, 1);
    KernelTimer kt("afstft_synthesis_generic");
    if (s.hop == 64) hipLaunchKernelGGL(afstft_synthesis_generic_kernel<64>, dim3(s.nCh, s.nInst), dim3(64), 0, stream(), g);
    else             hipLaunchKernelGGL(afstft_synthesis_generic_kernel<256>, dim3(s.nCh, s.nInst), dim3(256), 0, stream(), g);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
