/*
 * afstft_kernels.hip — afSTFT analysis / synthesis for gfx950 (MI355X).
 *
 * Replaces the per-hop, per-channel CPU loops of the reference
 *   afSTFTlib_forward   (framework/resources/afSTFT/afSTFT_internal.c:237-333)
 *   afHybridForward     (afSTFT_internal.c:523-623)
 *   afHybridInverse     (afSTFT_internal.c:625-653)
 *   afSTFTlib_inverse   (afSTFT_internal.c:335-453)
 * and the strided stores/loads of afSTFT_forward/backward_knownDimensions
 * (afSTFTlib.c:267-308, :390-431).
 *
 * The filterbank has no recursion, so any number of hops is processed in one
 * launch from (a) 15 hops of input history (9 for the 1280-tap prototype window,
 * 6 more for the hybrid half-band FIR and its 3-hop delay) and (b) the last 9
 * synthesised frames for the 10-segment overlap-add.
 *
 * Mapping: one workgroup (4 waves) per (instance, channel[, hop chunk]); one
 * 64-lane wave per hop computes the 256-point real FFT as a 128-point complex
 * FFT with two points per lane: six radix-2 stages exchange across lanes
 * (wave shuffles, no LDS traffic), the seventh is in-lane.  Spectra of a
 * 16-hop sub-chunk are staged in LDS so that every global store/load of the
 * [band][channel][time] layout is a full 128-byte line.
 */
#include "saf_hip_common.h"

namespace saf {

#define SUB      16      /* hops per sub-chunk */
#define SPEC_LD  129     /* odd leading dimension: conflict-free across slots */
#define RING     22      /* SUB + 6 spectra kept for the hybrid filter */
#define G_LD     288     /* 256 + 32: padded so the bit-reversed float4 stores spread over the banks */

#define COEFF1 0.031273141818515176604f   /* afSTFT_internal.h:74 */
#define COEFF2 0.28127313041521179171f    /* afSTFT_internal.h:75 */

__device__ __forceinline__ int bitrev6(int x) { return (int)(__brev((unsigned)x) >> 26); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { /* a * conj(b) */ return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ float2 shflx(float2 v, int h) { return make_float2(__shfl_xor(v.x, h), __shfl_xor(v.y, h)); }
__device__ __forceinline__ float2 shfl(float2 v, int src) { return make_float2(__shfl(v.x, src), __shfl(v.y, src)); }

/* per-lane twiddles: rows 0..5 = W_{2h}^{lane & (h-1)}, h = 1<<row; row 6 = W_128^lane; row 7 = W_256^lane (forward sign) */
struct LaneTw { float2 st[6]; float2 w128; float2 w256; };
__device__ __forceinline__ LaneTw load_tw(const float2* tw, int lane)
{
    LaneTw t;
#pragma unroll
    for (int s = 0; s < 6; s++) t.st[s] = tw[s * 64 + lane];
    t.w128 = tw[6 * 64 + lane];
    t.w256 = tw[7 * 64 + lane];
    return t;
}

/* 128-point complex forward FFT, decimation in time.  In: lane holds z[2q], z[2q+1], q = bitrev6(lane).
 * Out: A = Z[lane], B = Z[lane + 64]. */
__device__ __forceinline__ void fft128_fwd(float2 e, float2 o, const LaneTw& tw, int lane, float2& A, float2& B)
{
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int h = 1 << s;
        const bool up = (lane & h) != 0;
        float2 te = up ? cmul(e, tw.st[s]) : e;
        float2 to = up ? cmul(o, tw.st[s]) : o;
        float2 pe = shflx(te, h), po = shflx(to, h);
        e = up ? make_float2(pe.x - te.x, pe.y - te.y) : make_float2(te.x + pe.x, te.y + pe.y);
        o = up ? make_float2(po.x - to.x, po.y - to.y) : make_float2(to.x + po.x, to.y + po.y);
    }
    float2 t = cmul(o, tw.w128);
    A = make_float2(e.x + t.x, e.y + t.y);
    B = make_float2(e.x - t.x, e.y - t.y);
}

/* 128-point complex inverse FFT (unscaled), decimation in frequency.  In: a = Z[lane], b = Z[lane+64].
 * Out: lane holds z[2q] (a), z[2q+1] (b), q = bitrev6(lane). */
__device__ __forceinline__ void fft128_inv(float2& a, float2& b, const LaneTw& tw, int lane)
{
    float2 s = make_float2(a.x + b.x, a.y + b.y);
    float2 d = cmulc(make_float2(a.x - b.x, a.y - b.y), tw.w128);
    a = s; b = d;
#pragma unroll
    for (int st = 5; st >= 0; st--) {
        const int h = 1 << st;
        const bool up = (lane & h) != 0;
        float2 pa = shflx(a, h), pb = shflx(b, h);
        a = up ? cmulc(make_float2(pa.x - a.x, pa.y - a.y), tw.st[st]) : make_float2(a.x + pa.x, a.y + pa.y);
        b = up ? cmulc(make_float2(pb.x - b.x, pb.y - b.y), tw.st[st]) : make_float2(b.x + pb.x, b.y + pb.y);
    }
}

/* ========================================================================== */
/*                                 analysis                                   */
/* ========================================================================== */

struct AnaArgs {
    AnaLaunch a;
    const float* win;
    const float2* tw;
    int chunk;
};

__device__ __forceinline__ float4 ana_load4(const AnaArgs& g, int inst, int ch, int srcch, bool valid, float scale, int hop, int c4)
{
    if (hop < 0) {
        const float* p = g.a.hist_rd + (((long long)inst * g.a.nCh + ch) * SAF_ANA_HIST + (SAF_ANA_HIST + hop)) * SAF_HOP + c4 * 4;
        return *reinterpret_cast<const float4*>(p);
    }
    if (!valid) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int frame = hop / g.a.hopsPerFrame, sub = hop - frame * g.a.hopsPerFrame;
    const float* p = g.a.in + (long long)inst * g.a.in_inst + (long long)frame * g.a.in_frame + (long long)srcch * g.a.in_ch + sub * SAF_HOP + c4 * 4;
    float4 v = *reinterpret_cast<const float4*>(p);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    return v;
}

__global__ __launch_bounds__(256) void afstft_analysis_kernel(AnaArgs g)
{
    __shared__ __attribute__((aligned(16))) float s_in[(SUB + 9) * SAF_HOP];
    __shared__ float s_re[RING * SPEC_LD];
    __shared__ float s_im[RING * SPEC_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = blockIdx.y, inst = blockIdx.z;
    const int c0 = blockIdx.x * g.chunk;
    const int c1 = min(c0 + g.chunk, g.a.H);
    if (c0 >= c1) return;

    const int tabStride = g.a.tab_stride ? g.a.tab_stride : g.a.nCh;
    const int srcch = g.a.ch_map ? g.a.ch_map[inst * tabStride + ch] : ch;
    const bool valid = srcch >= 0 && srcch < g.a.nChIn;
    const float scale = g.a.ch_scale ? g.a.ch_scale[inst * tabStride + ch] : 1.0f;
    const int nBandsOut = g.a.hybrid ? SAF_NBANDS : SAF_NBINS;

    /* per-lane constants: which 4 folded samples this lane produces, and their 5 window taps each */
    const int q = bitrev6(lane);
    const int odd = q >> 5;                 /* 0: first half of the 256-frame (even k), 1: second half (odd k) */
    const int n0 = 4 * (q & 31);
    float4 w[5];
#pragma unroll
    for (int i = 0; i < 5; i++) w[i] = *reinterpret_cast<const float4*>(g.win + (2 * i + odd) * SAF_HOP + n0);
    const LaneTw tw = load_tw(g.tw, lane);

    for (int s0 = c0 - 6; s0 < c1;) {
        const int n = (s0 < c0) ? 6 : min(SUB, c1 - s0);
        /* stage input hops s0-9 .. s0+n-1 */
        for (int idx = tid; idx < (n + 9) * 32; idx += 256) {
            const int row = idx >> 5, c4 = idx & 31;
            float4 v = ana_load4(g, inst, ch, srcch, valid, scale, s0 - 9 + row, c4);
            *reinterpret_cast<float4*>(&s_in[row * SAF_HOP + c4 * 4]) = v;
        }
        __syncthreads();
        for (int t = wave; t < n; t += 4) {
            /* 1280-tap window + fold: f[(k&1)*128 + n] = sum_k x[hop-9+k][n] * w[k*128+n]   (afSTFT_internal.c:276-301) */
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int k = 2 * i + odd;
                const float4 x = *reinterpret_cast<const float4*>(&s_in[(t + k) * SAF_HOP + n0]);
                acc.x = fmaf(x.x, w[i].x, acc.x); acc.y = fmaf(x.y, w[i].y, acc.y);
                acc.z = fmaf(x.z, w[i].z, acc.z); acc.w = fmaf(x.w, w[i].w, acc.w);
            }
            float2 A, B;
            fft128_fwd(make_float2(acc.x, acc.y), make_float2(acc.z, acc.w), tw, lane, A, B);
            /* real-FFT split: X[k], X[128-k] from Z[k], Z[128-k]  (kiss_fftr.c:86-123 convention) */
            const float2 Zmk = shfl(B, (64 - lane) & 63);
            const float2 f1 = make_float2(A.x + Zmk.x, A.y - Zmk.y);
            const float2 f2 = make_float2(A.x - Zmk.x, A.y + Zmk.y);
            const float2 tt = cmul(f2, tw.w256);
            float2 Xk = make_float2(0.5f * (f1.x + tt.y), 0.5f * (f1.y - tt.x));
            float2 Xmk = make_float2(0.5f * (f1.x - tt.y), 0.5f * (-f1.y - tt.x));
            const int slot = (s0 + t - c0 + 6) % RING;
            float* re = &s_re[slot * SPEC_LD];
            float* im = &s_im[slot * SPEC_LD];
            if (lane == 0) {
                re[0] = A.x + A.y;   im[0] = 0.f;          /* DC */
                re[128] = A.x - A.y; im[128] = 0.f;        /* Nyquist */
                re[64] = B.x;        im[64] = -B.y;        /* bin 64 = conj(Z[64]) */
            } else {
                re[lane] = Xk.x;        im[lane] = Xk.y;
                re[128 - lane] = Xmk.x; im[128 - lane] = Xmk.y;
            }
        }
        __syncthreads();
        if (s0 >= c0) {
            /* hybrid split + 3-hop delay (afSTFT_internal.c:523-623), stored time-contiguous */
            for (int idx = tid; idx < nBandsOut * SUB; idx += 256) {
                const int band = idx >> 4, t = idx & 15;
                if (t >= n) continue;
                const int hop = s0 + t;
                const int base = hop - c0 + 6;              /* ring position of S_hop */
                const int sD = (base - 3) % RING;
                float2 v;
                if (!g.a.hybrid) {
                    const int p0 = base % RING;             /* plain STFT bins, no hybrid delay */
                    v = make_float2(s_re[p0 * SPEC_LD + band], s_im[p0 * SPEC_LD + band]);
                } else if (band == 0 || band >= 9) {
                    const int bin = band == 0 ? 0 : band - 4;
                    v = make_float2(s_re[sD * SPEC_LD + bin], s_im[sD * SPEC_LD + bin]);
                } else {
                    const int b = (band + 1) >> 1;
                    const int p0 = base % RING, p2 = (base - 2) % RING, p4 = (base - 4) % RING, p6 = (base - 6) % RING;
                    float gr, gi;
                    gr = -COEFF1 * s_im[p0 * SPEC_LD + b];
                    gi =  COEFF1 * s_re[p0 * SPEC_LD + b];
                    gr -= COEFF2 * s_im[p2 * SPEC_LD + b];
                    gi += COEFF2 * s_re[p2 * SPEC_LD + b];
                    gr += COEFF2 * s_im[p4 * SPEC_LD + b];
                    gi -= COEFF2 * s_re[p4 * SPEC_LD + b];
                    gr += COEFF1 * s_im[p6 * SPEC_LD + b];
                    gi -= COEFF1 * s_re[p6 * SPEC_LD + b];
                    const float dr = s_re[sD * SPEC_LD + b] * 0.5f, di = s_im[sD * SPEC_LD + b] * 0.5f;
                    /* lower half-band (odd band index) of bins 1,3 subtracts, of bins 2,4 adds (afSTFT_internal.c:606-619) */
                    const bool lower = (band & 1) != 0;
                    const bool minus = ((b & 1) != 0) == lower;
                    v = minus ? make_float2(dr - gr, di - gi) : make_float2(dr + gr, di + gi);
                }
                g.a.out[(long long)inst * g.a.out_inst + (long long)band * g.a.out_band + (long long)ch * g.a.out_ch + hop] = v;
            }
        }
        s0 += n;
        __syncthreads();
    }

    /* the workgroup that owns the last chunk records the new input history */
    if (c1 == g.a.H && g.a.hist_wr) {
        for (int idx = tid; idx < SAF_ANA_HIST * 32; idx += 256) {
            const int row = idx >> 5, c4 = idx & 31;
            const int hop = g.a.H - SAF_ANA_HIST + row;
            float4 v = ana_load4(g, inst, ch, srcch, valid, scale, hop, c4);
            float* p = g.a.hist_wr + (((long long)inst * g.a.nCh + ch) * SAF_ANA_HIST + row) * SAF_HOP + c4 * 4;
            *reinterpret_cast<float4*>(p) = v;
        }
    }
}

/* ========================================================================== */
/*                                 synthesis                                  */
/* ========================================================================== */

struct SynArgs {
    SynLaunch s;
    const float* win;
    const float2* tw;
};

__device__ __forceinline__ int g_pad(int n) { return n + 4 * (n >> 5); }

__global__ __launch_bounds__(256) void afstft_synthesis_kernel(SynArgs g)
{
    __shared__ float s_re[SUB * SPEC_LD];
    __shared__ float s_im[SUB * SPEC_LD];
    __shared__ __attribute__((aligned(16))) float s_g[SUB * G_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = blockIdx.x, inst = blockIdx.y;
    const int H = g.s.H;
    const LaneTw tw = load_tw(g.tw, lane);
    const int q = bitrev6(lane);

    /* overlap-add state of thread n (<128): the last 9 synthesised frames at n (first half) and 128+n (second half) */
    float gl[10], gr[10], wn[10];
    if (tid < 128) {
#pragma unroll
        for (int k = 0; k < 10; k++) wn[k] = g.win[k * SAF_HOP + tid];
        const float* h = g.s.hist_rd + ((long long)inst * g.s.nCh + ch) * SAF_SYN_HIST * 256;
#pragma unroll
        for (int k = 1; k < 10; k++) { gl[k] = h[(9 - k) * 256 + tid]; gr[k] = h[(9 - k) * 256 + 128 + tid]; }
        gl[0] = gr[0] = 0.f;
    }

    for (int s0 = 0; s0 < H; s0 += SUB) {
        const int n = min(SUB, H - s0);
        /* gather bands -> bins (afHybridInverse, afSTFT_internal.c:625-653): reads are time-contiguous */
        for (int idx = tid; idx < SAF_NBINS * SUB; idx += 256) {
            const int bin = idx >> 4, t = idx & 15;
            if (t >= n) continue;
            const float2* p = g.s.in + (long long)inst * g.s.in_inst + (long long)ch * g.s.in_ch + (s0 + t);
            float2 v;
            if (bin == 0) v = p[0];
            else if (!g.s.hybrid) v = p[(long long)bin * g.s.in_band];
            else if (bin < 5) {
                const float2 a = p[(long long)(2 * bin - 1) * g.s.in_band], b = p[(long long)(2 * bin) * g.s.in_band];
                v = make_float2(a.x + b.x, a.y + b.y);
            } else v = p[(long long)(bin + 4) * g.s.in_band];
            /* low-delay mode: odd bins change sign = circular half-frame shift (afSTFT_internal.c:366-369) */
            if (g.s.lowDelay && (bin & 1) && bin < SAF_HOP) { v.x = -v.x; v.y = -v.y; }
            s_re[t * SPEC_LD + bin] = v.x;
            s_im[t * SPEC_LD + bin] = v.y;
        }
        __syncthreads();
        for (int t = wave; t < n; t += 4) {
            const float* re = &s_re[t * SPEC_LD];
            const float* im = &s_im[t * SPEC_LD];
            /* half-complex -> packed: Z[k], Z[128-k] from X[k], X[128-k]  (kiss_fftr.c:125-161); Im of DC/Nyquist ignored */
            const int k = lane;
            const float2 fk = make_float2(re[k], im[k]);
            const float2 fnkc = make_float2(re[128 - k], -im[128 - k]);
            const float2 fek = make_float2(fk.x + fnkc.x, fk.y + fnkc.y);
            const float2 tmp = make_float2(fk.x - fnkc.x, fk.y - fnkc.y);
            const float2 fok = cmulc(tmp, tw.w256);                         /* * e^{+2 pi i k/256} */
            float2 Zk = make_float2(fek.x - fok.y, fek.y + fok.x);          /* fek + i fok */
            float2 Zmk = make_float2(fek.x + fok.y, -(fek.y - fok.x));      /* conj(fek - i fok) */
            if (lane == 0) {
                Zk = make_float2(re[0] + re[128], re[0] - re[128]);
                Zmk = make_float2(2.0f * re[64], -2.0f * im[64]);            /* Z[64] = 2 conj(X[64]) */
            }
            float2 a = Zk;
            float2 b = shfl(Zmk, (64 - lane) & 63);                          /* Z[lane + 64] */
            fft128_inv(a, b, tw, lane);
            const float sc = 1.0f / 256.0f;                                  /* saf_rfft_backward scaling (saf_utility_fft.c:751) */
            float4 o = make_float4(a.x * sc, a.y * sc, b.x * sc, b.y * sc); /* frame samples 4q .. 4q+3 */
            *reinterpret_cast<float4*>(&s_g[t * G_LD + g_pad(4 * q)]) = o;
        }
        __syncthreads();
        if (tid < 128) {
            for (int t = 0; t < n; t++) {
                gl[0] = s_g[t * G_LD + g_pad(tid)];
                gr[0] = s_g[t * G_LD + g_pad(128 + tid)];
                /* 10-segment overlap-add, oldest frame first (afSTFT_internal.c:396-444) */
                float acc = 0.f;
#pragma unroll
                for (int k = 9; k >= 0; k--) acc = fmaf(wn[k], (k & 1) ? gr[k] : gl[k], acc);
                const int hop = s0 + t;
                const int frame = hop / g.s.hopsPerFrame, sub = hop - frame * g.s.hopsPerFrame;
                g.s.out[(long long)inst * g.s.out_inst + (long long)frame * g.s.out_frame + (long long)ch * g.s.out_ch + sub * SAF_HOP + tid] = acc;
#pragma unroll
                for (int k = 9; k >= 1; k--) { gl[k] = gl[k - 1]; gr[k] = gr[k - 1]; }
            }
        }
        __syncthreads();
    }
    if (tid < 128 && g.s.hist_wr) {
        float* h = g.s.hist_wr + ((long long)inst * g.s.nCh + ch) * SAF_SYN_HIST * 256;
#pragma unroll
        for (int k = 1; k < 10; k++) { h[(9 - k) * 256 + tid] = gl[k]; h[(9 - k) * 256 + 128 + tid] = gr[k]; }
    }
}

/* ========================================================================== */
/*                        constant tables + launchers                         */
/* ========================================================================== */

static float* g_dev_win[2][2] = { { nullptr, nullptr }, { nullptr, nullptr } };
static float2* g_dev_tw = nullptr;

const float* dev_window(int lowDelay, int synthesis)
{
    float*& d = g_dev_win[lowDelay ? 1 : 0][synthesis ? 1 : 0];
    if (d) return d;
    /* afSTFTlib_init, hop 128 (afSTFT_internal.c:122-145): every 8th tap of the 10240-tap prototype,
     * reversed, times eq; the low-delay synthesis window is not reversed. */
    const float* p = table_required(lowDelay ? "afSTFT_protoFilter1024LD" : "afSTFT_protoFilter1024", 10240);
    const float eq = lowDelay ? 2.0f / sqrtf(4.544559956f) : 2.0f / sqrtf(5.487604141f);
    std::vector<float> w(1280);
    for (int k = 0; k < 1280; k++) {
        const float v = p[k * 8] * eq;
        if (lowDelay && synthesis) w[k] = v; else w[1280 - k - 1] = v;
    }
    HIP_CHECK(hipMalloc((void**)&d, 1280 * sizeof(float)));
    HIP_CHECK(hipMemcpy(d, w.data(), 1280 * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

const float2* dev_twiddles()
{
    if (g_dev_tw) return g_dev_tw;
    std::vector<float2> t(8 * 64);
    for (int s = 0; s < 6; s++) {
        const int h = 1 << s;
        for (int l = 0; l < 64; l++) {
            const double a = -2.0 * SAF_PId * (double)(l & (h - 1)) / (double)(2 * h);
            t[s * 64 + l] = make_float2((float)cos(a), (float)sin(a));
        }
    }
    for (int l = 0; l < 64; l++) {
        double a = -2.0 * SAF_PId * (double)l / 128.0;
        t[6 * 64 + l] = make_float2((float)cos(a), (float)sin(a));
        a = -2.0 * SAF_PId * (double)l / 256.0;
        t[7 * 64 + l] = make_float2((float)cos(a), (float)sin(a));
    }
    HIP_CHECK(hipMalloc((void**)&g_dev_tw, t.size() * sizeof(float2)));
    HIP_CHECK(hipMemcpy(g_dev_tw, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice));
    return g_dev_tw;
}

void launch_analysis(const AnaLaunch& a)
{
    if (a.H <= 0 || a.nCh <= 0 || a.nInst <= 0) return;
    if ((a.in_inst | a.in_ch | a.in_frame) & 3) SAF_FATAL("analysis: sample strides must be multiples of 4 floats");
    if (((uintptr_t)a.in & 15) != 0) SAF_FATAL("analysis: input must be 16-byte aligned");
    AnaArgs g;
    g.a = a;
    g.win = dev_window(a.lowDelay, 0);
    g.tw = dev_twiddles();
    /* time chunks add parallelism when few (instance, channel) pairs are in flight; each extra chunk
     * recomputes 6 warm-up FFTs, so long launches use long chunks */
    const long long pairs = (long long)a.nCh * a.nInst;
    int chunk = a.H;
    if (pairs < 2048 && a.H > 32) chunk = 32;
    if (pairs < 512 && a.H > 16) chunk = 16;
    g.chunk = chunk;
    dim3 grid((a.H + chunk - 1) / chunk, a.nCh, a.nInst);
    KernelTimer kt("afstft_analysis");
    hipLaunchKernelGGL(afstft_analysis_kernel, grid, dim3(256), 0, stream(), g);
    HIP_CHECK(hipGetLastError());
}

void launch_synthesis(const SynLaunch& s)
{
    if (s.H <= 0 || s.nCh <= 0 || s.nInst <= 0) return;
    SynArgs g;
    g.s = s;
    g.win = dev_window(s.lowDelay, 1);
    g.tw = dev_twiddles();
    dim3 grid(s.nCh, s.nInst);
    KernelTimer kt("afstft_synthesis");
    hipLaunchKernelGGL(afstft_synthesis_kernel, grid, dim3(256), 0, stream(), g);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
