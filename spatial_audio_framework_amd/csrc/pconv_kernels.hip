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
 * A real FFT of size N is a complex FFT of M = N/2 points followed by the usual split.  N is a power of two >= 2*hop;
 * any N >= hop + partitionLength - 1 gives the same linear convolution as the reference's N = 2*hop /
 * N = numOvrlpAddBlocks*hop, so arbitrary hop sizes are supported.
 *
 * The M-point FFT is a Stockham autosort FFT in LDS with ONE workgroup of M/8 threads per transform: every thread owns
 * the 8 elements tid + q*M/8 in every pass (conflict-free, coalesced), does one radix-8 butterfly in registers (the
 * first pass is radix 4 or 2 when log2 M is not a multiple of 3) and scatters its outputs; log8(M) passes, two barriers
 * each — a 1024-point real FFT is 3 passes of ONE wave, so its barriers are free and 16+ transforms run per CU.
 * (The first version: radix 2, 256 threads, 9 barrier-separated stages; 25 us for 4096 transforms, 31 us for 32.)
 *
 * Spectra rows hold exactly M complex numbers: bin 0 carries (Re X[0], Re X[M]) — both are real — so the rows are
 * power-of-two sized and the MAC treats bin 0 as two real products.
 */
#include "saf_hip_common.h"
#include "fft_butterflies.h"

namespace saf {

#define PC_PAD(i) ((i) + ((i) >> 3))          /* LDS index of element i: one pad element per 8 (radix-8 scatter without bank conflicts) */

/* exp(-2 pi i k / M) for 0 <= k < M from the half-circle table s_tw[k] = exp(-2 pi i k / M), k < M/2 */
__device__ __forceinline__ float2 pc_tw(const float2* s_tw, int k, int halfM)
{
    const float2 w = s_tw[k & (halfM - 1)];
    return k >= halfM ? make_float2(-w.x, -w.y) : w;
}

/* M-point complex FFT of the sequence whose elements tid + q*M/8 this thread holds in v[q]; the result is left in
 * natural order in s (padded: element i at PC_PAD(i)).  nthr = M/8 threads take part (tid < nthr); all threads of the
 * workgroup must call (barriers).  s_tw: half-circle twiddles in LDS.  INV: conjugated twiddles, unscaled. */
/* LOGM > 0: transform size known at compile time (passes unrolled, strides and twiddle steps constant); LOGM = 0: the
 * runtime M / logM are used. */
template <bool INV, int LOGM>
__device__ __forceinline__ void pc_fft(float2 (&v)[8], float2* s, const float2* s_tw, int M_rt, int logM_rt, int tid)
{
    const int logM = LOGM > 0 ? LOGM : logM_rt;
    const int M = LOGM > 0 ? (1 << LOGM) : M_rt;
    const int M8 = M >> 3, halfM = M >> 1;
    const bool on = tid < M8;
    int Ns = 1;
    /* first pass: radix 2 or 4 when log2 M is not a multiple of 3 (no twiddles at Ns = 1) */
    const int r0 = logM % 3;
    if (r0 == 1) {
        /* 4 radix-2 butterflies: jj = tid + u*M8, inputs v[u], v[u + 4]; outputs at 2*jj, 2*jj + 1 */
#pragma unroll
        for (int u = 0; u < 4; u++) { const float2 a = v[u], b = v[u + 4]; v[u] = cadd(a, b); v[u + 4] = csub(a, b); }
        __syncthreads();
        if (on) {
#pragma unroll
            for (int u = 0; u < 4; u++) { const int o = 2 * (tid + u * M8); s[PC_PAD(o)] = v[u]; s[PC_PAD(o + 1)] = v[u + 4]; }
        }
        Ns = 2;
    } else if (r0 == 2) {
        /* 2 radix-4 butterflies: jj = tid + u*M8, inputs v[u + 2r]; outputs at 4*jj + r */
#pragma unroll
        for (int u = 0; u < 2; u++) dft4<INV>(v[u], v[u + 2], v[u + 4], v[u + 6]);
        __syncthreads();
        if (on) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int o = 4 * (tid + u * M8);
#pragma unroll
                for (int r = 0; r < 4; r++) s[PC_PAD(o + r)] = v[u + 2 * r];
            }
        }
        Ns = 4;
    }
    if (r0 != 0) {
        __syncthreads();
        if (on) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = s[PC_PAD(tid + q * M8)];
        }
    }
    auto pass = [&](int Ns_, bool last) {
        const int jm = tid & (Ns_ - 1);
        if (Ns_ > 1) {
            const int step = jm * (M / (8 * Ns_));               /* exponent of the r = 1 twiddle in units of 1/M turns */
#pragma unroll
            for (int r = 1; r < 8; r++) { float2 w = pc_tw(s_tw, r * step, halfM); if (INV) w.y = -w.y; v[r] = cmul(v[r], w); }
        }
        dft8<INV>(v);
        __syncthreads();                                         /* every thread has read its inputs of this pass */
        if (on) {
            const int base = ((tid - jm) << 3) + jm;
#pragma unroll
            for (int r = 0; r < 8; r++) s[PC_PAD(base + r * Ns_)] = X8(v, r);
        }
        __syncthreads();
        if (!last && on) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = s[PC_PAD(tid + q * M8)];
        }
    };
    if (LOGM > 0) {
#pragma unroll
        for (int p = 0; p < LOGM / 3; p++) pass(((LOGM % 3) == 0 ? 1 : ((LOGM % 3) == 1 ? 2 : 4)) << (3 * p), p == LOGM / 3 - 1);
    } else {
        for (int rem = logM - r0; rem > 0; rem -= 3) { pass(Ns, rem == 3); Ns <<= 3; }
    }
}

struct FwdArgs {
    const float* src; long long s0, s1, s2;       /* element (x, y, z) of the grid starts at src + x*s0 + y*s1 + z*s2 */
    int nValid;                                   /* samples taken from src, the rest of the N-frame is zero */
    int yValidStep, yValidTotal;                  /* if yValidStep > 0: nValid = clamp(yValidTotal - y*yValidStep, 0, nValid) (last filter partition) */
    float2* dst; long long d0, d1, d2;
    int ringLen, ringHead;                        /* y is a time index: its slot is (ringHead + y) % ringLen (ringLen 0: plain y) */
    const float2* tw;                             /* exp(-2 pi i k / N), k < M */
    int M, logM;
    int g0, fpw;                                  /* transforms along x; transforms per workgroup */
};

/* grid (ceil(g0 / fpw), g1, g2); fpw transforms per workgroup, max(64, M/8) threads each (one-wave workgroups are
 * bound by the workgroup dispatch rate: 4096 of them took 19 us whatever they did).
 * dynamic LDS: fpw x (M + M/8) data + M/2 twiddles, float2 each */
template <int LOGM>
__global__ __launch_bounds__(1024) void pconv_rfft_fwd_kernel(FwdArgs a)
{
    extern __shared__ float2 s_pc[];
    const int M = LOGM > 0 ? (1 << LOGM) : a.M, logM = LOGM > 0 ? LOGM : a.logM, M8 = M >> 3;
    const int tpf = (int)blockDim.x / a.fpw;                       /* threads per transform */
    const int sub = threadIdx.x / tpf, tid = threadIdx.x - sub * tpf;
    const int fx = blockIdx.x * a.fpw + sub;
    float2* s_tw = s_pc;
    float2* s = s_pc + (M >> 1) + sub * (M + M8);
    const bool on = tid < M8 && fx < a.g0;
    const int tc = tid < M8 ? tid : 0;
    const int fxc = fx < a.g0 ? fx : a.g0 - 1;
    const float* src = a.src + (long long)fxc * a.s0 + (long long)blockIdx.y * a.s1 + (long long)blockIdx.z * a.s2;
    int nValid = a.nValid;
    if (a.yValidStep > 0) { int r = a.yValidTotal - (int)blockIdx.y * a.yValidStep; nValid = r < 0 ? 0 : (r < nValid ? r : nValid); }
    if (nValid <= 0) src = reinterpret_cast<const float*>(a.tw);         /* nothing to read: any valid address, every value is masked */
    const int last = nValid > 0 ? nValid - 1 : 0;
    /* z[m] = x[2m] + i x[2m+1], m = tid + q*M/8: unconditional loads (clamped index, masked value) */
    float2 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int i0 = 2 * (tc + q * M8), i1 = i0 + 1;
        const float re = src[i0 < last ? i0 : last], im = src[i1 < last ? i1 : last];
        v[q] = make_float2(i0 < nValid ? re : 0.0f, i1 < nValid ? im : 0.0f);
    }
    /* split twiddles of this thread's 8 bins and the FFT's half-circle table (exp(-2 pi i k / M) = tw[2k]) */
    float2 wk[8];
#pragma unroll
    for (int i = 0; i < 8; i++) wk[i] = a.tw[tc + i * M8];
    for (int k = threadIdx.x; k < (M >> 1); k += blockDim.x) s_tw[k] = a.tw[2 * k];
    __syncthreads();
    pc_fft<false, LOGM>(v, s, s_tw, M, logM, tid < M8 ? tid : M8);             /* tid >= M/8: takes part in the barriers only */
    if (!on) return;
    const int y = a.ringLen ? (a.ringHead + (int)blockIdx.y) % a.ringLen : (int)blockIdx.y;
    float2* dst = a.dst + (long long)fx * a.d0 + (long long)y * a.d1 + (long long)blockIdx.z * a.d2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = tid + i * M8;
        const float2 Zk = s[PC_PAD(k)], Zm = s[PC_PAD((M - k) & (M - 1))];
        const float2 e = make_float2(Zk.x + Zm.x, Zk.y - Zm.y);
        const float2 d = make_float2(Zk.x - Zm.x, Zk.y + Zm.y);
        const float2 t = cmul(wk[i], d);
        float2 X = make_float2(0.5f * (e.x + t.y), 0.5f * (e.y - t.x));
        if (k == 0) X = make_float2(Zk.x + Zk.y, Zk.x - Zk.y);             /* (Re X[0], Re X[M]) */
        dst[k] = X;
    }
}

/* -------------------------------------------------------------------------- */

#define MAC_TB 4      /* consecutive blocks that share one pass over the filter spectra */

struct MacArgs {
    const float2* Hf;      /* [nOut][nTerms = nFB*nIn][M] */
    const float2* Xr;      /* [ringLen][nX][M] */
    float2* P;             /* partial sums [T][nOut][kSplit][M] */
    int nIn, nFB, nOut, M, kSplit, termsPerSplit, tGroups;
    int ringLen, ringHead; /* slot of block t of this call = (ringHead + t) % ringLen */
    int T;
    int nX, diag;          /* channels per ring slot; diag = 1: output o convolves input channel o only (saf_multiConv), nIn = 1 */
};

/* spectral product with bin 0 = two real products (packed DC / Nyquist): hA = b0 ? 0 : h.y, hB = b0 ? h.y : h.x, xs = b0 ? 0 : x.x */
__device__ __forceinline__ void pc_mac(float2& acc, float2 h, float hA, float hB, float2 x, bool b0)
{
    const float xs = b0 ? 0.0f : x.x;
    acc.x = fmaf(h.x, x.x, acc.x); acc.x = fmaf(-hA, x.y, acc.x);
    acc.y = fmaf(hB, x.y, acc.y);  acc.y = fmaf(h.y, xs, acc.y);
}

#define MAC_BINS 16   /* bins per workgroup: 16 lanes x 8 B = one 128-byte line of every spectrum row */
#define MAC_TG   16   /* term groups per workgroup (256 threads = 16 bins x 16 groups) */

/* grid (ceil(M/16), ceil(nOut/OB), kSplit * tGroups).
 * A workgroup forms the sums of OB outputs x MAC_TB blocks over its share of the (partition, input) terms for 16 bins:
 * the input spectra are read once for both outputs, the filter spectra once for the 4 blocks; narrow bin tiles give
 * enough workgroups without splitting the term sum finely.  Term loops are uniform per 16-lane group and every load is
 * unconditional.  The kSplit partial sums are folded by the inverse-transform kernel.  (Folding them here, in the last
 * workgroup of a tile to arrive, needs device-scope fences: on this 8-XCD part they write back and invalidate the XCD's L2
 * — measured 85 us for this kernel instead of 11.) */
template <int OB>
__global__ __launch_bounds__(256) void pconv_mac_kernel(MacArgs a)
{
    __shared__ float2 s_red[OB * MAC_TB][MAC_TG][MAC_BINS];
    const int b = threadIdx.x & (MAC_BINS - 1), q = threadIdx.x / MAC_BINS;
    const int bin = blockIdx.x * MAC_BINS + b;
    const int binc = bin < a.M ? bin : a.M - 1;
    const bool b0 = binc == 0;
    const int o0 = blockIdx.y * OB;
    const int ks = blockIdx.z / a.tGroups, tg = blockIdx.z - ks * a.tGroups;
    const int t0 = tg * MAC_TB;
    const int nTerms = a.nFB * a.nIn;
    const int k0 = ks * a.termsPerSplit;
    const int k1 = k0 + a.termsPerSplit < nTerms ? k0 + a.termsPerSplit : nTerms;
    const float2* H[OB];
#pragma unroll
    for (int j = 0; j < OB; j++) { const int o = o0 + j < a.nOut ? o0 + j : a.nOut - 1; H[j] = a.Hf + (long long)o * nTerms * a.M + binc; }
    const float2* Xb = a.Xr + binc + (a.diag ? (long long)o0 * a.M : 0);
    const long long slotStride = (long long)a.nX * a.M;
    float2 acc[OB][MAC_TB];
#pragma unroll
    for (int j = 0; j < OB; j++)
#pragma unroll
        for (int u = 0; u < MAC_TB; u++) acc[j][u] = make_float2(0.f, 0.f);
    int k = k0 + q;
    int p = k / a.nIn, i = k - p * a.nIn;
    for (; k < k1; k += MAC_TG) {
        float2 h[OB];
#pragma unroll
        for (int j = 0; j < OB; j++) h[j] = H[j][(long long)k * a.M];
        int slot = (a.ringHead + t0 - p) % a.ringLen; if (slot < 0) slot += a.ringLen;
        float2 x[MAC_TB];
#pragma unroll
        for (int u = 0; u < MAC_TB; u++) {
            x[u] = Xb[(long long)slot * slotStride + (long long)i * a.M];      /* blocks beyond T read a valid slot; never stored */
            slot++; if (slot == a.ringLen) slot = 0;
        }
#pragma unroll
        for (int j = 0; j < OB; j++) {
            const float hA = b0 ? 0.0f : h[j].y, hB = b0 ? h[j].y : h[j].x;
#pragma unroll
            for (int u = 0; u < MAC_TB; u++) pc_mac(acc[j][u], h[j], hA, hB, x[u], b0);
        }
        i += MAC_TG; while (i >= a.nIn) { i -= a.nIn; p++; }
    }
#pragma unroll
    for (int j = 0; j < OB; j++)
#pragma unroll
        for (int u = 0; u < MAC_TB; u++) s_red[j * MAC_TB + u][q][b] = acc[j][u];
    __syncthreads();
    /* threads 0 .. OB*MAC_TB*16-1 fold the 16 term groups: one (output, block, bin) each */
    const int e = threadIdx.x;
    const int ju = e / MAC_BINS, bb = e & (MAC_BINS - 1), j = ju / MAC_TB, u = ju - j * MAC_TB;
    const int gb = blockIdx.x * MAC_BINS + bb;
    const bool mine = e < OB * MAC_TB * MAC_BINS && gb < a.M && o0 + j < a.nOut && t0 + u < a.T;
    const long long row = mine ? (long long)(t0 + u) * a.nOut + o0 + j : 0;
    if (mine) {
        float2 r = s_red[ju][0][bb];
#pragma unroll
        for (int qq = 1; qq < MAC_TG; qq++) { r.x += s_red[ju][qq][bb].x; r.y += s_red[ju][qq][bb].y; }
        a.P[(row * a.kSplit + ks) * a.M + gb] = r;
    }
}

/* -------------------------------------------------------------------------- */

struct InvArgs {
    const float2* P;       /* [T][nOut][kSplit][M] */
    float* zs;             /* [zRing][nOut][N] */
    const float2* tw;
    int nOut, kSplit, M, logM;
    int zRing, zHead;      /* slot of block t = (zHead + t) % zRing */
};

/* grid (nOut, T); max(256, M/8) threads, the first M/8 of which run the FFT.  Folds the kSplit partial sums (all
 * threads: group g = tid / (M/8) takes the splits g, g + G, ...; the groups meet in LDS), rebuilds the packed
 * half-size spectrum, inverse FFT, scale 1/M (= the 1/N of the real transform times the 2 of the packing).
 * dynamic LDS: (M + M/8) data + M/2 twiddles + G*M fold, float2 each. */
template <int LOGM>
__global__ __launch_bounds__(1024) void pconv_irfft_kernel(InvArgs a)
{
    extern __shared__ float2 s_pc[];
    const int M = LOGM > 0 ? (1 << LOGM) : a.M, logM = LOGM > 0 ? LOGM : a.logM, M8 = M >> 3, tid = threadIdx.x, o = blockIdx.x, t = blockIdx.y;
    float2* s = s_pc;
    float2* s_tw = s_pc + M + M8;
    float2* s_fold = s_tw + (M >> 1);
    const bool on = tid < M8;
    const int G = (int)blockDim.x / M8 > 0 ? (int)blockDim.x / M8 : 1;           /* fold groups */
    const int g = tid / M8, tb = tid - g * M8;                                  /* threads beyond G*M8 (M8 not a divisor) idle */
    const int tc = on ? tid : 0;
    const float2* P = a.P + ((long long)t * a.nOut + o) * a.kSplit * M;
    float2 X[8], wk[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { X[i] = make_float2(0.f, 0.f); wk[i] = a.tw[tc + i * M8]; }
    if (g < G)
        for (int sp = g; sp < a.kSplit; sp += G) {
#pragma unroll
            for (int i = 0; i < 8; i++) { const float2 u = P[(long long)sp * M + tb + i * M8]; X[i].x += u.x; X[i].y += u.y; }
        }
    for (int k = tid; k < (M >> 1); k += blockDim.x) s_tw[k] = a.tw[2 * k];
    if (G > 1) {
        if (g > 0 && g < G) {
#pragma unroll
            for (int i = 0; i < 8; i++) s_fold[(g - 1) * M + tb + i * M8] = X[i];
        }
        __syncthreads();
        if (on) {
            for (int gg = 1; gg < G; gg++) {
#pragma unroll
                for (int i = 0; i < 8; i++) { const float2 u = s_fold[(gg - 1) * M + tid + i * M8]; X[i].x += u.x; X[i].y += u.y; }
            }
        }
    }
    if (on) {
#pragma unroll
        for (int i = 0; i < 8; i++) s[PC_PAD(tid + i * M8)] = X[i];
    }
    __syncthreads();
    float2 Z[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = tc + i * M8;
        const float2 Xk = X[i], Xm = s[PC_PAD((M - k) & (M - 1))];
        const float2 E = make_float2(0.5f * (Xk.x + Xm.x), 0.5f * (Xk.y - Xm.y));
        const float2 D = make_float2(0.5f * (Xk.x - Xm.x), 0.5f * (Xk.y + Xm.y));
        float2 W = wk[i]; W.y = -W.y;                                   /* e^{+2 pi i k / N} */
        const float2 O = cmul(D, W);
        Z[i] = make_float2(E.x - O.y, E.y + O.x);                       /* E + i O */
        if (k == 0) Z[i] = make_float2(0.5f * (Xk.x + Xk.y), 0.5f * (Xk.x - Xk.y));     /* packed (Re X[0], Re X[M]) */
    }
    /* Z[i] is element tid + i*M/8 of the half-size spectrum: exactly the registers the FFT starts from */
    pc_fft<true, LOGM>(Z, s, s_tw, M, logM, tid);
    if (!on) return;
    const int slot = (a.zHead + t) % a.zRing;
    float2* z = reinterpret_cast<float2*>(a.zs + ((long long)slot * a.nOut + o) * (2 * M));
    const float sc = 1.0f / (float)M;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int m = tid + i * M8;
        const float2 vv = s[PC_PAD(m)];
        z[m] = make_float2(vv.x * sc, vv.y * sc);
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
    const float2* Hf;      /* [nIRs][nOut][nFB][M] */
    const float2* Xr;      /* [ringLen][M] (one input channel) */
    float2* P;             /* [T][nOut*3][M] */
    const int* irSel;      /* [T][3]: IR of block t, of t-1, of t-2 */
    int nFB, nOut, M, ringLen, ringHead;
};

/* grid (ceil(M/64), nOut*3, T); 256 threads = 64 bins x 4 partition groups */
__global__ __launch_bounds__(256) void tvconv_mac_kernel(TvMacArgs a)
{
    __shared__ float2 s_red[4][64];
    const int b = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int bin = blockIdx.x * 64 + b, ov = blockIdx.y, t = blockIdx.z;
    const bool live = bin < a.M;
    const int binc = live ? bin : a.M - 1;
    const bool b0 = binc == 0;
    const int o = ov / 3, v = ov - 3 * o;
    const int ir = a.irSel[t * 3 + v];
    float2 acc = make_float2(0.f, 0.f);
    const float2* H = a.Hf + ((long long)ir * a.nOut + o) * a.nFB * a.M + binc;
    const float2* Xb = a.Xr + binc;
    for (int p = q; p < a.nFB; p += 4) {
        int slot = (a.ringHead + t - p) % a.ringLen; if (slot < 0) slot += a.ringLen;
        const float2 h = H[(long long)p * a.M], x = Xb[(long long)slot * a.M];
        pc_mac(acc, h, b0 ? 0.0f : h.y, b0 ? h.y : h.x, x, b0);
    }
    s_red[q][b] = acc;
    __syncthreads();
    if (q == 0 && live) {
        float2 r = s_red[0][b];
#pragma unroll
        for (int qq = 1; qq < 4; qq++) { r.x += s_red[qq][b].x; r.y += s_red[qq][b].y; }
        a.P[((long long)t * a.nOut * 3 + ov) * a.M + bin] = r;
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

/* threads and dynamic LDS of the FFT kernels for a half-size M */
static int fft_threads(int M) { return M / 8 < 64 ? 64 : M / 8; }
static size_t fft_lds(int M) { return sizeof(float2) * (size_t)(M + M / 8 + M / 2); }
static int inv_threads(int M, int kSplit) { return (kSplit > 1 && M / 8 < 256) ? 256 : fft_threads(M); }
static size_t inv_lds(int M, int kSplit) { const int G = inv_threads(M, kSplit) / (M / 8); return fft_lds(M) + (G > 1 ? sizeof(float2) * (size_t)(G - 1) * M : 0); }
/* the common transform sizes (hop 64 .. 4096) get kernels specialised on log2 M */
#define PC_DISPATCH(LOGM_RT, CALL)                                                  \
    switch (LOGM_RT) {                                                              \
        case 6: { constexpr int L = 6; CALL; } break;                               \
        case 7: { constexpr int L = 7; CALL; } break;                               \
        case 8: { constexpr int L = 8; CALL; } break;                               \
        case 9: { constexpr int L = 9; CALL; } break;                               \
        case 10: { constexpr int L = 10; CALL; } break;                             \
        case 11: { constexpr int L = 11; CALL; } break;                             \
        case 12: { constexpr int L = 12; CALL; } break;                             \
        default: { constexpr int L = 0; CALL; } break;                              \
    }

static void fft_check(int N)
{
    const int M = N / 2, logM = ilog2(M);
    if ((1 << logM) != M || M < 8 || M > 8192) SAF_FATAL("matrixConv: FFT size %d unsupported (16 .. 16384; use the partitioned mode for long filters)", N);
    /* only M = 8192 (the runtime-size instantiation) needs more than the default 64 KiB of dynamic LDS */
    static bool raised = false;
    if (M > 4096 && !raised) {
        HIP_CHECK(hipFuncSetAttribute((const void*)pconv_rfft_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds(8192)));
        HIP_CHECK(hipFuncSetAttribute((const void*)pconv_irfft_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds(8192)));
        raised = true;
    }
}

void pconv_launch_fwd(const PconvFwd& f)
{
    FwdArgs a;
    a.src = f.src; a.s0 = f.s0; a.s1 = f.s1; a.s2 = f.s2; a.nValid = f.nValid; a.yValidStep = f.yValidStep; a.yValidTotal = f.yValidTotal;
    a.dst = f.dst; a.d0 = f.d0; a.d1 = f.d1; a.d2 = f.d2; a.ringLen = f.ringLen; a.ringHead = f.ringHead;
    a.tw = f.tw; a.M = f.N / 2; a.logM = ilog2(a.M);
    fft_check(f.N);
    const int tpf = fft_threads(a.M);
    int fpw = 1;
    while (fpw < 4 && 2 * fpw * tpf <= 256 && 2 * fpw <= f.g0) fpw *= 2;
    a.g0 = f.g0; a.fpw = fpw;
    const size_t lds = sizeof(float2) * (size_t)(a.M / 2 + fpw * (a.M + a.M / 8));
    KernelTimer kt("pconv_fft");
    PC_DISPATCH(a.logM, hipLaunchKernelGGL(pconv_rfft_fwd_kernel<L>, dim3((f.g0 + fpw - 1) / fpw, f.g1, f.g2), dim3(tpf * fpw), lds, stream(), a));
    HIP_CHECK(hipGetLastError());
}

void pconv_launch_apply(const PconvApply& p)
{
    const int M = p.N / 2, logM = ilog2(M);
    fft_check(p.N);
    /* split of the (partition, input) sum for THIS call: enough workgroups to fill the chip, no more — the blocks of
     * the call already provide parallelism, and every split is one more partial sum to fold */
    const int nTerms = p.nFB * p.nIn;
    const int OB = (!p.diag && p.nOut >= 2) ? 2 : 1;
    const int binTiles = (M + MAC_BINS - 1) / MAC_BINS, oGroups = (p.nOut + OB - 1) / OB, tGroups = (p.T + MAC_TB - 1) / MAC_TB;
    const long long tiles = (long long)binTiles * oGroups * tGroups;
    int kSplit = (int)((1024 + tiles - 1) / tiles);
    if (kSplit > p.kSplit) kSplit = p.kSplit;
    if (kSplit > (nTerms + MAC_TG - 1) / MAC_TG) kSplit = (nTerms + MAC_TG - 1) / MAC_TG;
    if (kSplit < 1) kSplit = 1;
    const int termsPerSplit = (nTerms + kSplit - 1) / kSplit;
    kSplit = (nTerms + termsPerSplit - 1) / termsPerSplit;
    {
        MacArgs a;
        a.Hf = p.Hf; a.Xr = p.Xr; a.P = p.P; a.nIn = p.nIn; a.nFB = p.nFB; a.nOut = p.nOut; a.M = M;
        a.kSplit = kSplit; a.termsPerSplit = termsPerSplit; a.tGroups = tGroups; a.ringLen = p.xRing; a.ringHead = p.xHead; a.T = p.T;
        a.diag = p.diag ? 1 : 0; a.nX = p.diag ? p.nOut : p.nIn;
        KernelTimer kt("pconv_mac");
        const dim3 grid(binTiles, oGroups, kSplit * tGroups);
        if (OB == 2) hipLaunchKernelGGL(pconv_mac_kernel<2>, grid, dim3(256), 0, stream(), a);
        else hipLaunchKernelGGL(pconv_mac_kernel<1>, grid, dim3(256), 0, stream(), a);
        HIP_CHECK(hipGetLastError());
    }
    {
        InvArgs a;
        a.P = p.P; a.zs = p.zs; a.tw = p.tw; a.nOut = p.nOut; a.kSplit = kSplit; a.M = M; a.logM = logM;
        a.zRing = p.zRing; a.zHead = p.zHead;
        KernelTimer kt("pconv_ifft");
        PC_DISPATCH(logM, hipLaunchKernelGGL(pconv_irfft_kernel<L>, dim3(p.nOut, p.T), dim3(inv_threads(M, kSplit)), inv_lds(M, kSplit), stream(), a));
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
    fft_check(p.N);
    {
        TvMacArgs a;
        a.Hf = p.Hf; a.Xr = p.Xr; a.P = p.P; a.irSel = p.irSel; a.nFB = p.nFB; a.nOut = p.nOut; a.M = M;
        a.ringLen = p.xRing; a.ringHead = p.xHead;
        KernelTimer kt("tvconv_mac");
        hipLaunchKernelGGL(tvconv_mac_kernel, dim3((M + 63) / 64, p.nOut * 3, p.T), dim3(256), 0, stream(), a);
        HIP_CHECK(hipGetLastError());
    }
    {
        InvArgs a;
        a.P = p.P; a.zs = p.zs; a.tw = p.tw; a.nOut = p.nOut * 3; a.kSplit = 1; a.M = M; a.logM = logM;
        a.zRing = p.zRing; a.zHead = p.zHead;
        KernelTimer kt("pconv_ifft");
        PC_DISPATCH(logM, hipLaunchKernelGGL(pconv_irfft_kernel<L>, dim3(p.nOut * 3, p.T), dim3(fft_threads(M)), fft_lds(M), stream(), a));
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
