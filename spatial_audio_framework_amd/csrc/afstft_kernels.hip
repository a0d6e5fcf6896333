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
 * Both kernels work through the hops of a launch in sub-chunks of 16 and keep the sub-chunk's spectra in an LDS ring
 * of 1 KiB slots (one slot = one hop of one channel):
 *   analysis  : workgroup = (instance, channel[, hop chunk]), 2 waves.  Per sub-chunk: window fold (thread = sample
 *               position, sliding 10-hop register window, every input sample read once from HBM, next sub-chunk's
 *               loads in flight under the FFT)  ->  16 FFTs  ->  real-FFT split (bins k and 128-k share their inputs),
 *               hybrid split and time-contiguous store of [band][ch][hop] (16 lanes = one 128-byte row segment).
 *   synthesis : workgroup = (instance, channel), 4 waves, wave-specialised: two producer waves gather the
 *               time-contiguous band rows of sub-chunk i+1, merge the hybrid bands, pack and inverse-FFT them into one
 *               LDS buffer while two consumer waves run the 10-segment overlap-add of sub-chunk i (thread = sample
 *               position, frame history in registers) out of the other buffer.
 * The 256-point real FFT is a 128-point complex FFT done by 8 lanes x 16 points: a radix-4x4 DFT-16 in registers, the
 * W128 twiddles, an 8x16 transpose through the FFT's own LDS slot (XOR-swizzled, conflict-free), two DFT-8 in
 * registers.  No cross-lane shuffles, ~45 wave-instructions per FFT.
 *
 * Lessons that shaped the code (measured on MI355X, profiles/): (1) a global load under a branch gets its own wait:
 * every load here is unconditional (clamped address, masked value); (2) __syncthreads() also drains vmcnt, so the
 * barriers order LDS only; (3) vmcnt retires in order: prefetched loads are consumed BEFORE the phase that issues the
 * spectrum / sample stores, otherwise their wait also waits for those stores to reach HBM.
 */
#include "saf_hip_common.h"
#include "afstft_device.h"
#include <mutex>

namespace saf {

/* ========================================================================== */
/*                                 analysis                                   */
/* ========================================================================== */

#ifdef ANA_STAMPS
static unsigned long long* g_ana_stamps = nullptr;
#endif
struct AnaArgs {
    AnaLaunch a;
    const float* win;      /* [1280] */
    const float2* twJ;     /* [8][16]  exp(-2 pi i j p / 128) */
    const float2* tw256;   /* [129]    exp(-2 pi i k / 256) */
    int chunk;
    unsigned long long* stamps;
};
#ifdef ANA_STAMPS        /* diagnostic build only: cycles per phase of every 64th workgroup (tools/ana_stamps.py) */
#define ASTAMP(i) do { if (stampOn && (tid & 63) == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stampAcc[i] += t_ - stampT; stampT = t_; } } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif

/* NCH = channels per workgroup (128 threads each).  NCH = 1 gives twice as many, half as large workgroups: more
 * independent fold / FFT / store pipelines per CU to overlap with each other. */
#ifndef ANA_BATCH
#define ANA_BATCH 4        /* regular items whose LDS reads are in flight together (split phase) */
#endif
template <int NCH>
__global__ __launch_bounds__(128 * NCH, 3) void afstft_analysis_kernel(AnaArgs g)
{
    __shared__ __attribute__((aligned(16))) float s_ring[NCH * ARING * SLOT];
    __shared__ float2 s_tw256[130];
    __shared__ float2 s_twJ[8 * 16];

    const int tid = threadIdx.x;
    const int inst = blockIdx.z;
    const int chBase = blockIdx.y * NCH;
#ifdef ANA_STAMPS
    const bool stampOn = g.stamps != nullptr && ((blockIdx.z * gridDim.y + blockIdx.y) & 63) == 0;
    unsigned long long stampAcc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stampT = __builtin_amdgcn_s_memtime();
#endif
    const int c0 = blockIdx.x * g.chunk;
    const int c1 = min(c0 + g.chunk, g.a.H);
    if (c0 >= c1) return;
    const int tabStride = g.a.tab_stride ? g.a.tab_stride : g.a.nCh;
    const int T = g.a.hopsPerFrame;

    for (int k = tid; k < 129; k += 128 * NCH) s_tw256[k] = g.tw256[k];
    if (tid < 128) load_twiddles_pj(s_twJ, g.twJ, tid);

    /* ---- fold role: thread = (channel of the pair, sample position) ---- */
    const int fc = tid >> 7, fn = tid & 127;
    const int fch = chBase + fc;
    const bool fOn = fch < g.a.nCh;
    const int fchc = fOn ? fch : g.a.nCh - 1;          /* the odd pair's missing channel shadows the last one (never stored) */
    const int srcch = g.a.ch_map ? g.a.ch_map[inst * tabStride + fchc] : fchc;
    const bool chValid = srcch >= 0 && srcch < g.a.nChIn;
    const float scale = chValid ? (g.a.ch_scale ? g.a.ch_scale[inst * tabStride + fchc] : 1.0f) : 0.0f;
    /* every load is  uniform 64-bit base  +  32-bit per-thread offset, unconditional (a load under a branch gets its own
     * wait and would serialise a sub-chunk's 16 loads into 16 HBM round trips) */
    const unsigned offIn = (unsigned)((chValid ? srcch : 0) * g.a.in_ch + fn);
    const unsigned offHist = (unsigned)(fchc * (SAF_ANA_HIST * SAF_HOP) + fn);
    const float* inBase = g.a.in + (long long)inst * g.a.in_inst;
    const float* histBase = g.a.hist_rd + (long long)inst * g.a.nCh * (SAF_ANA_HIST * SAF_HOP);
    /* uniform 64-bit base + 32-bit BYTE offset per lane (launch_analysis checks the extents): with a float index the
     * compiler must assume that 4 * index overflows 32 bits and does a 64-bit vector add per load */
    const unsigned offInB = offIn * 4u;
    auto ld_in = [&](const float* base) {
        /* `base` is uniform: pin it to scalar registers so that the access is  s[base] + v(32-bit offset) */
        const unsigned long long b = (unsigned long long)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo) + offInB);
    };
    const float invT = 1.0f / (float)T;
    const int inFrame = (int)g.a.in_frame;
    const char* inBaseB;                                    /* uniform: pinned to scalar registers */
    {
        const unsigned long long b = (unsigned long long)inBase;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        inBaseB = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    }
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; k++) w[k] = g.win[k * SAF_HOP + fn];

    /* ---- FFT role: thread = (FFT of the sub-chunk, lane j of its group of 8) ---- */
    const int ff = tid >> 3, fj = tid & 7;
    const int fftC = ff >> 4, fftT = ff & 15;
    const TwCol twJ{ s_twJ + fj };

    /* input cursor: element offset of the next non-negative hop inside this instance's input (uniform) */
    int curHop = c0 - SAF_ANA_HIST < 0 ? 0 : c0 - SAF_ANA_HIST;
    int curFrame = curHop / T, curSub = curHop - curFrame * T;
    long long curOff = (long long)curFrame * g.a.in_frame + curSub * SAF_HOP;

    /* window: xin[i] = x[hop h0 - 9 + i].  All loads a workgroup needs before its first FFT — the 9 hops before the
     * warm-up, the 6 warm-up hops and the first sub-chunk — are issued back to back: one memory round trip. */
    float xin[SUB + 9], xw[6];
#pragma unroll
    for (int i = 0; i < SAF_ANA_HIST; i++) {
        const int h = c0 - SAF_ANA_HIST + i;                       /* uniform */
        const float* base = h < 0 ? histBase + (SAF_ANA_HIST + h) * SAF_HOP : inBase + curOff;
        const unsigned off = h < 0 ? offHist : offIn;
        const float v = base[off] * (h < 0 ? 1.0f : scale);
        if (i < 6) xw[i] = v; else xin[i - 6] = v;
        if (h >= 0) { curSub++; curOff += SAF_HOP; if (curSub == T) { curSub = 0; curOff += g.a.in_frame - (long long)T * SAF_HOP; } }
    }
#pragma unroll
    for (int i = 0; i < SUB; i++) {
        xin[9 + i] = ld_in(inBase + curOff) * scale;
        if (c0 + i + 1 < c1) { curSub++; curOff += SAF_HOP; if (curSub == T) { curSub = 0; curOff += g.a.in_frame - (long long)T * SAF_HOP; } }
    }
    /* hop h of this chunk lives in ring position (h - (c0 - 6)) % ARING */
    /* ---- prologue: spectra of the 6 warm-up hops c0-6 .. c0-1 (hybrid FIR history; nothing is stored).
     *      Warm-up hop t folds x[c0-15+t .. c0-6+t] = xw[t..5], xin[0..t+3] ---- */
#pragma unroll
    for (int t = 0; t < 6; t++) {
        float fe = 0.0f, fo = 0.0f;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int q = t + k;                                   /* index into the 15 hops c0-15 .. c0-1 */
            const float xv = q < 6 ? xw[q < 6 ? q : 0] : xin[q >= 6 ? q - 6 : 0];
            if (k & 1) fo = fmaf(xv, w[k], fo); else fe = fmaf(xv, w[k], fe);
        }
        float* slot = s_ring + (fc * ARING + t) * SLOT;
        const int fa = 2 * ((fn >> 1) ^ SLOT_SG(t)) + (fn & 1);
        slot[fa] = fe; slot[128 + fa] = fo;
    }
    lds_barrier();
    if (fftT < 6) fft128_slot<false>(s_ring + (fftC * ARING + fftT) * SLOT, fj, twJ, SLOT_SG(fftT));
    /* (the barrier after the first fold below orders these spectra before their first use) */

    float2* outBase = g.a.out + (long long)inst * g.a.out_inst + (long long)chBase * g.a.out_ch;
    const unsigned ob32 = (unsigned)g.a.out_band;
    const int st = tid & 15, sr = tid >> 4;                /* store role: hop of the sub-chunk, item lane */

    int p0 = 6;                                            /* ring position of the sub-chunk's first hop */
    ASTAMP(7);
    for (int s0 = c0; s0 < c1; s0 += SUB) {
        const int n = min(SUB, c1 - s0);                   /* hops of this sub-chunk */
        if (s0 > c0) {                                     /* the hops requested in the previous iteration */
#pragma unroll
            for (int i = 0; i < SUB; i++) xin[9 + i] *= scale;
        }
        /* 1. window + fold (afSTFT_internal.c:276-301): f[(k&1)*128 + n] = sum_k x[hop-9+k][n] * w[k*128+n] */
#pragma unroll
        /* (hops beyond the end of a partial last sub-chunk fold their clamped loads into slots nobody reads: sixteen folds of
         * straight-line code) */
        for (int t = 0; t < SUB; t++) {
            float fe = 0.0f, fo = 0.0f;
#pragma unroll
            for (int i = 0; i < 5; i++) { fe = fmaf(xin[t + 2 * i], w[2 * i], fe); fo = fmaf(xin[t + 2 * i + 1], w[2 * i + 1], fo); }
            int pos = p0 + t; if (pos >= ARING) pos -= ARING;
            float* slot = s_ring + (fc * ARING + pos) * SLOT;
            const int fa = 2 * ((fn >> 1) ^ SLOT_SG(pos)) + (fn & 1);
            slot[fa] = fe; slot[128 + fa] = fo;
        }
        /* the workgroup that owns the end of the launch records the new input history (the last 15 hops) from its window */
        if (n == SUB && s0 + SUB == g.a.H && g.a.hist_wr && fOn) {
            float* dst = g.a.hist_wr + ((long long)inst * g.a.nCh + fch) * SAF_ANA_HIST * SAF_HOP + fn;
#pragma unroll
            for (int row = 0; row < SAF_ANA_HIST; row++) dst[row * SAF_HOP] = xin[SUB + 9 - SAF_ANA_HIST + row];
        }
        /* slide the window */
#pragma unroll
        for (int i = 0; i < 9; i++) xin[i] = xin[i + SUB];
        const bool more = s0 + SUB < c1;
#ifdef ANA_PF_EARLY
        float xl[SUB];
        if (more) {
#pragma unroll
            for (int i = 0; i < SUB; i++) {
                xl[i] = ld_in(inBase + curOff);
                if (s0 + SUB + i + 1 < c1) { curSub++; curOff += SAF_HOP; if (curSub == T) { curSub = 0; curOff += g.a.in_frame - (long long)T * SAF_HOP; } }
            }
        }
#endif
        ASTAMP(0);
        lds_barrier();
        ASTAMP(1);
        /* 2. 256-point real FFT as a 128-point complex FFT of z[m] = f[2m] + i f[2m+1], in place in the slot */
        if (fftT < n) {
            int pos = p0 + fftT; if (pos >= ARING) pos -= ARING;
            fft128_slot<false>(s_ring + (fftC * ARING + pos) * SLOT, fj, twJ, SLOT_SG(pos));
        }
        ASTAMP(2);
        lds_barrier();
        ASTAMP(3);
        /* The next sub-chunk's input is requested HERE, into the dead upper part of the window, and first used (scaled) by the
         * next fold: in flight under the split phase.  The spectrum stores below are younger than these loads, so the fold's
         * wait for them (vmcnt retires in order) does not wait for the stores; and no prefetch register is live across the FFT. */
#ifdef ANA_PF_EARLY
        if (more) {
#pragma unroll
            for (int i = 0; i < SUB; i++) xin[9 + i] = xl[i];
        }
#else
        if (more) {
            /* hop h starts at float index (h / T) * in_frame + (h % T) * 128: the offsets of the 16 hops are computed by 16 lanes at
             * once (float reciprocal, exact for h < 2^22: launch_analysis checks) and handed out with v_readlane — a scalar cursor
             * with its wrap test per hop is ~20 scalar instructions and two branches per load, here on the critical path */
            const int hq = min(s0 + SUB + (tid & 15), c1 - 1);
            const int fq = (int)(((float)hq + 0.5f) * invT);
            const unsigned offN = (unsigned)(fq * inFrame + (hq - fq * T) * SAF_HOP) * 4u;
#pragma unroll
            for (int i = 0; i < SUB; i++) xin[9 + i] = *reinterpret_cast<const float*>(inBaseB + (__builtin_amdgcn_readlane(offN, i) + offInB));
        }
#endif
        ASTAMP(4);
        /* 3. real-FFT split (bins k and 128-k share their inputs), hybrid split + 3-hop delay
         *    (afSTFT_internal.c:523-623), stored time-contiguous: 16 lanes = 16 hops = one 128-byte row segment.
         *    Regular items (every thread, branch-free): (hop st, bins k and 128-k) for k = 0, 5..64 — k = 5 + sr + 8 ii, the last
         *    round takes 61..63, 0 and 64 (k = 64 is its own partner: both of its stores write the same value to the same place).
         *    Hybrid items (lanes 0..31 of each wave: hop st, bin b = 1 + 2 wv + (lane >> 4)): the two half-bands of bin b from the
         *    spectra of hops t, t-2, t-4, t-6 and the bin itself 3 hops back (afSTFT_internal.c:595-619), and band 132 - b.
         *    Straight-line code with the LDS reads of many items in flight at once: as a chain of read -> wait -> store per item
         *    this phase was 46 % of the kernel. */
        static_assert(NCH == 1, "the item schedule below assumes 8 item lanes per channel (128 threads)");
        {
            const int stc = st < n ? st : 0;                                   /* hops beyond the end shadow hop 0 (not stored) */
            int pos = p0 + stc; if (pos >= ARING) pos -= ARING;                /* ring position of S_hop */
            int pD = pos - 3; if (pD < 0) pD += ARING;                         /* all bands are delayed 3 hops */
            if (!g.a.hybrid) pD = pos;                                         /* plain STFT bins, no hybrid delay */
            const bool stOn = st < n;
            const float* slotD = s_ring + pD * SLOT;
            const int sgD = SLOT_SG(pD);
            /* spectra stores: uniform 64-bit base + 32-bit byte offset per lane (launch_analysis checks that an instance's
             * spectra span less than 4 GiB): no 64-bit address arithmetic per store */
            char* const ob = reinterpret_cast<char*>(outBase);
            const unsigned ohop8 = (unsigned)(s0 + stc) << 3, ob8 = ob32 << 3;
#ifdef ANA_NOSTORE   /* experiment: everything but the spectrum stores */
            auto put = [&](unsigned off8, float2 v, bool on) { if (on && v.x == 1.2345e-30f) *reinterpret_cast<float2*>(ob + off8) = v; };
#else
            auto put = [&](unsigned off8, float2 v, bool on) { if (on) *reinterpret_cast<float2*>(ob + off8) = v; };
#endif
            const int hyb = g.a.hybrid;
            /* all LDS reads of the eight regular items first (one wait), then the arithmetic and the stores */
#pragma unroll
            for (int bt = 0; bt < 8; bt += ANA_BATCH) {
            float2 rZk[ANA_BATCH], rZm[ANA_BATCH], rW[ANA_BATCH];
#pragma unroll
            for (int i4 = 0; i4 < ANA_BATCH; i4++) {
                const int ii = bt + i4;
                int k = 5 + sr + 8 * ii;
                if (ii == 7) k = sr < 3 ? 61 + sr : (sr == 3 ? 0 : (sr == 4 ? 64 : sr - 4));        /* sr 5..7: bins 1..3, stored by the plain STFT only */
                rZk[i4] = *reinterpret_cast<const float2*>(slotD + 2 * (k ^ sgD));
                rZm[i4] = *reinterpret_cast<const float2*>(slotD + 2 * (((128 - k) & 127) ^ sgD));
                rW[i4] = s_tw256[k];
            }
#pragma unroll
            for (int i4 = 0; i4 < ANA_BATCH; i4++) {
                const int ii = bt + i4;
                int k = 5 + sr + 8 * ii;
                bool on = stOn;
                if (ii == 7) { k = sr < 3 ? 61 + sr : (sr == 3 ? 0 : (sr == 4 ? 64 : sr - 4)); on = stOn && (sr < 5 || !hyb); }
                const float2 Zk = rZk[i4], Zm = rZm[i4];
                const float2 e = make_float2(Zk.x + Zm.x, Zk.y - Zm.y);
                const float2 d = make_float2(Zk.x - Zm.x, Zk.y + Zm.y);
                const float2 t = cmul(rW[i4], d);
                const float2 Xk = make_float2(0.5f * (e.x + t.y), 0.5f * (e.y - t.x));
                const float2 Xm = make_float2(0.5f * (e.x - t.y), 0.5f * (-e.y - t.x));
                const unsigned bk = hyb ? (k == 0 ? 0u : (unsigned)(k + 4)) : (unsigned)k;
                const unsigned bm = hyb ? (unsigned)(132 - k) : (unsigned)(128 - k);
                put(bk * ob8 + ohop8, Xk, on);
                put(bm * ob8 + ohop8, Xm, on);
            }
            }
            /* hybrid items: ten LDS reads, then the arithmetic */
            const int hbin = 1 + 2 * (tid >> 6) + ((tid >> 4) & 1);
            float2 hS0, hS2, hS4, hS6, hXk, hXm;
            if (hyb) {
                int p2 = pos - 2; if (p2 < 0) p2 += ARING;
                int p4 = pos - 4; if (p4 < 0) p4 += ARING;
                int p6 = pos - 6; if (p6 < 0) p6 += ARING;
                const float2 W = s_tw256[hbin];
                hS0 = ana_bin_lo(s_ring + pos * SLOT, SLOT_SG(pos), hbin, W);
                hS2 = ana_bin_lo(s_ring + p2 * SLOT, SLOT_SG(p2), hbin, W);
                hS4 = ana_bin_lo(s_ring + p4 * SLOT, SLOT_SG(p4), hbin, W);
                hS6 = ana_bin_lo(s_ring + p6 * SLOT, SLOT_SG(p6), hbin, W);
                ana_bin_pair(slotD, sgD, hbin, W, hXk, hXm);
            }
            if (!hyb) {          /* plain STFT: bin 4 is left (bins 1..3 went with the last round) */
                float2 Xk, Xm;
                ana_bin_pair(slotD, sgD, 4, s_tw256[4], Xk, Xm);
                put(4u * ob8 + ohop8, Xk, stOn && sr == 0);
                put(124u * ob8 + ohop8, Xm, stOn && sr == 0);
            } else {
                float gr, gi;
                gr = -COEFF1 * hS0.y;          gi = COEFF1 * hS0.x;
                gr -= COEFF2 * hS2.y;          gi += COEFF2 * hS2.x;
                gr += COEFF2 * hS4.y;          gi -= COEFF2 * hS4.x;
                gr += COEFF1 * hS6.y;          gi -= COEFF1 * hS6.x;
                const float dr = hXk.x * 0.5f, di = hXk.y * 0.5f;
                /* lower half-band (band 2b-1) of bins 1,3 subtracts, of bins 2,4 adds (afSTFT_internal.c:606-619) */
                const float sgn = (hbin & 1) ? -1.0f : 1.0f;
                const bool hOn = stOn && (tid & 32) == 0;
                put((unsigned)(2 * hbin - 1) * ob8 + ohop8, make_float2(dr + sgn * gr, di + sgn * gi), hOn);
                put((unsigned)(2 * hbin) * ob8 + ohop8, make_float2(dr - sgn * gr, di - sgn * gi), hOn);
                put((unsigned)(132 - hbin) * ob8 + ohop8, hXm, hOn);
            }
        }
        p0 += n; if (p0 >= ARING) p0 -= ARING;
        ASTAMP(5);
        lds_barrier();
        ASTAMP(6);
    }
#ifdef ANA_STAMPS
    if (stampOn && (tid & 63) == 0) for (int i = 0; i < 8; i++) atomicAdd(&g.stamps[(tid >> 6) * 8 + i], stampAcc[i]);
#endif

    /* the workgroup that owns the last chunk records the new input history: the last 15 input hops */
    if (c1 == g.a.H && (g.a.H % SUB) != 0 && g.a.hist_wr && fOn) {        /* partial last sub-chunk: reload */
        int hh = g.a.H - SAF_ANA_HIST < 0 ? 0 : g.a.H - SAF_ANA_HIST;
        int fr = hh / T, sb = hh - fr * T;
        long long off = (long long)fr * g.a.in_frame + sb * SAF_HOP;
        float* dst = g.a.hist_wr + ((long long)inst * g.a.nCh + fch) * SAF_ANA_HIST * SAF_HOP + fn;
#pragma unroll
        for (int row = 0; row < SAF_ANA_HIST; row++) {
            const int h = g.a.H - SAF_ANA_HIST + row;
            const float* base = h < 0 ? histBase + (SAF_ANA_HIST + h) * SAF_HOP : inBase + off;
            const unsigned o = h < 0 ? offHist : offIn;
            dst[row * SAF_HOP] = base[o] * (h < 0 ? 1.0f : scale);
            if (h >= 0) { sb++; off += SAF_HOP; if (sb == T) { sb = 0; off += g.a.in_frame - (long long)T * SAF_HOP; } }
        }
    }
}

/* ========================================================================== */
/*                                 synthesis                                  */
/* ========================================================================== */

struct SynArgs {
    SynLaunch s;
    const float* win;
    const float2* twJ;
    const float2* tw256;
    int chunk;             /* hops per workgroup along time (a multiple of SUB, or H) */
};

/* Wave-specialised synthesis: one output channel per workgroup of 4 waves.
 *   waves 0-1 (producer): gather + hybrid merge + half-complex packing of sub-chunk i+1 into one LDS buffer, then its 16
 *                         inverse FFTs in place;
 *   waves 2-3 (consumer): 10-segment overlap-add of sub-chunk i from the other buffer, one thread per sample position,
 *                         frame history in registers, output stores.
 * The two roles need different registers (gather staging + FFT temporaries vs. the overlap-add window), so the kernel's
 * register count is the larger of the two instead of their sum, and they overlap in time: while the producer's loads are
 * in flight and its FFT runs, the consumer streams the previous sub-chunk out.  Two workgroup barriers per sub-chunk.
 * grid (channel, instance, time chunk): a chunk that does not start the launch re-synthesises the 16 hops before it to
 * rebuild the 9-frame overlap-add history (nothing is emitted for them); only used when few (instance, channel)
 * workgroups exist. */
__global__ __launch_bounds__(256, 4) void afstft_synthesis_ws_kernel(SynArgs g)
{
    __shared__ __attribute__((aligned(16))) float s_buf[2][SUB * SLOT];
    __shared__ float2 s_tw256[130];
    __shared__ float2 s_twJ[8 * 16];

    const int tid = threadIdx.x;
    const int ch = blockIdx.x, inst = blockIdx.y;
    const int c0 = blockIdx.z * g.chunk;
    const int H = min(c0 + g.chunk, g.s.H);                      /* end of this workgroup's hops */
    if (c0 >= H) return;
    const int hs = c0 > 0 ? c0 - SUB : 0;                        /* first hop synthesised (warm-up sub-chunk before c0) */
    const int nSub = (H - hs + SUB - 1) / SUB;
    const bool producer = tid < 128;
    for (int k = tid; k < 129; k += 256) s_tw256[k] = g.tw256[k];
    if (tid < 128) load_twiddles_pj(s_twJ, g.twJ, tid);

    if (producer) {
        const int ff = tid >> 3, fj = tid & 7;                    /* FFT role: 16 FFTs x 8 lanes */
        const TwCol twJ{ s_twJ + fj };
        const float2* inBase = g.s.in + (long long)inst * g.s.in_inst + (long long)ch * g.s.in_ch;
        const unsigned ib32 = (unsigned)g.s.in_band;
        /* spectra loads: uniform 64-bit base + 32-bit byte offset per lane (launch_synthesis checks the extent) */
        const char* const ibc = reinterpret_cast<const char*>(inBase);
        auto get = [&](unsigned idx) { return *reinterpret_cast<const float2*>(ibc + (idx << 3)); };
        const int gt = tid & 15, gq = (tid >> 4) & 7;             /* gather role: hop of the sub-chunk, item lane (8); the mask tells the compiler the range, so the per-item case analysis folds after unrolling */
        for (int it = -1; it <= nSub; it++) {
            const int s0 = hs + it * SUB;
            if (it >= 0 && it < nSub) {
                /* gather bands -> bins (afHybridInverse, afSTFT_internal.c:625-653): time-contiguous reads, unconditional
                 * loads, all issued before the first use; item = bin pair (k, 128-k), k = gq + 8 i.  The load latency is
                 * covered by the consumer waves, which are busy with the previous sub-chunk. */
                const int nn = min(SUB, H - s0);
                const unsigned ohop = (unsigned)(s0 + (gt < nn ? gt : 0));
                float2 rXk[9], rXm[9], rX2 = make_float2(0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    int k = gq + 8 * i; if (k > 64) k = 0;
                    int bk, bm;
                    if (!g.s.hybrid) { bk = k; bm = 128 - k; }
                    else { bm = 132 - k; bk = k == 0 ? 0 : (k < 5 ? 2 * k - 1 : k + 4); }      /* bin k = band 2k-1 (+ band 2k), k = 1..4 */
                    rXk[i] = get((unsigned)bk * ib32 + ohop);
                    rXm[i] = get((unsigned)bm * ib32 + ohop);
                    if (i == 0) {                   /* the only pass that can hold bins 1..4 */
                        const bool pair = g.s.hybrid && k >= 1 && k < 5;
                        const float2 u = get((unsigned)(pair ? 2 * k : bk) * ib32 + ohop);
                        rX2 = pair ? u : make_float2(0.f, 0.f);
                    }
                }
                /* half-complex -> packed (kiss_fftr.c:125-161; Im of DC/Nyquist ignored): bins k and 128-k give
                 * 2 Z[k] = E + i O and 2 Z[128-k] = conj(E - i O), E = X[k] + conj X[128-k], O = (X[k] - conj X[128-k]) e^{+2 pi i k/256} */
                float* ring = s_buf[it & 1];
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const int kk = gq + 8 * i;
                    const int k = kk > 64 ? 0 : kk;
                    float2 Xk = rXk[i], Xm = rXm[i];
                    if (i == 0) { Xk.x += rX2.x; Xk.y += rX2.y; }
                    /* low-delay mode: odd bins change sign = circular half-frame shift (afSTFT_internal.c:366-369) */
                    if (g.s.lowDelay && (k & 1)) { Xk.x = -Xk.x; Xk.y = -Xk.y; Xm.x = -Xm.x; Xm.y = -Xm.y; }
                    if (k == 0) { Xk.y = 0.0f; Xm.y = 0.0f; }
                    const float2 E = make_float2(Xk.x + Xm.x, Xk.y - Xm.y);
                    const float2 D = make_float2(Xk.x - Xm.x, Xk.y + Xm.y);
                    const float2 W = s_tw256[k];
                    const float2 O = make_float2(D.x * W.x + D.y * W.y, D.y * W.x - D.x * W.y);      /* D * conj(W) */
                    if (gt < nn && kk <= 64) {
                        float* slot = ring + gt * SLOT;
                        const int sg = SLOT_SG(gt);
                        *reinterpret_cast<float2*>(slot + 2 * (k ^ sg)) = make_float2(E.x - O.y, E.y + O.x);
                        if (k != 0 && k != 64) *reinterpret_cast<float2*>(slot + 2 * ((128 - k) ^ sg)) = make_float2(E.x + O.y, O.x - E.y);
                    }
                }
            }
            lds_barrier();                                       /* (A) */
            /* 128-point inverse FFT in place: frame sample 2m, 2m+1 = Re, Im z[m] (x 1/256 in the overlap-add: 1/2 of the
             * packing above and the 1/128 of saf_rfft_backward's 1/N, saf_utility_fft.c:751) */
            if (it >= 0 && it < nSub && ff < min(SUB, H - s0)) fft128_slot<true>(s_buf[it & 1] + ff * SLOT, fj, twJ, SLOT_SG(ff));
            lds_barrier();                                       /* (B) */
        }
    } else {
        /* overlap-add role: thread = sample position n: gl[i] / gr[i] = samples n / 128+n of the frame of hop h0 - 9 + i */
        const int on = tid - 128;
        const int T = g.s.hopsPerFrame;
        float wn[10], gl[OLA + 9], gr[OLA + 9];
#pragma unroll
        for (int k = 0; k < 10; k++) wn[k] = g.win[k * SAF_HOP + on];
        {
            const float* h = g.s.hist_rd + ((long long)inst * g.s.nCh + ch) * SAF_SYN_HIST * 256;
#pragma unroll
            for (int i = 0; i < 9; i++) { const float a = h[i * 256 + on], b = h[i * 256 + 128 + on]; gl[i] = c0 > 0 ? 0.0f : a; gr[i] = c0 > 0 ? 0.0f : b; }
#pragma unroll
            for (int i = 9; i < OLA + 9; i++) gl[i] = gr[i] = 0.0f;
        }
        float* outBase = g.s.out + (long long)inst * g.s.out_inst + (long long)ch * g.s.out_ch + on;
        int oFrame = c0 / T, oSub = c0 - oFrame * T;             /* output cursor (uniform): hop -> (frame, hop within the frame) */
        const float sc = 1.0f / 256.0f;
        /* 10-segment overlap-add, oldest frame first (afSTFT_internal.c:396-444): the hop emitted at s0+t is
         * sum_k w[k*128+n] * frame_{t-k}[(k&1)*128 + n]; one pass of OLA = 8 hops per barrier interval */
        for (int it = -1; it <= nSub; it++) {
            const int sp = hs + (it - 1) * SUB;                  /* sub-chunk being emitted */
            const int np = it >= 1 ? min(SUB, H - sp) : 0;
            const bool emit = sp >= c0;                          /* the warm-up sub-chunk only rebuilds the frame history */
            const float* ring = s_buf[(it + 1) & 1];
#pragma unroll
            for (int half = 0; half < SUB / OLA; half++) {       /* one pass of OLA = 8 hops per barrier interval */
                const int nh = min(OLA, np - half * OLA);
                if (nh > 0) {
#pragma unroll
                    for (int u = 0; u < OLA; u++) {
                        if (u < nh) {
                            const float* slot = ring + (half * OLA + u) * SLOT;
                            const int oa = 2 * ((on >> 1) ^ SLOT_SG(half * OLA + u)) + (on & 1);
                            gl[9 + u] = slot[oa] * sc; gr[9 + u] = slot[128 + oa] * sc;
                            float acc = 0.0f;
#pragma unroll
                            for (int k = 9; k >= 0; k--) acc = fmaf(wn[k], (k & 1) ? gr[9 + u - k] : gl[9 + u - k], acc);
                            if (emit) {
                                outBase[(long long)oFrame * g.s.out_frame + oSub * SAF_HOP] = acc;
                                oSub++; if (oSub == T) { oSub = 0; oFrame++; }
                            }
                        }
                    }
                    if (nh == OLA) {
#pragma unroll
                        for (int i = 0; i < 9; i++) { gl[i] = gl[i + OLA]; gr[i] = gr[i + OLA]; }
                    } else {                                     /* partial pass: the 9 newest frames sit at nh .. nh+8 */
#pragma unroll
                        for (int i = 0; i < 9; i++) {
                            float a = gl[i], b = gr[i];
#pragma unroll
                            for (int q = 1; q < OLA; q++) if (q == nh) { a = gl[i + q]; b = gr[i + q]; }
                            gl[i] = a; gr[i] = b;
                        }
                    }
                }
                lds_barrier();                                   /* (A) after the first pass, (B) after the second */
            }
        }
        if (g.s.hist_wr && H == g.s.H) {
            float* h = g.s.hist_wr + ((long long)inst * g.s.nCh + ch) * SAF_SYN_HIST * 256;
#pragma unroll
            for (int i = 0; i < 9; i++) { h[i * 256 + on] = gl[i]; h[i * 256 + 128 + on] = gr[i]; }
        }
    }
}

/* ========================================================================== */
/*                        constant tables + launchers                         */
/* ========================================================================== */

static float* g_dev_win[2][2] = { { nullptr, nullptr }, { nullptr, nullptr } };
static float2* g_dev_tw = nullptr;
static std::mutex g_tab_mutex;          /* first calls may come from several host threads at once: a table is published only when it is filled */

const float* dev_window(int lowDelay, int synthesis)
{
    float*& slot = g_dev_win[lowDelay ? 1 : 0][synthesis ? 1 : 0];
    if (float* q = __atomic_load_n(&slot, __ATOMIC_ACQUIRE)) return q;
    std::lock_guard<std::mutex> lk(g_tab_mutex);
    if (slot) return slot;
    float* d = nullptr;
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
    __atomic_store_n(&slot, d, __ATOMIC_RELEASE);
    return d;
}

/* [0 .. 127]: twJ[j][p] = exp(-2 pi i j p / 128), j < 8, p < 16;  [128 .. 256]: exp(-2 pi i k / 256), k <= 128 */
const float2* dev_twiddles()
{
    if (float2* q = __atomic_load_n(&g_dev_tw, __ATOMIC_ACQUIRE)) return q;
    std::lock_guard<std::mutex> lk(g_tab_mutex);
    if (g_dev_tw) return g_dev_tw;
    std::vector<float2> t(128 + 129);
    for (int j = 0; j < 8; j++)
        for (int p = 0; p < 16; p++) {
            const double a = -2.0 * SAF_PId * (double)(j * p) / 128.0;
            t[j * 16 + p] = make_float2((float)cos(a), (float)sin(a));
        }
    for (int k = 0; k <= 128; k++) {
        const double a = -2.0 * SAF_PId * (double)k / 256.0;
        t[128 + k] = make_float2((float)cos(a), (float)sin(a));
    }
    t[128 + 128] = make_float2(-1.0f, 0.0f);           /* exact: bin 128 must come out purely real */
    float2* d = nullptr;
    HIP_CHECK(hipMalloc((void**)&d, t.size() * sizeof(float2)));
    HIP_CHECK(hipMemcpy(d, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice));
    __atomic_store_n(&g_dev_tw, d, __ATOMIC_RELEASE);
    return d;
}

void launch_analysis(const AnaLaunch& a)
{
    if (a.H <= 0 || a.nCh <= 0 || a.nInst <= 0) return;
    if (a.hop != SAF_HOP) { launch_analysis_generic(a); return; }
    AnaArgs g;
    g.a = a;
    g.win = dev_window(a.lowDelay, 0);
    g.twJ = dev_twiddles();
    g.tw256 = g.twJ + 128;
    /* Time chunks add parallelism when few (instance, channel pair) workgroups are in flight, but every extra chunk
     * recomputes 6 warm-up FFTs.  Chunks are multiples of 16 hops (aligned 128-byte row segments).  Pick the chunk
     * count that minimises (rounds over the chip) x (sub-chunks per workgroup); 3 workgroups per CU on 256 CUs. */
    const int NCHW = 1;                                   /* channels per workgroup */
    const long long groups = (long long)((a.nCh + NCHW - 1) / NCHW) * a.nInst;
    const long long slots = (6 / NCHW) * 256;
    int chunk = a.H;
    {
        long long best = -1;
        const int nSub = (a.H + SUB - 1) / SUB;
        for (int per = nSub; per >= 1; per--) {                 /* sub-chunks per workgroup */
            const int nc = (nSub + per - 1) / per;
            const long long rounds = (groups * nc + slots - 1) / slots;
            const long long cost = rounds * (3 * per + 1);      /* prologue ~ a third of a sub-chunk */
            if (best < 0 || cost < best) { best = cost; chunk = per * SUB; }
        }
    }
    g.chunk = chunk;
    g.stamps = nullptr;
#ifdef ANA_STAMPS
    { static unsigned long long* buf = nullptr; if (!buf) { HIP_CHECK(hipMalloc((void**)&buf, 16 * 8)); HIP_CHECK(hipMemset(buf, 0, 16 * 8)); } g.stamps = buf; g_ana_stamps = buf; }
#endif
    /* the kernel addresses one instance's samples and spectra with 32-bit byte offsets from uniform bases */
    {
        const long long chSpan = (long long)(a.nChIn > 0 ? a.nChIn : 1) * a.in_ch;
        const long long hopSpan = (long long)((a.H + a.hopsPerFrame - 1) / a.hopsPerFrame) * a.in_frame + (long long)a.hopsPerFrame * SAF_HOP;
        if (a.in_ch < 0 || a.in_frame < 0 || a.H >= (1 << 22) || (chSpan + hopSpan) * 4 >= (1ll << 32))
            SAF_FATAL("afSTFT analysis: one call spans more than 4 GiB of one instance's input, 2^22 hops or uses negative strides: split the call");
    }
    if ((unsigned long long)(a.nChIn > 0 ? a.nChIn : 1) * (unsigned long long)(a.in_ch < 0 ? -a.in_ch : a.in_ch) * 4ull >= (1ull << 32) ||
        ((unsigned long long)SAF_NBANDS * (unsigned long long)a.out_band + (unsigned long long)a.nCh * (unsigned long long)a.out_ch + (unsigned long long)a.H) * 8ull >= (1ull << 32))
        SAF_FATAL("afSTFT analysis: one instance's channel block or spectra exceed 4 GiB (split the call)");
    dim3 grid((a.H + g.chunk - 1) / g.chunk, (a.nCh + NCHW - 1) / NCHW, a.nInst);
    KernelTimer kt("afstft_analysis");
    hipLaunchKernelGGL(afstft_analysis_kernel<NCHW>, grid, dim3(128 * NCHW), 0, stream(), g);
    HIP_CHECK(hipGetLastError());
}

void launch_synthesis(const SynLaunch& s)
{
    if (s.H <= 0 || s.nCh <= 0 || s.nInst <= 0) return;
    if (s.hop != SAF_HOP) { launch_synthesis_generic(s); return; }
    SynArgs g;
    g.s = s;
    g.win = dev_window(s.lowDelay, 1);
    g.twJ = dev_twiddles();
    g.tw256 = g.twJ + 128;
    /* time chunks only when the (instance, channel) grid leaves most of the chip idle: every extra chunk re-synthesises
     * 16 hops.  Aim at >= 512 workgroups with chunks of >= 64 hops (multiples of 16). */
    g.chunk = s.H;
    const long long wgs = (long long)s.nCh * s.nInst;
    if (wgs < 512 && s.H >= 128) {
        int nChunks = (int)((512 + wgs - 1) / wgs);
        if (nChunks > s.H / 64) nChunks = s.H / 64;
        if (nChunks > 1) g.chunk = ((s.H + nChunks - 1) / nChunks + SUB - 1) / SUB * SUB;
    }
    if (((unsigned long long)SAF_NBANDS * (unsigned long long)s.in_band + (unsigned long long)s.H) * 8ull >= (1ull << 32))
        SAF_FATAL("afSTFT synthesis: one channel's spectra span more than 4 GiB (split the call)");
    dim3 grid(s.nCh, s.nInst, (s.H + g.chunk - 1) / g.chunk);
    KernelTimer kt("afstft_synthesis");
    hipLaunchKernelGGL(afstft_synthesis_ws_kernel, grid, dim3(256), 0, stream(), g);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf

#ifdef ANA_STAMPS
extern "C" __attribute__((visibility("default"))) void saf_hip_debug_ana_stamps(unsigned long long* out16)
{
    HIP_CHECK(hipStreamSynchronize(saf::stream()));
    if (saf::g_ana_stamps) { HIP_CHECK(hipMemcpy(out16, saf::g_ana_stamps, 16 * 8, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemset(saf::g_ana_stamps, 0, 16 * 8)); }
}
#endif
