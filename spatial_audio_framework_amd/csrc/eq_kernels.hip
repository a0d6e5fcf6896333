/*
 * eq_kernels.hip — afSTFT analysis -> real gain per (channel, band) -> afSTFT synthesis of every channel in ONE kernel:
 * the spectra never leave the compute unit.
 *
 * Why it exists.  ambi_dec_process (examples/src/ambi_dec/ambi_dec.c:514-566) runs afSTFT_forward, one real matrix per
 * band (decoder d(band), order n(band), max-rE weights, M_norm: ambi_dec.c:518-540) and afSTFT_backward.  Every one of
 * those matrices is the SAME dense decoder M_d (ambi_dec.c:283-288 truncates the order-N matrix to the first (n+1)^2
 * columns) times a DIAGONAL of per-channel weights:  A_{d,n} = M_d diag(w_{d,n}),  w_{d,n}[k] = M_norm_{d,n} a_n[k]
 * (k < (n+1)^2, else 0).  The diagonal part commutes into the filterbank of each input channel, the dense part is
 * band-independent and commutes out of it:
 *
 *     out = sum_d  M_d  z_d,      z_d[ch] = synthesis( w_{d(band), n(band)}[ch] (.) analysis(x[ch]) )
 *
 * z_d is what this kernel computes (per SH channel: window fold, 256-point FFT, hybrid split, the gains, hybrid merge,
 * inverse FFT, 10-segment overlap-add: afSTFT_internal.c:237-653 restated per channel); the dense product is one
 * time-domain MFMA GEMM (gemm_kernels.hip).  HBM traffic per 64-channel block: samples in, z out, z in, samples out
 * (4 x 131 072 B for one dense matrix) instead of the 1 351 680 B of the three-kernel transform path.
 *
 * Channels whose gains are the same in every band need no transform at all: FFT and inverse FFT cancel, the hybrid
 * split + merge is its 3-hop delay, and the frame of output hop t is gain x (window fold of hop t-3): `uniform`.
 *
 * Work decomposition: workgroup = (channel, instance), 128 threads.  Hops are processed in sub-chunks of 16; the
 * sub-chunk's folds / spectra / frames live in a ring of 1 KiB LDS slots (one slot = one hop), transformed in place:
 *   1 fold     thread = sample position, sliding 10-hop register window, every input sample read once
 *   2 FFT      8 lanes x 16 points per hop (fft128_slot); bins 1..4 of the new hop -> s_low (hybrid FIR history)
 *   3 bins     lane = bin pair (k, 128-k) of one slot: real-FFT split, gains, half-complex packing — consecutive
 *              8-byte LDS accesses, conflict-free, in place.  The hybrid bins 1..4 (afSTFT_internal.c:595-619 + the gains
 *              of their two half-bands + merge) and DC / Nyquist are separate items of the same phase: every item reads
 *              and writes only its own two elements of the slot
 *   4 IFFT     in place
 *   5 OLA      thread = sample position, frame history in registers, output stores
 * Phases 3-5 run three hops behind phases 1-2: the hybrid filter of output hop t needs bins 1..4 of hops t, t-2, t-4,
 * t-6 and everything else of hop t-3.  Three workgroup barriers per sub-chunk: fold -> FFT (every FFT reads all sample
 * positions), FFT -> bins (the hybrid filter reads bins of hops transformed by the other wave), IFFT -> OLA.  A wave runs
 * phases 3 and 4 on the same 8 slots (LDS operations of a wave execute in order: no barrier), and phase 5 -> phase 1 of
 * the next sub-chunk needs none either: in both a thread touches only the two elements of its own sample position.
 */
#include "saf_hip_common.h"
#include "afstft_device.h"

namespace saf {

#define ERING 20        /* slots in the ring: 16 new hops + 3 lagged + 1 (a multiple of 4: the four FFT groups of a lane
                         * group stay 16 banks apart across the wrap) */
#define LOWR  32        /* hops of bins 1..4 kept for the hybrid FIR (power of two >= 16 + 7) */
#ifndef EQ_OLA
#define EQ_OLA 4
#endif
#ifndef EQ_MINWAVES
#define EQ_MINWAVES 3       /* waves per SIMD the one-output kernel is compiled for (168 registers) */
#endif

struct EqArgs { EqLaunch e; const float* win; const float2* twJ; const float2* tw256; int chunk; };

template <int D>
__global__ __launch_bounds__(128, D == 1 ? EQ_MINWAVES : 2) void afstft_eq_kernel(EqArgs g)
{
    __shared__ __attribute__((aligned(16))) float s_ring[ERING * SLOT];
    __shared__ __attribute__((aligned(16))) float s_out1[D > 1 ? SUB * SLOT : 4];      /* frames of the second output */
    __shared__ float2 s_low[LOWR][4];
    __shared__ float s_gain[D][136];
    __shared__ float2 s_twJ[8 * 16];
    __shared__ float2 s_twl[8];                              /* e^{-2 pi i k / 256}, k < 8 (bins 1..4) */

    const EqLaunch& e = g.e;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ch = blockIdx.x, inst = blockIdx.y;
    const int T = e.hopsPerFrame;
    /* time chunks (grid z) add parallelism when few (channel, instance) workgroups exist: a chunk that does not start the launch
     * first runs the 16 hops before it without emitting them, which rebuilds its overlap-add history (identical arithmetic:
     * the outputs do not depend on how a launch is cut) */
    const int c0 = blockIdx.z * g.chunk;                     /* first hop this workgroup emits */
    const int H = min(c0 + g.chunk, e.H);                    /* end of its hops */
    if (c0 >= H) return;
    const int hs = c0 > 0 ? c0 - SUB : 0;                    /* first hop it processes (launch_eq keeps chunks >= 32 hops: hs >= 15) */
    const bool last = H == e.H;                              /* the workgroup that owns the end of the launch records the state */
    const bool uni = e.uniform != nullptr && e.uniform[inst * SAF_MAXCH + ch] != 0;

    load_twiddles_pj(s_twJ, g.twJ, tid);
    if (tid < 8) s_twl[tid] = g.tw256[tid];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float* gsrc = e.gains + (((long long)inst * D + d) * SAF_MAXCH + ch) * 136;
        for (int b = tid; b < 136; b += 128) s_gain[d][b] = b < SAF_NBANDS ? gsrc[b] : 0.0f;
    }

    /* ---- fold role: thread = sample position ---- */
    const int fn = tid;
    const int srcch = e.ch_map ? e.ch_map[inst * SAF_MAXCH + ch] : ch;
    const bool chValid = srcch >= 0 && srcch < e.nChIn;
    const float scale = chValid ? (e.ch_scale ? e.ch_scale[inst * SAF_MAXCH + ch] : 1.0f) : 0.0f;
    const unsigned offInB = (unsigned)((chValid ? srcch : 0) * e.in_ch + fn) * 4u;      /* launch_eq checks the extent */
    const float* inBase = e.in + (long long)inst * e.in_inst;
    const float* hist = e.hist_rd + ((long long)inst * e.nCh + ch) * (SAF_ANA_HIST * SAF_HOP) + fn;
    auto ld_in = [&](const float* base) {       /* uniform 64-bit base in scalar registers + 32-bit byte offset per lane */
        const unsigned long long b = (unsigned long long)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo) + offInB);
    };
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; k++) w[k] = g.win[k * SAF_HOP + fn];

    /* ---- FFT role: thread = (hop of the sub-chunk, lane j of its group of 8) ---- */
    const int ff = tid >> 3, fj = tid & 7;
    const TwCol twJ{ s_twJ + fj };

    /* ---- main-pass role: lane = bin pair (k, 128-k), k = lane + 1 ---- */
    const int mk = lane + 1;
    const float2 Wk = g.tw256[mk];
    float gk[D], gm[D];

    /* ---- overlap-add role: thread = sample position; frame history of the 9 hops before the launch ---- */
    float gl[D][EQ_OLA + 9], gr[D][EQ_OLA + 9];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float* h = e.syn_rd + (long long)d * e.syn_d + ((long long)inst * e.nCh + ch) * SAF_SYN_HIST * 256;
#pragma unroll
        for (int i = 0; i < 9; i++) { const float a = h[i * 256 + tid], b = h[i * 256 + 128 + tid]; gl[d][i] = c0 > 0 ? 0.0f : a; gr[d][i] = c0 > 0 ? 0.0f : b; }
#pragma unroll
        for (int i = 9; i < EQ_OLA + 9; i++) gl[d][i] = gr[d][i] = 0.0f;
    }

    /* input cursor (uniform): element offset of the next hop inside this instance's input */
    const int hFirst = hs - SAF_ANA_HIST < 0 ? 0 : hs - SAF_ANA_HIST;      /* first hop read from the input buffer */
    int curSub = hFirst % T;
    long long curOff = (long long)(hFirst / T) * e.in_frame + curSub * SAF_HOP;
    auto advance = [&]() { curSub++; curOff += SAF_HOP; if (curSub == T) { curSub = 0; curOff += e.in_frame - (long long)T * SAF_HOP; } };

    /* the 15 hops before the first processed hop (state of the previous call, or — in a later chunk — the input itself) and the
     * first sub-chunk: one memory round trip */
    float xin[SUB + 9], xw[6];
#pragma unroll
    for (int i = 0; i < SAF_ANA_HIST; i++) {
        const int h = hs - SAF_ANA_HIST + i;                 /* uniform */
        float v;
        if (h < 0) v = hist[(SAF_ANA_HIST + h) * SAF_HOP];
        else { v = ld_in(inBase + curOff) * scale; advance(); }
        if (i < 6) xw[i] = v; else xin[i - 6] = v;
    }
#pragma unroll
    for (int i = 0; i < SUB; i++) {
        xin[9 + i] = ld_in(inBase + curOff) * scale;
        if (hs + i + 1 < H) advance();
    }
    __syncthreads();                                         /* s_gain, s_twJ */
    /* the gains carry the 1/2 of the real-FFT split and the 1/256 of the inverse transform (1/2 of the packing, 1/128 of
     * saf_rfft_backward, saf_utility_fft.c:751): powers of two, so nothing changes in the rounding */
    const float GS = 1.0f / 256.0f;
    float sc[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        gk[d] = 0.5f * GS * s_gain[d][mk + 4];              /* band of bin k >= 5 is k + 4 (lanes k < 5 idle in the bin phase) */
        gm[d] = 0.5f * GS * s_gain[d][132 - mk];            /* band of bin 128 - k */
        sc[d] = uni ? s_gain[d][0] : 1.0f;                  /* frame scale of the overlap-add */
    }
    const int hb = (lane & 3) + 1;                          /* hybrid items (wave 0): lane = (lagged hop u, bin b = 1..4) */

    /* ---- prologue: hops hs-6 .. hs-1 (ring positions 0 .. 5): bins 1..4 for the hybrid FIR; hops -3 .. -1 are the first
     *      three lagged slots of sub-chunk 0 ---- */
#pragma unroll
    for (int t = 0; t < 6; t++) {
        float fe = 0.0f, fo = 0.0f;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int q = t + k;                             /* index into the 15 hops -15 .. -1 */
            const float xv = q < 6 ? xw[q < 6 ? q : 0] : xin[q >= 6 ? q - 6 : 0];
            if (k & 1) fo = fmaf(xv, w[k], fo); else fe = fmaf(xv, w[k], fe);
        }
        float* slot = s_ring + t * SLOT;
        slot[fn] = fe; slot[128 + fn] = fo;
    }
    lds_barrier();
    if (!uni && ff < 6) {
        float* slot = s_ring + ff * SLOT;
        fft128_slot<false>(slot, fj, twJ, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (fj >= 1 && fj <= 4) s_low[(hs + ff - 6 + 64) & (LOWR - 1)][fj - 1] = ana_bin_lo(slot, 0, fj, s_twl[fj]);
    }
    lds_barrier();      /* the first fold below wraps into ring positions 0 and 1 (hops 14, 15): the warm-up FFTs must be done with them */

    float* zBase[D];                                         /* uniform: the stores address  scalar base + 4 * tid */
#pragma unroll
    for (int d = 0; d < D; d++) zBase[d] = e.z + (long long)d * e.z_d + (long long)inst * e.z_inst + (long long)ch * e.z_ch;
    int pN = 6;                                              /* ring position of hop s0 = (s0 + 6) % ERING */

    for (int s0 = hs; s0 < H; s0 += SUB) {
        const int n = min(SUB, H - s0);
        const bool emit = s0 >= c0;                         /* the warm-up sub-chunk of a later chunk only rebuilds the frame history */
        /* 1. window + fold of the new hops (afSTFT_internal.c:276-301) -> ring position (hop + 6) % ERING */
#pragma unroll
        for (int t = 0; t < SUB; t++) {
            if (t < n) {
                float fe = 0.0f, fo = 0.0f;
#pragma unroll
                for (int i = 0; i < 5; i++) { fe = fmaf(xin[t + 2 * i], w[2 * i], fe); fo = fmaf(xin[t + 2 * i + 1], w[2 * i + 1], fo); }
                const int pos = pN + t >= ERING ? pN + t - ERING : pN + t;
                float* slot = s_ring + pos * SLOT;
                slot[fn] = fe; slot[128 + fn] = fo;
            }
        }
        /* the last sub-chunk records the new input history (the last 15 hops); a partial one re-reads them below */
        const bool more = s0 + SUB < H;
        if (!more && n == SUB && e.hist_wr && last) {
            float* dst = e.hist_wr + ((long long)inst * e.nCh + ch) * SAF_ANA_HIST * SAF_HOP + fn;
#pragma unroll
            for (int row = 0; row < SAF_ANA_HIST; row++) dst[row * SAF_HOP] = xin[SUB + 9 - SAF_ANA_HIST + row];
        }
        /* slide the window, prefetch the next sub-chunk (consumed before the output stores of phase 5: vmcnt is in order) */
#pragma unroll
        for (int i = 0; i < 9; i++) xin[i] = xin[i + SUB];
        float xl[SUB];
        if (more) {
#pragma unroll
            for (int i = 0; i < SUB; i++) {
                xl[i] = ld_in(inBase + curOff);
                if (s0 + SUB + i + 1 < H) advance();
            }
        }
        const int pL = pN >= 3 ? pN - 3 : pN - 3 + ERING;      /* ring position of the first lagged hop s0 - 3 */
        auto lag_slot = [&](int u) { const int pos = pL + u >= ERING ? pL + u - ERING : pL + u; return s_ring + pos * SLOT; };
        if (!uni) {
            lds_barrier();                                   /* B1 */
            /* 2. 256-point real FFT as a 128-point complex FFT, in place; bins 1..4 of the new hop -> s_low */
            if (ff < n) {
                float* slot = s_ring + (pN + ff >= ERING ? pN + ff - ERING : pN + ff) * SLOT;
                fft128_slot<false>(slot, fj, twJ, 0);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (fj >= 1 && fj <= 4) s_low[(s0 + ff + 64) & (LOWR - 1)][fj - 1] = ana_bin_lo(slot, 0, fj, s_twl[fj]);
            }
            lds_barrier();                                   /* B2 */
            /* 3. bins k and 128-k of the lagged slots (lagged hop hl = s0 - 3 + u): real-FFT split (kiss_fftr.c:86-123), gains,
             *    half-complex packing (kiss_fftr.c:125-161), in place.  2 Z'[k] = E + i O, 2 Z'[128-k] = conj(E - i O) with
             *    E = B[k] + conj B[128-k], O = (B[k] - conj B[128-k]) e^{+2 pi i k / 256}. */
            auto bin_pair = [&](float* slot, int u, int k, float2 W, const float (&ga)[D], const float (&gb)[D], bool hybrid, const float2 (&Bh)[D]) {
                const float2 Zk = *reinterpret_cast<const float2*>(slot + 2 * k);
                const float2 Zm = *reinterpret_cast<const float2*>(slot + 2 * (128 - k));
                const float2 ee = make_float2(Zk.x + Zm.x, Zk.y - Zm.y);
                const float2 dd = make_float2(Zk.x - Zm.x, Zk.y + Zm.y);
                const float2 tt = cmul(W, dd);
                const float2 Xk = make_float2(ee.x + tt.y, ee.y - tt.x);          /* 2 X[k] */
                const float2 Xm = make_float2(ee.x - tt.y, -ee.y - tt.x);         /* 2 X[128-k] */
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const float2 Bk = hybrid ? Bh[d] : make_float2(ga[d] * Xk.x, ga[d] * Xk.y);
                    const float2 Bm = make_float2(gb[d] * Xm.x, gb[d] * Xm.y);
                    const float2 E = make_float2(Bk.x + Bm.x, Bk.y - Bm.y);
                    const float2 Dd = make_float2(Bk.x - Bm.x, Bk.y + Bm.y);
                    const float2 O = make_float2(Dd.x * W.x + Dd.y * W.y, Dd.y * W.x - Dd.x * W.y);      /* Dd * conj(W) */
                    float* o = d == 0 ? slot : s_out1 + u * SLOT;
                    *reinterpret_cast<float2*>(o + 2 * k) = make_float2(E.x - O.y, E.y + O.x);
                    if (k != 64) *reinterpret_cast<float2*>(o + 2 * (128 - k)) = make_float2(E.x + O.y, O.x - E.y);
                }
            };
            /* a wave owns the lagged slots 8 wv .. 8 wv + 7 through phases 3 and 4 */
            if (lane < 32) {
                /* hybrid bins (afSTFT_internal.c:595-619): band 2b-1 / 2b = 0.5 S_{hl}[b] -/+ (or +/-) g_b,
                 * g_b = i (C1 (S_{hl+3} - S_{hl-3}) + C2 (S_{hl+1} - S_{hl-1})); then the gains of the two half-bands and their sum
                 * (afHybridInverse, :625-653) */
                const int u = 8 * wv + (lane >> 2);
                if (u < n) {
                    const int hl = s0 - 3 + u + 64;
                    const float2 Dk = s_low[hl & (LOWR - 1)][hb - 1];
                    const float2 S0 = s_low[(hl + 3) & (LOWR - 1)][hb - 1], S2 = s_low[(hl + 1) & (LOWR - 1)][hb - 1];
                    const float2 S4 = s_low[(hl - 1) & (LOWR - 1)][hb - 1], S6 = s_low[(hl - 3) & (LOWR - 1)][hb - 1];
                    float gre, gim;
                    gre = -COEFF1 * S0.y;          gim = COEFF1 * S0.x;
                    gre -= COEFF2 * S2.y;          gim += COEFF2 * S2.x;
                    gre += COEFF2 * S4.y;          gim -= COEFF2 * S4.x;
                    gre += COEFF1 * S6.y;          gim -= COEFF1 * S6.x;
                    const float dr = Dk.x * 0.5f, di = Dk.y * 0.5f;
                    const float sgn = (hb & 1) ? -1.0f : 1.0f;
                    const float2 lo = make_float2(dr + sgn * gre, di + sgn * gim), hi = make_float2(dr - sgn * gre, di - sgn * gim);
                    float2 Bh[D]; float ghm[D];
#pragma unroll
                    for (int d = 0; d < D; d++) {
                        const float gh1 = GS * s_gain[d][2 * hb - 1], gh2 = GS * s_gain[d][2 * hb];
                        ghm[d] = 0.5f * GS * s_gain[d][132 - hb];
                        Bh[d] = make_float2(gh1 * lo.x + gh2 * hi.x, gh1 * lo.y + gh2 * hi.y);
                    }
                    bin_pair(lag_slot(u), u, hb, s_twl[hb], ghm, ghm, true, Bh);
                }
            } else if (lane < 40) {
                /* DC and Nyquist: X[0] = Re Z[0] + Im Z[0], X[128] = Re Z[0] - Im Z[0]; packed back as (B0 + B128, B0 - B128) */
                const int u = 8 * wv + lane - 32;
                if (u < n) {
                    float* slot = lag_slot(u);
                    const float2 Z0 = *reinterpret_cast<const float2*>(slot);
                    const float X0 = Z0.x + Z0.y, X128 = Z0.x - Z0.y;
#pragma unroll
                    for (int d = 0; d < D; d++) {
                        const float B0 = GS * s_gain[d][0] * X0, B128 = GS * s_gain[d][132] * X128;
                        float* o = d == 0 ? slot : s_out1 + u * SLOT;
                        *reinterpret_cast<float2*>(o) = make_float2(B0 + B128, B0 - B128);
                    }
                }
            }
            if (mk >= 5) {
                const float2 none[D] = {};
#pragma unroll 2
                for (int i = 0; i < SUB / 2; i++) {
                    const int u = 8 * wv + i;
                    if (u < n) bin_pair(lag_slot(u), u, mk, Wk, gk, gm, false, none);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            /* 4. inverse FFT of the wave's own 8 slots, in place: frame samples 2m, 2m+1 = Re, Im z[m] */
            if (ff < n) {
                fft128_slot<true>(lag_slot(ff), fj, twJ, 0);
                if (D > 1) fft128_slot<true>(s_out1 + ff * SLOT, fj, twJ, 0);
            }
            lds_barrier();                                   /* B3 */
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < SUB; i++) xin[9 + i] = xl[i] * scale;
        }
        /* 5. 10-segment overlap-add, oldest frame first (afSTFT_internal.c:396-444): the hop emitted at s0 + t is
         *    sum_k w[k*128+n] * frame_{t-k}[(k&1)*128 + n] */
#pragma unroll
        for (int half = 0; half < SUB / EQ_OLA; half++) {
            const int nh = min(EQ_OLA, n - half * EQ_OLA);
            if (nh > 0) {
#pragma unroll
                for (int d = 0; d < D; d++) {
#pragma unroll
                    for (int u = 0; u < EQ_OLA; u++) {
                        if (u < nh) {
                            const int uu = half * EQ_OLA + u;
                            const float* slot = (d == 0 || uni) ? lag_slot(uu) : s_out1 + uu * SLOT;
                            gl[d][9 + u] = slot[tid] * sc[d]; gr[d][9 + u] = slot[128 + tid] * sc[d];
                            float acc = 0.0f;
#pragma unroll
                            for (int k = 9; k >= 0; k--) acc = fmaf(w[k], (k & 1) ? gr[d][9 + u - k] : gl[d][9 + u - k], acc);
                            if (emit) (zBase[d] + (long long)(s0 + uu) * SAF_HOP)[tid] = acc;
                        }
                    }
                    if (nh == EQ_OLA) {
#pragma unroll
                        for (int i = 0; i < 9; i++) { gl[d][i] = gl[d][i + EQ_OLA]; gr[d][i] = gr[d][i + EQ_OLA]; }
                    } else {                                 /* partial pass: the 9 newest frames sit at nh .. nh+8 */
#pragma unroll
                        for (int i = 0; i < 9; i++) {
                            float a = gl[d][i], b = gr[d][i];
#pragma unroll
                            for (int q = 1; q < EQ_OLA; q++) if (q == nh) { a = gl[d][i + q]; b = gr[d][i + q]; }
                            gl[d][i] = a; gr[d][i] = b;
                        }
                    }
                }
            }
        }
        pN = pN + SUB >= ERING ? pN + SUB - ERING : pN + SUB;
        /* (no barrier: the next fold writes, and this overlap-add read, only the thread's own sample position of every slot) */
    }

    if (((H - hs) % SUB) != 0 && e.hist_wr && last) {       /* partial last sub-chunk: the last 15 hops of [old history | input] */
        int hh = H - SAF_ANA_HIST < 0 ? 0 : H - SAF_ANA_HIST;
        int fr = hh / T, sb = hh - fr * T;
        long long off = (long long)fr * e.in_frame + sb * SAF_HOP;
        float* dst = e.hist_wr + ((long long)inst * e.nCh + ch) * SAF_ANA_HIST * SAF_HOP + fn;
#pragma unroll
        for (int row = 0; row < SAF_ANA_HIST; row++) {
            const int h = H - SAF_ANA_HIST + row;
            const float v = h < 0 ? hist[(SAF_ANA_HIST + h) * SAF_HOP] : ld_in(inBase + off) * scale;
            dst[row * SAF_HOP] = v;
            if (h >= 0) { sb++; off += SAF_HOP; if (sb == T) { sb = 0; off += e.in_frame - (long long)T * SAF_HOP; } }
        }
    }
    if (e.syn_wr && last) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            float* h = e.syn_wr + (long long)d * e.syn_d + ((long long)inst * e.nCh + ch) * SAF_SYN_HIST * 256;
#pragma unroll
            for (int i = 0; i < 9; i++) { h[i * 256 + tid] = gl[d][i]; h[i * 256 + 128 + tid] = gr[d][i]; }
        }
    }
}

void launch_eq(const EqLaunch& e)
{
    if (e.H <= 0 || e.nCh <= 0 || e.nInst <= 0) return;
    if (e.D != 1 && e.D != 2) SAF_FATAL("filterbank equaliser: D must be 1 or 2");
    if ((unsigned long long)(e.nChIn > 0 ? e.nChIn : 1) * (unsigned long long)(e.in_ch < 0 ? -e.in_ch : e.in_ch) * 4ull >= (1ull << 32))
        SAF_FATAL("filterbank equaliser: one instance's channel block exceeds 4 GiB (split the call)");
    EqArgs g;
    g.e = e;
    g.win = dev_window(0, 0);
    g.twJ = dev_twiddles();
    g.tw256 = g.twJ + 128;
    /* time chunks only when the (channel, instance) grid leaves most of the chip idle: every extra chunk processes 16 more hops.
     * Aim at ~1024 workgroups with chunks of >= 64 hops (multiples of 16). */
    g.chunk = e.H;
    const long long wgs = (long long)e.nCh * e.nInst;
    if (wgs < 512 && e.H >= 128) {
        int nChunks = (int)((1024 + wgs - 1) / wgs);
        if (nChunks > e.H / 64) nChunks = e.H / 64;
        if (nChunks > 1) g.chunk = ((e.H + nChunks - 1) / nChunks + SUB - 1) / SUB * SUB;
    }
    const dim3 grid(e.nCh, e.nInst, (e.H + g.chunk - 1) / g.chunk);
    KernelTimer kt("afstft_eq");
    if (e.D == 1) hipLaunchKernelGGL(afstft_eq_kernel<1>, grid, dim3(128), 0, stream(), g);
    else          hipLaunchKernelGGL(afstft_eq_kernel<2>, grid, dim3(128), 0, stream(), g);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
