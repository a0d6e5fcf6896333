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
#include "mfma_tile.h"

namespace saf {

#define ERING 20        /* slots in the ring: 16 new hops + 3 lagged + 1 (a multiple of 4: the four FFT groups of a lane
                         * group stay 16 banks apart across the wrap) */
#define LOWR  32        /* hops of bins 1..4 kept for the hybrid FIR (power of two >= 16 + 7) */
#ifndef EQ_OLA
#define EQ_OLA 16          /* frames per overlap-add pass: their LDS reads are issued together, the history shifts once per pass */
#endif
#ifndef EQ_BINB
#define EQ_BINB 4           /* slots whose bin-pair reads are in flight together (bins phase) */
#endif
#ifndef EQ_OLA3
#define EQ_OLA3 8          /* ... of the cooperative form: the decode's operand is in flight during the overlap-add, 16 registers less for the history */
#endif
#ifndef EQ_MINWAVES
#define EQ_MINWAVES 3       /* waves per SIMD the one-output kernel is compiled for (168 registers) */
#endif

struct EqArgs { EqLaunch e; const float* win; const float2* twJ; const float2* tw256; int chunk; unsigned long long* stamps;
                unsigned* done; int prio;         /* publishing launches: [nInst] counters of finished workgroups (see the end of the kernel) */
                EqDecodeTail dec; unsigned target;      /* MODE 2: the decode blocks behind the channel blocks of every instance */
                EqCoop co;                              /* MODE 3: the cooperative decode (below) */
                const int* runFlag; };                  /* MODE 0, when set: the launch does nothing unless *runFlag != 0 (the re-run behind MODE 3) */
#ifdef EQ_STAMPS        /* diagnostic build only: cycles per phase of every 64th workgroup (tools/eq_stamps.py) */
#define STAMP(i) do { if (stampOn && lane == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stampAcc[i] += t_ - stampT; stampT = t_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

/* The load / store optimizer pairs neighbouring 8-byte LDS accesses into ds_read2_b64 / ds_write2_b64.  On gfx950 a ds_read2_b64
 * costs 9 LDS cycles against 2 x 3 for two ds_read_b64 (ds_write2_b64: 9 against 2 x 5; tools/probes/lds_bank.hip,
 * profiles/r02_lds_bank.txt), and this kernel keeps the LDS pipe busy 60 % of the time: unpaired it runs 1.8 % faster
 * (same box: 1.618 -> 1.589 ms).  Device pass only: the host pass does not know the feature. */
#if defined(__HIP_DEVICE_COMPILE__)
#define EQ_NO_DS_PAIRING __attribute__((target("no-load-store-opt")))
#else
#define EQ_NO_DS_PAIRING
#endif
/* MODE 2, small launches (one block of one handle: the host-pointer ambi_dec_process): the time-domain decode  out = sum_d M_d z_d
 * rides in the SAME launch, as extra workgroups (blockIdx.x >= nCh) behind the channel workgroups of their instance: they
 * wait until the instance's counter says that all its channels have published their z, then multiply (one 64 x 128 tile per
 * unit, half a tile's operands in flight at a time: latency is what matters here, not bandwidth).  One launch per call instead of
 * two: the host-pointer path is bound by launches (profiles/r03_host_pointer.txt).  The launcher only uses this mode when every
 * workgroup of the launch is resident at once, and a poll that is not answered gives up and reports through err[0]. */
template <int D>
__device__ __forceinline__ void decode_tail(const EqLaunch& e, const EqDecodeTail& r, const unsigned* done, unsigned target,
                                            float* sA0, float* sA1, int* s_ok, int inst, int gi)
{
    const int tid = threadIdx.x, lane = tid & 63, rt = tid >> 6, kh = lane >> 5;
    {
        const float4* Ag = reinterpret_cast<const float4*>(r.Mfrag + (long long)inst * r.m_inst);
#pragma unroll
        for (int i = 0; i < 8; i++) reinterpret_cast<float4*>(sA0)[tid + 128 * i] = Ag[tid + 128 * i];
        if (D > 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) reinterpret_cast<float4*>(sA1)[tid + 128 * i] = Ag[1024 + tid + 128 * i];
        }
    }
    if (tid == 0) {
        int ok = 0;
        for (int it = 0; it < 2000000; it++) {          /* bounded: ~1 s */
            const unsigned v = __hip_atomic_load(done + inst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(v - target) >= 0) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (ok) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else
            __hip_atomic_store(r.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *s_ok = ok;
    }
    __syncthreads();
    if (!*s_ok) return;
    const int nColTiles = r.F / 128, units = r.nFrames * nColTiles;
    const int u0 = gi * r.G, u1 = min(u0 + r.G, units);
    const int lastRow = e.nCh - 1;
    for (int u = u0; u < u1; u++) {
        const int fr = u / nColTiles, ct = u - fr * nColTiles;
        const int col = ct * 128 + 4 * (lane & 31);
        Tile128 t;
        tile_zero(t);
#pragma unroll
        for (int d = 0; d < D; d++) {
            const float* Z = e.z + (long long)d * e.z_d + (long long)inst * e.z_inst + (long long)fr * r.F + col;
            const float* As = (d == 0 ? sA0 : sA1) + rt * 2048 + lane;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                float4 b[16];
#pragma unroll
                for (int i = 0; i < 16; i++) b[i] = *reinterpret_cast<const float4*>(Z + (long long)min(2 * (16 * half + i) + kh, lastRow) * e.z_ch);
#pragma unroll
                for (int i = 0; i < 16; i++) tile_step(t, As[(16 * half + i) * 64], b[i]);
            }
        }
        float* Y = r.Y + (long long)inst * r.y_inst + (long long)fr * r.y_frame + col;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int row = rt * 32 + tile_row(q, lane);
            if (row < r.nRowsY) *reinterpret_cast<float4*>(Y + (long long)row * r.y_row) = make_float4(t.c[0][q], t.c[1][q], t.c[2][q], t.c[3][q]);
        }
    }
}

/* MODE 3 — the decode INSIDE the equaliser launch (order 7: 64 SH channels, 64 loudspeakers, one dense decoder; optional, off by
 * default: measured slower than the two kernels, profiles/r03_coop_experiment.txt).
 * The 64 channel workgroups of an instance hand their z to each other through the pipeline's z buffer (EqLaunch::z, the same layout
 * the stand-alone GEMM reads) and each of them computes 1/64 of the instance's decode per sub-chunk: workgroup ch owns columns
 * [32 ch, 32 ch + 32) of the sub-chunk's 2048 samples — wave w the loudspeaker rows 32 w .. 32 w + 31 — as 32 k-pair steps of
 * v_mfma_f32_32x32x2_f32 whose B operand is one 4-byte load per lane (row = SH channel 2 s + (lane >> 5), 128 contiguous bytes per
 * half wave).
 *   iteration it:  [filterbank of sub-chunk it]
 *                  [publish point: sub-chunk it - 1 published (its stores are a whole iteration old: s_waitcnt vmcnt(0) is free;
 *                   one counter add per wave); decode of sub-chunk it - L confirmed (normally known from the counter read one
 *                   iteration ago, else a bounded poll) + workgroup barrier; its B operand requested (sc1 loads, in flight during
 *                   the overlap-add); the counter of sub-chunk it - L + 1 requested]
 *                  [overlap-add; z re-packed through the dead frame slots into 16-byte WRITE-THROUGH stores]
 *                  [the decode's MFMAs (matrix fragments from L2, two groups of eight in flight; the next sub-chunk's input is
 *                   requested half way), plain stores of the output block]
 * Hand-over form: MI355X_MICROARCH.md "Hand-offs measured with sc1 loads in place of the acquire", third row (every 128-byte line
 * written whole by ONE 16-byte-per-lane sc1 store instruction of one wave; every storing wave adds to the counter after its wait;
 * a 4-byte sc1 poll; a workgroup barrier between the poll and every load; 4-byte sc1 loads).  No L2 write-back, no L1 invalidate.
 * sc1 loads are L2-served: a hand-off address is read only after its counter says so and is not written twice in a launch (no
 * ring), or a line read earlier would be hit again, stale.
 * All 64 workgroups of an instance wait for each other.  They need not all be resident at once for that: workgroups are dispatched
 * in order per XCD, so the lowest unfinished instance has all its workgroups resident or next in line (observed, not promised: a
 * poll that is not answered within ~0.5 s gives up, sets the host-visible flag, and the guarded re-run launches behind this one —
 * the plain kernel and the stand-alone GEMM — recompute the call from the untouched input state). */
#ifndef EQF_L
#define EQF_L 3
#endif
#ifndef EQF_NB
#define EQF_NB 32        /* B operand rows requested at the publish point (the rest after the first MFMAs) */
#endif

template <int D, int MODE>
__global__ __launch_bounds__(128, (D == 1 && MODE != 2) ? EQ_MINWAVES : 2) EQ_NO_DS_PAIRING void afstft_eq_kernel(EqArgs g)      /* (MODE 2: small launches, occupancy does not matter) */
{
    __shared__ __attribute__((aligned(16))) float s_ring[ERING * SLOT];
    __shared__ __attribute__((aligned(16))) float s_out1[D > 1 ? SUB * SLOT : 4];      /* frames of the second output */
    /* The two-output kernel keeps its small tables in the 64-byte pads of the slots (floats 256..271 of a slot are touched by no
     * transform): the W128 twiddles [p][j] in the pads of ring slots 0..15, e^{-2 pi i k/256} (k < 8) in that of slot 16, the
     * bins-1..4 history in the pads of the second output's slots.  42.4 -> 40.3 KB: four workgroups per CU instead of three. */
    constexpr bool PADTAB = D > 1;
    __shared__ float2 s_low_[PADTAB ? 1 : LOWR][4];
    __shared__ float s_gain[D][136];
    __shared__ float2 s_twJ_[PADTAB ? 1 : 8 * 16];
    __shared__ float2 s_twl_[PADTAB ? 1 : 8];                /* e^{-2 pi i k / 256}, k < 8 (bins 1..4) */
    auto lowp = [&](int h, int b) -> float2& {               /* bins 1..4 (b = 0..3) of hop h (mod LOWR) */
        const int hh = h & (LOWR - 1);
        return PADTAB ? *reinterpret_cast<float2*>(s_out1 + (hh & 15) * SLOT + 256 + 8 * (hh >> 4) + 2 * b) : s_low_[PADTAB ? 0 : hh][b];
    };
    auto twl = [&](int k) -> float2& { return PADTAB ? *reinterpret_cast<float2*>(s_ring + 16 * SLOT + 256 + 2 * k) : s_twl_[PADTAB ? 0 : k]; };

    constexpr int OLA_N = MODE == 3 ? EQ_OLA3 : EQ_OLA;      /* frames per overlap-add pass */
    const EqLaunch& e = g.e;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ch = blockIdx.x, inst = blockIdx.y;
    if (MODE == 2) {
        __shared__ int s_ok;
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) __hip_atomic_store(g.dec.err, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)blockIdx.x >= e.nCh) { decode_tail<D>(e, g.dec, g.done, g.target, s_ring, s_out1, &s_ok, inst, (int)blockIdx.x - e.nCh); return; }
    }
    if (MODE == 0 && g.runFlag != nullptr && *g.runFlag == 0) return;      /* re-run launch behind MODE 3: nothing to repair */
    if (MODE == 3 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) __hip_atomic_store(g.co.err, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int T = e.hopsPerFrame;
    /* Beside the decode kernel (launch_dec_stream) the waves of this kernel share their SIMDs with MFMA waves that were
     * dispatched earlier: arbitration is by priority, then age, so at equal priority the MFMA wave takes every issue slot it
     * can use and the vector waves beside it crawl (tools/probes/corun_clock.hip).  This kernel is the critical path of the
     * pair: its waves run at a raised priority, the MFMAs fill what they leave. */
    if (MODE != 3 && g.prio) __builtin_amdgcn_s_setprio(3);
#ifdef EQ_COOP_PRIO
    if (MODE == 3) __builtin_amdgcn_s_setprio(EQ_COOP_PRIO);
#endif
    /* time chunks (grid z) add parallelism when few (channel, instance) workgroups exist: a chunk that does not start the launch
     * first runs the 16 hops before it without emitting them, which rebuilds its overlap-add history (identical arithmetic:
     * the outputs do not depend on how a launch is cut) */
    const int c0 = blockIdx.z * g.chunk;                     /* first hop this workgroup emits */
    const int H = min(c0 + g.chunk, e.H);                    /* end of its hops */
    if (c0 >= H) return;
    const int hs = c0 > 0 ? c0 - SUB : 0;                    /* first hop it processes (launch_eq keeps chunks >= 32 hops: hs >= 15) */
    const bool last = H == e.H;                              /* the workgroup that owns the end of the launch records the state */
    const bool uni = e.uniform != nullptr && e.uniform[inst * SAF_MAXCH + ch] != 0;

    if (PADTAB) *reinterpret_cast<float2*>(s_ring + (tid & 15) * SLOT + 256 + 2 * (tid >> 4)) = g.twJ[tid];      /* [p][j], p = tid & 15 */
    else load_twiddles_pj(s_twJ_, g.twJ, tid);
    if (tid < 8) twl(tid) = g.tw256[tid];
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
    /* Input addressing: hop h of this instance starts at float index (h / T) * in_frame + (h % T) * 128.  The offsets of 16
     * consecutive hops are computed by 16 lanes at once (float reciprocal: exact for h < 2^22, checked by launch_eq) and
     * handed out with v_readlane; every load is  uniform 64-bit base + 32-bit byte offset per lane.  (A scalar cursor with
     * its wrap test per hop cost ~360 scalar instructions and 36 branches per sub-chunk: a seventh of the wave's issue slots.) */
    const float invT = 1.0f / (float)T;
    const int inFrame = (int)e.in_frame;
    auto hop_off_bytes = [&](int h) {                       /* per lane */
        const int fr = (int)(((float)h + 0.5f) * invT);
        return (unsigned)(fr * inFrame + (h - fr * T) * SAF_HOP) * 4u;
    };
    const gbase_t inBaseB = uniform_gbase(inBase);          /* uniform: pinned to scalar registers */
    auto ld_at = [&](unsigned hopOffBytes) { return gld<float>(inBaseB, hopOffBytes + offInB); };
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; k++) w[k] = g.win[k * SAF_HOP + fn];

    /* ---- FFT role: thread = (hop of the sub-chunk, lane j of its group of 8) ---- */
    const int ff = tid >> 3, fj = tid & 7;
    /* twiddle p of this lane: s_twJ[p * 8 + fj], or float2 fj of the pad of ring slot p */
    struct TwAny { const float2* p; int stride; __device__ __forceinline__ float2 operator[](int i) const { return p[i * stride]; } };
    const TwAny twJ{ PADTAB ? reinterpret_cast<const float2*>(s_ring + 256) + fj : s_twJ_ + fj, PADTAB ? SLOT / 2 : 8 };

    /* ---- main-pass role: lane = bin pair (k, 128-k), k = lane + 1 ---- */
    const int mk = lane == 0 ? 0 : lane + 1;                 /* lane 0: DC / Nyquist (k = 0); lanes 1..3 idle (bins 2..4 are hybrid items) */
    const bool mkOn = lane == 0 || lane >= 4;
    const bool mkPair = lane >= 4 && lane != 63;             /* writes Z'[128-k] too (not for k = 0 and k = 64: their partner is themselves) */
    float ca[D], cb[D], cg[D];                               /* the pair's 2 x 2 real map, see below */

    /* ---- overlap-add role: thread = sample position; frame history of the 9 hops before the launch ---- */
    float gl[D][OLA_N + 9], gr[D][OLA_N + 9];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float* h = e.syn_rd + (long long)d * e.syn_d + ((long long)inst * e.nCh + ch) * SAF_SYN_HIST * 256;
#pragma unroll
        for (int i = 0; i < 9; i++) { const float a = h[i * 256 + tid], b = h[i * 256 + 128 + tid]; gl[d][i] = c0 > 0 ? 0.0f : a; gr[d][i] = c0 > 0 ? 0.0f : b; }
#pragma unroll
        for (int i = 9; i < OLA_N + 9; i++) gl[d][i] = gr[d][i] = 0.0f;
    }

    /* input cursor (uniform): element offset of the next hop inside this instance's input */
    /* the 15 hops before the first processed hop (state of the previous call, or — in a later chunk — the input itself) and the
     * first sub-chunk: one memory round trip */
    float xin[SUB + 9], xw[6];
    {
        const int hq = hs - SAF_ANA_HIST + (lane & 15);     /* lanes 0..14: the history hops */
        const unsigned offH = hop_off_bytes(hq < 0 ? 0 : hq);
        const unsigned offN = hop_off_bytes(min(hs + (lane & 15), H - 1));      /* hops beyond the end re-read the last one (never used) */
#pragma unroll
        for (int i = 0; i < SAF_ANA_HIST; i++) {
            const int h = hs - SAF_ANA_HIST + i;             /* uniform */
            float v;
            if (h < 0) v = hist[(SAF_ANA_HIST + h) * SAF_HOP];
            else v = ld_at(__builtin_amdgcn_readlane(offH, i)) * scale;
            if (i < 6) xw[i] = v; else xin[i - 6] = v;
        }
#pragma unroll
        for (int i = 0; i < SUB; i++) xin[9 + i] = ld_at(__builtin_amdgcn_readlane(offN, i)) * scale;
    }
    __syncthreads();                                         /* s_gain, the twiddle tables */
    /* Bins k and 128-k (k >= 5) between the forward and the inverse transform: real-FFT split (kiss_fftr.c:86-123), the gains
     * g_k, g_m of their two bands, half-complex packing (kiss_fftr.c:125-161).  With a = Z[k], b = conj Z[128-k], W = e^{-2 pi i k/256}:
     *     2 X[k] = a (1 - iW) + b (1 + iW),      2 X[128-k] = conj( a (1 + iW) + b (1 - iW) )
     *     Z'[k] = B_k (1 + i W*) + conj(B_m) (1 - i W*),   Z'[128-k] = conj( B_k (1 - i W*) + conj(B_m) (1 + i W*) ),   B = g X
     * and because |W| = 1 every product of the brackets is real or purely imaginary:
     *     Z'[k]     = alpha Z[k]     + i beta conj Z[128-k],     alpha = g_k (1 + Im W) + g_m (1 - Im W)
     *     Z'[128-k] = gamma Z[128-k] + i beta conj Z[k],         gamma = g_k (1 - Im W) + g_m (1 + Im W),   beta = Re W (g_k - g_m)
     * — eight multiply-adds per pair instead of the split / scale / pack sequence (28).  The gains carry the 1/2 of the split
     * and the 1/256 of the inverse transform (1/2 of the packing, 1/128 of saf_rfft_backward, saf_utility_fft.c:751). */
    const float GS = 1.0f / 256.0f;
    float sc[D];
    {
        const float2 Wk = g.tw256[mk];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const float gkd = GS * s_gain[d][mk == 0 ? 0 : mk + 4];           /* band of bin k >= 5 is k + 4; bin 0 is band 0 */
            const float gmd = GS * s_gain[d][132 - mk];                       /* band of bin 128 - k (k = 0: the Nyquist band 132) */
            /* k = 0: Z[0] packs the two real bins: X[0] = Re + Im, X[128] = Re - Im, and back Z'[0] = (B0 + B128, B0 - B128):
             * the same 2 x 2 map with Z[128-k] := Z[0], alpha = g_0 + g_132, beta = g_0 - g_132 */
            ca[d] = mk == 0 ? gkd + gmd : gkd * (1.0f + Wk.y) + gmd * (1.0f - Wk.y);
            cg[d] = gkd * (1.0f - Wk.y) + gmd * (1.0f + Wk.y);
            cb[d] = mk == 0 ? gkd - gmd : Wk.x * (gkd - gmd);
            sc[d] = uni ? s_gain[d][0] : 1.0f;              /* frame scale of the overlap-add */
        }
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
        if (D == 1 && uni) { fe *= sc[0]; fo *= sc[0]; }      /* uniform channel: the folds of hops -3 .. -1 are the frames of output hops 0 .. 2 */
        slot[fn] = fe; slot[128 + fn] = fo;
    }
    lds_barrier();
    if (!uni && ff < 6) {
        float* slot = s_ring + ff * SLOT;
        fft128_slot<false>(slot, fj, twJ, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (fj >= 1 && fj <= 4) lowp(hs + ff - 6 + 64, fj - 1) = ana_bin_lo(slot, 0, fj, twl(fj));
    }
    lds_barrier();      /* the first fold below wraps into ring positions 0 and 1 (hops 14, 15): the warm-up FFTs must be done with them */

    float* zBase[D];                                         /* uniform: the stores address  scalar base + 4 * tid */
#pragma unroll
    for (int d = 0; d < D; d++) zBase[d] = e.z + (long long)d * e.z_d + (long long)inst * e.z_inst + (long long)ch * e.z_ch;
    int pN = 6;                                              /* ring position of hop s0 = (s0 + 6) % ERING */
#ifdef EQ_STAMPS
    const bool stampOn = g.stamps != nullptr && ((blockIdx.y * gridDim.x + blockIdx.x) & 63) == 0;
    unsigned long long stampAcc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, stampT = __builtin_amdgcn_s_memtime();
#endif

    /* ---- MODE 3: ring, counters, this workgroup's slice of the decode ---- */
#ifdef EQ_COOP_CHECK
    auto co_ok = [&](int kind, gbase_t base, unsigned off, int bytes, const void* lo, long long extent, int a, int b) -> bool {
        const long long o = (long long)((unsigned long long)base - (unsigned long long)lo) + off;
        if (o >= 0 && o + bytes <= extent) return true;
        if (g.co.dbg && atomicCAS(reinterpret_cast<unsigned long long*>(g.co.dbg), 0ull, (unsigned long long)kind) == 0ull) {
            g.co.dbg[1] = o; g.co.dbg[2] = extent; g.co.dbg[3] = inst; g.co.dbg[4] = ch; g.co.dbg[5] = a; g.co.dbg[6] = b; g.co.dbg[7] = tid;
        }
        return false;
    };
#define CO_OK(kind, base, off, bytes, lo, extent, a, b) co_ok(kind, base, off, bytes, lo, extent, a, b)
#else
#define CO_OK(kind, base, off, bytes, lo, extent, a, b) true
#endif
    bool coDead = false;                                     /* this wave gave up a poll: no more decodes (the re-run launches recompute the call) */
    bool coReady = false;                                    /* the counter of sub-chunk coIt - 3 had reached the target when it was read one iteration ago */
    unsigned coPoll = 0;                                     /* counter of sub-chunk coIt - 2, requested at the publish point, looked at after the overlap-add */
    float coB[32];                                            /* B operand of the decode in flight: requested at the publish point, multiplied after the overlap-add */
    const int coHl = ch >> 2, coSo = (ch & 3) * 32;          /* this workgroup's 32 columns of a sub-chunk: hop coHl, samples coSo .. coSo + 31 */
    auto co_publish = [&](int k) {                           /* this wave's z of sub-chunk k is in memory (the caller has waited for its stores) */
        if (lane == 0 && CO_OK(1, uniform_gbase(g.co.cnt + (long long)inst * g.co.nSub + k), 0u, 4, g.co.cnt, g.co.cntBytes, k, 0))
            __hip_atomic_fetch_add(g.co.cnt + (long long)inst * g.co.nSub + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto co_request = [&](int k) {                           /* ask for the counter of sub-chunk k (every lane the same word) */
        coPoll = __hip_atomic_load(g.co.cnt + (long long)inst * g.co.nSub + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    /* Every wave confirms for itself that all 128 waves of the instance have published sub-chunk d — normally it already knows
     * (coReady), otherwise one lane polls, bounded — and a workgroup barrier stands between that and every load of the bytes
     * (MI355X_MICROARCH.md "Valid forms", third row: sc1 stores of whole lines, per-wave adds after the wave's own wait, sc1 poll,
     * barrier, sc1 loads). */
    auto co_confirm = [&](int d) {
#ifdef EQ_COOP_KNOBS                                         /* timing experiments only (results are wrong): SAF_HIP_COOP_KNOBS bits */
        if (!(g.prio & 1))
#endif
        if (!coReady && !coDead) {
            int ok = 0;
#ifdef EQ_STAMPS
            if (stampOn && lane == 0) stampAcc[11] += 1;
#endif
            if (lane == 0) {
                const unsigned* c = g.co.cnt + (long long)inst * g.co.nSub + d;
                for (int itp = 0; itp < 400000 && CO_OK(2, uniform_gbase(c), 0u, 4, g.co.cnt, g.co.cntBytes, d, 0); itp++) {          /* bounded: ~0.5 s */
                    const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(v - g.co.target) >= 0) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            ok = __builtin_amdgcn_readfirstlane(ok);
            if (!ok) {
                if (lane == 0) {
                    __hip_atomic_store(g.co.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (g.co.giveUps) atomicAdd(g.co.giveUps, 1);
                }
                coDead = true;
            }
        }
        lds_barrier();
    };
    /* every access below is  uniform base (scalar registers) + 32-bit lane offset: 64-bit addresses per lane for 32 + 32 + 16
     * accesses would not fit beside the filterbank state */
    auto co_loadB = [&](int d, int first, int count) {       /* B operand rows: SH channel c = 2 s + kh of the ring, this workgroup's 32 columns */
        int lq = lane;
        asm volatile("" : "+v"(lq));                          /* (lane offsets recomputed here: hoisted out of the sub-chunk loop they cost registers the filterbank needs) */
        const int kh = lq >> 5, lr = lq & 31;
        int io = inst;
        asm volatile("" : "+s"(io));
        int zch = (int)e.z_ch;
        asm volatile("" : "+s"(zch));
        const gbase_t zb = uniform_gbase(e.z + (long long)io * e.z_inst + (d * SUB + coHl) * SAF_HOP + coSo);
        const unsigned zl = (unsigned)(kh * zch + lr) * 4u;
        const long long zstep = 8ll * zch;                                         /* bytes between the row pairs of consecutive steps */
#pragma unroll
        for (int i = 0; i < 32; i++)
            if (i >= first && i < first + count)
                coB[i] = !CO_OK(3, zb + i * zstep, zl, 4, e.z, g.co.ringBytes, d, i) ? 0.0f :
                        __hip_atomic_load(reinterpret_cast<const float __attribute__((address_space(1)))*>(zb + i * zstep + zl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* global_load_dword sc1 */
    };
    auto co_finish = [&](int d, auto&& midway) {             /* out[:, slice] of sub-chunk d = M z[:, slice]; midway(): run when half the operand registers are free again */
        int lq = lane, io = inst, yr = g.co.y_row;
        asm volatile("" : "+v"(lq));
        asm volatile("" : "+s"(io), "+s"(yr));                /* (bases and row offsets recomputed per decode too: as loop invariants they take the scalar registers of the loop) */
        const int kh = lq >> 5, lr = lq & 31;
        const gbase_t Af = uniform_gbase(g.co.Mfrag + (long long)io * g.co.m_inst + wv * 2048);
        const unsigned al = (unsigned)lq * 4u;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        /* matrix fragments (L1 / L2 hits): two groups of eight in flight, the next group requested when its registers have been read */
        float a0[8], a1[8];
        auto ldA = [&](float (&a)[8], int s0_) {
#pragma unroll
#if defined(EQ_COOP_Y) && (EQ_COOP_Y & 2)             /* timing experiment: no matrix loads */
            for (int i = 0; i < 8; i++) a[i] = (float)(s0_ + i);
#else
            for (int i = 0; i < 8; i++) a[i] = !CO_OK(4, Af + (s0_ + i) * 256u, al, 4, g.co.Mfrag, g.co.mBytes, d, s0_ + i) ? 0.0f : gld<float>(Af + (s0_ + i) * 256u, al);
#endif
        };
        auto mm = [&](const float (&a)[8], int s0_) {
#pragma unroll
#if defined(EQ_COOP_Y) && (EQ_COOP_Y & 1)             /* timing experiment: one vector FMA in place of every MFMA */
            for (int i = 0; i < 8; i++) acc[(s0_ + i) & 15] = fmaf(a[i], coB[s0_ + i], acc[(s0_ + i) & 15]);
#else
            for (int i = 0; i < 8; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], coB[s0_ + i], acc, 0, 0, 0);
#endif
        };
#ifdef EQ_COOP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        ldA(a0, 0); ldA(a1, 8);
        if (EQF_NB < 32) co_loadB(d, EQF_NB, 32 - EQF_NB);
        mm(a0, 0);  __builtin_amdgcn_sched_barrier(0); ldA(a0, 16); __builtin_amdgcn_sched_barrier(0);
        mm(a1, 8);  __builtin_amdgcn_sched_barrier(0); ldA(a1, 24); midway(); __builtin_amdgcn_sched_barrier(0);
        mm(a0, 16); mm(a1, 24);
#ifdef EQ_COOP_PRIO
        __builtin_amdgcn_s_setprio(EQ_COOP_PRIO);
#endif
        const int hg = d * SUB + coHl, fr = hg / T;
        const gbase_t Yb = uniform_gbase(g.co.Y + (long long)io * g.co.y_inst + (long long)fr * g.co.y_frame + (hg - fr * T) * SAF_HOP + coSo + (long long)(wv * 32) * yr);
        const unsigned yl = (unsigned)(4 * kh * yr + lr) * 4u;
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (CO_OK(5, Yb + (long long)((r & 3) + 8 * (r >> 2)) * yr * 4, yl, 4, g.co.Y, g.co.yBytes, d, r))
                gst<float>(Yb + (long long)((r & 3) + 8 * (r >> 2)) * yr * 4, yl, acc[r]);
    };
    int coIt = 0;

    for (int s0 = hs; s0 < H; s0 += SUB) {
        const int n = min(SUB, H - s0);
        if (s0 > hs) {                                       /* the hops requested at the end of the previous iteration */
#pragma unroll
            for (int i = 0; i < SUB; i++) xin[9 + i] *= scale;
        }
        const bool emit = s0 >= c0;                         /* the warm-up sub-chunk of a later chunk only rebuilds the frame history */
        /* 1. window + fold of the new hops (afSTFT_internal.c:276-301) -> ring position (hop + 6) % ERING */
#pragma unroll
        /* (hops beyond the end of a partial last sub-chunk fold the clamped loads into slots nobody reads: no guards, the
         * sixteen folds are straight-line code) */
        for (int t = 0; t < SUB; t++) {
            float fe = 0.0f, fo = 0.0f;
#pragma unroll
            for (int i = 0; i < 5; i++) { fe = fmaf(xin[t + 2 * i], w[2 * i], fe); fo = fmaf(xin[t + 2 * i + 1], w[2 * i + 1], fo); }
            const int pos = pN + t >= ERING ? pN + t - ERING : pN + t;
            float* slot = s_ring + pos * SLOT;
            if (D == 1 && uni) { fe *= sc[0]; fo *= sc[0]; }      /* uniform channel: the fold IS the frame; its gain goes in here */
            slot[fn] = fe; slot[128 + fn] = fo;
        }
        /* the last sub-chunk records the new input history (the last 15 hops); a partial one re-reads them below */
        const bool more = s0 + SUB < H;
        if (!more && n == SUB && e.hist_wr && last) {
            float* dst = e.hist_wr + ((long long)inst * e.nCh + ch) * SAF_ANA_HIST * SAF_HOP + fn;
#pragma unroll
            for (int row = 0; row < SAF_ANA_HIST; row++) dst[row * SAF_HOP] = xin[SUB + 9 - SAF_ANA_HIST + row];
        }
        /* slide the window */
#pragma unroll
        for (int i = 0; i < 9; i++) xin[i] = xin[i + SUB];
        const int pL = pN >= 3 ? pN - 3 : pN - 3 + ERING;      /* ring position of the first lagged hop s0 - 3 */
        auto lag_slot = [&](int u) { const int pos = pL + u >= ERING ? pL + u - ERING : pL + u; return s_ring + pos * SLOT; };
        STAMP(0);                                            /* fold */
        if (!uni) {
            lds_barrier();                                   /* B1 */
            STAMP(1);
            /* 2. 256-point real FFT as a 128-point complex FFT, in place; bins 1..4 of the new hop -> s_low */
            if (ff < n) {
                float* slot = s_ring + (pN + ff >= ERING ? pN + ff - ERING : pN + ff) * SLOT;
                fft128_slot<false>(slot, fj, twJ, 0);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (fj >= 1 && fj <= 4) lowp(s0 + ff + 64, fj - 1) = ana_bin_lo(slot, 0, fj, twl(fj));
            }
            STAMP(2);                                        /* FFT */
            lds_barrier();                                   /* B2 */
            STAMP(3);
            /* 3. bins of the wave's lagged slots 8 wv .. 8 wv + 7 (lagged hop hl = s0 - 3 + u), in place.  Every item reads and
             *    writes only its own two elements of the slot, so the items need no order among themselves.
             *    Hybrid items, lane = (slot u, bin b = 1..4) (afSTFT_internal.c:595-619): band 2b-1 / 2b = 0.5 S_{hl}[b] -/+ (or +/-) g_b,
             *    g_b = i (C1 (S_{hl+3} - S_{hl-3}) + C2 (S_{hl+1} - S_{hl-1})); the gains of the two half-bands, their sum
             *    (afHybridInverse, :625-653) = B[b]; then split / gain / pack against bin 128-b as in the general case:
             *    2 Z'[b] = E + i O, 2 Z'[128-b] = conj(E - i O), E = B[b] + conj B[128-b], O = (B[b] - conj B[128-b]) e^{+2 pi i b / 256}.
             *    Written branch-free (all lanes compute, lanes 32..63 shadow lanes 0..31, only the stores are guarded) and with
             *    its loads first: as a guarded block it was four dependent LDS round trips = 15 % of the sub-chunk. */
            const int uh = 8 * wv + ((lane >> 2) & 7);
            const bool hOn = lane < 32 && uh < n;
            float* hslot = lag_slot(uh < n ? uh : 0);
            const int hl = s0 - 3 + uh + 64;
            const float2 hDk = lowp(hl, hb - 1);
            const float2 hS0 = lowp(hl + 3, hb - 1), hS2 = lowp(hl + 1, hb - 1);
            const float2 hS4 = lowp(hl - 1, hb - 1), hS6 = lowp(hl - 3, hb - 1);
            const float2 hZk = *reinterpret_cast<const float2*>(hslot + 2 * hb);
            const float2 hZm = *reinterpret_cast<const float2*>(hslot + 2 * (128 - hb));
            const float2 hW = twl(hb);
            float hg1[D], hg2[D], hgm[D];
#pragma unroll
            for (int d = 0; d < D; d++) { hg1[d] = GS * s_gain[d][2 * hb - 1]; hg2[d] = GS * s_gain[d][2 * hb]; hgm[d] = 0.5f * GS * s_gain[d][132 - hb]; }
            /* general items: lane = bin pair (k, 128-k), see the derivation of ca / cb / cg above */
            if (mkOn) {
                /* four slots at a time: their eight LDS reads first (one wait), then the arithmetic and the writes */
#pragma unroll
                for (int i0 = 0; i0 < SUB / 2; i0 += EQ_BINB) {
                    float2 bZk[EQ_BINB], bZm[EQ_BINB];
#pragma unroll
                    for (int i = 0; i < EQ_BINB; i++) {
                        const float* slot = lag_slot(8 * wv + i0 + i);     /* (slots beyond the end of a partial last sub-chunk: transformed too, read by nobody) */
                        bZk[i] = *reinterpret_cast<const float2*>(slot + 2 * mk);
                        bZm[i] = *reinterpret_cast<const float2*>(slot + 2 * ((128 - mk) & 127));
                    }
#pragma unroll
                    for (int i = 0; i < EQ_BINB; i++) {
                        const int u = 8 * wv + i0 + i;
                        float* slot = lag_slot(u);
                        const float2 Zk = bZk[i], Zm = bZm[i];
#pragma unroll
                        for (int d = 0; d < D; d++) {
                            float* o = d == 0 ? slot : s_out1 + u * SLOT;
                            *reinterpret_cast<float2*>(o + 2 * mk) = make_float2(fmaf(ca[d], Zk.x, cb[d] * Zm.y), fmaf(ca[d], Zk.y, cb[d] * Zm.x));
                            if (mkPair) *reinterpret_cast<float2*>(o + 2 * (128 - mk)) = make_float2(fmaf(cg[d], Zm.x, cb[d] * Zk.y), fmaf(cg[d], Zm.y, cb[d] * Zk.x));
                        }
                    }
                }
            }
            {
                float gre, gim;
                gre = -COEFF1 * hS0.y;          gim = COEFF1 * hS0.x;
                gre -= COEFF2 * hS2.y;          gim += COEFF2 * hS2.x;
                gre += COEFF2 * hS4.y;          gim -= COEFF2 * hS4.x;
                gre += COEFF1 * hS6.y;          gim -= COEFF1 * hS6.x;
                const float dr = hDk.x * 0.5f, di = hDk.y * 0.5f;
                const float sgn = (hb & 1) ? -1.0f : 1.0f;
                const float2 lo = make_float2(dr + sgn * gre, di + sgn * gim), hi = make_float2(dr - sgn * gre, di - sgn * gim);
                const float2 ee = make_float2(hZk.x + hZm.x, hZk.y - hZm.y);
                const float2 dd = make_float2(hZk.x - hZm.x, hZk.y + hZm.y);
                const float2 tt = cmul(hW, dd);
                const float2 Xm = make_float2(ee.x - tt.y, -ee.y - tt.x);         /* 2 X[128-b] */
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const float2 Bk = make_float2(hg1[d] * lo.x + hg2[d] * hi.x, hg1[d] * lo.y + hg2[d] * hi.y);
                    const float2 Bm = make_float2(hgm[d] * Xm.x, hgm[d] * Xm.y);
                    const float2 E = make_float2(Bk.x + Bm.x, Bk.y - Bm.y);
                    const float2 Dd = make_float2(Bk.x - Bm.x, Bk.y + Bm.y);
                    const float2 O = make_float2(Dd.x * hW.x + Dd.y * hW.y, Dd.y * hW.x - Dd.x * hW.y);      /* Dd * conj(W) */
                    if (hOn) {
                        float* o = d == 0 ? hslot : s_out1 + uh * SLOT;
                        *reinterpret_cast<float2*>(o + 2 * hb) = make_float2(E.x - O.y, E.y + O.x);
                        *reinterpret_cast<float2*>(o + 2 * (128 - hb)) = make_float2(E.x + O.y, O.x - E.y);
                    }
                }
            }
            STAMP(4);                                        /* bins */
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            /* 4. inverse FFT of the wave's own 8 slots, in place: frame samples 2m, 2m+1 = Re, Im z[m] */
            if (ff < n) {
                fft128_slot<true>(lag_slot(ff), fj, twJ, 0);
                if (D > 1) fft128_slot<true>(s_out1 + ff * SLOT, fj, twJ, 0);
            }
            STAMP(5);                                        /* IFFT */
            lds_barrier();                                   /* B3 */
            STAMP(6);
        }
        /* The next sub-chunk's input is requested HERE, into the dead upper part of the window, and first used by the next fold:
         * in flight under the overlap-add.  Not earlier: 16 more live registers across the two FFT phases spill, and every
         * reload of a spilled register is a vector-memory load whose wait (vmcnt is in order) also waits for these 16 loads —
         * measured as 4 000 idle cycles per sub-chunk in the bin phase.  The output stores below are younger than the loads, so
         * the fold's wait for the loads does not wait for them. */
        if (MODE == 3) {
            /* The publish point.  The ring stores of sub-chunk coIt - 1 are a whole iteration old (the wait is free): publish it.
             * Then the decode of sub-chunk coIt - 3 starts — its B operand is requested HERE and multiplied after the overlap-add,
             * like the filterbank's own input — and the counter of sub-chunk coIt - 2 is requested for the next iteration. */
            if (coIt > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                co_publish(coIt - 1);
            }
#ifdef EQ_COOP_KNOBS
            if (!(g.prio & 4))
#endif
            if (coIt >= EQF_L) {
                STAMP(7);
                co_confirm(coIt - EQF_L);
                STAMP(9);
            }
            /* (written in EVERY iteration — zeros while there is nothing to decode yet, loads also by a wave that has given up: an
             * operand array that is written under a condition stays allocated around the whole loop.  No loads "ahead of time"
             * from the hand-off buffer: they would leave lines in this XCD's L2 that the real loads then hit, stale.) */
#if !defined(EQ_COOP_X) || EQ_COOP_X < 2
            if (coIt >= EQF_L) co_loadB(coIt - EQF_L, 0, EQF_NB);
            else {
#pragma unroll
                for (int i = 0; i < 32; i++) coB[i] = 0.0f;
            }
#endif
            if (coIt >= EQF_L - 1) co_request(coIt - (EQF_L - 1));
        }
        auto request_input = [&]() {
            const unsigned offN = hop_off_bytes(min(s0 + SUB + (lane & 15), H - 1));
#pragma unroll
            for (int i = 0; i < SUB; i++) xin[9 + i] = ld_at(__builtin_amdgcn_readlane(offN, i));
        };
        if (more && MODE != 3) request_input();              /* (MODE 3: after the overlap-add — the decode's operand is in flight during it instead) */
        /* 5. 10-segment overlap-add, oldest frame first (afSTFT_internal.c:396-444): the hop emitted at s0 + t is
         *    sum_k w[k*128+n] * frame_{t-k}[(k&1)*128 + n] */
#pragma unroll
        for (int half = 0; half < SUB / OLA_N; half++) {
            const int nh = min(OLA_N, n - half * OLA_N);
            if (nh > 0) {
#pragma unroll
                for (int d = 0; d < D; d++) {
#pragma unroll
                    /* the frames of the pass first (one LDS wait), then the sums; frames beyond the end of a partial pass are read
                     * (slots nobody uses) but neither stored nor kept (see the history update below) */
                    for (int u = 0; u < OLA_N; u++) {
                        const int uu = half * OLA_N + u;
                        const float* slot = (d == 0 || uni) ? lag_slot(uu) : s_out1 + uu * SLOT;
                        if (D == 1) { gl[d][9 + u] = slot[tid]; gr[d][9 + u] = slot[128 + tid]; }
                        else { gl[d][9 + u] = slot[tid] * sc[d]; gr[d][9 + u] = slot[128 + tid] * sc[d]; }
                    }
#pragma unroll
                    for (int u = 0; u < OLA_N; u++) {
                        const int uu = half * OLA_N + u;
                        float acc = 0.0f;
#pragma unroll
                        for (int k = 9; k >= 0; k--) acc = fmaf(w[k], (k & 1) ? gr[d][9 + u - k] : gl[d][9 + u - k], acc);
                        if (MODE == 3) lag_slot(uu)[tid] = acc;      /* the frame in this slot has been read: re-pack z through it (below) */
                        else if (emit && u < nh) {      /* uniform 64-bit base (scalar registers) + 4 * tid */
                            gst<float>(uniform_gbase(zBase[d] + (long long)(s0 + uu) * SAF_HOP), (unsigned)(tid * 4), acc);
                        }
                    }
                    if (nh == OLA_N) {
#pragma unroll
                        for (int i = 0; i < 9; i++) { gl[d][i] = gl[d][i + OLA_N]; gr[d][i] = gr[d][i + OLA_N]; }
                    } else {                                 /* partial pass: the 9 newest frames sit at nh .. nh+8 */
#pragma unroll
                        for (int i = 0; i < 9; i++) {
                            float a = gl[d][i], b = gr[d][i];
#pragma unroll
                            for (int q = 1; q < OLA_N; q++) if (q == nh) { a = gl[d][i + q]; b = gr[d][i + q]; }
                            gl[d][i] = a; gr[d][i] = b;
                        }
                    }
                }
            }
        }
        if (MODE == 3) {
            /* z of this sub-chunk sits in the 16 lagged slots at [tid]; a wave re-reads its own 64 sample positions as 16-byte
             * pieces (4 hops x 256 contiguous bytes per instruction) and stores them write-through: every 128-byte line of the ring
             * is written whole by one store instruction */
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const gbase_t rb = uniform_gbase(zBase[0] + (long long)coIt * (SUB * SAF_HOP) + 64 * wv);
            int lq = lane;
            asm volatile("" : "+v"(lq));
            const unsigned rl = (unsigned)((lq >> 4) * SAF_HOP + 4 * (lq & 15)) * 4u;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int hh = 4 * i + (lq >> 4);
                const saf_v4f v = *reinterpret_cast<const saf_v4f*>(lag_slot(hh) + 64 * wv + 4 * (lq & 15));
                const gbase_t q = rb + 4 * i * SAF_HOP * 4;
                /* The compiler's hazard pass does not see through an inline instruction, so the two hazards of this store are
                 * covered by hand: its scalar base comes from v_readfirstlane (a vector instruction writing a scalar register:
                 * five wait states before a memory instruction may read it — without them the store can go out with the OLD
                 * register contents, i.e. to a wild address), and a store of more than 8 bytes reads its data registers late
                 * (two wait states before they may be rewritten). */
                if (CO_OK(6, q, rl, 16, e.z, g.co.ringBytes, coIt, i))
                asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(rl), "v"(v), "s"(q) : "memory");
            }
            /* the next sub-chunk's input is requested in the middle of the decode (when half its operand registers are free) */
            bool asked = false;
#ifdef EQ_COOP_KNOBS
            if (!(g.prio & 6))
#endif
#if !defined(EQ_COOP_X) || EQ_COOP_X < 1
            if (coIt >= EQF_L && !coDead) { STAMP(7); co_finish(coIt - EQF_L, [&]() { if (more) request_input(); }); asked = true; STAMP(10); }
#endif
            if (more && !asked) request_input();
            coReady = coIt >= EQF_L - 1 && (int)((unsigned)__builtin_amdgcn_readfirstlane((int)coPoll) - g.co.target) >= 0;
            coIt++;
        }
        STAMP(7);                                            /* prefetch wait + OLA + stores */
        pN = pN + SUB >= ERING ? pN + SUB - ERING : pN + SUB;
        /* (no barrier: the next fold writes, and this overlap-add read, only the thread's own sample position of every slot) */
    }

    if (((H - hs) % SUB) != 0 && e.hist_wr && last) {       /* partial last sub-chunk: the last 15 hops of [old history | input] */
        float* dst = e.hist_wr + ((long long)inst * e.nCh + ch) * SAF_ANA_HIST * SAF_HOP + fn;
        const int hq = H - SAF_ANA_HIST + (lane & 15);
        const unsigned offH = hop_off_bytes(hq < 0 ? 0 : hq);
#pragma unroll
        for (int row = 0; row < SAF_ANA_HIST; row++) {
            const int h = H - SAF_ANA_HIST + row;
            dst[row * SAF_HOP] = h < 0 ? hist[(SAF_ANA_HIST + h) * SAF_HOP] : ld_at(__builtin_amdgcn_readlane(offH, row)) * scale;
        }
    }
#ifdef EQ_STAMPS
    if (stampOn && lane == 0) for (int i = 0; i < 12; i++) atomicAdd(&g.stamps[wv * 12 + i], stampAcc[i]);
#endif
    if (e.syn_wr && last) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            float* h = e.syn_wr + (long long)d * e.syn_d + ((long long)inst * e.nCh + ch) * SAF_SYN_HIST * 256;
#pragma unroll
            for (int i = 0; i < 9; i++) { h[i * 256 + tid] = gl[d][i]; h[i * 256 + 128 + tid] = gr[d][i]; }
        }
    }
    if (MODE == 3) {
        /* drain: the last sub-chunk is published, then the decodes that lag behind */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        co_publish(coIt - 1);
#ifdef EQ_COOP_KNOBS
        if (!(g.prio & 4))
#endif
        for (int d = coIt - EQF_L; d < coIt; d++)
            if (d >= 0) {
                if (d > coIt - EQF_L) coReady = false;       /* (the first of them was asked about in the last iteration) */
                STAMP(8);
                co_confirm(d);
                STAMP(9);
#ifdef EQ_COOP_KNOBS
                if (!(g.prio & 2))
#endif
#if !defined(EQ_COOP_X) || EQ_COOP_X < 3
                if (!coDead) { co_loadB(d, 0, EQF_NB); co_finish(d, []() {}); }
#endif
                STAMP(10);
            }
    }
    if (MODE == 1 || MODE == 2) {
        /* Publish this channel's z to the decode kernel that runs beside this one (launch_dec_stream, gemm_kernels.hip): every
         * wave's stores drained, workgroup barrier, agent-scope release (write-back of the XCD's L2), then the instance's counter
         * (MI355X_MICROARCH.md "Valid forms"; the consumer polls, acquires, then loads).  PUBLISH launches have no time chunks:
         * one workgroup per (channel, instance). */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(g.done + inst, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static unsigned long long* g_eq_stamps = nullptr;
static unsigned long long* eq_stamps_buffer()
{
#ifdef EQ_STAMPS
    if (!g_eq_stamps) { HIP_CHECK(hipMalloc((void**)&g_eq_stamps, 24 * sizeof(unsigned long long))); HIP_CHECK(hipMemset(g_eq_stamps, 0, 24 * sizeof(unsigned long long))); }
#endif
    return g_eq_stamps;
}

void launch_eq(const EqLaunch& e, unsigned* done)
{
    if (e.H <= 0 || e.nCh <= 0 || e.nInst <= 0) return;
    if (e.D != 1 && e.D != 2) SAF_FATAL("filterbank equaliser: D must be 1 or 2");
    /* the kernel addresses one instance's samples with 32-bit byte offsets (channel offset + hop offset) from a uniform base */
    const long long chSpan = (long long)(e.nChIn > 0 ? e.nChIn : 1) * e.in_ch;
    const long long hopSpan = (long long)((e.H + e.hopsPerFrame - 1) / e.hopsPerFrame) * e.in_frame + (long long)e.hopsPerFrame * SAF_HOP;
    if (e.in_ch < 0 || e.in_frame < 0 || e.H >= (1 << 22) || (chSpan + hopSpan) * 4 >= (1ll << 32))
        SAF_FATAL("filterbank equaliser: one call spans more than 4 GiB of one instance's input, 2^22 hops or uses negative strides: split the call");
    EqArgs g;
    g.e = e;
    g.win = dev_window(0, 0);
    g.twJ = dev_twiddles();
    g.tw256 = g.twJ + 128;
    g.stamps = eq_stamps_buffer();
    /* time chunks only when the (channel, instance) grid leaves most of the chip idle: every extra chunk processes 16 more hops.
     * Aim at ~1024 workgroups with chunks of >= 64 hops (multiples of 16). */
    g.chunk = e.H;
    const long long wgs = (long long)e.nCh * e.nInst;
    if (wgs < 512 && e.H >= 128 && !done) {
        int nChunks = (int)((1024 + wgs - 1) / wgs);
        if (nChunks > e.H / 64) nChunks = e.H / 64;
        if (nChunks > 1) g.chunk = ((e.H + nChunks - 1) / nChunks + SUB - 1) / SUB * SUB;
    }
    const dim3 grid(e.nCh, e.nInst, (e.H + g.chunk - 1) / g.chunk);
    KernelTimer kt("afstft_eq");
    g.done = done; g.target = 0; g.dec = EqDecodeTail{}; g.co = EqCoop{}; g.runFlag = e.runFlag;
    { static const int p = []() { const char* v = getenv("SAF_HIP_EQ_PRIO"); return v ? atoi(v) : -1; }(); g.prio = p >= 0 ? p : (done != nullptr); }
    if (done) {
        if (e.D == 1) hipLaunchKernelGGL((afstft_eq_kernel<1, 1>), grid, dim3(128), 0, stream(), g);
        else          hipLaunchKernelGGL((afstft_eq_kernel<2, 1>), grid, dim3(128), 0, stream(), g);
    } else {
        /* (experiments: SAF_HIP_EQ_EXTRA_LDS reserves unused LDS per workgroup, i.e. lowers the occupancy) */
        static const int extraLds = []() { const char* v = getenv("SAF_HIP_EQ_EXTRA_LDS"); return v ? atoi(v) : 0; }();
        if (e.D == 1) hipLaunchKernelGGL((afstft_eq_kernel<1, 0>), grid, dim3(128), extraLds, stream(), g);
        else          hipLaunchKernelGGL((afstft_eq_kernel<2, 0>), grid, dim3(128), 0, stream(), g);
    }
    HIP_CHECK(hipGetLastError());
}

bool launch_eq_decode(const EqLaunch& e, const EqDecodeTail& d, unsigned* done, unsigned target)
{
    if (e.H <= 0 || e.nCh <= 0 || e.nInst <= 0) return true;
    if (e.D != 1 && e.D != 2) SAF_FATAL("filterbank equaliser: D must be 1 or 2");
    if (d.F % 128 != 0 || d.nFrames * (d.F / SAF_HOP) != e.H || d.G < 1) return false;
    if (((d.y_inst | d.y_frame | d.y_row | e.z_inst | e.z_ch | e.z_d | d.m_inst) & 3) || ((uintptr_t)d.Y & 15) || ((uintptr_t)e.z & 15)) return false;
    const int units = d.nFrames * (d.F / 128), nDec = (units + d.G - 1) / d.G;
    /* every workgroup of the launch must be resident at once (6 per compute unit on 256 compute units, with a margin for other
     * streams' launches): a waiting decode workgroup then never keeps a channel workgroup of its launch off the chip */
    if ((long long)e.nInst * (e.nCh + nDec) > 768) return false;
    /* ... and only for a few blocks per call: the decode tail is built for latency (half a tile's operands in flight), the
     * stand-alone GEMM for throughput */
    if (units > 16) return false;
    const long long chSpan = (long long)(e.nChIn > 0 ? e.nChIn : 1) * e.in_ch;
    const long long hopSpan = (long long)((e.H + e.hopsPerFrame - 1) / e.hopsPerFrame) * e.in_frame + (long long)e.hopsPerFrame * SAF_HOP;
    if (e.in_ch < 0 || e.in_frame < 0 || e.H >= (1 << 22) || (chSpan + hopSpan) * 4 >= (1ll << 32))
        SAF_FATAL("filterbank equaliser: one call spans more than 4 GiB of one instance's input, 2^22 hops or uses negative strides: split the call");
    EqArgs g;
    g.e = e;
    g.win = dev_window(0, 0);
    g.twJ = dev_twiddles();
    g.tw256 = g.twJ + 128;
    g.stamps = nullptr;
    g.chunk = e.H;
    g.done = done; g.target = target; g.dec = d; g.prio = 0; g.co = EqCoop{}; g.runFlag = nullptr;
    const dim3 grid(e.nCh + nDec, e.nInst, 1);
    KernelTimer kt("afstft_eq_decode");
    if (e.D == 1) hipLaunchKernelGGL((afstft_eq_kernel<1, 2>), grid, dim3(128), 0, stream(), g);
    else          hipLaunchKernelGGL((afstft_eq_kernel<2, 2>), grid, dim3(128), 0, stream(), g);
    HIP_CHECK(hipGetLastError());
    return true;
}

bool launch_eq_coop(const EqLaunch& e, const EqCoop& c)
{
    if (e.H <= 0 || e.nInst <= 0) return true;
    /* order 7 -> 64 loudspeakers, one dense decoder, whole sub-chunks; every lane stores 4 bytes of the output: no alignment rule */
    if (e.D != 1 || e.nCh != SAF_MAXCH || c.nRowsY != 64 || e.H % SUB != 0 || c.nSub * SUB < e.H || c.T < 1 || SAF_HOP * c.T != c.F) return false;
    const long long chSpan = (long long)(e.nChIn > 0 ? e.nChIn : 1) * e.in_ch;
    const long long hopSpan = (long long)((e.H + e.hopsPerFrame - 1) / e.hopsPerFrame) * e.in_frame + (long long)e.hopsPerFrame * SAF_HOP;
    if (e.in_ch < 0 || e.in_frame < 0 || e.H >= (1 << 22) || (chSpan + hopSpan) * 4 >= (1ll << 32))
        SAF_FATAL("filterbank equaliser: one call spans more than 4 GiB of one instance's input, 2^22 hops or uses negative strides: split the call");
    EqArgs g;
    g.e = e;
    g.win = dev_window(0, 0);
    g.twJ = dev_twiddles();
    g.tw256 = g.twJ + 128;
    g.stamps = eq_stamps_buffer();
    g.chunk = e.H;
    g.done = nullptr; g.target = 0; g.dec = EqDecodeTail{}; g.prio = 0; g.co = c; g.runFlag = nullptr;
#ifdef EQ_COOP_KNOBS
    { const char* v = getenv("SAF_HIP_COOP_KNOBS"); g.prio = v ? atoi(v) : 0; }
#endif
    KernelTimer kt("afstft_eq_coop");
    hipLaunchKernelGGL((afstft_eq_kernel<1, 3>), dim3(e.nCh, e.nInst, 1), dim3(128), 0, stream(), g);
    HIP_CHECK(hipGetLastError());
    return true;
}


}  // namespace saf

#ifdef EQ_STAMPS
extern "C" __attribute__((visibility("default"))) void saf_hip_debug_eq_stamps(unsigned long long* out16)
{
    HIP_CHECK(hipStreamSynchronize(saf::stream()));
    if (saf::g_eq_stamps) { HIP_CHECK(hipMemcpy(out16, saf::g_eq_stamps, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost)); HIP_CHECK(hipMemset(saf::g_eq_stamps, 0, 24 * sizeof(unsigned long long))); }
}
#endif
