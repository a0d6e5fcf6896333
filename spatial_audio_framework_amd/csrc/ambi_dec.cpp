/*
 * ambi_dec.cpp — the ambi_dec operator (examples/include/ambi_dec.h:114-520,
 * examples/src/ambi_dec/ambi_dec.c, ambi_dec_internal.c) with its per-block path
 * on the GPU:
 *
 *   inputs -> [afSTFT analysis kernel: SN3D/FuMa conversion folded in]
 *          -> [band-batched MFMA GEMM with the (decoder, order, maxrE, norm) matrix of each band]
 *          -> [afSTFT synthesis kernel] -> outputs
 *
 * The codec state machine, parameter snapshotting and "zero the output when not
 * ready" behaviour follow the reference.  The device pipeline is shared by the
 * single-handle ambi_dec_process (host pointers, one block) and the batched
 * device-pointer entry point (many instances x many blocks per launch).
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "afstft_state.h"
#include "design_host.h"
#include "presets.h"
#include "hrtf_tables.h"
#include "hrir_host.h"
#include <thread>
#include <chrono>

namespace saf {

#define NUM_DECODERS 2
#define NMAT (NUM_DECODERS * SAF_MAX_ORDER)
static int g_ambi_dec_frame_size = 128;    /* default of the reference (ambi_dec_internal.h:65) */

/* Which block path loudspeaker decoding takes (DecPipeline::process):
 *   0  the three-kernel transform path: afSTFT analysis -> per-band MFMA GEMM -> afSTFT synthesis (spectra through HBM)
 *   1  (default) the equaliser path: per-channel filterbank equaliser (eq_kernels.hip) + ONE time-domain GEMM with the dense
 *      decoder(s); channels whose weights are the same in every band skip the transforms
 *   2  as 1, but every channel runs the transforms (what holds for ANY per-band order / decoder / weighting assignment) */
static int g_ambi_dec_time_domain = []() { const char* e = getenv("SAF_HIP_AMBI_DEC_TIME_DOMAIN"); return e ? atoi(e) : 1; }();

/* Decode beside the equaliser (launch_dec_stream): 0 (default) never — the two kernels run one after the other; 1 when the launch
 * is large enough to fill the chip several times; 2 whenever the shape allows (tests).  Off by default because it is slower on
 * MI355X (profiles/r03_overlap_experiment.txt): side by side the two kernels take as long as one after the other (the pair is
 * bound by the 9.8 GB it moves through HBM, not by the vector or matrix pipes), and publishing z per workgroup (an agent-scope
 * release = L2 write-back per workgroup) costs the equaliser kernel another 0.4 ms. */
static unsigned g_coop_target_bias = 0;   /* tests: a bias that no counter reaches makes every cooperative decode give up (the re-run launches must then produce the block) */
static int g_ambi_dec_overlap = []() { const char* e = getenv("SAF_HIP_AMBI_DEC_OVERLAP"); return e ? atoi(e) : 0; }();

static inline void sleep_ms(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }

struct AmbiDec {
    int F, T;
    /* designed codec tables (ambi_dec_codecPars, ambi_dec_internal.h:86-121) */
    std::vector<float> M_dec[NUM_DECODERS][SAF_MAX_ORDER], M_dec_maxrE[NUM_DECODERS][SAF_MAX_ORDER];
    float M_norm[NUM_DECODERS][SAF_MAX_ORDER][2];
    std::string sofa_filepath;
    int hrir_fs = 0;
    std::shared_ptr<HrtfTables> hrtf;      /* itds, VBAP interpolation table, hrtf_fb(_mag) of ambi_dec_codecPars (ambi_dec_internal.h:100-118) */
    /* (ambi_dec_data, ambi_dec_internal.h:127-173) */
    float freqVector[SAF_NBANDS];
    int fs = 48000;
    volatile CODEC_STATUS codecStatus;
    volatile PROC_STATUS procStatus;
    float progressBar0_1;
    char progressBarText[PROGRESSBARTEXT_CHAR_LENGTH];
    int loudpkrs_nDims, new_nLoudpkrs, new_binauraliseLS, new_masterOrder;
    int reinit_hrtfsFLAG, recalc_hrtf_interpFLAG[SAF_MAXCH];
    int masterOrder, orderPerBand[SAF_NBANDS];
    int dec_method[NUM_DECODERS], rE_WEIGHT[NUM_DECODERS], diffEQmode[NUM_DECODERS];
    float transitionFreq;
    int nLoudpkrs;
    float loudpkrs_dirs_deg[SAF_MAXCH][2];
    int useDefaultHRIRsFLAG, enableHRIRsPreProc, binauraliseLS;
    CH_ORDER chOrdering;
    NORM_TYPES norm;
    /* device side */
    bool haveSTFT = false;
    unsigned long long codecEpoch = 0;     /* bumped by every initCodec; pipelines re-read the tables when it changes */
    struct DecPipeline* pipe = nullptr;    /* single-instance pipeline of ambi_dec_process */
    hipStream_t own = nullptr;             /* ... and its stream: handles driven from different host threads do not share a queue */
    PinBuf<float> h_in, h_out;
    DevBuf<float> d_in, d_out;
};

/* -------------------------------------------------------------------------- */
/*  device pipeline for nInst instances                                       */
/* -------------------------------------------------------------------------- */
struct DecPipeline {
    int nInst = 0, F = 0, T = 0, maxFrames = 0, Hmax = 0, nSH = 0, nLS = 0;
    std::vector<AmbiDec*> inst;
    AfState st;
    DevBuf<float2> X, Y;            /* [nInst][133][64][Hmax] */
    DevBuf<float> Afrag;            /* [nInst][NMAT][2][32][64] */
    DevBuf<int> band2mat;           /* [nInst][133] */
    DevBuf<float> chScale;          /* [nInst][64] */
    DevBuf<int> chMap;              /* [nInst][64] */
    /* equaliser path (see eq_kernels.hip): every per-band matrix of ambi_dec is M_d diag(w_{d,n}); the diagonal goes into a
     * per-channel filterbank equaliser, the dense M_d into one time-domain GEMM:  out = sum_d M_d z_d */
    DevBuf<float> eqGains;          /* [nInst][eqD][64][136] w_{d(band), n(band)}[ch] */
    DevBuf<int> eqUni;              /* [nInst][64] 1: the channel's gains are the same in every band */
    DevBuf<float> Mfrag;            /* [nInst][2][2][32][64] the dense decoders M_0, M_1 in MFMA fragment order */
    DevBuf<float> zbuf;             /* [eqD][nInst][64][Hmax * 128] */
    DevBuf<float> zsyn[2];          /* ping-pong: [2][nInst][nSH][9][256] synthesised-frame history of z_d (SH domain) */
    DevBuf<int> zerosI;             /* [nInst][maxFrames] "band -> matrix 0" table of the time-domain GEMM */
    PinBuf<int> errPin;             /* [1] host-visible: a decode workgroup of the one-launch form gave up waiting */
    bool deferFixup = false;        /* host-pointer caller: it synchronises anyway and then calls fixup_if_needed() instead of paying a guarded fix-up launch */
    bool fixPending = false; BandGemmLaunch fixGemm{};
    DevBuf<unsigned> eqDone;        /* [nInst] equaliser workgroups finished, monotonic over the publishing launches */
    DevBuf<int> eqErr;              /* [2] see DecStreamLaunch::err */
    unsigned eqDoneBase = 0;
    int lastOverlap = 0;            /* 1: the last call ran the decode kernel beside the equaliser kernel; 3: inside it (cooperative form) */
    DevBuf<unsigned> coCnt;         /* [nInst][coNSub] waves that have stored their z of a sub-chunk (zeroed before every launch) */
    int coNSub = 0;
    DevBuf<long long> coDbg;        /* -DEQ_COOP_CHECK builds: the first out-of-range access of the cooperative form */
    int zsynPar = 0, eqD = 1;
    std::vector<char> eqDirty, eqTwo;        /* per instance: tables stale; the two decoders are different matrices */
    std::vector<unsigned long long> eqTwoEpoch;
    PinBuf<float> stageG, stageM; PinBuf<int> stageU;
    bool mode2Shadow = false;
    /* The overlap-add history lives in the loudspeaker domain on the transform path (AfState::syn) and in the SH domain on
     * the equaliser path (zsyn).  SH -> loudspeaker is exact (st.syn = sum_d M_d zsyn_d, one small GEMM); the other
     * direction does not exist, so a pipeline that has run the transform path stays on it until its state is cleared. */
    enum { DOM_NONE, DOM_LS, DOM_SH } synDomain = DOM_NONE;
    int lastPath = -1;              /* 0 transform, 1 equaliser (saf_hip_ambi_dec_batch_lastPath) */
    /* binauralised output (ambi_dec.c:543-563) */
    bool bin = false;
    std::shared_ptr<HrtfTables> hrtf;
    DevBuf<float2> Z;               /* [nInst][133][2][Hmax] */
    DevBuf<float2> hrtfInterp;      /* [nInst][64][133][2] */
    DevBuf<float2> HM;              /* [nInst][64][133][2]  HRTFs x decoder: one 2 x nSH matrix per band */
    DevBuf<float> Arow;             /* [nInst][NMAT][64][64] the decoder matrices, row-major, for the fold */
    PinBuf<float> stageArow;
    bool foldDirty = true;
    DevBuf<float> lsDirs;           /* [nInst][64][2] */
    DevBuf<float> freq;             /* [133] centre frequencies at the time of the last interpolation */
    DevBuf<int> lsRecalc;           /* [nInst][64] */
    PinBuf<float> stageD; PinBuf<int> stageR;
    /* host shadows to detect parameter changes between calls */
    struct Shadow {
        unsigned long long epoch = ~0ull;
        int rE[2] = { -1, -1 }, eq[2] = { -1, -1 };
        int b2m[SAF_NBANDS];
        int norm = -1, chOrd = -1;
        bool b2mValid = false;
    };
    std::vector<Shadow> shadow;
    PinBuf<float> stageA; PinBuf<int> stageI; PinBuf<float> stageS;
    /* Staging: every instance has its own slot in the pinned blocks of the small tables, so the uploads of a call are queued
     * without a synchronisation between instances; the call that next has something to upload first waits for these copies. */
    bool stagePending = false;
    std::vector<char> afragDirty;   /* per instance: the 14 effective matrices of the transform path / binaural fold are stale */
    void begin_staging() { if (stagePending) { HIP_CHECK(hipStreamSynchronize(stream())); stagePending = false; } }

    void create(AmbiDec* const* handles, int n, int maxFrames_)
    {
        nInst = n; inst.assign(handles, handles + n);
        F = inst[0]->F; T = F / SAF_HOP; maxFrames = maxFrames_;
        Hmax = (T * maxFrames + 15) & ~15;      /* spectra rows are 128-byte multiples: every kernel moves them as 16-byte vectors */
        nSH = ORDER2NSH(inst[0]->masterOrder); nLS = inst[0]->nLoudpkrs;
        for (int i = 0; i < n; i++) {
            if (inst[i]->F != F) SAF_FATAL("ambi_dec batch: all instances must use the same block size");
            if (inst[i]->codecStatus != CODEC_STATUS_INITIALISED) SAF_FATAL("ambi_dec batch: instance %d is not initialised (call ambi_dec_initCodec)", i);
            if (ORDER2NSH(inst[i]->masterOrder) != nSH || inst[i]->nLoudpkrs != nLS)
                SAF_FATAL("ambi_dec batch: all instances must share master order and loudspeaker count");
            if ((inst[i]->binauraliseLS != 0) != (inst[0]->binauraliseLS != 0)) SAF_FATAL("ambi_dec batch: all instances must agree on the binauraliseLS flag");
            if (inst[i]->binauraliseLS && inst[i]->hrtf != inst[0]->hrtf) SAF_FATAL("ambi_dec batch: binauralising instances must share HRIR set, pre-processing flag and sample rate");
        }
        bin = inst[0]->binauraliseLS != 0;
        st.create(nInst, nSH, bin ? 2 : nLS);
        if (bin) {
            hrtf = inst[0]->hrtf;
            Z.alloc((size_t)nInst * SAF_NBANDS * 2 * Hmax, true);
            hrtfInterp.alloc((size_t)nInst * SAF_MAXCH * SAF_NBANDS * 2); HM.alloc((size_t)nInst * SAF_MAXCH * SAF_NBANDS * 2);
            Arow.alloc((size_t)nInst * NMAT * 64 * 64); stageArow.ensure((size_t)NMAT * 64 * 64);
            lsDirs.alloc((size_t)nInst * SAF_MAXCH * 2); lsRecalc.alloc((size_t)nInst * SAF_MAXCH); freq.alloc(SAF_NBANDS);
            stageD.ensure((size_t)nInst * SAF_MAXCH * 2 + SAF_NBANDS); stageR.ensure((size_t)nInst * SAF_MAXCH);
            for (int i = 0; i < n; i++) for (int ch = 0; ch < SAF_MAXCH; ch++) inst[i]->recalc_hrtf_interpFLAG[ch] = 1;     /* a new pipeline starts without interpolated HRTFs */
        }
        if (bin) ensure_transform_buffers();
        Afrag.alloc((size_t)nInst * NMAT * 64 * 64);
        band2mat.alloc((size_t)nInst * SAF_NBANDS);
        chScale.alloc((size_t)nInst * SAF_MAXCH);
        chMap.alloc((size_t)nInst * SAF_MAXCH);
        shadow.assign(nInst, Shadow());
        eqDirty.assign(nInst, 1); eqTwo.assign(nInst, 0); eqTwoEpoch.assign(nInst, ~0ull); afragDirty.assign(nInst, 1);
        stageA.ensure((size_t)NMAT * 64 * 64); stageI.ensure((size_t)nInst * (SAF_NBANDS + SAF_MAXCH)); stageS.ensure((size_t)nInst * SAF_MAXCH);
    }

    /* spectra of the transform path (4.5 GB each at 256 instances x 64 blocks): allocated when that path first runs */
    void ensure_transform_buffers()
    {
        /* zeroed once: the analysis only ever writes the first nSH channel rows, the GEMM reads all 64
         * (against zero matrix columns) — stale NaNs there would poison the product */
        if (!X.p) X.alloc((size_t)nInst * SAF_NBANDS * SAF_MAXCH * Hmax, true);
        if (!bin && !Y.p) Y.alloc((size_t)nInst * SAF_NBANDS * SAF_MAXCH * Hmax, true);      /* binauralised output never forms the loudspeaker spectra */
    }
    void ensure_eq_buffers(int D)
    {
        if (!eqGains.p || D > eqD) {
            HIP_CHECK(hipStreamSynchronize(stream()));
            const int oldD = eqGains.p ? eqD : 0;
            eqD = D;
            eqGains.alloc((size_t)nInst * eqD * SAF_MAXCH * 136);
            zbuf.alloc((size_t)eqD * nInst * SAF_MAXCH * Hmax * SAF_HOP, true);
            std::fill(eqDirty.begin(), eqDirty.end(), 1);
            if (!eqUni.p) {
                eqUni.alloc((size_t)nInst * SAF_MAXCH); Mfrag.alloc((size_t)nInst * 2 * 64 * 64); zerosI.alloc((size_t)nInst * maxFrames);
                stageG.ensure((size_t)nInst * 2 * SAF_MAXCH * 136); stageM.ensure((size_t)nInst * 2 * 64 * 64); stageU.ensure((size_t)nInst * SAF_MAXCH);
                for (int i = 0; i < 2; i++) zsyn[i].alloc((size_t)2 * nInst * nSH * SAF_SYN_HIST * 256);
                eqDone.alloc(nInst); eqErr.alloc(2); eqDoneBase = 0;
                errPin.ensure(1); errPin.p[0] = 0;
            }
            (void)oldD;     /* zsyn holds both outputs from the start: a pipeline that goes from one dense matrix to two keeps z_0's history, z_1's starts from zero */
        }
    }
    /* after the caller's synchronisation: the decode of the last one-launch call, should one of its workgroups have given up */
    void fixup_if_needed()
    {
        if (!fixPending) return;
        fixPending = false;
        if (__atomic_load_n(errPin.p, __ATOMIC_ACQUIRE) == 0) return;
        fixGemm.runFlag = nullptr;
        launch_band_gemm(fixGemm);
        HIP_CHECK(hipStreamSynchronize(stream()));
    }
    /* afSTFT_clearBuffers for the whole pipeline (ambi_dec.c:218,225) */
    void clear_state()
    {
        st.clear();
        for (int i = 0; i < 2; i++) zsyn[i].zero();
        synDomain = DOM_NONE;
    }
    /* ... and for one instance (its codec was re-initialised: ambi_dec_initCodec ends with afSTFT_clearBuffers) */
    void clear_instance(int i)
    {
        const size_t na = (size_t)nSH * SAF_ANA_HIST * SAF_HOP, ns = (size_t)(bin ? 2 : nLS) * SAF_SYN_HIST * 256, nz = (size_t)nSH * SAF_SYN_HIST * 256;
        HIP_CHECK(hipMemsetAsync(st.ana[st.anaPar].p + i * na, 0, na * sizeof(float), stream()));
        HIP_CHECK(hipMemsetAsync(st.syn[st.synPar].p + i * ns, 0, ns * sizeof(float), stream()));
        if (zsyn[0].p)
            for (int d = 0; d < 2; d++) HIP_CHECK(hipMemsetAsync(zsyn[zsynPar].p + ((size_t)d * nInst + i) * nz, 0, nz * sizeof(float), stream()));
    }

    /* Do the two decoder slots of any instance hold different dense matrices?  Then the equaliser kernel has to emit two
     * signals per channel (one per dense matrix: 1 forward, 2 inverse transforms, 40 KB of LDS per workgroup) and the GEMM has
     * two terms: 3.86 M frames/s against 3.69 M on the transform path (same box, profiles/r02_*) since the kernel's tables moved
     * into the slot pads (four workgroups per CU instead of three). */
    bool two_dense_matrices()
    {
        bool two = false;
        for (int i = 0; i < nInst; i++) {
            AmbiDec* p = inst[i];
            if (eqTwoEpoch[i] != p->codecEpoch) {
                const std::vector<float>& M0 = p->M_dec[0][p->masterOrder - 1];
                const std::vector<float>& M1 = p->M_dec[1][p->masterOrder - 1];
                eqTwo[i] = !(M0.size() == M1.size() && memcmp(M0.data(), M1.data(), M0.size() * sizeof(float)) == 0);
                eqTwoEpoch[i] = p->codecEpoch;
            }
            two = two || eqTwo[i];
        }
        return two;
    }

    /* tables of the equaliser path for the instances whose parameters changed (called after refresh()) */
    void refresh_eq(int mode)
    {
        ensure_eq_buffers(two_dense_matrices() ? 2 : 1);
        const bool force = mode == 2;
        if (force != mode2Shadow) { std::fill(eqDirty.begin(), eqDirty.end(), 1); mode2Shadow = force; }
        bool began = false;
        for (int i = 0; i < nInst; i++) {
            if (!eqDirty[i]) continue;
            AmbiDec* p = inst[i];
            const Shadow& s = shadow[i];
            if (!began) { begin_staging(); began = true; }
            /* w_{d,n}[k] = M_norm_{d,n} * (max-rE ? a_n[k] : 1) for k < (n+1)^2 (ambi_dec.c:524-539) */
            float wtab[NUM_DECODERS][SAF_MAX_ORDER][SAF_MAXCH];
            memset(wtab, 0, sizeof(wtab));
            for (int d = 0; d < NUM_DECODERS; d++)
                for (int n = 1; n <= p->masterOrder; n++) {
                    std::vector<float> a_n; maxre_weights(n, a_n);
                    const float msc = p->M_norm[d][n - 1][s.eq[d] == AMPLITUDE_PRESERVING ? 0 : 1];
                    for (int k = 0; k < ORDER2NSH(n); k++) wtab[d][n - 1][k] = (s.rE[d] ? a_n[k] : 1.0f) * msc;
                }
            const size_t nG = (size_t)eqD * SAF_MAXCH * 136;
            float* G = stageG.p + (size_t)i * nG; int* U = stageU.p + (size_t)i * SAF_MAXCH;
            memset(G, 0, sizeof(float) * nG);
            for (int ch = 0; ch < SAF_MAXCH; ch++) {
                bool uniform = true;
                for (int band = 0; band < SAF_NBANDS; band++) {
                    const int mi = s.b2m[band], d = mi / SAF_MAX_ORDER, n = mi % SAF_MAX_ORDER;
                    const float wv = wtab[d][n][ch];
                    if (eqD == 1) G[(size_t)ch * 136 + band] = wv;
                    else G[((size_t)d * SAF_MAXCH + ch) * 136 + band] = wv;
                }
                for (int d = 0; d < eqD; d++)
                    for (int band = 1; band < SAF_NBANDS; band++)
                        uniform = uniform && G[((size_t)d * SAF_MAXCH + ch) * 136 + band] == G[((size_t)d * SAF_MAXCH + ch) * 136];
                U[ch] = uniform && !force ? 1 : 0;
            }
            HIP_CHECK(hipMemcpyAsync(eqGains.p + (size_t)i * nG, G, sizeof(float) * nG, hipMemcpyHostToDevice, stream()));
            HIP_CHECK(hipMemcpyAsync(eqUni.p + (size_t)i * SAF_MAXCH, U, sizeof(int) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
            /* the dense decoders: the order-N matrices without max-rE / normalisation (ambi_dec.c:283-288), zero-padded */
            std::vector<float> A(64 * 64);
            const int nSHo = ORDER2NSH(p->masterOrder);
            float* Ms = stageM.p + (size_t)i * 2 * 64 * 64;
            for (int d = 0; d < NUM_DECODERS; d++) {
                std::fill(A.begin(), A.end(), 0.0f);
                const std::vector<float>& M = p->M_dec[d][p->masterOrder - 1];
                for (int l = 0; l < p->nLoudpkrs; l++) for (int k = 0; k < nSHo; k++) A[l * 64 + k] = M[(size_t)l * nSHo + k];
                pack_A(A.data(), Ms + (size_t)d * 64 * 64);
            }
            HIP_CHECK(hipMemcpyAsync(Mfrag.p + (size_t)i * 2 * 64 * 64, Ms, sizeof(float) * 2 * 64 * 64, hipMemcpyHostToDevice, stream()));
            eqDirty[i] = 0;
        }
        if (began) stagePending = true;
    }

    /* the 14 effective matrices M_norm * (maxrE ? M_dec_maxrE : M_dec), zero-padded 64 x 64 (ambi_dec.c:524-539), of the
     * instances whose codec or weighting changed: only the transform path and the binaural fold read them */
    void upload_afrag()
    {
        for (int i = 0; i < nInst; i++) {
            if (!afragDirty[i]) continue;
            AmbiDec* p = inst[i];
            const Shadow& s = shadow[i];
            HIP_CHECK(hipStreamSynchronize(stream()));       /* one staging block for all instances: rare path */
            stagePending = false;
            std::vector<float> A(64 * 64);
            for (int d = 0; d < NUM_DECODERS; d++)
                for (int n = 1; n <= SAF_MAX_ORDER; n++) {
                    std::fill(A.begin(), A.end(), 0.0f);
                    if (n <= p->masterOrder) {
                        const int nSHo = ORDER2NSH(n);
                        const std::vector<float>& M = s.rE[d] ? p->M_dec_maxrE[d][n - 1] : p->M_dec[d][n - 1];
                        const float sc = p->M_norm[d][n - 1][s.eq[d] == AMPLITUDE_PRESERVING ? 0 : 1];
                        for (int l = 0; l < p->nLoudpkrs; l++)
                            for (int k = 0; k < nSHo; k++) A[l * 64 + k] = M[(size_t)l * nSHo + k] * sc;
                    }
                    pack_A(A.data(), stageA.p + (size_t)(d * SAF_MAX_ORDER + n - 1) * 64 * 64);
                    if (bin) memcpy(stageArow.p + (size_t)(d * SAF_MAX_ORDER + n - 1) * 64 * 64, A.data(), sizeof(float) * 64 * 64);
                }
            HIP_CHECK(hipMemcpyAsync(Afrag.p + (size_t)i * NMAT * 64 * 64, stageA.p, sizeof(float) * NMAT * 64 * 64, hipMemcpyHostToDevice, stream()));
            if (bin) { HIP_CHECK(hipMemcpyAsync(Arow.p + (size_t)i * NMAT * 64 * 64, stageArow.p, sizeof(float) * NMAT * 64 * 64, hipMemcpyHostToDevice, stream())); foldDirty = true; }
            HIP_CHECK(hipStreamSynchronize(stream()));
            afragDirty[i] = 0;
        }
    }

    /* push per-instance tables whose inputs changed since the previous call (parameters are
     * snapshotted at the start of a block like ambi_dec.c:479-488) */
    void refresh()
    {
        bool began = false;
        auto stage = [&]() { if (!began) { begin_staging(); began = true; } };
        for (int i = 0; i < nInst; i++) {
            AmbiDec* p = inst[i];
            Shadow& s = shadow[i];
            const int rE[2] = { p->rE_WEIGHT[0], p->rE_WEIGHT[1] }, eq[2] = { p->diffEQmode[0], p->diffEQmode[1] };
            if (s.epoch != p->codecEpoch || s.rE[0] != rE[0] || s.rE[1] != rE[1] || s.eq[0] != eq[0] || s.eq[1] != eq[1]) {
                if (s.epoch != ~0ull && s.epoch != p->codecEpoch) clear_instance(i);      /* re-initialised codec: ambi_dec_initCodec clears the filterbank (ambi_dec.c:218,225) */
                s.epoch = p->codecEpoch; s.rE[0] = rE[0]; s.rE[1] = rE[1]; s.eq[0] = eq[0]; s.eq[1] = eq[1];
                s.b2mValid = false; s.norm = -1; eqDirty[i] = 1; afragDirty[i] = 1;
                if (bin) foldDirty = true;
            }
            int b2m[SAF_NBANDS];
            for (int band = 0; band < SAF_NBANDS; band++) {
                int ob = p->orderPerBand[band] < p->masterOrder ? p->orderPerBand[band] : p->masterOrder;
                if (ob < 1) ob = 1;
                const int decIdx = p->freqVector[band] < p->transitionFreq ? 0 : 1;      /* ambi_dec.c:519-523 */
                b2m[band] = decIdx * SAF_MAX_ORDER + ob - 1;
            }
            int* slotI = stageI.p + (size_t)i * (SAF_NBANDS + SAF_MAXCH);
            if (!s.b2mValid || memcmp(b2m, s.b2m, sizeof(b2m)) != 0) {
                stage();
                memcpy(slotI, b2m, sizeof(b2m));
                HIP_CHECK(hipMemcpyAsync(band2mat.p + (size_t)i * SAF_NBANDS, slotI, sizeof(b2m), hipMemcpyHostToDevice, stream()));
                memcpy(s.b2m, b2m, sizeof(b2m)); s.b2mValid = true; foldDirty = true; eqDirty[i] = 1;
            }
            if (s.norm != (int)p->norm || s.chOrd != (int)p->chOrdering) {
                /* input conventions -> ACN/N3D (ambi_dec.c:500-511, saf_hoa.c:40-116) as a gather map + row scale */
                stage();
                int* map = slotI + SAF_NBANDS; float* sc = stageS.p + (size_t)i * SAF_MAXCH;
                for (int ch = 0; ch < SAF_MAXCH; ch++) { map[ch] = ch; sc[ch] = 1.0f; }
                if (p->chOrdering == CH_FUMA) {
                    /* FuMa WXYZ -> ACN WYZX, first order only; higher channels are zeroed (saf_hoa.c:58-69) */
                    map[1] = 2; map[2] = 3; map[3] = 1;
                    for (int ch = 4; ch < SAF_MAXCH; ch++) map[ch] = -1;
                }
                if (p->norm == NORM_SN3D) {
                    for (int n = 0; n <= p->masterOrder; n++)
                        for (int ch = n * n; ch < ORDER2NSH(n); ch++) sc[ch] = sqrtf(2.0f * (float)n + 1.0f);
                } else if (p->norm == NORM_FUMA) {
                    sc[0] = sqrtf(2.0f);
                    for (int ch = 1; ch < 4; ch++) sc[ch] = sqrtf(3.0f);
                }
                HIP_CHECK(hipMemcpyAsync(chMap.p + (size_t)i * SAF_MAXCH, map, sizeof(int) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
                HIP_CHECK(hipMemcpyAsync(chScale.p + (size_t)i * SAF_MAXCH, sc, sizeof(float) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
                s.norm = (int)p->norm; s.chOrd = (int)p->chOrdering;
            }
        }
        if (began) stagePending = true;
    }

    /* interpolated HRTFs of the loudspeakers whose direction changed (ambi_dec.c:549-553, ambi_dec_interpHRTFs =
     * the magnitude + ITD mode of the binauraliser's interpolation kernel) */
    void refresh_hrtfs()
    {
        bool any = false;
        for (int i = 0; i < nInst && !any; i++) for (int ch = 0; ch < nLS; ch++) any = any || inst[i]->recalc_hrtf_interpFLAG[ch];
        if (!any) return;
        foldDirty = true;
        HIP_CHECK(hipStreamSynchronize(stream()));
        for (int i = 0; i < nInst; i++)
            for (int ch = 0; ch < SAF_MAXCH; ch++) {
                stageD.p[((size_t)i * SAF_MAXCH + ch) * 2] = inst[i]->loudpkrs_dirs_deg[ch][0];
                stageD.p[((size_t)i * SAF_MAXCH + ch) * 2 + 1] = inst[i]->loudpkrs_dirs_deg[ch][1];
                stageR.p[(size_t)i * SAF_MAXCH + ch] = ch < nLS ? inst[i]->recalc_hrtf_interpFLAG[ch] : 0;
                if (ch < nLS) inst[i]->recalc_hrtf_interpFLAG[ch] = 0;
            }
        memcpy(stageD.p + (size_t)nInst * SAF_MAXCH * 2, inst[0]->freqVector, sizeof(float) * SAF_NBANDS);       /* ambi_dec_internal.c:106-108 reads pData->freqVector */
        HIP_CHECK(hipMemcpyAsync(lsDirs.p, stageD.p, sizeof(float) * (size_t)nInst * SAF_MAXCH * 2, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(freq.p, stageD.p + (size_t)nInst * SAF_MAXCH * 2, sizeof(float) * SAF_NBANDS, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(lsRecalc.p, stageR.p, sizeof(int) * (size_t)nInst * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
        HrtfInterpLaunch l{};
        l.srcDirs = lsDirs.p; l.recalc = lsRecalc.p; l.gtComp = hrtf->d_gtComp.p; l.gtIdx = hrtf->d_gtIdx.p;
        l.hrtf_fb = hrtf->d_hrtf_fb.p; l.hrtf_mag = hrtf->d_mag.p; l.itds = hrtf->d_itds.p; l.freq = freq.p;
        l.hrtf_interp = hrtfInterp.p; l.nSrc = nLS; l.N = hrtf->N; l.mode = 2 /* INTERP_TRI_PS */;
        l.aziRes = hrtf->vbapTableRes[0]; l.elevRes = hrtf->vbapTableRes[1];
        l.nInst = nInst; l.srcStride = SAF_MAXCH;
        launch_hrtf_interp(l);
    }

    void process(const float* d_in, long long in_inst, long long in_frame, long long in_ch, int nChPresent,
                 float* d_out, long long out_inst, long long out_frame, long long out_ch, int nFrames)
    {
        if (nFrames <= 0) return;
        if (nFrames > maxFrames) SAF_FATAL("ambi_dec batch: nFrames %d exceeds the maxFramesPerCall %d given at creation", nFrames, maxFrames);
        refresh();
        if (bin) {
            upload_afrag();
            refresh_hrtfs();
            if (foldDirty) {
                BinFoldLaunch f{};
                f.h = hrtfInterp.p; f.A = Arow.p; f.band2mat = band2mat.p; f.HM = HM.p; f.nInst = nInst; f.nMat = NMAT; f.nLS = nLS;
                launch_binaural_fold(f);
                foldDirty = false;
            }
        }
        const int H = nFrames * T;
        const int mode = g_ambi_dec_time_domain;
        /* (an output block at an odd offset stays on this path too: the time-domain GEMM then writes it with 4-byte stores —
         * a pipeline only leaves the equaliser path for an explicit mode 0, a two-decoder pipeline with such an output, or
         * binauralised output) */
        const bool outOdd = ((out_inst | out_frame | out_ch) & 3) != 0 || (((uintptr_t)d_out) & 15) != 0;
        bool eq = !bin && mode != 0 && synDomain != DOM_LS && !(outOdd && two_dense_matrices());
        if (eq) {
            refresh_eq(mode);
            const long long zCh = (long long)Hmax * SAF_HOP, zInst = (long long)SAF_MAXCH * zCh, zD = (long long)nInst * zInst;
            const long long synD = (long long)nInst * nSH * SAF_SYN_HIST * 256;
            EqLaunch q{};
            q.in = d_in; q.in_inst = in_inst; q.in_ch = in_ch; q.in_frame = in_frame; q.hopsPerFrame = T; q.nChIn = nChPresent;
            q.hist_rd = st.ana[st.anaPar].p; q.hist_wr = st.ana[st.anaPar ^ 1].p;
            q.ch_scale = chScale.p; q.ch_map = chMap.p;
            q.gains = eqGains.p; q.uniform = eqUni.p; q.D = eqD;
            q.z = zbuf.p; q.z_d = zD; q.z_inst = zInst; q.z_ch = zCh;
            q.syn_rd = zsyn[zsynPar].p; q.syn_wr = zsyn[zsynPar ^ 1].p; q.syn_d = synD;
            q.nCh = nSH; q.nInst = nInst; q.H = H;
            /* out = M_0 z_0 (+ M_1 z_1): "band" = block, columns = the F samples of the block */
            BandGemmLaunch gn{};
            gn.X = zbuf.p; gn.x_inst = zInst; gn.x_band = F; gn.x_row = zCh; gn.nTerms = eqD; gn.x_term = zD;
            gn.Y = d_out; gn.y_inst = out_inst; gn.y_band = out_frame; gn.y_row = out_ch; gn.nRowsY = nLS;
            gn.Afrag = Mfrag.p; gn.a_inst = 2 * 64 * 64; gn.band2mat = zerosI.p;
            gn.nBands = nFrames; gn.nInst = nInst; gn.N = F; gn.nRowsX = nSH;
            /* Optional (saf_hip_ambi_dec_setOverlap): the decode runs BESIDE the equaliser kernel (persistent MFMA workgroups on the
             * side stream that take the instances as their equaliser workgroups finish) instead of after it — the reference
             * decodes each frame right after transforming it (ambi_dec.c:514-566).  The GEMM launch that follows is then the
             * fix-up: its workgroups leave at once unless a decode workgroup gave up waiting (see launch_dec_stream). */
            DecStreamLaunch ds{};
            ds.z = zbuf.p; ds.z_d = zD; ds.z_inst = zInst; ds.z_ch = zCh; ds.D = eqD; ds.nCh = nSH; ds.nInst = nInst;
            ds.Y = d_out; ds.y_inst = out_inst; ds.y_frame = out_frame; ds.y_row = out_ch;
            ds.Mfrag = Mfrag.p; ds.m_inst = 2 * 64 * 64; ds.nRowsY = nLS; ds.F = F; ds.nFrames = nFrames;
            const bool overlap = (g_ambi_dec_overlap == 1 || g_ambi_dec_overlap == 2) && (g_ambi_dec_overlap == 2 || ((long long)nInst * nSH >= 3072 && H >= 32)) &&
                                 !on_private_stream() && dec_stream_supported(ds);      /* (one side stream per process: not from a handle's own stream) */
            /* Small launches (the one-block host-pointer call is bound by launches, not by the chip): equaliser and decode in ONE
             * launch — the decode workgroups ride behind the channel workgroups and wait on the instance's counter. */
            static const int fuseSmall = []() { const char* v = getenv("SAF_HIP_AMBI_DEC_ONE_LAUNCH"); return v ? atoi(v) : 1; }();
            /* Optional (setOverlap(3)): the decode INSIDE the equaliser launch.  The 64 channel workgroups of an instance hand
             * their z to each other through zbuf (write-through stores, per-sub-chunk counters) and each decodes its 32 columns
             * of every sub-chunk three iterations later, while the data is still in the memory-side cache.  A wave that gives
             * up waiting sets errPin and the two guarded launches behind recompute the call the ordinary way (the histories are
             * still unflipped). */
            if (g_ambi_dec_overlap == 3 && eqD == 1 && nSH == SAF_MAXCH && nLS == 64 && H % 16 == 0 &&
                out_frame < (1ll << 31) && out_ch < (1ll << 29) && out_frame >= 0 && out_ch >= 0) {
                const int nSub = H / 16;
                if (coNSub < nSub) { coCnt.alloc((size_t)nInst * nSub); coNSub = nSub; }
                errPin.ensure(1);
                HIP_CHECK(hipMemsetAsync(coCnt.p, 0, (size_t)nInst * coNSub * sizeof(unsigned), stream()));
                EqCoop c{};
                c.cnt = coCnt.p; c.target = 2u * SAF_MAXCH + g_coop_target_bias; c.nSub = coNSub;
                c.Y = d_out; c.y_inst = out_inst; c.y_frame = (int)out_frame; c.y_row = (int)out_ch; c.nRowsY = nLS; c.F = F; c.T = T;
                c.Mfrag = Mfrag.p; c.m_inst = 2 * 64 * 64; c.err = errPin.p; c.giveUps = eqErr.p ? eqErr.p + 1 : nullptr;
#ifdef EQ_COOP_CHECK
                if (!coDbg.p) coDbg.alloc(8);
                c.dbg = coDbg.p; c.ringBytes = (long long)zbuf.n * 4; c.cntBytes = (long long)coCnt.n * 4; c.mBytes = (long long)Mfrag.n * 4;
                c.yBytes = ((long long)(nInst - 1) * out_inst + (long long)(nFrames - 1) * out_frame + (long long)(nLS - 1) * out_ch + F) * 4;
#endif
                if (launch_eq_coop(q, c)) {
                    EqLaunch q2 = q; q2.runFlag = errPin.p;
                    launch_eq(q2);                                      /* both leave at once unless a workgroup gave up */
                    gn.runFlag = errPin.p;
                    launch_band_gemm(gn);
                    st.anaPar ^= 1; zsynPar ^= 1;
                    synDomain = DOM_SH; lastPath = 1; lastOverlap = 3;
                    return;
                }
            }
            if (fuseSmall && !overlap) {
                EqDecodeTail t{};
                t.Y = d_out; t.y_inst = out_inst; t.y_frame = out_frame; t.y_row = out_ch; t.Mfrag = Mfrag.p; t.m_inst = 2 * 64 * 64;
                t.nRowsY = nLS; t.F = F; t.nFrames = nFrames; t.err = errPin.p;
                const int units = nFrames * (F / 128);
                t.G = (units + 7) / 8;                                  /* at most 8 decode workgroups per instance */
                if (launch_eq_decode(q, t, eqDone.p, eqDoneBase + (unsigned)nSH)) {
                    eqDoneBase += (unsigned)nSH;
                    st.anaPar ^= 1; zsynPar ^= 1;
                    gn.runFlag = errPin.p;
                    if (deferFixup) { fixGemm = gn; fixPending = true; }
                    else launch_band_gemm(gn);                          /* leaves at once unless a decode workgroup gave up */
                    synDomain = DOM_SH; lastPath = 1; lastOverlap = 0;
                    return;
                }
            }
            if (overlap) {
                ds.done = eqDone.p; ds.target = eqDoneBase + (unsigned)nSH; ds.err = eqErr.p;
                eqDoneBase += (unsigned)nSH;
                side_fork();
                launch_dec_stream(ds, side_stream());
                launch_eq(q, eqDone.p);
                side_join();
                gn.runFlag = eqErr.p;
            } else
                launch_eq(q);
            st.anaPar ^= 1; zsynPar ^= 1;
            launch_band_gemm(gn);
            synDomain = DOM_SH; lastPath = 1; lastOverlap = overlap ? 1 : 0;
            return;
        }
        ensure_transform_buffers();
        upload_afrag();
        if (synDomain == DOM_SH) {
            /* the overlap-add history of the equaliser path, taken to the loudspeaker domain: st.syn = sum_d M_d zsyn_d */
            const long long fr = (long long)SAF_SYN_HIST * 256;
            BandGemmLaunch gc{};
            gc.X = zsyn[zsynPar].p; gc.x_inst = (long long)nSH * fr; gc.x_band = 0; gc.x_row = fr; gc.nTerms = eqD; gc.x_term = (long long)nInst * nSH * fr;
            gc.Y = st.syn[st.synPar].p; gc.y_inst = (long long)nLS * fr; gc.y_band = 0; gc.y_row = fr; gc.nRowsY = nLS;
            gc.Afrag = Mfrag.p; gc.a_inst = 2 * 64 * 64; gc.band2mat = zerosI.p;
            gc.nBands = 1; gc.nInst = nInst; gc.N = (int)fr; gc.nRowsX = nSH;
            launch_band_gemm(gc);
            for (int i = 0; i < 2; i++) zsyn[i].zero();
        }
        if (!bin) synDomain = DOM_LS;
        lastPath = 0;
        AnaLaunch a{};
        a.in = d_in; a.in_inst = in_inst; a.in_ch = in_ch; a.in_frame = in_frame; a.hopsPerFrame = T; a.nChIn = nChPresent;
        a.hist_rd = st.ana[st.anaPar].p; a.hist_wr = st.ana[st.anaPar ^ 1].p;
        a.out = X.p; a.out_inst = (long long)SAF_NBANDS * SAF_MAXCH * Hmax; a.out_band = (long long)SAF_MAXCH * Hmax; a.out_ch = Hmax;
        a.ch_scale = chScale.p; a.ch_map = chMap.p;
        a.nCh = nSH; a.nInst = nInst; a.H = H; a.lowDelay = 0; a.hybrid = 1;
        a.tab_stride = SAF_MAXCH;
        launch_analysis(a);
        st.anaPar ^= 1;

        if (!bin) {
            BandGemmLaunch g{};
            g.X = (const float*)X.p; g.x_inst = 2 * a.out_inst; g.x_band = 2 * a.out_band; g.x_row = 2 * a.out_ch;
            g.Y = (float*)Y.p; g.y_inst = g.x_inst; g.y_band = g.x_band; g.y_row = g.x_row;
            g.Afrag = Afrag.p; g.a_inst = (long long)NMAT * 64 * 64; g.band2mat = band2mat.p;
            g.nBands = SAF_NBANDS; g.nInst = nInst; g.N = 2 * H;
            launch_band_gemm(g);
        }

        SynLaunch s{};
        s.in = Y.p; s.in_inst = a.out_inst; s.in_band = a.out_band; s.in_ch = a.out_ch;
        s.nCh = nLS;
        if (bin) {
            /* decode + binauralise (ambi_dec.c:518-563) as ONE band MAC: ears = (H_b M_b) x_b / sqrt(nLS).  The reference
             * forms the loudspeaker spectra M_b x_b first and then applies the HRTFs; both steps are linear, so the
             * 2 x nSH product matrix gives the same ears without materialising 64 loudspeaker channels. */
            BinMacLaunch m{};
            m.X = X.p; m.x_inst = a.out_inst; m.x_band = a.out_band; m.x_ch = a.out_ch;
            m.h = HM.p; m.h_inst = (long long)SAF_MAXCH * SAF_NBANDS * 2;
            m.Y = Z.p; m.y_inst = (long long)SAF_NBANDS * 2 * Hmax; m.y_band = (long long)2 * Hmax; m.y_ch = Hmax;
            m.nSrc = nSH; m.H = H; m.scale = 1.0f / sqrtf((float)nLS); m.nInst = nInst;
            launch_binaural_mac(m);
            s.in = Z.p; s.in_inst = m.y_inst; s.in_band = m.y_band; s.in_ch = m.y_ch; s.nCh = 2;
        }
        s.out = d_out; s.out_inst = out_inst; s.out_ch = out_ch; s.out_frame = out_frame; s.hopsPerFrame = T;
        s.hist_rd = st.syn[st.synPar].p; s.hist_wr = st.syn[st.synPar ^ 1].p;
        s.nInst = nInst; s.H = H; s.lowDelay = 0; s.hybrid = 1;
        launch_synthesis(s);
        st.synPar ^= 1;
    }
};

/* -------------------------------------------------------------------------- */

static void set_codec_status(AmbiDec* p, CODEC_STATUS s)     /* ambi_dec_internal.c:48-57 */
{
    if (s == CODEC_STATUS_NOT_INITIALISED)
        while (p->codecStatus == CODEC_STATUS_INITIALISING) sleep_ms(10);
    p->codecStatus = s;
}

}  // namespace saf

using namespace saf;

extern "C" {

void saf_hip_ambi_dec_setTimeDomainPath(int mode) { g_ambi_dec_time_domain = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
int saf_hip_ambi_dec_getTimeDomainPath(void) { return g_ambi_dec_time_domain; }

void saf_hip_ambi_dec_setFrameSize(int frameSize)
{
    if (frameSize <= 0 || frameSize % SAF_HOP != 0) SAF_FATAL("ambi_dec frame size must be a positive multiple of 128");
    g_ambi_dec_frame_size = frameSize;
}

void ambi_dec_create(void** const phAmbi)
{
    AmbiDec* p = new AmbiDec();
    *phAmbi = p;
    p->F = g_ambi_dec_frame_size; p->T = p->F / SAF_HOP;
    load_loudspeaker_preset(LOUDSPEAKER_ARRAY_PRESET_T_DESIGN_24, p->loudpkrs_dirs_deg, &p->new_nLoudpkrs, &p->loudpkrs_nDims);
    p->masterOrder = p->new_masterOrder = 1;
    for (int b = 0; b < SAF_NBANDS; b++) p->orderPerBand[b] = 1;
    p->useDefaultHRIRsFLAG = 1; p->enableHRIRsPreProc = 1;
    p->nLoudpkrs = p->new_nLoudpkrs;
    p->chOrdering = CH_ACN; p->norm = NORM_SN3D;
    p->dec_method[0] = p->dec_method[1] = DECODING_METHOD_ALLRAD;
    p->rE_WEIGHT[0] = p->rE_WEIGHT[1] = 1;
    p->diffEQmode[0] = p->diffEQmode[1] = ENERGY_PRESERVING;
    p->transitionFreq = 800.0f;
    p->progressBar0_1 = 0.0f; p->progressBarText[0] = 0;
    p->codecStatus = CODEC_STATUS_NOT_INITIALISED;
    p->binauraliseLS = p->new_binauraliseLS = 0;
    p->procStatus = PROC_STATUS_NOT_ONGOING;
    p->reinit_hrtfsFLAG = 1;
    for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc_hrtf_interpFLAG[ch] = 1;
    memset(p->freqVector, 0, sizeof(p->freqVector));
    memset(p->M_norm, 0, sizeof(p->M_norm));
}

void ambi_dec_destroy(void** const phAmbi)
{
    AmbiDec* p = (AmbiDec*)*phAmbi;
    if (!p) return;
    while (p->codecStatus == CODEC_STATUS_INITIALISING || p->procStatus == PROC_STATUS_ONGOING) sleep_ms(10);
    {
        StreamScope onOwn(p->own);
        if (p->pipe) { HIP_CHECK(hipStreamSynchronize(stream())); delete p->pipe; }
    }
    if (p->own) HIP_CHECK(hipStreamDestroy(p->own));
    delete p;
    *phAmbi = nullptr;
}

void ambi_dec_init(void* const hAmbi, int sampleRate)
{
    AmbiDec* p = (AmbiDec*)hAmbi;
    p->fs = sampleRate;
    /* before the first initCodec there is no filterbank yet: the NULL-handle table branch (ambi_dec.c:178, afSTFTlib.c:554-563) */
    if (!p->haveSTFT) afSTFT_getCentreFreqs(nullptr, (float)sampleRate, SAF_NBANDS, p->freqVector);
    else {   /* valid-handle branch of afSTFT_getCentreFreqs (afSTFTlib.c:565-587), hop 128 hybrid */
        static const float w[9] = { 1.0f, 0.7501f, 1.2499f, 0.8751f, 1.1249f, 0.9167f, 1.0833f, 0.9375f, 1.0625f };
        static const int bin[9] = { 0, 1, 1, 2, 2, 3, 3, 4, 4 };
        for (int i = 0; i < 9; i++) p->freqVector[i] = w[i] * ((float)bin[i] * (float)sampleRate / 256.0f);
        for (int i = 9, j = 5; i < SAF_NBANDS; i++, j++) p->freqVector[i] = (float)j * (float)sampleRate / 256.0f;
    }
}

void ambi_dec_initCodec(void* const hAmbi)
{
    AmbiDec* p = (AmbiDec*)hAmbi;
    if (p->codecStatus != CODEC_STATUS_NOT_INITIALISED) return;          /* ambi_dec.c:196-197 */
    while (p->procStatus == PROC_STATUS_ONGOING) { p->codecStatus = CODEC_STATUS_INITIALISING; sleep_ms(10); }
    ensure_device();
    p->codecStatus = CODEC_STATUS_INITIALISING;
    strcpy(p->progressBarText, "Initialising");
    p->progressBar0_1 = 0.0f;

    const int masterOrder = p->new_masterOrder;
    const int max_nSH = ORDER2NSH(masterOrder);
    int nLS = p->new_nLoudpkrs;
    /* (re)create the filterbank state: channel change + clearBuffers == fresh zero state (ambi_dec.c:213-226) */
    if (p->pipe) { StreamScope onOwn(p->own); HIP_CHECK(hipStreamSynchronize(stream())); delete p->pipe; p->pipe = nullptr; }
    p->haveSTFT = true;
    p->binauraliseLS = p->new_binauraliseLS;
    p->nLoudpkrs = nLS;

    strcpy(p->progressBarText, "Computing decoder");
    p->progressBar0_1 = 0.2f;
    float sum_elev = 0.0f;
    for (int ch = 0; ch < nLS; ch++) sum_elev += fabsf(p->loudpkrs_dirs_deg[ch][1]);
    p->loudpkrs_nDims = (((sum_elev < 5.0f) && (sum_elev > -5.0f)) || (nLS < 4)) ? 2 : 3;
    const bool virt = p->loudpkrs_nDims == 2 && (p->dec_method[0] == DECODING_METHOD_ALLRAD || p->dec_method[1] == DECODING_METHOD_ALLRAD);
    if (virt) {      /* virtual loudspeakers above/below so a 2-D layout triangulates (ambi_dec.c:241-249) */
        if (nLS > SAF_MAXCH - 2) SAF_FATAL("ambi_dec: a 2-D layout needs two spare loudspeaker slots for AllRAD");
        p->loudpkrs_dirs_deg[nLS][0] = 0.0f; p->loudpkrs_dirs_deg[nLS][1] = -90.0f;
        p->loudpkrs_dirs_deg[nLS + 1][0] = 0.0f; p->loudpkrs_dirs_deg[nLS + 1][1] = 90.0f;
        nLS += 2;
    }

    /* SH of the 480-point t-design, evaluated once per order on the GPU (ambi_dec.c:304-320 evaluates getSHreal per direction) */
    const int nGrid = 480;
    const float* grid = table_required("Tdesign_degree_30_dirs_deg", nGrid * 2);
    std::vector<float> grid_rad((size_t)nGrid * 2);
    for (int ng = 0; ng < nGrid; ng++) {
        grid_rad[ng * 2] = grid[ng * 2] * SAF_PI / 180.0f;
        grid_rad[ng * 2 + 1] = SAF_PI / 2.0f - grid[ng * 2 + 1] * SAF_PI / 180.0f;
    }
    std::vector<float> Ygrid((size_t)max_nSH * nGrid);
    sh_eval_host(0, masterOrder, grid_rad.data(), nGrid, Ygrid.data());

    static const int method_map[5] = { LOUDSPEAKER_DECODER_SAD, LOUDSPEAKER_DECODER_SAD, LOUDSPEAKER_DECODER_MMD, LOUDSPEAKER_DECODER_EPAD, LOUDSPEAKER_DECODER_ALLRAD };
    std::vector<float> M_tmp((size_t)nLS * max_nSH), g(nLS);
    for (int d = 0; d < NUM_DECODERS; d++) {
        const int dm = p->dec_method[d];
        decoder_matrix(&p->loudpkrs_dirs_deg[0][0], nLS, (dm >= 1 && dm <= 4) ? method_map[dm] : LOUDSPEAKER_DECODER_SAD, masterOrder, 0, M_tmp.data());
        for (int n = 1; n <= masterOrder; n++) {
            const int nSHo = ORDER2NSH(n);
            std::vector<float>& M = p->M_dec[d][n - 1];
            std::vector<float>& Mr = p->M_dec_maxrE[d][n - 1];
            M.assign((size_t)nLS * nSHo, 0.0f); Mr.assign((size_t)nLS * nSHo, 0.0f);
            for (int i = 0; i < nLS; i++) for (int j = 0; j < nSHo; j++) M[(size_t)i * nSHo + j] = M_tmp[(size_t)i * max_nSH + j];
            std::vector<float> a_n; maxre_weights(n, a_n);
            for (int i = 0; i < nLS; i++) for (int j = 0; j < nSHo; j++) Mr[(size_t)i * nSHo + j] = M[(size_t)i * nSHo + j] * a_n[j];
            /* omni amplitude / energy of the non-maxrE decoder over the t-design (ambi_dec.c:304-331).
             * The SH of order n are the first (n+1)^2 rows of the order-N evaluation. */
            float a_avg = 0.0f, e_avg = 0.0f;
            for (int ng = 0; ng < nGrid; ng++) {
                float a = 0.0f, e = 0.0f;
                for (int i = 0; i < nLS; i++) {
                    float acc = 0.0f;
                    for (int j = 0; j < nSHo; j++) acc += M[(size_t)i * nSHo + j] * Ygrid[(size_t)j * nGrid + ng];
                    g[i] = acc;
                }
                for (int i = 0; i < nLS; i++) { a += g[i]; e += powf(g[i], 2.0f); }
                a_avg += a; e_avg += e;
            }
            a_avg /= (float)nGrid; e_avg /= (float)nGrid;
            p->M_norm[d][n - 1][0] = 1.0f / (a_avg + 2.23e-6f);
            p->M_norm[d][n - 1][1] = sqrtf(1.0f / (e_avg + 2.23e-6f));
            if (virt) { M.resize((size_t)p->nLoudpkrs * nSHo); Mr.resize((size_t)p->nLoudpkrs * nSHo); }   /* ambi_dec.c:336-341 */
        }
    }
    p->masterOrder = p->new_masterOrder;

    /* Binaural-related initialisations (ambi_dec.c:349-445).  The reference runs them on every re-init whether or not the
     * output is binauralised; its default HRIR set is not part of the checkout, so here they run when binauralised output
     * is requested (and abort with a message when no set has been installed with saf_hip_setDefaultHRIRs). */
    if (p->binauraliseLS && (p->reinit_hrtfsFLAG || !p->hrtf || p->hrtf->hrirEpoch != default_hrirs().epoch)) {
        strcpy(p->progressBarText, "Computing VBAP gain table");
        p->progressBar0_1 = 0.4f;
        p->useDefaultHRIRsFLAG = 1;                         /* "can only load the default HRIR data" (ambi_dec.c:381) */
        p->hrtf = ambi_dec_hrtf_tables(p->freqVector, p->enableHRIRsPreProc);
        p->hrir_fs = p->hrtf->fs;
        for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc_hrtf_interpFLAG[ch] = 1;
        p->reinit_hrtfsFLAG = 0;
    }

    p->codecEpoch++;
    strcpy(p->progressBarText, "Done!");
    p->progressBar0_1 = 1.0f;
    p->codecStatus = CODEC_STATUS_INITIALISED;
}

void ambi_dec_process(void* const hAmbi, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)
{
    AmbiDec* p = (AmbiDec*)hAmbi;
    const int F = p->F;
    if (nSamples == F && p->codecStatus == CODEC_STATUS_INITIALISED) {
        p->procStatus = PROC_STATUS_ONGOING;
        const int nSH = ORDER2NSH(p->masterOrder), nLS = p->nLoudpkrs;
        /* every handle works on a stream of its own: the call is synchronous for its caller (it ends with a stream sync), and
         * handles driven from different host threads (SURVEY 8b "Threading": one call at a time per handle) run side by side
         * instead of queueing on the process-wide stream */
        if (!p->own) p->own = new_stream();
        StreamScope onOwn(p->own);
        if (!p->pipe) {
            p->pipe = new DecPipeline();
            AmbiDec* self = p;
            p->pipe->create(&self, 1, 1);
            p->pipe->deferFixup = true;
            p->h_in.ensure((size_t)SAF_MAXCH * F); p->h_out.ensure((size_t)SAF_MAXCH * F);
            p->d_in.alloc((size_t)SAF_MAXCH * F, false); p->d_out.alloc((size_t)SAF_MAXCH * F, false);
        }
        int i;
        const int nPresent = nSH < nInputs ? nSH : nInputs;
        for (i = 0; i < nPresent; i++) memcpy(p->h_in.p + (size_t)i * F, inputs[i], sizeof(float) * F);
        /* a FuMa gather may read rows up to 3: make them defined */
        for (; i < (nSH < 4 ? 4 : nSH) && i < SAF_MAXCH; i++) memset(p->h_in.p + (size_t)i * F, 0, sizeof(float) * F);
        const int nRows = (nSH < 4 ? 4 : nSH);
        const int nOutCh = p->binauraliseLS ? 2 : nLS;                     /* NUM_EARS or the loudspeakers (ambi_dec.c:570) */
        if (zero_copy_io()) {
            /* one block: the kernels read the pinned input and write the pinned output directly (hipHostMalloc memory is
             * device-accessible): every sample crosses the link once, and the two DMA copies with their latencies drop
             * out of the dependent chain */
            p->pipe->process(p->h_in.p, 0, 0, F, nRows, p->h_out.p, 0, 0, F, 1);
        } else {
            HIP_CHECK(hipMemcpyAsync(p->d_in.p, p->h_in.p, sizeof(float) * (size_t)nRows * F, hipMemcpyHostToDevice, stream()));
            p->pipe->process(p->d_in.p, 0, 0, F, nRows, p->d_out.p, 0, 0, F, 1);
            HIP_CHECK(hipMemcpyAsync(p->h_out.p, p->d_out.p, sizeof(float) * (size_t)nOutCh * F, hipMemcpyDeviceToHost, stream()));
        }
        wait_stream(stream());
        p->pipe->fixup_if_needed();
        int ch;
        for (ch = 0; ch < (nOutCh < nOutputs ? nOutCh : nOutputs); ch++) memcpy(outputs[ch], p->h_out.p + (size_t)ch * F, sizeof(float) * F);
        for (; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    } else
        for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);      /* ambi_dec.c:575-577 */
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}

/* ------------------------------- set functions (ambi_dec.c:585-810) ------------------------------- */
#define PD AmbiDec* p = (AmbiDec*)hAmbi
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void ambi_dec_refreshSettings(void* const hAmbi)
{
    PD;
    for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc_hrtf_interpFLAG[ch] = 1;
    p->reinit_hrtfsFLAG = 1;
    set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
}
void ambi_dec_setMasterDecOrder(void* const hAmbi, int newValue)
{
    PD;
    p->new_masterOrder = clampi(newValue, 1, SAF_MAX_ORDER);
    set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
    if (p->new_masterOrder != 1 && p->chOrdering == CH_FUMA) p->chOrdering = CH_ACN;     /* FuMa is first-order only */
    if (p->new_masterOrder != 1 && p->norm == NORM_FUMA) p->norm = NORM_SN3D;
}
void ambi_dec_setDecOrder(void* const hAmbi, int newValue, int bandIdx) { PD; p->orderPerBand[bandIdx] = clampi(newValue, 1, p->new_masterOrder); }
void ambi_dec_setDecOrderAllBands(void* const hAmbi, int newValue) { PD; for (int b = 0; b < SAF_NBANDS; b++) p->orderPerBand[b] = clampi(newValue, 1, p->new_masterOrder); }
void ambi_dec_setLoudspeakerAzi_deg(void* const hAmbi, int index, float v)
{
    PD;
    if (v > 180.0f) v = -360.0f + v;
    v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    if (p->loudpkrs_dirs_deg[index][0] != v) { p->loudpkrs_dirs_deg[index][0] = v; p->recalc_hrtf_interpFLAG[index] = 1; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
}
void ambi_dec_setLoudspeakerElev_deg(void* const hAmbi, int index, float v)
{
    PD;
    v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    if (p->loudpkrs_dirs_deg[index][1] != v) { p->loudpkrs_dirs_deg[index][1] = v; p->recalc_hrtf_interpFLAG[index] = 1; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
}
void ambi_dec_setNumLoudspeakers(void* const hAmbi, int n)
{
    PD;
    p->new_nLoudpkrs = clampi(n, 4 /* MIN_NUM_LOUDSPEAKERS */, SAF_MAXCH);
    if (p->nLoudpkrs != p->new_nLoudpkrs) {
        for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc_hrtf_interpFLAG[ch] = 1;
        set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
    }
}
void ambi_dec_setBinauraliseLSflag(void* const hAmbi, int newState) { PD; p->new_binauraliseLS = newState; if (p->new_binauraliseLS != p->binauraliseLS) set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
void ambi_dec_setUseDefaultHRIRsflag(void* const hAmbi, int newState) { PD; if (!p->useDefaultHRIRsFLAG && newState) { p->useDefaultHRIRsFLAG = newState; ambi_dec_refreshSettings(hAmbi); } }
void ambi_dec_setSofaFilePath(void* const hAmbi, const char* path) { PD; p->sofa_filepath = path; p->useDefaultHRIRsFLAG = 0; ambi_dec_refreshSettings(hAmbi); }
void ambi_dec_setEnableHRIRsPreProc(void* const hAmbi, int newState) { PD; if (newState != p->enableHRIRsPreProc) { p->enableHRIRsPreProc = newState; ambi_dec_refreshSettings(hAmbi); } }
void ambi_dec_setOutputConfigPreset(void* const hAmbi, int newPresetID)
{
    PD;
    load_loudspeaker_preset(newPresetID, p->loudpkrs_dirs_deg, &p->new_nLoudpkrs, &p->loudpkrs_nDims);
    for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc_hrtf_interpFLAG[ch] = 1;
    set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
}
void ambi_dec_setSourcePreset(void* const hAmbi, int newPresetID)
{
    PD;
    mic_preset_order_per_band(newPresetID, p->masterOrder, p->freqVector, SAF_NBANDS, p->orderPerBand);
}
void ambi_dec_setChOrder(void* const hAmbi, int v) { PD; if ((CH_ORDER)v != CH_FUMA || p->new_masterOrder == 1) p->chOrdering = (CH_ORDER)v; }
void ambi_dec_setNormType(void* const hAmbi, int v) { PD; if ((NORM_TYPES)v != NORM_FUMA || p->new_masterOrder == 1) p->norm = (NORM_TYPES)v; }
void ambi_dec_setDecMethod(void* const hAmbi, int index, int newID) { PD; p->dec_method[index] = newID; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
void ambi_dec_setDecEnableMaxrE(void* const hAmbi, int index, int newID) { PD; p->rE_WEIGHT[index] = newID; }
void ambi_dec_setDecNormType(void* const hAmbi, int index, int newID) { PD; p->diffEQmode[index] = newID; }
void ambi_dec_setTransitionFreq(void* const hAmbi, float v) { PD; p->transitionFreq = v < 500.0f ? 500.0f : (v > 2000.0f ? 2000.0f : v); }   /* ambi_dec.h:99-102 */

/* ------------------------------- get functions (ambi_dec.c:813-988) ------------------------------- */
int ambi_dec_getFrameSize(void) { return g_ambi_dec_frame_size; }
CODEC_STATUS ambi_dec_getCodecStatus(void* const hAmbi) { PD; return p->codecStatus; }
float ambi_dec_getProgressBar0_1(void* const hAmbi) { PD; return p->progressBar0_1; }
void ambi_dec_getProgressBarText(void* const hAmbi, char* text) { PD; memcpy(text, p->progressBarText, PROGRESSBARTEXT_CHAR_LENGTH); }
int ambi_dec_getMasterDecOrder(void* const hAmbi) { PD; return p->new_masterOrder; }
int ambi_dec_getDecOrder(void* const hAmbi, int bandIdx) { PD; return p->orderPerBand[bandIdx]; }
int ambi_dec_getDecOrderAllBands(void* const hAmbi) { PD; return p->orderPerBand[0]; }
void ambi_dec_getDecOrderHandle(void* const hAmbi, float** pX_vector, int** pY_values, int* pNpoints) { PD; *pX_vector = &p->freqVector[0]; *pY_values = &p->orderPerBand[0]; *pNpoints = SAF_NBANDS; }
int ambi_dec_getNumberOfBands(void) { return SAF_NBANDS; }
float ambi_dec_getLoudspeakerAzi_deg(void* const hAmbi, int index) { PD; return p->loudpkrs_dirs_deg[index][0]; }
float ambi_dec_getLoudspeakerElev_deg(void* const hAmbi, int index) { PD; return p->loudpkrs_dirs_deg[index][1]; }
int ambi_dec_getNumLoudspeakers(void* const hAmbi) { PD; return p->new_nLoudpkrs; }
int ambi_dec_getMaxNumLoudspeakers(void) { return SAF_MAXCH; }
int ambi_dec_getNSHrequired(void* const hAmbi) { PD; return ORDER2NSH(p->masterOrder); }
int ambi_dec_getBinauraliseLSflag(void* const hAmbi) { PD; return p->new_binauraliseLS; }
int ambi_dec_getUseDefaultHRIRsflag(void* const hAmbi) { PD; return p->useDefaultHRIRsFLAG; }
char* ambi_dec_getSofaFilePath(void* const hAmbi) { PD; return p->sofa_filepath.empty() ? (char*)"no_file" : (char*)p->sofa_filepath.c_str(); }
int ambi_dec_getEnableHRIRsPreProc(void* const hAmbi) { PD; return p->enableHRIRsPreProc; }
int ambi_dec_getChOrder(void* const hAmbi) { PD; return (int)p->chOrdering; }
int ambi_dec_getNormType(void* const hAmbi) { PD; return (int)p->norm; }
int ambi_dec_getDecMethod(void* const hAmbi, int index) { PD; return p->dec_method[index]; }
int ambi_dec_getDecEnableMaxrE(void* const hAmbi, int index) { PD; return p->rE_WEIGHT[index]; }
int ambi_dec_getDecNormType(void* const hAmbi, int index) { PD; return p->diffEQmode[index]; }
float ambi_dec_getTransitionFreq(void* const hAmbi) { PD; return p->transitionFreq; }
int ambi_dec_getHRIRsamplerate(void* const hAmbi) { PD; return p->hrir_fs; }
int ambi_dec_getDAWsamplerate(void* const hAmbi) { PD; return p->fs; }
int ambi_dec_getProcessingDelay(void) { return 12 * SAF_HOP; }

void saf_hip_ambi_dec_getDecoderMtx(void* const hAmbi, int decIdx, int order, int maxrE, float* out)
{
    PD;
    const std::vector<float>& M = maxrE ? p->M_dec_maxrE[decIdx][order - 1] : p->M_dec[decIdx][order - 1];
    memcpy(out, M.data(), sizeof(float) * M.size());
}
float saf_hip_ambi_dec_getDecoderNorm(void* const hAmbi, int decIdx, int order, int ampOrEnergy) { PD; return p->M_norm[decIdx][order - 1][ampOrEnergy]; }

/* ------------------------------- batched entry point ------------------------------- */
void* saf_hip_ambi_dec_batch_create(void* const* hAmbis, int nInst, int maxFramesPerCall)
{
    if (nInst <= 0 || maxFramesPerCall <= 0) SAF_FATAL("ambi_dec batch: nInst and maxFramesPerCall must be positive");
    ensure_device();
    DecPipeline* b = new DecPipeline();
    b->create((AmbiDec* const*)hAmbis, nInst, maxFramesPerCall);
    return b;
}
void saf_hip_ambi_dec_batch_destroy(void** const phBatch)
{
    if (!phBatch || !*phBatch) return;
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete (DecPipeline*)*phBatch;
    *phBatch = nullptr;
}
void saf_hip_ambi_dec_batch_clear(void* const hBatch) { ((DecPipeline*)hBatch)->clear_state(); }
int saf_hip_ambi_dec_batch_lastPath(void* const hBatch) { return ((DecPipeline*)hBatch)->lastPath; }
int saf_hip_ambi_dec_batch_lastOverlap(void* const hBatch) { return ((DecPipeline*)hBatch)->lastOverlap; }
int saf_hip_ambi_dec_batch_decodeGiveUps(void* const hBatch)
{
    DecPipeline* b = (DecPipeline*)hBatch;
    if (!b->eqErr.p) return 0;
    int v[2] = { 0, 0 };
    HIP_CHECK(hipStreamSynchronize(stream()));
    HIP_CHECK(hipMemcpy(v, b->eqErr.p, sizeof(v), hipMemcpyDeviceToHost));
    return v[1];
}
void saf_hip_ambi_dec_setOverlap(int mode) { g_ambi_dec_overlap = mode < 0 ? 0 : (mode > 3 ? 3 : mode); }
__attribute__((visibility("default"))) long long saf_hip_debug_batch_fetch(void* const hBatch, int which, float* dst, long long n)
{   /* tests: the equaliser output buffer (0) or the cooperative form's ring (1), copied to the host */
    DecPipeline* b = (DecPipeline*)hBatch;
    HIP_CHECK(hipStreamSynchronize(stream()));
    const float* src = which == 0 ? b->zbuf.p : (const float*)b->coDbg.p;
    const long long have = (long long)(which == 0 ? b->zbuf.n : b->coDbg.n * 2);
    if (!src) return 0;
    if (n > have) n = have;
    HIP_CHECK(hipMemcpy(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return n;
}
__attribute__((visibility("default"))) void saf_hip_debug_coop_target_bias(unsigned bias) { g_coop_target_bias = bias; }
int saf_hip_ambi_dec_getOverlap(void) { return g_ambi_dec_overlap; }
int saf_hip_ambi_dec_lastPath(void* const hAmbi) { AmbiDec* p = (AmbiDec*)hAmbi; return p->pipe ? p->pipe->lastPath : -1; }
void saf_hip_ambi_dec_batch_process(void* const hBatch,
                                    const float* d_in, long long in_inst_stride, long long in_frame_stride, long long in_ch_stride,
                                    float* d_out, long long out_inst_stride, long long out_frame_stride, long long out_ch_stride,
                                    int nFrames)
{
    DecPipeline* b = (DecPipeline*)hBatch;
    b->process(d_in, in_inst_stride, in_frame_stride, in_ch_stride, b->nSH < 4 ? 4 : b->nSH,
               d_out, out_inst_stride, out_frame_stride, out_ch_stride, nFrames);
}

}  // extern "C"
