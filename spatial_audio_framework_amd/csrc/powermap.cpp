/*
 * powermap.cpp — the powermap operator (examples/include/powermap.h:87-371, examples/src/powermap/powermap.c,
 * powermap_internal.c), PWD mode, with its per-frame path on the GPU:
 *
 *   FIFO (host, sample-wise like the reference) -> [afSTFT analysis, channel / norm conventions folded in]
 *        -> [per-band covariance update kernel] -> on request: [grouped covariance + PWD map + smoothing kernels]
 *        -> display interpolation / 0..1 normalisation of the 812-point map (host: three table gains per pixel)
 *
 * The sub-space and adaptive modes (MVDR, CroPaC-LCMV, MUSIC, MinNorm; saf_sh.c:1586-1858) need Hermitian
 * eigen-decompositions / solves and are listed as "next" (SURVEY §8f-4): requesting a map in one of them aborts.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "afstft_state.h"
#include "design_host.h"
#include "presets.h"
#include <thread>
#include <chrono>

namespace saf {

static int g_pm_frame_size = 1024;        /* default of the reference (powermap_internal.h:47-52) */
#define PM_NUM_DISP_SLOTS 2
#define PM_MAX_COV_AVG_COEFF 0.45f        /* powermap_internal.h:58 */

static inline void psleep_ms(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }

struct Powermap {
    int F, T;
    float fs = 48000.0f;
    float freqVector[SAF_NBANDS];
    std::vector<float> inFIFO;           /* [64][F] */
    int FIFO_idx = 0;
    /* codec pars (powermap_internal.h:64-73) */
    const float* grid_dirs_deg = nullptr; int grid_nDirs = 0;
    std::vector<float> interp_dirs_deg, interpComp; std::vector<int> interpIdx;
    int interp_nDirs = 0, interp_nTri = 0;
    std::vector<float> pmap, pmap_grid[PM_NUM_DISP_SLOTS];
    float pmap_grid_minVal = 0.0f, pmap_grid_maxVal = 0.0f;
    int dispSlotIdx = 0, pmapReady = 0, recalcPmap = 1, dispWidth = 140;
    volatile CODEC_STATUS codecStatus; volatile PROC_STATUS procStatus;
    float progressBar0_1 = 0.0f; char progressBarText[PROGRESSBARTEXT_CHAR_LENGTH];
    /* user parameters */
    int masterOrder = 1, new_masterOrder = 1, analysisOrderPerBand[SAF_NBANDS], nSources = 1, pmap_mode, HFOVoption = 1, aspectRatioOption = 1;
    float pmapEQ[SAF_NBANDS], covAvgCoeff = 0.0f, pmapAvgCoeff = 0.666f;
    CH_ORDER chOrdering = CH_ACN; NORM_TYPES norm = NORM_SN3D;
    /* device side */
    AfState st; int stftCh = 0;
    int Hmax = 0;
    DevBuf<float2> X, Cx;
    DevBuf<float> Ygrid[SAF_MAX_ORDER], Cg, d_pmap, d_prev, d_bandScale, d_chScale, d_in, d_adapt;
    DevBuf<double2> d_chol;
    DevBuf<int> d_bandNSH, d_chMap;
    PinBuf<float> stF, h_in, h_pmap; PinBuf<int> stI;
    int shadowNorm = -1, shadowChOrd = -1, shadowOrder = -1;
};

static void finish_map_on_host(Powermap* p, const float* pmapDev);

static void set_codec_status(Powermap* p, CODEC_STATUS s)      /* powermap_internal.c:32-44 */
{
    if (s == CODEC_STATUS_NOT_INITIALISED) while (p->codecStatus == CODEC_STATUS_INITIALISING) psleep_ms(10);
    p->codecStatus = s;
}

/* one or more full frames of device-resident samples: analysis + covariance update (+ map on request) */
static void analyse_frames_dev(Powermap* p, const float* d_in, long long in_frame, long long in_ch, int nChPresent, int nFrames)
{
    const int masterOrder = p->masterOrder, nSH = ORDER2NSH(masterOrder), T = p->T, H = nFrames * T;
    if (H > p->Hmax) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->Hmax = (H + 15) & ~15;
        p->X.alloc((size_t)SAF_NBANDS * SAF_MAXCH * p->Hmax, true);
    }
    if (p->shadowNorm != (int)p->norm || p->shadowChOrd != (int)p->chOrdering || p->shadowOrder != masterOrder) {
        /* input conventions -> ACN/N3D (powermap.c:241-252, saf_hoa.c:40-116) as a gather map + row scale */
        HIP_CHECK(hipStreamSynchronize(stream()));
        int* map = p->stI.p; float* sc = p->stF.p;
        for (int ch = 0; ch < SAF_MAXCH; ch++) { map[ch] = ch; sc[ch] = 1.0f; }
        if (p->chOrdering == CH_FUMA) { map[1] = 2; map[2] = 3; map[3] = 1; for (int ch = 4; ch < SAF_MAXCH; ch++) map[ch] = -1; }
        if (p->norm == NORM_SN3D) { for (int n = 0; n <= masterOrder; n++) for (int ch = n * n; ch < ORDER2NSH(n); ch++) sc[ch] = sqrtf(2.0f * (float)n + 1.0f); }
        else if (p->norm == NORM_FUMA) { sc[0] = sqrtf(2.0f); for (int ch = 1; ch < 4; ch++) sc[ch] = sqrtf(3.0f); }
        HIP_CHECK(hipMemcpyAsync(p->d_chMap.p, map, sizeof(int) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(p->d_chScale.p, sc, sizeof(float) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->shadowNorm = (int)p->norm; p->shadowChOrd = (int)p->chOrdering; p->shadowOrder = masterOrder;
    }
    AnaLaunch a{};
    a.in = d_in; a.in_inst = 0; a.in_ch = in_ch; a.in_frame = in_frame; a.hopsPerFrame = T; a.nChIn = nChPresent;
    a.hist_rd = p->st.ana[p->st.anaPar].p; a.hist_wr = p->st.ana[p->st.anaPar ^ 1].p;
    a.out = p->X.p; a.out_inst = 0; a.out_band = (long long)SAF_MAXCH * p->Hmax; a.out_ch = p->Hmax;
    a.ch_scale = p->d_chScale.p; a.ch_map = p->d_chMap.p; a.tab_stride = SAF_MAXCH;
    a.nCh = nSH; a.nInst = 1; a.H = H; a.lowDelay = 0; a.hybrid = 1;
    launch_analysis(a);
    p->st.anaPar ^= 1;

    CovLaunch c{};
    c.X = p->X.p; c.x_band = a.out_band; c.x_ch = a.out_ch; c.Cx = p->Cx.p;
    c.nSH = nSH; c.T = T; c.nFrames = nFrames;
    c.alpha = p->covAvgCoeff < PM_MAX_COV_AVG_COEFF ? p->covAvgCoeff : PM_MAX_COV_AVG_COEFF;
    launch_cov_update(c);

    if (p->recalcPmap == 1) {
        p->recalcPmap = 0; p->pmapReady = 0;
        int maxOrder = 1;
        for (int i = 0; i < SAF_NBANDS; i++) { const int o = p->analysisOrderPerBand[i] < masterOrder ? p->analysisOrderPerBand[i] : masterOrder; if (o > maxOrder) maxOrder = o; }
        HIP_CHECK(hipStreamSynchronize(stream()));
        for (int band = 0; band < SAF_NBANDS; band++) {
            int ob = p->analysisOrderPerBand[band] < masterOrder ? p->analysisOrderPerBand[band] : masterOrder; if (ob < 1) ob = 1;
            float eq = p->pmapEQ[band]; eq = eq < 0.0f ? 0.0f : (eq > 2.0f ? 2.0f : eq);
            p->stI.p[band] = ORDER2NSH(ob); p->stF.p[band] = 1e3f * eq;
        }
        HIP_CHECK(hipMemcpyAsync(p->d_bandNSH.p, p->stI.p, sizeof(int) * SAF_NBANDS, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(p->d_bandScale.p, p->stF.p, sizeof(float) * SAF_NBANDS, hipMemcpyHostToDevice, stream()));
        if (p->pmap_mode >= 2 && p->pmap_mode <= 7) {       /* PM_MODE_MVDR .. PM_MODE_MINNORM_LOG (powermap.c:299-341) */
            if (!p->d_adapt.p) {
                /* scratch: complex C_grp [64][64] | eigenvectors [64][64] | Un [64] | eig [64] | status, and the float64 factor */
                p->d_adapt.alloc((size_t)2 * 64 * 64 * 2 + 64 * 2 + 64 + 4);
                p->d_chol.alloc((size_t)64 * 64);
            }
            AdaptMapLaunch m{};
            m.Cx = p->Cx.p; m.bandScale = p->d_bandScale.p; m.bandNSH = p->d_bandNSH.p;
            float* sc = p->d_adapt.p;
            m.Cg = (float2*)sc; m.Veig = (float2*)(sc + 64 * 64 * 2); m.Un = (float2*)(sc + 2 * 64 * 64 * 2); m.eig = sc + 2 * 64 * 64 * 2 + 64 * 2;
            m.status = (int*)(sc + 2 * 64 * 64 * 2 + 64 * 2 + 64); m.Lchol = p->d_chol.p;
            m.Ygrid = p->Ygrid[maxOrder - 1].p; m.pmap = p->d_pmap.p; m.prev_pmap = p->d_prev.p;
            m.nM = ORDER2NSH(maxOrder); m.G = p->grid_nDirs; m.mode = p->pmap_mode; m.nSources = p->nSources; m.avg = p->pmapAvgCoeff;
            m.regPar = 8.0f; m.lambda = 0.0f;
            launch_adaptive_map(m);
        } else {                                             /* PM_MODE_PWD and unknown ids (the switch's default, powermap.c:295-298) */
            PwdLaunch w{};
            w.Cx = p->Cx.p; w.bandScale = p->d_bandScale.p; w.bandNSH = p->d_bandNSH.p; w.Cg = p->Cg.p;
            w.Ygrid = p->Ygrid[maxOrder - 1].p; w.pmap = p->d_pmap.p; w.prev_pmap = p->d_prev.p;
            w.nM = ORDER2NSH(maxOrder); w.G = p->grid_nDirs; w.avg = p->pmapAvgCoeff;
            launch_pwd_map(w);
        }
        const int G = p->grid_nDirs;
        HIP_CHECK(hipMemcpyAsync(p->h_pmap.p, p->d_pmap.p, sizeof(float) * G, hipMemcpyDeviceToHost, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        /* interpolate to the display grid (powermap.c:350-353: sgemm with the VBAP table, at most three non-zeros per row,
         * added in ascending column order like the dense product) and normalise to 0..1 (:356-364) */
        finish_map_on_host(p, p->h_pmap.p);
    }
}

/* host part of a map update (powermap.c:350-364): interpolation to the display grid, 0..1 normalisation, display slot */
static void finish_map_on_host(Powermap* p, const float* pmapDev)
{
    const int G = p->grid_nDirs;
    memcpy(p->pmap.data(), pmapDev, sizeof(float) * G);
    std::vector<float>& out = p->pmap_grid[p->dispSlotIdx];
    for (int i = 0; i < p->interp_nDirs; i++) {
        float s = 0.0f;
        for (int q = 0; q < 3; q++) s += p->interpComp[(size_t)i * 3 + q] * p->pmap[p->interpIdx[(size_t)i * 3 + q]];
        out[i] = s;
    }
    float mn = out[0], mx = out[0];
    for (int i = 1; i < p->interp_nDirs; i++) { if (out[i] < mn) mn = out[i]; if (out[i] > mx) mx = out[i]; }
    p->pmap_grid_minVal = mn; p->pmap_grid_maxVal = mx;
    for (int i = 0; i < p->interp_nDirs; i++) out[i] = (out[i] - mn) / (mx - mn + 1e-11f);
    p->dispSlotIdx++; if (p->dispSlotIdx >= PM_NUM_DISP_SLOTS) p->dispSlotIdx = 0;
    p->pmapReady = 1;
}

/* ---- batch of powermaps (PWD mode): nInst initialised handles with the same frame size and master order; every call advances all
 *      of them by nFrames frames (own filterbank state and covariances, like the other batches).  The handles keep their parameters
 *      (per-band orders, EQ, averaging coefficients, conventions) and receive the maps they asked for (powermap_requestPmapUpdate):
 *      powermap_getPmap / saf_hip_powermap_getRawPmap on a member handle return the batch's result. ---- */
struct PmBatch {
    std::vector<Powermap*> inst;
    int nInst = 0, F = 0, T = 0, order = 0, nSH = 0, maxFrames = 0, Hmax = 0, G = 0;
    AfState st;
    DevBuf<float2> X, Cx;
    DevBuf<float> Cg, pmap, prev, bandScale, chScale, alpha, avg;
    DevBuf<int> bandNSH, chMap, mapOrder;
    PinBuf<float> stF, hPmap; PinBuf<int> stI;
    std::vector<int> shNorm, shChOrd;
    std::vector<float> shAlpha;
    bool stagePending = false;

    void create(Powermap* const* h, int n, int maxFrames_)
    {
        nInst = n; inst.assign(h, h + n); maxFrames = maxFrames_;
        Powermap* p0 = inst[0];
        F = p0->F; T = p0->T; order = p0->masterOrder; nSH = ORDER2NSH(order); G = p0->grid_nDirs;
        for (int i = 0; i < n; i++) {
            Powermap* p = inst[i];
            if (p->codecStatus != CODEC_STATUS_INITIALISED) SAF_FATAL("powermap batch: instance %d is not initialised (call powermap_initCodec)", i);
            if (p->F != F || p->masterOrder != order) SAF_FATAL("powermap batch: all instances must share frame size and master order");
            if (p->FIFO_idx != 0) SAF_FATAL("powermap batch: instance %d has a partly filled input FIFO", i);
        }
        Hmax = (T * maxFrames + 15) & ~15;
        st.create(nInst, nSH, 0);
        X.alloc((size_t)nInst * SAF_NBANDS * SAF_MAXCH * Hmax, true);
        Cx.alloc((size_t)nInst * SAF_NBANDS * 64 * 64);
        Cg.alloc((size_t)nInst * 64 * 64); pmap.alloc((size_t)nInst * G); prev.alloc((size_t)nInst * G);
        bandScale.alloc((size_t)nInst * SAF_NBANDS); bandNSH.alloc((size_t)nInst * SAF_NBANDS);
        chScale.alloc((size_t)nInst * SAF_MAXCH); chMap.alloc((size_t)nInst * SAF_MAXCH); alpha.alloc(nInst); avg.alloc(nInst); mapOrder.alloc(nInst);
        stF.ensure((size_t)nInst * (SAF_NBANDS + SAF_MAXCH + 2)); stI.ensure((size_t)nInst * (SAF_NBANDS + SAF_MAXCH + 1)); hPmap.ensure((size_t)nInst * G);
        shNorm.assign(nInst, -1); shChOrd.assign(nInst, -1); shAlpha.assign(nInst, -1.0f);
    }

    void analysis(const float* d_in, long long in_inst, long long in_frame, long long in_ch, int nIn, int nFrames)
    {
        if (nFrames <= 0) return;
        if (nFrames > maxFrames) SAF_FATAL("powermap batch: nFrames %d exceeds the maxFramesPerCall %d given at creation", nFrames, maxFrames);
        const int H = nFrames * T;
        /* per-instance tables whose inputs changed (conventions, covariance averaging), staged per instance: no sync between instances */
        bool began = false;
        auto stage = [&]() { if (!began) { if (stagePending) { HIP_CHECK(hipStreamSynchronize(stream())); stagePending = false; } began = true; } };
        float* fS = stF.p; int* iS = stI.p;
        const size_t fPer = SAF_NBANDS + SAF_MAXCH + 2, iPer = SAF_NBANDS + SAF_MAXCH + 1;
        for (int i = 0; i < nInst; i++) {
            Powermap* p = inst[i];
            if (p->pmap_mode != 1 && p->recalcPmap == 1) SAF_FATAL("powermap batch: only the PWD map is batched (instance %d asks for mode %d)", i, p->pmap_mode);
            if (shNorm[i] != (int)p->norm || shChOrd[i] != (int)p->chOrdering) {
                stage();
                int* map = iS + i * iPer + SAF_NBANDS; float* sc = fS + i * fPer + SAF_NBANDS;
                for (int ch = 0; ch < SAF_MAXCH; ch++) { map[ch] = ch; sc[ch] = 1.0f; }
                if (p->chOrdering == CH_FUMA) { map[1] = 2; map[2] = 3; map[3] = 1; for (int ch = 4; ch < SAF_MAXCH; ch++) map[ch] = -1; }
                if (p->norm == NORM_SN3D) { for (int n = 0; n <= order; n++) for (int ch = n * n; ch < ORDER2NSH(n); ch++) sc[ch] = sqrtf(2.0f * (float)n + 1.0f); }
                else if (p->norm == NORM_FUMA) { sc[0] = sqrtf(2.0f); for (int ch = 1; ch < 4; ch++) sc[ch] = sqrtf(3.0f); }
                HIP_CHECK(hipMemcpyAsync(chMap.p + (size_t)i * SAF_MAXCH, map, sizeof(int) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
                HIP_CHECK(hipMemcpyAsync(chScale.p + (size_t)i * SAF_MAXCH, sc, sizeof(float) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
                shNorm[i] = (int)p->norm; shChOrd[i] = (int)p->chOrdering;
            }
            const float al = p->covAvgCoeff < PM_MAX_COV_AVG_COEFF ? p->covAvgCoeff : PM_MAX_COV_AVG_COEFF;
            if (shAlpha[i] != al) {
                stage();
                float* slot = fS + i * fPer + SAF_NBANDS + SAF_MAXCH;
                slot[0] = al;
                HIP_CHECK(hipMemcpyAsync(alpha.p + i, slot, sizeof(float), hipMemcpyHostToDevice, stream()));
                shAlpha[i] = al;
            }
        }
        AnaLaunch a{};
        a.in = d_in; a.in_inst = in_inst; a.in_ch = in_ch; a.in_frame = in_frame; a.hopsPerFrame = T; a.nChIn = nIn;
        a.hist_rd = st.ana[st.anaPar].p; a.hist_wr = st.ana[st.anaPar ^ 1].p;
        a.out = X.p; a.out_inst = (long long)SAF_NBANDS * SAF_MAXCH * Hmax; a.out_band = (long long)SAF_MAXCH * Hmax; a.out_ch = Hmax;
        a.ch_scale = chScale.p; a.ch_map = chMap.p; a.tab_stride = SAF_MAXCH;
        a.nCh = nSH; a.nInst = nInst; a.H = H; a.lowDelay = 0; a.hybrid = 1;
        launch_analysis(a);
        st.anaPar ^= 1;

        CovLaunch c{};
        c.X = X.p; c.x_inst = a.out_inst; c.x_band = a.out_band; c.x_ch = a.out_ch; c.Cx = Cx.p; c.cx_inst = (long long)SAF_NBANDS * 64 * 64;
        c.nSH = nSH; c.T = T; c.nFrames = nFrames; c.nInst = nInst; c.alphaInst = alpha.p; c.alpha = 0.0f;
        launch_cov_update(c);

        /* maps for the instances that asked (powermap.c:270-272): one launch pair over all instances, the others' workgroups leave */
        std::vector<int> want;
        for (int i = 0; i < nInst; i++) if (inst[i]->recalcPmap == 1) want.push_back(i);
        if (!want.empty()) {
            stage();
            for (int i = 0; i < nInst; i++) {
                Powermap* p = inst[i];
                int* io = iS + i * iPer; float* fo = fS + i * fPer;
                int maxOrder = 1;
                for (int band = 0; band < SAF_NBANDS; band++) {
                    int ob = p->analysisOrderPerBand[band] < order ? p->analysisOrderPerBand[band] : order; if (ob < 1) ob = 1;
                    if (ob > maxOrder) maxOrder = ob;
                    float eq = p->pmapEQ[band]; eq = eq < 0.0f ? 0.0f : (eq > 2.0f ? 2.0f : eq);
                    io[band] = ORDER2NSH(ob); fo[band] = 1e3f * eq;
                }
                io[SAF_NBANDS + SAF_MAXCH] = p->recalcPmap == 1 ? maxOrder : 0;
                fo[SAF_NBANDS + SAF_MAXCH + 1] = p->pmapAvgCoeff;
                if (p->recalcPmap != 1) continue;
                HIP_CHECK(hipMemcpyAsync(bandNSH.p + (size_t)i * SAF_NBANDS, io, sizeof(int) * SAF_NBANDS, hipMemcpyHostToDevice, stream()));
                HIP_CHECK(hipMemcpyAsync(bandScale.p + (size_t)i * SAF_NBANDS, fo, sizeof(float) * SAF_NBANDS, hipMemcpyHostToDevice, stream()));
                HIP_CHECK(hipMemcpyAsync(avg.p + i, fo + SAF_NBANDS + SAF_MAXCH + 1, sizeof(float), hipMemcpyHostToDevice, stream()));
            }
            for (int i = 0; i < nInst; i++)
                HIP_CHECK(hipMemcpyAsync(mapOrder.p + i, iS + i * iPer + SAF_NBANDS + SAF_MAXCH, sizeof(int), hipMemcpyHostToDevice, stream()));
            PwdLaunch w{};
            w.Cx = Cx.p; w.cx_inst = c.cx_inst; w.bandScale = bandScale.p; w.bandNSH = bandNSH.p; w.Cg = Cg.p;
            w.pmap = pmap.p; w.prev_pmap = prev.p; w.G = G; w.nInst = nInst; w.mapOrder = mapOrder.p; w.avgInst = avg.p;
            for (int n = 0; n < order; n++) w.YgridByOrder[n] = inst[0]->Ygrid[n].p;
            w.Ygrid = inst[0]->Ygrid[order - 1].p; w.nM = nSH; w.avg = 0.0f;
            launch_pwd_map(w);
            HIP_CHECK(hipMemcpyAsync(hPmap.p, pmap.p, sizeof(float) * (size_t)nInst * G, hipMemcpyDeviceToHost, stream()));
            HIP_CHECK(hipStreamSynchronize(stream()));
            stagePending = false; began = false;
            for (int i : want) { inst[i]->recalcPmap = 0; inst[i]->pmapReady = 0; finish_map_on_host(inst[i], hPmap.p + (size_t)i * G); }
        }
        if (began) stagePending = true;
    }
};

}  // namespace saf

using namespace saf;

extern "C" {

void* saf_hip_powermap_batch_create(void* const* hPms, int nInst, int maxFramesPerCall)
{
    if (nInst <= 0 || maxFramesPerCall <= 0) SAF_FATAL("powermap batch: nInst and maxFramesPerCall must be positive");
    ensure_device();
    PmBatch* b = new PmBatch();
    b->create((Powermap* const*)hPms, nInst, maxFramesPerCall);
    return b;
}
void saf_hip_powermap_batch_destroy(void** const phBatch)
{
    if (!phBatch || !*phBatch) return;
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete (PmBatch*)*phBatch;
    *phBatch = nullptr;
}
void saf_hip_powermap_batch_analysis(void* const hBatch, const float* d_in, long long in_inst_stride, long long in_frame_stride, long long in_ch_stride, int nInputs, int nFrames)
{
    PmBatch* b = (PmBatch*)hBatch;
    b->analysis(d_in, in_inst_stride, in_frame_stride, in_ch_stride, nInputs, nFrames);
}
void saf_hip_powermap_batch_getCx(void* const hBatch, int instIdx, float_complex* Cx)
{
    PmBatch* b = (PmBatch*)hBatch;
    if (instIdx < 0 || instIdx >= b->nInst) SAF_FATAL("powermap batch: instance index out of range");
    const int nSH = b->nSH;
    std::vector<float2> h((size_t)SAF_NBANDS * 64 * 64);
    HIP_CHECK(hipStreamSynchronize(stream()));
    HIP_CHECK(hipMemcpy(h.data(), b->Cx.p + (size_t)instIdx * SAF_NBANDS * 64 * 64, sizeof(float2) * h.size(), hipMemcpyDeviceToHost));
    float2* o = reinterpret_cast<float2*>(Cx);
    for (int bd = 0; bd < SAF_NBANDS; bd++) for (int i = 0; i < nSH; i++) for (int j = 0; j < nSH; j++) o[((size_t)bd * nSH + i) * nSH + j] = h[(size_t)bd * 4096 + i * 64 + j];
}

void saf_hip_powermap_setFrameSize(int frameSize)
{
    if (frameSize <= 0 || frameSize % SAF_HOP != 0 || frameSize > 2048) SAF_FATAL("powermap frame size must be a multiple of 128, at most 2048");
    g_pm_frame_size = frameSize;
}

void powermap_create(void** const phPm)
{
    Powermap* p = new Powermap();
    *phPm = p;
    p->F = g_pm_frame_size; p->T = p->F / SAF_HOP;
    for (int b = 0; b < SAF_NBANDS; b++) { p->analysisOrderPerBand[b] = 1; p->pmapEQ[b] = 1.0f; }
    p->pmap_mode = 4;      /* PM_MODE_MUSIC, the reference's default (powermap.c:50) */
    p->codecStatus = CODEC_STATUS_NOT_INITIALISED; p->procStatus = PROC_STATUS_NOT_ONGOING;
    p->progressBarText[0] = 0;
    p->inFIFO.assign((size_t)SAF_MAXCH * p->F, 0.0f);
    memset(p->freqVector, 0, sizeof(p->freqVector));
}

void powermap_destroy(void** const phPm)
{
    Powermap* p = (Powermap*)*phPm;
    if (!p) return;
    while (p->codecStatus == CODEC_STATUS_INITIALISING || p->procStatus == PROC_STATUS_ONGOING) psleep_ms(10);
    if (p->stftCh) HIP_CHECK(hipStreamSynchronize(stream()));
    delete p;
    *phPm = nullptr;
}

void powermap_init(void* const hPm, float sampleRate)
{
    Powermap* p = (Powermap*)hPm;
    p->fs = sampleRate;
    /* the reference creates its filterbank in powermap_create, so this is always the valid-handle branch of
     * afSTFT_getCentreFreqs (afSTFTlib.c:565-587), hop 128 hybrid */
    static const float w[9] = { 1.0f, 0.7501f, 1.2499f, 0.8751f, 1.1249f, 0.9167f, 1.0833f, 0.9375f, 1.0625f };
    static const int bin[9] = { 0, 1, 1, 2, 2, 3, 3, 4, 4 };
    for (int i = 0; i < 9; i++) p->freqVector[i] = w[i] * ((float)bin[i] * sampleRate / 256.0f);
    for (int i = 9, j = 5; i < SAF_NBANDS; i++, j++) p->freqVector[i] = (float)j * sampleRate / 256.0f;
    if (p->Cx.p) p->Cx.zero();
    if (p->d_prev.p) p->d_prev.zero();
    p->pmapReady = 0; p->dispSlotIdx = 0;
}

void powermap_initCodec(void* const hPm)
{
    Powermap* p = (Powermap*)hPm;
    if (p->codecStatus != CODEC_STATUS_NOT_INITIALISED) return;
    while (p->procStatus == PROC_STATUS_ONGOING) { p->codecStatus = CODEC_STATUS_INITIALISING; psleep_ms(10); }
    ensure_device();
    p->codecStatus = CODEC_STATUS_INITIALISING;
    strcpy(p->progressBarText, "Initialising"); p->progressBar0_1 = 0.0f;
    HIP_CHECK(hipStreamSynchronize(stream()));
    const int order = p->new_masterOrder, new_nSH = ORDER2NSH(order);
    /* powermap_initTFT (powermap_internal.c:123-136): new channel count -> fresh filterbank state and covariances */
    if (!p->stftCh) {
        p->st.create(1, new_nSH, 0); p->stftCh = new_nSH;
        p->Cx.alloc((size_t)SAF_NBANDS * 64 * 64); p->Cg.alloc(64 * 64);
        p->d_bandScale.alloc(SAF_NBANDS); p->d_bandNSH.alloc(SAF_NBANDS); p->d_chScale.alloc(SAF_MAXCH); p->d_chMap.alloc(SAF_MAXCH);
        p->stF.ensure(SAF_NBANDS + SAF_MAXCH); p->stI.ensure(SAF_NBANDS + SAF_MAXCH);
    } else if (new_nSH != p->stftCh) {
        p->st.channelChange(new_nSH, 0); p->st.clear(); p->stftCh = new_nSH;
        p->Cx.zero();
    }
    /* powermap_initAna (powermap_internal.c:46-121) */
    int d0 = 0, d1 = 0;
    p->grid_dirs_deg = table("geosphere_ico_9_0_dirs_deg", &d0, &d1);
    if (!p->grid_dirs_deg) SAF_FATAL("table geosphere_ico_9_0_dirs_deg missing");
    p->grid_nDirs = d0;
    const int G = p->grid_nDirs;
    std::vector<float> Y((size_t)new_nSH * G), Yn;
    sh_eval_host(1, order, p->grid_dirs_deg, G, Y.data());                  /* getRSH */
    for (int n = 1; n <= order; n++) {
        const int ns = ORDER2NSH(n); const float sc = 1.0f / (float)ns;
        Yn.assign(Y.begin(), Y.begin() + (size_t)ns * G);
        for (float& v : Yn) v *= sc;
        p->Ygrid[n - 1].alloc((size_t)ns * G, false);
        HIP_CHECK(hipMemcpy(p->Ygrid[n - 1].p, Yn.data(), sizeof(float) * Yn.size(), hipMemcpyHostToDevice));
    }
    const float hfov = 360.0f, aspect = 2.0f;
    const int N_azi = p->dispWidth, N_ele = (int)((float)p->dispWidth / aspect + 0.5f);
    const float vfov = hfov / aspect;
    std::vector<float> gx(N_azi), gy(N_ele);
    { float fi = -hfov / 2.0f; for (int i = 0; i < N_azi; fi += hfov / N_azi, i++) gx[i] = fi; }
    { float fi = -vfov / 2.0f; for (int i = 0; i < N_ele; fi += vfov / N_ele, i++) gy[i] = fi; }
    p->interp_dirs_deg.resize((size_t)N_azi * N_ele * 2);
    for (int i = 0; i < N_ele; i++) for (int j = 0; j < N_azi; j++) { p->interp_dirs_deg[(i * N_azi + j) * 2] = gx[j]; p->interp_dirs_deg[(i * N_azi + j) * 2 + 1] = gy[i]; }
    p->interp_nDirs = N_azi * N_ele;
    std::vector<float> gt;
    if (!vbap_table(p->interp_dirs_deg.data(), p->interp_nDirs, p->grid_dirs_deg, G, 0, 0, 0.0f, gt, &p->interp_nTri))
        SAF_FATAL("powermap: the scanning grid could not be triangulated");
    VBAPgainTable2InterpTable(gt.data(), p->interp_nDirs, G);
    /* keep the (at most three) non-zeros of every row, in ascending column order */
    p->interpComp.assign((size_t)p->interp_nDirs * 3, 0.0f); p->interpIdx.assign((size_t)p->interp_nDirs * 3, 0);
    for (int i = 0; i < p->interp_nDirs; i++) {
        int q = 0;
        for (int j = 0; j < G && q < 3; j++) if (gt[(size_t)i * G + j] != 0.0f) { p->interpComp[(size_t)i * 3 + q] = gt[(size_t)i * G + j]; p->interpIdx[(size_t)i * 3 + q] = j; q++; }
    }
    p->pmap.assign(G, 0.0f);
    p->d_pmap.alloc(G); p->d_prev.alloc(G); p->h_pmap.ensure(G);
    for (int i = 0; i < PM_NUM_DISP_SLOTS; i++) p->pmap_grid[i].assign(p->interp_nDirs, 0.0f);
    p->masterOrder = order;
    strcpy(p->progressBarText, "Done!"); p->progressBar0_1 = 1.0f;
    p->codecStatus = CODEC_STATUS_INITIALISED;
}

void powermap_analysis(void* const hPm, const float* const* inputs, int nInputs, int nSamples, int isPlaying)
{
    Powermap* p = (Powermap*)hPm;
    const int nSH = ORDER2NSH(p->masterOrder), F = p->F;
    for (int s = 0; s < nSamples; s++) {
        int ch;
        for (ch = 0; ch < (nInputs < nSH ? nInputs : nSH); ch++) p->inFIFO[(size_t)ch * F + p->FIFO_idx] = inputs[ch][s];
        for (; ch < nSH; ch++) p->inFIFO[(size_t)ch * F + p->FIFO_idx] = 0.0f;
        p->FIFO_idx++;
        if (p->FIFO_idx >= F && p->codecStatus == CODEC_STATUS_INITIALISED && isPlaying) {
            p->FIFO_idx = 0;
            p->procStatus = PROC_STATUS_ONGOING;
            const int rows = nSH < 4 ? 4 : nSH;                 /* a FuMa gather may read rows up to 3 */
            p->h_in.ensure((size_t)SAF_MAXCH * F);
            if (p->d_in.n < (size_t)SAF_MAXCH * F) p->d_in.alloc((size_t)SAF_MAXCH * F, true);
            memcpy(p->h_in.p, p->inFIFO.data(), sizeof(float) * (size_t)rows * F);
            if (zero_copy_io()) analyse_frames_dev(p, p->h_in.p, 0, F, rows, 1);                             /* kernels on the pinned block */
            else {
                HIP_CHECK(hipMemcpyAsync(p->d_in.p, p->h_in.p, sizeof(float) * (size_t)rows * F, hipMemcpyHostToDevice, stream()));
                analyse_frames_dev(p, p->d_in.p, 0, F, rows, 1);
            }
            HIP_CHECK(hipStreamSynchronize(stream()));         /* h_in is reused by the next frame */
        } else if (p->FIFO_idx >= F) p->FIFO_idx = 0;
    }
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}

void saf_hip_powermap_analysis_dev(void* const hPm, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs, int nFrames)
{
    Powermap* p = (Powermap*)hPm;
    if (p->codecStatus != CODEC_STATUS_INITIALISED) SAF_FATAL("powermap: analysis_dev on a handle that is not initialised (call powermap_initCodec)");
    if (p->FIFO_idx != 0) SAF_FATAL("powermap: analysis_dev needs an empty input FIFO (do not mix it with partial powermap_analysis blocks)");
    p->procStatus = PROC_STATUS_ONGOING;
    analyse_frames_dev(p, d_in, in_frame_stride, in_ch_stride, nInputs, nFrames);
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}

/* ------------------------------- set functions (powermap.c:384-570) ------------------------------- */
#define PPM Powermap* p = (Powermap*)hPm
void powermap_refreshSettings(void* const hPm) { PPM; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
void powermap_setPowermapMode(void* const hPm, int newMode) { PPM; p->pmap_mode = newMode; if (p->d_prev.p) p->d_prev.zero(); }
void powermap_setMasterOrder(void* const hPm, int newValue)
{
    PPM;
    if (p->new_masterOrder != newValue) { p->new_masterOrder = newValue; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
    if (p->new_masterOrder != 1 && p->chOrdering == CH_FUMA) p->chOrdering = CH_ACN;       /* FuMa is first-order only */
    if (p->new_masterOrder != 1 && p->norm == NORM_FUMA) p->norm = NORM_SN3D;
}
void powermap_setCovAvgCoeff(void* const hPm, float v) { PPM; p->covAvgCoeff = v < 0.0f ? 0.0f : (v > 0.99999999f ? 0.99999999f : v); }
void powermap_setNumSources(void* const hPm, int v) { PPM; p->nSources = v; }
void powermap_setSourcePreset(void* const hPm, int newPresetID)
{
    PPM;
    mic_preset_order_per_band(newPresetID, p->new_masterOrder, p->freqVector, SAF_NBANDS, p->analysisOrderPerBand);
    /* above the spatial-aliasing limit of a real array the map is EQ'd out (powermap.c:441-443, :460-462, :479-481) */
    const char* tab = newPresetID == 2 ? "Zylia_freqRange" : newPresetID == 3 ? "Eigenmike32_freqRange" : newPresetID == 4 ? "DTU_mic_freqRange" : nullptr;
    const int maxOrder = newPresetID == 2 ? 3 : newPresetID == 3 ? 4 : 6;
    if (tab) {
        int d0 = 0, d1 = 0; const float* range = table(tab, &d0, &d1);
        if (range) for (int b = 0; b < SAF_NBANDS; b++) if (p->freqVector[b] > range[(maxOrder - 1) * 2 - 1]) p->pmapEQ[b] = 0.0f;
    }
}
void powermap_setAnaOrder(void* const hPm, int v, int bandIdx) { PPM; p->analysisOrderPerBand[bandIdx] = v < 1 ? 1 : (v > p->new_masterOrder ? p->new_masterOrder : v); }
void powermap_setAnaOrderAllBands(void* const hPm, int v) { PPM; for (int b = 0; b < SAF_NBANDS; b++) p->analysisOrderPerBand[b] = v < 1 ? 1 : (v > p->new_masterOrder ? p->new_masterOrder : v); }
void powermap_setPowermapEQ(void* const hPm, float v, int bandIdx) { PPM; p->pmapEQ[bandIdx] = v; }
void powermap_setPowermapEQAllBands(void* const hPm, float v) { PPM; for (int b = 0; b < SAF_NBANDS; b++) p->pmapEQ[b] = v; }
void powermap_setChOrder(void* const hPm, int v) { PPM; if ((CH_ORDER)v != CH_FUMA || p->new_masterOrder == 1) p->chOrdering = (CH_ORDER)v; }
void powermap_setNormType(void* const hPm, int v) { PPM; if ((NORM_TYPES)v != NORM_FUMA || p->new_masterOrder == 1) p->norm = (NORM_TYPES)v; }
void powermap_setDispFOV(void* const hPm, int v) { PPM; if (p->HFOVoption != v) { p->HFOVoption = v; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); } }
void powermap_setAspectRatio(void* const hPm, int v) { PPM; if (p->aspectRatioOption != v) { p->aspectRatioOption = v; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); } }
void powermap_setPowermapAvgCoeff(void* const hPm, float v) { PPM; p->pmapAvgCoeff = v < 0.0f ? 0.0f : (v > 0.99999999f ? 0.99999999f : v); }
void powermap_requestPmapUpdate(void* const hPm) { PPM; p->recalcPmap = 1; }

/* ------------------------------- get functions (powermap.c:573-745) ------------------------------- */
int powermap_getFrameSize(void) { return g_pm_frame_size; }
CODEC_STATUS powermap_getCodecStatus(void* const hPm) { PPM; return p->codecStatus; }
float powermap_getProgressBar0_1(void* const hPm) { PPM; return p->progressBar0_1; }
void powermap_getProgressBarText(void* const hPm, char* text) { PPM; memcpy(text, p->progressBarText, PROGRESSBARTEXT_CHAR_LENGTH); }
int powermap_getMasterOrder(void* const hPm) { PPM; return p->new_masterOrder; }
int powermap_getPowermapMode(void* const hPm) { PPM; return p->pmap_mode; }
int powermap_getSamplingRate(void* const hPm) { PPM; return (int)(p->fs + 0.5f); }
float powermap_getCovAvgCoeff(void* const hPm) { PPM; return p->covAvgCoeff; }
int powermap_getNumberOfBands(void) { return SAF_NBANDS; }
int powermap_getNSHrequired(void* const hPm) { PPM; return (p->new_masterOrder + 1) * (p->new_masterOrder + 1); }
float powermap_getPowermapEQ(void* const hPm, int bandIdx) { PPM; return p->pmapEQ[bandIdx]; }
float powermap_getPowermapEQAllBands(void* const hPm) { PPM; return p->pmapEQ[0]; }
void powermap_getPowermapEQHandle(void* const hPm, float** pX_vector, float** pY_values, int* pNpoints) { PPM; *pX_vector = p->freqVector; *pY_values = p->pmapEQ; *pNpoints = SAF_NBANDS; }
int powermap_getAnaOrder(void* const hPm, int bandIdx) { PPM; return p->analysisOrderPerBand[bandIdx]; }
int powermap_getAnaOrderAllBands(void* const hPm) { PPM; return p->analysisOrderPerBand[0]; }
void powermap_getAnaOrderHandle(void* const hPm, float** pX_vector, int** pY_values, int* pNpoints) { PPM; *pX_vector = p->freqVector; *pY_values = p->analysisOrderPerBand; *pNpoints = SAF_NBANDS; }
int powermap_getChOrder(void* const hPm) { PPM; return (int)p->chOrdering; }
int powermap_getNormType(void* const hPm) { PPM; return (int)p->norm; }
int powermap_getNumSources(void* const hPm) { PPM; return p->nSources; }
int powermap_getDispFOV(void* const hPm) { PPM; return p->HFOVoption; }
int powermap_getAspectRatio(void* const hPm) { PPM; return p->aspectRatioOption; }
float powermap_getPowermapAvgCoeff(void* const hPm) { PPM; return p->pmapAvgCoeff; }
int powermap_getPmap(void* const hPm, float** grid_dirs, float** pmap, int* nDirs, int* pmapWidth, int* hfov, int* aspectRatio)
{
    PPM;
    if (p->codecStatus == CODEC_STATUS_INITIALISED && p->pmapReady) {
        *grid_dirs = p->interp_dirs_deg.data();
        *pmap = p->pmap_grid[p->dispSlotIdx - 1 < 0 ? PM_NUM_DISP_SLOTS - 1 : p->dispSlotIdx - 1].data();
        *nDirs = p->interp_nDirs; *pmapWidth = p->dispWidth; *hfov = 360; *aspectRatio = 2;
    }
    return p->pmapReady;
}
int powermap_getProcessingDelay(void) { return g_pm_frame_size + 12 * SAF_HOP; }

/* read-back for parity checks: covariance matrices as [133][nSH][nSH] and the smoothed 812-point map */
void saf_hip_powermap_getCx(void* const hPm, float_complex* Cx)
{
    PPM;
    const int nSH = ORDER2NSH(p->masterOrder);
    std::vector<float2> h((size_t)SAF_NBANDS * 64 * 64);
    HIP_CHECK(hipStreamSynchronize(stream()));
    HIP_CHECK(hipMemcpy(h.data(), p->Cx.p, sizeof(float2) * h.size(), hipMemcpyDeviceToHost));
    float2* o = reinterpret_cast<float2*>(Cx);
    for (int b = 0; b < SAF_NBANDS; b++) for (int i = 0; i < nSH; i++) for (int j = 0; j < nSH; j++) o[((size_t)b * nSH + i) * nSH + j] = h[(size_t)b * 4096 + i * 64 + j];
}
int saf_hip_powermap_getRawPmap(void* const hPm, float* pmap) { PPM; memcpy(pmap, p->pmap.data(), sizeof(float) * p->pmap.size()); return (int)p->pmap.size(); }

/* ---- stand-alone map generators (saf_sh.h / saf_sh.c:1544-1858): one nSH x nSH covariance, host pointers.
 *      Y_grid is passed as complex like in the reference but must be real-valued (all its callers build it from real SH). ---- */
static void run_generate_map(int mode, int order, const float_complex* Cx, const float_complex* Y_grid, int G, float regPar, float lambda,
                             int nSources, float* pmap, float_complex* w_out)
{
    ensure_device();
    const int nSH = ORDER2NSH(order);
    if (order < 1 || order > SAF_MAX_ORDER || G < 1) SAF_FATAL("generate*map: order 1..7 and at least one grid direction");
    std::vector<float2> C((size_t)64 * 64, make_float2(0.f, 0.f));
    const float2* cx = reinterpret_cast<const float2*>(Cx);
    for (int i = 0; i < nSH; i++) for (int j = 0; j < nSH; j++) C[(size_t)i * 64 + j] = cx[(size_t)i * nSH + j];
    std::vector<float> Y((size_t)nSH * G);
    const float2* yg = reinterpret_cast<const float2*>(Y_grid);
    for (size_t i = 0; i < Y.size(); i++) { if (yg[i].y != 0.0f) SAF_FATAL("generate*map: Y_grid must be real-valued (zero imaginary parts)"); Y[i] = yg[i].x; }
    DevBuf<float2> dC, dSc, dW; DevBuf<float> dY, dP, dPrev, dScale, dEig; DevBuf<int> dN, dSt; DevBuf<double2> dL;
    dC.alloc(C.size(), false); dSc.alloc((size_t)2 * 64 * 64 + 64); dY.alloc(Y.size(), false); dP.alloc(G); dPrev.alloc(G); dScale.alloc(1, false); dN.alloc(1, false);
    dSt.alloc(1); dEig.alloc(64); dL.alloc((size_t)64 * 64);
    if (w_out) dW.alloc((size_t)nSH * G);
    const float one = 1.0f;
    HIP_CHECK(hipMemcpyAsync(dC.p, C.data(), sizeof(float2) * C.size(), hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(dY.p, Y.data(), sizeof(float) * Y.size(), hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(dScale.p, &one, sizeof(float), hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(dN.p, &nSH, sizeof(int), hipMemcpyHostToDevice, stream()));
    if (mode == 1) {
        DevBuf<float> dCg; dCg.alloc((size_t)64 * 64);
        PwdLaunch w{};
        w.Cx = dC.p; w.bandScale = dScale.p; w.bandNSH = dN.p; w.Cg = dCg.p; w.Ygrid = dY.p; w.pmap = dP.p; w.prev_pmap = dPrev.p;
        w.nM = nSH; w.G = G; w.avg = 0.0f; w.nBands = 1;
        launch_pwd_map(w);
        HIP_CHECK(hipStreamSynchronize(stream()));
    } else {
        AdaptMapLaunch m{};
        m.Cx = dC.p; m.bandScale = dScale.p; m.bandNSH = dN.p; m.Cg = dSc.p; m.Veig = dSc.p + 64 * 64; m.Un = dSc.p + 2 * 64 * 64; m.eig = dEig.p; m.status = dSt.p; m.Lchol = dL.p;
        m.Ygrid = dY.p; m.pmap = dP.p; m.prev_pmap = dPrev.p; m.nM = nSH; m.G = G; m.mode = mode; m.nSources = nSources; m.avg = 0.0f; m.regPar = regPar; m.lambda = lambda;
        m.nBands = 1; m.Wout = w_out ? dW.p : nullptr;
        launch_adaptive_map(m);
    }
    HIP_CHECK(hipMemcpyAsync(pmap, dP.p, sizeof(float) * G, hipMemcpyDeviceToHost, stream()));
    if (w_out) HIP_CHECK(hipMemcpyAsync((void*)w_out, dW.p, sizeof(float2) * (size_t)nSH * G, hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
}
void generatePWDmap(int order, float_complex* Cx, float_complex* Y_grid, int nGrid_dirs, float* pmap) { run_generate_map(1, order, Cx, Y_grid, nGrid_dirs, 0.f, 0.f, 0, pmap, nullptr); }
void generateMVDRmap(int order, float_complex* Cx, float_complex* Y_grid, int nGrid_dirs, float regPar, float* pmap, float_complex* w_MVDR_out)
{ run_generate_map(2, order, Cx, Y_grid, nGrid_dirs, regPar, 0.f, 0, pmap, w_MVDR_out); }
void generateCroPaCLCMVmap(int order, float_complex* Cx, float_complex* Y_grid, int nGrid_dirs, float regPar, float lambda, float* pmap)
{ run_generate_map(3, order, Cx, Y_grid, nGrid_dirs, regPar, lambda, 0, pmap, nullptr); }
void generateMUSICmap(int order, float_complex* Cx, float_complex* Y_grid, int nSources, int nGrid_dirs, int logScaleFlag, float* pmap)
{ run_generate_map(logScaleFlag ? 5 : 4, order, Cx, Y_grid, nGrid_dirs, 0.f, 0.f, nSources, pmap, nullptr); }
void generateMinNormMap(int order, float_complex* Cx, float_complex* Y_grid, int nSources, int nGrid_dirs, int logScaleFlag, float* pmap)
{ run_generate_map(logScaleFlag ? 7 : 6, order, Cx, Y_grid, nGrid_dirs, 0.f, 0.f, nSources, pmap, nullptr); }

/* ---- sphPWD / sphMUSIC objects (saf_sh.h; saf_sh.c:1042-1306): steering vectors = orthonormal real SH of the scanning grid,
 *      map on the GPU, the von-Mises-masked peak search (:1139-1169, :1275-1305) on the host ---- */
struct SphScan {
    int order, nSH, nDirs;
    std::vector<float> svecs;       /* [nSH][nDirs] */
    std::vector<float> xyz;         /* [nDirs][3] */
    std::vector<float> pSpec;
};
static SphScan* sph_scan_create(int order, const float* grid_dirs_deg, int nDirs)
{
    ensure_device();
    if (order < 1 || order > SAF_MAX_ORDER || nDirs < 1) SAF_FATAL("sphPWD/sphMUSIC: order 1..7 and at least one grid direction");
    SphScan* h = new SphScan();
    h->order = order; h->nSH = ORDER2NSH(order); h->nDirs = nDirs;
    std::vector<float> rad((size_t)nDirs * 2);
    for (int i = 0; i < nDirs; i++) { rad[i * 2] = grid_dirs_deg[i * 2] * SAF_PI / 180.0f; rad[i * 2 + 1] = SAF_PI / 2.0f - grid_dirs_deg[i * 2 + 1] * SAF_PI / 180.0f; }
    h->svecs.resize((size_t)h->nSH * nDirs);
    getSHreal(order, rad.data(), nDirs, h->svecs.data());
    h->xyz.resize((size_t)nDirs * 3);
    for (int i = 0; i < nDirs; i++) {               /* unitSph2cart, degrees */
        const float az = grid_dirs_deg[i * 2] * SAF_PI / 180.0f, el = grid_dirs_deg[i * 2 + 1] * SAF_PI / 180.0f;
        h->xyz[i * 3] = cosf(el) * cosf(az); h->xyz[i * 3 + 1] = cosf(el) * sinf(az); h->xyz[i * 3 + 2] = sinf(el);
    }
    h->pSpec.resize(nDirs);
    return h;
}
static void sph_scan_peaks(SphScan* h, int nSrcs, int* peak_inds)
{
    const float kappa = 50.0f, scale = kappa / (2.0f * SAF_PI * expf(kappa) - expf(-kappa));
    std::vector<float> P(h->pSpec);
    for (int k = 0; k < nSrcs; k++) {
        int pk = 0;
        for (int i = 1; i < h->nDirs; i++) if (P[i] > P[pk]) pk = i;           /* utility_simaxv: index of the maximum */
        peak_inds[k] = pk;
        if (k == nSrcs - 1) break;
        const float* m = &h->xyz[(size_t)pk * 3];
        for (int i = 0; i < h->nDirs; i++) {
            float d = h->xyz[i * 3] * m[0] + h->xyz[i * 3 + 1] * m[1] + h->xyz[i * 3 + 2] * m[2];
            d = expf(d * kappa) * scale;                                         /* von Mises distribution around the peak */
            P[i] = P[i] * (1.0f / (0.00001f + d));                               /* ... inverted, as a mask */
        }
    }
}
void sphPWD_create(void** const phPWD, int order, float* grid_dirs_deg, int nDirs) { *phPWD = sph_scan_create(order, grid_dirs_deg, nDirs); }
void sphPWD_destroy(void** const phPWD) { delete (SphScan*)*phPWD; *phPWD = nullptr; }
void sphPWD_compute(void* const hPWD, float_complex* Cx, int nSrcs, float* P_map, int* peak_inds)      /* saf_sh.c:1109-1170 */
{
    SphScan* h = (SphScan*)hPWD;
    std::vector<float2> Y((size_t)h->nSH * h->nDirs);
    for (size_t i = 0; i < Y.size(); i++) Y[i] = make_float2(h->svecs[i], 0.0f);
    run_generate_map(1, h->order, Cx, reinterpret_cast<const float_complex*>(Y.data()), h->nDirs, 0.f, 0.f, 0, h->pSpec.data(), nullptr);
    if (P_map) memcpy(P_map, h->pSpec.data(), sizeof(float) * h->nDirs);
    if (peak_inds) sph_scan_peaks(h, nSrcs, peak_inds);
}
void sphMUSIC_create(void** const phMUSIC, int order, float* grid_dirs_deg, int nDirs) { *phMUSIC = sph_scan_create(order, grid_dirs_deg, nDirs); }
void sphMUSIC_destroy(void** const phMUSIC) { delete (SphScan*)*phMUSIC; *phMUSIC = nullptr; }
void sphMUSIC_compute(void* const hMUSIC, float_complex* Vn, int nSrcs, float* P_music, int* peak_inds)   /* saf_sh.c:1243-1306 */
{
    SphScan* h = (SphScan*)hMUSIC;
    const int nSH = h->nSH, G = h->nDirs, VnD2 = nSH - nSrcs;
    if (nSrcs < 1 || VnD2 < 1) SAF_FATAL("sphMUSIC_compute: 1 <= nSrcs < nSH");
    /* the noise-subspace basis goes where the sub-space kernel expects it: columns nSrcs .. nSH-1 of a [64][64] matrix */
    std::vector<float2> V((size_t)64 * 64, make_float2(0.f, 0.f));
    const float2* vn = reinterpret_cast<const float2*>(Vn);
    for (int i = 0; i < nSH; i++) for (int j = 0; j < VnD2; j++) V[(size_t)i * 64 + nSrcs + j] = vn[(size_t)i * VnD2 + j];
    DevBuf<float2> dV, dUn; DevBuf<float> dY, dP, dPrev; DevBuf<int> dSt;
    dV.alloc(V.size(), false); dUn.alloc(64); dY.alloc(h->svecs.size(), false); dP.alloc(G); dPrev.alloc(G); dSt.alloc(1, false);
    const int one = 1;
    HIP_CHECK(hipMemcpyAsync(dV.p, V.data(), sizeof(float2) * V.size(), hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(dY.p, h->svecs.data(), sizeof(float) * h->svecs.size(), hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(dSt.p, &one, sizeof(int), hipMemcpyHostToDevice, stream()));
    AdaptMapLaunch m{};
    m.Veig = dV.p; m.Un = dUn.p; m.status = dSt.p; m.Ygrid = dY.p; m.pmap = dP.p; m.prev_pmap = dPrev.p;
    m.nM = nSH; m.G = G; m.mode = 4; m.nSources = nSrcs; m.avg = 0.0f;
    launch_subspace_map(m);
    HIP_CHECK(hipMemcpyAsync(h->pSpec.data(), dP.p, sizeof(float) * G, hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
    if (P_music) memcpy(P_music, h->pSpec.data(), sizeof(float) * G);
    if (peak_inds) sph_scan_peaks(h, nSrcs, peak_inds);
}

}
