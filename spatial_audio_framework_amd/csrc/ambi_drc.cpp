/*
 * ambi_drc.cpp — the ambi_drc operator (examples/include/ambi_drc.h:97-270, examples/src/ambi_drc/ambi_drc.c): frequency-
 * dependent dynamic range compression of an Ambisonic scene, driven by the omni channel, with the block path on the GPU:
 *
 *   afSTFT analysis of the nSH channels -> [drc_gain_kernel: per band, gain computer + attack / release smoothing]
 *   -> [drc_apply_kernel: boost * gain * make-up on every channel] -> afSTFT synthesis      ambi_drc.c:161-222
 *
 * The gain factors of every time slot are also kept in the display ring of the reference (ambi_drc_getGainTF).
 * The reference holds no test for this operator ("parity unpinned"): tests/ compares with a CPU restatement.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "afstft_state.h"

namespace saf {

static int g_drc_frame_size = 128;                 /* default of the reference (ambi_drc_internal.h:59) */
#define DRC_DISPLAY_SLOTS ((int)(8 * 48000.0f / (float)128))      /* AMBI_DRC_NUM_DISPLAY_TIME_SLOTS (ambi_drc.h:67-69) */

struct AmbiDrc {
    int F, T;
    float fs = 48000.0f;
    float threshold, ratio, knee, inGain, outGain, attack_ms, release_ms;
    CH_ORDER chOrdering; NORM_TYPES norm; SH_ORDERS currentOrder;
    int nSH, new_nSH, reInitTFT;
    float freqVector[SAF_NBANDS];
    /* display ring */
    std::vector<float> bank[2]; std::vector<float*> bankRows[2];
    int wIdx = 0, rIdx = 0, storeIdx = 0;
    /* device side */
    bool haveSTFT = false;
    int Hmax = 0;
    AfState st;
    DevBuf<float2> X;                               /* [133][64][Hmax] */
    DevBuf<float> gains, yL, d_in, d_out;           /* [133][Hmax], [133] */
    PinBuf<float> h_in, h_out, h_g;
};

static void drc_init_tft(AmbiDrc* p, int maxFrames)     /* ambi_drc_initTFT (ambi_drc_internal.c:90-104) */
{
    ensure_device();
    const int Hneed = (p->T * maxFrames + 15) & ~15;
    if (!p->haveSTFT || Hneed > p->Hmax) {
        p->Hmax = Hneed;
        p->X.alloc((size_t)SAF_NBANDS * SAF_MAXCH * p->Hmax, true);
        p->gains.alloc((size_t)SAF_NBANDS * p->Hmax, true);
        if (!p->haveSTFT) { p->yL.alloc(SAF_NBANDS, true); p->st.create(1, p->new_nSH, p->new_nSH); }
        p->haveSTFT = true;
    }
    if (p->st.nCHin != p->new_nSH) { p->st.channelChange(p->new_nSH, p->new_nSH); p->st.clear(); }
    p->nSH = p->new_nSH;
}

static void drc_run(AmbiDrc* p, const float* in, long long in_frame, long long in_ch, int nIn, float* out, long long out_frame, long long out_ch, int nFrames)
{
    const int nSH = p->nSH, T = p->T, H = nFrames * T;
    AnaLaunch a{};
    a.in = in; a.in_inst = 0; a.in_ch = in_ch; a.in_frame = in_frame; a.hopsPerFrame = T; a.nChIn = nIn;
    a.hist_rd = p->st.ana[p->st.anaPar].p; a.hist_wr = p->st.ana[p->st.anaPar ^ 1].p;
    a.out = p->X.p; a.out_inst = 0; a.out_band = (long long)SAF_MAXCH * p->Hmax; a.out_ch = p->Hmax;
    a.ch_scale = nullptr; a.ch_map = nullptr; a.nCh = nSH; a.nInst = 1; a.H = H; a.lowDelay = 0; a.hybrid = 1;
    launch_analysis(a);
    p->st.anaPar ^= 1;
    DrcLaunch d{};
    d.X = p->X.p; d.x_band = a.out_band; d.x_ch = a.out_ch; d.gains = p->gains.p; d.g_band = p->Hmax; d.yL_z1 = p->yL.p;
    d.alpha_a = expf(-1.0f / ((p->attack_ms / ((float)p->F / (float)T)) * p->fs * 0.001f));       /* ambi_drc.c:147-153 */
    d.alpha_r = expf(-1.0f / ((p->release_ms / ((float)p->F / (float)T)) * p->fs * 0.001f));
    d.boost = powf(10.0f, p->inGain / 20.0f); d.makeup = powf(10.0f, p->outGain / 20.0f);
    d.threshold = p->threshold; d.ratio = p->ratio; d.knee = p->knee; d.floor = 0.1585f;          /* AMBI_DRC_SPECTRAL_FLOOR */
    d.nCh = nSH; d.H = H;
    launch_drc(d);
    SynLaunch s{};
    s.in = p->X.p; s.in_inst = 0; s.in_band = a.out_band; s.in_ch = a.out_ch;
    s.out = out; s.out_inst = 0; s.out_ch = out_ch; s.out_frame = out_frame; s.hopsPerFrame = T;
    s.hist_rd = p->st.syn[p->st.synPar].p; s.hist_wr = p->st.syn[p->st.synPar ^ 1].p;
    s.nCh = nSH; s.nInst = 1; s.H = H; s.lowDelay = 0; s.hybrid = 1;
    launch_synthesis(s);
    p->st.synPar ^= 1;
}

/* the call's gain factors into the display ring (ambi_drc.c:185-209) */
static void drc_store_display(AmbiDrc* p, int H)
{
    p->h_g.ensure((size_t)SAF_NBANDS * p->Hmax);
    HIP_CHECK(hipMemcpyAsync(p->h_g.p, p->gains.p, sizeof(float) * (size_t)SAF_NBANDS * p->Hmax, hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
    for (int t = 0; t < H; t++) {
        for (int band = 0; band < SAF_NBANDS; band++) p->bank[p->storeIdx][(size_t)band * DRC_DISPLAY_SLOTS + p->wIdx] = p->h_g.p[(size_t)band * p->Hmax + t];
        p->wIdx++; p->rIdx++;
        if (p->wIdx >= DRC_DISPLAY_SLOTS) { p->wIdx = 0; p->storeIdx = p->storeIdx == 0 ? 1 : 0; }
        if (p->rIdx >= DRC_DISPLAY_SLOTS) p->rIdx = 0;
    }
}

}  // namespace saf

using namespace saf;

extern "C" {

void saf_hip_ambi_drc_setFrameSize(int frameSize)
{
    if (frameSize <= 0 || frameSize % SAF_HOP != 0) SAF_FATAL("ambi_drc frame size must be a positive multiple of 128");
    g_drc_frame_size = frameSize;
}

#define PD AmbiDrc* p = (AmbiDrc*)hAmbi
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

void ambi_drc_create(void** const phAmbi)          /* ambi_drc.c:43-80 */
{
    AmbiDrc* p = new AmbiDrc();
    *phAmbi = p;
    p->F = g_drc_frame_size; p->T = p->F / SAF_HOP;
    for (int b = 0; b < 2; b++) {
        p->bank[b].assign((size_t)SAF_NBANDS * DRC_DISPLAY_SLOTS, 0.0f);
        p->bankRows[b].resize(SAF_NBANDS);
        for (int band = 0; band < SAF_NBANDS; band++) p->bankRows[b][band] = p->bank[b].data() + (size_t)band * DRC_DISPLAY_SLOTS;
    }
    p->threshold = 0.0f; p->ratio = 8.0f; p->knee = 0.0f; p->inGain = 0.0f; p->outGain = 0.0f; p->attack_ms = 50.0f; p->release_ms = 100.0f;
    p->chOrdering = CH_ACN; p->norm = NORM_SN3D; p->currentOrder = SH_ORDER_FIRST;
    p->new_nSH = p->nSH = 4;
    p->reInitTFT = 1;
    memset(p->freqVector, 0, sizeof(p->freqVector));
}

void ambi_drc_destroy(void** const phAmbi)
{
    AmbiDrc* p = (AmbiDrc*)*phAmbi;
    if (!p) return;
    if (p->haveSTFT) HIP_CHECK(hipStreamSynchronize(stream()));
    delete p;
    *phAmbi = nullptr;
}

void ambi_drc_init(void* const hAmbi, int sampleRate)      /* ambi_drc.c:104-134 */
{
    PD;
    p->fs = (float)sampleRate;
    if (p->haveSTFT) p->yL.zero();
    /* the reference passes its afSTFT handle, NULL before the first initTFT: the two centre-frequency tables of a5 */
    if (!p->haveSTFT) afSTFT_getCentreFreqs(nullptr, (float)sampleRate, SAF_NBANDS, p->freqVector);
    else {
        void* h = nullptr;
        afSTFT_create(&h, 1, 1, SAF_HOP, 0, 1, AFSTFT_BANDS_CH_TIME);
        afSTFT_getCentreFreqs(h, (float)sampleRate, SAF_NBANDS, p->freqVector);
        afSTFT_destroy(&h);
    }
    p->rIdx = 0; p->wIdx = 1; p->storeIdx = 0;                          /* ambi_drc.c:118-120 */
    for (int b = 0; b < 2; b++) std::fill(p->bank[b].begin(), p->bank[b].end(), 0.0f);
    if (p->reInitTFT == 1) { p->reInitTFT = 2; drc_init_tft(p, 1); p->reInitTFT = 0; }
}

void ambi_drc_process(void* const hAmbi, const float* const* inputs, float** const outputs, int nCh, int nSamples)
{
    PD;
    const int F = p->F;
    if (p->reInitTFT == 1) { p->reInitTFT = 2; drc_init_tft(p, 1); p->reInitTFT = 0; }
    if (nSamples != F || p->reInitTFT != 0) {                         /* ambi_drc.c:224-227 */
        for (int ch = 0; ch < nCh; ch++) memset(outputs[ch], 0, sizeof(float) * F);
        return;
    }
    const int nSH = p->nSH, nIn = nSH < nCh ? nSH : (nCh < 0 ? 0 : nCh);
    p->h_in.ensure((size_t)SAF_MAXCH * F); p->h_out.ensure((size_t)SAF_MAXCH * F);
    if (p->d_in.n < (size_t)SAF_MAXCH * F) { p->d_in.alloc((size_t)SAF_MAXCH * F, true); p->d_out.alloc((size_t)SAF_MAXCH * F, true); }
    for (int i = 0; i < nIn; i++) memcpy(p->h_in.p + (size_t)i * F, inputs[i], sizeof(float) * F);
    for (int i = nIn; i < nSH; i++) memset(p->h_in.p + (size_t)i * F, 0, sizeof(float) * F);
    if (zero_copy_io()) drc_run(p, p->h_in.p, 0, F, nSH, p->h_out.p, 0, F, 1);
    else {
        HIP_CHECK(hipMemcpyAsync(p->d_in.p, p->h_in.p, sizeof(float) * (size_t)nSH * F, hipMemcpyHostToDevice, stream()));
        drc_run(p, p->d_in.p, 0, F, nSH, p->d_out.p, 0, F, 1);
        HIP_CHECK(hipMemcpyAsync(p->h_out.p, p->d_out.p, sizeof(float) * (size_t)nSH * F, hipMemcpyDeviceToHost, stream()));
    }
    drc_store_display(p, p->T);                                        /* synchronises the stream */
    for (int ch = 0; ch < nIn; ch++) memcpy(outputs[ch], p->h_out.p + (size_t)ch * F, sizeof(float) * F);
    for (int ch = nIn; ch < nCh; ch++) memset(outputs[ch], 0, sizeof(float) * F);
}

void saf_hip_ambi_drc_process_dev(void* const hAmbi, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                  float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames)
{
    PD;
    if (nFrames <= 0) return;
    if (p->reInitTFT == 1 || !p->haveSTFT || p->T * nFrames > p->Hmax) { p->reInitTFT = 2; drc_init_tft(p, nFrames); p->reInitTFT = 0; }
    drc_run(p, d_in, in_frame_stride, in_ch_stride, nInputs < p->nSH ? nInputs : p->nSH, d_out, out_frame_stride, out_ch_stride, nFrames);
}

void ambi_drc_refreshSettings(void* const hAmbi) { PD; p->reInitTFT = 1; }
void ambi_drc_setThreshold(void* const hAmbi, float v) { PD; p->threshold = clampf(v, -60.0f, 0.0f); }
void ambi_drc_setRatio(void* const hAmbi, float v) { PD; p->ratio = clampf(v, 1.0f, 30.0f); }
void ambi_drc_setKnee(void* const hAmbi, float v) { PD; p->knee = clampf(v, 0.0f, 10.0f); }
void ambi_drc_setInGain(void* const hAmbi, float v) { PD; p->inGain = clampf(v, -40.0f, 20.0f); }
void ambi_drc_setOutGain(void* const hAmbi, float v) { PD; p->outGain = clampf(v, -20.0f, 40.0f); }
void ambi_drc_setAttack(void* const hAmbi, float v) { PD; p->attack_ms = clampf(v, 10.0f, 200.0f); }
void ambi_drc_setRelease(void* const hAmbi, float v) { PD; p->release_ms = clampf(v, 50.0f, 1000.0f); }
void ambi_drc_setChOrder(void* const hAmbi, int o) { PD; if ((CH_ORDER)o != CH_FUMA || p->currentOrder == SH_ORDER_FIRST) p->chOrdering = (CH_ORDER)o; }
void ambi_drc_setNormType(void* const hAmbi, int t) { PD; if ((NORM_TYPES)t != NORM_FUMA || p->currentOrder == SH_ORDER_FIRST) p->norm = (NORM_TYPES)t; }
void ambi_drc_setInputPreset(void* const hAmbi, SH_ORDERS newPreset)
{
    PD;
    const int o = (int)newPreset < 1 ? 1 : ((int)newPreset > SAF_MAX_ORDER ? SAF_MAX_ORDER : (int)newPreset);
    p->new_nSH = ORDER2NSH(o);
    p->currentOrder = (SH_ORDERS)o;
    if (p->new_nSH != p->nSH) p->reInitTFT = 1;
    if (p->currentOrder != SH_ORDER_FIRST && p->chOrdering == CH_FUMA) p->chOrdering = CH_ACN;
    if (p->currentOrder != SH_ORDER_FIRST && p->norm == NORM_FUMA) p->norm = NORM_SN3D;
}
int ambi_drc_getFrameSize(void) { return g_drc_frame_size; }
float** ambi_drc_getGainTF(void* const hAmbi) { PD; return p->bankRows[p->storeIdx].data(); }
int ambi_drc_getGainTFwIdx(void* const hAmbi) { PD; return p->wIdx; }
int ambi_drc_getGainTFrIdx(void* const hAmbi) { PD; return p->rIdx; }
float* ambi_drc_getFreqVector(void* const hAmbi, int* nFreqPoints) { PD; *nFreqPoints = SAF_NBANDS; return p->freqVector; }
float ambi_drc_getThreshold(void* const hAmbi) { PD; return p->threshold; }
float ambi_drc_getRatio(void* const hAmbi) { PD; return p->ratio; }
float ambi_drc_getKnee(void* const hAmbi) { PD; return p->knee; }
float ambi_drc_getInGain(void* const hAmbi) { PD; return p->inGain; }
float ambi_drc_getOutGain(void* const hAmbi) { PD; return p->outGain; }
float ambi_drc_getAttack(void* const hAmbi) { PD; return p->attack_ms; }
float ambi_drc_getRelease(void* const hAmbi) { PD; return p->release_ms; }
int ambi_drc_getChOrder(void* const hAmbi) { PD; return (int)p->chOrdering; }
int ambi_drc_getNormType(void* const hAmbi) { PD; return (int)p->norm; }
SH_ORDERS ambi_drc_getInputPreset(void* const hAmbi) { PD; return p->currentOrder; }
int ambi_drc_getNSHrequired(void* const hAmbi) { PD; return p->nSH; }
int ambi_drc_getSamplerate(void* const hAmbi) { PD; return (int)(p->fs + 0.5f); }
int ambi_drc_getProcessingDelay(void) { return 12 * SAF_HOP; }

}
