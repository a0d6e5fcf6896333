/*
 * ambi_bin.cpp — the binaural Ambisonic decoder (examples/include/ambi_bin.h:161-482, examples/src/ambi_bin/ambi_bin.c) with
 * its per-block path on the GPU:
 *
 *   SH inputs -> [afSTFT analysis, channel/normalisation conventions folded in] -> [band MAC with the 2 x nSH decoder,
 *   sound-field rotation baked into it]                                         -> [afSTFT synthesis, 2 ears] -> outputs
 *
 * Init (ambi_bin_initCodec, ambi_bin.c:167-378): ITDs, HRIR -> filterbank coefficients (GPU analysis), Voronoi weights,
 * diffuse-field EQ / phase simplification, decoder design (binaural_design.cpp), truncation EQ.
 * The HRIR set is the one installed with saf_hip_setDefaultHRIRs (the reference's default set is absent from its checkout).
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "afstft_state.h"
#include "binaural_design.h"
#include "hrir_host.h"
#include <complex>
#include <thread>
#include <chrono>

namespace saf {

static int g_abin_frame_size = 128;      /* default of the reference (ambi_bin_internal.h) */
static inline void asleep_ms(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }

struct AmbiBin {
    int F, T, fs = 0;
    float freqVector[SAF_NBANDS];
    bool haveSTFT = false;
    /* codec parameters (ambi_bin_codecPars) */
    std::string sofa_filepath;
    std::vector<float> hrirs, hrir_dirs_deg, itds_s, weights;
    std::vector<float2> hrtf_fb;
    int N_hrir_dirs = 0, hrir_len = 0, hrir_fs = 0;
    std::vector<std::complex<float>> M_dec;                 /* [133][2][64]; M_dec_rot lives on the device only (d_decRot) */
    volatile CODEC_STATUS codecStatus;
    volatile PROC_STATUS procStatus;
    float progressBar0_1 = 0.0f; char progressBarText[PROGRESSBARTEXT_CHAR_LENGTH];
    int reinit_hrtfsFLAG = 1, recalc_M_rotFLAG = 1;
    /* user parameters */
    int order = 1, new_order = 1, nSH = 4;
    int useDefaultHRIRsFLAG = 1, preProc = HRIR_PREPROC_EQ, enableMaxRE = 1, enableDiffuseMatching = 0, enableTruncationEQ = 1, enableRotation = 0;
    int method = DECODING_METHOD_MAGLS, useRollPitchYawFlag = 0, bFlipYaw = 0, bFlipPitch = 0, bFlipRoll = 0;
    CH_ORDER chOrdering = CH_ACN; NORM_TYPES norm = NORM_SN3D;
    float yaw = 0.0f, pitch = 0.0f, roll = 0.0f;
    /* device side */
    AfState st;
    int Hmax = 0;
    DevBuf<float2> X, Y, d_dec /* M_dec, MAC layout */, d_decRot /* M_dec_rot, MAC layout */, d_MdecBase /* [133][2][64] */;
    DevBuf<float> d_Mrot; PinBuf<float> stR;
    bool stagingBusy = false;               /* stD / stR may still be read by a copy enqueued by a device-entry call */
    DevBuf<float> d_scale, d_in, d_out; DevBuf<int> d_map;
    PinBuf<float> stS, h_in, h_out; PinBuf<int> stM; PinBuf<float2> stD;
    bool decDirty = true;
    int shNorm = -1, shOrd = -1, shOrder = -1;
};

static void set_codec_status(AmbiBin* p, CODEC_STATUS s)
{
    if (s == CODEC_STATUS_NOT_INITIALISED) while (p->codecStatus == CODEC_STATUS_INITIALISING) asleep_ms(10);
    p->codecStatus = s;
}

/* the block path for nFrames consecutive blocks of device-resident samples */
static void process_dev(AmbiBin* p, const float* d_in, long long in_frame, long long in_ch, int nIn, float* d_out, long long out_frame, long long out_ch, int nFrames)
{
    const int order = p->order, nSH = ORDER2NSH(order), T = p->T, H = nFrames * T;
    if (H > p->Hmax) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->Hmax = (H + 15) & ~15;
        p->X.alloc((size_t)SAF_NBANDS * SAF_MAXCH * p->Hmax, true);
        p->Y.alloc((size_t)SAF_NBANDS * 2 * p->Hmax, true);
    }
    /* input conventions -> ACN/N3D (ambi_bin.c:419-431, saf_hoa.c:40-116) as the analysis kernel's gather map + row scale */
    if (p->shNorm != (int)p->norm || p->shOrd != (int)p->chOrdering || p->shOrder != order) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        int* map = p->stM.p; float* sc = p->stS.p;
        for (int ch = 0; ch < SAF_MAXCH; ch++) { map[ch] = ch; sc[ch] = 1.0f; }
        if (p->chOrdering == CH_FUMA) { map[1] = 2; map[2] = 3; map[3] = 1; for (int ch = 4; ch < SAF_MAXCH; ch++) map[ch] = -1; }
        if (p->norm == NORM_SN3D) { for (int n = 0; n <= order; n++) for (int ch = n * n; ch < ORDER2NSH(n); ch++) sc[ch] = sqrtf(2.0f * (float)n + 1.0f); }
        else if (p->norm == NORM_FUMA) { sc[0] = sqrtf(2.0f); for (int ch = 1; ch < 4; ch++) sc[ch] = sqrtf(3.0f); }
        HIP_CHECK(hipMemcpyAsync(p->d_map.p, map, sizeof(int) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(p->d_scale.p, sc, sizeof(float) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->shNorm = (int)p->norm; p->shOrd = (int)p->chOrdering; p->shOrder = order;
    }
    AnaLaunch a{};
    a.in = d_in; a.in_inst = 0; a.in_ch = in_ch; a.in_frame = in_frame; a.hopsPerFrame = T; a.nChIn = nIn;
    a.hist_rd = p->st.ana[p->st.anaPar].p; a.hist_wr = p->st.ana[p->st.anaPar ^ 1].p;
    a.out = p->X.p; a.out_inst = 0; a.out_band = (long long)SAF_MAXCH * p->Hmax; a.out_ch = p->Hmax;
    a.ch_scale = p->d_scale.p; a.ch_map = p->d_map.p; a.tab_stride = SAF_MAXCH;
    a.nCh = nSH; a.nInst = 1; a.H = H; a.lowDelay = 0; a.hybrid = 1;
    launch_analysis(a);
    p->st.anaPar ^= 1;

    /* rotation baked into the decoder when flagged (ambi_bin.c:437-456) */
    const int useRot = order > 0 && p->enableRotation ? 1 : 0;
    if (p->decDirty) {                                       /* a new decoder: its rows to the device, M_dec in the MAC's layout */
        if (p->stagingBusy) { HIP_CHECK(hipStreamSynchronize(stream())); p->stagingBusy = false; }
        for (size_t i = 0; i < p->M_dec.size(); i++) p->stD.p[i] = make_float2(p->M_dec[i].real(), p->M_dec[i].imag());
        HIP_CHECK(hipMemcpyAsync(p->d_MdecBase.p, p->stD.p, sizeof(float2) * p->M_dec.size(), hipMemcpyHostToDevice, stream()));
        DecRotLaunch r{}; r.Mdec = p->d_MdecBase.p; r.Mrot = nullptr; r.out = p->d_dec.p; r.nSH = nSH;
        launch_dec_rotate(r);
        p->decDirty = false; p->stagingBusy = true;
    }
    if (useRot && p->recalc_M_rotFLAG) {
        float R[3][3];
        yaw_pitch_roll_to_Rzyx(p->yaw, p->pitch, p->roll, p->useRollPitchYawFlag, R);
        if (p->stagingBusy) { HIP_CHECK(hipStreamSynchronize(stream())); p->stagingBusy = false; }
        sh_rot_matrix_real(R, p->stR.p, order);
        HIP_CHECK(hipMemcpyAsync(p->d_Mrot.p, p->stR.p, sizeof(float) * (size_t)nSH * nSH, hipMemcpyHostToDevice, stream()));
        DecRotLaunch r{}; r.Mdec = p->d_MdecBase.p; r.Mrot = p->d_Mrot.p; r.out = p->d_decRot.p; r.nSH = nSH;
        launch_dec_rotate(r);
        p->recalc_M_rotFLAG = 0; p->stagingBusy = true;
    }
    /* (with rotation enabled the reference multiplies by M_dec_rot whatever its age, ambi_bin.c:459-464) */
    const int wantRot = p->enableRotation ? 1 : 0;
    BinMacLaunch m{};
    m.X = p->X.p; m.x_band = a.out_band; m.x_ch = a.out_ch;
    m.h = wantRot ? p->d_decRot.p : p->d_dec.p;
    m.Y = p->Y.p; m.y_band = (long long)2 * p->Hmax; m.y_ch = p->Hmax;
    m.nSrc = nSH; m.H = H; m.scale = 1.0f;
    launch_binaural_mac(m);

    SynLaunch s{};
    s.in = p->Y.p; s.in_inst = 0; s.in_band = m.y_band; s.in_ch = m.y_ch;
    s.out = d_out; s.out_inst = 0; s.out_ch = out_ch; s.out_frame = out_frame; s.hopsPerFrame = T;
    s.hist_rd = p->st.syn[p->st.synPar].p; s.hist_wr = p->st.syn[p->st.synPar ^ 1].p;
    s.nCh = 2; s.nInst = 1; s.H = H; s.lowDelay = 0; s.hybrid = 1;
    launch_synthesis(s);
    p->st.synPar ^= 1;
}

}  // namespace saf

using namespace saf;

extern "C" {

void saf_hip_ambi_bin_setFrameSize(int frameSize)
{
    if (frameSize <= 0 || frameSize % SAF_HOP != 0) SAF_FATAL("ambi_bin frame size must be a positive multiple of 128");
    g_abin_frame_size = frameSize;
}

void ambi_bin_create(void** const phAmbi)        /* ambi_bin.c:48-109 */
{
    AmbiBin* p = new AmbiBin();
    *phAmbi = p;
    p->F = g_abin_frame_size; p->T = p->F / SAF_HOP;
    p->M_dec.assign((size_t)SAF_NBANDS * 2 * SAF_MAXCH, 0.0f);
    p->codecStatus = CODEC_STATUS_NOT_INITIALISED; p->procStatus = PROC_STATUS_NOT_ONGOING;
    p->progressBarText[0] = 0;
    memset(p->freqVector, 0, sizeof(p->freqVector));
}

void ambi_bin_destroy(void** const phAmbi)
{
    AmbiBin* p = (AmbiBin*)*phAmbi;
    if (!p) return;
    while (p->codecStatus == CODEC_STATUS_INITIALISING || p->procStatus == PROC_STATUS_ONGOING) asleep_ms(10);
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete p;
    *phAmbi = nullptr;
}

void ambi_bin_init(void* const hAmbi, int sampleRate)      /* ambi_bin.c:147-165 */
{
    AmbiBin* p = (AmbiBin*)hAmbi;
    if (p->fs != sampleRate) { p->fs = sampleRate; p->reinit_hrtfsFLAG = 1; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
    if (!p->haveSTFT) afSTFT_getCentreFreqs(nullptr, (float)p->fs, SAF_NBANDS, p->freqVector);
    else {
        static const float w[9] = { 1.0f, 0.7501f, 1.2499f, 0.8751f, 1.1249f, 0.9167f, 1.0833f, 0.9375f, 1.0625f };
        static const int bin[9] = { 0, 1, 1, 2, 2, 3, 3, 4, 4 };
        for (int i = 0; i < 9; i++) p->freqVector[i] = w[i] * ((float)bin[i] * (float)p->fs / 256.0f);
        for (int i = 9, j = 5; i < SAF_NBANDS; i++, j++) p->freqVector[i] = (float)j * (float)p->fs / 256.0f;
    }
    p->recalc_M_rotFLAG = 1;
}

void ambi_bin_initCodec(void* const hAmbi)                  /* ambi_bin.c:167-378 */
{
    AmbiBin* p = (AmbiBin*)hAmbi;
    if (p->codecStatus != CODEC_STATUS_NOT_INITIALISED) return;
    while (p->procStatus == PROC_STATUS_ONGOING) { p->codecStatus = CODEC_STATUS_INITIALISING; asleep_ms(10); }
    ensure_device();
    p->codecStatus = CODEC_STATUS_INITIALISING;
    strcpy(p->progressBarText, "Preparing HRIRs"); p->progressBar0_1 = 0.0f;
    const int order = p->new_order, nSH = ORDER2NSH(order);
    HIP_CHECK(hipStreamSynchronize(stream()));
    if (!p->haveSTFT) {
        p->st.create(1, nSH, 2);
        p->d_dec.alloc((size_t)SAF_MAXCH * SAF_NBANDS * 2); p->d_decRot.alloc((size_t)SAF_MAXCH * SAF_NBANDS * 2); p->d_MdecBase.alloc((size_t)SAF_NBANDS * 2 * SAF_MAXCH);
        p->d_Mrot.alloc(64 * 64); p->stR.ensure(64 * 64); p->d_scale.alloc(SAF_MAXCH); p->d_map.alloc(SAF_MAXCH);
        p->stS.ensure(SAF_MAXCH); p->stM.ensure(SAF_MAXCH); p->stD.ensure((size_t)SAF_MAXCH * SAF_NBANDS * 2);
        p->haveSTFT = true;
    } else if (p->nSH != nSH) { p->st.channelChange(nSH, 2); p->st.clear(); }
    p->nSH = nSH;

    if (p->reinit_hrtfsFLAG) {
        const DefaultHRIRs& D = default_hrirs();
        if (D.N == 0)
            SAF_FATAL("ambi_bin: no HRIR set installed.  The reference's default set (saf_default_hrirs.c) is not part of its checkout and "
                      "SOFA loading is outside this library: call saf_hip_setDefaultHRIRs() before ambi_bin_initCodec().");
        p->useDefaultHRIRsFLAG = 1;
        p->hrir_fs = D.fs; p->hrir_len = D.len; p->N_hrir_dirs = D.N; p->hrirs = D.hrirs; p->hrir_dirs_deg = D.dirs_deg;
        const int N = D.N;
        p->progressBar0_1 = 0.3f;
        p->itds_s.resize(N);
        estimateITDs(p->hrirs.data(), N, p->hrir_len, p->hrir_fs, p->itds_s.data());
        p->progressBar0_1 = 0.4f;
        p->hrtf_fb.resize((size_t)SAF_NBANDS * 2 * N);
        HRIRs2HRTFs_afSTFT(p->hrirs.data(), N, p->hrir_len, SAF_HOP, 0, 1, reinterpret_cast<float_complex*>(p->hrtf_fb.data()));
        p->progressBar0_1 = 0.6f;
        if (N <= 1000) { p->weights.resize(N); voronoi_weights(p->hrir_dirs_deg.data(), N, p->weights.data()); } else p->weights.clear();
        p->progressBar0_1 = 0.75f;
        diffuseFieldEqualiseHRTFs(N, p->itds_s.data(), p->freqVector, SAF_NBANDS, p->weights.empty() ? nullptr : p->weights.data(),
                                  p->preProc == HRIR_PREPROC_EQ || p->preProc == HRIR_PREPROC_ALL ? 1 : 0,
                                  p->preProc == HRIR_PREPROC_PHASE || p->preProc == HRIR_PREPROC_ALL ? 1 : 0,
                                  reinterpret_cast<float_complex*>(p->hrtf_fb.data()));
        p->reinit_hrtfsFLAG = 0;
    }
    strcpy(p->progressBarText, "Computing Decoder"); p->progressBar0_1 = 0.95f;
    const int N = p->N_hrir_dirs;
    std::vector<std::complex<float>> dec((size_t)SAF_NBANDS * 2 * nSH);
    BINAURAL_AMBI_DECODER_METHODS bm = BINAURAL_DECODER_LS;
    switch (p->method) {
        default: case DECODING_METHOD_LS: bm = BINAURAL_DECODER_LS; break;
        case DECODING_METHOD_LSDIFFEQ: bm = BINAURAL_DECODER_LSDIFFEQ; break;
        case DECODING_METHOD_SPR: bm = BINAURAL_DECODER_SPR; break;
        case DECODING_METHOD_TA: bm = BINAURAL_DECODER_TA; break;
        case DECODING_METHOD_MAGLS: bm = BINAURAL_DECODER_MAGLS; break;
    }
    getBinauralAmbiDecoderMtx(reinterpret_cast<float_complex*>(p->hrtf_fb.data()), p->hrir_dirs_deg.data(), N, SAF_NBANDS, bm, order, p->freqVector,
                              p->itds_s.data(), p->weights.empty() ? nullptr : p->weights.data(), p->enableDiffuseMatching, p->enableMaxRE,
                              reinterpret_cast<float_complex*>(dec.data()));
    /* truncation EQ (ambi_bin.c:311-364) */
    if (p->enableTruncationEQ && p->method == DECODING_METHOD_LS && p->preProc != HRIR_PREPROC_PHASE && p->preProc != HRIR_PREPROC_ALL) {
        std::vector<double> kr(SAF_NBANDS);
        for (int k = 0; k < SAF_NBANDS; k++) kr[k] = 2.0 * SAF_PId / 343.0 * (double)p->freqVector[k] * 0.085;
        std::vector<float> w_n(order + 1, 1.0f), eq(SAF_NBANDS);
        if (p->enableMaxRE) {
            std::vector<float> c(order + 1);
            beamWeightsMaxEV(order, c.data());
            for (int n = 0; n <= order; n++) w_n[n] = c[n] / sqrtf((float)(2 * n + 1) / (4.0f * SAF_PI));
            const float w0 = w_n[0];
            for (int n = 0; n <= order; n++) w_n[n] /= w0;
        }
        truncationEQ(w_n.data(), order, 42, kr.data(), SAF_NBANDS, 9.0f, eq.data());
        for (int b = 0; b < SAF_NBANDS; b++) for (int i = 0; i < 2 * nSH; i++) dec[(size_t)b * 2 * nSH + i] *= eq[b];
    }
    std::fill(p->M_dec.begin(), p->M_dec.end(), std::complex<float>(0.0f, 0.0f));
    for (int b = 0; b < SAF_NBANDS; b++) for (int e = 0; e < 2; e++) for (int j = 0; j < nSH; j++) p->M_dec[((size_t)b * 2 + e) * SAF_MAXCH + j] = dec[((size_t)b * 2 + e) * nSH + j];
    p->decDirty = true;
    p->order = order;
    strcpy(p->progressBarText, "Done!"); p->progressBar0_1 = 1.0f;
    p->codecStatus = CODEC_STATUS_INITIALISED;
}

void ambi_bin_process(void* const hAmbi, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)   /* ambi_bin.c:380-480 */
{
    AmbiBin* p = (AmbiBin*)hAmbi;
    const int F = p->F;
    if (nSamples == F && p->codecStatus == CODEC_STATUS_INITIALISED) {
        p->procStatus = PROC_STATUS_ONGOING;
        const int nSH = ORDER2NSH(p->order);
        const int nRows = nSH < 4 ? 4 : nSH;                            /* a FuMa gather may read rows up to 3 */
        p->h_in.ensure((size_t)SAF_MAXCH * F); p->h_out.ensure((size_t)2 * F);
        if (p->d_in.n < (size_t)SAF_MAXCH * F) { p->d_in.alloc((size_t)SAF_MAXCH * F, true); p->d_out.alloc((size_t)2 * F, true); }
        int i;
        for (i = 0; i < (nSH < nInputs ? nSH : nInputs); i++) memcpy(p->h_in.p + (size_t)i * F, inputs[i], sizeof(float) * F);
        for (; i < nRows; i++) memset(p->h_in.p + (size_t)i * F, 0, sizeof(float) * F);
        if (zero_copy_io()) process_dev(p, p->h_in.p, 0, F, nRows, p->h_out.p, 0, F, 1);                  /* kernels on the pinned blocks */
        else {
            HIP_CHECK(hipMemcpyAsync(p->d_in.p, p->h_in.p, sizeof(float) * (size_t)nRows * F, hipMemcpyHostToDevice, stream()));
            process_dev(p, p->d_in.p, 0, F, nRows, p->d_out.p, 0, F, 1);
            HIP_CHECK(hipMemcpyAsync(p->h_out.p, p->d_out.p, sizeof(float) * (size_t)2 * F, hipMemcpyDeviceToHost, stream()));
        }
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->stagingBusy = false;
        int ch;
        for (ch = 0; ch < (2 < nOutputs ? 2 : nOutputs); ch++) memcpy(outputs[ch], p->h_out.p + (size_t)ch * F, sizeof(float) * F);
        for (; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    } else
        for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}

void saf_hip_ambi_bin_process_dev(void* const hAmbi, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                  float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames)
{
    AmbiBin* p = (AmbiBin*)hAmbi;
    if (p->codecStatus != CODEC_STATUS_INITIALISED) SAF_FATAL("ambi_bin: process_dev on a handle that is not initialised (call ambi_bin_initCodec)");
    p->procStatus = PROC_STATUS_ONGOING;
    process_dev(p, d_in, in_frame_stride, in_ch_stride, nInputs < 0 ? 0 : nInputs, d_out, out_frame_stride, out_ch_stride, nFrames);
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}
void saf_hip_ambi_bin_getDecoderMtx(void* const hAmbi, float_complex* M /* [133][2][nSH] */)
{
    AmbiBin* p = (AmbiBin*)hAmbi;
    const int nSH = ORDER2NSH(p->order);
    std::complex<float>* o = reinterpret_cast<std::complex<float>*>(M);
    for (int b = 0; b < SAF_NBANDS; b++) for (int e = 0; e < 2; e++) for (int j = 0; j < nSH; j++) o[((size_t)b * 2 + e) * nSH + j] = p->M_dec[((size_t)b * 2 + e) * SAF_MAXCH + j];
}

/* ------------------------------- set / get functions (ambi_bin.c:485-824) ------------------------------- */
#define PAB AmbiBin* p = (AmbiBin*)hAmbi
void ambi_bin_refreshParams(void* const hAmbi) { PAB; p->reinit_hrtfsFLAG = 1; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
void ambi_bin_setUseDefaultHRIRsflag(void* const hAmbi, int s) { PAB; if (!p->useDefaultHRIRsFLAG && s) { p->useDefaultHRIRsFLAG = s; ambi_bin_refreshParams(hAmbi); } }
void ambi_bin_setSofaFilePath(void* const hAmbi, const char* path) { PAB; p->sofa_filepath = path; p->useDefaultHRIRsFLAG = 0; ambi_bin_refreshParams(hAmbi); }
void ambi_bin_setInputOrderPreset(void* const hAmbi, SH_ORDERS newOrder)
{
    PAB;
    if (p->order != (int)newOrder) { p->new_order = (int)newOrder; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
    if (p->new_order != SH_ORDER_FIRST && p->chOrdering == CH_FUMA) p->chOrdering = CH_ACN;
    if (p->new_order != SH_ORDER_FIRST && p->norm == NORM_FUMA) p->norm = NORM_SN3D;
}
void ambi_bin_setDecodingMethod(void* const hAmbi, AMBI_BIN_DECODING_METHODS m) { PAB; p->method = (int)m; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
void ambi_bin_setChOrder(void* const hAmbi, int v) { PAB; if ((CH_ORDER)v != CH_FUMA || p->new_order == SH_ORDER_FIRST) p->chOrdering = (CH_ORDER)v; }
void ambi_bin_setNormType(void* const hAmbi, int v) { PAB; if ((NORM_TYPES)v != NORM_FUMA || p->new_order == SH_ORDER_FIRST) p->norm = (NORM_TYPES)v; }
void ambi_bin_setEnableMaxRE(void* const hAmbi, int s) { PAB; if (p->enableMaxRE != s) { p->enableMaxRE = s; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); } }
void ambi_bin_setEnableDiffuseMatching(void* const hAmbi, int s) { PAB; if (p->enableDiffuseMatching != s) { p->enableDiffuseMatching = s; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); } }
void ambi_bin_setEnableTruncationEQ(void* const hAmbi, int s) { PAB; if (p->enableTruncationEQ != s) { p->enableTruncationEQ = s; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); } }
void ambi_bin_setHRIRsPreProc(void* const hAmbi, AMBI_BIN_PREPROC t) { PAB; if (p->preProc != (int)t) { p->preProc = (int)t; ambi_bin_refreshParams(hAmbi); } }
void ambi_bin_setEnableRotation(void* const hAmbi, int s) { PAB; p->enableRotation = s; }
void ambi_bin_setYaw(void* const hAmbi, float v) { PAB; p->yaw = p->bFlipYaw == 1 ? -(v * SAF_PI / 180.0f) : v * SAF_PI / 180.0f; p->recalc_M_rotFLAG = 1; }
void ambi_bin_setPitch(void* const hAmbi, float v) { PAB; p->pitch = p->bFlipPitch == 1 ? -(v * SAF_PI / 180.0f) : v * SAF_PI / 180.0f; p->recalc_M_rotFLAG = 1; }
void ambi_bin_setRoll(void* const hAmbi, float v) { PAB; p->roll = p->bFlipRoll == 1 ? -(v * SAF_PI / 180.0f) : v * SAF_PI / 180.0f; p->recalc_M_rotFLAG = 1; }
float ambi_bin_getYaw(void* const hAmbi) { PAB; return p->bFlipYaw ? -(p->yaw * 180.0f / SAF_PI) : p->yaw * 180.0f / SAF_PI; }
float ambi_bin_getPitch(void* const hAmbi) { PAB; return p->bFlipPitch ? -(p->pitch * 180.0f / SAF_PI) : p->pitch * 180.0f / SAF_PI; }
float ambi_bin_getRoll(void* const hAmbi) { PAB; return p->bFlipRoll ? -(p->roll * 180.0f / SAF_PI) : p->roll * 180.0f / SAF_PI; }
void ambi_bin_setFlipYaw(void* const hAmbi, int s) { PAB; if (s != p->bFlipYaw) { p->bFlipYaw = s; ambi_bin_setYaw(hAmbi, -ambi_bin_getYaw(hAmbi)); } }
void ambi_bin_setFlipPitch(void* const hAmbi, int s) { PAB; if (s != p->bFlipPitch) { p->bFlipPitch = s; ambi_bin_setPitch(hAmbi, -ambi_bin_getPitch(hAmbi)); } }
void ambi_bin_setFlipRoll(void* const hAmbi, int s) { PAB; if (s != p->bFlipRoll) { p->bFlipRoll = s; ambi_bin_setRoll(hAmbi, -ambi_bin_getRoll(hAmbi)); } }
void ambi_bin_setRPYflag(void* const hAmbi, int s) { PAB; p->useRollPitchYawFlag = s; }
int ambi_bin_getFrameSize(void) { return g_abin_frame_size; }
CODEC_STATUS ambi_bin_getCodecStatus(void* const hAmbi) { PAB; return p->codecStatus; }
float ambi_bin_getProgressBar0_1(void* const hAmbi) { PAB; return p->progressBar0_1; }
void ambi_bin_getProgressBarText(void* const hAmbi, char* text) { PAB; memcpy(text, p->progressBarText, PROGRESSBARTEXT_CHAR_LENGTH); }
int ambi_bin_getUseDefaultHRIRsflag(void* const hAmbi) { PAB; return p->useDefaultHRIRsFLAG; }
int ambi_bin_getInputOrderPreset(void* const hAmbi) { PAB; return p->new_order; }
AMBI_BIN_DECODING_METHODS ambi_bin_getDecodingMethod(void* const hAmbi) { PAB; return (AMBI_BIN_DECODING_METHODS)p->method; }
char* ambi_bin_getSofaFilePath(void* const hAmbi) { PAB; return p->sofa_filepath.empty() ? (char*)"no_file" : (char*)p->sofa_filepath.c_str(); }
int ambi_bin_getChOrder(void* const hAmbi) { PAB; return (int)p->chOrdering; }
int ambi_bin_getNormType(void* const hAmbi) { PAB; return (int)p->norm; }
int ambi_bin_getNumEars(void) { return 2; }
int ambi_bin_getNSHrequired(void* const hAmbi) { PAB; return ORDER2NSH(p->order); }
int ambi_bin_getEnableMaxRE(void* const hAmbi) { PAB; return p->enableMaxRE; }
int ambi_bin_getEnableDiffuseMatching(void* const hAmbi) { PAB; return p->enableDiffuseMatching; }
int ambi_bin_getEnableTruncationEQ(void* const hAmbi) { PAB; return p->enableTruncationEQ; }
AMBI_BIN_PREPROC ambi_bin_getHRIRsPreProc(void* const hAmbi) { PAB; return (AMBI_BIN_PREPROC)p->preProc; }
int ambi_bin_getEnableRotation(void* const hAmbi) { PAB; return p->enableRotation; }
int ambi_bin_getFlipYaw(void* const hAmbi) { PAB; return p->bFlipYaw; }
int ambi_bin_getFlipPitch(void* const hAmbi) { PAB; return p->bFlipPitch; }
int ambi_bin_getFlipRoll(void* const hAmbi) { PAB; return p->bFlipRoll; }
int ambi_bin_getRPYflag(void* const hAmbi) { PAB; return p->useRollPitchYawFlag; }
int ambi_bin_getNDirs(void* const hAmbi) { PAB; return p->N_hrir_dirs; }
int ambi_bin_getHRIRlength(void* const hAmbi) { PAB; return p->hrir_len; }
int ambi_bin_getHRIRsamplerate(void* const hAmbi) { PAB; return p->hrir_fs; }
int ambi_bin_getDAWsamplerate(void* const hAmbi) { PAB; return p->fs; }
int ambi_bin_getProcessingDelay(void) { return 12 * SAF_HOP; }

}
