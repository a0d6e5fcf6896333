/*
 * panner.cpp — the frequency-dependent VBAP panner (examples/include/panner.h:83-325, examples/src/panner/panner.c,
 * panner_internal.c) with its per-block path on the GPU:
 *
 *   inputs -> [afSTFT analysis, 1/sqrt(nSources) folded in] -> [per moved source: table row -> per-band p-norm gains]
 *          -> [band GEMM  out_b = G_b^T x_b  on MFMA]        -> [afSTFT synthesis, nLoudspeakers] -> outputs
 *
 * FORCE_3D_LAYOUT is defined in the reference (panner_internal.h:66): the gain table is always the 3-D one at 1 x 1
 * degree with dummy loudspeakers and the large-triangle filter (panner_internal.c:59-100), so only the 3-D branch of
 * panner_process (panner.c:230-273) exists here.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "afstft_state.h"
#include "design_host.h"
#include "presets.h"
#include <thread>
#include <chrono>

namespace saf {

static int g_pan_frame_size = 128;       /* default of the reference (panner_internal.h:67-73) */

static inline void psleep_ms(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }

struct Panner {
    int F, T;
    int fs = 48000;
    float freqVector[SAF_NBANDS], pValue[SAF_NBANDS];
    bool haveSTFT = false;
    /* gain table (panner_internal.h:96-99) */
    std::vector<float> vbap_gtable;
    int N_vbap_gtable = 0, nTriangles = 0;
    int vbapTableRes[2] = { 1, 1 };
    /* flags / status */
    volatile CODEC_STATUS codecStatus;
    volatile PROC_STATUS procStatus;
    float progressBar0_1 = 0.0f;
    char progressBarText[PROGRESSBARTEXT_CHAR_LENGTH];
    int recalc_gainsFLAG[SAF_MAXCH];
    int recalc_M_rotFLAG = 1, reInitGainTables = 1;
    /* user parameters */
    int nSources, new_nSources, nLoudpkrs, new_nLoudpkrs, output_nDims = 3;
    float src_dirs_deg[SAF_MAXCH][2], src_dirs_rot_deg[SAF_MAXCH][2], loudpkrs_dirs_deg[SAF_MAXCH][2];
    float DTT = 0.5f, spread_deg = 0.0f, yaw = 0.0f, pitch = 0.0f, roll = 0.0f;
    int bFlipYaw = 0, bFlipPitch = 0, bFlipRoll = 0;
    /* device side */
    AfState st;
    int Hmax = 0;
    bool tableDirty = true, pDirty = true;
    DevBuf<float2> X, Y;
    DevBuf<float> d_gtable, d_pValue, d_A, d_Afrag, d_scale, d_in, d_out;
    DevBuf<int> d_row, d_recalc, d_band2mat;
    PinBuf<int> stI; PinBuf<float> stF;
    float scaleOnDevice = -1.0f;
    PinBuf<float> h_in, h_out;
};

static void set_codec_status(Panner* p, CODEC_STATUS s)     /* panner_internal.c:47-57 */
{
    if (s == CODEC_STATUS_NOT_INITIALISED)
        while (p->codecStatus == CODEC_STATUS_INITIALISING) psleep_ms(10);
    p->codecStatus = s;
}

/* panner_loadLoudspeakerPreset (panner_internal.c:326-518): default and unknown ids are stereo; 22.2 is allowed here */
static void pan_ls_preset(int preset, float dirs[][2], int* n)
{
    int dims;
    if (preset == 12) {
        const float* t = table_required("9_10_3p2_dirs_deg", 48);
        float tmp[SAF_MAXCH][2]; int m;
        load_source_preset(3, tmp, &m);                                  /* stereo: fills the tail with the default coordinates */
        for (int ch = 0; ch < 24; ch++) { tmp[ch][0] = t[2 * ch]; tmp[ch][1] = t[2 * ch + 1]; }
        memcpy(dirs, tmp, sizeof(tmp)); *n = 24; return;
    }
    if (preset < 3 || preset > 29 || preset == 18 /* Zylia: not in the panner's list */) { load_source_preset(3, dirs, n); return; }
    load_loudspeaker_preset(preset, dirs, n, &dims);
}
/* panner_loadSourcePreset (panner_internal.c:119-324) */
static void pan_src_preset(int preset, float dirs[][2], int* n)
{
    if (preset >= 3 && preset <= 30 && preset != 19) { pan_ls_preset(preset - 1, dirs, n); return; }
    load_source_preset(2, dirs, n);                                       /* mono table + default tail */
    if (preset != 2) dirs[0][0] = dirs[0][1] = 0.0f;                      /* SOURCE_CONFIG_PRESET_DEFAULT: one source at (0, 0) */
    *n = 1;
}

/* yawPitchRoll2Rzyx with rollPitchYawFLAG = 0 (saf_utility_geometry.c:213-270) */
static void rot_zyx(float yaw, float pitch, float roll, float R[3][3])
{
    const float Rx[3][3] = { { 1, 0, 0 }, { 0, cosf(roll), sinf(roll) }, { 0, -sinf(roll), cosf(roll) } };
    const float Ry[3][3] = { { cosf(pitch), 0, -sinf(pitch) }, { 0, 1, 0 }, { sinf(pitch), 0, cosf(pitch) } };
    const float Rz[3][3] = { { cosf(yaw), sinf(yaw), 0 }, { -sinf(yaw), cosf(yaw), 0 }, { 0, 0, 1 } };
    float Tm[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += Ry[i][k] * Rz[k][j]; Tm[i][j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += Rx[i][k] * Tm[k][j]; R[i][j] = a; }
}
static float matlab_fmodf(float x, float y) { const float t = fmodf(x, y); return t >= 0 ? t : t + y; }   /* saf_utility_misc.c:188-191 */

/* panner_initGainTables (panner_internal.c:59-100), FORCE_3D_LAYOUT branch */
static void init_gain_tables(Panner* p)
{
    p->vbapTableRes[0] = 1; p->vbapTableRes[1] = 1; p->output_nDims = 3;
    std::vector<float> grid;
    vbap_grid_dirs(1, 1, grid);
    p->N_vbap_gtable = (int)grid.size() / 2;
    if (!vbap_table(grid.data(), p->N_vbap_gtable, &p->loudpkrs_dirs_deg[0][0], p->nLoudpkrs, 1, 1, p->spread_deg, p->vbap_gtable, &p->nTriangles)) {
        p->vbap_gtable.clear(); p->N_vbap_gtable = 0;         /* the reference keeps a NULL table: process then outputs zeros (panner.c:192) */
    }
    p->tableDirty = true;
}

/* the block path for nFrames consecutive blocks of device-resident samples */
static void process_dev(Panner* p, const float* d_in, long long in_frame, long long in_ch, int nIn,
                        float* d_out, long long out_frame, long long out_ch, int nFrames)
{
    const int nS = p->nSources, nL = p->nLoudpkrs, T = p->T, H = nFrames * T;
    if (H > p->Hmax) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->Hmax = (H + 15) & ~15;
        p->X.alloc((size_t)SAF_NBANDS * SAF_MAXCH * p->Hmax, true);
        p->Y.alloc((size_t)SAF_NBANDS * SAF_MAXCH * p->Hmax, true);
    }
    if (p->tableDirty) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->d_gtable.alloc(p->vbap_gtable.size(), false);
        HIP_CHECK(hipMemcpy(p->d_gtable.p, p->vbap_gtable.data(), sizeof(float) * p->vbap_gtable.size(), hipMemcpyHostToDevice));
        p->d_A.zero();                                      /* columns of sources that no longer exist must not survive */
        p->tableDirty = false;
    }
    if (p->pDirty) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        HIP_CHECK(hipMemcpy(p->d_pValue.p, p->pValue, sizeof(float) * SAF_NBANDS, hipMemcpyHostToDevice));
        p->pDirty = false;
    }
    /* 1/sqrt(nSources) (panner.c:308-310) rides on the analysis kernel's per-channel scale: every stage is linear */
    const float sc = 1.0f / sqrtf((float)nS);
    if (sc != p->scaleOnDevice) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        for (int ch = 0; ch < SAF_MAXCH; ch++) p->stF.p[ch] = sc;
        HIP_CHECK(hipMemcpyAsync(p->d_scale.p, p->stF.p, sizeof(float) * SAF_MAXCH, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->scaleOnDevice = sc;
    }
    AnaLaunch a{};
    a.in = d_in; a.in_inst = 0; a.in_ch = in_ch; a.in_frame = in_frame; a.hopsPerFrame = T; a.nChIn = nS < nIn ? nS : nIn;
    a.hist_rd = p->st.ana[p->st.anaPar].p; a.hist_wr = p->st.ana[p->st.anaPar ^ 1].p;
    a.out = p->X.p; a.out_inst = 0; a.out_band = (long long)SAF_MAXCH * p->Hmax; a.out_ch = p->Hmax;
    a.ch_scale = p->d_scale.p; a.ch_map = nullptr; a.tab_stride = SAF_MAXCH;
    a.nCh = nS; a.nInst = 1; a.H = H; a.lowDelay = 0; a.hybrid = 1;
    launch_analysis(a);
    p->st.anaPar ^= 1;

    /* rotate the source directions (panner.c:206-226); same float operations, so the same table rows are picked */
    if (p->recalc_M_rotFLAG) {
        float R[3][3];
        rot_zyx(p->yaw, p->pitch, p->roll, R);
        for (int i = 0; i < nS; i++) {
            const float az = p->src_dirs_deg[i][0] * SAF_PI / 180.0f, el = p->src_dirs_deg[i][1] * SAF_PI / 180.0f;
            const float x[3] = { cosf(el) * cosf(az), cosf(el) * sinf(az), sinf(el) };
            float r[3];
            for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += x[k] * R[k][j]; r[j] = s; }
            const float hyp = sqrtf(powf(r[0], 2.0f) + powf(r[1], 2.0f));
            p->src_dirs_rot_deg[i][0] = atan2f(r[1], r[0]) * 180.0f / SAF_PI;
            p->src_dirs_rot_deg[i][1] = atan2f(r[2], hyp) * 180.0f / SAF_PI;
            p->recalc_gainsFLAG[i] = 1;
        }
        p->recalc_M_rotFLAG = 0;
    }
    bool any = false;
    for (int ch = 0; ch < nS; ch++) any = any || p->recalc_gainsFLAG[ch];
    if (any) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        const float aziRes = (float)p->vbapTableRes[0], elevRes = (float)p->vbapTableRes[1];
        const int N_azi = (int)(360.0f / aziRes + 0.5f) + 1;
        for (int ch = 0; ch < nS; ch++) {
            const int aziIndex = (int)(matlab_fmodf(p->src_dirs_rot_deg[ch][0] + 180.0f, 360.0f) / aziRes + 0.5f);
            const int elevIndex = (int)((p->src_dirs_rot_deg[ch][1] + 90.0f) / elevRes + 0.5f);
            p->stI.p[ch] = elevIndex * N_azi + aziIndex;
            p->stI.p[SAF_MAXCH + ch] = p->recalc_gainsFLAG[ch];
            p->recalc_gainsFLAG[ch] = 0;
        }
        HIP_CHECK(hipMemcpyAsync(p->d_row.p, p->stI.p, sizeof(int) * nS, hipMemcpyHostToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(p->d_recalc.p, p->stI.p + SAF_MAXCH, sizeof(int) * nS, hipMemcpyHostToDevice, stream()));
        PanGainLaunch l{};
        l.gtable = p->d_gtable.p; l.row = p->d_row.p; l.recalc = p->d_recalc.p; l.pValue = p->d_pValue.p; l.A = p->d_A.p;
        l.nSrc = nS; l.nLS = nL;
        launch_panner_gains(l);
        launch_pack_A(p->d_A.p, p->d_Afrag.p, SAF_NBANDS);
    }
    BandGemmLaunch g{};
    g.X = (const float*)p->X.p; g.x_inst = 0; g.x_band = 2 * a.out_band; g.x_row = 2 * a.out_ch;
    g.Y = (float*)p->Y.p; g.y_inst = 0; g.y_band = g.x_band; g.y_row = g.x_row;
    g.Afrag = p->d_Afrag.p; g.a_inst = 0; g.band2mat = p->d_band2mat.p;
    g.nBands = SAF_NBANDS; g.nInst = 1; g.N = 2 * H;
    launch_band_gemm(g);

    SynLaunch s{};
    s.in = p->Y.p; s.in_inst = 0; s.in_band = a.out_band; s.in_ch = a.out_ch;
    s.out = d_out; s.out_inst = 0; s.out_ch = out_ch; s.out_frame = out_frame; s.hopsPerFrame = T;
    s.hist_rd = p->st.syn[p->st.synPar].p; s.hist_wr = p->st.syn[p->st.synPar ^ 1].p;
    s.nCh = nL; s.nInst = 1; s.H = H; s.lowDelay = 0; s.hybrid = 1;
    launch_synthesis(s);
    p->st.synPar ^= 1;
}

}  // namespace saf

using namespace saf;

extern "C" {

void saf_hip_panner_setFrameSize(int frameSize)
{
    if (frameSize <= 0 || frameSize % SAF_HOP != 0) SAF_FATAL("panner frame size must be a positive multiple of 128");
    g_pan_frame_size = frameSize;
}

/* getPvalues (saf_vbap.c:475-492) */
void getPvalues(float DTT, float* freq, int nFreq, float* pValues)
{
    const float a1 = 0.00045f, a2 = 0.000085f;
    for (int i = 0; i < nFreq; i++) {
        const float lim = 1.0f - a2 * freq[i];
        const float p0 = 1.5f - 0.5f * cosf(4.7f * tanhf(a1 * freq[i])) * (lim > 0.0f ? lim : 0.0f);
        pValues[i] = (p0 - 2.0f) * sqrtf(DTT) + 2.0f;
    }
}

void panner_create(void** const phPan)        /* panner.c:46-91 */
{
    Panner* p = new Panner();
    *phPan = p;
    p->F = g_pan_frame_size; p->T = p->F / SAF_HOP;
    pan_src_preset(SOURCE_CONFIG_PRESET_DEFAULT, p->src_dirs_deg, &p->new_nSources);
    p->nSources = p->new_nSources;
    pan_ls_preset(LOUDSPEAKER_ARRAY_PRESET_STEREO, p->loudpkrs_dirs_deg, &p->new_nLoudpkrs);
    p->nLoudpkrs = p->new_nLoudpkrs;
    memset(p->src_dirs_rot_deg, 0, sizeof(p->src_dirs_rot_deg));
    p->codecStatus = CODEC_STATUS_NOT_INITIALISED; p->procStatus = PROC_STATUS_NOT_ONGOING;
    p->progressBarText[0] = 0;
    for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc_gainsFLAG[ch] = 1;
    memset(p->freqVector, 0, sizeof(p->freqVector)); memset(p->pValue, 0, sizeof(p->pValue));
}

void panner_destroy(void** const phPan)       /* panner.c:93-116 */
{
    Panner* p = (Panner*)*phPan;
    if (!p) return;
    while (p->codecStatus == CODEC_STATUS_INITIALISING || p->procStatus == PROC_STATUS_ONGOING) psleep_ms(10);
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete p;
    *phPan = nullptr;
}

void panner_init(void* const hPan, int sampleRate)     /* panner.c:118-135 */
{
    Panner* p = (Panner*)hPan;
    p->fs = sampleRate;
    if (!p->haveSTFT) afSTFT_getCentreFreqs(nullptr, (float)sampleRate, SAF_NBANDS, p->freqVector);      /* NULL-handle table branch (afSTFTlib.c:554-563) */
    else {   /* valid-handle branch (afSTFTlib.c:565-587), hop 128 hybrid */
        static const float w[9] = { 1.0f, 0.7501f, 1.2499f, 0.8751f, 1.1249f, 0.9167f, 1.0833f, 0.9375f, 1.0625f };
        static const int bin[9] = { 0, 1, 1, 2, 2, 3, 3, 4, 4 };
        for (int i = 0; i < 9; i++) p->freqVector[i] = w[i] * ((float)bin[i] * (float)sampleRate / 256.0f);
        for (int i = 9, j = 5; i < SAF_NBANDS; i++, j++) p->freqVector[i] = (float)j * (float)sampleRate / 256.0f;
    }
    getPvalues(p->DTT, p->freqVector, SAF_NBANDS, p->pValue);
    p->pDirty = true;
    p->recalc_M_rotFLAG = 1;
}

void panner_initCodec(void* const hPan)       /* panner.c:137-171 */
{
    Panner* p = (Panner*)hPan;
    if (p->codecStatus != CODEC_STATUS_NOT_INITIALISED) return;
    while (p->procStatus == PROC_STATUS_ONGOING) { p->codecStatus = CODEC_STATUS_INITIALISING; psleep_ms(10); }
    ensure_device();
    p->codecStatus = CODEC_STATUS_INITIALISING;
    strcpy(p->progressBarText, "Initialising"); p->progressBar0_1 = 0.0f;
    /* panner_initTFT (panner_internal.c:102-117) */
    HIP_CHECK(hipStreamSynchronize(stream()));
    if (!p->haveSTFT) {
        p->st.create(1, p->new_nSources, p->new_nLoudpkrs);
        p->d_A.alloc((size_t)SAF_NBANDS * 64 * 64); p->d_Afrag.alloc((size_t)SAF_NBANDS * 64 * 64);
        p->d_pValue.alloc(SAF_NBANDS); p->d_scale.alloc(SAF_MAXCH); p->d_row.alloc(SAF_MAXCH); p->d_recalc.alloc(SAF_MAXCH);
        p->d_band2mat.alloc(SAF_NBANDS, false);
        p->stI.ensure(2 * SAF_MAXCH); p->stF.ensure(SAF_MAXCH);
        std::vector<int> b2m(SAF_NBANDS);
        for (int b = 0; b < SAF_NBANDS; b++) b2m[b] = b;
        HIP_CHECK(hipMemcpy(p->d_band2mat.p, b2m.data(), sizeof(int) * SAF_NBANDS, hipMemcpyHostToDevice));
        p->haveSTFT = true;
    } else if (p->new_nSources != p->nSources || p->new_nLoudpkrs != p->nLoudpkrs) {
        p->st.channelChange(p->new_nSources, p->new_nLoudpkrs); p->st.clear();
        if (p->new_nSources < p->nSources) { p->d_A.zero(); for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc_gainsFLAG[ch] = 1; }
    }
    p->nSources = p->new_nSources; p->nLoudpkrs = p->new_nLoudpkrs;
    if (p->reInitGainTables) { init_gain_tables(p); p->reInitGainTables = 0; }
    strcpy(p->progressBarText, "Done!"); p->progressBar0_1 = 1.0f;
    p->codecStatus = CODEC_STATUS_INITIALISED;
}

void panner_process(void* const hPan, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)   /* panner.c:173-323 */
{
    Panner* p = (Panner*)hPan;
    const int F = p->F, nS = p->nSources, nL = p->nLoudpkrs;
    if (nSamples == F && !p->vbap_gtable.empty() && p->codecStatus == CODEC_STATUS_INITIALISED) {
        p->procStatus = PROC_STATUS_ONGOING;
        const int nIn = nS < nInputs ? nS : (nInputs < 0 ? 0 : nInputs);
        p->h_in.ensure((size_t)SAF_MAXCH * F); p->h_out.ensure((size_t)SAF_MAXCH * F);
        if (p->d_in.n < (size_t)SAF_MAXCH * F) { p->d_in.alloc((size_t)SAF_MAXCH * F, true); p->d_out.alloc((size_t)SAF_MAXCH * F, true); }
        for (int i = 0; i < nIn; i++) memcpy(p->h_in.p + (size_t)i * F, inputs[i], sizeof(float) * F);
        if (zero_copy_io()) process_dev(p, p->h_in.p, 0, F, nIn, p->h_out.p, 0, F, 1);                    /* kernels on the pinned blocks */
        else {
            if (nIn) HIP_CHECK(hipMemcpyAsync(p->d_in.p, p->h_in.p, sizeof(float) * (size_t)nIn * F, hipMemcpyHostToDevice, stream()));
            process_dev(p, p->d_in.p, 0, F, nIn, p->d_out.p, 0, F, 1);
            HIP_CHECK(hipMemcpyAsync(p->h_out.p, p->d_out.p, sizeof(float) * (size_t)nL * F, hipMemcpyDeviceToHost, stream()));
        }
        HIP_CHECK(hipStreamSynchronize(stream()));
        int ch;
        for (ch = 0; ch < (nL < nOutputs ? nL : nOutputs); ch++) memcpy(outputs[ch], p->h_out.p + (size_t)ch * F, sizeof(float) * F);
        for (; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    } else
        for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}

void saf_hip_panner_process_dev(void* const hPan, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames)
{
    Panner* p = (Panner*)hPan;
    if (p->vbap_gtable.empty() || p->codecStatus != CODEC_STATUS_INITIALISED) SAF_FATAL("panner: process_dev on a handle that is not initialised (call panner_initCodec)");
    p->procStatus = PROC_STATUS_ONGOING;
    process_dev(p, d_in, in_frame_stride, in_ch_stride, nInputs < 0 ? 0 : nInputs, d_out, out_frame_stride, out_ch_stride, nFrames);
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}

/* G_src read-back for parity checks: [133][64][64] band, source, loudspeaker (panner_internal.h:99) */
void saf_hip_panner_getGains(void* const hPan, float* G)
{
    Panner* p = (Panner*)hPan;
    std::vector<float> A((size_t)SAF_NBANDS * 64 * 64);
    HIP_CHECK(hipStreamSynchronize(stream()));
    HIP_CHECK(hipMemcpy(A.data(), p->d_A.p, sizeof(float) * A.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < SAF_NBANDS; b++)
        for (int s = 0; s < 64; s++)
            for (int l = 0; l < 64; l++) G[((size_t)b * 64 + s) * 64 + l] = A[((size_t)b * 64 + l) * 64 + s];
}

/* ------------------------------- set functions (panner.c:335-540) ------------------------------- */
#define PPN Panner* p = (Panner*)hPan
static void flag_all(Panner* p) { for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc_gainsFLAG[ch] = 1; }
void panner_refreshSettings(void* const hPan) { PPN; p->reInitGainTables = 1; flag_all(p); set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
void panner_setSourceAzi_deg(void* const hPan, int index, float v)
{
    PPN;
    if (v > 180.0f) v = -360.0f + v;
    v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    if (p->src_dirs_deg[index][0] != v) { p->src_dirs_deg[index][0] = v; p->recalc_gainsFLAG[index] = 1; p->recalc_M_rotFLAG = 1; }
}
void panner_setSourceElev_deg(void* const hPan, int index, float v)
{
    PPN;
    v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    if (p->src_dirs_deg[index][1] != v) { p->src_dirs_deg[index][1] = v; p->recalc_gainsFLAG[index] = 1; p->recalc_M_rotFLAG = 1; }
}
void panner_setNumSources(void* const hPan, int n)
{
    PPN;
    n = n > SAF_MAXCH ? SAF_MAXCH : n;
    if (p->nSources != n) {
        p->new_nSources = n;
        for (int ch = p->nSources; ch < p->new_nSources; ch++) p->recalc_gainsFLAG[ch] = 1;
        p->recalc_M_rotFLAG = 1;
        set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
    }
}
void panner_setLoudspeakerAzi_deg(void* const hPan, int index, float v)
{
    PPN;
    if (v > 180.0f) v = -360.0f + v;
    v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    if (p->loudpkrs_dirs_deg[index][0] != v) {
        p->loudpkrs_dirs_deg[index][0] = v; p->reInitGainTables = 1; flag_all(p); p->recalc_M_rotFLAG = 1;
        set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
    }
}
void panner_setLoudspeakerElev_deg(void* const hPan, int index, float v)
{
    PPN;
    v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    if (p->loudpkrs_dirs_deg[index][1] != v) {
        p->loudpkrs_dirs_deg[index][1] = v; p->reInitGainTables = 1; flag_all(p); p->recalc_M_rotFLAG = 1;
        set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
    }
}
void panner_setNumLoudspeakers(void* const hPan, int n)
{
    PPN;
    n = n > SAF_MAXCH ? SAF_MAXCH : n;
    if (p->new_nLoudpkrs != n) {
        p->new_nLoudpkrs = n; p->reInitGainTables = 1; flag_all(p); p->recalc_M_rotFLAG = 1;
        set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
    }
}
void panner_setOutputConfigPreset(void* const hPan, int newPresetID)
{
    PPN;
    pan_ls_preset(newPresetID, p->loudpkrs_dirs_deg, &p->new_nLoudpkrs);
    p->reInitGainTables = 1; flag_all(p); p->recalc_M_rotFLAG = 1;
    set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
}
void panner_setInputConfigPreset(void* const hPan, int newPresetID)
{
    PPN;
    pan_src_preset(newPresetID, p->src_dirs_deg, &p->new_nSources);
    for (int ch = 0; ch < p->new_nSources; ch++) p->recalc_gainsFLAG[ch] = 1;
    p->recalc_M_rotFLAG = 1;
    set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
}
void panner_setDTT(void* const hPan, float v)
{
    PPN;
    if (p->DTT != v) {
        p->DTT = v;
        getPvalues(p->DTT, p->freqVector, SAF_NBANDS, p->pValue);
        p->pDirty = true;
        for (int ch = 0; ch < p->new_nSources; ch++) p->recalc_gainsFLAG[ch] = 1;
        p->recalc_M_rotFLAG = 1;
        set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
    }
}
void panner_setSpread(void* const hPan, float v)
{
    PPN;
    if (p->spread_deg != v) {
        p->spread_deg = v < PANNER_SPREAD_MIN_VALUE ? PANNER_SPREAD_MIN_VALUE : (v > PANNER_SPREAD_MAX_VALUE ? PANNER_SPREAD_MAX_VALUE : v);
        p->reInitGainTables = 1; flag_all(p); p->recalc_M_rotFLAG = 1;
        set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
    }
}
void panner_setYaw(void* const hPan, float v) { PPN; p->yaw = p->bFlipYaw == 1 ? -(v * SAF_PI / 180.0f) : v * SAF_PI / 180.0f; p->recalc_M_rotFLAG = 1; }
void panner_setPitch(void* const hPan, float v) { PPN; p->pitch = p->bFlipPitch == 1 ? -(v * SAF_PI / 180.0f) : v * SAF_PI / 180.0f; p->recalc_M_rotFLAG = 1; }
void panner_setRoll(void* const hPan, float v) { PPN; p->roll = p->bFlipRoll == 1 ? -(v * SAF_PI / 180.0f) : v * SAF_PI / 180.0f; p->recalc_M_rotFLAG = 1; }
float panner_getYaw(void* const hPan) { PPN; return p->bFlipYaw == 1 ? -(p->yaw * 180.0f / SAF_PI) : p->yaw * 180.0f / SAF_PI; }
float panner_getPitch(void* const hPan) { PPN; return p->bFlipPitch == 1 ? -(p->pitch * 180.0f / SAF_PI) : p->pitch * 180.0f / SAF_PI; }
float panner_getRoll(void* const hPan) { PPN; return p->bFlipRoll == 1 ? -(p->roll * 180.0f / SAF_PI) : p->roll * 180.0f / SAF_PI; }
void panner_setFlipYaw(void* const hPan, int s) { PPN; if (s != p->bFlipYaw) { p->bFlipYaw = s; panner_setYaw(hPan, -panner_getYaw(hPan)); } }
void panner_setFlipPitch(void* const hPan, int s) { PPN; if (s != p->bFlipPitch) { p->bFlipPitch = s; panner_setPitch(hPan, -panner_getPitch(hPan)); } }
void panner_setFlipRoll(void* const hPan, int s) { PPN; if (s != p->bFlipRoll) { p->bFlipRoll = s; panner_setRoll(hPan, -panner_getRoll(hPan)); } }

/* ------------------------------- get functions (panner.c:543-664) ------------------------------- */
int panner_getFrameSize(void) { return g_pan_frame_size; }
CODEC_STATUS panner_getCodecStatus(void* const hPan) { PPN; return p->codecStatus; }
float panner_getProgressBar0_1(void* const hPan) { PPN; return p->progressBar0_1; }
void panner_getProgressBarText(void* const hPan, char* text) { PPN; memcpy(text, p->progressBarText, PROGRESSBARTEXT_CHAR_LENGTH); }
float panner_getSourceAzi_deg(void* const hPan, int index) { PPN; return p->src_dirs_deg[index][0]; }
float panner_getSourceElev_deg(void* const hPan, int index) { PPN; return p->src_dirs_deg[index][1]; }
int panner_getNumSources(void* const hPan) { PPN; return p->new_nSources; }
int panner_getMaxNumSources(void) { return SAF_MAXCH; }
float panner_getLoudspeakerAzi_deg(void* const hPan, int index) { PPN; return p->loudpkrs_dirs_deg[index][0]; }
float panner_getLoudspeakerElev_deg(void* const hPan, int index) { PPN; return p->loudpkrs_dirs_deg[index][1]; }
int panner_getNumLoudspeakers(void* const hPan) { PPN; return p->new_nLoudpkrs; }
int panner_getMaxNumLoudspeakers(void) { return SAF_MAXCH; }
int panner_getDAWsamplerate(void* const hPan) { PPN; return p->fs; }
float panner_getDTT(void* const hPan) { PPN; return p->DTT; }
float panner_getSpread(void* const hPan) { PPN; return p->spread_deg; }
int panner_getFlipYaw(void* const hPan) { PPN; return p->bFlipYaw; }
int panner_getFlipPitch(void* const hPan) { PPN; return p->bFlipPitch; }
int panner_getFlipRoll(void* const hPan) { PPN; return p->bFlipRoll; }
int panner_getProcessingDelay(void) { return 12 * SAF_HOP; }

}
