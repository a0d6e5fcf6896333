/*
 * binauraliser.cpp — the binauraliser operator (examples/include/binauraliser.h:73-376,
 * examples/src/binauraliser/binauraliser.c, binauraliser_internal.c) with its per-block path on the GPU:
 *
 *   inputs -> [afSTFT analysis, source gains folded in] -> [per moved source: HRTF interpolation kernel]
 *          -> [band MAC over the sources, 1/sqrt(nSources)] -> [afSTFT synthesis, 2 ears] -> outputs
 *
 * Init (binauraliser_initHRTFsAndGainTables, binauraliser_internal.c:125-263) runs once per HRIR set: ITDs,
 * VBAP interpolation table over the HRIR grid, HRIR -> filterbank coefficients (GPU analysis), optional
 * diffuse-field equalisation with spherical-Voronoi weights.
 *
 * HRIR data: the reference's default set is missing from its checkout and SOFA loading (libmysofa/netCDF) is
 * file I/O outside the hot path, so the set in use is the one installed with saf_hip_setDefaultHRIRs.
 * The reference caps the sources at MAX_NUM_INPUTS = 64 (_common.h:231); saf_hip_binauraliser_setMaxNumSources
 * raises that cap for handles created afterwards (BASELINE configs[2] renders 256 sources).
 *
 * binauraliser_nf (examples/include/binauraliser_nf.h:76-194, examples/src/binauraliser_nf): the same operator with a
 * distance per source.  A handle from binauraliserNF_create is a binauraliser handle (every binauraliser_* call applies,
 * as in the reference, whose NF struct begins with the binauraliser members); binauraliserNF_process adds one step: for
 * each moved source the two DVF shelves (dvf_host.cpp) are re-derived on the host and one small kernel multiplies their
 * band responses onto the interpolated HRTFs of the near sources before the band MAC.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "afstft_state.h"
#include "design_host.h"
#include "hrir_host.h"
#include "presets.h"
#include "dvf_host.h"
#include <thread>
#include <chrono>

namespace saf {

static int g_bin_frame_size = 128;       /* default of the reference (binauraliser_internal.h:60-66) */
static int g_bin_max_sources = SAF_MAXCH;

static inline void bsleep_ms(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }

struct Binauraliser {
    int F, T, maxSrc;
    int fs = 48000;
    float freqVector[SAF_NBANDS];
    bool haveSTFT = false;
    /* HRIR-derived tables (binauraliser_internal.h:95-118) */
    int N_hrir_dirs = 0, hrir_loaded_len = 0, hrir_runtime_len = 0, hrir_loaded_fs = -1, hrir_runtime_fs = -1;
    std::vector<float> hrirs, hrir_dirs_deg, itds_s, weights, hrtf_fb_mag, gtableComp;
    std::vector<float2> hrtf_fb;
    std::vector<int> gtableIdx;
    int N_hrtf_vbap_gtable = 0, nTriangles = 0;
    int hrtf_vbapTableRes[2] = { 2, 5 };
    std::string sofa_filepath;
    /* flags / status */
    volatile CODEC_STATUS codecStatus;
    volatile PROC_STATUS procStatus;
    float progressBar0_1 = 0.0f;
    char progressBarText[PROGRESSBARTEXT_CHAR_LENGTH];
    int reInitHRTFsAndGainTables = 1, recalc_M_rotFLAG = 1;
    std::vector<int> recalc_hrtf_interpFLAG;
    /* user parameters */
    int nSources, new_nSources;
    std::vector<float> src_dirs_deg, src_dirs_rot_deg, src_gains;     /* [maxSrc][2], [maxSrc][2], [maxSrc] */
    int useDefaultHRIRsFLAG = 1, enableHRIRsDiffuseEQ = 1, enableRotation = 0, useRollPitchYawFlag = 0;
    int bFlipYaw = 0, bFlipPitch = 0, bFlipRoll = 0, interpMode = 1 /* INTERP_TRI */;
    float yaw = 0.0f, pitch = 0.0f, roll = 0.0f;
    /* device side */
    AfState st;
    int Hmax = 0;
    unsigned long long tablesEpoch = 0, tablesOnDevice = ~0ull;
    DevBuf<float2> X, Y, d_hrtf_fb, d_hrtf_interp;
    DevBuf<float> d_mag, d_itds, d_gtComp, d_freq, d_dirs, d_gains;
    DevBuf<int> d_gtIdx, d_recalc;
    PinBuf<float> stF; PinBuf<int> stI;
    std::vector<float> shadowGains, freqOnDevice;
    bool stagingBusy = false;               /* a kernel or copy enqueued by a device-entry call may still read stF / stI / stDvf */
    PinBuf<float> h_in, h_out;
    DevBuf<float> d_in, d_out;
    /* binauraliser_nf (binauraliser_nf_internal.h:140-158) */
    bool nf = false;
    int curRot = 0;                         /* src_dirs_cur points at the rotated directions (binauraliser_nf.c:285-289) */
    std::vector<int> recalc_dvfCoeffFLAG;
    std::vector<float> src_dists_m, dvfCoef, dvfCoefOnDevice;   /* [maxSrc], [maxSrc][2][4] = b0 b1 a1 near */
    float head_radius = 0.09096f, head_radius_recip = 0.0f, farfield_thresh_m = 0.0f, farfield_headroom = 1.05f, nearfield_limit_m = 0.15f;
    DevBuf<float2> d_hrtf_nf;
    DevBuf<float> d_dvfCoef;
    PinBuf<float> stDvf;
};

/* binauraliserNF_process (binauraliser_nf.c:291-318): shelf coefficients of the sources whose direction or distance changed.
 * Fills p->dvfCoef; returns true when the device copy is stale. */
static bool nf_refresh_coeffs(Binauraliser* p)
{
    const int nS = p->nSources;
    const std::vector<float>& dirs = p->curRot ? p->src_dirs_rot_deg : p->src_dirs_deg;
    for (int ch = 0; ch < nS; ch++) {
        float* k = &p->dvfCoef[(size_t)ch * 8];
        if (p->recalc_dvfCoeffFLAG[ch]) {
            const float rho = p->src_dists_m[ch] * p->head_radius_recip;
            float alphaLR[2] = { 0.0f, 0.0f };
            dvf_lateral_angles(dirs[ch * 2], dirs[ch * 2 + 1], alphaLR, nullptr);
            for (int e = 0; e < 2; e++) {
                float b[2], a[2] = { 1.0f, 0.0f };
                dvf_coeffs(alphaLR[e], rho, (float)p->fs, b, a);
                k[e * 4 + 0] = b[0]; k[e * 4 + 1] = b[1]; k[e * 4 + 2] = a[1];
            }
            p->recalc_dvfCoeffFLAG[ch] = 0;
        }
        const float near = p->src_dists_m[ch] < p->farfield_thresh_m ? 1.0f : 0.0f;       /* binauraliser_nf.c:321 */
        k[3] = k[7] = near;
    }
    return p->dvfCoefOnDevice.size() != p->dvfCoef.size() || memcmp(p->dvfCoefOnDevice.data(), p->dvfCoef.data(), sizeof(float) * (size_t)nS * 8) != 0;
}

static void set_codec_status(Binauraliser* p, CODEC_STATUS s)     /* binauraliser_internal.c:32-44 */
{
    if (s == CODEC_STATUS_NOT_INITIALISED)
        while (p->codecStatus == CODEC_STATUS_INITIALISING) bsleep_ms(10);
    p->codecStatus = s;
}

/* yawPitchRoll2Rzyx (saf_utility_geometry.c:213-270) */
static void rot_matrix(float yaw, float pitch, float roll, int rollPitchYaw, float R[3][3])
{
    auto Rx = [](float t, float M[3][3]) { const float m[3][3] = { { 1, 0, 0 }, { 0, cosf(t), sinf(t) }, { 0, -sinf(t), cosf(t) } }; memcpy(M, m, sizeof(m)); };
    auto Ry = [](float t, float M[3][3]) { const float m[3][3] = { { cosf(t), 0, -sinf(t) }, { 0, 1, 0 }, { sinf(t), 0, cosf(t) } }; memcpy(M, m, sizeof(m)); };
    auto Rz = [](float t, float M[3][3]) { const float m[3][3] = { { cosf(t), sinf(t), 0 }, { -sinf(t), cosf(t), 0 }, { 0, 0, 1 } }; memcpy(M, m, sizeof(m)); };
    float R1[3][3], R2[3][3], R3[3][3], Tm[3][3];
    if (rollPitchYaw) { Rx(yaw, R1); Ry(pitch, R2); Rz(roll, R3); }      /* EULER_ROTATION_ROLL_PITCH_YAW with (alpha, beta, gamma) = (yaw, pitch, roll) */
    else { Rz(yaw, R1); Ry(pitch, R2); Rx(roll, R3); }                    /* EULER_ROTATION_YAW_PITCH_ROLL */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += R2[i][k] * R1[k][j]; Tm[i][j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += R3[i][k] * Tm[k][j]; R[i][j] = a; }
}

/* binauraliser_initHRTFsAndGainTables (binauraliser_internal.c:125-263) */
static void init_hrtfs_and_tables(Binauraliser* p)
{
    strcpy(p->progressBarText, "Loading HRIRs"); p->progressBar0_1 = 0.2f;
    const DefaultHRIRs& D = default_hrirs();
    if (D.N == 0)
        SAF_FATAL("binauraliser: no HRIR set installed.  The reference's default set (saf_default_hrirs.c) is not part of its checkout and "
                  "SOFA loading is outside this library: call saf_hip_setDefaultHRIRs() before binauraliser_initCodec().");
    p->useDefaultHRIRsFLAG = 1;                         /* "can only load the default HRIR data" (:167) */
    p->hrir_loaded_fs = D.fs; p->hrir_loaded_len = D.len; p->N_hrir_dirs = D.N;
    p->hrirs = D.hrirs; p->hrir_dirs_deg = D.dirs_deg;
    const int N = p->N_hrir_dirs;
    for (int i = 0; i < N; i++) if (p->hrir_dirs_deg[i * 2] > 180.0f) p->hrir_dirs_deg[i * 2] = -360.0f + p->hrir_dirs_deg[i * 2];   /* convert_0_360To_m180_180 */
    strcpy(p->progressBarText, "Estimating ITDs"); p->progressBar0_1 = 0.4f;
    p->itds_s.resize(N);
    estimateITDs(p->hrirs.data(), N, p->hrir_loaded_len, p->hrir_loaded_fs, p->itds_s.data());
    if (p->hrir_loaded_fs != p->fs)
        SAF_FATAL("binauraliser: the installed HRIR set is at %d Hz but the host runs at %d Hz; resampling (speex) is outside this library: "
                  "install a set at the host rate.", p->hrir_loaded_fs, p->fs);
    p->hrir_runtime_fs = p->hrir_loaded_fs; p->hrir_runtime_len = p->hrir_loaded_len;
    strcpy(p->progressBarText, "Generating interpolation table"); p->progressBar0_1 = 0.6f;
    p->hrtf_vbapTableRes[0] = 2; p->hrtf_vbapTableRes[1] = 5;
    std::vector<float> grid, gtable;
    vbap_grid_dirs(p->hrtf_vbapTableRes[0], p->hrtf_vbapTableRes[1], grid);
    p->N_hrtf_vbap_gtable = (int)grid.size() / 2;
    if (!vbap_table(grid.data(), p->N_hrtf_vbap_gtable, p->hrir_dirs_deg.data(), N, 1, 0, 0.0f, gtable, &p->nTriangles))
        SAF_FATAL("binauraliser: the HRIR measurement grid could not be triangulated");
    p->gtableComp.resize((size_t)p->N_hrtf_vbap_gtable * 3); p->gtableIdx.resize((size_t)p->N_hrtf_vbap_gtable * 3);
    compressVBAPgainTable3D(gtable.data(), p->N_hrtf_vbap_gtable, N, p->gtableComp.data(), p->gtableIdx.data());
    p->hrtf_fb.resize((size_t)SAF_NBANDS * 2 * N);
    HRIRs2HRTFs_afSTFT(p->hrirs.data(), N, p->hrir_runtime_len, SAF_HOP, 0, 1, reinterpret_cast<float_complex*>(p->hrtf_fb.data()));
    if (p->enableHRIRsDiffuseEQ) {
        strcpy(p->progressBarText, "Applying HRIR diffuse-field EQ"); p->progressBar0_1 = 0.9f;
        p->weights.resize(N);
        if (N <= 1000) voronoi_weights(p->hrir_dirs_deg.data(), N, p->weights.data());
        else for (int i = 0; i < N; i++) p->weights[i] = 4.f * SAF_PI / (float)N;
        diffuseFieldEqualiseHRTFs(N, p->itds_s.data(), p->freqVector, SAF_NBANDS, p->weights.data(), 1, 0, reinterpret_cast<float_complex*>(p->hrtf_fb.data()));
    }
    p->hrtf_fb_mag.resize(p->hrtf_fb.size());
    for (size_t i = 0; i < p->hrtf_fb.size(); i++) p->hrtf_fb_mag[i] = hypotf(p->hrtf_fb[i].x, p->hrtf_fb[i].y);
    for (int i = 0; i < p->maxSrc; i++) p->recalc_hrtf_interpFLAG[i] = 1;
    p->tablesEpoch++;
}

static void upload_tables(Binauraliser* p)
{
    if (p->tablesOnDevice == p->tablesEpoch) return;
    const int N = p->N_hrir_dirs;
    HIP_CHECK(hipStreamSynchronize(stream()));
    p->d_hrtf_fb.alloc(p->hrtf_fb.size(), false); p->d_mag.alloc(p->hrtf_fb_mag.size(), false); p->d_itds.alloc(N, false);
    p->d_gtComp.alloc(p->gtableComp.size(), false); p->d_gtIdx.alloc(p->gtableIdx.size(), false);
    HIP_CHECK(hipMemcpy(p->d_hrtf_fb.p, p->hrtf_fb.data(), sizeof(float2) * p->hrtf_fb.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(p->d_mag.p, p->hrtf_fb_mag.data(), sizeof(float) * p->hrtf_fb_mag.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(p->d_itds.p, p->itds_s.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(p->d_gtComp.p, p->gtableComp.data(), sizeof(float) * p->gtableComp.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(p->d_gtIdx.p, p->gtableIdx.data(), sizeof(int) * p->gtableIdx.size(), hipMemcpyHostToDevice));
    p->tablesOnDevice = p->tablesEpoch;
}

/* band centre frequencies on the device; they change only with the sample rate (binauraliser_init) */
static void upload_freq(Binauraliser* p)
{
    if (p->freqOnDevice.size() == SAF_NBANDS && memcmp(p->freqOnDevice.data(), p->freqVector, sizeof(p->freqVector)) == 0) return;
    HIP_CHECK(hipStreamSynchronize(stream()));
    HIP_CHECK(hipMemcpy(p->d_freq.p, p->freqVector, sizeof(p->freqVector), hipMemcpyHostToDevice));
    p->freqOnDevice.assign(p->freqVector, p->freqVector + SAF_NBANDS);
}

/* rotate source directions (binauraliser.c:230-248) */
static void rotate_sources(Binauraliser* p)
{
    const int nS = p->nSources;
    if (p->enableRotation && p->recalc_M_rotFLAG) {
        float R[3][3];
        rot_matrix(p->yaw, p->pitch, p->roll, p->useRollPitchYawFlag, R);
        for (int i = 0; i < nS; i++) {
            const float az = p->src_dirs_deg[i * 2] * SAF_PI / 180.0f, el = p->src_dirs_deg[i * 2 + 1] * SAF_PI / 180.0f;
            const float x[3] = { cosf(el) * cosf(az), cosf(el) * sinf(az), sinf(el) };
            float r[3];
            for (int j = 0; j < 3; j++) { float s = 0; for (int k = 0; k < 3; k++) s += x[k] * R[k][j]; r[j] = s; }
            const float hyp = sqrtf(powf(r[0], 2.0f) + powf(r[1], 2.0f));
            p->src_dirs_rot_deg[i * 2] = atan2f(r[1], r[0]) * 180.0f / SAF_PI;
            p->src_dirs_rot_deg[i * 2 + 1] = atan2f(r[2], hyp) * 180.0f / SAF_PI;
            p->recalc_hrtf_interpFLAG[i] = 1;
        }
        p->recalc_M_rotFLAG = 0;
    }
}

/* the block path for nFrames consecutive blocks of device-resident samples */
static void process_dev(Binauraliser* p, const float* d_in, long long in_frame, long long in_ch, int nIn,
                        float* d_out, long long out_frame, long long out_ch, int nOut, int nFrames, bool nearField = false)
{
    const int nS = p->nSources, T = p->T, H = nFrames * T;
    if (H > p->Hmax) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->Hmax = (H + 15) & ~15;
        p->X.alloc((size_t)SAF_NBANDS * p->maxSrc * p->Hmax, true);
        p->Y.alloc((size_t)SAF_NBANDS * 2 * p->Hmax, true);
    }
    upload_tables(p);
    /* source gains (binauraliser.c:221-224) as the analysis kernel's per-channel scale */
    {
        std::vector<float> g(p->maxSrc);
        for (int ch = 0; ch < p->maxSrc; ch++) g[ch] = fabsf(p->src_gains[ch] - 1.f) > 1e-6f ? p->src_gains[ch] : 1.0f;
        if (g != p->shadowGains) {
            HIP_CHECK(hipStreamSynchronize(stream()));
            memcpy(p->stF.p, g.data(), sizeof(float) * p->maxSrc);
            HIP_CHECK(hipMemcpyAsync(p->d_gains.p, p->stF.p, sizeof(float) * p->maxSrc, hipMemcpyHostToDevice, stream()));
            HIP_CHECK(hipStreamSynchronize(stream()));
            p->shadowGains = g;
        }
    }
    AnaLaunch a{};
    a.in = d_in; a.in_inst = 0; a.in_ch = in_ch; a.in_frame = in_frame; a.hopsPerFrame = T; a.nChIn = nS < nIn ? nS : nIn;
    a.hist_rd = p->st.ana[p->st.anaPar].p; a.hist_wr = p->st.ana[p->st.anaPar ^ 1].p;
    a.out = p->X.p; a.out_inst = 0; a.out_band = (long long)p->maxSrc * p->Hmax; a.out_ch = p->Hmax;
    a.ch_scale = p->d_gains.p; a.ch_map = nullptr; a.tab_stride = p->maxSrc;
    a.nCh = nS; a.nInst = 1; a.H = H; a.lowDelay = 0; a.hybrid = 1;
    launch_analysis(a);
    p->st.anaPar ^= 1;

    rotate_sources(p);
    /* interpolate the HRTFs of the sources that moved (binauraliser.c:252-260) */
    bool any = false;
    for (int ch = 0; ch < nS; ch++) any = any || p->recalc_hrtf_interpFLAG[ch];
    /* zero-copy mode: the small parameter tables are read by the kernels straight from the pinned staging blocks (no copies
     * on the stream); the host-pointer entry ends every call with a stream sync, so nothing is waited for here either */
    const bool zc = zero_copy_io();
    if (any) {
        if (p->stagingBusy) { HIP_CHECK(hipStreamSynchronize(stream())); p->stagingBusy = false; }
        const std::vector<float>& dirs = p->enableRotation ? p->src_dirs_rot_deg : p->src_dirs_deg;
        memcpy(p->stF.p, dirs.data(), sizeof(float) * 2 * nS);
        for (int ch = 0; ch < nS; ch++) {
            p->stI.p[ch] = p->recalc_hrtf_interpFLAG[ch];
            if (p->nf && p->recalc_hrtf_interpFLAG[ch]) { p->recalc_dvfCoeffFLAG[ch] = 1; p->curRot = p->enableRotation; p->dvfCoefOnDevice.clear(); }
            p->recalc_hrtf_interpFLAG[ch] = 0;
        }
        if (!zc) {
            HIP_CHECK(hipMemcpyAsync(p->d_dirs.p, p->stF.p, sizeof(float) * 2 * nS, hipMemcpyHostToDevice, stream()));
            HIP_CHECK(hipMemcpyAsync(p->d_recalc.p, p->stI.p, sizeof(int) * nS, hipMemcpyHostToDevice, stream()));
        }
        upload_freq(p);
        HrtfInterpLaunch l{};
        l.srcDirs = zc ? p->stF.p : p->d_dirs.p; l.recalc = zc ? p->stI.p : p->d_recalc.p; l.gtComp = p->d_gtComp.p; l.gtIdx = p->d_gtIdx.p;
        l.hrtf_fb = p->d_hrtf_fb.p; l.hrtf_mag = p->d_mag.p; l.itds = p->d_itds.p; l.freq = p->d_freq.p;
        l.hrtf_interp = p->d_hrtf_interp.p; l.nSrc = nS; l.N = p->N_hrir_dirs; l.mode = p->interpMode;
        l.aziRes = p->hrtf_vbapTableRes[0]; l.elevRes = p->hrtf_vbapTableRes[1];
        launch_hrtf_interp(l);
        p->stagingBusy = true;
    }
    const bool useNF = p->nf && nearField;
    if (useNF && nf_refresh_coeffs(p)) {
        if (p->stagingBusy && !any) { HIP_CHECK(hipStreamSynchronize(stream())); p->stagingBusy = false; }      /* stDvf is only read by this kernel */
        memcpy(p->stDvf.p, p->dvfCoef.data(), sizeof(float) * (size_t)nS * 8);
        if (!zc) HIP_CHECK(hipMemcpyAsync(p->d_dvfCoef.p, p->stDvf.p, sizeof(float) * (size_t)nS * 8, hipMemcpyHostToDevice, stream()));
        upload_freq(p);
        p->dvfCoefOnDevice = p->dvfCoef;
        DvfScaleLaunch d{};
        d.hrtf_interp = p->d_hrtf_interp.p; d.coef = zc ? p->stDvf.p : p->d_dvfCoef.p; d.freq = p->d_freq.p; d.hrtf_nf = p->d_hrtf_nf.p;
        d.fs = (float)p->fs; d.nSrc = nS;
        launch_dvf_scale(d);
        p->stagingBusy = true;
    }
    BinMacLaunch m{};
    m.X = p->X.p; m.x_band = a.out_band; m.x_ch = a.out_ch;
    m.h = useNF ? p->d_hrtf_nf.p : p->d_hrtf_interp.p;
    m.Y = p->Y.p; m.y_band = (long long)2 * p->Hmax; m.y_ch = p->Hmax;
    m.nSrc = nS; m.H = H; m.scale = 1.0f / sqrtf((float)nS);
    launch_binaural_mac(m);

    SynLaunch s{};
    s.in = p->Y.p; s.in_inst = 0; s.in_band = m.y_band; s.in_ch = m.y_ch;
    s.out = d_out; s.out_inst = 0; s.out_ch = out_ch; s.out_frame = out_frame; s.hopsPerFrame = T;
    s.hist_rd = p->st.syn[p->st.synPar].p; s.hist_wr = p->st.syn[p->st.synPar ^ 1].p;
    s.nCh = 2; s.nInst = 1; s.H = H; s.lowDelay = 0; s.hybrid = 1;
    (void)nOut;
    launch_synthesis(s);
    p->st.synPar ^= 1;
}

/* ---- batch of binauralisers: nInst initialised handles with the same block size, source count, HRIR tables and
 *      interpolation mode; every call advances all of them by nFrames blocks (own filterbank state, like the ambi_dec batch) ---- */
struct BinBatch {
    std::vector<Binauraliser*> inst;
    int nInst = 0, F = 0, T = 0, maxSrc = 0, nS = 0, maxFrames = 0, Hmax = 0;
    AfState st;
    DevBuf<float2> X, Y, hrtfInterp;
    DevBuf<float> dirs, gains;
    DevBuf<int> recalc;
    PinBuf<float> stD, stG; PinBuf<int> stR;
    std::vector<float> shadowGains;
    bool nf = false;                        /* binauraliser_nf instances: DVF-scaled HRTFs feed the MAC */
    DevBuf<float2> hrtfNf; DevBuf<float> dvfCoef; PinBuf<float> stK;

    void create(Binauraliser* const* h, int n, int maxFrames_)
    {
        nInst = n; inst.assign(h, h + n); maxFrames = maxFrames_;
        Binauraliser* p0 = inst[0];
        F = p0->F; T = p0->T; maxSrc = p0->maxSrc; nS = p0->nSources;
        for (int i = 0; i < n; i++) {
            Binauraliser* p = inst[i];
            if (p->hrtf_fb.empty() || p->codecStatus != CODEC_STATUS_INITIALISED) SAF_FATAL("binauraliser batch: instance %d is not initialised (call binauraliser_initCodec)", i);
            if (p->F != F || p->maxSrc != maxSrc || p->nSources != nS) SAF_FATAL("binauraliser batch: all instances must share block size, source cap and source count");
            if (p->N_hrir_dirs != p0->N_hrir_dirs || p->enableHRIRsDiffuseEQ != p0->enableHRIRsDiffuseEQ || p->fs != p0->fs)
                SAF_FATAL("binauraliser batch: all instances must share the HRIR set, the diffuse-field EQ flag and the sample rate");
            if (p->nf != p0->nf) SAF_FATAL("binauraliser batch: binauraliser and binauraliser_nf handles cannot be mixed");
        }
        nf = p0->nf;
        if (nf) { hrtfNf.alloc((size_t)n * maxSrc * SAF_NBANDS * 2); dvfCoef.alloc((size_t)n * maxSrc * 8); stK.ensure((size_t)n * maxSrc * 8); }
        Hmax = (T * maxFrames + 15) & ~15;
        st.create(nInst, nS, 2);
        X.alloc((size_t)nInst * SAF_NBANDS * maxSrc * Hmax, true);
        Y.alloc((size_t)nInst * SAF_NBANDS * 2 * Hmax, true);
        hrtfInterp.alloc((size_t)nInst * maxSrc * SAF_NBANDS * 2);
        dirs.alloc((size_t)nInst * maxSrc * 2); gains.alloc((size_t)nInst * maxSrc); recalc.alloc((size_t)nInst * maxSrc);
        stD.ensure((size_t)nInst * maxSrc * 2 + SAF_NBANDS); stG.ensure((size_t)nInst * maxSrc); stR.ensure((size_t)nInst * maxSrc);
        for (int i = 0; i < n; i++) for (int ch = 0; ch < maxSrc; ch++) inst[i]->recalc_hrtf_interpFLAG[ch] = 1;      /* a new pipeline starts without interpolated HRTFs */
    }

    void process(const float* d_in, long long in_inst, long long in_frame, long long in_ch, int nIn,
                 float* d_out, long long out_inst, long long out_frame, long long out_ch, int nFrames)
    {
        if (nFrames <= 0) return;
        if (nFrames > maxFrames) SAF_FATAL("binauraliser batch: nFrames %d exceeds the maxFramesPerCall %d given at creation", nFrames, maxFrames);
        Binauraliser* p0 = inst[0];
        upload_tables(p0);
        const int H = nFrames * T;
        /* source gains (binauraliser.c:221-224) */
        {
            std::vector<float> g((size_t)nInst * maxSrc);
            for (int i = 0; i < nInst; i++) for (int ch = 0; ch < maxSrc; ch++) g[(size_t)i * maxSrc + ch] = fabsf(inst[i]->src_gains[ch] - 1.f) > 1e-6f ? inst[i]->src_gains[ch] : 1.0f;
            if (g != shadowGains) {
                HIP_CHECK(hipStreamSynchronize(stream()));
                memcpy(stG.p, g.data(), sizeof(float) * g.size());
                HIP_CHECK(hipMemcpyAsync(gains.p, stG.p, sizeof(float) * g.size(), hipMemcpyHostToDevice, stream()));
                HIP_CHECK(hipStreamSynchronize(stream()));
                shadowGains = g;
            }
        }
        AnaLaunch a{};
        a.in = d_in; a.in_inst = in_inst; a.in_ch = in_ch; a.in_frame = in_frame; a.hopsPerFrame = T; a.nChIn = nS < nIn ? nS : nIn;
        a.hist_rd = st.ana[st.anaPar].p; a.hist_wr = st.ana[st.anaPar ^ 1].p;
        a.out = X.p; a.out_inst = (long long)SAF_NBANDS * maxSrc * Hmax; a.out_band = (long long)maxSrc * Hmax; a.out_ch = Hmax;
        a.ch_scale = gains.p; a.ch_map = nullptr; a.tab_stride = maxSrc;
        a.nCh = nS; a.nInst = nInst; a.H = H; a.lowDelay = 0; a.hybrid = 1;
        launch_analysis(a);
        st.anaPar ^= 1;
        /* head rotation and moved sources of every instance (binauraliser.c:230-260) */
        bool any = false;
        for (int i = 0; i < nInst; i++) {
            Binauraliser* p = inst[i];
            rotate_sources(p);
            for (int ch = 0; ch < nS; ch++) any = any || p->recalc_hrtf_interpFLAG[ch];
        }
        if (any) {
            HIP_CHECK(hipStreamSynchronize(stream()));
            for (int i = 0; i < nInst; i++) {
                Binauraliser* p = inst[i];
                const std::vector<float>& d = p->enableRotation ? p->src_dirs_rot_deg : p->src_dirs_deg;
                memcpy(stD.p + (size_t)i * maxSrc * 2, d.data(), sizeof(float) * 2 * maxSrc);
                for (int ch = 0; ch < maxSrc; ch++) {
                    stR.p[(size_t)i * maxSrc + ch] = ch < nS ? p->recalc_hrtf_interpFLAG[ch] : 0;
                    if (ch < nS && p->nf && p->recalc_hrtf_interpFLAG[ch]) { p->recalc_dvfCoeffFLAG[ch] = 1; p->curRot = p->enableRotation; p->dvfCoefOnDevice.clear(); }
                    if (ch < nS) p->recalc_hrtf_interpFLAG[ch] = 0;
                }
            }
            memcpy(stD.p + (size_t)nInst * maxSrc * 2, p0->freqVector, sizeof(float) * SAF_NBANDS);
            HIP_CHECK(hipMemcpyAsync(dirs.p, stD.p, sizeof(float) * (size_t)nInst * maxSrc * 2, hipMemcpyHostToDevice, stream()));
            HIP_CHECK(hipMemcpyAsync(recalc.p, stR.p, sizeof(int) * (size_t)nInst * maxSrc, hipMemcpyHostToDevice, stream()));
            HIP_CHECK(hipMemcpyAsync(p0->d_freq.p, stD.p + (size_t)nInst * maxSrc * 2, sizeof(float) * SAF_NBANDS, hipMemcpyHostToDevice, stream()));
            HrtfInterpLaunch l{};
            l.srcDirs = dirs.p; l.recalc = recalc.p; l.gtComp = p0->d_gtComp.p; l.gtIdx = p0->d_gtIdx.p;
            l.hrtf_fb = p0->d_hrtf_fb.p; l.hrtf_mag = p0->d_mag.p; l.itds = p0->d_itds.p; l.freq = p0->d_freq.p;
            l.hrtf_interp = hrtfInterp.p; l.nSrc = nS; l.N = p0->N_hrir_dirs; l.mode = p0->interpMode;
            l.aziRes = p0->hrtf_vbapTableRes[0]; l.elevRes = p0->hrtf_vbapTableRes[1];
            l.nInst = nInst; l.srcStride = maxSrc;
            launch_hrtf_interp(l);
        }
        if (nf) {
            bool stale = false;
            for (int i = 0; i < nInst; i++) stale = nf_refresh_coeffs(inst[i]) || stale;
            if (stale) {
                HIP_CHECK(hipStreamSynchronize(stream()));
                for (int i = 0; i < nInst; i++) {
                    memcpy(stK.p + (size_t)i * maxSrc * 8, inst[i]->dvfCoef.data(), sizeof(float) * (size_t)maxSrc * 8);
                    inst[i]->dvfCoefOnDevice = inst[i]->dvfCoef;
                }
                HIP_CHECK(hipMemcpyAsync(dvfCoef.p, stK.p, sizeof(float) * (size_t)nInst * maxSrc * 8, hipMemcpyHostToDevice, stream()));
                DvfScaleLaunch d{};
                d.hrtf_interp = hrtfInterp.p; d.coef = dvfCoef.p; d.freq = p0->d_freq.p; d.hrtf_nf = hrtfNf.p;
                d.fs = (float)p0->fs; d.nSrc = nS; d.nInst = nInst; d.srcStride = maxSrc;
                launch_dvf_scale(d);
            }
        }
        BinMacLaunch m{};
        m.X = X.p; m.x_inst = a.out_inst; m.x_band = a.out_band; m.x_ch = a.out_ch;
        m.h = nf ? hrtfNf.p : hrtfInterp.p; m.h_inst = (long long)maxSrc * SAF_NBANDS * 2;
        m.Y = Y.p; m.y_inst = (long long)SAF_NBANDS * 2 * Hmax; m.y_band = (long long)2 * Hmax; m.y_ch = Hmax;
        m.nSrc = nS; m.H = H; m.scale = 1.0f / sqrtf((float)nS); m.nInst = nInst;
        launch_binaural_mac(m);
        SynLaunch s{};
        s.in = Y.p; s.in_inst = m.y_inst; s.in_band = m.y_band; s.in_ch = m.y_ch;
        s.out = d_out; s.out_inst = out_inst; s.out_ch = out_ch; s.out_frame = out_frame; s.hopsPerFrame = T;
        s.hist_rd = st.syn[st.synPar].p; s.hist_wr = st.syn[st.synPar ^ 1].p;
        s.nCh = 2; s.nInst = nInst; s.H = H; s.lowDelay = 0; s.hybrid = 1;
        launch_synthesis(s);
        st.synPar ^= 1;
    }
};

}  // namespace saf

using namespace saf;

extern "C" {

void saf_hip_binauraliser_setFrameSize(int frameSize)
{
    if (frameSize <= 0 || frameSize % SAF_HOP != 0) SAF_FATAL("binauraliser frame size must be a positive multiple of 128");
    g_bin_frame_size = frameSize;
}
void saf_hip_binauraliser_setMaxNumSources(int n)
{
    if (n < 1 || n > 1024) SAF_FATAL("binauraliser: the source cap must be in 1..1024");
    g_bin_max_sources = n;
}

void binauraliser_create(void** const phBin)
{
    Binauraliser* p = new Binauraliser();
    *phBin = p;
    p->F = g_bin_frame_size; p->T = p->F / SAF_HOP; p->maxSrc = g_bin_max_sources;
    p->src_dirs_deg.assign((size_t)p->maxSrc * 2, 0.0f); p->src_dirs_rot_deg.assign((size_t)p->maxSrc * 2, 0.0f);
    p->src_gains.assign(p->maxSrc, 1.0f);
    p->recalc_hrtf_interpFLAG.assign(p->maxSrc, 1);
    /* SOURCE_CONFIG_PRESET_DEFAULT of binauraliser_loadPreset (binauraliser_internal.c:291-300): one source at (0, 0) */
    p->new_nSources = p->nSources = 1;
    p->codecStatus = CODEC_STATUS_NOT_INITIALISED; p->procStatus = PROC_STATUS_NOT_ONGOING;
    p->progressBarText[0] = 0;
    memset(p->freqVector, 0, sizeof(p->freqVector));
}

void binauraliser_destroy(void** const phBin)
{
    Binauraliser* p = (Binauraliser*)*phBin;
    if (!p) return;
    while (p->codecStatus == CODEC_STATUS_INITIALISING || p->procStatus == PROC_STATUS_ONGOING) bsleep_ms(10);
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete p;
    *phBin = nullptr;
}

void binauraliser_init(void* const hBin, int sampleRate)
{
    Binauraliser* p = (Binauraliser*)hBin;
    p->fs = sampleRate;
    if (!p->haveSTFT) afSTFT_getCentreFreqs(nullptr, (float)sampleRate, SAF_NBANDS, p->freqVector);      /* NULL-handle table branch (afSTFTlib.c:554-563) */
    else {   /* valid-handle branch (afSTFTlib.c:565-587), hop 128 hybrid */
        static const float w[9] = { 1.0f, 0.7501f, 1.2499f, 0.8751f, 1.1249f, 0.9167f, 1.0833f, 0.9375f, 1.0625f };
        static const int bin[9] = { 0, 1, 1, 2, 2, 3, 3, 4, 4 };
        for (int i = 0; i < 9; i++) p->freqVector[i] = w[i] * ((float)bin[i] * (float)sampleRate / 256.0f);
        for (int i = 9, j = 5; i < SAF_NBANDS; i++, j++) p->freqVector[i] = (float)j * (float)sampleRate / 256.0f;
    }
    if (p->hrir_runtime_fs != p->fs) { p->reInitHRTFsAndGainTables = 1; set_codec_status(p, CODEC_STATUS_NOT_INITIALISED); }
    p->recalc_M_rotFLAG = 1;
}

void binauraliser_initCodec(void* const hBin)
{
    Binauraliser* p = (Binauraliser*)hBin;
    if (p->codecStatus != CODEC_STATUS_NOT_INITIALISED) return;
    while (p->procStatus == PROC_STATUS_ONGOING) { p->codecStatus = CODEC_STATUS_INITIALISING; bsleep_ms(10); }
    ensure_device();
    p->codecStatus = CODEC_STATUS_INITIALISING;
    strcpy(p->progressBarText, "Initialising"); p->progressBar0_1 = 0.0f;
    /* binauraliser_initTFT (binauraliser_internal.c:265-279) */
    HIP_CHECK(hipStreamSynchronize(stream()));
    if (!p->haveSTFT) {
        p->st.create(1, p->new_nSources, 2);
        p->d_hrtf_interp.alloc((size_t)p->maxSrc * SAF_NBANDS * 2);
        p->d_dirs.alloc((size_t)p->maxSrc * 2); p->d_recalc.alloc(p->maxSrc); p->d_gains.alloc(p->maxSrc); p->d_freq.alloc(SAF_NBANDS);
        p->stF.ensure((size_t)2 * p->maxSrc + SAF_NBANDS); p->stI.ensure(p->maxSrc);
        if (p->nf) { p->d_hrtf_nf.alloc((size_t)p->maxSrc * SAF_NBANDS * 2); p->d_dvfCoef.alloc((size_t)p->maxSrc * 8); p->stDvf.ensure((size_t)p->maxSrc * 8 + SAF_NBANDS); }
        p->haveSTFT = true;
    } else if (p->new_nSources != p->nSources) { p->st.channelChange(p->new_nSources, 2); p->st.clear(); }
    p->nSources = p->new_nSources;
    if (p->reInitHRTFsAndGainTables) { init_hrtfs_and_tables(p); p->reInitHRTFsAndGainTables = 0; }
    strcpy(p->progressBarText, "Done!"); p->progressBar0_1 = 1.0f;
    p->codecStatus = CODEC_STATUS_INITIALISED;
}

static void process_host(Binauraliser* p, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples, bool nearField)
{
    const int F = p->F, nS = p->nSources;
    if (nSamples == F && !p->hrtf_fb.empty() && p->codecStatus == CODEC_STATUS_INITIALISED) {
        p->procStatus = PROC_STATUS_ONGOING;
        const int nIn = nS < nInputs ? nS : (nInputs < 0 ? 0 : nInputs);
        p->h_in.ensure((size_t)p->maxSrc * F); p->h_out.ensure((size_t)2 * F);
        if (p->d_in.n < (size_t)p->maxSrc * F) p->d_in.alloc((size_t)p->maxSrc * F, true);
        for (int i = 0; i < nIn; i++) memcpy(p->h_in.p + (size_t)i * F, inputs[i], sizeof(float) * F);
        DevBuf<float>& o = p->d_out;
        if (o.n < (size_t)2 * F) o.alloc((size_t)2 * F, false);
        /* always synthesise both ears; the copy-out below honours nOutputs (binauraliser.c:274-277) */
        if (zero_copy_io()) process_dev(p, p->h_in.p, 0, F, nIn, p->h_out.p, 0, F, 2, 1, nearField);                 /* kernels on the pinned blocks */
        else {
            if (nIn) HIP_CHECK(hipMemcpyAsync(p->d_in.p, p->h_in.p, sizeof(float) * (size_t)nIn * F, hipMemcpyHostToDevice, stream()));
            float* d_o = o.p;
            process_dev(p, p->d_in.p, 0, F, nIn, d_o, 0, F, 2, 1, nearField);
            HIP_CHECK(hipMemcpyAsync(p->h_out.p, d_o, sizeof(float) * (size_t)2 * F, hipMemcpyDeviceToHost, stream()));
        }
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->stagingBusy = false;
        int ch;
        for (ch = 0; ch < (2 < nOutputs ? 2 : nOutputs); ch++) memcpy(outputs[ch], p->h_out.p + (size_t)ch * F, sizeof(float) * F);
        for (; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    } else
        for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);      /* binauraliser.c:279-282 */
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}

void binauraliser_process(void* const hBin, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)
{
    process_host((Binauraliser*)hBin, inputs, outputs, nInputs, nOutputs, nSamples, false);
}

void saf_hip_binauraliser_process_dev(void* const hBin, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                      float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames)
{
    Binauraliser* p = (Binauraliser*)hBin;
    if (p->hrtf_fb.empty() || p->codecStatus != CODEC_STATUS_INITIALISED) SAF_FATAL("binauraliser: process_dev on a handle that is not initialised (call binauraliser_initCodec)");
    p->procStatus = PROC_STATUS_ONGOING;
    process_dev(p, d_in, in_frame_stride, in_ch_stride, nInputs < 0 ? 0 : nInputs, d_out, out_frame_stride, out_ch_stride, 2, nFrames);
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}

/* ------------------------------- set functions (binauraliser.c:289-470) ------------------------------- */
#define PBN Binauraliser* p = (Binauraliser*)hBin
void binauraliser_refreshSettings(void* const hBin)
{
    PBN;
    p->reInitHRTFsAndGainTables = 1;
    for (int ch = 0; ch < p->maxSrc; ch++) p->recalc_hrtf_interpFLAG[ch] = 1;
    set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
}
void binauraliser_setSourceAzi_deg(void* const hBin, int index, float v)
{
    PBN;
    if (v > 180.0f) v = -360.0f + v;
    v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    if (p->src_dirs_deg[index * 2] != v) { p->src_dirs_deg[index * 2] = v; p->recalc_hrtf_interpFLAG[index] = 1; p->recalc_M_rotFLAG = 1; }
}
void binauraliser_setSourceElev_deg(void* const hBin, int index, float v)
{
    PBN;
    v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    if (p->src_dirs_deg[index * 2 + 1] != v) { p->src_dirs_deg[index * 2 + 1] = v; p->recalc_hrtf_interpFLAG[index] = 1; p->recalc_M_rotFLAG = 1; }
}
void binauraliser_setNumSources(void* const hBin, int n)
{
    PBN;
    p->new_nSources = n < 1 ? 1 : (n > p->maxSrc ? p->maxSrc : n);
    p->recalc_M_rotFLAG = 1;
    set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
}
void binauraliser_setUseDefaultHRIRsflag(void* const hBin, int newState) { PBN; if (!p->useDefaultHRIRsFLAG && newState) { p->useDefaultHRIRsFLAG = newState; binauraliser_refreshSettings(hBin); } }
void binauraliser_setSofaFilePath(void* const hBin, const char* path) { PBN; p->sofa_filepath = path; p->useDefaultHRIRsFLAG = 0; binauraliser_refreshSettings(hBin); }
void binauraliser_setEnableHRIRsDiffuseEQ(void* const hBin, int newState) { PBN; if (newState != p->enableHRIRsDiffuseEQ) { p->enableHRIRsDiffuseEQ = newState; binauraliser_refreshSettings(hBin); } }
void binauraliser_setInputConfigPreset(void* const hBin, int newPresetID)
{
    PBN;
    float dirs[SAF_MAXCH][2]; int n = 1;
    if (newPresetID <= 1) { for (int ch = 0; ch < SAF_MAXCH; ch++) dirs[ch][0] = dirs[ch][1] = 0.0f; }    /* default: one source, all zero */
    else load_source_preset(newPresetID, dirs, &n);
    for (int ch = 0; ch < SAF_MAXCH && ch < p->maxSrc; ch++) { p->src_dirs_deg[ch * 2] = dirs[ch][0]; p->src_dirs_deg[ch * 2 + 1] = dirs[ch][1]; }
    p->new_nSources = n;
    if (p->nSources != p->new_nSources) set_codec_status(p, CODEC_STATUS_NOT_INITIALISED);
    for (int ch = 0; ch < p->maxSrc; ch++) p->recalc_hrtf_interpFLAG[ch] = 1;
}
void binauraliser_setEnableRotation(void* const hBin, int newState) { PBN; p->enableRotation = newState; if (!p->enableRotation) for (int ch = 0; ch < p->maxSrc; ch++) p->recalc_hrtf_interpFLAG[ch] = 1; }
void binauraliser_setYaw(void* const hBin, float v) { PBN; p->yaw = p->bFlipYaw == 1 ? -(v * SAF_PI / 180.0f) : v * SAF_PI / 180.0f; p->recalc_M_rotFLAG = 1; }
void binauraliser_setPitch(void* const hBin, float v) { PBN; p->pitch = p->bFlipPitch == 1 ? -(v * SAF_PI / 180.0f) : v * SAF_PI / 180.0f; p->recalc_M_rotFLAG = 1; }
void binauraliser_setRoll(void* const hBin, float v) { PBN; p->roll = p->bFlipRoll == 1 ? -(v * SAF_PI / 180.0f) : v * SAF_PI / 180.0f; p->recalc_M_rotFLAG = 1; }
float binauraliser_getYaw(void* const hBin) { PBN; return p->bFlipYaw ? -(p->yaw * 180.0f / SAF_PI) : p->yaw * 180.0f / SAF_PI; }
float binauraliser_getPitch(void* const hBin) { PBN; return p->bFlipPitch ? -(p->pitch * 180.0f / SAF_PI) : p->pitch * 180.0f / SAF_PI; }
float binauraliser_getRoll(void* const hBin) { PBN; return p->bFlipRoll ? -(p->roll * 180.0f / SAF_PI) : p->roll * 180.0f / SAF_PI; }
void binauraliser_setFlipYaw(void* const hBin, int s) { PBN; if (s != p->bFlipYaw) { p->bFlipYaw = s; binauraliser_setYaw(hBin, -binauraliser_getYaw(hBin)); } }
void binauraliser_setFlipPitch(void* const hBin, int s) { PBN; if (s != p->bFlipPitch) { p->bFlipPitch = s; binauraliser_setPitch(hBin, -binauraliser_getPitch(hBin)); } }
void binauraliser_setFlipRoll(void* const hBin, int s) { PBN; if (s != p->bFlipRoll) { p->bFlipRoll = s; binauraliser_setRoll(hBin, -binauraliser_getRoll(hBin)); } }
void binauraliser_setRPYflag(void* const hBin, int s) { PBN; p->useRollPitchYawFlag = s; }
void binauraliser_setInterpMode(void* const hBin, int m) { PBN; p->interpMode = m; for (int ch = 0; ch < p->maxSrc; ch++) p->recalc_hrtf_interpFLAG[ch] = 1; }
void binauraliser_setSourceGain(void* const hBin, int srcIdx, float g) { PBN; p->src_gains[srcIdx] = g; }
void binauraliser_setSourceSolo(void* const hBin, int srcIdx) { PBN; for (int i = 0; i < p->nSources; i++) p->src_gains[i] = i == srcIdx ? 1.f : 0.f; }
void binauraliser_setUnSolo(void* const hBin) { PBN; for (int i = 0; i < p->nSources; i++) p->src_gains[i] = 1.f; }

/* ------------------------------- get functions (binauraliser.c:473-640) ------------------------------- */
int binauraliser_getFrameSize(void) { return g_bin_frame_size; }
CODEC_STATUS binauraliser_getCodecStatus(void* const hBin) { PBN; return p->codecStatus; }
float binauraliser_getProgressBar0_1(void* const hBin) { PBN; return p->progressBar0_1; }
void binauraliser_getProgressBarText(void* const hBin, char* text) { PBN; memcpy(text, p->progressBarText, PROGRESSBARTEXT_CHAR_LENGTH); }
float binauraliser_getSourceAzi_deg(void* const hBin, int index) { PBN; return p->src_dirs_deg[index * 2]; }
float binauraliser_getSourceElev_deg(void* const hBin, int index) { PBN; return p->src_dirs_deg[index * 2 + 1]; }
int binauraliser_getNumSources(void* const hBin) { PBN; return p->new_nSources; }
int binauraliser_getMaxNumSources(void) { return g_bin_max_sources; }
int binauraliser_getNumEars(void) { return 2; }
int binauraliser_getNDirs(void* const hBin) { PBN; return p->N_hrir_dirs; }
int binauraliser_getNTriangles(void* const hBin) { PBN; return p->nTriangles; }
float binauraliser_getHRIRAzi_deg(void* const hBin, int index) { PBN; return p->hrir_dirs_deg.empty() ? 0.0f : p->hrir_dirs_deg[index * 2]; }
float binauraliser_getHRIRElev_deg(void* const hBin, int index) { PBN; return p->hrir_dirs_deg.empty() ? 0.0f : p->hrir_dirs_deg[index * 2 + 1]; }
int binauraliser_getHRIRlength(void* const hBin) { PBN; return p->hrir_loaded_len; }
int binauraliser_getHRIRsamplerate(void* const hBin) { PBN; return p->hrir_loaded_fs; }
int binauraliser_getUseDefaultHRIRsflag(void* const hBin) { PBN; return p->useDefaultHRIRsFLAG; }
char* binauraliser_getSofaFilePath(void* const hBin) { PBN; return p->sofa_filepath.empty() ? (char*)"no_file" : (char*)p->sofa_filepath.c_str(); }
int binauraliser_getEnableHRIRsDiffuseEQ(void* const hBin) { PBN; return p->enableHRIRsDiffuseEQ; }
int binauraliser_getDAWsamplerate(void* const hBin) { PBN; return p->fs; }
int binauraliser_getEnableRotation(void* const hBin) { PBN; return p->enableRotation; }
int binauraliser_getFlipYaw(void* const hBin) { PBN; return p->bFlipYaw; }
int binauraliser_getFlipPitch(void* const hBin) { PBN; return p->bFlipPitch; }
int binauraliser_getFlipRoll(void* const hBin) { PBN; return p->bFlipRoll; }
int binauraliser_getRPYflag(void* const hBin) { PBN; return p->useRollPitchYawFlag; }
int binauraliser_getInterpMode(void* const hBin) { PBN; return p->interpMode; }
int binauraliser_getProcessingDelay(void) { return 12 * SAF_HOP; }

/* table read-back for parity checks (what binauraliser_data holds, binauraliser_internal.h:95-118) */
void saf_hip_binauraliser_getITDs(void* const hBin, float* itds_s) { PBN; memcpy(itds_s, p->itds_s.data(), sizeof(float) * p->itds_s.size()); }
void saf_hip_binauraliser_getWeights(void* const hBin, float* w) { PBN; memcpy(w, p->weights.data(), sizeof(float) * p->weights.size()); }
void saf_hip_binauraliser_getHRTFfb(void* const hBin, float_complex* hrtf_fb) { PBN; memcpy((void*)hrtf_fb, p->hrtf_fb.data(), sizeof(float2) * p->hrtf_fb.size()); }
void saf_hip_binauraliser_getHRTFinterp(void* const hBin, float_complex* hrtf_interp)
{
    PBN;
    HIP_CHECK(hipStreamSynchronize(stream()));
    HIP_CHECK(hipMemcpy((void*)hrtf_interp, p->d_hrtf_interp.p, sizeof(float2) * (size_t)p->nSources * SAF_NBANDS * 2, hipMemcpyDeviceToHost));
}

/* ------------------------------- binauraliser_nf (examples/include/binauraliser_nf.h:76-194) ------------------------------- */
static void nf_reset_distances(Binauraliser* p)         /* binauraliserNF_resetSourceDistances (binauraliser_nf_internal.c:62-70) */
{
    for (int i = 0; i < p->maxSrc; i++) p->src_dists_m[i] = p->farfield_thresh_m * p->farfield_headroom;
}
void binauraliserNF_create(void** const phBin)
{
    binauraliser_create(phBin);
    Binauraliser* p = (Binauraliser*)*phBin;
    p->nf = true;
    /* binauraliser_nf.c:64-80: head radius of the DVF model, far field from rho = 34 (about 3.09 m), stable down to 0.15 m */
    p->head_radius_recip = 1.f / p->head_radius;
    p->farfield_thresh_m = p->head_radius * 34.f;
    p->src_dists_m.assign(p->maxSrc, 0.0f);
    nf_reset_distances(p);
    p->recalc_dvfCoeffFLAG.assign(p->maxSrc, 1);
    p->dvfCoef.assign((size_t)p->maxSrc * 8, 0.0f);
}
void binauraliserNF_destroy(void** const phBin) { binauraliser_destroy(phBin); }
void binauraliserNF_init(void* const hBin, int sampleRate) { binauraliser_init(hBin, sampleRate); }       /* binauraliser_nf.c:172-178 */
void binauraliserNF_initCodec(void* const hBin) { binauraliser_initCodec(hBin); }                         /* binauraliser_nf.c:187-224: the same steps with the 2-channel synthesis */
void binauraliserNF_process(void* const hBin, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)
{
    Binauraliser* p = (Binauraliser*)hBin;
    if (!p->nf) SAF_FATAL("binauraliserNF_process on a handle that was not made by binauraliserNF_create");
    process_host(p, inputs, outputs, nInputs, nOutputs, nSamples, true);
}
/* binauraliser_nf.h:135 declares this "alternate version that performs frequency-domain DVF filtering"; the reference's
 * source never defines it — its binauraliserNF_process IS the frequency-domain version (binauraliser_nf.c:224) — so both
 * names are the same function here. */
void binauraliserNF_processFD(void* const hBin, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)
{
    binauraliserNF_process(hBin, inputs, outputs, nInputs, nOutputs, nSamples);
}
void saf_hip_binauraliserNF_process_dev(void* const hBin, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                        float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames)
{
    Binauraliser* p = (Binauraliser*)hBin;
    if (!p->nf) SAF_FATAL("binauraliserNF process_dev on a handle that was not made by binauraliserNF_create");
    if (p->hrtf_fb.empty() || p->codecStatus != CODEC_STATUS_INITIALISED) SAF_FATAL("binauraliserNF: process_dev on a handle that is not initialised (call binauraliserNF_initCodec)");
    p->procStatus = PROC_STATUS_ONGOING;
    process_dev(p, d_in, in_frame_stride, in_ch_stride, nInputs < 0 ? 0 : nInputs, d_out, out_frame_stride, out_ch_stride, 2, nFrames, true);
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}
void binauraliserNF_setSourceDist_m(void* const hBin, int index, float newDist_m)       /* binauraliser_nf.c:372-380 */
{
    PBN;
    newDist_m = newDist_m > p->nearfield_limit_m ? newDist_m : p->nearfield_limit_m;
    if (p->src_dists_m[index] != newDist_m) { p->src_dists_m[index] = newDist_m; p->recalc_dvfCoeffFLAG[index] = 1; }
}
void binauraliserNF_setInputConfigPreset(void* const hBin, int newPresetID)               /* binauraliser_nf.c:382-398: presets put the sources back in the far field */
{
    PBN;
    binauraliser_setInputConfigPreset(hBin, newPresetID);
    nf_reset_distances(p);
    for (int ch = 0; ch < p->maxSrc; ch++) p->recalc_dvfCoeffFLAG[ch] = 1;
}
float binauraliserNF_getSourceDist_m(void* const hBin, int index) { PBN; return p->src_dists_m[index]; }
float binauraliserNF_getFarfieldThresh_m(void* const hBin) { PBN; return p->farfield_thresh_m; }
float binauraliserNF_getFarfieldHeadroom(void* const hBin) { PBN; return p->farfield_headroom; }
float binauraliserNF_getNearfieldLimit_m(void* const hBin) { PBN; return p->nearfield_limit_m; }
/* read-back for parity checks: the filters the band MAC applies, [nSources][133][2] (HRTF x DVF for near sources) */
void saf_hip_binauraliserNF_getHRTFnf(void* const hBin, float_complex* hrtf_nf)
{
    PBN;
    HIP_CHECK(hipStreamSynchronize(stream()));
    HIP_CHECK(hipMemcpy((void*)hrtf_nf, p->d_hrtf_nf.p, sizeof(float2) * (size_t)p->nSources * SAF_NBANDS * 2, hipMemcpyDeviceToHost));
}

void* saf_hip_binauraliser_batch_create(void* const* hBins, int nInst, int maxFramesPerCall)
{
    if (nInst < 1 || maxFramesPerCall < 1) SAF_FATAL("binauraliser batch: nInst and maxFramesPerCall must be positive");
    BinBatch* b = new BinBatch();
    b->create((Binauraliser* const*)hBins, nInst, maxFramesPerCall);
    return b;
}
void saf_hip_binauraliser_batch_destroy(void** const phBatch)
{
    BinBatch* b = (BinBatch*)*phBatch;
    if (!b) return;
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete b;
    *phBatch = nullptr;
}
void saf_hip_binauraliser_batch_process(void* const hBatch, const float* d_in, long long in_inst_stride, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                        float* d_out, long long out_inst_stride, long long out_frame_stride, long long out_ch_stride, int nFrames)
{
    ((BinBatch*)hBatch)->process(d_in, in_inst_stride, in_frame_stride, in_ch_stride, nInputs < 0 ? 0 : nInputs, d_out, out_inst_stride, out_frame_stride, out_ch_stride, nFrames);
}

}
