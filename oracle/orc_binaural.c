/*
 * orc_binaural.c — CPU restatement of the HRIR/HRTF processing of saf_hrir and of the binauraliser
 * operator (examples/src/binauraliser).  TEST INFRASTRUCTURE ONLY (see saf_oracle.h).
 *
 * The reference's default HRIR set (saf_default_hrirs.c) is absent from the checkout, so the set is
 * injected through orc_binauraliser_setHRIRs — the same arrays the reference would memcpy from
 * (binauraliser_internal.c:169-178).  Parity of this chain is therefore "unpinned" by reference-side data:
 * no reference test covers the binauraliser; it is pinned only by the closed-form checks in
 * tests/test_oracle_cpu.py.
 */
#include "saf_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_PI 3.14159265358979323846264338327950288f
#define NB 133
#define HOP 128

static float matlab_fmodf(float x, float y) { float t = fmodf(x, y); return t >= 0 ? t : t + y; }   /* saf_utility_misc.c:188-191 */

/* estimateITDs (saf_hrir.c:40-108): 750 Hz biquad low-pass (DAFX, direct form 2), full cross-correlation
 * (cxcorr, saf_utility_misc.c:193-223), ITD = (len - argmax - 1) / fs clamped to +-sqrt(2)/2 ms */
void orc_estimateITDs(const float* hrirs, int N_dirs, int hrir_len, int fs, float* itds_s)
{
    const float fc = 750.0f, Q = 0.7071f;
    const float K = tanf(ORC_PI * fc / (float)fs), KK = K * K, D = KK * Q + K + Q;
    const float b[3] = { (KK * Q) / D, (2.0f * KK * Q) / D, (KK * Q) / D };
    const float a[3] = { 1.0f, (2.0f * Q * (KK - 1.0f)) / D, (KK * Q - K + Q) / D };
    const int xl = 2 * hrir_len - 1;
    const float bound = sqrtf(2.0f) / 2e3f;
    float* xc = (float*)malloc(sizeof(float) * xl);
    float* L = (float*)malloc(sizeof(float) * hrir_len);
    float* R = (float*)malloc(sizeof(float) * hrir_len);
    for (int i = 0; i < N_dirs; i++) {
        float Wz1[2] = { 0, 0 }, Wz2[2] = { 0, 0 };
        for (int n = 0; n < hrir_len; n++)
            for (int j = 0; j < 2; j++) {
                const float wn = hrirs[((size_t)i * 2 + j) * hrir_len + n] - a[1] * Wz1[j] - a[2] * Wz2[j];
                const float y = b[0] * wn + b[1] * Wz1[j] + b[2] * Wz2[j];
                if (j == 0) L[n] = y; else R[n] = y;
                Wz2[j] = Wz1[j]; Wz1[j] = wn;
            }
        memset(xc, 0, sizeof(float) * xl);
        for (int m = 1; m <= xl; m++) {
            const int arg = m - hrir_len;
            const int lim = arg < 0 ? hrir_len + arg : hrir_len - arg;
            for (int n = 1; n <= lim; n++)
                xc[m - 1] += arg >= 0 ? L[arg + n - 1] * R[n - 1] : L[n - 1] * R[n - arg - 1];
        }
        int maxIdx = 0; float maxVal = 0.0f;
        for (int j = 0; j < xl; j++) if (xc[j] > maxVal) { maxIdx = j; maxVal = xc[j]; }
        float v = ((float)hrir_len - (float)maxIdx - 1.0f) / (float)fs;
        v = v > bound ? bound : v; v = v < -bound ? -bound : v;
        itds_s[i] = v;
    }
    free(xc); free(L); free(R);
}

/* diffuseFieldEqualiseHRTFs (saf_hrir.c:173-239), applyEQ = 1, applyPhase = 0 (as binauraliser_internal.c:244 uses it) */
void orc_diffuseFieldEqualiseHRTFs(int N_dirs, int N_bands, const float* weights, orc_cpx* hrtfs)
{
    for (int band = 0; band < N_bands; band++)
        for (int e = 0; e < 2; e++) {
            orc_cpx* h = &hrtfs[((size_t)band * 2 + e) * N_dirs];
            float acc = 0.0f;
            for (int j = 0; j < N_dirs; j++) {
                const float w = weights ? weights[j] : 4.f * ORC_PI / (float)N_dirs;
                acc += w / (4.f * ORC_PI) * powf(hypotf(h[j].re, h[j].im), 2.0f);
            }
            const float d = sqrtf(acc > 0.00001f ? acc : 0.00001f) + 2.23e-8f;
            for (int j = 0; j < N_dirs; j++) { h[j].re /= d; h[j].im /= d; }     /* ccdivf by a real */
        }
}

/* ------------------------------- binauraliser ------------------------------- */
typedef struct {
    int F, T, maxSrc, fs;
    float freqVector[NB];
    void* hSTFT;
    int haveSTFT;
    /* injected HRIR set */
    float* set_hrirs; float* set_dirs; int set_N, set_len, set_fs;
    /* runtime tables (binauraliser_internal.h:73-139) */
    int N_dirs, hrir_len, hrir_fs, N_gtable, nTriangles;
    float* hrir_dirs_deg; float* itds_s; float* weights; orc_cpx* hrtf_fb; float* hrtf_fb_mag;
    float* gtableComp; int* gtableIdx;
    orc_cpx* hrtf_interp;           /* [maxSrc][NB][2] */
    int* recalc;
    float* src_dirs_deg; float* src_gains;
    float* src_rot_deg;
    int nSources, new_nSources, interpMode, enableDiffEQ, enableRotation, recalcRot, useRPY, flip[3], codecReady, reinit;
    float ypr[3];
    /* binauraliser_nf (binauraliser_nf_internal.h:140-158) */
    int nf, curRot /* src_dirs_cur points at the rotated directions (binauraliser_nf.c:285-289) */; int* recalcDvf; float* src_dists_m; float* dvfmags; float* dvfphases; float* b_dvf; float* a_dvf;   /* [maxSrc][2][NB] x2, [maxSrc][2][2] x2 */
    float head_radius_recip, farfield_thresh_m, farfield_headroom, nearfield_limit_m;
} orc_bin;

void orc_binauraliser_create(void** ph, int frameSize, int maxSources)
{
    orc_bin* p = (orc_bin*)calloc(1, sizeof(orc_bin));
    p->F = frameSize; p->T = frameSize / HOP; p->maxSrc = maxSources; p->fs = 48000;
    p->hrtf_interp = (orc_cpx*)calloc((size_t)maxSources * NB * 2, sizeof(orc_cpx));
    p->recalc = (int*)malloc(sizeof(int) * maxSources);
    p->src_dirs_deg = (float*)calloc((size_t)maxSources * 2, sizeof(float));
    p->src_rot_deg = (float*)calloc((size_t)maxSources * 2, sizeof(float));
    p->src_gains = (float*)malloc(sizeof(float) * maxSources);
    for (int i = 0; i < maxSources; i++) { p->recalc[i] = 1; p->src_gains[i] = 1.0f; }
    p->nSources = p->new_nSources = 1;                      /* SOURCE_CONFIG_PRESET_DEFAULT: one source at (0, 0) */
    p->interpMode = 1 /* INTERP_TRI (binauraliser.h:58-61) */; p->enableDiffEQ = 1; p->reinit = 1; p->recalcRot = 1;
    *ph = p;
}
void orc_binauraliser_destroy(void** ph)
{
    orc_bin* p = (orc_bin*)*ph; if (!p) return;
    if (p->hSTFT) orc_afSTFT_destroy(&p->hSTFT);
    free(p->set_hrirs); free(p->set_dirs); free(p->hrir_dirs_deg); free(p->itds_s); free(p->weights); free(p->hrtf_fb); free(p->hrtf_fb_mag);
    free(p->gtableComp); free(p->gtableIdx); free(p->hrtf_interp); free(p->recalc); free(p->src_dirs_deg); free(p->src_rot_deg); free(p->src_gains);
    free(p->recalcDvf); free(p->src_dists_m); free(p->dvfmags); free(p->dvfphases); free(p->b_dvf); free(p->a_dvf);
    free(p); *ph = NULL;
}
void orc_binauraliser_setHRIRs(void* h, const float* hrirs, const float* dirs_deg, int N, int len, int fs)
{
    orc_bin* p = (orc_bin*)h;
    free(p->set_hrirs); free(p->set_dirs);
    p->set_hrirs = (float*)malloc(sizeof(float) * (size_t)N * 2 * len); memcpy(p->set_hrirs, hrirs, sizeof(float) * (size_t)N * 2 * len);
    p->set_dirs = (float*)malloc(sizeof(float) * (size_t)N * 2); memcpy(p->set_dirs, dirs_deg, sizeof(float) * (size_t)N * 2);
    p->set_N = N; p->set_len = len; p->set_fs = fs;
    p->reinit = 1; p->codecReady = 0;
}
/* binauraliser_init (binauraliser.c:133-150) */
void orc_binauraliser_init(void* h, int sampleRate)
{
    orc_bin* p = (orc_bin*)h;
    p->fs = sampleRate;
    orc_afSTFT_getCentreFreqs(p->haveSTFT ? p->hSTFT : NULL, (float)sampleRate, NB, p->freqVector);
    if (p->hrir_fs != p->fs) { p->reinit = 1; p->codecReady = 0; }
    p->recalcRot = 1;
}
/* binauraliser_initCodec (binauraliser.c:152-189) + binauraliser_initTFT / initHRTFsAndGainTables (binauraliser_internal.c:125-279) */
void orc_binauraliser_initCodec(void* h)
{
    orc_bin* p = (orc_bin*)h;
    if (p->codecReady) return;
    if (!p->hSTFT) orc_afSTFT_create(&p->hSTFT, p->new_nSources, 2, HOP, 0, 1, ORC_AFSTFT_BANDS_CH_TIME);
    else if (p->new_nSources != p->nSources) { orc_afSTFT_channelChange(p->hSTFT, p->new_nSources, 2); orc_afSTFT_clearBuffers(p->hSTFT); }
    p->haveSTFT = 1;
    p->nSources = p->new_nSources;
    if (p->reinit) {
        const int N = p->set_N, len = p->set_len;
        p->N_dirs = N; p->hrir_len = len; p->hrir_fs = p->set_fs;        /* no resampling path here: hrir fs must equal the host fs */
        p->hrir_dirs_deg = (float*)realloc(p->hrir_dirs_deg, sizeof(float) * 2 * N);
        memcpy(p->hrir_dirs_deg, p->set_dirs, sizeof(float) * 2 * N);
        for (int i = 0; i < N; i++) if (p->hrir_dirs_deg[i * 2] > 180.0f) p->hrir_dirs_deg[i * 2] = -360.0f + p->hrir_dirs_deg[i * 2];   /* convert_0_360To_m180_180 */
        p->itds_s = (float*)realloc(p->itds_s, sizeof(float) * N);
        orc_estimateITDs(p->set_hrirs, N, len, p->hrir_fs, p->itds_s);
        float* gtable = NULL;
        orc_generateVBAPgainTable3D(p->hrir_dirs_deg, N, 2, 5, 1, 0, 0.0f, &gtable, &p->N_gtable, &p->nTriangles);
        p->gtableComp = (float*)realloc(p->gtableComp, sizeof(float) * 3 * p->N_gtable);
        p->gtableIdx = (int*)realloc(p->gtableIdx, sizeof(int) * 3 * p->N_gtable);
        orc_compressVBAPgainTable3D(gtable, p->N_gtable, N, p->gtableComp, p->gtableIdx);
        free(gtable);
        p->hrtf_fb = (orc_cpx*)realloc(p->hrtf_fb, sizeof(orc_cpx) * (size_t)NB * 2 * N);
        orc_afSTFT_FIRtoFilterbankCoeffs(p->set_hrirs, N, 2, len, HOP, 0, 1, p->hrtf_fb);
        if (p->enableDiffEQ) {
            p->weights = (float*)realloc(p->weights, sizeof(float) * N);
            if (N <= 1000) orc_getVoronoiWeights(p->hrir_dirs_deg, N, p->weights);
            else for (int i = 0; i < N; i++) p->weights[i] = 4.f * ORC_PI / (float)N;
            orc_diffuseFieldEqualiseHRTFs(N, NB, p->weights, p->hrtf_fb);
        }
        p->hrtf_fb_mag = (float*)realloc(p->hrtf_fb_mag, sizeof(float) * (size_t)NB * 2 * N);
        for (size_t i = 0; i < (size_t)NB * 2 * N; i++) p->hrtf_fb_mag[i] = hypotf(p->hrtf_fb[i].re, p->hrtf_fb[i].im);
        for (int i = 0; i < p->maxSrc; i++) p->recalc[i] = 1;
        p->reinit = 0;
    }
    p->codecReady = 1;
}

/* binauraliser_interpHRTFs (binauraliser_internal.c:46-123) */
static void interp_hrtfs(orc_bin* p, int mode, float azi, float elev, orc_cpx* hout /* [NB][2] */)
{
    const float aziRes = 2.0f, elevRes = 5.0f;
    const int N_azi = (int)(360.0f / aziRes + 0.5f) + 1;
    const int aziIndex = (int)(matlab_fmodf(azi + 180.0f, 360.0f) / aziRes + 0.5f);
    const int elevIndex = (int)((elev + 90.0f) / elevRes + 0.5f);
    const int idx3d = elevIndex * N_azi + aziIndex;
    const float* w = &p->gtableComp[idx3d * 3]; const int* id = &p->gtableIdx[idx3d * 3];
    const int N = p->N_dirs;
    if (mode == 1) {      /* INTERP_TRI */
        for (int band = 0; band < NB; band++)
            for (int e = 0; e < 2; e++) {
                float re = 0.0f, im = 0.0f;
                for (int i = 0; i < 3; i++) { const orc_cpx v = p->hrtf_fb[((size_t)band * 2 + e) * N + id[i]]; re += v.re * w[i]; im += v.im * w[i]; }
                hout[band * 2 + e].re = re; hout[band * 2 + e].im = im;
            }
    } else {
        float itd = 0.0f;
        for (int i = 0; i < 3; i++) itd += w[i] * p->itds_s[id[i]];
        for (int band = 0; band < NB; band++) {
            float mag[2] = { 0, 0 };
            for (int i = 0; i < 3; i++) for (int e = 0; e < 2; e++) mag[e] += w[i] * p->hrtf_fb_mag[((size_t)band * 2 + e) * N + id[i]];
            const float ipd = p->freqVector[band] < 1.5e3f ? (matlab_fmodf(2.0f * ORC_PI * p->freqVector[band] * itd + ORC_PI, 2.0f * ORC_PI) - ORC_PI) / 2.0f : 0.0f;
            const float c = cosf(ipd), s = sinf(ipd);
            hout[band * 2 + 0].re = c * mag[0]; hout[band * 2 + 0].im = s * mag[0];
            hout[band * 2 + 1].re = c * mag[1]; hout[band * 2 + 1].im = -s * mag[1];
        }
    }
}

/* ambi_dec_interpHRTFs (ambi_dec_internal.c:59-115): the magnitude + ITD interpolation (identical to INTERP_TRI_PS of
 * binauraliser_interpHRTFs) on caller-held tables; 2 x 5 degree VBAP grid */
void orc_interpHRTFs_ps(const float* gtableComp, const int* gtableIdx, const float* itds_s, const float* hrtf_fb_mag, int N,
                        const float* freqVector, float azi, float elev, orc_cpx* hout /* [NB][2] */)
{
    const float aziRes = 2.0f, elevRes = 5.0f;
    const int N_azi = (int)(360.0f / aziRes + 0.5f) + 1;
    const int aziIndex = (int)(matlab_fmodf(azi + 180.0f, 360.0f) / aziRes + 0.5f);
    const int elevIndex = (int)((elev + 90.0f) / elevRes + 0.5f);
    const int idx3d = elevIndex * N_azi + aziIndex;
    const float* w = &gtableComp[idx3d * 3]; const int* id = &gtableIdx[idx3d * 3];
    float itd = 0.0f;
    for (int i = 0; i < 3; i++) itd += w[i] * itds_s[id[i]];
    for (int band = 0; band < NB; band++) {
        float mag[2] = { 0, 0 };
        for (int i = 0; i < 3; i++) for (int e = 0; e < 2; e++) mag[e] += w[i] * hrtf_fb_mag[((size_t)band * 2 + e) * N + id[i]];
        const float ipd = freqVector[band] < 1.5e3f ? (matlab_fmodf(2.0f * ORC_PI * freqVector[band] * itd + ORC_PI, 2.0f * ORC_PI) - ORC_PI) / 2.0f : 0.0f;
        const float c = cosf(ipd), s = sinf(ipd);
        hout[band * 2 + 0].re = c * mag[0]; hout[band * 2 + 0].im = s * mag[0];
        hout[band * 2 + 1].re = c * mag[1]; hout[band * 2 + 1].im = -s * mag[1];
    }
}

static void rot_mtx(float yaw, float pitch, float roll, int rpy, float R[3][3])    /* yawPitchRoll2Rzyx (saf_utility_geometry.c:213-270) */
{
    float Rx[3][3] = { { 1, 0, 0 }, { 0, cosf(roll), sinf(roll) }, { 0, -sinf(roll), cosf(roll) } };
    float Ry[3][3] = { { cosf(pitch), 0, -sinf(pitch) }, { 0, 1, 0 }, { sinf(pitch), 0, cosf(pitch) } };
    float Rz[3][3] = { { cosf(yaw), sinf(yaw), 0 }, { -sinf(yaw), cosf(yaw), 0 }, { 0, 0, 1 } };
    float (*R1)[3], (*R2)[3] = Ry, (*R3)[3];
    if (rpy) {      /* EULER_ROTATION_ROLL_PITCH_YAW is called with (yaw, pitch, roll) as (alpha, beta, gamma): Rx(alpha) Ry(beta) Rz(gamma) */
        float Rxa[3][3] = { { 1, 0, 0 }, { 0, cosf(yaw), sinf(yaw) }, { 0, -sinf(yaw), cosf(yaw) } };
        float Rzg[3][3] = { { cosf(roll), sinf(roll), 0 }, { -sinf(roll), cosf(roll), 0 }, { 0, 0, 1 } };
        memcpy(Rx, Rxa, sizeof(Rx)); memcpy(Rz, Rzg, sizeof(Rz));
        R1 = Rx; R3 = Rz;
    } else { R1 = Rz; R3 = Rx; }
    float T[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += R2[i][k] * R1[k][j]; T[i][j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += R3[i][k] * T[k][j]; R[i][j] = a; }
}

/* binauraliser_process (binauraliser.c:191-285); with p->nf the same loop as restated by binauraliserNF_process
 * (binauraliser_nf.c:226-367): DVF update after the HRTF interpolation and the per-source near / far choice */
static void bin_process(orc_bin* p, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)
{
    const int F = p->F, T = p->T, nS = p->nSources;
    if (nSamples != F || !p->hrtf_fb || !p->codecReady) { for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F); return; }
    float* inTD = (float*)calloc((size_t)nS * F, sizeof(float));
    for (int i = 0; i < (nS < nInputs ? nS : nInputs); i++) memcpy(&inTD[(size_t)i * F], inputs[i], sizeof(float) * F);
    for (int ch = 0; ch < nS; ch++)
        if (fabsf(p->src_gains[ch] - 1.f) > 1e-6f) for (int n = 0; n < F; n++) inTD[(size_t)ch * F + n] *= p->src_gains[ch];
    orc_cpx* inTF = (orc_cpx*)calloc((size_t)NB * nS * T, sizeof(orc_cpx));
    orc_afSTFT_forward_knownDimensions(p->hSTFT, inTD, F, nS, T, inTF);
    if (p->enableRotation && p->recalcRot) {
        float R[3][3];
        rot_mtx(p->ypr[0], p->ypr[1], p->ypr[2], p->useRPY, R);
        for (int i = 0; i < nS; i++) {
            const float az = p->src_dirs_deg[i * 2] * ORC_PI / 180.0f, el = p->src_dirs_deg[i * 2 + 1] * ORC_PI / 180.0f;
            const float x[3] = { cosf(el) * cosf(az), cosf(el) * sinf(az), sinf(el) };
            float r[3];
            for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += x[k] * R[k][j]; r[j] = a; }
            const float hyp = sqrtf(powf(r[0], 2.0f) + powf(r[1], 2.0f));
            p->src_rot_deg[i * 2] = atan2f(r[1], r[0]) * 180.0f / ORC_PI;
            p->src_rot_deg[i * 2 + 1] = atan2f(r[2], hyp) * 180.0f / ORC_PI;
            p->recalc[i] = 1;
        }
        p->recalcRot = 0;
    }
    orc_cpx* outTF = (orc_cpx*)calloc((size_t)NB * 2 * T, sizeof(orc_cpx));
    for (int ch = 0; ch < nS; ch++) {
        if (p->recalc[ch]) {
            const float* d = p->enableRotation ? &p->src_rot_deg[ch * 2] : &p->src_dirs_deg[ch * 2];
            interp_hrtfs(p, p->interpMode, d[0], d[1], &p->hrtf_interp[(size_t)ch * NB * 2]);
            p->recalc[ch] = 0;
            if (p->nf) { p->recalcDvf[ch] = 1; p->curRot = p->enableRotation; }
        }
        int near = 0;
        if (p->nf) {
            /* binauraliser_nf.c:299-318: shelf coefficients per ear from the lateral angle and the normalised distance, then their
             * response at the band centre frequencies */
            if (p->recalcDvf[ch]) {
                const float* d = p->curRot ? &p->src_rot_deg[ch * 2] : &p->src_dirs_deg[ch * 2];
                const float rho = p->src_dists_m[ch] * p->head_radius_recip;
                float alphaLR[2] = { 0.0f, 0.0f };
                orc_doaToIpsiInteraural(d[0], d[1], alphaLR, NULL);
                for (int e = 0; e < 2; e++) {
                    float* b = &p->b_dvf[(ch * 2 + e) * 2]; float* a = &p->a_dvf[(ch * 2 + e) * 2];
                    orc_calcDVFCoeffs(alphaLR[e], rho, (float)p->fs, b, a);
                    orc_evalIIRTransferFunctionf(b, a, 2, p->freqVector, NB, (float)p->fs, 0, &p->dvfmags[((size_t)ch * 2 + e) * NB], &p->dvfphases[((size_t)ch * 2 + e) * NB]);
                }
                p->recalcDvf[ch] = 0;
            }
            near = p->src_dists_m[ch] < p->farfield_thresh_m;
        }
        for (int band = 0; band < NB; band++)
            for (int e = 0; e < 2; e++) {
                orc_cpx a = p->hrtf_interp[((size_t)ch * NB + band) * 2 + e];
                if (near) {     /* binauraliser_nf.c:331: cmplxf(mag, phase) * hrtf — magnitude as the real and phase as the imaginary part */
                    const float mr = p->dvfmags[((size_t)ch * 2 + e) * NB + band], mi = p->dvfphases[((size_t)ch * 2 + e) * NB + band];
                    const orc_cpx hh = a;
                    a.re = mr * hh.re - mi * hh.im; a.im = mr * hh.im + mi * hh.re;
                }
                const orc_cpx* x = &inTF[((size_t)band * nS + ch) * T];
                orc_cpx* y = &outTF[((size_t)band * 2 + e) * T];
                for (int t = 0; t < T; t++) { y[t].re += a.re * x[t].re - a.im * x[t].im; y[t].im += a.re * x[t].im + a.im * x[t].re; }
            }
    }
    const float sc = 1.0f / sqrtf((float)nS);
    for (size_t i = 0; i < (size_t)NB * 2 * T; i++) { outTF[i].re *= sc; outTF[i].im *= sc; }
    float* outTD = (float*)calloc((size_t)2 * F, sizeof(float));
    orc_afSTFT_backward_knownDimensions(p->hSTFT, outTF, F, 2, T, outTD);
    int ch;
    for (ch = 0; ch < (2 < nOutputs ? 2 : nOutputs); ch++) memcpy(outputs[ch], &outTD[(size_t)ch * F], sizeof(float) * F);
    for (; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    free(inTD); free(inTF); free(outTF); free(outTD);
}
void orc_binauraliser_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)
{
    bin_process((orc_bin*)h, inputs, outputs, nInputs, nOutputs, nSamples);
}

/* ------------------------------- binauraliser_nf ------------------------------- */
/* binauraliserNF_create (binauraliser_nf.c:36-133): the binauraliser state plus distances and DVF responses */
void orc_binauraliserNF_create(void** ph, int frameSize, int maxSources)
{
    orc_binauraliser_create(ph, frameSize, maxSources);
    orc_bin* p = (orc_bin*)*ph;
    p->nf = 1;
    const float head_radius = 0.09096f;
    p->head_radius_recip = 1.f / head_radius;
    p->farfield_thresh_m = head_radius * 34.f;
    p->farfield_headroom = 1.05f;
    p->nearfield_limit_m = 0.15f;
    p->recalcDvf = (int*)malloc(sizeof(int) * maxSources);
    p->src_dists_m = (float*)malloc(sizeof(float) * maxSources);
    p->dvfmags = (float*)calloc((size_t)maxSources * 2 * NB, sizeof(float));
    p->dvfphases = (float*)calloc((size_t)maxSources * 2 * NB, sizeof(float));
    p->b_dvf = (float*)calloc((size_t)maxSources * 4, sizeof(float));
    p->a_dvf = (float*)calloc((size_t)maxSources * 4, sizeof(float));
    for (int i = 0; i < maxSources; i++) {
        p->recalcDvf[i] = 1;
        p->src_dists_m[i] = p->farfield_thresh_m * p->farfield_headroom;      /* binauraliserNF_resetSourceDistances (binauraliser_nf_internal.c:62-70) */
        p->a_dvf[(i * 2 + 0) * 2] = p->a_dvf[(i * 2 + 1) * 2] = 1.f;
    }
}
void orc_binauraliserNF_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)
{
    bin_process((orc_bin*)h, inputs, outputs, nInputs, nOutputs, nSamples);
}
/* binauraliserNF_setSourceDist_m (binauraliser_nf.c:372-380) */
void orc_binauraliserNF_setSourceDist_m(void* h, int i, float d)
{
    orc_bin* p = (orc_bin*)h;
    d = d > p->nearfield_limit_m ? d : p->nearfield_limit_m;
    if (p->src_dists_m[i] != d) { p->src_dists_m[i] = d; p->recalcDvf[i] = 1; }
}
float orc_binauraliserNF_getSourceDist_m(void* h, int i) { return ((orc_bin*)h)->src_dists_m[i]; }
float orc_binauraliserNF_getFarfieldThresh_m(void* h) { return ((orc_bin*)h)->farfield_thresh_m; }
float orc_binauraliserNF_getFarfieldHeadroom(void* h) { return ((orc_bin*)h)->farfield_headroom; }
float orc_binauraliserNF_getNearfieldLimit_m(void* h) { return ((orc_bin*)h)->nearfield_limit_m; }
const float* orc_binauraliserNF_getDVFmags(void* h) { return ((orc_bin*)h)->dvfmags; }
const float* orc_binauraliserNF_getDVFphases(void* h) { return ((orc_bin*)h)->dvfphases; }

/* setters (binauraliser.c:289-470) */
#define PB orc_bin* p = (orc_bin*)h
void orc_binauraliser_setSourceAzi_deg(void* h, int i, float v) { PB; if (v > 180.0f) v = -360.0f + v; v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    if (p->src_dirs_deg[i * 2] != v) { p->src_dirs_deg[i * 2] = v; p->recalc[i] = 1; p->recalcRot = 1; } }
void orc_binauraliser_setSourceElev_deg(void* h, int i, float v) { PB; v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    if (p->src_dirs_deg[i * 2 + 1] != v) { p->src_dirs_deg[i * 2 + 1] = v; p->recalc[i] = 1; p->recalcRot = 1; } }
void orc_binauraliser_setNumSources(void* h, int n) { PB; p->new_nSources = n < 1 ? 1 : (n > p->maxSrc ? p->maxSrc : n); p->recalcRot = 1; p->codecReady = 0; }
void orc_binauraliser_setEnableHRIRsDiffuseEQ(void* h, int s) { PB; if (s != p->enableDiffEQ) { p->enableDiffEQ = s; p->reinit = 1; for (int i = 0; i < p->maxSrc; i++) p->recalc[i] = 1; p->codecReady = 0; } }
void orc_binauraliser_setEnableRotation(void* h, int s) { PB; p->enableRotation = s; if (!s) for (int i = 0; i < p->maxSrc; i++) p->recalc[i] = 1; }
void orc_binauraliser_setYaw(void* h, float v) { PB; p->ypr[0] = (p->flip[0] ? -1.0f : 1.0f) * v * ORC_PI / 180.0f; p->recalcRot = 1; }
void orc_binauraliser_setPitch(void* h, float v) { PB; p->ypr[1] = (p->flip[1] ? -1.0f : 1.0f) * v * ORC_PI / 180.0f; p->recalcRot = 1; }
void orc_binauraliser_setRoll(void* h, float v) { PB; p->ypr[2] = (p->flip[2] ? -1.0f : 1.0f) * v * ORC_PI / 180.0f; p->recalcRot = 1; }
void orc_binauraliser_setRPYflag(void* h, int s) { PB; p->useRPY = s; }
void orc_binauraliser_setInterpMode(void* h, int m) { PB; p->interpMode = m; for (int i = 0; i < p->maxSrc; i++) p->recalc[i] = 1; }
void orc_binauraliser_setSourceGain(void* h, int i, float g) { PB; p->src_gains[i] = g; }
/* table read-back for parity checks */
int orc_binauraliser_getNDirs(void* h) { PB; return p->N_dirs; }
int orc_binauraliser_getNTriangles(void* h) { PB; return p->nTriangles; }
const float* orc_binauraliser_getITDs(void* h) { PB; return p->itds_s; }
const float* orc_binauraliser_getWeights(void* h) { PB; return p->weights; }
const orc_cpx* orc_binauraliser_getHRTFfb(void* h) { PB; return p->hrtf_fb; }
const orc_cpx* orc_binauraliser_getHRTFinterp(void* h) { PB; return p->hrtf_interp; }
