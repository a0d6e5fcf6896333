/*
 * orc_panner.c — CPU restatement of the reference's frequency-dependent VBAP panner
 * (examples/src/panner/panner.c, panner_internal.c) and getPvalues (saf_vbap.c:475-492).
 *
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  Parity status: the reference holds no test or golden vector for
 * panner_process or for saf_vbap (its test file is empty), and its BLAS dependency makes it unbuildable here:
 * "parity unpinned" beyond the pieces pinned elsewhere (afSTFT by test__afSTFT; triangulation face counts).
 */
#include "saf_oracle.h"
#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NB 133
#define HOP 128
#define MAXCH 64
#define ORC_PI 3.14159265358979323846264338327950288f   /* SAF_PI (saf_utilities.h:70) */

static float matlab_fmodf(float x, float y) { float t = fmodf(x, y); return t >= 0 ? t : t + y; }   /* saf_utility_misc.c:188-191 */

/* getPvalues (saf_vbap.c:475-492) */
void orc_getPvalues(float DTT, const float* freq, int nFreq, float* pValues)
{
    const float a1 = 0.00045f, a2 = 0.000085f;
    for (int i = 0; i < nFreq; i++) {
        const float lim = 1.0f - a2 * freq[i];
        const float p0 = 1.5f - 0.5f * cosf(4.7f * tanhf(a1 * freq[i])) * (lim > 0.0f ? lim : 0.0f);
        pValues[i] = (p0 - 2.0f) * sqrtf(DTT) + 2.0f;
    }
}

typedef struct {
    int F, T;
    void* hSTFT;
    int fs;
    float freqVector[NB], pValue[NB];
    float* vbap_gtable; int N_vbap_gtable, nTriangles;
    float* G_src;                       /* [NB][MAXCH][MAXCH] real (the reference stores complex with zero imaginary part) */
    int codecReady, reInitGainTables, recalcRot;
    int recalc[MAXCH];
    float src_dirs_deg[MAXCH][2], src_rot_deg[MAXCH][2], ls_dirs_deg[MAXCH][2];
    int nSources, new_nSources, nLoudpkrs, new_nLoudpkrs;
    float DTT, spread_deg, ypr[3];
    int flip[3];
} orc_pan;

/* panner_loadSourcePreset / panner_loadLoudspeakerPreset (panner_internal.c:119-518): ids follow _common.h; the
 * loudspeaker list is the source list without DEFAULT/MONO, so one table with the loudspeaker ids serves both */
static const struct { int id; const char* tab; int n; } g_ls[] = {
    { 2, "stereo_dirs_deg", 2 }, { 3, "5pX_dirs_deg", 5 }, { 4, "7pX_dirs_deg", 7 }, { 5, "8pX_dirs_deg", 8 }, { 6, "9pX_dirs_deg", 9 },
    { 7, "10pX_dirs_deg", 10 }, { 8, "11pX_dirs_deg", 11 }, { 9, "11pX_7_4_dirs_deg", 11 }, { 10, "13pX_dirs_deg", 13 },
    { 11, "22pX_dirs_deg", 22 }, { 12, "9_10_3p2_dirs_deg", 24 }, { 13, "Aalto_MCC_dirs_deg", 45 }, { 14, "Aalto_MCCsubset_dirs_deg", 37 },
    { 15, "Aalto_Apaja_dirs_deg", 29 }, { 16, "Aalto_LR_dirs_deg", 13 }, { 17, "DTU_AVIL_dirs_deg", 64 },
    { 19, "Tdesign_degree_2_dirs_deg", 4 }, { 20, "Tdesign_degree_4_dirs_deg", 12 }, { 21, "Tdesign_degree_6_dirs_deg", 24 },
    { 22, "Tdesign_degree_8_dirs_deg", 36 }, { 23, "Tdesign_degree_9_dirs_deg", 48 }, { 24, "Tdesign_degree_10_dirs_deg", 60 },
    { 25, "SphCovering_9_dirs_deg", 9 }, { 26, "SphCovering_16_dirs_deg", 16 }, { 27, "SphCovering_25_dirs_deg", 25 },
    { 28, "SphCovering_49_dirs_deg", 49 }, { 29, "SphCovering_64_dirs_deg", 64 },
};
static void fill_dirs(const char* tab, int n, float dirs[MAXCH][2])
{
    int d0, d1, ch;
    const float* t = orc_table(tab, &d0, &d1);
    assert(t);
    for (ch = 0; ch < n; ch++) { dirs[ch][0] = t[ch * 2]; dirs[ch][1] = t[ch * 2 + 1]; }
    const float* def = orc_table("default_LScoords64_rad", &d0, &d1);
    assert(def);
    for (; ch < MAXCH; ch++) for (int i = 0; i < 2; i++) dirs[ch][i] = def[ch * 2 + i] * (180.0f / ORC_PI);
}
static void load_ls_preset(int preset, float dirs[MAXCH][2], int* nCH)
{
    int found = 0;                                   /* default and unknown ids: stereo (panner_internal.c:337-345) */
    for (unsigned i = 0; i < sizeof(g_ls) / sizeof(g_ls[0]); i++) if (g_ls[i].id == preset) found = (int)i;
    fill_dirs(g_ls[found].tab, g_ls[found].n, dirs);
    *nCH = g_ls[found].n;
}
static void load_src_preset(int preset, float dirs[MAXCH][2], int* nCH)
{
    /* SOURCE_CONFIG_PRESETS: 1 default (one source at 0,0), 2 mono, 3 stereo, k >= 4 = loudspeaker preset k-1 */
    if (preset >= 3) { load_ls_preset(preset - 1, dirs, nCH); if (preset == 3 || *nCH != 2) return; }
    if (preset == 2) { fill_dirs("mono_dirs_deg", 1, dirs); *nCH = 1; return; }
    fill_dirs("mono_dirs_deg", 1, dirs); dirs[0][0] = dirs[0][1] = 0.0f; *nCH = 1;
}

/* panner_create (panner.c:46-91) */
void orc_panner_create(void** ph, int frameSize)
{
    orc_pan* p = (orc_pan*)calloc(1, sizeof(orc_pan));
    assert(frameSize % HOP == 0);
    p->F = frameSize; p->T = frameSize / HOP;
    load_src_preset(1, p->src_dirs_deg, &p->new_nSources); p->nSources = p->new_nSources;
    p->DTT = 0.5f; p->spread_deg = 0.0f;
    load_ls_preset(2, p->ls_dirs_deg, &p->new_nLoudpkrs); p->nLoudpkrs = p->new_nLoudpkrs;
    p->G_src = (float*)calloc((size_t)NB * MAXCH * MAXCH, sizeof(float));
    for (int ch = 0; ch < MAXCH; ch++) p->recalc[ch] = 1;
    p->recalcRot = 1; p->reInitGainTables = 1;
    *ph = p;
}
void orc_panner_destroy(void** ph)
{
    orc_pan* p = (orc_pan*)*ph; if (!p) return;
    if (p->hSTFT) orc_afSTFT_destroy(&p->hSTFT);
    free(p->vbap_gtable); free(p->G_src); free(p); *ph = NULL;
}
/* panner_init (panner.c:118-135) */
void orc_panner_init(void* h, int sampleRate)
{
    orc_pan* p = (orc_pan*)h;
    p->fs = sampleRate;
    orc_afSTFT_getCentreFreqs(p->hSTFT, (float)sampleRate, NB, p->freqVector);
    orc_getPvalues(p->DTT, p->freqVector, NB, p->pValue);
    p->recalcRot = 1;
}
/* panner_initCodec (panner.c:137-171), panner_initTFT and panner_initGainTables (panner_internal.c:59-117); FORCE_3D_LAYOUT is
 * defined (panner_internal.h:66), so the table is always the 3-D one at 1 x 1 degree with dummies and the large-triangle filter */
void orc_panner_initCodec(void* h)
{
    orc_pan* p = (orc_pan*)h;
    if (p->codecReady) return;
    if (!p->hSTFT) orc_afSTFT_create(&p->hSTFT, p->new_nSources, p->new_nLoudpkrs, HOP, 0, 1, ORC_AFSTFT_BANDS_CH_TIME);
    else if (p->new_nSources != p->nSources || p->new_nLoudpkrs != p->nLoudpkrs) {
        orc_afSTFT_channelChange(p->hSTFT, p->new_nSources, p->new_nLoudpkrs);
        orc_afSTFT_clearBuffers(p->hSTFT);
    }
    p->nSources = p->new_nSources; p->nLoudpkrs = p->new_nLoudpkrs;
    if (p->reInitGainTables) {
        free(p->vbap_gtable); p->vbap_gtable = NULL;
        orc_generateVBAPgainTable3D(&p->ls_dirs_deg[0][0], p->nLoudpkrs, 1, 1, 1, 1, p->spread_deg, &p->vbap_gtable, &p->N_vbap_gtable, &p->nTriangles);
        p->reInitGainTables = 0;
    }
    p->codecReady = 1;
}

static void rot_zyx(float yaw, float pitch, float roll, float R[3][3])       /* yawPitchRoll2Rzyx, rollPitchYawFLAG = 0 (saf_utility_geometry.c:213-270) */
{
    const float Rx[3][3] = { { 1, 0, 0 }, { 0, cosf(roll), sinf(roll) }, { 0, -sinf(roll), cosf(roll) } };
    const float Ry[3][3] = { { cosf(pitch), 0, -sinf(pitch) }, { 0, 1, 0 }, { sinf(pitch), 0, cosf(pitch) } };
    const float Rz[3][3] = { { cosf(yaw), sinf(yaw), 0 }, { -sinf(yaw), cosf(yaw), 0 }, { 0, 0, 1 } };
    float T[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += Ry[i][k] * Rz[k][j]; T[i][j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += Rx[i][k] * T[k][j]; R[i][j] = a; }
}

/* panner_process (panner.c:173-323), 3-D branch */
void orc_panner_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)
{
    orc_pan* p = (orc_pan*)h;
    const int F = p->F, T = p->T, nS = p->nSources, nL = p->nLoudpkrs;
    if (nSamples != F || !p->vbap_gtable || !p->codecReady) { for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F); return; }
    float* inTD = (float*)calloc((size_t)nS * F, sizeof(float));
    for (int i = 0; i < (nS < nInputs ? nS : nInputs); i++) memcpy(&inTD[(size_t)i * F], inputs[i], sizeof(float) * F);
    orc_cpx* inTF = (orc_cpx*)calloc((size_t)NB * nS * T, sizeof(orc_cpx));
    orc_afSTFT_forward_knownDimensions(p->hSTFT, inTD, F, nS, T, inTF);
    if (p->recalcRot) {
        float R[3][3];
        rot_zyx(p->ypr[0], p->ypr[1], p->ypr[2], R);
        for (int i = 0; i < nS; i++) {
            const float az = p->src_dirs_deg[i][0] * ORC_PI / 180.0f, el = p->src_dirs_deg[i][1] * ORC_PI / 180.0f;
            const float x[3] = { cosf(el) * cosf(az), cosf(el) * sinf(az), sinf(el) };
            float r[3];
            for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += x[k] * R[k][j]; r[j] = a; }
            const float hyp = sqrtf(powf(r[0], 2.0f) + powf(r[1], 2.0f));
            p->src_rot_deg[i][0] = atan2f(r[1], r[0]) * 180.0f / ORC_PI;
            p->src_rot_deg[i][1] = atan2f(r[2], hyp) * 180.0f / ORC_PI;
            p->recalc[i] = 1;
        }
        p->recalcRot = 0;
    }
    const int N_azi = (int)(360.0f / 1.0f + 0.5f) + 1;
    for (int ch = 0; ch < nS; ch++) {
        if (!p->recalc[ch]) continue;
        const int aziIndex = (int)(matlab_fmodf(p->src_rot_deg[ch][0] + 180.0f, 360.0f) / 1.0f + 0.5f);
        const int elevIndex = (int)((p->src_rot_deg[ch][1] + 90.0f) / 1.0f + 0.5f);
        const float* g = &p->vbap_gtable[(size_t)(elevIndex * N_azi + aziIndex) * nL];
        for (int band = 0; band < NB; band++) {
            float* G = &p->G_src[((size_t)band * MAXCH + ch) * MAXCH];
            const float pv = p->pValue[band];
            if (pv != 2.0f) {
                float s = 0.0f;
                for (int ls = 0; ls < nL; ls++) s += powf(g[ls] > 0.0f ? g[ls] : 0.0f, pv);
                s = powf(s, 1.0f / (pv + 2.23e-9f));
                for (int ls = 0; ls < nL; ls++) G[ls] = g[ls] / (s + 2.23e-9f);
            } else
                for (int ls = 0; ls < nL; ls++) G[ls] = g[ls];
        }
        p->recalc[ch] = 0;
    }
    orc_cpx* outTF = (orc_cpx*)calloc((size_t)NB * nL * T, sizeof(orc_cpx));
    const float sc = 1.0f / sqrtf((float)nS);
    for (int band = 0; band < NB; band++)
        for (int ls = 0; ls < nL; ls++) {
            orc_cpx* y = &outTF[((size_t)band * nL + ls) * T];
            for (int ch = 0; ch < nS; ch++) {
                const float g = p->G_src[((size_t)band * MAXCH + ch) * MAXCH + ls];
                const orc_cpx* x = &inTF[((size_t)band * nS + ch) * T];
                for (int t = 0; t < T; t++) { y[t].re += g * x[t].re; y[t].im += g * x[t].im; }
            }
            for (int t = 0; t < T; t++) { y[t].re *= sc; y[t].im *= sc; }
        }
    float* outTD = (float*)calloc((size_t)nL * F, sizeof(float));
    orc_afSTFT_backward_knownDimensions(p->hSTFT, outTF, F, nL, T, outTD);
    int ch;
    for (ch = 0; ch < (nL < nOutputs ? nL : nOutputs); ch++) memcpy(outputs[ch], &outTD[(size_t)ch * F], sizeof(float) * F);
    for (; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    free(inTD); free(inTF); free(outTF); free(outTD);
}

/* setters (panner.c:335-540) */
#define PP orc_pan* p = (orc_pan*)h
static void all_recalc(orc_pan* p) { for (int ch = 0; ch < MAXCH; ch++) p->recalc[ch] = 1; }
void orc_panner_setSourceAzi_deg(void* h, int i, float v) { PP; if (v > 180.0f) v = -360.0f + v; v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    if (p->src_dirs_deg[i][0] != v) { p->src_dirs_deg[i][0] = v; p->recalc[i] = 1; p->recalcRot = 1; } }
void orc_panner_setSourceElev_deg(void* h, int i, float v) { PP; v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    if (p->src_dirs_deg[i][1] != v) { p->src_dirs_deg[i][1] = v; p->recalc[i] = 1; p->recalcRot = 1; } }
void orc_panner_setNumSources(void* h, int n) { PP; n = n > MAXCH ? MAXCH : n;
    if (p->nSources != n) { p->new_nSources = n; for (int ch = p->nSources; ch < n; ch++) p->recalc[ch] = 1; p->recalcRot = 1; p->codecReady = 0; } }
void orc_panner_setLoudspeakerAzi_deg(void* h, int i, float v) { PP; if (v > 180.0f) v = -360.0f + v; v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    if (p->ls_dirs_deg[i][0] != v) { p->ls_dirs_deg[i][0] = v; p->reInitGainTables = 1; all_recalc(p); p->recalcRot = 1; p->codecReady = 0; } }
void orc_panner_setLoudspeakerElev_deg(void* h, int i, float v) { PP; v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    if (p->ls_dirs_deg[i][1] != v) { p->ls_dirs_deg[i][1] = v; p->reInitGainTables = 1; all_recalc(p); p->recalcRot = 1; p->codecReady = 0; } }
void orc_panner_setNumLoudspeakers(void* h, int n) { PP; n = n > MAXCH ? MAXCH : n;
    if (p->new_nLoudpkrs != n) { p->new_nLoudpkrs = n; p->reInitGainTables = 1; all_recalc(p); p->recalcRot = 1; p->codecReady = 0; } }
void orc_panner_setOutputConfigPreset(void* h, int id) { PP; load_ls_preset(id, p->ls_dirs_deg, &p->new_nLoudpkrs); p->reInitGainTables = 1; all_recalc(p); p->recalcRot = 1; p->codecReady = 0; }
void orc_panner_setInputConfigPreset(void* h, int id) { PP; load_src_preset(id, p->src_dirs_deg, &p->new_nSources);
    for (int ch = 0; ch < p->new_nSources; ch++) { p->recalc[ch] = 1; }
    p->recalcRot = 1; p->codecReady = 0; }
void orc_panner_setDTT(void* h, float v) { PP; if (p->DTT != v) { p->DTT = v; orc_getPvalues(p->DTT, p->freqVector, NB, p->pValue);
    for (int ch = 0; ch < p->new_nSources; ch++) { p->recalc[ch] = 1; }
    p->recalcRot = 1; p->codecReady = 0; } }
void orc_panner_setSpread(void* h, float v) { PP; if (p->spread_deg != v) { p->spread_deg = v < 0.0f ? 0.0f : (v > 90.0f ? 90.0f : v); p->reInitGainTables = 1; all_recalc(p); p->recalcRot = 1; p->codecReady = 0; } }
void orc_panner_setYaw(void* h, float v) { PP; p->ypr[0] = (p->flip[0] ? -1.0f : 1.0f) * (v * ORC_PI / 180.0f); p->recalcRot = 1; }
void orc_panner_setPitch(void* h, float v) { PP; p->ypr[1] = (p->flip[1] ? -1.0f : 1.0f) * (v * ORC_PI / 180.0f); p->recalcRot = 1; }
void orc_panner_setRoll(void* h, float v) { PP; p->ypr[2] = (p->flip[2] ? -1.0f : 1.0f) * (v * ORC_PI / 180.0f); p->recalcRot = 1; }
int orc_panner_getNumSources(void* h) { PP; return p->new_nSources; }
int orc_panner_getNumLoudspeakers(void* h) { PP; return p->new_nLoudpkrs; }
int orc_panner_getNTriangles(void* h) { PP; return p->nTriangles; }
const float* orc_panner_getGains(void* h) { PP; return p->G_src; }
const float* orc_panner_getPvalue(void* h) { PP; return p->pValue; }
