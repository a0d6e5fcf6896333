/*
 * orc_ambi_drc.c — CPU restatement of the ambi_drc example (examples/src/ambi_drc/ambi_drc.c:43-428,
 * ambi_drc_internal.c:46-129).  TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  The reference holds no test for this
 * operator: "parity unpinned" by reference-side data.
 */
#include "saf_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <complex.h>

#define NB 133
#define MAXSH 64

typedef struct {
    int F, T, nSH, new_nSH, order, reInit;
    float fs, threshold, ratio, knee, inGain, outGain, attack_ms, release_ms;
    float yL_z1[NB];
    void* hSTFT;
    float* frameTD; orc_cpx* inTF; orc_cpx* outTF;       /* [64][F], [133][64][T] */
    float* gains;                                          /* [133][T] of the last call */
} orc_drc;

static float gain_computer(float xG, float T, float R, float W)     /* ambi_drc_internal.c:46-66 */
{
    float yG;
    if (2.0f * (xG - T) < -W) yG = xG;
    else if (2.0f * (fabsf(xG - T)) <= W) yG = xG + (1.0f / R - 1.0f) * powf(xG - T + W / 2.0f, 2.0f) / (2.0f * W);
    else if (2.0f * (xG - T) > W) yG = T + (xG - T) / R;
    else yG = 0.0f;
    return yG;
}
static float peak_detector(float xL, float yL_z1, float aa, float ar)  /* :71-88 */
{
    return xL > yL_z1 ? aa * yL_z1 + (1.0f - aa) * xL : ar * yL_z1 + (1.0f - ar) * xL;
}

static void init_tft(orc_drc* p)       /* ambi_drc_internal.c:90-104 */
{
    if (!p->hSTFT) orc_afSTFT_create(&p->hSTFT, p->new_nSH, p->new_nSH, 128, 0, 1, 0 /* bands x ch x time */);
    else if (p->nSH != p->new_nSH) { orc_afSTFT_channelChange(p->hSTFT, p->new_nSH, p->new_nSH); orc_afSTFT_clearBuffers(p->hSTFT); }
    p->nSH = p->new_nSH;
}

void orc_ambi_drc_create(void** ph, int F)
{
    orc_drc* p = (orc_drc*)calloc(1, sizeof(orc_drc));
    p->F = F; p->T = F / 128; p->fs = 48000.0f;
    p->frameTD = (float*)calloc((size_t)MAXSH * F, sizeof(float));
    p->inTF = (orc_cpx*)calloc((size_t)NB * MAXSH * p->T, sizeof(orc_cpx));
    p->outTF = (orc_cpx*)calloc((size_t)NB * MAXSH * p->T, sizeof(orc_cpx));
    p->gains = (float*)calloc((size_t)NB * p->T, sizeof(float));
    p->threshold = 0.0f; p->ratio = 8.0f; p->knee = 0.0f; p->inGain = 0.0f; p->outGain = 0.0f; p->attack_ms = 50.0f; p->release_ms = 100.0f;
    p->order = 1; p->new_nSH = p->nSH = 4; p->reInit = 1;
    *ph = p;
}
void orc_ambi_drc_destroy(void** ph)
{
    orc_drc* p = (orc_drc*)*ph; if (!p) return;
    if (p->hSTFT) orc_afSTFT_destroy(&p->hSTFT);
    free(p->frameTD); free(p->inTF); free(p->outTF); free(p->gains); free(p); *ph = NULL;
}
void orc_ambi_drc_init(void* h, int fs)
{
    orc_drc* p = (orc_drc*)h;
    p->fs = (float)fs;
    memset(p->yL_z1, 0, sizeof(p->yL_z1));
    if (p->reInit == 1) { p->reInit = 2; init_tft(p); p->reInit = 0; }
}

void orc_ambi_drc_process(void* h, const float* const* inputs, float* const* outputs, int nCh, int nSamples)   /* ambi_drc.c:134-228 */
{
    orc_drc* p = (orc_drc*)h;
    const int F = p->F, T = p->T;
    if (p->reInit == 1) { p->reInit = 2; init_tft(p); p->reInit = 0; }
    const float alpha_a = expf(-1.0f / ((p->attack_ms / ((float)F / (float)T)) * p->fs * 0.001f));
    const float alpha_r = expf(-1.0f / ((p->release_ms / ((float)F / (float)T)) * p->fs * 0.001f));
    const float boost = powf(10.0f, p->inGain / 20.0f), makeup = powf(10.0f, p->outGain / 20.0f);
    if (nSamples != F || p->reInit != 0) { for (int ch = 0; ch < nCh; ch++) memset(outputs[ch], 0, sizeof(float) * F); return; }
    const int nSH = p->nSH;
    int i;
    for (i = 0; i < (nSH < nCh ? nSH : nCh); i++) memcpy(p->frameTD + (size_t)i * F, inputs[i], sizeof(float) * F);
    for (; i < nSH; i++) memset(p->frameTD + (size_t)i * F, 0, sizeof(float) * F);
    orc_afSTFT_forward_knownDimensions(p->hSTFT, p->frameTD, F, MAXSH, T, p->inTF);
    for (int t = 0; t < T; t++)
        for (int band = 0; band < NB; band++) {
            for (int ch = 0; ch < nSH; ch++) {
                orc_cpx* v = &p->inTF[((size_t)band * MAXSH + ch) * T + t];
                v->re *= boost; v->im *= boost;
            }
            const orc_cpx om = p->inTF[((size_t)band * MAXSH + 0) * T + t];
            const float xG = 10.0f * log10f(powf(cabsf(om.re + I * om.im), 2.0f) + 2e-13f);
            const float yG = gain_computer(xG, p->threshold, p->ratio, p->knee);
            const float xL = xG - yG;
            const float yL = peak_detector(xL, p->yL_z1[band], alpha_a, alpha_r);
            p->yL_z1[band] = yL;
            float cdB = -yL;
            cdB = fmaxf(0.1585f, sqrtf(powf(10.0f, cdB / 20.0f)));
            p->gains[(size_t)band * T + t] = cdB;
            for (int ch = 0; ch < nSH; ch++) {
                const orc_cpx v = p->inTF[((size_t)band * MAXSH + ch) * T + t];
                orc_cpx* o = &p->outTF[((size_t)band * MAXSH + ch) * T + t];
                o->re = v.re * (cdB * makeup); o->im = v.im * (cdB * makeup);
            }
        }
    orc_afSTFT_backward_knownDimensions(p->hSTFT, p->outTF, F, MAXSH, T, p->frameTD);
    int ch;
    for (ch = 0; ch < (nSH < nCh ? nSH : nCh); ch++) memcpy(outputs[ch], p->frameTD + (size_t)ch * F, sizeof(float) * F);
    for (; ch < nCh; ch++) memset(outputs[ch], 0, sizeof(float) * F);
}

const float* orc_ambi_drc_getLastGains(void* h) { return ((orc_drc*)h)->gains; }      /* [133][T] */
static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
#define DP orc_drc* p = (orc_drc*)h
void orc_ambi_drc_setThreshold(void* h, float v) { DP; p->threshold = clampf(v, -60.0f, 0.0f); }
void orc_ambi_drc_setRatio(void* h, float v) { DP; p->ratio = clampf(v, 1.0f, 30.0f); }
void orc_ambi_drc_setKnee(void* h, float v) { DP; p->knee = clampf(v, 0.0f, 10.0f); }
void orc_ambi_drc_setInGain(void* h, float v) { DP; p->inGain = clampf(v, -40.0f, 20.0f); }
void orc_ambi_drc_setOutGain(void* h, float v) { DP; p->outGain = clampf(v, -20.0f, 40.0f); }
void orc_ambi_drc_setAttack(void* h, float v) { DP; p->attack_ms = clampf(v, 10.0f, 200.0f); }
void orc_ambi_drc_setRelease(void* h, float v) { DP; p->release_ms = clampf(v, 50.0f, 1000.0f); }
void orc_ambi_drc_setInputPreset(void* h, int o) { DP; p->new_nSH = (o + 1) * (o + 1); p->order = o; if (p->new_nSH != p->nSH) p->reInit = 1; }
void orc_ambi_drc_refreshSettings(void* h) { DP; p->reInit = 1; }
int orc_ambi_drc_getNSHrequired(void* h) { DP; return p->nSH; }
