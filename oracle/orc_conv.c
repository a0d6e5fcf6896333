/*
 * orc_conv.c — CPU restatement of saf_multiConv_* and saf_TVConv_*
 * (framework/modules/saf_utilities/saf_utility_matrixConv.c:237-620).
 *
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  Parity status: the reference's tests only smoke-run these
 * (test__utilities_module.c) and hold no golden vectors: the restatement is additionally checked against float64 direct
 * time-domain convolution in tests/test_oracle_cpu.py; beyond that "parity unpinned".
 */
#include "saf_oracle.h"
#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void cvvmul(const orc_cpx* a, const orc_cpx* b, int n, orc_cpx* c)      /* utility_cvvmul */
{
    for (int i = 0; i < n; i++) { c[i].re = a[i].re * b[i].re - a[i].im * b[i].im; c[i].im = a[i].re * b[i].im + a[i].im * b[i].re; }
}

/* ------------------------------------------------------------------ multi-channel convolver */
typedef struct {
    int hopSize, fftSize, nBins, length_h, nCH, numOvrlpAddBlocks, numFilterBlocks, usePart;
    void* hFFT;
    float *x_pad, *z_n, *ovrlpAddBuffer, *hx_n, *y_n_overlap;
    orc_cpx *X_n, *HX_n, *Z_n, *H_f, *Hpart_f;
} orc_mulc;

/* saf_multiConv_create (saf_utility_matrixConv.c:257-328) */
void orc_multiConv_create(void** ph, int hopSize, const float* H, int length_h, int nCH, int usePartFLAG)
{
    orc_mulc* h = (orc_mulc*)calloc(1, sizeof(orc_mulc));
    h->hopSize = hopSize; h->length_h = length_h; h->nCH = nCH; h->usePart = usePartFLAG;
    if (!usePartFLAG) {
        h->numOvrlpAddBlocks = (int)(ceilf((float)(hopSize + length_h - 1) / (float)hopSize) + 0.1f);
        h->fftSize = h->numOvrlpAddBlocks * hopSize; h->nBins = h->fftSize / 2 + 1;
        h->ovrlpAddBuffer = (float*)calloc((size_t)nCH * h->fftSize, sizeof(float));
        float* h_pad = (float*)calloc(h->fftSize, sizeof(float));
        h->H_f = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nCH * h->nBins);
        h->X_n = (orc_cpx*)calloc((size_t)nCH * h->nBins, sizeof(orc_cpx));
        h->Z_n = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nCH * h->nBins);
        h->x_pad = (float*)calloc(h->fftSize, sizeof(float));
        h->z_n = (float*)malloc(sizeof(float) * (size_t)nCH * h->fftSize);
        orc_rfft_create(&h->hFFT, h->fftSize);
        for (int nc = 0; nc < nCH; nc++) {
            memcpy(h_pad, &H[(size_t)nc * length_h], sizeof(float) * length_h);
            orc_rfft_forward(h->hFFT, h_pad, &h->H_f[(size_t)nc * h->nBins]);
        }
        free(h_pad);
    } else {
        h->fftSize = 2 * hopSize; h->nBins = hopSize + 1;
        h->numFilterBlocks = (int)ceilf((float)length_h / (float)hopSize);
        assert(h->numFilterBlocks >= 1);
        const int nFB = h->numFilterBlocks;
        float* h_pad = (float*)calloc((size_t)nFB * hopSize, sizeof(float));
        float* h_pad2 = (float*)calloc(2 * hopSize, sizeof(float));
        h->Hpart_f = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nFB * nCH * h->nBins);
        h->X_n = (orc_cpx*)calloc((size_t)nFB * nCH * h->nBins, sizeof(orc_cpx));
        h->HX_n = (orc_cpx*)calloc((size_t)nFB * nCH * h->nBins, sizeof(orc_cpx));
        h->x_pad = (float*)calloc(2 * hopSize, sizeof(float));
        h->hx_n = (float*)malloc(sizeof(float) * (size_t)nFB * nCH * h->fftSize);
        h->z_n = (float*)calloc(h->fftSize, sizeof(float));
        h->y_n_overlap = (float*)calloc((size_t)nCH * hopSize, sizeof(float));
        orc_rfft_create(&h->hFFT, h->fftSize);
        for (int nc = 0; nc < nCH; nc++) {
            memset(h_pad, 0, sizeof(float) * (size_t)nFB * hopSize);   /* (the reference leaves the previous channel's tail in h_pad; with equal lengths it is overwritten) */
            memcpy(h_pad, &H[(size_t)nc * length_h], sizeof(float) * length_h);
            for (int nb = 0; nb < nFB; nb++) {
                memcpy(h_pad2, &h_pad[(size_t)nb * hopSize], sizeof(float) * hopSize);
                orc_rfft_forward(h->hFFT, h_pad2, &h->Hpart_f[((size_t)nb * nCH + nc) * h->nBins]);
            }
        }
        free(h_pad); free(h_pad2);
    }
    *ph = h;
}
void orc_multiConv_destroy(void** ph)
{
    orc_mulc* h = (orc_mulc*)*ph; if (!h) return;
    orc_rfft_destroy(&h->hFFT);
    free(h->X_n); free(h->x_pad); free(h->z_n); free(h->ovrlpAddBuffer); free(h->Z_n); free(h->H_f);
    free(h->HX_n); free(h->hx_n); free(h->y_n_overlap); free(h->Hpart_f);
    free(h); *ph = NULL;
}
/* saf_multiConv_apply (saf_utility_matrixConv.c:356-416) */
void orc_multiConv_apply(void* hh, const float* in, float* out)
{
    orc_mulc* h = (orc_mulc*)hh;
    const int hop = h->hopSize, nB = h->nBins, fft = h->fftSize, nCH = h->nCH;
    if (!h->usePart) {
        for (int nc = 0; nc < nCH; nc++) {
            memcpy(h->x_pad, &in[(size_t)nc * hop], sizeof(float) * hop);
            orc_rfft_forward(h->hFFT, h->x_pad, &h->X_n[(size_t)nc * nB]);
        }
        cvvmul(h->H_f, h->X_n, nCH * nB, h->Z_n);
        for (int nc = 0; nc < nCH; nc++) {
            orc_rfft_backward(h->hFFT, &h->Z_n[(size_t)nc * nB], &h->z_n[(size_t)nc * fft]);
            float* ob = &h->ovrlpAddBuffer[(size_t)nc * fft];
            memmove(ob, ob + hop, sizeof(float) * (size_t)(h->numOvrlpAddBlocks - 1) * hop);
            memset(ob + (size_t)(h->numOvrlpAddBlocks - 1) * hop, 0, sizeof(float) * hop);
            for (int n = 0; n < fft; n++) ob[n] += h->z_n[(size_t)nc * fft + n];
            memcpy(&out[(size_t)nc * hop], ob, sizeof(float) * hop);
        }
    } else {
        const int nFB = h->numFilterBlocks;
        memmove(&h->X_n[(size_t)nCH * nB], h->X_n, sizeof(orc_cpx) * (size_t)(nFB - 1) * nCH * nB);
        for (int nc = 0; nc < nCH; nc++) {
            memcpy(h->x_pad, &in[(size_t)nc * hop], sizeof(float) * hop);
            orc_rfft_forward(h->hFFT, h->x_pad, &h->X_n[(size_t)nc * nB]);
        }
        cvvmul(h->Hpart_f, h->X_n, nFB * nCH * nB, h->HX_n);
        for (int nc = 0; nc < nCH; nc++) {
            for (int nb = 0; nb < nFB; nb++)
                orc_rfft_backward(h->hFFT, &h->HX_n[((size_t)nb * nCH + nc) * nB], &h->hx_n[((size_t)nb * nCH + nc) * fft]);
            memset(h->z_n, 0, sizeof(float) * fft);
            for (int nb = 0; nb < nFB; nb++) for (int n = 0; n < fft; n++) h->z_n[n] += h->hx_n[((size_t)nb * nCH + nc) * fft + n];
            for (int n = 0; n < hop; n++) out[(size_t)nc * hop + n] = h->z_n[n] + h->y_n_overlap[(size_t)nc * hop + n];
            memcpy(&h->y_n_overlap[(size_t)nc * hop], &h->z_n[hop], sizeof(float) * hop);
        }
    }
}

/* ------------------------------------------------------------------ time-varying convolver */
typedef struct {
    int hopSize, fftSize, nBins, length_h, nIRs, nCHout, numFilterBlocks;
    void* hFFT;
    float *x_pad, *hx_n, *z_n, *z_n_last, *z_n_last2, *y_n_overlap, *y_n_overlap_last, *fadeIn, *fadeOut;
    orc_cpx *X_n, *HX_n, *Hpart_f;     /* Hpart_f [nIRs][nCHout][nFB][nBins] */
    int posIdx_last, posIdx_last2;
} orc_tvc;

/* saf_TVConv_create (saf_utility_matrixConv.c:438-513); H is [nIRs][nCHout][length_h] flat */
void orc_TVConv_create(void** ph, int hopSize, const float* H, int length_h, int nIRs, int nCHout, int initIdx)
{
    orc_tvc* h = (orc_tvc*)calloc(1, sizeof(orc_tvc));
    h->hopSize = hopSize; h->length_h = length_h; h->nIRs = nIRs; h->nCHout = nCHout;
    h->posIdx_last = h->posIdx_last2 = initIdx < nIRs ? initIdx : 0;
    h->fftSize = 2 * hopSize; h->nBins = hopSize + 1;
    h->numFilterBlocks = (int)ceilf((float)length_h / (float)hopSize);
    assert(h->numFilterBlocks >= 1);
    const int nFB = h->numFilterBlocks, nB = h->nBins;
    float* h_pad = (float*)calloc((size_t)nFB * hopSize, sizeof(float));
    float* h_pad2 = (float*)calloc(2 * hopSize, sizeof(float));
    h->Hpart_f = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nIRs * nCHout * nFB * nB);
    h->X_n = (orc_cpx*)calloc((size_t)nFB * nB, sizeof(orc_cpx));
    h->HX_n = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nFB * nB);
    h->x_pad = (float*)calloc(2 * hopSize, sizeof(float));
    h->hx_n = (float*)malloc(sizeof(float) * (size_t)nFB * h->fftSize);
    h->y_n_overlap = (float*)calloc((size_t)nCHout * hopSize, sizeof(float));
    h->y_n_overlap_last = (float*)calloc((size_t)nCHout * hopSize, sizeof(float));
    h->z_n = (float*)malloc(sizeof(float) * h->fftSize);
    h->z_n_last = (float*)malloc(sizeof(float) * h->fftSize);
    h->z_n_last2 = (float*)malloc(sizeof(float) * h->fftSize);
    h->fadeIn = (float*)malloc(sizeof(float) * hopSize); h->fadeOut = (float*)malloc(sizeof(float) * hopSize);
    for (int n = 0; n < hopSize; n++) { h->fadeIn[n] = (float)n / (float)(hopSize - 1); h->fadeOut[n] = (float)(hopSize - 1 - n) / (float)(hopSize - 1); }
    orc_rfft_create(&h->hFFT, h->fftSize);
    for (int np = 0; np < nIRs; np++)
        for (int no = 0; no < nCHout; no++) {
            memset(h_pad, 0, sizeof(float) * (size_t)nFB * hopSize);
            memcpy(h_pad, &H[((size_t)np * nCHout + no) * length_h], sizeof(float) * length_h);
            for (int nb = 0; nb < nFB; nb++) {
                memcpy(h_pad2, &h_pad[(size_t)nb * hopSize], sizeof(float) * hopSize);
                orc_rfft_forward(h->hFFT, h_pad2, &h->Hpart_f[(((size_t)np * nCHout + no) * nFB + nb) * nB]);
            }
        }
    free(h_pad); free(h_pad2);
    *ph = h;
}
void orc_TVConv_destroy(void** ph)
{
    orc_tvc* h = (orc_tvc*)*ph; if (!h) return;
    orc_rfft_destroy(&h->hFFT);
    free(h->X_n); free(h->x_pad); free(h->z_n); free(h->z_n_last); free(h->z_n_last2); free(h->hx_n); free(h->HX_n);
    free(h->y_n_overlap); free(h->y_n_overlap_last); free(h->fadeIn); free(h->fadeOut); free(h->Hpart_f);
    free(h); *ph = NULL;
}
static void tv_conv_one(orc_tvc* h, int ir, int no, float* z)
{
    const int nFB = h->numFilterBlocks, nB = h->nBins, fft = h->fftSize;
    cvvmul(&h->Hpart_f[((size_t)ir * h->nCHout + no) * nFB * nB], h->X_n, nFB * nB, h->HX_n);
    for (int nb = 0; nb < nFB; nb++) orc_rfft_backward(h->hFFT, &h->HX_n[(size_t)nb * nB], &h->hx_n[(size_t)nb * fft]);
    memset(z, 0, sizeof(float) * fft);
    for (int nb = 0; nb < nFB; nb++) for (int n = 0; n < fft; n++) z[n] += h->hx_n[(size_t)nb * fft + n];
}
/* saf_TVConv_apply (saf_utility_matrixConv.c:554-620) */
void orc_TVConv_apply(void* hh, const float* in, float* out, int irIdx)
{
    orc_tvc* h = (orc_tvc*)hh;
    const int hop = h->hopSize, nB = h->nBins, fft = h->fftSize, nFB = h->numFilterBlocks;
    memmove(&h->X_n[nB], h->X_n, sizeof(orc_cpx) * (size_t)(nFB - 1) * nB);
    memcpy(h->x_pad, in, sizeof(float) * hop);
    orc_rfft_forward(h->hFFT, h->x_pad, h->X_n);
    for (int no = 0; no < h->nCHout; no++) {
        tv_conv_one(h, irIdx, no, h->z_n);
        if (irIdx != h->posIdx_last) tv_conv_one(h, h->posIdx_last, no, h->z_n_last); else memcpy(h->z_n_last, h->z_n, sizeof(float) * fft);
        if (h->posIdx_last != h->posIdx_last2) tv_conv_one(h, h->posIdx_last2, no, h->z_n_last2); else memcpy(h->z_n_last2, h->z_n_last, sizeof(float) * fft);
        for (int n = 0; n < hop; n++) {
            const float o1 = h->z_n_last[n] + h->y_n_overlap[(size_t)no * hop + n];
            const float o2 = h->z_n_last2[n] + h->y_n_overlap_last[(size_t)no * hop + n];
            out[(size_t)no * hop + n] = o1 * h->fadeIn[n] + o2 * h->fadeOut[n];
        }
        memcpy(&h->y_n_overlap[(size_t)no * hop], &h->z_n[hop], sizeof(float) * hop);
        memcpy(&h->y_n_overlap_last[(size_t)no * hop], &h->z_n_last[hop], sizeof(float) * hop);
    }
    h->posIdx_last2 = h->posIdx_last;
    h->posIdx_last = irIdx;
}
