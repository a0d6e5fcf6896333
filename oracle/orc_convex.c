/*
 * orc_convex.c — CPU restatement of the matrixconv / multiconv example operators
 * (examples/src/matrixconv/matrixconv.c, examples/src/multiconv/multiconv.c): sample-wise FIFO around the convolvers.
 *
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  The reference has no test for these wrappers: "parity unpinned"
 * (the convolvers underneath are checked against direct convolution).
 */
#include "saf_oracle.h"
#include <stdlib.h>
#include <string.h>

#define MAXCH 64
#define MINF 512
#define MAXF 8192

typedef struct {
    int matrix, FIFO_idx;
    float* inFIFO; float* outFIFO;        /* [MAXCH][MAXF] */
    float* inTD; float* outTD;            /* [MAXCH][B] */
    void* hConv;
    int hostBlockSize, B;
    float* filters;
    int nfilters, wavLen, filter_length, reInit, nOut, nIn, part;
} orc_cx;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void cx_destroy_conv(orc_cx* p) { if (!p->hConv) return; if (p->matrix) orc_matrixConv_destroy(&p->hConv); else orc_multiConv_destroy(&p->hConv); p->hConv = NULL; }

/* matrixconv_checkReInit (matrixconv.c:164-203) / multiconv_checkReInit (multiconv.c:164-192) */
static void cx_check(orc_cx* p)
{
    if (p->reInit == 1 && p->filters) {
        p->reInit = 2;
        cx_destroy_conv(p);
        p->B = clampi(p->hostBlockSize, MINF, MAXF);
        if (p->matrix) { if (p->filter_length > 0) orc_matrixConv_create(&p->hConv, p->B, p->filters, p->filter_length, p->nIn, p->nOut, p->part); }
        else orc_multiConv_create(&p->hConv, p->B, p->filters, p->filter_length, p->nfilters, p->part);
        p->inTD = (float*)realloc(p->inTD, sizeof(float) * MAXCH * p->B); p->outTD = (float*)realloc(p->outTD, sizeof(float) * MAXCH * p->B);
        memset(p->inTD, 0, sizeof(float) * MAXCH * p->B); memset(p->outTD, 0, sizeof(float) * MAXCH * p->B);
        p->FIFO_idx = 0;
        memset(p->inFIFO, 0, sizeof(float) * MAXCH * MAXF); memset(p->outFIFO, 0, sizeof(float) * MAXCH * MAXF);
        p->reInit = 0;
    }
}

void orc_convex_create(void** ph, int matrix)
{
    orc_cx* p = (orc_cx*)calloc(1, sizeof(orc_cx));
    p->matrix = matrix; p->nIn = 1; p->hostBlockSize = -1; p->B = MINF; p->reInit = 1;
    p->inFIFO = (float*)calloc((size_t)MAXCH * MAXF, sizeof(float)); p->outFIFO = (float*)calloc((size_t)MAXCH * MAXF, sizeof(float));
    *ph = p;
}
void orc_convex_destroy(void** ph)
{
    orc_cx* p = (orc_cx*)*ph; if (!p) return;
    cx_destroy_conv(p); free(p->inFIFO); free(p->outFIFO); free(p->inTD); free(p->outTD); free(p->filters); free(p); *ph = NULL;
}
void orc_convex_init(void* h, int sampleRate, int hostBlockSize)
{
    orc_cx* p = (orc_cx*)h; (void)sampleRate;
    if (p->hostBlockSize != hostBlockSize) { p->hostBlockSize = hostBlockSize; p->B = clampi(hostBlockSize, MINF, MAXF); p->reInit = 1; }
    cx_check(p);
}
/* matrixconv_setFilters (matrixconv.c:205-236) / multiconv_setFilters (multiconv.c:194-211); H flat [numChannels][numSamples] */
void orc_convex_setFilters(void* h, const float* H, int numChannels, int numSamples)
{
    orc_cx* p = (orc_cx*)h;
    p->filters = (float*)realloc(p->filters, sizeof(float) * (size_t)numChannels * numSamples);
    memcpy(p->filters, H, sizeof(float) * (size_t)numChannels * numSamples);
    if (p->matrix) {
        p->nOut = numChannels < MAXCH ? numChannels : MAXCH; p->wavLen = numSamples; p->nfilters = p->nOut * p->nIn;
        p->filter_length = p->wavLen % p->nIn == 0 ? p->wavLen / p->nIn : 0;
    } else { p->nfilters = numChannels; p->filter_length = numSamples; }
    p->reInit = 1;
}
void orc_convex_setEnablePart(void* h, int s) { orc_cx* p = (orc_cx*)h; if (p->part != s) { p->part = s; p->reInit = 1; } }
void orc_convex_setNumInputChannels(void* h, int n)
{
    orc_cx* p = (orc_cx*)h;
    p->nIn = clampi(n, 1, MAXCH);
    if (p->matrix) {
        p->nfilters = p->nOut * p->nIn;
        p->filter_length = (p->nOut > 0 && p->wavLen % p->nIn == 0) ? p->wavLen / p->nIn : 0;
        p->reInit = 1;
    }
}
int orc_convex_getProcessingDelay(void* h) { return ((orc_cx*)h)->B; }
int orc_convex_getFilterLength(void* h) { return ((orc_cx*)h)->filter_length; }
/* matrixconv_process (matrixconv.c:97-157) / multiconv_process (multiconv.c:95-153) */
void orc_convex_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)
{
    orc_cx* p = (orc_cx*)h;
    cx_check(p);
    const int numIn = p->nIn, numOut = p->matrix ? p->nOut : p->nIn, B = p->B;
    for (int s = 0; s < nSamples; s++) {
        int ch, lim = nInputs < numIn ? nInputs : numIn; if (lim > MAXCH) lim = MAXCH;
        for (ch = 0; ch < lim; ch++) p->inFIFO[(size_t)ch * MAXF + p->FIFO_idx] = inputs[ch][s];
        for (; ch < numIn; ch++) p->inFIFO[(size_t)ch * MAXF + p->FIFO_idx] = 0.0f;
        lim = nOutputs < numOut ? nOutputs : numOut; if (lim > MAXCH) lim = MAXCH;
        for (ch = 0; ch < lim; ch++) outputs[ch][s] = p->outFIFO[(size_t)ch * MAXF + p->FIFO_idx];
        for (; ch < nOutputs; ch++) outputs[ch][s] = 0.0f;
        p->FIFO_idx++;
        if (p->FIFO_idx >= B && p->reInit == 0) {
            p->FIFO_idx = 0;
            for (int i = 0; i < numIn; i++) memcpy(&p->inTD[(size_t)i * B], &p->inFIFO[(size_t)i * MAXF], sizeof(float) * B);
            if (p->hConv && (!p->matrix || p->filter_length > 0)) {
                if (p->matrix) orc_matrixConv_apply(p->hConv, p->inTD, p->outTD); else orc_multiConv_apply(p->hConv, p->inTD, p->outTD);
            } else memset(p->outTD, 0, sizeof(float) * MAXCH * B);
            for (int i = 0; i < (numOut < MAXCH ? numOut : MAXCH); i++) memcpy(&p->outFIFO[(size_t)i * MAXF], &p->outTD[(size_t)i * B], sizeof(float) * B);
        } else if (p->FIFO_idx >= B) {
            p->FIFO_idx = 0;
            memset(p->outFIFO, 0, sizeof(float) * MAXCH * MAXF);
        }
    }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * tvconv example (examples/src/tvconv/tvconv.c, tvconv_internal.c): FIFO around saf_TVConv; the IR set follows the listener
 * position nearest to the target.  IRs / positions are injected (the reference reads them from a SOFA file).
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int FIFO_idx, hostBlockSize, B, nIr, irLen, nPos, nOut, posIdx, reInit, ready;
    float* inFIFO; float* outFIFO; float* inTD; float* outTD; float* irs; float* pos;
    float target[3];
    void* hTV;
} orc_tvx;
void orc_tvconvex_create(void** ph)
{
    orc_tvx* p = (orc_tvx*)calloc(1, sizeof(orc_tvx));
    p->hostBlockSize = -1; p->B = MINF; p->reInit = 1;
    p->inFIFO = (float*)calloc((size_t)MAXCH * MAXF, sizeof(float)); p->outFIFO = (float*)calloc((size_t)MAXCH * MAXF, sizeof(float));
    *ph = p;
}
void orc_tvconvex_destroy(void** ph)
{
    orc_tvx* p = (orc_tvx*)*ph; if (!p) return;
    if (p->hTV) orc_TVConv_destroy(&p->hTV);
    free(p->inFIFO); free(p->outFIFO); free(p->inTD); free(p->outTD); free(p->irs); free(p->pos); free(p); *ph = NULL;
}
static void tvx_check(orc_tvx* p)          /* tvconv_checkReInit (tvconv.c:196-232) */
{
    if (p->reInit == 1 && p->irs) {
        p->reInit = 2;
        if (p->hTV) orc_TVConv_destroy(&p->hTV);
        p->B = clampi(p->hostBlockSize, MINF, MAXF);
        if (p->irLen > 0) orc_TVConv_create(&p->hTV, p->B, p->irs, p->irLen, p->nPos, p->nOut, p->posIdx);
        p->inTD = (float*)realloc(p->inTD, sizeof(float) * MAXCH * p->B); p->outTD = (float*)realloc(p->outTD, sizeof(float) * MAXCH * p->B);
        memset(p->inTD, 0, sizeof(float) * MAXCH * p->B); memset(p->outTD, 0, sizeof(float) * MAXCH * p->B);
        p->FIFO_idx = 0;
        memset(p->inFIFO, 0, sizeof(float) * MAXCH * MAXF); memset(p->outFIFO, 0, sizeof(float) * MAXCH * MAXF);
        p->reInit = 0; p->ready = 1;
    }
}
void orc_tvconvex_init(void* h, int hostBlockSize)
{
    orc_tvx* p = (orc_tvx*)h;
    if (p->hostBlockSize != hostBlockSize) { p->hostBlockSize = hostBlockSize; p->B = clampi(hostBlockSize, MINF, MAXF); p->reInit = 1; p->ready = 0; }
    tvx_check(p);
}
/* irs flat [nPos][nIr][irLen]; positions [nPos][3] (tvconv.c:262-312) */
void orc_tvconvex_setIRsAndPositions(void* h, const float* irs, const float* positions, int nPos, int nIr, int irLen)
{
    orc_tvx* p = (orc_tvx*)h;
    p->irs = (float*)realloc(p->irs, sizeof(float) * (size_t)nPos * nIr * irLen); memcpy(p->irs, irs, sizeof(float) * (size_t)nPos * nIr * irLen);
    p->pos = (float*)realloc(p->pos, sizeof(float) * (size_t)nPos * 3); memcpy(p->pos, positions, sizeof(float) * (size_t)nPos * 3);
    p->nPos = nPos; p->nIr = nIr; p->irLen = irLen; p->nOut = nIr < MAXCH ? nIr : MAXCH;
    for (int d = 0; d < 3; d++) { float mn = positions[d]; for (int i = 1; i < nPos; i++) if (positions[i * 3 + d] < mn) mn = positions[i * 3 + d]; p->target[d] = mn; }
    p->posIdx = 0; p->ready = 1; p->reInit = 1;
}
void orc_tvconvex_setTargetPosition(void* h, float v, int dim)       /* tvconv.c:333-338, tvconv_internal.c:42-61 */
{
    orc_tvx* p = (orc_tvx*)h;
    p->target[dim] = v;
    int mi = 0; float md = 0.0f;
    for (int i = 0; i < p->nPos; i++) {
        float dist = 0.0f;
        for (int d = 0; d < 3; d++) dist += (p->target[d] - p->pos[i * 3 + d]) * (p->target[d] - p->pos[i * 3 + d]);
        if (dist < md || i == 0) { md = dist; mi = i; }
    }
    p->posIdx = mi;
}
int orc_tvconvex_getListenerPositionIdx(void* h) { return ((orc_tvx*)h)->posIdx; }
void orc_tvconvex_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)     /* tvconv.c:119-186 */
{
    orc_tvx* p = (orc_tvx*)h;
    tvx_check(p);
    const int B = p->B, numOut = p->nOut;
    for (int s = 0; s < nSamples; s++) {
        int ch;
        for (ch = 0; ch < (nInputs < 1 ? nInputs : 1); ch++) p->inFIFO[(size_t)ch * MAXF + p->FIFO_idx] = inputs[ch][s];
        for (; ch < 1; ch++) p->inFIFO[(size_t)ch * MAXF + p->FIFO_idx] = 0.0f;
        int lim = nOutputs < numOut ? nOutputs : numOut;
        for (ch = 0; ch < lim; ch++) outputs[ch][s] = p->outFIFO[(size_t)ch * MAXF + p->FIFO_idx];
        for (; ch < nOutputs; ch++) outputs[ch][s] = 0.0f;
        p->FIFO_idx++;
        if (p->FIFO_idx >= B && p->reInit == 0 && p->ready) {
            p->FIFO_idx = 0;
            memcpy(p->inTD, p->inFIFO, sizeof(float) * B);
            if (p->hTV && p->irLen > 0) orc_TVConv_apply(p->hTV, p->inTD, p->outTD, p->posIdx); else memset(p->outTD, 0, sizeof(float) * MAXCH * B);
            for (int i = 0; i < numOut; i++) memcpy(&p->outFIFO[(size_t)i * MAXF], &p->outTD[(size_t)i * B], sizeof(float) * B);
        } else if (p->FIFO_idx >= B) { p->FIFO_idx = 0; memset(p->outFIFO, 0, sizeof(float) * MAXCH * MAXF); }
    }
}
