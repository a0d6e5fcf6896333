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
