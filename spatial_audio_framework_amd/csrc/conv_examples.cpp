/*
 * conv_examples.cpp — the matrixconv and multiconv example operators (examples/include/matrixconv.h:53-191,
 * multiconv.h:53-168; examples/src/matrixconv/matrixconv.c, examples/src/multiconv/multiconv.c): a sample-wise FIFO that
 * collects one host block (clamped to 512..8192 samples), runs the convolver on it and plays the result back one block
 * later.  The convolvers are the GPU ones of matrixconv.cpp; everything here is host bookkeeping.
 * (tvconv, the third wrapper of this family, takes its filters and listener positions from a SOFA file only —
 * file I/O outside this library; saf_TVConv itself is provided.)
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"

namespace saf {

#define CX_MIN_FRAME 512       /* matrixconv_internal.h:40 / multiconv_internal.h:40 */
#define CX_MAX_FRAME 8192      /* :41 */

struct ConvExample {
    bool matrix;                       /* true: matrixconv, false: multiconv */
    int FIFO_idx = 0;
    std::vector<float> inFIFO, outFIFO;          /* [64][8192] */
    std::vector<float> inputFrameTD, outputFrameTD;   /* [64][hostBlockSize_clamped] */
    void* hConv = nullptr;
    int hostBlockSize = -1, hostBlockSize_clamped = CX_MIN_FRAME;
    std::vector<float> filters; bool haveFilters = false;
    int nfilters = 0, input_wav_length = 0, filter_length = 0, filter_fs = 0, host_fs = 0, reInitFilters = 1, nOutputChannels = 0;
    int nInputChannels = 1;            /* matrixconv: inputs; multiconv: nChannels */
    int enablePartitionedConv = 0;
};

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void destroy_conv(ConvExample* p)
{
    if (!p->hConv) return;
    if (p->matrix) saf_matrixConv_destroy(&p->hConv); else saf_multiConv_destroy(&p->hConv);
    p->hConv = nullptr;
}

/* matrixconv_checkReInit (matrixconv.c:164-203) / multiconv_checkReInit (multiconv.c:164-192) */
static void check_reinit(ConvExample* p)
{
    if (p->reInitFilters == 1 && p->haveFilters) {
        p->reInitFilters = 2;
        destroy_conv(p);
        p->hostBlockSize_clamped = clampi(p->hostBlockSize, CX_MIN_FRAME, CX_MAX_FRAME);
        if (p->matrix) {
            /* if the wav length is not divisible by the number of inputs the handle stays NULL and nothing is convolved */
            if (p->filter_length > 0)
                saf_matrixConv_create(&p->hConv, p->hostBlockSize_clamped, p->filters.data(), p->filter_length, p->nInputChannels, p->nOutputChannels, p->enablePartitionedConv);
        } else {
            if (p->nfilters > SAF_MAXCH) SAF_FATAL("multiconv: %d filters exceed the %d channels of the example's frame buffers", p->nfilters, SAF_MAXCH);
            saf_multiConv_create(&p->hConv, p->hostBlockSize_clamped, p->filters.data(), p->filter_length, p->nfilters, p->enablePartitionedConv);
        }
        p->inputFrameTD.assign((size_t)SAF_MAXCH * p->hostBlockSize_clamped, 0.0f);
        p->outputFrameTD.assign((size_t)SAF_MAXCH * p->hostBlockSize_clamped, 0.0f);
        p->FIFO_idx = 0;
        std::fill(p->inFIFO.begin(), p->inFIFO.end(), 0.0f); std::fill(p->outFIFO.begin(), p->outFIFO.end(), 0.0f);
        p->reInitFilters = 0;
    }
}

static ConvExample* create_example(bool matrix)
{
    ConvExample* p = new ConvExample();
    p->matrix = matrix;
    p->inFIFO.assign((size_t)SAF_MAXCH * CX_MAX_FRAME, 0.0f); p->outFIFO.assign((size_t)SAF_MAXCH * CX_MAX_FRAME, 0.0f);
    return p;
}

/* matrixconv_process (matrixconv.c:97-157) / multiconv_process (multiconv.c:95-153) */
static void process_example(ConvExample* p, const float* const* inputs, float** outputs, int nInputs, int nOutputs, int nSamples)
{
    check_reinit(p);
    const int numIn = p->nInputChannels, numOut = p->matrix ? p->nOutputChannels : p->nInputChannels;
    const int B = p->hostBlockSize_clamped;
    for (int s = 0; s < nSamples; s++) {
        int ch;
        for (ch = 0; ch < std::min(std::min(nInputs, numIn), SAF_MAXCH); ch++) p->inFIFO[(size_t)ch * CX_MAX_FRAME + p->FIFO_idx] = inputs[ch][s];
        for (; ch < numIn; ch++) p->inFIFO[(size_t)ch * CX_MAX_FRAME + p->FIFO_idx] = 0.0f;
        for (ch = 0; ch < std::min(std::min(nOutputs, numOut), SAF_MAXCH); ch++) outputs[ch][s] = p->outFIFO[(size_t)ch * CX_MAX_FRAME + p->FIFO_idx];
        for (; ch < nOutputs; ch++) outputs[ch][s] = 0.0f;
        p->FIFO_idx++;
        if (p->FIFO_idx >= B && p->reInitFilters == 0) {
            p->FIFO_idx = 0;
            for (int i = 0; i < numIn; i++) memcpy(&p->inputFrameTD[(size_t)i * B], &p->inFIFO[(size_t)i * CX_MAX_FRAME], sizeof(float) * B);
            if (p->hConv && (!p->matrix || p->filter_length > 0)) {
                if (p->matrix) saf_matrixConv_apply(p->hConv, p->inputFrameTD.data(), p->outputFrameTD.data());
                else saf_multiConv_apply(p->hConv, p->inputFrameTD.data(), p->outputFrameTD.data());
            } else
                std::fill(p->outputFrameTD.begin(), p->outputFrameTD.end(), 0.0f);
            for (int i = 0; i < std::min(numOut, SAF_MAXCH); i++) memcpy(&p->outFIFO[(size_t)i * CX_MAX_FRAME], &p->outputFrameTD[(size_t)i * B], sizeof(float) * B);
        } else if (p->FIFO_idx >= B) {
            p->FIFO_idx = 0;                                        /* clear outFIFO if the codec was not ready */
            std::fill(p->outFIFO.begin(), p->outFIFO.end(), 0.0f);
        }
    }
}

static void init_example(ConvExample* p, int sampleRate, int hostBlockSize)
{
    p->host_fs = sampleRate;
    if (p->hostBlockSize != hostBlockSize) {
        p->hostBlockSize = hostBlockSize;
        p->hostBlockSize_clamped = clampi(hostBlockSize, CX_MIN_FRAME, CX_MAX_FRAME);
        p->reInitFilters = 1;
    }
    check_reinit(p);
}

}  // namespace saf

using namespace saf;

extern "C" {

#define CXP ConvExample* p = (ConvExample*)hMCnv
/* ---------------- matrixconv (matrixconv.h:53-191) ---------------- */
void matrixconv_create(void** const phMCnv) { *phMCnv = create_example(true); }
void matrixconv_destroy(void** const phMCnv) { ConvExample* p = (ConvExample*)*phMCnv; if (!p) return; destroy_conv(p); delete p; *phMCnv = nullptr; }
void matrixconv_init(void* const hMCnv, int sampleRate, int hostBlockSize) { CXP; init_example(p, sampleRate, hostBlockSize); }
void matrixconv_process(void* const hMCnv, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples) { CXP; process_example(p, inputs, outputs, nInputs, nOutputs, nSamples); }
void matrixconv_refreshParams(void* const hMCnv) { CXP; p->reInitFilters = 1; }
void matrixconv_checkReInit(void* const hMCnv) { CXP; check_reinit(p); }
void matrixconv_setFilters(void* const hMCnv, const float** H, int numChannels, int numSamples, int sampleRate)      /* matrixconv.c:205-236 */
{
    CXP;
    if (!(numChannels <= 1024 && numChannels > 0 && numSamples > 0)) SAF_FATAL("matrixconv_setFilters: WAV is limited to 1024 channels");
    p->nOutputChannels = numChannels < SAF_MAXCH ? numChannels : SAF_MAXCH;
    p->input_wav_length = numSamples;
    p->nfilters = p->nOutputChannels * p->nInputChannels;
    p->filters.resize((size_t)numChannels * numSamples);
    for (int i = 0; i < numChannels; i++) memcpy(&p->filters[(size_t)i * numSamples], H[i], sizeof(float) * numSamples);
    p->haveFilters = true;
    p->filter_fs = sampleRate;
    p->filter_length = p->input_wav_length % p->nInputChannels == 0 ? p->input_wav_length / p->nInputChannels : 0;
    p->reInitFilters = 1;
}
void matrixconv_setEnablePart(void* const hMCnv, int newState) { CXP; if (p->enablePartitionedConv != newState) { p->enablePartitionedConv = newState; p->reInitFilters = 1; } }
void matrixconv_setNumInputChannels(void* const hMCnv, int newValue)
{
    CXP;
    p->nInputChannels = clampi(newValue, 1, SAF_MAXCH);
    p->nfilters = p->nOutputChannels * p->nInputChannels;
    p->filter_length = (p->nOutputChannels > 0 && p->input_wav_length % p->nInputChannels == 0) ? p->input_wav_length / p->nInputChannels : 0;
    p->reInitFilters = 1;
}
int matrixconv_getEnablePart(void* const hMCnv) { CXP; return p->enablePartitionedConv; }
int matrixconv_getNumInputChannels(void* const hMCnv) { CXP; return p->nInputChannels; }
int matrixconv_getNumOutputChannels(void* const hMCnv) { CXP; return p->nOutputChannels; }
int matrixconv_getHostBlockSize(void* const hMCnv) { CXP; return p->hostBlockSize; }
int matrixconv_getNfilters(void* const hMCnv) { CXP; return p->nfilters; }
int matrixconv_getFilterLength(void* const hMCnv) { CXP; return p->filter_length; }
int matrixconv_getFilterFs(void* const hMCnv) { CXP; return p->filter_fs; }
int matrixconv_getHostFs(void* const hMCnv) { CXP; return p->host_fs; }
int matrixconv_getProcessingDelay(void* const hMCnv) { CXP; return p->hostBlockSize_clamped; }

/* ---------------- multiconv (multiconv.h:53-168) ---------------- */
void multiconv_create(void** const phMCnv) { *phMCnv = create_example(false); }
void multiconv_destroy(void** const phMCnv) { matrixconv_destroy(phMCnv); }
void multiconv_init(void* const hMCnv, int sampleRate, int hostBlockSize) { CXP; init_example(p, sampleRate, hostBlockSize); }
void multiconv_process(void* const hMCnv, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples) { CXP; process_example(p, inputs, outputs, nInputs, nOutputs, nSamples); }
void multiconv_refreshParams(void* const hMCnv) { CXP; p->reInitFilters = 1; }
void multiconv_checkReInit(void* const hMCnv) { CXP; check_reinit(p); }
void multiconv_setFilters(void* const hMCnv, const float** H, int numChannels, int numSamples, int sampleRate)      /* multiconv.c:194-211 */
{
    CXP;
    p->filters.resize((size_t)numChannels * numSamples);
    p->nfilters = numChannels; p->filter_length = numSamples;
    for (int i = 0; i < numChannels; i++) memcpy(&p->filters[(size_t)i * numSamples], H[i], sizeof(float) * numSamples);
    p->haveFilters = true;
    p->filter_fs = sampleRate;
    p->reInitFilters = 1;
}
void multiconv_setEnablePart(void* const hMCnv, int newState) { matrixconv_setEnablePart(hMCnv, newState); }
void multiconv_setNumChannels(void* const hMCnv, int newValue) { CXP; p->nInputChannels = clampi(newValue, 1, SAF_MAXCH); }
int multiconv_getEnablePart(void* const hMCnv) { CXP; return p->enablePartitionedConv; }
int multiconv_getNumChannels(void* const hMCnv) { CXP; return p->nInputChannels; }
int multiconv_getHostBlockSize(void* const hMCnv) { CXP; return p->hostBlockSize; }
int multiconv_getNfilters(void* const hMCnv) { CXP; return p->nfilters; }
int multiconv_getFilterLength(void* const hMCnv) { CXP; return p->filter_length; }
int multiconv_getFilterFs(void* const hMCnv) { CXP; return p->filter_fs; }
int multiconv_getHostFs(void* const hMCnv) { CXP; return p->host_fs; }
int multiconv_getProcessingDelay(void* const hMCnv) { CXP; return p->hostBlockSize_clamped; }

}
