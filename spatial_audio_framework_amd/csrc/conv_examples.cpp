/*
 * conv_examples.cpp — the matrixconv and multiconv example operators (examples/include/matrixconv.h:53-191,
 * multiconv.h:53-168; examples/src/matrixconv/matrixconv.c, examples/src/multiconv/multiconv.c): a sample-wise FIFO that
 * collects one host block (clamped to 512..8192 samples), runs the convolver on it and plays the result back one block
 * later.  The convolvers are the GPU ones of matrixconv.cpp; everything here is host bookkeeping.
 * tvconv (examples/include/tvconv.h:42-174, examples/src/tvconv/tvconv.c) is the same FIFO around saf_TVConv, with the IR set
 * chosen by the listener position nearest to a target position.  The reference fills its IRs and positions from a SOFA file
 * (file I/O, not available here: tvconv_setSofaFilePath behaves like the reference built without its SOFA reader, tvconv.c:316-319);
 * saf_hip_tvconv_setIRsAndPositions installs the same arrays directly.
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
int matrixconv_getFrameSize(void) { return CX_MIN_FRAME; }
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
int multiconv_getFrameSize(void) { return CX_MIN_FRAME; }
int multiconv_getHostBlockSize(void* const hMCnv) { CXP; return p->hostBlockSize; }
int multiconv_getNfilters(void* const hMCnv) { CXP; return p->nfilters; }
int multiconv_getFilterLength(void* const hMCnv) { CXP; return p->filter_length; }
int multiconv_getFilterFs(void* const hMCnv) { CXP; return p->filter_fs; }
int multiconv_getHostFs(void* const hMCnv) { CXP; return p->host_fs; }
int multiconv_getProcessingDelay(void* const hMCnv) { CXP; return p->hostBlockSize_clamped; }

/* ---------------- tvconv (tvconv.h:42-174) ---------------- */
namespace saf {
struct TvExample {
    int FIFO_idx = 0;
    std::vector<float> inFIFO, outFIFO, inputFrameTD, outputFrameTD;
    void* hTVConv = nullptr;
    int hostBlockSize = -1, hostBlockSize_clamped = CX_MIN_FRAME, host_fs = 0;
    std::vector<std::vector<float>> irs;         /* [nListenerPositions][nIrChannels * ir_length] */
    std::vector<float> listenerPositions;        /* [n][3] */
    int nIrChannels = 0, ir_length = 0, ir_fs = 0, nInputChannels = 1, nOutputChannels = 0, nListenerPositions = 0, position_idx = 0, reInitFilters = 1;
    float sourcePosition[3] = { 0, 0, 0 }, targetPosition[3] = { 0, 0, 0 }, minDimensions[3] = { 0, 0, 0 }, maxDimensions[3] = { 0, 0, 0 };
    std::string sofa_filepath;
    volatile CODEC_STATUS codecStatus = CODEC_STATUS_NOT_INITIALISED;
    volatile PROC_STATUS procStatus = PROC_STATUS_NOT_ONGOING;
    float progressBar0_1 = 0.0f; char progressBarText[PROGRESSBARTEXT_CHAR_LENGTH] = { 0 };
};
/* tvconv_checkReInit (tvconv.c:196-232) */
static void tv_check_reinit(TvExample* p)
{
    if (p->reInitFilters == 1 && !p->irs.empty()) {
        p->reInitFilters = 2;
        if (p->hTVConv) { saf_TVConv_destroy(&p->hTVConv); p->hTVConv = nullptr; }
        p->hostBlockSize_clamped = clampi(p->hostBlockSize, CX_MIN_FRAME, CX_MAX_FRAME);
        if (p->ir_length > 0) {
            std::vector<float*> rows(p->nListenerPositions);
            for (int i = 0; i < p->nListenerPositions; i++) rows[i] = p->irs[i].data();
            saf_TVConv_create(&p->hTVConv, p->hostBlockSize_clamped, rows.data(), p->ir_length, p->nListenerPositions, p->nOutputChannels, p->position_idx);
        }
        p->inputFrameTD.assign((size_t)SAF_MAXCH * p->hostBlockSize_clamped, 0.0f); p->outputFrameTD.assign((size_t)SAF_MAXCH * p->hostBlockSize_clamped, 0.0f);
        p->FIFO_idx = 0;
        std::fill(p->inFIFO.begin(), p->inFIFO.end(), 0.0f); std::fill(p->outFIFO.begin(), p->outFIFO.end(), 0.0f);
        p->reInitFilters = 0;
        p->codecStatus = CODEC_STATUS_INITIALISED;
    }
}
/* tvconv_findNearestNeigbour (tvconv_internal.c:42-61) */
static void tv_nearest(TvExample* p)
{
    int min_idx = 0; float minDist = 0.0f;
    for (int i = 0; i < p->nListenerPositions; i++) {
        float dist = 0.0f;
        for (int d = 0; d < 3; d++) dist += (p->targetPosition[d] - p->listenerPositions[(size_t)i * 3 + d]) * (p->targetPosition[d] - p->listenerPositions[(size_t)i * 3 + d]);
        if (dist < minDist || i == 0) { minDist = dist; min_idx = i; }
    }
    p->position_idx = min_idx;
}
}  // namespace saf

#define TVP TvExample* p = (TvExample*)hTVCnv
void tvconv_create(void** const phTVCnv)
{
    TvExample* p = new TvExample();
    p->inFIFO.assign((size_t)SAF_MAXCH * CX_MAX_FRAME, 0.0f); p->outFIFO.assign((size_t)SAF_MAXCH * CX_MAX_FRAME, 0.0f);
    *phTVCnv = p;
}
void tvconv_destroy(void** const phTVCnv)
{
    TvExample* p = (TvExample*)*phTVCnv; if (!p) return;
    if (p->hTVConv) saf_TVConv_destroy(&p->hTVConv);
    delete p; *phTVCnv = nullptr;
}
void tvconv_init(void* const hTVCnv, int sampleRate, int hostBlockSize)          /* tvconv.c:98-117 */
{
    TVP;
    p->host_fs = sampleRate;
    if (p->hostBlockSize != hostBlockSize) {
        p->hostBlockSize = hostBlockSize; p->hostBlockSize_clamped = clampi(hostBlockSize, CX_MIN_FRAME, CX_MAX_FRAME);
        p->reInitFilters = 1; p->codecStatus = CODEC_STATUS_NOT_INITIALISED;
    }
    tv_check_reinit(p);
}
void tvconv_process(void* const hTVCnv, float** const inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)     /* tvconv.c:119-186 */
{
    TVP;
    tv_check_reinit(p);
    p->procStatus = PROC_STATUS_ONGOING;
    const int numIn = p->nInputChannels, numOut = p->nOutputChannels, B = p->hostBlockSize_clamped;
    for (int s = 0; s < nSamples; s++) {
        int ch;
        for (ch = 0; ch < std::min(std::min(nInputs, numIn), SAF_MAXCH); ch++) p->inFIFO[(size_t)ch * CX_MAX_FRAME + p->FIFO_idx] = inputs[ch][s];
        for (; ch < numIn; ch++) p->inFIFO[(size_t)ch * CX_MAX_FRAME + p->FIFO_idx] = 0.0f;
        for (ch = 0; ch < std::min(std::min(nOutputs, numOut), SAF_MAXCH); ch++) outputs[ch][s] = p->outFIFO[(size_t)ch * CX_MAX_FRAME + p->FIFO_idx];
        for (; ch < nOutputs; ch++) outputs[ch][s] = 0.0f;
        p->FIFO_idx++;
        if (p->FIFO_idx >= B && p->reInitFilters == 0 && p->codecStatus == CODEC_STATUS_INITIALISED) {
            p->FIFO_idx = 0;
            for (int i = 0; i < numIn; i++) memcpy(&p->inputFrameTD[(size_t)i * B], &p->inFIFO[(size_t)i * CX_MAX_FRAME], sizeof(float) * B);
            if (p->hTVConv && p->ir_length > 0) saf_TVConv_apply(p->hTVConv, p->inputFrameTD.data(), p->outputFrameTD.data(), p->position_idx);
            else std::fill(p->outputFrameTD.begin(), p->outputFrameTD.end(), 0.0f);
            for (int i = 0; i < std::min(numOut, SAF_MAXCH); i++) memcpy(&p->outFIFO[(size_t)i * CX_MAX_FRAME], &p->outputFrameTD[(size_t)i * B], sizeof(float) * B);
        } else if (p->FIFO_idx >= B) {
            p->FIFO_idx = 0;
            std::fill(p->outFIFO.begin(), p->outFIFO.end(), 0.0f);
        }
    }
    p->procStatus = PROC_STATUS_NOT_ONGOING;
}
void tvconv_refreshParams(void* const hTVCnv) { TVP; p->reInitFilters = 1; }
void tvconv_checkReInit(void* const hTVCnv) { TVP; tv_check_reinit(p); }
/* what tvconv_setFiltersAndPositions does with the contents of a SOFA file (tvconv.c:262-312) */
void saf_hip_tvconv_setIRsAndPositions(void* const hTVCnv, const float* const* irs, const float* listenerPositions, const float* sourcePosition,
                                       int nListenerPositions, int nIrChannels, int irLength, int irFs)
{
    TVP;
    p->codecStatus = CODEC_STATUS_INITIALISING;
    p->ir_fs = irFs; p->ir_length = irLength; p->nIrChannels = nIrChannels; p->nListenerPositions = nListenerPositions;
    for (int d = 0; d < 3; d++) p->sourcePosition[d] = sourcePosition ? sourcePosition[d] : 0.0f;
    p->irs.resize(nListenerPositions);
    for (int i = 0; i < nListenerPositions; i++) p->irs[i].assign(irs[i], irs[i] + (size_t)nIrChannels * irLength);
    p->listenerPositions.assign(listenerPositions, listenerPositions + (size_t)nListenerPositions * 3);
    p->nOutputChannels = nIrChannels < SAF_MAXCH ? nIrChannels : SAF_MAXCH;
    for (int d = 0; d < 3; d++) {       /* tvconv_setMinMaxDimensions (tvconv_internal.c:63-83) */
        p->minDimensions[d] = p->maxDimensions[d] = p->listenerPositions[d];
        for (int i = 1; i < nListenerPositions; i++) {
            const float v = p->listenerPositions[(size_t)i * 3 + d];
            if (v < p->minDimensions[d]) p->minDimensions[d] = v; else if (v > p->maxDimensions[d]) p->maxDimensions[d] = v;
        }
        p->targetPosition[d] = p->minDimensions[d];
    }
    p->position_idx = 0;
    p->codecStatus = CODEC_STATUS_INITIALISED;
    p->reInitFilters = 1;
    strcpy(p->progressBarText, "Done!"); p->progressBar0_1 = 1.0f;
}
void tvconv_setFiltersAndPositions(void* const hTVCnv)       /* the reference built without its SOFA reader (tvconv.c:316-324) */
{
    TVP;
    if (p->codecStatus != CODEC_STATUS_NOT_INITIALISED) return;
    p->ir_length = 0;
    fprintf(stderr, "libsaf_hip: tvconv reads IRs from SOFA files only in the reference; use saf_hip_tvconv_setIRsAndPositions\n");
    p->position_idx = 0; p->codecStatus = CODEC_STATUS_INITIALISED; p->reInitFilters = 1;
}
void tvconv_setSofaFilePath(void* const hTVCnv, const char* path) { TVP; p->sofa_filepath = path; p->codecStatus = CODEC_STATUS_NOT_INITIALISED; tvconv_setFiltersAndPositions(hTVCnv); }
void tvconv_setTargetPosition(void* const hTVCnv, float position, int dim)
{
    TVP;
    if (dim < 0 || dim >= 3) SAF_FATAL("tvconv_setTargetPosition: dimension out of scope");
    p->targetPosition[dim] = position;
    tv_nearest(p);
}
int tvconv_getNumInputChannels(void* const hTVCnv) { TVP; return p->nInputChannels; }
int tvconv_getNumOutputChannels(void* const hTVCnv) { TVP; return p->nOutputChannels; }
int tvconv_getFrameSize(void) { return CX_MIN_FRAME; }
int tvconv_getHostBlockSize(void* const hTVCnv) { TVP; return p->hostBlockSize; }
int tvconv_getNumIRs(void* const hTVCnv) { TVP; return p->nIrChannels; }
int tvconv_getNumListenerPositions(void* const hTVCnv) { TVP; return p->codecStatus == CODEC_STATUS_INITIALISED ? p->nListenerPositions : 0; }
float tvconv_getListenerPosition(void* const hTVCnv, int index, int dim) { TVP; return p->codecStatus == CODEC_STATUS_INITIALISED ? p->listenerPositions[(size_t)index * 3 + dim] : 0.0f; }
int tvconv_getListenerPositionIdx(void* const hTVCnv) { TVP; return p->position_idx; }
float tvconv_getTargetPosition(void* const hTVCnv, int dim) { TVP; return p->targetPosition[dim]; }
float tvconv_getSourcePosition(void* const hTVCnv, int dim) { TVP; return p->sourcePosition[dim]; }
float tvconv_getMinDimension(void* const hTVCnv, int dim) { TVP; return p->minDimensions[dim]; }
float tvconv_getMaxDimension(void* const hTVCnv, int dim) { TVP; return p->maxDimensions[dim]; }
int tvconv_getIRLength(void* const hTVCnv) { TVP; return p->ir_length; }
int tvconv_getIRFs(void* const hTVCnv) { TVP; return p->ir_fs; }
int tvconv_getHostFs(void* const hTVCnv) { TVP; return p->host_fs; }
int tvconv_getProcessingDelay(void* const hTVCnv) { TVP; return p->hostBlockSize_clamped; }
char* tvconv_getSofaFilePath(void* const hTVCnv) { TVP; return p->sofa_filepath.empty() ? (char*)"no_file" : (char*)p->sofa_filepath.c_str(); }
CODEC_STATUS tvconv_getCodecStatus(void* const hTVCnv) { TVP; return p->codecStatus; }

}
