/*
 * ambi_enc.cpp — the ambi_enc operator (examples/include/ambi_enc.h:55-222,
 * examples/src/ambi_enc/ambi_enc.c) with its per-block path on the GPU:
 *
 *   [per moved source: getRSH_recur on the device -> column of Y]            ambi_enc.c:120-131
 *   prev frame (after gains) -> [MFMA GEMM with Y (and prev_Y + linear cross-fade when a direction
 *   changed) -> 1/sqrt(nSources) -> ACN/N3D to the output convention] -> outputs   ambi_enc.c:138-190
 *
 * Like the reference, a call encodes the PREVIOUS block (one block of latency,
 * ambi_enc_getProcessingDelay).  The device pipeline is shared by the single-handle
 * ambi_enc_process (host pointers, one block) and the batched device-pointer entry point.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "design_host.h"
#include "presets.h"

namespace saf {

static int g_ambi_enc_frame_size = 64;      /* default of the reference (ambi_enc_internal.h:45) */

struct EncPipeline;

struct AmbiEnc {
    int F;
    float fs = 48000.0f;
    int recalc_SH_FLAG[SAF_MAXCH];
    float src_dirs_deg[SAF_MAXCH][2];
    float src_gains[SAF_MAXCH];
    int nSources, new_nSources, order, enablePostScaling;
    CH_ORDER chOrdering;
    NORM_TYPES norm;
    unsigned long long initEpoch = 1;       /* bumped by ambi_enc_init: pipelines zero prev_Y / the previous frame */
    EncPipeline* pipe = nullptr;
    PinBuf<float> h_in, h_out;
    DevBuf<float> d_in, d_out;
};

struct EncPipeline {
    int nInst = 0, F = 0, maxFrames = 0;
    std::vector<AmbiEnc*> inst;
    DevBuf<float> YY;        /* [nInst][2 = {Y, prev_Y}][64][64] row-major, row stride MAX_NUM_INPUTS */
    DevBuf<float> Afrag;     /* [nInst][2][2][32][64] MFMA fragment order */
    DevBuf<float> prev[2];   /* [nInst][64][F] previous block after gains (ping-pong) */
    DevBuf<float> fpar;      /* [nInst] postScale | [nInst][64] gains | [nInst][64] rowScale | [nInst][64][2] dirs */
    DevBuf<int> ipar;        /* [nInst] nSrc | [nInst] mix | [nInst] order | [nInst][64] rowMap | [nInst][64] recalc */
    PinBuf<float> hf; PinBuf<int> hi;
    std::vector<float> shadowF; std::vector<int> shadowI;
    std::vector<unsigned long long> epoch;
    int par = 0;
    bool first = true;

    size_t fN() const { return (size_t)nInst * (1 + 64 + 64 + 128); }
    size_t iN() const { return (size_t)nInst * (3 + 64 + 64); }

    void create(AmbiEnc* const* handles, int n, int maxFrames_)
    {
        nInst = n; inst.assign(handles, handles + n); F = inst[0]->F; maxFrames = maxFrames_;
        for (int i = 0; i < n; i++) if (inst[i]->F != F) SAF_FATAL("ambi_enc batch: all instances must use the same block size");
        YY.alloc((size_t)n * 2 * 4096);
        Afrag.alloc((size_t)n * 2 * 4096);
        prev[0].alloc((size_t)n * SAF_MAXCH * F); prev[1].alloc((size_t)n * SAF_MAXCH * F);
        fpar.alloc(fN()); ipar.alloc(iN());
        hf.ensure(fN()); hi.ensure(iN());
        shadowF.assign(fN(), -12345.0f); shadowI.assign(iN(), -12345);
        epoch.assign(n, 0);
    }

    void process(const float* d_in, long long in_inst, long long in_frame, long long in_ch, int nIn,
                 float* d_out, long long out_inst, long long out_frame, long long out_ch, int nOut, int nFrames)
    {
        if (nFrames <= 0) return;
        if (nFrames > maxFrames) SAF_FATAL("ambi_enc batch: nFrames %d exceeds the maxFramesPerCall %d given at creation", nFrames, maxFrames);
        const int n = nInst;
        /* ---- snapshot of the user parameters (ambi_enc.c:100-111) into flat tables ---- */
        std::vector<float> f(fN()); std::vector<int> ii(iN());
        float* postScale = f.data(); float* gains = postScale + n; float* rowScale = gains + 64 * n; float* dirs = rowScale + 64 * n;
        int* nSrc = ii.data(); int* mix = nSrc + n; int* order = mix + n; int* rowMap = order + n; int* recalc = rowMap + 64 * n;
        bool anyRecalc = false;
        for (int i = 0; i < n; i++) {
            AmbiEnc* p = inst[i];
            if (epoch[i] != p->initEpoch) {
                /* ambi_enc_init (ambi_enc.c:80-83): prev_Y and the previous frame are cleared */
                HIP_CHECK(hipMemsetAsync(YY.p + ((size_t)i * 2 + 1) * 4096, 0, sizeof(float) * 4096, stream()));
                HIP_CHECK(hipMemsetAsync(prev[par].p + (size_t)i * SAF_MAXCH * F, 0, sizeof(float) * SAF_MAXCH * F, stream()));
                epoch[i] = p->initEpoch;
            }
            const int ord = p->order < SAF_MAX_ORDER ? p->order : SAF_MAX_ORDER, nSH = ORDER2NSH(ord);
            const int nS = p->nSources;
            order[i] = ord;
            nSrc[i] = nS < nIn ? nS : nIn;
            mix[i] = 0;
            for (int ch = 0; ch < SAF_MAXCH; ch++) {
                const bool rc = ch < nS && p->recalc_SH_FLAG[ch];
                recalc[i * 64 + ch] = rc ? 1 : 0;
                if (rc) { mix[i] = 1; anyRecalc = true; p->recalc_SH_FLAG[ch] = 0; }
                dirs[(i * 64 + ch) * 2 + 0] = p->src_dirs_deg[ch][0];
                dirs[(i * 64 + ch) * 2 + 1] = p->src_dirs_deg[ch][1];
                gains[i * 64 + ch] = fabsf(p->src_gains[ch] - 1.0f) > 1e-6f ? p->src_gains[ch] : 1.0f;      /* ambi_enc.c:134-135 */
                rowScale[i * 64 + ch] = 1.0f;
                rowMap[i * 64 + ch] = ch;
            }
            postScale[i] = p->enablePostScaling ? 1.0f / sqrtf((float)nS) : 1.0f;                          /* ambi_enc.c:168-171 */
            if (p->chOrdering == CH_FUMA) { rowMap[i * 64 + 1] = 2; rowMap[i * 64 + 2] = 3; rowMap[i * 64 + 3] = 1; }   /* ACN WYZX -> FuMa WXYZ (saf_hoa.c:40-70) */
            if (p->norm == NORM_SN3D) {
                for (int k = 0; k <= ord; k++)
                    for (int ch = k * k; ch < ORDER2NSH(k); ch++) rowScale[i * 64 + ch] = 1.0f / sqrtf(2.0f * (float)k + 1.0f);   /* saf_hoa.c:88-93 */
            } else if (p->norm == NORM_FUMA) {
                rowScale[i * 64] = 1.0f / sqrtf(2.0f);
                for (int ch = 1; ch < 4 && ch < nSH; ch++) rowScale[i * 64 + ch] = 1.0f / sqrtf(3.0f);
            }
        }
        const bool fchg = first || memcmp(f.data(), shadowF.data(), sizeof(float) * f.size()) != 0;
        /* mix / recalc are only consumed by a call that recalculates: they never force an upload on their own */
        if (!anyRecalc) { memcpy(mix, shadowI.data() + (mix - ii.data()), sizeof(int) * n); memcpy(recalc, shadowI.data() + (recalc - ii.data()), sizeof(int) * 64 * n); }
        const bool ichg = first || memcmp(ii.data(), shadowI.data(), sizeof(int) * ii.size()) != 0;
        if (fchg || ichg) {
            HIP_CHECK(hipStreamSynchronize(stream()));          /* staging may still be in flight from an earlier call */
            if (fchg) { memcpy(hf.p, f.data(), sizeof(float) * f.size()); HIP_CHECK(hipMemcpyAsync(fpar.p, hf.p, sizeof(float) * f.size(), hipMemcpyHostToDevice, stream())); shadowF = f; }
            if (ichg) { memcpy(hi.p, ii.data(), sizeof(int) * ii.size()); HIP_CHECK(hipMemcpyAsync(ipar.p, hi.p, sizeof(int) * ii.size(), hipMemcpyHostToDevice, stream())); shadowI = ii; }
            first = false;
        }
        const float* d_post = fpar.p; const float* d_gains = d_post + n; const float* d_rowScale = d_gains + 64 * n; const float* d_dirs = d_rowScale + 64 * n;
        const int* d_nSrc = ipar.p; const int* d_mix = d_nSrc + n; const int* d_order = d_mix + n; const int* d_rowMap = d_order + n; const int* d_recalc = d_rowMap + 64 * n;

        if (anyRecalc) {
            launch_enc_update_Y(d_order, d_dirs, d_recalc, YY.p, 2 * 4096, n);
            launch_pack_A(YY.p, Afrag.p, 2 * n);
        }
        EncLaunch e{};
        e.in = d_in; e.in_inst = in_inst; e.in_frame = in_frame; e.in_ch = in_ch;
        e.out = d_out; e.out_inst = out_inst; e.out_frame = out_frame; e.out_ch = out_ch;
        e.prev_rd = prev[par].p; e.prev_wr = prev[par ^ 1].p;
        e.Afrag = Afrag.p; e.gains = d_gains; e.postScale = d_post; e.rowScale = d_rowScale; e.rowMap = d_rowMap;
        e.nSrc = d_nSrc; e.order = d_order; e.mix = anyRecalc ? d_mix : nullptr;
        e.F = F; e.nFrames = nFrames; e.nInst = n; e.nOut = nOut < SAF_MAXCH ? nOut : SAF_MAXCH;
        for (int i = 0; i < n; i++) { e.maxSteps = std::max(e.maxSteps, (shadowI[i] + 1) / 2); e.rowsIn = std::max(e.rowsIn, shadowI[i]); }
        launch_enc_gemm(e);
        par ^= 1;
        if (anyRecalc)      /* prev_Y <- Y for the instances that mixed (ambi_enc.c:162) */
            for (int i = 0; i < n; i++)
                if (shadowI[n + i]) HIP_CHECK(hipMemcpyAsync(YY.p + ((size_t)i * 2 + 1) * 4096, YY.p + (size_t)i * 2 * 4096, sizeof(float) * 4096, hipMemcpyDeviceToDevice, stream()));
    }
};

}  // namespace saf

using namespace saf;

extern "C" {

void saf_hip_ambi_enc_setFrameSize(int frameSize)
{
    if (frameSize <= 0 || frameSize % 4 != 0) SAF_FATAL("ambi_enc frame size must be a positive multiple of 4");
    g_ambi_enc_frame_size = frameSize;
}

void ambi_enc_create(void** const phAmbi)
{
    AmbiEnc* p = new AmbiEnc();
    *phAmbi = p;
    p->F = g_ambi_enc_frame_size;
    load_source_preset(SOURCE_CONFIG_PRESET_DEFAULT, p->src_dirs_deg, &p->new_nSources);
    p->nSources = p->new_nSources;
    for (int i = 0; i < SAF_MAXCH; i++) { p->recalc_SH_FLAG[i] = 1; p->src_gains[i] = 1.0f; }
    p->chOrdering = CH_ACN; p->norm = NORM_SN3D; p->order = 1; p->enablePostScaling = 1;
}

void ambi_enc_destroy(void** const phAmbi)
{
    AmbiEnc* p = (AmbiEnc*)*phAmbi;
    if (!p) return;
    if (p->pipe) { HIP_CHECK(hipStreamSynchronize(stream())); delete p->pipe; }
    delete p;
    *phAmbi = nullptr;
}

void ambi_enc_init(void* const hAmbi, int sampleRate)
{
    AmbiEnc* p = (AmbiEnc*)hAmbi;
    p->fs = (float)sampleRate;
    p->initEpoch++;
    for (int i = 0; i < SAF_MAXCH; i++) p->recalc_SH_FLAG[i] = 1;
}

void ambi_enc_process(void* const hAmbi, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)
{
    AmbiEnc* p = (AmbiEnc*)hAmbi;
    const int F = p->F;
    if (nSamples != F) {                                                  /* ambi_enc.c:192-195 */
        for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
        return;
    }
    ensure_device();
    if (!p->pipe) { p->pipe = new EncPipeline(); AmbiEnc* h = p; p->pipe->create(&h, 1, 1); }
    const int nIn = nInputs < SAF_MAXCH ? (nInputs < 0 ? 0 : nInputs) : SAF_MAXCH;
    const int nOut = nOutputs < SAF_MAXCH ? (nOutputs < 0 ? 0 : nOutputs) : SAF_MAXCH;
    p->h_in.ensure((size_t)SAF_MAXCH * F); p->h_out.ensure((size_t)SAF_MAXCH * F);
    if (!p->d_in.p) { p->d_in.alloc((size_t)SAF_MAXCH * F); p->d_out.alloc((size_t)SAF_MAXCH * F); }
    for (int ch = 0; ch < nIn; ch++) memcpy(p->h_in.p + (size_t)ch * F, inputs[ch], sizeof(float) * F);
    if (zero_copy_io()) p->pipe->process(p->h_in.p, 0, 0, F, nIn, p->h_out.p, 0, 0, F, nOut, 1);      /* kernels on the pinned blocks */
    else {
        if (nIn) HIP_CHECK(hipMemcpyAsync(p->d_in.p, p->h_in.p, sizeof(float) * (size_t)nIn * F, hipMemcpyHostToDevice, stream()));
        p->pipe->process(p->d_in.p, 0, 0, F, nIn, p->d_out.p, 0, 0, F, nOut, 1);
        if (nOut) HIP_CHECK(hipMemcpyAsync(p->h_out.p, p->d_out.p, sizeof(float) * (size_t)nOut * F, hipMemcpyDeviceToHost, stream()));
    }
    HIP_CHECK(hipStreamSynchronize(stream()));
    for (int ch = 0; ch < nOut; ch++) memcpy(outputs[ch], p->h_out.p + (size_t)ch * F, sizeof(float) * F);
    for (int ch = nOut; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
}

/* ------------------------------- set functions (ambi_enc.c:205-330) ------------------------------- */
#define PE AmbiEnc* p = (AmbiEnc*)hAmbi
void ambi_enc_refreshParams(void* const hAmbi) { PE; for (int i = 0; i < SAF_MAXCH; i++) p->recalc_SH_FLAG[i] = 1; }
void ambi_enc_setOutputOrder(void* const hAmbi, int newOrder)
{
    PE;
    if (newOrder != p->order) {
        p->order = newOrder;
        for (int i = 0; i < SAF_MAXCH; i++) p->recalc_SH_FLAG[i] = 1;
        if (p->order != 1 && p->chOrdering == CH_FUMA) p->chOrdering = CH_ACN;      /* FuMa is first-order only */
        if (p->order != 1 && p->norm == NORM_FUMA) p->norm = NORM_SN3D;
    }
}
void ambi_enc_setSourceAzi_deg(void* const hAmbi, int index, float v)
{
    PE;
    if (v > 180.0f) v = -360.0f + v;
    v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    p->recalc_SH_FLAG[index] = 1; p->src_dirs_deg[index][0] = v;
}
void ambi_enc_setSourceElev_deg(void* const hAmbi, int index, float v)
{
    PE;
    v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    p->recalc_SH_FLAG[index] = 1; p->src_dirs_deg[index][1] = v;
}
void ambi_enc_setNumSources(void* const hAmbi, int n)
{
    PE;
    p->new_nSources = n < 1 ? 1 : (n > SAF_MAXCH ? SAF_MAXCH : n);
    p->nSources = p->new_nSources;
    for (int i = 0; i < SAF_MAXCH; i++) p->recalc_SH_FLAG[i] = 1;
}
void ambi_enc_setInputConfigPreset(void* const hAmbi, int newPresetID)
{
    PE;
    load_source_preset(newPresetID, p->src_dirs_deg, &p->new_nSources);
    p->nSources = p->new_nSources;
    for (int i = 0; i < SAF_MAXCH; i++) p->recalc_SH_FLAG[i] = 1;
}
void ambi_enc_setChOrder(void* const hAmbi, int v) { PE; if ((CH_ORDER)v != CH_FUMA || p->order == 1) p->chOrdering = (CH_ORDER)v; }
void ambi_enc_setNormType(void* const hAmbi, int v) { PE; if ((NORM_TYPES)v != NORM_FUMA || p->order == 1) p->norm = (NORM_TYPES)v; }
void ambi_enc_setEnablePostScaling(void* const hAmbi, int v) { PE; p->enablePostScaling = v; }
void ambi_enc_setSourceGain(void* const hAmbi, int srcIdx, float g) { PE; p->src_gains[srcIdx] = g; }
void ambi_enc_setSourceSolo(void* const hAmbi, int srcIdx) { PE; for (int i = 0; i < p->nSources; i++) p->src_gains[i] = i == srcIdx ? 1.0f : 0.0f; }
void ambi_enc_setUnSolo(void* const hAmbi) { PE; for (int i = 0; i < p->nSources; i++) p->src_gains[i] = 1.0f; }

/* ------------------------------- get functions (ambi_enc.c:333-400) ------------------------------- */
int ambi_enc_getFrameSize(void) { return g_ambi_enc_frame_size; }
int ambi_enc_getOutputOrder(void* const hAmbi) { PE; return p->order; }
float ambi_enc_getSourceAzi_deg(void* const hAmbi, int index) { PE; return p->src_dirs_deg[index][0]; }
float ambi_enc_getSourceElev_deg(void* const hAmbi, int index) { PE; return p->src_dirs_deg[index][1]; }
int ambi_enc_getNumSources(void* const hAmbi) { PE; return p->new_nSources; }
int ambi_enc_getMaxNumSources(void) { return SAF_MAXCH; }
int ambi_enc_getNSHrequired(void* const hAmbi) { PE; return (p->order + 1) * (p->order + 1); }
int ambi_enc_getChOrder(void* const hAmbi) { PE; return (int)p->chOrdering; }
int ambi_enc_getNormType(void* const hAmbi) { PE; return (int)p->norm; }
int ambi_enc_getEnablePostScaling(void* const hAmbi) { PE; return p->enablePostScaling; }
int ambi_enc_getProcessingDelay(void) { return g_ambi_enc_frame_size; }

/* ------------------------------- batched device-pointer entry point ------------------------------- */
void* saf_hip_ambi_enc_batch_create(void* const* hAmbis, int nInst, int maxFramesPerCall)
{
    ensure_device();
    if (nInst <= 0 || maxFramesPerCall <= 0) SAF_FATAL("ambi_enc batch: nInst and maxFramesPerCall must be positive");
    EncPipeline* b = new EncPipeline();
    b->create((AmbiEnc* const*)hAmbis, nInst, maxFramesPerCall);
    return b;
}
void saf_hip_ambi_enc_batch_destroy(void** const phBatch)
{
    EncPipeline* b = (EncPipeline*)*phBatch;
    if (!b) return;
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete b;
    *phBatch = nullptr;
}
void saf_hip_ambi_enc_batch_process(void* const hBatch,
                                    const float* d_in, long long in_inst_stride, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                    float* d_out, long long out_inst_stride, long long out_frame_stride, long long out_ch_stride, int nOutputs,
                                    int nFrames)
{
    EncPipeline* b = (EncPipeline*)hBatch;
    b->process(d_in, in_inst_stride, in_frame_stride, in_ch_stride, nInputs < SAF_MAXCH ? nInputs : SAF_MAXCH,
               d_out, out_inst_stride, out_frame_stride, out_ch_stride, nOutputs, nFrames);
}

}
