/*
 * afstft_api.cpp — the afSTFT_* C API (framework/resources/afSTFT/afSTFTlib.h:85-278)
 * on top of the device kernels in afstft_kernels.hip.
 *
 * Host-pointer calls stage the block through pinned memory, run the kernels and
 * copy the result into the caller's layout; the `_dev` entry points skip the
 * staging.  Filterbank state (input history, synthesised-frame history) lives
 * in device memory per handle.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "afstft_state.h"

namespace saf {

void AfState::create(int nInst_, int nIn, int nOut, int hop_)
{
    nInst = nInst_; nCHin = nIn; nCHout = nOut; hop = hop_;
    for (int i = 0; i < 2; i++) {
        ana[i].alloc((size_t)nInst * (nIn > 0 ? nIn : 1) * SAF_ANA_HIST * hop);
        syn[i].alloc((size_t)nInst * (nOut > 0 ? nOut : 1) * SAF_SYN_HIST * 2 * hop);
    }
    anaPar = synPar = 0;
}

void AfState::clear()
{
    for (int i = 0; i < 2; i++) { ana[i].zero(); syn[i].zero(); }
}

/* surviving channels keep their state, new channels start from zero (afSTFT_internal.c:158-211) */
void AfState::channelChange(int newIn, int newOut)
{
    if (nInst != 1) SAF_FATAL("channelChange on a batched state is not supported");
    if (newIn != nCHin) {
        DevBuf<float> n0, n1;
        n0.alloc((size_t)(newIn > 0 ? newIn : 1) * SAF_ANA_HIST * hop);
        n1.alloc((size_t)(newIn > 0 ? newIn : 1) * SAF_ANA_HIST * hop);
        const int keep = newIn < nCHin ? newIn : nCHin;
        if (keep > 0)
            HIP_CHECK(hipMemcpyAsync(n0.p, ana[anaPar].p, (size_t)keep * SAF_ANA_HIST * hop * sizeof(float), hipMemcpyDeviceToDevice, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        std::swap(ana[0].p, n0.p); std::swap(ana[0].n, n0.n);
        std::swap(ana[1].p, n1.p); std::swap(ana[1].n, n1.n);
        anaPar = 0;
        nCHin = newIn;
    }
    if (newOut != nCHout) {
        DevBuf<float> n0, n1;
        n0.alloc((size_t)(newOut > 0 ? newOut : 1) * SAF_SYN_HIST * 2 * hop);
        n1.alloc((size_t)(newOut > 0 ? newOut : 1) * SAF_SYN_HIST * 2 * hop);
        const int keep = newOut < nCHout ? newOut : nCHout;
        if (keep > 0)
            HIP_CHECK(hipMemcpyAsync(n0.p, syn[synPar].p, (size_t)keep * SAF_SYN_HIST * 2 * hop * sizeof(float), hipMemcpyDeviceToDevice, stream()));
        HIP_CHECK(hipStreamSynchronize(stream()));
        std::swap(syn[0].p, n0.p); std::swap(syn[0].n, n0.n);
        std::swap(syn[1].p, n1.p); std::swap(syn[1].n, n1.n);
        synPar = 0;
        nCHout = newOut;
    }
}

struct AfSTFT {
    int hop, lowDelay, hybrid, nBands, delay;
    AFSTFT_FDDATA_FORMAT format;
    AfState st;
    DevBuf<float> d_td;
    DevBuf<float2> d_fd;
    PinBuf<float> h_td;
    PinBuf<float2> h_fd;
    void ensure(size_t td, size_t fd) {
        if (d_td.n < td) d_td.alloc(td, false);
        if (d_fd.n < fd) d_fd.alloc(fd, false);
        h_td.ensure(td); h_fd.ensure(fd);
    }
};

static void run_forward(AfSTFT* h, const float* d_td, long long td_ch, int nHops, float2* d_fd, long long fd_band, long long fd_ch)
{
    if (h->st.nCHin <= 0 || nHops <= 0) return;
    AnaLaunch a{};
    a.in = d_td; a.in_inst = 0; a.in_ch = td_ch; a.in_frame = 0; a.hopsPerFrame = nHops; a.nChIn = h->st.nCHin;
    a.hist_rd = h->st.ana[h->st.anaPar].p; a.hist_wr = h->st.ana[h->st.anaPar ^ 1].p;
    a.out = d_fd; a.out_inst = 0; a.out_band = fd_band; a.out_ch = fd_ch;
    a.ch_scale = nullptr; a.ch_map = nullptr;
    a.nCh = h->st.nCHin; a.nInst = 1; a.H = nHops; a.lowDelay = h->lowDelay; a.hybrid = h->hybrid; a.hop = h->hop;
    launch_analysis(a);
    h->st.anaPar ^= 1;
}

static void run_backward(AfSTFT* h, const float2* d_fd, long long fd_band, long long fd_ch, int nHops, float* d_td, long long td_ch)
{
    if (h->st.nCHout <= 0 || nHops <= 0) return;
    SynLaunch s{};
    s.in = d_fd; s.in_inst = 0; s.in_band = fd_band; s.in_ch = fd_ch;
    s.out = d_td; s.out_inst = 0; s.out_ch = td_ch; s.out_frame = 0; s.hopsPerFrame = nHops;
    s.hist_rd = h->st.syn[h->st.synPar].p; s.hist_wr = h->st.syn[h->st.synPar ^ 1].p;
    s.nCh = h->st.nCHout; s.nInst = 1; s.H = nHops; s.lowDelay = h->lowDelay; s.hybrid = h->hybrid; s.hop = h->hop;
    launch_synthesis(s);
    h->st.synPar ^= 1;
}

/* forward on host data gathered through `rd(ch)`; result lands in h->h_fd as [band][nCHin][nHops] */
template <typename Rd>
static void forward_host(AfSTFT* h, int framesize, Rd rd)
{
    if (framesize % h->hop != 0) SAF_FATAL("afSTFT: framesize must be a multiple of hopsize");   /* afSTFTlib.c:240 */
    const int nHops = framesize / h->hop, nCH = h->st.nCHin;
    h->ensure((size_t)(nCH > h->st.nCHout ? nCH : h->st.nCHout) * framesize, (size_t)h->nBands * (nCH > h->st.nCHout ? nCH : h->st.nCHout) * nHops);
    for (int ch = 0; ch < nCH; ch++) memcpy(h->h_td.p + (size_t)ch * framesize, rd(ch), sizeof(float) * framesize);
    if (zero_copy_io()) run_forward(h, h->h_td.p, framesize, nHops, h->d_fd.p, (long long)nCH * nHops, nHops);       /* the kernel reads the pinned samples directly */
    else {
        HIP_CHECK(hipMemcpyAsync(h->d_td.p, h->h_td.p, sizeof(float) * (size_t)nCH * framesize, hipMemcpyHostToDevice, stream()));
        run_forward(h, h->d_td.p, framesize, nHops, h->d_fd.p, (long long)nCH * nHops, nHops);
    }
    HIP_CHECK(hipMemcpyAsync(h->h_fd.p, h->d_fd.p, sizeof(float2) * (size_t)h->nBands * nCH * nHops, hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
}

/* backward: h->h_fd must hold [band][nCHout][nHops]; result in h->h_td as [nCHout][framesize] */
static void backward_host(AfSTFT* h, int framesize)
{
    const int nHops = framesize / h->hop, nCH = h->st.nCHout;
    HIP_CHECK(hipMemcpyAsync(h->d_fd.p, h->h_fd.p, sizeof(float2) * (size_t)h->nBands * nCH * nHops, hipMemcpyHostToDevice, stream()));
    if (zero_copy_io()) run_backward(h, h->d_fd.p, (long long)nCH * nHops, nHops, nHops, h->h_td.p, framesize);      /* the kernel writes the pinned samples directly */
    else {
        run_backward(h, h->d_fd.p, (long long)nCH * nHops, nHops, nHops, h->d_td.p, framesize);
        HIP_CHECK(hipMemcpyAsync(h->h_td.p, h->d_td.p, sizeof(float) * (size_t)nCH * framesize, hipMemcpyDeviceToHost, stream()));
    }
    HIP_CHECK(hipStreamSynchronize(stream()));
}

}  // namespace saf

using namespace saf;

extern "C" {

void afSTFT_create(void** const phSTFT, int nCHin, int nCHout, int hopsize, int lowDelayMode, int hybridmode, AFSTFT_FDDATA_FORMAT format)
{
    ensure_device();
    /* afSTFTlib.c:158-159 restricts the hop to 64, 128 or 256 in HYBRID mode only; without the hybrid filters the reference takes any
     * hop that divides 1024.  Here 64 / 128 / 256 exist in both modes (128: the tuned kernels, 64 / 256: the generic ones); the other
     * non-hybrid hops (32, 512, 1024 ...) are a stated limitation of this library (include/saf_hip.h), not of the reference. */
    if (hopsize != 64 && hopsize != 128 && hopsize != 256)
        SAF_FATAL("afSTFT_create: hopsize %d is not supported by libsaf_hip (64, 128 or 256; the reference also accepts other divisors of 1024 when hybridmode = 0)", hopsize);
    AfSTFT* h = new AfSTFT();
    h->hop = hopsize; h->lowDelay = lowDelayMode ? 1 : 0; h->hybrid = hybridmode ? 1 : 0; h->format = format;
    h->nBands = hybridmode ? hopsize + 5 : hopsize + 1;                       /* afSTFTlib.c:165 */
    if (lowDelayMode) h->delay = hybridmode ? 7 * hopsize : 4 * hopsize;      /* afSTFTlib.c:166-169 */
    else              h->delay = hybridmode ? 12 * hopsize : 9 * hopsize;
    h->st.create(1, nCHin, nCHout, hopsize);
    *phSTFT = h;
}

/* afAnalyse (afSTFTlib.c:78-119): one-shot analysis of nCH interleaved signals with a FRESH filterbank: the input is
 * zero-padded to whole hops, outTF is [nBands][nTimeslots][nCH].  Same device path as afSTFT_forward. */
void afAnalyse(float* inTD, int nSamplesTD, int nCH, int hopSize, int LDmode, int hybridmode, float_complex* outTF)
{
    if (nCH <= 0 || nSamplesTD <= 0) return;
    void* hv = nullptr;
    afSTFT_create(&hv, nCH, 1, hopSize, LDmode, hybridmode, AFSTFT_TIME_CH_BANDS);
    AfSTFT* h = (AfSTFT*)hv;
    const int nTimeSlots = (int)((float)nSamplesTD / (float)hopSize + 0.9999f);      /* the reference's "ceil" */
    const int framesize = nTimeSlots * hopSize;
    std::vector<float> td((size_t)nCH * framesize, 0.0f);
    for (int ch = 0; ch < nCH; ch++)
        for (int n = 0; n < nSamplesTD; n++) td[(size_t)ch * framesize + n] = inTD[(size_t)n * nCH + ch];
    forward_host(h, framesize, [&](int ch) { return td.data() + (size_t)ch * framesize; });
    const float2* r = h->h_fd.p;                                                     /* [band][ch][hop] */
    for (int band = 0; band < h->nBands; band++)
        for (int t = 0; t < nTimeSlots; t++)
            for (int ch = 0; ch < nCH; ch++) {
                const float2 v = r[((size_t)band * nCH + ch) * nTimeSlots + t];
                outTF[((size_t)band * nTimeSlots + t) * nCH + ch] = float_complex(v.x, v.y);
            }
    afSTFT_destroy(&hv);
}

void afSTFT_destroy(void** const phSTFT)
{
    if (!phSTFT || !*phSTFT) return;
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete (AfSTFT*)*phSTFT;
    *phSTFT = nullptr;                                                        /* afSTFTlib.c:223-225 */
}

void afSTFT_forward(void* const hSTFT, float** dataTD, int framesize, float_complex*** dataFD)
{
    AfSTFT* h = (AfSTFT*)hSTFT;
    forward_host(h, framesize, [&](int ch) { return dataTD[ch]; });
    const int nHops = framesize / h->hop, nCH = h->st.nCHin;
    const float2* r = h->h_fd.p;
    for (int band = 0; band < h->nBands; band++)
        for (int ch = 0; ch < nCH; ch++)
            for (int t = 0; t < nHops; t++) {
                const float2 v = r[((size_t)band * nCH + ch) * nHops + t];
                if (h->format == AFSTFT_BANDS_CH_TIME) dataFD[band][ch][t] = float_complex(v.x, v.y);
                else dataFD[t][ch][band] = float_complex(v.x, v.y);
            }
}

void afSTFT_forward_knownDimensions(void* const hSTFT, float** dataTD, int framesize, int dataFD_nCH, int dataFD_nHops, float_complex*** dataFD)
{
    AfSTFT* h = (AfSTFT*)hSTFT;
    forward_host(h, framesize, [&](int ch) { return dataTD[ch]; });
    const int nHops = framesize / h->hop, nCH = h->st.nCHin;
    const float2* r = h->h_fd.p;
    if (h->format == AFSTFT_BANDS_CH_TIME) {
        float2* flat = (float2*)&dataFD[0][0][0];                             /* afSTFTlib.c:283,296-297 */
        for (int band = 0; band < h->nBands; band++)
            for (int ch = 0; ch < nCH; ch++)
                memcpy(&flat[(size_t)band * dataFD_nCH * dataFD_nHops + (size_t)ch * dataFD_nHops],
                       &r[((size_t)band * nCH + ch) * nHops], sizeof(float2) * nHops);
    } else {
        for (int t = 0; t < nHops; t++)
            for (int ch = 0; ch < nCH; ch++)
                for (int band = 0; band < h->nBands; band++) {
                    const float2 v = r[((size_t)band * nCH + ch) * nHops + t];
                    dataFD[t][ch][band] = float_complex(v.x, v.y);
                }
    }
}

void afSTFT_forward_flat(void* const hSTFT, float* dataTD, int framesize, float_complex* dataFD)
{
    AfSTFT* h = (AfSTFT*)hSTFT;
    forward_host(h, framesize, [&](int ch) { return dataTD + (size_t)ch * framesize; });
    const int nHops = framesize / h->hop, nCH = h->st.nCHin;
    const float2* r = h->h_fd.p;
    if (h->format == AFSTFT_BANDS_CH_TIME)
        memcpy((void*)dataFD, r, sizeof(float2) * (size_t)h->nBands * nCH * nHops);
    else
        for (int t = 0; t < nHops; t++)
            for (int ch = 0; ch < nCH; ch++)
                for (int band = 0; band < h->nBands; band++) {
                    const float2 v = r[((size_t)band * nCH + ch) * nHops + t];
                    dataFD[((size_t)t * nCH + ch) * h->nBands + band] = float_complex(v.x, v.y);
                }
}

void afSTFT_backward(void* const hSTFT, float_complex*** dataFD, int framesize, float** dataTD)
{
    AfSTFT* h = (AfSTFT*)hSTFT;
    if (framesize % h->hop != 0) SAF_FATAL("afSTFT: framesize must be a multiple of hopsize");
    const int nHops = framesize / h->hop, nCH = h->st.nCHout;
    h->ensure((size_t)(nCH > h->st.nCHin ? nCH : h->st.nCHin) * framesize, (size_t)h->nBands * (nCH > h->st.nCHin ? nCH : h->st.nCHin) * nHops);
    float2* w = h->h_fd.p;
    for (int band = 0; band < h->nBands; band++)
        for (int ch = 0; ch < nCH; ch++)
            for (int t = 0; t < nHops; t++) {
                const float_complex v = (h->format == AFSTFT_BANDS_CH_TIME) ? dataFD[band][ch][t] : dataFD[t][ch][band];
                w[((size_t)band * nCH + ch) * nHops + t] = make_float2(v.real(), v.imag());
            }
    backward_host(h, framesize);
    for (int ch = 0; ch < nCH; ch++) memcpy(dataTD[ch], h->h_td.p + (size_t)ch * framesize, sizeof(float) * framesize);
}

void afSTFT_backward_knownDimensions(void* const hSTFT, float_complex*** dataFD, int framesize, int dataFD_nCH, int dataFD_nHops, float** dataTD)
{
    AfSTFT* h = (AfSTFT*)hSTFT;
    if (framesize % h->hop != 0) SAF_FATAL("afSTFT: framesize must be a multiple of hopsize");
    const int nHops = framesize / h->hop, nCH = h->st.nCHout;
    h->ensure((size_t)(nCH > h->st.nCHin ? nCH : h->st.nCHin) * framesize, (size_t)h->nBands * (nCH > h->st.nCHin ? nCH : h->st.nCHin) * nHops);
    float2* w = h->h_fd.p;
    if (h->format == AFSTFT_BANDS_CH_TIME) {
        const float2* flat = (const float2*)&dataFD[0][0][0];                 /* afSTFTlib.c:406,414-415 */
        for (int band = 0; band < h->nBands; band++)
            for (int ch = 0; ch < nCH; ch++)
                memcpy(&w[((size_t)band * nCH + ch) * nHops],
                       &flat[(size_t)band * dataFD_nCH * dataFD_nHops + (size_t)ch * dataFD_nHops], sizeof(float2) * nHops);
    } else {
        for (int t = 0; t < nHops; t++)
            for (int ch = 0; ch < nCH; ch++)
                for (int band = 0; band < h->nBands; band++) {
                    const float_complex v = dataFD[t][ch][band];
                    w[((size_t)band * nCH + ch) * nHops + t] = make_float2(v.real(), v.imag());
                }
    }
    backward_host(h, framesize);
    for (int ch = 0; ch < nCH; ch++) memcpy(dataTD[ch], h->h_td.p + (size_t)ch * framesize, sizeof(float) * framesize);
}

void afSTFT_backward_flat(void* const hSTFT, float_complex* dataFD, int framesize, float* dataTD)
{
    AfSTFT* h = (AfSTFT*)hSTFT;
    if (framesize % h->hop != 0) SAF_FATAL("afSTFT: framesize must be a multiple of hopsize");
    const int nHops = framesize / h->hop, nCH = h->st.nCHout;
    h->ensure((size_t)(nCH > h->st.nCHin ? nCH : h->st.nCHin) * framesize, (size_t)h->nBands * (nCH > h->st.nCHin ? nCH : h->st.nCHin) * nHops);
    float2* w = h->h_fd.p;
    if (h->format == AFSTFT_BANDS_CH_TIME)
        memcpy(w, (const void*)dataFD, sizeof(float2) * (size_t)h->nBands * nCH * nHops);
    else
        for (int t = 0; t < nHops; t++)
            for (int ch = 0; ch < nCH; ch++)
                for (int band = 0; band < h->nBands; band++) {
                    const float_complex v = dataFD[((size_t)t * nCH + ch) * h->nBands + band];
                    w[((size_t)band * nCH + ch) * nHops + t] = make_float2(v.real(), v.imag());
                }
    backward_host(h, framesize);
    memcpy(dataTD, h->h_td.p, sizeof(float) * (size_t)nCH * framesize);
}

void afSTFT_channelChange(void* const hSTFT, int new_nCHin, int new_nCHout)
{
    ((AfSTFT*)hSTFT)->st.channelChange(new_nCHin, new_nCHout);
}

void afSTFT_clearBuffers(void* const hSTFT) { ((AfSTFT*)hSTFT)->st.clear(); }
int afSTFT_getNBands(void* const hSTFT) { return ((AfSTFT*)hSTFT)->nBands; }
int afSTFT_getProcDelay(void* const hSTFT) { return ((AfSTFT*)hSTFT)->delay; }

/* afSTFTlib.c:545-590.  NULL handle: the measured 48k / 44.1k centre-frequency tables; valid
 * handle: uniform bin centres with the first five mapped through the 9x5 hybrid matrix
 * (afSTFTlib.c:65-74: one non-zero per row). */
void afSTFT_getCentreFreqs(void* const hSTFT, float fs, int nBands, float* freqVector)
{
    if (!hSTFT) {
        if (nBands < SAF_NBANDS) SAF_FATAL("afSTFT_getCentreFreqs: nBands must be >= 133 with a NULL handle");
        const float* tab = table_required(fs == 44100.0f ? "afCenterFreq44100" : "afCenterFreq48e3", SAF_NBANDS);
        for (int b = 0; b < nBands; b++) freqVector[b] = tab[b];
        return;
    }
    AfSTFT* h = (AfSTFT*)hSTFT;
    if (nBands < h->nBands) SAF_FATAL("afSTFT_getCentreFreqs: freqVector too short");
    const int fftSize = 2 * h->hop;
    if (h->hybrid) {
        static const float w[9] = { 1.0f, 0.7501f, 1.2499f, 0.8751f, 1.1249f, 0.9167f, 1.0833f, 0.9375f, 1.0625f };
        static const int bin[9] = { 0, 1, 1, 2, 2, 3, 3, 4, 4 };
        for (int i = 0; i < 9; i++) freqVector[i] = w[i] * ((float)bin[i] * fs / (float)fftSize);
        for (int i = 9, j = 5; i < h->nBands; i++, j++) freqVector[i] = (float)j * fs / (float)fftSize;
    } else
        for (int k = 0; k <= h->hop; k++) freqVector[k] = (float)k * fs / (float)fftSize;
}

void saf_hip_afSTFT_forward_dev(void* const hSTFT, const float* d_td, long long td_ch_stride, int nHops,
                                float_complex* d_fd, long long fd_band_stride, long long fd_ch_stride)
{
    run_forward((AfSTFT*)hSTFT, d_td, td_ch_stride, nHops, (float2*)d_fd, fd_band_stride, fd_ch_stride);
}

void saf_hip_afSTFT_backward_dev(void* const hSTFT, const float_complex* d_fd, long long fd_band_stride, long long fd_ch_stride,
                                 int nHops, float* d_td, long long td_ch_stride)
{
    run_backward((AfSTFT*)hSTFT, (const float2*)d_fd, fd_band_stride, fd_ch_stride, nHops, d_td, td_ch_stride);
}

/* afSTFT_FIRtoFilterbankCoeffs (afSTFTlib.c:592-674): every IR (and a centred unit impulse) is
 * analysed by a fresh zero-state filterbank; per band the energy ratio gives the gain and the
 * cross-spectrum with the impulse the phase.  All N_dirs*nCH IRs go through ONE analysis launch
 * (they are independent channels of a zero-state bank). */
void afSTFT_FIRtoFilterbankCoeffs(float* hIR, int N_dirs, int nCH, int ir_len, int hopSize, int LDmode, int hybridmode, float_complex* hFB)
{
    ensure_device();
    if (hopSize != 64 && hopSize != 128 && hopSize != 256) SAF_FATAL("afSTFT_FIRtoFilterbankCoeffs: hopSize %d is not supported (64, 128 or 256)", hopSize);
    const int nBands = hopSize + (hybridmode ? 5 : 1);
    const int ir_pad = 1024;
    const int maxlen = (ir_len > hopSize ? ir_len : hopSize) + ir_pad;
    const int nT = (int)((float)maxlen / (float)hopSize + 0.9999f);
    const int L = nT * hopSize;
    /* centre of the FIR delays, from the FIRST direction (afSTFTlib.c:617-634) */
    float idxDel = 0.0f;
    for (int j = 0; j < nCH; j++) {
        float maxVal = 2.23e-13f; int mi = 0;
        for (int i = 0; i < ir_len; i++) if (hIR[j * ir_len + i] > maxVal) { maxVal = hIR[j * ir_len + i]; mi = i; }
        idxDel += (float)mi;
    }
    idxDel /= (float)nCH;
    idxDel = idxDel + 1.5f;
    const int nSig = N_dirs * nCH + 1;                      /* channel 0 = the impulse */
    PinBuf<float> h_td; h_td.ensure((size_t)nSig * L);
    memset(h_td.p, 0, sizeof(float) * (size_t)nSig * L);
    h_td.p[(int)idxDel] = 1.0f;
    for (int nd = 0; nd < N_dirs; nd++)
        for (int c = 0; c < nCH; c++)
            memcpy(h_td.p + (size_t)(1 + nd * nCH + c) * L, hIR + ((size_t)nd * nCH + c) * ir_len, sizeof(float) * ir_len);
    AfState st; st.create(1, nSig, 0, hopSize);
    DevBuf<float> d_td; d_td.alloc((size_t)nSig * L, false);
    DevBuf<float2> d_fd; d_fd.alloc((size_t)nBands * nSig * nT, false);
    HIP_CHECK(hipMemcpyAsync(d_td.p, h_td.p, sizeof(float) * (size_t)nSig * L, hipMemcpyHostToDevice, stream()));
    AnaLaunch a{};
    a.in = d_td.p; a.in_ch = L; a.hopsPerFrame = nT; a.nChIn = nSig;
    a.hist_rd = st.ana[0].p; a.hist_wr = nullptr;
    a.out = d_fd.p; a.out_band = (long long)nSig * nT; a.out_ch = nT;
    a.nCh = nSig; a.nInst = 1; a.H = nT; a.lowDelay = LDmode ? 1 : 0; a.hybrid = hybridmode ? 1 : 0; a.hop = hopSize;
    launch_analysis(a);
    std::vector<float2> fd((size_t)nBands * nSig * nT);
    HIP_CHECK(hipMemcpyAsync(fd.data(), d_fd.p, sizeof(float2) * fd.size(), hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
    for (int b = 0; b < nBands; b++) {
        const float2* imp = &fd[((size_t)b * nSig + 0) * nT];
        float cE = 0.0f;
        for (int t = 0; t < nT; t++) cE += powf(hypotf(imp[t].x, imp[t].y), 2.0f);
        const float denom = cE > 2.23e-8f ? cE : 2.23e-8f;
        for (int nd = 0; nd < N_dirs; nd++)
            for (int c = 0; c < nCH; c++) {
                const float2* ir = &fd[((size_t)b * nSig + 1 + nd * nCH + c) * nT];
                float e = 0.0f, cr = 0.0f, ci = 0.0f;
                for (int t = 0; t < nT; t++) {
                    e += powf(hypotf(ir[t].x, ir[t].y), 2.0f);
                    cr += ir[t].x * imp[t].x + ir[t].y * imp[t].y;
                    ci += ir[t].y * imp[t].x - ir[t].x * imp[t].y;
                }
                const float gain = sqrtf(e / denom);
                const float phase = atan2f(ci, cr);
                hFB[((size_t)b * nCH + c) * N_dirs + nd] = float_complex(cosf(phase) * gain, sinf(phase) * gain);
            }
    }
}

}  // extern "C"
