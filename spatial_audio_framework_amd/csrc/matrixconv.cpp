/*
 * matrixconv.cpp — saf_matrixConv_*, saf_multiConv_* and saf_TVConv_*
 * (framework/modules/saf_utilities/saf_utility_matrixConv.h:55-200, .c:37-620) on the GPU.
 *
 * Both reference modes are linear convolution with zero latency
 *     y_o[n] = sum_i sum_k H[o][i][k] x_i[n-k]
 * computed block-wise in the frequency domain.  Partitioned mode splits the filters into
 * nFB = ceil(len/hop) partitions and keeps a delay line of input spectra; non-partitioned mode
 * is the same machinery with one partition and a longer transform.  State per handle (device):
 * ring of input spectra, ring of the last inverse transforms (the overlap-add buffer).
 *
 * saf_multiConv (one filter per channel, y_c = h_c * x_c) is the same object with a diagonal filter matrix: one term
 * per partition, the spectral MAC reads input channel o for output o.
 * saf_TVConv (one input, nCHout outputs, nIRs switchable filter sets) convolves every block with the IR sets selected
 * for this block and for the two previous ones and cross-fades linearly from the IR of block t-2 to the IR of block
 * t-1 (.c:572-611): three spectral products + inverse transforms per output and block, one mix kernel.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"

namespace saf {

struct MatrixConv {
    int hop, len, nIn, nOut, part, diag = 0;
    int nFB, N, nBinsP, nOB, kSplit, termsPerSplit, maxBlocks;
    int xRing, zRing;
    long long blk = 0;              /* absolute index of the next block */
    DevBuf<float2> tw, Hf, Xr, P;
    DevBuf<float> zs, d_in, d_out;
    PinBuf<float> h_in, h_out;

    void size_scratch(int T)
    {
        if (T <= maxBlocks) return;
        /* growing the rings would lose their phase: rings are sized once for maxBlocks at creation */
        SAF_FATAL("matrixConv: %d blocks per call exceed the %d the handle was created for", T, maxBlocks);
    }

    /* diag_: H is [nCH][len] and channel c is filtered by H[c] only (saf_multiConv); nCHin must then be 1 */
    void create(int hopSize, const float* H, int length_h, int nCHin, int nCHout, int usePart, int maxBlocks_, int diag_ = 0)
    {
        diag = diag_;
        hop = hopSize; len = length_h; nIn = nCHin; nOut = nCHout; part = usePart ? 1 : 0; maxBlocks = maxBlocks_;
        if (hop < 1 || len < 1 || nIn < 1 || nOut < 1) SAF_FATAL("matrixConv: bad dimensions");
        int partLen;
        if (part) { nFB = (len + hop - 1) / hop; partLen = hop; }                  /* saf_utility_matrixConv.c:107 */
        else      { nFB = 1; partLen = len; }
        const int Lz = hop + partLen - 1;                                            /* length of one block's linear convolution */
        nOB = (Lz + hop - 1) / hop;                                                  /* = numOvrlpAddBlocks (:71) / 2 in partitioned mode */
        N = 16; while (N < Lz || N < 2 * hop) N <<= 1;
        if (N > 16384) SAF_FATAL("matrixConv: transform size %d exceeds the supported 16384 (use the partitioned mode)", N);
        nBinsP = N / 2;                                                              /* bin 0 carries (Re X[0], Re X[N/2]) */
        xRing = nFB - 1 + maxBlocks; zRing = nOB - 1 + maxBlocks;
        const int nTerms = nFB * nIn;
        /* enough workgroups to pull the filter spectra at HBM rate: split the (partition, input) sum */
        /* upper bound of the split of the (partition, input) sum; the launcher picks the split per call (pconv_launch_apply) */
        const int tiles = ((N / 2 + 15) / 16) * ((nOut + 1) / 2);
        kSplit = (1024 + tiles - 1) / tiles; if (kSplit > (nTerms + 15) / 16) kSplit = (nTerms + 15) / 16; if (kSplit < 1) kSplit = 1;
        termsPerSplit = (nTerms + kSplit - 1) / kSplit;
        pconv_twiddles(N, tw);
        Hf.alloc((size_t)nOut * nTerms * nBinsP);
        Xr.alloc((size_t)xRing * (diag ? nOut : nIn) * nBinsP);
        P.alloc((size_t)maxBlocks * nOut * kSplit * nBinsP);
        zs.alloc((size_t)zRing * nOut * N);
        /* filter spectra: partition p of H[o][i] = taps [p*partLen, (p+1)*partLen) zero-padded to N (:116-125) */
        DevBuf<float> dH; dH.alloc((size_t)nOut * nIn * len, false);
        HIP_CHECK(hipMemcpyAsync(dH.p, H, sizeof(float) * (size_t)nOut * nIn * len, hipMemcpyHostToDevice, stream()));
        PconvFwd f{};
        f.src = dH.p; f.s0 = len; f.s1 = partLen; f.s2 = (long long)nIn * len;
        f.nValid = partLen; f.yValidStep = partLen; f.yValidTotal = len;
        f.dst = Hf.p; f.d0 = nBinsP; f.d1 = (long long)nIn * nBinsP; f.d2 = (long long)nTerms * nBinsP;
        f.ringLen = 0; f.ringHead = 0; f.tw = tw.p; f.N = N; f.g0 = nIn; f.g1 = nFB; f.g2 = nOut;
        pconv_launch_fwd(f);
        HIP_CHECK(hipStreamSynchronize(stream()));      /* dH is released on return */
    }

    void apply_dev(const float* in, long long in_ch, long long in_blk, float* out, long long out_ch, long long out_blk, int T)
    {
        if (T <= 0) return;
        size_scratch(T);
        const int xHead = (int)(blk % xRing), zHead = (int)(blk % zRing);
        PconvFwd f{};
        f.src = in; f.s0 = in_ch; f.s1 = in_blk; f.s2 = 0; f.nValid = hop; f.yValidStep = 0; f.yValidTotal = 0;
        const int nX = diag ? nOut : nIn;                 /* channels of the input */
        f.dst = Xr.p; f.d0 = nBinsP; f.d1 = (long long)nX * nBinsP; f.d2 = 0; f.ringLen = xRing; f.ringHead = xHead;
        f.tw = tw.p; f.N = N; f.g0 = nX; f.g1 = T; f.g2 = 1;
        pconv_launch_fwd(f);
        PconvApply a{};
        a.Hf = Hf.p; a.Xr = Xr.p; a.P = P.p; a.zs = zs.p; a.out = out; a.out_ch = out_ch; a.out_blk = out_blk; a.tw = tw.p;
        a.nIn = nIn; a.nOut = nOut; a.nFB = nFB; a.N = N; a.hop = hop; a.nOB = nOB; a.nBinsP = nBinsP; a.kSplit = kSplit; a.termsPerSplit = termsPerSplit;
        a.xRing = xRing; a.xHead = xHead; a.zRing = zRing; a.zHead = zHead; a.T = T; a.diag = diag;
        pconv_launch_apply(a);
        blk += T;
    }
};

struct TVConv {
    int hop, len, nIRs, nOut, nFB, N, nBinsP, maxBlocks, xRing, zRing;
    int posIdx_last, posIdx_last2;
    long long blk = 0;
    DevBuf<float2> tw, Hf, Xr, P;
    DevBuf<float> zs, d_in, d_out;
    DevBuf<int> irSel;
    PinBuf<int> h_sel;
    PinBuf<float> h_in, h_out;

    void create(int hopSize, float** H, int length_h, int nIRs_, int nCHout, int initIdx, int maxBlocks_)
    {
        hop = hopSize; len = length_h; nIRs = nIRs_; nOut = nCHout; maxBlocks = maxBlocks_;
        if (hop < 2 || len < 1 || nIRs < 1 || nOut < 1) SAF_FATAL("TVConv: bad dimensions");
        posIdx_last = posIdx_last2 = initIdx < nIRs ? initIdx : 0;                  /* saf_utility_matrixConv.c:456-462 */
        nFB = (len + hop - 1) / hop;                                                 /* :468 */
        N = 16; while (N < 2 * hop) N <<= 1;
        if (N > 16384) SAF_FATAL("TVConv: transform size %d exceeds the supported 16384", N);
        nBinsP = N / 2;
        xRing = nFB - 1 + maxBlocks; zRing = 1 + maxBlocks;
        pconv_twiddles(N, tw);
        Hf.alloc((size_t)nIRs * nOut * nFB * nBinsP);
        Xr.alloc((size_t)xRing * nBinsP);
        P.alloc((size_t)maxBlocks * nOut * 3 * nBinsP);
        zs.alloc((size_t)zRing * nOut * 3 * N);
        irSel.alloc((size_t)maxBlocks * 3); h_sel.ensure((size_t)maxBlocks * 3);
        /* partition p of H[ir][o] = taps [p*hop, (p+1)*hop) zero-padded to N (:493-503) */
        DevBuf<float> dH; dH.alloc((size_t)nIRs * nOut * len, false);
        for (int ir = 0; ir < nIRs; ir++)
            HIP_CHECK(hipMemcpyAsync(dH.p + (size_t)ir * nOut * len, H[ir], sizeof(float) * (size_t)nOut * len, hipMemcpyHostToDevice, stream()));
        PconvFwd f{};
        f.src = dH.p; f.s0 = 0; f.s1 = hop; f.s2 = len;
        f.nValid = hop; f.yValidStep = hop; f.yValidTotal = len;
        f.dst = Hf.p; f.d0 = 0; f.d1 = nBinsP; f.d2 = (long long)nFB * nBinsP;
        f.ringLen = 0; f.ringHead = 0; f.tw = tw.p; f.N = N; f.g0 = 1; f.g1 = nFB;
        for (int z0 = 0; z0 < nIRs * nOut; z0 += 32768) {        /* grid.z limit */
            f.g2 = nIRs * nOut - z0 < 32768 ? nIRs * nOut - z0 : 32768;
            f.src = dH.p + (size_t)z0 * len; f.dst = Hf.p + (size_t)z0 * nFB * nBinsP;
            pconv_launch_fwd(f);
        }
        HIP_CHECK(hipStreamSynchronize(stream()));
    }

    void apply_dev(const float* in, long long in_blk, float* out, long long out_ch, long long out_blk, const int* irIdx, int T)
    {
        if (T <= 0) return;
        if (T > maxBlocks) SAF_FATAL("TVConv: %d blocks per call exceed the %d the handle was created for", T, maxBlocks);
        HIP_CHECK(hipStreamSynchronize(stream()));             /* the selection staging buffer may still be in flight */
        for (int t = 0; t < T; t++) {
            int ir = irIdx[t];
            if (ir < 0 || ir >= nIRs) SAF_FATAL("TVConv: IR index %d outside 0..%d", ir, nIRs - 1);
            h_sel.p[t * 3] = ir; h_sel.p[t * 3 + 1] = posIdx_last; h_sel.p[t * 3 + 2] = posIdx_last2;
            posIdx_last2 = posIdx_last; posIdx_last = ir;                           /* :618-619 */
        }
        HIP_CHECK(hipMemcpyAsync(irSel.p, h_sel.p, sizeof(int) * (size_t)T * 3, hipMemcpyHostToDevice, stream()));
        const int xHead = (int)(blk % xRing), zHead = (int)(blk % zRing);
        PconvFwd f{};
        f.src = in; f.s0 = 0; f.s1 = in_blk; f.s2 = 0; f.nValid = hop; f.yValidStep = 0; f.yValidTotal = 0;
        f.dst = Xr.p; f.d0 = 0; f.d1 = nBinsP; f.d2 = 0; f.ringLen = xRing; f.ringHead = xHead;
        f.tw = tw.p; f.N = N; f.g0 = 1; f.g1 = T; f.g2 = 1;
        pconv_launch_fwd(f);
        TvApply a{};
        a.Hf = Hf.p; a.Xr = Xr.p; a.P = P.p; a.zs = zs.p; a.out = out; a.out_ch = out_ch; a.out_blk = out_blk; a.tw = tw.p; a.irSel = irSel.p;
        a.nOut = nOut; a.nFB = nFB; a.N = N; a.hop = hop; a.nBinsP = nBinsP; a.xRing = xRing; a.xHead = xHead; a.zRing = zRing; a.zHead = zHead; a.T = T;
        tvconv_launch_apply(a);
        blk += T;
    }
};

}  // namespace saf

using namespace saf;

static int g_matrixconv_max_blocks = 1;

extern "C" {

void saf_hip_matrixConv_setMaxBlocksPerCall(int n) { g_matrixconv_max_blocks = n < 1 ? 1 : n; }

void saf_matrixConv_create(void** const phMC, int hopSize, float* H, int length_h, int nCHin, int nCHout, int usePartFLAG)
{
    ensure_device();
    MatrixConv* h = new MatrixConv();
    h->create(hopSize, H, length_h, nCHin, nCHout, usePartFLAG, g_matrixconv_max_blocks);
    *phMC = h;
}

void saf_matrixConv_destroy(void** const phMC)
{
    MatrixConv* h = (MatrixConv*)*phMC;
    if (!h) return;
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete h;
    *phMC = nullptr;
}

void saf_matrixConv_apply(void* const hMC, float* inputSig, float* outputSig)
{
    MatrixConv* h = (MatrixConv*)hMC;
    const size_t nin = (size_t)h->nIn * h->hop, nout = (size_t)h->nOut * h->hop;
    h->h_in.ensure(nin); h->h_out.ensure(nout);
    if (!h->d_in.p) { h->d_in.alloc(nin, false); h->d_out.alloc(nout, false); }
    memcpy(h->h_in.p, inputSig, sizeof(float) * nin);
    if (zero_copy_io()) h->apply_dev(h->h_in.p, h->hop, 0, h->h_out.p, h->hop, 0, 1);                    /* kernels on the pinned blocks */
    else {
        HIP_CHECK(hipMemcpyAsync(h->d_in.p, h->h_in.p, sizeof(float) * nin, hipMemcpyHostToDevice, stream()));
        h->apply_dev(h->d_in.p, h->hop, 0, h->d_out.p, h->hop, 0, 1);
        HIP_CHECK(hipMemcpyAsync(h->h_out.p, h->d_out.p, sizeof(float) * nout, hipMemcpyDeviceToHost, stream()));
    }
    HIP_CHECK(hipStreamSynchronize(stream()));
    memcpy(outputSig, h->h_out.p, sizeof(float) * nout);
}

void saf_hip_matrixConv_apply_dev(void* const hMC, const float* d_in, long long in_ch_stride, long long in_block_stride,
                                  float* d_out, long long out_ch_stride, long long out_block_stride, int nBlocks)
{
    ((MatrixConv*)hMC)->apply_dev(d_in, in_ch_stride, in_block_stride, d_out, out_ch_stride, out_block_stride, nBlocks);
}

/* ---------------- saf_multiConv (saf_utility_matrixConv.h:109-137, .c:257-416) ---------------- */
void saf_multiConv_create(void** const phMC, int hopSize, float* H, int length_h, int nCH, int usePartFLAG)
{
    ensure_device();
    MatrixConv* h = new MatrixConv();
    h->create(hopSize, H, length_h, 1, nCH, usePartFLAG, g_matrixconv_max_blocks, 1);
    *phMC = h;
}
void saf_multiConv_destroy(void** const phMC) { saf_matrixConv_destroy(phMC); }
void saf_multiConv_apply(void* const hMC, float* inputSig, float* outputSig)
{
    MatrixConv* h = (MatrixConv*)hMC;
    const size_t n = (size_t)h->nOut * h->hop;
    h->h_in.ensure(n); h->h_out.ensure(n);
    if (!h->d_in.p) { h->d_in.alloc(n, false); h->d_out.alloc(n, false); }
    memcpy(h->h_in.p, inputSig, sizeof(float) * n);
    if (zero_copy_io()) h->apply_dev(h->h_in.p, h->hop, 0, h->h_out.p, h->hop, 0, 1);
    else {
        HIP_CHECK(hipMemcpyAsync(h->d_in.p, h->h_in.p, sizeof(float) * n, hipMemcpyHostToDevice, stream()));
        h->apply_dev(h->d_in.p, h->hop, 0, h->d_out.p, h->hop, 0, 1);
        HIP_CHECK(hipMemcpyAsync(h->h_out.p, h->d_out.p, sizeof(float) * n, hipMemcpyDeviceToHost, stream()));
    }
    HIP_CHECK(hipStreamSynchronize(stream()));
    memcpy(outputSig, h->h_out.p, sizeof(float) * n);
}
void saf_hip_multiConv_apply_dev(void* const hMC, const float* d_in, long long in_ch_stride, long long in_block_stride,
                                 float* d_out, long long out_ch_stride, long long out_block_stride, int nBlocks)
{
    ((MatrixConv*)hMC)->apply_dev(d_in, in_ch_stride, in_block_stride, d_out, out_ch_stride, out_block_stride, nBlocks);
}

/* ---------------- saf_TVConv (saf_utility_matrixConv.h:157-200, .c:438-620) ---------------- */
void saf_TVConv_create(void** const phTVC, int hopSize, float** H, int length_h, int nIRs, int nCHout, int initIdx)
{
    ensure_device();
    TVConv* h = new TVConv();
    h->create(hopSize, H, length_h, nIRs, nCHout, initIdx, g_matrixconv_max_blocks);
    *phTVC = h;
}
void saf_TVConv_destroy(void** const phTVC)
{
    TVConv* h = (TVConv*)*phTVC;
    if (!h) return;
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete h;
    *phTVC = nullptr;
}
void saf_TVConv_apply(void* const hTVC, float* inputSig, float* outputSig, int irIdx)
{
    TVConv* h = (TVConv*)hTVC;
    const size_t nin = h->hop, nout = (size_t)h->nOut * h->hop;
    h->h_in.ensure(nin); h->h_out.ensure(nout);
    if (!h->d_in.p) { h->d_in.alloc(nin, false); h->d_out.alloc(nout, false); }
    memcpy(h->h_in.p, inputSig, sizeof(float) * nin);
    if (zero_copy_io()) h->apply_dev(h->h_in.p, 0, h->h_out.p, h->hop, 0, &irIdx, 1);
    else {
        HIP_CHECK(hipMemcpyAsync(h->d_in.p, h->h_in.p, sizeof(float) * nin, hipMemcpyHostToDevice, stream()));
        h->apply_dev(h->d_in.p, 0, h->d_out.p, h->hop, 0, &irIdx, 1);
        HIP_CHECK(hipMemcpyAsync(h->h_out.p, h->d_out.p, sizeof(float) * nout, hipMemcpyDeviceToHost, stream()));
    }
    HIP_CHECK(hipStreamSynchronize(stream()));
    memcpy(outputSig, h->h_out.p, sizeof(float) * nout);
}
void saf_hip_TVConv_apply_dev(void* const hTVC, const float* d_in, long long in_block_stride,
                              float* d_out, long long out_ch_stride, long long out_block_stride, const int* irIdx, int nBlocks)
{
    ((TVConv*)hTVC)->apply_dev(d_in, in_block_stride, d_out, out_ch_stride, out_block_stride, irIdx, nBlocks);
}

}
