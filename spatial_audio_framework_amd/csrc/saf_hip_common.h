/*
 * saf_hip_common.h — internal helpers shared by the libsaf_hip translation units.
 * (Not part of the public C-ABI; see include/saf_hip.h for that.)
 */
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>

#define SAF_HOP        128          /* afSTFT hop size of all operators (ambi_dec_internal.h:68) */
#define SAF_NBINS      129
#define SAF_NBANDS     133          /* hybrid bands (ambi_dec_internal.h:69) */
#define SAF_MAXCH      64           /* MAX_NUM_CHANNELS (_common.h:228) */
#define SAF_MAX_ORDER  7            /* MAX_SH_ORDER (_common.h:50) */
#define SAF_ANA_HIST   15           /* input hops of history the analysis kernel needs (9 window + 6 hybrid) */
#define SAF_SYN_HIST   9            /* synthesised frames of history the overlap-add needs */

#define SAF_PI   3.14159265358979323846264338327950288f
#define SAF_PId  3.14159265358979323846264338327950288
#define SAF_SQRT4PI 3.544907701811032f
#define ORDER2NSH(o) (((o) + 1) * ((o) + 1))

/* The product has no CPU fallback: any HIP failure is fatal and loud. */
#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "libsaf_hip: HIP error %s (%d) at %s:%d: %s\n", hipGetErrorString(e_),  \
                    (int)e_, __FILE__, __LINE__, #expr);                                             \
            abort();                                                                                 \
        }                                                                                            \
    } while (0)

#define SAF_FATAL(...)                                                   \
    do {                                                                 \
        fprintf(stderr, "libsaf_hip: " __VA_ARGS__);                    \
        fprintf(stderr, "\n");                                           \
        abort();                                                         \
    } while (0)

namespace saf {

/* ---- runtime (runtime.cpp) ---- */
hipStream_t stream();                 /* stream all library work is enqueued on */
hipStream_t side_stream();            /* second stream: kernels that run beside one on stream() (fork / join with the two calls below) */
void        side_fork();              /* work enqueued on side_stream() after this call starts after what is on stream() now */
void        side_join();              /* work enqueued on stream() after this call starts after what is on side_stream() now */
void        set_stream(hipStream_t);  /* adopt a caller's stream (e.g. torch's current stream) */
hipStream_t new_stream();             /* a non-blocking stream of the library's device */
void        wait_stream(hipStream_t); /* wait for a stream from a host-pointer call: polls, yields the CPU between polls when it takes long */
bool        on_private_stream();      /* this thread is inside a StreamScope */
struct StreamScope {                  /* RAII: stream() of the calling thread returns `s` until the scope ends (s == nullptr: no change) */
    explicit StreamScope(hipStream_t s);
    ~StreamScope();
    hipStream_t prev;
    StreamScope(const StreamScope&) = delete;
    StreamScope& operator=(const StreamScope&) = delete;
};
void        ensure_device();          /* aborts with a clear message when no GPU is usable */

/* ---- host-pointer entry points (runtime.cpp) ---- */
/* One-block host-pointer calls (X_process, saf_matrixConv_apply ...): the kernels read the pinned input block and write
 * the pinned output block directly (hipHostMalloc memory is device-accessible) instead of two DMA copies in the
 * dependent chain: every sample still crosses the link once.  Default on; env SAF_HIP_ZERO_COPY=0 / saf_hip_setZeroCopyIO(0). */
bool zero_copy_io();

/* ---- optional per-kernel timing with HIP events on the library stream (runtime.cpp) ---- */
struct KernelTimer {            /* RAII: brackets one kernel launch when profiling is enabled */
    explicit KernelTimer(const char* name, hipStream_t on = nullptr);     /* on: the stream the kernel is launched on (default: stream()) */
    ~KernelTimer();
    int slot;
};

/* ---- data tables (tables.cpp) ---- */
const float* table(const char* name, int* d0, int* d1);   /* host pointer, NULL if absent */
const float* table_required(const char* name, int d0);

/* ---- constant device tables shared by the afSTFT kernels (afstft_kernels.hip) ---- */
const float*  dev_window(int lowDelay, int synthesis);   /* [1280] */
const float2* dev_twiddles();                            /* [8][64] */

/* simple owning device buffer */
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count, bool zero = true) {
        release();
        n = count;
        if (!count) return;
        HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
        if (zero) HIP_CHECK(hipMemsetAsync(p, 0, count * sizeof(T), stream()));
    }
    void zero() { if (p) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), stream())); }
    void release() { if (p) { HIP_CHECK(hipFree(p)); p = nullptr; n = 0; } }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

/* pinned host staging buffer */
template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    void ensure(size_t count) {
        if (count <= n) return;
        if (p) HIP_CHECK(hipHostFree(p));
        HIP_CHECK(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
        n = count;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
};

/* ---- afSTFT device core (afstft_kernels.hip) ---- */
struct AnaLaunch {
    const float* in;            /* samples: in[inst*in_inst + frame*in_frame + ch*in_ch + (hop%T)*128 + n] */
    long long in_inst, in_ch, in_frame;
    int hopsPerFrame;           /* T */
    int nChIn;                  /* channels physically present in `in`; the rest are zero */
    const float* hist_rd;       /* [inst][nCh][15][128] */
    float* hist_wr;
    float2* out;                /* spectra out[inst*out_inst + band*out_band + ch*out_ch + hop] */
    long long out_inst, out_band, out_ch;
    const float* ch_scale;      /* [nInst][nCh] or null */
    const int* ch_map;          /* [nInst][nCh] input channel gather map or null */
    int tab_stride;             /* instance stride of ch_scale / ch_map (0: nCh) */
    int nCh, nInst, H;          /* H = hops in this launch */
    int lowDelay, hybrid;
    int hop = SAF_HOP;          /* 128: the tuned kernels; 64 / 256: the generic ones (afstft_generic.hip), state sized by the hop */
};
void launch_analysis(const AnaLaunch& a);
void launch_analysis_generic(const AnaLaunch& a);

struct SynLaunch {
    const float2* in;           /* spectra in[inst*in_inst + band*in_band + ch*in_ch + hop] */
    long long in_inst, in_band, in_ch;
    float* out;                 /* samples, addressed like AnaLaunch::in */
    long long out_inst, out_ch, out_frame;
    int hopsPerFrame;
    const float* hist_rd;       /* [inst][nCh][9][256] */
    float* hist_wr;
    int nCh, nInst, H;
    int lowDelay, hybrid;
    int hop = SAF_HOP;
};
void launch_synthesis(const SynLaunch& s);
void launch_synthesis_generic(const SynLaunch& s);

/* ---- per-band dynamic range compression on the spectra (drc_kernels.hip; ambi_drc.c:168-199) ---- */
struct DrcLaunch {
    float2* X; long long x_band, x_ch;        /* spectra [band][ch][hop], compressed in place */
    float* gains; long long g_band;           /* out: gain factor per (band, hop) of this call */
    float* yL_z1;                             /* [133] peak detector state */
    float boost, makeup, threshold, ratio, knee, alpha_a, alpha_r, floor;
    int nCh, H;
};
void launch_drc(const DrcLaunch& l);

/* ---- afSTFT analysis -> real gain per (channel, band) -> afSTFT synthesis in one kernel (eq_kernels.hip) ----
 * z_d[ch] = synthesis( gains[d][ch][band] (.) analysis(x[ch]) ) for d < D (1 or 2) and every channel; hybrid mode, normal
 * delay.  Input addressing, conventions (ch_map / ch_scale) and the input history are those of AnaLaunch; the frame
 * history (9 synthesised frames per channel and output d) has the layout of SynLaunch::hist per d. */
struct EqLaunch {
    const float* in; long long in_inst, in_ch, in_frame; int hopsPerFrame, nChIn;
    const float* hist_rd; float* hist_wr;          /* [inst][nCh][15][128] */
    const float* ch_scale; const int* ch_map;      /* [inst][64] or null */
    const float* gains;                            /* [inst][D][64][136] (bands 133..135 unused) */
    const int* uniform;                            /* [inst][64] 1: the gains of this channel are the same in every band (for every d), or null */
    int D;
    float* z; long long z_d, z_inst, z_ch;         /* z[d*z_d + inst*z_inst + ch*z_ch + hop*128 + n] */
    const float* syn_rd; float* syn_wr; long long syn_d;      /* [d][inst][nCh][9][256] */
    int nCh, nInst, H;
    const int* runFlag = nullptr;                  /* when set: the launch does nothing unless *runFlag != 0 (the re-run behind launch_eq_coop) */
};
/* done != nullptr: every workgroup (one per channel and instance, no time chunks) publishes its z and adds 1 to done[inst] when
 * it has finished — the decode kernel of launch_dec_stream consumes the instances as they complete */
void launch_eq(const EqLaunch& e, unsigned* done = nullptr);

/* Small launches (the one-block host-pointer call): equaliser AND decode in ONE launch — the decode workgroups ride behind the
 * channel workgroups of their instance and wait on done[inst] (eq_kernels.hip, MODE 2).  Returns false when the launch is too
 * large for every workgroup to be resident at once or the shapes do not fit (then: launch_eq + launch_band_gemm). */
struct EqDecodeTail {
    float* Y; long long y_inst, y_frame, y_row;   /* out[inst*y_inst + frame*y_frame + row*y_row + n], n < F */
    const float* Mfrag; long long m_inst;         /* dense decoder(s) of an instance in MFMA fragment order: [D][2][32][64] */
    int nRowsY, F, nFrames, G;                    /* G: 128-column units per decode workgroup */
    int* err;                                     /* [1] host-visible: set when a decode workgroup gave up waiting (the caller then runs the GEMM) */
};
bool launch_eq_decode(const EqLaunch& e, const EqDecodeTail& d, unsigned* done, unsigned target);

/* The decode INSIDE the equaliser launch (eq_kernels.hip MODE 3): order 7 (64 SH channels), 64 loudspeakers, one dense decoder,
 * whole 16-hop sub-chunks.  The 64 channel workgroups of an instance exchange z through EqLaunch::z (write-through stores) with
 * per-(instance, sub-chunk) arrival counters `cnt` ([nInst][nSub], zeroed by the caller before the launch; `target` = arrivals
 * that complete a sub-chunk = 2 waves x 64 workgroups) and each decodes 32 of every sub-chunk's 2048 columns.  Returns false when
 * the shape does not fit.  `err` (host-visible) is set when a wave gave up waiting: the caller's guarded re-run launches
 * (launch_eq with EqLaunch::runFlag, launch_band_gemm with runFlag) then recompute the call. */
struct EqCoop {
    unsigned* cnt; unsigned target; int nSub;
    float* Y; long long y_inst; int y_frame, y_row; int nRowsY, F, T;      /* (block and row strides as ints: scalar registers are short in this kernel) */
    const float* Mfrag; int m_inst;
    int* err;
    int* giveUps;                                  /* device counter of the workgroups that gave up (diagnostics) */
    /* -DEQ_COOP_CHECK builds only: buffer extents in bytes and a record of the first access outside them (the access is skipped) */
    long long ringBytes = 0, cntBytes = 0, yBytes = 0, mBytes = 0; long long* dbg = nullptr;
};
bool launch_eq_coop(const EqLaunch& e, const EqCoop& c);

/* ---- the time-domain decode  out = sum_d M_d z_d  running BESIDE the equaliser kernel (gemm_kernels.hip) ----
 * A persistent grid of register-lean MFMA workgroups on the library's second stream: workgroup p takes the work items
 * p, p + P, ... (item = G column tiles of one instance, instances in launch order), waits until the equaliser workgroups of the
 * item's instance have all published (done[inst] reaches `target`) and multiplies.  The equaliser is bound by vector issue and
 * LDS, the decode by HBM and the matrix cores: side by side they share the compute units instead of taking turns. */
struct DecStreamLaunch {
    const float* z; long long z_d, z_inst, z_ch;  /* as EqLaunch */
    int D, nCh, nInst;
    float* Y; long long y_inst, y_frame, y_row;   /* out[inst*y_inst + frame*y_frame + row*y_row + n], n < F */
    const float* Mfrag; long long m_inst;         /* dense decoder(s) of an instance in MFMA fragment order: [D][2][32][64] */
    int nRowsY, F, nFrames;
    const unsigned* done; unsigned target;
    int* err;                                     /* [2] [0]: a workgroup gave up waiting in this launch (the caller's fix-up GEMM then runs), [1]: total */
};
bool dec_stream_supported(const DecStreamLaunch& l);
void launch_dec_stream(const DecStreamLaunch& l, hipStream_t s);

/* ---- band-batched real GEMM on MFMA (gemm_kernels.hip) ----
 * For every (inst, band): Y[64 x N] = A[mat(inst,band)][64 x 64] * X[64 x N], N = 2*H floats
 * (interleaved re/im of H time slots).  A is stored in MFMA fragment order, see pack_A(). */
struct BandGemmLaunch {
    const float* X; long long x_inst, x_band, x_row;      /* float strides */
    float* Y;       long long y_inst, y_band, y_row;
    const float* Afrag;            /* [nInst][nMat][2][32][64] */
    long long a_inst;              /* float stride between instances' matrix sets */
    const int* band2mat;           /* [nInst][nBands] */
    int nBands, nInst, N;
    int nRowsX = 64;               /* rows physically present in X: higher rows re-read the last one (their matrix columns are zero) */
    /* nTerms = 2:  Y = A_0 X_0 + A_1 X_1  with X_1 = X + x_term and A_1 = the matrix after A_0 (Afrag + 4096 floats) */
    int nTerms = 1; long long x_term = 0;
    int nRowsY = 64;               /* rows of Y that exist: rows beyond are computed (against zero matrix rows) but not stored */
    const int* runFlag = nullptr;  /* when set: the launch does nothing unless *runFlag != 0 (the fix-up behind launch_eq_dec) */
};
void launch_band_gemm(const BandGemmLaunch& g);
void pack_A(const float* A /* [64][64] row-major, zero padded */, float* Afrag /* [2][32][64] */);
/* device-side pack_A for nMat matrices: A [nMat][64][64] row-major -> Afrag [nMat][2][32][64] */
void launch_pack_A(const float* d_A, float* d_Afrag, int nMat);

/* ---- SH encode GEMM of ambi_enc (gemm_kernels.hip; ambi_enc.c:138-171) ----
 * For every (inst, frame f): out = post( fade( Y * p , prevY * p ) ), p = the PREVIOUS frame after its
 * source gains (frame 0 of a call reads the saved state).  One extra block row saves the last frame. */
struct EncLaunch {
    const float* in; long long in_inst, in_frame, in_ch;
    float* out;      long long out_inst, out_frame, out_ch;
    const float* prev_rd; float* prev_wr;     /* [nInst][64][F] */
    const float* Afrag;                       /* [nInst][2 = {Y, prev_Y}][2][32][64] */
    const float* AfragPrev = nullptr;         /* when set: prev_Y of instance i is AfragPrev + i*2*4096 instead of the second half of its Afrag
                                               * entry (operators that keep two matrix slots and swap them instead of copying) */
    const float* gains;                       /* [nInst][64] effective per-source gain */
    const float* postScale;                   /* [nInst] 1/sqrt(nSources) or 1 */
    const float* rowScale;                    /* [nInst][64] N3D -> output norm, by ACN row */
    const int* rowMap;                        /* [nInst][64] ACN row -> output channel */
    const int* nSrc;                          /* [nInst] min(nSources, nInputs present) */
    const int* order;                         /* [nInst] encoding order: rows >= (order+1)^2 of Y are zero */
    const int* mix;                           /* [nInst] 1: frame 0 cross-fades Y with prev_Y; null: nobody mixes */
    int F, nFrames, nInst, nOut;
    int maxSteps;                             /* max over instances of ceil(nSrc / 2) (0: unknown) */
    int rowsIn = 0;                           /* max over instances of the input rows read (0: unknown; used by the overlap check) */
};
void launch_enc_gemm(const EncLaunch& e);

/* ---- FFT-domain matrix convolution (pconv_kernels.hip) ---- */
struct PconvFwd {           /* zero-padded real FFT of size N for every (x, y, z) of the grid */
    const float* src; long long s0, s1, s2; int nValid, yValidStep, yValidTotal;
    float2* dst; long long d0, d1, d2; int ringLen, ringHead;
    const float2* tw; int N; int g0, g1, g2;
};
void pconv_launch_fwd(const PconvFwd& f);
struct PconvApply {         /* MAC over (partition, input) + one inverse FFT per output + overlap-add, for T blocks */
    const float2* Hf; const float2* Xr; float2* P; float* zs; float* out; long long out_ch, out_blk;
    const float2* tw;
    int nIn, nOut, nFB, N, hop, nOB, nBinsP, kSplit, termsPerSplit, xRing, xHead, zRing, zHead, T;
    int diag = 0;           /* 1: saf_multiConv — output o filters input channel o only (nIn = 1 term per partition) */
};
void pconv_launch_apply(const PconvApply& p);
struct TvApply {            /* saf_TVConv_apply for T blocks: three IR selections per block, inverse FFTs, cross-fade */
    const float2* Hf; const float2* Xr; float2* P; float* zs; float* out; long long out_ch, out_blk;
    const float2* tw; const int* irSel;
    int nOut, nFB, N, hop, nBinsP, xRing, xHead, zRing, zHead, T;
};
void tvconv_launch_apply(const TvApply& p);
void pconv_twiddles(int N, DevBuf<float2>& tw);

/* ---- binaural rendering kernels (binaural_kernels.hip) ---- */
struct HrtfInterpLaunch {       /* binauraliser_interpHRTFs (binauraliser_internal.c:46-123) for every flagged source */
    const float* srcDirs;       /* [nSrc][2] azimuth, elevation in degrees (after any head rotation) */
    const int* recalc;          /* [nSrc] */
    const float* gtComp; const int* gtIdx;      /* compressed VBAP table over the HRIR grid: [nTable][3] gains / indices */
    const float2* hrtf_fb;      /* [133][2][N] */
    const float* hrtf_mag;      /* [133][2][N] */
    const float* itds;          /* [N] */
    const float* freq;          /* [133] */
    float2* hrtf_interp;        /* [nSrc][133][2] */
    int nSrc, N, mode, aziRes, elevRes;
    int nInst = 1, srcStride = 0;   /* batches: instance i uses srcDirs / recalc / hrtf_interp entries i*srcStride + src */
};
void launch_hrtf_interp(const HrtfInterpLaunch& l);
struct DvfScaleLaunch {         /* binauraliser_nf.c:299-338: per source and ear the response of the first-order DVF shelf at the band centres
                                 * (evalIIRTransferFunctionf, saf_utility_filters.c:609-671) combined with the interpolated HRTF */
    const float2* hrtf_interp;  /* [nInst * srcStride][133][2] */
    const float* coef;          /* [nInst * srcStride][2 ears][4]: b0, b1, a1, near-field flag (0: far field, HRTF passes unchanged) */
    const float* freq;          /* [133] */
    float2* hrtf_nf;            /* out, layout of hrtf_interp */
    float fs;
    int nSrc, nInst = 1, srcStride = 0;
};
void launch_dvf_scale(const DvfScaleLaunch& l);
struct BinMacLaunch {           /* out[band][ear][t] = scale * sum_src h[src][band][ear] * X[band][src][t]  (binauraliser.c:252-268) */
    const float2* X; long long x_band, x_ch;
    const float2* h;
    float2* Y; long long y_band, y_ch;
    int nSrc, H; float scale;
    int nInst = 1; long long x_inst = 0, y_inst = 0, h_inst = 0;    /* batches: per-instance strides in float2 elements */
};
void launch_binaural_mac(const BinMacLaunch& l);
struct DecRotLaunch {           /* ambi_bin.c:437-456: M_dec_rot[band] = M_dec[band] (2 x nSH, complex) * M_rot (nSH x nSH, real), written in the
                                 * band MAC's operand layout; Mrot == nullptr copies M_dec into that layout */
    const float2* Mdec;         /* [133][2][64] */
    const float* Mrot;          /* [nSH][nSH] row-major, or nullptr */
    float2* out;                /* [64][133][2]; rows >= nSH are written as zero */
    int nSH;
};
void launch_dec_rotate(const DecRotLaunch& l);
struct BinFoldLaunch {          /* HM[inst][sh][band][ear] = sum_ls h[inst][ls][band][ear] * A[inst][band2mat[band]][ls][sh]: decoder and HRTFs as ONE 2 x nSH matrix per band */
    const float2* h;            /* [nInst][64][133][2] */
    const float* A;             /* [nInst][nMat][64][64] row-major [ls][sh] */
    const int* band2mat;        /* [nInst][133] */
    float2* HM;                 /* [nInst][64][133][2] */
    int nInst, nMat, nLS;
};
void launch_binaural_fold(const BinFoldLaunch& l);

/* ---- panner gains (panner_kernels.hip) ---- */
struct PanGainLaunch {          /* per moved source: table row -> per-band p-norm gains -> column `src` of A[band][ls][src] (panner.c:230-262) */
    const float* gtable;        /* [nTable][nLS] */
    const int* row;             /* [nSrc] table row of each source */
    const int* recalc;          /* [nSrc] */
    const float* pValue;        /* [133] */
    float* A;                   /* [133][64][64] row = loudspeaker, column = source */
    int nSrc, nLS;
};
void launch_panner_gains(const PanGainLaunch& l);

/* ---- powermap kernels (powermap_kernels.hip) ---- */
struct CovLaunch {              /* Cx[b] <- a Cx[b] + (1-a) X_f X_f^H for consecutive frames f (powermap.c:258-267), for nInst instances */
    const float2* X; long long x_band, x_ch;     /* spectra [band][ch][hop] */
    float2* Cx;                 /* [133][64][64] */
    int nSH, T, nFrames; float alpha;
    int nInst = 1; long long x_inst = 0, cx_inst = 0;      /* batches: per-instance strides in float2 elements */
    const float* alphaInst = nullptr;                       /* [nInst] per-instance averaging coefficient (else `alpha` for all) */
};
void launch_cov_update(const CovLaunch& l);
struct PwdLaunch {              /* grouped covariance + PWD map + temporal smoothing (powermap.c:276-347, saf_sh.c:1544-1584) */
    const float2* Cx; const float* bandScale /* [133] 1e3*EQ */; const int* bandNSH /* [133] */;
    float* Cg;                  /* scratch [64][64] (real part) */
    const float* Ygrid;         /* [nM][G] scaled 1/nM */
    float* pmap; float* prev_pmap;
    int nM, G; float avg;
    int nBands = SAF_NBANDS;
    /* batches: instance i uses Cx + i*cx_inst, bandScale / bandNSH + i*133, Cg + i*4096, pmap / prev_pmap + i*G, and the grid table of
     * its own order mapOrder[i] (1..7; 0: no map asked for this instance, its workgroups leave); avgInst[i] replaces avg */
    int nInst = 1; long long cx_inst = 0;
    const int* mapOrder = nullptr; const float* avgInst = nullptr;
    const float* YgridByOrder[SAF_MAX_ORDER] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
};
void launch_pwd_map(const PwdLaunch& l);
struct AdaptMapLaunch {         /* MVDR / CroPaC-LCMV / MUSIC / MinNorm maps (powermap.c:294-341, saf_sh.c:1586-1858) */
    const float2* Cx; const float* bandScale; const int* bandNSH;
    float2* Cg;                 /* scratch [64][64] grouped complex covariance */
    double2* Lchol;             /* scratch [64][64] Cholesky factor */
    float2* Veig;               /* scratch [64][64] eigenvectors in columns, descending eigenvalues */
    float* eig;                 /* scratch [64] */
    float2* Un;                 /* scratch [64] min-norm vector */
    int* status;                /* scratch [1] */
    const float* Ygrid;         /* [nM][G] scaled 1/nM */
    float* pmap; float* prev_pmap;
    int nM, G, mode, nSources; float avg, regPar, lambda;
    int nBands = SAF_NBANDS;    /* covariance matrices to group (1 for the stand-alone generate*map entry points) */
    float2* Wout = nullptr;     /* optional [nM][G] MVDR weights (generateMVDRmap's w_MVDR_out) */
};
void launch_adaptive_map(const AdaptMapLaunch& l);
void launch_subspace_map(const AdaptMapLaunch& l);      /* only the per-direction projection (eigenvectors supplied in Veig) */

}  // namespace saf
