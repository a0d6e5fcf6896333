/*
 * beamformer.cpp — the beamformer operator (examples/include/beamformer.h:50-190, examples/src/beamformer/beamformer.c):
 * static axisymmetric beams (cardioid / hyper-cardioid / max-EV) steered over an Ambisonic scene, with the block path on
 * the GPU:
 *
 *   previous block (N3D/ACN) -> [MFMA GEMM with the beam weights (and the previous weights + linear cross-fade when a
 *   beam moved)] -> one output per beam                                                       beamformer.c:118-175
 *
 * Same shape as ambi_enc's encode step ([nBeams x nSH] instead of [nSH x nSources]): it runs on the enc_gemm kernels,
 * the SN3D / FuMa -> N3D scaling as the per-row input gain.  The weights are built on the host:
 *   beamWeightsCardioid2Spherical / beamWeightsHypercardioid2Spherical (saf_sh.c:716-745), beamWeightsMaxEV (:747),
 *   rotateAxisCoeffsReal (saf_sh.c:839-882 via getSHcomplex :333-382 and complex2realCoeffs :384-475).
 * The reference holds no test for this operator ("parity unpinned"): tests/ compares with a literal CPU restatement and
 * with the closed-form beam patterns.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"

namespace saf {

static int g_beamformer_frame_size = 128;       /* default of the reference (beamformer_internal.h:46) */

static double fact(int n) { double f = 1.0; for (int i = 2; i <= n; i++) f *= (double)i; return f; }

/* associated Legendre functions P_n^m(x), m = 0..n, with the Condon-Shortley phase (what unnorm_legendreP returns) */
static void legendre_cs(int n, double x, double* P)
{
    /* P_m^m = (-1)^m (2m-1)!! (1-x^2)^(m/2);  P_{m+1}^m = x (2m+1) P_m^m;  (l-m) P_l^m = x (2l-1) P_{l-1}^m - (l+m-1) P_{l-2}^m */
    const double s = sqrt(1.0 - x * x > 0.0 ? 1.0 - x * x : 0.0);
    for (int m = 0; m <= n; m++) {
        double pmm = 1.0;
        for (int i = 1; i <= m; i++) pmm *= -(2.0 * i - 1.0) * s;
        if (m == n) { P[m] = pmm; continue; }
        double p1 = x * (2.0 * m + 1.0) * pmm, p0 = pmm;
        for (int l = m + 2; l <= n; l++) { const double p2 = (x * (2.0 * l - 1.0) * p1 - (l + m - 1.0) * p0) / (double)(l - m); p0 = p1; p1 = p2; }
        P[m] = p1;
    }
}

/* rotateAxisCoeffsReal (saf_sh.c:839-882): c_nm = Re( conj(T_c2r) * [ conj(Y_nm^complex(theta0, phi0)) sqrt(4 pi/(2n+1)) c_n ] ),
 * written out per (n, m) from the entries of complex2realSHMtx (saf_sh.c:384-414) */
void rotate_axis_coeffs_real(int order, const float* c_n, float theta_0, float phi_0, float* c_nm)
{
    const double ct = cos((double)theta_0);
    double P[SAF_MAX_ORDER + 2];
    for (int n = 0, q0 = 0; n <= order; q0 += 2 * n + 1, n++) {
        legendre_cs(n, ct, P);
        const float sc = sqrtf(4.0f * SAF_PI / (2.0f * (float)n + 1.0f)) * c_n[n];
        /* complex coefficients C[n, m] = conj(Y_nm) * sc, m = -n..n */
        float cre[2 * SAF_MAX_ORDER + 1], cim[2 * SAF_MAX_ORDER + 1];
        for (int m = 0; m <= n; m++) {
            const double norm = sqrt((2.0 * n + 1.0) * fact(n - m) / (4.0 * SAF_PId * fact(n + m)));
            const double yr = cos((double)m * (double)phi_0) * norm * P[m], yi = sin((double)m * (double)phi_0) * norm * P[m];   /* Y_n^m, m >= 0 */
            const float yrf = (float)yr, yif = (float)yi;
            cre[n + m] = yrf * sc; cim[n + m] = -yif * sc;                               /* conj(Y_n^m) */
            if (m > 0) {
                const double sg = (m & 1) ? -1.0 : 1.0;                                   /* Y_n^{-m} = (-1)^m conj(Y_n^m) */
                const float yr2 = (float)(sg * yr), yi2 = (float)(-sg * yi);
                cre[n - m] = yr2 * sc; cim[n - m] = -yi2 * sc;
            }
        }
        const float r2 = 1.0f / sqrtf(2.0f);
        for (int m = -n; m <= n; m++) {
            float v;
            if (m == 0) v = cre[n];
            else if (m < 0) {
                const int am = -m;
                const float sg = (am & 1) ? -1.0f : 1.0f;
                /* -i/sqrt2 C[n,-|m|] + i (-1)^|m| / sqrt2 C[n,+|m|]: real part */
                v = r2 * cim[n - am] - sg * r2 * cim[n + am];
            } else {
                const float sg = (m & 1) ? -1.0f : 1.0f;
                v = sg * r2 * cre[n + m] + r2 * cre[n - m];
            }
            c_nm[q0 + n + m] = v;
        }
    }
}

struct Beamformer {
    int F, fs = 48000;
    int beamOrder, nBeams, beamType;
    float beam_dirs_deg[SAF_MAXCH][2];
    CH_ORDER chOrdering; NORM_TYPES norm;
    int recalc[SAF_MAXCH];
    float W[64 * 64], prevW[64 * 64];           /* [beam][SH] row-major, zero padded */
    bool ready = false, clearState = true;
    int par = 0;
    DevBuf<float> Afrag, prev[2], fpar, d_in, d_out;
    DevBuf<int> ipar;
    PinBuf<float> hf, hA, h_in, h_out;
    PinBuf<int> hi;
    float shadowGain[64];
    int shadowI[3] = { -1, -1, -1 };
    int cur = 0;                                /* Afrag slot holding W; the other one holds prevW (swapped, not copied) */
    bool stagingBusy = false;                   /* hA may still be read by a copy enqueued by a device-entry call */
};

static void bf_setup(Beamformer* p)
{
    if (p->ready) return;
    ensure_device();
    const int F = p->F;
    p->Afrag.alloc(2 * 4096);
    p->prev[0].alloc((size_t)SAF_MAXCH * F); p->prev[1].alloc((size_t)SAF_MAXCH * F);
    p->fpar.alloc(1 + 64 + 64); p->ipar.alloc(3 + 64);
    p->hf.ensure(1 + 64 + 64); p->hi.ensure(3 + 64); p->hA.ensure(2 * 4096);
    p->hf.p[0] = 1.0f;
    for (int i = 0; i < 64; i++) { p->hf.p[1 + i] = 1.0f; p->hf.p[65 + i] = 1.0f; p->hi.p[3 + i] = i; p->shadowGain[i] = 1.0f; }
    HIP_CHECK(hipMemcpyAsync(p->fpar.p, p->hf.p, sizeof(float) * 129, hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(p->ipar.p + 3, p->hi.p + 3, sizeof(int) * 64, hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
    p->ready = true;
}

/* beamformer.c:118-175 for nFrames consecutive blocks at device-accessible addresses (ACN channel order) */
static void bf_run(Beamformer* p, const float* in, long long in_frame, long long in_ch, int nIn, float* out, long long out_frame, long long out_ch, int nOut, int nFrames)
{
    bf_setup(p);
    const int F = p->F, order = p->beamOrder, nSH = ORDER2NSH(order), nBeams = p->nBeams;
    if (p->clearState) {                                     /* beamformer_init (beamformer.c:82-84) */
        HIP_CHECK(hipMemsetAsync(p->prev[p->par].p, 0, sizeof(float) * (size_t)SAF_MAXCH * F, stream()));
        HIP_CHECK(hipMemsetAsync(p->Afrag.p, 0, sizeof(float) * 2 * 4096, stream()));      /* beamWeights and prev_beamWeights are zeroed with the state */
        p->clearState = false;
    }
    /* input normalisation -> N3D as the per-row input gain (saf_hoa.c:72-116) */
    float gain[64];
    for (int i = 0; i < 64; i++) gain[i] = 1.0f;
    if (p->norm == NORM_SN3D) { for (int n = 0; n <= order; n++) for (int ch = n * n; ch < ORDER2NSH(n); ch++) gain[ch] = sqrtf(2.0f * (float)n + 1.0f); }
    else if (p->norm == NORM_FUMA) { gain[0] = sqrtf(2.0f); for (int ch = 1; ch < 4; ch++) gain[ch] = sqrtf(3.0f); }
    if (memcmp(gain, p->shadowGain, sizeof(gain)) != 0) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        memcpy(p->hf.p + 1, gain, sizeof(gain));
        HIP_CHECK(hipMemcpyAsync(p->fpar.p + 1, p->hf.p + 1, sizeof(gain), hipMemcpyHostToDevice, stream()));
        memcpy(p->shadowGain, gain, sizeof(gain));
    }
    int mix = 0;
    for (int bi = 0; bi < nBeams; bi++) {
        if (!p->recalc[bi]) continue;
        float c_n[SAF_MAX_ORDER + 1], w[64];
        switch (p->beamType) {
            case STATIC_BEAM_TYPE_CARDIOID: beamWeightsCardioid2Spherical(order, c_n); break;
            case STATIC_BEAM_TYPE_HYPERCARDIOID: beamWeightsHypercardioid2Spherical(order, c_n); break;
            default: beamWeightsMaxEV(order, c_n); break;
        }
        rotate_axis_coeffs_real(order, c_n, SAF_PI / 2.0f - p->beam_dirs_deg[bi][1] * SAF_PI / 180.0f, p->beam_dirs_deg[bi][0] * SAF_PI / 180.0f, w);
        memset(p->W + bi * 64, 0, sizeof(float) * 64);
        memcpy(p->W + bi * 64, w, sizeof(float) * nSH);
        p->recalc[bi] = 0;
        mix = 1;
    }
    if (mix) {
        if (p->stagingBusy) { HIP_CHECK(hipStreamSynchronize(stream())); p->stagingBusy = false; }
        /* rows beyond nBeams stay as they are in the reference's gemm (not computed); here they are not stored (nOut) */
        p->cur ^= 1;
        pack_A(p->W, p->hA.p);
        HIP_CHECK(hipMemcpyAsync(p->Afrag.p + p->cur * 4096, p->hA.p, sizeof(float) * 4096, hipMemcpyHostToDevice, stream()));
        p->stagingBusy = true;
    }
    const int nSrc = nSH < nIn ? nSH : nIn;
    /* the kernels bound the output rows by (order+1)^2: give them the order that covers the beams */
    int rowOrder = 0; while (ORDER2NSH(rowOrder) < nBeams) rowOrder++;
    if (p->shadowI[0] != nSrc || p->shadowI[1] != mix || p->shadowI[2] != rowOrder) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->hi.p[0] = nSrc; p->hi.p[1] = mix; p->hi.p[2] = rowOrder;
        HIP_CHECK(hipMemcpyAsync(p->ipar.p, p->hi.p, sizeof(int) * 3, hipMemcpyHostToDevice, stream()));
        p->shadowI[0] = nSrc; p->shadowI[1] = mix; p->shadowI[2] = rowOrder;
    }
    EncLaunch e{};
    e.in = in; e.in_inst = 0; e.in_frame = in_frame; e.in_ch = in_ch;
    e.out = out; e.out_inst = 0; e.out_frame = out_frame; e.out_ch = out_ch;
    e.prev_rd = p->prev[p->par].p; e.prev_wr = p->prev[p->par ^ 1].p;
    e.Afrag = p->Afrag.p + p->cur * 4096; e.AfragPrev = p->Afrag.p + (p->cur ^ 1) * 4096; e.postScale = p->fpar.p; e.gains = p->fpar.p + 1; e.rowScale = p->fpar.p + 65;
    e.nSrc = p->ipar.p; e.mix = mix ? p->ipar.p + 1 : nullptr; e.order = p->ipar.p + 2; e.rowMap = p->ipar.p + 3;
    e.F = F; e.nFrames = nFrames; e.nInst = 1; e.nOut = nOut < nBeams ? nOut : nBeams;
    e.maxSteps = (nSrc + 1) / 2; e.rowsIn = nSrc;
    launch_enc_gemm(e);
    p->par ^= 1;
    if (mix) memcpy(p->prevW, p->W, sizeof(p->W));            /* prev_beamWeights <- beamWeights (beamformer.c:171): on the device the slots swap at the next change */
}

}  // namespace saf

using namespace saf;

extern "C" {

/* ---------------- beam weights (saf_sh.h; saf_sh.c:716-745, 839-857) ---------------- */
void beamWeightsCardioid2Spherical(int N, float* b_n)
{
    for (int n = 0; n < N + 1; n++)
        b_n[n] = sqrtf(4.0f * SAF_PI * (2.0f * (float)n + 1.0f)) * (float)fact(N) * (float)fact(N + 1) / ((float)fact(N + n + 1) * (float)fact(N - n)) / ((float)N + 1.0f);
}

void beamWeightsHypercardioid2Spherical(int N, float* b_n)
{
    /* c_n = getSHreal(N, [0, 0]) (azimuth 0, inclination 0): only the m = 0 terms are non-zero there, Y_n0 = sqrt((2n+1)/(4 pi)) */
    float dirs[2] = { 0.0f, 0.0f };
    std::vector<float> Y((size_t)(N + 1) * (N + 1));
    getSHreal(N, dirs, 1, Y.data());
    for (int n = 0; n < N + 1; n++) b_n[n] = Y[(n + 1) * (n + 1) - n - 1] * 4.0f * SAF_PI / powf((float)N + 1.0f, 2.0f);
}

void rotateAxisCoeffsReal(int order, float* c_n, float theta_0, float phi_0, float* c_nm) { rotate_axis_coeffs_real(order, c_n, theta_0, phi_0, c_nm); }

/* ---------------- beamformer (beamformer.h) ---------------- */
void saf_hip_beamformer_setFrameSize(int frameSize)
{
    if (frameSize <= 0 || frameSize % 4 != 0) SAF_FATAL("beamformer frame size must be a positive multiple of 4");
    g_beamformer_frame_size = frameSize;
}

#define PB Beamformer* p = (Beamformer*)hBeam

void beamformer_create(void** const phBeam)
{
    Beamformer* p = new Beamformer();
    *phBeam = p;
    p->F = g_beamformer_frame_size;
    p->beamOrder = 1;
    const float* def = table_required("default_LScoords64_rad", 128);
    for (int i = 0; i < SAF_MAXCH; i++) {
        p->beam_dirs_deg[i][0] = def[i * 2] * 180.0f / SAF_PI;
        p->beam_dirs_deg[i][1] = (def[i * 2 + 1] - SAF_PI / 2.0f) < -SAF_PI / 2.0f ? (SAF_PI / 2.0f + def[i * 2 + 1]) : (def[i * 2 + 1] - SAF_PI / 2.0f);
        p->beam_dirs_deg[i][1] *= 180.0f / SAF_PI;
        p->recalc[i] = 1;
    }
    p->nBeams = 1; p->beamType = STATIC_BEAM_TYPE_HYPERCARDIOID; p->chOrdering = CH_ACN; p->norm = NORM_SN3D;
    memset(p->W, 0, sizeof(p->W)); memset(p->prevW, 0, sizeof(p->prevW));
}

void beamformer_destroy(void** const phBeam)
{
    Beamformer* p = (Beamformer*)*phBeam;
    if (!p) return;
    if (p->ready) HIP_CHECK(hipStreamSynchronize(stream()));
    delete p;
    *phBeam = nullptr;
}

void beamformer_init(void* const hBeam, int sampleRate)
{
    PB;
    p->fs = sampleRate;
    memset(p->W, 0, sizeof(p->W)); memset(p->prevW, 0, sizeof(p->prevW));
    p->clearState = true;
    for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc[ch] = 1;
}

void beamformer_process(void* const hBeam, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)
{
    PB;
    const int F = p->F, order = p->beamOrder, nSH = ORDER2NSH(order), nBeams = p->nBeams;
    if (nSamples != F) {                                                  /* beamformer.c:184-186 */
        for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
        return;
    }
    bf_setup(p);
    p->h_in.ensure((size_t)SAF_MAXCH * F); p->h_out.ensure((size_t)SAF_MAXCH * F);
    if (!p->d_in.p) { p->d_in.alloc((size_t)SAF_MAXCH * F); p->d_out.alloc((size_t)SAF_MAXCH * F); }
    static const int fuma2acn[4] = { 0, 2, 3, 1 };                        /* ACN channel c reads FuMa channel fuma2acn[c] (saf_hoa.c:40-70) */
    const bool fuma = p->chOrdering == CH_FUMA && order == 1;
    for (int c = 0; c < nSH; c++) {
        const int src = fuma ? fuma2acn[c] : c;
        if (src < nInputs) memcpy(p->h_in.p + (size_t)c * F, inputs[src], sizeof(float) * F);
        else memset(p->h_in.p + (size_t)c * F, 0, sizeof(float) * F);
    }
    const int nOut = nBeams < nOutputs ? nBeams : (nOutputs < 0 ? 0 : nOutputs);
    if (zero_copy_io()) bf_run(p, p->h_in.p, 0, F, nSH, p->h_out.p, 0, F, nOut, 1);
    else {
        HIP_CHECK(hipMemcpyAsync(p->d_in.p, p->h_in.p, sizeof(float) * (size_t)nSH * F, hipMemcpyHostToDevice, stream()));
        bf_run(p, p->d_in.p, 0, F, nSH, p->d_out.p, 0, F, nOut, 1);
        if (nOut) HIP_CHECK(hipMemcpyAsync(p->h_out.p, p->d_out.p, sizeof(float) * (size_t)nOut * F, hipMemcpyDeviceToHost, stream()));
    }
    HIP_CHECK(hipStreamSynchronize(stream()));
    p->stagingBusy = false;
    for (int ch = 0; ch < nOut; ch++) memcpy(outputs[ch], p->h_out.p + (size_t)ch * F, sizeof(float) * F);
    for (int ch = nOut; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
}

void saf_hip_beamformer_process_dev(void* const hBeam, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                    float* d_out, long long out_frame_stride, long long out_ch_stride, int nOutputs, int nFrames)
{
    PB;
    if (nFrames <= 0) return;
    if (p->chOrdering == CH_FUMA) SAF_FATAL("saf_hip_beamformer_process_dev takes ACN channel order (convert FuMa on the host entry)");
    bf_run(p, d_in, in_frame_stride, in_ch_stride, nInputs, d_out, out_frame_stride, out_ch_stride, nOutputs, nFrames);
}

void beamformer_refreshSettings(void* const hBeam) { PB; for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc[ch] = 1; }
void beamformer_setBeamOrder(void* const hBeam, int v)
{
    PB;
    p->beamOrder = v < 1 ? 1 : (v > SAF_MAX_ORDER ? SAF_MAX_ORDER : v);
    for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc[ch] = 1;
    if (p->beamOrder != SH_ORDER_FIRST && p->chOrdering == CH_FUMA) p->chOrdering = CH_ACN;
    if (p->beamOrder != SH_ORDER_FIRST && p->norm == NORM_FUMA) p->norm = NORM_SN3D;
}
void beamformer_setBeamAzi_deg(void* const hBeam, int index, float v)
{
    PB;
    if (v > 180.0f) v = -360.0f + v;
    v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    p->beam_dirs_deg[index][0] = v; p->recalc[index] = 1;
}
void beamformer_setBeamElev_deg(void* const hBeam, int index, float v)
{
    PB;
    v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    p->beam_dirs_deg[index][1] = v; p->recalc[index] = 1;
}
void beamformer_setNumBeams(void* const hBeam, int n)
{
    PB;
    n = n < 1 ? 1 : (n > SAF_MAXCH ? SAF_MAXCH : n);         /* the reference stores any value: MAX_NUM_BEAMS = 64 is the array bound */
    if (p->nBeams != n) { p->nBeams = n; for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc[ch] = 1; }
}
void beamformer_setChOrder(void* const hBeam, int o) { PB; if ((CH_ORDER)o != CH_FUMA || p->beamOrder == SH_ORDER_FIRST) p->chOrdering = (CH_ORDER)o; }
void beamformer_setNormType(void* const hBeam, int t) { PB; if ((NORM_TYPES)t != NORM_FUMA || p->beamOrder == SH_ORDER_FIRST) p->norm = (NORM_TYPES)t; }
void beamformer_setBeamType(void* const hBeam, int id) { PB; p->beamType = id; for (int ch = 0; ch < SAF_MAXCH; ch++) p->recalc[ch] = 1; }
int beamformer_getFrameSize(void) { return g_beamformer_frame_size; }
int beamformer_getBeamOrder(void* const hBeam) { PB; return p->beamOrder; }
float beamformer_getBeamAzi_deg(void* const hBeam, int index) { PB; return p->beam_dirs_deg[index][0]; }
float beamformer_getBeamElev_deg(void* const hBeam, int index) { PB; return p->beam_dirs_deg[index][1]; }
int beamformer_getNumBeams(void* const hBeam) { PB; return p->nBeams; }
int beamformer_getMaxNumBeams(void) { return SAF_MAXCH; }
int beamformer_getNSHrequired(void* const hBeam) { PB; return ORDER2NSH(p->beamOrder); }
int beamformer_getChOrder(void* const hBeam) { PB; return (int)p->chOrdering; }
int beamformer_getNormType(void* const hBeam) { PB; return (int)p->norm; }
int beamformer_getBeamType(void* const hBeam) { PB; return p->beamType; }
int beamformer_getProcessingDelay(void) { return g_beamformer_frame_size; }

}
