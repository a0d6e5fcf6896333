/*
 * orc_examples.c — oracle: the operator-level block paths (ambi_dec, ambi_enc),
 * the matrix convolver and the binauraliser band MAC.
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  Reference paths relative to
 * /root/reference.
 */
#include "saf_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include <time.h>

#define ORC_PI 3.14159265358979323846264338327950288f
#define NSH(o) (((o) + 1) * ((o) + 1))
#define MAX_SH_ORDER 7            /* _common.h:50 */
#define MAX_CH 64                 /* _common.h:228 */
#define HOP 128                   /* ambi_dec_internal.h:68 */
#define NBANDS 133                /* ambi_dec_internal.h:69 */
#define NUM_DECODERS 2

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* loudspeaker presets used by loadLoudspeakerArrayPreset (ambi_dec_internal.c:117-313); enum values _common.h */
static const struct { int id; const char* tab; int n; } g_presets[] = {
    { 1, "5pX_dirs_deg", 5 }, { 3, "5pX_dirs_deg", 5 }, { 4, "7pX_dirs_deg", 7 }, { 5, "8pX_dirs_deg", 8 },
    { 6, "9pX_dirs_deg", 9 }, { 7, "10pX_dirs_deg", 10 }, { 8, "11pX_dirs_deg", 11 }, { 9, "11pX_7_4_dirs_deg", 11 },
    { 10, "13pX_dirs_deg", 13 }, { 11, "22pX_dirs_deg", 22 }, { 13, "Aalto_MCC_dirs_deg", 45 },
    { 14, "Aalto_MCCsubset_dirs_deg", 37 }, { 15, "Aalto_Apaja_dirs_deg", 29 }, { 16, "Aalto_LR_dirs_deg", 13 },
    { 17, "DTU_AVIL_dirs_deg", 64 }, { 18, "Zylia_Lab_dirs_deg", 22 }, { 19, "Tdesign_degree_2_dirs_deg", 4 },
    { 20, "Tdesign_degree_4_dirs_deg", 12 }, { 21, "Tdesign_degree_6_dirs_deg", 24 }, { 22, "Tdesign_degree_8_dirs_deg", 36 },
    { 23, "Tdesign_degree_9_dirs_deg", 48 }, { 24, "Tdesign_degree_10_dirs_deg", 60 }, { 25, "SphCovering_9_dirs_deg", 9 },
    { 26, "SphCovering_16_dirs_deg", 16 }, { 27, "SphCovering_25_dirs_deg", 25 }, { 28, "SphCovering_49_dirs_deg", 49 },
    { 29, "SphCovering_64_dirs_deg", 64 },
};

/* ========================================================================== */
/*                                  ambi_dec                                  */
/* ========================================================================== */

typedef struct {
    int F, T;
    void* hSTFT;
    float* SHFrameTD;        /* [64][F] */
    float* outputFrameTD;    /* [64][F] */
    orc_cpx* SHframeTF;      /* [133][64][T] */
    orc_cpx* outputframeTF;  /* [133][64][T] */
    float freqVector[NBANDS];
    int fs;
    int codecInitialised;
    float* M_dec[NUM_DECODERS][MAX_SH_ORDER];
    float* M_dec_maxrE[NUM_DECODERS][MAX_SH_ORDER];
    float M_norm[NUM_DECODERS][MAX_SH_ORDER][2];
    int new_nLoudpkrs, nLoudpkrs, loudpkrs_nDims;
    int masterOrder, new_masterOrder;
    int orderPerBand[NBANDS];
    int dec_method[NUM_DECODERS];
    int rE_WEIGHT[NUM_DECODERS];
    int diffEQmode[NUM_DECODERS];   /* 1 amplitude preserving, 2 energy preserving (ambi_dec.h:101-105) */
    float transitionFreq;
    float loudpkrs_dirs_deg[MAX_CH][2];
    int chOrdering, norm;
    double t_fwd, t_dec, t_bwd;
    /* binauralised output (ambi_dec.c:349-445, 543-563); the HRIR set is injected (the reference's default set is absent) */
    int binauraliseLS, new_binauraliseLS, enableHRIRsPreProc, reinit_hrtfs, recalc_hrtf[MAX_CH];
    float* set_hrirs; float* set_dirs; int set_N, set_len, set_fs;
    float* itds_s; float* weights; float* gtableComp; int* gtableIdx; orc_cpx* hrtf_fb; float* hrtf_fb_mag;
    int N_hrir_dirs, N_gtable, hrtf_nTriangles;
    orc_cpx* hrtf_interp;      /* [MAX_CH][NBANDS][2] */
    orc_cpx* binframeTF;       /* [NBANDS][2][T] */
} orc_ambi_dec;

static void load_preset(int preset, float dirs[MAX_CH][2], int* nCH, int* nDims)
{
    int found = -1;
    for (unsigned i = 0; i < sizeof(g_presets) / sizeof(g_presets[0]); i++) if (g_presets[i].id == preset) found = (int)i;
    if (found < 0) found = 0;     /* default: 5.x */
    int d0, d1;
    const float* t = orc_table(g_presets[found].tab, &d0, &d1);
    assert(t);
    int n = g_presets[found].n, ch;
    for (ch = 0; ch < n; ch++) { dirs[ch][0] = t[ch * 2]; dirs[ch][1] = t[ch * 2 + 1]; }
    const float* def = orc_table("default_LScoords64_rad", &d0, &d1);
    assert(def);
    for (; ch < MAX_CH; ch++) for (int i = 0; i < 2; i++) dirs[ch][i] = def[ch * 2 + i] * (180.0f / ORC_PI);
    *nCH = n;
    float sum_elev = 0.0f;
    for (int i = 0; i < n; i++) sum_elev += fabsf(dirs[i][1]);
    *nDims = sum_elev < 0.01f ? 2 : 3;
}

/* ambi_dec_create (ambi_dec.c:48-116) */
void orc_ambi_dec_create(void** ph, int frameSize)
{
    orc_ambi_dec* p = (orc_ambi_dec*)calloc(1, sizeof(orc_ambi_dec));
    assert(frameSize % HOP == 0 && frameSize / HOP <= 64);
    p->F = frameSize; p->T = frameSize / HOP;
    load_preset(21 /* T_DESIGN_24 */, p->loudpkrs_dirs_deg, &p->new_nLoudpkrs, &p->loudpkrs_nDims);
    p->masterOrder = p->new_masterOrder = 1;
    for (int b = 0; b < NBANDS; b++) p->orderPerBand[b] = 1;
    p->nLoudpkrs = p->new_nLoudpkrs;
    p->chOrdering = 1; p->norm = 2;     /* CH_ACN, NORM_SN3D */
    p->dec_method[0] = p->dec_method[1] = ORC_DECODER_ALLRAD;
    p->rE_WEIGHT[0] = p->rE_WEIGHT[1] = 1;
    p->diffEQmode[0] = p->diffEQmode[1] = 2;
    p->transitionFreq = 800.0f;
    p->SHFrameTD = (float*)calloc((size_t)MAX_CH * p->F, sizeof(float));
    p->outputFrameTD = (float*)calloc((size_t)MAX_CH * p->F, sizeof(float));
    p->SHframeTF = (orc_cpx*)calloc((size_t)NBANDS * MAX_CH * p->T, sizeof(orc_cpx));
    p->outputframeTF = (orc_cpx*)calloc((size_t)NBANDS * MAX_CH * p->T, sizeof(orc_cpx));
    p->enableHRIRsPreProc = 1; p->reinit_hrtfs = 1;
    for (int ch = 0; ch < MAX_CH; ch++) p->recalc_hrtf[ch] = 1;
    p->hrtf_interp = (orc_cpx*)calloc((size_t)MAX_CH * NBANDS * 2, sizeof(orc_cpx));
    p->binframeTF = (orc_cpx*)calloc((size_t)NBANDS * 2 * p->T, sizeof(orc_cpx));
    *ph = p;
}

void orc_ambi_dec_destroy(void** ph)
{
    orc_ambi_dec* p = (orc_ambi_dec*)*ph;
    if (!p) return;
    if (p->hSTFT) orc_afSTFT_destroy(&p->hSTFT);
    free(p->SHFrameTD); free(p->outputFrameTD); free(p->SHframeTF); free(p->outputframeTF);
    free(p->set_hrirs); free(p->set_dirs); free(p->itds_s); free(p->weights); free(p->gtableComp); free(p->gtableIdx);
    free(p->hrtf_fb); free(p->hrtf_fb_mag); free(p->hrtf_interp); free(p->binframeTF);
    for (int d = 0; d < NUM_DECODERS; d++) for (int n = 0; n < MAX_SH_ORDER; n++) { free(p->M_dec[d][n]); free(p->M_dec_maxrE[d][n]); }
    free(p); *ph = NULL;
}

/* ambi_dec_init (ambi_dec.c:168-179) */
void orc_ambi_dec_init(void* h, int sampleRate)
{
    orc_ambi_dec* p = (orc_ambi_dec*)h;
    p->fs = sampleRate;
    orc_afSTFT_getCentreFreqs(p->hSTFT, (float)sampleRate, NBANDS, p->freqVector);
}

/* ambi_dec_initCodec (ambi_dec.c:181-455) */
void orc_ambi_dec_initCodec(void* h)
{
    orc_ambi_dec* p = (orc_ambi_dec*)h;
    const int masterOrder = p->new_masterOrder;
    const int max_nSH = NSH(masterOrder);
    int nLS = p->new_nLoudpkrs;
    if (!p->hSTFT) orc_afSTFT_create(&p->hSTFT, max_nSH, p->new_binauraliseLS ? 2 : nLS, HOP, 0, 1, ORC_AFSTFT_BANDS_CH_TIME);
    else orc_afSTFT_channelChange(p->hSTFT, max_nSH, p->new_binauraliseLS ? 2 : nLS);
    orc_afSTFT_clearBuffers(p->hSTFT);
    p->binauraliseLS = p->new_binauraliseLS;
    p->nLoudpkrs = nLS;
    float sum_elev = 0.0f;
    for (int ch = 0; ch < nLS; ch++) sum_elev += fabsf(p->loudpkrs_dirs_deg[ch][1]);
    p->loudpkrs_nDims = (((sum_elev < 5.0f) && (sum_elev > -5.0f)) || (nLS < 4)) ? 2 : 3;
    const int virt = p->loudpkrs_nDims == 2 && (p->dec_method[0] == ORC_DECODER_ALLRAD || p->dec_method[1] == ORC_DECODER_ALLRAD);
    if (virt) {
        assert(nLS <= MAX_CH - 2);
        p->loudpkrs_dirs_deg[nLS][0] = 0.0f; p->loudpkrs_dirs_deg[nLS][1] = -90.0f;
        p->loudpkrs_dirs_deg[nLS + 1][0] = 0.0f; p->loudpkrs_dirs_deg[nLS + 1][1] = 90.0f;
        nLS += 2;
    }
    const int nGrid = 480;
    int d0, d1;
    const float* grid = orc_table("Tdesign_degree_30_dirs_deg", &d0, &d1);
    assert(grid && d0 == nGrid);
    float* g = (float*)malloc(sizeof(float) * nLS);
    for (int d = 0; d < NUM_DECODERS; d++) {
        float* M_tmp = (float*)malloc(sizeof(float) * nLS * max_nSH);
        orc_getLoudspeakerDecoderMtx(&p->loudpkrs_dirs_deg[0][0], nLS, p->dec_method[d], masterOrder, 0, M_tmp);
        for (int n = 1; n <= masterOrder; n++) {
            const int nSHo = NSH(n);
            free(p->M_dec[d][n - 1]); free(p->M_dec_maxrE[d][n - 1]);
            float* M = p->M_dec[d][n - 1] = (float*)malloc(sizeof(float) * nLS * nSHo);
            float* Mr = p->M_dec_maxrE[d][n - 1] = (float*)malloc(sizeof(float) * nLS * nSHo);
            for (int i = 0; i < nLS; i++) for (int j = 0; j < nSHo; j++) M[i * nSHo + j] = M_tmp[i * max_nSH + j];
            float* a_n = (float*)malloc(sizeof(float) * nSHo);
            orc_getMaxREweights(n, 0, a_n);
            for (int i = 0; i < nLS; i++) for (int j = 0; j < nSHo; j++) Mr[i * nSHo + j] = M[i * nSHo + j] * a_n[j];
            free(a_n);
            float* Y = (float*)malloc(sizeof(float) * nSHo);
            float a_avg = 0.0f, e_avg = 0.0f;
            float* a = (float*)malloc(sizeof(float) * nGrid);
            float* e = (float*)malloc(sizeof(float) * nGrid);
            for (int ng = 0; ng < nGrid; ng++) {
                float azi_incl[2];
                azi_incl[0] = grid[ng * 2] * ORC_PI / 180.0f;
                azi_incl[1] = ORC_PI / 2.0f - grid[ng * 2 + 1] * ORC_PI / 180.0f;
                orc_getSHreal(n, azi_incl, 1, Y);
                for (int i = 0; i < nLS; i++) {
                    float acc = 0.0f;
                    for (int j = 0; j < nSHo; j++) acc += M[i * nSHo + j] * Y[j];
                    g[i] = acc;
                }
                a[ng] = e[ng] = 0.0f;
                for (int i = 0; i < nLS; i++) { a[ng] += g[i]; e[ng] += powf(g[i], 2.0f); }
            }
            for (int ng = 0; ng < nGrid; ng++) { a_avg += a[ng]; e_avg += e[ng]; }
            a_avg /= (float)nGrid; e_avg /= (float)nGrid;
            p->M_norm[d][n - 1][0] = 1.0f / (a_avg + 2.23e-6f);
            p->M_norm[d][n - 1][1] = sqrtf(1.0f / (e_avg + 2.23e-6f));
            free(Y); free(a); free(e);
            /* dropping the virtual loudspeakers = keeping the first nLoudpkrs rows (ambi_dec.c:336-341) */
        }
        free(M_tmp);
    }
    free(g);
    p->masterOrder = p->new_masterOrder;
    /* binaural-related initialisations (ambi_dec.c:349-445); skipped while no HRIR set has been injected */
    if (p->reinit_hrtfs && p->set_hrirs) {
        const int N = p->set_N, len = p->set_len;
        p->N_hrir_dirs = N;
        p->itds_s = (float*)realloc(p->itds_s, sizeof(float) * N);
        orc_estimateITDs(p->set_hrirs, N, len, p->set_fs, p->itds_s);
        float* gtable = NULL;
        orc_generateVBAPgainTable3D(p->set_dirs, N, 2, 5, 1, 0, 0.0f, &gtable, &p->N_gtable, &p->hrtf_nTriangles);
        assert(gtable);
        p->gtableComp = (float*)realloc(p->gtableComp, sizeof(float) * 3 * p->N_gtable);
        p->gtableIdx = (int*)realloc(p->gtableIdx, sizeof(int) * 3 * p->N_gtable);
        orc_compressVBAPgainTable3D(gtable, p->N_gtable, N, p->gtableComp, p->gtableIdx);
        free(gtable);
        p->hrtf_fb = (orc_cpx*)realloc(p->hrtf_fb, sizeof(orc_cpx) * (size_t)NBANDS * 2 * N);
        orc_afSTFT_FIRtoFilterbankCoeffs(p->set_hrirs, N, 2, len, HOP, 0, 1, p->hrtf_fb);
        if (p->enableHRIRsPreProc) {
            p->weights = (float*)realloc(p->weights, sizeof(float) * N);
            if (N <= 3600) orc_getVoronoiWeights(p->set_dirs, N, p->weights);
            else for (int i = 0; i < N; i++) p->weights[i] = 4.f * ORC_PI / (float)N;
            orc_diffuseFieldEqualiseHRTFs(N, NBANDS, p->weights, p->hrtf_fb);
        }
        p->hrtf_fb_mag = (float*)realloc(p->hrtf_fb_mag, sizeof(float) * (size_t)NBANDS * 2 * N);
        for (size_t i = 0; i < (size_t)NBANDS * 2 * N; i++) p->hrtf_fb_mag[i] = hypotf(p->hrtf_fb[i].re, p->hrtf_fb[i].im);
        p->reinit_hrtfs = 0;
    }
    p->codecInitialised = 1;
}

/* ambi_dec_process (ambi_dec.c:457-580) */
void orc_ambi_dec_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)
{
    orc_ambi_dec* p = (orc_ambi_dec*)h;
    const int F = p->F, T = p->T;
    const int masterOrder = p->masterOrder, nSH = NSH(masterOrder), nLS = p->nLoudpkrs;
    if (nSamples == F && p->codecInitialised) {
        int i;
        for (i = 0; i < (nSH < nInputs ? nSH : nInputs); i++) memcpy(&p->SHFrameTD[(size_t)i * F], inputs[i], sizeof(float) * F);
        for (; i < nSH; i++) memset(&p->SHFrameTD[(size_t)i * F], 0, sizeof(float) * F);
        if (p->chOrdering == 2) orc_convertHOAChannelConvention(p->SHFrameTD, masterOrder, F, 2, 1);
        if (p->norm == 2) orc_convertHOANormConvention(p->SHFrameTD, masterOrder, F, 2, 1);
        else if (p->norm == 3) orc_convertHOANormConvention(p->SHFrameTD, masterOrder, F, 3, 1);
        double t0 = now_s();
        orc_afSTFT_forward_knownDimensions(p->hSTFT, p->SHFrameTD, F, MAX_CH, T, p->SHframeTF);
        double t1 = now_s();
        memset(p->outputframeTF, 0, sizeof(orc_cpx) * (size_t)NBANDS * MAX_CH * T);
        for (int band = 0; band < NBANDS; band++) {
            int ob = p->orderPerBand[band] < masterOrder ? p->orderPerBand[band] : masterOrder;
            if (ob < 1) ob = 1;
            const int nSHb = NSH(ob);
            const int decIdx = p->freqVector[band] < p->transitionFreq ? 0 : 1;
            const float* M = p->rE_WEIGHT[decIdx] ? p->M_dec_maxrE[decIdx][ob - 1] : p->M_dec[decIdx][ob - 1];
            const float sc = p->M_norm[decIdx][ob - 1][p->diffEQmode[decIdx] == 1 ? 0 : 1];
            const orc_cpx* X = &p->SHframeTF[(size_t)band * MAX_CH * T];
            orc_cpx* Yo = &p->outputframeTF[(size_t)band * MAX_CH * T];
            /* cgemm with a real-valued complex matrix, then sscal (ambi_dec.c:525-539).
             * Inner loop runs over the 2T contiguous floats (re/im of T slots) of one SH row so it
             * vectorises; every output element still accumulates over k in ascending order. */
            const int T2 = 2 * T;
            const float* Xf = (const float*)X;
            float* Yf = (float*)Yo;
            if (T2 == 8) {
                /* fixed-width case of the headline config (F = 512): 8 floats = one SIMD register,
                 * 4 loudspeaker rows at a time (GCC vector extension; plain mul + add, no FMA) */
                typedef float v8f __attribute__((vector_size(32), aligned(4)));
                int l = 0;
                for (; l + 4 <= nLS; l += 4) {
                    v8f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
                    const float* m0 = &M[(l + 0) * nSHb]; const float* m1 = &M[(l + 1) * nSHb];
                    const float* m2 = &M[(l + 2) * nSHb]; const float* m3 = &M[(l + 3) * nSHb];
                    for (int k = 0; k < nSHb; k++) {
                        const v8f xr = *(const v8f*)&Xf[(size_t)k * 8];
                        a0 += m0[k] * xr; a1 += m1[k] * xr; a2 += m2[k] * xr; a3 += m3[k] * xr;
                    }
                    *(v8f*)&Yf[(size_t)(l + 0) * 8] = a0 * sc; *(v8f*)&Yf[(size_t)(l + 1) * 8] = a1 * sc;
                    *(v8f*)&Yf[(size_t)(l + 2) * 8] = a2 * sc; *(v8f*)&Yf[(size_t)(l + 3) * 8] = a3 * sc;
                }
                for (; l < nLS; l++) {
                    v8f a0 = {0};
                    for (int k = 0; k < nSHb; k++) a0 += M[l * nSHb + k] * *(const v8f*)&Xf[(size_t)k * 8];
                    *(v8f*)&Yf[(size_t)l * 8] = a0 * sc;
                }
            } else {
                for (int l = 0; l < nLS; l++) {
                    float acc[128];
                    for (int q = 0; q < T2; q++) acc[q] = 0.0f;
                    for (int k = 0; k < nSHb; k++) {
                        const float m = M[l * nSHb + k];
                        const float* xr = &Xf[(size_t)k * T2];
                        for (int q = 0; q < T2; q++) acc[q] += m * xr[q];
                    }
                    for (int q = 0; q < T2; q++) Yf[(size_t)l * T2 + q] = acc[q] * sc;
                }
            }
        }
        const int bin = p->binauraliseLS && p->hrtf_fb_mag;
        if (bin) {      /* binauralise the loudspeaker signals (ambi_dec.c:543-563) */
            memset(p->binframeTF, 0, sizeof(orc_cpx) * (size_t)NBANDS * 2 * T);
            for (int ch = 0; ch < nLS; ch++) {
                orc_cpx* hi = &p->hrtf_interp[(size_t)ch * NBANDS * 2];
                if (p->recalc_hrtf[ch]) {
                    orc_interpHRTFs_ps(p->gtableComp, p->gtableIdx, p->itds_s, p->hrtf_fb_mag, p->N_hrir_dirs, p->freqVector,
                                       p->loudpkrs_dirs_deg[ch][0], p->loudpkrs_dirs_deg[ch][1], hi);
                    p->recalc_hrtf[ch] = 0;
                }
                for (int band = 0; band < NBANDS; band++)
                    for (int e = 0; e < 2; e++) {
                        const orc_cpx a = hi[band * 2 + e];
                        const orc_cpx* x = &p->outputframeTF[((size_t)band * MAX_CH + ch) * T];
                        orc_cpx* y = &p->binframeTF[((size_t)band * 2 + e) * T];
                        for (int t = 0; t < T; t++) { y[t].re += a.re * x[t].re - a.im * x[t].im; y[t].im += a.re * x[t].im + a.im * x[t].re; }
                    }
            }
            const float sc = 1.0f / sqrtf((float)nLS);
            for (size_t i = 0; i < (size_t)NBANDS * 2 * T; i++) { p->binframeTF[i].re *= sc; p->binframeTF[i].im *= sc; }
        }
        double t2 = now_s();
        orc_afSTFT_backward_knownDimensions(p->hSTFT, bin ? p->binframeTF : p->outputframeTF, F, bin ? 2 : MAX_CH, T, p->outputFrameTD);
        double t3 = now_s();
        p->t_fwd += t1 - t0; p->t_dec += t2 - t1; p->t_bwd += t3 - t2;
        int ch;
        const int nOutCh = bin ? 2 : nLS;
        for (ch = 0; ch < (nOutCh < nOutputs ? nOutCh : nOutputs); ch++) memcpy(outputs[ch], &p->outputFrameTD[(size_t)ch * F], sizeof(float) * F);
        for (; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    } else
        for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
}

#define CLAMPI(v, lo, hi) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
void orc_ambi_dec_setMasterDecOrder(void* h, int o) { orc_ambi_dec* p = (orc_ambi_dec*)h; p->new_masterOrder = CLAMPI(o, 1, MAX_SH_ORDER); p->codecInitialised = 0;
    if (p->new_masterOrder != 1 && p->chOrdering == 2) p->chOrdering = 1;
    if (p->new_masterOrder != 1 && p->norm == 3) p->norm = 2; }
void orc_ambi_dec_setDecOrder(void* h, int o, int band) { orc_ambi_dec* p = (orc_ambi_dec*)h; p->orderPerBand[band] = CLAMPI(o, 1, p->new_masterOrder); }
void orc_ambi_dec_setDecOrderAllBands(void* h, int o) { orc_ambi_dec* p = (orc_ambi_dec*)h; for (int b = 0; b < NBANDS; b++) p->orderPerBand[b] = CLAMPI(o, 1, p->new_masterOrder); }
void orc_ambi_dec_setLoudspeakers(void* h, const float* dirs_deg, int nLS)
{
    orc_ambi_dec* p = (orc_ambi_dec*)h;
    for (int i = 0; i < nLS; i++) { p->loudpkrs_dirs_deg[i][0] = dirs_deg[2 * i]; p->loudpkrs_dirs_deg[i][1] = dirs_deg[2 * i + 1]; }
    p->new_nLoudpkrs = nLS; p->codecInitialised = 0;
}
void orc_ambi_dec_setOutputConfigPreset(void* h, int id) { orc_ambi_dec* p = (orc_ambi_dec*)h; load_preset(id, p->loudpkrs_dirs_deg, &p->new_nLoudpkrs, &p->loudpkrs_nDims); p->codecInitialised = 0;
    for (int ch = 0; ch < MAX_CH; ch++) p->recalc_hrtf[ch] = 1; }
void orc_ambi_dec_setHRIRs(void* h, const float* hrirs, const float* dirs_deg, int N, int len, int fs)
{
    orc_ambi_dec* p = (orc_ambi_dec*)h;
    free(p->set_hrirs); free(p->set_dirs);
    p->set_hrirs = (float*)malloc(sizeof(float) * (size_t)N * 2 * len); memcpy(p->set_hrirs, hrirs, sizeof(float) * (size_t)N * 2 * len);
    p->set_dirs = (float*)malloc(sizeof(float) * (size_t)N * 2); memcpy(p->set_dirs, dirs_deg, sizeof(float) * (size_t)N * 2);
    p->set_N = N; p->set_len = len; p->set_fs = fs;
    p->reinit_hrtfs = 1; p->codecInitialised = 0;
    for (int ch = 0; ch < MAX_CH; ch++) p->recalc_hrtf[ch] = 1;
}
void orc_ambi_dec_setBinauraliseLSflag(void* h, int s) { orc_ambi_dec* p = (orc_ambi_dec*)h; p->new_binauraliseLS = s; if (p->new_binauraliseLS != p->binauraliseLS) p->codecInitialised = 0; }
void orc_ambi_dec_setEnableHRIRsPreProc(void* h, int s) { orc_ambi_dec* p = (orc_ambi_dec*)h;
    if (s != p->enableHRIRsPreProc) { p->enableHRIRsPreProc = s; p->reinit_hrtfs = 1; p->codecInitialised = 0; for (int ch = 0; ch < MAX_CH; ch++) p->recalc_hrtf[ch] = 1; } }
void orc_ambi_dec_setLoudspeakerAzi_deg(void* h, int i, float v) { orc_ambi_dec* p = (orc_ambi_dec*)h; if (v > 180.0f) v = -360.0f + v; v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v);
    if (p->loudpkrs_dirs_deg[i][0] != v) { p->loudpkrs_dirs_deg[i][0] = v; p->recalc_hrtf[i] = 1; p->codecInitialised = 0; } }
void orc_ambi_dec_setLoudspeakerElev_deg(void* h, int i, float v) { orc_ambi_dec* p = (orc_ambi_dec*)h; v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v);
    if (p->loudpkrs_dirs_deg[i][1] != v) { p->loudpkrs_dirs_deg[i][1] = v; p->recalc_hrtf[i] = 1; p->codecInitialised = 0; } }
const orc_cpx* orc_ambi_dec_getHRTFinterp(void* h) { return ((orc_ambi_dec*)h)->hrtf_interp; }
void orc_ambi_dec_setChOrder(void* h, int v) { orc_ambi_dec* p = (orc_ambi_dec*)h; if (v != 2 || p->new_masterOrder == 1) p->chOrdering = v; }
void orc_ambi_dec_setNormType(void* h, int v) { orc_ambi_dec* p = (orc_ambi_dec*)h; if (v != 3 || p->new_masterOrder == 1) p->norm = v; }
void orc_ambi_dec_setDecMethod(void* h, int index, int id) { orc_ambi_dec* p = (orc_ambi_dec*)h; p->dec_method[index] = id; p->codecInitialised = 0; }
void orc_ambi_dec_setDecEnableMaxrE(void* h, int index, int id) { ((orc_ambi_dec*)h)->rE_WEIGHT[index] = id; }
void orc_ambi_dec_setDecNormType(void* h, int index, int id) { ((orc_ambi_dec*)h)->diffEQmode[index] = id; }
void orc_ambi_dec_setTransitionFreq(void* h, float v) { ((orc_ambi_dec*)h)->transitionFreq = v < 500.0f ? 500.0f : (v > 2000.0f ? 2000.0f : v); }
int orc_ambi_dec_getNumLoudspeakers(void* h) { return ((orc_ambi_dec*)h)->new_nLoudpkrs; }
const float* orc_ambi_dec_getDecMtx(void* h, int dec, int order, int maxrE) { orc_ambi_dec* p = (orc_ambi_dec*)h; return maxrE ? p->M_dec_maxrE[dec][order - 1] : p->M_dec[dec][order - 1]; }
float orc_ambi_dec_getMnorm(void* h, int dec, int order, int which) { return ((orc_ambi_dec*)h)->M_norm[dec][order - 1][which]; }
const float* orc_ambi_dec_getFreqVector(void* h) { return ((orc_ambi_dec*)h)->freqVector; }
void orc_ambi_dec_getStageTimes(void* h, double* a, double* b, double* c) { orc_ambi_dec* p = (orc_ambi_dec*)h; *a = p->t_fwd; *b = p->t_dec; *c = p->t_bwd; }

/* ========================================================================== */
/*                                  ambi_enc                                  */
/* ========================================================================== */

typedef struct {
    int F;
    float* inputFrameTD;       /* [64][F] */
    float* prev_inputFrameTD;  /* [64][F] */
    float* tempFrame;          /* [64][F] */
    float* outputFrameTD;      /* [64][F] */
    float* fadeIn; float* fadeOut;
    float Y[MAX_CH][MAX_CH], prev_Y[MAX_CH][MAX_CH];
    int recalc_SH_FLAG[MAX_CH];
    float src_dirs_deg[MAX_CH][2], src_gains[MAX_CH];
    int nSources, chOrdering, norm, order, enablePostScaling;
    float fs;
} orc_ambi_enc;

/* ambi_enc_create (ambi_enc.c:28-51); default source preset = mono at (0,0) padded with the default 64 coords */
void orc_ambi_enc_create(void** ph, int frameSize)
{
    orc_ambi_enc* p = (orc_ambi_enc*)calloc(1, sizeof(orc_ambi_enc));
    p->F = frameSize;
    p->inputFrameTD = (float*)calloc((size_t)MAX_CH * frameSize, sizeof(float));
    p->prev_inputFrameTD = (float*)calloc((size_t)MAX_CH * frameSize, sizeof(float));
    p->tempFrame = (float*)calloc((size_t)MAX_CH * frameSize, sizeof(float));
    p->outputFrameTD = (float*)calloc((size_t)MAX_CH * frameSize, sizeof(float));
    p->fadeIn = (float*)calloc(frameSize, sizeof(float));
    p->fadeOut = (float*)calloc(frameSize, sizeof(float));
    int d0, d1;
    const float* def = orc_table("default_LScoords64_rad", &d0, &d1);
    for (int ch = 0; ch < MAX_CH; ch++) for (int i = 0; i < 2; i++) p->src_dirs_deg[ch][i] = def ? def[ch * 2 + i] * (180.0f / ORC_PI) : 0.0f;
    p->src_dirs_deg[0][0] = 0.0f; p->src_dirs_deg[0][1] = 0.0f;
    p->nSources = 1;
    for (int i = 0; i < MAX_CH; i++) { p->recalc_SH_FLAG[i] = 1; p->src_gains[i] = 1.f; }
    p->chOrdering = 1; p->norm = 2; p->order = 1; p->enablePostScaling = 1;
    *ph = p;
}
void orc_ambi_enc_destroy(void** ph)
{
    orc_ambi_enc* p = (orc_ambi_enc*)*ph; if (!p) return;
    free(p->inputFrameTD); free(p->prev_inputFrameTD); free(p->tempFrame); free(p->outputFrameTD); free(p->fadeIn); free(p->fadeOut);
    free(p); *ph = NULL;
}
/* ambi_enc_init (ambi_enc.c:67-84) */
void orc_ambi_enc_init(void* h, int sampleRate)
{
    orc_ambi_enc* p = (orc_ambi_enc*)h;
    p->fs = (float)sampleRate;
    for (int i = 1; i <= p->F; i++) { p->fadeIn[i - 1] = (float)i * 1.0f / (float)p->F; p->fadeOut[i - 1] = 1.0f - p->fadeIn[i - 1]; }
    memset(p->prev_Y, 0, sizeof(p->prev_Y));
    memset(p->prev_inputFrameTD, 0, sizeof(float) * (size_t)MAX_CH * p->F);
    for (int i = 0; i < MAX_CH; i++) p->recalc_SH_FLAG[i] = 1;
}
/* ambi_enc_process (ambi_enc.c:86-196) */
void orc_ambi_enc_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)
{
    orc_ambi_enc* p = (orc_ambi_enc*)h;
    const int F = p->F, nSources = p->nSources;
    const int order = p->order < MAX_SH_ORDER ? p->order : MAX_SH_ORDER, nSH = NSH(order);
    if (nSamples != F) { for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F); return; }
    int i;
    for (i = 0; i < (nSources < nInputs ? nSources : nInputs); i++) memcpy(&p->inputFrameTD[(size_t)i * F], inputs[i], sizeof(float) * F);
    for (; i < MAX_CH; i++) memset(&p->inputFrameTD[(size_t)i * F], 0, sizeof(float) * F);
    int mix = 0;
    float Y_src[MAX_CH];
    for (int ch = 0; ch < nSources; ch++) {
        if (p->recalc_SH_FLAG[ch]) {
            orc_getRSH_recur(order, p->src_dirs_deg[ch], 1, Y_src);
            int j;
            for (j = 0; j < nSH; j++) p->Y[j][ch] = Y_src[j];
            for (; j < MAX_CH; j++) p->Y[j][ch] = 0.0f;
            p->recalc_SH_FLAG[ch] = 0;
            mix = 1;
        }
        if (fabsf(p->src_gains[ch] - 1.f) > 1e-6f)
            for (int n = 0; n < F; n++) p->inputFrameTD[(size_t)ch * F + n] *= p->src_gains[ch];
    }
    /* encodes the PREVIOUS frame (ambi_enc.c:140-143) */
    for (int s = 0; s < nSH; s++)
        for (int n = 0; n < F; n++) {
            float acc = 0.0f;
            for (int c = 0; c < nSources; c++) acc += p->Y[s][c] * p->prev_inputFrameTD[(size_t)c * F + n];
            p->outputFrameTD[(size_t)s * F + n] = acc;
        }
    if (mix) {
        for (int s = 0; s < nSH; s++)
            for (int n = 0; n < F; n++) {
                float acc = 0.0f;
                for (int c = 0; c < nSources; c++) acc += p->prev_Y[s][c] * p->prev_inputFrameTD[(size_t)c * F + n];
                p->tempFrame[(size_t)s * F + n] = acc;
            }
        for (int s = 0; s < nSH; s++)
            for (int n = 0; n < F; n++) {
                float a = p->fadeIn[n] * p->outputFrameTD[(size_t)s * F + n];
                float b = p->fadeOut[n] * p->tempFrame[(size_t)s * F + n];
                p->outputFrameTD[(size_t)s * F + n] = a + b;
            }
        memcpy(p->prev_Y, p->Y, sizeof(p->Y));
    }
    memcpy(p->prev_inputFrameTD, p->inputFrameTD, sizeof(float) * (size_t)MAX_CH * F);
    if (p->enablePostScaling) {
        float scale = 1.0f / sqrtf((float)nSources);
        for (size_t q = 0; q < (size_t)nSH * F; q++) p->outputFrameTD[q] *= scale;
    }
    if (p->chOrdering == 2) orc_convertHOAChannelConvention(p->outputFrameTD, order, F, 1, 2);
    if (p->norm == 2) orc_convertHOANormConvention(p->outputFrameTD, order, F, 1, 2);
    else if (p->norm == 3) orc_convertHOANormConvention(p->outputFrameTD, order, F, 1, 3);
    for (i = 0; i < (nSH < nOutputs ? nSH : nOutputs); i++) memcpy(outputs[i], &p->outputFrameTD[(size_t)i * F], sizeof(float) * F);
    for (; i < nOutputs; i++) memset(outputs[i], 0, sizeof(float) * F);
}
/* setters (ambi_enc.c:205-330) */
void orc_ambi_enc_setOutputOrder(void* h, int o) { orc_ambi_enc* p = (orc_ambi_enc*)h; if (o != p->order) { p->order = o; for (int i = 0; i < MAX_CH; i++) p->recalc_SH_FLAG[i] = 1;
    if (p->order != 1 && p->chOrdering == 2) p->chOrdering = 1;
    if (p->order != 1 && p->norm == 3) p->norm = 2; } }
void orc_ambi_enc_setNumSources(void* h, int n) { orc_ambi_enc* p = (orc_ambi_enc*)h; p->nSources = CLAMPI(n, 1, MAX_CH); for (int i = 0; i < MAX_CH; i++) p->recalc_SH_FLAG[i] = 1; }
void orc_ambi_enc_setSourceAzi_deg(void* h, int idx, float a) { orc_ambi_enc* p = (orc_ambi_enc*)h; if (a > 180.0f) a = -360.0f + a; a = a < -180.0f ? -180.0f : (a > 180.0f ? 180.0f : a); p->recalc_SH_FLAG[idx] = 1; p->src_dirs_deg[idx][0] = a; }
void orc_ambi_enc_setSourceElev_deg(void* h, int idx, float e) { orc_ambi_enc* p = (orc_ambi_enc*)h; e = e < -90.0f ? -90.0f : (e > 90.0f ? 90.0f : e); p->recalc_SH_FLAG[idx] = 1; p->src_dirs_deg[idx][1] = e; }
void orc_ambi_enc_setSourceGain(void* h, int idx, float g) { ((orc_ambi_enc*)h)->src_gains[idx] = g; }
void orc_ambi_enc_setChOrder(void* h, int v) { orc_ambi_enc* p = (orc_ambi_enc*)h; if (v != 2 || p->order == 1) p->chOrdering = v; }
void orc_ambi_enc_setNormType(void* h, int v) { orc_ambi_enc* p = (orc_ambi_enc*)h; if (v != 3 || p->order == 1) p->norm = v; }
void orc_ambi_enc_setEnablePostScaling(void* h, int v) { ((orc_ambi_enc*)h)->enablePostScaling = v; }

/* ========================================================================== */
/*                              matrix convolver                              */
/* ========================================================================== */

typedef struct {
    int hopSize, fftSize, nBins, length_h, nCHin, nCHout, numFilterBlocks, numOvrlpAddBlocks, usePart;
    void* hFFT;
    float *x_pad, *hx_n, *z_n, *ovrlpAddBuffer, *y_n_overlap;
    orc_cpx *H_f, *X_n, *HX_n;
    orc_cpx** Hpart_f;
} orc_mc;

/* saf_matrixConv_create (saf_utility_matrixConv.c:49-130) */
void orc_matrixConv_create(void** ph, int hopSize, const float* H, int length_h, int nCHin, int nCHout, int usePartFLAG)
{
    orc_mc* h = (orc_mc*)calloc(1, sizeof(orc_mc));
    h->hopSize = hopSize; h->length_h = length_h; h->nCHin = nCHin; h->nCHout = nCHout; h->usePart = usePartFLAG;
    if (!usePartFLAG) {
        h->numOvrlpAddBlocks = (int)(ceilf((float)(hopSize + length_h - 1) / (float)hopSize) + 0.1f);
        h->fftSize = h->numOvrlpAddBlocks * hopSize;
        h->nBins = h->fftSize / 2 + 1;
        h->ovrlpAddBuffer = (float*)calloc((size_t)nCHout * h->fftSize, sizeof(float));
        h->x_pad = (float*)calloc((size_t)nCHin * h->fftSize, sizeof(float));
        h->hx_n = (float*)malloc(sizeof(float) * h->fftSize);
        h->z_n = (float*)malloc(sizeof(float) * h->fftSize);
        h->H_f = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nCHout * nCHin * h->nBins);
        h->X_n = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nCHin * h->nBins);
        h->HX_n = (orc_cpx*)malloc(sizeof(orc_cpx) * h->nBins);
        orc_rfft_create(&h->hFFT, h->fftSize);
        float* h_pad = (float*)calloc(h->fftSize, sizeof(float));
        for (int no = 0; no < nCHout; no++)
            for (int ni = 0; ni < nCHin; ni++) {
                memcpy(h_pad, &H[((size_t)no * nCHin + ni) * length_h], sizeof(float) * length_h);
                orc_rfft_forward(h->hFFT, h_pad, &h->H_f[((size_t)no * nCHin + ni) * h->nBins]);
            }
        free(h_pad);
    } else {
        h->fftSize = 2 * hopSize; h->nBins = hopSize + 1;
        h->numFilterBlocks = (int)ceilf((float)length_h / (float)hopSize);
        assert(h->numFilterBlocks >= 1);
        float* h_pad = (float*)calloc((size_t)h->numFilterBlocks * hopSize, sizeof(float));
        float* h_pad2 = (float*)calloc(2 * hopSize, sizeof(float));
        h->Hpart_f = (orc_cpx**)malloc(sizeof(orc_cpx*) * nCHout);
        h->X_n = (orc_cpx*)calloc((size_t)h->numFilterBlocks * nCHin * h->nBins, sizeof(orc_cpx));
        h->HX_n = (orc_cpx*)malloc(sizeof(orc_cpx) * h->nBins);
        h->x_pad = (float*)calloc(2 * hopSize, sizeof(float));
        h->hx_n = (float*)malloc(sizeof(float) * h->fftSize);
        h->y_n_overlap = (float*)calloc((size_t)nCHout * hopSize, sizeof(float));
        h->z_n = (float*)malloc(sizeof(float) * h->fftSize);
        orc_rfft_create(&h->hFFT, h->fftSize);
        for (int no = 0; no < nCHout; no++) {
            h->Hpart_f[no] = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)h->numFilterBlocks * nCHin * h->nBins);
            for (int ni = 0; ni < nCHin; ni++) {
                memset(h_pad, 0, sizeof(float) * (size_t)h->numFilterBlocks * hopSize);
                memcpy(h_pad, &H[((size_t)no * nCHin + ni) * length_h], sizeof(float) * length_h);
                for (int nb = 0; nb < h->numFilterBlocks; nb++) {
                    memcpy(h_pad2, &h_pad[(size_t)nb * hopSize], sizeof(float) * hopSize);
                    orc_rfft_forward(h->hFFT, h_pad2, &h->Hpart_f[no][((size_t)nb * nCHin + ni) * h->nBins]);
                }
            }
        }
        free(h_pad); free(h_pad2);
    }
    *ph = h;
}

void orc_matrixConv_destroy(void** ph)
{
    orc_mc* h = (orc_mc*)*ph; if (!h) return;
    orc_rfft_destroy(&h->hFFT);
    free(h->X_n); free(h->x_pad); free(h->z_n); free(h->hx_n); free(h->HX_n);
    if (!h->usePart) { free(h->ovrlpAddBuffer); free(h->H_f); }
    else { free(h->y_n_overlap); for (int no = 0; no < h->nCHout; no++) free(h->Hpart_f[no]); free(h->Hpart_f); }
    free(h); *ph = NULL;
}

/* saf_matrixConv_apply (saf_utility_matrixConv.c:165-236): every (partition, input) product is
 * inverse-transformed separately and the blocks are summed in the time domain, as the reference does */
void orc_matrixConv_apply(void* hh, const float* in, float* out)
{
    orc_mc* h = (orc_mc*)hh;
    const int hop = h->hopSize, nB = h->nBins, fft = h->fftSize;
    if (!h->usePart) {
        for (int ni = 0; ni < h->nCHin; ni++) {
            memcpy(&h->x_pad[(size_t)ni * fft], &in[(size_t)ni * hop], sizeof(float) * hop);
            orc_rfft_forward(h->hFFT, &h->x_pad[(size_t)ni * fft], &h->X_n[(size_t)ni * nB]);
        }
        for (int no = 0; no < h->nCHout; no++) {
            memset(h->z_n, 0, sizeof(float) * fft);
            for (int ni = 0; ni < h->nCHin; ni++) {
                const orc_cpx* Hf = &h->H_f[((size_t)no * h->nCHin + ni) * nB];
                const orc_cpx* X = &h->X_n[(size_t)ni * nB];
                for (int b = 0; b < nB; b++) { h->HX_n[b].re = Hf[b].re * X[b].re - Hf[b].im * X[b].im; h->HX_n[b].im = Hf[b].re * X[b].im + Hf[b].im * X[b].re; }
                orc_rfft_backward(h->hFFT, h->HX_n, h->hx_n);
                for (int n = 0; n < fft; n++) h->z_n[n] += h->hx_n[n];
            }
            float* ola = &h->ovrlpAddBuffer[(size_t)no * fft];
            memmove(ola, ola + hop, sizeof(float) * (size_t)(h->numOvrlpAddBlocks - 1) * hop);
            memset(ola + (size_t)(h->numOvrlpAddBlocks - 1) * hop, 0, sizeof(float) * hop);
            for (int n = 0; n < fft; n++) ola[n] += h->z_n[n];
            memcpy(&out[(size_t)no * hop], ola, sizeof(float) * hop);
        }
    } else {
        const size_t slot = (size_t)h->nCHin * nB;
        memmove(&h->X_n[slot], h->X_n, sizeof(orc_cpx) * (size_t)(h->numFilterBlocks - 1) * slot);
        for (int ni = 0; ni < h->nCHin; ni++) {
            memcpy(h->x_pad, &in[(size_t)ni * hop], sizeof(float) * hop);   /* second half stays zero */
            orc_rfft_forward(h->hFFT, h->x_pad, &h->X_n[(size_t)ni * nB]);
        }
        for (int no = 0; no < h->nCHout; no++) {
            memset(h->z_n, 0, sizeof(float) * fft);
            for (int q = 0; q < h->numFilterBlocks * h->nCHin; q++) {
                const orc_cpx* Hf = &h->Hpart_f[no][(size_t)q * nB];
                const orc_cpx* X = &h->X_n[(size_t)q * nB];
                for (int b = 0; b < nB; b++) { h->HX_n[b].re = Hf[b].re * X[b].re - Hf[b].im * X[b].im; h->HX_n[b].im = Hf[b].re * X[b].im + Hf[b].im * X[b].re; }
                orc_rfft_backward(h->hFFT, h->HX_n, h->hx_n);
                for (int n = 0; n < fft; n++) h->z_n[n] += h->hx_n[n];
            }
            for (int n = 0; n < hop; n++) out[(size_t)no * hop + n] = h->z_n[n] + h->y_n_overlap[(size_t)no * hop + n];
            memcpy(&h->y_n_overlap[(size_t)no * hop], &h->z_n[hop], sizeof(float) * hop);
        }
    }
}

/* ========================================================================== */
/*           binauraliser band MAC (binauraliser.c:252-268)                   */
/* ========================================================================== */
void orc_binaural_mac(const orc_cpx* inTF, const orc_cpx* hrtf, int nBands, int nSrc, int nSrcStride, int T, float scale, orc_cpx* outTF)
{
    memset(outTF, 0, sizeof(orc_cpx) * (size_t)nBands * 2 * T);
    for (int ch = 0; ch < nSrc; ch++)
        for (int band = 0; band < nBands; band++)
            for (int ear = 0; ear < 2; ear++) {
                const orc_cpx a = hrtf[((size_t)ch * nBands + band) * 2 + ear];
                const orc_cpx* x = &inTF[((size_t)band * nSrcStride + ch) * T];
                orc_cpx* y = &outTF[((size_t)band * 2 + ear) * T];
                for (int t = 0; t < T; t++) {     /* cblas_caxpy */
                    y[t].re += a.re * x[t].re - a.im * x[t].im;
                    y[t].im += a.re * x[t].im + a.im * x[t].re;
                }
            }
    for (size_t i = 0; i < (size_t)nBands * 2 * T; i++) { outTF[i].re *= scale; outTF[i].im *= scale; }
}
