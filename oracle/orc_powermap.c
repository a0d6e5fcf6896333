/*
 * orc_powermap.c — CPU restatement of the powermap operator (all map modes; the adaptive ones live in orc_pmaps.c)
 * (examples/src/powermap/powermap.c:185-380, powermap_internal.c:46-136; generatePWDmap saf_sh.c:1544-1584).
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  No reference test covers powermap: parity "unpinned"
 * by reference-side data; pinned by closed forms in tests/test_oracle_cpu.py.
 */
#include "saf_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define NB 133
#define HOP 128
#define MAXSH 64
#define NSLOTS 2
#define NSH(o) (((o) + 1) * ((o) + 1))

typedef struct {
    int F, T;
    float fs, freqVector[NB];
    void* hSTFT; int stftCh;
    float* inFIFO; int FIFO_idx;
    orc_cpx* Cx;                     /* [NB][64*64] (row stride nSH of the current master order, as the reference indexes it) */
    int masterOrder, new_masterOrder, analysisOrderPerBand[NB], nSources, pmap_mode, chOrdering, norm, dispWidth;
    float pmapEQ[NB], covAvgCoeff, pmapAvgCoeff;
    int codecReady, recalcPmap, pmapReady, dispSlotIdx;
    /* codec pars */
    const float* grid_dirs_deg; int grid_nDirs;
    float* Y_grid[7];
    float* interp_dirs_deg; float* interp_table; int interp_nDirs, interp_nTri;
    float* pmap; float* prev_pmap; float* pmap_grid[NSLOTS];
    float minVal, maxVal;
} orc_pm;

void orc_powermap_create(void** ph, int frameSize)
{
    orc_pm* p = (orc_pm*)calloc(1, sizeof(orc_pm));
    p->F = frameSize; p->T = frameSize / HOP;
    p->masterOrder = p->new_masterOrder = 1;
    for (int b = 0; b < NB; b++) { p->analysisOrderPerBand[b] = 1; p->pmapEQ[b] = 1.0f; }
    p->covAvgCoeff = 0.0f; p->pmapAvgCoeff = 0.666f; p->nSources = 1; p->pmap_mode = 4 /* PM_MODE_MUSIC */;
    p->chOrdering = 1; p->norm = 2; p->dispWidth = 140; p->recalcPmap = 1;
    orc_afSTFT_create(&p->hSTFT, MAXSH, 0, HOP, 0, 1, ORC_AFSTFT_BANDS_CH_TIME); p->stftCh = MAXSH;      /* powermap.c:59 */
    p->inFIFO = (float*)calloc((size_t)MAXSH * frameSize, sizeof(float));
    p->Cx = (orc_cpx*)calloc((size_t)NB * MAXSH * MAXSH, sizeof(orc_cpx));
    *ph = p;
}
void orc_powermap_destroy(void** ph)
{
    orc_pm* p = (orc_pm*)*ph; if (!p) return;
    orc_afSTFT_destroy(&p->hSTFT);
    free(p->inFIFO); free(p->Cx);
    for (int n = 0; n < 7; n++) free(p->Y_grid[n]);
    free(p->interp_dirs_deg); free(p->interp_table); free(p->pmap); free(p->prev_pmap);
    for (int i = 0; i < NSLOTS; i++) free(p->pmap_grid[i]);
    free(p); *ph = NULL;
}
/* powermap_init (powermap.c:134-151) */
void orc_powermap_init(void* h, float sampleRate)
{
    orc_pm* p = (orc_pm*)h;
    p->fs = sampleRate;
    orc_afSTFT_getCentreFreqs(p->hSTFT, sampleRate, NB, p->freqVector);
    memset(p->Cx, 0, sizeof(orc_cpx) * (size_t)NB * MAXSH * MAXSH);
    if (p->prev_pmap) memset(p->prev_pmap, 0, sizeof(float) * p->grid_nDirs);
    p->pmapReady = 0; p->dispSlotIdx = 0;
}
/* powermap_initCodec (powermap.c:153-181) = initTFT + initAna (powermap_internal.c:46-136) */
void orc_powermap_initCodec(void* h)
{
    orc_pm* p = (orc_pm*)h;
    if (p->codecReady) return;
    const int nSH = NSH(p->masterOrder), new_nSH = NSH(p->new_masterOrder);
    if (nSH != new_nSH) {
        orc_afSTFT_channelChange(p->hSTFT, new_nSH, 0); orc_afSTFT_clearBuffers(p->hSTFT); p->stftCh = new_nSH;
        memset(p->Cx, 0, sizeof(orc_cpx) * (size_t)NB * MAXSH * MAXSH);
    }
    const int order = p->new_masterOrder;
    int d0, d1;
    p->grid_dirs_deg = orc_table("geosphere_ico_9_0_dirs_deg", &d0, &d1);
    p->grid_nDirs = d0;
    const int G = p->grid_nDirs;
    float* Y = (float*)malloc(sizeof(float) * (size_t)NSH(order) * G);
    orc_getRSH(order, p->grid_dirs_deg, G, Y);
    for (int n = 1; n <= order; n++) {
        const int ns = NSH(n); const float sc = 1.0f / (float)ns;
        p->Y_grid[n - 1] = (float*)realloc(p->Y_grid[n - 1], sizeof(float) * (size_t)ns * G);
        for (size_t i = 0; i < (size_t)ns * G; i++) p->Y_grid[n - 1][i] = Y[i] * sc;
    }
    free(Y);
    const float hfov = 360.0f, aspect = 2.0f;
    const int N_azi = p->dispWidth, N_ele = (int)((float)p->dispWidth / aspect + 0.5f);
    const float vfov = hfov / aspect;
    float* gx = (float*)malloc(sizeof(float) * N_azi); float* gy = (float*)malloc(sizeof(float) * N_ele);
    { float fi = -hfov / 2.0f; for (int i = 0; i < N_azi; fi += hfov / N_azi, i++) gx[i] = fi; }
    { float fi = -vfov / 2.0f; for (int i = 0; i < N_ele; fi += vfov / N_ele, i++) gy[i] = fi; }
    p->interp_dirs_deg = (float*)realloc(p->interp_dirs_deg, sizeof(float) * 2 * N_azi * N_ele);
    for (int i = 0; i < N_ele; i++) for (int j = 0; j < N_azi; j++) { p->interp_dirs_deg[(i * N_azi + j) * 2] = gx[j]; p->interp_dirs_deg[(i * N_azi + j) * 2 + 1] = gy[i]; }
    free(gx); free(gy);
    free(p->interp_table); p->interp_table = NULL;
    orc_generateVBAPgainTable3D_srcs(p->interp_dirs_deg, N_azi * N_ele, p->grid_dirs_deg, G, 0, 0, 0.0f, &p->interp_table, &p->interp_nDirs, &p->interp_nTri);
    for (int i = 0; i < p->interp_nDirs; i++) {          /* VBAPgainTable2InterpTable (saf_vbap.c:369-388) */
        float s = 0.0f;
        for (int j = 0; j < G; j++) s += p->interp_table[(size_t)i * G + j];
        for (int j = 0; j < G; j++) p->interp_table[(size_t)i * G + j] /= s;
    }
    p->pmap = (float*)realloc(p->pmap, sizeof(float) * G);
    free(p->prev_pmap); p->prev_pmap = (float*)calloc(G, sizeof(float));
    for (int i = 0; i < NSLOTS; i++) { free(p->pmap_grid[i]); p->pmap_grid[i] = (float*)calloc(p->interp_nDirs, sizeof(float)); }
    p->masterOrder = order;
    p->codecReady = 1;
}

/* one full frame: conventions, afSTFT, covariance update, optional map (powermap.c:232-371) */
static void analyse_frame(orc_pm* p)
{
    const int F = p->F, T = p->T, masterOrder = p->masterOrder, nSH = NSH(masterOrder);
    float* td = (float*)calloc((size_t)p->stftCh * F, sizeof(float));
    for (int ch = 0; ch < nSH; ch++) memcpy(&td[(size_t)ch * F], &p->inFIFO[(size_t)ch * F], sizeof(float) * F);
    if (p->chOrdering == 2) orc_convertHOAChannelConvention(td, masterOrder, F, 2, 1);
    if (p->norm == 2) orc_convertHOANormConvention(td, masterOrder, F, 2, 1);
    else if (p->norm == 3) orc_convertHOANormConvention(td, masterOrder, F, 3, 1);
    orc_cpx* tf = (orc_cpx*)calloc((size_t)NB * p->stftCh * T, sizeof(orc_cpx));
    orc_afSTFT_forward_knownDimensions(p->hSTFT, td, F, p->stftCh, T, tf);
    const float a = p->covAvgCoeff < 0.45f ? p->covAvgCoeff : 0.45f;
    for (int band = 0; band < NB; band++) {
        orc_cpx* C = &p->Cx[(size_t)band * MAXSH * MAXSH];
        for (int i = 0; i < nSH; i++)
            for (int j = 0; j < nSH; j++) {
                float re = 0.0f, im = 0.0f;
                const orc_cpx* xi = &tf[((size_t)band * p->stftCh + i) * T]; const orc_cpx* xj = &tf[((size_t)band * p->stftCh + j) * T];
                for (int t = 0; t < T; t++) { re += xi[t].re * xj[t].re + xi[t].im * xj[t].im; im += xi[t].im * xj[t].re - xi[t].re * xj[t].im; }
                orc_cpx* c = &C[i * nSH + j];
                c->re = c->re * a; c->im = c->im * a;                         /* cblas_sscal */
                c->re += (1.0f - a) * re; c->im += (1.0f - a) * im;           /* cblas_saxpy */
            }
    }
    free(td); free(tf);
    if (p->recalcPmap == 1) {
        p->recalcPmap = 0; p->pmapReady = 0;
        int maxOrder = 1;
        for (int i = 0; i < NB; i++) { int o = p->analysisOrderPerBand[i] < masterOrder ? p->analysisOrderPerBand[i] : masterOrder; if (o > maxOrder) maxOrder = o; }
        const int nM = NSH(maxOrder), G = p->grid_nDirs;
        orc_cpx* Cg = (orc_cpx*)calloc((size_t)nM * nM, sizeof(orc_cpx));
        for (int band = 0; band < NB; band++) {
            int ob = p->analysisOrderPerBand[band] < masterOrder ? p->analysisOrderPerBand[band] : masterOrder; if (ob < 1) ob = 1;
            const int ns = NSH(ob);
            float eq = p->pmapEQ[band]; eq = eq < 0.0f ? 0.0f : (eq > 2.0f ? 2.0f : eq);
            const orc_cpx* C = &p->Cx[(size_t)band * MAXSH * MAXSH];
            for (int i = 0; i < ns; i++) for (int j = 0; j < ns; j++) { Cg[i * nM + j].re += C[i * nSH + j].re * (1e3f * eq); Cg[i * nM + j].im += C[i * nSH + j].im * (1e3f * eq); }
        }
        /* generate powermap (powermap.c:291-341) */
        const float* Yg = p->Y_grid[maxOrder - 1];
        float trace = 0.0f;
        for (int i = 0; i < nM; i++) trace += Cg[i * nM + i].re;
        switch (p->pmap_mode) {
            default:
            case 1: {   /* PM_MODE_PWD: generatePWDmap (saf_sh.c:1544-1584): pmap[d] = Re( y_d^T (C y_d) ) */
                for (int d = 0; d < G; d++) {
                    float accr = 0.0f;
                    for (int i = 0; i < nM; i++) {
                        float cr = 0.0f;
                        for (int j = 0; j < nM; j++) cr += Cg[i * nM + j].re * Yg[(size_t)j * G + d];
                        accr += Yg[(size_t)i * G + d] * cr;
                    }
                    p->pmap[d] = accr;
                }
            } break;
            case 2: if (trace > 1e-8f) orc_generateMVDRmap(maxOrder, Cg, Yg, G, 8.0f, p->pmap, NULL); else memset(p->pmap, 0, sizeof(float) * G); break;
            case 3: if (trace > 1e-8f) orc_generateCroPaCLCMVmap(maxOrder, Cg, Yg, G, 8.0f, 0.0f, p->pmap); else memset(p->pmap, 0, sizeof(float) * G); break;
            case 4: case 5: if (trace > 1e-8f) orc_generateMUSICmap(maxOrder, Cg, Yg, p->nSources, G, p->pmap_mode == 5, p->pmap); else memset(p->pmap, 0, sizeof(float) * G); break;
            case 6: case 7: if (trace > 1e-8f) orc_generateMinNormMap(maxOrder, Cg, Yg, p->nSources, G, p->pmap_mode == 7, p->pmap); else memset(p->pmap, 0, sizeof(float) * G); break;
        }
        free(Cg);
        for (int i = 0; i < G; i++) p->pmap[i] = (1.0f - p->pmapAvgCoeff) * p->pmap[i] + p->pmapAvgCoeff * p->prev_pmap[i];
        memcpy(p->prev_pmap, p->pmap, sizeof(float) * G);
        float* out = p->pmap_grid[p->dispSlotIdx];
        for (int i = 0; i < p->interp_nDirs; i++) {
            float s = 0.0f;
            for (int j = 0; j < G; j++) s += p->interp_table[(size_t)i * G + j] * p->pmap[j];
            out[i] = s;
        }
        float mn = out[0], mx = out[0];
        for (int i = 1; i < p->interp_nDirs; i++) { if (out[i] < mn) mn = out[i]; if (out[i] > mx) mx = out[i]; }
        p->minVal = mn; p->maxVal = mx;
        for (int i = 0; i < p->interp_nDirs; i++) out[i] = (out[i] - mn) / (mx - mn + 1e-11f);
        p->dispSlotIdx++; if (p->dispSlotIdx >= NSLOTS) p->dispSlotIdx = 0;
        p->pmapReady = 1;
    }
}

/* powermap_analysis (powermap.c:185-380): sample-wise FIFO */
void orc_powermap_analysis(void* h, const float* const* inputs, int nInputs, int nSamples, int isPlaying)
{
    orc_pm* p = (orc_pm*)h;
    const int nSH = NSH(p->masterOrder), F = p->F;
    for (int s = 0; s < nSamples; s++) {
        int ch;
        for (ch = 0; ch < (nInputs < nSH ? nInputs : nSH); ch++) p->inFIFO[(size_t)ch * F + p->FIFO_idx] = inputs[ch][s];
        for (; ch < nSH; ch++) p->inFIFO[(size_t)ch * F + p->FIFO_idx] = 0.0f;
        p->FIFO_idx++;
        if (p->FIFO_idx >= F && p->codecReady && isPlaying) { p->FIFO_idx = 0; analyse_frame(p); }
        else if (p->FIFO_idx >= F) p->FIFO_idx = 0;
    }
}
#define PP orc_pm* p = (orc_pm*)h
void orc_powermap_setPowermapMode(void* h, int m) { PP; p->pmap_mode = m; if (p->prev_pmap) memset(p->prev_pmap, 0, sizeof(float) * p->grid_nDirs); }
void orc_powermap_setMasterOrder(void* h, int o) { PP; if (p->new_masterOrder != o) { p->new_masterOrder = o; p->codecReady = 0; }
    if (p->new_masterOrder != 1 && p->chOrdering == 2) p->chOrdering = 1;
    if (p->new_masterOrder != 1 && p->norm == 3) p->norm = 2; }
void orc_powermap_setCovAvgCoeff(void* h, float a) { PP; p->covAvgCoeff = a < 0.0f ? 0.0f : (a > 0.99999999f ? 0.99999999f : a); }
void orc_powermap_setAnaOrder(void* h, int o, int band) { PP; p->analysisOrderPerBand[band] = o < 1 ? 1 : (o > p->new_masterOrder ? p->new_masterOrder : o); }
void orc_powermap_setAnaOrderAllBands(void* h, int o) { PP; for (int b = 0; b < NB; b++) p->analysisOrderPerBand[b] = o < 1 ? 1 : (o > p->new_masterOrder ? p->new_masterOrder : o); }
void orc_powermap_setPowermapEQ(void* h, float v, int band) { PP; p->pmapEQ[band] = v; }
void orc_powermap_setChOrder(void* h, int v) { PP; if (v != 2 || p->new_masterOrder == 1) p->chOrdering = v; }
void orc_powermap_setNormType(void* h, int v) { PP; if (v != 3 || p->new_masterOrder == 1) p->norm = v; }
void orc_powermap_setPowermapAvgCoeff(void* h, float v) { PP; p->pmapAvgCoeff = v < 0.0f ? 0.0f : (v > 0.99999999f ? 0.99999999f : v); }
void orc_powermap_setNumSources(void* h, int n) { PP; p->nSources = n; }      /* powermap.c:418-422 */
void orc_powermap_requestPmapUpdate(void* h) { PP; p->recalcPmap = 1; }
int orc_powermap_getPmap(void* h, const float** grid_dirs, const float** pmap, int* nDirs)
{
    PP;
    if (p->codecReady && p->pmapReady) { *grid_dirs = p->interp_dirs_deg; *pmap = p->pmap_grid[p->dispSlotIdx - 1 < 0 ? NSLOTS - 1 : p->dispSlotIdx - 1]; *nDirs = p->interp_nDirs; }
    return p->pmapReady;
}
const orc_cpx* orc_powermap_getCx(void* h) { PP; return p->Cx; }
const float* orc_powermap_getRawPmap(void* h) { PP; return p->pmap; }
int orc_powermap_getGridNDirs(void* h) { PP; return p->grid_nDirs; }
