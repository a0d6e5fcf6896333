/*
 * orc_beamformer.c — CPU restatement of the beamformer example (examples/src/beamformer/beamformer.c:30-339) and of the
 * SH helpers it uses: getSHcomplex (saf_sh.c:333-382), complex2realSHMtx / complex2realCoeffs (:384-475),
 * rotateAxisCoeffsComplex / Real (:839-882), beamWeightsCardioid2Spherical / Hypercardioid2Spherical (:716-745).
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  The reference holds no test for this operator: "parity unpinned" by
 * reference-side data; tests/test_oracle_cpu.py checks the closed-form beam patterns.
 */
#include "saf_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <complex.h>

#define BPI 3.14159265358979323846264338327950288f
#define BPId 3.14159265358979323846264338327950288
#define NMAX 64

static double factd(int n) { double f = 1.0; for (int i = 2; i <= n; i++) f *= (double)i; return f; }

/* getSHcomplex, one direction (azimuth, inclination) */
static void sh_complex(int order, float azi, float incl, float complex* Y)
{
    double ci = cos((double)incl);
    double* L = (double*)malloc(sizeof(double) * (order + 1));
    int idx = 0;
    for (int n = 0; n <= order; n++) {
        orc_unnorm_legendreP(n, &ci, 1, L);                   /* includes the Condon-Shortley phase */
        for (int m = -n, j = 0; m <= n; m++, j++) {
            const int am = abs(m);
            const double nr = sqrt((2.0 * n + 1.0) * factd(n - am) / (4.0 * BPId * factd(n + am)));
            double complex y = cexp(I * (double)am * (double)azi) * (nr * L[am]);
            if (m < 0) y = conj(y) * pow(-1.0, (double)am);
            Y[idx + j] = (float)creal(y) + I * (float)cimag(y);
        }
        idx += 2 * n + 1;
    }
    free(L);
}

/* complex2realSHMtx (saf_sh.c:384-414) */
static void c2r_mtx(int order, float complex* T)
{
    const int nSH = (order + 1) * (order + 1);
    memset(T, 0, sizeof(float complex) * nSH * nSH);
    T[0] = 1.0f;
    int idx = 1;
    for (int n = 1, q = 1; n <= order; n++) {
        idx += 2 * n + 1;
        for (int m = -n, p = 0; m <= n; m++, q++, p++) {
            if (m < 0) { T[q * nSH + q] = I * (1.0f / sqrtf(2.0f)); T[(idx - p - 1) * nSH + q] = 1.0f / sqrtf(2.0f); }
            else if (m == 0) T[q * nSH + q] = 1.0f;
            else { T[q * nSH + q] = powf(-1.0f, (float)m) / sqrtf(2.0f); T[(idx - p - 1) * nSH + q] = -I * (powf(-1.0f, (float)abs(m)) / sqrtf(2.0f)); }
        }
    }
}

/* rotateAxisCoeffsReal (saf_sh.c:839-882) */
void orc_rotateAxisCoeffsReal(int order, const float* c_n, float theta_0, float phi_0, float* c_nm)
{
    const int nSH = (order + 1) * (order + 1);
    float complex* Y = (float complex*)malloc(sizeof(float complex) * nSH);
    float complex* c = (float complex*)malloc(sizeof(float complex) * nSH);
    float complex* T = (float complex*)malloc(sizeof(float complex) * nSH * nSH);
    sh_complex(order, phi_0, theta_0, Y);
    for (int n = 0, q = 0; n <= order; n++)
        for (int m = -n; m <= n; m++, q++) c[q] = conjf(Y[q]) * (sqrtf(4.0f * BPI / (2.0f * (float)n + 1.0f)) * c_n[n]);
    c2r_mtx(order, T);
    for (int i = 0; i < nSH; i++) {
        float complex a = 0.0f;
        for (int j = 0; j < nSH; j++) a += conjf(T[i * nSH + j]) * c[j];
        c_nm[i] = crealf(a);
    }
    free(Y); free(c); free(T);
}

void orc_beamWeightsCardioid2Spherical(int N, float* b_n)
{
    for (int n = 0; n < N + 1; n++)
        b_n[n] = sqrtf(4.0f * BPI * (2.0f * (float)n + 1.0f)) * (float)factd(N) * (float)factd(N + 1) / ((float)factd(N + n + 1) * (float)factd(N - n)) / ((float)N + 1.0f);
}

void orc_beamWeightsHypercardioid2Spherical(int N, float* b_n)
{
    float dirs[2] = { 0.0f, 0.0f };
    float* c = (float*)malloc(sizeof(float) * (N + 1) * (N + 1));
    orc_getSHreal(N, dirs, 1, c);
    for (int n = 0; n < N + 1; n++) b_n[n] = c[(n + 1) * (n + 1) - n - 1] * 4.0f * BPI / powf((float)N + 1.0f, 2.0f);
    free(c);
}

typedef struct {
    int F, order, nBeams, beamType, chOrdering, norm;
    float dirs[NMAX][2];
    int recalc[NMAX];
    float W[NMAX][NMAX], prevW[NMAX][NMAX];
    float* prevIn; float* fadeIn; float* fadeOut;
} orc_bf;

void orc_beamformer_create(void** ph, int F)
{
    orc_bf* p = (orc_bf*)calloc(1, sizeof(orc_bf));
    p->F = F; p->order = 1; p->nBeams = 1; p->beamType = 2; p->chOrdering = 1; p->norm = 2;
    const float* def = orc_table("default_LScoords64_rad", NULL, NULL);
    for (int i = 0; i < NMAX; i++) {
        p->dirs[i][0] = def[i * 2] * 180.0f / BPI;
        p->dirs[i][1] = (def[i * 2 + 1] - BPI / 2.0f) < -BPI / 2.0f ? (BPI / 2.0f + def[i * 2 + 1]) : (def[i * 2 + 1] - BPI / 2.0f);
        p->dirs[i][1] *= 180.0f / BPI;
        p->recalc[i] = 1;
    }
    p->prevIn = (float*)calloc((size_t)NMAX * F, sizeof(float));
    p->fadeIn = (float*)malloc(sizeof(float) * F); p->fadeOut = (float*)malloc(sizeof(float) * F);
    *ph = p;
}
void orc_beamformer_destroy(void** ph) { orc_bf* p = (orc_bf*)*ph; if (!p) return; free(p->prevIn); free(p->fadeIn); free(p->fadeOut); free(p); *ph = NULL; }

void orc_beamformer_init(void* h, int fs)       /* beamformer.c:71-92 */
{
    orc_bf* p = (orc_bf*)h; (void)fs;
    memset(p->W, 0, sizeof(p->W)); memset(p->prevW, 0, sizeof(p->prevW));
    memset(p->prevIn, 0, sizeof(float) * (size_t)NMAX * p->F);
    for (int ch = 0; ch < NMAX; ch++) p->recalc[ch] = 1;
    for (int i = 1; i <= p->F; i++) { p->fadeIn[i - 1] = (float)i * 1.0f / (float)p->F; p->fadeOut[i - 1] = 1.0f - p->fadeIn[i - 1]; }
}

void orc_beamformer_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)   /* beamformer.c:94-187 */
{
    orc_bf* p = (orc_bf*)h;
    const int F = p->F, order = p->order, nSH = (order + 1) * (order + 1), nBeams = p->nBeams;
    if (nSamples != F) { for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F); return; }
    float* in = (float*)calloc((size_t)NMAX * F, sizeof(float));
    float* out = (float*)calloc((size_t)NMAX * F, sizeof(float));
    for (int i = 0; i < (nSH < nInputs ? nSH : nInputs); i++) memcpy(in + (size_t)i * F, inputs[i], sizeof(float) * F);
    if (p->chOrdering == 2 && order == 1) {                      /* FuMa WXYZ -> ACN WYZX */
        float* t = (float*)malloc(sizeof(float) * 4 * F);
        memcpy(t, in, sizeof(float) * 4 * F);
        memcpy(in + F, t + 2 * F, sizeof(float) * F); memcpy(in + 2 * F, t + 3 * F, sizeof(float) * F); memcpy(in + 3 * F, t + F, sizeof(float) * F);
        free(t);
    }
    if (p->norm == 2) { for (int n = 0; n <= order; n++) for (int ch = n * n; ch < (n + 1) * (n + 1); ch++) for (int s = 0; s < F; s++) in[(size_t)ch * F + s] *= sqrtf(2.0f * (float)n + 1.0f); }
    else if (p->norm == 3) { for (int s = 0; s < F; s++) in[s] *= sqrtf(2.0f); for (int ch = 1; ch < 4; ch++) for (int s = 0; s < F; s++) in[(size_t)ch * F + s] *= sqrtf(3.0f); }
    int mix = 0;
    for (int bi = 0; bi < nBeams; bi++) {
        if (!p->recalc[bi]) continue;
        float c_n[8], w[NMAX];
        memset(p->W[bi], 0, sizeof(float) * NMAX);
        if (p->beamType == 1) orc_beamWeightsCardioid2Spherical(order, c_n);
        else if (p->beamType == 2) orc_beamWeightsHypercardioid2Spherical(order, c_n);
        else orc_beamWeightsMaxEV(order, c_n);
        orc_rotateAxisCoeffsReal(order, c_n, BPI / 2.0f - p->dirs[bi][1] * BPI / 180.0f, p->dirs[bi][0] * BPI / 180.0f, w);
        memcpy(p->W[bi], w, sizeof(float) * nSH);
        p->recalc[bi] = 0; mix = 1;
    }
    for (int b = 0; b < nBeams; b++)
        for (int s = 0; s < F; s++) {
            float a = 0.0f;
            for (int j = 0; j < nSH; j++) a += p->W[b][j] * p->prevIn[(size_t)j * F + s];
            out[(size_t)b * F + s] = a;
        }
    if (mix) {
        for (int b = 0; b < nBeams; b++)
            for (int s = 0; s < F; s++) {
                float t = 0.0f;
                for (int j = 0; j < nSH; j++) t += p->prevW[b][j] * p->prevIn[(size_t)j * F + s];
                const float fi = p->fadeIn[s] * out[(size_t)b * F + s], fo = p->fadeOut[s] * t;
                out[(size_t)b * F + s] = fi + fo;
            }
        memcpy(p->prevW, p->W, sizeof(p->W));
    }
    memcpy(p->prevIn, in, sizeof(float) * (size_t)NMAX * F);
    int ch;
    for (ch = 0; ch < (nBeams < nOutputs ? nBeams : nOutputs); ch++) memcpy(outputs[ch], out + (size_t)ch * F, sizeof(float) * F);
    for (; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    free(in); free(out);
}

#define BP orc_bf* p = (orc_bf*)h
void orc_beamformer_setBeamOrder(void* h, int v)
{
    BP;
    p->order = v < 1 ? 1 : (v > 7 ? 7 : v);
    for (int ch = 0; ch < NMAX; ch++) p->recalc[ch] = 1;
    if (p->order != 1 && p->chOrdering == 2) p->chOrdering = 1;
    if (p->order != 1 && p->norm == 3) p->norm = 2;
}
void orc_beamformer_setBeamAzi_deg(void* h, int i, float v) { BP; if (v > 180.0f) v = -360.0f + v; v = v < -180.0f ? -180.0f : (v > 180.0f ? 180.0f : v); p->dirs[i][0] = v; p->recalc[i] = 1; }
void orc_beamformer_setBeamElev_deg(void* h, int i, float v) { BP; v = v < -90.0f ? -90.0f : (v > 90.0f ? 90.0f : v); p->dirs[i][1] = v; p->recalc[i] = 1; }
void orc_beamformer_setNumBeams(void* h, int n) { BP; if (p->nBeams != n) { p->nBeams = n; for (int ch = 0; ch < NMAX; ch++) p->recalc[ch] = 1; } }
void orc_beamformer_setChOrder(void* h, int o) { BP; if (o != 2 || p->order == 1) p->chOrdering = o; }
void orc_beamformer_setNormType(void* h, int t) { BP; if (t != 3 || p->order == 1) p->norm = t; }
void orc_beamformer_setBeamType(void* h, int id) { BP; p->beamType = id; for (int ch = 0; ch < NMAX; ch++) p->recalc[ch] = 1; }
