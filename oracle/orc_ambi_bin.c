/*
 * orc_ambi_bin.c — CPU restatement of the binaural Ambisonic decoders and of the ambi_bin operator:
 *   getSHrotMtxReal (saf_sh.c:479-560, saf_sh_internal.c:151-261; Ivanic & Ruedenberg 1996/1998),
 *   getBinDecoder_LS / _LSDIFFEQ / _SPR / _TA / _MAGLS (saf_hoa_internal.c:162-623), getBinauralAmbiDecoderMtx and
 *   applyDiffCovMatching (saf_hoa.c:394-450, 502-603), diffuseFieldEqualiseHRTFs with the phase option (saf_hrir.c:173-239),
 *   ambi_bin_initCodec / ambi_bin_process (examples/src/ambi_bin/ambi_bin.c:167-480).
 *
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  Linear solves / factorisations that the reference hands to
 * single-precision LAPACK (cgesv, cpotrf, cgesvd, sgesvd) are done in float64 here.  The reference has no test for any of
 * this and its default HRIR set is absent: "parity unpinned"; tests/test_oracle_cpu.py checks closed forms (rotation
 * matrices against rotated SH, LS decoding of low-order HRTF sets, diffuse-field covariance after matching).
 */
#include "saf_oracle.h"
#include <assert.h>
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NB 133
#define HOP 128
#define MAXSH 64
#define ORC_PI 3.14159265358979323846264338327950288f
typedef float complex fc;
typedef double complex zc;
#define C2F(c) ((c).re + I * (c).im)

/* ------------------------------------------------------------------ SH rotation (Ivanic & Ruedenberg) */
static float rP(int M, int i, int l, int a, int b, float R1[3][3], const float* Rlm1)
{
    const float ri1 = R1[i + 1][2], rim1 = R1[i + 1][0], ri0 = R1[i + 1][1];
    if (b == -l) return ri1 * Rlm1[(a + l - 1) * M + 0] + rim1 * Rlm1[(a + l - 1) * M + (2 * l - 2)];
    if (b == l) return ri1 * Rlm1[(a + l - 1) * M + (2 * l - 2)] - rim1 * Rlm1[(a + l - 1) * M];
    return ri0 * Rlm1[(a + l - 1) * M + (b + l - 1)];
}
static float rV(int M, int l, int m, int n, float R1[3][3], const float* Rlm1)
{
    if (m == 0) return rP(M, 1, l, 1, n, R1, Rlm1) + rP(M, -1, l, -1, n, R1, Rlm1);
    if (m > 0) { const int d = m == 1; return rP(M, 1, l, m - 1, n, R1, Rlm1) * sqrtf(1.0f + d) - rP(M, -1, l, -m + 1, n, R1, Rlm1) * (1.0f - d); }
    { const int d = m == -1; return rP(M, 1, l, m + 1, n, R1, Rlm1) * (1.0f - (float)d) + rP(M, -1, l, -m - 1, n, R1, Rlm1) * sqrtf(1.0f + (float)d); }
}
static float rW(int M, int l, int m, int n, float R1[3][3], const float* Rlm1)
{
    if (m == 0) return 0.0f;
    if (m > 0) return rP(M, 1, l, m + 1, n, R1, Rlm1) + rP(M, -1, l, -m - 1, n, R1, Rlm1);
    return rP(M, 1, l, m - 1, n, R1, Rlm1) - rP(M, -1, l, -m + 1, n, R1, Rlm1);
}
void orc_getSHrotMtxReal(const float Rxyz[9], float* RotMtx, int L)
{
    const int M = (L + 1) * (L + 1);
    float R1[3][3];
    float* Rlm1 = (float*)calloc((size_t)M * M, sizeof(float)); float* Rl = (float*)calloc((size_t)M * M, sizeof(float));
    memset(RotMtx, 0, sizeof(float) * M * M);
    RotMtx[0] = 1.0f;
    if (L >= 1) {
        /* band 1 is the rotation matrix itself in (y, z, x) order */
        const int p[3] = { 1, 2, 0 };
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R1[i][j] = Rxyz[p[i] * 3 + p[j]];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rlm1[i * M + j] = R1[i][j]; RotMtx[(i + 1) * M + j + 1] = R1[i][j]; }
    }
    int bandIdx = 4;
    for (int l = 2; l <= L; l++) {
        for (int m = -l; m <= l; m++)
            for (int n = -l; n <= l; n++) {
                const int d = m == 0;
                const int denom = abs(n) == l ? (2 * l) * (2 * l - 1) : (l * l - n * n);
                float u = sqrtf((float)(l * l - m * m) / (float)denom);
                float v = sqrtf((float)((1 + d) * (l + abs(m) - 1) * (l + abs(m))) / (float)denom) * (float)(1 - 2 * d) * 0.5f;
                float w = sqrtf((float)((l - abs(m) - 1) * (l - abs(m))) / (float)denom) * (float)(1 - d) * (-0.5f);
                if (u != 0) u = u * rP(M, 0, l, m, n, R1, Rlm1);
                if (v != 0) v = v * rV(M, l, m, n, R1, Rlm1);
                if (w != 0) w = w * rW(M, l, m, n, R1, Rlm1);
                Rl[(m + l) * M + (n + l)] = u + v + w;
            }
        for (int i = 0; i < 2 * l + 1; i++) for (int j = 0; j < 2 * l + 1; j++) { RotMtx[(bandIdx + i) * M + bandIdx + j] = Rl[i * M + j]; Rlm1[i * M + j] = Rl[i * M + j]; }
        bandIdx += 2 * l + 1;
    }
    free(Rlm1); free(Rl);
}
void orc_yawPitchRoll2Rzyx(float yaw, float pitch, float roll, int rpy, float R[9])       /* saf_utility_geometry.c:213-270 */
{
    float Rx[3][3] = { { 1, 0, 0 }, { 0, cosf(roll), sinf(roll) }, { 0, -sinf(roll), cosf(roll) } };
    float Ry[3][3] = { { cosf(pitch), 0, -sinf(pitch) }, { 0, 1, 0 }, { sinf(pitch), 0, cosf(pitch) } };
    float Rz[3][3] = { { cosf(yaw), sinf(yaw), 0 }, { -sinf(yaw), cosf(yaw), 0 }, { 0, 0, 1 } };
    float (*R1)[3], (*R3)[3];
    if (rpy) {
        float Rxa[3][3] = { { 1, 0, 0 }, { 0, cosf(yaw), sinf(yaw) }, { 0, -sinf(yaw), cosf(yaw) } };
        float Rzg[3][3] = { { cosf(roll), sinf(roll), 0 }, { -sinf(roll), cosf(roll), 0 }, { 0, 0, 1 } };
        memcpy(Rx, Rxa, sizeof(Rx)); memcpy(Rz, Rzg, sizeof(Rz)); R1 = Rx; R3 = Rz;
    } else { R1 = Rz; R3 = Rx; }
    float T[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += Ry[i][k] * R1[k][j]; T[i][j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += R3[i][k] * T[k][j]; R[i * 3 + j] = a; }
}

/* ------------------------------------------------------------------ diffuse-field EQ / phase simplification (saf_hrir.c:173-239) */
static float mfmodf(float x, float y) { float t = fmodf(x, y); return t >= 0 ? t : t + y; }
void orc_diffuseFieldEqualiseHRTFs_full(int N, const float* itds_s, const float* centreFreq, int nBands, const float* weights, int applyEQ, int applyPhase, orc_cpx* hrtfs)
{
    if (applyEQ) orc_diffuseFieldEqualiseHRTFs(N, nBands, weights, hrtfs);
    if (applyPhase)
        for (int band = 0; band < nBands; band++)
            for (int nd = 0; nd < N; nd++) {
                const float ipd = (mfmodf(2.0f * ORC_PI * (centreFreq[band] * itds_s[nd]) + ORC_PI, 2.0f * ORC_PI) - ORC_PI) / 2.0f;
                orc_cpx* l = &hrtfs[((size_t)band * 2 + 0) * N + nd]; orc_cpx* r = &hrtfs[((size_t)band * 2 + 1) * N + nd];
                const float ml = cabsf(C2F(*l)), mr = cabsf(C2F(*r));
                const fc el = cexpf(I * ipd) * ml, er = cexpf(-I * ipd) * mr;
                l->re = crealf(el); l->im = cimagf(el); r->re = crealf(er); r->im = cimagf(er);
            }
}

/* ------------------------------------------------------------------ decoders */
/* solve the real symmetric positive definite G [n x n] against complex right-hand sides B [n x m] (utility_cglslv on a
 * real-valued complex matrix), float64 Cholesky */
static void spd_solve(int n, const float* G, const fc* B, int m, fc* X)
{
    double* L = (double*)calloc((size_t)n * n, sizeof(double));
    for (int j = 0; j < n; j++) {
        double d = G[j * n + j];
        for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
        assert(d > 0.0);
        L[j * n + j] = sqrt(d);
        for (int i = j + 1; i < n; i++) { double s = G[i * n + j]; for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k]; L[i * n + j] = s / L[j * n + j]; }
    }
    zc* y = (zc*)malloc(sizeof(zc) * n);
    for (int c = 0; c < m; c++) {
        for (int i = 0; i < n; i++) { zc s = B[i * m + c]; for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
        for (int i = n - 1; i >= 0; i--) { zc s = y[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * y[k]; y[i] = s / L[i * n + i]; }
        for (int i = 0; i < n; i++) X[i * m + c] = (fc)y[i];
    }
    free(L); free(y);
}
static void weights_vec(const float* weights, int N, float* w) { for (int i = 0; i < N; i++) w[i] = weights ? weights[i] : 1.0f / (float)N; }
/* common part of LS / LSDIFFEQ / TA / MAGLS: Y [nSH][N] (getRSH), YW = Y diag(w), G = YW Y^T */
static void ls_prep(int order, const float* dirs_deg, int N, const float* weights, float** Y, float** YW, float** G)
{
    const int nSH = (order + 1) * (order + 1);
    *Y = (float*)malloc(sizeof(float) * nSH * N); *YW = (float*)malloc(sizeof(float) * nSH * N); *G = (float*)malloc(sizeof(float) * nSH * nSH);
    float* w = (float*)malloc(sizeof(float) * N);
    weights_vec(weights, N, w);
    orc_getRSH(order, dirs_deg, N, *Y);
    for (int i = 0; i < nSH; i++) for (int j = 0; j < N; j++) (*YW)[i * N + j] = (*Y)[i * N + j] * w[j];
    for (int i = 0; i < nSH; i++) for (int j = 0; j < nSH; j++) { float s = 0.0f; for (int k = 0; k < N; k++) s += (*YW)[i * N + k] * (*Y)[j * N + k]; (*G)[i * nSH + j] = s; }
    free(w);
}
/* B = G^-1 (YW H^H)  for H [2][N]; decoder rows = B^H */
static void ls_band(int nSH, int N, const float* YW, const float* G, const fc* H, fc* B)
{
    fc* R = (fc*)malloc(sizeof(fc) * nSH * 2);
    for (int i = 0; i < nSH; i++) for (int e = 0; e < 2; e++) { fc s = 0.0f; for (int k = 0; k < N; k++) s += YW[i * N + k] * conjf(H[e * N + k]); R[i * 2 + e] = s; }
    spd_solve(nSH, G, R, 2, B);
    free(R);
}
static int band_cutoff_idx(const float* freqVector, int nBands)
{
    float minVal = 2.23e10f; int bc = 0;
    for (int band = 0; band < nBands; band++) if (minVal > fabsf(freqVector[band] - 1.5e3f)) { minVal = fabsf(freqVector[band] - 1.5e3f); bc = band; }
    return bc;
}
static void diffuse_cov(int N, const float* w, const fc* H, fc C[2][2])     /* H diag(w) H^H */
{
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { fc s = 0.0f; for (int k = 0; k < N; k++) s += (H[a * N + k] * w[k]) * conjf(H[b * N + k]); C[a][b] = s; }
}

void orc_getBinauralAmbiDecoderMtx(const orc_cpx* hrtfs_, const float* dirs_deg, int N, int nBands, int method, int order, const float* freqVector,
                                   const float* itd_s, const float* weights, int enableDiffCovMatching, int enableMaxRE, orc_cpx* decMtx_)
{
    const int nSH = (order + 1) * (order + 1);
    const fc* hrtfs = (const fc*)hrtfs_; fc* decMtx = (fc*)decMtx_;
    float *Y, *YW, *G;
    ls_prep(order, dirs_deg, N, weights, &Y, &YW, &G);
    fc* B = (fc*)malloc(sizeof(fc) * nSH * 2); fc* Hm = (fc*)malloc(sizeof(fc) * 2 * N);
    float* w = (float*)malloc(sizeof(float) * N); weights_vec(weights, N, w);
    (void)itd_s;
    if (method == 3) {          /* BINAURAL_DECODER_SPR (saf_hoa_internal.c:332-430) */
        int Nh_max = (int)(sqrtf((float)N) - 1.0f); if (Nh_max > 20) Nh_max = 20;
        float* rad = (float*)malloc(sizeof(float) * 2 * N);
        for (int i = 0; i < N; i++) { rad[i * 2] = dirs_deg[i * 2] * (ORC_PI / 180.0f); rad[i * 2 + 1] = ORC_PI / 2.0f - dirs_deg[i * 2 + 1] * (ORC_PI / 180.0f); }
        /* checkCondNumberSHTReal (saf_sh.c:884-960): condition number of Y_n^T W Y_n per order (orthonormal SH) */
        const int nSHmax = (Nh_max + 1) * (Nh_max + 1);
        float* YN = (float*)malloc(sizeof(float) * nSHmax * N);
        orc_getSHreal(Nh_max, rad, N, YN);
        int Nh = 0;
        float* YY = (float*)malloc(sizeof(float) * nSHmax * nSHmax); float* sv = (float*)malloc(sizeof(float) * nSHmax);
        for (int n = 0; n <= Nh_max; n++) {
            const int ns = (n + 1) * (n + 1);
            for (int i = 0; i < ns; i++) for (int j = 0; j < ns; j++) { float s = 0.0f; for (int k = 0; k < N; k++) s += YN[i * N + k] * (weights ? weights[k] : 1.0f) * YN[j * N + k]; YY[i * ns + j] = s; }
            orc_singular_values(YY, ns, ns, sv);
            float mx = sv[0], mn = sv[0];
            for (int i = 1; i < ns; i++) { if (sv[i] > mx) mx = sv[i]; if (sv[i] < mn) mn = sv[i]; }
            if (mx / (mn + 2.23e-7f) < 100.0f) Nh = n;
        }
        assert(Nh >= order);
        const int nSHh = (Nh + 1) * (Nh + 1);
        float* Ynh = (float*)malloc(sizeof(float) * nSHh * N);
        orc_getRSH(Nh, dirs_deg, N, Ynh);
        char tname[64]; int K, d1;
        snprintf(tname, sizeof(tname), "Tdesign_degree_%d_dirs_deg", 2 * order);
        const float* td = orc_table(tname, &K, &d1); assert(td);
        float* Ytd = (float*)malloc(sizeof(float) * nSHh * K);
        orc_getRSH(Nh, td, K, Ytd);
        /* W (Y_nh^T Y_td) with W = w / 4pi (or 1/N) */
        float* WYY = (float*)malloc(sizeof(float) * N * K);
        for (int i = 0; i < N; i++) for (int j = 0; j < K; j++) { float s = 0.0f; for (int k = 0; k < nSHh; k++) s += Ynh[k * N + i] * Ytd[k * K + j]; WYY[i * K + j] = s * (weights ? weights[i] / (4.0f * ORC_PI) : 1.0f / (float)N); }
        fc* Htd = (fc*)malloc(sizeof(fc) * 2 * K);
        for (int band = 0; band < nBands; band++) {
            const fc* H = &hrtfs[(size_t)band * 2 * N];
            for (int e = 0; e < 2; e++) for (int j = 0; j < K; j++) { fc s = 0.0f; for (int k = 0; k < N; k++) s += H[e * N + k] * WYY[k * K + j]; Htd[e * K + j] = s; }
            for (int i = 0; i < nSH; i++) for (int e = 0; e < 2; e++) {
                fc s = 0.0f; for (int j = 0; j < K; j++) s += Ytd[i * K + j] * conjf(Htd[e * K + j]);
                decMtx[(size_t)band * 2 * nSH + e * nSH + i] = conjf(s) * (1.0f / (float)K);
            }
        }
        free(rad); free(YN); free(YY); free(sv); free(Ynh); free(Ytd); free(WYY); free(Htd);
    } else {
        const int bc = (method == 4 || method == 5) ? band_cutoff_idx(freqVector, nBands) : 0;
        for (int band = 0; band < nBands; band++) {
            const fc* H = &hrtfs[(size_t)band * 2 * N];
            float Gh = 1.0f;
            if (method == 4 && band >= bc) {
                /* BINAURAL_DECODER_TA (:432-523): the phase term of the reference is exp(0 * itd/2) = 1, i.e. the HRTFs of the
                 * cut-off band are used unmodified above the cut-off */
                ls_band(nSH, N, YW, G, &hrtfs[(size_t)bc * 2 * N], B);
            } else if (method == 5 && band > bc) {
                /* BINAURAL_DECODER_MAGLS (:525-623): magnitudes of this band with the phase the previous band's decoder gives */
                const fc* Dp = &decMtx[(size_t)(band - 1) * 2 * nSH];
                for (int e = 0; e < 2; e++) for (int k = 0; k < N; k++) {
                    fc s = 0.0f; for (int i = 0; i < nSH; i++) s += Dp[e * nSH + i] * Y[i * N + k];
                    Hm[e * N + k] = cabsf(H[e * N + k]) * cexpf(I * atan2f(cimagf(s), crealf(s)));
                }
                ls_band(nSH, N, YW, G, Hm, B);
            } else {
                ls_band(nSH, N, YW, G, H, B);
                if (method == 2) {      /* BINAURAL_DECODER_LSDIFFEQ (:230-330): diffuse-field gain of the order-limited HRTFs */
                    for (int e = 0; e < 2; e++) for (int k = 0; k < N; k++) { fc s = 0.0f; for (int i = 0; i < nSH; i++) s += conjf(B[i * 2 + e]) * Y[i * N + k]; Hm[e * N + k] = s; }
                    fc Cr[2][2], Cl[2][2];
                    diffuse_cov(N, w, H, Cr); diffuse_cov(N, w, Hm, Cl);
                    Gh = (sqrtf(crealf(Cr[0][0]) / (crealf(Cl[0][0]) + 2.23e-7f)) + sqrtf(crealf(Cr[1][1]) / (crealf(Cl[1][1]) + 2.23e-7f))) / 2.0f;
                }
            }
            for (int i = 0; i < nSH; i++) for (int e = 0; e < 2; e++) decMtx[(size_t)band * 2 * nSH + e * nSH + i] = conjf(B[i * 2 + e]) * Gh;
        }
    }
    if (enableMaxRE) {          /* saf_hoa.c:427-444 */
        float* a = (float*)malloc(sizeof(float) * nSH * nSH);
        orc_getMaxREweights(order, 1, a);
        for (int band = 0; band < nBands; band++) for (int e = 0; e < 2; e++) for (int i = 0; i < nSH; i++) decMtx[(size_t)band * 2 * nSH + e * nSH + i] *= a[i * nSH + i];
        free(a);
    }
    if (enableDiffCovMatching) {        /* applyDiffCovMatching (saf_hoa.c:502-603); Nyquist band skipped */
        fc* Ha = (fc*)malloc(sizeof(fc) * 2 * N);
        for (int band = 0; band < nBands - 1; band++) {
            const fc* H = &hrtfs[(size_t)band * 2 * N]; fc* D = &decMtx[(size_t)band * 2 * nSH];
            fc Cr[2][2], Ca[2][2];
            diffuse_cov(N, w, H, Cr);
            for (int e = 0; e < 2; e++) for (int k = 0; k < N; k++) { fc s = 0.0f; for (int i = 0; i < nSH; i++) s += D[e * nSH + i] * Y[i * N + k]; Ha[e * N + k] = s; }
            diffuse_cov(N, w, Ha, Ca);
            /* upper Cholesky factors X^H X = C (diagonals forced real), float64 */
            zc X[2][2] = { { 0 } }, Xa[2][2] = { { 0 } };
            X[0][0] = sqrt((double)crealf(Cr[0][0])); X[0][1] = (zc)Cr[0][1] / X[0][0]; X[1][1] = sqrt((double)crealf(Cr[1][1]) - creal(X[0][1] * conj(X[0][1])));
            Xa[0][0] = sqrt((double)crealf(Ca[0][0])); Xa[0][1] = (zc)Ca[0][1] / Xa[0][0]; Xa[1][1] = sqrt((double)crealf(Ca[1][1]) - creal(Xa[0][1] * conj(Xa[0][1])));
            /* A = Xa^H X = U S V^H;  V U^H = (unitary polar factor of A)^H = (A^H A)^(-1/2) A^H */
            zc A[2][2], P[2][2];
            for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { A[i][j] = 0; for (int k = 0; k < 2; k++) A[i][j] += conj(Xa[k][i]) * X[k][j]; }
            for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { P[i][j] = 0; for (int k = 0; k < 2; k++) P[i][j] += conj(A[k][i]) * A[k][j]; }
            /* inverse square root of the 2 x 2 Hermitian positive definite P via its eigen-decomposition */
            const double a = creal(P[0][0]), d = creal(P[1][1]); const zc b = P[0][1];
            const double tr = a + d, det = a * d - creal(b * conj(b));
            const double sdet = sqrt(det), t = sqrt(tr + 2.0 * sdet);          /* sqrt(P) = (P + sqrt(det) I) / t */
            zc S[2][2] = { { (a + sdet) / t, b / t }, { conj(b) / t, (d + sdet) / t } }, Si[2][2];
            const zc dS = S[0][0] * S[1][1] - S[0][1] * S[1][0];
            Si[0][0] = S[1][1] / dS; Si[0][1] = -S[0][1] / dS; Si[1][0] = -S[1][0] / dS; Si[1][1] = S[0][0] / dS;
            zc VU[2][2], VUX[2][2], Mx[2][2];
            for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { VU[i][j] = 0; for (int k = 0; k < 2; k++) VU[i][j] += Si[i][k] * conj(A[j][k]); }
            for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { VUX[i][j] = 0; for (int k = 0; k < 2; k++) VUX[i][j] += VU[i][k] * X[k][j]; }
            /* M = Xa^-1 VUX (Xa upper triangular) */
            for (int j = 0; j < 2; j++) { Mx[1][j] = VUX[1][j] / Xa[1][1]; Mx[0][j] = (VUX[0][j] - Xa[0][1] * Mx[1][j]) / Xa[0][0]; }
            fc Dn[2 * MAXSH];
            for (int e = 0; e < 2; e++) for (int i = 0; i < nSH; i++) { zc s = 0; for (int k = 0; k < 2; k++) s += conj(Mx[k][e]) * (zc)D[k * nSH + i]; Dn[e * nSH + i] = (fc)s; }
            memcpy(D, Dn, sizeof(fc) * 2 * nSH);
        }
        free(Ha);
    }
    free(Y); free(YW); free(G); free(B); free(Hm); free(w);
}

/* ------------------------------------------------------------------ truncation EQ (saf_hoa.c:269-324, ambi_bin.c:311-364) */
/* spherical Bessel functions after Zhang & Jin, "Computation of Special Functions" (SPHJ, SPHY, MSTA1, MSTA2, ENVJ), the
 * routines behind bessel_jn_ALL / hankel_hn2_ALL (saf_utility_bessel.c:40-353, 662-721, 1125-1190) */
static double envj(int n, double x) { return 0.5 * log(6.28 * n) - n * log(1.36 * x / n); }
static int msta_iter(double a0, int n0, double obj)
{
    double f0 = envj(n0, a0) - obj; int n1 = n0 + 5; double f1 = envj(n1, a0) - obj; int nn = 0;
    for (int it = 1; it <= 20; it++) {
        nn = n1 - (int)((double)(n1 - n0) / (1.0 - f0 / f1));
        const double f = envj(nn, a0) - obj;
        if (abs(nn - n1) < 1) break;
        n0 = n1; f0 = f1; n1 = nn; f1 = f;
    }
    return nn;
}
static int msta1(double x, int mp) { const double a0 = fabs(x); return msta_iter(a0, (int)(floor(1.1 * a0) + 1.0), mp); }
static int msta2(double x, int n, int mp)
{
    const double a0 = fabs(x), hmp = 0.5 * mp, ejn = envj(n, a0);
    return (ejn <= hmp ? msta_iter(a0, (int)floor(1.1 * a0), mp) : msta_iter(a0, n, hmp + ejn)) + 10;
}
void orc_sphj(int N, double X, int* NM, double* SJ, double* DJ)
{
    *NM = N;
    if (fabs(X) < 1e-80) { for (int k = 0; k <= N; k++) { SJ[k] = 0.0; DJ[k] = 0.0; } SJ[0] = 1.0; if (N > 0) DJ[1] = 0.333333333333333; return; }
    SJ[0] = sin(X) / X; if (N >= 1) SJ[1] = (SJ[0] - cos(X)) / X;
    if (N >= 2) {
        const double SA = SJ[0], SB = SJ[1];
        int M = msta1(X, 200);
        if (M < N) *NM = M; else M = msta2(X, N, 15);
        int i = 0;
        while (M < 0) { M = msta2(X, N, 14 - i); i++; if (i == 14) M = 0; }
        double F0 = 0.0, F1 = 1.0 - 100, F = 1;
        for (int K = M; K > -1; K--) { F = (2.0 * K + 3.0) * F1 / X - F0; if (K <= *NM) SJ[K] = F; F0 = F1; F1 = F; }
        double CS = 1;
        if (fabs(SA) > fabs(SB)) CS = SA / F;
        if (fabs(SA) <= fabs(SB)) CS = SB / F0;
        for (int K = 0; K <= *NM; K++) SJ[K] *= CS;
    }
    DJ[0] = (cos(X) - sin(X) / X) / X;
    for (int K = 1; K <= *NM; K++) DJ[K] = SJ[K - 1] - (K + 1.0) * SJ[K] / X;
}
void orc_sphy(int N, double X, int* NM, double* SY, double* DY)
{
    *NM = N;
    if (X < 1e-20) { for (int k = 0; k <= N; k++) { SY[k] = -1.0e+300; DY[k] = 1e+300; } return; }
    SY[0] = -cos(X) / X; if (N >= 1) SY[1] = (SY[0] - sin(X)) / X;
    double F0 = SY[0], F1 = N >= 1 ? SY[1] : 0.0; int K;
    for (K = 2; K <= N; K++) { const double F = (2.0 * K - 1.0) * F1 / X - F0; SY[K] = F; if (fabs(F) >= 1e+300) break; F0 = F1; F1 = F; }
    *NM = K - 1;
    DY[0] = (sin(X) + cos(X) / X) / X;
    for (K = 1; K <= *NM; K++) DY[K] = SY[K - 1] - (K + 1.0) * SY[K] / X;
}
/* sphModalCoeffs, ARRAY_CONSTRUCTION_RIGID (saf_sh.c:2018-2048): b_N [nBands][order+1] */
static void modal_rigid(int order, const double* kr, int nBands, zc* b_N)
{
    const int S = order + 1;
    double* jn = (double*)calloc((size_t)nBands * S, sizeof(double)); double* djn = (double*)calloc((size_t)nBands * S, sizeof(double));
    zc* hn = (zc*)calloc((size_t)nBands * S, sizeof(zc)); zc* dhn = (zc*)calloc((size_t)nBands * S, sizeof(zc));
    double* tj = (double*)calloc(S, sizeof(double)); double* tdj = (double*)calloc(S, sizeof(double)); double* ty = (double*)calloc(S, sizeof(double)); double* tdy = (double*)calloc(S, sizeof(double));
    int maxN = 1000000000;
    for (int i = 0; i < nBands; i++) {
        if (kr[i] <= 1e-15) {       /* the reference writes these defaults into row 0 (saf_utility_bessel.c:679-688, 1146-1152) */
            memset(jn, 0, sizeof(double) * S); memset(djn, 0, sizeof(double) * S); memset(hn, 0, sizeof(zc) * S); memset(dhn, 0, sizeof(zc) * S);
            jn[0] = 1.0; if (order > 0) djn[1] = 1.0 / 3.0; hn[0] = 1.0;
            continue;
        }
        int n1, n2;
        orc_sphj(order, kr[i], &n1, tj, tdj); if (n1 < maxN) maxN = n1;
        for (int n = 0; n <= n1; n++) { jn[(size_t)i * S + n] = tj[n]; djn[(size_t)i * S + n] = tdj[n]; }
        orc_sphy(order, kr[i], &n2, ty, tdy); if (n2 < maxN) maxN = n2;
        for (int n = 0; n <= (n1 < n2 ? n1 : n2); n++) { hn[(size_t)i * S + n] = tj[n] - I * ty[n]; dhn[(size_t)i * S + n] = tdj[n] - I * tdy[n]; }
    }
    if (maxN > order) maxN = order;
    memset(b_N, 0, sizeof(zc) * (size_t)nBands * S);
    for (int i = 0; i < nBands; i++)
        for (int n = 0; n <= maxN; n++) {
            if (n == 0 && kr[i] <= 1e-20) b_N[(size_t)i * S + n] = 4.0 * M_PI;
            else if (kr[i] <= 1e-20) b_N[(size_t)i * S + n] = 0.0;
            else b_N[(size_t)i * S + n] = cpow(I, (double)n) * 4.0 * M_PI * (jn[(size_t)i * S + n] - (djn[(size_t)i * S + n] / dhn[(size_t)i * S + n]) * hn[(size_t)i * S + n]);
        }
    free(jn); free(djn); free(hn); free(dhn); free(tj); free(tdj); free(ty); free(tdy);
}
void orc_truncationEQ(const float* w_n, int order_truncated, int order_target, const double* kr, int nBands, float softThreshold, float* gain)
{
    zc* bt = (zc*)malloc(sizeof(zc) * (size_t)nBands * (order_target + 1)); zc* bq = (zc*)malloc(sizeof(zc) * (size_t)nBands * (order_truncated + 1));
    modal_rigid(order_target, kr, nBands, bt); modal_rigid(order_truncated, kr, nBands, bq);
    const float clipFactor = powf(10.0f, softThreshold / 20.0f);
    for (int b = 0; b < nBands; b++) {
        double pt = 0.0, pq = 0.0;
        for (int n = 0; n <= order_target; n++) pt += (2.0 * n + 1.0) * pow(cabs(bt[(size_t)b * (order_target + 1) + n]), 2.0);
        for (int n = 0; n <= order_truncated; n++) pq += w_n[n] * (2.0 * n + 1.0) * pow(cabs(bq[(size_t)b * (order_truncated + 1) + n]), 2.0);
        pt = 1.0 / (4.0 * ORC_PI) * sqrt(pt); pq = 1.0 / (4.0 * ORC_PI) * sqrt(pq);
        float g = (float)(pt / (pq + 2.23e-13));
        g = g / clipFactor;
        if (g > 1.0f) g = 1.0f + tanhf(g - 1.0f);
        gain[b] = g * clipFactor;
    }
    free(bt); free(bq);
}
/* beamWeightsMaxEV (saf_sh.c:751-776) */
void orc_beamWeightsMaxEV(int N, float* b_n)
{
    float norm = 0.0f;
    const double x = cos(2.4068f / ((double)N + 1.51));
    double Pm2 = 1.0, Pm1 = x;
    for (int n = 0; n <= N; n++) {
        double P = n == 0 ? 1.0 : (n == 1 ? x : ((2.0 * n - 1.0) * x * Pm1 - (n - 1.0) * Pm2) / (double)n);
        if (n >= 2) { Pm2 = Pm1; Pm1 = P; }
        b_n[n] = sqrtf((2.0f * (float)n + 1.0f) / (4.0f * ORC_PI)) * (float)P;
        norm += sqrtf((2.0f * (float)n + 1.0f) / (4.0f * ORC_PI)) * b_n[n];
    }
    for (int n = 0; n <= N; n++) b_n[n] /= norm;
}

/* ------------------------------------------------------------------ ambi_bin operator */
typedef struct {
    int F, T, fs, order, new_order, nSH, codecReady, reinit_hrtfs, recalcRot;
    void* hSTFT;
    float freqVector[NB];
    float* set_hrirs; float* set_dirs; int set_N, set_len, set_fs;
    float* itds_s; float* weights; orc_cpx* hrtf_fb; int haveWeights;
    orc_cpx* M_dec; orc_cpx* M_dec_rot;     /* [NB][2][MAXSH] */
    int preProc, chOrdering, norm, enableMaxRE, enableDiffM, enableRot, enableTruncEQ, method, useRPY, flip[3];
    float ypr[3];
} orc_abin;

void orc_ambi_bin_create(void** ph, int frameSize)
{
    orc_abin* p = (orc_abin*)calloc(1, sizeof(orc_abin));
    p->F = frameSize; p->T = frameSize / HOP;
    p->preProc = 2; p->chOrdering = 1; p->norm = 2; p->enableMaxRE = 1; p->enableTruncEQ = 1; p->method = 5; p->order = p->new_order = 1; p->nSH = 4;
    p->recalcRot = 1; p->reinit_hrtfs = 1;
    p->M_dec = (orc_cpx*)calloc((size_t)NB * 2 * MAXSH, sizeof(orc_cpx)); p->M_dec_rot = (orc_cpx*)calloc((size_t)NB * 2 * MAXSH, sizeof(orc_cpx));
    *ph = p;
}
void orc_ambi_bin_destroy(void** ph)
{
    orc_abin* p = (orc_abin*)*ph; if (!p) return;
    if (p->hSTFT) orc_afSTFT_destroy(&p->hSTFT);
    free(p->set_hrirs); free(p->set_dirs); free(p->itds_s); free(p->weights); free(p->hrtf_fb); free(p->M_dec); free(p->M_dec_rot); free(p); *ph = NULL;
}
void orc_ambi_bin_setHRIRs(void* h, const float* hrirs, const float* dirs_deg, int N, int len, int fs)
{
    orc_abin* p = (orc_abin*)h;
    free(p->set_hrirs); free(p->set_dirs);
    p->set_hrirs = (float*)malloc(sizeof(float) * (size_t)N * 2 * len); memcpy(p->set_hrirs, hrirs, sizeof(float) * (size_t)N * 2 * len);
    p->set_dirs = (float*)malloc(sizeof(float) * (size_t)N * 2); memcpy(p->set_dirs, dirs_deg, sizeof(float) * (size_t)N * 2);
    p->set_N = N; p->set_len = len; p->set_fs = fs; p->reinit_hrtfs = 1; p->codecReady = 0;
}
void orc_ambi_bin_init(void* h, int sampleRate)       /* ambi_bin.c:147-165 */
{
    orc_abin* p = (orc_abin*)h;
    if (p->fs != sampleRate) { p->fs = sampleRate; p->reinit_hrtfs = 1; p->codecReady = 0; }
    orc_afSTFT_getCentreFreqs(p->hSTFT, (float)p->fs, NB, p->freqVector);
    p->recalcRot = 1;
}
void orc_ambi_bin_initCodec(void* h)                  /* ambi_bin.c:167-378 */
{
    orc_abin* p = (orc_abin*)h;
    if (p->codecReady) return;
    const int order = p->new_order, nSH = (order + 1) * (order + 1);
    if (!p->hSTFT) orc_afSTFT_create(&p->hSTFT, nSH, 2, HOP, 0, 1, ORC_AFSTFT_BANDS_CH_TIME);
    else if (p->nSH != nSH) { orc_afSTFT_channelChange(p->hSTFT, nSH, 2); orc_afSTFT_clearBuffers(p->hSTFT); }
    p->nSH = nSH;
    const int N = p->set_N;
    assert(p->set_hrirs);
    if (p->reinit_hrtfs) {
        p->itds_s = (float*)realloc(p->itds_s, sizeof(float) * N);
        orc_estimateITDs(p->set_hrirs, N, p->set_len, p->set_fs, p->itds_s);
        p->hrtf_fb = (orc_cpx*)realloc(p->hrtf_fb, sizeof(orc_cpx) * (size_t)NB * 2 * N);
        orc_afSTFT_FIRtoFilterbankCoeffs(p->set_hrirs, N, 2, p->set_len, HOP, 0, 1, p->hrtf_fb);
        p->haveWeights = N <= 1000;
        if (p->haveWeights) { p->weights = (float*)realloc(p->weights, sizeof(float) * N); orc_getVoronoiWeights(p->set_dirs, N, p->weights); }
        orc_diffuseFieldEqualiseHRTFs_full(N, p->itds_s, p->freqVector, NB, p->haveWeights ? p->weights : NULL,
                                           p->preProc == 2 || p->preProc == 4, p->preProc == 3 || p->preProc == 4, p->hrtf_fb);
        p->reinit_hrtfs = 0;
    }
    orc_cpx* dec = (orc_cpx*)calloc((size_t)NB * 2 * nSH, sizeof(orc_cpx));
    orc_getBinauralAmbiDecoderMtx(p->hrtf_fb, p->set_dirs, N, NB, p->method, order, p->freqVector, p->itds_s, p->haveWeights ? p->weights : NULL,
                                  p->enableDiffM, p->enableMaxRE, dec);
    if (p->enableTruncEQ && p->method == 1 && p->preProc != 3 && p->preProc != 4) {        /* ambi_bin.c:311-364 */
        double kr[NB]; float w_n[8], eq[NB];
        for (int k = 0; k < NB; k++) kr[k] = 2.0 * M_PI / 343.0 * (double)p->freqVector[k] * 0.085;
        for (int n = 0; n <= order; n++) w_n[n] = 1.0f;
        if (p->enableMaxRE) {
            float c[8];
            orc_beamWeightsMaxEV(order, c);
            for (int n = 0; n <= order; n++) w_n[n] = c[n] / sqrtf((float)(2 * n + 1) / (4.0f * ORC_PI));
            const float w0 = w_n[0];
            for (int n = 0; n <= order; n++) w_n[n] /= w0;
        }
        orc_truncationEQ(w_n, order, 42, kr, NB, 9.0f, eq);
        for (int b = 0; b < NB; b++) for (int i = 0; i < 2 * nSH; i++) { dec[(size_t)b * 2 * nSH + i].re *= eq[b]; dec[(size_t)b * 2 * nSH + i].im *= eq[b]; }
    }
    memset(p->M_dec, 0, sizeof(orc_cpx) * (size_t)NB * 2 * MAXSH);
    for (int b = 0; b < NB; b++) for (int e = 0; e < 2; e++) for (int j = 0; j < nSH; j++) p->M_dec[((size_t)b * 2 + e) * MAXSH + j] = dec[((size_t)b * 2 + e) * nSH + j];
    free(dec);
    p->order = order;
    p->codecReady = 1;
}
void orc_ambi_bin_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)   /* ambi_bin.c:380-480 */
{
    orc_abin* p = (orc_abin*)h;
    const int F = p->F, T = p->T, order = p->order, nSH = (order + 1) * (order + 1);
    if (nSamples != F || !p->codecReady) { for (int ch = 0; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F); return; }
    float* td = (float*)calloc((size_t)MAXSH * F, sizeof(float));
    for (int i = 0; i < (nSH < nInputs ? nSH : nInputs); i++) memcpy(&td[(size_t)i * F], inputs[i], sizeof(float) * F);
    if (p->chOrdering == 2) orc_convertHOAChannelConvention(td, order, F, 2, 1);
    if (p->norm == 2) orc_convertHOANormConvention(td, order, F, 2, 1);
    else if (p->norm == 3) orc_convertHOANormConvention(td, order, F, 3, 1);
    orc_cpx* tf = (orc_cpx*)calloc((size_t)NB * MAXSH * T, sizeof(orc_cpx));
    orc_afSTFT_forward_knownDimensions(p->hSTFT, td, F, MAXSH, T, tf);
    if (order > 0 && p->enableRot && p->recalcRot) {
        float R[9]; float* Mr = (float*)malloc(sizeof(float) * nSH * nSH);
        orc_yawPitchRoll2Rzyx(p->ypr[0], p->ypr[1], p->ypr[2], p->useRPY, R);
        orc_getSHrotMtxReal(R, Mr, order);
        for (int b = 0; b < NB; b++) for (int e = 0; e < 2; e++) for (int j = 0; j < nSH; j++) {
            fc s = 0.0f; for (int k = 0; k < nSH; k++) s += C2F(p->M_dec[((size_t)b * 2 + e) * MAXSH + k]) * Mr[k * nSH + j];
            p->M_dec_rot[((size_t)b * 2 + e) * MAXSH + j].re = crealf(s); p->M_dec_rot[((size_t)b * 2 + e) * MAXSH + j].im = cimagf(s);
        }
        free(Mr); p->recalcRot = 0;
    }
    const orc_cpx* M = p->enableRot ? p->M_dec_rot : p->M_dec;
    orc_cpx* out = (orc_cpx*)calloc((size_t)NB * 2 * T, sizeof(orc_cpx));
    for (int b = 0; b < NB; b++) for (int e = 0; e < 2; e++) for (int t = 0; t < T; t++) {
        fc s = 0.0f; for (int k = 0; k < nSH; k++) s += C2F(M[((size_t)b * 2 + e) * MAXSH + k]) * C2F(tf[((size_t)b * MAXSH + k) * T + t]);
        out[((size_t)b * 2 + e) * T + t].re = crealf(s); out[((size_t)b * 2 + e) * T + t].im = cimagf(s);
    }
    float* bt = (float*)calloc((size_t)2 * F, sizeof(float));
    orc_afSTFT_backward_knownDimensions(p->hSTFT, out, F, 2, T, bt);
    int ch;
    for (ch = 0; ch < (2 < nOutputs ? 2 : nOutputs); ch++) memcpy(outputs[ch], &bt[(size_t)ch * F], sizeof(float) * F);
    for (; ch < nOutputs; ch++) memset(outputs[ch], 0, sizeof(float) * F);
    free(td); free(tf); free(out); free(bt);
}
#define PA orc_abin* p = (orc_abin*)h
void orc_ambi_bin_setInputOrderPreset(void* h, int o) { PA; if (p->order != o) { p->new_order = o; p->codecReady = 0; } }
void orc_ambi_bin_setDecodingMethod(void* h, int m) { PA; p->method = m; p->codecReady = 0; }
void orc_ambi_bin_setChOrder(void* h, int v) { PA; if (v != 2 || p->order == 1) p->chOrdering = v; }
void orc_ambi_bin_setNormType(void* h, int v) { PA; if (v != 3 || p->order == 1) p->norm = v; }
void orc_ambi_bin_setEnableMaxRE(void* h, int s) { PA; if (p->enableMaxRE != s) { p->enableMaxRE = s; p->codecReady = 0; } }
void orc_ambi_bin_setEnableDiffuseMatching(void* h, int s) { PA; if (p->enableDiffM != s) { p->enableDiffM = s; p->codecReady = 0; } }
void orc_ambi_bin_setEnableTruncationEQ(void* h, int s) { PA; if (p->enableTruncEQ != s) { p->enableTruncEQ = s; p->codecReady = 0; } }
void orc_ambi_bin_setHRIRsPreProc(void* h, int s) { PA; if (p->preProc != s) { p->preProc = s; p->reinit_hrtfs = 1; p->codecReady = 0; } }
void orc_ambi_bin_setEnableRotation(void* h, int s) { PA; p->enableRot = s; }
void orc_ambi_bin_setYaw(void* h, float v) { PA; p->ypr[0] = (p->flip[0] ? -1.0f : 1.0f) * (v * ORC_PI / 180.0f); p->recalcRot = 1; }
void orc_ambi_bin_setPitch(void* h, float v) { PA; p->ypr[1] = (p->flip[1] ? -1.0f : 1.0f) * (v * ORC_PI / 180.0f); p->recalcRot = 1; }
void orc_ambi_bin_setRoll(void* h, float v) { PA; p->ypr[2] = (p->flip[2] ? -1.0f : 1.0f) * (v * ORC_PI / 180.0f); p->recalcRot = 1; }
void orc_ambi_bin_setRPYflag(void* h, int s) { PA; p->useRPY = s; }
const orc_cpx* orc_ambi_bin_getDecMtx(void* h) { PA; return p->M_dec; }

/* getBinauralAmbiDecoderFilters (saf_hoa.c:452-500) */
void orc_getBinauralAmbiDecoderFilters(const orc_cpx* hrtfs, const float* dirs_deg, int N, int fftSize, float fs, int method, int order,
                                       const float* itd_s, const float* weights, int diffMatching, int maxRE, float* decFilters)
{
    const int nBins = fftSize / 2 + 1, nSH = (order + 1) * (order + 1);
    float* fv = (float*)malloc(sizeof(float) * nBins);
    for (int k = 0; k < nBins; k++) fv[k] = (float)k * fs / (float)fftSize;
    orc_cpx* dec = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nBins * 2 * nSH);
    orc_cpx* b = (orc_cpx*)malloc(sizeof(orc_cpx) * nBins);
    orc_getBinauralAmbiDecoderMtx(hrtfs, dirs_deg, N, nBins, method, order, fv, itd_s, weights, diffMatching, maxRE, dec);
    void* h = NULL;
    orc_rfft_create(&h, fftSize);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < nSH; j++) {
            for (int k = 0; k < nBins; k++) b[k] = dec[(size_t)k * 2 * nSH + i * nSH + j];
            orc_rfft_backward(h, b, decFilters + ((size_t)i * nSH + j) * fftSize);
        }
    orc_rfft_destroy(&h);
    free(fv); free(dec); free(b);
}
