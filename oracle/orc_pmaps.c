/*
 * orc_pmaps.c — CPU restatement of the adaptive / sub-space activity-map generators of saf_sh
 * (framework/modules/saf_sh/saf_sh.c:1586-1858): generateMVDRmap, generateCroPaCLCMVmap, generateMUSICmap,
 * generateMinNormMap, as called by powermap_analysis (powermap.c:294-341).
 *
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).
 * The reference factorises with single-precision LAPACK (cposv / cgesv / cheev / cgeev), which is not available here;
 * the restatement uses the same mathematics with float64 factorisations (Cholesky, LU with partial pivoting, cyclic
 * Jacobi) and float32 everywhere else.  Parity status: the reference has no test for these four generators: "parity
 * unpinned"; tests/test_oracle_cpu.py checks closed forms (source directions are the arg-max; MVDR of a single plane
 * wave; distortionless weights).
 * Implementation-defined in the reference and fixed here: generateMinNormMap takes its eigenvectors from cgeev, whose
 * eigenvalue ORDER is unspecified for this input; here they are sorted by descending eigenvalue (what utility_cseig
 * does for MUSIC) and phase-normalised like cgeev does (unit norm, largest component real and positive).
 */
#include "saf_oracle.h"
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double complex zc;

/* generatePWDmap (saf_sh.c:1544-1584) with complex weights: pmap[d] = Re( w_d^T C w_d )  (no conjugation) */
static void pwd_map_w(int nSH, const orc_cpx* Cx, const float complex* W /* [nSH][G] */, int G, float* pmap)
{
    for (int d = 0; d < G; d++) {
        float complex acc = 0.0f;
        for (int i = 0; i < nSH; i++) {
            float complex cw = 0.0f;
            for (int j = 0; j < nSH; j++) cw += (Cx[i * nSH + j].re + I * Cx[i * nSH + j].im) * W[(size_t)j * G + d];
            acc += W[(size_t)i * G + d] * cw;
        }
        pmap[d] = crealf(acc);
    }
}

/* lower Cholesky factor of the Hermitian positive definite A (row-major), float64; returns 0 on success */
static int chol(int n, const zc* A, zc* L)
{
    memset(L, 0, sizeof(zc) * n * n);
    for (int j = 0; j < n; j++) {
        double d = creal(A[j * n + j]);
        for (int k = 0; k < j; k++) d -= creal(L[j * n + k] * conj(L[j * n + k]));
        if (!(d > 0.0)) return 1;
        const double ljj = sqrt(d);
        L[j * n + j] = ljj;
        for (int i = j + 1; i < n; i++) {
            zc s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= L[i * n + k] * conj(L[j * n + k]);
            L[i * n + j] = s / ljj;
        }
    }
    return 0;
}
static void chol_solve(int n, const zc* L, zc* b)        /* b <- (L L^H)^-1 b */
{
    for (int i = 0; i < n; i++) { zc s = b[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k]; b[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { zc s = b[i]; for (int k = i + 1; k < n; k++) s -= conj(L[k * n + i]) * b[k]; b[i] = s / L[i * n + i]; }
}

/* generateMVDRmap (saf_sh.c:1586-1644) */
void orc_generateMVDRmap(int order, const orc_cpx* Cx, const float* Y_grid /* [nSH][G] real */, int G, float regPar, float* pmap, orc_cpx* w_out)
{
    const int nSH = (order + 1) * (order + 1);
    float tr = 0.0f;
    for (int i = 0; i < nSH; i++) tr += Cx[i * nSH + i].re;
    tr /= (float)nSH;
    zc* A = (zc*)malloc(sizeof(zc) * nSH * nSH); zc* L = (zc*)malloc(sizeof(zc) * nSH * nSH); zc* z = (zc*)malloc(sizeof(zc) * nSH);
    for (int i = 0; i < nSH * nSH; i++) A[i] = (double)Cx[i].re + I * (double)Cx[i].im;
    for (int i = 0; i < nSH; i++) A[i * nSH + i] = (double)(Cx[i * nSH + i].re + regPar * tr) + I * (double)Cx[i * nSH + i].im;   /* craddf in float */
    float complex* W = (float complex*)malloc(sizeof(float complex) * (size_t)nSH * G);
    if (chol(nSH, A, L)) { memset(pmap, 0, sizeof(float) * G); if (w_out) memset(w_out, 0, sizeof(orc_cpx) * (size_t)nSH * G); free(A); free(L); free(z); free(W); return; }
    for (int d = 0; d < G; d++) {
        for (int j = 0; j < nSH; j++) z[j] = Y_grid[(size_t)j * G + d];
        chol_solve(nSH, L, z);
        float complex den = 0.0f;                                   /* utility_cvvdot(Y, conj(invCx_Y), NO_CONJ) */
        for (int j = 0; j < nSH; j++) den += Y_grid[(size_t)j * G + d] * conjf((float complex)z[j]);
        for (int j = 0; j < nSH; j++) W[(size_t)j * G + d] = (float complex)z[j] / den;
    }
    pwd_map_w(nSH, Cx, W, G, pmap);
    if (w_out) for (size_t i = 0; i < (size_t)nSH * G; i++) { w_out[i].re = crealf(W[i]); w_out[i].im = cimagf(W[i]); }
    free(A); free(L); free(z); free(W);
}

/* generateCroPaCLCMVmap (saf_sh.c:1650-1752) */
void orc_generateCroPaCLCMVmap(int order, const orc_cpx* Cx, const float* Y_grid, int G, float regPar, float lambda, float* pmap)
{
    const int nSH = (order + 1) * (order + 1);
    float* mvdr = (float*)malloc(sizeof(float) * G);
    orc_cpx* wq = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nSH * G);
    orc_generateMVDRmap(order, Cx, Y_grid, G, regPar, mvdr, wq);
    float complex* W = (float complex*)malloc(sizeof(float complex) * (size_t)nSH * G);
    for (size_t i = 0; i < (size_t)nSH * G; i++) W[i] = wq[i].re + I * wq[i].im;
    float tr = 0.0f;
    for (int i = 0; i < nSH; i++) tr += Cx[i * nSH + i].re;
    tr /= (float)nSH;
    zc* A = (zc*)malloc(sizeof(zc) * nSH * nSH); zc* L = (zc*)malloc(sizeof(zc) * nSH * nSH);
    for (int i = 0; i < nSH * nSH; i++) A[i] = (double)Cx[i].re + I * (double)Cx[i].im;
    for (int i = 0; i < nSH; i++) A[i * nSH + i] = (double)(Cx[i * nSH + i].re + regPar * tr) + I * (double)Cx[i * nSH + i].im;
    const int ok = !chol(nSH, A, L);
    zc* a0 = (zc*)malloc(sizeof(zc) * nSH); zc* a1 = (zc*)malloc(sizeof(zc) * nSH);
    for (int d = 0; d < G && ok; d++) {
        /* constraint matrix A = [y, y .* diag(Cx)] (:1705-1708) and Cx_d^-1 A */
        float complex c0[64], c1[64];
        for (int j = 0; j < nSH; j++) {
            c0[j] = Y_grid[(size_t)j * G + d];
            c1[j] = c0[j] * (Cx[j * nSH + j].re + I * Cx[j * nSH + j].im);
            a0[j] = c0[j]; a1[j] = c1[j];
        }
        chol_solve(nSH, L, a0); chol_solve(nSH, L, a1);
        /* A^H conj(Cx_d^-1 A) (:1712-1717): the reference conjugates the solve before the product */
        float complex M[2][2] = { { 0, 0 }, { 0, 0 } };
        for (int j = 0; j < nSH; j++) {
            const float complex s0 = conjf((float complex)a0[j]), s1 = conjf((float complex)a1[j]);
            M[0][0] += conjf(c0[j]) * s0; M[0][1] += conjf(c0[j]) * s1;
            M[1][0] += conjf(c1[j]) * s0; M[1][1] += conjf(c1[j]) * s1;
        }
        /* w_LCMV_s = M^-1 [row k = (Cx_d^-1 A)(:,k)^T]  (2 x nSH), then wo = w_LCMV_s^T b with b = [1 0]^T: first row (:1718-1726) */
        const float complex det = M[0][0] * M[1][1] - M[0][1] * M[1][0];
        float complex xs = 0.0f;                                   /* cross-spectrum wo . (Cx y)  (:1729-1731) */
        for (int j = 0; j < nSH; j++) {
            const float complex wo = (M[1][1] * (float complex)a0[j] - M[0][1] * (float complex)a1[j]) / det;
            float complex cy = 0.0f;
            for (int k = 0; k < nSH; k++) cy += (Cx[j * nSH + k].re + I * Cx[j * nSH + k].im) * Y_grid[(size_t)k * G + d];
            xs += wo * cy;
        }
        float S = cabsf(xs); if (mvdr[d] < S) S = mvdr[d];
        float Gn = sqrtf(S / (mvdr[d] + 2.23e-10f));
        if (Gn < lambda) Gn = lambda;
        for (int j = 0; j < nSH; j++) W[(size_t)j * G + d] *= Gn;
    }
    pwd_map_w(nSH, Cx, W, G, pmap);
    free(mvdr); free(wq); free(W); free(A); free(L); free(a0); free(a1);
}

/* eigen-decomposition of the Hermitian A (row-major) by cyclic Jacobi in float64; eigenvalues descending,
 * eigenvectors = columns of V (row-major), unit norm, largest component real positive */
void orc_herm_eig(int n, const orc_cpx* Ain, double* eig, double* Vre, double* Vim)
{
    zc* A = (zc*)malloc(sizeof(zc) * n * n); zc* V = (zc*)calloc((size_t)n * n, sizeof(zc));
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++)
        A[i * n + j] = 0.5 * (((double)Ain[i * n + j].re + I * (double)Ain[i * n + j].im) + conj((double)Ain[j * n + i].re + I * (double)Ain[j * n + i].im));
    for (int i = 0; i < n; i++) V[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { const double m = creal(A[i * n + j] * conj(A[i * n + j])); if (i == j) diag += m; else off += m; }
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const zc apq = A[p * n + q];
                const double g = cabs(apq);
                if (g == 0.0) continue;
                const double app = creal(A[p * n + p]), aqq = creal(A[q * n + q]);
                const zc ph = apq / g;                                  /* e^{i phi} */
                const double tau = (aqq - app) / (2.0 * g);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                /* columns: A <- A R,  R = [[c, s ph], [-s conj(ph), c]] acting on columns (p, q) */
                for (int k = 0; k < n; k++) {
                    const zc akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * conj(ph) * akq;
                    A[k * n + q] = s * ph * akp + c * akq;
                    const zc vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * conj(ph) * vkq;
                    V[k * n + q] = s * ph * vkp + c * vkq;
                }
                /* rows: A <- R^H A */
                for (int k = 0; k < n; k++) {
                    const zc apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * ph * aqk;
                    A[q * n + k] = s * conj(ph) * apk + c * aqk;
                }
            }
    }
    int* idx = (int*)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) idx[i] = i;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (creal(A[idx[j] * n + idx[j]]) > creal(A[idx[i] * n + idx[i]])) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    for (int c = 0; c < n; c++) {
        const int s = idx[c];
        eig[c] = creal(A[s * n + s]);
        double nrm = 0.0, big = -1.0; int kb = 0;
        for (int k = 0; k < n; k++) { const double m = creal(V[k * n + s] * conj(V[k * n + s])); nrm += m; if (m > big) { big = m; kb = k; } }
        const zc sc = conj(V[kb * n + s]) / (sqrt(big) * sqrt(nrm));
        for (int k = 0; k < n; k++) { const zc v = V[k * n + s] * sc; Vre[k * n + c] = creal(v); Vim[k * n + c] = cimag(v); }
    }
    free(idx); free(A); free(V);
}

/* generateMUSICmap (saf_sh.c:1754-1799) */
void orc_generateMUSICmap(int order, const orc_cpx* Cx, const float* Y_grid, int nSources, int G, int logScaleFlag, float* pmap)
{
    const int nSH = (order + 1) * (order + 1);
    if (nSources > nSH / 2) nSources = nSH / 2;
    double* eig = (double*)malloc(sizeof(double) * nSH); double* Vr = (double*)malloc(sizeof(double) * nSH * nSH); double* Vi = (double*)malloc(sizeof(double) * nSH * nSH);
    orc_herm_eig(nSH, Cx, eig, Vr, Vi);
    for (int d = 0; d < G; d++) {
        float tmp = 0.0f;
        for (int j = nSources; j < nSH; j++) {
            float complex s = 0.0f;                                 /* (Vn^T Y)[j][d]: no conjugation (:1781-1784) */
            for (int i = 0; i < nSH; i++) s += ((float)Vr[i * nSH + j] + I * (float)Vi[i * nSH + j]) * Y_grid[(size_t)i * G + d];
            tmp += crealf(conjf(s) * s);
        }
        pmap[d] = logScaleFlag ? logf(1.0f / (tmp + 2.23e-10f)) : 1.0f / (tmp + 2.23e-10f);
    }
    free(eig); free(Vr); free(Vi);
}

/* generateMinNormMap (saf_sh.c:1801-1858) */
void orc_generateMinNormMap(int order, const orc_cpx* Cx, const float* Y_grid, int nSources, int G, int logScaleFlag, float* pmap)
{
    const int nSH = (order + 1) * (order + 1);
    if (nSources > nSH / 2) nSources = nSH / 2;
    double* eig = (double*)malloc(sizeof(double) * nSH); double* Vr = (double*)malloc(sizeof(double) * nSH * nSH); double* Vi = (double*)malloc(sizeof(double) * nSH * nSH);
    orc_herm_eig(nSH, Cx, eig, Vr, Vi);
    const int nN = nSH - nSources;
    float complex dot = 0.0f;                                       /* utility_cvvdot(Vn1, Vn1, NO_CONJ): sum of squares, not of moduli */
    for (int j = 0; j < nN; j++) { const float complex v = (float)Vr[j + nSources] + I * (float)Vi[j + nSources]; dot += v * v; }
    float complex Un[64];
    for (int i = 0; i < nSH; i++) {
        float complex s = 0.0f;                                     /* Vn * Vn1^H */
        for (int j = 0; j < nN; j++) s += ((float)Vr[i * nSH + j + nSources] + I * (float)Vi[i * nSH + j + nSources]) * conjf((float)Vr[j + nSources] + I * (float)Vi[j + nSources]);
        Un[i] = s / (dot + 2.23e-9f);
    }
    for (int d = 0; d < G; d++) {
        float complex s = 0.0f;                                     /* Un^H Y */
        for (int i = 0; i < nSH; i++) s += conjf(Un[i]) * Y_grid[(size_t)i * G + d];
        const float m = powf(cabsf(s), 2.0f) + 2.23e-9f;
        pmap[d] = logScaleFlag ? logf(1.0f / m) : 1.0f / m;
    }
    free(eig); free(Vr); free(Vi);
}
