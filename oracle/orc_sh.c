/*
 * orc_sh.c — oracle: spherical harmonics, HOA conventions, loudspeaker decoder
 * design, small dense linear algebra, VBAP.
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  Reference paths relative to
 * /root/reference.
 */
#include "saf_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>

#define ORC_PI   3.14159265358979323846264338327950288f   /* SAF_PI  (saf_utilities.h:70) */
#define ORC_PId  3.14159265358979323846264338327950288    /* SAF_PId (saf_utilities.h:73) */
#define ORC_SQRT4PI 3.544907701811032f                     /* saf_utilities.h:89 */
#define NSH(o) (((o) + 1) * ((o) + 1))

/* factorial (saf_utility_misc.c:174-186): long double; below 15 the reference reads
 * a table (saf_utility_misc.c:33-34) whose entry for 14! is 8.71782891e10, i.e.
 * 2.4e-8 below the exact 87178291200 — kept, since it feeds the order-7 norms. */
static long double orc_factorial(int n)
{
    if (n == 14) return 8.71782891e10L;
    long double ff = 1.0L;
    for (int i = 2; i <= n; i++) ff *= (long double)i;
    return ff;
}

/* ========================================================================== */
/*                       Legendre functions / real SH                         */
/* ========================================================================== */

/* unnorm_legendreP (saf_sh.c:53-127) */
void orc_unnorm_legendreP(int n, const double* x, int lenX, double* y)
{
    if (n == 0) { for (int i = 0; i < lenX; i++) y[i] = 1.0; return; }
    double* P = (double*)calloc((size_t)(n + 3) * lenX, sizeof(double));
    double* s_n = (double*)malloc(sizeof(double) * lenX);
    double* tc = (double*)malloc(sizeof(double) * lenX);
    double* sqrt_n = (double*)malloc(sizeof(double) * (2 * n + 1));
    for (int i = 0; i < lenX; i++) {
        double s = sqrt(1.0 - pow(x[i], 2.0)) + 2.23e-20;
        s_n[i] = pow(-s, (double)n);
        tc[i] = -2.0 * x[i] / s;
    }
    for (int i = 0; i < 2 * n + 1; i++) sqrt_n[i] = sqrt((double)i);
    double norm = 1.0;
    for (int i = 1; i <= n; i++) norm *= 1.0 - 1.0 / (2.0 * (double)i);
    for (int i = 0; i < lenX; i++) {
        P[n * lenX + i] = sqrt(norm) * s_n[i];
        P[(n - 1) * lenX + i] = P[n * lenX + i] * tc[i] * (double)n / sqrt_n[2 * n];
    }
    for (int m = n - 2; m >= 0; m--)
        for (int i = 0; i < lenX; i++)
            P[m * lenX + i] = (P[(m + 1) * lenX + i] * tc[i] * ((double)m + 1.0)
                               - P[(m + 2) * lenX + i] * sqrt_n[n + m + 2] * sqrt_n[n - m - 1])
                              / (sqrt_n[n + m + 1] * sqrt_n[n - m]);
    for (int i = 0; i < n + 1; i++) memcpy(&y[i * lenX], &P[i * lenX], sizeof(double) * lenX);
    for (int i = 0; i < lenX; i++)
        if (sqrt(1.0 - pow(x[i], 2.0)) == 0) y[i] = pow(x[i], (double)n);
    for (int m = 1; m < n; m++) {
        double scale = 1.0;
        for (int i = n - m + 1; i < n + m + 1; i++) scale *= sqrt_n[i];
        for (int i = 0; i < lenX; i++) y[m * lenX + i] *= scale;
    }
    double scale = 1.0;
    for (int i = 1; i < 2 * n + 1; i++) scale *= sqrt_n[i];
    for (int i = 0; i < lenX; i++) y[n * lenX + i] *= scale;
    free(P); free(s_n); free(tc); free(sqrt_n);
}

/* unnorm_legendreP_recur (saf_sh.c:129-183) */
static void legendreP_recur(int n, const float* x, int lenX, const float* Pnm_minus1, const float* Pnm_minus2, float* Pnm)
{
    if (n == 0) { for (int i = 0; i < lenX; i++) Pnm[i] = 1.0f; return; }
    for (int i = 0; i < lenX; i++) {
        float x2 = x[i] * x[i];
        switch (n) {
            case 1:
                Pnm[0 * lenX + i] = x[i];
                Pnm[1 * lenX + i] = sqrtf(1.0f - x2);
                break;
            case 2:
                Pnm[0 * lenX + i] = (3.0f * x2 - 1.0f) / 2.0f;
                Pnm[1 * lenX + i] = x[i] * 3.0f * sqrtf(1.0f - x2);
                Pnm[2 * lenX + i] = 3.0f * (1.0f - x2);
                break;
            default: {
                float one_min_x2 = 1.0f - x2;
                int k = 2 * n - 1;
                float dfact_k = 1.0f;
                if ((k % 2) == 0) for (int kk = 1; kk < k / 2 + 1; kk++) dfact_k *= 2.0f * (float)kk;
                else              for (int kk = 1; kk < (k + 1) / 2 + 1; kk++) dfact_k *= (2.0f * (float)kk - 1.0f);
                Pnm[n * lenX + i] = dfact_k * powf(one_min_x2, (float)n / 2.0f);
                Pnm[(n - 1) * lenX + i] = (float)k * x[i] * Pnm_minus1[(n - 1) * lenX + i];
                for (int m = 0; m < n - 1; m++)
                    Pnm[m * lenX + i] = (((float)k * x[i] * Pnm_minus1[m * lenX + i])
                                         - ((float)(n + m - 1) * Pnm_minus2[m * lenX + i])) / (float)(n - m);
            } break;
        }
    }
}

/* getSHreal (saf_sh.c:190-253) */
void orc_getSHreal(int order, const float* dirs_rad, int nDirs, float* Y)
{
    if (nDirs < 1) return;
    double* Lnm = (double*)malloc(sizeof(double) * (2 * order + 1) * nDirs);
    double* norm_real = (double*)malloc(sizeof(double) * (2 * order + 1));
    double* cos_incl = (double*)malloc(sizeof(double) * nDirs);
    double* p_nm = (double*)malloc(sizeof(double) * (order + 1) * nDirs);
    for (int dir = 0; dir < nDirs; dir++) cos_incl[dir] = cos((double)dirs_rad[dir * 2 + 1]);
    int idx_Y = 0;
    for (int n = 0; n <= order; n++) {
        orc_unnorm_legendreP(n, cos_incl, nDirs, p_nm);
        for (int dir = 0; dir < nDirs; dir++) {
            if (n != 0) {
                for (int m = -n, j = 0; m <= n; m++, j++)
                    Lnm[j * nDirs + dir] = pow(-1.0, (double)abs(m)) * p_nm[abs(m) * nDirs + dir];
            } else Lnm[dir] = p_nm[dir];
        }
        for (int m = -n, j = 0; m <= n; m++, j++)
            norm_real[j] = sqrt((2.0 * (double)n + 1.0) * (double)orc_factorial(n - abs(m))
                                / (4.0 * ORC_PId * (double)orc_factorial(n + abs(m))));
        for (int dir = 0; dir < nDirs; dir++)
            for (int m = -n, j = 0; m <= n; m++, j++) {
                if (j < n)
                    Y[(j + idx_Y) * nDirs + dir] = (float)(norm_real[j] * Lnm[j * nDirs + dir] * sqrt(2.0) * sin((double)(n - j) * (double)dirs_rad[dir * 2]));
                else if (j == n)
                    Y[(j + idx_Y) * nDirs + dir] = (float)(norm_real[j] * Lnm[j * nDirs + dir]);
                else
                    Y[(j + idx_Y) * nDirs + dir] = (float)(norm_real[j] * Lnm[j * nDirs + dir] * sqrt(2.0) * cos((double)(abs(m)) * (double)dirs_rad[dir * 2]));
            }
        idx_Y += 2 * n + 1;
    }
    free(p_nm); free(Lnm); free(norm_real); free(cos_incl);
}

/* shared body of getSHreal_recur (saf_sh.c:255-331) and getRSH_recur (saf_hoa.c:152-228) */
static void sh_recur(int N, const float* dirs, int nDirs, float* Y, int rsh)
{
    if (nDirs < 1) return;
    float* factorials_n = (float*)malloc(sizeof(float) * (2 * N + 1));
    float* leg_n = (float*)calloc((size_t)(N + 1) * nDirs, sizeof(float));
    float* leg_n_1 = (float*)calloc((size_t)(N + 1) * nDirs, sizeof(float));
    float* leg_n_2 = (float*)calloc((size_t)(N + 1) * nDirs, sizeof(float));
    float* ci = (float*)malloc(sizeof(float) * nDirs);
    for (int i = 0; i < 2 * N + 1; i++) factorials_n[i] = (float)orc_factorial(i);
    for (int dir = 0; dir < nDirs; dir++)
        ci[dir] = rsh ? sinf(dirs[dir * 2 + 1] * ORC_PI / 180.0f) : cosf(dirs[dir * 2 + 1]);
    int index_n = 0;
    for (int n = 0; n < N + 1; n++) {
        if (n == 0) {
            for (int dir = 0; dir < nDirs; dir++) Y[dir] = rsh ? 1.0f : 1.0f / ORC_SQRT4PI;
            index_n = 1;
        } else {
            legendreP_recur(n, ci, nDirs, leg_n_1, leg_n_2, leg_n);
            float Nn0 = sqrtf(2.0f * (float)n + 1.0f);
            for (int dir = 0; dir < nDirs; dir++)
                for (int m = 0; m < n + 1; m++) {
                    if (m == 0) {
                        Y[(index_n + n) * nDirs + dir] = rsh ? Nn0 * leg_n[m * nDirs + dir]
                                                             : Nn0 / ORC_SQRT4PI * leg_n[m * nDirs + dir];
                    } else {
                        float Nnm = Nn0 * sqrtf(2.0f * factorials_n[n - m] / factorials_n[n + m]);
                        if (rsh) {
                            Y[(index_n + n - m) * nDirs + dir] = Nnm * leg_n[m * nDirs + dir] * sinf((float)m * (dirs[dir * 2]) * ORC_PI / 180.0f);
                            Y[(index_n + n + m) * nDirs + dir] = Nnm * leg_n[m * nDirs + dir] * cosf((float)m * (dirs[dir * 2]) * ORC_PI / 180.0f);
                        } else {
                            Y[(index_n + n - m) * nDirs + dir] = Nnm / ORC_SQRT4PI * leg_n[m * nDirs + dir] * sinf((float)m * (dirs[dir * 2]));
                            Y[(index_n + n + m) * nDirs + dir] = Nnm / ORC_SQRT4PI * leg_n[m * nDirs + dir] * cosf((float)m * (dirs[dir * 2]));
                        }
                    }
                }
            index_n += 2 * n + 1;
        }
        memcpy(leg_n_2, leg_n_1, sizeof(float) * (size_t)(N + 1) * nDirs);
        memcpy(leg_n_1, leg_n, sizeof(float) * (size_t)(N + 1) * nDirs);
    }
    free(factorials_n); free(leg_n); free(leg_n_1); free(leg_n_2); free(ci);
}
void orc_getSHreal_recur(int N, const float* dirs_rad, int nDirs, float* Y) { sh_recur(N, dirs_rad, nDirs, Y, 0); }
void orc_getRSH_recur(int N, const float* dirs_deg, int nDirs, float* Y) { sh_recur(N, dirs_deg, nDirs, Y, 1); }

/* getRSH (saf_hoa.c:118-150) */
void orc_getRSH(int N, const float* dirs_deg, int nDirs, float* Y)
{
    if (nDirs < 1) return;
    const int nSH = NSH(N);
    const float scale = sqrtf(4.0f * ORC_PI);
    float* dirs_rad = (float*)malloc(sizeof(float) * nDirs * 2);
    for (int i = 0; i < nDirs; i++) {
        dirs_rad[i * 2 + 0] = dirs_deg[i * 2 + 0] * ORC_PI / 180.0f;
        dirs_rad[i * 2 + 1] = ORC_PI / 2.0f - (dirs_deg[i * 2 + 1] * ORC_PI / 180.0f);
    }
    orc_getSHreal(N, dirs_rad, nDirs, Y);
    for (int i = 0; i < nSH * nDirs; i++) Y[i] *= scale;
    free(dirs_rad);
}

/* getMaxREweights (saf_hoa.c:235-267) */
void orc_getMaxREweights(int order, int diagMtxFlag, float* a_n)
{
    double x = cosf(137.9f * (ORC_PI / 180.0f) / ((float)order + 1.51f));
    const int nSH = NSH(order);
    memset(a_n, 0, sizeof(float) * (diagMtxFlag ? nSH * nSH : nSH));
    double* ppm = (double*)calloc(order + 1, sizeof(double));
    int idx = 0;
    for (int n = 0; n <= order; n++) {
        orc_unnorm_legendreP(n, &x, 1, ppm);
        for (int i = 0; i < 2 * n + 1; i++) {
            if (diagMtxFlag) a_n[(idx + i) * nSH + (idx + i)] = (float)ppm[0];
            else a_n[idx + i] = (float)ppm[0];
        }
        idx += 2 * n + 1;
    }
    free(ppm);
}

/* convertHOAChannelConvention (saf_hoa.c:40-70); 1 = ACN, 2 = FuMa */
void orc_convertHOAChannelConvention(float* insig, int order, int len, int inConv, int outConv)
{
    const int nSH = NSH(order);
    if (order == 0 || inConv == outConv) return;
    float* t = (float*)malloc(sizeof(float) * len);
#define SWAP(a, b) do { memcpy(t, &insig[(a) * len], sizeof(float) * len); memcpy(&insig[(a) * len], &insig[(b) * len], sizeof(float) * len); memcpy(&insig[(b) * len], t, sizeof(float) * len); } while (0)
    if (inConv == 2 && outConv == 1) { SWAP(1, 3); SWAP(1, 2); }
    else if (inConv == 1 && outConv == 2) { SWAP(1, 2); SWAP(1, 3); }
#undef SWAP
    for (int i = 4; i < nSH; i++) memset(&insig[i * len], 0, sizeof(float) * len);
    free(t);
}

/* convertHOANormConvention (saf_hoa.c:72-116); 1 = N3D, 2 = SN3D, 3 = FuMa */
void orc_convertHOANormConvention(float* insig, int order, int len, int inConv, int outConv)
{
    if (order == 0 || inConv == outConv) return;
#define SCAL(ch, s) do { float sc_ = (s); for (int i_ = 0; i_ < len; i_++) insig[(ch) * len + i_] *= sc_; } while (0)
    if (inConv == 1) {
        if (outConv == 2) {
            for (int n = 0; n < order + 1; n++)
                for (int ch = n * n; ch < NSH(n); ch++) SCAL(ch, 1.0f / sqrtf(2.0f * (float)n + 1.0f));
        } else if (outConv == 3) {
            SCAL(0, 1.0f / sqrtf(2.0f));
            for (int ch = 1; ch < 4; ch++) SCAL(ch, 1.0f / sqrtf(3.0f));
        }
    } else if (inConv == 2) {
        if (outConv == 1) {
            for (int n = 0; n < order + 1; n++)
                for (int ch = n * n; ch < NSH(n); ch++) SCAL(ch, sqrtf(2.0f * (float)n + 1.0f));
        } else if (outConv == 3) SCAL(0, 1.0f / sqrtf(2.0f));
    } else if (inConv == 3) {
        if (outConv == 1) {
            SCAL(0, sqrtf(2.0f));
            for (int ch = 1; ch < 4; ch++) SCAL(ch, sqrtf(3.0f));
        } else if (outConv == 2) SCAL(0, sqrtf(2.0f));
    }
#undef SCAL
}

/* ========================================================================== */
/*                    small dense linear algebra (own code)                   */
/* ========================================================================== */

/* One-sided Jacobi SVD of A [m x n], m >= n, double precision.
 * On return A's columns hold U*diag(s) -> normalised into U [m x n], s[n], V [n x n]. */
static void jacobi_svd(double* A, int m, int n, double* s, double* V)
{
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < m; i++) {
                    double ap = A[i * n + p], aq = A[i * n + q];
                    alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
                }
                if (gamma == 0.0) continue;
                double lim = fabs(gamma) / sqrt(alpha * beta + 1e-300);
                if (lim > off) off = lim;
                if (lim < 1e-15) continue;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < m; i++) {
                    double ap = A[i * n + p], aq = A[i * n + q];
                    A[i * n + p] = c * ap - sn * aq; A[i * n + q] = sn * ap + c * aq;
                }
                for (int i = 0; i < n; i++) {
                    double vp = V[i * n + p], vq = V[i * n + q];
                    V[i * n + p] = c * vp - sn * vq; V[i * n + q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-14) break;
    }
    for (int j = 0; j < n; j++) {
        double nr = 0; for (int i = 0; i < m; i++) nr += A[i * n + j] * A[i * n + j];
        s[j] = sqrt(nr);
    }
    /* sort descending */
    for (int a = 0; a < n - 1; a++) {
        int best = a;
        for (int b = a + 1; b < n; b++) if (s[b] > s[best]) best = b;
        if (best != a) {
            double ts = s[a]; s[a] = s[best]; s[best] = ts;
            for (int i = 0; i < m; i++) { double t = A[i * n + a]; A[i * n + a] = A[i * n + best]; A[i * n + best] = t; }
            for (int i = 0; i < n; i++) { double t = V[i * n + a]; V[i * n + a] = V[i * n + best]; V[i * n + best] = t; }
        }
    }
    for (int j = 0; j < n; j++)
        if (s[j] > 0) for (int i = 0; i < m; i++) A[i * n + j] /= s[j];
}

/* thin SVD of a general real matrix M [r x c]: M = U diag(s) V^T, k = min(r,c);
 * U [r x k], V [c x k] (row-major, columns are singular vectors). */
static void thin_svd(const float* M, int r, int c, double* U, double* s, double* V)
{
    const int k = r < c ? r : c;
    if (r >= c) {
        double* A = (double*)malloc(sizeof(double) * r * c);
        for (int i = 0; i < r * c; i++) A[i] = M[i];
        double* Vf = (double*)malloc(sizeof(double) * c * c);
        jacobi_svd(A, r, c, s, Vf);
        memcpy(U, A, sizeof(double) * r * k);
        memcpy(V, Vf, sizeof(double) * c * k);
        free(A); free(Vf);
    } else {
        /* decompose M^T [c x r] = V diag(s) U^T */
        double* A = (double*)malloc(sizeof(double) * r * c);
        for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) A[j * r + i] = M[i * c + j];
        double* Uf = (double*)malloc(sizeof(double) * r * r);
        jacobi_svd(A, c, r, s, Uf);
        memcpy(V, A, sizeof(double) * c * k);
        memcpy(U, Uf, sizeof(double) * r * k);
        free(A); free(Uf);
    }
}

/* singular values of a general real matrix (utility_ssvd with only `sing` requested), descending */
void orc_singular_values(const float* M, int r, int c, float* s)
{
    const int k = r < c ? r : c;
    double* U = (double*)malloc(sizeof(double) * (size_t)r * k); double* sv = (double*)malloc(sizeof(double) * k); double* V = (double*)malloc(sizeof(double) * (size_t)c * k);
    thin_svd(M, r, c, U, sv, V);
    for (int i = 0; i < k; i++) s[i] = (float)sv[i];
    free(U); free(sv); free(V);
}

/* utility_spinv (saf_utility_veclib.c:3466-3560): singular values <= 1e-5 are
 * multiplied in rather than inverted. out [dim2 x dim1]. */
void orc_pinv(const float* inM, int dim1, int dim2, float* outM)
{
    const int k = dim1 < dim2 ? dim1 : dim2;
    double* U = (double*)malloc(sizeof(double) * dim1 * k);
    double* V = (double*)malloc(sizeof(double) * dim2 * k);
    double* s = (double*)malloc(sizeof(double) * k);
    thin_svd(inM, dim1, dim2, U, s, V);
    for (int j = 0; j < dim2; j++)
        for (int i = 0; i < dim1; i++) {
            double acc = 0;
            for (int q = 0; q < k; q++) {
                float sf = (float)s[q];
                double ss = sf > 1.0e-5f ? 1.0 / s[q] : s[q];
                acc += V[j * k + q] * ss * U[i * k + q];
            }
            outM[j * dim1 + i] = (float)acc;
        }
    free(U); free(V); free(s);
}

/* ========================================================================== */
/*                    convex hull of points on the sphere                     */
/* ========================================================================== */
/* The reference triangulates with quickhull plus rand() noise
 * (convhull_3d.c:367-839, noise at :400), so its face ORDER (and, on layouts with
 * co-spherical coplanar quads, its face SET) is not a function of the input.
 * This is an own incremental hull in double precision; faces come out
 * outward-oriented and canonically ordered (each face rotated to start at its
 * smallest vertex index, faces sorted lexicographically) so results are a pure
 * function of the input.  On non-degenerate layouts the face set is unique and
 * therefore identical to the reference's. */

typedef struct { int v[3]; double n[3]; double d; int alive; } hface;

static void face_plane(const double* P, hface* f)
{
    const double* a = &P[3 * f->v[0]]; const double* b = &P[3 * f->v[1]]; const double* c = &P[3 * f->v[2]];
    double u[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
    double w[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
    f->n[0] = u[1] * w[2] - u[2] * w[1];
    f->n[1] = u[2] * w[0] - u[0] * w[2];
    f->n[2] = u[0] * w[1] - u[1] * w[0];
    double l = sqrt(f->n[0] * f->n[0] + f->n[1] * f->n[1] + f->n[2] * f->n[2]);
    if (l > 0) { f->n[0] /= l; f->n[1] /= l; f->n[2] /= l; }
    f->d = f->n[0] * a[0] + f->n[1] * a[1] + f->n[2] * a[2];
}
static double face_dist(const hface* f, const double* p) { return f->n[0] * p[0] + f->n[1] * p[1] + f->n[2] * p[2] - f->d; }

static int cmp_face(const void* a, const void* b)
{
    const int* x = (const int*)a; const int* y = (const int*)b;
    for (int i = 0; i < 3; i++) if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
}

/* returns number of faces, faces malloc'd into *out (nf x 3) */
static int sphere_hull(const double* P, int n, int** out)
{
    const double EPS = 1e-9;
    *out = NULL;
    if (n < 4) return 0;
    /* initial tetrahedron: i0 = 0, i1 farthest from i0, i2 farthest from line, i3 farthest from plane */
    int i0 = 0, i1 = -1, i2 = -1, i3 = -1; double best = -1;
    for (int i = 1; i < n; i++) {
        double d = 0; for (int k = 0; k < 3; k++) d += (P[3 * i + k] - P[k]) * (P[3 * i + k] - P[k]);
        if (d > best) { best = d; i1 = i; }
    }
    best = -1;
    for (int i = 0; i < n; i++) {
        if (i == i0 || i == i1) continue;
        double u[3], w[3], c[3];
        for (int k = 0; k < 3; k++) { u[k] = P[3 * i1 + k] - P[3 * i0 + k]; w[k] = P[3 * i + k] - P[3 * i0 + k]; }
        c[0] = u[1] * w[2] - u[2] * w[1]; c[1] = u[2] * w[0] - u[0] * w[2]; c[2] = u[0] * w[1] - u[1] * w[0];
        double d = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        if (d > best) { best = d; i2 = i; }
    }
    hface t; t.v[0] = i0; t.v[1] = i1; t.v[2] = i2; face_plane(P, &t);
    best = -1;
    for (int i = 0; i < n; i++) {
        if (i == i0 || i == i1 || i == i2) continue;
        double d = fabs(face_dist(&t, &P[3 * i]));
        if (d > best) { best = d; i3 = i; }
    }
    if (i3 < 0 || best < EPS) return 0;   /* all coplanar */
    double cen[3];
    for (int k = 0; k < 3; k++) cen[k] = (P[3 * i0 + k] + P[3 * i1 + k] + P[3 * i2 + k] + P[3 * i3 + k]) / 4.0;

    int cap = 4 * n + 64, nf = 0;
    hface* F = (hface*)malloc(sizeof(hface) * cap);
    int tet[4][3] = { { i0, i1, i2 }, { i0, i1, i3 }, { i0, i2, i3 }, { i1, i2, i3 } };
    for (int f = 0; f < 4; f++) {
        hface* h = &F[nf++];
        h->v[0] = tet[f][0]; h->v[1] = tet[f][1]; h->v[2] = tet[f][2]; h->alive = 1;
        face_plane(P, h);
        if (face_dist(h, cen) > 0) { int s = h->v[1]; h->v[1] = h->v[2]; h->v[2] = s; face_plane(P, h); }
    }
    char* used = (char*)calloc(n, 1);
    used[i0] = used[i1] = used[i2] = used[i3] = 1;
    int* hor = (int*)malloc(sizeof(int) * 2 * cap);
    for (int p = 0; p < n; p++) {
        if (used[p]) continue;
        used[p] = 1;
        /* visible faces */
        int nvis = 0;
        for (int f = 0; f < nf; f++) if (F[f].alive && face_dist(&F[f], &P[3 * p]) > EPS) { F[f].alive = 2; nvis++; }
        if (!nvis) continue;   /* inside / on the hull */
        /* horizon: directed edges of visible faces whose reverse edge is not on a visible face */
        int nh = 0;
        for (int f = 0; f < nf; f++) {
            if (F[f].alive != 2) continue;
            for (int e = 0; e < 3; e++) {
                int a = F[f].v[e], b = F[f].v[(e + 1) % 3];
                int shared = 0;
                for (int g = 0; g < nf && !shared; g++) {
                    if (F[g].alive != 2 || g == f) continue;
                    for (int e2 = 0; e2 < 3; e2++)
                        if (F[g].v[e2] == b && F[g].v[(e2 + 1) % 3] == a) { shared = 1; break; }
                }
                if (!shared) { hor[2 * nh] = a; hor[2 * nh + 1] = b; nh++; }
            }
        }
        for (int f = 0; f < nf; f++) if (F[f].alive == 2) F[f].alive = 0;
        /* compact occasionally */
        if (nf + nh >= cap) {
            int w = 0;
            for (int f = 0; f < nf; f++) if (F[f].alive) F[w++] = F[f];
            nf = w;
            if (nf + nh >= cap) { cap = 2 * (nf + nh) + 64; F = (hface*)realloc(F, sizeof(hface) * cap); hor = (int*)realloc(hor, sizeof(int) * 2 * cap); }
        }
        for (int e = 0; e < nh; e++) {
            hface* h = &F[nf++];
            h->v[0] = hor[2 * e]; h->v[1] = hor[2 * e + 1]; h->v[2] = p; h->alive = 1;
            face_plane(P, h);
        }
    }
    int cnt = 0;
    for (int f = 0; f < nf; f++) if (F[f].alive) cnt++;
    int* o = (int*)malloc(sizeof(int) * 3 * (cnt > 0 ? cnt : 1));
    int w = 0;
    for (int f = 0; f < nf; f++) {
        if (!F[f].alive) continue;
        int* v = F[f].v; int r = 0;
        if (v[1] < v[r]) r = 1;
        if (v[2] < v[r]) r = 2;
        o[3 * w] = v[r]; o[3 * w + 1] = v[(r + 1) % 3]; o[3 * w + 2] = v[(r + 2) % 3];
        w++;
    }
    qsort(o, cnt, 3 * sizeof(int), cmp_face);
    free(F); free(used); free(hor);
    *out = o;
    return cnt;
}

/* ========================================================================== */
/*                                   VBAP                                     */
/* ========================================================================== */

/* findLsTriplets (saf_vbap.c:499-674) */
int orc_findLsTriplets(const float* ls_dirs_deg, int L, int omitLargeTriangles, float** out_vertices, int* numOutVertices, int** out_faces, int* numOutFaces)
{
    *numOutVertices = L;
    *out_vertices = (float*)malloc(sizeof(float) * L * 3);
    double* P = (double*)malloc(sizeof(double) * L * 3);
    for (int i = 0; i < L; i++) {
        (*out_vertices)[i * 3 + 2] = (float)sin((double)ls_dirs_deg[i * 2 + 1] * ORC_PId / 180.0);
        double rcoselev = cos((double)ls_dirs_deg[i * 2 + 1] * ORC_PId / 180.0);
        (*out_vertices)[i * 3 + 0] = (float)(rcoselev * cos((double)ls_dirs_deg[i * 2 + 0] * ORC_PId / 180.0));
        (*out_vertices)[i * 3 + 1] = (float)(rcoselev * sin((double)ls_dirs_deg[i * 2 + 0] * ORC_PId / 180.0));
        for (int k = 0; k < 3; k++) P[3 * i + k] = (*out_vertices)[i * 3 + k];
    }
    int* faces = NULL;
    int nFaces = sphere_hull(P, L, &faces);
    free(P);
    if (!faces) { *out_faces = NULL; *numOutFaces = 0; return -1; }
    const float* V = *out_vertices;
    int* keep = (int*)malloc(sizeof(int) * nFaces);
    int nValid = 0;
    for (int i = 0; i < nFaces; i++) {
        float vecs[3][3], a[3], b[3], cvec[3], centroid[3];
        for (int j = 0; j < 3; j++) for (int q = 0; q < 3; q++) vecs[q][j] = V[faces[i * 3 + q] * 3 + j];
        for (int j = 0; j < 3; j++) { a[j] = vecs[1][j] - vecs[0][j]; b[j] = vecs[2][j] - vecs[1][j]; }
        cvec[0] = a[1] * b[2] - a[2] * b[1]; cvec[1] = a[2] * b[0] - a[0] * b[2]; cvec[2] = a[0] * b[1] - a[1] * b[0];
        for (int j = 0; j < 3; j++) centroid[j] = (vecs[0][j] + vecs[1][j] + vecs[2][j]) / 3.0f;
        float dotcc = cvec[0] * centroid[0] + cvec[1] * centroid[1] + cvec[2] * centroid[2];
        float cl = dotcc < 0.99999999f ? dotcc : 0.99999999f; if (cl < -0.99999999f) cl = -0.99999999f;
        keep[i] = acosf(cl) < (ORC_PI / 2.0f);
        if (keep[i] && omitLargeTriangles) {
            const float aperture_lim = 180.0f * ORC_PI / 180.0f;   /* APERTURE_LIMIT_DEG, saf_vbap_internal.h:50 */
            float abc[3];
            for (int q = 0; q < 3; q++) {
                const float* x = vecs[q]; const float* y = vecs[(q + 1) % 3];
                abc[q] = acosf(x[0] * y[0] + x[1] * y[1] + x[2] * y[2]);
            }
            keep[i] = abc[0] < aperture_lim && abc[1] < aperture_lim && abc[2] < aperture_lim;
        }
        nValid += keep[i];
    }
    *out_faces = (int*)malloc(sizeof(int) * 3 * (nValid > 0 ? nValid : 1));
    for (int i = 0, j = 0; i < nFaces; i++)
        if (keep[i]) { memcpy(&(*out_faces)[3 * j], &faces[3 * i], sizeof(int) * 3); j++; }
    *numOutFaces = nValid;
    free(keep); free(faces);
    return 0;
}

/* invertLsMtx3D (saf_vbap.c:676-705): per face, inverse of the 3x3 matrix whose COLUMNS are the unit vectors */
void orc_invertLsMtx3D(const float* U, const int* g, int N_group, float* inv)
{
    for (int n = 0; n < N_group; n++) {
        double m[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[j * 3 + i] = U[g[n * 3 + i] * 3 + j];
        double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
        double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
        double id = 1.0 / det;
        inv[n * 9 + 0] = (float)(c00 * id);
        inv[n * 9 + 1] = (float)((m[2] * m[7] - m[1] * m[8]) * id);
        inv[n * 9 + 2] = (float)((m[1] * m[5] - m[2] * m[4]) * id);
        inv[n * 9 + 3] = (float)(c01 * id);
        inv[n * 9 + 4] = (float)((m[0] * m[8] - m[2] * m[6]) * id);
        inv[n * 9 + 5] = (float)((m[2] * m[3] - m[0] * m[5]) * id);
        inv[n * 9 + 6] = (float)(c02 * id);
        inv[n * 9 + 7] = (float)((m[1] * m[6] - m[0] * m[7]) * id);
        inv[n * 9 + 8] = (float)((m[0] * m[4] - m[1] * m[3]) * id);
    }
}

/* getSpreadSrcDirs3D (saf_vbap.c:707-783) */
static void spread_dirs(float azi, float elev, float spread, int num_src, int num_rings, float* Us)
{
    float u[3] = { cosf(elev) * cosf(azi), cosf(elev) * sinf(azi), sinf(elev) };
    float uxu[3][3], ux[3][3] = { { 0, -u[2], u[1] }, { u[2], 0, -u[0] }, { -u[1], u[0], 0 } }, R[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) uxu[i][j] = (i == j) ? powf(u[i], 2.0f) : u[i] * u[j];
    float theta = 2.0f * ORC_PI / (float)num_src, st = sinf(theta), ct = cosf(theta);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = st * ux[i][j] + (1.0f - ct) * uxu[i][j] + (i == j ? ct : 0.0f);
    float* sb = (float*)calloc(num_src * 3, sizeof(float));
    if ((elev > ORC_PI / 2.0f - 0.01f) || (elev < -(ORC_PI / 2.0f - 0.01f))) sb[0] = 1.0f;
    else {
        float uu2[3] = { u[1] * 1.0f - u[2] * 0.0f, u[2] * 0.0f - u[0] * 1.0f, 0.0f };
        float sc = sqrtf(uu2[0] * uu2[0] + uu2[1] * uu2[1] + uu2[2] * uu2[2]);
        for (int i = 0; i < 3; i++) sb[i] = uu2[i] / sc;
    }
    for (int ns = 1; ns < num_src; ns++)
        for (int i = 0; i < 3; i++) {
            float acc = 0; for (int j = 0; j < 3; j++) acc += R[i][j] * sb[(ns - 1) * 3 + j];
            sb[ns * 3 + i] = acc;
        }
    float spread_rad = (spread / 2.0f) * ORC_PI / 180.0f, ring_rad = spread_rad / (float)num_rings;
    for (int nr = 0; nr < num_rings; nr++)
        for (int ns = 0; ns < num_src; ns++)
            for (int i = 0; i < 3; i++)
                Us[(nr * num_src + ns) * 3 + i] = u[i] + sb[ns * 3 + i] * tanf(ring_rad * (float)(nr + 1));
    float nrm = sqrtf(Us[0] * Us[0] + Us[1] * Us[1] + Us[2] * Us[2]);
    for (int i = 0; i < num_rings * num_src * 3; i++) Us[i] /= nrm;
    for (int i = 0; i < 3; i++) Us[(num_rings * num_src) * 3 + i] = u[i];
    free(sb);
}

/* vbap3D (saf_vbap.c:786-896) */
void orc_vbap3D(const float* src_dirs, int src_num, int ls_num, const int* grp, int nFaces, float spread, const float* inv, float** GainMtx)
{
    *GainMtx = (float*)malloc(sizeof(float) * (size_t)src_num * ls_num);
    float* gains = (float*)malloc(sizeof(float) * ls_num);
    const int mdap = spread > 0.1f;
    const int nSpr = mdap ? 9 : 1;
    float Us[9 * 3];
    for (int ns = 0; ns < src_num; ns++) {
        float azi = src_dirs[ns * 2 + 0] * ORC_PI / 180.0f, elev = src_dirs[ns * 2 + 1] * ORC_PI / 180.0f;
        if (mdap) spread_dirs(azi, elev, spread, 8, 1, Us);
        else { Us[0] = cosf(azi) * cosf(elev); Us[1] = sinf(azi) * cosf(elev); Us[2] = sinf(elev); }
        memset(gains, 0, sizeof(float) * ls_num);
        for (int sp = 0; sp < nSpr; sp++) {
            const float* u = &Us[sp * 3];
            for (int i = 0; i < nFaces; i++) {
                float g[3], mn = 2.23e13f, rms = 0.0f;
                for (int r = 0; r < 3; r++) {
                    /* utility_svvdot: sequential float dot product */
                    g[r] = inv[i * 9 + r * 3 + 0] * u[0] + inv[i * 9 + r * 3 + 1] * u[1] + inv[i * 9 + r * 3 + 2] * u[2];
                    mn = mn < g[r] ? mn : g[r];
                    rms += powf(g[r], 2.0f);
                }
                rms = sqrtf(rms);
                if (mn > -0.001) {
                    if (mdap) { for (int j = 0; j < 3; j++) gains[grp[i * 3 + j]] += g[j] / rms; }
                    else { for (int j = 0; j < 3; j++) gains[grp[i * 3 + j]] = g[j] / rms; break; }
                }
            }
        }
        float grms = 0.0f;
        for (int i = 0; i < ls_num; i++) grms += powf(gains[i], 2.0f);
        grms = sqrtf(grms);
        for (int i = 0; i < ls_num; i++) { float v = gains[i] / grms; (*GainMtx)[(size_t)ns * ls_num + i] = v > 0.0f ? v : 0.0f; }
    }
    free(gains);
}

static void vbap_table_common(const float* src_dirs, int S, const float* ls_dirs_deg, int L, int omitLarge, int enableDummies, float spread,
                              float** gtable, int* N_gtable, int* nTriangles)
{
    /* saf_vbap.c:52-169 / :171-310 share this body */
    int needDummy[2] = { 1, 1 };
    float* verts = NULL; int* faces = NULL; int nV = 0, nF = 0;
    float* dirs = NULL; int Ld = L;
    if (enableDummies) {
        for (int i = 0; i < L; i++) {
            if (ls_dirs_deg[i * 2 + 1] <= -60.0f) needDummy[0] = 0;   /* ADD_DUMMY_LIMIT saf_vbap_internal.h:46 */
            if (ls_dirs_deg[i * 2 + 1] >= 60.0f) needDummy[1] = 0;
        }
    } else needDummy[0] = needDummy[1] = 0;
    if (needDummy[0] || needDummy[1]) {
        Ld = L + needDummy[0] + needDummy[1];
        dirs = (float*)malloc(sizeof(float) * Ld * 2);
        memcpy(dirs, ls_dirs_deg, sizeof(float) * L * 2);
        int i = L;
        if (needDummy[0]) { dirs[i * 2] = 0.0f; dirs[i * 2 + 1] = -90.0f; i++; }
        if (needDummy[1]) { dirs[i * 2] = 0.0f; dirs[i * 2 + 1] = 90.0f; }
        orc_findLsTriplets(dirs, Ld, omitLarge, &verts, &nV, &faces, &nF);
        free(dirs);
    } else
        orc_findLsTriplets(ls_dirs_deg, L, omitLarge, &verts, &nV, &faces, &nF);
    if (!faces) { *gtable = NULL; *N_gtable = 0; *nTriangles = 0; free(verts); return; }
    float* inv = (float*)malloc(sizeof(float) * 9 * (nF > 0 ? nF : 1));
    orc_invertLsMtx3D(verts, faces, nF, inv);
    orc_vbap3D(src_dirs, S, nV, faces, nF, spread, inv, gtable);
    if (needDummy[0] || needDummy[1]) {
        for (int i = 0; i < S; i++) memmove(&(*gtable)[(size_t)i * L], &(*gtable)[(size_t)i * nV], sizeof(float) * L);
        *gtable = (float*)realloc(*gtable, sizeof(float) * (size_t)S * L);
    }
    *N_gtable = S; *nTriangles = nF;
    free(verts); free(faces); free(inv);
}

void orc_generateVBAPgainTable3D_srcs(const float* src_dirs_deg, int S, const float* ls_dirs_deg, int L, int omitLarge, int enableDummies, float spread,
                                      float** gtable, int* N_gtable, int* nTriangles)
{
    vbap_table_common(src_dirs_deg, S, ls_dirs_deg, L, omitLarge, enableDummies, spread, gtable, N_gtable, nTriangles);
}

/* generateVBAPgainTable3D (saf_vbap.c:171-310) */
void orc_generateVBAPgainTable3D(const float* ls_dirs_deg, int L, int az_res_deg, int el_res_deg, int omitLarge, int enableDummies, float spread,
                                 float** gtable, int* N_gtable, int* nTriangles)
{
    const int N_azi = (int)((360.0f / (float)az_res_deg) + 1.5f);
    const int N_ele = (int)((180.0f / (float)el_res_deg) + 1.5f);
    float* azi = (float*)malloc(sizeof(float) * N_azi);
    float* ele = (float*)malloc(sizeof(float) * N_ele);
    float fi; int i;
    for (fi = -180.0f, i = 0; i < N_azi; fi += (float)az_res_deg, i++) azi[i] = fi;
    for (fi = -90.0f, i = 0; i < N_ele; fi += (float)el_res_deg, i++) ele[i] = fi;
    float* src = (float*)malloc(sizeof(float) * (size_t)N_azi * N_ele * 2);
    for (i = 0; i < N_ele; i++)
        for (int j = 0; j < N_azi; j++) { src[(i * N_azi + j) * 2] = azi[j]; src[(i * N_azi + j) * 2 + 1] = ele[i]; }
    vbap_table_common(src, N_azi * N_ele, ls_dirs_deg, L, omitLarge, enableDummies, spread, gtable, N_gtable, nTriangles);
    free(azi); free(ele); free(src);
}

/* compressVBAPgainTable3D (saf_vbap.c:312-367) */
void orc_compressVBAPgainTable3D(const float* gt, int nTable, int nDirs, float* comp, int* idx)
{
    memset(comp, 0, sizeof(float) * nTable * 3);
    memset(idx, 0, sizeof(int) * nTable * 3);
    for (int nt = 0; nt < nTable; nt++) {
        float gains_nt[3] = { 0, 0, 0 }, sum = 0.0f; int idx_nt[3] = { 0, 0, 0 }, j = 0;
        for (int i = 0; i < nDirs; i++)
            if (gt[(size_t)nt * nDirs + i] > 0.0000001f && j < 3) {
                gains_nt[j] = gt[(size_t)nt * nDirs + i]; sum += gains_nt[j]; idx_nt[j] = i; j++;
            }
        for (int i = 0; i < j; i++) {
            float v = gains_nt[i] / sum;
            comp[nt * 3 + i] = v > 0.0f ? v : 0.0f;
            idx[nt * 3 + i] = idx_nt[i];
        }
    }
}

/* ========================================================================== */
/*                         loudspeaker decoder design                         */
/* ========================================================================== */

/* getEPAD (saf_hoa_internal.c:41-98) */
static void get_epad(int order, const float* ls_dirs_deg, int nLS, float* decMtx)
{
    const int nSH = NSH(order);
    float* Y = (float*)malloc(sizeof(float) * nSH * nLS);
    orc_getRSH(order, ls_dirs_deg, nLS, Y);
    for (int i = 0; i < nSH * nLS; i++) Y[i] *= 1.0f / ORC_SQRT4PI;
    const int k = nSH < nLS ? nSH : nLS;
    double* U = (double*)malloc(sizeof(double) * nSH * k);
    double* V = (double*)malloc(sizeof(double) * nLS * k);
    double* s = (double*)malloc(sizeof(double) * k);
    thin_svd(Y, nSH, nLS, U, s, V);
    /* both branches of the reference reduce to V[:, :k] * U[:, :k]^T */
    const float scale = sqrtf(4.0f * ORC_PI / (float)nLS);
    for (int i = 0; i < nLS; i++)
        for (int j = 0; j < nSH; j++) {
            double acc = 0;
            for (int q = 0; q < k; q++) acc += V[i * k + q] * U[j * k + q];
            decMtx[i * nSH + j] = (float)acc * scale;
        }
    free(Y); free(U); free(V); free(s);
}

/* getAllRAD (saf_hoa_internal.c:100-155) */
static void get_allrad(int order, const float* ls_dirs_deg, int nLS, float* decMtx)
{
    const int nSH = NSH(order);
    int d0, d1;
    const float* t_dirs = orc_table("Tdesign_degree_100_dirs_deg", &d0, &d1);
    assert(t_dirs && d0 == 5100);
    const int nT = 5100;
    float* G = NULL; int Ng, nTri;
    orc_generateVBAPgainTable3D_srcs(t_dirs, nT, ls_dirs_deg, nLS, 0, 0, 0.0f, &G, &Ng, &nTri);
    float* Y = (float*)malloc(sizeof(float) * (size_t)nSH * nT);
    orc_getRSH(order, t_dirs, nT, Y);
    for (size_t i = 0; i < (size_t)nSH * nT; i++) Y[i] *= 1.0f / ORC_SQRT4PI;
    const float sc = (4.0f * ORC_PI) / (float)nT;
    for (int i = 0; i < nLS; i++)
        for (int j = 0; j < nSH; j++) {
            float acc = 0.0f;
            for (int t = 0; t < nT; t++) acc += G[(size_t)t * nLS + i] * Y[(size_t)j * nT + t];
            decMtx[i * nSH + j] = acc * sc;
        }
    free(Y); free(G);
}

/* getLoudspeakerDecoderMtx (saf_hoa.c:326-392) */
void orc_getLoudspeakerDecoderMtx(const float* ls_dirs_deg, int nLS, int method, int order, int enableMaxrE, float* decMtx)
{
    const int nSH = NSH(order);
    const float scale = 1.0f / ORC_SQRT4PI;
    switch (method) {
        default:
        case ORC_DECODER_DEFAULT:
        case ORC_DECODER_SAD: {
            float* Y = (float*)malloc(sizeof(float) * nSH * nLS);
            orc_getRSH(order, ls_dirs_deg, nLS, Y);
            for (int i = 0; i < nSH * nLS; i++) Y[i] *= scale;
            for (int i = 0; i < nLS; i++)
                for (int j = 0; j < nSH; j++)
                    decMtx[i * nSH + j] = (4.0f * ORC_PI) * Y[j * nLS + i] / (float)nLS;
            free(Y);
        } break;
        case ORC_DECODER_MMD: {
            float* Y = (float*)malloc(sizeof(float) * nSH * nLS);
            orc_getRSH(order, ls_dirs_deg, nLS, Y);
            for (int i = 0; i < nSH * nLS; i++) Y[i] *= scale;
            orc_pinv(Y, nSH, nLS, decMtx);
            free(Y);
        } break;
        case ORC_DECODER_EPAD: get_epad(order, ls_dirs_deg, nLS, decMtx); break;
        case ORC_DECODER_ALLRAD: get_allrad(order, ls_dirs_deg, nLS, decMtx); break;
    }
    if (enableMaxrE) {
        float* a_n = (float*)malloc(sizeof(float) * nSH);
        orc_getMaxREweights(order, 0, a_n);
        /* right-multiplication by a diagonal matrix (saf_hoa.c:379-391) */
        for (int i = 0; i < nLS; i++) for (int j = 0; j < nSH; j++) decMtx[i * nSH + j] *= a_n[j];
        free(a_n);
    }
}

/* ========================================================================== */
/*            spherical Voronoi integration weights (saf_utility_geometry.c)  */
/* ========================================================================== */

static void cross3f(const float* a, const float* b, float* c)        /* crossProduct3 (saf_utility_geometry.c:452-462) */
{
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
static float norm3f(const float* v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* getVoronoiWeights (saf_utility_geometry.c:937-983): sphDelaunay (:659-691, float trigonometry) -> sphVoronoi
 * (:693-868: Voronoi vertex of a Delaunay triangle = its unit normal; triangles around a point ordered by walking
 * through shared neighbours; vertices closer than 1e-5 per coordinate are merged) -> sphVoronoiAreas (:870-935:
 * spherical excess of the polygon, interior angles from the tangents of the great circles).  The reference's
 * quickhull is replaced by the deterministic hull of this file; the Voronoi cells do not depend on the face order. */
void orc_getVoronoiWeights(const float* dirs_deg, int nDirs, float* weights)
{
    float* V = (float*)malloc(sizeof(float) * 3 * nDirs);
    double* P = (double*)malloc(sizeof(double) * 3 * nDirs);
    for (int i = 0; i < nDirs; i++) {
        V[i * 3 + 2] = sinf(dirs_deg[i * 2 + 1] * ORC_PI / 180.0f);
        const float rcoselev = cosf(dirs_deg[i * 2 + 1] * ORC_PI / 180.0f);
        V[i * 3 + 0] = rcoselev * cosf(dirs_deg[i * 2 + 0] * ORC_PI / 180.0f);
        V[i * 3 + 1] = rcoselev * sinf(dirs_deg[i * 2 + 0] * ORC_PI / 180.0f);
        for (int k = 0; k < 3; k++) P[3 * i + k] = V[3 * i + k];
    }
    int* faces = NULL;
    const int nFaces = sphere_hull(P, nDirs, &faces);
    free(P);
    for (int i = 0; i < nDirs; i++) weights[i] = 0.0f;
    if (!faces) { free(V); return; }
    /* Voronoi vertices */
    float* vert = (float*)malloc(sizeof(float) * 3 * nFaces);
    for (int n = 0; n < nFaces; n++) {
        float r12[3], r13[3], nr[3];
        for (int k = 0; k < 3; k++) { r12[k] = V[faces[n * 3 + 1] * 3 + k] - V[faces[n * 3] * 3 + k]; r13[k] = V[faces[n * 3 + 2] * 3 + k] - V[faces[n * 3] * 3 + k]; }
        cross3f(r12, r13, nr);
        const float inv = 1.0f / norm3f(nr);
        for (int k = 0; k < 3; k++) vert[n * 3 + k] = nr[k] * inv;
    }
    int* dup = (int*)calloc(nFaces, sizeof(int));
    for (int n = 0; n < nFaces; n++)
        if (dup[n] == 0)
            for (int m = 0; m < nFaces; m++)
                if (n != m && fabsf(vert[n * 3] - vert[m * 3]) < 1.0e-5f && fabsf(vert[n * 3 + 1] - vert[m * 3 + 1]) < 1.0e-5f && fabsf(vert[n * 3 + 2] - vert[m * 3 + 2]) < 1.0e-5f)
                    dup[m] = n;
    int* ring = (int*)malloc(sizeof(int) * nFaces);
    int* poly = (int*)malloc(sizeof(int) * nFaces);
    float* theta = (float*)malloc(sizeof(float) * nFaces);
    for (int n = 0; n < nDirs; n++) {
        int nR = 0;
        for (int m = 0; m < nFaces; m++) if (faces[m * 3] == n || faces[m * 3 + 1] == n || faces[m * 3 + 2] == n) ring[nR++] = m;
        if (nR < 3) continue;
        /* walk around the point: next triangle = the unvisited one sharing the current "outer" vertex */
        int cur = ring[0], curv = -1, nS = 0;
        for (int j = 0; j < 3; j++) if (faces[cur * 3 + j] != n) { curv = faces[cur * 3 + j]; break; }
        poly[nS++] = cur;
        while (nS < nR) {
            int found = -1;
            for (int l = 0; l < nR && found < 0; l++) {
                const int f = ring[l];
                if (f == cur) continue;
                int seen = 0; for (int q = 0; q < nS; q++) if (poly[q] == f) seen = 1;
                if (seen) continue;
                if (faces[f * 3] == curv || faces[f * 3 + 1] == curv || faces[f * 3 + 2] == curv) found = f;
            }
            if (found < 0) break;
            poly[nS++] = found;
            for (int j = 0; j < 3; j++) if (faces[found * 3 + j] != n && faces[found * 3 + j] != curv) { curv = faces[found * 3 + j]; break; }
            cur = found;
        }
        /* merge duplicate vertices, keeping first occurrences in walking order */
        int nU = 0;
        for (int i = 0; i < nS; i++) {
            const int id = dup[poly[i]] != 0 ? dup[poly[i]] : poly[i];
            int seen = 0; for (int q = 0; q < nU; q++) if (poly[q] == id) seen = 1;
            if (!seen) poly[nU++] = id;
        }
        if (nU < 3) continue;
        for (int k = 0; k < nU; k++) {
            const float* r01 = &vert[poly[k] * 3]; const float* r02 = &vert[poly[(k + 1) % nU] * 3]; const float* r03 = &vert[poly[(k + 2) % nU] * 3];
            float r2x1[3], r21[3], r2x3[3], r23[3];
            cross3f(r02, r01, r2x1); cross3f(r2x1, r02, r21);
            cross3f(r02, r03, r2x3); cross3f(r2x3, r02, r23);
            const float n21 = 1.0f / norm3f(r21), n23 = 1.0f / norm3f(r23);
            float d = 0.0f;
            for (int q = 0; q < 3; q++) d += (r21[q] * n21) * (r23[q] * n23);
            theta[k] = acosf(d);
        }
        float tmp = 0.0f;
        for (int k = 0; k < nU; k++) tmp += theta[k];
        weights[n] = tmp - ((float)nU - 2.0f) * ORC_PI;
    }
    free(V); free(faces); free(vert); free(dup); free(ring); free(poly); free(theta);
}

/* ---------------- 2-D VBAP (saf_vbap.c:390-473, 898-1024) ---------------- */
/* findLsPairs: ascending azimuth order, neighbouring loudspeakers form the pairs, the last pair wraps around */
void orc_findLsPairs(const float* ls_dirs_deg, int L, int* pairs)
{
    int* idx = (int*)malloc(sizeof(int) * (L + 1));
    for (int n = 0; n < L; n++) idx[n] = n;
    for (int i = 1; i < L; i++) {                      /* insertion sort (stable) */
        int k = idx[i], j = i - 1;
        while (j >= 0 && ls_dirs_deg[idx[j] * 2] > ls_dirs_deg[k * 2]) { idx[j + 1] = idx[j]; j--; }
        idx[j + 1] = k;
    }
    idx[L] = idx[0];
    for (int n = 0; n < L; n++) { pairs[n * 2] = idx[n]; pairs[n * 2 + 1] = idx[n + 1]; }
    free(idx);
}

/* generateVBAPgainTable2D_srcs / vbap2D on S azimuths: gtable is [S][L] (caller-allocated here) */
void orc_vbap2D_table(const float* src_azi_deg, int S, const float* ls_dirs_deg, int L, float* gtable)
{
    int* pairs = (int*)malloc(sizeof(int) * 2 * L);
    orc_findLsPairs(ls_dirs_deg, L, pairs);
    float* inv = (float*)malloc(sizeof(float) * 4 * L);
    for (int n = 0; n < L; n++) {
        /* tempGroup[j*2+i] = U[pair_i][j]: columns = unit vectors of the pair; invertLsMtx2D stores the inverse row-major */
        float m00 = cosf(ls_dirs_deg[pairs[n * 2] * 2] * ORC_PI / 180.0f), m10 = sinf(ls_dirs_deg[pairs[n * 2] * 2] * ORC_PI / 180.0f);
        float m01 = cosf(ls_dirs_deg[pairs[n * 2 + 1] * 2] * ORC_PI / 180.0f), m11 = sinf(ls_dirs_deg[pairs[n * 2 + 1] * 2] * ORC_PI / 180.0f);
        double det = (double)m00 * m11 - (double)m01 * m10;
        inv[n * 4 + 0] = (float)(m11 / det); inv[n * 4 + 1] = (float)(-m01 / det);
        inv[n * 4 + 2] = (float)(-m10 / det); inv[n * 4 + 3] = (float)(m00 / det);
    }
    float* gains = (float*)malloc(sizeof(float) * L);
    for (int ns = 0; ns < S; ns++) {
        float azi = src_azi_deg[ns] * ORC_PI / 180.0f, u0 = cosf(azi), u1 = sinf(azi);
        memset(gains, 0, sizeof(float) * L);
        for (int i = 0; i < L; i++) {
            float g0 = inv[i * 4] * u0 + inv[i * 4 + 1] * u1, g1 = inv[i * 4 + 2] * u0 + inv[i * 4 + 3] * u1;
            float mn = g0 < g1 ? g0 : g1, rms = sqrtf(powf(g0, 2.0f) + powf(g1, 2.0f));
            if (mn > -0.001) { gains[pairs[i * 2]] = g0 / rms; gains[pairs[i * 2 + 1]] = g1 / rms; }
        }
        float e = 0.0f;
        for (int i = 0; i < L; i++) e += powf(gains[i], 2.0f);
        e = sqrtf(e);
        for (int i = 0; i < L; i++) { float v = gains[i] / e; gtable[(size_t)ns * L + i] = v > 0.0f ? v : 0.0f; }
    }
    free(gains); free(inv); free(pairs);
}

/* generateVBAPgainTable2D: azimuth grid -180 : res : 180; returns the number of grid points (gtable NULL: count only) */
int orc_generateVBAPgainTable2D(const float* ls_dirs_deg, int L, int az_res_deg, float* gtable)
{
    const int N_azi = (int)((360.0f / (float)az_res_deg) + 1.5f);
    if (!gtable) return N_azi;
    float* azi = (float*)malloc(sizeof(float) * N_azi);
    float fi; int i;
    for (fi = -180.0f, i = 0; i < N_azi; fi += (float)az_res_deg, i++) azi[i] = fi;
    orc_vbap2D_table(azi, N_azi, ls_dirs_deg, L, gtable);
    free(azi);
    return N_azi;
}

void orc_getSpreadSrcDirs3D(float azi, float elev, float spread, int num_src, int num_rings, float* Us) { spread_dirs(azi, elev, spread, num_src, num_rings, Us); }
