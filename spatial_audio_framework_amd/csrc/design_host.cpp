/*
 * design_host.cpp — init-time design code that stays on the host: small dense
 * linear algebra (SVD / pseudo-inverse), sphere triangulation, VBAP tables and
 * the loudspeaker decoder matrices.  None of this runs per audio block; the
 * spherical-harmonic matrices it needs are evaluated by the GPU kernels in
 * sh_kernels.hip.
 *
 * Reference interfaces replaced (relative to the SAF checkout):
 *   framework/modules/saf_hoa/saf_hoa.c:40-116,235-267,326-392
 *   framework/modules/saf_hoa/saf_hoa_internal.c:41-155
 *   framework/modules/saf_vbap/saf_vbap.c:52-388,499-896
 *   framework/modules/saf_utilities/saf_utility_veclib.c:3466-3560 (utility_spinv)
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "design_host.h"
#include <algorithm>
#include <array>
#include <map>
#include <cstdint>

namespace saf {

void sh_eval_host(int kind, int order, const float* dirs, int nDirs, float* Y);   /* sh_kernels.hip */

/* ========================================================================== */
/*                         SVD (Hestenes one-sided Jacobi)                    */
/* ========================================================================== */
/* Decomposes a tall matrix G (rows >= cols, column-major vectors of length `rows`) in place:
 * on return column j of G is u_j * sigma_j and V accumulates the right rotations. */
static void hestenes(std::vector<std::vector<double>>& G, std::vector<std::vector<double>>& V)
{
    const int cols = (int)G.size();
    const int rows = cols ? (int)G[0].size() : 0;
    V.assign(cols, std::vector<double>(cols, 0.0));
    for (int i = 0; i < cols; i++) V[i][i] = 1.0;
    for (int sweep = 0; sweep < 80; sweep++) {
        bool rotated = false;
        for (int i = 0; i + 1 < cols; i++)
            for (int j = i + 1; j < cols; j++) {
                double aii = 0, ajj = 0, aij = 0;
                for (int r = 0; r < rows; r++) { aii += G[i][r] * G[i][r]; ajj += G[j][r] * G[j][r]; aij += G[i][r] * G[j][r]; }
                if (std::fabs(aij) <= 1e-16 * std::sqrt(aii * ajj) || aij == 0.0) continue;
                rotated = true;
                const double tau = (ajj - aii) / (2.0 * aij);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < rows; r++) { const double gi = G[i][r], gj = G[j][r]; G[i][r] = c * gi - s * gj; G[j][r] = s * gi + c * gj; }
                for (int r = 0; r < cols; r++) { const double vi = V[i][r], vj = V[j][r]; V[i][r] = c * vi - s * vj; V[j][r] = s * vi + c * vj; }
            }
        if (!rotated) break;
    }
}

void thin_svd(const float* M, int r, int c, std::vector<double>& U, std::vector<double>& S, std::vector<double>& Vout)
{
    const bool tall = r >= c;
    const int rows = tall ? r : c, cols = tall ? c : r;
    std::vector<std::vector<double>> G(cols, std::vector<double>(rows));
    for (int j = 0; j < cols; j++)
        for (int i = 0; i < rows; i++) G[j][i] = tall ? (double)M[i * c + j] : (double)M[j * c + i];
    std::vector<std::vector<double>> V;
    hestenes(G, V);
    std::vector<double> sig(cols);
    for (int j = 0; j < cols; j++) { double n2 = 0; for (double v : G[j]) n2 += v * v; sig[j] = std::sqrt(n2); }
    std::vector<int> ord(cols);
    for (int j = 0; j < cols; j++) ord[j] = j;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return sig[a] > sig[b]; });
    const int k = cols;
    /* left vectors of the tall problem (length rows) and right vectors (length cols) */
    std::vector<double> L((size_t)rows * k), R((size_t)cols * k);
    S.resize(k);
    for (int q = 0; q < k; q++) {
        const int j = ord[q];
        S[q] = sig[j];
        for (int i = 0; i < rows; i++) L[(size_t)i * k + q] = sig[j] > 0 ? G[j][i] / sig[j] : 0.0;
        for (int i = 0; i < cols; i++) R[(size_t)i * k + q] = V[j][i];
    }
    if (tall) { U = L; Vout = R; } else { U = R; Vout = L; }     /* M^T = L S R^T  =>  M = R S L^T */
}

/* utility_spinv (saf_utility_veclib.c:3466-3560): out [dim2 x dim1]; singular values <= 1e-5 are
 * multiplied in (not inverted, not zeroed) exactly as the reference does (:3535-3540). */
void pinv_f(const float* inM, int dim1, int dim2, float* outM)
{
    std::vector<double> U, S, V;
    thin_svd(inM, dim1, dim2, U, S, V);
    const int k = (int)S.size();
    for (int j = 0; j < dim2; j++)
        for (int i = 0; i < dim1; i++) {
            double acc = 0;
            for (int q = 0; q < k; q++) {
                const double ss = ((float)S[q] > 1.0e-5f) ? 1.0 / S[q] : S[q];
                acc += V[(size_t)j * k + q] * ss * U[(size_t)i * k + q];
            }
            outM[(size_t)j * dim1 + i] = (float)acc;
        }
}

/* ========================================================================== */
/*                  triangulation of directions on the sphere                 */
/* ========================================================================== */
/* The reference runs quickhull on the unit vectors after adding rand()-scaled 1e-7 noise
 * (convhull_3d.c:400), so its face order — and on layouts with coplanar point groups its face
 * set — changes from call to call.  Here: incremental hull with exact-sign-robust orient tests
 * in double, faces outward-oriented, each rotated to start at its smallest vertex and the list
 * sorted lexicographically, so the result is a function of the input alone. */
namespace {
struct Tri { int a, b, c; bool live; };

inline double orient(const double* P, int a, int b, int c, const double* p)
{
    const double* A = P + 3 * a; const double* B = P + 3 * b; const double* C = P + 3 * c;
    const double ux = B[0] - A[0], uy = B[1] - A[1], uz = B[2] - A[2];
    const double vx = C[0] - A[0], vy = C[1] - A[1], vz = C[2] - A[2];
    double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const double nl = std::sqrt(nx * nx + ny * ny + nz * nz);
    if (nl > 0) { nx /= nl; ny /= nl; nz /= nl; }
    return nx * (p[0] - A[0]) + ny * (p[1] - A[1]) + nz * (p[2] - A[2]);
}
}  // namespace

bool sphere_triangulate(const std::vector<double>& P, std::vector<std::array<int, 3>>& faces)
{
    const int n = (int)P.size() / 3;
    const double tol = 1e-9;
    faces.clear();
    if (n < 4) return false;
    /* seed tetrahedron */
    int s0 = 0, s1 = -1, s2 = -1, s3 = -1;
    double best = -1;
    for (int i = 1; i < n; i++) {
        double d = 0; for (int k = 0; k < 3; k++) d += (P[3 * i + k] - P[k]) * (P[3 * i + k] - P[k]);
        if (d > best) { best = d; s1 = i; }
    }
    best = -1;
    for (int i = 0; i < n; i++) {
        if (i == s0 || i == s1) continue;
        const double ux = P[3 * s1] - P[0], uy = P[3 * s1 + 1] - P[1], uz = P[3 * s1 + 2] - P[2];
        const double vx = P[3 * i] - P[0], vy = P[3 * i + 1] - P[1], vz = P[3 * i + 2] - P[2];
        const double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
        const double d = cx * cx + cy * cy + cz * cz;
        if (d > best) { best = d; s2 = i; }
    }
    best = -1;
    for (int i = 0; i < n; i++) {
        if (i == s0 || i == s1 || i == s2) continue;
        const double d = std::fabs(orient(P.data(), s0, s1, s2, &P[3 * i]));
        if (d > best) { best = d; s3 = i; }
    }
    if (s3 < 0 || best < tol) return false;
    double cen[3];
    for (int k = 0; k < 3; k++) cen[k] = (P[3 * s0 + k] + P[3 * s1 + k] + P[3 * s2 + k] + P[3 * s3 + k]) / 4.0;
    std::vector<Tri> T;
    auto add_oriented = [&](int a, int b, int c) {
        if (orient(P.data(), a, b, c, cen) > 0) std::swap(b, c);
        T.push_back({ a, b, c, true });
    };
    add_oriented(s0, s1, s2); add_oriented(s0, s1, s3); add_oriented(s0, s2, s3); add_oriented(s1, s2, s3);
    std::vector<char> done(n, 0);
    done[s0] = done[s1] = done[s2] = done[s3] = 1;
    for (int p = 0; p < n; p++) {
        if (done[p]) continue;
        done[p] = 1;
        std::vector<int> vis;
        for (int f = 0; f < (int)T.size(); f++)
            if (T[f].live && orient(P.data(), T[f].a, T[f].b, T[f].c, &P[3 * p]) > tol) vis.push_back(f);
        if (vis.empty()) continue;
        /* horizon = directed edges of visible faces whose twin is not on a visible face */
        std::map<std::pair<int, int>, int> edges;
        for (int f : vis) {
            const int v[3] = { T[f].a, T[f].b, T[f].c };
            for (int e = 0; e < 3; e++) edges[{ v[e], v[(e + 1) % 3] }] = f;
        }
        std::vector<std::pair<int, int>> horizon;
        for (auto& kv : edges)
            if (!edges.count({ kv.first.second, kv.first.first })) horizon.push_back(kv.first);
        for (int f : vis) T[f].live = false;
        for (auto& e : horizon) T.push_back({ e.first, e.second, p, true });
        if (T.size() > (size_t)(16 * n + 64)) {       /* drop dead entries now and then */
            std::vector<Tri> L;
            for (auto& t : T) if (t.live) L.push_back(t);
            T.swap(L);
        }
    }
    for (auto& t : T) {
        if (!t.live) continue;
        std::array<int, 3> f = { t.a, t.b, t.c };
        const int r = (int)(std::min_element(f.begin(), f.end()) - f.begin());
        faces.push_back({ f[r], f[(r + 1) % 3], f[(r + 2) % 3] });
    }
    std::sort(faces.begin(), faces.end());
    return !faces.empty();
}

/* ========================================================================== */
/*                                   VBAP                                     */
/* ========================================================================== */

/* findLsTriplets (saf_vbap.c:499-674) */
bool find_ls_triplets(const float* ls_dirs_deg, int L, int omitLargeTriangles, std::vector<float>& verts, std::vector<int>& faces)
{
    verts.resize((size_t)L * 3);
    std::vector<double> P((size_t)L * 3);
    for (int i = 0; i < L; i++) {
        verts[i * 3 + 2] = (float)std::sin((double)ls_dirs_deg[i * 2 + 1] * SAF_PId / 180.0);
        const double rc = std::cos((double)ls_dirs_deg[i * 2 + 1] * SAF_PId / 180.0);
        verts[i * 3 + 0] = (float)(rc * std::cos((double)ls_dirs_deg[i * 2 + 0] * SAF_PId / 180.0));
        verts[i * 3 + 1] = (float)(rc * std::sin((double)ls_dirs_deg[i * 2 + 0] * SAF_PId / 180.0));
        for (int k = 0; k < 3; k++) P[3 * i + k] = verts[i * 3 + k];
    }
    std::vector<std::array<int, 3>> F;
    faces.clear();
    if (!sphere_triangulate(P, F)) return false;
    for (auto& f : F) {
        float v[3][3];
        for (int q = 0; q < 3; q++) for (int j = 0; j < 3; j++) v[q][j] = verts[f[q] * 3 + j];
        float a[3], b[3], cen[3];
        for (int j = 0; j < 3; j++) { a[j] = v[1][j] - v[0][j]; b[j] = v[2][j] - v[1][j]; cen[j] = (v[0][j] + v[1][j] + v[2][j]) / 3.0f; }
        const float cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
        float d = cx * cen[0] + cy * cen[1] + cz * cen[2];
        d = std::max(std::min(d, 0.99999999f), -0.99999999f);
        bool ok = acosf(d) < (SAF_PI / 2.0f);                    /* normal must point away from the origin (:586-609) */
        if (ok && omitLargeTriangles) {
            const float lim = 180.0f * SAF_PI / 180.0f;          /* APERTURE_LIMIT_DEG (saf_vbap_internal.h:50) */
            for (int q = 0; q < 3 && ok; q++) {
                const float* x = v[q]; const float* y = v[(q + 1) % 3];
                ok = acosf(x[0] * y[0] + x[1] * y[1] + x[2] * y[2]) < lim;
            }
        }
        if (ok) { faces.push_back(f[0]); faces.push_back(f[1]); faces.push_back(f[2]); }
    }
    return true;
}

/* invertLsMtx3D (saf_vbap.c:676-705): inverse of the matrix whose columns are the three unit vectors */
void invert_ls_mtx(const float* U, const int* grp, int nGroups, float* inv)
{
    for (int n = 0; n < nGroups; n++) {
        double m[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[j][i] = U[grp[n * 3 + i] * 3 + j];
        const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1])
                         - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
                         + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        const double r = 1.0 / det;
        float* o = inv + n * 9;
        o[0] = (float)((m[1][1] * m[2][2] - m[1][2] * m[2][1]) * r);
        o[1] = (float)((m[0][2] * m[2][1] - m[0][1] * m[2][2]) * r);
        o[2] = (float)((m[0][1] * m[1][2] - m[0][2] * m[1][1]) * r);
        o[3] = (float)((m[1][2] * m[2][0] - m[1][0] * m[2][2]) * r);
        o[4] = (float)((m[0][0] * m[2][2] - m[0][2] * m[2][0]) * r);
        o[5] = (float)((m[0][2] * m[1][0] - m[0][0] * m[1][2]) * r);
        o[6] = (float)((m[1][0] * m[2][1] - m[1][1] * m[2][0]) * r);
        o[7] = (float)((m[0][1] * m[2][0] - m[0][0] * m[2][1]) * r);
        o[8] = (float)((m[0][0] * m[1][1] - m[0][1] * m[1][0]) * r);
    }
}

/* getSpreadSrcDirs3D (saf_vbap.c:707-783): 8 directions on a ring + the source itself */
void spread_ring(float azi, float elev, float spread, int nSrc, int nRings, float* Us)
{
    const float u[3] = { cosf(elev) * cosf(azi), cosf(elev) * sinf(azi), sinf(elev) };
    const float theta = 2.0f * SAF_PI / (float)nSrc, st = sinf(theta), ct = cosf(theta);
    const float ux[3][3] = { { 0.0f, -u[2], u[1] }, { u[2], 0.0f, -u[0] }, { -u[1], u[0], 0.0f } };
    float R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const float outer = (i == j) ? powf(u[i], 2.0f) : u[i] * u[j];
            R[i][j] = st * ux[i][j] + (1.0f - ct) * outer + (i == j ? ct : 0.0f);
        }
    std::vector<float> base((size_t)nSrc * 3, 0.0f);
    if (elev > SAF_PI / 2.0f - 0.01f || elev < -(SAF_PI / 2.0f - 0.01f)) base[0] = 1.0f;
    else {
        const float c[3] = { u[1], -u[0], 0.0f };             /* u x (0,0,1) */
        const float nrm = sqrtf(powf(c[0], 2.0f) + powf(c[1], 2.0f) + powf(c[2], 2.0f));
        for (int i = 0; i < 3; i++) base[i] = c[i] / nrm;
    }
    for (int ns = 1; ns < nSrc; ns++)
        for (int i = 0; i < 3; i++) {
            float acc = 0.0f;
            for (int j = 0; j < 3; j++) acc += R[i][j] * base[(ns - 1) * 3 + j];
            base[ns * 3 + i] = acc;
        }
    const float ring = ((spread / 2.0f) * SAF_PI / 180.0f) / (float)nRings;
    for (int nr = 0; nr < nRings; nr++)
        for (int ns = 0; ns < nSrc; ns++)
            for (int i = 0; i < 3; i++) Us[(nr * nSrc + ns) * 3 + i] = u[i] + base[ns * 3 + i] * tanf(ring * (float)(nr + 1));
    const float n0 = sqrtf(powf(Us[0], 2.0f) + powf(Us[1], 2.0f) + powf(Us[2], 2.0f));
    for (int i = 0; i < nRings * nSrc * 3; i++) Us[i] /= n0;
    for (int i = 0; i < 3; i++) Us[nRings * nSrc * 3 + i] = u[i];
}

/* vbap3D (saf_vbap.c:786-896): the first face (in list order) whose three gains all exceed -0.001 wins */
void vbap_gains(const float* src_dirs_deg, int S, int L, const int* grp, int nFaces, float spread, const float* inv, float* G)
{
    std::vector<float> gains(L);
    const bool mdap = spread > 0.1f;
    const int nDir = mdap ? 9 : 1;
    float Us[27];
    for (int ns = 0; ns < S; ns++) {
        const float azi = src_dirs_deg[ns * 2] * SAF_PI / 180.0f, elev = src_dirs_deg[ns * 2 + 1] * SAF_PI / 180.0f;
        if (mdap) spread_ring(azi, elev, spread, 8, 1, Us);
        else { Us[0] = cosf(azi) * cosf(elev); Us[1] = sinf(azi) * cosf(elev); Us[2] = sinf(elev); }
        std::fill(gains.begin(), gains.end(), 0.0f);
        for (int d = 0; d < nDir; d++) {
            const float* u = Us + 3 * d;
            for (int f = 0; f < nFaces; f++) {
                float g[3], mn = 2.23e13f, e = 0.0f;
                for (int r = 0; r < 3; r++) {
                    const float* row = inv + f * 9 + r * 3;
                    g[r] = row[0] * u[0] + row[1] * u[1] + row[2] * u[2];
                    mn = std::min(mn, g[r]);
                    e += powf(g[r], 2.0f);
                }
                e = sqrtf(e);
                if (mn > -0.001) {
                    if (mdap) for (int j = 0; j < 3; j++) gains[grp[f * 3 + j]] += g[j] / e;
                    else { for (int j = 0; j < 3; j++) gains[grp[f * 3 + j]] = g[j] / e; break; }
                }
            }
        }
        float e = 0.0f;
        for (int i = 0; i < L; i++) e += powf(gains[i], 2.0f);
        e = sqrtf(e);
        for (int i = 0; i < L; i++) G[(size_t)ns * L + i] = std::max(gains[i] / e, 0.0f);
    }
}

/* shared body of generateVBAPgainTable3D_srcs (saf_vbap.c:52-169) and generateVBAPgainTable3D (:171-310) */
bool vbap_table(const float* src_dirs_deg, int S, const float* ls_dirs_deg, int L, int omitLarge, int enableDummies, float spread,
                std::vector<float>& gtable, int* nTriangles)
{
    bool need[2] = { false, false };
    if (enableDummies) {
        need[0] = need[1] = true;
        for (int i = 0; i < L; i++) {
            if (ls_dirs_deg[i * 2 + 1] <= -60.0f) need[0] = false;      /* ADD_DUMMY_LIMIT (saf_vbap_internal.h:46) */
            if (ls_dirs_deg[i * 2 + 1] >= 60.0f) need[1] = false;
        }
    }
    std::vector<float> dirs(ls_dirs_deg, ls_dirs_deg + (size_t)L * 2);
    if (need[0]) { dirs.push_back(0.0f); dirs.push_back(-90.0f); }
    if (need[1]) { dirs.push_back(0.0f); dirs.push_back(90.0f); }
    const int Ld = (int)dirs.size() / 2;
    std::vector<float> verts; std::vector<int> faces;
    if (!find_ls_triplets(dirs.data(), Ld, omitLarge, verts, faces)) { gtable.clear(); *nTriangles = 0; return false; }
    const int nF = (int)faces.size() / 3;
    std::vector<float> inv((size_t)std::max(nF, 1) * 9);
    invert_ls_mtx(verts.data(), faces.data(), nF, inv.data());
    std::vector<float> G((size_t)S * Ld);
    vbap_gains(src_dirs_deg, S, Ld, faces.data(), nF, spread, inv.data(), G.data());
    gtable.resize((size_t)S * L);
    for (int i = 0; i < S; i++) memcpy(&gtable[(size_t)i * L], &G[(size_t)i * Ld], sizeof(float) * L);   /* dummies dropped */
    *nTriangles = nF;
    return true;
}

void vbap_grid_dirs(int az_res_deg, int el_res_deg, std::vector<float>& src)
{
    const int N_azi = (int)((360.0f / (float)az_res_deg) + 1.5f);       /* saf_vbap.c:194-208 */
    const int N_ele = (int)((180.0f / (float)el_res_deg) + 1.5f);
    std::vector<float> azi(N_azi), ele(N_ele);
    float fi; int i;
    for (fi = -180.0f, i = 0; i < N_azi; fi += (float)az_res_deg, i++) azi[i] = fi;
    for (fi = -90.0f, i = 0; i < N_ele; fi += (float)el_res_deg, i++) ele[i] = fi;
    src.resize((size_t)N_azi * N_ele * 2);
    for (i = 0; i < N_ele; i++)
        for (int j = 0; j < N_azi; j++) { src[(i * N_azi + j) * 2] = azi[j]; src[(i * N_azi + j) * 2 + 1] = ele[i]; }
}

/* ========================================================================== */
/*                           HOA helpers + decoders                           */
/* ========================================================================== */

/* getMaxREweights (saf_hoa.c:235-267): a_n = P_n(cos(137.9 deg / (N + 1.51))), argument evaluated in float */
void maxre_weights(int order, std::vector<float>& a)
{
    const double x = cosf(137.9f * (SAF_PI / 180.0f) / ((float)order + 1.51f));
    a.assign(ORDER2NSH(order), 0.0f);
    double pm1 = 1.0, p = x;      /* Bonnet recursion for the Legendre polynomials */
    int idx = 0;
    for (int n = 0; n <= order; n++) {
        double pn;
        if (n == 0) pn = 1.0;
        else if (n == 1) pn = x;
        else { pn = ((2.0 * n - 1.0) * x * p - (n - 1.0) * pm1) / (double)n; pm1 = p; p = pn; }
        for (int i = 0; i < 2 * n + 1; i++) a[idx + i] = (float)pn;
        idx += 2 * n + 1;
    }
}

/* getEPAD (saf_hoa_internal.c:41-98): both truncation branches equal V_k U_k^T with k = min(nSH, nLS) */
static void epad(int order, const float* ls_dirs_deg, int nLS, float* dec)
{
    const int nSH = ORDER2NSH(order);
    std::vector<float> Y((size_t)nSH * nLS);
    sh_eval_host(1, order, ls_dirs_deg, nLS, Y.data());
    for (auto& v : Y) v *= 1.0f / SAF_SQRT4PI;
    std::vector<double> U, S, V;
    thin_svd(Y.data(), nSH, nLS, U, S, V);
    const int k = (int)S.size();
    const float scale = sqrtf(4.0f * SAF_PI / (float)nLS);
    for (int i = 0; i < nLS; i++)
        for (int j = 0; j < nSH; j++) {
            double acc = 0;
            for (int q = 0; q < k; q++) acc += V[(size_t)i * k + q] * U[(size_t)j * k + q];
            dec[(size_t)i * nSH + j] = (float)acc * scale;
        }
}

/* getAllRAD (saf_hoa_internal.c:100-155): VBAP gains of the 5100-point t-design times its SH matrix */
static void allrad(int order, const float* ls_dirs_deg, int nLS, float* dec)
{
    const int nSH = ORDER2NSH(order), nT = 5100;
    const float* t_dirs = table_required("Tdesign_degree_100_dirs_deg", nT * 2);
    std::vector<float> G; int nTri = 0;
    if (!vbap_table(t_dirs, nT, ls_dirs_deg, nLS, 0, 0, 0.0f, G, &nTri))
        SAF_FATAL("getLoudspeakerDecoderMtx(AllRAD): the loudspeaker directions could not be triangulated");
    std::vector<float> Y((size_t)nSH * nT);
    sh_eval_host(1, order, t_dirs, nT, Y.data());
    for (auto& v : Y) v *= 1.0f / SAF_SQRT4PI;
    const float sc = (4.0f * SAF_PI) / (float)nT;
    for (int i = 0; i < nLS; i++)
        for (int j = 0; j < nSH; j++) {
            float acc = 0.0f;
            for (int t = 0; t < nT; t++) acc += G[(size_t)t * nLS + i] * Y[(size_t)j * nT + t];
            dec[(size_t)i * nSH + j] = acc * sc;
        }
}

/* getLoudspeakerDecoderMtx (saf_hoa.c:326-392) */
void decoder_matrix(const float* ls_dirs_deg, int nLS, int method, int order, int maxrE, float* dec)
{
    const int nSH = ORDER2NSH(order);
    switch (method) {
        default:
        case LOUDSPEAKER_DECODER_DEFAULT:
        case LOUDSPEAKER_DECODER_SAD: {
            std::vector<float> Y((size_t)nSH * nLS);
            sh_eval_host(1, order, ls_dirs_deg, nLS, Y.data());
            for (auto& v : Y) v *= 1.0f / SAF_SQRT4PI;
            for (int i = 0; i < nLS; i++)
                for (int j = 0; j < nSH; j++) dec[(size_t)i * nSH + j] = (4.0f * SAF_PI) * Y[(size_t)j * nLS + i] / (float)nLS;
        } break;
        case LOUDSPEAKER_DECODER_MMD: {
            std::vector<float> Y((size_t)nSH * nLS);
            sh_eval_host(1, order, ls_dirs_deg, nLS, Y.data());
            for (auto& v : Y) v *= 1.0f / SAF_SQRT4PI;
            pinv_f(Y.data(), nSH, nLS, dec);
        } break;
        case LOUDSPEAKER_DECODER_EPAD: epad(order, ls_dirs_deg, nLS, dec); break;
        case LOUDSPEAKER_DECODER_ALLRAD: allrad(order, ls_dirs_deg, nLS, dec); break;
    }
    if (maxrE) {
        std::vector<float> a; maxre_weights(order, a);
        for (int i = 0; i < nLS; i++) for (int j = 0; j < nSH; j++) dec[(size_t)i * nSH + j] *= a[j];
    }
}

}  // namespace saf

using namespace saf;

extern "C" {

void getMaxREweights(int order, int diagMtxFlag, float* a_n)
{
    std::vector<float> a; maxre_weights(order, a);
    const int nSH = ORDER2NSH(order);
    if (diagMtxFlag) { memset(a_n, 0, sizeof(float) * nSH * nSH); for (int i = 0; i < nSH; i++) a_n[i * nSH + i] = a[i]; }
    else memcpy(a_n, a.data(), sizeof(float) * nSH);
}

void getLoudspeakerDecoderMtx(float* ls_dirs_deg, int nLS, LOUDSPEAKER_AMBI_DECODER_METHODS method, int order, int enableMaxReWeighting, float* decMtx)
{
    decoder_matrix(ls_dirs_deg, nLS, (int)method, order, enableMaxReWeighting, decMtx);
}

/* saf_hoa.c:40-70 (host arrays; the block path folds this into the analysis kernel's channel map) */
void convertHOAChannelConvention(float* insig, int order, int len, HOA_CH_ORDER inC, HOA_CH_ORDER outC)
{
    if (order == 0 || inC == outC) return;
    auto swap_rows = [&](int a, int b) { for (int i = 0; i < len; i++) std::swap(insig[a * len + i], insig[b * len + i]); };
    if (inC == HOA_CH_ORDER_FUMA && outC == HOA_CH_ORDER_ACN) { swap_rows(1, 3); swap_rows(1, 2); }
    else if (inC == HOA_CH_ORDER_ACN && outC == HOA_CH_ORDER_FUMA) { swap_rows(1, 2); swap_rows(1, 3); }
    for (int i = 4; i < ORDER2NSH(order); i++) memset(&insig[i * len], 0, sizeof(float) * len);
}

/* saf_hoa.c:72-116 */
void convertHOANormConvention(float* insig, int order, int len, HOA_NORM inC, HOA_NORM outC)
{
    if (order == 0 || inC == outC) return;
    auto scal = [&](int ch, float s) { for (int i = 0; i < len; i++) insig[ch * len + i] *= s; };
    if (inC == HOA_NORM_N3D && outC == HOA_NORM_SN3D) {
        for (int n = 0; n <= order; n++) for (int ch = n * n; ch < ORDER2NSH(n); ch++) scal(ch, 1.0f / sqrtf(2.0f * (float)n + 1.0f));
    } else if (inC == HOA_NORM_N3D && outC == HOA_NORM_FUMA) {
        scal(0, 1.0f / sqrtf(2.0f));
        for (int ch = 1; ch < 4; ch++) scal(ch, 1.0f / sqrtf(3.0f));
    } else if (inC == HOA_NORM_SN3D && outC == HOA_NORM_N3D) {
        for (int n = 0; n <= order; n++) for (int ch = n * n; ch < ORDER2NSH(n); ch++) scal(ch, sqrtf(2.0f * (float)n + 1.0f));
    } else if (inC == HOA_NORM_SN3D && outC == HOA_NORM_FUMA) {
        scal(0, 1.0f / sqrtf(2.0f));
    } else if (inC == HOA_NORM_FUMA && outC == HOA_NORM_N3D) {
        scal(0, sqrtf(2.0f));
        for (int ch = 1; ch < 4; ch++) scal(ch, sqrtf(3.0f));
    } else if (inC == HOA_NORM_FUMA && outC == HOA_NORM_SN3D) {
        scal(0, sqrtf(2.0f));
    }
}

/* ---- VBAP C API: out-params are malloc'd here and free()'d by the caller, as in the reference ---- */
void findLsTriplets(float* ls_dirs_deg, int L, int omitLargeTriangles, float** out_vertices, int* numOutVertices, int** out_faces, int* numOutFaces)
{
    std::vector<float> v; std::vector<int> f;
    const bool ok = find_ls_triplets(ls_dirs_deg, L, omitLargeTriangles, v, f);
    *numOutVertices = L;
    *out_vertices = (float*)malloc(sizeof(float) * 3 * (size_t)std::max(L, 1));
    memcpy(*out_vertices, v.data(), sizeof(float) * 3 * (size_t)L);
    if (!ok) { *out_faces = nullptr; *numOutFaces = 0; return; }
    *numOutFaces = (int)f.size() / 3;
    *out_faces = (int*)malloc(sizeof(int) * std::max<size_t>(f.size(), 1));
    memcpy(*out_faces, f.data(), sizeof(int) * f.size());
}

void invertLsMtx3D(float* U_spkr, int* ls_groups, int N_group, float** layoutInvMtx)
{
    *layoutInvMtx = (float*)malloc(sizeof(float) * 9 * (size_t)std::max(N_group, 1));
    invert_ls_mtx(U_spkr, ls_groups, N_group, *layoutInvMtx);
}

void vbap3D(float* src_dirs, int src_num, int ls_num, int* ls_groups, int nFaces, float spread, float* layoutInvMtx, float** GainMtx)
{
    *GainMtx = (float*)malloc(sizeof(float) * (size_t)std::max(src_num, 1) * ls_num);
    vbap_gains(src_dirs, src_num, ls_num, ls_groups, nFaces, spread, layoutInvMtx, *GainMtx);
}

void generateVBAPgainTable3D_srcs(float* src_dirs_deg, int S, float* ls_dirs_deg, int L, int omitLargeTriangles, int enableDummies, float spread,
                                  float** gtable, int* N_gtable, int* nTriangles)
{
    std::vector<float> G;
    if (!vbap_table(src_dirs_deg, S, ls_dirs_deg, L, omitLargeTriangles, enableDummies, spread, G, nTriangles)) { *gtable = nullptr; *N_gtable = 0; return; }
    *gtable = (float*)malloc(sizeof(float) * G.size());
    memcpy(*gtable, G.data(), sizeof(float) * G.size());
    *N_gtable = S;
}

void generateVBAPgainTable3D(float* ls_dirs_deg, int L, int az_res_deg, int el_res_deg, int omitLargeTriangles, int enableDummies, float spread,
                             float** gtable, int* N_gtable, int* nTriangles)
{
    std::vector<float> src; vbap_grid_dirs(az_res_deg, el_res_deg, src);
    const int S = (int)src.size() / 2;
    std::vector<float> G;
    if (!vbap_table(src.data(), S, ls_dirs_deg, L, omitLargeTriangles, enableDummies, spread, G, nTriangles)) { *gtable = nullptr; *N_gtable = 0; return; }
    *gtable = (float*)malloc(sizeof(float) * G.size());
    memcpy(*gtable, G.data(), sizeof(float) * G.size());
    *N_gtable = S;
}

/* saf_vbap.c:312-367: keep the (up to 3) gains above 1e-7 in ascending loudspeaker order, amplitude-normalised */
void compressVBAPgainTable3D(float* vbap_gtable, int nTable, int nDirs, float* comp, int* idx)
{
    memset(comp, 0, sizeof(float) * 3 * (size_t)nTable);
    memset(idx, 0, sizeof(int) * 3 * (size_t)nTable);
    for (int nt = 0; nt < nTable; nt++) {
        float g[3] = { 0, 0, 0 }, sum = 0.0f; int id[3] = { 0, 0, 0 }, j = 0;
        for (int i = 0; i < nDirs && j < 3; i++) {
            const float v = vbap_gtable[(size_t)nt * nDirs + i];
            if (v > 0.0000001f) { g[j] = v; sum += v; id[j] = i; j++; }
        }
        for (int i = 0; i < j; i++) { comp[nt * 3 + i] = std::max(g[i] / sum, 0.0f); idx[nt * 3 + i] = id[i]; }
    }
}

/* saf_vbap.c:369-388 */
void VBAPgainTable2InterpTable(float* vbap_gtable, int nTable, int nDirs)
{
    for (int i = 0; i < nTable; i++) {
        float s = 0.0f;
        for (int j = 0; j < nDirs; j++) s += vbap_gtable[(size_t)i * nDirs + j];
        for (int j = 0; j < nDirs; j++) vbap_gtable[(size_t)i * nDirs + j] /= s;
    }
}


/* ---------------- 2-D (horizontal) VBAP and the spread ring (saf_vbap.h:277-306, 360-430) ---------------- */
/* findLsPairs (saf_vbap.c:898-928): neighbours after sorting the azimuths; L pairs, the last one closes the circle */
void findLsPairs(float* ls_dirs_deg, int L, int** out_pairs, int* numOutPairs)
{
    std::vector<int> order(L);
    for (int n = 0; n < L; n++) order[n] = n;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ls_dirs_deg[a * 2] < ls_dirs_deg[b * 2]; });
    *out_pairs = (int*)malloc(sizeof(int) * 2 * (size_t)std::max(L, 1));
    for (int n = 0; n < L; n++) { (*out_pairs)[n * 2] = order[n]; (*out_pairs)[n * 2 + 1] = order[(n + 1) % L]; }
    *numOutPairs = L;
}

/* invertLsMtx2D (saf_vbap.c:930-960): inverse of the 2 x 2 matrix whose columns are the pair's unit vectors */
void invertLsMtx2D(float* U_spkr, int* ls_pairs, int N_pairs, float** layoutInvMtx)
{
    *layoutInvMtx = (float*)malloc(sizeof(float) * 4 * (size_t)std::max(N_pairs, 1));
    for (int n = 0; n < N_pairs; n++) {
        const double a = U_spkr[ls_pairs[n * 2] * 2], c = U_spkr[ls_pairs[n * 2] * 2 + 1];           /* first column  */
        const double b = U_spkr[ls_pairs[n * 2 + 1] * 2], d = U_spkr[ls_pairs[n * 2 + 1] * 2 + 1];   /* second column */
        const double r = 1.0 / (a * d - b * c);
        float* o = *layoutInvMtx + n * 4;
        o[0] = (float)(d * r); o[1] = (float)(-b * r); o[2] = (float)(-c * r); o[3] = (float)(a * r);
    }
}

/* vbap2D (saf_vbap.c:962-1024): every pair whose two gains exceed -0.001 writes its (pair-normalised) gains — a later pair
 * overwrites an earlier one on a shared loudspeaker —, then unit energy over all loudspeakers and clipping at 0.
 * src_dirs holds AZIMUTHS only (one float per source), as the reference reads it. */
void vbap2D(float* src_dirs, int src_num, int ls_num, int* ls_pairs, int N_pairs, float* layoutInvMtx, float** GainMtx)
{
    *GainMtx = (float*)malloc(sizeof(float) * (size_t)std::max(src_num, 1) * std::max(ls_num, 1));
    std::vector<float> gains(ls_num);
    for (int ns = 0; ns < src_num; ns++) {
        const float azi = src_dirs[ns] * SAF_PI / 180.0f;
        const float u[2] = { cosf(azi), sinf(azi) };
        std::fill(gains.begin(), gains.end(), 0.0f);
        for (int i = 0; i < N_pairs; i++) {
            float g[2];
            g[0] = layoutInvMtx[i * 4 + 0] * u[0] + layoutInvMtx[i * 4 + 1] * u[1];
            g[1] = layoutInvMtx[i * 4 + 2] * u[0] + layoutInvMtx[i * 4 + 3] * u[1];
            const float mn = std::min(g[0], g[1]);
            const float rms = sqrtf(powf(g[0], 2.0f) + powf(g[1], 2.0f));
            if (mn > -0.001f)
                for (int j = 0; j < 2; j++) gains[ls_pairs[i * 2 + j]] = g[j] / rms;
        }
        float e = 0.0f;
        for (int i = 0; i < ls_num; i++) e += powf(gains[i], 2.0f);
        e = sqrtf(e);
        for (int i = 0; i < ls_num; i++) (*GainMtx)[(size_t)ns * ls_num + i] = std::max(gains[i] / e, 0.0f);
    }
}

static void vbap2d_table(float* src_azi, int S, float* ls_dirs_deg, int L, float** gtable, int* N_gtable, int* nPairs)
{
    int* pairs = nullptr; int nP = 0;
    findLsPairs(ls_dirs_deg, L, &pairs, &nP);
    std::vector<float> verts((size_t)L * 2);
    for (int i = 0; i < L; i++) { verts[i * 2] = cosf(ls_dirs_deg[i * 2] * SAF_PI / 180.0f); verts[i * 2 + 1] = sinf(ls_dirs_deg[i * 2] * SAF_PI / 180.0f); }
    float* inv = nullptr;
    invertLsMtx2D(verts.data(), pairs, nP, &inv);
    vbap2D(src_azi, S, L, pairs, nP, inv, gtable);
    *nPairs = nP; *N_gtable = S;
    free(pairs); free(inv);
}

/* generateVBAPgainTable2D_srcs (saf_vbap.c:390-426): src_dirs_deg is handed to vbap2D as it is, i.e. read as S azimuths */
void generateVBAPgainTable2D_srcs(float* src_dirs_deg, int S, float* ls_dirs_deg, int L, float** gtable, int* N_gtable, int* nPairs)
{
    vbap2d_table(src_dirs_deg, S, ls_dirs_deg, L, gtable, N_gtable, nPairs);
}

/* generateVBAPgainTable2D (saf_vbap.c:428-473): azimuth grid -180 : res : 180 */
void generateVBAPgainTable2D(float* ls_dirs_deg, int L, int az_res_deg, float** gtable, int* N_gtable, int* nPairs)
{
    const int N_azi = (int)((360.0f / (float)az_res_deg) + 1.5f);
    std::vector<float> azi(N_azi);
    float fi = -180.0f;
    for (int i = 0; i < N_azi; fi += (float)az_res_deg, i++) azi[i] = fi;
    vbap2d_table(azi.data(), N_azi, ls_dirs_deg, L, gtable, N_gtable, nPairs);
}

/* getSpreadSrcDirs3D (saf_vbap.c:707-783): U_spread is [(num_rings_3d * num_src + 1) x 3] */
void getSpreadSrcDirs3D(float src_azi_rad, float src_elev_rad, float spread, int num_src, int num_rings_3d, float* U_spread)
{
    saf::spread_ring(src_azi_rad, src_elev_rad, spread, num_src, num_rings_3d, U_spread);
}

}
