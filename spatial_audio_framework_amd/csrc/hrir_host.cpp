/*
 * hrir_host.cpp — init-time HRIR/HRTF processing of saf_hrir (framework/modules/saf_hrir/saf_hrir.c) and the
 * spherical-Voronoi integration weights (framework/modules/saf_utilities/saf_utility_geometry.c:659-983).
 * These run once per HRIR set (host C++); the FIR -> filterbank conversion itself is the GPU path of
 * afSTFT_FIRtoFilterbankCoeffs (afstft_api.cpp).
 *
 * The reference's default HRIR set (saf_default_hrirs.c) is absent from the reference checkout
 * (.MISSING_LARGE_BLOBS), so this library ships none: callers install one with saf_hip_setDefaultHRIRs.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "design_host.h"
#include "hrir_host.h"
#include <algorithm>

/* The spherical excess of a Voronoi cell is a difference of numbers ~1e3 times larger than the cell area, so its float32
 * value depends on every rounding: keep multiply and add separate (what a plain C build of the reference does, and
 * what the parity tests assume). */
#pragma clang fp contract(off)

namespace saf {

static DefaultHRIRs g_default;
const DefaultHRIRs& default_hrirs() { return g_default; }

static inline float matlab_fmodf(float x, float y) { float t = fmodf(x, y); return t >= 0 ? t : t + y; }

static inline void cross3(const float* a, const float* b, float* c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; }
static inline float norm3(const float* v) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* getVoronoiWeights (saf_utility_geometry.c:937-983) with diagFLAG = 0 */
void voronoi_weights(const float* dirs_deg, int nDirs, float* weights)
{
    std::vector<float> V((size_t)nDirs * 3);
    std::vector<double> P((size_t)nDirs * 3);
    for (int i = 0; i < nDirs; i++) {                       /* sphDelaunay (:659-691): float trigonometry */
        V[i * 3 + 2] = sinf(dirs_deg[i * 2 + 1] * SAF_PI / 180.0f);
        const float rc = cosf(dirs_deg[i * 2 + 1] * SAF_PI / 180.0f);
        V[i * 3 + 0] = rc * cosf(dirs_deg[i * 2 + 0] * SAF_PI / 180.0f);
        V[i * 3 + 1] = rc * sinf(dirs_deg[i * 2 + 0] * SAF_PI / 180.0f);
        for (int k = 0; k < 3; k++) P[3 * i + k] = V[3 * i + k];
    }
    std::fill(weights, weights + nDirs, 0.0f);
    std::vector<std::array<int, 3>> F;
    if (!sphere_triangulate(P, F)) return;
    const int nF = (int)F.size();
    /* sphVoronoi (:693-868): the Voronoi vertex of a Delaunay triangle on the unit sphere is its unit normal */
    std::vector<float> vert((size_t)nF * 3);
    for (int n = 0; n < nF; n++) {
        float r12[3], r13[3], nr[3];
        for (int k = 0; k < 3; k++) { r12[k] = V[F[n][1] * 3 + k] - V[F[n][0] * 3 + k]; r13[k] = V[F[n][2] * 3 + k] - V[F[n][0] * 3 + k]; }
        cross3(r12, r13, nr);
        const float inv = 1.0f / norm3(nr);
        for (int k = 0; k < 3; k++) vert[n * 3 + k] = nr[k] * inv;
    }
    std::vector<int> dup(nF, 0);
    for (int n = 0; n < nF; n++)
        if (dup[n] == 0)
            for (int m = 0; m < nF; m++)
                if (n != m && fabsf(vert[n * 3] - vert[m * 3]) < 1.0e-5f && fabsf(vert[n * 3 + 1] - vert[m * 3 + 1]) < 1.0e-5f && fabsf(vert[n * 3 + 2] - vert[m * 3 + 2]) < 1.0e-5f)
                    dup[m] = n;
    /* triangles around every point */
    std::vector<std::vector<int>> around(nDirs);
    for (int m = 0; m < nF; m++) for (int q = 0; q < 3; q++) around[F[m][q]].push_back(m);
    std::vector<int> poly; std::vector<float> theta;
    for (int n = 0; n < nDirs; n++) {
        const std::vector<int>& ring = around[n];
        const int nR = (int)ring.size();
        if (nR < 3) continue;
        poly.clear();
        int cur = ring[0], curv = -1;
        for (int j = 0; j < 3; j++) if (F[cur][j] != n) { curv = F[cur][j]; break; }
        poly.push_back(cur);
        while ((int)poly.size() < nR) {                     /* walk through the neighbours sharing the outer vertex */
            int found = -1;
            for (int f : ring) {
                if (f == cur || std::find(poly.begin(), poly.end(), f) != poly.end()) continue;
                if (F[f][0] == curv || F[f][1] == curv || F[f][2] == curv) { found = f; break; }
            }
            if (found < 0) break;
            poly.push_back(found);
            for (int j = 0; j < 3; j++) if (F[found][j] != n && F[found][j] != curv) { curv = F[found][j]; break; }
            cur = found;
        }
        std::vector<int> uniq;
        for (int f : poly) { const int id = dup[f] != 0 ? dup[f] : f; if (std::find(uniq.begin(), uniq.end(), id) == uniq.end()) uniq.push_back(id); }
        const int nU = (int)uniq.size();
        if (nU < 3) continue;
        /* sphVoronoiAreas (:870-935): spherical excess, interior angles between the great-circle tangents */
        float tmp = 0.0f;
        for (int k = 0; k < nU; k++) {
            const float* r01 = &vert[uniq[k] * 3]; const float* r02 = &vert[uniq[(k + 1) % nU] * 3]; const float* r03 = &vert[uniq[(k + 2) % nU] * 3];
            float a[3], r21[3], b[3], r23[3];
            cross3(r02, r01, a); cross3(a, r02, r21);
            cross3(r02, r03, b); cross3(b, r02, r23);
            const float n21 = 1.0f / norm3(r21), n23 = 1.0f / norm3(r23);
            float d = 0.0f;
            for (int q = 0; q < 3; q++) d += (r21[q] * n21) * (r23[q] * n23);
            tmp += acosf(d);
        }
        weights[n] = tmp - ((float)nU - 2.0f) * SAF_PI;
    }
}

}  // namespace saf

using namespace saf;

extern "C" {

void saf_hip_setDefaultHRIRs(const float* hrirs, const float* hrir_dirs_deg, int N_hrir_dirs, int hrir_len, int hrir_fs)
{
    if (N_hrir_dirs < 4 || hrir_len < 1) SAF_FATAL("saf_hip_setDefaultHRIRs: need at least 4 directions and 1 tap");
    /* installing the set that is already installed changes nothing (tables derived from it stay valid and shared) */
    if (g_default.N == N_hrir_dirs && g_default.len == hrir_len && g_default.fs == hrir_fs &&
        memcmp(g_default.hrirs.data(), hrirs, sizeof(float) * (size_t)N_hrir_dirs * 2 * hrir_len) == 0 &&
        memcmp(g_default.dirs_deg.data(), hrir_dirs_deg, sizeof(float) * (size_t)N_hrir_dirs * 2) == 0) return;
    g_default.hrirs.assign(hrirs, hrirs + (size_t)N_hrir_dirs * 2 * hrir_len);
    g_default.dirs_deg.assign(hrir_dirs_deg, hrir_dirs_deg + (size_t)N_hrir_dirs * 2);
    g_default.N = N_hrir_dirs; g_default.len = hrir_len; g_default.fs = hrir_fs;
    g_default.epoch++;
}

/* estimateITDs (saf_hrir.c:40-108) */
void estimateITDs(float* hrirs, int N_dirs, int hrir_len, int fs, float* itds_s)
{
    const float fc = 750.0f, Q = 0.7071f;                  /* 2nd-order low-pass, DAFX (2nd ed.) p50 */
    const float K = tanf(SAF_PI * fc / (float)fs), KK = K * K, D = KK * Q + K + Q;
    const float b[3] = { (KK * Q) / D, (2.0f * KK * Q) / D, (KK * Q) / D };
    const float a[3] = { 1.0f, (2.0f * Q * (KK - 1.0f)) / D, (KK * Q - K + Q) / D };
    const int xl = 2 * hrir_len - 1;
    const float bound = sqrtf(2.0f) / 2e3f;
    std::vector<float> xc(xl), L(hrir_len), R(hrir_len);
    for (int i = 0; i < N_dirs; i++) {
        float Wz1[2] = { 0, 0 }, Wz2[2] = { 0, 0 };
        for (int n = 0; n < hrir_len; n++)
            for (int j = 0; j < 2; j++) {                   /* biquad, direct form 2 */
                const float wn = hrirs[((size_t)i * 2 + j) * hrir_len + n] - a[1] * Wz1[j] - a[2] * Wz2[j];
                const float y = b[0] * wn + b[1] * Wz1[j] + b[2] * Wz2[j];
                (j == 0 ? L : R)[n] = y;
                Wz2[j] = Wz1[j]; Wz1[j] = wn;
            }
        std::fill(xc.begin(), xc.end(), 0.0f);              /* cxcorr (saf_utility_misc.c:193-223) */
        for (int m = 1; m <= xl; m++) {
            const int arg = m - hrir_len;
            const int lim = arg < 0 ? hrir_len + arg : hrir_len - arg;
            float acc = 0.0f;
            for (int n = 1; n <= lim; n++) acc += arg >= 0 ? L[arg + n - 1] * R[n - 1] : L[n - 1] * R[n - arg - 1];
            xc[m - 1] = acc;
        }
        int maxIdx = 0; float maxVal = 0.0f;
        for (int j = 0; j < xl; j++) if (xc[j] > maxVal) { maxIdx = j; maxVal = xc[j]; }
        float v = ((float)hrir_len - (float)maxIdx - 1.0f) / (float)fs;
        v = v > bound ? bound : v; v = v < -bound ? -bound : v;
        itds_s[i] = v;
    }
}

/* HRIRs2HRTFs_afSTFT (saf_hrir.c:110-123) */
void HRIRs2HRTFs_afSTFT(float* hrirs, int N_dirs, int hrir_len, int hopsize, int LDmode, int hybridmode, float_complex* hrtf_fb)
{
    afSTFT_FIRtoFilterbankCoeffs(hrirs, N_dirs, 2, hrir_len, hopsize, LDmode, hybridmode, hrtf_fb);
}

/* diffuseFieldEqualiseHRTFs (saf_hrir.c:173-239) */
void diffuseFieldEqualiseHRTFs(int N_dirs, float* itds_s, float* centreFreq, int N_bands, float* weights, int applyEQ, int applyPhase, float_complex* hrtfs)
{
    if (!(applyEQ + applyPhase)) return;
    float2* H = reinterpret_cast<float2*>(hrtfs);
    if (applyEQ) {
        for (int band = 0; band < N_bands; band++)
            for (int e = 0; e < 2; e++) {
                float2* h = &H[((size_t)band * 2 + e) * N_dirs];
                float acc = 0.0f;
                for (int j = 0; j < N_dirs; j++) {
                    const float w = weights ? weights[j] : 4.f * SAF_PI / (float)N_dirs;
                    acc += w / (4.f * SAF_PI) * powf(hypotf(h[j].x, h[j].y), 2.0f);
                }
                const float d = sqrtf(acc > 0.00001f ? acc : 0.00001f) + 2.23e-8f;
                for (int j = 0; j < N_dirs; j++) { h[j].x /= d; h[j].y /= d; }
            }
    }
    if (applyPhase) {       /* complex HRTFs from the magnitudes and the interaural phase differences */
        for (int band = 0; band < N_bands; band++)
            for (int nd = 0; nd < N_dirs; nd++) {
                const float ipd = (matlab_fmodf(2.0f * SAF_PI * (centreFreq[band] * itds_s[nd]) + SAF_PI, 2.0f * SAF_PI) - SAF_PI) / 2.0f;
                float2& l = H[((size_t)band * 2 + 0) * N_dirs + nd]; float2& r = H[((size_t)band * 2 + 1) * N_dirs + nd];
                const float ml = hypotf(l.x, l.y), mr = hypotf(r.x, r.y);
                l = make_float2(cosf(ipd) * ml, sinf(ipd) * ml);
                r = make_float2(cosf(-ipd) * mr, sinf(-ipd) * mr);
            }
    }
}

/* getVoronoiWeights (saf_utility_geometry.c:937-983) */
void getVoronoiWeights(float* dirs_deg, int nDirs, int diagFLAG, float* weights)
{
    if (!diagFLAG) { voronoi_weights(dirs_deg, nDirs, weights); return; }
    std::vector<float> w(nDirs);
    voronoi_weights(dirs_deg, nDirs, w.data());
    memset(weights, 0, sizeof(float) * (size_t)nDirs * nDirs);
    for (int i = 0; i < nDirs; i++) weights[(size_t)i * nDirs + i] = w[i];
}

}
