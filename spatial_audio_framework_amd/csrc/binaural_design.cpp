/*
 * binaural_design.cpp — host-side design of the binaural Ambisonic decoders:
 *   getSHrotMtxReal (saf_sh.h; saf_sh.c:479-560 — Ivanic & Ruedenberg's recursion for real SH rotation matrices),
 *   getBinauralAmbiDecoderMtx with its five decoders and applyDiffCovMatching (saf_hoa.h:394-450 / saf_hoa.c:394-603,
 *   saf_hoa_internal.c:162-623), truncationEQ (saf_hoa.c:269-324).
 * Init-time work on small matrices: plain C++ in float64 (the reference calls single-precision BLAS / LAPACK).
 */
#include "binaural_design.h"
#include "../../include/saf_hip.h"
#include "design_host.h"
#include <complex>

namespace saf {

typedef std::complex<double> zd;
typedef std::complex<float> zf;

/* ------------------------------------------------------------------ SH rotation */
namespace {
struct RotRec {
    int M; float R1[3][3]; std::vector<float> prev;          /* prev: band l-1, row stride M */
    float P(int i, int l, int a, int b) const
    {
        const float ri1 = R1[i + 1][2], rim1 = R1[i + 1][0], ri0 = R1[i + 1][1];
        const float* row = &prev[(size_t)(a + l - 1) * M];
        if (b == -l) return ri1 * row[0] + rim1 * row[2 * l - 2];
        if (b == l) return ri1 * row[2 * l - 2] - rim1 * row[0];
        return ri0 * row[b + l - 1];
    }
    float V(int l, int m, int n) const
    {
        if (m == 0) return P(1, l, 1, n) + P(-1, l, -1, n);
        if (m > 0) { const float d = m == 1 ? 1.0f : 0.0f; return P(1, l, m - 1, n) * sqrtf(1.0f + d) - P(-1, l, -m + 1, n) * (1.0f - d); }
        const float d = m == -1 ? 1.0f : 0.0f;
        return P(1, l, m + 1, n) * (1.0f - d) + P(-1, l, -m - 1, n) * sqrtf(1.0f + d);
    }
    float W(int l, int m, int n) const
    {
        if (m == 0) return 0.0f;
        return m > 0 ? P(1, l, m + 1, n) + P(-1, l, -m - 1, n) : P(1, l, m - 1, n) - P(-1, l, -m + 1, n);
    }
};
}  // namespace

void sh_rot_matrix_real(const float Rxyz[3][3], float* RotMtx, int L)
{
    const int M = (L + 1) * (L + 1);
    std::fill(RotMtx, RotMtx + (size_t)M * M, 0.0f);
    RotMtx[0] = 1.0f;                                   /* band 0 is invariant */
    if (L < 1) return;
    RotRec r; r.M = M; r.prev.assign((size_t)M * M, 0.0f);
    static const int perm[3] = { 1, 2, 0 };             /* band 1 = the rotation matrix in (y, z, x) order */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { r.R1[i][j] = Rxyz[perm[i]][perm[j]]; r.prev[(size_t)i * M + j] = r.R1[i][j]; RotMtx[(size_t)(i + 1) * M + j + 1] = r.R1[i][j]; }
    std::vector<float> cur((size_t)M * M, 0.0f);
    int bandIdx = 4;
    for (int l = 2; l <= L; l++) {
        for (int m = -l; m <= l; m++)
            for (int n = -l; n <= l; n++) {
                const int d = m == 0 ? 1 : 0;
                const int denom = std::abs(n) == l ? (2 * l) * (2 * l - 1) : (l * l - n * n);
                float u = sqrtf((float)(l * l - m * m) / (float)denom);
                float v = sqrtf((float)((1 + d) * (l + std::abs(m) - 1) * (l + std::abs(m))) / (float)denom) * (float)(1 - 2 * d) * 0.5f;
                float w = sqrtf((float)((l - std::abs(m) - 1) * (l - std::abs(m))) / (float)denom) * (float)(1 - d) * (-0.5f);
                if (u != 0) u *= r.P(0, l, m, n);
                if (v != 0) v *= r.V(l, m, n);
                if (w != 0) w *= r.W(l, m, n);
                cur[(size_t)(m + l) * M + (n + l)] = u + v + w;
            }
        for (int i = 0; i < 2 * l + 1; i++) for (int j = 0; j < 2 * l + 1; j++) { RotMtx[(size_t)(bandIdx + i) * M + bandIdx + j] = cur[(size_t)i * M + j]; r.prev[(size_t)i * M + j] = cur[(size_t)i * M + j]; }
        bandIdx += 2 * l + 1;
    }
}

/* yawPitchRoll2Rzyx (saf_utility_geometry.c:213-270) */
void yaw_pitch_roll_to_Rzyx(float yaw, float pitch, float roll, int rollPitchYaw, float R[3][3])
{
    auto Rx = [](float t, float M[3][3]) { const float m[3][3] = { { 1, 0, 0 }, { 0, cosf(t), sinf(t) }, { 0, -sinf(t), cosf(t) } }; memcpy(M, m, sizeof(m)); };
    auto Ry = [](float t, float M[3][3]) { const float m[3][3] = { { cosf(t), 0, -sinf(t) }, { 0, 1, 0 }, { sinf(t), 0, cosf(t) } }; memcpy(M, m, sizeof(m)); };
    auto Rz = [](float t, float M[3][3]) { const float m[3][3] = { { cosf(t), sinf(t), 0 }, { -sinf(t), cosf(t), 0 }, { 0, 0, 1 } }; memcpy(M, m, sizeof(m)); };
    float A[3][3], B[3][3], Cm[3][3], T[3][3];
    if (rollPitchYaw) { Rx(yaw, A); Ry(pitch, B); Rz(roll, Cm); } else { Rz(yaw, A); Ry(pitch, B); Rx(roll, Cm); }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += B[i][k] * A[k][j]; T[i][j] = a; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { float a = 0; for (int k = 0; k < 3; k++) a += Cm[i][k] * T[k][j]; R[i][j] = a; }
}

/* ------------------------------------------------------------------ decoders */
namespace {

struct LsSystem {              /* Y [nSH][N], YW = Y diag(w), Cholesky factor of G = YW Y^T */
    int nSH, N;
    std::vector<double> Y, YW, L, w;
    void build(int order, const float* dirs_deg, int N_, const float* weights)
    {
        nSH = (order + 1) * (order + 1); N = N_;
        std::vector<float> Yf((size_t)nSH * N);
        getRSH(order, const_cast<float*>(dirs_deg), N, Yf.data());
        Y.assign(Yf.begin(), Yf.end());
        w.resize(N);
        for (int i = 0; i < N; i++) w[i] = weights ? (double)weights[i] : 1.0 / (double)N;
        YW.resize(Y.size());
        for (int i = 0; i < nSH; i++) for (int k = 0; k < N; k++) YW[(size_t)i * N + k] = Y[(size_t)i * N + k] * w[k];
        std::vector<double> G((size_t)nSH * nSH);
        for (int i = 0; i < nSH; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k < N; k++) s += YW[(size_t)i * N + k] * Y[(size_t)j * N + k]; G[(size_t)i * nSH + j] = G[(size_t)j * nSH + i] = s; }
        L.assign((size_t)nSH * nSH, 0.0);
        for (int j = 0; j < nSH; j++) {
            double d = G[(size_t)j * nSH + j];
            for (int k = 0; k < j; k++) d -= L[(size_t)j * nSH + k] * L[(size_t)j * nSH + k];
            if (!(d > 0.0)) SAF_FATAL("getBinauralAmbiDecoderMtx: the HRTF grid cannot resolve order %d (Y W Y^T is singular)", order);
            L[(size_t)j * nSH + j] = sqrt(d);
            for (int i = j + 1; i < nSH; i++) { double s = G[(size_t)i * nSH + j]; for (int k = 0; k < j; k++) s -= L[(size_t)i * nSH + k] * L[(size_t)j * nSH + k]; L[(size_t)i * nSH + j] = s / L[(size_t)j * nSH + j]; }
        }
    }
    /* rows of the least-squares decoder for the two-ear HRTF set H [2][N]:  D = (G^-1 YW H^H)^H  -> D[e][i] */
    void decode(const zd* H, zd* D) const
    {
        std::vector<zd> y(nSH);
        for (int e = 0; e < 2; e++) {
            for (int i = 0; i < nSH; i++) { zd s = 0; for (int k = 0; k < N; k++) s += YW[(size_t)i * N + k] * std::conj(H[(size_t)e * N + k]); y[i] = s; }
            for (int i = 0; i < nSH; i++) { zd s = y[i]; for (int k = 0; k < i; k++) s -= L[(size_t)i * nSH + k] * y[k]; y[i] = s / L[(size_t)i * nSH + i]; }
            for (int i = nSH - 1; i >= 0; i--) { zd s = y[i]; for (int k = i + 1; k < nSH; k++) s -= L[(size_t)k * nSH + i] * y[k]; y[i] = s / L[(size_t)i * nSH + i]; }
            for (int i = 0; i < nSH; i++) D[(size_t)e * nSH + i] = std::conj(y[i]);
        }
    }
    void apply(const zd* D, zd* Ha) const      /* Ha [2][N] = D Y */
    {
        for (int e = 0; e < 2; e++) for (int k = 0; k < N; k++) { zd s = 0; for (int i = 0; i < nSH; i++) s += D[(size_t)e * nSH + i] * Y[(size_t)i * N + k]; Ha[(size_t)e * N + k] = s; }
    }
    void diffuse_cov(const zd* H, zd C[2][2]) const   /* H diag(w) H^H */
    {
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { zd s = 0; for (int k = 0; k < N; k++) s += H[(size_t)a * N + k] * w[k] * std::conj(H[(size_t)b * N + k]); C[a][b] = s; }
    }
};

int cutoff_band(const float* freqVector, int nBands)      /* band nearest to 1.5 kHz (saf_hoa_internal.c:461-468) */
{
    float minVal = 2.23e10f; int bc = 0;
    for (int b = 0; b < nBands; b++) if (minVal > fabsf(freqVector[b] - 1.5e3f)) { minVal = fabsf(freqVector[b] - 1.5e3f); bc = b; }
    return bc;
}

/* spatial re-sampling decoder (saf_hoa_internal.c:332-430) */
void decoder_spr(const zf* hrtfs, const float* dirs_deg, int N, int nBands, int order, const float* weights, zf* dec)
{
    const int nSH = (order + 1) * (order + 1);
    int Nh_max = (int)(sqrtf((float)N) - 1.0f); if (Nh_max > 20) Nh_max = 20;
    std::vector<float> rad((size_t)N * 2);
    for (int i = 0; i < N; i++) { rad[i * 2] = dirs_deg[i * 2] * (SAF_PI / 180.0f); rad[i * 2 + 1] = SAF_PI / 2.0f - dirs_deg[i * 2 + 1] * (SAF_PI / 180.0f); }
    /* highest order whose weighted SH Gram matrix is conditioned below 100 (checkCondNumberSHTReal, saf_sh.c:884-960) */
    const int nSHmax = (Nh_max + 1) * (Nh_max + 1);
    std::vector<float> YN((size_t)nSHmax * N);
    getSHreal(Nh_max, rad.data(), N, YN.data());
    int Nh = 0;
    for (int n = 0; n <= Nh_max; n++) {
        const int ns = (n + 1) * (n + 1);
        std::vector<float> YY((size_t)ns * ns);
        for (int i = 0; i < ns; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k < N; k++) s += (double)YN[(size_t)i * N + k] * (weights ? weights[k] : 1.0f) * YN[(size_t)j * N + k]; YY[(size_t)i * ns + j] = YY[(size_t)j * ns + i] = (float)s; }
        std::vector<double> U, S, V;
        thin_svd(YY.data(), ns, ns, U, S, V);
        double mx = S[0], mn = S[0];
        for (double v : S) { mx = std::max(mx, v); mn = std::min(mn, v); }
        if ((float)mx / ((float)mn + 2.23e-7f) < 100.0f) Nh = n;
    }
    if (Nh < order) SAF_FATAL("getBinauralAmbiDecoderMtx (SPR): input order %d exceeds the modal order %d of the HRTF grid", order, Nh);
    const int nSHh = (Nh + 1) * (Nh + 1);
    std::vector<float> Ynh((size_t)nSHh * N);
    getRSH(Nh, const_cast<float*>(dirs_deg), N, Ynh.data());
    char name[64]; snprintf(name, sizeof(name), "Tdesign_degree_%d_dirs_deg", 2 * order);
    int K = 0, d1 = 0;
    const float* td = table(name, &K, &d1);
    if (!td) SAF_FATAL("table %s missing", name);
    std::vector<float> Ytd((size_t)nSHh * K);
    getRSH(Nh, const_cast<float*>(td), K, Ytd.data());
    std::vector<double> WYY((size_t)N * K);
    for (int i = 0; i < N; i++) for (int j = 0; j < K; j++) { double s = 0; for (int k = 0; k < nSHh; k++) s += (double)Ynh[(size_t)k * N + i] * Ytd[(size_t)k * K + j]; WYY[(size_t)i * K + j] = s * (weights ? (double)weights[i] / (4.0 * SAF_PId) : 1.0 / (double)N); }
    std::vector<zd> Htd((size_t)2 * K);
    for (int band = 0; band < nBands; band++) {
        const zf* H = hrtfs + (size_t)band * 2 * N;
        for (int e = 0; e < 2; e++) for (int j = 0; j < K; j++) { zd s = 0; for (int k = 0; k < N; k++) s += zd(H[(size_t)e * N + k]) * WYY[(size_t)k * K + j]; Htd[(size_t)e * K + j] = s; }
        for (int e = 0; e < 2; e++) for (int i = 0; i < nSH; i++) {
            zd s = 0; for (int j = 0; j < K; j++) s += (double)Ytd[(size_t)i * K + j] * std::conj(Htd[(size_t)e * K + j]);
            dec[(size_t)band * 2 * nSH + (size_t)e * nSH + i] = zf(std::conj(s) / (double)K);
        }
    }
}

}  // namespace
}  // namespace saf

using namespace saf;

extern "C" {

void getSHrotMtxReal(float Rxyz[3][3], float* RotMtx, int L) { sh_rot_matrix_real(Rxyz, RotMtx, L); }       /* saf_sh.c:479 */
void yawPitchRoll2Rzyx(float yaw, float pitch, float roll, int rollPitchYawFLAG, float R[3][3]) { yaw_pitch_roll_to_Rzyx(yaw, pitch, roll, rollPitchYawFLAG, R); }

/* saf_hoa.c:502-603 */
void applyDiffCovMatching(float_complex* hrtfs_, float* hrtf_dirs_deg, int N_dirs, int N_bands, int order, float* weights, float_complex* decMtx_)
{
    const zf* hrtfs = reinterpret_cast<const zf*>(hrtfs_); zf* dec = reinterpret_cast<zf*>(decMtx_);
    LsSystem sys; sys.build(order, hrtf_dirs_deg, N_dirs, weights);
    const int nSH = sys.nSH, N = N_dirs;
    std::vector<zd> H((size_t)2 * N), Ha((size_t)2 * N), D((size_t)2 * nSH), Dn((size_t)2 * nSH);
    for (int band = 0; band < N_bands - 1 /* Nyquist skipped */; band++) {
        for (int i = 0; i < 2 * N; i++) H[i] = zd(hrtfs[(size_t)band * 2 * N + i]);
        for (int i = 0; i < 2 * nSH; i++) D[i] = zd(dec[(size_t)band * 2 * nSH + i]);
        zd Cr[2][2], Ca[2][2];
        sys.diffuse_cov(H.data(), Cr);
        sys.apply(D.data(), Ha.data());
        sys.diffuse_cov(Ha.data(), Ca);
        /* upper Cholesky factors X^H X = C (diagonals forced real, :556-573) */
        zd X[2][2] = { { 0, 0 }, { 0, 0 } }, Xa[2][2] = { { 0, 0 }, { 0, 0 } };
        X[0][0] = sqrt(Cr[0][0].real()); X[0][1] = Cr[0][1] / X[0][0]; X[1][1] = sqrt(Cr[1][1].real() - std::norm(X[0][1]));
        Xa[0][0] = sqrt(Ca[0][0].real()); Xa[0][1] = Ca[0][1] / Xa[0][0]; Xa[1][1] = sqrt(Ca[1][1].real() - std::norm(Xa[0][1]));
        /* A = Xa^H X = U S V^H (:576-580); the matching matrix needs V U^H = (A^H A)^(-1/2) A^H, the conjugate-transposed unitary
         * polar factor of A — unique, unlike U and V themselves */
        zd A[2][2], P[2][2];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { A[i][j] = 0; for (int k = 0; k < 2; k++) A[i][j] += std::conj(Xa[k][i]) * X[k][j]; }
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { P[i][j] = 0; for (int k = 0; k < 2; k++) P[i][j] += std::conj(A[k][i]) * A[k][j]; }
        const double a = P[0][0].real(), d = P[1][1].real(); const zd b = P[0][1];
        const double sdet = sqrt(a * d - std::norm(b)), t = sqrt(a + d + 2.0 * sdet);      /* sqrt(P) = (P + sqrt(det P) I) / t */
        const zd S[2][2] = { { (a + sdet) / t, b / t }, { std::conj(b) / t, (d + sdet) / t } };
        const zd dS = S[0][0] * S[1][1] - S[0][1] * S[1][0];
        const zd Si[2][2] = { { S[1][1] / dS, -S[0][1] / dS }, { -S[1][0] / dS, S[0][0] / dS } };
        zd VU[2][2], VUX[2][2], M[2][2];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { VU[i][j] = 0; for (int k = 0; k < 2; k++) VU[i][j] += Si[i][k] * std::conj(A[j][k]); }
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { VUX[i][j] = 0; for (int k = 0; k < 2; k++) VUX[i][j] += VU[i][k] * X[k][j]; }
        for (int j = 0; j < 2; j++) { M[1][j] = VUX[1][j] / Xa[1][1]; M[0][j] = (VUX[0][j] - Xa[0][1] * M[1][j]) / Xa[0][0]; }      /* Xa M = VUX */
        for (int e = 0; e < 2; e++) for (int i = 0; i < nSH; i++) { zd s = 0; for (int k = 0; k < 2; k++) s += std::conj(M[k][e]) * D[(size_t)k * nSH + i]; Dn[(size_t)e * nSH + i] = s; }
        for (int i = 0; i < 2 * nSH; i++) dec[(size_t)band * 2 * nSH + i] = zf(Dn[i]);
    }
}

/* saf_hoa.c:394-450 */
void getBinauralAmbiDecoderMtx(float_complex* hrtfs_, float* hrtf_dirs_deg, int N_dirs, int N_bands, BINAURAL_AMBI_DECODER_METHODS method, int order,
                               float* freqVector, float* itd_s, float* weights, int enableDiffCovMatching, int enableMaxReWeighting, float_complex* decMtx_)
{
    const zf* hrtfs = reinterpret_cast<const zf*>(hrtfs_); zf* dec = reinterpret_cast<zf*>(decMtx_);
    const int nSH = (order + 1) * (order + 1), N = N_dirs;
    (void)itd_s;      /* the time-alignment phase term of the reference multiplies the ITD by zero (saf_hoa_internal.c:495-498) */
    if (method == BINAURAL_DECODER_SPR) decoder_spr(hrtfs, hrtf_dirs_deg, N, N_bands, order, weights, dec);
    else {
        LsSystem sys; sys.build(order, hrtf_dirs_deg, N, weights);
        if ((method == BINAURAL_DECODER_TA || method == BINAURAL_DECODER_MAGLS) && !freqVector) SAF_FATAL("getBinauralAmbiDecoderMtx: the TA and MagLS decoders need freqVector");
        const int bc = (method == BINAURAL_DECODER_TA || method == BINAURAL_DECODER_MAGLS) ? cutoff_band(freqVector, N_bands) : 0;
        std::vector<zd> H((size_t)2 * N), Hm((size_t)2 * N), D((size_t)2 * nSH), Dprev((size_t)2 * nSH);
        for (int band = 0; band < N_bands; band++) {
            for (int i = 0; i < 2 * N; i++) H[i] = zd(hrtfs[(size_t)band * 2 * N + i]);
            double Gh = 1.0;
            if (method == BINAURAL_DECODER_TA && band >= bc) {
                for (int i = 0; i < 2 * N; i++) Hm[i] = zd(hrtfs[(size_t)bc * 2 * N + i]);
                sys.decode(Hm.data(), D.data());
            } else if (method == BINAURAL_DECODER_MAGLS && band > bc) {
                /* magnitudes of this band, phases of what the previous band's decoder renders (saf_hoa_internal.c:596-610) */
                for (int i = 0; i < 2 * nSH; i++) Dprev[i] = zd(dec[(size_t)(band - 1) * 2 * nSH + i]);
                sys.apply(Dprev.data(), Hm.data());
                for (int i = 0; i < 2 * N; i++) { const float ph = atan2f((float)Hm[i].imag(), (float)Hm[i].real()); Hm[i] = (double)std::abs(zf(H[i])) * zd(std::exp(zf(0.0f, ph))); }
                sys.decode(Hm.data(), D.data());
            } else {
                sys.decode(H.data(), D.data());
                if (method == BINAURAL_DECODER_LSDIFFEQ) {
                    sys.apply(D.data(), Hm.data());
                    zd Cr[2][2], Cl[2][2];
                    sys.diffuse_cov(H.data(), Cr); sys.diffuse_cov(Hm.data(), Cl);
                    Gh = (sqrtf((float)Cr[0][0].real() / ((float)Cl[0][0].real() + 2.23e-7f)) + sqrtf((float)Cr[1][1].real() / ((float)Cl[1][1].real() + 2.23e-7f))) / 2.0f;
                }
            }
            for (int i = 0; i < 2 * nSH; i++) dec[(size_t)band * 2 * nSH + i] = zf(D[i] * Gh);
        }
    }
    if (enableMaxReWeighting) {
        std::vector<float> a; maxre_weights(order, a);
        for (int band = 0; band < N_bands; band++) for (int e = 0; e < 2; e++) for (int i = 0; i < nSH; i++) dec[(size_t)band * 2 * nSH + (size_t)e * nSH + i] *= a[i];
    }
    if (enableDiffCovMatching) applyDiffCovMatching(hrtfs_, hrtf_dirs_deg, N_dirs, N_bands, order, weights, decMtx_);
}

}

/* ------------------------------------------------------------------ truncation EQ (saf_hoa.c:269-324) */
namespace saf {
namespace {

/* Spherical Bessel functions j_n, y_n and derivatives for n = 0..N by Zhang & Jin's scheme ("Computation of Special
 * Functions", routines SPHJ / SPHY / MSTA1 / MSTA2 — what saf_utility_bessel.c:40-353 implements): j_n by downward
 * recurrence from a starting order chosen with the envelope estimate, normalised against j_0 or j_1; y_n by upward
 * recurrence until it overflows.  `nm` returns the highest order actually computed. */
double envelope(int n, double x) { return 0.5 * log(6.28 * n) - n * log(1.36 * x / n); }
int secant_order(double a0, int n0, double obj)
{
    double f0 = envelope(n0, a0) - obj;
    int n1 = n0 + 5;
    double f1 = envelope(n1, a0) - obj;
    int nn = 0;
    for (int it = 0; it < 20; it++) {
        nn = n1 - (int)((double)(n1 - n0) / (1.0 - f0 / f1));
        const double f = envelope(nn, a0) - obj;
        if (std::abs(nn - n1) < 1) break;
        n0 = n1; f0 = f1; n1 = nn; f1 = f;
    }
    return nn;
}
int start_order_magnitude(double x, int mp) { const double a0 = fabs(x); return secant_order(a0, (int)(floor(1.1 * a0) + 1.0), (double)mp); }           /* MSTA1 */
int start_order_digits(double x, int n, int mp)                                                                                                     /* MSTA2 */
{
    const double a0 = fabs(x), hmp = 0.5 * mp, ejn = envelope(n, a0);
    if (ejn <= hmp) return secant_order(a0, (int)floor(1.1 * a0), (double)mp) + 10;
    return secant_order(a0, n, hmp + ejn) + 10;
}
void sph_bessel_j(int N, double x, int* nm, double* sj, double* dj)
{
    *nm = N;
    if (fabs(x) < 1e-80) { for (int k = 0; k <= N; k++) sj[k] = dj[k] = 0.0; sj[0] = 1.0; if (N > 0) dj[1] = 0.333333333333333; return; }
    sj[0] = sin(x) / x;
    if (N >= 1) sj[1] = (sj[0] - cos(x)) / x;
    if (N >= 2) {
        const double sa = sj[0], sb = sj[1];
        int m = start_order_magnitude(x, 200);
        if (m < N) *nm = m; else m = start_order_digits(x, N, 15);
        for (int i = 0; m < 0; i++) { m = start_order_digits(x, N, 14 - i); if (i == 13) m = m < 0 ? 0 : m; }     /* fewer digits rather than NaNs */
        double f0 = 0.0, f1 = 1.0 - 100, f = 1.0;           /* any non-zero seed: the result is rescaled below */
        for (int k = m; k >= 0; k--) { f = (2.0 * k + 3.0) * f1 / x - f0; if (k <= *nm) sj[k] = f; f0 = f1; f1 = f; }
        const double cs = fabs(sa) > fabs(sb) ? sa / f : sb / f0;
        for (int k = 0; k <= *nm; k++) sj[k] *= cs;
    }
    dj[0] = (cos(x) - sin(x) / x) / x;
    for (int k = 1; k <= *nm; k++) dj[k] = sj[k - 1] - (k + 1.0) * sj[k] / x;
}
void sph_bessel_y(int N, double x, int* nm, double* sy, double* dy)
{
    *nm = N;
    if (x < 1e-20) { for (int k = 0; k <= N; k++) { sy[k] = -1.0e+300; dy[k] = 1e+300; } return; }
    sy[0] = -cos(x) / x;
    if (N >= 1) sy[1] = (sy[0] - sin(x)) / x;
    double f0 = sy[0], f1 = N >= 1 ? sy[1] : 0.0;
    int k = 2;
    for (; k <= N; k++) { const double f = (2.0 * k - 1.0) * f1 / x - f0; sy[k] = f; if (fabs(f) >= 1e+300) break; f0 = f1; f1 = f; }
    *nm = k - 1;
    dy[0] = (sin(x) + cos(x) / x) / x;
    for (int q = 1; q <= *nm; q++) dy[q] = sy[q - 1] - (q + 1.0) * sy[q] / x;
}

/* |b_n|^2 of the rigid-sphere modal coefficients 4 pi i^n (j_n - j_n'/h_n2' h_n2) (sphModalCoeffs, saf_sh.c:2018-2048), orders
 * 0..N for every kr; orders above the highest one computable for ALL kr stay zero, as in the reference */
void rigid_modal_power(int N, const double* kr, int nBands, std::vector<double>& b2)
{
    b2.assign((size_t)nBands * (N + 1), 0.0);
    std::vector<zd> hn((size_t)nBands * (N + 1), zd(0, 0)), dhn((size_t)nBands * (N + 1), zd(0, 0));
    std::vector<double> jn((size_t)nBands * (N + 1), 0.0), djn((size_t)nBands * (N + 1), 0.0);
    std::vector<double> tj(N + 1), tdj(N + 1), ty(N + 1), tdy(N + 1);
    int maxJ = 1000000000, maxH = 1000000000;
    for (int i = 0; i < nBands; i++) {
        if (kr[i] <= 1e-15) {
            /* (the reference writes these defaults into row 0 whatever i is, saf_utility_bessel.c:679-688, 1146-1152; the
             * only kr = 0 of the afSTFT centre frequencies IS band 0) */
            for (int n = 0; n <= N; n++) { jn[n] = 0.0; djn[n] = 0.0; hn[n] = dhn[n] = zd(0, 0); }
            jn[0] = 1.0; if (N > 0) djn[1] = 1.0 / 3.0; hn[0] = zd(1.0, 0.0);
            continue;
        }
        int n1, n2;
        sph_bessel_j(N, kr[i], &n1, tj.data(), tdj.data());
        maxJ = std::min(maxJ, n1);
        for (int n = 0; n <= n1; n++) { jn[(size_t)i * (N + 1) + n] = tj[n]; djn[(size_t)i * (N + 1) + n] = tdj[n]; }
        sph_bessel_y(N, kr[i], &n2, ty.data(), tdy.data());
        maxH = std::min(maxH, std::min(n1, n2));
        for (int n = 0; n <= std::min(n1, n2); n++) { hn[(size_t)i * (N + 1) + n] = zd(tj[n], -ty[n]); dhn[(size_t)i * (N + 1) + n] = zd(tdj[n], -tdy[n]); }
    }
    const int maxN = std::min(std::min(maxJ, maxH), N);
    for (int i = 0; i < nBands; i++)
        for (int n = 0; n <= maxN; n++) {
            zd b;
            if (n == 0 && kr[i] <= 1e-20) b = zd(4.0 * SAF_PId, 0.0);
            else if (kr[i] <= 1e-20) b = zd(0.0, 0.0);
            else b = 4.0 * SAF_PId * (zd(jn[(size_t)i * (N + 1) + n], 0.0) - zd(djn[(size_t)i * (N + 1) + n], 0.0) / dhn[(size_t)i * (N + 1) + n] * hn[(size_t)i * (N + 1) + n]);      /* |i^n| = 1 */
            b2[(size_t)i * (N + 1) + n] = std::norm(b);
        }
}

}  // namespace
}  // namespace saf

extern "C" {

/* saf_sh.c:751-776 */
void beamWeightsMaxEV(int N, float* b_n)
{
    float norm = 0.0f;
    const double x = cos(2.4068f / ((double)N + 1.51));
    double pm1 = 1.0, p = x;
    for (int n = 0; n <= N; n++) {
        double pn;                                  /* Legendre polynomial P_n(x) */
        if (n == 0) pn = 1.0; else if (n == 1) pn = x; else { pn = ((2.0 * n - 1.0) * x * p - (n - 1.0) * pm1) / (double)n; pm1 = p; p = pn; }
        b_n[n] = sqrtf((2.0f * (float)n + 1.0f) / (4.0f * SAF_PI)) * (float)pn;
        norm += sqrtf((2.0f * (float)n + 1.0f) / (4.0f * SAF_PI)) * b_n[n];
    }
    for (int n = 0; n <= N; n++) b_n[n] /= norm;
}

/* saf_hoa.c:269-324 */
void truncationEQ(float* w_n, int order_truncated, int order_target, double* kr, int nBands, float softThreshold, float* gain)
{
    std::vector<double> bt, bq;
    saf::rigid_modal_power(order_target, kr, nBands, bt);
    saf::rigid_modal_power(order_truncated, kr, nBands, bq);
    const float clip = powf(10.0f, softThreshold / 20.0f);
    for (int b = 0; b < nBands; b++) {
        double pt = 0.0, pq = 0.0;
        for (int n = 0; n <= order_target; n++) pt += (2.0 * n + 1.0) * bt[(size_t)b * (order_target + 1) + n];
        for (int n = 0; n <= order_truncated; n++) pq += w_n[n] * (2.0 * n + 1.0) * bq[(size_t)b * (order_truncated + 1) + n];
        pt = 1.0 / (4.0 * SAF_PI) * sqrt(pt); pq = 1.0 / (4.0 * SAF_PI) * sqrt(pq);
        float g = (float)(pt / (pq + 2.23e-13));
        g /= clip;                                  /* soft clip at the threshold */
        if (g > 1.0f) g = 1.0f + tanhf(g - 1.0f);
        gain[b] = g * clip;
    }
}


/* getBinauralAmbiDecoderFilters (saf_hoa.h:452-500 / saf_hoa.c:452-500): the decoder per uniformly spaced bin
 * (getUniformFreqVector, saf_utility_fft.c:145-155), then one inverse real FFT per (ear, SH channel).
 * hrtfs: (fftSize/2+1) x 2 x N_dirs; decFilters: 2 x nSH x fftSize. */
void getBinauralAmbiDecoderFilters(float_complex* hrtfs, float* hrtf_dirs_deg, int N_dirs, int fftSize, float fs, BINAURAL_AMBI_DECODER_METHODS method, int order,
                                   float* itd_s, float* weights, int enableDiffCovMatching, int enableMaxReWeighting, float* decFilters)
{
    const int nBins = fftSize / 2 + 1, nSH = (order + 1) * (order + 1);
    std::vector<float> freq(nBins);
    for (int k = 0; k < nBins; k++) freq[k] = (float)k * fs / (float)fftSize;
    std::vector<float_complex> dec((size_t)nBins * 2 * nSH), bins(nBins);
    getBinauralAmbiDecoderMtx(hrtfs, hrtf_dirs_deg, N_dirs, nBins, method, order, freq.data(), itd_s, weights, enableDiffCovMatching, enableMaxReWeighting, dec.data());
    void* hFFT = nullptr;
    saf_rfft_create(&hFFT, fftSize);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < nSH; j++) {
            for (int k = 0; k < nBins; k++) bins[k] = dec[(size_t)k * 2 * nSH + i * nSH + j];
            saf_rfft_backward(hFFT, bins.data(), decFilters + ((size_t)i * nSH + j) * fftSize);
        }
    saf_rfft_destroy(&hFFT);
}

}
