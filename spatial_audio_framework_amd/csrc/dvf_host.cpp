/*
 * dvf_host.cpp — distance variation function (DVF) near-field filters on the host: what binauraliser_nf evaluates per moved
 * source and ear before the band MAC (framework/modules/saf_utilities/saf_utility_dvf.h:62-153, saf_utility_dvf.c).
 *
 * Model: S. Spagnol, E. Tavazzi, F. Avanzini, "Distance rendering and perception of nearby virtual sound sources with a
 * near-field filter model", Applied Acoustics 115 (2017): the near-field correction of a rigid-sphere head as a first-order
 * high shelf whose DC gain, high-frequency gain and cut-off are rational functions of the normalised distance rho, fitted
 * at 19 lateral angles (0..180 degrees in 10-degree steps) and interpolated linearly in between.
 *
 * The per-band response of the shelf (evalIIRTransferFunctionf, saf_utility_filters.c:609-671) is evaluated on the GPU
 * for the operator (binaural_kernels.hip, dvf_scale_kernel); the host form here serves callers of the utility itself.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "dvf_host.h"
#include <cmath>

namespace saf {

/* fitted rational functions per table angle (the numbers of saf_utility_dvf.c:34-46):
 *   g0(rho)   = (n[0] rho + n[1]) / (rho^2 + n[2] rho + n[3])            [dB]
 *   gInf(rho) = same form                                                [dB]
 *   fc(rho)   = (n[0] rho^2 + n[1] rho + n[2]) / (rho^2 + n[3] rho + n[4])   [normalised; x c / (2 pi a_head) -> Hz] */
struct DvfFit { double g0[4], gInf[4], fc[5]; };
static const DvfFit kDvfFit[19] = {
    { { 12.97, -9.69, -1.14, 0.219 }, { -4.39, 2.123, -0.55, -0.06 }, { 0.457, -0.67, 0.174, -1.75, 0.699 } },
    { { 13.19, 234.2, 18.48, -8.5 }, { -4.31, -2.78, 0.59, -0.17 }, { 0.455, 0.142, -0.11, -0.01, -0.35 } },
    { { 12.13, -11.2, -1.25, 0.346 }, { -4.18, 4.224, -1.01, -0.02 }, { -0.87, 3404., -1699., 7354., -5350. } },
    { { 11.19, -9.03, -1.02, 0.336 }, { -4.01, 3.039, -0.56, -0.32 }, { 0.465, -0.91, 0.437, -2.18, 1.188 } },
    { { 9.91, -7.87, -0.83, 0.379 }, { -3.87, -0.57, 0.665, -1.13 }, { 0.494, -0.67, 0.658, -1.2, 0.256 } },
    { { 8.328, -7.42, -0.67, 0.421 }, { -4.1, -34.7, 11.39, -8.3 }, { 0.549, -1.21, 2.02, -1.59, 0.816 } },
    { { 6.493, -7.31, -0.5, 0.423 }, { -3.87, 3.271, -1.57, 0.637 }, { 0.663, -1.76, 6.815, -1.23, 1.166 } },
    { { 4.455, -7.28, -0.32, 0.382 }, { -5.02, 0.023, -0.87, 0.325 }, { 0.691, 4.655, 0.614, -0.89, 0.76 } },
    { { 2.274, -7.29, -0.11, 0.314 }, { -6.72, -8.96, 0.37, -0.08 }, { 3.507, 55.09, 589.3, 29.23, 59.51 } },
    { { 0.018, -7.48, -0.13, 0.24 }, { -8.69, -58.4, 5.446, -1.19 }, { -27.4, 10336., 16818., 1945., 1707. } },
    { { -2.24, -8.04, 0.395, 0.177 }, { -11.2, 11.47, -1.13, 0.103 }, { 6.371, 1.735, -9.39, -0.06, -1.12 } },
    { { -4.43, -9.23, 0.699, 0.132 }, { -12.1, 8.716, -0.63, -0.12 }, { 7.032, 40.88, -44.1, 5.635, -6.18 } },
    { { -6.49, -11.6, 1.084, 0.113 }, { -11.1, 21.8, -2.01, 0.098 }, { 7.092, 23.86, -23.6, 3.308, -3.39 } },
    { { -8.34, -17.4, 1.757, 0.142 }, { -11.1, 1.91, 0.15, -0.4 }, { 7.463, 102.8, -92.3, 13.88, -12.7 } },
    { { -9.93, -48.4, 4.764, 0.462 }, { -9.72, -0.04, 0.243, -0.41 }, { 7.453, -6.14, -1.81, -0.88, -0.19 } },
    { { -11.3, 9.149, -0.64, -0.14 }, { -8.42, -0.66, 0.147, -0.34 }, { 8.101, -18.1, 10.54, -2.23, 1.295 } },
    { { -12.2, 1.905, 0.109, -0.08 }, { -7.44, 0.395, -0.18, -0.18 }, { 8.702, -9.05, 0.532, -0.96, -0.02 } },
    { { -12.8, -0.75, 0.386, -0.06 }, { -6.78, 2.662, -0.67, 0.05 }, { 8.925, -9.03, 0.285, -0.9, -0.08 } },
    { { -13.0, -1.32, 0.45, -0.05 }, { -6.58, 3.387, -0.84, 0.131 }, { 9.317, -6.89, -2.08, -0.57, -0.4 } },
};
static const int kDvfAngles = 19;
static const float kHeadRadius = 0.09096f;                                  /* a_head, saf_utility_dvf.c:48-49 */
static const float kShelfWarp = SAF_PI * (0.0875f / kHeadRadius);           /* pi a_0 / a_head */
static const float kFcScale = 343.0f / (2.0f * SAF_PI * kHeadRadius);       /* c / (2 pi a_head) */

void dvf_shelf_params(int idx, float rhoIn, float* g0, float* gInf, float* fc)
{
    const DvfFit& f = kDvfFit[idx];
    const double r = (double)rhoIn, r2 = r * r;
    *g0 = (float)((f.g0[0] * r + f.g0[1]) / (r2 + f.g0[2] * r + f.g0[3]));
    *gInf = (float)((f.gInf[0] * r + f.gInf[1]) / (r2 + f.gInf[2] * r + f.gInf[3]));
    const float fn = (float)((f.fc[0] * r2 + f.fc[1] * r + f.fc[2]) / (r2 + f.fc[3] * r + f.fc[4]));
    *fc = fn * kFcScale;
}

void dvf_interp_params(float theta, float rho, float* g0, float* gInf, float* fc)
{
    theta = fminf(fmaxf(theta, 0.f), 180.f);
    rho = fmaxf(rho, 1.0f);
    const float pos = theta / 10.f;
    int lo = (int)pos, hi = lo + 1;
    if (hi >= kDvfAngles) { hi = kDvfAngles - 1; lo = hi - 1; }
    float l[3], h[3];
    dvf_shelf_params(lo, rho, &l[0], &l[1], &l[2]);
    dvf_shelf_params(hi, rho, &h[0], &h[1], &h[2]);
    const float frac = pos - lo;
    *g0 = l[0] + (h[0] - l[0]) * frac;
    *gInf = l[1] + (h[1] - l[1]) * frac;
    *fc = l[2] + (h[2] - l[2]) * frac;
}

void dvf_shelf_coeffs(float g0_dB, float gInf_dB, float fc, float fs, float* b0, float* b1, float* a1)
{
    const float V0 = powf(10.f, gInf_dB / 20.f);
    const float G0 = powf(10.f, g0_dB / 20.f);
    const float t = tanf((kShelfWarp / fs) * fc);
    const float V0t = V0 * t;
    const float ac = (V0t - 1.f) / (V0t + 1.f);
    const float V = (V0 - 1.f) * 0.5f;
    const float Vac = V * ac;
    *b0 = G0 * (V - Vac + 1.f);
    *b1 = G0 * (Vac - V + ac);
    *a1 = ac;
}

void dvf_coeffs(float alpha, float rho, float fs, float* b, float* a)
{
    float g0, gInf, fc;
    dvf_interp_params(alpha, rho, &g0, &gInf, &fc);
    dvf_shelf_coeffs(g0, gInf, fc, fs, &b[0], &b[1], &a[1]);
}

void dvf_lateral_angles(float azimuth, float elevation, float* alphaLR, float* betaLR)
{
    const float az = azimuth * SAF_PI / 180.0f, el = elevation * SAF_PI / 180.0f;
    const float sAz = sinf(az), sEl = sinf(el), cAz = cosf(az), cEl = cosf(el);
    float alpha = SAF_PI / 2.f - acosf(sAz * cEl);
    float beta = asinf(sEl / sqrtf(powf(sEl, 2.f) + (powf(cAz, 2.f) * powf(cEl, 2.f))));
    if (beta > SAF_PI / 2.f) { alpha = SAF_PI - alpha; beta = SAF_PI - beta; }
    alpha = fabsf(SAF_PI / 2.f - alpha);
    if (alpha > SAF_PI) alpha = 2 * SAF_PI - alpha;
    const float aDeg = alpha * 180.0f / SAF_PI;
    alphaLR[0] = aDeg; alphaLR[1] = 180.f - aDeg;
    if (betaLR) { const float bDeg = beta * 180.0f / SAF_PI; betaLR[0] = bDeg; betaLR[1] = 180.f - bDeg; }
}

void iir_response_f(const float* bc, const float* ac, int nCoeffs, const float* freqs, int nFreqs, float fs, int mag2dB, float* magnitude, float* phase_rad)
{
    const float wScale = -2.0 * SAF_PI / fs;        /* z^-n = e^{-j n w} */
    for (int k = 0; k < nFreqs; k++) {
        const float w = freqs[k] * wScale;
        float nr = bc[0], ni = 0.0f, dr = ac[0], di = 0.0f;
        for (int n = 1; n < nCoeffs; n++) {
            const float x = n * w, cx = cosf(x), sx = sinf(x);
            nr += bc[n] * cx; ni += bc[n] * sx; dr += ac[n] * cx; di += ac[n] * sx;
        }
        const double inv = 1.0 / (powf(dr, 2.f) + powf(di, 2.f) + 2.23e-7f);
        if (magnitude) {
            magnitude[k] = (float)sqrt((powf(nr, 2.0f) + powf(ni, 2.0f)) * inv);
            if (mag2dB) magnitude[k] = 20.0f * log10f(magnitude[k]);
        }
        if (phase_rad) {
            const float hr = (nr * dr + ni * di) * inv, hi = (ni * dr - nr * di) * inv;
            phase_rad[k] = (float)atan2(hi, hr);
        }
    }
}

}  // namespace saf

extern "C" {
void calcDVFShelfParams(int i, float rho, float* g0, float* gInf, float* fc) { saf::dvf_shelf_params(i, rho, g0, gInf, fc); }
void interpDVFShelfParams(float theta, float rho, float* iG0, float* iGInf, float* iFc) { saf::dvf_interp_params(theta, rho, iG0, iGInf, iFc); }
void dvfShelfCoeffs(float g0, float gInf, float fc, float fs, float* b0, float* b1, float* a1) { saf::dvf_shelf_coeffs(g0, gInf, fc, fs, b0, b1, a1); }
void calcDVFCoeffs(float alpha, float rho, float fs, float* b, float* a) { saf::dvf_coeffs(alpha, rho, fs, b, a); }
void doaToIpsiInteraural(float azimuth, float elevation, float* alphaLR, float* betaLR) { saf::dvf_lateral_angles(azimuth, elevation, alphaLR, betaLR); }
void evalIIRTransferFunctionf(float* b_coeff, float* a_coeff, int nCoeffs, float* freqs, int nFreqs, float fs, int mag2dB, float* magnitude, float* phase_rad)
{
    saf::iir_response_f(b_coeff, a_coeff, nCoeffs, freqs, nFreqs, fs, mag2dB, magnitude, phase_rad);
}
}
