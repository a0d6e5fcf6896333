/*
 * orc_dvf.c — TEST INFRASTRUCTURE ONLY (CPU restatement; never linked into or called by the product library).
 *
 * Distance variation function (DVF) near-field filters: framework/modules/saf_utilities/saf_utility_dvf.c
 * (rational fits of Spagnol, Tavazzi, Avanzini, "Distance rendering and perception of nearby virtual sound sources
 * with a near-field filter model", Applied Acoustics 115, 2017) and the first-order response evaluation
 * evalIIRTransferFunctionf (saf_utility_filters.c:609-671).
 *
 * Pinned by the reference's own known-answer tests test__dvf_calcDVFShelfParams / test__dvf_interpDVFShelfParams /
 * test__dvf_dvfShelfCoeffs and the 12 DVF cases of test__evalIIRTransferFunction (test__utilities_module.c:1114-1190,
 * 1304-1440), restated in tests/test_oracle_cpu.py.
 */
#include "saf_oracle.h"
#include <math.h>
#define ORC_PI 3.14159265358979323846264338327950288f   /* SAF_PI (saf_utilities.h:70) */

/* one row per 10-degree step of the lateral angle: g0 fit (p1 p2 q1 q2), gInf fit (p1 p2 q1 q2), fc fit (p1 p2 p3 q1 q2)
 * — the numbers of saf_utility_dvf.c:34-46, transposed */
static const double dvf_fit[19][13] = {
    { 12.97, -9.69, -1.14, 0.219, -4.39, 2.123, -0.55, -0.06, 0.457, -0.67, 0.174, -1.75, 0.699 },
    { 13.19, 234.2, 18.48, -8.5, -4.31, -2.78, 0.59, -0.17, 0.455, 0.142, -0.11, -0.01, -0.35 },
    { 12.13, -11.2, -1.25, 0.346, -4.18, 4.224, -1.01, -0.02, -0.87, 3404., -1699., 7354., -5350. },
    { 11.19, -9.03, -1.02, 0.336, -4.01, 3.039, -0.56, -0.32, 0.465, -0.91, 0.437, -2.18, 1.188 },
    { 9.91, -7.87, -0.83, 0.379, -3.87, -0.57, 0.665, -1.13, 0.494, -0.67, 0.658, -1.2, 0.256 },
    { 8.328, -7.42, -0.67, 0.421, -4.1, -34.7, 11.39, -8.3, 0.549, -1.21, 2.02, -1.59, 0.816 },
    { 6.493, -7.31, -0.5, 0.423, -3.87, 3.271, -1.57, 0.637, 0.663, -1.76, 6.815, -1.23, 1.166 },
    { 4.455, -7.28, -0.32, 0.382, -5.02, 0.023, -0.87, 0.325, 0.691, 4.655, 0.614, -0.89, 0.76 },
    { 2.274, -7.29, -0.11, 0.314, -6.72, -8.96, 0.37, -0.08, 3.507, 55.09, 589.3, 29.23, 59.51 },
    { 0.018, -7.48, -0.13, 0.24, -8.69, -58.4, 5.446, -1.19, -27.4, 10336., 16818., 1945., 1707. },
    { -2.24, -8.04, 0.395, 0.177, -11.2, 11.47, -1.13, 0.103, 6.371, 1.735, -9.39, -0.06, -1.12 },
    { -4.43, -9.23, 0.699, 0.132, -12.1, 8.716, -0.63, -0.12, 7.032, 40.88, -44.1, 5.635, -6.18 },
    { -6.49, -11.6, 1.084, 0.113, -11.1, 21.8, -2.01, 0.098, 7.092, 23.86, -23.6, 3.308, -3.39 },
    { -8.34, -17.4, 1.757, 0.142, -11.1, 1.91, 0.15, -0.4, 7.463, 102.8, -92.3, 13.88, -12.7 },
    { -9.93, -48.4, 4.764, 0.462, -9.72, -0.04, 0.243, -0.41, 7.453, -6.14, -1.81, -0.88, -0.19 },
    { -11.3, 9.149, -0.64, -0.14, -8.42, -0.66, 0.147, -0.34, 8.101, -18.1, 10.54, -2.23, 1.295 },
    { -12.2, 1.905, 0.109, -0.08, -7.44, 0.395, -0.18, -0.18, 8.702, -9.05, 0.532, -0.96, -0.02 },
    { -12.8, -0.75, 0.386, -0.06, -6.78, 2.662, -0.67, 0.05, 8.925, -9.03, 0.285, -0.9, -0.08 },
    { -13.0, -1.32, 0.45, -0.05, -6.58, 3.387, -0.84, 0.131, 9.317, -6.89, -2.08, -0.57, -0.4 },
};

/* calcDVFShelfParams (saf_utility_dvf.c:78-101) */
void orc_calcDVFShelfParams(int i, float rhoIn, float* g0, float* gInf, float* fc)
{
    const double* c = dvf_fit[i];
    const double rho = (double)rhoIn, rho2 = rho * rho;
    *g0 = (float)((c[0] * rho + c[1]) / (rho2 + c[2] * rho + c[3]));
    *gInf = (float)((c[4] * rho + c[5]) / (rho2 + c[6] * rho + c[7]));
    const float f = (float)((c[8] * rho2 + c[9] * rho + c[10]) / (rho2 + c[11] * rho + c[12]));
    *fc = f * (343.0f / (2.0f * ORC_PI * 0.09096f));         /* c / (2 pi a_head), :49 */
}

/* interpDVFShelfParams (saf_utility_dvf.c:107-143) */
void orc_interpDVFShelfParams(float theta, float rho, float* iG0, float* iGInf, float* iFc)
{
    theta = theta < 0.f ? 0.f : (theta > 180.f ? 180.f : theta);
    rho = rho < 1.0f ? 1.0f : rho;
    const float t10 = theta / 10.f;
    int lo = (int)t10, hi = lo + 1;
    if (hi >= 19) { hi = 18; lo = 17; }
    float a[3], b[3];
    orc_calcDVFShelfParams(lo, rho, &a[0], &a[1], &a[2]);
    orc_calcDVFShelfParams(hi, rho, &b[0], &b[1], &b[2]);
    const float w = t10 - lo;
    *iG0 = a[0] + (b[0] - a[0]) * w; *iGInf = a[1] + (b[1] - a[1]) * w; *iFc = a[2] + (b[2] - a[2]) * w;
}

/* dvfShelfCoeffs (saf_utility_dvf.c:149-175) */
void orc_dvfShelfCoeffs(float g0, float gInf, float fc, float fs, float* b0, float* b1, float* a1)
{
    const float headDim = ORC_PI * (0.0875f / 0.09096f);     /* :48 */
    const float v0 = powf(10.f, gInf / 20.f), g0m = powf(10.f, g0 / 20.f);
    const float tf = tanf((headDim / fs) * fc), v0t = v0 * tf;
    const float ac = (v0t - 1.f) / (v0t + 1.f);
    const float v = (v0 - 1.f) * 0.5f, vac = v * ac;
    *b0 = g0m * (v - vac + 1.f); *b1 = g0m * (vac - v + ac); *a1 = ac;
}

/* calcDVFCoeffs (saf_utility_dvf.c:177-190): writes b[0], b[1], a[1] */
void orc_calcDVFCoeffs(float alpha, float rho, float fs, float* b, float* a)
{
    float g0, gInf, fc;
    orc_interpDVFShelfParams(alpha, rho, &g0, &gInf, &fc);
    orc_dvfShelfCoeffs(g0, gInf, fc, fs, &b[0], &b[1], &a[1]);
}

/* doaToIpsiInteraural (saf_utility_dvf.c:192-232) */
void orc_doaToIpsiInteraural(float azimuth, float elevation, float* alphaLR, float* betaLR)
{
    const float az = azimuth * ORC_PI / 180.0f, el = elevation * ORC_PI / 180.0f;
    const float sa = sinf(az), se = sinf(el), ca = cosf(az), ce = cosf(el);
    float alpha = ORC_PI / 2.f - acosf(sa * ce);
    float beta = asinf(se / sqrtf(powf(se, 2.f) + (powf(ca, 2.f) * powf(ce, 2.f))));
    if (beta > ORC_PI / 2.f) { alpha = ORC_PI - alpha; beta = ORC_PI - beta; }
    alpha = fabsf(ORC_PI / 2.f - alpha);
    if (alpha > ORC_PI) alpha = 2 * ORC_PI - alpha;
    const float ad = alpha * 180.0f / ORC_PI;
    alphaLR[0] = ad; alphaLR[1] = 180.f - ad;
    if (betaLR) { const float bd = beta * 180.0f / ORC_PI; betaLR[0] = bd; betaLR[1] = 180.f - bd; }
}

/* evalIIRTransferFunctionf (saf_utility_filters.c:609-671) */
void orc_evalIIRTransferFunctionf(const float* b_coeff, const float* a_coeff, int nCoeffs, const float* freqs, int nFreqs, float fs, int mag2dB,
                                  float* magnitude, float* phase_rad)
{
    const float norm_frq = -2.0 * ORC_PI / fs;
    for (int ff = 0; ff < nFreqs; ff++) {
        const float w = freqs[ff] * norm_frq;
        float a = b_coeff[0], b = 0.0f, c = a_coeff[0], d = 0.0f;
        for (int n = 1; n < nCoeffs; n++) {
            const float x = n * w, cx = cosf(x), sx = sinf(x);
            a += b_coeff[n] * cx; b += b_coeff[n] * sx; c += a_coeff[n] * cx; d += a_coeff[n] * sx;
        }
        const double dvsr = 1.0 / (powf(c, 2.f) + powf(d, 2.f) + 2.23e-7f);
        if (magnitude) {
            magnitude[ff] = (float)sqrt((powf(a, 2.0f) + powf(b, 2.0f)) * dvsr);
            if (mag2dB) magnitude[ff] = 20.0f * log10f(magnitude[ff]);
        }
        if (phase_rad) {
            const float hr = (a * c + b * d) * dvsr, hi = (b * c - a * d) * dvsr;
            phase_rad[ff] = (float)atan2(hi, hr);
        }
    }
}
