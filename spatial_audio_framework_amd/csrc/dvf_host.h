/* dvf_host.h — near-field DVF shelf filters on the host (saf_utility_dvf.h:62-153) */
#pragma once
namespace saf {
void dvf_shelf_params(int idx, float rho, float* g0, float* gInf, float* fc);
void dvf_interp_params(float theta, float rho, float* g0, float* gInf, float* fc);
void dvf_shelf_coeffs(float g0_dB, float gInf_dB, float fc, float fs, float* b0, float* b1, float* a1);
void dvf_coeffs(float alpha, float rho, float fs, float* b /* [2] */, float* a /* [2], a[0] untouched */);
void dvf_lateral_angles(float azimuth_deg, float elevation_deg, float* alphaLR /* [2] */, float* betaLR /* [2] or null */);
void iir_response_f(const float* b, const float* a, int nCoeffs, const float* freqs, int nFreqs, float fs, int mag2dB, float* magnitude, float* phase_rad);
}
