/*
 * drc_kernels.hip — the per-band dynamic range compressor of ambi_drc (examples/src/ambi_drc/ambi_drc.c:168-199,
 * ambi_drc_internal.c:46-88) on the filterbank spectra, for gfx950.
 *
 *   drc_gain_kernel   thread = band, hops in order: level of the boosted omni channel -> gain computer (threshold, ratio,
 *                     knee) -> smooth peak detector with attack / release (one recursive state per band) -> gain factor
 *                     with the spectral floor; the factors of the call are kept for the display ring
 *   drc_apply_kernel  every channel of the band times boost * gain * make-up, in place
 */
#include "saf_hip_common.h"

namespace saf {

struct DrcArgs { DrcLaunch l; };

__device__ __forceinline__ float drc_gain_computer(float xG, float T, float R, float W)       /* ambi_drc_internal.c:46-66 */
{
    float yG;
    if (2.0f * (xG - T) < -W) yG = xG;
    else if (2.0f * fabsf(xG - T) <= W) yG = xG + (1.0f / R - 1.0f) * powf(xG - T + W / 2.0f, 2.0f) / (2.0f * W);
    else if (2.0f * (xG - T) > W) yG = T + (xG - T) / R;
    else yG = 0.0f;
    return yG;
}

__global__ __launch_bounds__(192) void drc_gain_kernel(DrcArgs a)
{
#pragma clang fp contract(off)
    const DrcLaunch& l = a.l;
    const int band = threadIdx.x;
    if (band >= SAF_NBANDS) return;
    const float2* X0 = l.X + (long long)band * l.x_band;          /* omni channel */
    float yL = l.yL_z1[band];
    for (int t = 0; t < l.H; t++) {
        float2 x = X0[t];
        x.x *= l.boost; x.y *= l.boost;
        const float mag = hypotf(x.x, x.y);                                               /* cabsf */
        const float xG = 10.0f * log10f(powf(mag, 2.0f) + 2e-13f);
        const float yG = drc_gain_computer(xG, l.threshold, l.ratio, l.knee);
        const float xL = xG - yG;
        yL = xL > yL ? l.alpha_a * yL + (1.0f - l.alpha_a) * xL : l.alpha_r * yL + (1.0f - l.alpha_r) * xL;      /* ambi_drc_internal.c:71-88 */
        const float c = fmaxf(l.floor, sqrtf(powf(10.0f, -yL / 20.0f)));
        l.gains[(long long)band * l.g_band + t] = c;
    }
    l.yL_z1[band] = yL;
}

/* grid (ceil(H/64), nCh, 133) */
__global__ __launch_bounds__(64) void drc_apply_kernel(DrcArgs a)
{
#pragma clang fp contract(off)
    const DrcLaunch& l = a.l;
    const int t = blockIdx.x * 64 + threadIdx.x, ch = blockIdx.y, band = blockIdx.z;
    if (t >= l.H) return;
    float2* p = l.X + (long long)band * l.x_band + (long long)ch * l.x_ch + t;
    float2 v = *p;
    v.x *= l.boost; v.y *= l.boost;                                                        /* crmulf(in, boost) */
    const float g = l.gains[(long long)band * l.g_band + t] * l.makeup;                    /* crmulf(in, cdB * makeup) */
    *p = make_float2(v.x * g, v.y * g);
}

void launch_drc(const DrcLaunch& l)
{
    if (l.H <= 0 || l.nCh <= 0) return;
    DrcArgs a; a.l = l;
    KernelTimer kt("drc_gain");
    hipLaunchKernelGGL(drc_gain_kernel, dim3(1), dim3(192), 0, stream(), a);
    hipLaunchKernelGGL(drc_apply_kernel, dim3((l.H + 63) / 64, l.nCh, SAF_NBANDS), dim3(64), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
