/* hrtf_tables.h — init-time HRTF tables of ambi_dec's binauralised output (ambi_dec.c:349-445), built once per
 * (installed HRIR set, pre-processing flag) and shared by every handle that asks for them. */
#pragma once
#include "saf_hip_common.h"
#include <memory>

namespace saf {
struct HrtfTables {
    int N = 0, len = 0, fs = 0, N_gtable = 0, nTriangles = 0;
    int vbapTableRes[2] = { 2, 5 };
    std::vector<float> dirs_deg, itds_s, weights, hrtf_fb_mag, gtableComp;
    std::vector<int> gtableIdx;
    std::vector<float2> hrtf_fb;
    DevBuf<float> d_mag, d_itds, d_gtComp;
    DevBuf<int> d_gtIdx;
    DevBuf<float2> d_hrtf_fb;
    /* cache key */
    unsigned long long hrirEpoch = 0; int preProc = 0;
};
/* aborts with a message when no HRIR set is installed (saf_hip_setDefaultHRIRs) */
std::shared_ptr<HrtfTables> ambi_dec_hrtf_tables(const float* freqVector, int enablePreProc);
}  // namespace saf
