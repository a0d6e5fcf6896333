/* hrir_host.h — HRIR set registry and init-time HRTF helpers shared by the binaural operators (hrir_host.cpp). */
#pragma once
#include <vector>
namespace saf {
struct DefaultHRIRs {               /* stands in for __default_hrirs & co. (saf_hrir.h:49-61), installed by the caller */
    std::vector<float> hrirs;       /* [N][2][len] */
    std::vector<float> dirs_deg;    /* [N][2] */
    int N = 0, len = 0, fs = 0;
    unsigned long long epoch = 0;
};
const DefaultHRIRs& default_hrirs();
void voronoi_weights(const float* dirs_deg, int nDirs, float* weights);
}  // namespace saf
