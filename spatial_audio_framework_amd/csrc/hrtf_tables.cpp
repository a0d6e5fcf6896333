/*
 * hrtf_tables.cpp — the binaural-related initialisation of ambi_dec_initCodec (examples/src/ambi_dec/ambi_dec.c:349-445):
 * ITDs, the 2 x 5 degree VBAP interpolation table over the HRIR grid (compressed to 3 gains + 3 indices), HRIR ->
 * filterbank coefficients (GPU analysis), optional diffuse-field equalisation with spherical-Voronoi weights
 * (N <= 3600, :425-433; amplitude only, so the centre frequencies passed along are not used), magnitudes.  The reference
 * rebuilds these per handle; they depend only on the HRIR set and the pre-processing flag, so handles share one copy.
 */
#include "hrtf_tables.h"
#include "../../include/saf_hip.h"
#include "design_host.h"
#include "hrir_host.h"
#include <mutex>

namespace saf {

static std::mutex g_mu;
static std::vector<std::shared_ptr<HrtfTables>> g_cache;

std::shared_ptr<HrtfTables> ambi_dec_hrtf_tables(const float* freqVector, int enablePreProc)
{
    const DefaultHRIRs& D = default_hrirs();
    if (D.N == 0)
        SAF_FATAL("ambi_dec: binauralised output needs an HRIR set.  The reference's default set (saf_default_hrirs.c) is not part of its "
                  "checkout and SOFA loading is outside this library: call saf_hip_setDefaultHRIRs() before ambi_dec_initCodec().");
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& t : g_cache)
        if (t->hrirEpoch == D.epoch && t->preProc == enablePreProc) return t;
    /* drop tables of HRIR sets that are no longer installed and that nobody holds */
    for (size_t i = 0; i < g_cache.size();)
        if (g_cache[i]->hrirEpoch != D.epoch && g_cache[i].use_count() == 1) g_cache.erase(g_cache.begin() + i); else i++;

    auto t = std::make_shared<HrtfTables>();
    t->hrirEpoch = D.epoch; t->preProc = enablePreProc;
    const int N = t->N = D.N; t->len = D.len; t->fs = D.fs;
    t->dirs_deg = D.dirs_deg;
    std::vector<float> hrirs = D.hrirs;
    t->itds_s.resize(N);
    estimateITDs(hrirs.data(), N, t->len, t->fs, t->itds_s.data());
    std::vector<float> grid, gtable;
    vbap_grid_dirs(t->vbapTableRes[0], t->vbapTableRes[1], grid);
    t->N_gtable = (int)grid.size() / 2;
    if (!vbap_table(grid.data(), t->N_gtable, t->dirs_deg.data(), N, 1, 0, 0.0f, gtable, &t->nTriangles))
        SAF_FATAL("ambi_dec: the HRIR measurement grid could not be triangulated");
    t->gtableComp.resize((size_t)t->N_gtable * 3); t->gtableIdx.resize((size_t)t->N_gtable * 3);
    compressVBAPgainTable3D(gtable.data(), t->N_gtable, N, t->gtableComp.data(), t->gtableIdx.data());
    t->hrtf_fb.resize((size_t)SAF_NBANDS * 2 * N);
    HRIRs2HRTFs_afSTFT(hrirs.data(), N, t->len, SAF_HOP, 0, 1, reinterpret_cast<float_complex*>(t->hrtf_fb.data()));
    if (enablePreProc) {
        t->weights.resize(N);
        if (N <= 3600) voronoi_weights(t->dirs_deg.data(), N, t->weights.data());
        else for (int i = 0; i < N; i++) t->weights[i] = 4.f * SAF_PI / (float)N;
        std::vector<float> fv(freqVector, freqVector + SAF_NBANDS);
        diffuseFieldEqualiseHRTFs(N, t->itds_s.data(), fv.data(), SAF_NBANDS, t->weights.data(), 1, 0, reinterpret_cast<float_complex*>(t->hrtf_fb.data()));
    }
    t->hrtf_fb_mag.resize(t->hrtf_fb.size());
    for (size_t i = 0; i < t->hrtf_fb.size(); i++) t->hrtf_fb_mag[i] = hypotf(t->hrtf_fb[i].x, t->hrtf_fb[i].y);

    HIP_CHECK(hipStreamSynchronize(stream()));
    t->d_hrtf_fb.alloc(t->hrtf_fb.size(), false); t->d_mag.alloc(t->hrtf_fb_mag.size(), false); t->d_itds.alloc(N, false);
    t->d_gtComp.alloc(t->gtableComp.size(), false); t->d_gtIdx.alloc(t->gtableIdx.size(), false);
    HIP_CHECK(hipMemcpy(t->d_hrtf_fb.p, t->hrtf_fb.data(), sizeof(float2) * t->hrtf_fb.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(t->d_mag.p, t->hrtf_fb_mag.data(), sizeof(float) * t->hrtf_fb_mag.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(t->d_itds.p, t->itds_s.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(t->d_gtComp.p, t->gtableComp.data(), sizeof(float) * t->gtableComp.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(t->d_gtIdx.p, t->gtableIdx.data(), sizeof(int) * t->gtableIdx.size(), hipMemcpyHostToDevice));
    g_cache.push_back(t);
    return t;
}

}  // namespace saf
