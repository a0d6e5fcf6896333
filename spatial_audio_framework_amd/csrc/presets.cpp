/*
 * presets.cpp — preset tables of the operators.  The direction sets themselves are data
 * (tables blob); this file only maps preset ids (examples/include/_common.h:88-160) to them.
 */
#include "saf_hip_common.h"
#include "presets.h"

namespace saf {

struct PresetRow { int id; const char* table; int n; };
/* ids follow LOUDSPEAKER_ARRAY_PRESETS / SOURCE_CONFIG_PRESETS (identical numbering from 3 upwards) */
static const PresetRow kRows[] = {
    { 3, "5pX_dirs_deg", 5 },   { 4, "7pX_dirs_deg", 7 },   { 5, "8pX_dirs_deg", 8 },   { 6, "9pX_dirs_deg", 9 },
    { 7, "10pX_dirs_deg", 10 }, { 8, "11pX_dirs_deg", 11 }, { 9, "11pX_7_4_dirs_deg", 11 }, { 10, "13pX_dirs_deg", 13 },
    { 11, "22pX_dirs_deg", 22 }, { 13, "Aalto_MCC_dirs_deg", 45 }, { 14, "Aalto_MCCsubset_dirs_deg", 37 },
    { 15, "Aalto_Apaja_dirs_deg", 29 }, { 16, "Aalto_LR_dirs_deg", 13 }, { 17, "DTU_AVIL_dirs_deg", 64 },
    { 18, "Zylia_Lab_dirs_deg", 22 }, { 19, "Tdesign_degree_2_dirs_deg", 4 }, { 20, "Tdesign_degree_4_dirs_deg", 12 },
    { 21, "Tdesign_degree_6_dirs_deg", 24 }, { 22, "Tdesign_degree_8_dirs_deg", 36 }, { 23, "Tdesign_degree_9_dirs_deg", 48 },
    { 24, "Tdesign_degree_10_dirs_deg", 60 }, { 25, "SphCovering_9_dirs_deg", 9 }, { 26, "SphCovering_16_dirs_deg", 16 },
    { 27, "SphCovering_25_dirs_deg", 25 }, { 28, "SphCovering_49_dirs_deg", 49 }, { 29, "SphCovering_64_dirs_deg", 64 },
};

static void fill(const char* tab, int n, float dirs[][2])
{
    const float* t = table_required(tab, 2 * n);
    for (int ch = 0; ch < n; ch++) { dirs[ch][0] = t[2 * ch]; dirs[ch][1] = t[2 * ch + 1]; }
    const float* def = table_required("default_LScoords64_rad", 128);
    for (int ch = n; ch < SAF_MAXCH; ch++)
        for (int i = 0; i < 2; i++) dirs[ch][i] = def[2 * ch + i] * (180.0f / SAF_PI);
}

void load_loudspeaker_preset(int preset, float dirs[][2], int* newNCH, int* nDims)
{
    const PresetRow* r = &kRows[0];                     /* default / unknown id: 5.x */
    if (preset == 12) SAF_FATAL("loudspeaker preset 22.2 (9+10+3) is not suitable, since it contains LFE channels");
    for (const PresetRow& q : kRows) if (q.id == preset) r = &q;
    fill(r->table, r->n, dirs);
    *newNCH = r->n;
    float sum_elev = 0.0f;
    for (int i = 0; i < r->n; i++) sum_elev += fabsf(dirs[i][1]);
    *nDims = sum_elev < 0.01f ? 2 : 3;
}

void load_source_preset(int preset, float dirs[][2], int* newNCH)
{
    /* SOURCE_CONFIG_PRESETS (_common.h:120-152) = 1 default, 2 mono, 3 stereo, then the loudspeaker list shifted by one */
    const PresetRow* r = nullptr;
    if (preset >= 4) for (const PresetRow& q : kRows) if (q.id == preset - 1) r = &q;
    if (r) { fill(r->table, r->n, dirs); *newNCH = r->n; }
    else if (preset == 3) { fill("stereo_dirs_deg", 2, dirs); *newNCH = 2; }
    else { fill("mono_dirs_deg", 1, dirs); *newNCH = 1; }
}

void mic_preset_order_per_band(int preset, int masterOrder, const float* freqVector, int nBands, int* orderPerBand)
{
    /* MIC_PRESETS (_common.h:78-84): 1 ideal, 2 Zylia, 3 Eigenmike32, 4 DTU */
    if (preset == 1) { for (int b = 0; b < nBands; b++) orderPerBand[b] = masterOrder; return; }
    const char* tab = preset == 2 ? "Zylia_freqRange" : preset == 3 ? "Eigenmike32_freqRange" : preset == 4 ? "DTU_mic_freqRange" : nullptr;
    const int maxOrder = preset == 2 ? 3 : preset == 3 ? 4 : 6;   /* saf_utility_sensorarray_presets.c:329-332 */
    if (!tab) return;
    int d0 = 0, d1 = 0;
    const float* range = table(tab, &d0, &d1);
    if (!range) SAF_FATAL("table %s missing", tab);
    int rangeIdx = 0, curOrder = 1, reverse = 0;
    for (int b = 0; b < nBands; b++) {
        if (rangeIdx < 2 * (maxOrder - 1) && freqVector[b] > range[rangeIdx]) {
            if (!reverse) curOrder++; else curOrder--;
            reverse = (curOrder == maxOrder) || reverse ? 1 : 0;
            rangeIdx++;
        }
        orderPerBand[b] = masterOrder < curOrder ? masterOrder : curOrder;
    }
}

}  // namespace saf
