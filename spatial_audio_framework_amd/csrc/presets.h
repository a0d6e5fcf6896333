/* presets.h — loudspeaker / source / microphone presets shared by the operators (presets.cpp). */
#pragma once
namespace saf {
/* loadLoudspeakerArrayPreset (examples/src/ambi_dec/ambi_dec_internal.c:117-313); preset ids: _common.h LOUDSPEAKER_ARRAY_PRESETS */
void load_loudspeaker_preset(int preset, float dirs_deg[][2], int* newNCH, int* nDims);
/* loadSourceConfigPreset (examples/src/ambi_enc/ambi_enc_internal.c:30-230); preset ids: _common.h SOURCE_CONFIG_PRESETS */
void load_source_preset(int preset, float dirs_deg[][2], int* newNCH);
/* ambi_dec_setSourcePreset (examples/src/ambi_dec/ambi_dec.c:703-767): usable order per band of a microphone array */
void mic_preset_order_per_band(int preset, int masterOrder, const float* freqVector, int nBands, int* orderPerBand);
}
