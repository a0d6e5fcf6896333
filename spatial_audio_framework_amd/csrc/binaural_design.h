/* binaural_design.h — init-time design of binaural Ambisonic decoders and SH rotation (binaural_design.cpp). */
#pragma once
#include "saf_hip_common.h"
namespace saf {
void sh_rot_matrix_real(const float Rxyz[3][3], float* RotMtx, int L);
void yaw_pitch_roll_to_Rzyx(float yaw, float pitch, float roll, int rollPitchYaw, float R[3][3]);
}  // namespace saf
