/*
 * rotator.cpp — the rotator operator (examples/include/rotator.h:55-263, examples/src/rotator/rotator.c): rotation of an
 * Ambisonic sound scene by a real SH rotation matrix, with its per-block path on the GPU:
 *
 *   previous block -> [MFMA GEMM with M_rot (and prev_M_rot + linear cross-fade when the rotation changed)] -> outputs
 *                                                                                                rotator.c:143-176
 * Same shape as ambi_enc's encode step (the matrix is [nSH x nSH] instead of [nSH x nSources]), so it runs on the same
 * kernels (enc_gemm_*).  The rotation matrix is built on the host (getSHrotMtxReal, saf_sh.c:479-560) when yaw / pitch /
 * roll or the quaternion changed.  Like the reference a call rotates the PREVIOUS block (rotator_getProcessingDelay);
 * order 0 passes the omni channel through without delay (rotator.c:181-182).
 * Also here: the quaternion helpers of saf_utility_geometry.c:89-214 the operator needs.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include "binaural_design.h"

namespace saf {

static int g_rotator_frame_size = 64;       /* default of the reference (rotator_internal.h:53) */

enum { M_ROT_READY = 1, M_ROT_RECOMPUTE_EULER, M_ROT_RECOMPUTE_QUATERNION };     /* rotator_internal.h:62-66 */

struct Rotator {
    int F, fs = 48000;
    quaternion_data Q;
    int bFlipQuaternion, bFlipYaw, bFlipPitch, bFlipRoll, useRollPitchYawFlag;
    float yaw, pitch, roll;
    CH_ORDER chOrdering; NORM_TYPES norm; int inputOrder;
    int M_rot_status;
    float M_rot[64 * 64], prev_M_rot[64 * 64];     /* row-major, zero padded (rotator_internal.h: MAX_NUM_SH_SIGNALS^2) */
    /* device side */
    bool ready = false, clearState = true;
    int maxFrames = 1, par = 0;
    DevBuf<float> Afrag;                     /* two slots [2][32][64]: M_rot in slot `cur`, prev_M_rot in the other (swapped, not copied) */
    int cur = 0;
    DevBuf<float> prev[2];                   /* [64][F] previous block (ping-pong) */
    DevBuf<float> fpar;                      /* postScale | gains[64] | rowScale[64] */
    DevBuf<int> ipar;                        /* nSrc | mix | order | rowMap[64] */
    PinBuf<float> hf, hA, h_in, h_out;
    PinBuf<int> hi;
    DevBuf<float> d_in, d_out;
    int shadowI[3] = { -1, -1, -1 };
    bool stagingBusy = false;                /* hA may still be read by a copy enqueued by a device-entry call */
};

static void rot_setup(Rotator* p)
{
    if (p->ready) return;
    ensure_device();
    const int F = p->F;
    p->Afrag.alloc(2 * 4096);
    p->prev[0].alloc((size_t)SAF_MAXCH * F); p->prev[1].alloc((size_t)SAF_MAXCH * F);
    p->fpar.alloc(1 + 64 + 64); p->ipar.alloc(3 + 64);
    p->hf.ensure(1 + 64 + 64); p->hi.ensure(3 + 64); p->hA.ensure(2 * 4096);
    p->hf.p[0] = 1.0f;
    for (int i = 0; i < 64; i++) { p->hf.p[1 + i] = 1.0f; p->hf.p[65 + i] = 1.0f; p->hi.p[3 + i] = i; }
    HIP_CHECK(hipMemcpyAsync(p->fpar.p, p->hf.p, sizeof(float) * 129, hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
    p->ready = true;
}

/* rotator.c:126-176 for nFrames consecutive blocks at device-accessible addresses (ACN channel order) */
static void rot_run(Rotator* p, const float* in, long long in_frame, long long in_ch, int nIn, float* out, long long out_frame, long long out_ch, int nOut, int nFrames)
{
    rot_setup(p);
    const int F = p->F, order = p->inputOrder, nSH = ORDER2NSH(order);
    if (p->clearState) {                                     /* rotator_init (rotator.c:93-96) */
        HIP_CHECK(hipMemsetAsync(p->prev[p->par].p, 0, sizeof(float) * (size_t)SAF_MAXCH * F, stream()));
        HIP_CHECK(hipMemsetAsync(p->Afrag.p, 0, sizeof(float) * 2 * 4096, stream()));      /* M_rot and prev_M_rot are zeroed with the state (rotator.c:93-96) */
        p->clearState = false;
    }
    int mix = 0;
    if (p->M_rot_status != M_ROT_READY) {
        float Rxyz[3][3];
        if (p->M_rot_status == M_ROT_RECOMPUTE_EULER) {
            yaw_pitch_roll_to_Rzyx(p->yaw, p->pitch, p->roll, p->useRollPitchYawFlag, Rxyz);
            euler2Quaternion(p->yaw, p->pitch, p->roll, 0, p->useRollPitchYawFlag ? EULER_ROTATION_ROLL_PITCH_YAW : EULER_ROTATION_YAW_PITCH_ROLL, &p->Q);
        } else {
            quaternion2rotationMatrix(&p->Q, Rxyz);
            quaternion2euler(&p->Q, 0, p->useRollPitchYawFlag ? EULER_ROTATION_ROLL_PITCH_YAW : EULER_ROTATION_YAW_PITCH_ROLL, &p->yaw, &p->pitch, &p->roll);
        }
        std::vector<float> M((size_t)nSH * nSH);
        sh_rot_matrix_real(Rxyz, M.data(), order);
        memset(p->M_rot, 0, sizeof(p->M_rot));
        for (int i = 0; i < nSH; i++) for (int j = 0; j < nSH; j++) p->M_rot[i * 64 + j] = M[(size_t)i * nSH + j];
        mix = 1;
        p->M_rot_status = M_ROT_READY;
        if (p->stagingBusy) { HIP_CHECK(hipStreamSynchronize(stream())); p->stagingBusy = false; }     /* the staging buffer may still be in flight */
        p->cur ^= 1;                                         /* the slot of the matrix before last is free; the other one holds prev_M_rot */
        pack_A(p->M_rot, p->hA.p);
        HIP_CHECK(hipMemcpyAsync(p->Afrag.p + p->cur * 4096, p->hA.p, sizeof(float) * 4096, hipMemcpyHostToDevice, stream()));
        p->stagingBusy = true;
    }
    const int nSrc = nSH < nIn ? nSH : nIn;
    if (p->shadowI[0] != nSrc || p->shadowI[1] != mix || p->shadowI[2] != order) {
        HIP_CHECK(hipStreamSynchronize(stream()));
        p->hi.p[0] = nSrc; p->hi.p[1] = mix; p->hi.p[2] = order;
        HIP_CHECK(hipMemcpyAsync(p->ipar.p, p->hi.p, sizeof(int) * 3, hipMemcpyHostToDevice, stream()));
        if (p->shadowI[0] < 0) HIP_CHECK(hipMemcpyAsync(p->ipar.p + 3, p->hi.p + 3, sizeof(int) * 64, hipMemcpyHostToDevice, stream()));
        p->shadowI[0] = nSrc; p->shadowI[1] = mix; p->shadowI[2] = order;
    }
    EncLaunch e{};
    e.in = in; e.in_inst = 0; e.in_frame = in_frame; e.in_ch = in_ch;
    e.out = out; e.out_inst = 0; e.out_frame = out_frame; e.out_ch = out_ch;
    e.prev_rd = p->prev[p->par].p; e.prev_wr = p->prev[p->par ^ 1].p;
    e.Afrag = p->Afrag.p + p->cur * 4096; e.AfragPrev = p->Afrag.p + (p->cur ^ 1) * 4096; e.postScale = p->fpar.p; e.gains = p->fpar.p + 1; e.rowScale = p->fpar.p + 65;
    e.nSrc = p->ipar.p; e.mix = mix ? p->ipar.p + 1 : nullptr; e.order = p->ipar.p + 2; e.rowMap = p->ipar.p + 3;
    e.F = F; e.nFrames = nFrames; e.nInst = 1; e.nOut = nOut < SAF_MAXCH ? nOut : SAF_MAXCH;
    e.maxSteps = (nSrc + 1) / 2; e.rowsIn = nSrc;
    launch_enc_gemm(e);
    p->par ^= 1;
    if (mix) memcpy(p->prev_M_rot, p->M_rot, sizeof(p->M_rot));     /* prev_M_rot <- M_rot (rotator.c:170): on the device the slots swap at the next change */
}

}  // namespace saf

using namespace saf;

extern "C" {

/* ---------------- quaternion helpers (saf_utility_geometry.h; saf_utility_geometry.c:89-214) ---------------- */
void quaternion2rotationMatrix(quaternion_data* Q, float R[3][3])
{
    R[0][0] = 2.0f * (Q->w * Q->w + Q->z * Q->z) - 1.0f; R[0][1] = 2.0f * (Q->z * Q->y - Q->w * Q->x); R[0][2] = 2.0f * (Q->z * Q->x + Q->w * Q->y);
    R[1][0] = 2.0f * (Q->z * Q->y + Q->w * Q->x); R[1][1] = 2.0f * (Q->w * Q->w + Q->y * Q->y) - 1.0f; R[1][2] = 2.0f * (Q->y * Q->x - Q->w * Q->z);
    R[2][0] = 2.0f * (Q->z * Q->x - Q->w * Q->y); R[2][1] = 2.0f * (Q->y * Q->x + Q->w * Q->z); R[2][2] = 2.0f * (Q->w * Q->w + Q->x * Q->x) - 1.0f;
}

void rotationMatrix2quaternion(float R[3][3], quaternion_data* Q)
{
    Q->w = sqrtf(fmaxf(0.0f, 1.0f + R[0][0] + R[1][1] + R[2][2])) / 2.0f;
    Q->z = sqrtf(fmaxf(0.0f, 1.0f + R[0][0] - R[1][1] - R[2][2])) / 2.0f;
    Q->y = sqrtf(fmaxf(0.0f, 1.0f - R[0][0] + R[1][1] - R[2][2])) / 2.0f;
    Q->x = sqrtf(fmaxf(0.0f, 1.0f - R[0][0] - R[1][1] + R[2][2])) / 2.0f;
    Q->z = copysignf(Q->z, R[2][1] - R[1][2]);
    Q->y = copysignf(Q->y, R[0][2] - R[2][0]);
    Q->x = copysignf(Q->x, R[1][0] - R[0][1]);
}

void euler2Quaternion(float alpha, float beta, float gamma, int degreesFlag, EULER_ROTATION_CONVENTIONS convention, quaternion_data* Q)
{
    const float k = degreesFlag ? SAF_PI / 180.0f : 1.0f;
    float a, c;                                              /* the angles paired with "y" (yaw slot) and "r" (roll slot) */
    switch (convention) {
        case EULER_ROTATION_YAW_PITCH_ROLL: a = alpha; c = gamma; break;
        case EULER_ROTATION_ROLL_PITCH_YAW: a = gamma; c = alpha; break;
        default: SAF_FATAL("euler2Quaternion: this convention is not supported");
    }
    const float cy = cosf(a * k * 0.5f), sy = sinf(a * k * 0.5f), cp = cosf(beta * k * 0.5f), sp = sinf(beta * k * 0.5f);
    const float cr = cosf(c * k * 0.5f), sr = sinf(c * k * 0.5f);
    Q->w = cy * cr * cp + sy * sr * sp;
    Q->x = cy * sr * cp - sy * cr * sp;
    Q->y = cy * cr * sp + sy * sr * cp;
    Q->z = sy * cr * cp - cy * sr * sp;
}

void quaternion2euler(quaternion_data* Q, int degreesFlag, EULER_ROTATION_CONVENTIONS convention, float* alpha, float* beta, float* gamma)
{
    const float sinr_cosp = 2.0f * (Q->w * Q->x + Q->y * Q->z), cosr_cosp = 1.0f - 2.0f * (Q->x * Q->x + Q->y * Q->y);
    const float sinp = 2.0f * (Q->w * Q->y - Q->z * Q->x);
    const float siny_cosp = 2.0f * (Q->w * Q->z + Q->x * Q->y), cosy_cosp = 1.0f - 2.0f * (Q->y * Q->y + Q->z * Q->z);
    const float pitch = fabsf(sinp) >= 1.0f ? copysignf(SAF_PI / 2.0f, sinp) : asinf(sinp);
    switch (convention) {
        case EULER_ROTATION_YAW_PITCH_ROLL: *gamma = atan2f(sinr_cosp, cosr_cosp); *beta = pitch; *alpha = atan2f(siny_cosp, cosy_cosp); break;
        case EULER_ROTATION_ROLL_PITCH_YAW: *alpha = atan2f(sinr_cosp, cosr_cosp); *beta = pitch; *gamma = atan2f(siny_cosp, cosy_cosp); break;
        default: SAF_FATAL("quaternion2euler: this convention is not supported");
    }
    if (degreesFlag) { *alpha *= 180.0f / SAF_PI; *beta *= 180.0f / SAF_PI; *gamma *= 180.0f / SAF_PI; }
}

/* ---------------- rotator (rotator.h) ---------------- */
void saf_hip_rotator_setFrameSize(int frameSize)
{
    if (frameSize <= 0 || frameSize % 4 != 0) SAF_FATAL("rotator frame size must be a positive multiple of 4");
    g_rotator_frame_size = frameSize;
}

#define PR Rotator* p = (Rotator*)hRot
static inline float deg2rad(float d) { return d * SAF_PI / 180.0f; }
static inline float rad2deg(float r) { return r * 180.0f / SAF_PI; }

void rotator_setOrder(void* const hRot, int newOrder);

void rotator_create(void** const phRot)
{
    Rotator* p = new Rotator();
    *phRot = p;
    p->F = g_rotator_frame_size;
    p->M_rot_status = M_ROT_RECOMPUTE_QUATERNION;
    p->Q.w = 1.0f; p->Q.x = p->Q.y = p->Q.z = 0.0f;
    p->bFlipQuaternion = 0; p->yaw = p->pitch = p->roll = 0.0f; p->bFlipYaw = p->bFlipPitch = p->bFlipRoll = 0;
    p->chOrdering = CH_ACN; p->norm = NORM_SN3D; p->useRollPitchYawFlag = 0;
    memset(p->M_rot, 0, sizeof(p->M_rot)); memset(p->prev_M_rot, 0, sizeof(p->prev_M_rot));
    rotator_setOrder(p, SH_ORDER_FIRST);
}

void rotator_destroy(void** const phRot)
{
    Rotator* p = (Rotator*)*phRot;
    if (!p) return;
    if (p->ready) HIP_CHECK(hipStreamSynchronize(stream()));
    delete p;
    *phRot = nullptr;
}

void rotator_init(void* const hRot, int sampleRate)
{
    PR;
    p->fs = sampleRate;
    memset(p->M_rot, 0, sizeof(p->M_rot)); memset(p->prev_M_rot, 0, sizeof(p->prev_M_rot));
    p->clearState = true;
    p->M_rot_status = M_ROT_RECOMPUTE_QUATERNION;
}

void rotator_process(void* const hRot, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples)
{
    PR;
    const int F = p->F, order = p->inputOrder, nSH = ORDER2NSH(order);
    if (nSamples != F) {                                                  /* rotator.c:197-200 */
        for (int i = 0; i < nOutputs; i++) memset(outputs[i], 0, sizeof(float) * F);
        return;
    }
    if (order <= 0) {                                                     /* the omni cannot be rotated: passed through, no delay */
        if (nOutputs > 0) { if (nInputs > 0) memcpy(outputs[0], inputs[0], sizeof(float) * F); else memset(outputs[0], 0, sizeof(float) * F); }
        for (int i = 1; i < nOutputs; i++) memset(outputs[i], 0, sizeof(float) * F);
        return;
    }
    rot_setup(p);
    p->h_in.ensure((size_t)SAF_MAXCH * F); p->h_out.ensure((size_t)SAF_MAXCH * F);
    if (!p->d_in.p) { p->d_in.alloc((size_t)SAF_MAXCH * F); p->d_out.alloc((size_t)SAF_MAXCH * F); }
    /* FuMa WXYZ <-> ACN WYZX (first order only, saf_hoa.c:40-70): done while gathering / scattering the channels */
    static const int fuma2acn[4] = { 0, 2, 3, 1 };                        /* ACN channel c reads FuMa channel fuma2acn[c] */
    const bool fuma = p->chOrdering == CH_FUMA && order == 1;
    for (int c = 0; c < nSH; c++) {
        const int src = fuma ? fuma2acn[c] : c;
        if (src < nInputs) memcpy(p->h_in.p + (size_t)c * F, inputs[src], sizeof(float) * F);
        else memset(p->h_in.p + (size_t)c * F, 0, sizeof(float) * F);
    }
    if (zero_copy_io()) rot_run(p, p->h_in.p, 0, F, nSH, p->h_out.p, 0, F, nSH, 1);
    else {
        HIP_CHECK(hipMemcpyAsync(p->d_in.p, p->h_in.p, sizeof(float) * (size_t)nSH * F, hipMemcpyHostToDevice, stream()));
        rot_run(p, p->d_in.p, 0, F, nSH, p->d_out.p, 0, F, nSH, 1);
        HIP_CHECK(hipMemcpyAsync(p->h_out.p, p->d_out.p, sizeof(float) * (size_t)nSH * F, hipMemcpyDeviceToHost, stream()));
    }
    HIP_CHECK(hipStreamSynchronize(stream()));
    p->stagingBusy = false;
    for (int c = 0; c < nSH; c++) {
        const int dst = fuma ? fuma2acn[c] : c;                           /* ACN channel c goes to FuMa channel fuma2acn[c] */
        if (dst < nOutputs) memcpy(outputs[dst], p->h_out.p + (size_t)c * F, sizeof(float) * F);
    }
    for (int i = nSH; i < nOutputs; i++) memset(outputs[i], 0, sizeof(float) * F);
}

void saf_hip_rotator_process_dev(void* const hRot, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                 float* d_out, long long out_frame_stride, long long out_ch_stride, int nOutputs, int nFrames)
{
    PR;
    if (nFrames <= 0) return;
    if (p->inputOrder <= 0) SAF_FATAL("saf_hip_rotator_process_dev: order 0 has nothing to rotate");
    if (p->chOrdering == CH_FUMA) SAF_FATAL("saf_hip_rotator_process_dev takes ACN channel order (convert FuMa on the host entry)");
    rot_run(p, d_in, in_frame_stride, in_ch_stride, nInputs, d_out, out_frame_stride, out_ch_stride, nOutputs, nFrames);
}

int rotator_getFrameSize(void) { return g_rotator_frame_size; }
void rotator_setYaw(void* const hRot, float v) { PR; p->yaw = p->bFlipYaw == 1 ? -deg2rad(v) : deg2rad(v); p->M_rot_status = M_ROT_RECOMPUTE_EULER; }
void rotator_setPitch(void* const hRot, float v) { PR; p->pitch = p->bFlipPitch == 1 ? -deg2rad(v) : deg2rad(v); p->M_rot_status = M_ROT_RECOMPUTE_EULER; }
void rotator_setRoll(void* const hRot, float v) { PR; p->roll = p->bFlipRoll == 1 ? -deg2rad(v) : deg2rad(v); p->M_rot_status = M_ROT_RECOMPUTE_EULER; }
void rotator_setQuaternionW(void* const hRot, float v) { PR; p->Q.w = v; p->M_rot_status = M_ROT_RECOMPUTE_QUATERNION; }
void rotator_setQuaternionX(void* const hRot, float v) { PR; p->Q.x = p->bFlipQuaternion == 1 ? -v : v; p->M_rot_status = M_ROT_RECOMPUTE_QUATERNION; }
void rotator_setQuaternionY(void* const hRot, float v) { PR; p->Q.y = p->bFlipQuaternion == 1 ? -v : v; p->M_rot_status = M_ROT_RECOMPUTE_QUATERNION; }
void rotator_setQuaternionZ(void* const hRot, float v) { PR; p->Q.z = p->bFlipQuaternion == 1 ? -v : v; p->M_rot_status = M_ROT_RECOMPUTE_QUATERNION; }
float rotator_getYaw(void* const hRot) { PR; return p->bFlipYaw == 1 ? -rad2deg(p->yaw) : rad2deg(p->yaw); }
float rotator_getPitch(void* const hRot) { PR; return p->bFlipPitch == 1 ? -rad2deg(p->pitch) : rad2deg(p->pitch); }
float rotator_getRoll(void* const hRot) { PR; return p->bFlipRoll == 1 ? -rad2deg(p->roll) : rad2deg(p->roll); }
float rotator_getQuaternionW(void* const hRot) { PR; return p->Q.w; }
float rotator_getQuaternionX(void* const hRot) { PR; return p->bFlipQuaternion == 1 ? -p->Q.x : p->Q.x; }
float rotator_getQuaternionY(void* const hRot) { PR; return p->bFlipQuaternion == 1 ? -p->Q.y : p->Q.y; }
float rotator_getQuaternionZ(void* const hRot) { PR; return p->bFlipQuaternion == 1 ? -p->Q.z : p->Q.z; }
void rotator_setFlipYaw(void* const hRot, int s) { PR; if (s != p->bFlipYaw) { p->bFlipYaw = s; rotator_setYaw(hRot, -rotator_getYaw(hRot)); } }
void rotator_setFlipPitch(void* const hRot, int s) { PR; if (s != p->bFlipPitch) { p->bFlipPitch = s; rotator_setPitch(hRot, -rotator_getPitch(hRot)); } }
void rotator_setFlipRoll(void* const hRot, int s) { PR; if (s != p->bFlipRoll) { p->bFlipRoll = s; rotator_setRoll(hRot, -rotator_getRoll(hRot)); } }
void rotator_setFlipQuaternion(void* const hRot, int s)
{
    PR;
    if (s != p->bFlipQuaternion) {
        p->bFlipQuaternion = s;
        rotator_setQuaternionX(hRot, -rotator_getQuaternionX(hRot));
        rotator_setQuaternionY(hRot, -rotator_getQuaternionY(hRot));
        rotator_setQuaternionZ(hRot, -rotator_getQuaternionZ(hRot));
    }
}
void rotator_setRPYflag(void* const hRot, int s) { PR; p->useRollPitchYawFlag = s; }
void rotator_setChOrder(void* const hRot, int o) { PR; if ((CH_ORDER)o != CH_FUMA || p->inputOrder == SH_ORDER_FIRST) p->chOrdering = (CH_ORDER)o; }
void rotator_setNormType(void* const hRot, int t) { PR; if ((NORM_TYPES)t != NORM_FUMA || p->inputOrder == SH_ORDER_FIRST) p->norm = (NORM_TYPES)t; }
void rotator_setOrder(void* const hRot, int newOrder)
{
    PR;
    p->inputOrder = newOrder < 0 ? 0 : (newOrder > SAF_MAX_ORDER ? SAF_MAX_ORDER : newOrder);
    p->M_rot_status = M_ROT_RECOMPUTE_QUATERNION;
    if (p->inputOrder != SH_ORDER_FIRST && p->chOrdering == CH_FUMA) p->chOrdering = CH_ACN;
    if (p->inputOrder != SH_ORDER_FIRST && p->norm == NORM_FUMA) p->norm = NORM_SN3D;
}
int rotator_getFlipYaw(void* const hRot) { PR; return p->bFlipYaw; }
int rotator_getFlipPitch(void* const hRot) { PR; return p->bFlipPitch; }
int rotator_getFlipRoll(void* const hRot) { PR; return p->bFlipRoll; }
int rotator_getFlipQuaternion(void* const hRot) { PR; return p->bFlipQuaternion; }
int rotator_getRPYflag(void* const hRot) { PR; return p->useRollPitchYawFlag; }
int rotator_getChOrder(void* const hRot) { PR; return (int)p->chOrdering; }
int rotator_getNormType(void* const hRot) { PR; return (int)p->norm; }
int rotator_getOrder(void* const hRot) { PR; return p->inputOrder; }
int rotator_getNSHrequired(void* const hRot) { PR; return (p->inputOrder + 1) * (p->inputOrder + 1); }
int rotator_getProcessingDelay(void) { return g_rotator_frame_size; }

}
