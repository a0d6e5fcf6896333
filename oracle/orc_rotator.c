/*
 * orc_rotator.c — CPU restatement of the rotator example (examples/src/rotator/rotator.c:34-436) and of the quaternion
 * helpers it uses (saf_utility_geometry.c:89-214).  TEST INFRASTRUCTURE ONLY (see saf_oracle.h): the product never calls it.
 * Pinned by the reference's own known-answer test test__saf_example_rotator (test/src/test__examples.c:357-440, 1e-6).
 */
#include "saf_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define RPI 3.14159265358979323846264338327950288f
#define NSHMAX 64

typedef struct {
    int F, order, chOrdering, norm, useRPY, status;            /* status: 1 ready, 2 recompute from Euler angles, 3 from the quaternion */
    int flipQ, flipYaw, flipPitch, flipRoll;
    float q[4], yaw, pitch, roll;                              /* q = w x y z; angles in radians */
    float M[NSHMAX][NSHMAX], prevM[NSHMAX][NSHMAX];
    float* prevIn;                                             /* [64][F] */
    float* fadeIn; float* fadeOut;
} orc_rot;

void orc_quaternion2rotationMatrix(const float q[4], float R[9])
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 2.0f * (w * w + z * z) - 1.0f; R[1] = 2.0f * (z * y - w * x); R[2] = 2.0f * (z * x + w * y);
    R[3] = 2.0f * (z * y + w * x); R[4] = 2.0f * (w * w + y * y) - 1.0f; R[5] = 2.0f * (y * x - w * z);
    R[6] = 2.0f * (z * x - w * y); R[7] = 2.0f * (y * x + w * z); R[8] = 2.0f * (w * w + x * x) - 1.0f;
}

/* saf_utility_geometry.c:107-121: component magnitudes from the diagonal, signs from the antisymmetric part */
void orc_rotationMatrix2quaternion(const float R[9], float q[4])
{
    q[0] = sqrtf(fmaxf(0.0f, 1.0f + R[0] + R[4] + R[8])) / 2.0f;
    q[3] = copysignf(sqrtf(fmaxf(0.0f, 1.0f + R[0] - R[4] - R[8])) / 2.0f, R[7] - R[5]);
    q[2] = copysignf(sqrtf(fmaxf(0.0f, 1.0f - R[0] + R[4] - R[8])) / 2.0f, R[2] - R[6]);
    q[1] = copysignf(sqrtf(fmaxf(0.0f, 1.0f - R[0] - R[4] + R[8])) / 2.0f, R[3] - R[1]);
}

/* convention: 2 = yaw-pitch-roll, 3 = roll-pitch-yaw (EULER_ROTATION_CONVENTIONS); radians */
void orc_euler2Quaternion(float alpha, float beta, float gamma, int convention, float q[4])
{
    const float a = convention == 2 ? alpha : gamma, c = convention == 2 ? gamma : alpha;
    const float cy = cosf(a * 0.5f), sy = sinf(a * 0.5f), cp = cosf(beta * 0.5f), sp = sinf(beta * 0.5f), cr = cosf(c * 0.5f), sr = sinf(c * 0.5f);
    q[0] = cy * cr * cp + sy * sr * sp;
    q[1] = cy * sr * cp - sy * cr * sp;
    q[2] = cy * cr * sp + sy * sr * cp;
    q[3] = sy * cr * cp - cy * sr * sp;
}

void orc_quaternion2euler(const float q[4], int convention, float* alpha, float* beta, float* gamma)
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float sinr_cosp = 2.0f * (w * x + y * z), cosr_cosp = 1.0f - 2.0f * (x * x + y * y);
    const float sinp = 2.0f * (w * y - z * x);
    const float siny_cosp = 2.0f * (w * z + x * y), cosy_cosp = 1.0f - 2.0f * (y * y + z * z);
    const float pitch = fabsf(sinp) >= 1.0f ? copysignf(RPI / 2.0f, sinp) : asinf(sinp);
    if (convention == 2) { *gamma = atan2f(sinr_cosp, cosr_cosp); *beta = pitch; *alpha = atan2f(siny_cosp, cosy_cosp); }
    else { *alpha = atan2f(sinr_cosp, cosr_cosp); *beta = pitch; *gamma = atan2f(siny_cosp, cosy_cosp); }
}

void orc_rotator_setOrder(void* h, int o);

void orc_rotator_create(void** ph, int F)
{
    orc_rot* p = (orc_rot*)calloc(1, sizeof(orc_rot));
    p->F = F; p->status = 3; p->q[0] = 1.0f; p->chOrdering = 1; p->norm = 2;
    p->prevIn = (float*)calloc((size_t)NSHMAX * F, sizeof(float));
    p->fadeIn = (float*)malloc(sizeof(float) * F); p->fadeOut = (float*)malloc(sizeof(float) * F);
    *ph = p;
    orc_rotator_setOrder(p, 1);
}
void orc_rotator_destroy(void** ph) { orc_rot* p = (orc_rot*)*ph; if (!p) return; free(p->prevIn); free(p->fadeIn); free(p->fadeOut); free(p); *ph = NULL; }

void orc_rotator_init(void* h, int fs)       /* rotator.c:76-98 */
{
    orc_rot* p = (orc_rot*)h; (void)fs;
    for (int i = 1; i <= p->F; i++) { p->fadeIn[i - 1] = (float)i * 1.0f / (float)p->F; p->fadeOut[i - 1] = 1.0f - p->fadeIn[i - 1]; }
    memset(p->M, 0, sizeof(p->M)); memset(p->prevM, 0, sizeof(p->prevM));
    memset(p->prevIn, 0, sizeof(float) * (size_t)NSHMAX * p->F);
    p->status = 3;
}

void orc_rotator_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples)   /* rotator.c:100-201 */
{
    orc_rot* p = (orc_rot*)h;
    const int F = p->F, order = p->order, nSH = (order + 1) * (order + 1);
    if (nSamples != F) { for (int i = 0; i < nOutputs; i++) memset(outputs[i], 0, sizeof(float) * F); return; }
    float* in = (float*)calloc((size_t)NSHMAX * F, sizeof(float));
    float* out = (float*)calloc((size_t)NSHMAX * F, sizeof(float));
    for (int i = 0; i < (nSH < nInputs ? nSH : nInputs); i++) memcpy(in + (size_t)i * F, inputs[i], sizeof(float) * F);
    if (p->chOrdering == 2 && order == 1) {                      /* FuMa WXYZ -> ACN WYZX (saf_hoa.c:40-70) */
        float* t = (float*)malloc(sizeof(float) * 4 * F);
        memcpy(t, in, sizeof(float) * 4 * F);
        memcpy(in + F, t + 2 * F, sizeof(float) * F); memcpy(in + 2 * F, t + 3 * F, sizeof(float) * F); memcpy(in + 3 * F, t + F, sizeof(float) * F);
        free(t);
    }
    if (order > 0) {
        int mix = 0;
        if (p->status != 1) {
            float R[9];
            memset(p->M, 0, sizeof(p->M));
            if (p->status == 2) {
                orc_yawPitchRoll2Rzyx(p->yaw, p->pitch, p->roll, p->useRPY, R);
                orc_euler2Quaternion(p->yaw, p->pitch, p->roll, p->useRPY ? 3 : 2, p->q);
            } else {
                orc_quaternion2rotationMatrix(p->q, R);
                orc_quaternion2euler(p->q, p->useRPY ? 3 : 2, &p->yaw, &p->pitch, &p->roll);
            }
            float* Mt = (float*)malloc(sizeof(float) * nSH * nSH);
            orc_getSHrotMtxReal(R, Mt, order);
            for (int i = 0; i < nSH; i++) for (int j = 0; j < nSH; j++) p->M[i][j] = Mt[i * nSH + j];
            free(Mt);
            mix = 1; p->status = 1;
        }
        for (int i = 0; i < nSH; i++)
            for (int n = 0; n < F; n++) {
                float a = 0.0f;
                for (int j = 0; j < nSH; j++) a += p->M[i][j] * p->prevIn[(size_t)j * F + n];
                out[(size_t)i * F + n] = a;
            }
        if (mix) {
            for (int i = 0; i < nSH; i++)
                for (int n = 0; n < F; n++) {
                    float b = 0.0f;
                    for (int j = 0; j < nSH; j++) b += p->prevM[i][j] * p->prevIn[(size_t)j * F + n];
                    const float fi = p->fadeIn[n] * out[(size_t)i * F + n], fo = p->fadeOut[n] * b;
                    out[(size_t)i * F + n] = fi + fo;
                }
            memcpy(p->prevM, p->M, sizeof(p->M));
        }
        memcpy(p->prevIn, in, sizeof(float) * (size_t)NSHMAX * F);
    } else
        memcpy(out, in, sizeof(float) * F);
    if (p->chOrdering == 2 && order == 1) {                      /* ACN -> FuMa */
        float* t = (float*)malloc(sizeof(float) * 4 * F);
        memcpy(t, out, sizeof(float) * 4 * F);
        memcpy(out + F, t + 3 * F, sizeof(float) * F); memcpy(out + 2 * F, t + F, sizeof(float) * F); memcpy(out + 3 * F, t + 2 * F, sizeof(float) * F);
        free(t);
    }
    int i;
    for (i = 0; i < (nSH < nOutputs ? nSH : nOutputs); i++) memcpy(outputs[i], out + (size_t)i * F, sizeof(float) * F);
    for (; i < nOutputs; i++) memset(outputs[i], 0, sizeof(float) * F);
    free(in); free(out);
}

#define RP orc_rot* p = (orc_rot*)h
void orc_rotator_setYaw(void* h, float v) { RP; p->yaw = p->flipYaw == 1 ? -(v * RPI / 180.0f) : v * RPI / 180.0f; p->status = 2; }
void orc_rotator_setPitch(void* h, float v) { RP; p->pitch = p->flipPitch == 1 ? -(v * RPI / 180.0f) : v * RPI / 180.0f; p->status = 2; }
void orc_rotator_setRoll(void* h, float v) { RP; p->roll = p->flipRoll == 1 ? -(v * RPI / 180.0f) : v * RPI / 180.0f; p->status = 2; }
void orc_rotator_setQuaternionW(void* h, float v) { RP; p->q[0] = v; p->status = 3; }
void orc_rotator_setQuaternionX(void* h, float v) { RP; p->q[1] = p->flipQ == 1 ? -v : v; p->status = 3; }
void orc_rotator_setQuaternionY(void* h, float v) { RP; p->q[2] = p->flipQ == 1 ? -v : v; p->status = 3; }
void orc_rotator_setQuaternionZ(void* h, float v) { RP; p->q[3] = p->flipQ == 1 ? -v : v; p->status = 3; }
float orc_rotator_getYaw(void* h) { RP; return p->flipYaw == 1 ? -(p->yaw * 180.0f / RPI) : p->yaw * 180.0f / RPI; }
float orc_rotator_getPitch(void* h) { RP; return p->flipPitch == 1 ? -(p->pitch * 180.0f / RPI) : p->pitch * 180.0f / RPI; }
float orc_rotator_getRoll(void* h) { RP; return p->flipRoll == 1 ? -(p->roll * 180.0f / RPI) : p->roll * 180.0f / RPI; }
float orc_rotator_getQuaternionW(void* h) { RP; return p->q[0]; }
float orc_rotator_getQuaternionX(void* h) { RP; return p->flipQ == 1 ? -p->q[1] : p->q[1]; }
float orc_rotator_getQuaternionY(void* h) { RP; return p->flipQ == 1 ? -p->q[2] : p->q[2]; }
float orc_rotator_getQuaternionZ(void* h) { RP; return p->flipQ == 1 ? -p->q[3] : p->q[3]; }
void orc_rotator_setFlipYaw(void* h, int s) { RP; if (s != p->flipYaw) { p->flipYaw = s; orc_rotator_setYaw(h, -orc_rotator_getYaw(h)); } }
void orc_rotator_setFlipPitch(void* h, int s) { RP; if (s != p->flipPitch) { p->flipPitch = s; orc_rotator_setPitch(h, -orc_rotator_getPitch(h)); } }
void orc_rotator_setFlipRoll(void* h, int s) { RP; if (s != p->flipRoll) { p->flipRoll = s; orc_rotator_setRoll(h, -orc_rotator_getRoll(h)); } }
void orc_rotator_setFlipQuaternion(void* h, int s)
{
    RP;
    if (s != p->flipQ) {
        p->flipQ = s;
        orc_rotator_setQuaternionX(h, -orc_rotator_getQuaternionX(h)); orc_rotator_setQuaternionY(h, -orc_rotator_getQuaternionY(h)); orc_rotator_setQuaternionZ(h, -orc_rotator_getQuaternionZ(h));
    }
}
void orc_rotator_setRPYflag(void* h, int s) { RP; p->useRPY = s; }
void orc_rotator_setChOrder(void* h, int o) { RP; if (o != 2 || p->order == 1) p->chOrdering = o; }
void orc_rotator_setNormType(void* h, int t) { RP; if (t != 3 || p->order == 1) p->norm = t; }
void orc_rotator_setOrder(void* h, int o)
{
    RP;
    p->order = o; p->status = 3;
    if (p->order != 1 && p->chOrdering == 2) p->chOrdering = 1;
    if (p->order != 1 && p->norm == 3) p->norm = 2;
}
int orc_rotator_getChOrder(void* h) { RP; return p->chOrdering; }
int orc_rotator_getOrder(void* h) { RP; return p->order; }
