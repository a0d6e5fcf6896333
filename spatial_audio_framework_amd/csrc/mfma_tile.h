/*
 * mfma_tile.h — the fp32 matrix-core tile shared by the GEMM kernels of libsaf_hip.
 *
 * One 64-lane wave computes a 32-row x 128-column tile of  D = A[32 x 2*KS] * B[2*KS x 128]
 * with v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain; bf16/fp16 MFMA cannot meet the 1e-5 budget).
 *
 * Operand layout of the 32x32x2 instruction:
 *   A: lane l holds A[i = l&31][k = l>>5];   B: lane l holds B[k = l>>5][j = l&31];
 *   C/D: register r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
 * The instruction treats its 32 columns independently, so "column j" may be any memory column:
 * here lane l loads FOUR consecutive floats of its B row with one 16-byte load and feeds them to
 * four accumulators.  Accumulator c, column j is memory column 4*j + c: a wave covers 128
 * contiguous floats (512 B) of each B row and every global access is a full dwordx4.
 */
#pragma once
#include <hip/hip_runtime.h>

namespace saf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Tile128 { f32x16 c[4]; };

__device__ __forceinline__ void tile_zero(Tile128& t)
{
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 16; r++) t.c[c][r] = 0.0f;
}

/* row of accumulator register r for this lane, relative to the 32-row tile */
__device__ __forceinline__ int tile_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

/* One k-pair step: a = this lane's A fragment value for the step, b = this lane's four B values. */
__device__ __forceinline__ void tile_step(Tile128& t, float a, const float4& b)
{
    t.c[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.x, t.c[0], 0, 0, 0);
    t.c[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.y, t.c[1], 0, 0, 0);
    t.c[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.z, t.c[2], 0, 0, 0);
    t.c[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.w, t.c[3], 0, 0, 0);
}

/* 16-byte load of 4 consecutive floats with a column bound (nValid = number of valid floats from p) */
__device__ __forceinline__ float4 load4_bounded(const float* p, int nValid)
{
    if (nValid >= 4) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nValid > 0) v.x = p[0];
    if (nValid > 1) v.y = p[1];
    if (nValid > 2) v.z = p[2];
    return v;
}

__device__ __forceinline__ void store4_bounded(float* p, const float4& v, int nValid)
{
    if (nValid >= 4) { *reinterpret_cast<float4*>(p) = v; return; }
    if (nValid > 0) p[0] = v.x;
    if (nValid > 1) p[1] = v.y;
    if (nValid > 2) p[2] = v.z;
}

}  // namespace saf
