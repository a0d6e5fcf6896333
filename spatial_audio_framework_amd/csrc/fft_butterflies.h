/*
 * fft_butterflies.h — radix-4 / 8 / 16 DFT butterflies in registers, shared by the afSTFT kernels (128-point FFT of the
 * filterbank) and the convolver kernels (power-of-two FFTs of the partitioned convolution).
 * INV = false: forward transform (e^{-i...}); INV = true: inverse (conjugated twiddles, unscaled).
 */
#pragma once
#include <hip/hip_runtime.h>

namespace saf {

#define RSQRT2 0.70710678118654752440f

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
/* multiply by -i (forward) or +i (inverse) */
template <bool INV> __device__ __forceinline__ float2 rot90(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }
/* multiply by a constant twiddle given for the FORWARD transform (conjugated for the inverse) */
template <bool INV> __device__ __forceinline__ float2 twc(float2 a, float wr, float wi) { return cmul(a, make_float2(wr, INV ? -wi : wi)); }

template <bool INV> __device__ __forceinline__ void dft4(float2& x0, float2& x1, float2& x2, float2& x3)
{
    const float2 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = rot90<INV>(csub(x1, x3));
    x0 = cadd(t0, t2); x1 = cadd(t1, t3); x2 = csub(t0, t2); x3 = csub(t1, t3);
}

/* 16-point DFT in registers, radix 4 x 4.  In: v[m].  Out: X[p] is left in v[4*(p&3) + (p>>2)]. */
template <bool INV> __device__ __forceinline__ void dft16(float2 (&v)[16])
{
#pragma unroll
    for (int b = 0; b < 4; b++) dft4<INV>(v[b], v[4 + b], v[8 + b], v[12 + b]);     /* v[4c+b] = y_b[c] */
    /* y_b[c] *= W16^(b*c) */
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;       /* cos, sin(pi/8) */
    v[4 * 1 + 1] = twc<INV>(v[4 * 1 + 1], c1, -s1);                                 /* W16^1 */
    v[4 * 1 + 2] = twc<INV>(v[4 * 1 + 2], RSQRT2, -RSQRT2);                         /* W16^2 */
    v[4 * 1 + 3] = twc<INV>(v[4 * 1 + 3], s1, -c1);                                 /* W16^3 */
    v[4 * 2 + 1] = twc<INV>(v[4 * 2 + 1], RSQRT2, -RSQRT2);                         /* W16^2 */
    v[4 * 2 + 2] = rot90<INV>(v[4 * 2 + 2]);                                        /* W16^4 = -i */
    v[4 * 2 + 3] = twc<INV>(v[4 * 2 + 3], -RSQRT2, -RSQRT2);                        /* W16^6 */
    v[4 * 3 + 1] = twc<INV>(v[4 * 3 + 1], s1, -c1);                                 /* W16^3 */
    v[4 * 3 + 2] = twc<INV>(v[4 * 3 + 2], -RSQRT2, -RSQRT2);                        /* W16^6 */
    v[4 * 3 + 3] = twc<INV>(v[4 * 3 + 3], -c1, s1);                                 /* W16^9 */
#pragma unroll
    for (int c = 0; c < 4; c++) dft4<INV>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);   /* v[4c+d] = X[c+4d] */
}
#define X16(v, p) v[4 * ((p) & 3) + ((p) >> 2)]

/* 8-point DFT in registers.  In: v[j].  Out: X[q] is left in v[2*(q&3) + (q>>2)]. */
template <bool INV> __device__ __forceinline__ void dft8(float2 (&v)[8])
{
    dft4<INV>(v[0], v[2], v[4], v[6]);          /* v[2c]   = y_0[c] */
    dft4<INV>(v[1], v[3], v[5], v[7]);          /* v[2c+1] = y_1[c] */
    v[3] = twc<INV>(v[3], RSQRT2, -RSQRT2);     /* W8^1 */
    v[5] = rot90<INV>(v[5]);                    /* W8^2 */
    v[7] = twc<INV>(v[7], -RSQRT2, -RSQRT2);    /* W8^3 */
#pragma unroll
    for (int c = 0; c < 4; c++) { const float2 u = v[2 * c], w = v[2 * c + 1]; v[2 * c] = cadd(u, w); v[2 * c + 1] = csub(u, w); }
}
#define X8(v, q) v[2 * ((q) & 3) + ((q) >> 2)]

}  // namespace saf
