/*
 * afstft_device.h — device helpers shared by the afSTFT kernels (afstft_kernels.hip) and the filterbank equaliser
 * (eq_kernels.hip): LDS slot geometry, the LDS-only barrier, the 128-point FFT of one slot, the real-FFT split.
 * Reference arithmetic: framework/resources/afSTFT/afSTFT_internal.c:237-653, kissFFT/kiss_fftr.c:86-161.
 */
#pragma once
#include "saf_hip_common.h"
#include "fft_butterflies.h"

namespace saf {

#define SUB       16      /* hops per sub-chunk */
#define ARING     22      /* analysis ring: SUB + 6 spectra kept for the hybrid filter */
#define OLA       8       /* hops per overlap-add register window */
#define SLOT      272     /* floats per LDS slot: 256 + 16, slot stride = 16 banks -> the 4 FFTs of a lane group never collide */

#define COEFF1 0.031273141818515176604f   /* afSTFT_internal.h:74 */
#define COEFF2 0.28127313041521179171f    /* afSTFT_internal.h:75 */

/* Global memory through  uniform 64-bit base (scalar registers) + 32-bit byte offset of the lane.  The base is an
 * address-space-1 pointer: a pointer that went through an integer (readfirstlane) is otherwise taken for a FLAT address, and a
 * flat access counts in lgkmcnt as well as vmcnt — every LDS wait behind it would also wait for the HBM round trip. */
typedef const char __attribute__((address_space(1)))* gbase_t;
__device__ __forceinline__ gbase_t uniform_gbase(const void* q)
{
    const unsigned long long b = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return (gbase_t)(((unsigned long long)hi << 32) | lo);
}
/* (re-)pin a uniform base to scalar registers after uniform arithmetic on it: the address selection then takes the
 * "scalar base + 32-bit lane offset" form of global_load / global_store instead of a 64-bit add per lane */
__device__ __forceinline__ gbase_t uniform_gbase(gbase_t q) { return uniform_gbase((const void*)q); }
template <typename T> __device__ __forceinline__ T gld(gbase_t b, unsigned off) { return *(const T __attribute__((address_space(1)))*)(b + off); }
template <typename T> __device__ __forceinline__ void gst(gbase_t b, unsigned off, const T& v) { *(T __attribute__((address_space(1)))*)(b + off) = v; }
/* (float4 is a class: its copy constructor does not take an address-space-1 reference; the 16-byte forms go through the
 * native vector type) */
typedef float saf_v4f __attribute__((ext_vector_type(4)));
template <> __device__ __forceinline__ float4 gld<float4>(gbase_t b, unsigned off)
{
    const saf_v4f v = *(const saf_v4f __attribute__((address_space(1)))*)(b + off);
    return make_float4(v.x, v.y, v.z, v.w);
}
template <> __device__ __forceinline__ void gst<float4>(gbase_t b, unsigned off, const float4& v)
{
    saf_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    *(saf_v4f __attribute__((address_space(1)))*)(b + off) = w;
}

/* Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: every barrier would then wait
 * for the prefetched input loads and for the spectra / sample stores still on their way to HBM. */
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

/* 128-point complex FFT of the sequence stored in one LDS slot (z[n] at floats 2n, 2n+1), in place, by the
 * 8 lanes j = 0..7 of one FFT group (all in one wave: LDS operations of a wave execute in order, so the
 * group needs no barrier).  twJ[p] = exp(-2 pi i j p / 128).  Result Z[k] at floats 2k, 2k+1. */
/* Slot addressing: complex element m of a slot lives at floats 2*(m ^ sg), 2*(m ^ sg) + 1, where sg = SLOT_SG(position of
 * the slot) in 0..7.  The XOR only permutes elements inside groups of 8, so every access of the FFT (lane j <-> element
 * j + 8m) stays conflict-free, while the "column" accesses of the split / pack phases — the same element of 16 consecutive
 * slots, whose bases are only 16 banks apart — are spread over all banks instead of colliding 4- to 8-fold. */
#define SLOT_SG(pos) (((pos) >> 1) & 7)

/* The W128 twiddles of the 8 lanes live in LDS as [p][j] (lane j reads s_tw[p*8 + j]): the 8 lanes of an FFT group read 8
 * consecutive float2 and all groups of a wave read the same ones (broadcast).  Stored [j][p] the 8 lanes hit 8 different
 * rows 128 bytes apart = ONE bank pair: an 8-way conflict on each of the 16 twiddle reads of an FFT, which was more than
 * half of the LDS cycles of the filterbank kernels (tools/probes/lds_probe.hip, profiles/r02_lds_probe.txt). */
struct TwCol {
    const float2* p;
    __device__ __forceinline__ float2 operator[](int i) const { return p[i * 8]; }
};
/* fill s_tw[128] ([p][j]) from the device table ([j][p]); call with tid < 128 */
__device__ __forceinline__ void load_twiddles_pj(float2* s_tw, const float2* g_tw, int tid) { s_tw[(tid & 15) * 8 + (tid >> 4)] = g_tw[tid]; }

template <bool INV, typename TW> __device__ __forceinline__ void fft128_slot(float* slot, int j, const TW& twJ, int sg)
{
    const int js = j ^ sg;
    float2 v[16];
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = *reinterpret_cast<const float2*>(slot + 2 * js + 16 * m);      /* z[j + 8m] */
    dft16<INV>(v);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    /* transpose buffer T[p][j] at floats 16p + 2*(j ^ (p&7)) */
#pragma unroll
    for (int p = 0; p < 16; p++) {
        float2 w = twJ[p]; if (INV) w.y = -w.y;
        const float2 c = p == 0 ? X16(v, 0) : cmul(X16(v, p), w);
        *reinterpret_cast<float2*>(slot + 16 * p + 2 * (j ^ (p & 7))) = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float2 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = *reinterpret_cast<const float2*>(slot + 16 * j + 2 * (i ^ j));
        b[i] = *reinterpret_cast<const float2*>(slot + 16 * (j + 8) + 2 * (i ^ j));
    }
    dft8<INV>(a); dft8<INV>(b);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int q = 0; q < 8; q++) {
        *reinterpret_cast<float2*>(slot + 2 * (js + 16 * q)) = X8(a, q);          /* Z[j + 16q] */
        *reinterpret_cast<float2*>(slot + 2 * (js + 8 + 16 * q)) = X8(b, q);      /* Z[j + 8 + 16q] */
    }
}

/* bins k and 128-k (k = 0..64) of the 256-point real FFT from the packed 128-point spectrum in an LDS slot
 * (kiss_fftr.c:86-123 convention).  k = 0 gives X[0] and X[128] (Z[128] := Z[0], W256^0 = 1). */
__device__ __forceinline__ void ana_bin_pair(const float* slot, int sg, int k, float2 W, float2& Xk, float2& Xm)
{
    const float2 Zk = *reinterpret_cast<const float2*>(slot + 2 * (k ^ sg));
    const float2 Zm = *reinterpret_cast<const float2*>(slot + 2 * (((128 - k) & 127) ^ sg));
    const float2 e = make_float2(Zk.x + Zm.x, Zk.y - Zm.y);
    const float2 d = make_float2(Zk.x - Zm.x, Zk.y + Zm.y);
    const float2 t = cmul(W, d);
    Xk = make_float2(0.5f * (e.x + t.y), 0.5f * (e.y - t.x));
    Xm = make_float2(0.5f * (e.x - t.y), 0.5f * (-e.y - t.x));
}
__device__ __forceinline__ float2 ana_bin_lo(const float* slot, int sg, int k, float2 W)
{
    float2 a, b;
    ana_bin_pair(slot, sg, k, W, a, b);
    return a;
}

}  // namespace saf
