/*
 * gemm_kernels.hip — the dense real GEMMs of the block path on the fp32 matrix cores (gfx950).
 *
 * band_gemm_kernel replaces the per-band `cblas_cgemm` + `cblas_sscal` of the reference decode
 * step (examples/src/ambi_dec/ambi_dec.c:518-540; the same shape recurs in panner.c:266-274).
 * The decoder matrices are real, so instead of a complex GEMM the interleaved re/im time slots
 * are treated as 2*H real columns:
 *
 *     Y_b [64 x 2H] = A_{mat(b)} [64 x 64] * X_b [64 x 2H]      for every (instance, band b)
 *
 * A already contains the M_norm scale and is zero-padded to 64 x 64, so lower per-band orders
 * and fewer loudspeakers need no special cases.
 *
 * enc_gemm_kernel replaces the `cblas_sgemm` pair, the cross-fade, the 1/sqrt(nSources) scale and
 * the output convention conversions of ambi_enc_process (examples/src/ambi_enc/ambi_enc.c:138-190).
 *
 * Both use the 32 x 128 wave tile of mfma_tile.h: every global access is a 16-byte load/store of
 * a 512-byte contiguous row segment.
 */
#include "saf_hip_common.h"
#include "mfma_tile.h"
#include "afstft_device.h"
#include <algorithm>
#include <climits>
#include <mutex>
#include <utility>

namespace saf {

/* ========================================================================== */
/*                               band GEMM                                    */
/* ========================================================================== */

struct GemmArgs { BandGemmLaunch g; int unitsPerInst, nColTiles, G; int scalarStores; };      /* scalarStores: Y is not 16-byte aligned / strided: four 4-byte stores per lane */

/* Workgroup = 2 waves = the two 32-row halves of 64 x 128 output tiles.  A workgroup owns G consecutive
 * (band, column tile) units of one instance and software-pipelines them: while the matrix cores work on one half
 * of K (16 k-pair steps = 64 MFMAs), the loads of the next half — or of the next unit's first half — are in flight.
 * (Measured without this overlap: load 33 us + MFMA 25 us + store 28 us were additive.) */
/* FULL: every column tile is complete (N % 128 == 0): plain 16-byte loads.  Otherwise masked loads without
 * branches — a load under a branch is followed by its own wait, which would serialise the loads of a tile. */
template <bool FULL>
__device__ __forceinline__ void gemm_load_tile(float4 (&b)[32], const float* X, long long x_row, int nValid, int kh, int nRows)
{
    /* this lane's B row of k-pair step i is 2 i + kh; rows beyond nRows re-read the last present row */
    const long long lastOff = (long long)(nRows - 1) * x_row;
    long long off = (long long)kh * x_row;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const float* p = X + (off < lastOff ? off : lastOff);
        off += 2 * x_row;
        if (FULL) b[i] = *reinterpret_cast<const float4*>(p);
        else {
            /* columns beyond N read column (nValid - 1 clamped) of the same row — valid memory — and are zeroed */
            const int last = nValid > 0 ? nValid - 1 : 0;
            const float* q = nValid > 0 ? p : p + (nValid - 1);          /* lanes wholly outside step back into the row */
            float4 v;
            v.x = q[0 < last ? 0 : last]; v.y = q[1 < last ? 1 : last]; v.z = q[2 < last ? 2 : last]; v.w = q[3 < last ? 3 : last];
            v.x = nValid > 0 ? v.x : 0.0f; v.y = nValid > 1 ? v.y : 0.0f; v.z = nValid > 2 ? v.z : 0.0f; v.w = nValid > 3 ? v.w : 0.0f;
            b[i] = v;
        }
    }
}

/* Workgroup = 2 waves = the two 32-row halves of 64 x 128 output tiles, ONE wave per SIMD (the kernel takes the whole
 * 512-entry register file).  A workgroup owns G consecutive (band, column tile) units of one instance and
 * double-buffers them in registers: while the matrix cores work through the 128 MFMAs of one unit (3.4 us), the 32 KiB
 * of the next unit are in flight.  (Measured without this overlap: load 33 us + MFMA 25 us + store 28 us were additive.) */
template <bool FULL>
__global__ __launch_bounds__(128, 1) void band_gemm_kernel(GemmArgs a)
{
    const BandGemmLaunch& g = a.g;
    if (g.runFlag != nullptr && *g.runFlag == 0) return;      /* fix-up launch behind the fused equaliser + decode kernel: nothing to repair */
    const int inst = blockIdx.y;
    const int lane = threadIdx.x & 63, rt = threadIdx.x >> 6;
    const int kh = lane >> 5;
    const int u0 = blockIdx.x * a.G;
    const int u1 = min(u0 + a.G, a.unitsPerInst);
    if (u0 >= u1) return;
    const float* Xi = g.X + (long long)inst * g.x_inst;
    float* Yi = g.Y + (long long)inst * g.y_inst;
    const float* Ai = g.Afrag + (long long)inst * g.a_inst + (long long)rt * 32 * 64 + lane;
    const int* b2m = g.band2mat + inst * g.nBands;

    float av[32];
    int curMat = -1;
    float4 bA[32], bB[32];
    auto unit_ptr = [&](int u, int& band, int& col) {
        band = u / a.nColTiles; const int ct = u - band * a.nColTiles;
        col = ct * 128 + 4 * (lane & 31);
        return Xi + (long long)band * g.x_band + col;
    };
    auto compute_store = [&](const float4 (&b)[32], int band, int col) {
        const int mat = b2m[band];
        if (mat != curMat) {                      /* rare: the decoder / order of the band changed */
#pragma unroll
            for (int s = 0; s < 32; s++) av[s] = Ai[(long long)mat * 2 * 32 * 64 + s * 64];
            curMat = mat;
        }
        Tile128 t;
        tile_zero(t);
#pragma unroll
        for (int i = 0; i < 32; i++) tile_step(t, av[i], b[i]);
        const int nValid = g.N - col;
        if (nValid > 0) {
            float* Y = Yi + (long long)band * g.y_band + col;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = rt * 32 + tile_row(r, lane);
                const float4 v = make_float4(t.c[0][r], t.c[1][r], t.c[2][r], t.c[3][r]);
                if (row >= g.nRowsY) continue;
                if (FULL) *reinterpret_cast<float4*>(Y + (long long)row * g.y_row) = v;
                else if (a.scalarStores) {
                    float* q = Y + (long long)row * g.y_row;
                    q[0] = v.x; if (nValid > 1) q[1] = v.y; if (nValid > 2) q[2] = v.z; if (nValid > 3) q[3] = v.w;
                } else store4_bounded(Y + (long long)row * g.y_row, v, nValid);
            }
        }
    };
    int bandA, colA, bandB = 0, colB = 0;
    const float* pA = unit_ptr(u0, bandA, colA);
    gemm_load_tile<FULL>(bA, pA, g.x_row, g.N - colA, kh, g.nRowsX);
    for (int u = u0; u < u1; u += 2) {
        if (u + 1 < u1) { const float* pB = unit_ptr(u + 1, bandB, colB); gemm_load_tile<FULL>(bB, pB, g.x_row, g.N - colB, kh, g.nRowsX); }
        compute_store(bA, bandA, colA);
        if (u + 1 < u1) {
            if (u + 2 < u1) { pA = unit_ptr(u + 2, bandA, colA); gemm_load_tile<FULL>(bA, pA, g.x_row, g.N - colA, kh, g.nRowsX); }
            compute_store(bB, bandB, colB);
        }
    }
}

/* Two-term variant:  Y = A_0 X_0 + A_1 X_1  (the time-domain decode after the filterbank equaliser when the two decoders
 * of ambi_dec are different matrices: out = M_0 z_0 + M_1 z_1).  Same tile and register budget; the two operand tiles of a
 * unit take the places of the two units in flight of the one-term kernel: term 1 is in flight under the MFMAs of term 0,
 * the next unit's term 0 under those of term 1. */
__global__ __launch_bounds__(128, 1) void band_gemm2_kernel(GemmArgs a)
{
    constexpr bool FULL = true;          /* complete column tiles only (N % 128 == 0) */
    const BandGemmLaunch& g = a.g;
    if (g.runFlag != nullptr && *g.runFlag == 0) return;
    const int inst = blockIdx.y;
    const int lane = threadIdx.x & 63, rt = threadIdx.x >> 6;
    const int kh = lane >> 5;
    const int u0 = blockIdx.x * a.G;
    const int u1 = min(u0 + a.G, a.unitsPerInst);
    if (u0 >= u1) return;
    const float* Xi = g.X + (long long)inst * g.x_inst;
    float* Yi = g.Y + (long long)inst * g.y_inst;
    const float* Ai = g.Afrag + (long long)inst * g.a_inst + (long long)rt * 32 * 64 + lane;
    const int* b2m = g.band2mat + inst * g.nBands;

    float av0[32], av1[32];
    int curMat = -1;
    float4 bA[32], bB[32];
    auto unit_ptr = [&](int u, int& band, int& col) {
        band = u / a.nColTiles; const int ct = u - band * a.nColTiles;
        col = ct * 128 + 4 * (lane & 31);
        return Xi + (long long)band * g.x_band + col;
    };
    int band, col;
    const float* p = unit_ptr(u0, band, col);
    gemm_load_tile<FULL>(bA, p, g.x_row, g.N - col, kh, g.nRowsX);
    gemm_load_tile<FULL>(bB, p + g.x_term, g.x_row, g.N - col, kh, g.nRowsX);
    for (int u = u0; u < u1; u++) {
        const int mat = b2m[band];
        if (mat != curMat) {
#pragma unroll
            for (int s = 0; s < 32; s++) { av0[s] = Ai[(long long)mat * 2 * 32 * 64 + s * 64]; av1[s] = Ai[(long long)(mat + 1) * 2 * 32 * 64 + s * 64]; }
            curMat = mat;
        }
        Tile128 t;
        tile_zero(t);
#pragma unroll
        for (int i = 0; i < 32; i++) tile_step(t, av0[i], bA[i]);
        int bandN = band, colN = col;
        const float* pN = p;
        if (u + 1 < u1) { pN = unit_ptr(u + 1, bandN, colN); gemm_load_tile<FULL>(bA, pN, g.x_row, g.N - colN, kh, g.nRowsX); }
#pragma unroll
        for (int i = 0; i < 32; i++) tile_step(t, av1[i], bB[i]);
        if (u + 1 < u1) gemm_load_tile<FULL>(bB, pN + g.x_term, g.x_row, g.N - colN, kh, g.nRowsX);
        const int nValid = g.N - col;
        if (nValid > 0) {
            float* Y = Yi + (long long)band * g.y_band + col;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = rt * 32 + tile_row(r, lane);
                const float4 v = make_float4(t.c[0][r], t.c[1][r], t.c[2][r], t.c[3][r]);
                if (row >= g.nRowsY) continue;
                if (FULL) *reinterpret_cast<float4*>(Y + (long long)row * g.y_row) = v;
                else store4_bounded(Y + (long long)row * g.y_row, v, nValid);
            }
        }
        band = bandN; col = colN; p = pN;
    }
}

void launch_band_gemm(const BandGemmLaunch& g)
{
    if (g.N <= 0 || g.nInst <= 0) return;
    if ((g.x_row | g.x_band | g.x_inst) & 3) SAF_FATAL("band gemm: the strides of X must be multiples of 4 floats");
    if (((uintptr_t)g.X) & 15) SAF_FATAL("band gemm: X must be 16-byte aligned");
    /* an output that is not 16-byte aligned or whose strides are not multiples of 4 floats (a caller's block at an odd offset)
     * is written with 4-byte stores by the bounded variant of the one-term kernel */
    const bool yOdd = ((g.y_row | g.y_band | g.y_inst) & 3) != 0 || (((uintptr_t)g.Y) & 15) != 0;
    if (yOdd && g.nTerms != 1) SAF_FATAL("band gemm: the two-term form needs a 16-byte aligned output with strides that are multiples of 4 floats");
    GemmArgs a;
    a.g = g;
    a.scalarStores = yOdd ? 1 : 0;
    a.nColTiles = (g.N + 127) / 128;
    a.unitsPerInst = g.nBands * a.nColTiles;
    /* as many units per workgroup as it takes for the whole launch to be resident at once (1 wave per SIMD on
     * 1024 SIMDs, 2 waves per workgroup): no second round, and the pipeline prologue is amortised over G units */
    const long long units = (long long)a.unitsPerInst * g.nInst;
    int G = (int)((units + 511) / 512);
    if (G < 1) G = 1;
    a.G = G;
    dim3 grid((a.unitsPerInst + G - 1) / G, g.nInst);
    KernelTimer kt("band_gemm");
    if (g.nTerms == 2) {
        if (g.x_term & 3) SAF_FATAL("band gemm: the term stride must be a multiple of 4 floats");
        if (g.N % 128 != 0) SAF_FATAL("band gemm: the two-term form needs N to be a multiple of 128");
        hipLaunchKernelGGL(band_gemm2_kernel, grid, dim3(128), 0, stream(), a);
    } else if (g.nTerms != 1) SAF_FATAL("band gemm: 1 or 2 terms");
    else if (g.N % 128 == 0 && !yOdd) hipLaunchKernelGGL(band_gemm_kernel<true>, grid, dim3(128), 0, stream(), a);
    else                     hipLaunchKernelGGL(band_gemm_kernel<false>, grid, dim3(128), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

/* ========================================================================== */
/*        the time-domain decode beside the filterbank equaliser kernel       */
/* ========================================================================== */
/* The reference decodes the frame it has just transformed while it is in cache (ambi_dec.c:514-566).  Here the equaliser
 * kernel (eq_kernels.hip) is bound by vector issue and LDS, the decode  out = sum_d M_d z_d  by HBM and the matrix cores: run
 * one after the other each leaves half the chip idle.  This kernel runs BESIDE the equaliser kernel, on the library's side
 * stream: a persistent grid of P workgroups (about one per compute unit; a workgroup = 2 waves with at most 184 registers,
 * so that it fits next to two of the equaliser's 160-register waves on a SIMD and takes one of a CU's six workgroup slots).
 * Workgroup p takes the items p, p + P, ... — item = G consecutive units of one instance, instances in the order in which the
 * equaliser kernel's workgroups are dispatched — and before an item it waits for the instance: every equaliser workgroup
 * publishes its z (stores drained, barrier, agent-scope release) and adds 1 to done[inst]; here one lane polls the counter,
 * then agent-scope acquire, barrier, plain loads (MI355X_MICROARCH.md "Valid forms").  Nothing depends on dispatch order or
 * placement: the equaliser workgroups never wait, and at most P slots of the chip are held by waiting workgroups.  A poll
 * that is not answered within ~2 s (the equaliser kernel was never launched) sets err[0] and the workgroup leaves; the
 * caller's fix-up launch (launch_band_gemm guarded by err[0]) then computes the output.
 *
 * Unit = 128 columns of one block (frame) of one instance; wave = 32 rows x 128 columns on v_mfma_f32_32x32x2_f32 (exact
 * fp32).  Register-lean on purpose: 64 accumulators + a ring of DEC_RING 16-byte operand loads per lane that is refilled as
 * it is consumed (band_gemm_kernel double-buffers whole units in 256 registers and owns its SIMD); the matrix sits in LDS in
 * fragment order (one conflict-free ds_read_b32 per k-pair step). */
#ifndef DEC_RING
#define DEC_RING 16             /* operand loads (16 bytes per lane each) in flight per wave */
#endif
#ifndef DEC_UNROLL
#define DEC_UNROLL 4            /* units per trip of the unit loop: inside the straight-line body the compiler counts the loads in
                                 * flight exactly (s_waitcnt vmcnt(DEC_RING - 1) per step), at the top of a trip it drains them all */
#endif
#ifndef DEC_VGPRS
#define DEC_VGPRS 184
#endif
struct DecArgs { DecStreamLaunch l; int nColTiles, unitsPerInst, G, nG, nItems; };

template <int D>
__global__ __launch_bounds__(128) __attribute__((amdgpu_num_vgpr(DEC_VGPRS))) void dec_stream_kernel(DecArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_A[D * 2 * 32 * 64];
    __shared__ int s_ok;
    const DecStreamLaunch& l = a.l;
    const int tid = threadIdx.x, lane = tid & 63, rt = tid >> 6, kh = lane >> 5;
    if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(l.err, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* (a workgroup gives up 2 s later at the earliest) */
    /* Addressing: every load and store is  uniform 64-bit base (scalar registers) + 32-bit byte offset of the lane.  Step s of a
     * unit reads rows 2 s + kh: the step part goes into the scalar base, the lane keeps (kh, column).  Steps beyond the last
     * channel pair re-read that pair (their matrix columns are zero).  With an odd channel count the kh = 1 lanes of the last
     * pair read row nCh, one beyond the channels: dec_stream_supported requires that row to exist in z (the operators allocate
     * 64 zero-filled rows per instance); its matrix column is zero.  An instance's z and output blocks span less than 4 GiB. */
    const int sLast = (l.nCh - 1) >> 1;
    const unsigned laneA = (unsigned)(kh * l.z_ch + 4 * (lane & 31)) * 4u;
    const unsigned laneY = (unsigned)((rt * 32 + 4 * kh) * l.y_row + 4 * (lane & 31)) * 4u;
    const unsigned rowPair = (unsigned)(2 * l.z_ch * 4);     /* bytes between the row pairs of consecutive steps */
    const long long termB = l.z_d * 4;
    typedef const float __attribute__((address_space(3)))* lds_cf;
    const lds_cf As = (lds_cf)s_A + rt * 2048 + lane;
    constexpr int NS = 32 * D, RING = DEC_RING, UU = DEC_UNROLL;
    static_assert(NS % RING == 0, "the ring position of a step must not depend on the unit");

    for (int w = blockIdx.x; w < a.nItems; w += gridDim.x) {
        const int inst = w / a.nG, gi = w - inst * a.nG;
        __syncthreads();                                        /* the previous item's reads of s_A */
        {
            const float4* Ag = reinterpret_cast<const float4*>(l.Mfrag + (long long)inst * l.m_inst);
#pragma unroll
            for (int i = 0; i < 8 * D; i++) reinterpret_cast<float4*>(s_A)[tid + 128 * i] = Ag[tid + 128 * i];
        }
        if (tid == 0) {
            int ok = 0;
            for (int it = 0; it < 1000000; it++) {              /* bounded: ~2 s of 1.7 us naps */
                const unsigned v = __hip_atomic_load(l.done + inst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(v - l.target) >= 0) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(64);
            }
            if (ok) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                __hip_atomic_store(l.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(l.err + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) return;

        const int u0 = gi * a.G, u1 = min(u0 + a.G, a.unitsPerInst);
        const gbase_t Zi = uniform_gbase(l.z + (long long)inst * l.z_inst);
        const gbase_t Yi = uniform_gbase(l.Y + (long long)inst * l.y_inst);
        auto unit_base = [&](int u) { const int fr = u / a.nColTiles, ct = u - fr * a.nColTiles; return uniform_gbase(Zi + ((long long)fr * l.F + ct * 128) * 4); };
        /* The operand loads form one stream over (unit, k-pair step), RING steps ahead of the MFMAs: `lp` is the scalar base of
         * the stream's next load and moves by one row pair per step (steps beyond the last channel pair stay on it: their matrix
         * columns are zero), to the second term's rows after step 31 and to the next unit after the last step.  (A table of the
         * 32 row offsets would be loop-invariant: hoisted, it costs 64 scalar registers and pushes the address arithmetic into the
         * vector unit.) */
        gbase_t ubL = unit_base(u0), lp = ubL;
        auto next_load = [&](int j, int uAfter) {               /* j: step of this load within its unit (compile-time constant) */
            const float4 v = gld<float4>(lp, laneA);
            if (j == NS - 1) { ubL = unit_base(uAfter); lp = ubL; }
            else if (D > 1 && j == 31) lp = ubL + termB;
            else lp += (j & 31) < sLast ? rowPair : 0u;
            return v;
        };
        float4 ring[RING];
#pragma unroll
        for (int s = 0; s < RING; s++) ring[s] = next_load(s, u0);
        for (int ub = u0; ub < u1; ub += UU) {
#pragma unroll
            for (int uu = 0; uu < UU; uu++) {
                const int u = min(ub + uu, u1 - 1);             /* (a short last trip repeats its last unit: same values, same places) */
                const int uN = min(ub + uu + 1, u1 - 1), uNN = min(ub + uu + 2, u1 - 1);      /* (the last unit re-requests its own first steps: no load under a branch) */
                Tile128 t;
                tile_zero(t);
                float aNext = As[0];
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const float av = aNext;
                    if (s + 1 < NS) aNext = As[((s + 1) >> 5) * 4096 + ((s + 1) & 31) * 64];      /* [term][row tile][step][lane] */
                    tile_step(t, av, ring[s % RING]);
                    /* the refill goes where it is written, behind the MFMAs that read the slot (so that it takes the slot's
                     * registers): left alone the scheduler either hoists it above them (a second set of registers) or sinks it to
                     * just before its use (one load in flight instead of DEC_RING) */
                    __builtin_amdgcn_sched_barrier(0);
                    const int sn = s + RING;                    /* the stream is RING steps ahead: step sn of this unit, or sn - NS of the next */
                    ring[s % RING] = next_load(sn % NS, sn < NS ? uN : uNN);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int fr = u / a.nColTiles, ct = u - fr * a.nColTiles;
                const gbase_t Y = uniform_gbase(Yi + ((long long)fr * l.y_frame + ct * 128) * 4);
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int rq = (q & 3) + 8 * (q >> 2);      /* row = rt * 32 + 4 * kh + rq */
                    gst<float4>(Y + (long long)rq * l.y_row * 4, laneY, make_float4(t.c[0][q], t.c[1][q], t.c[2][q], t.c[3][q]));
                    __builtin_amdgcn_sched_barrier(0);          /* (one address at a time) */
                }
            }
        }
    }
}

bool dec_stream_supported(const DecStreamLaunch& l)
{
    /* the tile stores 64 rows unguarded and moves 128 columns with 16-byte accesses; with an odd channel count the row behind the
     * last channel must exist (see the kernel) */
    if (l.nRowsY != 64 || l.F % 128 != 0 || l.nFrames <= 0 || l.nInst <= 0 || (l.D != 1 && l.D != 2)) return false;
    if (((l.y_inst | l.y_frame | l.y_row | l.z_inst | l.z_ch | l.z_d | l.m_inst) & 3) || ((uintptr_t)l.Y & 15) || ((uintptr_t)l.z & 15) || ((uintptr_t)l.Mfrag & 15)) return false;
    if (l.z_ch < 0 || l.y_row < 0 || l.y_frame < 0) return false;
    const long long rows = l.nCh + (l.nCh & 1);
    if (rows * l.z_ch > l.z_inst && l.nInst > 1) return false;
    if ((64 * l.z_ch + (long long)l.nFrames * l.F) * 4 >= (1ll << 32)) return false;
    if ((64 * l.y_row + (long long)l.nFrames * l.y_frame + l.F) * 4 >= (1ll << 32)) return false;
    return true;
}

static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static int g_dec_P = env_int("SAF_HIP_DEC_WGS", 0), g_dec_G = env_int("SAF_HIP_DEC_G", 0);

void launch_dec_stream(const DecStreamLaunch& l, hipStream_t s)
{
    if (!dec_stream_supported(l)) SAF_FATAL("launch_dec_stream: unsupported shape (check dec_stream_supported first)");
    DecArgs a;
    a.l = l;
    a.nColTiles = l.F / 128;
    a.unitsPerInst = l.nFrames * a.nColTiles;
    /* G units per item: long enough to amortise the matrix load and the acquire (a unit is ~5 us), short enough that the
     * workgroups stay close behind the equaliser kernel */
    a.G = g_dec_G > 0 ? g_dec_G : 16;
    if (a.G > a.unitsPerInst) a.G = a.unitsPerInst;
    a.nG = (a.unitsPerInst + a.G - 1) / a.G;
    a.nItems = l.nInst * a.nG;
    int P = g_dec_P > 0 ? g_dec_P : 256;                         /* one per compute unit */
    if (P > a.nItems) P = a.nItems;
    KernelTimer kt("dec_stream", s);
    if (l.D == 1) hipLaunchKernelGGL(dec_stream_kernel<1>, dim3(P), dim3(128), 0, s, a);
    else          hipLaunchKernelGGL(dec_stream_kernel<2>, dim3(P), dim3(128), 0, s, a);
    HIP_CHECK(hipGetLastError());
}

/* A [64][64] row-major -> MFMA fragment order [rt][s][lane]: lane l of step s holds A[rt*32 + (l&31)][2s + (l>>5)] */
void pack_A(const float* A, float* Afrag)
{
    for (int rt = 0; rt < 2; rt++)
        for (int s = 0; s < 32; s++)
            for (int l = 0; l < 64; l++)
                Afrag[(rt * 32 + s) * 64 + l] = A[(rt * 32 + (l & 31)) * 64 + 2 * s + (l >> 5)];
}

__global__ void pack_A_kernel(const float* A, float* Afrag, int nMat)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nMat * 4096) return;
    const int m = idx >> 12, e = idx & 4095;
    const int rt = e >> 11, s = (e >> 6) & 31, l = e & 63;
    Afrag[idx] = A[(long long)m * 4096 + (rt * 32 + (l & 31)) * 64 + 2 * s + (l >> 5)];
}

void launch_pack_A(const float* d_A, float* d_Afrag, int nMat)
{
    if (nMat <= 0) return;
    hipLaunchKernelGGL(pack_A_kernel, dim3((nMat * 4096 + 255) / 256), dim3(256), 0, stream(), d_A, d_Afrag, nMat);
    HIP_CHECK(hipGetLastError());
}

/* ========================================================================== */
/*                          ambi_enc encode GEMM                              */
/* ========================================================================== */

struct EncArgs { EncLaunch e; int frameBase; };

/* one block of one instance, columns [128 blockIdx.x, +128); block index nFrames only saves the last
 * input frame (after gains) as the next call's "previous frame" (ambi_enc.c:165). */
template <bool CANMIX, bool SMALL>
__device__ __forceinline__ void enc_frame(const EncLaunch& e, int frame, int inst)
{
#pragma clang fp contract(off)      /* the reference scales / fades in separate BLAS + veclib calls: keep the roundings */
    const int lane = threadIdx.x & 63, rt = threadIdx.x >> 6;
    const int nSrc = e.nSrc[inst];
    const float* gains = e.gains + inst * SAF_MAXCH;

    if (frame == e.nFrames) {
        /* prev_inputFrameTD <- gained input of the last frame; rows beyond the present sources are zero */
        const float* src = e.in + (long long)inst * e.in_inst + (long long)(e.nFrames - 1) * e.in_frame;
        float* dst = e.prev_wr + (long long)inst * SAF_MAXCH * e.F;
        const int c4 = blockIdx.x * 32 + (threadIdx.x & 31);        /* float4 column of this thread */
        if (c4 * 4 < e.F)
            for (int row = threadIdx.x >> 5; row < SAF_MAXCH; row += 4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < nSrc) {
                    v = load4_bounded(src + (long long)row * e.in_ch + c4 * 4, e.F - c4 * 4);
                    const float gq = gains[row];
                    v.x *= gq; v.y *= gq; v.z *= gq; v.w *= gq;
                }
                store4_bounded(dst + (long long)row * e.F + c4 * 4, v, e.F - c4 * 4);
            }
        return;
    }

    const int col = blockIdx.x * 128 + 4 * (lane & 31);
    const int nValid = e.F - col;
    const int kh = lane >> 5;
    const bool fromState = frame == 0;
    const float* X = fromState ? e.prev_rd + (long long)inst * SAF_MAXCH * e.F
                               : e.in + (long long)inst * e.in_inst + (long long)(frame - 1) * e.in_frame;
    const long long xrow = fromState ? e.F : e.in_ch;
    const bool mix = CANMIX && fromState && e.mix != nullptr && e.mix[inst] != 0;
    const float* A = e.Afrag + ((long long)inst * 2 * 2 + rt) * 32 * 64 + lane;          /* Y */
    const float* Ap = e.AfragPrev ? e.AfragPrev + ((long long)inst * 2 * 2 + rt) * 32 * 64 + lane : A + 2 * 32 * 64;     /* prev_Y */

    /* SMALL (few sources): the output row tables are fetched before the product so that their latency hides under it,
     * and since rows >= nSH of Y and columns >= nSrc are zero, a row tile without SH rows and the k-steps past the last
     * source — which only add zeros — are skipped (4 sources at first order need 2 of the 32 steps of one tile). */
    float rsv[16]; int orv[16];
    if (SMALL) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = rt * 32 + tile_row(r, lane);
            rsv[r] = e.rowScale[inst * SAF_MAXCH + row]; orv[r] = e.rowMap[inst * SAF_MAXCH + row];
        }
    }
    Tile128 t, tp;
    tile_zero(t);
    if (mix) tile_zero(tp);
#define ENC_STEP(s)                                                                                          \
    {                                                                                                        \
        const int k = 2 * (s) + kh;                                                                          \
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
        if (k < nSrc) {                                                                                      \
            b = load4_bounded(X + (long long)k * xrow + col, nValid);                                        \
            if (!fromState) { const float gq = gains[k]; b.x *= gq; b.y *= gq; b.z *= gq; b.w *= gq; }       \
        }                                                                                                    \
        tile_step(t, A[(s) * 64], b);                                                                        \
        if (mix) tile_step(tp, Ap[(s) * 64], b);                                                             \
    }
    if (SMALL) {
        const int nSH = (e.order[inst] + 1) * (e.order[inst] + 1);
        const int nSteps = rt * 32 < nSH ? (nSrc + 1) >> 1 : 0;
        for (int s = 0; s < nSteps; s++) ENC_STEP(s)
    } else {
#pragma unroll 8
        for (int s = 0; s < 32; s++) ENC_STEP(s)
    }
#undef ENC_STEP

    if (nValid <= 0) return;
    const float post = e.postScale[inst];
    float fin[4], fout[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {                       /* interpolators of ambi_enc_init (ambi_enc.c:76-79) */
        fin[c] = (float)(col + c + 1) * 1.0f / (float)e.F;
        fout[c] = 1.0f - fin[c];
    }
    float* O = e.out + (long long)inst * e.out_inst + (long long)frame * e.out_frame + col;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = rt * 32 + tile_row(r, lane);
        const int orow = SMALL ? orv[r] : e.rowMap[inst * SAF_MAXCH + row];
        if (orow < 0 || orow >= e.nOut) continue;
        float v[4] = { t.c[0][r], t.c[1][r], t.c[2][r], t.c[3][r] };
        if (mix) {
            const float w[4] = { tp.c[0][r], tp.c[1][r], tp.c[2][r], tp.c[3][r] };
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = fin[c] * v[c] + fout[c] * w[c];
        }
        const float rs = SMALL ? rsv[r] : e.rowScale[inst * SAF_MAXCH + row];
#pragma unroll
        for (int c = 0; c < 4; c++) { v[c] = v[c] * post; v[c] = v[c] * rs; }
        store4_bounded(O + (long long)orow * e.out_ch, make_float4(v[0], v[1], v[2], v[3]), nValid);
    }
}

/* grid (column tiles of 128 samples, nFrames + 1 blocks, nInst) */
template <bool CANMIX, bool SMALL>
__global__ __launch_bounds__(128) void enc_gemm_kernel(EncArgs a)
{
    enc_frame<CANMIX, SMALL>(a.e, a.frameBase + blockIdx.y, blockIdx.z);
}

/* Blocks that do not cross-fade, full-size scenes: the product is software-pipelined in groups of 4 k-pair steps (the
 * loads of group g+1 are in flight under the 16 MFMAs of group g).  Every sample load is unconditional — rows beyond
 * the present sources re-read the last present row and are zeroed by a select, columns beyond F re-read column 0 —
 * because a load under a branch gets its own wait (the first version of this loop ran load -> wait -> 4 MFMAs 32 times
 * in a row).  The Y fragments, the per-source gains and the output row tables sit in LDS: their reads do not share the
 * samples' vmcnt queue.  128 registers = 4 waves per SIMD. */
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void enc_gemm_full_kernel(EncArgs a)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float s_A[2 * 32 * 64];
    __shared__ float s_g[SAF_MAXCH], s_rs[SAF_MAXCH];
    __shared__ int s_map[SAF_MAXCH];
    const EncLaunch& e = a.e;
    const int frame = a.frameBase + blockIdx.y, inst = blockIdx.z;
    if (frame == e.nFrames) { enc_frame<false, false>(e, frame, inst); return; }      /* the "previous frame" copy */
    const int tid = threadIdx.x, lane = tid & 63, rt = tid >> 6, kh = lane >> 5;
    const int nSrc = e.nSrc[inst];
    const bool fromState = frame == 0;
    const float* X = fromState ? e.prev_rd + (long long)inst * SAF_MAXCH * e.F
                               : e.in + (long long)inst * e.in_inst + (long long)(frame - 1) * e.in_frame;
    const long long xrow = fromState ? e.F : e.in_ch;
    const int col = blockIdx.x * 128 + 4 * (lane & 31);
    const bool colOn = col < e.F;
    const float* Xc = X + (colOn ? col : 0);
    const long long lastOff = (long long)(nSrc > 0 ? nSrc - 1 : 0) * xrow;
    if (nSrc <= 0) Xc = e.prev_rd;                      /* nothing to read: any valid address, every value is masked */

    float4 b[2][4];
    long long off = (long long)kh * xrow;
#pragma unroll
    for (int i = 0; i < 4; i++) { b[0][i] = *reinterpret_cast<const float4*>(Xc + (off < lastOff ? off : lastOff)); off += 2 * xrow; }
    {
        const float4* Ag = reinterpret_cast<const float4*>(e.Afrag + (long long)inst * 2 * 2 * 32 * 64);
#pragma unroll
        for (int i = 0; i < 8; i++) reinterpret_cast<float4*>(s_A)[tid + 128 * i] = Ag[tid + 128 * i];
        if (tid < SAF_MAXCH) {
            s_g[tid] = tid < nSrc ? (fromState ? 1.0f : e.gains[inst * SAF_MAXCH + tid]) : 0.0f;
            s_rs[tid] = e.rowScale[inst * SAF_MAXCH + tid];
            s_map[tid] = e.rowMap[inst * SAF_MAXCH + tid];
        }
    }
    const float post = e.postScale[inst];
    __syncthreads();

    Tile128 t;
    tile_zero(t);
    const float* Aw = s_A + rt * 32 * 64 + lane;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        if (g + 1 < 8) {
#pragma unroll
            for (int i = 0; i < 4; i++) { b[(g + 1) & 1][i] = *reinterpret_cast<const float4*>(Xc + (off < lastOff ? off : lastOff)); off += 2 * xrow; }
        }
        __builtin_amdgcn_sched_barrier(0);          /* keep the prefetch ahead of the group's MFMAs (the scheduler sinks it otherwise) */
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int s = 4 * g + i, k = 2 * s + kh;
            const float gq = s_g[k];
            const bool on = k < nSrc;
            float4 v = b[g & 1][i];
            v.x = on ? v.x * gq : 0.0f; v.y = on ? v.y * gq : 0.0f; v.z = on ? v.z * gq : 0.0f; v.w = on ? v.w * gq : 0.0f;
            tile_step(t, Aw[s * 64], v);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!colOn) return;
    float* O = e.out + (long long)inst * e.out_inst + (long long)frame * e.out_frame + col;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = rt * 32 + tile_row(r, lane);
        const int orow = s_map[row];
        if (orow < 0 || orow >= e.nOut) continue;
        const float rs = s_rs[row];
        float4 v = make_float4(t.c[0][r], t.c[1][r], t.c[2][r], t.c[3][r]);
        v.x = v.x * post; v.x = v.x * rs; v.y = v.y * post; v.y = v.y * rs; v.z = v.z * post; v.z = v.z * rs; v.w = v.w * post; v.w = v.w * rs;
        *reinterpret_cast<float4*>(O + (long long)orow * e.out_ch) = v;
    }
}

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void enc_gemm_small_kernel(EncArgs a)
{
    enc_frame<false, true>(a.e, a.frameBase + blockIdx.y, blockIdx.z);
}

/* Do two sets of F-float segments  base + i0 s0 + i1 s1 + i2 s2  (0 <= ik < nk, any sign of the strides) share a float?
 * 1. extents (anchored at the right end for negative strides) apart: no.  2. the same strides on both sides: the sets are
 * translates of one lattice, so they meet iff the distance of the bases is within F of a lattice difference — (2 n0 - 1)(2 n1 - 1)
 * candidates, the third index solved for.  3. otherwise: sort the segments of both sets by start and sweep (the verdict for the
 * last shape seen is kept: a caller repeats its layout).  Exact in all three. */
struct StridedSet { const float* base; long long s[3]; int n[3]; int F; };
static bool strided_sets_overlap(const StridedSet& a, const StridedSet& b)
{
    auto lo_hi = [](const StridedSet& x, long long& lo, long long& hi) {
        lo = 0; hi = x.F;
        for (int k = 0; k < 3; k++) { const long long ext = (long long)(x.n[k] - 1) * x.s[k]; if (ext < 0) lo += ext; else hi += ext; }
    };
    long long alo, ahi, blo, bhi;
    lo_hi(a, alo, ahi); lo_hi(b, blo, bhi);
    const long long d = b.base - a.base;                                  /* floats */
    if (d + bhi <= alo || ahi <= d + blo) return false;
    if (a.s[0] == b.s[0] && a.s[1] == b.s[1] && a.s[2] == b.s[2]) {
        /* a-segment (i) meets b-segment (j) iff |d + (j - i) . s| < F */
        for (long long d0 = -(a.n[0] - 1); d0 <= b.n[0] - 1; d0++)
            for (long long d1 = -(a.n[1] - 1); d1 <= b.n[1] - 1; d1++) {
                const long long r = d + d0 * a.s[0] + d1 * a.s[1];
                if (a.s[2] == 0) { if (r > -a.F && r < a.F) return true; continue; }
                const long long q = -r / a.s[2];
                for (long long d2 = q - 1; d2 <= q + 1; d2++) {
                    if (d2 < -(a.n[2] - 1) || d2 > b.n[2] - 1) continue;
                    const long long v = r + d2 * a.s[2];
                    if (v > -a.F && v < a.F) return true;
                }
            }
        return false;
    }
    static std::mutex mtx;
    static StridedSet lastA{}, lastB{}; static bool lastV = false, have = false;
    std::lock_guard<std::mutex> lk(mtx);
    if (have && !memcmp(&lastA, &a, sizeof(a)) && !memcmp(&lastB, &b, sizeof(b))) return lastV;
    std::vector<std::pair<long long, int>> seg;
    seg.reserve((size_t)a.n[0] * a.n[1] * a.n[2] + (size_t)b.n[0] * b.n[1] * b.n[2]);
    for (int w = 0; w < 2; w++) {
        const StridedSet& x = w ? b : a;
        for (int i = 0; i < x.n[0]; i++) for (int j = 0; j < x.n[1]; j++) for (int k = 0; k < x.n[2]; k++)
            seg.emplace_back((w ? d : 0) + i * x.s[0] + j * x.s[1] + k * x.s[2], w);
    }
    std::sort(seg.begin(), seg.end());
    bool v = false;
    long long endOf[2] = { LLONG_MIN, LLONG_MIN };                        /* furthest end seen so far, per set */
    for (const auto& sg : seg) {
        if (sg.first < endOf[sg.second ^ 1]) { v = true; break; }
        if (sg.first + a.F > endOf[sg.second]) endOf[sg.second] = sg.first + a.F;
    }
    lastA = a; lastB = b; lastV = v; have = true;
    return v;
}

void launch_enc_gemm(const EncLaunch& e)
{
    if (e.nFrames <= 0 || e.nInst <= 0) return;
    if ((e.in_inst | e.in_frame | e.in_ch | e.out_inst | e.out_frame | e.out_ch | e.F) & 3) SAF_FATAL("ambi_enc: strides and block size must be multiples of 4 floats");
    if ((((uintptr_t)e.in) | ((uintptr_t)e.out)) & 15) SAF_FATAL("ambi_enc: sample buffers must be 16-byte aligned");
    {
        /* in-place use is not possible: within one launch the workgroup of block f writes output block f while the workgroup
         * of block f + 1 (and the state-save block) reads input block f.  What is forbidden is an input segment [p, p + F) and an
         * output segment sharing a float — not merely overlapping extents: in = t[:, 0], out = t[:, 1] of one [blocks][2][ch][F]
         * tensor, or an [inst][chIn + chOut][F] workspace, are disjoint although their extents interleave. */
        const int rowsIn = e.rowsIn > 0 && e.rowsIn < SAF_MAXCH ? e.rowsIn : SAF_MAXCH;       /* rows the kernels can touch (higher rows re-read the last one) */
        const int rowsOut = e.nOut < SAF_MAXCH ? e.nOut : SAF_MAXCH;
        const StridedSet si{ e.in, { e.in_inst, e.in_frame, e.in_ch }, { e.nInst, e.nFrames, rowsIn }, e.F };
        const StridedSet so{ e.out, { e.out_inst, e.out_frame, e.out_ch }, { e.nInst, e.nFrames, rowsOut }, e.F };
        if (strided_sets_overlap(si, so))
            SAF_FATAL("encode GEMM (ambi_enc / rotator / beamformer *_process_dev, batch_process): an input block and an output block share memory; the device entry points do not work in place");
    }
    EncArgs a;
    a.e = e;
    KernelTimer kt("sh_encode");
    /* only block 0 of a call can cross-fade (direction changes arrive between calls): it alone runs the
     * variant that carries a second accumulator tile */
    const int first = e.mix ? 1 : 0;
    if (first && e.nFrames == 1) {      /* one-block call with a cross-fade (head tracking): the block and the state copy in one launch */
        a.frameBase = 0;
        hipLaunchKernelGGL((enc_gemm_kernel<true, false>), dim3((e.F + 127) / 128, 2, e.nInst), dim3(128), 0, stream(), a);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (first) {
        a.frameBase = 0;
        hipLaunchKernelGGL((enc_gemm_kernel<true, false>), dim3((e.F + 127) / 128, 1, e.nInst), dim3(128), 0, stream(), a);
    }
    a.frameBase = first;
    const int nBlk = e.nFrames + 1 - first;
    const dim3 grid((e.F + 127) / 128, nBlk, e.nInst);
    if (e.maxSteps > 0 && e.maxSteps <= 8) hipLaunchKernelGGL(enc_gemm_small_kernel, grid, dim3(128), 0, stream(), a);
    else hipLaunchKernelGGL(enc_gemm_full_kernel, grid, dim3(128), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf

/* the overlap test of launch_enc_gemm, callable without a GPU (tests/test_lib_cpu.py compares it with brute force) */
extern "C" __attribute__((visibility("default"))) int saf_hip_debug_segments_overlap(const float* a, long long a0, long long a1, long long a2, int an0, int an1, int an2,
                                                                                       const float* b, long long b0, long long b1, long long b2, int bn0, int bn1, int bn2, int F)
{
    const saf::StridedSet sa{ a, { a0, a1, a2 }, { an0, an1, an2 }, F }, sb{ b, { b0, b1, b2 }, { bn0, bn1, bn2 }, F };
    return saf::strided_sets_overlap(sa, sb) ? 1 : 0;
}
