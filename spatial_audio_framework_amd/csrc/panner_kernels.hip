/*
 * panner_kernels.hip — frequency-dependent VBAP gains of the panner (examples/src/panner/panner.c:230-262).
 *
 * For every source whose direction changed: read its row of the VBAP gain table (the row index is computed on the
 * host with the reference's float arithmetic, so the SAME table row is chosen bit for bit), and for every band b
 *     G[b][ls] = g[ls] / ( (sum_ls max(g[ls], 0)^p_b)^(1/(p_b + 2.23e-9)) + 2.23e-9 )     (p_b != 2)
 *     G[b][ls] = g[ls]                                                                       (p_b == 2)
 * written as column `src` of the band's [loudspeaker x source] matrix, the A operand of the band GEMM.
 */
#include "saf_hip_common.h"

namespace saf {

struct PanGainArgs { PanGainLaunch l; };

/* grid (nSrc); 192 threads: thread = band */
__global__ __launch_bounds__(192) void panner_gains_kernel(PanGainArgs a)
{
    __shared__ float s_g[SAF_MAXCH];
    const PanGainLaunch& l = a.l;
    const int src = blockIdx.x;
    if (!l.recalc[src]) return;
    const int tid = threadIdx.x;
    if (tid < SAF_MAXCH) s_g[tid] = tid < l.nLS ? l.gtable[(long long)l.row[src] * l.nLS + tid] : 0.0f;
    __syncthreads();
    if (tid >= SAF_NBANDS) return;
    const float pv = l.pValue[tid];
    float inv = 1.0f;
    if (pv != 2.0f) {
        float s = 0.0f;
        for (int ls = 0; ls < l.nLS; ls++) s += powf(fmaxf(s_g[ls], 0.0f), pv);
        s = powf(s, 1.0f / (pv + 2.23e-9f));
        inv = s + 2.23e-9f;
    }
    float* A = l.A + (long long)tid * SAF_MAXCH * SAF_MAXCH + src;
    for (int ls = 0; ls < SAF_MAXCH; ls++) A[ls * SAF_MAXCH] = pv != 2.0f ? s_g[ls] / inv : s_g[ls];
}

void launch_panner_gains(const PanGainLaunch& l)
{
    if (l.nSrc <= 0) return;
    PanGainArgs a; a.l = l;
    KernelTimer kt("panner_gains");
    hipLaunchKernelGGL(panner_gains_kernel, dim3(l.nSrc), dim3(192), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
