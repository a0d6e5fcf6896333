/*
 * sh_kernels.hip — real spherical harmonics on the GPU, one thread per direction.
 *
 * Replaces getSHreal / unnorm_legendreP (framework/modules/saf_sh/saf_sh.c:53-127,190-253),
 * getSHreal_recur / unnorm_legendreP_recur (saf_sh.c:129-183,255-331),
 * getRSH and getRSH_recur (framework/modules/saf_hoa/saf_hoa.c:118-228).
 * The arithmetic type of each path is the reference's: float64 for the direct
 * form (cast to float at the end), float32 for the recursive form.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"

namespace saf {

#define SH_MAX_ORDER 40

__constant__ double c_fact_d[2 * SH_MAX_ORDER + 2];
__constant__ float  c_fact_f[2 * SH_MAX_ORDER + 2];
static bool g_fact_ready = false;

static void ensure_factorials()
{
    if (g_fact_ready) return;
    double fd[2 * SH_MAX_ORDER + 2]; float ff[2 * SH_MAX_ORDER + 2];
    for (int n = 0; n < 2 * SH_MAX_ORDER + 2; n++) {
        /* factorial() (saf_utility_misc.c:174-186) works in long double and below 15 reads a table
         * (saf_utility_misc.c:33-34) whose 14! entry is 8.71782891e10 instead of 87178291200; the
         * order-7 normalisation inherits that, so it is kept. */
        long double v = 1.0L;
        for (int i = 2; i <= n; i++) v *= (long double)i;
        if (n == 14) v = 8.71782891e10L;
        fd[n] = (double)v; ff[n] = (float)v;
    }
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_fact_d), fd, sizeof(fd)));
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_fact_f), ff, sizeof(ff)));
    g_fact_ready = true;
}

/* unnorm_legendreP (saf_sh.c:53-127) for a single x: y[0..n] */
__device__ void legendreP_d(int n, double x, double* y)
{
    if (n == 0) { y[0] = 1.0; return; }
    double P[SH_MAX_ORDER + 3];
    double sqrt_n[2 * SH_MAX_ORDER + 1];
    for (int i = 0; i < n + 3; i++) P[i] = 0.0;
    const double s = sqrt(1.0 - pow(x, 2.0)) + 2.23e-20;
    const double s_n = pow(-s, (double)n);
    const double tc = -2.0 * x / s;
    for (int i = 0; i < 2 * n + 1; i++) sqrt_n[i] = sqrt((double)i);
    double norm = 1.0;
    for (int i = 1; i <= n; i++) norm *= 1.0 - 1.0 / (2.0 * (double)i);
    P[n] = sqrt(norm) * s_n;
    P[n - 1] = P[n] * tc * (double)n / sqrt_n[2 * n];
    for (int m = n - 2; m >= 0; m--)
        P[m] = (P[m + 1] * tc * ((double)m + 1.0) - P[m + 2] * sqrt_n[n + m + 2] * sqrt_n[n - m - 1]) / (sqrt_n[n + m + 1] * sqrt_n[n - m]);
    for (int i = 0; i < n + 1; i++) y[i] = P[i];
    if (sqrt(1.0 - pow(x, 2.0)) == 0) y[0] = pow(x, (double)n);
    for (int m = 1; m < n; m++) {
        double scale = 1.0;
        for (int i = n - m + 1; i < n + m + 1; i++) scale *= sqrt_n[i];
        y[m] *= scale;
    }
    double scale = 1.0;
    for (int i = 1; i < 2 * n + 1; i++) scale *= sqrt_n[i];
    y[n] *= scale;
}

/* mode 0: dirs = [azi, inclination] rad (getSHreal); mode 1: dirs = [azi, elev] deg, result * sqrt(4 pi) (getRSH) */
__global__ void sh_real_kernel(int order, const float* dirs, int nDirs, float* Y, int mode)
{
    const int dir = blockIdx.x * blockDim.x + threadIdx.x;
    if (dir >= nDirs) return;
    float azi_f, incl_f;
    if (mode == 1) {
        azi_f = dirs[dir * 2 + 0] * SAF_PI / 180.0f;                       /* saf_hoa.c:139-140, float arithmetic */
        incl_f = SAF_PI / 2.0f - (dirs[dir * 2 + 1] * SAF_PI / 180.0f);
    } else { azi_f = dirs[dir * 2 + 0]; incl_f = dirs[dir * 2 + 1]; }
    const double x = cos((double)incl_f);
    const float post = sqrtf(4.0f * SAF_PI);
    double p_nm[SH_MAX_ORDER + 1];
    int idx_Y = 0;
    for (int n = 0; n <= order; n++) {
        legendreP_d(n, x, p_nm);
        for (int m = -n, j = 0; m <= n; m++, j++) {
            const int am = m < 0 ? -m : m;
            const double L = (n != 0) ? pow(-1.0, (double)am) * p_nm[am] : p_nm[0];
            const double norm = sqrt((2.0 * (double)n + 1.0) * c_fact_d[n - am] / (4.0 * SAF_PId * c_fact_d[n + am]));
            double v;
            if (j < n)       v = norm * L * sqrt(2.0) * sin((double)(n - j) * (double)azi_f);
            else if (j == n) v = norm * L;
            else             v = norm * L * sqrt(2.0) * cos((double)am * (double)azi_f);
            float f = (float)v;
            if (mode == 1) f *= post;                                       /* utility_svsmul, saf_hoa.c:147 */
            Y[(long long)(j + idx_Y) * nDirs + dir] = f;
        }
        idx_Y += 2 * n + 1;
    }
}

/* unnorm_legendreP_recur (saf_sh.c:129-183) for a single x */
__device__ void legendreP_recur_f(int n, float x, const float* Pm1, const float* Pm2, float* P)
{
    /* the three-term float recursion amplifies rounding differences: keep the reference's
     * multiply-then-subtract sequence instead of letting the compiler fuse it into FMAs */
#pragma clang fp contract(off)
    const float x2 = x * x;
    switch (n) {
        case 0: P[0] = 1.0f; break;
        case 1: P[0] = x; P[1] = sqrtf(1.0f - x2); break;
        case 2: P[0] = (3.0f * x2 - 1.0f) / 2.0f; P[1] = x * 3.0f * sqrtf(1.0f - x2); P[2] = 3.0f * (1.0f - x2); break;
        default: {
            const float one_min_x2 = 1.0f - x2;
            const int k = 2 * n - 1;
            float dfact_k = 1.0f;
            for (int kk = 1; kk < (k + 1) / 2 + 1; kk++) dfact_k *= (2.0f * (float)kk - 1.0f);   /* k is odd */
            P[n] = dfact_k * powf(one_min_x2, (float)n / 2.0f);
            P[n - 1] = (float)k * x * Pm1[n - 1];
            for (int m = 0; m < n - 1; m++)
                P[m] = (((float)k * x * Pm1[m]) - ((float)(n + m - 1) * Pm2[m])) / (float)(n - m);
        } break;
    }
}

/* One direction of getSHreal_recur (mode 0: dirs rad, orthonormal incl. 1/sqrt(4pi)) / getRSH_recur (mode 1: dirs deg,
 * N3D): writes Y[idx * stride], idx < (N+1)^2 (saf_sh.c:255-331, saf_hoa.c:152-228). */
__device__ void sh_recur_one(int N, float a, float b, int mode, float* Y, long long stride)
{
#pragma clang fp contract(off)
    float leg_n[SH_MAX_ORDER + 1], leg_n_1[SH_MAX_ORDER + 1], leg_n_2[SH_MAX_ORDER + 1];
    for (int i = 0; i <= N; i++) leg_n[i] = leg_n_1[i] = leg_n_2[i] = 0.0f;
    const float ci = mode ? sinf(b * SAF_PI / 180.0f) : cosf(b);
    int index_n = 0;
    for (int n = 0; n < N + 1; n++) {
        if (n == 0) {
            Y[0] = mode ? 1.0f : 1.0f / SAF_SQRT4PI;
            index_n = 1;
        } else {
            legendreP_recur_f(n, ci, leg_n_1, leg_n_2, leg_n);
            const float Nn0 = sqrtf(2.0f * (float)n + 1.0f);
            for (int m = 0; m < n + 1; m++) {
                if (m == 0) {
                    Y[(long long)(index_n + n) * stride] = mode ? Nn0 * leg_n[m] : Nn0 / SAF_SQRT4PI * leg_n[m];
                } else {
                    const float Nnm = Nn0 * sqrtf(2.0f * c_fact_f[n - m] / c_fact_f[n + m]);
                    if (mode) {
                        Y[(long long)(index_n + n - m) * stride] = Nnm * leg_n[m] * sinf((float)m * a * SAF_PI / 180.0f);
                        Y[(long long)(index_n + n + m) * stride] = Nnm * leg_n[m] * cosf((float)m * a * SAF_PI / 180.0f);
                    } else {
                        Y[(long long)(index_n + n - m) * stride] = Nnm / SAF_SQRT4PI * leg_n[m] * sinf((float)m * a);
                        Y[(long long)(index_n + n + m) * stride] = Nnm / SAF_SQRT4PI * leg_n[m] * cosf((float)m * a);
                    }
                }
            }
            index_n += 2 * n + 1;
        }
        for (int i = 0; i <= N; i++) { leg_n_2[i] = leg_n_1[i]; leg_n_1[i] = leg_n[i]; }
    }
}

__global__ void sh_recur_kernel(int N, const float* dirs, int nDirs, float* Y, int mode)
{
    const int dir = blockIdx.x * blockDim.x + threadIdx.x;
    if (dir >= nDirs) return;
    sh_recur_one(N, dirs[dir * 2 + 0], dirs[dir * 2 + 1], mode, Y + dir, nDirs);
}

/* ambi_enc_process, "recalculate SHs" (ambi_enc.c:120-131): for every flagged source of every instance
 * Y[inst][:, ch] = getRSH_recur(order, dir), rows beyond nSH zeroed.  Y[inst] is [64][64] at instance stride y_inst, row stride 64
 * (= MAX_NUM_INPUTS).  grid = nInst, block = 64 (one thread per source). */
__global__ void enc_update_Y_kernel(const int* order, const float* dirs, const int* recalc, float* Y, long long y_inst)
{
    const int inst = blockIdx.x, ch = threadIdx.x;
    if (!recalc[inst * SAF_MAXCH + ch]) return;
    const int N = order[inst];
    float* y = Y + (long long)inst * y_inst + ch;
    sh_recur_one(N, dirs[(inst * SAF_MAXCH + ch) * 2 + 0], dirs[(inst * SAF_MAXCH + ch) * 2 + 1], 1, y, SAF_MAXCH);
    for (int j = ORDER2NSH(N); j < SAF_MAXCH; j++) y[j * SAF_MAXCH] = 0.0f;
}

void launch_enc_update_Y(const int* d_order, const float* d_dirs, const int* d_recalc, float* d_Y, long long y_inst, int nInst)
{
    ensure_factorials();
    hipLaunchKernelGGL(enc_update_Y_kernel, dim3(nInst), dim3(SAF_MAXCH), 0, stream(), d_order, d_dirs, d_recalc, d_Y, y_inst);
    HIP_CHECK(hipGetLastError());
}

/* kind: 0 getSHreal, 1 getRSH, 2 getSHreal_recur, 3 getRSH_recur; device pointers */
void sh_eval_dev(int kind, int order, const float* d_dirs, int nDirs, float* d_Y)
{
    if (nDirs < 1) return;
    if (order > SH_MAX_ORDER) SAF_FATAL("spherical harmonic order %d exceeds the supported maximum %d", order, SH_MAX_ORDER);
    ensure_factorials();
    dim3 grid((nDirs + 63) / 64), block(64);
    if (kind < 2) hipLaunchKernelGGL(sh_real_kernel, grid, block, 0, stream(), order, d_dirs, nDirs, d_Y, kind);
    else          hipLaunchKernelGGL(sh_recur_kernel, grid, block, 0, stream(), order, d_dirs, nDirs, d_Y, kind - 2);
    HIP_CHECK(hipGetLastError());
}

/* host-pointer wrapper used by the C API and by the init-time design code */
void sh_eval_host(int kind, int order, const float* dirs, int nDirs, float* Y)
{
    if (nDirs < 1) return;
    ensure_device();
    const int nSH = ORDER2NSH(order);
    DevBuf<float> d_dirs, d_Y;
    d_dirs.alloc((size_t)nDirs * 2, false);
    d_Y.alloc((size_t)nSH * nDirs, false);
    HIP_CHECK(hipMemcpyAsync(d_dirs.p, dirs, sizeof(float) * nDirs * 2, hipMemcpyHostToDevice, stream()));
    sh_eval_dev(kind, order, d_dirs.p, nDirs, d_Y.p);
    HIP_CHECK(hipMemcpyAsync(Y, d_Y.p, sizeof(float) * (size_t)nSH * nDirs, hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
}

}  // namespace saf

extern "C" {
void getSHreal(int order, float* dirs_rad, int nDirs, float* Y) { saf::sh_eval_host(0, order, dirs_rad, nDirs, Y); }
void getRSH(int order, float* dirs_deg, int nDirs, float* Y) { saf::sh_eval_host(1, order, dirs_deg, nDirs, Y); }
void getSHreal_recur(int order, float* dirs_rad, int nDirs, float* Y) { saf::sh_eval_host(2, order, dirs_rad, nDirs, Y); }
void getRSH_recur(int order, float* dirs_deg, int nDirs, float* Y) { saf::sh_eval_host(3, order, dirs_deg, nDirs, Y); }
void saf_hip_getRSH_recur_dev(int order, const float* d_dirs_deg, int nDirs, float* d_Y) { saf::sh_eval_dev(3, order, d_dirs_deg, nDirs, d_Y); }
}
