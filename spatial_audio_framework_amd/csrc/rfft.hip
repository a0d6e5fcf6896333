/*
 * rfft.hip — saf_rfft_create / _forward / _backward / _destroy (framework/modules/saf_utilities/saf_utility_fft.h,
 * saf_utility_fft.c:531-753) as a stand-alone GPU object: the real FFT of size N as a complex FFT of M = N/2 points
 * (packing of kiss_fftr.c:69-161) — forward unscaled with N/2+1 bins, backward scaled 1/N and ignoring the imaginary
 * parts of DC and Nyquist.  Sizes: any even N whose half factors into 2, 3, 5 (and other primes up to 31 through a
 * generic butterfly), i.e. every size of the reference's test__saf_rfft (16 … 1 048 576, 80 … 30 720).
 *
 * The hot kernels do not use this object (their transforms live in registers / LDS: afstft_kernels.hip,
 * pconv_kernels.hip); it is the drop-in for callers of saf_rfft_* and the target of the reference's own FFT test.
 * Complex FFT: Stockham autosort, decimation in frequency, one pass per factor r, ping-pong in global memory:
 *     y[q + s (r p + j)] = ( sum_k x[q + s (p + m k)] w_r^{jk} ) w_n^{jp},   n = r m the remaining length, s the stride.
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"

namespace saf {

struct RfftHandle {
    int N, M;
    std::vector<int> factors;
    DevBuf<float2> twM;      /* exp(-2 pi i k / M), k < M */
    DevBuf<float2> twN;      /* exp(-2 pi i k / N), k <= M */
    DevBuf<float2> A, B;     /* M + 1 each */
    PinBuf<float> h;         /* staging, N + 2 floats */
};

__device__ __forceinline__ float2 rf_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

/* one Stockham pass of radix R (R = 0: generic radix r <= 31) */
template <int R>
__global__ __launch_bounds__(256) void stockham_pass_kernel(const float2* __restrict__ x, float2* __restrict__ y, const float2* __restrict__ tw,
                                                             int M, int n, int s, int rgen, int inverse)
{
    const int r = R ? R : rgen;
    const int m = n / r;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * s) return;
    const int p = idx / s, q = idx - p * s;
    float2 a[R ? R : 31];
    for (int k = 0; k < r; k++) a[k] = x[q + s * (p + m * k)];
    const int stepN = M / n;                       /* w_n^t = tw[t * stepN] */
    const int stepR = M / r;                       /* w_r^t = tw[t * stepR] */
    for (int j = 0; j < r; j++) {
        float2 acc = a[0];
        for (int k = 1; k < r; k++) {
            float2 w = tw[((j * k) % r) * stepR];
            if (inverse) w.y = -w.y;
            const float2 t = rf_cmul(a[k], w);
            acc.x += t.x; acc.y += t.y;
        }
        float2 w = tw[(long long)(j * p) * stepN];
        if (inverse) w.y = -w.y;
        y[q + s * (r * p + j)] = rf_cmul(acc, w);
    }
}

/* real spectrum from the packed transform (kiss_fftr.c:86-123): X[k], k = 0..M */
__global__ __launch_bounds__(256) void rfft_split_kernel(const float2* __restrict__ Z, float2* __restrict__ X, const float2* __restrict__ twN, int M)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > M) return;
    const float2 Zk = Z[k == M ? 0 : k], Zm = Z[k == 0 || k == M ? 0 : M - k];
    const float2 W = twN[k];
    const float2 e = make_float2(Zk.x + Zm.x, Zk.y - Zm.y), d = make_float2(Zk.x - Zm.x, Zk.y + Zm.y);
    const float2 t = rf_cmul(W, d);
    float2 o = make_float2(0.5f * (e.x + t.y), 0.5f * (e.y - t.x));
    if (k == 0 || k == M) o.y = 0.0f;
    X[k] = o;
}

/* packed spectrum for the inverse (kiss_fftr.c:125-161), already scaled by 1/N: Z[k], k < M */
__global__ __launch_bounds__(256) void rfft_merge_kernel(const float2* __restrict__ X, float2* __restrict__ Z, const float2* __restrict__ twN, int M, float scale)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= M) return;
    float2 Xk = X[k], Xm = X[M - k];
    if (k == 0) { Xk.y = 0.0f; Xm.y = 0.0f; }                       /* Im of DC and Nyquist ignored */
    const float2 E = make_float2(Xk.x + Xm.x, Xk.y - Xm.y), D = make_float2(Xk.x - Xm.x, Xk.y + Xm.y);
    float2 W = twN[k]; W.y = -W.y;
    const float2 O = rf_cmul(D, W);
    Z[k] = make_float2((E.x - O.y) * scale, (E.y + O.x) * scale);      /* (E + i O) / N: 1/2 of the packing times 1/M of the inverse */
}

static void complex_fft(RfftHandle* h, float2*& src, float2*& dst, bool inverse)
{
    int n = h->M, s = 1;
    for (int r : h->factors) {
        const int threads = (n / r) * s;
        const dim3 grid((threads + 255) / 256);
        switch (r) {
            case 2: hipLaunchKernelGGL(stockham_pass_kernel<2>, grid, dim3(256), 0, stream(), src, dst, h->twM.p, h->M, n, s, r, inverse ? 1 : 0); break;
            case 3: hipLaunchKernelGGL(stockham_pass_kernel<3>, grid, dim3(256), 0, stream(), src, dst, h->twM.p, h->M, n, s, r, inverse ? 1 : 0); break;
            case 4: hipLaunchKernelGGL(stockham_pass_kernel<4>, grid, dim3(256), 0, stream(), src, dst, h->twM.p, h->M, n, s, r, inverse ? 1 : 0); break;
            case 5: hipLaunchKernelGGL(stockham_pass_kernel<5>, grid, dim3(256), 0, stream(), src, dst, h->twM.p, h->M, n, s, r, inverse ? 1 : 0); break;
            default: hipLaunchKernelGGL(stockham_pass_kernel<0>, grid, dim3(256), 0, stream(), src, dst, h->twM.p, h->M, n, s, r, inverse ? 1 : 0); break;
        }
        HIP_CHECK(hipGetLastError());
        std::swap(src, dst);
        n /= r; s *= r;
    }
}

}  // namespace saf

using namespace saf;

extern "C" {

void saf_rfft_create(void** const phFFT, int N)       /* saf_utility_fft.c:531-640 */
{
    ensure_device();
    if (N < 2 || (N & 1)) SAF_FATAL("saf_rfft_create: N must be even and >= 2 (got %d)", N);
    RfftHandle* h = new RfftHandle();
    h->N = N; h->M = N / 2;
    int m = h->M;
    while (m % 4 == 0) { h->factors.push_back(4); m /= 4; }
    for (int r : { 2, 3, 5 }) while (m % r == 0) { h->factors.push_back(r); m /= r; }
    for (int r = 7; r <= 31 && m > 1; r += 2) while (m % r == 0) { h->factors.push_back(r); m /= r; }
    if (m != 1) SAF_FATAL("saf_rfft_create: N/2 = %d has a prime factor above 31, which this build does not transform", h->M);
    std::vector<float2> tM(h->M), tN(h->M + 1);
    for (int k = 0; k < h->M; k++) { const double a = -2.0 * SAF_PId * (double)k / (double)h->M; tM[k] = make_float2((float)cos(a), (float)sin(a)); }
    for (int k = 0; k <= h->M; k++) { const double a = -2.0 * SAF_PId * (double)k / (double)N; tN[k] = make_float2((float)cos(a), (float)sin(a)); }
    tN[h->M] = make_float2(-1.0f, 0.0f);
    h->twM.alloc(h->M, false); h->twN.alloc(h->M + 1, false); h->A.alloc(h->M + 1); h->B.alloc(h->M + 1);
    HIP_CHECK(hipMemcpy(h->twM.p, tM.data(), sizeof(float2) * h->M, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(h->twN.p, tN.data(), sizeof(float2) * (h->M + 1), hipMemcpyHostToDevice));
    h->h.ensure((size_t)N + 2);
    *phFFT = h;
}

void saf_rfft_destroy(void** const phFFT)
{
    RfftHandle* h = (RfftHandle*)*phFFT;
    if (!h) return;
    HIP_CHECK(hipStreamSynchronize(stream()));
    delete h;
    *phFFT = nullptr;
}

void saf_rfft_forward(void* const hFFT, float* inputTD, float_complex* outputFD)      /* saf_utility_fft.c:690-726 */
{
    RfftHandle* h = (RfftHandle*)hFFT;
    memcpy(h->h.p, inputTD, sizeof(float) * h->N);
    HIP_CHECK(hipMemcpyAsync(h->A.p, h->h.p, sizeof(float) * h->N, hipMemcpyHostToDevice, stream()));     /* z[m] = x[2m] + i x[2m+1] */
    float2 *src = h->A.p, *dst = h->B.p;
    complex_fft(h, src, dst, false);
    hipLaunchKernelGGL(rfft_split_kernel, dim3((h->M + 256) / 256), dim3(256), 0, stream(), src, dst, h->twN.p, h->M);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(h->h.p, dst, sizeof(float2) * (h->M + 1), hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
    memcpy((void*)outputFD, h->h.p, sizeof(float2) * (h->M + 1));
}

void saf_rfft_backward(void* const hFFT, float_complex* inputFD, float* outputTD)     /* saf_utility_fft.c:728-753 */
{
    RfftHandle* h = (RfftHandle*)hFFT;
    memcpy(h->h.p, (const void*)inputFD, sizeof(float2) * (h->M + 1));
    HIP_CHECK(hipMemcpyAsync(h->A.p, h->h.p, sizeof(float2) * (h->M + 1), hipMemcpyHostToDevice, stream()));
    float2 *src = h->B.p, *dst = h->A.p;
    hipLaunchKernelGGL(rfft_merge_kernel, dim3((h->M + 255) / 256), dim3(256), 0, stream(), h->A.p, h->B.p, h->twN.p, h->M, 1.0f / (float)h->N);
    HIP_CHECK(hipGetLastError());
    complex_fft(h, src, dst, true);
    HIP_CHECK(hipMemcpyAsync(h->h.p, src, sizeof(float) * h->N, hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipStreamSynchronize(stream()));
    memcpy(outputTD, h->h.p, sizeof(float) * h->N);
}

}
