/*
 * pmap_adaptive_kernels.hip — the adaptive and sub-space activity maps of powermap (powermap.c:294-341) for gfx950:
 * generateMVDRmap, generateCroPaCLCMVmap, generateMUSICmap, generateMinNormMap (saf_sh.c:1586-1858) on the ONE grouped
 * nM x nM covariance matrix of a map update (nM <= 64), for the 812-direction scanning grid.
 *
 *   cgrp_cplx_kernel   grouped complex covariance C = sum_b 1e3 EQ_b Cx_b (leading nSH_b x nSH_b blocks), powermap.c:279-289
 *   chol_kernel        C_d = C + regPar * trace/nM * I, lower Cholesky factor in LDS (float64), one workgroup
 *   mvdr_kernel        per direction: z = C_d^-1 y (two triangular solves out of LDS), w = z / (y^T conj z),
 *                      map = Re(w^T C w); CroPaC adds the two-constraint LCMV weight and the cross-spectrum gain
 *   herm_eig_kernel    Hermitian eigen-decomposition by parallel cyclic Jacobi (round-robin pairs) in LDS (float64),
 *                      one workgroup; eigenvalues sorted descending, vectors unit-norm with the largest component
 *                      real positive; also the min-norm vector Un
 *   subspace_kernel    per direction: MUSIC 1 / sum_j |Vn_j^T y|^2 or MinNorm 1 / |Un^H y|^2 (+ log)
 *
 * The reference factorises in single precision with LAPACK (cposv, cgesv, cheev, cgeev); float64 in LDS costs nothing at
 * this size (MI355X vector FP64 = half the FP32 rate) and makes the result independent of elimination order.
 * All maps end with the temporal smoothing of powermap.c:343-346.
 */
#include "saf_hip_common.h"

namespace saf {

typedef double2 zc;
__device__ __forceinline__ zc zmul(zc a, zc b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ zc zmulc(zc a, zc b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   /* a * conj(b) */
__device__ __forceinline__ zc zsub(zc a, zc b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cdivf(float2 a, float2 b)
{
    const float d = b.x * b.x + b.y * b.y;
    return make_float2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

struct AdArgs { AdaptMapLaunch l; };

/* grid (64 rows); 256 threads = 64 columns x 4 band groups; zero outside the leading nM x nM block.  Unconditional
 * loads with a select (see cgrp_kernel in powermap_kernels.hip); contiguous band ranges summed in ascending order, the
 * four partial sums added in group order. */
__global__ __launch_bounds__(256) void cgrp_cplx_kernel(AdArgs a)
{
    __shared__ float2 s_p[4][64];
    const AdaptMapLaunch& l = a.l;
    const int i = blockIdx.x, j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int per = (l.nBands + 3) / 4;
    const int b0 = g * per, b1 = b0 + per < l.nBands ? b0 + per : l.nBands;
    float re = 0.0f, im = 0.0f;
    const float2* Ce = l.Cx + i * 64 + j;
#pragma unroll 17
    for (int band = b0; band < b1; band++) {
        const int ns = l.bandNSH[band];
        const float2 c = Ce[(long long)band * 64 * 64];
        const float sc = l.bandScale[band];
        const bool on = i < ns && j < ns;
        re += on ? c.x * sc : 0.0f; im += on ? c.y * sc : 0.0f;
    }
    s_p[g][j] = make_float2(re, im);
    __syncthreads();
    if (g == 0)
        l.Cg[i * 64 + j] = make_float2(((s_p[0][j].x + s_p[1][j].x) + s_p[2][j].x) + s_p[3][j].x, ((s_p[0][j].y + s_p[1][j].y) + s_p[2][j].y) + s_p[3][j].y);
}

/* one workgroup, 256 threads.  status[0] = 1 when trace > 1e-8 and the factorisation succeeded, else 0 (map = 0) */
__global__ __launch_bounds__(256) void chol_kernel(AdArgs a)
{
    extern __shared__ zc s_A[];                       /* [nM][65] */
    const AdaptMapLaunch& l = a.l;
    const int n = l.nM, tid = threadIdx.x;
    __shared__ float s_tr;
    __shared__ int s_ok;
    if (tid == 0) {
        float tr = 0.0f;
        for (int i = 0; i < n; i++) tr += l.Cg[i * 64 + i].x;
        s_tr = tr; s_ok = tr > 1e-8f ? 1 : 0;
    }
    __syncthreads();
    const float load = l.regPar * (s_tr / (float)n);
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        const float2 c = l.Cg[i * 64 + j];
        s_A[i * 65 + j] = i == j ? make_double2((double)(c.x + load), (double)c.y) : make_double2((double)c.x, (double)c.y);
    }
    __syncthreads();
    /* right-looking Cholesky, lower triangle in place */
    for (int j = 0; j < n && s_ok; j++) {
        __shared__ double s_d;
        if (tid == 0) { const double d = s_A[j * 65 + j].x; if (!(d > 0.0)) s_ok = 0; s_d = d > 0.0 ? sqrt(d) : 1.0; }
        __syncthreads();
        const double ljj = s_d;
        for (int i = j + 1 + tid; i < n; i += 256) { zc v = s_A[i * 65 + j]; v.x /= ljj; v.y /= ljj; s_A[i * 65 + j] = v; }
        if (tid == 0) s_A[j * 65 + j] = make_double2(ljj, 0.0);
        __syncthreads();
        const int m = n - j - 1;                      /* trailing update: A[i][k] -= L[i][j] conj(L[k][j]), k <= i */
        for (int e = tid; e < m * m; e += 256) {
            const int i = j + 1 + e / m, k = j + 1 + e % m;
            if (k <= i) s_A[i * 65 + k] = zsub(s_A[i * 65 + k], zmulc(s_A[i * 65 + j], s_A[k * 65 + j]));
        }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        l.Lchol[i * 64 + j] = j <= i ? s_A[i * 65 + j] : make_double2(0.0, 0.0);
    }
    if (tid == 0) l.status[0] = s_ok;
}

/* TPB directions per workgroup; LDS: L [nM][nM] + TPB solve vectors (x2 for CroPaC) */
template <int TPB, bool CROPAC>
__global__ __launch_bounds__(TPB) void mvdr_kernel(AdArgs a)
{
    extern __shared__ zc s_mem[];
    const AdaptMapLaunch& l = a.l;
    const int n = l.nM, G = l.G, tid = threadIdx.x;
    zc* s_L = s_mem;                                  /* [n][n] */
    zc* s_z = s_mem + n * n;                          /* [n][TPB] */
    zc* s_z1 = s_z + n * TPB;                         /* [n][TPB] (CroPaC) */
    for (int e = tid; e < n * n; e += TPB) s_L[e] = l.Lchol[(e / n) * 64 + e % n];
    __syncthreads();
    const int d = blockIdx.x * TPB + tid;
    if (d >= G) return;
    float out = 0.0f;
    if (l.status[0]) {
        const float* Y = l.Ygrid + d;
        auto solve = [&](zc* z) {                     /* z <- (L L^H)^-1 z */
            for (int i = 0; i < n; i++) {
                zc s = z[i * TPB + tid];
                for (int k = 0; k < i; k++) s = zsub(s, zmul(s_L[i * n + k], z[k * TPB + tid]));
                const double r = s_L[i * n + i].x;
                z[i * TPB + tid] = make_double2(s.x / r, s.y / r);
            }
            for (int i = n - 1; i >= 0; i--) {
                zc s = z[i * TPB + tid];
                for (int k = i + 1; k < n; k++) { const zc lk = s_L[k * n + i]; s = zsub(s, zmul(make_double2(lk.x, -lk.y), z[k * TPB + tid])); }
                const double r = s_L[i * n + i].x;
                z[i * TPB + tid] = make_double2(s.x / r, s.y / r);
            }
        };
        for (int j = 0; j < n; j++) s_z[j * TPB + tid] = make_double2((double)Y[(long long)j * G], 0.0);
        solve(s_z);
        float2 den = make_float2(0.f, 0.f);           /* sum_j y_j conj(z_j) (saf_sh.c:1622-1627) */
        for (int j = 0; j < n; j++) { const float y = Y[(long long)j * G]; const zc z = s_z[j * TPB + tid]; den.x += y * (float)z.x; den.y -= y * (float)z.y; }
        float gain = 1.0f;
        float mv = 0.0f;
        /* MVDR map = Re(w^T C w), w = z / den (saf_sh.c:1629-1634 then generatePWDmap).  w_j is parked in the second solve
         * vector's slots when CroPaC still needs z, else over z itself. */
        {
            zc* s_w = CROPAC ? s_z1 : s_z;
            for (int j = 0; j < n; j++) {
                const zc z = s_z[j * TPB + tid];
                const float2 w = cdivf(make_float2((float)z.x, (float)z.y), den);
                s_w[j * TPB + tid] = make_double2((double)w.x, (double)w.y);
                if (l.Wout) l.Wout[(long long)j * G + d] = w;
            }
            float2 acc = make_float2(0.f, 0.f);
            for (int i = 0; i < n; i++) {
                float2 cw = make_float2(0.f, 0.f);
                for (int j = 0; j < n; j++) {
                    const zc wz = s_w[j * TPB + tid];
                    const float2 t = cmulf(l.Cg[i * 64 + j], make_float2((float)wz.x, (float)wz.y));
                    cw.x += t.x; cw.y += t.y;
                }
                const zc wi = s_w[i * TPB + tid];
                const float2 t = cmulf(make_float2((float)wi.x, (float)wi.y), cw);
                acc.x += t.x; acc.y += t.y;
            }
            mv = acc.x;
        }
        out = mv;
        if (CROPAC) {
            /* second constraint column a1 = y .* diag(C), solved with the same factor (saf_sh.c:1704-1711) */
            for (int j = 0; j < n; j++) { const float2 c1 = cmulf(make_float2(Y[(long long)j * G], 0.f), l.Cg[j * 64 + j]); s_z1[j * TPB + tid] = make_double2((double)c1.x, (double)c1.y); }
            solve(s_z1);
            float2 M00 = make_float2(0.f, 0.f), M01 = M00, M10 = M00, M11 = M00;       /* A^H conj(C_d^-1 A) (:1712-1717) */
            for (int j = 0; j < n; j++) {
                const float y = Y[(long long)j * G];
                const float2 c0 = make_float2(y, 0.f), c1 = cmulf(c0, l.Cg[j * 64 + j]);
                const zc a0 = s_z[j * TPB + tid], a1 = s_z1[j * TPB + tid];
                const float2 s0 = make_float2((float)a0.x, -(float)a0.y), s1 = make_float2((float)a1.x, -(float)a1.y);
                const float2 c0c = make_float2(c0.x, -c0.y), c1c = make_float2(c1.x, -c1.y);
                float2 t;
                t = cmulf(c0c, s0); M00.x += t.x; M00.y += t.y;
                t = cmulf(c0c, s1); M01.x += t.x; M01.y += t.y;
                t = cmulf(c1c, s0); M10.x += t.x; M10.y += t.y;
                t = cmulf(c1c, s1); M11.x += t.x; M11.y += t.y;
            }
            const float2 p = cmulf(M00, M11), q = cmulf(M01, M10);
            const float2 det = make_float2(p.x - q.x, p.y - q.y);
            float2 xs = make_float2(0.f, 0.f);        /* wo . (C y) (:1729-1731) */
            for (int j = 0; j < n; j++) {
                const zc a0 = s_z[j * TPB + tid], a1 = s_z1[j * TPB + tid];
                const float2 u = cmulf(M11, make_float2((float)a0.x, (float)a0.y)), v = cmulf(M01, make_float2((float)a1.x, (float)a1.y));
                const float2 wo = cdivf(make_float2(u.x - v.x, u.y - v.y), det);
                float2 cy = make_float2(0.f, 0.f);
                for (int k = 0; k < n; k++) { const float y = Y[(long long)k * G]; const float2 c = l.Cg[j * 64 + k]; cy.x += c.x * y; cy.y += c.y * y; }
                const float2 t = cmulf(wo, cy);
                xs.x += t.x; xs.y += t.y;
            }
            float S = hypotf(xs.x, xs.y); if (mv < S) S = mv;
            gain = sqrtf(S / (mv + 2.23e-10f));
            if (gain < l.lambda) gain = l.lambda;
            out = mv * gain * gain;                   /* Re((G w)^T C (G w)) with real G */
        }
    }
    const float v = (1.0f - l.avg) * out + l.avg * l.prev_pmap[d];
    l.pmap[d] = v;
    l.prev_pmap[d] = v;
}

/* one workgroup of 256 threads: Hermitian Jacobi in LDS (float64).  n_e = nM rounded up to even; round-robin pairing. */
__global__ __launch_bounds__(256) void herm_eig_kernel(AdArgs a)
{
    extern __shared__ zc s_mem[];
    const AdaptMapLaunch& l = a.l;
    const int n = l.nM, tid = threadIdx.x;
    const int ne = n + (n & 1), half = ne / 2;
    zc* s_A = s_mem;                                  /* [n][65] */
    zc* s_V = s_mem + 64 * 65;                        /* [n][65] */
    __shared__ double s_c[32], s_s[32]; __shared__ zc s_ph[32]; __shared__ int s_p[32], s_q[32];
    __shared__ double s_red[256]; __shared__ int s_ord[64]; __shared__ zc s_scl[64];
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        const float2 u = l.Cg[i * 64 + j], v = l.Cg[j * 64 + i];
        s_A[i * 65 + j] = make_double2(0.5 * ((double)u.x + (double)v.x), 0.5 * ((double)u.y - (double)v.y));    /* Hermitian part */
        s_V[i * 65 + j] = make_double2(i == j ? 1.0 : 0.0, 0.0);
    }
    if (tid == 0) {
        float tr = 0.0f;
        for (int i = 0; i < n; i++) tr += l.Cg[i * 64 + i].x;
        l.status[0] = tr > 1e-8f ? 1 : 0;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 24; sweep++) {
        /* convergence: off-diagonal mass against the diagonal */
        double off = 0.0, dg = 0.0;
        for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e - i * n; const zc v = s_A[i * 65 + j]; const double m = v.x * v.x + v.y * v.y; if (i == j) dg += m; else off += m; }
        s_red[tid] = off; __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (tid < st) s_red[tid] += s_red[tid + st]; __syncthreads(); }
        const double offT = s_red[0]; __syncthreads();
        s_red[tid] = dg; __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (tid < st) s_red[tid] += s_red[tid + st]; __syncthreads(); }
        const double dgT = s_red[0]; __syncthreads();
        if (offT <= 1e-30 * (dgT + 1e-300)) break;
        for (int step = 0; step < ne - 1; step++) {
            /* round-robin tournament: player ne-1 fixed, the others rotate */
            if (tid < half) {
                int p = tid == 0 ? ne - 1 : (step + tid) % (ne - 1);
                int q = (step + ne - 1 - tid) % (ne - 1);
                if (p > q) { const int t = p; p = q; q = t; }
                s_p[tid] = p; s_q[tid] = q;
                double c = 1.0, s = 0.0; zc ph = make_double2(1.0, 0.0);
                if (q < n) {
                    const zc apq = s_A[p * 65 + q];
                    const double g = sqrt(apq.x * apq.x + apq.y * apq.y);
                    if (g > 0.0) {
                        const double app = s_A[p * 65 + p].x, aqq = s_A[q * 65 + q].x;
                        ph = make_double2(apq.x / g, apq.y / g);
                        const double tau = (aqq - app) / (2.0 * g);
                        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t); s = t * c;
                    }
                }
                s_c[tid] = c; s_s[tid] = s; s_ph[tid] = ph;
            }
            __syncthreads();
            /* columns: A <- A R, V <- V R, R = [[c, s ph], [-s conj(ph), c]] on (p, q) */
            for (int e = tid; e < half * n; e += 256) {
                const int pr = e / n, k = e - pr * n, p = s_p[pr], q = s_q[pr];
                if (q >= n) continue;
                const double c = s_c[pr], s = s_s[pr]; const zc ph = s_ph[pr];
                const zc akp = s_A[k * 65 + p], akq = s_A[k * 65 + q];
                const zc t1 = zmulc(akq, ph), t2 = zmul(akp, ph);
                s_A[k * 65 + p] = make_double2(c * akp.x - s * t1.x, c * akp.y - s * t1.y);
                s_A[k * 65 + q] = make_double2(s * t2.x + c * akq.x, s * t2.y + c * akq.y);
                const zc vkp = s_V[k * 65 + p], vkq = s_V[k * 65 + q];
                const zc u1 = zmulc(vkq, ph), u2 = zmul(vkp, ph);
                s_V[k * 65 + p] = make_double2(c * vkp.x - s * u1.x, c * vkp.y - s * u1.y);
                s_V[k * 65 + q] = make_double2(s * u2.x + c * vkq.x, s * u2.y + c * vkq.y);
            }
            __syncthreads();
            /* rows: A <- R^H A */
            for (int e = tid; e < half * n; e += 256) {
                const int pr = e / n, k = e - pr * n, p = s_p[pr], q = s_q[pr];
                if (q >= n) continue;
                const double c = s_c[pr], s = s_s[pr]; const zc ph = s_ph[pr];
                const zc apk = s_A[p * 65 + k], aqk = s_A[q * 65 + k];
                const zc t1 = zmul(aqk, ph), t2 = zmulc(apk, ph);
                s_A[p * 65 + k] = make_double2(c * apk.x - s * t1.x, c * apk.y - s * t1.y);
                s_A[q * 65 + k] = make_double2(s * t2.x + c * aqk.x, s * t2.y + c * aqk.y);
            }
            __syncthreads();
        }
    }
    /* descending order (utility_cseig with sortDecFLAG, saf_utility_veclib.c:2086-2105); ties by index */
    if (tid < n) {
        const double e = s_A[tid * 65 + tid].x;
        int rank = 0;
        for (int k = 0; k < n; k++) { const double ek = s_A[k * 65 + k].x; if (ek > e || (ek == e && k < tid)) rank++; }
        s_ord[rank] = tid;
        /* unit norm, largest component real and positive (cgeev's normalisation) */
        double nrm = 0.0, big = -1.0; int kb = 0;
        for (int k = 0; k < n; k++) { const zc v = s_V[k * 65 + tid]; const double m = v.x * v.x + v.y * v.y; nrm += m; if (m > big) { big = m; kb = k; } }
        const zc vb = s_V[kb * 65 + tid];
        const double sc = 1.0 / (sqrt(big) * sqrt(nrm));
        s_scl[tid] = make_double2(vb.x * sc, -vb.y * sc);
    }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
        const int i = e >> 6, c = e & 63;
        float2 o = make_float2(0.f, 0.f);
        if (i < n && c < n) { const int s = s_ord[c]; const zc v = zmul(s_V[i * 65 + s], s_scl[s]); o = make_float2((float)v.x, (float)v.y); }
        l.Veig[i * 64 + c] = o;
    }
    if (tid < n) l.eig[tid] = (float)s_A[s_ord[tid] * 65 + s_ord[tid]].x;
    __syncthreads();
    /* min-norm vector Un = Vn Vn1^H / (Vn1 . Vn1 + 2.23e-9) (saf_sh.c:1832-1842; the dot product is NOT conjugated) */
    int nS = l.nSources < n / 2 ? l.nSources : n / 2;
    if (tid < n) {
        float2 dot = make_float2(0.f, 0.f);
        for (int j = nS; j < n; j++) { const int s = s_ord[j]; const zc vz = zmul(s_V[0 * 65 + s], s_scl[s]); const float2 v = make_float2((float)vz.x, (float)vz.y); const float2 t = cmulf(v, v); dot.x += t.x; dot.y += t.y; }
        float2 acc = make_float2(0.f, 0.f);
        for (int j = nS; j < n; j++) {
            const int s = s_ord[j];
            const zc v1 = zmul(s_V[0 * 65 + s], s_scl[s]), vi = zmul(s_V[tid * 65 + s], s_scl[s]);
            const float2 t = cmulf(make_float2((float)vi.x, (float)vi.y), make_float2((float)v1.x, -(float)v1.y));
            acc.x += t.x; acc.y += t.y;
        }
        l.Un[tid] = cdivf(acc, make_float2(dot.x + 2.23e-9f, dot.y));
    }
}

/* grid ceil(G/256) x 256: mode 4/5 MUSIC (log), 6/7 MinNorm (log) */
__global__ __launch_bounds__(256) void subspace_kernel(AdArgs a)
{
    __shared__ float2 s_V[64 * 64];
    __shared__ float2 s_Un[64];
    const AdaptMapLaunch& l = a.l;
    const int n = l.nM, G = l.G;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) s_V[e] = l.Veig[e];
    if (threadIdx.x < 64) s_Un[threadIdx.x] = threadIdx.x < n ? l.Un[threadIdx.x] : make_float2(0.f, 0.f);
    __syncthreads();
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= G) return;
    float out = 0.0f;
    if (l.status[0]) {
        const float* Y = l.Ygrid + d;
        const int nS = l.nSources < n / 2 ? l.nSources : n / 2;
        if (l.mode == 4 || l.mode == 5) {
            float tmp = 0.0f;
            for (int j = nS; j < n; j++) {
                float2 s = make_float2(0.f, 0.f);
                for (int i = 0; i < n; i++) { const float y = Y[(long long)i * G]; const float2 v = s_V[i * 64 + j]; s.x += v.x * y; s.y += v.y * y; }
                tmp += s.x * s.x + s.y * s.y;
            }
            out = l.mode == 5 ? logf(1.0f / (tmp + 2.23e-10f)) : 1.0f / (tmp + 2.23e-10f);
        } else {
            float2 s = make_float2(0.f, 0.f);
            for (int i = 0; i < n; i++) { const float y = Y[(long long)i * G]; const float2 u = s_Un[i]; s.x += u.x * y; s.y -= u.y * y; }
            const float m = powf(hypotf(s.x, s.y), 2.0f) + 2.23e-9f;
            out = l.mode == 7 ? logf(1.0f / m) : 1.0f / m;
        }
    }
    const float v = (1.0f - l.avg) * out + l.avg * l.prev_pmap[d];
    l.pmap[d] = v;
    l.prev_pmap[d] = v;
}

void launch_adaptive_map(const AdaptMapLaunch& l)
{
    AdArgs a; a.l = l;
    static bool raised = false;
    if (!raised) {
        /* dynamic LDS beyond the 64 KB default; static __shared__ of each kernel comes on top and the sum must stay below 160 KB */
        HIP_CHECK(hipFuncSetAttribute((const void*)chol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(zc) * 64 * 65));
        HIP_CHECK(hipFuncSetAttribute((const void*)herm_eig_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(zc) * 2 * 64 * 65));
        HIP_CHECK(hipFuncSetAttribute((const void*)mvdr_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(zc) * (64 * 64 + 64 * 64)));
        HIP_CHECK(hipFuncSetAttribute((const void*)mvdr_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(zc) * (64 * 64 + 2 * 64 * 32)));
        raised = true;
    }
    const int n = l.nM;
    KernelTimer kt("adaptive_map");
    hipLaunchKernelGGL(cgrp_cplx_kernel, dim3(64), dim3(256), 0, stream(), a);
    if (l.mode == 2 || l.mode == 3) {
        hipLaunchKernelGGL(chol_kernel, dim3(1), dim3(256), sizeof(zc) * 64 * 65, stream(), a);
        if (l.mode == 2)
            hipLaunchKernelGGL((mvdr_kernel<64, false>), dim3((l.G + 63) / 64), dim3(64), sizeof(zc) * (n * n + n * 64), stream(), a);
        else
            hipLaunchKernelGGL((mvdr_kernel<32, true>), dim3((l.G + 31) / 32), dim3(32), sizeof(zc) * (n * n + 2 * n * 32), stream(), a);
    } else {
        hipLaunchKernelGGL(herm_eig_kernel, dim3(1), dim3(256), sizeof(zc) * 2 * 64 * 65, stream(), a);
        hipLaunchKernelGGL(subspace_kernel, dim3((l.G + 255) / 256), dim3(256), 0, stream(), a);
    }
    HIP_CHECK(hipGetLastError());
}

void launch_subspace_map(const AdaptMapLaunch& l)
{
    AdArgs a; a.l = l;
    KernelTimer kt("adaptive_map");
    hipLaunchKernelGGL(subspace_kernel, dim3((l.G + 255) / 256), dim3(256), 0, stream(), a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace saf
