/*
 * orc_core.c — oracle: data tables, real FFT, afSTFT filterbank.
 * TEST INFRASTRUCTURE ONLY (see saf_oracle.h).  Reference paths are relative to
 * /root/reference.
 */
#include "saf_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>

/* ========================================================================== */
/*                                 tables                                     */
/* ========================================================================== */

typedef struct { char name[64]; int d0, d1; float* data; } orc_tab;
static orc_tab* g_tabs = NULL;
static int g_ntabs = 0;

int orc_tables_load(const char* path)
{
    if (g_tabs) return 0;
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    char magic[4]; unsigned ver, n;
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "SAFT", 4)) { fclose(f); return -2; }
    if (fread(&ver, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) { fclose(f); return -2; }
    orc_tab* t = (orc_tab*)calloc(n, sizeof(orc_tab));
    for (unsigned i = 0; i < n; i++) {
        unsigned nl, d0, d1;
        if (fread(&nl, 4, 1, f) != 1 || nl >= 64) { fclose(f); return -3; }
        if (fread(t[i].name, 1, nl, f) != nl) { fclose(f); return -3; }
        t[i].name[nl] = 0;
        if (fread(&d0, 4, 1, f) != 1 || fread(&d1, 4, 1, f) != 1) { fclose(f); return -3; }
        t[i].d0 = (int)d0; t[i].d1 = (int)d1;
        t[i].data = (float*)malloc(sizeof(float) * d0 * d1);
        if (fread(t[i].data, sizeof(float), (size_t)d0 * d1, f) != (size_t)d0 * d1) { fclose(f); return -3; }
    }
    fclose(f);
    g_tabs = t; g_ntabs = (int)n;
    return 0;
}

const float* orc_table(const char* name, int* d0, int* d1)
{
    for (int i = 0; i < g_ntabs; i++)
        if (!strcmp(g_tabs[i].name, name)) {
            if (d0) *d0 = g_tabs[i].d0;
            if (d1) *d1 = g_tabs[i].d1;
            return g_tabs[i].data;
        }
    return NULL;
}

/* ========================================================================== */
/*   real FFT with saf_rfft semantics (saf_utility_fft.c:531-753):            */
/*   forward unscaled, N/2+1 bins; backward scaled by 1/N; N even.            */
/*   The reference's default backend packs N reals as N/2 complex points      */
/*   (kiss_fftr.c:69-161); the same packing is used here, with an own         */
/*   decimation-in-time mixed-radix complex FFT underneath.                   */
/* ========================================================================== */

typedef struct {
    int N, M;            /* real length, complex length N/2 */
    int nfac, fac[32];
    orc_cpx* tw;         /* W_M^k, k<M */
    orc_cpx* rtw;        /* exp(-2 pi i k / N), k<=M/2+... (k<M) */
    orc_cpx* z;          /* work M */
    orc_cpx* zo;         /* work M */
} orc_rfft;

static void factorise(int n, int* fac, int* nfac)
{
    int k = 0;
    while (n % 4 == 0) { fac[k++] = 4; n /= 4; }
    while (n % 2 == 0) { fac[k++] = 2; n /= 2; }
    for (int p = 3; p * p <= n; p += 2)
        while (n % p == 0) { fac[k++] = p; n /= p; }
    if (n > 1) fac[k++] = n;
    *nfac = k;
}

/* out[0..n) = DFT_n of in[0], in[stride], ... ; sign=-1 forward, +1 inverse.
 * tw holds W_M^k for the TOP-level size M; twstride = M/n. */
static void cfft_rec(const orc_cpx* in, orc_cpx* out, int n, int stride, const int* fac,
                     const orc_cpx* tw, int twstride, int M, int sign)
{
    const int p = fac[0];
    const int m = n / p;
    if (m == 1) {
        for (int q = 0; q < p; q++) out[q] = in[q * stride];
    } else {
        for (int q = 0; q < p; q++)
            cfft_rec(in + q * stride, out + q * m, m, stride * p, fac + 1, tw, twstride * p, M, sign);
    }
    /* butterflies */
    if (p == 2) {
        for (int k = 0; k < m; k++) {
            orc_cpx w = tw[(k * twstride) % M];
            if (sign > 0) w.im = -w.im;
            orc_cpx a = out[k], b = out[k + m], t;
            t.re = b.re * w.re - b.im * w.im;
            t.im = b.re * w.im + b.im * w.re;
            out[k].re = a.re + t.re; out[k].im = a.im + t.im;
            out[k + m].re = a.re - t.re; out[k + m].im = a.im - t.im;
        }
    } else if (p == 4) {
        for (int k = 0; k < m; k++) {
            orc_cpx x[4];
            x[0] = out[k];
            for (int q = 1; q < 4; q++) {
                orc_cpx w = tw[(q * k * twstride) % M];
                if (sign > 0) w.im = -w.im;
                orc_cpx b = out[k + q * m];
                x[q].re = b.re * w.re - b.im * w.im;
                x[q].im = b.re * w.im + b.im * w.re;
            }
            orc_cpx s0 = { x[0].re + x[2].re, x[0].im + x[2].im };
            orc_cpx s1 = { x[0].re - x[2].re, x[0].im - x[2].im };
            orc_cpx s2 = { x[1].re + x[3].re, x[1].im + x[3].im };
            orc_cpx s3 = { x[1].re - x[3].re, x[1].im - x[3].im };
            /* forward: X1 = s1 - i s3 ; inverse: X1 = s1 + i s3 */
            out[k].re = s0.re + s2.re;          out[k].im = s0.im + s2.im;
            out[k + 2 * m].re = s0.re - s2.re;  out[k + 2 * m].im = s0.im - s2.im;
            if (sign < 0) {
                out[k + m].re = s1.re + s3.im;      out[k + m].im = s1.im - s3.re;
                out[k + 3 * m].re = s1.re - s3.im;  out[k + 3 * m].im = s1.im + s3.re;
            } else {
                out[k + m].re = s1.re - s3.im;      out[k + m].im = s1.im + s3.re;
                out[k + 3 * m].re = s1.re + s3.im;  out[k + 3 * m].im = s1.im - s3.re;
            }
        }
    } else {
        orc_cpx* x = (orc_cpx*)malloc(sizeof(orc_cpx) * p);
        for (int k = 0; k < m; k++) {
            for (int q = 0; q < p; q++) {
                orc_cpx w = tw[(q * k * twstride) % M];
                if (sign > 0) w.im = -w.im;
                orc_cpx b = out[k + q * m];
                x[q].re = b.re * w.re - b.im * w.im;
                x[q].im = b.re * w.im + b.im * w.re;
            }
            for (int r = 0; r < p; r++) {
                float sr = 0.f, si = 0.f;
                for (int q = 0; q < p; q++) {
                    orc_cpx w = tw[(int)(((long long)q * r * m * twstride) % M)];
                    if (sign > 0) w.im = -w.im;
                    sr += x[q].re * w.re - x[q].im * w.im;
                    si += x[q].re * w.im + x[q].im * w.re;
                }
                out[k + r * m].re = sr; out[k + r * m].im = si;
            }
        }
        free(x);
    }
}

void orc_rfft_create(void** ph, int N)
{
    assert(N % 2 == 0);
    orc_rfft* h = (orc_rfft*)calloc(1, sizeof(orc_rfft));
    h->N = N; h->M = N / 2;
    factorise(h->M, h->fac, &h->nfac);
    if (h->nfac == 0) { h->fac[0] = 1; h->nfac = 1; }
    h->tw = (orc_cpx*)malloc(sizeof(orc_cpx) * h->M);
    h->rtw = (orc_cpx*)malloc(sizeof(orc_cpx) * (h->M + 1));
    h->z = (orc_cpx*)malloc(sizeof(orc_cpx) * h->M);
    h->zo = (orc_cpx*)malloc(sizeof(orc_cpx) * h->M);
    for (int k = 0; k < h->M; k++) {
        double a = -2.0 * M_PI * (double)k / (double)h->M;
        h->tw[k].re = (float)cos(a); h->tw[k].im = (float)sin(a);
    }
    for (int k = 0; k <= h->M; k++) {
        double a = -2.0 * M_PI * (double)k / (double)N;
        h->rtw[k].re = (float)cos(a); h->rtw[k].im = (float)sin(a);
    }
    *ph = h;
}

void orc_rfft_destroy(void** ph)
{
    orc_rfft* h = (orc_rfft*)*ph;
    if (!h) return;
    free(h->tw); free(h->rtw); free(h->z); free(h->zo); free(h);
    *ph = NULL;
}

void orc_rfft_forward(void* hh, const float* in, orc_cpx* out)
{
    orc_rfft* h = (orc_rfft*)hh;
    const int M = h->M;
    if (M == 1) { out[0].re = in[0] + in[1]; out[0].im = 0; out[1].re = in[0] - in[1]; out[1].im = 0; return; }
    /* z[n] = x[2n] + i x[2n+1]  (kiss_fftr.c:69-84) */
    cfft_rec((const orc_cpx*)in, h->zo, M, 1, h->fac, h->tw, 1, M, -1);
    const orc_cpx* Z = h->zo;
    /* X[k] = (Z[k] + conj(Z[M-k]))/2 - i/2 e^{-2 pi i k/N} (Z[k] - conj(Z[M-k]))  (kiss_fftr.c:86-123) */
    out[0].re = Z[0].re + Z[0].im; out[0].im = 0.f;
    out[M].re = Z[0].re - Z[0].im; out[M].im = 0.f;
    for (int k = 1; k <= M / 2; k++) {
        orc_cpx a = Z[k], b = { Z[M - k].re, -Z[M - k].im };
        orc_cpx f1 = { a.re + b.re, a.im + b.im };
        orc_cpx f2 = { a.re - b.re, a.im - b.im };
        orc_cpx w = h->rtw[k];
        orc_cpx t = { f2.re * w.re - f2.im * w.im, f2.re * w.im + f2.im * w.re };
        out[k].re = 0.5f * (f1.re + t.im);
        out[k].im = 0.5f * (f1.im - t.re);
        out[M - k].re = 0.5f * (f1.re - t.im);
        out[M - k].im = 0.5f * (-f1.im - t.re);
    }
}

void orc_rfft_backward(void* hh, const orc_cpx* in, float* out)
{
    orc_rfft* h = (orc_rfft*)hh;
    const int M = h->M, N = h->N;
    if (M == 1) { out[0] = 0.5f * (in[0].re + in[1].re); out[1] = 0.5f * (in[0].re - in[1].re); return; }
    orc_cpx* Z = h->z;
    /* imaginary parts of DC and Nyquist are ignored (kiss_fftr.c:125-161) */
    Z[0].re = in[0].re + in[M].re;
    Z[0].im = in[0].re - in[M].re;
    for (int k = 1; k <= M / 2; k++) {
        orc_cpx fk = in[k], fnkc = { in[M - k].re, -in[M - k].im };
        orc_cpx fek = { fk.re + fnkc.re, fk.im + fnkc.im };
        orc_cpx tmp = { fk.re - fnkc.re, fk.im - fnkc.im };
        orc_cpx w = { h->rtw[k].re, -h->rtw[k].im };   /* e^{+2 pi i k/N} */
        orc_cpx fok = { tmp.re * w.re - tmp.im * w.im, tmp.re * w.im + tmp.im * w.re };
        /* Z[k] = fek + i*fok ; Z[M-k] = conj(fek - i*fok) */
        Z[k].re = fek.re - fok.im;  Z[k].im = fek.im + fok.re;
        Z[M - k].re = fek.re + fok.im; Z[M - k].im = -(fek.im - fok.re);
    }
    cfft_rec(Z, (orc_cpx*)out, M, 1, h->fac, h->tw, 1, M, +1);
    const float sc = 1.0f / (float)N;   /* saf_utility_fft.c:751 */
    for (int i = 0; i < N; i++) out[i] *= sc;
}

/* ========================================================================== */
/*                                 afSTFT                                     */
/* ========================================================================== */

#define COEFF1 0.031273141818515176604f   /* afSTFT_internal.h:74 */
#define COEFF2 0.28127313041521179171f    /* afSTFT_internal.h:75 */

typedef struct {
    /* afSTFTlib_internal_data (afSTFT_internal.h:93-123) */
    int inChannels, outChannels, hopSize, hLen, LDmode, hopIndexIn, hopIndexOut, totalHops;
    float *protoFilter, *protoFilterI;
    float **inBuffer, **outBuffer;
    float *fftProcessFrameTD, *tempHopBuffer;
    orc_cpx *fftProcessFrameFD;
    void* hFFT;
    int hybridMode;
    /* afHybrid (afSTFT_internal.h:128-135): [ch][7 slots] x {re[hop+1], im[hop+1]} */
    float** hybRe; float** hybIm; int hybInCh; int loopPointer;
    /* afSTFT_data (afSTFTlib.c:122-135) */
    int nBands, format, afSTFTdelay;
    float **inRe, **inIm, **outRe, **outIm;   /* STFTInput/OutputFrameTF */
    float** tempHopFrameTD; int tempHopCh;
} orc_afstft;

static float** alloc2(int n, int len) {
    float** p = (float**)malloc(sizeof(float*) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) p[i] = (float*)calloc(len, sizeof(float));
    return p;
}
static void free2(float** p, int n) { if (!p) return; for (int i = 0; i < n; i++) free(p[i]); free(p); }
static float** resize2(float** p, int oldn, int newn, int len) {
    /* surviving rows keep their contents, new rows start zeroed (afSTFT_internal.c:158-211) */
    for (int i = newn; i < oldn; i++) free(p[i]);
    p = (float**)realloc(p, sizeof(float*) * (newn > 0 ? newn : 1));
    for (int i = oldn; i < newn; i++) p[i] = (float*)calloc(len, sizeof(float));
    return p;
}

/* afSTFT_create (afSTFTlib.c:142-196) + afSTFTlib_init (afSTFT_internal.c:57-156) */
void orc_afSTFT_create(void** ph, int nCHin, int nCHout, int hopsize, int lowDelayMode, int hybridmode, int format)
{
    orc_afstft* h = (orc_afstft*)calloc(1, sizeof(orc_afstft));
    if (hybridmode) assert(hopsize == 64 || hopsize == 128 || hopsize == 256);
    assert(1024 % hopsize == 0);
    h->inChannels = nCHin; h->outChannels = nCHout; h->hopSize = hopsize;
    const int dsFactor = 1024 / hopsize;
    h->hLen = 10240 / dsFactor;
    h->totalHops = 10;
    h->LDmode = lowDelayMode;
    h->hybridMode = hybridmode;
    h->nBands = hybridmode ? hopsize + 5 : hopsize + 1;
    if (lowDelayMode) h->afSTFTdelay = hybridmode ? 7 * hopsize : 4 * hopsize;
    else              h->afSTFTdelay = hybridmode ? 12 * hopsize : 9 * hopsize;
    h->format = format;
    h->protoFilter = (float*)malloc(sizeof(float) * h->hLen);
    h->protoFilterI = (float*)malloc(sizeof(float) * h->hLen);
    h->fftProcessFrameTD = (float*)calloc(hopsize * 2, sizeof(float));
    h->fftProcessFrameFD = (orc_cpx*)calloc(hopsize + 1, sizeof(orc_cpx));
    h->tempHopBuffer = (float*)malloc(sizeof(float) * hopsize);
    orc_rfft_create(&h->hFFT, hopsize * 2);
    int d0 = 0, d1 = 0;
    if (!lowDelayMode) {
        const float* p = orc_table("afSTFT_protoFilter1024", &d0, &d1);
        assert(p && d0 * d1 == 10240);
        const float eq = 2.0f / sqrtf(5.487604141f);             /* afSTFT_internal.c:125 */
        for (int k = 0; k < h->hLen; k++) {
            h->protoFilter[h->hLen - k - 1] = p[k * dsFactor] * eq;
            h->protoFilterI[h->hLen - k - 1] = p[k * dsFactor] * eq;
        }
    } else {
        const float* p = orc_table("afSTFT_protoFilter1024LD", &d0, &d1);
        assert(p && d0 * d1 == 10240);
        const float eq = 2.0f / sqrtf(4.544559956f);             /* afSTFT_internal.c:137 */
        for (int k = 0; k < h->hLen; k++) {
            h->protoFilter[h->hLen - k - 1] = p[k * dsFactor] * eq;
            h->protoFilterI[k] = p[k * dsFactor] * eq;
        }
    }
    h->inBuffer = alloc2(nCHin, h->hLen);
    h->outBuffer = alloc2(nCHout, h->hLen);
    if (hybridmode) {
        h->hybInCh = nCHin;
        h->hybRe = alloc2(nCHin, 7 * (hopsize + 1));
        h->hybIm = alloc2(nCHin, 7 * (hopsize + 1));
    }
    h->inRe = alloc2(nCHin, h->nBands);  h->inIm = alloc2(nCHin, h->nBands);
    h->outRe = alloc2(nCHout, h->nBands); h->outIm = alloc2(nCHout, h->nBands);
    h->tempHopCh = nCHin > nCHout ? nCHin : nCHout;
    h->tempHopFrameTD = alloc2(h->tempHopCh, hopsize);
    *ph = h;
}

void orc_afSTFT_destroy(void** ph)
{
    orc_afstft* h = (orc_afstft*)*ph;
    if (!h) return;
    free(h->protoFilter); free(h->protoFilterI); free(h->fftProcessFrameTD); free(h->fftProcessFrameFD);
    free(h->tempHopBuffer); orc_rfft_destroy(&h->hFFT);
    free2(h->inBuffer, h->inChannels); free2(h->outBuffer, h->outChannels);
    if (h->hybridMode) { free2(h->hybRe, h->hybInCh); free2(h->hybIm, h->hybInCh); }
    free2(h->inRe, h->inChannels); free2(h->inIm, h->inChannels);
    free2(h->outRe, h->outChannels); free2(h->outIm, h->outChannels);
    free2(h->tempHopFrameTD, h->tempHopCh);
    free(h); *ph = NULL;
}

/* afHybridForward (afSTFT_internal.c:523-623) */
static void hybrid_forward(orc_afstft* h)
{
    const int hs = h->hopSize, L = hs + 1;
    h->loopPointer++;
    if (h->loopPointer == 7) h->loopPointer = 0;
    for (int ch = 0; ch < h->inChannels; ch++) {
        float* re = h->inRe[ch]; float* im = h->inIm[ch];
        float* bre = h->hybRe[ch]; float* bim = h->hybIm[ch];
        memcpy(bre + h->loopPointer * L, re, sizeof(float) * L);
        memcpy(bim + h->loopPointer * L, im, sizeof(float) * L);
        int lp = h->loopPointer - 3; if (lp < 0) lp += 7;
        for (int ri = 0; ri < 2; ri++) {
            float* pr1 = ri ? im : re;
            const float* pr2 = (ri ? bim : bre) + lp * L;
            pr1[0] = pr2[0];
            pr1[1] = pr2[1] * 0.5f; pr1[2] = pr1[1];
            pr1[3] = pr2[2] * 0.5f; pr1[4] = pr1[3];
            pr1[5] = pr2[3] * 0.5f; pr1[6] = pr1[5];
            pr1[7] = pr2[4] * 0.5f; pr1[8] = pr1[7];
            memcpy(pr1 + 9, pr2 + 5, sizeof(float) * (hs - 4));
        }
        int si[7];
        for (int s = 0; s < 7; s++) { si[s] = h->loopPointer + 1 + s; if (si[s] > 6) si[s] -= 7; }
        for (int band = 1; band < 5; band++) {
            float r, i;
            r = -COEFF1 * bim[si[6] * L + band];
            i =  COEFF1 * bre[si[6] * L + band];
            r -= COEFF2 * bim[si[4] * L + band];
            i += COEFF2 * bre[si[4] * L + band];
            r += COEFF2 * bim[si[2] * L + band];
            i -= COEFF2 * bre[si[2] * L + band];
            r += COEFF1 * bim[si[0] * L + band];
            i -= COEFF1 * bre[si[0] * L + band];
            if (band == 1 || band == 3) {
                re[band * 2 - 1] -= r; im[band * 2 - 1] -= i;
                re[band * 2] += r;     im[band * 2] += i;
            } else {
                re[band * 2 - 1] += r; im[band * 2 - 1] += i;
                re[band * 2] -= r;     im[band * 2] -= i;
            }
        }
    }
}

/* afSTFTlib_forward (afSTFT_internal.c:237-333): one hop for all input channels */
static void core_forward(orc_afstft* h, float** inTD)
{
    const int hs = h->hopSize;
    for (int ch = 0; ch < h->inChannels; ch++) {
        int hopIndex_this2 = h->hopIndexIn;
        memcpy(&h->inBuffer[ch][hopIndex_this2 * hs], inTD[ch], sizeof(float) * hs);
        hopIndex_this2++;
        if (hopIndex_this2 >= h->totalHops) hopIndex_this2 = 0;
        memset(h->fftProcessFrameTD, 0, sizeof(float) * hs * 2);
        int lr = 0, hopIndex_this = hopIndex_this2;
        for (int k = 0; k < h->totalHops; k++) {
            const float* p1 = &h->inBuffer[ch][hs * hopIndex_this];
            const float* p2 = &h->protoFilter[k * hs];
            float* p3;
            if (lr == 1) { p3 = &h->fftProcessFrameTD[hs]; lr = 0; }
            else         { p3 = &h->fftProcessFrameTD[0];  lr = 1; }
            /* utility_svvmul then cblas_saxpy: product rounded, then added (afSTFT_internal.c:291-292) */
            for (int n = 0; n < hs; n++) h->tempHopBuffer[n] = p1[n] * p2[n];
            for (int n = 0; n < hs; n++) p3[n] += h->tempHopBuffer[n];
            hopIndex_this++;
            if (hopIndex_this >= h->totalHops) hopIndex_this = 0;
        }
        orc_rfft_forward(h->hFFT, h->fftProcessFrameTD, h->fftProcessFrameFD);
        for (int b = 0; b <= hs; b++) { h->inRe[ch][b] = h->fftProcessFrameFD[b].re; h->inIm[ch][b] = h->fftProcessFrameFD[b].im; }
    }
    h->hopIndexIn++;
    if (h->hopIndexIn >= h->totalHops) h->hopIndexIn = 0;
    if (h->hybridMode) hybrid_forward(h);
}

/* afHybridInverse (afSTFT_internal.c:625-653) + afSTFTlib_inverse (:335-453) */
static void core_inverse(orc_afstft* h, float** outTD)
{
    const int hs = h->hopSize;
    if (h->hybridMode) {
        for (int ch = 0; ch < h->outChannels; ch++)
            for (int ri = 0; ri < 2; ri++) {
                float* pr = ri ? h->outIm[ch] : h->outRe[ch];
                pr[1] = pr[1] + pr[2];
                pr[2] = pr[3] + pr[4];
                pr[3] = pr[5] + pr[6];
                pr[4] = pr[7] + pr[8];
                memmove(pr + 5, pr + 9, sizeof(float) * (hs - 4));
            }
    }
    for (int ch = 0; ch < h->outChannels; ch++) {
        int hopIndex_this2 = h->hopIndexOut;
        for (int b = 0; b <= hs; b++) { h->fftProcessFrameFD[b].re = h->outRe[ch][b]; h->fftProcessFrameFD[b].im = h->outIm[ch][b]; }
        if (h->LDmode == 1)
            for (int k = 1; k < hs; k += 2) { h->fftProcessFrameFD[k].re = -h->fftProcessFrameFD[k].re; h->fftProcessFrameFD[k].im = -h->fftProcessFrameFD[k].im; }
        orc_rfft_backward(h->hFFT, h->fftProcessFrameFD, h->fftProcessFrameTD);
        memset(&h->outBuffer[ch][hopIndex_this2 * hs], 0, sizeof(float) * hs);
        hopIndex_this2++;
        if (hopIndex_this2 >= h->totalHops) hopIndex_this2 = 0;
        int hopIndex_this = hopIndex_this2, lr = 0;
        for (int k = 0; k < h->totalHops; k++) {
            float* p1 = &h->outBuffer[ch][hs * hopIndex_this];
            const float* p2 = &h->protoFilterI[k * hs];
            const float* p3;
            if (lr == 1) { p3 = &h->fftProcessFrameTD[hs]; lr = 0; }
            else         { p3 = &h->fftProcessFrameTD[0];  lr = 1; }
            for (int n = 0; n < hs; n++) h->tempHopBuffer[n] = p2[n] * p3[n];
            for (int n = 0; n < hs; n++) p1[n] += h->tempHopBuffer[n];
            hopIndex_this++;
            if (hopIndex_this >= h->totalHops) hopIndex_this = 0;
        }
        memcpy(outTD[ch], &h->outBuffer[ch][hs * hopIndex_this], sizeof(float) * hs);
    }
    h->hopIndexOut++;
    if (h->hopIndexOut >= h->totalHops) h->hopIndexOut = 0;
}

/* afSTFT_forward_knownDimensions (afSTFTlib.c:267-308) */
void orc_afSTFT_forward_knownDimensions(void* hh, const float* dataTD, int framesize, int dataFD_nCH, int dataFD_nHops, orc_cpx* dataFD)
{
    orc_afstft* h = (orc_afstft*)hh;
    assert(framesize % h->hopSize == 0);
    const int nHops = framesize / h->hopSize;
    for (int t = 0; t < nHops; t++) {
        for (int ch = 0; ch < h->inChannels; ch++)
            memcpy(h->tempHopFrameTD[ch], &dataTD[(size_t)ch * framesize + t * h->hopSize], sizeof(float) * h->hopSize);
        core_forward(h, h->tempHopFrameTD);
        if (h->format == ORC_AFSTFT_BANDS_CH_TIME) {
            for (int ch = 0; ch < h->inChannels; ch++)
                for (int b = 0; b < h->nBands; b++) {
                    orc_cpx* d = &dataFD[(size_t)b * dataFD_nCH * dataFD_nHops + (size_t)ch * dataFD_nHops + t];
                    d->re = h->inRe[ch][b]; d->im = h->inIm[ch][b];
                }
        } else {   /* [t][ch][band] */
            for (int ch = 0; ch < h->inChannels; ch++)
                for (int b = 0; b < h->nBands; b++) {
                    orc_cpx* d = &dataFD[((size_t)t * dataFD_nCH + ch) * h->nBands + b];
                    d->re = h->inRe[ch][b]; d->im = h->inIm[ch][b];
                }
        }
    }
}

/* afSTFT_backward_knownDimensions (afSTFTlib.c:390-431) */
void orc_afSTFT_backward_knownDimensions(void* hh, const orc_cpx* dataFD, int framesize, int dataFD_nCH, int dataFD_nHops, float* dataTD)
{
    orc_afstft* h = (orc_afstft*)hh;
    assert(framesize % h->hopSize == 0);
    const int nHops = framesize / h->hopSize;
    for (int t = 0; t < nHops; t++) {
        for (int ch = 0; ch < h->outChannels; ch++)
            for (int b = 0; b < h->nBands; b++) {
                const orc_cpx* d = (h->format == ORC_AFSTFT_BANDS_CH_TIME)
                    ? &dataFD[(size_t)b * dataFD_nCH * dataFD_nHops + (size_t)ch * dataFD_nHops + t]
                    : &dataFD[((size_t)t * dataFD_nCH + ch) * h->nBands + b];
                h->outRe[ch][b] = d->re; h->outIm[ch][b] = d->im;
            }
        core_inverse(h, h->tempHopFrameTD);
        for (int ch = 0; ch < h->outChannels; ch++)
            memcpy(&dataTD[(size_t)ch * framesize + t * h->hopSize], h->tempHopFrameTD[ch], sizeof(float) * h->hopSize);
    }
}

/* afSTFT_channelChange (afSTFTlib.c:476-516) + afSTFTlib_channelChange (afSTFT_internal.c:158-211) */
void orc_afSTFT_channelChange(void* hh, int new_in, int new_out)
{
    orc_afstft* h = (orc_afstft*)hh;
    h->inBuffer = resize2(h->inBuffer, h->inChannels, new_in, h->hLen);
    h->outBuffer = resize2(h->outBuffer, h->outChannels, new_out, h->hLen);
    if (h->hybridMode) {
        h->hybRe = resize2(h->hybRe, h->hybInCh, new_in, 7 * (h->hopSize + 1));
        h->hybIm = resize2(h->hybIm, h->hybInCh, new_in, 7 * (h->hopSize + 1));
        h->hybInCh = new_in;
    }
    h->inRe = resize2(h->inRe, h->inChannels, new_in, h->nBands);
    h->inIm = resize2(h->inIm, h->inChannels, new_in, h->nBands);
    h->outRe = resize2(h->outRe, h->outChannels, new_out, h->nBands);
    h->outIm = resize2(h->outIm, h->outChannels, new_out, h->nBands);
    int newTemp = new_in > new_out ? new_in : new_out;
    h->tempHopFrameTD = resize2(h->tempHopFrameTD, h->tempHopCh, newTemp, h->hopSize);
    h->tempHopCh = newTemp;
    h->inChannels = new_in; h->outChannels = new_out;
}

/* afSTFTlib_clearBuffers (afSTFT_internal.c:213-235) */
void orc_afSTFT_clearBuffers(void* hh)
{
    orc_afstft* h = (orc_afstft*)hh;
    for (int i = 0; i < h->inChannels; i++) memset(h->inBuffer[i], 0, sizeof(float) * h->hLen);
    for (int i = 0; i < h->outChannels; i++) memset(h->outBuffer[i], 0, sizeof(float) * h->hLen);
    if (h->hybridMode)
        for (int ch = 0; ch < h->hybInCh; ch++) {
            memset(h->hybRe[ch], 0, sizeof(float) * 7 * (h->hopSize + 1));
            memset(h->hybIm[ch], 0, sizeof(float) * 7 * (h->hopSize + 1));
        }
}

int orc_afSTFT_getNBands(void* hh) { return ((orc_afstft*)hh)->nBands; }
int orc_afSTFT_getProcDelay(void* hh) { return ((orc_afstft*)hh)->afSTFTdelay; }

/* afSTFT_getCentreFreqs (afSTFTlib.c:545-590).  With a NULL handle the reference
 * returns one of two measured 133-entry tables (afSTFTlib.c:54-59); they are data
 * and come from the tables blob ("afCenterFreq48e3" / "afCenterFreq44100"). */
static const float stft2hyb[9] = { 1.0f, 0.7501f, 1.2499f, 0.8751f, 1.1249f, 0.9167f, 1.0833f, 0.9375f, 1.0625f };
static const int   stft2hybBin[9] = { 0, 1, 1, 2, 2, 3, 3, 4, 4 };

void orc_afSTFT_getCentreFreqs(void* hh, float fs, int nBands, float* freqVector)
{
    extern const float* orc_centre_table(int is441);
    if (hh == NULL) {
        assert(nBands >= 133);
        const float* tab = orc_centre_table(fs == 44100.0f);
        for (int b = 0; b < nBands; b++) freqVector[b] = tab[b];
        return;
    }
    orc_afstft* h = (orc_afstft*)hh;
    assert(nBands >= h->nBands);
    /* getUniformFreqVector(fftSize, fs): k * fs / fftSize  (saf_utility_fft.c) */
    const int fftSize = h->hopSize * 2;
    if (h->hybridMode) {
        for (int i = 0; i < 9; i++)
            freqVector[i] = stft2hyb[i] * ((float)stft2hybBin[i] * fs / (float)fftSize);
        for (int i = 9, j = 5; i < h->nBands; i++, j++)
            freqVector[i] = (float)j * fs / (float)fftSize;
    } else {
        for (int k = 0; k <= h->hopSize; k++) freqVector[k] = (float)k * fs / (float)fftSize;
    }
}

const float* orc_centre_table(int is441)
{
    int d0, d1;
    const float* t = orc_table(is441 ? "afCenterFreq44100" : "afCenterFreq48e3", &d0, &d1);
    assert(t && d0 * d1 == 133);
    return t;
}

/* afAnalyse (afSTFTlib.c:78-119): out [nBands][nTimeSlots][nCH] */
static void af_analyse(const float* inTD /* nSamples x nCH */, int nSamplesTD, int nCH, int hopSize, int LDmode, int hybridmode, orc_cpx* outTF)
{
    const int nBands = hopSize + (hybridmode ? 5 : 1);
    const int nTimeSlots = (int)((float)nSamplesTD / (float)hopSize + 0.9999f);
    void* hSTFT;
    orc_afSTFT_create(&hSTFT, nCH, 1, hopSize, LDmode, hybridmode, ORC_AFSTFT_TIME_CH_BANDS);
    float* tmp = (float*)calloc((size_t)nCH * nTimeSlots * hopSize, sizeof(float));
    orc_cpx* FrameTF = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nTimeSlots * nCH * nBands);
    for (int ch = 0; ch < nCH; ch++)
        for (int s = 0; s < nSamplesTD; s++)
            tmp[(size_t)ch * nTimeSlots * hopSize + s] = inTD[(size_t)s * nCH + ch];
    orc_afSTFT_forward_knownDimensions(hSTFT, tmp, nTimeSlots * hopSize, nCH, nTimeSlots, FrameTF);
    for (int band = 0; band < nBands; band++)
        for (int t = 0; t < nTimeSlots; t++)
            for (int ch = 0; ch < nCH; ch++)
                outTF[((size_t)band * nTimeSlots + t) * nCH + ch] = FrameTF[((size_t)t * nCH + ch) * nBands + band];
    orc_afSTFT_destroy(&hSTFT);
    free(tmp); free(FrameTF);
}

/* afSTFT_FIRtoFilterbankCoeffs (afSTFTlib.c:592-674) */
void orc_afSTFT_FIRtoFilterbankCoeffs(const float* hIR, int N_dirs, int nCH, int ir_len, int hopSize, int LDmode, int hybridmode, orc_cpx* hFB)
{
    const int nBands = hopSize + (hybridmode ? 5 : 1);
    const int ir_pad = 1024;
    const int maxlen = (ir_len > hopSize ? ir_len : hopSize) + ir_pad;
    const int nTimeSlots = (int)((float)maxlen / (float)hopSize + 0.9999f);
    int* maxIdx = (int*)calloc(nCH, sizeof(int));
    float* centerImpulse = (float*)calloc(maxlen, sizeof(float));
    for (int j = 0; j < nCH; j++) {
        float maxVal = 2.23e-13f;
        for (int i = 0; i < ir_len; i++)
            if (hIR[j * ir_len + i] > maxVal) { maxVal = hIR[j * ir_len + i]; maxIdx[j] = i; }
    }
    float idxDel = 0.0f;
    for (int j = 0; j < nCH; j++) idxDel += (float)maxIdx[j];
    idxDel /= (float)nCH;
    idxDel = idxDel + 1.5f;
    centerImpulse[(int)idxDel] = 1.0f;
    orc_cpx* cFB = (orc_cpx*)malloc(sizeof(orc_cpx) * (size_t)nBands * nTimeSlots);
    af_analyse(centerImpulse, maxlen, 1, hopSize, LDmode, hybridmode, cFB);
    float* cE = (float*)calloc(nBands, sizeof(float));
    for (int i = 0; i < nBands; i++)
        for (int t = 0; t < nTimeSlots; t++) {
            float a = hypotf(cFB[i * nTimeSlots + t].re, cFB[i * nTimeSlots + t].im);
            cE[i] += powf(a, 2.0f);
        }
    float* ir = (float*)calloc((size_t)maxlen * nCH, sizeof(float));
    orc_cpx* irFB = (orc_cpx*)calloc((size_t)nBands * nTimeSlots * nCH, sizeof(orc_cpx));
    for (int nd = 0; nd < N_dirs; nd++) {
        for (int j = 0; j < ir_len; j++)
            for (int i = 0; i < nCH; i++)
                ir[(size_t)j * nCH + i] = hIR[(size_t)nd * nCH * ir_len + (size_t)i * ir_len + j];
        af_analyse(ir, maxlen, nCH, hopSize, LDmode, hybridmode, irFB);
        for (int nm = 0; nm < nCH; nm++)
            for (int i = 0; i < nBands; i++) {
                float e = 0.f;
                for (int t = 0; t < nTimeSlots; t++) {
                    orc_cpx v = irFB[((size_t)i * nTimeSlots + t) * nCH + nm];
                    e += powf(hypotf(v.re, v.im), 2.0f);
                }
                float denom = cE[i] > 2.23e-8f ? cE[i] : 2.23e-8f;
                float gain = sqrtf(e / denom);
                float cr = 0.f, ci = 0.f;
                for (int t = 0; t < nTimeSlots; t++) {
                    orc_cpx a = irFB[((size_t)i * nTimeSlots + t) * nCH + nm];
                    orc_cpx b = cFB[i * nTimeSlots + t];     /* conj(b) */
                    cr += a.re * b.re + a.im * b.im;
                    ci += a.im * b.re - a.re * b.im;
                }
                float phase = atan2f(ci, cr);
                orc_cpx* o = &hFB[((size_t)i * nCH + nm) * N_dirs + nd];
                o->re = cosf(phase) * gain; o->im = sinf(phase) * gain;
            }
    }
    free(maxIdx); free(centerImpulse); free(cFB); free(cE); free(ir); free(irFB);
}
