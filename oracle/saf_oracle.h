/*
 * saf_oracle.h — CPU restatement of the SAF per-block rendering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may build, load or call it, and only as the checker.  The product
 * (spatial_audio_framework_amd/csrc -> libsaf_hip.so) never links or calls it.
 *
 * Plain C (gcc), scalar, single thread, float arithmetic in the same order as
 * the reference wherever that order is observable.  Every function cites the
 * reference file:line it restates (paths relative to /root/reference).
 *
 * Pinning: the full reference cannot be built in this image
 * (it needs CBLAS/LAPACKE headers and a default-HRIR source file that are
 * absent), so the oracle is pinned by
 *   (1) the reference's own unit tests restated in tests/test_oracle_*.py
 *       (test__afSTFT, test__saf_rfft, test__getSHreal(_recur),
 *        test__getLoudspeakerDecoderMtx, test__saf_example_ambi_enc/_ambi_dec),
 *   (2) the known-answer values recorded from a reference run in SURVEY.md §8c,
 *   (3) oracle/_ref: the reference's vendored KissFFT, which DOES compile from
 *       its own two source files, for the real-FFT convention,
 *   (4) independent float64 closed forms (direct convolution, direct DFT).
 */
#ifndef SAF_ORACLE_H
#define SAF_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_cpx;

/* ---- data tables (spatial_audio_framework_amd/data/saf_tables.bin) ---- */
int          orc_tables_load(const char* path);              /* 0 on success */
const float* orc_table(const char* name, int* d0, int* d1);   /* NULL if absent */

/* ---- real FFT, saf_rfft semantics (saf_utility_fft.c:531-753) ---- */
void orc_rfft_create(void** ph, int N);
void orc_rfft_destroy(void** ph);
void orc_rfft_forward(void* h, const float* in, orc_cpx* out);   /* unscaled, N/2+1 bins */
void orc_rfft_backward(void* h, const orc_cpx* in, float* out);  /* scaled 1/N */

/* ---- afSTFT (afSTFTlib.c / afSTFT_internal.c) ---- */
#define ORC_AFSTFT_BANDS_CH_TIME 0
#define ORC_AFSTFT_TIME_CH_BANDS 1
void orc_afSTFT_create(void** ph, int nCHin, int nCHout, int hopsize, int lowDelayMode, int hybridmode, int format);
void orc_afSTFT_destroy(void** ph);
/* dataTD: flat [nCH][framesize]; dataFD flat with the "knownDimensions" strides */
void orc_afSTFT_forward_knownDimensions(void* h, const float* dataTD, int framesize, int dataFD_nCH, int dataFD_nHops, orc_cpx* dataFD);
void orc_afSTFT_backward_knownDimensions(void* h, const orc_cpx* dataFD, int framesize, int dataFD_nCH, int dataFD_nHops, float* dataTD);
void orc_afSTFT_channelChange(void* h, int new_nCHin, int new_nCHout);
void orc_afSTFT_clearBuffers(void* h);
int  orc_afSTFT_getNBands(void* h);
int  orc_afSTFT_getProcDelay(void* h);
void orc_afSTFT_getCentreFreqs(void* h /* may be NULL */, float fs, int nBands, float* freqVector);
void orc_afSTFT_FIRtoFilterbankCoeffs(const float* hIR, int N_dirs, int nCH, int ir_len, int hopSize, int LDmode, int hybridmode, orc_cpx* hFB);

/* ---- spherical harmonics (saf_sh.c, saf_hoa.c) ---- */
void orc_unnorm_legendreP(int n, const double* x, int lenX, double* y);
void orc_getSHreal(int order, const float* dirs_rad, int nDirs, float* Y);
void orc_getSHreal_recur(int order, const float* dirs_rad, int nDirs, float* Y);
void orc_getRSH(int order, const float* dirs_deg, int nDirs, float* Y);
void orc_getRSH_recur(int order, const float* dirs_deg, int nDirs, float* Y);
void orc_getMaxREweights(int order, int diagMtxFlag, float* a_n);
void orc_convertHOAChannelConvention(float* insig, int order, int signalLength, int inConv, int outConv); /* 1 ACN, 2 FuMa */
void orc_convertHOANormConvention(float* insig, int order, int signalLength, int inConv, int outConv);    /* 1 N3D, 2 SN3D, 3 FuMa */

/* ---- loudspeaker decoders (saf_hoa.c:326-392, saf_hoa_internal.c:41-155) ---- */
#define ORC_DECODER_DEFAULT 0
#define ORC_DECODER_SAD     1
#define ORC_DECODER_MMD     2
#define ORC_DECODER_EPAD    3
#define ORC_DECODER_ALLRAD  4
void orc_getLoudspeakerDecoderMtx(const float* ls_dirs_deg, int nLS, int method, int order, int enableMaxrE, float* decMtx);
void orc_pinv(const float* inM, int dim1, int dim2, float* outM);  /* utility_spinv, saf_utility_veclib.c:3466 */

/* ---- VBAP (saf_vbap.c) ---- */
int  orc_findLsTriplets(const float* ls_dirs_deg, int L, int omitLargeTriangles, float** out_vertices, int* numOutVertices, int** out_faces, int* numOutFaces);
void orc_findLsPairs(const float* ls_dirs_deg, int L, int* pairs /* [L][2] */);
void orc_vbap2D_table(const float* src_azi_deg, int S, const float* ls_dirs_deg, int L, float* gtable /* [S][L] */);
int  orc_generateVBAPgainTable2D(const float* ls_dirs_deg, int L, int az_res_deg, float* gtable /* NULL: returns the row count */);
void orc_getSpreadSrcDirs3D(float azi, float elev, float spread, int num_src, int num_rings, float* Us);
void orc_invertLsMtx3D(const float* U_spkr, const int* ls_groups, int N_group, float* layoutInvMtx);
void orc_vbap3D(const float* src_dirs, int src_num, int ls_num, const int* ls_groups, int nFaces, float spread, const float* layoutInvMtx, float** GainMtx);
void orc_generateVBAPgainTable3D_srcs(const float* src_dirs_deg, int S, const float* ls_dirs_deg, int L, int omitLargeTriangles, int enableDummies, float spread, float** gtable, int* N_gtable, int* nTriangles);
void orc_generateVBAPgainTable3D(const float* ls_dirs_deg, int L, int az_res_deg, int el_res_deg, int omitLargeTriangles, int enableDummies, float spread, float** gtable, int* N_gtable, int* nTriangles);
void orc_compressVBAPgainTable3D(const float* vbap_gtable, int nTable, int nDirs, float* vbap_gtableComp, int* vbap_gtableIdx);

/* ---- ambi_dec (examples/src/ambi_dec/ambi_dec.c) — loudspeaker output path ---- */
void orc_ambi_dec_create(void** ph, int frameSize);
void orc_ambi_dec_destroy(void** ph);
void orc_ambi_dec_init(void* h, int sampleRate);
void orc_ambi_dec_initCodec(void* h);
void orc_ambi_dec_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);
void orc_ambi_dec_setMasterDecOrder(void* h, int order);
void orc_ambi_dec_setDecOrder(void* h, int order, int band);
void orc_ambi_dec_setDecOrderAllBands(void* h, int order);
void orc_ambi_dec_setLoudspeakers(void* h, const float* dirs_deg, int nLS);
void orc_ambi_dec_setOutputConfigPreset(void* h, int presetID);
void orc_ambi_dec_setHRIRs(void* h, const float* hrirs, const float* dirs_deg, int N, int len, int fs);   /* stands in for the absent default set */
void orc_ambi_dec_setBinauraliseLSflag(void* h, int s);
void orc_ambi_dec_setEnableHRIRsPreProc(void* h, int s);
void orc_ambi_dec_setLoudspeakerAzi_deg(void* h, int i, float v);
void orc_ambi_dec_setLoudspeakerElev_deg(void* h, int i, float v);
const orc_cpx* orc_ambi_dec_getHRTFinterp(void* h);     /* [64][133][2] */
void orc_interpHRTFs_ps(const float* gtableComp, const int* gtableIdx, const float* itds_s, const float* hrtf_fb_mag, int N,
                        const float* freqVector, float azi, float elev, orc_cpx* hout);
void orc_ambi_dec_setChOrder(void* h, int newOrder);
void orc_ambi_dec_setNormType(void* h, int newType);
void orc_ambi_dec_setDecMethod(void* h, int index, int newID);
void orc_ambi_dec_setDecEnableMaxrE(void* h, int index, int newID);
void orc_ambi_dec_setDecNormType(void* h, int index, int newID);
void orc_ambi_dec_setTransitionFreq(void* h, float newValue);
int  orc_ambi_dec_getNumLoudspeakers(void* h);
const float* orc_ambi_dec_getDecMtx(void* h, int dec, int order, int maxrE);   /* [nLS x nSH_order] */
float orc_ambi_dec_getMnorm(void* h, int dec, int order, int which);
const float* orc_ambi_dec_getFreqVector(void* h);
/* stage timers for the cpu_baseline split (seconds accumulated) */
void orc_ambi_dec_getStageTimes(void* h, double* fwd, double* dec, double* bwd);

/* ---- ambi_enc (examples/src/ambi_enc/ambi_enc.c) ---- */
void orc_ambi_enc_create(void** ph, int frameSize);
void orc_ambi_enc_destroy(void** ph);
void orc_ambi_enc_init(void* h, int sampleRate);
void orc_ambi_enc_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);
void orc_ambi_enc_setOutputOrder(void* h, int order);
void orc_ambi_enc_setNumSources(void* h, int n);
void orc_ambi_enc_setSourceAzi_deg(void* h, int idx, float azi);
void orc_ambi_enc_setSourceElev_deg(void* h, int idx, float elev);
void orc_ambi_enc_setSourceGain(void* h, int idx, float g);
void orc_ambi_enc_setChOrder(void* h, int v);
void orc_ambi_enc_setNormType(void* h, int v);
void orc_ambi_enc_setEnablePostScaling(void* h, int v);

/* ---- matrix convolver (saf_utility_matrixConv.c:37-236) ---- */
void orc_matrixConv_create(void** ph, int hopSize, const float* H, int length_h, int nCHin, int nCHout, int usePartFLAG);
void orc_matrixConv_destroy(void** ph);
void orc_matrixConv_apply(void* h, const float* in, float* out);
/* ---- multi-channel and time-varying convolvers (saf_utility_matrixConv.c:237-620) ---- */
void orc_multiConv_create(void** ph, int hopSize, const float* H /* nCH x length_h */, int length_h, int nCH, int usePartFLAG);
void orc_multiConv_destroy(void** ph);
void orc_multiConv_apply(void* h, const float* in /* nCH x hop */, float* out /* nCH x hop */);
void orc_TVConv_create(void** ph, int hopSize, const float* H /* nIRs x nCHout x length_h, flat */, int length_h, int nIRs, int nCHout, int initIdx);
void orc_TVConv_destroy(void** ph);
void orc_TVConv_apply(void* h, const float* in /* hop */, float* out /* nCHout x hop */, int irIdx);

/* ---- binauraliser block path (binauraliser.c:191-285) with caller-supplied per-source HRTF band coefficients ---- */
void orc_binaural_mac(const orc_cpx* inTF /* [nBands][nSrcStride][T] */, const orc_cpx* hrtf /* [nSrc][nBands][2] */,
                      int nBands, int nSrc, int nSrcStride, int T, float scale, orc_cpx* outTF /* [nBands][2][T] */);

/* ---- HRIR processing (saf_hrir.c) and spherical Voronoi weights (saf_utility_geometry.c:937) ---- */
void orc_estimateITDs(const float* hrirs /* N x 2 x len */, int N_dirs, int hrir_len, int fs, float* itds_s);
void orc_diffuseFieldEqualiseHRTFs(int N_dirs, int N_bands, const float* weights /* may be NULL */, orc_cpx* hrtfs /* bands x 2 x N */);
void orc_getVoronoiWeights(const float* dirs_deg, int nDirs, float* weights);

/* ---- binauraliser (examples/src/binauraliser); the HRIR set is injected (the reference's default set is absent) ---- */
void orc_binauraliser_create(void** ph, int frameSize, int maxSources);
void orc_binauraliser_destroy(void** ph);
void orc_binauraliser_setHRIRs(void* h, const float* hrirs, const float* dirs_deg, int N, int len, int fs);
void orc_binauraliser_init(void* h, int sampleRate);
void orc_binauraliser_initCodec(void* h);
void orc_binauraliser_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);
void orc_binauraliser_setSourceAzi_deg(void* h, int i, float v);
void orc_binauraliser_setSourceElev_deg(void* h, int i, float v);
void orc_binauraliser_setNumSources(void* h, int n);
void orc_binauraliser_setEnableHRIRsDiffuseEQ(void* h, int s);
void orc_binauraliser_setEnableRotation(void* h, int s);
void orc_binauraliser_setYaw(void* h, float v);
void orc_binauraliser_setPitch(void* h, float v);
void orc_binauraliser_setRoll(void* h, float v);
void orc_binauraliser_setRPYflag(void* h, int s);
void orc_binauraliser_setInterpMode(void* h, int m);
void orc_binauraliser_setSourceGain(void* h, int i, float g);
int  orc_binauraliser_getNDirs(void* h);
int  orc_binauraliser_getNTriangles(void* h);
const float* orc_binauraliser_getITDs(void* h);
const float* orc_binauraliser_getWeights(void* h);
const orc_cpx* orc_binauraliser_getHRTFfb(void* h);
const orc_cpx* orc_binauraliser_getHRTFinterp(void* h);

/* ---- near-field DVF filters (saf_utility_dvf.c) and first-order response (saf_utility_filters.c:609-671) ---- */
void orc_calcDVFShelfParams(int i, float rho, float* g0, float* gInf, float* fc);
void orc_interpDVFShelfParams(float theta, float rho, float* iG0, float* iGInf, float* iFc);
void orc_dvfShelfCoeffs(float g0, float gInf, float fc, float fs, float* b0, float* b1, float* a1);
void orc_calcDVFCoeffs(float alpha, float rho, float fs, float* b, float* a);
void orc_doaToIpsiInteraural(float azimuth, float elevation, float* alphaLR, float* betaLR /* may be NULL */);
void orc_evalIIRTransferFunctionf(const float* b_coeff, const float* a_coeff, int nCoeffs, const float* freqs, int nFreqs, float fs, int mag2dB,
                                  float* magnitude /* may be NULL */, float* phase_rad /* may be NULL */);

/* ---- binauraliser_nf (examples/src/binauraliser_nf): an orc_binauraliser handle with per-source distances; every
 *      orc_binauraliser_* setter applies to it, process through orc_binauraliserNF_process ---- */
void orc_binauraliserNF_create(void** ph, int frameSize, int maxSources);
void orc_binauraliserNF_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);
void orc_binauraliserNF_setSourceDist_m(void* h, int i, float d);
float orc_binauraliserNF_getSourceDist_m(void* h, int i);
float orc_binauraliserNF_getFarfieldThresh_m(void* h);
float orc_binauraliserNF_getFarfieldHeadroom(void* h);
float orc_binauraliserNF_getNearfieldLimit_m(void* h);
const float* orc_binauraliserNF_getDVFmags(void* h);      /* [maxSrc][2][133] */
const float* orc_binauraliserNF_getDVFphases(void* h);    /* [maxSrc][2][133] */

/* ---- powermap, PWD mode (examples/src/powermap) ---- */
void orc_powermap_create(void** ph, int frameSize);
void orc_powermap_destroy(void** ph);
void orc_powermap_init(void* h, float sampleRate);
void orc_powermap_initCodec(void* h);
void orc_powermap_analysis(void* h, const float* const* inputs, int nInputs, int nSamples, int isPlaying);
void orc_powermap_setPowermapMode(void* h, int m);
void orc_powermap_setMasterOrder(void* h, int o);
void orc_powermap_setCovAvgCoeff(void* h, float a);
void orc_powermap_setAnaOrder(void* h, int o, int band);
void orc_powermap_setAnaOrderAllBands(void* h, int o);
void orc_powermap_setPowermapEQ(void* h, float v, int band);
void orc_powermap_setChOrder(void* h, int v);
void orc_powermap_setNormType(void* h, int v);
void orc_powermap_setPowermapAvgCoeff(void* h, float v);
void orc_powermap_setNumSources(void* h, int n);
void orc_powermap_requestPmapUpdate(void* h);
int  orc_powermap_getPmap(void* h, const float** grid_dirs, const float** pmap, int* nDirs);
const orc_cpx* orc_powermap_getCx(void* h);          /* [133][64*64], row stride nSH of the master order */
const float* orc_powermap_getRawPmap(void* h);       /* [grid_nDirs] after temporal smoothing */
int  orc_powermap_getGridNDirs(void* h);

/* ---- binaural Ambisonic decoders, SH rotation, ambi_bin (saf_hoa_internal.c:162-623, saf_hoa.c:394-603, saf_sh.c:479-560, examples/src/ambi_bin) ---- */
void orc_singular_values(const float* M, int r, int c, float* s);
void orc_getSHrotMtxReal(const float Rxyz[9], float* RotMtx /* (L+1)^2 x (L+1)^2 */, int L);
void orc_yawPitchRoll2Rzyx(float yaw, float pitch, float roll, int rollPitchYawFLAG, float R[9]);
void orc_diffuseFieldEqualiseHRTFs_full(int N, const float* itds_s, const float* centreFreq, int nBands, const float* weights, int applyEQ, int applyPhase, orc_cpx* hrtfs);
/* method: 1 LS, 2 LSDIFFEQ, 3 SPR, 4 TA, 5 MAGLS (BINAURAL_AMBI_DECODER_METHODS, saf_hoa.h:134-171); hrtfs [nBands][2][N]; decMtx [nBands][2][nSH] */
/* ---- ambi_drc example (orc_ambi_drc.c) ---- */
void orc_ambi_drc_create(void** ph, int frameSize);
void orc_ambi_drc_destroy(void** ph);
void orc_ambi_drc_init(void* h, int fs);
void orc_ambi_drc_process(void* h, const float* const* inputs, float* const* outputs, int nCh, int nSamples);
const float* orc_ambi_drc_getLastGains(void* h);      /* [133][T] gain factors of the last block */
/* ---- beamformer example and its SH helpers (orc_beamformer.c) ---- */
void orc_rotateAxisCoeffsReal(int order, const float* c_n, float theta_0 /* inclination */, float phi_0 /* azimuth */, float* c_nm);
void orc_beamWeightsCardioid2Spherical(int N, float* b_n);
void orc_beamWeightsHypercardioid2Spherical(int N, float* b_n);
void orc_beamformer_create(void** ph, int frameSize);
void orc_beamformer_destroy(void** ph);
void orc_beamformer_init(void* h, int fs);
void orc_beamformer_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);
/* ---- rotator example and quaternion helpers (orc_rotator.c) ---- */
void orc_quaternion2rotationMatrix(const float q[4] /* w x y z */, float R[9]);
void orc_rotationMatrix2quaternion(const float R[9], float q[4]);
void orc_euler2Quaternion(float alpha, float beta, float gamma, int convention /* 2 ypr, 3 rpy */, float q[4]);
void orc_quaternion2euler(const float q[4], int convention, float* alpha, float* beta, float* gamma);
void orc_rotator_create(void** ph, int frameSize);
void orc_rotator_destroy(void** ph);
void orc_rotator_init(void* h, int fs);
void orc_rotator_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);
void orc_getBinauralAmbiDecoderFilters(const orc_cpx* hrtfs, const float* dirs_deg, int N, int fftSize, float fs, int method, int order,
                                       const float* itd_s, const float* weights, int diffMatching, int maxRE, float* decFilters /* [2][nSH][fftSize] */);
void orc_getBinauralAmbiDecoderMtx(const orc_cpx* hrtfs, const float* dirs_deg, int N, int nBands, int method, int order, const float* freqVector,
                                   const float* itd_s, const float* weights, int enableDiffCovMatching, int enableMaxRE, orc_cpx* decMtx);
void orc_ambi_bin_create(void** ph, int frameSize);
void orc_ambi_bin_destroy(void** ph);
void orc_ambi_bin_setHRIRs(void* h, const float* hrirs, const float* dirs_deg, int N, int len, int fs);
void orc_ambi_bin_init(void* h, int sampleRate);
void orc_ambi_bin_initCodec(void* h);
void orc_ambi_bin_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);
void orc_ambi_bin_setInputOrderPreset(void* h, int o);
void orc_ambi_bin_setDecodingMethod(void* h, int m);
void orc_ambi_bin_setChOrder(void* h, int v);
void orc_ambi_bin_setNormType(void* h, int v);
void orc_ambi_bin_setEnableMaxRE(void* h, int s);
void orc_ambi_bin_setEnableDiffuseMatching(void* h, int s);
void orc_ambi_bin_setEnableTruncationEQ(void* h, int s);
void orc_ambi_bin_setHRIRsPreProc(void* h, int s);
void orc_truncationEQ(const float* w_n, int order_truncated, int order_target, const double* kr, int nBands, float softThreshold, float* gain);
void orc_beamWeightsMaxEV(int N, float* b_n);
void orc_sphj(int N, double X, int* NM, double* SJ, double* DJ);      /* spherical Bessel j_n, j_n' (Zhang & Jin's SPHJ) */
void orc_sphy(int N, double X, int* NM, double* SY, double* DY);      /* spherical Bessel y_n, y_n' (SPHY) */
void orc_ambi_bin_setEnableRotation(void* h, int s);
void orc_ambi_bin_setYaw(void* h, float v);
void orc_ambi_bin_setPitch(void* h, float v);
void orc_ambi_bin_setRoll(void* h, float v);
void orc_ambi_bin_setRPYflag(void* h, int s);
const orc_cpx* orc_ambi_bin_getDecMtx(void* h);      /* [133][2][64] */

/* ---- matrixconv / multiconv example operators (examples/src/matrixconv, examples/src/multiconv): FIFO around the convolvers ---- */
void orc_convex_create(void** ph, int matrix /* 1: matrixconv, 0: multiconv */);
void orc_convex_destroy(void** ph);
void orc_convex_init(void* h, int sampleRate, int hostBlockSize);
void orc_convex_setFilters(void* h, const float* H /* [numChannels][numSamples] */, int numChannels, int numSamples);
void orc_convex_setEnablePart(void* h, int s);
void orc_convex_setNumInputChannels(void* h, int n);       /* multiconv: setNumChannels */
int  orc_convex_getProcessingDelay(void* h);
int  orc_convex_getFilterLength(void* h);
void orc_convex_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);

void orc_tvconvex_create(void** ph);
void orc_tvconvex_destroy(void** ph);
void orc_tvconvex_init(void* h, int hostBlockSize);
void orc_tvconvex_setIRsAndPositions(void* h, const float* irs /* [nPos][nIr][irLen] */, const float* positions /* [nPos][3] */, int nPos, int nIr, int irLen);
void orc_tvconvex_setTargetPosition(void* h, float v, int dim);
int  orc_tvconvex_getListenerPositionIdx(void* h);
void orc_tvconvex_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);

/* ---- adaptive / sub-space activity maps (saf_sh.c:1586-1858); Y_grid is the REAL [nSH][G] matrix (the reference passes it as complex with zero imaginary part) ---- */
void orc_generateMVDRmap(int order, const orc_cpx* Cx, const float* Y_grid, int G, float regPar, float* pmap, orc_cpx* w_MVDR_out /* [nSH][G] or NULL */);
void orc_generateCroPaCLCMVmap(int order, const orc_cpx* Cx, const float* Y_grid, int G, float regPar, float lambda, float* pmap);
void orc_generateMUSICmap(int order, const orc_cpx* Cx, const float* Y_grid, int nSources, int G, int logScaleFlag, float* pmap);
void orc_generateMinNormMap(int order, const orc_cpx* Cx, const float* Y_grid, int nSources, int G, int logScaleFlag, float* pmap);
void orc_herm_eig(int n, const orc_cpx* A, double* eig /* descending */, double* Vre, double* Vim /* [n][n], columns = eigenvectors */);

/* ---- panner (examples/src/panner) and getPvalues (saf_vbap.c:475-492) ---- */
void orc_getPvalues(float DTT, const float* freq, int nFreq, float* pValues);
void orc_panner_create(void** ph, int frameSize);
void orc_panner_destroy(void** ph);
void orc_panner_init(void* h, int sampleRate);
void orc_panner_initCodec(void* h);
void orc_panner_process(void* h, const float* const* inputs, float* const* outputs, int nInputs, int nOutputs, int nSamples);
void orc_panner_setSourceAzi_deg(void* h, int i, float v);
void orc_panner_setSourceElev_deg(void* h, int i, float v);
void orc_panner_setNumSources(void* h, int n);
void orc_panner_setLoudspeakerAzi_deg(void* h, int i, float v);
void orc_panner_setLoudspeakerElev_deg(void* h, int i, float v);
void orc_panner_setNumLoudspeakers(void* h, int n);
void orc_panner_setOutputConfigPreset(void* h, int id);
void orc_panner_setInputConfigPreset(void* h, int id);
void orc_panner_setDTT(void* h, float v);
void orc_panner_setSpread(void* h, float v);
void orc_panner_setYaw(void* h, float v);
void orc_panner_setPitch(void* h, float v);
void orc_panner_setRoll(void* h, float v);
int  orc_panner_getNumSources(void* h);
int  orc_panner_getNumLoudspeakers(void* h);
int  orc_panner_getNTriangles(void* h);
const float* orc_panner_getGains(void* h);      /* [133][64][64]: band, source, loudspeaker */
const float* orc_panner_getPvalue(void* h);

#ifdef __cplusplus
}
#endif
#endif
