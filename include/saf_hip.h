/*
 * saf_hip.h — C-ABI of libsaf_hip: the MI355X (gfx950) implementation of the SAF
 * per-block rendering hot path.
 *
 * Every function below that carries a SAF name keeps the reference's signature,
 * argument meaning and error behaviour, so a SAF caller links against this
 * library instead of the reference objects for this path.
 * The reference interface each entry replaces is cited as file:line relative to
 * the SAF checkout.  Functions prefixed `saf_hip_` are additions: device-pointer
 * and batched entry points (throughput needs many frames/instances per launch)
 * and runtime control.  No torch / C++ types cross this boundary.
 *
 * Error behaviour: no return codes.  A process call with the wrong
 * block size or an un-initialised codec zero-fills its outputs.  Violated
 * preconditions and any HIP failure abort with a message — there is no CPU
 * fallback anywhere in this library.
 */
#ifndef SAF_HIP_H_INCLUDED
#define SAF_HIP_H_INCLUDED

#ifdef __cplusplus
# include <complex>
typedef std::complex<float> float_complex;     /* framework/modules/saf_utilities/saf_utility_complex.h:34 */
extern "C" {
#else
typedef float _Complex float_complex;          /* saf_utility_complex.h:70 */
#endif

#define SAF_API __attribute__((visibility("default")))

/* ========================================================================== */
/*                                 runtime                                    */
/* ========================================================================== */
/** One-block host-pointer calls (X_process, saf_matrixConv_apply, ...) let the kernels read / write the library's pinned
 *  staging blocks directly instead of copying them to and from device memory (default 1; env SAF_HIP_ZERO_COPY). */
SAF_API void saf_hip_setZeroCopyIO(int enable);
SAF_API int  saf_hip_getZeroCopyIO(void);
/** Adopt a caller-owned hipStream_t: all work of the device-pointer entry points is enqueued on it, in order with the caller's own
 *  work on that stream.  NULL selects a library-owned NON-BLOCKING stream — the legacy default stream (whose handle is also 0)
 *  cannot be adopted, and work the caller queued on it is NOT ordered against the library's: a caller that produces inputs on
 *  the default stream synchronises (or uses a stream of its own, as tests/conftest.py does) before calling. */
SAF_API void  saf_hip_set_stream(void* hipStream);
SAF_API void* saf_hip_get_stream(void);
SAF_API void  saf_hip_synchronize(void);
SAF_API int   saf_hip_device_count(void);
SAF_API void  saf_hip_set_device(int dev);
SAF_API const char* saf_hip_version(void);
/** Per-kernel timing with HIP events recorded on the library stream around every kernel launch
 *  ("afstft_analysis", "band_gemm", "afstft_synthesis", ...). read() synchronises, returns the number
 *  of launches of `kernelName` since reset() and their summed duration. */
SAF_API void saf_hip_profile_enable(int on);
SAF_API void saf_hip_profile_reset(void);
SAF_API int  saf_hip_profile_read(const char* kernelName, double* total_ms);
/** Two HIP events on the library stream: start() records the first, stop_ms() records the second, waits for it and returns
 *  the time between them (what the device spent on everything the library enqueued in between). */
SAF_API void   saf_hip_stopwatch_start(void);
SAF_API double saf_hip_stopwatch_stop_ms(void);

/* ========================================================================== */
/*      afSTFT  (framework/resources/afSTFT/afSTFTlib.h:85-278)               */
/* ========================================================================== */
typedef enum {                      /* afSTFTlib.h:79-83 */
    AFSTFT_BANDS_CH_TIME,
    AFSTFT_TIME_CH_BANDS
} AFSTFT_FDDATA_FORMAT;

/** afSTFTlib.h:107 / afSTFTlib.c:142.  hopsize 64, 128 or 256 — exactly what the reference accepts in hybrid mode
 *  (afSTFTlib.c:158-159).  LIMITATION: with hybridmode = 0 the reference also takes any other hop that divides 1024 (32, 512,
 *  1024 ...); this library aborts with a message for those.
 *  128 — the value every operator of the path fixes (ambi_dec_internal.h:68, binauraliser_internal.h:63, powermap_internal.h:67)
 *  — runs the tuned kernels; 64 and 256 run generic, untuned kernels (csrc/afstft_generic.hip) of the same algorithm. */
SAF_API void afSTFT_create(void** const phSTFT, int nCHin, int nCHout, int hopsize, int lowDelayMode, int hybridmode, AFSTFT_FDDATA_FORMAT format);
SAF_API void afSTFT_destroy(void** const phSTFT);                                                         /* afSTFTlib.h:120 */
/** afSTFTlib.h:85 / afSTFTlib.c:78-119: one-shot analysis with a fresh filterbank; inTD [nSamplesTD][nCH], outTF [nBands][ceil(nSamplesTD/hop)][nCH]. */
SAF_API void afAnalyse(float* inTD, int nSamplesTD, int nCH, int hopSize, int LDmode, int hybridmode, float_complex* outTF);
SAF_API void afSTFT_forward(void* const hSTFT, float** dataTD, int framesize, float_complex*** dataFD);   /* afSTFTlib.h:130 */
SAF_API void afSTFT_forward_knownDimensions(void* const hSTFT, float** dataTD, int framesize, int dataFD_nCH, int dataFD_nHops, float_complex*** dataFD); /* afSTFTlib.h:149 */
SAF_API void afSTFT_forward_flat(void* const hSTFT, float* dataTD, int framesize, float_complex* dataFD); /* afSTFTlib.h:164 */
SAF_API void afSTFT_backward(void* const hSTFT, float_complex*** dataFD, int framesize, float** dataTD);  /* afSTFTlib.h:177 */
SAF_API void afSTFT_backward_knownDimensions(void* const hSTFT, float_complex*** dataFD, int framesize, int dataFD_nCH, int dataFD_nHops, float** dataTD); /* afSTFTlib.h:196 */
SAF_API void afSTFT_backward_flat(void* const hSTFT, float_complex* dataFD, int framesize, float* dataTD); /* afSTFTlib.h:211 */
SAF_API void afSTFT_channelChange(void* const hSTFT, int new_nCHin, int new_nCHout);                      /* afSTFTlib.h:224 */
SAF_API void afSTFT_clearBuffers(void* const hSTFT);                                                      /* afSTFTlib.h:229 */
SAF_API int  afSTFT_getNBands(void* const hSTFT);                                                         /* afSTFTlib.h:232 */
SAF_API int  afSTFT_getProcDelay(void* const hSTFT);                                                      /* afSTFTlib.h:247 */
SAF_API void afSTFT_getCentreFreqs(void* const hSTFT, float fs, int nBands, float* freqVector);           /* afSTFTlib.h:250 */
SAF_API void afSTFT_FIRtoFilterbankCoeffs(float* hIR, int N_dirs, int nCH, int ir_len, int hopSize, int LDmode, int hybridmode, float_complex* hFB); /* afSTFTlib.h:269 */

/** Device-pointer transform of `nHops` hops for all channels of the handle.
 *  d_td[ch*td_ch_stride + t]  (floats, 16-byte aligned, strides multiple of 4);
 *  d_fd[band*fd_band_stride + ch*fd_ch_stride + hop] (complex). State carries over between calls
 *  exactly like consecutive afSTFT_forward calls. */
SAF_API void saf_hip_afSTFT_forward_dev(void* const hSTFT, const float* d_td, long long td_ch_stride, int nHops,
                                        float_complex* d_fd, long long fd_band_stride, long long fd_ch_stride);
SAF_API void saf_hip_afSTFT_backward_dev(void* const hSTFT, const float_complex* d_fd, long long fd_band_stride, long long fd_ch_stride,
                                         int nHops, float* d_td, long long td_ch_stride);

/* ========================================================================== */
/*      spherical harmonics / HOA (saf_sh.h, saf_hoa.h)                       */
/* ========================================================================== */
SAF_API void getSHreal(int order, float* dirs_rad, int nDirs, float* Y);        /* saf_sh.h:176 / saf_sh.c:190 */
SAF_API void getSHreal_recur(int order, float* dirs_rad, int nDirs, float* Y);  /* saf_sh.h:211 / saf_sh.c:255 */
SAF_API void getRSH(int order, float* dirs_deg, int nDirs, float* Y);           /* saf_hoa.h:293 / saf_hoa.c:118 */
SAF_API void getRSH_recur(int order, float* dirs_deg, int nDirs, float* Y);     /* saf_hoa.h:328 / saf_hoa.c:152 */
/** Same as getRSH_recur but with device pointers (dirs [nDirs][2] degrees, Y [nSH][nDirs]). */
SAF_API void saf_hip_getRSH_recur_dev(int order, const float* d_dirs_deg, int nDirs, float* d_Y);

typedef enum { HOA_CH_ORDER_ACN, HOA_CH_ORDER_FUMA } HOA_CH_ORDER;              /* saf_hoa.h:183-188 */
typedef enum { HOA_NORM_N3D, HOA_NORM_SN3D, HOA_NORM_FUMA } HOA_NORM;           /* saf_hoa.h:205-211 */
SAF_API void convertHOAChannelConvention(float* insig, int order, int signalLength, HOA_CH_ORDER inConvention, HOA_CH_ORDER outConvention); /* saf_hoa.h:237 */
SAF_API void convertHOANormConvention(float* insig, int order, int signalLength, HOA_NORM inConvention, HOA_NORM outConvention);           /* saf_hoa.h:262 */
SAF_API void getMaxREweights(int order, int diagMtxFlag, float* a_n);          /* saf_hoa.h:363 / saf_hoa.c:235 */

typedef enum {                                                                  /* saf_hoa.h:61-111 */
    LOUDSPEAKER_DECODER_DEFAULT,
    LOUDSPEAKER_DECODER_SAD,
    LOUDSPEAKER_DECODER_MMD,
    LOUDSPEAKER_DECODER_EPAD,
    LOUDSPEAKER_DECODER_ALLRAD
} LOUDSPEAKER_AMBI_DECODER_METHODS;
SAF_API void getLoudspeakerDecoderMtx(float* ls_dirs_deg, int nLS, LOUDSPEAKER_AMBI_DECODER_METHODS method, int order, int enableMaxReWeighting, float* decMtx); /* saf_hoa.h:413 / saf_hoa.c:326 */

/* ========================================================================== */
/*      VBAP (saf_vbap.h:73-450)                                              */
/* ========================================================================== */
SAF_API void generateVBAPgainTable3D_srcs(float* src_dirs_deg, int S, float* ls_dirs_deg, int L, int omitLargeTriangles, int enableDummies, float spread, float** gtable, int* N_gtable, int* nTriangles); /* saf_vbap.h:73 */
SAF_API void generateVBAPgainTable3D(float* ls_dirs_deg, int L, int az_res_deg, int el_res_deg, int omitLargeTriangles, int enableDummies, float spread, float** gtable, int* N_gtable, int* nTriangles);   /* saf_vbap.h:129 */
SAF_API void compressVBAPgainTable3D(float* vbap_gtable, int nTable, int nDirs, float* vbap_gtableComp, int* vbap_gtableIdx); /* saf_vbap.h:174 */
SAF_API void VBAPgainTable2InterpTable(float* vbap_gtable, int nTable, int nDirs);                                           /* saf_vbap.h:192 */
SAF_API void findLsTriplets(float* ls_dirs_deg, int L, int omitLargeTriangles, float** out_vertices, int* numOutVertices, int** out_faces, int* numOutFaces); /* saf_vbap.h:328 */
/* 2-D (horizontal) VBAP — not reached by any operator of this library (panner forces the 3-D layout, panner_internal.h:59)
 * but part of the module's interface; host code.  Source directions are AZIMUTHS only (one float per source), as
 * vbap2D reads them. */
SAF_API void generateVBAPgainTable2D_srcs(float* src_dirs_deg, int S, float* ls_dirs_deg, int L, float** gtable, int* N_gtable, int* nPairs); /* saf_vbap.h:277 */
SAF_API void generateVBAPgainTable2D(float* ls_dirs_deg, int L, int az_res_deg, float** gtable, int* N_gtable, int* nPairs);                   /* saf_vbap.h:299 */
SAF_API void findLsPairs(float* ls_dirs_deg, int L, int** out_pairs, int* numOutPairs);                                                         /* saf_vbap.h:404 */
SAF_API void invertLsMtx2D(float* U_spkr, int* ls_pairs, int N_pairs, float** layoutInvMtx);                                                    /* saf_vbap.h:417 */
SAF_API void vbap2D(float* src_dirs, int src_num, int ls_num, int* ls_pairs, int N_pairs, float* layoutInvMtx, float** GainMtx);               /* saf_vbap.h:431 */
SAF_API void getSpreadSrcDirs3D(float src_azi_rad, float src_elev_rad, float spread, int num_src, int num_rings_3d, float* U_spread);           /* saf_vbap.h:366 */
SAF_API void invertLsMtx3D(float* U_spkr, int* ls_groups, int N_group, float** layoutInvMtx);                                /* saf_vbap.h:348 */
SAF_API void vbap3D(float* src_dirs, int src_num, int ls_num, int* ls_groups, int nFaces, float spread, float* layoutInvMtx, float** GainMtx); /* saf_vbap.h:393 */

/* ========================================================================== */
/*      shared operator enums (examples/include/_common.h)                    */
/* ========================================================================== */
typedef enum { CH_ACN = 1, CH_FUMA } CH_ORDER;                                   /* _common.h:57-61 */
typedef enum { NORM_N3D = 1, NORM_SN3D, NORM_FUMA } NORM_TYPES;                  /* _common.h:70-75 */
typedef enum { CODEC_STATUS_INITIALISED = 0, CODEC_STATUS_NOT_INITIALISED, CODEC_STATUS_INITIALISING } CODEC_STATUS; /* _common.h:199-207 */
typedef enum { PROC_STATUS_ONGOING = 0, PROC_STATUS_NOT_ONGOING } PROC_STATUS;   /* _common.h:215-220 */
typedef enum { MIC_PRESET_IDEAL = 1, MIC_PRESET_ZYLIA, MIC_PRESET_EIGENMIKE32, MIC_PRESET_DTU_MIC } MIC_PRESETS;   /* _common.h:78-84 */
typedef enum {                                                                   /* _common.h:88-119 */
    LOUDSPEAKER_ARRAY_PRESET_DEFAULT = 1, LOUDSPEAKER_ARRAY_PRESET_STEREO, LOUDSPEAKER_ARRAY_PRESET_5PX, LOUDSPEAKER_ARRAY_PRESET_7PX,
    LOUDSPEAKER_ARRAY_PRESET_8PX, LOUDSPEAKER_ARRAY_PRESET_9PX, LOUDSPEAKER_ARRAY_PRESET_10PX, LOUDSPEAKER_ARRAY_PRESET_11PX,
    LOUDSPEAKER_ARRAY_PRESET_11PX_7_4, LOUDSPEAKER_ARRAY_PRESET_13PX, LOUDSPEAKER_ARRAY_PRESET_22PX, LOUDSPEAKER_ARRAY_PRESET_22P2_9_10_3,
    LOUDSPEAKER_ARRAY_PRESET_AALTO_MCC, LOUDSPEAKER_ARRAY_PRESET_AALTO_MCC_SUBSET, LOUDSPEAKER_ARRAY_PRESET_AALTO_APAJA,
    LOUDSPEAKER_ARRAY_PRESET_AALTO_LR, LOUDSPEAKER_ARRAY_PRESET_DTU_AVIL, LOUDSPEAKER_ARRAY_PRESET_ZYLIA_LAB,
    LOUDSPEAKER_ARRAY_PRESET_T_DESIGN_4, LOUDSPEAKER_ARRAY_PRESET_T_DESIGN_12, LOUDSPEAKER_ARRAY_PRESET_T_DESIGN_24,
    LOUDSPEAKER_ARRAY_PRESET_T_DESIGN_36, LOUDSPEAKER_ARRAY_PRESET_T_DESIGN_48, LOUDSPEAKER_ARRAY_PRESET_T_DESIGN_60,
    LOUDSPEAKER_ARRAY_PRESET_SPH_COV_9, LOUDSPEAKER_ARRAY_PRESET_SPH_COV_16, LOUDSPEAKER_ARRAY_PRESET_SPH_COV_25,
    LOUDSPEAKER_ARRAY_PRESET_SPH_COV_49, LOUDSPEAKER_ARRAY_PRESET_SPH_COV_64
} LOUDSPEAKER_ARRAY_PRESETS;
typedef enum {                                                                   /* _common.h:123-155 */
    SOURCE_CONFIG_PRESET_DEFAULT = 1, SOURCE_CONFIG_PRESET_MONO, SOURCE_CONFIG_PRESET_STEREO, SOURCE_CONFIG_PRESET_5PX, SOURCE_CONFIG_PRESET_7PX,
    SOURCE_CONFIG_PRESET_8PX, SOURCE_CONFIG_PRESET_9PX, SOURCE_CONFIG_PRESET_10PX, SOURCE_CONFIG_PRESET_11PX, SOURCE_CONFIG_PRESET_11PX_7_4,
    SOURCE_CONFIG_PRESET_13PX, SOURCE_CONFIG_PRESET_22PX, SOURCE_CONFIG_PRESET_22P2_9_10_3, SOURCE_CONFIG_PRESET_AALTO_MCC,
    SOURCE_CONFIG_PRESET_AALTO_MCC_SUBSET, SOURCE_CONFIG_PRESET_AALTO_APAJA, SOURCE_CONFIG_PRESET_AALTO_LR, SOURCE_CONFIG_PRESET_DTU_AVIL,
    SOURCE_CONFIG_PRESET_ZYLIA_LAB, SOURCE_CONFIG_PRESET_T_DESIGN_4, SOURCE_CONFIG_PRESET_T_DESIGN_12, SOURCE_CONFIG_PRESET_T_DESIGN_24,
    SOURCE_CONFIG_PRESET_T_DESIGN_36, SOURCE_CONFIG_PRESET_T_DESIGN_48, SOURCE_CONFIG_PRESET_T_DESIGN_60, SOURCE_CONFIG_PRESET_SPH_COV_9,
    SOURCE_CONFIG_PRESET_SPH_COV_16, SOURCE_CONFIG_PRESET_SPH_COV_25, SOURCE_CONFIG_PRESET_SPH_COV_49, SOURCE_CONFIG_PRESET_SPH_COV_64
} SOURCE_CONFIG_PRESETS;
#define PROGRESSBARTEXT_CHAR_LENGTH ( 256 )                                      /* _common.h:225 */
#define MAX_NUM_CHANNELS ( 64 )                                                  /* _common.h:228 */
#define MAX_SH_ORDER ( 7 )                                                       /* _common.h:50 */

/* ========================================================================== */
/*      ambi_dec (examples/include/ambi_dec.h:114-520)                        */
/* ========================================================================== */
typedef enum { DECODING_METHOD_SAD = 1, DECODING_METHOD_MMD, DECODING_METHOD_EPAD, DECODING_METHOD_ALLRAD } AMBI_DEC_DECODING_METHODS; /* ambi_dec.h:73-78 */
typedef enum { AMPLITUDE_PRESERVING = 1, ENERGY_PRESERVING } AMBI_DEC_DIFFUSE_FIELD_EQ_APPROACH;                                  /* ambi_dec.h:92-95 */

/** The reference fixes the block size at compile time (-DAMBI_DEC_FRAME_SIZE, ambi_dec_internal.h:61-67, default 128).
 *  Here it is a process-wide setting read by ambi_dec_create; must be a multiple of 128. */
/** Block path of loudspeaker decoding.  Every per-band matrix ambi_dec_process applies (ambi_dec.c:518-540) is the
 *  dense order-N decoder of its band's decoder slot (ambi_dec.c:283-288 only truncates columns) times a diagonal of
 *  per-SH-channel weights (max-rE, M_norm, order truncation).  The diagonal commutes into the filterbank of each input
 *  channel and the dense matrix out of it:  out = sum_d M_d synthesis( w_d(band)[ch] (.) analysis(x[ch]) ).
 *    mode 1 (default): "equaliser path" — one kernel runs analysis -> per-band gains -> synthesis per SH channel with the
 *            spectra kept on chip, then ONE time-domain MFMA GEMM applies the dense decoder(s); channels whose weights are
 *            the same in every band skip the transforms (FFT / inverse FFT cancel, the hybrid split + merge is its delay);
 *    mode 2: as 1, every channel runs the transforms;
 *    mode 0: the three-kernel transform path (afSTFT analysis -> per-band MFMA GEMM -> afSTFT synthesis).
 *  Same outputs to rounding (1e-6).  The overlap-add history lives in the SH domain on the equaliser path and in the
 *  loudspeaker domain on the transform path: switching 1/2 -> 0 converts it (exact); a pipeline that has run mode 0 stays
 *  on the transform path until its state is cleared (initCodec / saf_hip_ambi_dec_batch_clear).  Binauralised output
 *  always takes the transform path.  saf_hip_ambi_dec(_batch)_lastPath: 0 transform, 1 equaliser, -1 nothing run yet. */
SAF_API void saf_hip_ambi_dec_setTimeDomainPath(int mode);
SAF_API int  saf_hip_ambi_dec_getTimeDomainPath(void);
SAF_API int  saf_hip_ambi_dec_lastPath(void* const hAmbi);
SAF_API void saf_hip_ambi_dec_setFrameSize(int frameSize);

SAF_API void ambi_dec_create(void** const phAmbi);                               /* ambi_dec.h:114 */
SAF_API void ambi_dec_destroy(void** const phAmbi);                              /* ambi_dec.h:121 */
SAF_API void ambi_dec_init(void* const hAmbi, int samplerate);                   /* ambi_dec.h:131 */
SAF_API void ambi_dec_initCodec(void* const hAmbi);                              /* ambi_dec.h:149 */
SAF_API void ambi_dec_process(void* const hAmbi, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples); /* ambi_dec.h:161 */
SAF_API void ambi_dec_refreshSettings(void* const hAmbi);                        /* ambi_dec.h:177 */
SAF_API void ambi_dec_setMasterDecOrder(void* const hAmbi, int newValue);        /* ambi_dec.h:188 */
SAF_API void ambi_dec_setDecOrder(void* const hAmbi, int newValue, int bandIdx); /* ambi_dec.h:199 */
SAF_API void ambi_dec_setDecOrderAllBands(void* const hAmbi, int newValue);      /* ambi_dec.h:209 */
SAF_API void ambi_dec_setLoudspeakerAzi_deg(void* const hAmbi, int index, float newAzi_deg);   /* ambi_dec.h:218 */
SAF_API void ambi_dec_setLoudspeakerElev_deg(void* const hAmbi, int index, float newElev_deg); /* ambi_dec.h:229 */
SAF_API void ambi_dec_setNumLoudspeakers(void* const hAmbi, int new_nLoudspeakers);            /* ambi_dec.h:236 */
SAF_API void ambi_dec_setBinauraliseLSflag(void* const hAmbi, int newState);     /* ambi_dec.h:246 */
SAF_API void ambi_dec_setUseDefaultHRIRsflag(void* const hAmbi, int newState);   /* ambi_dec.h:259 */
SAF_API void ambi_dec_setSofaFilePath(void* const hAmbi, const char* path);      /* ambi_dec.h:272 */
SAF_API void ambi_dec_setEnableHRIRsPreProc(void* const hAmbi, int newState);    /* ambi_dec.h:275 */
SAF_API void ambi_dec_setSourcePreset(void* const hAmbi, int newPresetID);       /* ambi_dec.h:287 */
SAF_API void ambi_dec_setOutputConfigPreset(void* const hAmbi, int newPresetID); /* ambi_dec.h:295 */
SAF_API void ambi_dec_setChOrder(void* const hAmbi, int newOrder);               /* ambi_dec.h:301 */
SAF_API void ambi_dec_setNormType(void* const hAmbi, int newType);               /* ambi_dec.h:307 */
SAF_API void ambi_dec_setDecMethod(void* const hAmbi, int index, int newID);     /* ambi_dec.h:318 */
SAF_API void ambi_dec_setDecEnableMaxrE(void* const hAmbi, int index, int newID);/* ambi_dec.h:328 */
SAF_API void ambi_dec_setDecNormType(void* const hAmbi, int index, int newID);   /* ambi_dec.h:343 */
SAF_API void ambi_dec_setTransitionFreq(void* const hAmbi, float newValue);      /* ambi_dec.h:352 */
SAF_API int  ambi_dec_getFrameSize(void);                                        /* ambi_dec.h:363 */
SAF_API CODEC_STATUS ambi_dec_getCodecStatus(void* const hAmbi);                 /* ambi_dec.h:366 */
SAF_API float ambi_dec_getProgressBar0_1(void* const hAmbi);                     /* ambi_dec.h:373 */
SAF_API void ambi_dec_getProgressBarText(void* const hAmbi, char* text);         /* ambi_dec.h:381 */
SAF_API int  ambi_dec_getMasterDecOrder(void* const hAmbi);                      /* ambi_dec.h:384 */
SAF_API int  ambi_dec_getDecOrder(void* const hAmbi, int bandIdx);               /* ambi_dec.h:390 */
SAF_API int  ambi_dec_getDecOrderAllBands(void* const hAmbi);                    /* ambi_dec.h:393 */
SAF_API void ambi_dec_getDecOrderHandle(void* const hAmbi, float** pX_vector, int** pY_values, int* pNpoints); /* ambi_dec.h:403 */
SAF_API int  ambi_dec_getNumberOfBands(void);                                    /* ambi_dec.h:409 */
SAF_API float ambi_dec_getLoudspeakerAzi_deg(void* const hAmbi, int index);      /* ambi_dec.h:412 */
SAF_API float ambi_dec_getLoudspeakerElev_deg(void* const hAmbi, int index);     /* ambi_dec.h:415 */
SAF_API int  ambi_dec_getNumLoudspeakers(void* const hAmbi);                     /* ambi_dec.h:418 */
SAF_API int  ambi_dec_getMaxNumLoudspeakers(void);                               /* ambi_dec.h:421 */
SAF_API int  ambi_dec_getNSHrequired(void* const hAmbi);                         /* ambi_dec.h:427 */
SAF_API int  ambi_dec_getBinauraliseLSflag(void* const hAmbi);                   /* ambi_dec.h:434 */
SAF_API int  ambi_dec_getUseDefaultHRIRsflag(void* const hAmbi);                 /* ambi_dec.h:444 */
SAF_API char* ambi_dec_getSofaFilePath(void* const hAmbi);                       /* ambi_dec.h:453 */
SAF_API int  ambi_dec_getEnableHRIRsPreProc(void* const hAmbi);                  /* ambi_dec.h:459 */
SAF_API int  ambi_dec_getChOrder(void* const hAmbi);                             /* ambi_dec.h:466 */
SAF_API int  ambi_dec_getNormType(void* const hAmbi);                            /* ambi_dec.h:473 */
SAF_API int  ambi_dec_getDecMethod(void* const hAmbi, int index);                /* ambi_dec.h:479 */
SAF_API int  ambi_dec_getDecEnableMaxrE(void* const hAmbi, int index);           /* ambi_dec.h:489 */
SAF_API int  ambi_dec_getDecNormType(void* const hAmbi, int index);              /* ambi_dec.h:502 */
SAF_API float ambi_dec_getTransitionFreq(void* const hAmbi);                     /* ambi_dec.h:508 */
SAF_API int  ambi_dec_getHRIRsamplerate(void* const hAmbi);                      /* ambi_dec.h:511 */
SAF_API int  ambi_dec_getDAWsamplerate(void* const hAmbi);                       /* ambi_dec.h:514 */
SAF_API int  ambi_dec_getProcessingDelay(void);                                  /* ambi_dec.h:520 */

/** Read-back of the designed decoder (what ambi_dec_codecPars holds, ambi_dec_internal.h:88-96): [nLS x (order+1)^2]. */
SAF_API void saf_hip_ambi_dec_getDecoderMtx(void* const hAmbi, int decIdx, int order, int maxrE, float* out);
SAF_API float saf_hip_ambi_dec_getDecoderNorm(void* const hAmbi, int decIdx, int order, int ampOrEnergy);

/* ---- batched, device-resident entry point (throughput path) ----
 * A batch groups nInst initialised ambi_dec handles with the same block size, master order and
 * loudspeaker count (decoders may differ per instance).  One call processes nFrames consecutive
 * blocks of every instance; filterbank state carries over between calls exactly as between
 * consecutive ambi_dec_process calls.  Sample addressing (floats):
 *     x[inst*inst_stride + frame*frame_stride + ch*ch_stride + n],  n < frameSize
 * Pointers are device pointers, 16-byte aligned, strides multiples of 4.  Work is enqueued on the
 * library stream and NOT synchronised. */
SAF_API void* saf_hip_ambi_dec_batch_create(void* const* hAmbis, int nInst, int maxFramesPerCall);
SAF_API void  saf_hip_ambi_dec_batch_destroy(void** const phBatch);
SAF_API void  saf_hip_ambi_dec_batch_clear(void* const hBatch);
SAF_API int   saf_hip_ambi_dec_batch_lastPath(void* const hBatch);
/** Equaliser path, optional: the dense time-domain decode runs BESIDE the filterbank equaliser kernel instead of after it — a
 *  persistent grid of MFMA workgroups on a second stream takes each instance as soon as its equaliser workgroups have published
 *  their output (the reference decodes every frame right after transforming it, ambi_dec.c:514-566).  Same arithmetic, same
 *  results (bit-identical to the two kernels run in sequence; 64 loudspeakers only).  setOverlap: 0 (default) never, 1 for launches
 *  of >= 3072 (instance, SH channel) pairs, 2 whenever the shape allows (tests).  Off by default: measured slower on MI355X
 *  (profiles/r03_overlap_experiment.txt).  3: the decode INSIDE the equaliser launch — the channel workgroups of an instance pass z
 *  to each other through write-through stores and per-sub-chunk counters and each decodes 1/64 of it (order 7, 64 loudspeakers,
 *  one dense decoder, blocks in multiples of 16 hops; results within 1e-6 of the two kernels'); also measured slower
 *  (profiles/r03_coop_experiment.txt).  batch_lastOverlap: 1 / 3 when the last call of the batch ran that way;
 *  batch_decodeGiveUps: how many decode workgroups ever gave up waiting for the equaliser kernel (their blocks were then computed
 *  by the fix-up launch): 0 in normal operation. */
SAF_API void  saf_hip_ambi_dec_setOverlap(int mode);
SAF_API int   saf_hip_ambi_dec_getOverlap(void);
SAF_API int   saf_hip_ambi_dec_batch_lastOverlap(void* const hBatch);
SAF_API int   saf_hip_ambi_dec_batch_decodeGiveUps(void* const hBatch);
SAF_API void  saf_hip_ambi_dec_batch_process(void* const hBatch,
                                             const float* d_in, long long in_inst_stride, long long in_frame_stride, long long in_ch_stride,
                                             float* d_out, long long out_inst_stride, long long out_frame_stride, long long out_ch_stride,
                                             int nFrames);

/* ========================================================================== */
/*      ambi_enc (examples/include/ambi_enc.h:55-222)                         */
/* ========================================================================== */
/** The reference fixes the block size at compile time (-DAMBI_ENC_FRAME_SIZE, ambi_enc_internal.h:41-47, default 64).
 *  Here it is a process-wide setting read by ambi_enc_create; must be a multiple of 4. */
SAF_API void saf_hip_ambi_enc_setFrameSize(int frameSize);

SAF_API void ambi_enc_create(void** const phAmbi);                               /* ambi_enc.h:55 */
SAF_API void ambi_enc_destroy(void** const phAmbi);                              /* ambi_enc.h:62 */
SAF_API void ambi_enc_init(void* const hAmbi, int samplerate);                   /* ambi_enc.h:70 */
SAF_API void ambi_enc_process(void* const hAmbi, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples); /* ambi_enc.h:84 */
SAF_API void ambi_enc_refreshParams(void* const hAmbi);                          /* ambi_enc.h:100 */
SAF_API void ambi_enc_setOutputOrder(void* const hAmbi, int newValue);           /* ambi_enc.h:105 */
SAF_API void ambi_enc_setSourceAzi_deg(void* const hAmbi, int index, float newAzi_deg);    /* ambi_enc.h:114 */
SAF_API void ambi_enc_setSourceElev_deg(void* const hAmbi, int index, float newElev_deg);  /* ambi_enc.h:123 */
SAF_API void ambi_enc_setNumSources(void* const hAmbi, int new_nSources);        /* ambi_enc.h:126 */
SAF_API void ambi_enc_setInputConfigPreset(void* const hAmbi, int newPresetID);  /* ambi_enc.h:129 */
SAF_API void ambi_enc_setChOrder(void* const hAmbi, int newOrder);               /* ambi_enc.h:135 */
SAF_API void ambi_enc_setNormType(void* const hAmbi, int newType);               /* ambi_enc.h:141 */
SAF_API void ambi_enc_setEnablePostScaling(void* const hAmbi, int newStatus);    /* ambi_enc.h:147 */
SAF_API void ambi_enc_setSourceGain(void* const hAmbi, int srcIdx, float newGain); /* ambi_enc.h:153 */
SAF_API void ambi_enc_setSourceSolo(void* const hAmbi, int srcIdx);              /* ambi_enc.h:158 */
SAF_API void ambi_enc_setUnSolo(void* const hAmbi);                              /* ambi_enc.h:163 */
SAF_API int  ambi_enc_getFrameSize(void);                                        /* ambi_enc.h:174 */
SAF_API int  ambi_enc_getOutputOrder(void* const hAmbi);                         /* ambi_enc.h:183 */
SAF_API float ambi_enc_getSourceAzi_deg(void* const hAmbi, int index);           /* ambi_enc.h:186 */
SAF_API float ambi_enc_getSourceElev_deg(void* const hAmbi, int index);          /* ambi_enc.h:189 */
SAF_API int  ambi_enc_getNumSources(void* const hAmbi);                          /* ambi_enc.h:192 */
SAF_API int  ambi_enc_getMaxNumSources(void);                                    /* ambi_enc.h:195 */
SAF_API int  ambi_enc_getNSHrequired(void* const hAmbi);                         /* ambi_enc.h:201 */
SAF_API int  ambi_enc_getChOrder(void* const hAmbi);                             /* ambi_enc.h:207 */
SAF_API int  ambi_enc_getNormType(void* const hAmbi);                            /* ambi_enc.h:213 */
SAF_API int  ambi_enc_getEnablePostScaling(void* const hAmbi);                   /* ambi_enc.h:216 */
SAF_API int  ambi_enc_getProcessingDelay(void);                                  /* ambi_enc.h:222 */

/* ---- batched, device-resident entry point: nInst ambi_enc handles with the same block size; one call
 * encodes nFrames consecutive blocks of every instance (each output block is the encoding of the block
 * before it, exactly like consecutive ambi_enc_process calls; direction changes made through the set
 * functions take effect, cross-faded, on the first block of the next call).  Sample addressing as for
 * saf_hip_ambi_dec_batch_process; `nInputs` / `nOutputs` are the channel rows present in d_in / d_out.
 * d_in and d_out must not overlap (no in-place use: one launch reads block f while another workgroup writes block f);
 * the same holds for saf_hip_rotator_process_dev and saf_hip_beamformer_process_dev.  Overlap aborts with a message. */
SAF_API void* saf_hip_ambi_enc_batch_create(void* const* hAmbis, int nInst, int maxFramesPerCall);
SAF_API void  saf_hip_ambi_enc_batch_destroy(void** const phBatch);
SAF_API void  saf_hip_ambi_enc_batch_process(void* const hBatch,
                                             const float* d_in, long long in_inst_stride, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                             float* d_out, long long out_inst_stride, long long out_frame_stride, long long out_ch_stride, int nOutputs,
                                             int nFrames);

/* ========================================================================== */
/*      matrix convolver (saf_utility_matrixConv.h:55-86)                     */
/* ========================================================================== */
/** H is nCHout x nCHin x length_h (host, copied); inputSig nCHin x hopSize, outputSig nCHout x hopSize (host, flat). */
SAF_API void saf_matrixConv_create(void** const phMC, int hopSize, float* H, int length_h, int nCHin, int nCHout, int usePartFLAG); /* saf_utility_matrixConv.h:55 */
SAF_API void saf_matrixConv_destroy(void** const phMC);                                                                           /* saf_utility_matrixConv.h:68 */
SAF_API void saf_matrixConv_apply(void* const hMC, float* inputSig, float* outputSig);                                             /* saf_utility_matrixConv.h:81 */
/** Device-pointer entry point: nBlocks consecutive blocks per call,
 *  in[blk*in_block_stride + ch*in_ch_stride + n], out likewise (floats).  State carries over between calls exactly as
 *  between consecutive saf_matrixConv_apply calls.  Handles are created for at most the number of blocks per call set by
 *  saf_hip_matrixConv_setMaxBlocksPerCall (process-wide, read by saf_matrixConv_create; default 1). */
SAF_API void saf_hip_matrixConv_setMaxBlocksPerCall(int nBlocks);
SAF_API void saf_hip_matrixConv_apply_dev(void* const hMC, const float* d_in, long long in_ch_stride, long long in_block_stride,
                                          float* d_out, long long out_ch_stride, long long out_block_stride, int nBlocks);

/* Multi-channel convolver: channel c is filtered by H[c] (saf_utility_matrixConv.h:109-137 / .c:257-416).
 *   H: nCH x length_h;  inputSig / outputSig: nCH x hopSize */
SAF_API void saf_multiConv_create(void** const phMC, int hopSize, float* H, int length_h, int nCH, int usePartFLAG);   /* saf_utility_matrixConv.h:109 */
SAF_API void saf_multiConv_destroy(void** const phMC);                                                                /* saf_utility_matrixConv.h:122 */
SAF_API void saf_multiConv_apply(void* const hMC, float* inputSig, float* outputSig);                                 /* saf_utility_matrixConv.h:132 */
SAF_API void saf_hip_multiConv_apply_dev(void* const hMC, const float* d_in, long long in_ch_stride, long long in_block_stride,
                                         float* d_out, long long out_ch_stride, long long out_block_stride, int nBlocks);

/* Time-varying convolver: one input, nCHout outputs, nIRs filter sets selected per block by irIdx with a linear
 * cross-fade one block later (saf_utility_matrixConv.h:157-200 / .c:438-620).
 *   H: nIRs pointers to FLAT(nCHout x length_h);  inputSig: hopSize;  outputSig: nCHout x hopSize */
SAF_API void saf_TVConv_create(void** const phTVC, int hopSize, float** H, int length_h, int nIRs, int nCHout, int initIdx);   /* saf_utility_matrixConv.h:157 */
SAF_API void saf_TVConv_destroy(void** const phTVC);                                                                          /* saf_utility_matrixConv.h:171 */
SAF_API void saf_TVConv_apply(void* const hTVC, float* inputSig, float* outputSig, int irIdx);                                /* saf_utility_matrixConv.h:185 */
/** nBlocks consecutive blocks of device-resident samples; irIdx is a HOST array of nBlocks indices. */
SAF_API void saf_hip_TVConv_apply_dev(void* const hTVC, const float* d_in, long long in_block_stride,
                                      float* d_out, long long out_ch_stride, long long out_block_stride, const int* irIdx, int nBlocks);

/* ========================================================================== */
/*      HRIR processing (saf_hrir.h) and Voronoi weights (saf_utility_geometry.h) */
/* ========================================================================== */
SAF_API void estimateITDs(float* hrirs, int N_dirs, int hrir_len, int fs, float* itds_s);                                              /* saf_hrir.h:94 / saf_hrir.c:40 */
SAF_API void HRIRs2HRTFs_afSTFT(float* hrirs, int N_dirs, int hrir_len, int hopsize, int LDmode, int hybridmode, float_complex* hrtf_fb); /* saf_hrir.h:113 / saf_hrir.c:110 */
SAF_API void diffuseFieldEqualiseHRTFs(int N_dirs, float* itds_s, float* centreFreq, int N_bands, float* weights, int applyEQ, int applyPhase, float_complex* hrtfs); /* saf_hrir.h:186 / saf_hrir.c:173 */
SAF_API void getVoronoiWeights(float* dirs_deg, int nDirs, int diagFLAG, float* weights);                                              /* saf_utility_geometry.h:430 / .c:937 */
/** Installs the HRIR set used wherever the reference would read its built-in default set (__default_hrirs & co.,
 *  saf_hrir.h:49-61; the data file is absent from the reference checkout, so this library ships none).
 *  hrirs: N x 2 x len, dirs: N x 2 [azimuth, elevation] degrees.  Copied; applies to codecs initialised afterwards. */
SAF_API void saf_hip_setDefaultHRIRs(const float* hrirs, const float* hrir_dirs_deg, int N_hrir_dirs, int hrir_len, int hrir_fs);

/* ========================================================================== */
/*      binauraliser (examples/include/binauraliser.h:73-376)                 */
/* ========================================================================== */
typedef enum { INTERP_TRI = 1, INTERP_TRI_PS } INTERP_MODES;                     /* binauraliser.h:58-61 */
/** The reference fixes the block size (-DBINAURALISER_FRAME_SIZE, default 128) and the source cap (MAX_NUM_INPUTS = 64)
 *  at compile time; here both are process-wide settings read by binauraliser_create. */
SAF_API void saf_hip_binauraliser_setFrameSize(int frameSize);
SAF_API void saf_hip_binauraliser_setMaxNumSources(int maxSources);

SAF_API void binauraliser_create(void** const phBin);                            /* binauraliser.h:73 */
SAF_API void binauraliser_destroy(void** const phBin);                           /* binauraliser.h:80 */
SAF_API void binauraliser_init(void* const hBin, int samplerate);                /* binauraliser.h:90 */
SAF_API void binauraliser_initCodec(void* const hBin);                           /* binauraliser.h:108 */
SAF_API void binauraliser_process(void* const hBin, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples); /* binauraliser.h:120 */
SAF_API void binauraliser_refreshSettings(void* const hBin);                     /* binauraliser.h:136 */
SAF_API void binauraliser_setSourceAzi_deg(void* const hBin, int index, float newAzi_deg);   /* binauraliser.h:139 */
SAF_API void binauraliser_setSourceElev_deg(void* const hBin, int index, float newElev_deg); /* binauraliser.h:144 */
SAF_API void binauraliser_setNumSources(void* const hBin, int new_nSources);     /* binauraliser.h:149 */
SAF_API void binauraliser_setUseDefaultHRIRsflag(void* const hBin, int newState);/* binauraliser.h:159 */
SAF_API void binauraliser_setSofaFilePath(void* const hBin, const char* path);   /* binauraliser.h:172 */
SAF_API void binauraliser_setEnableHRIRsDiffuseEQ(void* const hBin, int newState);/* binauraliser.h:175 */
SAF_API void binauraliser_setInputConfigPreset(void* const hBin, int newPresetID);/* binauraliser.h:178 */
SAF_API void binauraliser_setEnableRotation(void* const hBin, int newState);     /* binauraliser.h:181 */
SAF_API void binauraliser_setYaw(void* const hBin, float newYaw);                /* binauraliser.h:184 */
SAF_API void binauraliser_setPitch(void* const hBin, float newPitch);            /* binauraliser.h:187 */
SAF_API void binauraliser_setRoll(void* const hBin, float newRoll);              /* binauraliser.h:190 */
SAF_API void binauraliser_setFlipYaw(void* const hBin, int newState);            /* binauraliser.h:196 */
SAF_API void binauraliser_setFlipPitch(void* const hBin, int newState);          /* binauraliser.h:202 */
SAF_API void binauraliser_setFlipRoll(void* const hBin, int newState);           /* binauraliser.h:208 */
SAF_API void binauraliser_setRPYflag(void* const hBin, int newState);            /* binauraliser.h:214 */
SAF_API void binauraliser_setInterpMode(void* const hBin, int newMode);          /* binauraliser.h:217 */
SAF_API void binauraliser_setSourceGain(void* const hBin, int srcIdx, float newGain); /* binauraliser.h:222 */
SAF_API void binauraliser_setSourceSolo(void* const hBin, int srcIdx);           /* binauraliser.h:227 */
SAF_API void binauraliser_setUnSolo(void* const hBin);                           /* binauraliser.h:232 */
SAF_API int  binauraliser_getFrameSize(void);                                    /* binauraliser.h:243 */
SAF_API CODEC_STATUS binauraliser_getCodecStatus(void* const hBin);              /* binauraliser.h:246 */
SAF_API float binauraliser_getProgressBar0_1(void* const hBin);                  /* binauraliser.h:253 */
SAF_API void binauraliser_getProgressBarText(void* const hBin, char* text);      /* binauraliser.h:261 */
SAF_API float binauraliser_getSourceAzi_deg(void* const hBin, int index);        /* binauraliser.h:264 */
SAF_API float binauraliser_getSourceElev_deg(void* const hBin, int index);       /* binauraliser.h:267 */
SAF_API int  binauraliser_getNumSources(void* const hBin);                       /* binauraliser.h:270 */
SAF_API int  binauraliser_getMaxNumSources(void);                                /* binauraliser.h:273 */
SAF_API int  binauraliser_getNumEars(void);                                      /* binauraliser.h:276 */
SAF_API int  binauraliser_getNDirs(void* const hBin);                            /* binauraliser.h:279 */
SAF_API int  binauraliser_getNTriangles(void* const hBin);                       /* binauraliser.h:285 */
SAF_API float binauraliser_getHRIRAzi_deg(void* const hBin, int index);          /* binauraliser.h:288 */
SAF_API float binauraliser_getHRIRElev_deg(void* const hBin, int index);         /* binauraliser.h:291 */
SAF_API int  binauraliser_getHRIRlength(void* const hBin);                       /* binauraliser.h:294 */
SAF_API int  binauraliser_getHRIRsamplerate(void* const hBin);                   /* binauraliser.h:297 */
SAF_API int  binauraliser_getUseDefaultHRIRsflag(void* const hBin);              /* binauraliser.h:307 */
SAF_API char* binauraliser_getSofaFilePath(void* const hBin);                    /* binauraliser.h:319 */
SAF_API int  binauraliser_getEnableHRIRsDiffuseEQ(void* const hBin);             /* binauraliser.h:325 */
SAF_API int  binauraliser_getDAWsamplerate(void* const hBin);                    /* binauraliser.h:328 */
SAF_API int  binauraliser_getEnableRotation(void* const hBin);                   /* binauraliser.h:334 */
SAF_API float binauraliser_getYaw(void* const hBin);                             /* binauraliser.h:337 */
SAF_API float binauraliser_getPitch(void* const hBin);                           /* binauraliser.h:340 */
SAF_API float binauraliser_getRoll(void* const hBin);                            /* binauraliser.h:343 */
SAF_API int  binauraliser_getFlipYaw(void* const hBin);                          /* binauraliser.h:349 */
SAF_API int  binauraliser_getFlipPitch(void* const hBin);                        /* binauraliser.h:355 */
SAF_API int  binauraliser_getFlipRoll(void* const hBin);                         /* binauraliser.h:361 */
SAF_API int  binauraliser_getRPYflag(void* const hBin);                          /* binauraliser.h:367 */
SAF_API int  binauraliser_getInterpMode(void* const hBin);                       /* binauraliser.h:370 */
SAF_API int  binauraliser_getProcessingDelay(void);                              /* binauraliser.h:376 */
/** Device-pointer entry: nFrames consecutive blocks, in[frame*in_frame_stride + ch*in_ch_stride + n], two output rows
 *  out[frame*out_frame_stride + ear*out_ch_stride + n].  State carries over exactly as between binauraliser_process calls. */
SAF_API void saf_hip_binauraliser_process_dev(void* const hBin, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                              float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames);
/** Read-back of the tables binauraliser_data holds (binauraliser_internal.h:95-118), for parity checks. */
/** Batch of nInst initialised binauralisers (same block size, source count, HRIR set, flags): device-resident blocks
 *  in[inst*in_inst + frame*in_frame + ch*in_ch + n] -> out[inst*out_inst + frame*out_frame + ear*out_ch + n]; enqueues only. */
SAF_API void* saf_hip_binauraliser_batch_create(void* const* hBins, int nInst, int maxFramesPerCall);
SAF_API void  saf_hip_binauraliser_batch_destroy(void** const phBatch);
SAF_API void  saf_hip_binauraliser_batch_process(void* const hBatch, const float* d_in, long long in_inst_stride, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                                 float* d_out, long long out_inst_stride, long long out_frame_stride, long long out_ch_stride, int nFrames);
SAF_API void saf_hip_binauraliser_getITDs(void* const hBin, float* itds_s);
SAF_API void saf_hip_binauraliser_getWeights(void* const hBin, float* weights);
SAF_API void saf_hip_binauraliser_getHRTFfb(void* const hBin, float_complex* hrtf_fb);        /* [133][2][N] */
SAF_API void saf_hip_binauraliser_getHRTFinterp(void* const hBin, float_complex* hrtf_interp);/* [nSources][133][2] */

/* ========================================================================== */
/*      binauraliser_nf (examples/include/binauraliser_nf.h:76-194)           */
/* ========================================================================== */
/** A handle from binauraliserNF_create is a binauraliser handle with a distance per source: every binauraliser_* setter and
 *  getter above applies to it (the reference's NF struct begins with the binauraliser members, binauraliser_nf_internal.h:61-137)
 *  and it is accepted by saf_hip_binauraliser_batch_create (all instances of a batch NF, or none).  Sources nearer than the
 *  far-field threshold (34 head radii) get the DVF shelf of their lateral angle and distance on each ear.
 *  binauraliserNF_processFD (binauraliser_nf.h:135) is declared by the reference but has no definition there: its
 *  binauraliserNF_process is the frequency-domain version (binauraliser_nf.c:224), so here both names are one function. */
SAF_API void binauraliserNF_create(void** const phBin);                              /* binauraliser_nf.h:76 */
SAF_API void binauraliserNF_destroy(void** const phBin);                             /* binauraliser_nf.h:83 */
SAF_API void binauraliserNF_init(void* const hBin, int samplerate);                  /* binauraliser_nf.h:93 */
SAF_API void binauraliserNF_initCodec(void* const hBin);                             /* binauraliser_nf.h:112 */
SAF_API void binauraliserNF_process(void* const hBin, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples); /* binauraliser_nf.h:124 */
SAF_API void binauraliserNF_processFD(void* const hBin, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples); /* binauraliser_nf.h:135 */
SAF_API void binauraliserNF_setSourceDist_m(void* const hBin, int index, float newDist_m);   /* binauraliser_nf.h:154 */
SAF_API void binauraliserNF_setInputConfigPreset(void* const hBin, int newPresetID); /* binauraliser_nf.h:163 */
SAF_API float binauraliserNF_getSourceDist_m(void* const hBin, int index);           /* binauraliser_nf.h:177 */
SAF_API float binauraliserNF_getFarfieldThresh_m(void* const hBin);                  /* binauraliser_nf.h:183 */
SAF_API float binauraliserNF_getFarfieldHeadroom(void* const hBin);                  /* binauraliser_nf.h:189 */
SAF_API float binauraliserNF_getNearfieldLimit_m(void* const hBin);                  /* binauraliser_nf.h:194 */
/** Device-pointer entry, layout of saf_hip_binauraliser_process_dev. */
SAF_API void saf_hip_binauraliserNF_process_dev(void* const hBin, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                                float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames);
/** Read-back for parity checks: the filters the band MAC applies, [nSources][133][2] (HRTF x DVF response for near sources). */
SAF_API void saf_hip_binauraliserNF_getHRTFnf(void* const hBin, float_complex* hrtf_nf);

/* ---- near-field DVF filters (framework/modules/saf_utilities/saf_utility_dvf.h:62-153) and the response of a filter at given
 *      frequencies (saf_utility_filters.h, evalIIRTransferFunctionf; saf_utility_filters.c:609-671).  Host functions. ---- */
SAF_API void calcDVFCoeffs(float alpha, float rho, float fs, float* b, float* a);                                /* saf_utility_dvf.h:62 */
SAF_API void interpDVFShelfParams(float theta, float rho, float* iG0, float* iGInf, float* iFc);                 /* saf_utility_dvf.h:82 */
SAF_API void dvfShelfCoeffs(float g0, float gInf, float fc, float fs, float* b0, float* b1, float* a1);          /* saf_utility_dvf.h:102 */
SAF_API void calcDVFShelfParams(int i, float rho, float* g0, float* gInf, float* fc);                            /* saf_utility_dvf.h:124 */
SAF_API void doaToIpsiInteraural(float azimuth, float elevation, float* alphaLR, float* betaLR);                 /* saf_utility_dvf.h:149 */
SAF_API void evalIIRTransferFunctionf(float* b_coeff, float* a_coeff, int nCoeffs, float* freqs, int nFreqs, float fs, int mag2dB,
                                      float* magnitude, float* phase_rad);

/* ========================================================================== */
/*      powermap (examples/include/powermap.h:87-371), PWD mode               */
/* ========================================================================== */
typedef enum { PM_MODE_PWD = 1, PM_MODE_MVDR, PM_MODE_CROPAC_LCMV, PM_MODE_MUSIC, PM_MODE_MUSIC_LOG, PM_MODE_MINNORM, PM_MODE_MINNORM_LOG } POWERMAP_MODES; /* powermap.h:58-74 */
/** The reference fixes the frame size at compile time (-DPOWERMAP_FRAME_SIZE, default 1024); here it is a process-wide
 *  setting read by powermap_create (multiple of 128, at most 2048). */
SAF_API void saf_hip_powermap_setFrameSize(int frameSize);
SAF_API void powermap_create(void** const phPm);                                  /* powermap.h:87 */
SAF_API void powermap_destroy(void** const phPm);                                 /* powermap.h:94 */
SAF_API void powermap_init(void* const hPm, float samplerate);                    /* powermap.h:104 */
SAF_API void powermap_initCodec(void* const hPm);                                 /* powermap.h:122 */
SAF_API void powermap_analysis(void* const hPm, const float* const* inputs, int nInputs, int nSamples, int isPlaying); /* powermap.h:134 */
SAF_API void powermap_refreshSettings(void* const hPm);                           /* powermap.h:149 */
SAF_API void powermap_setPowermapMode(void* const hPm, int newMode);              /* powermap.h:152 */
SAF_API void powermap_setMasterOrder(void* const hPm, int newValue);              /* powermap.h:155 */
SAF_API void powermap_setAnaOrder(void* const hPm, int newValue, int bandIdx);    /* powermap.h:158 */
SAF_API void powermap_setAnaOrderAllBands(void* const hPm, int newValue);         /* powermap.h:161 */
SAF_API void powermap_setPowermapEQ(void* const hPm, float newValue, int bandIdx);/* powermap.h:167 */
SAF_API void powermap_setPowermapEQAllBands(void* const hPm, float newValue);     /* powermap.h:170 */
SAF_API void powermap_setCovAvgCoeff(void* const hPm, float newAvg);              /* powermap.h:173 */
SAF_API void powermap_setChOrder(void* const hPm, int newOrder);                  /* powermap.h:179 */
SAF_API void powermap_setNormType(void* const hPm, int newType);                  /* powermap.h:185 */
SAF_API void powermap_setSourcePreset(void* const hPm, int newPresetID);          /* powermap.h:191 */
SAF_API void powermap_setNumSources(void* const hPm, int newValue);               /* powermap.h:194 */
SAF_API void powermap_setDispFOV(void* const hPm, int newOption);                 /* powermap.h:200 */
SAF_API void powermap_setAspectRatio(void* const hPm, int newOption);             /* powermap.h:206 */
SAF_API void powermap_setPowermapAvgCoeff(void* const hPm, float newValue);       /* powermap.h:209 */
SAF_API void powermap_requestPmapUpdate(void* const hPm);                         /* powermap.h:215 */
SAF_API int  powermap_getFrameSize(void);                                         /* powermap.h:226 */
SAF_API CODEC_STATUS powermap_getCodecStatus(void* const hPm);                    /* powermap.h:229 */
SAF_API float powermap_getProgressBar0_1(void* const hPm);                        /* powermap.h:236 */
SAF_API void powermap_getProgressBarText(void* const hPm, char* text);            /* powermap.h:243 */
SAF_API int  powermap_getMasterOrder(void* const hPm);                            /* powermap.h:248 */
SAF_API int  powermap_getPowermapMode(void* const hPm);                           /* powermap.h:254 */
SAF_API int  powermap_getSamplingRate(void* const hPm);                           /* powermap.h:257 */
SAF_API float powermap_getCovAvgCoeff(void* const hPm);                           /* powermap.h:260 */
SAF_API int  powermap_getNumberOfBands(void);                                     /* powermap.h:263 */
SAF_API int  powermap_getNSHrequired(void* const hPm);                            /* powermap.h:269 */
SAF_API float powermap_getPowermapEQ(void* const hPm, int bandIdx);               /* powermap.h:275 */
SAF_API float powermap_getPowermapEQAllBands(void* const hPm);                    /* powermap.h:280 */
SAF_API void powermap_getPowermapEQHandle(void* const hPm, float** pX_vector, float** pY_values, int* pNpoints); /* powermap.h:290 */
SAF_API int  powermap_getAnaOrder(void* const hPm, int bandIdx);                  /* powermap.h:296 */
SAF_API int  powermap_getAnaOrderAllBands(void* const hPm);                       /* powermap.h:299 */
SAF_API void powermap_getAnaOrderHandle(void* const hPm, float** pX_vector, int** pY_values, int* pNpoints);     /* powermap.h:309 */
SAF_API int  powermap_getChOrder(void* const hPm);                                /* powermap.h:319 */
SAF_API int  powermap_getNormType(void* const hPm);                               /* powermap.h:326 */
SAF_API int  powermap_getNumSources(void* const hPm);                             /* powermap.h:329 */
SAF_API int  powermap_getDispFOV(void* const hPm);                                /* powermap.h:335 */
SAF_API int  powermap_getAspectRatio(void* const hPm);                            /* powermap.h:341 */
SAF_API float powermap_getPowermapAvgCoeff(void* const hPm);                      /* powermap.h:344 */
SAF_API int  powermap_getPmap(void* const hPm, float** grid_dirs, float** pmap, int* nDirs, int* pmapWidth, int* hfov, int* aspectRatio); /* powermap.h:359 */
SAF_API int  powermap_getProcessingDelay(void);                                   /* powermap.h:371 */
/** Device-pointer entry: nFrames whole frames (the input FIFO must be empty), in[frame*in_frame_stride + ch*in_ch_stride + n]. */
SAF_API void saf_hip_powermap_analysis_dev(void* const hPm, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs, int nFrames);
/** Read-back for parity checks: Cx as [133][nSH][nSH]; the smoothed map on the 812-point scanning grid (returns its length). */
SAF_API void saf_hip_powermap_getCx(void* const hPm, float_complex* Cx);
/* ---- batched, device-resident entry point (PWD mode): nInst initialised handles with the same frame size and master order.  One
 * call advances every instance by nFrames frames of device-resident samples x[inst*inst_stride + frame*frame_stride + ch*ch_stride + n]
 * (covariances and filterbank state are the batch's own, zero at creation).  The member handles keep their parameters and receive
 * the maps they asked for with powermap_requestPmapUpdate: powermap_getPmap / saf_hip_powermap_getRawPmap on a member return the
 * batch's result.  The call synchronises only when a map was asked for. */
SAF_API void* saf_hip_powermap_batch_create(void* const* hPms, int nInst, int maxFramesPerCall);
SAF_API void  saf_hip_powermap_batch_destroy(void** const phBatch);
SAF_API void  saf_hip_powermap_batch_analysis(void* const hBatch, const float* d_in, long long in_inst_stride, long long in_frame_stride, long long in_ch_stride, int nInputs, int nFrames);
SAF_API void  saf_hip_powermap_batch_getCx(void* const hBatch, int instIdx, float_complex* Cx);
SAF_API int  saf_hip_powermap_getRawPmap(void* const hPm, float* pmap);
/* Activity-map generators on one nSH x nSH covariance matrix (saf_sh.h / saf_sh.c:1544-1858); host pointers.
 * Y_grid: nSH x nGrid_dirs, passed as complex like in the reference but real-valued (it is built from real SH). */
SAF_API void generatePWDmap(int order, float_complex* Cx, float_complex* Y_grid, int nGrid_dirs, float* pmap);                                                  /* saf_sh.c:1544 */
SAF_API void generateMVDRmap(int order, float_complex* Cx, float_complex* Y_grid, int nGrid_dirs, float regPar, float* pmap, float_complex* w_MVDR_out);       /* saf_sh.c:1586 */
SAF_API void generateCroPaCLCMVmap(int order, float_complex* Cx, float_complex* Y_grid, int nGrid_dirs, float regPar, float lambda, float* pmap);              /* saf_sh.c:1650 */
SAF_API void generateMUSICmap(int order, float_complex* Cx, float_complex* Y_grid, int nSources, int nGrid_dirs, int logScaleFlag, float* pmap);               /* saf_sh.c:1754 */
SAF_API void generateMinNormMap(int order, float_complex* Cx, float_complex* Y_grid, int nSources, int nGrid_dirs, int logScaleFlag, float* pmap);             /* saf_sh.c:1801 */
/* Scanning objects with peak search (saf_sh.h; saf_sh.c:1042-1306): steering vectors = orthonormal real SH of the grid. */
SAF_API void sphPWD_create(void** const phPWD, int order, float* grid_dirs_deg, int nDirs);                      /* saf_sh.c:1042 */
SAF_API void sphPWD_destroy(void** const phPWD);                                                                 /* saf_sh.c:1088 */
SAF_API void sphPWD_compute(void* const hPWD, float_complex* Cx, int nSrcs, float* P_map, int* peak_inds);        /* saf_sh.c:1109 */
SAF_API void sphMUSIC_create(void** const phMUSIC, int order, float* grid_dirs_deg, int nDirs);                  /* saf_sh.c:1172 */
SAF_API void sphMUSIC_destroy(void** const phMUSIC);                                                             /* saf_sh.c:1220 */
SAF_API void sphMUSIC_compute(void* const hMUSIC, float_complex* Vn, int nSrcs, float* P_music, int* peak_inds);  /* saf_sh.c:1243 */

/* ========================================================================== */
/*      panner (examples/include/panner.h:83-325) + getPvalues                 */
/* ========================================================================== */
#define PANNER_SPREAD_MIN_VALUE ( 0.0f )                                           /* panner.h:68 */
#define PANNER_SPREAD_MAX_VALUE ( 90.0f )                                          /* panner.h:71 */
SAF_API void getPvalues(float DTT, float* freq, int nFreq, float* pValues);        /* saf_vbap.h:292 */
/** Replaces -DPANNER_FRAME_SIZE (panner_internal.h:67-73); call before panner_create. */
SAF_API void saf_hip_panner_setFrameSize(int frameSize);
SAF_API void panner_create(void** const phPan);                                    /* panner.h:83 */
SAF_API void panner_destroy(void** const phPan);                                   /* panner.h:90 */
SAF_API void panner_init(void* const hPan, int samplerate);                        /* panner.h:100 */
SAF_API void panner_initCodec(void* const hPan);                                   /* panner.h:118 */
SAF_API void panner_process(void* const hPan, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples); /* panner.h:144 */
SAF_API void panner_refreshSettings(void* const hPan);                             /* panner.h:160 */
SAF_API void panner_setSourceAzi_deg(void* const hPan, int index, float newAzi_deg);   /* panner.h:163 */
SAF_API void panner_setSourceElev_deg(void* const hPan, int index, float newElev_deg); /* panner.h:166 */
SAF_API void panner_setNumSources(void* const hPan, int new_nSources);             /* panner.h:169 */
SAF_API void panner_setLoudspeakerAzi_deg(void* const hPan, int index, float newAzi_deg);   /* panner.h:172 */
SAF_API void panner_setLoudspeakerElev_deg(void* const hPan, int index, float newElev_deg); /* panner.h:175 */
SAF_API void panner_setNumLoudspeakers(void* const hPan, int new_nLoudspeakers);   /* panner.h:178 */
SAF_API void panner_setOutputConfigPreset(void* const hPan, int newPresetID);      /* panner.h:184 */
SAF_API void panner_setInputConfigPreset(void* const hPan, int newPresetID);       /* panner.h:189 */
SAF_API void panner_setDTT(void* const hPan, float newValue);                      /* panner.h:200 */
SAF_API void panner_setSpread(void* const hPan, float newValue);                   /* panner.h:203 */
SAF_API void panner_setYaw(void* const hPan, float newYaw);                        /* panner.h:206 */
SAF_API void panner_setPitch(void* const hPan, float newPitch);                    /* panner.h:209 */
SAF_API void panner_setRoll(void* const hPan, float newRoll);                      /* panner.h:212 */
SAF_API void panner_setFlipYaw(void* const hPan, int newState);                    /* panner.h:218 */
SAF_API void panner_setFlipPitch(void* const hPan, int newState);                  /* panner.h:224 */
SAF_API void panner_setFlipRoll(void* const hPan, int newState);                   /* panner.h:230 */
SAF_API int  panner_getFrameSize(void);                                            /* panner.h:241 */
SAF_API CODEC_STATUS panner_getCodecStatus(void* const hPan);                      /* panner.h:244 */
SAF_API float panner_getProgressBar0_1(void* const hPan);                          /* panner.h:251 */
SAF_API void panner_getProgressBarText(void* const hPan, char* text);              /* panner.h:259 */
SAF_API float panner_getSourceAzi_deg(void* const hPan, int index);                /* panner.h:262 */
SAF_API float panner_getSourceElev_deg(void* const hPan, int index);               /* panner.h:265 */
SAF_API int  panner_getNumSources(void* const hPan);                               /* panner.h:268 */
SAF_API int  panner_getMaxNumSources(void);                                        /* panner.h:271 */
SAF_API float panner_getLoudspeakerAzi_deg(void* const hPan, int index);           /* panner.h:274 */
SAF_API float panner_getLoudspeakerElev_deg(void* const hPan, int index);          /* panner.h:277 */
SAF_API int  panner_getNumLoudspeakers(void* const hPan);                          /* panner.h:280 */
SAF_API int  panner_getMaxNumLoudspeakers(void);                                   /* panner.h:283 */
SAF_API int  panner_getDAWsamplerate(void* const hPan);                            /* panner.h:286 */
SAF_API float panner_getDTT(void* const hPan);                                     /* panner.h:289 */
SAF_API float panner_getSpread(void* const hPan);                                  /* panner.h:292 */
SAF_API float panner_getYaw(void* const hPan);                                     /* panner.h:295 */
SAF_API float panner_getPitch(void* const hPan);                                   /* panner.h:298 */
SAF_API float panner_getRoll(void* const hPan);                                    /* panner.h:301 */
SAF_API int  panner_getFlipYaw(void* const hPan);                                  /* panner.h:307 */
SAF_API int  panner_getFlipPitch(void* const hPan);                                /* panner.h:313 */
SAF_API int  panner_getFlipRoll(void* const hPan);                                 /* panner.h:319 */
SAF_API int  panner_getProcessingDelay(void);                                      /* panner.h:325 */
/** Device-pointer entry: nFrames consecutive blocks, in[frame*in_frame_stride + ch*in_ch_stride + n] (same for out); enqueues only. */
SAF_API void saf_hip_panner_process_dev(void* const hPan, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                        float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames);
/** Read-back for parity checks: G_src as [133][64][64] (band, source, loudspeaker), panner_internal.h:99. */
SAF_API void saf_hip_panner_getGains(void* const hPan, float* G);

/* ========================================================================== */
/*      matrixconv / multiconv example operators (sample-wise FIFO around the convolvers)  */
/* ========================================================================== */
SAF_API void matrixconv_create(void** const phMCnv);                                                                  /* matrixconv.h:53 */
SAF_API void matrixconv_destroy(void** const phMCnv);                                                                 /* matrixconv.h:60 */
SAF_API void matrixconv_init(void* const hMCnv, int samplerate, int hostBlockSize);                                   /* matrixconv.h:69 */
SAF_API void matrixconv_process(void* const hMCnv, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples); /* matrixconv.h:83 */
SAF_API void matrixconv_refreshParams(void* const hMCnv);                                                             /* matrixconv.h:99 */
SAF_API void matrixconv_checkReInit(void* const hMCnv);                                                               /* matrixconv.h:104 */
SAF_API void matrixconv_setFilters(void* const hMCnv, const float** H, int numChannels, int numSamples, int sampleRate); /* matrixconv.h:124 */
SAF_API void matrixconv_setEnablePart(void* const hMCnv, int newState);                                               /* matrixconv.h:131 */
SAF_API void matrixconv_setNumInputChannels(void* const hMCnv, int newValue);                                         /* matrixconv.h:141 */
SAF_API int  matrixconv_getEnablePart(void* const hMCnv);                                                             /* matrixconv.h:158 */
SAF_API int  matrixconv_getNumInputChannels(void* const hMCnv);                                                       /* matrixconv.h:161 */
SAF_API int  matrixconv_getNumOutputChannels(void* const hMCnv);                                                      /* matrixconv.h:167 */
SAF_API int  matrixconv_getFrameSize(void);     /* matrixconv.h:152 — declared there, defined nowhere in the reference; here: the smallest internal block (MIN_FRAME_SIZE, 512) */
SAF_API int  matrixconv_getHostBlockSize(void* const hMCnv);                                                          /* matrixconv.h:170 */
SAF_API int  matrixconv_getNfilters(void* const hMCnv);                                                               /* matrixconv.h:176 */
SAF_API int  matrixconv_getFilterLength(void* const hMCnv);                                                           /* matrixconv.h:179 */
SAF_API int  matrixconv_getFilterFs(void* const hMCnv);                                                               /* matrixconv.h:182 */
SAF_API int  matrixconv_getHostFs(void* const hMCnv);                                                                 /* matrixconv.h:185 */
SAF_API int  matrixconv_getProcessingDelay(void* const hMCnv);                                                        /* matrixconv.h:191 */
SAF_API void multiconv_create(void** const phMCnv);                                                                   /* multiconv.h:53 */
SAF_API void multiconv_destroy(void** const phMCnv);                                                                  /* multiconv.h:60 */
SAF_API void multiconv_init(void* const hMCnv, int samplerate, int hostBlockSize);                                    /* multiconv.h:69 */
SAF_API void multiconv_process(void* const hMCnv, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples); /* multiconv.h:83 */
SAF_API void multiconv_refreshParams(void* const hMCnv);                                                              /* multiconv.h:99 */
SAF_API void multiconv_checkReInit(void* const hMCnv);                                                                /* multiconv.h:104 */
SAF_API void multiconv_setFilters(void* const hMCnv, const float** H, int numChannels, int numSamples, int sampleRate); /* multiconv.h:117 */
SAF_API void multiconv_setEnablePart(void* const hMCnv, int newState);                                                /* multiconv.h:124 */
SAF_API void multiconv_setNumChannels(void* const hMCnv, int newValue);                                               /* multiconv.h:127 */
SAF_API int  multiconv_getEnablePart(void* const hMCnv);                                                              /* multiconv.h:144 */
SAF_API int  multiconv_getNumChannels(void* const hMCnv);                                                             /* multiconv.h:147 */
SAF_API int  multiconv_getFrameSize(void);      /* multiconv.h:138 — as above */
SAF_API int  multiconv_getHostBlockSize(void* const hMCnv);                                                           /* multiconv.h:150 */
SAF_API int  multiconv_getNfilters(void* const hMCnv);                                                                /* multiconv.h:153 */
SAF_API int  multiconv_getFilterLength(void* const hMCnv);                                                            /* multiconv.h:156 */
SAF_API int  multiconv_getFilterFs(void* const hMCnv);                                                                /* multiconv.h:159 */
SAF_API int  multiconv_getHostFs(void* const hMCnv);                                                                  /* multiconv.h:162 */
SAF_API int  multiconv_getProcessingDelay(void* const hMCnv);                                                         /* multiconv.h:168 */

SAF_API void tvconv_create(void** const phTVCnv);                                                                     /* tvconv.h:42 */
SAF_API void tvconv_destroy(void** const phTVCnv);                                                                    /* tvconv.h:49 */
SAF_API void tvconv_init(void* const hTVCnv, int samplerate, int hostBlockSize);                                      /* tvconv.h:58 */
SAF_API void tvconv_process(void* const hTVCnv, float** const inputs, float** const outputs, int nInputs, int nOutputs, int nSamples);   /* tvconv.h:72 */
SAF_API void tvconv_refreshParams(void* const hTVCnv);                                                                /* tvconv.h:88 */
SAF_API void tvconv_checkReInit(void* const hTVCnv);                                                                  /* tvconv.h:93 */
SAF_API void tvconv_setFiltersAndPositions(void* const hTVCnv);                                                       /* tvconv.h:96 */
SAF_API void tvconv_setSofaFilePath(void* const hTVCnv, const char* path);                                            /* tvconv.h:99 */
SAF_API void tvconv_setTargetPosition(void* const hTVCnv, float position, int dim);                                   /* tvconv.h:108 */
SAF_API int  tvconv_getNumInputChannels(void* const hTVCnv);                                                          /* tvconv.h:123 */
SAF_API int  tvconv_getNumOutputChannels(void* const hTVCnv);                                                         /* tvconv.h:129 */
SAF_API int  tvconv_getFrameSize(void);         /* tvconv.h:119 — as above */
SAF_API int  tvconv_getHostBlockSize(void* const hTVCnv);                                                             /* tvconv.h:132 */
SAF_API int  tvconv_getNumIRs(void* const hTVCnv);                                                                    /* tvconv.h:135 */
SAF_API int  tvconv_getNumListenerPositions(void* const hTVCnv);                                                      /* tvconv.h:138 */
SAF_API float tvconv_getListenerPosition(void* const hTVCnv, int index, int dim);                                     /* tvconv.h:141 */
SAF_API int  tvconv_getListenerPositionIdx(void* const hTVCnv);                                                       /* tvconv.h:144 */
SAF_API float tvconv_getTargetPosition(void* const hTVCnv, int dim);                                                  /* tvconv.h:147 */
SAF_API float tvconv_getSourcePosition(void* const hTVCnv, int dim);                                                  /* tvconv.h:150 */
SAF_API float tvconv_getMinDimension(void* const hTVCnv, int dim);                                                    /* tvconv.h:153 */
SAF_API float tvconv_getMaxDimension(void* const hTVCnv, int dim);                                                    /* tvconv.h:156 */
SAF_API int  tvconv_getIRLength(void* const hTVCnv);                                                                  /* tvconv.h:159 */
SAF_API int  tvconv_getIRFs(void* const hTVCnv);                                                                      /* tvconv.h:162 */
SAF_API int  tvconv_getHostFs(void* const hTVCnv);                                                                    /* tvconv.h:165 */
SAF_API int  tvconv_getProcessingDelay(void* const hTVCnv);                                                           /* tvconv.h:171 */
SAF_API char* tvconv_getSofaFilePath(void* const hTVCnv);                                                             /* tvconv.h:174 */
SAF_API CODEC_STATUS tvconv_getCodecStatus(void* const hTVCnv);                                                       /* tvconv.h */
/** Installs what the reference reads from a SOFA file (tvconv.c:262-312): irs[nListenerPositions] each FLAT(nIrChannels x irLength),
 *  listenerPositions [n][3] (Cartesian), sourcePosition [3] (may be NULL). */
SAF_API void saf_hip_tvconv_setIRsAndPositions(void* const hTVCnv, const float* const* irs, const float* listenerPositions, const float* sourcePosition,
                                               int nListenerPositions, int nIrChannels, int irLength, int irFs);
/* ========================================================================== */
/*      real FFT object (saf_utility_fft.h / saf_utility_fft.c:531-753)        */
/* ========================================================================== */
/* N even, N/2 = 2^a 3^b 5^c (x primes <= 31): every size of test__saf_rfft.  forward: unscaled, N/2+1 bins;
 * backward: scaled 1/N, imaginary parts of DC and Nyquist ignored.  Host pointers. */
SAF_API void saf_rfft_create(void** const phFFT, int N);                                   /* saf_utility_fft.h (saf_utility_fft.c:531) */
SAF_API void saf_rfft_destroy(void** const phFFT);                                         /* saf_utility_fft.c:642 */
SAF_API void saf_rfft_forward(void* const hFFT, float* inputTD, float_complex* outputFD);  /* saf_utility_fft.c:690 */
SAF_API void saf_rfft_backward(void* const hFFT, float_complex* inputFD, float* outputTD); /* saf_utility_fft.c:728 */

/* ========================================================================== */
/*      binaural Ambisonic decoding: design functions and the ambi_bin operator  */
/* ========================================================================== */
typedef enum {                                                                    /* saf_hoa.h:134-171 */
    BINAURAL_DECODER_DEFAULT = 0, BINAURAL_DECODER_LS, BINAURAL_DECODER_LSDIFFEQ, BINAURAL_DECODER_SPR, BINAURAL_DECODER_TA, BINAURAL_DECODER_MAGLS
} BINAURAL_AMBI_DECODER_METHODS;
typedef enum { SH_ORDER_FIRST = 1, SH_ORDER_SECOND, SH_ORDER_THIRD, SH_ORDER_FOURTH, SH_ORDER_FIFTH, SH_ORDER_SIXTH, SH_ORDER_SEVENTH } SH_ORDERS;   /* _common.h:38-48 */
typedef enum { DECODING_METHOD_LS = 1, DECODING_METHOD_LSDIFFEQ, DECODING_METHOD_SPR, DECODING_METHOD_TA, DECODING_METHOD_MAGLS } AMBI_BIN_DECODING_METHODS;   /* ambi_bin.h:126-135 */
typedef enum { HRIR_PREPROC_OFF = 1, HRIR_PREPROC_EQ, HRIR_PREPROC_PHASE, HRIR_PREPROC_ALL } AMBI_BIN_PREPROC;                                               /* ambi_bin.h:141-146 */
SAF_API void getSHrotMtxReal(float Rxyz[3][3], float* RotMtx, int L);                                                  /* saf_sh.h / saf_sh.c:479 */
SAF_API void yawPitchRoll2Rzyx(float yaw, float pitch, float roll, int rollPitchYawFLAG, float R[3][3]);               /* saf_utility_geometry.c:213 */
SAF_API void beamWeightsMaxEV(int N, float* b_n);                                                                      /* saf_sh.c:751 */
SAF_API void truncationEQ(float* w_n, int order_truncated, int order_target, double* kr, int nBands, float softThreshold, float* gain);   /* saf_hoa.c:269 */
/** hrtfs: N_bands x 2 x N_dirs; decMtx: N_bands x 2 x (order+1)^2 (saf_hoa.h:394-450).  itd_s is accepted and unused, as in the
 *  reference (its time-alignment phase term multiplies the ITD by zero, saf_hoa_internal.c:495-498). */
SAF_API void getBinauralAmbiDecoderMtx(float_complex* hrtfs, float* hrtf_dirs_deg, int N_dirs, int N_bands, BINAURAL_AMBI_DECODER_METHODS method, int order,
                                       float* freqVector, float* itd_s, float* weights, int enableDiffCovMatching, int enableMaxReWeighting, float_complex* decMtx);
/** decFilters: 2 x (order+1)^2 x fftSize time-domain decoding filters; hrtfs: (fftSize/2+1) x 2 x N_dirs (saf_hoa.h:452-500). */
SAF_API void getBinauralAmbiDecoderFilters(float_complex* hrtfs, float* hrtf_dirs_deg, int N_dirs, int fftSize, float fs, BINAURAL_AMBI_DECODER_METHODS method, int order,
                                           float* itd_s, float* weights, int enableDiffCovMatching, int enableMaxReWeighting, float* decFilters);
SAF_API void applyDiffCovMatching(float_complex* hrtfs, float* hrtf_dirs_deg, int N_dirs, int N_bands, int order, float* weights, float_complex* decMtx);   /* saf_hoa.c:502 */
/* ------------------------------------------------------------------------------------------------------------
 * rotator: rotation of an Ambisonic scene (examples/include/rotator.h:55-263; examples/src/rotator/rotator.c).
 * Same block path as ambi_enc with the SH rotation matrix in place of the encoding matrix (one block of latency,
 * linear cross-fade when the rotation changed).  The quaternion helpers are those of saf_utility_geometry.h:35-106.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct _quaternion_data { union { struct { float w, x, y, z; }; float Q[4]; }; } quaternion_data;       /* saf_utility_geometry.h:36-45 */
typedef enum { EULER_ROTATION_Y_CONVENTION, EULER_ROTATION_X_CONVENTION, EULER_ROTATION_YAW_PITCH_ROLL, EULER_ROTATION_ROLL_PITCH_YAW } EULER_ROTATION_CONVENTIONS;   /* :48-54 */
SAF_API void quaternion2rotationMatrix(quaternion_data* Q, float R[3][3]);                                        /* saf_utility_geometry.c:89 */
SAF_API void rotationMatrix2quaternion(float R[3][3], quaternion_data* Q);                                        /* :107 */
SAF_API void euler2Quaternion(float alpha, float beta, float gamma, int degreesFlag, EULER_ROTATION_CONVENTIONS convention, quaternion_data* Q);   /* :123 (the y- and x-conventions are unsupported there too) */
SAF_API void quaternion2euler(quaternion_data* Q, int degreesFlag, EULER_ROTATION_CONVENTIONS convention, float* alpha, float* beta, float* gamma); /* :163 */
/** Replaces -DROTATOR_FRAME_SIZE (default 64); call before rotator_create. */
SAF_API void saf_hip_rotator_setFrameSize(int frameSize);
SAF_API void rotator_create(void** const phRot);                                   /* rotator.h:61 */
SAF_API void rotator_destroy(void** const phRot);                                  /* rotator.h:68 */
SAF_API void rotator_init(void* const hRot, int samplerate);                       /* rotator.h:76 */
SAF_API void rotator_process(void* const hRot, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples);   /* rotator.h:89 */
/** nFrames consecutive blocks of device-resident ACN signals with the rotation set before the call (the cross-fade, if any, is on the first block). */
SAF_API void saf_hip_rotator_process_dev(void* const hRot, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                         float* d_out, long long out_frame_stride, long long out_ch_stride, int nOutputs, int nFrames);
SAF_API int  rotator_getFrameSize(void);                                           /* rotator.h:105 */
SAF_API void rotator_setYaw(void* const hRot, float newYaw);                       /* rotator.h:108-171 */
SAF_API void rotator_setPitch(void* const hRot, float newPitch);
SAF_API void rotator_setRoll(void* const hRot, float newRoll);
SAF_API void rotator_setQuaternionW(void* const hRot, float newValue);
SAF_API void rotator_setQuaternionX(void* const hRot, float newValue);
SAF_API void rotator_setQuaternionY(void* const hRot, float newValue);
SAF_API void rotator_setQuaternionZ(void* const hRot, float newValue);
SAF_API void rotator_setFlipYaw(void* const hRot, int newState);
SAF_API void rotator_setFlipPitch(void* const hRot, int newState);
SAF_API void rotator_setFlipRoll(void* const hRot, int newState);
SAF_API void rotator_setFlipQuaternion(void* const hRot, int newState);
SAF_API void rotator_setChOrder(void* const hRot, int newOrder);
SAF_API void rotator_setNormType(void* const hRot, int newType);
SAF_API void rotator_setOrder(void* const hRot, int newOrder);
SAF_API void rotator_setRPYflag(void* const hRot, int newState);
SAF_API float rotator_getYaw(void* const hRot);                                    /* rotator.h:179-256 */
SAF_API float rotator_getPitch(void* const hRot);
SAF_API float rotator_getRoll(void* const hRot);
SAF_API float rotator_getQuaternionW(void* const hRot);
SAF_API float rotator_getQuaternionX(void* const hRot);
SAF_API float rotator_getQuaternionY(void* const hRot);
SAF_API float rotator_getQuaternionZ(void* const hRot);
SAF_API int  rotator_getFlipYaw(void* const hRot);
SAF_API int  rotator_getFlipPitch(void* const hRot);
SAF_API int  rotator_getFlipRoll(void* const hRot);
SAF_API int  rotator_getFlipQuaternion(void* const hRot);
SAF_API int  rotator_getRPYflag(void* const hRot);
SAF_API int  rotator_getChOrder(void* const hRot);
SAF_API int  rotator_getNormType(void* const hRot);
SAF_API int  rotator_getOrder(void* const hRot);
SAF_API int  rotator_getNSHrequired(void* const hRot);
SAF_API int  rotator_getProcessingDelay(void);

/* ------------------------------------------------------------------------------------------------------------
 * beamformer: static axisymmetric beams over an Ambisonic scene (examples/include/beamformer.h:50-190;
 * examples/src/beamformer/beamformer.c).  Same block path as ambi_enc with the beam weights as the matrix.
 * ---------------------------------------------------------------------------------------------------------- */
typedef enum { STATIC_BEAM_TYPE_CARDIOID = 1, STATIC_BEAM_TYPE_HYPERCARDIOID, STATIC_BEAM_TYPE_MAX_EV } STATIC_BEAM_TYPES;      /* _common.h:166-171 */
SAF_API void beamWeightsCardioid2Spherical(int N, float* b_n);                                                   /* saf_sh.c:716 */
SAF_API void beamWeightsHypercardioid2Spherical(int N, float* b_n);                                              /* saf_sh.c:733 */
SAF_API void rotateAxisCoeffsReal(int order, float* c_n, float theta_0, float phi_0, float* c_nm);               /* saf_sh.c:839 (inclination, azimuth in rad) */
/** Replaces -DBEAMFORMER_FRAME_SIZE (default 128); call before beamformer_create. */
SAF_API void saf_hip_beamformer_setFrameSize(int frameSize);
SAF_API void beamformer_create(void** const phBeam);                               /* beamformer.h:56 */
SAF_API void beamformer_destroy(void** const phBeam);                              /* beamformer.h:63 */
SAF_API void beamformer_init(void* const hBeam, int samplerate);                   /* beamformer.h:71 */
SAF_API void beamformer_process(void* const hBeam, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples);   /* beamformer.h:84 */
/** nFrames consecutive blocks of device-resident ACN signals with the beams set before the call (the cross-fade, if any, is on the first block). */
SAF_API void saf_hip_beamformer_process_dev(void* const hBeam, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                            float* d_out, long long out_frame_stride, long long out_ch_stride, int nOutputs, int nFrames);
SAF_API void beamformer_refreshSettings(void* const hBeam);                        /* beamformer.h:100-134 */
SAF_API void beamformer_setBeamOrder(void* const hBeam, int newValue);
SAF_API void beamformer_setBeamAzi_deg(void* const hBeam, int index, float newAzi_deg);
SAF_API void beamformer_setBeamElev_deg(void* const hBeam, int index, float newElev_deg);
SAF_API void beamformer_setNumBeams(void* const hBeam, int new_nBeams);
SAF_API void beamformer_setChOrder(void* const hBeam, int newOrder);
SAF_API void beamformer_setNormType(void* const hBeam, int newType);
SAF_API void beamformer_setBeamType(void* const hBeam, int newID);
SAF_API int  beamformer_getFrameSize(void);                                        /* beamformer.h:145-189 */
SAF_API int  beamformer_getBeamOrder(void* const hBeam);
SAF_API float beamformer_getBeamAzi_deg(void* const hBeam, int index);
SAF_API float beamformer_getBeamElev_deg(void* const hBeam, int index);
SAF_API int  beamformer_getNumBeams(void* const hBeam);
SAF_API int  beamformer_getMaxNumBeams(void);
SAF_API int  beamformer_getNSHrequired(void* const hBeam);
SAF_API int  beamformer_getChOrder(void* const hBeam);
SAF_API int  beamformer_getNormType(void* const hBeam);
SAF_API int  beamformer_getBeamType(void* const hBeam);
SAF_API int  beamformer_getProcessingDelay(void);

/* ------------------------------------------------------------------------------------------------------------
 * ambi_drc: frequency-dependent dynamic range compression of an Ambisonic scene (examples/include/ambi_drc.h:97-270;
 * examples/src/ambi_drc/ambi_drc.c).  afSTFT analysis -> per-band gain from the omni channel -> synthesis.
 * ---------------------------------------------------------------------------------------------------------- */
/** Replaces -DAMBI_DRC_FRAME_SIZE (default 128); call before ambi_drc_create. */
SAF_API void saf_hip_ambi_drc_setFrameSize(int frameSize);
SAF_API void ambi_drc_create(void** const phAmbi);                                  /* ambi_drc.h:103 */
SAF_API void ambi_drc_destroy(void** const phAmbi);                                 /* ambi_drc.h:110 */
SAF_API void ambi_drc_init(void* const hAmbi, int samplerate);                      /* ambi_drc.h:118 */
SAF_API void ambi_drc_process(void* const hAmbi, const float* const* inputs, float** const outputs, int nCH, int nSamples);   /* ambi_drc.h:131 */
/** nFrames consecutive blocks of device-resident signals (the display ring is not fed by this entry). */
SAF_API void saf_hip_ambi_drc_process_dev(void* const hAmbi, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                          float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames);
SAF_API void ambi_drc_refreshSettings(void* const hAmbi);                           /* ambi_drc.h:146-189 */
SAF_API void ambi_drc_setThreshold(void* const hAmbi, float newValue);
SAF_API void ambi_drc_setRatio(void* const hAmbi, float newValue);
SAF_API void ambi_drc_setKnee(void* const hAmbi, float newValue);
SAF_API void ambi_drc_setInGain(void* const hAmbi, float newValue);
SAF_API void ambi_drc_setOutGain(void* const hAmbi, float newValue);
SAF_API void ambi_drc_setAttack(void* const hAmbi, float newValue);
SAF_API void ambi_drc_setRelease(void* const hAmbi, float newValue);
SAF_API void ambi_drc_setChOrder(void* const hAmbi, int newOrder);
SAF_API void ambi_drc_setNormType(void* const hAmbi, int newType);
SAF_API void ambi_drc_setInputPreset(void* const hAmbi, SH_ORDERS newPreset);
SAF_API int  ambi_drc_getFrameSize(void);                                           /* ambi_drc.h:200-268 */
SAF_API float** ambi_drc_getGainTF(void* const hAmbi);                              /* [133][3000] gain factors of the last 8 s */
SAF_API int  ambi_drc_getGainTFwIdx(void* const hAmbi);
SAF_API int  ambi_drc_getGainTFrIdx(void* const hAmbi);
SAF_API float* ambi_drc_getFreqVector(void* const hAmbi, int* nFreqPoints);
SAF_API float ambi_drc_getThreshold(void* const hAmbi);
SAF_API float ambi_drc_getRatio(void* const hAmbi);
SAF_API float ambi_drc_getKnee(void* const hAmbi);
SAF_API float ambi_drc_getInGain(void* const hAmbi);
SAF_API float ambi_drc_getOutGain(void* const hAmbi);
SAF_API float ambi_drc_getAttack(void* const hAmbi);
SAF_API float ambi_drc_getRelease(void* const hAmbi);
SAF_API int  ambi_drc_getChOrder(void* const hAmbi);
SAF_API int  ambi_drc_getNormType(void* const hAmbi);
SAF_API SH_ORDERS ambi_drc_getInputPreset(void* const hAmbi);
SAF_API int  ambi_drc_getNSHrequired(void* const hAmbi);
SAF_API int  ambi_drc_getSamplerate(void* const hAmbi);
SAF_API int  ambi_drc_getProcessingDelay(void);

/** Replaces -DAMBI_BIN_FRAME_SIZE; call before ambi_bin_create. */
SAF_API void saf_hip_ambi_bin_setFrameSize(int frameSize);
SAF_API void ambi_bin_create(void** const phAmbi);                                 /* ambi_bin.h:161 */
SAF_API void ambi_bin_destroy(void** const phAmbi);                                /* ambi_bin.h:168 */
SAF_API void ambi_bin_init(void* const hAmbi, int samplerate);                     /* ambi_bin.h:178 */
SAF_API void ambi_bin_initCodec(void* const hAmbi);                                /* ambi_bin.h:196 */
SAF_API void ambi_bin_process(void* const hAmbi, const float* const* inputs, float** const outputs, int nInputs, int nOutputs, int nSamples);   /* ambi_bin.h:208 */
SAF_API void ambi_bin_refreshParams(void* const hAmbi);                            /* ambi_bin.h:224 */
SAF_API void ambi_bin_setUseDefaultHRIRsflag(void* const hAmbi, int newState);     /* ambi_bin.h:237 */
SAF_API void ambi_bin_setSofaFilePath(void* const hAmbi, const char* path);        /* ambi_bin.h:250 */
SAF_API void ambi_bin_setInputOrderPreset(void* const hAmbi, SH_ORDERS newPreset); /* ambi_bin.h:260 */
SAF_API void ambi_bin_setDecodingMethod(void* const hAmbi, AMBI_BIN_DECODING_METHODS newMethod);   /* ambi_bin.h:266 */
SAF_API void ambi_bin_setChOrder(void* const hAmbi, int newOrder);                 /* ambi_bin.h:273 */
SAF_API void ambi_bin_setNormType(void* const hAmbi, int newType);                 /* ambi_bin.h:279 */
SAF_API void ambi_bin_setEnableMaxRE(void* const hAmbi, int newState);             /* ambi_bin.h:282 */
SAF_API void ambi_bin_setEnableDiffuseMatching(void* const hAmbi, int newState);   /* ambi_bin.h:285 */
SAF_API void ambi_bin_setEnableTruncationEQ(void* const hAmbi, int newState);      /* ambi_bin.h:288 */
SAF_API void ambi_bin_setHRIRsPreProc(void* const hAmbi, AMBI_BIN_PREPROC newType);/* ambi_bin.h:291 */
SAF_API void ambi_bin_setEnableRotation(void* const hAmbi, int newState);          /* ambi_bin.h:294 */
SAF_API void ambi_bin_setYaw(void* const hAmbi, float newYaw_deg);                 /* ambi_bin.h:297 */
SAF_API void ambi_bin_setPitch(void* const hAmbi, float newPitch);                 /* ambi_bin.h:300 */
SAF_API void ambi_bin_setRoll(void* const hAmbi, float newRoll);                   /* ambi_bin.h:303 */
SAF_API void ambi_bin_setFlipYaw(void* const hAmbi, int newState);                 /* ambi_bin.h:306 */
SAF_API void ambi_bin_setFlipPitch(void* const hAmbi, int newState);               /* ambi_bin.h:309 */
SAF_API void ambi_bin_setFlipRoll(void* const hAmbi, int newState);                /* ambi_bin.h:312 */
SAF_API void ambi_bin_setRPYflag(void* const hAmbi, int newState);                 /* ambi_bin.h:318 */
SAF_API int  ambi_bin_getFrameSize(void);                                          /* ambi_bin.h:329 */
SAF_API CODEC_STATUS ambi_bin_getCodecStatus(void* const hAmbi);                   /* ambi_bin.h:332 */
SAF_API float ambi_bin_getProgressBar0_1(void* const hAmbi);                       /* ambi_bin.h:335 */
SAF_API void ambi_bin_getProgressBarText(void* const hAmbi, char* text);           /* ambi_bin.h:343 */
SAF_API int  ambi_bin_getUseDefaultHRIRsflag(void* const hAmbi);                   /* ambi_bin.h:353 */
SAF_API int  ambi_bin_getInputOrderPreset(void* const hAmbi);                      /* ambi_bin.h:362 */
SAF_API AMBI_BIN_DECODING_METHODS ambi_bin_getDecodingMethod(void* const hAmbi);   /* ambi_bin.h:368 */
SAF_API char* ambi_bin_getSofaFilePath(void* const hAmbi);                         /* ambi_bin.h:378 */
SAF_API int  ambi_bin_getChOrder(void* const hAmbi);                               /* ambi_bin.h:385 */
SAF_API int  ambi_bin_getNormType(void* const hAmbi);                              /* ambi_bin.h:392 */
SAF_API int  ambi_bin_getNumEars(void);                                            /* ambi_bin.h:395 */
SAF_API int  ambi_bin_getNSHrequired(void* const hAmbi);                           /* ambi_bin.h:401 */
SAF_API int  ambi_bin_getEnableMaxRE(void* const hAmbi);                           /* ambi_bin.h:407 */
SAF_API int  ambi_bin_getEnableDiffuseMatching(void* const hAmbi);                 /* ambi_bin.h:413 */
SAF_API int  ambi_bin_getEnableTruncationEQ(void* const hAmbi);                    /* ambi_bin.h:419 */
SAF_API AMBI_BIN_PREPROC ambi_bin_getHRIRsPreProc(void* const hAmbi);              /* ambi_bin.h:425 */
SAF_API int  ambi_bin_getEnableRotation(void* const hAmbi);                        /* ambi_bin.h:431 */
SAF_API float ambi_bin_getYaw(void* const hAmbi);                                  /* ambi_bin.h:434 */
SAF_API float ambi_bin_getPitch(void* const hAmbi);                                /* ambi_bin.h:437 */
SAF_API float ambi_bin_getRoll(void* const hAmbi);                                 /* ambi_bin.h:440 */
SAF_API int  ambi_bin_getFlipYaw(void* const hAmbi);                               /* ambi_bin.h:446 */
SAF_API int  ambi_bin_getFlipPitch(void* const hAmbi);                             /* ambi_bin.h:452 */
SAF_API int  ambi_bin_getFlipRoll(void* const hAmbi);                              /* ambi_bin.h:458 */
SAF_API int  ambi_bin_getRPYflag(void* const hAmbi);                               /* ambi_bin.h:464 */
SAF_API int  ambi_bin_getNDirs(void* const hAmbi);                                 /* ambi_bin.h:467 */
SAF_API int  ambi_bin_getHRIRlength(void* const hAmbi);                            /* ambi_bin.h:470 */
SAF_API int  ambi_bin_getHRIRsamplerate(void* const hAmbi);                        /* ambi_bin.h:473 */
SAF_API int  ambi_bin_getDAWsamplerate(void* const hAmbi);                         /* ambi_bin.h:476 */
SAF_API int  ambi_bin_getProcessingDelay(void);                                    /* ambi_bin.h:482 */
/** Device-pointer entry: nFrames consecutive blocks, in[frame*in_frame_stride + ch*in_ch_stride + n] -> out[frame*..., ear*..., n]. */
SAF_API void saf_hip_ambi_bin_process_dev(void* const hAmbi, const float* d_in, long long in_frame_stride, long long in_ch_stride, int nInputs,
                                          float* d_out, long long out_frame_stride, long long out_ch_stride, int nFrames);
/** Read-back for parity checks: the decoding matrix as [133][2][nSH]. */
SAF_API void saf_hip_ambi_bin_getDecoderMtx(void* const hAmbi, float_complex* M);

#ifdef __cplusplus
}
#endif
#endif /* SAF_HIP_H_INCLUDED */
