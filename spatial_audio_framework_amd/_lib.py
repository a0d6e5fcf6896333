"""ctypes loader for libsaf_hip.so — the only compute path of this package.

There is deliberately no fallback: if the HIP library is missing the import of
any compute entry point fails loudly.
"""
import ctypes as C
import os
import os
import re
from pathlib import Path

PKG = Path(__file__).resolve().parent
# SAF_HIP_LIB: another build of the same library (kernel A/B runs: tools/ab_variants.sh); the default is the in-tree build
SO = Path(os.environ["SAF_HIP_LIB"]).resolve() if os.environ.get("SAF_HIP_LIB") else PKG / "libsaf_hip.so"
HEADER = PKG.parent / "include" / "saf_hip.h"
_lib = None


class SafHipMissing(RuntimeError):
    pass


def declared_symbols():
    """Names of every function include/saf_hip.h declares (lines tagged SAF_API)."""
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"SAF_API\s+[\w\s\*]+?\b(\w+)\s*\(", text)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not SO.exists():
        raise SafHipMissing(
            f"{SO} not found. Build it with `python -m spatial_audio_framework_amd.build` "
            "(hipcc, --offload-arch=gfx950). This package has no CPU fallback.")
    L = C.CDLL(str(SO), mode=C.RTLD_GLOBAL)
    vp, ci, cf, cll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("saf_hip_set_stream", None, vp)
    sig("saf_hip_stopwatch_start", None); sig("saf_hip_stopwatch_stop_ms", C.c_double)
    sig("saf_hip_setZeroCopyIO", None, ci); sig("saf_hip_getZeroCopyIO", ci)
    sig("saf_hip_get_stream", vp)
    sig("saf_hip_synchronize", None)
    sig("saf_hip_device_count", ci)
    sig("saf_hip_set_device", None, ci)
    sig("saf_hip_version", C.c_char_p)
    sig("saf_hip_profile_enable", None, ci)
    sig("saf_hip_profile_reset", None)
    sig("saf_hip_profile_read", ci, C.c_char_p, C.POINTER(C.c_double))
    # afSTFT
    sig("afSTFT_create", None, C.POINTER(vp), ci, ci, ci, ci, ci, ci)
    sig("afAnalyse", None, fp, ci, ci, ci, ci, ci, vp)
    sig("afSTFT_destroy", None, C.POINTER(vp))
    sig("afSTFT_forward_flat", None, vp, fp, ci, vp)
    sig("afSTFT_backward_flat", None, vp, vp, ci, fp)
    sig("afSTFT_forward_knownDimensions", None, vp, C.POINTER(fp), ci, ci, ci, vp)
    sig("afSTFT_backward_knownDimensions", None, vp, vp, ci, ci, ci, C.POINTER(fp))
    sig("afSTFT_forward", None, vp, C.POINTER(fp), ci, vp)
    sig("afSTFT_backward", None, vp, vp, ci, C.POINTER(fp))
    sig("afSTFT_channelChange", None, vp, ci, ci)
    sig("afSTFT_clearBuffers", None, vp)
    sig("afSTFT_getNBands", ci, vp)
    sig("afSTFT_getProcDelay", ci, vp)
    sig("afSTFT_getCentreFreqs", None, vp, cf, ci, fp)
    sig("afSTFT_FIRtoFilterbankCoeffs", None, fp, ci, ci, ci, ci, ci, ci, vp)
    sig("saf_hip_afSTFT_forward_dev", None, vp, vp, cll, ci, vp, cll, cll)
    sig("saf_hip_afSTFT_backward_dev", None, vp, vp, cll, cll, ci, vp, cll)
    # SH / HOA
    for n in ("getSHreal", "getSHreal_recur", "getRSH", "getRSH_recur"):
        sig(n, None, ci, fp, ci, fp)
    sig("saf_hip_getRSH_recur_dev", None, ci, vp, ci, vp)
    sig("getMaxREweights", None, ci, ci, fp)
    sig("convertHOAChannelConvention", None, fp, ci, ci, ci, ci)
    sig("convertHOANormConvention", None, fp, ci, ci, ci, ci)
    sig("getLoudspeakerDecoderMtx", None, fp, ci, ci, ci, ci, fp)
    # VBAP
    sig("findLsTriplets", None, fp, ci, ci, C.POINTER(fp), ip, C.POINTER(ip), ip)
    sig("invertLsMtx3D", None, fp, ip, ci, C.POINTER(fp))
    sig("vbap3D", None, fp, ci, ci, ip, ci, cf, fp, C.POINTER(fp))
    sig("generateVBAPgainTable3D_srcs", None, fp, ci, fp, ci, ci, ci, cf, C.POINTER(fp), ip, ip)
    sig("generateVBAPgainTable3D", None, fp, ci, ci, ci, ci, ci, cf, C.POINTER(fp), ip, ip)
    sig("generateVBAPgainTable2D_srcs", None, fp, ci, fp, ci, C.POINTER(fp), ip, ip)
    sig("generateVBAPgainTable2D", None, fp, ci, ci, C.POINTER(fp), ip, ip)
    sig("findLsPairs", None, fp, ci, C.POINTER(ip), ip)
    sig("invertLsMtx2D", None, fp, ip, ci, C.POINTER(fp))
    sig("vbap2D", None, fp, ci, ci, ip, ci, fp, C.POINTER(fp))
    sig("getSpreadSrcDirs3D", None, cf, cf, cf, ci, ci, fp)
    sig("compressVBAPgainTable3D", None, fp, ci, ci, fp, ip)
    sig("VBAPgainTable2InterpTable", None, fp, ci, ci)
    # ambi_dec
    sig("saf_hip_ambi_dec_setFrameSize", None, ci)
    sig("saf_hip_ambi_dec_setTimeDomainPath", None, ci)
    sig("saf_hip_ambi_dec_getTimeDomainPath", ci)
    sig("saf_hip_ambi_dec_lastPath", ci, vp)
    sig("saf_hip_ambi_dec_batch_lastPath", ci, vp)
    sig("saf_hip_ambi_dec_setOverlap", None, ci)
    sig("saf_hip_ambi_dec_getOverlap", ci)
    sig("saf_hip_ambi_dec_batch_lastOverlap", ci, vp)
    sig("saf_hip_ambi_dec_batch_decodeGiveUps", ci, vp)
    sig("ambi_dec_create", None, C.POINTER(vp))
    sig("ambi_dec_destroy", None, C.POINTER(vp))
    sig("ambi_dec_init", None, vp, ci)
    sig("ambi_dec_initCodec", None, vp)
    sig("ambi_dec_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    sig("ambi_dec_refreshSettings", None, vp)
    for n in ("setMasterDecOrder", "setDecOrderAllBands", "setNumLoudspeakers", "setBinauraliseLSflag", "setUseDefaultHRIRsflag",
              "setEnableHRIRsPreProc", "setSourcePreset", "setOutputConfigPreset", "setChOrder", "setNormType"):
        sig("ambi_dec_" + n, None, vp, ci)
    for n in ("setDecOrder", "setDecMethod", "setDecEnableMaxrE", "setDecNormType"):
        sig("ambi_dec_" + n, None, vp, ci, ci)
    sig("ambi_dec_setLoudspeakerAzi_deg", None, vp, ci, cf)
    sig("ambi_dec_setLoudspeakerElev_deg", None, vp, ci, cf)
    sig("ambi_dec_setSofaFilePath", None, vp, C.c_char_p)
    sig("ambi_dec_setTransitionFreq", None, vp, cf)
    for n in ("getFrameSize", "getNumberOfBands", "getMaxNumLoudspeakers", "getProcessingDelay"):
        sig("ambi_dec_" + n, ci)
    for n in ("getCodecStatus", "getMasterDecOrder", "getDecOrderAllBands", "getNumLoudspeakers", "getNSHrequired", "getBinauraliseLSflag",
              "getUseDefaultHRIRsflag", "getEnableHRIRsPreProc", "getChOrder", "getNormType", "getHRIRsamplerate", "getDAWsamplerate"):
        sig("ambi_dec_" + n, ci, vp)
    for n in ("getDecOrder", "getDecMethod", "getDecEnableMaxrE", "getDecNormType"):
        sig("ambi_dec_" + n, ci, vp, ci)
    sig("ambi_dec_getProgressBar0_1", cf, vp)
    sig("ambi_dec_getProgressBarText", None, vp, C.c_char_p)
    sig("ambi_dec_getDecOrderHandle", None, vp, C.POINTER(fp), C.POINTER(ip), ip)
    sig("ambi_dec_getLoudspeakerAzi_deg", cf, vp, ci)
    sig("ambi_dec_getLoudspeakerElev_deg", cf, vp, ci)
    sig("ambi_dec_getSofaFilePath", C.c_char_p, vp)
    sig("ambi_dec_getTransitionFreq", cf, vp)
    sig("saf_hip_ambi_dec_getDecoderMtx", None, vp, ci, ci, ci, fp)
    sig("saf_hip_ambi_dec_getDecoderNorm", cf, vp, ci, ci, ci)
    sig("saf_hip_ambi_dec_batch_create", vp, C.POINTER(vp), ci, ci)
    sig("saf_hip_ambi_dec_batch_destroy", None, C.POINTER(vp))
    sig("saf_hip_ambi_dec_batch_clear", None, vp)
    sig("saf_hip_ambi_dec_batch_process", None, vp, vp, cll, cll, cll, vp, cll, cll, cll, ci)
    # ambi_enc
    sig("saf_hip_ambi_enc_setFrameSize", None, ci)
    sig("ambi_enc_create", None, C.POINTER(vp))
    sig("ambi_enc_destroy", None, C.POINTER(vp))
    sig("ambi_enc_init", None, vp, ci)
    sig("ambi_enc_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    sig("ambi_enc_refreshParams", None, vp)
    for n in ("setOutputOrder", "setNumSources", "setInputConfigPreset", "setChOrder", "setNormType", "setEnablePostScaling", "setSourceSolo"):
        sig("ambi_enc_" + n, None, vp, ci)
    for n in ("setSourceAzi_deg", "setSourceElev_deg", "setSourceGain"):
        sig("ambi_enc_" + n, None, vp, ci, cf)
    sig("ambi_enc_setUnSolo", None, vp)
    for n in ("getFrameSize", "getMaxNumSources", "getProcessingDelay"):
        sig("ambi_enc_" + n, ci)
    for n in ("getOutputOrder", "getNumSources", "getNSHrequired", "getChOrder", "getNormType", "getEnablePostScaling"):
        sig("ambi_enc_" + n, ci, vp)
    sig("ambi_enc_getSourceAzi_deg", cf, vp, ci)
    sig("ambi_enc_getSourceElev_deg", cf, vp, ci)
    sig("saf_hip_ambi_enc_batch_create", vp, C.POINTER(vp), ci, ci)
    sig("saf_hip_ambi_enc_batch_destroy", None, C.POINTER(vp))
    sig("saf_hip_ambi_enc_batch_process", None, vp, vp, cll, cll, cll, ci, vp, cll, cll, cll, ci, ci)
    # matrix convolver
    sig("saf_matrixConv_create", None, C.POINTER(vp), ci, fp, ci, ci, ci, ci)
    sig("saf_matrixConv_destroy", None, C.POINTER(vp))
    sig("saf_matrixConv_apply", None, vp, fp, fp)
    sig("saf_hip_matrixConv_setMaxBlocksPerCall", None, ci)
    sig("saf_hip_matrixConv_apply_dev", None, vp, vp, cll, cll, vp, cll, cll, ci)
    sig("saf_multiConv_create", None, C.POINTER(vp), ci, fp, ci, ci, ci)
    sig("saf_multiConv_destroy", None, C.POINTER(vp))
    sig("saf_multiConv_apply", None, vp, fp, fp)
    sig("saf_hip_multiConv_apply_dev", None, vp, vp, cll, cll, vp, cll, cll, ci)
    sig("saf_TVConv_create", None, C.POINTER(vp), ci, C.POINTER(fp), ci, ci, ci, ci)
    sig("saf_TVConv_destroy", None, C.POINTER(vp))
    sig("saf_TVConv_apply", None, vp, fp, fp, ci)
    sig("saf_hip_TVConv_apply_dev", None, vp, vp, cll, vp, cll, cll, C.POINTER(ci), ci)
    # HRIR processing + binauraliser
    sig("estimateITDs", None, fp, ci, ci, ci, fp)
    sig("HRIRs2HRTFs_afSTFT", None, fp, ci, ci, ci, ci, ci, vp)
    sig("diffuseFieldEqualiseHRTFs", None, ci, fp, fp, ci, fp, ci, ci, vp)
    sig("getVoronoiWeights", None, fp, ci, ci, fp)
    sig("saf_hip_setDefaultHRIRs", None, fp, fp, ci, ci, ci)
    sig("saf_hip_binauraliser_setFrameSize", None, ci)
    sig("saf_hip_binauraliser_setMaxNumSources", None, ci)
    sig("binauraliser_create", None, C.POINTER(vp))
    sig("binauraliser_destroy", None, C.POINTER(vp))
    sig("binauraliser_init", None, vp, ci)
    sig("binauraliser_initCodec", None, vp)
    sig("binauraliser_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    sig("binauraliser_refreshSettings", None, vp)
    for n in ("setNumSources", "setUseDefaultHRIRsflag", "setEnableHRIRsDiffuseEQ", "setInputConfigPreset", "setEnableRotation", "setFlipYaw",
              "setFlipPitch", "setFlipRoll", "setRPYflag", "setInterpMode", "setSourceSolo"):
        sig("binauraliser_" + n, None, vp, ci)
    for n in ("setSourceAzi_deg", "setSourceElev_deg", "setSourceGain"):
        sig("binauraliser_" + n, None, vp, ci, cf)
    for n in ("setYaw", "setPitch", "setRoll"):
        sig("binauraliser_" + n, None, vp, cf)
    sig("binauraliser_setSofaFilePath", None, vp, C.c_char_p)
    sig("binauraliser_setUnSolo", None, vp)
    for n in ("getFrameSize", "getMaxNumSources", "getNumEars", "getProcessingDelay"):
        sig("binauraliser_" + n, ci)
    for n in ("getCodecStatus", "getNumSources", "getNDirs", "getNTriangles", "getHRIRlength", "getHRIRsamplerate", "getUseDefaultHRIRsflag",
              "getEnableHRIRsDiffuseEQ", "getDAWsamplerate", "getEnableRotation", "getFlipYaw", "getFlipPitch", "getFlipRoll", "getRPYflag", "getInterpMode"):
        sig("binauraliser_" + n, ci, vp)
    for n in ("getProgressBar0_1", "getYaw", "getPitch", "getRoll"):
        sig("binauraliser_" + n, cf, vp)
    for n in ("getSourceAzi_deg", "getSourceElev_deg", "getHRIRAzi_deg", "getHRIRElev_deg"):
        sig("binauraliser_" + n, cf, vp, ci)
    sig("binauraliser_getProgressBarText", None, vp, C.c_char_p)
    sig("binauraliser_getSofaFilePath", C.c_char_p, vp)
    sig("saf_hip_binauraliser_process_dev", None, vp, vp, cll, cll, ci, vp, cll, cll, ci)
    sig("saf_hip_binauraliser_batch_create", vp, C.POINTER(vp), ci, ci)
    sig("saf_hip_binauraliser_batch_destroy", None, C.POINTER(vp))
    sig("saf_hip_binauraliser_batch_process", None, vp, vp, cll, cll, cll, ci, vp, cll, cll, cll, ci)
    sig("saf_hip_binauraliser_getITDs", None, vp, fp)
    sig("saf_hip_binauraliser_getWeights", None, vp, fp)
    sig("saf_hip_binauraliser_getHRTFfb", None, vp, vp)
    sig("saf_hip_binauraliser_getHRTFinterp", None, vp, vp)
    # binauraliser_nf + DVF utilities
    sig("binauraliserNF_create", None, C.POINTER(vp))
    sig("binauraliserNF_destroy", None, C.POINTER(vp))
    sig("binauraliserNF_init", None, vp, ci)
    sig("binauraliserNF_initCodec", None, vp)
    sig("binauraliserNF_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    sig("binauraliserNF_processFD", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    sig("binauraliserNF_setSourceDist_m", None, vp, ci, cf)
    sig("binauraliserNF_setInputConfigPreset", None, vp, ci)
    sig("binauraliserNF_getSourceDist_m", cf, vp, ci)
    for n in ("getFarfieldThresh_m", "getFarfieldHeadroom", "getNearfieldLimit_m"):
        sig("binauraliserNF_" + n, cf, vp)
    sig("saf_hip_binauraliserNF_process_dev", None, vp, vp, cll, cll, ci, vp, cll, cll, ci)
    sig("saf_hip_binauraliserNF_getHRTFnf", None, vp, vp)
    sig("calcDVFCoeffs", None, cf, cf, cf, fp, fp)
    sig("interpDVFShelfParams", None, cf, cf, fp, fp, fp)
    sig("dvfShelfCoeffs", None, cf, cf, cf, cf, fp, fp, fp)
    sig("calcDVFShelfParams", None, ci, cf, fp, fp, fp)
    sig("doaToIpsiInteraural", None, cf, cf, fp, fp)
    sig("evalIIRTransferFunctionf", None, fp, fp, ci, fp, ci, cf, ci, fp, fp)
    # panner
    sig("getPvalues", None, cf, fp, ci, fp)
    sig("saf_hip_panner_setFrameSize", None, ci)
    sig("panner_create", None, C.POINTER(vp))
    sig("panner_destroy", None, C.POINTER(vp))
    sig("panner_init", None, vp, ci)
    sig("panner_initCodec", None, vp)
    sig("panner_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    sig("panner_refreshSettings", None, vp)
    for n in ("setNumSources", "setNumLoudspeakers", "setOutputConfigPreset", "setInputConfigPreset", "setFlipYaw", "setFlipPitch", "setFlipRoll"):
        sig("panner_" + n, None, vp, ci)
    for n in ("setSourceAzi_deg", "setSourceElev_deg", "setLoudspeakerAzi_deg", "setLoudspeakerElev_deg"):
        sig("panner_" + n, None, vp, ci, cf)
    for n in ("setDTT", "setSpread", "setYaw", "setPitch", "setRoll"):
        sig("panner_" + n, None, vp, cf)
    for n in ("getFrameSize", "getMaxNumSources", "getMaxNumLoudspeakers", "getProcessingDelay"):
        sig("panner_" + n, ci)
    for n in ("getCodecStatus", "getNumSources", "getNumLoudspeakers", "getDAWsamplerate", "getFlipYaw", "getFlipPitch", "getFlipRoll"):
        sig("panner_" + n, ci, vp)
    for n in ("getProgressBar0_1", "getDTT", "getSpread", "getYaw", "getPitch", "getRoll"):
        sig("panner_" + n, cf, vp)
    for n in ("getSourceAzi_deg", "getSourceElev_deg", "getLoudspeakerAzi_deg", "getLoudspeakerElev_deg"):
        sig("panner_" + n, cf, vp, ci)
    sig("panner_getProgressBarText", None, vp, C.c_char_p)
    sig("saf_hip_panner_process_dev", None, vp, vp, cll, cll, ci, vp, cll, cll, ci)
    sig("saf_hip_panner_getGains", None, vp, fp)
    # matrixconv / multiconv example operators
    for pre, nin in (("matrixconv", "setNumInputChannels"), ("multiconv", "setNumChannels")):
        sig(pre + "_create", None, C.POINTER(vp)); sig(pre + "_destroy", None, C.POINTER(vp))
        sig(pre + "_init", None, vp, ci, ci)
        sig(pre + "_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
        sig(pre + "_refreshParams", None, vp); sig(pre + "_checkReInit", None, vp)
        sig(pre + "_setFilters", None, vp, C.POINTER(fp), ci, ci, ci)
        sig(pre + "_setEnablePart", None, vp, ci); sig(pre + "_" + nin, None, vp, ci)
        for g in ("getEnablePart", "getHostBlockSize", "getNfilters", "getFilterLength", "getFilterFs", "getHostFs", "getProcessingDelay", nin.replace("set", "get")):
            sig(pre + "_" + g, ci, vp)
    sig("matrixconv_getNumOutputChannels", ci, vp)
    # binaural Ambisonic decoding
    sig("getSHrotMtxReal", None, fp, fp, ci)
    sig("yawPitchRoll2Rzyx", None, cf, cf, cf, ci, fp)
    sig("beamWeightsMaxEV", None, ci, fp)
    sig("truncationEQ", None, fp, ci, ci, C.POINTER(C.c_double), ci, cf, fp)
    sig("getBinauralAmbiDecoderMtx", None, vp, fp, ci, ci, ci, ci, fp, fp, fp, ci, ci, vp)
    sig("getBinauralAmbiDecoderFilters", None, vp, fp, ci, ci, cf, ci, ci, fp, fp, ci, ci, fp)
    sig("applyDiffCovMatching", None, vp, fp, ci, ci, ci, fp, vp)
    sig("saf_hip_ambi_bin_setFrameSize", None, ci)
    sig("ambi_bin_create", None, C.POINTER(vp)); sig("ambi_bin_destroy", None, C.POINTER(vp))
    sig("ambi_bin_init", None, vp, ci); sig("ambi_bin_initCodec", None, vp); sig("ambi_bin_refreshParams", None, vp)
    sig("ambi_bin_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    for n in ("setUseDefaultHRIRsflag", "setInputOrderPreset", "setDecodingMethod", "setChOrder", "setNormType", "setEnableMaxRE", "setEnableDiffuseMatching",
              "setEnableTruncationEQ", "setHRIRsPreProc", "setEnableRotation", "setFlipYaw", "setFlipPitch", "setFlipRoll", "setRPYflag"):
        sig("ambi_bin_" + n, None, vp, ci)
    for n in ("setYaw", "setPitch", "setRoll"):
        sig("ambi_bin_" + n, None, vp, cf)
    sig("ambi_bin_setSofaFilePath", None, vp, C.c_char_p)
    for n in ("getFrameSize", "getNumEars", "getProcessingDelay"):
        sig("ambi_bin_" + n, ci)
    for n in ("getCodecStatus", "getUseDefaultHRIRsflag", "getInputOrderPreset", "getDecodingMethod", "getChOrder", "getNormType", "getNSHrequired", "getEnableMaxRE",
              "getEnableDiffuseMatching", "getEnableTruncationEQ", "getHRIRsPreProc", "getEnableRotation", "getFlipYaw", "getFlipPitch", "getFlipRoll", "getRPYflag",
              "getNDirs", "getHRIRlength", "getHRIRsamplerate", "getDAWsamplerate"):
        sig("ambi_bin_" + n, ci, vp)
    for n in ("getProgressBar0_1", "getYaw", "getPitch", "getRoll"):
        sig("ambi_bin_" + n, cf, vp)
    sig("ambi_bin_getProgressBarText", None, vp, C.c_char_p)
    sig("ambi_bin_getSofaFilePath", C.c_char_p, vp)
    sig("saf_hip_ambi_bin_process_dev", None, vp, vp, cll, cll, ci, vp, cll, cll, ci)
    sig("saf_hip_ambi_bin_getDecoderMtx", None, vp, vp)
    # real FFT object
    sig("saf_rfft_create", None, C.POINTER(vp), ci)
    sig("saf_rfft_destroy", None, C.POINTER(vp))
    sig("saf_rfft_forward", None, vp, fp, vp)
    sig("saf_rfft_backward", None, vp, vp, fp)
    for m in ("matrixconv", "multiconv", "tvconv"):
        sig(m + "_getFrameSize", ci)
    sig("quaternion2rotationMatrix", None, fp, fp); sig("rotationMatrix2quaternion", None, fp, fp)
    sig("euler2Quaternion", None, cf, cf, cf, ci, ci, fp); sig("quaternion2euler", None, fp, ci, ci, fp, fp, fp)
    sig("beamWeightsCardioid2Spherical", None, ci, fp); sig("beamWeightsHypercardioid2Spherical", None, ci, fp)
    sig("rotateAxisCoeffsReal", None, ci, fp, cf, cf, fp)
    sig("saf_hip_beamformer_setFrameSize", None, ci)
    sig("beamformer_create", None, C.POINTER(vp)); sig("beamformer_destroy", None, C.POINTER(vp)); sig("beamformer_init", None, vp, ci)
    sig("beamformer_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    sig("saf_hip_beamformer_process_dev", None, vp, vp, cll, cll, ci, vp, cll, cll, ci, ci)
    sig("beamformer_refreshSettings", None, vp)
    for g in ("BeamOrder", "NumBeams", "ChOrder", "NormType", "BeamType"):
        sig("beamformer_set" + g, None, vp, ci); sig("beamformer_get" + g, ci, vp)
    for g in ("BeamAzi_deg", "BeamElev_deg"):
        sig("beamformer_set" + g, None, vp, ci, cf); sig("beamformer_get" + g, cf, vp, ci)
    sig("beamformer_getFrameSize", ci); sig("beamformer_getMaxNumBeams", ci); sig("beamformer_getProcessingDelay", ci)
    sig("beamformer_getNSHrequired", ci, vp)
    sig("saf_hip_ambi_drc_setFrameSize", None, ci)
    sig("ambi_drc_create", None, C.POINTER(vp)); sig("ambi_drc_destroy", None, C.POINTER(vp)); sig("ambi_drc_init", None, vp, ci)
    sig("ambi_drc_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci)
    sig("saf_hip_ambi_drc_process_dev", None, vp, vp, cll, cll, ci, vp, cll, cll, ci)
    sig("ambi_drc_refreshSettings", None, vp)
    for g in ("Threshold", "Ratio", "Knee", "InGain", "OutGain", "Attack", "Release"):
        sig("ambi_drc_set" + g, None, vp, cf); sig("ambi_drc_get" + g, cf, vp)
    for g in ("ChOrder", "NormType", "InputPreset"):
        sig("ambi_drc_set" + g, None, vp, ci); sig("ambi_drc_get" + g, ci, vp)
    sig("ambi_drc_getFrameSize", ci); sig("ambi_drc_getProcessingDelay", ci)
    for g in ("GainTFwIdx", "GainTFrIdx", "NSHrequired", "Samplerate"):
        sig("ambi_drc_get" + g, ci, vp)
    sig("ambi_drc_getGainTF", C.POINTER(fp), vp); sig("ambi_drc_getFreqVector", fp, vp, ip)
    sig("saf_hip_rotator_setFrameSize", None, ci)
    sig("rotator_create", None, C.POINTER(vp)); sig("rotator_destroy", None, C.POINTER(vp)); sig("rotator_init", None, vp, ci)
    sig("rotator_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    sig("saf_hip_rotator_process_dev", None, vp, vp, cll, cll, ci, vp, cll, cll, ci, ci)
    sig("rotator_getFrameSize", ci); sig("rotator_getProcessingDelay", ci)
    for g in ("Yaw", "Pitch", "Roll", "QuaternionW", "QuaternionX", "QuaternionY", "QuaternionZ"):
        sig("rotator_set" + g, None, vp, cf); sig("rotator_get" + g, cf, vp)
    for g in ("FlipYaw", "FlipPitch", "FlipRoll", "FlipQuaternion", "ChOrder", "NormType", "Order", "RPYflag"):
        sig("rotator_set" + g, None, vp, ci); sig("rotator_get" + g, ci, vp)
    sig("rotator_getNSHrequired", ci, vp)
    sig("tvconv_create", None, C.POINTER(vp)); sig("tvconv_destroy", None, C.POINTER(vp))
    sig("tvconv_init", None, vp, ci, ci)
    sig("tvconv_process", None, vp, C.POINTER(fp), C.POINTER(fp), ci, ci, ci)
    sig("tvconv_refreshParams", None, vp); sig("tvconv_checkReInit", None, vp); sig("tvconv_setFiltersAndPositions", None, vp)
    sig("tvconv_setSofaFilePath", None, vp, C.c_char_p)
    sig("tvconv_setTargetPosition", None, vp, cf, ci)
    for g in ("getNumInputChannels", "getNumOutputChannels", "getHostBlockSize", "getNumIRs", "getNumListenerPositions", "getListenerPositionIdx", "getIRLength", "getIRFs",
              "getHostFs", "getProcessingDelay", "getCodecStatus"):
        sig("tvconv_" + g, ci, vp)
    for g in ("getTargetPosition", "getSourcePosition", "getMinDimension", "getMaxDimension"):
        sig("tvconv_" + g, cf, vp, ci)
    sig("tvconv_getListenerPosition", cf, vp, ci, ci)
    sig("tvconv_getSofaFilePath", C.c_char_p, vp)
    sig("saf_hip_tvconv_setIRsAndPositions", None, vp, C.POINTER(fp), fp, fp, ci, ci, ci, ci)
    # activity-map generators
    sig("generatePWDmap", None, ci, vp, vp, ci, fp)
    sig("generateMVDRmap", None, ci, vp, vp, ci, cf, fp, vp)
    sig("generateCroPaCLCMVmap", None, ci, vp, vp, ci, cf, cf, fp)
    sig("generateMUSICmap", None, ci, vp, vp, ci, ci, ci, fp)
    sig("generateMinNormMap", None, ci, vp, vp, ci, ci, ci, fp)
    for n in ("sphPWD", "sphMUSIC"):
        sig(n + "_create", None, C.POINTER(vp), ci, fp, ci)
        sig(n + "_destroy", None, C.POINTER(vp))
        sig(n + "_compute", None, vp, vp, ci, fp, C.POINTER(ci))
    # powermap
    sig("saf_hip_powermap_setFrameSize", None, ci)
    sig("powermap_create", None, C.POINTER(vp))
    sig("powermap_destroy", None, C.POINTER(vp))
    sig("powermap_init", None, vp, cf)
    sig("powermap_initCodec", None, vp)
    sig("powermap_analysis", None, vp, C.POINTER(fp), ci, ci, ci)
    sig("powermap_refreshSettings", None, vp)
    sig("powermap_requestPmapUpdate", None, vp)
    for n in ("setPowermapMode", "setMasterOrder", "setAnaOrderAllBands", "setChOrder", "setNormType", "setSourcePreset", "setNumSources", "setDispFOV", "setAspectRatio"):
        sig("powermap_" + n, None, vp, ci)
    sig("powermap_setAnaOrder", None, vp, ci, ci)
    sig("powermap_setPowermapEQ", None, vp, cf, ci)
    for n in ("setPowermapEQAllBands", "setCovAvgCoeff", "setPowermapAvgCoeff"):
        sig("powermap_" + n, None, vp, cf)
    for n in ("getFrameSize", "getNumberOfBands", "getProcessingDelay"):
        sig("powermap_" + n, ci)
    for n in ("getCodecStatus", "getMasterOrder", "getPowermapMode", "getSamplingRate", "getNSHrequired", "getAnaOrderAllBands", "getChOrder", "getNormType",
              "getNumSources", "getDispFOV", "getAspectRatio"):
        sig("powermap_" + n, ci, vp)
    for n in ("getProgressBar0_1", "getCovAvgCoeff", "getPowermapEQAllBands", "getPowermapAvgCoeff"):
        sig("powermap_" + n, cf, vp)
    sig("powermap_getPowermapEQ", cf, vp, ci)
    sig("powermap_getAnaOrder", ci, vp, ci)
    sig("powermap_getProgressBarText", None, vp, C.c_char_p)
    sig("powermap_getPowermapEQHandle", None, vp, C.POINTER(fp), C.POINTER(fp), ip)
    sig("powermap_getAnaOrderHandle", None, vp, C.POINTER(fp), C.POINTER(ip), ip)
    sig("powermap_getPmap", ci, vp, C.POINTER(fp), C.POINTER(fp), ip, ip, ip, ip)
    sig("saf_hip_powermap_analysis_dev", None, vp, vp, cll, cll, ci, ci)
    sig("saf_hip_powermap_getCx", None, vp, vp)
    sig("saf_hip_powermap_batch_create", vp, C.POINTER(vp), ci, ci)
    sig("saf_hip_powermap_batch_destroy", None, C.POINTER(vp))
    sig("saf_hip_powermap_batch_analysis", None, vp, vp, cll, cll, cll, ci, ci)
    sig("saf_hip_powermap_batch_getCx", None, vp, ci, vp)
    sig("saf_hip_powermap_getRawPmap", ci, vp, fp)
    _lib = L
    return L
