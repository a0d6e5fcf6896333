"""Python mirror of the SAF operator / framework interface for the hot path,
implemented by calling libsaf_hip.so through its C-ABI (include/saf_hip.h).

Names, argument meaning and error behaviour follow the reference headers
(examples/include/ambi_dec.h, framework/resources/afSTFT/afSTFTlib.h,
framework/modules/saf_hoa/saf_hoa.h, saf_sh.h, saf_vbap.h).  This layer only
marshals numpy arrays / device pointers; all arithmetic happens in the library.
"""
import ctypes as C

import numpy as np

from ._lib import load

vp = C.c_void_p
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)

# enums (include/saf_hip.h)
AFSTFT_BANDS_CH_TIME, AFSTFT_TIME_CH_BANDS = 0, 1
CH_ACN, CH_FUMA = 1, 2
NORM_N3D, NORM_SN3D, NORM_FUMA = 1, 2, 3
DECODING_METHOD_SAD, DECODING_METHOD_MMD, DECODING_METHOD_EPAD, DECODING_METHOD_ALLRAD = 1, 2, 3, 4
LOUDSPEAKER_DECODER_SAD, LOUDSPEAKER_DECODER_MMD, LOUDSPEAKER_DECODER_EPAD, LOUDSPEAKER_DECODER_ALLRAD = 1, 2, 3, 4
AMPLITUDE_PRESERVING, ENERGY_PRESERVING = 1, 2
CODEC_STATUS_INITIALISED, CODEC_STATUS_NOT_INITIALISED, CODEC_STATUS_INITIALISING = 0, 1, 2
LOUDSPEAKER_ARRAY_PRESET_22PX = 11
LOUDSPEAKER_ARRAY_PRESET_T_DESIGN_24 = 21
LOUDSPEAKER_ARRAY_PRESET_T_DESIGN_60 = 24
LOUDSPEAKER_ARRAY_PRESET_SPH_COV_49 = 28
LOUDSPEAKER_ARRAY_PRESET_SPH_COV_64 = 29


def _f(a):
    return a.ctypes.data_as(fp)


def _rows(a):
    """channel-pointer table (float**) of a 2-D float32 array with contiguous rows"""
    base, stride = a.ctypes.data, a.strides[0]
    return C.cast((C.c_void_p * a.shape[0])(*[base + i * stride for i in range(a.shape[0])]), C.POINTER(fp))


def set_stream(ptr):
    load().saf_hip_set_stream(vp(ptr))


def synchronize():
    load().saf_hip_synchronize()


# ---------------------------------------------------------------- afSTFT
class AfSTFT:
    def __init__(self, nCHin, nCHout, hopsize=128, lowDelay=0, hybrid=1, fmt=AFSTFT_BANDS_CH_TIME):
        self.L = load()
        self.h = vp()
        self.nCHin, self.nCHout, self.hop, self.fmt = nCHin, nCHout, hopsize, fmt
        self.L.afSTFT_create(C.byref(self.h), nCHin, nCHout, hopsize, lowDelay, hybrid, fmt)
        self.nBands = self.L.afSTFT_getNBands(self.h)
        self.delay = self.L.afSTFT_getProcDelay(self.h)

    def forward(self, x):
        """x [nCHin][framesize] -> [nBands][nCHin][nHops] (afSTFT_forward_flat)."""
        x = np.ascontiguousarray(x, np.float32)
        nH = x.shape[1] // self.hop
        shape = (self.nBands, self.nCHin, nH) if self.fmt == AFSTFT_BANDS_CH_TIME else (nH, self.nCHin, self.nBands)
        out = np.zeros(shape, np.complex64)
        self.L.afSTFT_forward_flat(self.h, _f(x), x.shape[1], out.ctypes.data_as(vp))
        return out

    def forward_knownDimensions(self, x, nCH_alloc, nHops_alloc):
        """Mimics a caller that owns a malloc3d [nBands][nCH_alloc][nHops_alloc] buffer."""
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((self.nBands, nCH_alloc, nHops_alloc), np.complex64)
        # build the pointer tables of malloc3d so &dataFD[0][0][0] is the flat payload
        rows2 = (vp * (self.nBands * nCH_alloc))()
        base = out.ctypes.data
        for i in range(self.nBands * nCH_alloc):
            rows2[i] = base + i * nHops_alloc * 8
        rows1 = (vp * self.nBands)()
        for b in range(self.nBands):
            rows1[b] = C.addressof(rows2) + b * nCH_alloc * C.sizeof(vp)
        self.L.afSTFT_forward_knownDimensions(self.h, _rows(x), x.shape[1], nCH_alloc, nHops_alloc, C.cast(rows1, vp))
        return out

    def backward(self, X):
        X = np.ascontiguousarray(X, np.complex64)
        nH = X.shape[2] if self.fmt == AFSTFT_BANDS_CH_TIME else X.shape[0]
        out = np.zeros((self.nCHout, nH * self.hop), np.float32)
        self.L.afSTFT_backward_flat(self.h, X.ctypes.data_as(vp), nH * self.hop, _f(out))
        return out

    def channelChange(self, nin, nout):
        self.L.afSTFT_channelChange(self.h, nin, nout)
        self.nCHin, self.nCHout = nin, nout

    def clearBuffers(self):
        self.L.afSTFT_clearBuffers(self.h)

    def centreFreqs(self, fs):
        f = np.zeros(self.nBands, np.float32)
        self.L.afSTFT_getCentreFreqs(self.h, C.c_float(fs), self.nBands, _f(f))
        return f

    def forward_dev(self, d_td_ptr, td_ch_stride, nHops, d_fd_ptr, fd_band_stride, fd_ch_stride):
        self.L.saf_hip_afSTFT_forward_dev(self.h, vp(d_td_ptr), td_ch_stride, nHops, vp(d_fd_ptr), fd_band_stride, fd_ch_stride)

    def backward_dev(self, d_fd_ptr, fd_band_stride, fd_ch_stride, nHops, d_td_ptr, td_ch_stride):
        self.L.saf_hip_afSTFT_backward_dev(self.h, vp(d_fd_ptr), fd_band_stride, fd_ch_stride, nHops, vp(d_td_ptr), td_ch_stride)

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.afSTFT_destroy(C.byref(self.h))


def afAnalyse(x_td, hopSize=128, LDmode=0, hybridmode=1):
    """afAnalyse (afSTFTlib.h:85): x_td [nSamples][nCH] -> [nBands][ceil(nSamples / hop)][nCH], fresh filterbank"""
    x = np.ascontiguousarray(x_td, np.float32); nS, nCH = x.shape
    nB = hopSize + (5 if hybridmode else 1); nT = int(np.float32(nS) / np.float32(hopSize) + np.float32(0.9999))
    out = np.zeros((nB, nT, nCH), np.complex64)
    load().afAnalyse(_f(x), nS, nCH, hopSize, LDmode, hybridmode, out.ctypes.data_as(vp))
    return out


def convertHOAChannelConvention(sig, order, inConv, outConv):
    """in place on a copy: sig [(order+1)^2][signalLength]; HOA_CH_ORDER: 0 = ACN, 1 = FuMa (saf_hoa.h:183-192, 237)"""
    x = np.ascontiguousarray(sig, np.float32).copy()
    load().convertHOAChannelConvention(_f(x), order, x.shape[1], inConv, outConv); return x


def convertHOANormConvention(sig, order, inConv, outConv):
    """HOA_NORM: 0 = N3D, 1 = SN3D, 2 = FuMa (saf_hoa.h:203-213, 262)"""
    x = np.ascontiguousarray(sig, np.float32).copy()
    load().convertHOANormConvention(_f(x), order, x.shape[1], inConv, outConv); return x


def afSTFT_getCentreFreqs_nullHandle(fs):
    f = np.zeros(133, np.float32)
    load().afSTFT_getCentreFreqs(None, C.c_float(fs), 133, _f(f))
    return f


def afSTFT_FIRtoFilterbankCoeffs(hIR, hop=128, LD=0, hybrid=1):
    hIR = np.ascontiguousarray(hIR, np.float32)
    nd, nch, Lh = hIR.shape
    out = np.zeros((hop + (5 if hybrid else 1), nch, nd), np.complex64)
    load().afSTFT_FIRtoFilterbankCoeffs(_f(hIR), nd, nch, Lh, hop, LD, hybrid, out.ctypes.data_as(vp))
    return out


# ---------------------------------------------------------------- SH / HOA / VBAP
def _sh(name, order, dirs):
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 2)
    Y = np.zeros(((order + 1) ** 2, d.shape[0]), np.float32)
    getattr(load(), name)(order, _f(d), d.shape[0], _f(Y))
    return Y


def getSHreal(order, dirs_rad): return _sh("getSHreal", order, dirs_rad)
def getSHreal_recur(order, dirs_rad): return _sh("getSHreal_recur", order, dirs_rad)
def getRSH(order, dirs_deg): return _sh("getRSH", order, dirs_deg)
def getRSH_recur(order, dirs_deg): return _sh("getRSH_recur", order, dirs_deg)


def getMaxREweights(order, diag=False):
    n = (order + 1) ** 2
    a = np.zeros((n, n) if diag else n, np.float32)
    load().getMaxREweights(order, int(diag), _f(a))
    return a


def getLoudspeakerDecoderMtx(ls_dirs_deg, method, order, maxrE=0):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    M = np.zeros((d.shape[0], (order + 1) ** 2), np.float32)
    load().getLoudspeakerDecoderMtx(_f(d), d.shape[0], method, order, maxrE, _f(M))
    return M


def _take(ptr, shape, ctype=np.float32):
    """Copy a malloc'd out-param into numpy and free it (caller-frees convention of the reference)."""
    n = int(np.prod(shape))
    a = np.ctypeslib.as_array(ptr, shape=(n,)).copy().reshape(shape) if n else np.zeros(shape, ctype)
    C.CDLL(None).free(C.cast(ptr, vp))
    return a


def findLsTriplets(ls_dirs_deg, omitLarge=0):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    v, f = fp(), ip()
    nv, nf = C.c_int(), C.c_int()
    load().findLsTriplets(_f(d), d.shape[0], omitLarge, C.byref(v), C.byref(nv), C.byref(f), C.byref(nf))
    V = _take(v, (nv.value, 3))
    Fc = _take(f, (nf.value, 3), np.int32) if nf.value else np.zeros((0, 3), np.int32)
    return V, Fc


def generateVBAPgainTable3D_srcs(src_dirs_deg, ls_dirs_deg, omitLarge=0, dummies=0, spread=0.0):
    s = np.ascontiguousarray(src_dirs_deg, np.float32).reshape(-1, 2)
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    g = fp()
    n, nt = C.c_int(), C.c_int()
    load().generateVBAPgainTable3D_srcs(_f(s), s.shape[0], _f(d), d.shape[0], omitLarge, dummies, C.c_float(spread), C.byref(g), C.byref(n), C.byref(nt))
    return _take(g, (n.value, d.shape[0])), nt.value


def generateVBAPgainTable3D(ls_dirs_deg, az_res, el_res, omitLarge=0, dummies=0, spread=0.0):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    g = fp()
    n, nt = C.c_int(), C.c_int()
    load().generateVBAPgainTable3D(_f(d), d.shape[0], az_res, el_res, omitLarge, dummies, C.c_float(spread), C.byref(g), C.byref(n), C.byref(nt))
    return _take(g, (n.value, d.shape[0])), nt.value


def generateVBAPgainTable2D(ls_dirs_deg, az_res):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    g = fp()
    n, npairs = C.c_int(), C.c_int()
    load().generateVBAPgainTable2D(_f(d), d.shape[0], az_res, C.byref(g), C.byref(n), C.byref(npairs))
    return _take(g, (n.value, d.shape[0])), npairs.value


def generateVBAPgainTable2D_srcs(src_azi_deg, ls_dirs_deg):
    a = np.ascontiguousarray(src_azi_deg, np.float32).reshape(-1)
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    g = fp()
    n, npairs = C.c_int(), C.c_int()
    load().generateVBAPgainTable2D_srcs(_f(a), a.shape[0], _f(d), d.shape[0], C.byref(g), C.byref(n), C.byref(npairs))
    return _take(g, (n.value, d.shape[0])), npairs.value


def findLsPairs(ls_dirs_deg):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    pairs = ip(); n = C.c_int()
    load().findLsPairs(_f(d), d.shape[0], C.byref(pairs), C.byref(n))
    out = np.ctypeslib.as_array(pairs, shape=(n.value * 2,)).copy().reshape(n.value, 2)
    C.CDLL(None).free(C.cast(pairs, vp))
    return out


def getSpreadSrcDirs3D(azi_rad, elev_rad, spread_deg, num_src=8, num_rings=1):
    U = np.zeros((num_rings * num_src + 1, 3), np.float32)
    load().getSpreadSrcDirs3D(C.c_float(azi_rad), C.c_float(elev_rad), C.c_float(spread_deg), num_src, num_rings, _f(U))
    return U


def compressVBAPgainTable3D(gt):
    gt = np.ascontiguousarray(gt, np.float32)
    comp = np.zeros((gt.shape[0], 3), np.float32)
    idx = np.zeros((gt.shape[0], 3), np.int32)
    load().compressVBAPgainTable3D(_f(gt), gt.shape[0], gt.shape[1], _f(comp), idx.ctypes.data_as(ip))
    return comp, idx


# ---------------------------------------------------------------- ambi_dec
class AmbiDec:
    """examples/include/ambi_dec.h.  `frameSize` plays the role of -DAMBI_DEC_FRAME_SIZE."""

    def __init__(self, frameSize=128):
        self.L = load()
        self.L.saf_hip_ambi_dec_setFrameSize(frameSize)
        self.h = vp()
        self.F = frameSize
        self.L.ambi_dec_create(C.byref(self.h))

    def __getattr__(self, name):
        fn = getattr(load(), "ambi_dec_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def setLoudspeakersDeg(self, dirs):
        d = np.asarray(dirs, np.float32).reshape(-1, 2)
        self.L.ambi_dec_setNumLoudspeakers(self.h, d.shape[0])
        for i in range(d.shape[0]):
            self.L.ambi_dec_setLoudspeakerAzi_deg(self.h, i, C.c_float(float(d[i, 0])))
            self.L.ambi_dec_setLoudspeakerElev_deg(self.h, i, C.c_float(float(d[i, 1])))

    def process(self, x, nOut, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((nOut, max(ns, self.F)), np.nan, np.float32)
        self.L.ambi_dec_process(self.h, _rows(x), _rows(y), x.shape[0], nOut, ns)
        return y[:, :self.F]

    def setHRIRs(self, hrirs, dirs_deg, fs):
        """install the process-wide default HRIR set (what the reference links as saf_default_hrirs.c) and flag a re-init"""
        setDefaultHRIRs(hrirs, dirs_deg, fs)
        self.L.ambi_dec_refreshSettings(self.h)

    def decMtx(self, dec, order, maxrE, nLS):
        M = np.zeros((nLS, (order + 1) ** 2), np.float32)
        self.L.saf_hip_ambi_dec_getDecoderMtx(self.h, dec, order, maxrE, _f(M))
        return M

    def Mnorm(self, dec, order, which):
        return self.L.saf_hip_ambi_dec_getDecoderNorm(self.h, dec, order, which)

    def lastPath(self):
        """0: transform path, 1: equaliser path, -1: no block processed yet (saf_hip_ambi_dec_setTimeDomainPath)"""
        return self.L.saf_hip_ambi_dec_lastPath(self.h)

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.ambi_dec_destroy(C.byref(self.h))


class AmbiDecBatch:
    """saf_hip_ambi_dec_batch_*: nInst initialised handles, device-resident blocks."""

    def __init__(self, decoders, maxFramesPerCall):
        self.L = load()
        self.decoders = list(decoders)          # keep the handles alive
        arr = (vp * len(self.decoders))(*[d.h for d in self.decoders])
        self.nInst = len(self.decoders)
        self.hb = vp(self.L.saf_hip_ambi_dec_batch_create(arr, self.nInst, maxFramesPerCall))

    def process_ptr(self, d_in, in_strides, d_out, out_strides, nFrames):
        """strides = (inst, frame, ch) in floats."""
        self.L.saf_hip_ambi_dec_batch_process(self.hb, vp(d_in), *in_strides, vp(d_out), *out_strides, nFrames)

    def clear(self):
        self.L.saf_hip_ambi_dec_batch_clear(self.hb)

    def lastPath(self):
        return self.L.saf_hip_ambi_dec_batch_lastPath(self.hb)

    def lastOverlap(self):
        """1: the last call ran the decode kernel beside the equaliser kernel (saf_hip_ambi_dec_setOverlap)"""
        return self.L.saf_hip_ambi_dec_batch_lastOverlap(self.hb)

    def decodeGiveUps(self):
        return self.L.saf_hip_ambi_dec_batch_decodeGiveUps(self.hb)

    def __del__(self):
        if getattr(self, "hb", None) and C is not None:
            self.L.saf_hip_ambi_dec_batch_destroy(C.byref(self.hb))


# ---------------------------------------------------------------- ambi_enc
class AmbiEnc:
    """examples/include/ambi_enc.h.  `frameSize` plays the role of -DAMBI_ENC_FRAME_SIZE."""

    def __init__(self, frameSize=64):
        self.L = load()
        self.L.saf_hip_ambi_enc_setFrameSize(frameSize)
        self.h = vp()
        self.F = frameSize
        self.L.ambi_enc_create(C.byref(self.h))

    def __getattr__(self, name):
        fn = getattr(load(), "ambi_enc_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nOut, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((nOut, max(ns, self.F)), np.nan, np.float32)
        self.L.ambi_enc_process(self.h, _rows(x), _rows(y), x.shape[0], nOut, ns)
        return y[:, :self.F]

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.ambi_enc_destroy(C.byref(self.h))


class AmbiEncBatch:
    """saf_hip_ambi_enc_batch_*: nInst handles, device-resident blocks."""

    def __init__(self, encoders, maxFramesPerCall):
        self.L = load()
        self.encoders = list(encoders)
        arr = (vp * len(self.encoders))(*[e.h for e in self.encoders])
        self.nInst = len(self.encoders)
        self.hb = vp(self.L.saf_hip_ambi_enc_batch_create(arr, self.nInst, maxFramesPerCall))

    def process_ptr(self, d_in, in_strides, nIn, d_out, out_strides, nOut, nFrames):
        """strides = (inst, frame, ch) in floats."""
        self.L.saf_hip_ambi_enc_batch_process(self.hb, vp(d_in), *in_strides, nIn, vp(d_out), *out_strides, nOut, nFrames)

    def __del__(self):
        if getattr(self, "hb", None) and C is not None:
            self.L.saf_hip_ambi_enc_batch_destroy(C.byref(self.hb))


# ---------------------------------------------------------------- matrix convolver
class MatrixConv:
    """saf_matrixConv_* (saf_utility_matrixConv.h:55-86); H [nOut][nIn][len]."""

    def __init__(self, hop, H, part=1, maxBlocks=1):
        H = np.ascontiguousarray(H, np.float32)
        self.nOut, self.nIn, self.len = H.shape
        self.hop = hop
        self.L = load()
        self.h = vp()
        self.L.saf_hip_matrixConv_setMaxBlocksPerCall(maxBlocks)
        self.L.saf_matrixConv_create(C.byref(self.h), hop, _f(H), self.len, self.nIn, self.nOut, part)
        self.L.saf_hip_matrixConv_setMaxBlocksPerCall(1)

    def apply(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((self.nOut, self.hop), np.float32)
        self.L.saf_matrixConv_apply(self.h, _f(x), _f(y))
        return y

    def apply_dev(self, d_in, in_strides, d_out, out_strides, nBlocks):
        """strides = (ch, block) in floats."""
        self.L.saf_hip_matrixConv_apply_dev(self.h, vp(d_in), *in_strides, vp(d_out), *out_strides, nBlocks)

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.saf_matrixConv_destroy(C.byref(self.h))


class MultiConv:
    """saf_multiConv_* (saf_utility_matrixConv.h:109-137); H [nCH][len]: channel c is filtered by H[c]."""

    def __init__(self, hop, H, part=1, maxBlocks=1):
        H = np.ascontiguousarray(H, np.float32)
        self.nCH, self.len = H.shape
        self.hop = hop
        self.L = load()
        self.h = vp()
        self.L.saf_hip_matrixConv_setMaxBlocksPerCall(maxBlocks)
        self.L.saf_multiConv_create(C.byref(self.h), hop, _f(H), self.len, self.nCH, part)
        self.L.saf_hip_matrixConv_setMaxBlocksPerCall(1)

    def apply(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((self.nCH, self.hop), np.float32)
        self.L.saf_multiConv_apply(self.h, _f(x), _f(y))
        return y

    def apply_dev(self, d_in, in_strides, d_out, out_strides, nBlocks):
        """strides = (ch, block) in floats."""
        self.L.saf_hip_multiConv_apply_dev(self.h, vp(d_in), *in_strides, vp(d_out), *out_strides, nBlocks)

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.saf_multiConv_destroy(C.byref(self.h))


class TVConv:
    """saf_TVConv_* (saf_utility_matrixConv.h:157-200); H [nIRs][nCHout][len], one input channel."""

    def __init__(self, hop, H, initIdx=0, maxBlocks=1):
        H = np.ascontiguousarray(H, np.float32)
        self.nIRs, self.nOut, self.len = H.shape
        self.hop = hop
        self.L = load()
        self.h = vp()
        rows = (C.POINTER(C.c_float) * self.nIRs)(*[_f(H[i]) for i in range(self.nIRs)])
        self.L.saf_hip_matrixConv_setMaxBlocksPerCall(maxBlocks)
        self.L.saf_TVConv_create(C.byref(self.h), hop, rows, self.len, self.nIRs, self.nOut, initIdx)
        self.L.saf_hip_matrixConv_setMaxBlocksPerCall(1)

    def apply(self, x, irIdx):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((self.nOut, self.hop), np.float32)
        self.L.saf_TVConv_apply(self.h, _f(x), _f(y), int(irIdx))
        return y

    def apply_dev(self, d_in, in_block_stride, d_out, out_strides, irIdx, nBlocks):
        """out_strides = (ch, block) in floats; irIdx: nBlocks host ints."""
        idx = (C.c_int * nBlocks)(*[int(i) for i in irIdx])
        self.L.saf_hip_TVConv_apply_dev(self.h, vp(d_in), in_block_stride, vp(d_out), *out_strides, idx, nBlocks)

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.saf_TVConv_destroy(C.byref(self.h))


class ConvExample:
    """matrixconv (matrix=1, examples/include/matrixconv.h) / multiconv (matrix=0, multiconv.h) example operators"""

    def __init__(self, matrix):
        self.L = load()
        self.pre = "matrixconv" if matrix else "multiconv"
        self.h = vp()
        getattr(self.L, self.pre + "_create")(C.byref(self.h))

    def __getattr__(self, name):
        if name == "setNumInputChannels" and self.pre == "multiconv":
            name = "setNumChannels"
        fn = getattr(load(), self.pre + "_" + name)
        return lambda *a: fn(self.h, *a)

    def setFilters(self, H, fs=48000):
        H = np.ascontiguousarray(H, np.float32)
        rows = (C.POINTER(C.c_float) * H.shape[0])(*[_f(H[i]) for i in range(H.shape[0])])
        getattr(self.L, self.pre + "_setFilters")(self.h, rows, H.shape[0], H.shape[1], fs)

    def process(self, x, nOut):
        x = np.ascontiguousarray(x, np.float32)
        y = np.full((nOut, x.shape[1]), np.nan, np.float32)
        getattr(self.L, self.pre + "_process")(self.h, _rows(x), _rows(y), x.shape[0], nOut, x.shape[1])
        return y

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            getattr(self.L, self.pre + "_destroy")(C.byref(self.h))


class Rfft:
    """saf_rfft_* (saf_utility_fft.c:531-753)"""

    def __init__(self, N):
        self.L = load(); self.N = N; self.h = vp()
        self.L.saf_rfft_create(C.byref(self.h), N)

    def forward(self, x):
        x = np.ascontiguousarray(x, np.float32); X = np.zeros(self.N // 2 + 1, np.complex64)
        self.L.saf_rfft_forward(self.h, _f(x), X.ctypes.data_as(vp)); return X

    def backward(self, X):
        X = np.ascontiguousarray(X, np.complex64); x = np.zeros(self.N, np.float32)
        self.L.saf_rfft_backward(self.h, X.ctypes.data_as(vp), _f(x)); return x

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.saf_rfft_destroy(C.byref(self.h))


class TvConvExample:
    """tvconv example operator (examples/include/tvconv.h) with IRs / listener positions installed directly"""

    def __init__(self):
        self.L = load(); self.h = vp()
        self.L.tvconv_create(C.byref(self.h))

    def __getattr__(self, name):
        fn = getattr(load(), "tvconv_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def setIRsAndPositions(self, irs, positions, fs=48000):
        irs = np.ascontiguousarray(irs, np.float32); pos = np.ascontiguousarray(positions, np.float32)
        flat = irs.reshape(irs.shape[0], -1)
        rows = (C.POINTER(C.c_float) * irs.shape[0])(*[_f(flat[i]) for i in range(irs.shape[0])])
        self.L.saf_hip_tvconv_setIRsAndPositions(self.h, rows, _f(pos), None, irs.shape[0], irs.shape[1], irs.shape[2], fs)

    def process(self, x, nOut):
        x = np.ascontiguousarray(x, np.float32).reshape(1, -1)
        y = np.full((nOut, x.shape[1]), np.nan, np.float32)
        self.L.tvconv_process(self.h, _rows(x), _rows(y), 1, nOut, x.shape[1])
        return y

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.tvconv_destroy(C.byref(self.h))


class AmbiDrc:
    """ambi_drc operator (examples/include/ambi_drc.h)"""

    def __init__(self, frameSize=128):
        self.L = load(); self.h = vp(); self.F = frameSize
        self.L.saf_hip_ambi_drc_setFrameSize(frameSize)
        self.L.ambi_drc_create(C.byref(self.h))

    def __getattr__(self, name):
        fn = getattr(load(), "ambi_drc_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nSamples=None):
        """x [nCh][F] -> y [nCh][F] (the reference's process has one channel count for both)"""
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((x.shape[0], max(ns, self.F)), np.nan, np.float32)
        self.L.ambi_drc_process(self.h, _rows(x), _rows(y), x.shape[0], ns)
        return y[:, :self.F]

    def process_dev(self, d_in, in_strides, nIn, d_out, out_strides, nFrames):
        self.L.saf_hip_ambi_drc_process_dev(self.h, vp(d_in), *in_strides, nIn, vp(d_out), *out_strides, nFrames)

    def gainTF(self):
        """display ring [133][3000] of the bank being written, write index"""
        rows = self.L.ambi_drc_getGainTF(self.h)
        return np.stack([np.ctypeslib.as_array(rows[b], shape=(3000,)).copy() for b in range(133)]), self.L.ambi_drc_getGainTFwIdx(self.h)

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.ambi_drc_destroy(C.byref(self.h))


class Beamformer:
    """beamformer operator (examples/include/beamformer.h)"""

    def __init__(self, frameSize=128):
        self.L = load(); self.h = vp(); self.F = frameSize
        self.L.saf_hip_beamformer_setFrameSize(frameSize)
        self.L.beamformer_create(C.byref(self.h))

    def __getattr__(self, name):
        fn = getattr(load(), "beamformer_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nOut, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((nOut, max(ns, self.F)), np.nan, np.float32)
        self.L.beamformer_process(self.h, _rows(x), _rows(y), x.shape[0], nOut, ns)
        return y[:, :self.F]

    def process_dev(self, d_in, in_strides, nIn, d_out, out_strides, nOut, nFrames):
        """strides = (frame, ch) in floats"""
        self.L.saf_hip_beamformer_process_dev(self.h, vp(d_in), *in_strides, nIn, vp(d_out), *out_strides, nOut, nFrames)

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.beamformer_destroy(C.byref(self.h))


def rotateAxisCoeffsReal(order, c_n, theta_0, phi_0):
    c = np.ascontiguousarray(c_n, np.float32); out = np.zeros((order + 1) ** 2, np.float32)
    load().rotateAxisCoeffsReal(order, _f(c), C.c_float(theta_0), C.c_float(phi_0), _f(out)); return out


def beamWeights(kind, N):
    """kind: 1 cardioid, 2 hyper-cardioid, 3 max-EV -> b_n[N+1]"""
    b = np.zeros(N + 1, np.float32)
    {1: load().beamWeightsCardioid2Spherical, 2: load().beamWeightsHypercardioid2Spherical, 3: load().beamWeightsMaxEV}[kind](N, _f(b)); return b


def quaternion2rotationMatrix(q):
    """q = (w, x, y, z) -> 3 x 3 (saf_utility_geometry.h:60)"""
    q = np.ascontiguousarray(q, np.float32); R = np.zeros((3, 3), np.float32)
    load().quaternion2rotationMatrix(_f(q), _f(R)); return R


def rotationMatrix2quaternion(R):
    R = np.ascontiguousarray(R, np.float32); q = np.zeros(4, np.float32)
    load().rotationMatrix2quaternion(_f(R), _f(q)); return q


def euler2Quaternion(alpha, beta, gamma, degrees=False, convention=2):
    q = np.zeros(4, np.float32)
    load().euler2Quaternion(C.c_float(alpha), C.c_float(beta), C.c_float(gamma), int(degrees), convention, _f(q)); return q


def quaternion2euler(q, degrees=False, convention=2):
    q = np.ascontiguousarray(q, np.float32); a, b, c = C.c_float(), C.c_float(), C.c_float()
    load().quaternion2euler(_f(q), int(degrees), convention, C.byref(a), C.byref(b), C.byref(c))
    return np.array([a.value, b.value, c.value], np.float32)


class Rotator:
    """rotator operator (examples/include/rotator.h)"""

    def __init__(self, frameSize=64):
        self.L = load(); self.h = vp(); self.F = frameSize
        self.L.saf_hip_rotator_setFrameSize(frameSize)
        self.L.rotator_create(C.byref(self.h))

    def __getattr__(self, name):
        fn = getattr(load(), "rotator_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nOut, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((nOut, max(ns, self.F)), np.nan, np.float32)
        self.L.rotator_process(self.h, _rows(x), _rows(y), x.shape[0], nOut, ns)
        return y[:, :self.F]

    def process_dev(self, d_in, in_strides, nIn, d_out, out_strides, nOut, nFrames):
        """strides = (frame, ch) in floats"""
        self.L.saf_hip_rotator_process_dev(self.h, vp(d_in), *in_strides, nIn, vp(d_out), *out_strides, nOut, nFrames)

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.rotator_destroy(C.byref(self.h))


# ---------------------------------------------------------------- HRIR processing / binauraliser
def estimateITDs(hrirs, fs):
    hrirs = np.ascontiguousarray(hrirs, np.float32)
    N, _, L = hrirs.shape
    out = np.zeros(N, np.float32)
    load().estimateITDs(_f(hrirs), N, L, fs, _f(out))
    return out


def getVoronoiWeights(dirs_deg):
    d = np.ascontiguousarray(dirs_deg, np.float32)
    out = np.zeros(d.shape[0], np.float32)
    load().getVoronoiWeights(_f(d), d.shape[0], 0, _f(out))
    return out


def diffuseFieldEqualiseHRTFs(hrtfs, weights=None):
    h = np.ascontiguousarray(hrtfs, np.complex64).copy()
    nB, _, N = h.shape
    w = None if weights is None else np.ascontiguousarray(weights, np.float32)
    load().diffuseFieldEqualiseHRTFs(N, None, None, nB, _f(w) if w is not None else None, 1, 0, h.ctypes.data_as(vp))
    return h


def setDefaultHRIRs(hrirs, dirs_deg, fs):
    hrirs = np.ascontiguousarray(hrirs, np.float32); d = np.ascontiguousarray(dirs_deg, np.float32)
    load().saf_hip_setDefaultHRIRs(_f(hrirs), _f(d), hrirs.shape[0], hrirs.shape[2], fs)


class Binauraliser:
    """examples/include/binauraliser.h.  `frameSize` plays the role of -DBINAURALISER_FRAME_SIZE, `maxSources` of MAX_NUM_INPUTS."""

    def __init__(self, frameSize=128, maxSources=64):
        self.L = load()
        self.L.saf_hip_binauraliser_setFrameSize(frameSize)
        self.L.saf_hip_binauraliser_setMaxNumSources(maxSources)
        self.h = vp()
        self.F = frameSize
        self.maxSources = maxSources
        self.L.binauraliser_create(C.byref(self.h))
        self.L.saf_hip_binauraliser_setMaxNumSources(64)

    def __getattr__(self, name):
        fn = getattr(load(), "binauraliser_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def setHRIRs(self, hrirs, dirs_deg, fs):
        setDefaultHRIRs(hrirs, dirs_deg, fs)
        self.L.binauraliser_refreshSettings(self.h)

    def process(self, x, nOut=2, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((nOut, max(ns, self.F)), np.nan, np.float32)
        self.L.binauraliser_process(self.h, _rows(x), _rows(y), x.shape[0], nOut, ns)
        return y[:, :self.F]

    def process_dev(self, d_in, in_strides, nIn, d_out, out_strides, nFrames):
        """strides = (frame, ch) in floats."""
        self.L.saf_hip_binauraliser_process_dev(self.h, vp(d_in), *in_strides, nIn, vp(d_out), *out_strides, nFrames)

    def itds(self):
        out = np.zeros(self.L.binauraliser_getNDirs(self.h), np.float32); self.L.saf_hip_binauraliser_getITDs(self.h, _f(out)); return out

    def weights(self):
        out = np.zeros(self.L.binauraliser_getNDirs(self.h), np.float32); self.L.saf_hip_binauraliser_getWeights(self.h, _f(out)); return out

    def hrtf_fb(self):
        out = np.zeros((133, 2, self.L.binauraliser_getNDirs(self.h)), np.complex64)
        self.L.saf_hip_binauraliser_getHRTFfb(self.h, out.ctypes.data_as(vp)); return out

    def hrtf_interp(self, nSrc):
        out = np.zeros((nSrc, 133, 2), np.complex64)
        self.L.saf_hip_binauraliser_getHRTFinterp(self.h, out.ctypes.data_as(vp)); return out

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.binauraliser_destroy(C.byref(self.h))


class BinauraliserNF(Binauraliser):
    """examples/include/binauraliser_nf.h: a Binauraliser whose sources carry distances (binauraliser_* calls apply to it)."""

    def __init__(self, frameSize=128, maxSources=64):
        self.L = load()
        self.L.saf_hip_binauraliser_setFrameSize(frameSize)
        self.L.saf_hip_binauraliser_setMaxNumSources(maxSources)
        self.h = vp()
        self.F = frameSize
        self.maxSources = maxSources
        self.L.binauraliserNF_create(C.byref(self.h))
        self.L.saf_hip_binauraliser_setMaxNumSources(64)

    def initCodec(self):
        self.L.binauraliserNF_initCodec(self.h)

    def process(self, x, nOut=2, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((nOut, max(ns, self.F)), np.nan, np.float32)
        self.L.binauraliserNF_process(self.h, _rows(x), _rows(y), x.shape[0], nOut, ns)
        return y[:, :self.F]

    def processFD(self, x, nOut=2, nSamples=None):
        """binauraliserNF_processFD (binauraliser_nf.h:135)"""
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((nOut, max(ns, self.F)), np.nan, np.float32)
        self.L.binauraliserNF_processFD(self.h, _rows(x), _rows(y), x.shape[0], nOut, ns)
        return y[:, :self.F]

    def process_dev(self, d_in, in_strides, nIn, d_out, out_strides, nFrames):
        self.L.saf_hip_binauraliserNF_process_dev(self.h, vp(d_in), *in_strides, nIn, vp(d_out), *out_strides, nFrames)

    def setSourceDist_m(self, i, d):
        self.L.binauraliserNF_setSourceDist_m(self.h, i, C.c_float(d))

    def getSourceDist_m(self, i):
        return self.L.binauraliserNF_getSourceDist_m(self.h, i)

    def setInputConfigPreset(self, preset):
        self.L.binauraliserNF_setInputConfigPreset(self.h, preset)

    def getFarfieldThresh_m(self):
        return self.L.binauraliserNF_getFarfieldThresh_m(self.h)

    def getFarfieldHeadroom(self):
        return self.L.binauraliserNF_getFarfieldHeadroom(self.h)

    def getNearfieldLimit_m(self):
        return self.L.binauraliserNF_getNearfieldLimit_m(self.h)

    def hrtf_nf(self, nSrc):
        out = np.zeros((nSrc, 133, 2), np.complex64)
        self.L.saf_hip_binauraliserNF_getHRTFnf(self.h, out.ctypes.data_as(vp)); return out

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.binauraliserNF_destroy(C.byref(self.h))


# DVF near-field filters (saf_utility_dvf.h) and evalIIRTransferFunctionf (saf_utility_filters.h); host functions
def calcDVFShelfParams(i, rho):
    o = np.zeros(3, np.float32)
    load().calcDVFShelfParams(int(i), C.c_float(rho), _f(o[0:1]), _f(o[1:2]), _f(o[2:3])); return tuple(o)


def interpDVFShelfParams(theta, rho):
    o = np.zeros(3, np.float32)
    load().interpDVFShelfParams(C.c_float(theta), C.c_float(rho), _f(o[0:1]), _f(o[1:2]), _f(o[2:3])); return tuple(o)


def dvfShelfCoeffs(g0, gInf, fc, fs):
    o = np.zeros(3, np.float32)
    load().dvfShelfCoeffs(C.c_float(g0), C.c_float(gInf), C.c_float(fc), C.c_float(fs), _f(o[0:1]), _f(o[1:2]), _f(o[2:3])); return tuple(o)


def calcDVFCoeffs(alpha, rho, fs):
    b = np.zeros(2, np.float32); a = np.ones(2, np.float32)
    load().calcDVFCoeffs(C.c_float(alpha), C.c_float(rho), C.c_float(fs), _f(b), _f(a)); return b, a


def doaToIpsiInteraural(azi, elev):
    al = np.zeros(2, np.float32); be = np.zeros(2, np.float32)
    load().doaToIpsiInteraural(C.c_float(azi), C.c_float(elev), _f(al), _f(be)); return al, be


def evalIIRTransferFunctionf(b, a, freqs, fs, mag2dB=0):
    b = np.ascontiguousarray(b, np.float32); a = np.ascontiguousarray(a, np.float32); f = np.ascontiguousarray(freqs, np.float32)
    mag = np.zeros(f.size, np.float32); ph = np.zeros(f.size, np.float32)
    load().evalIIRTransferFunctionf(_f(b), _f(a), b.size, _f(f), f.size, C.c_float(fs), mag2dB, _f(mag), _f(ph)); return mag, ph


class BinauraliserBatch:
    """saf_hip_binauraliser_batch_*: nInst initialised handles, device-resident blocks."""

    def __init__(self, bins, maxFramesPerCall):
        self.L = load()
        self.bins = list(bins)
        arr = (vp * len(self.bins))(*[b.h for b in self.bins])
        self.hb = vp(self.L.saf_hip_binauraliser_batch_create(arr, len(self.bins), maxFramesPerCall))

    def process_ptr(self, d_in, in_strides, nIn, d_out, out_strides, nFrames):
        """strides = (inst, frame, ch) in floats."""
        self.L.saf_hip_binauraliser_batch_process(self.hb, vp(d_in), *in_strides, nIn, vp(d_out), *out_strides, nFrames)

    def __del__(self):
        if getattr(self, "hb", None) and C is not None:
            self.L.saf_hip_binauraliser_batch_destroy(C.byref(self.hb))


# ---------------------------------------------------------------- binaural Ambisonic decoding / ambi_bin
def getSHrotMtxReal(R, order):
    R = np.ascontiguousarray(R, np.float32).reshape(9); n = (order + 1) ** 2
    out = np.zeros((n, n), np.float32)
    load().getSHrotMtxReal(_f(R), _f(out), order); return out


def yawPitchRoll2Rzyx(yaw, pitch, roll, rpy=0):
    R = np.zeros(9, np.float32)
    load().yawPitchRoll2Rzyx(C.c_float(yaw), C.c_float(pitch), C.c_float(roll), rpy, _f(R)); return R.reshape(3, 3)


def getBinauralAmbiDecoderMtx(hrtfs, dirs_deg, method, order, freqVector=None, itd_s=None, weights=None, diffMatching=0, maxRE=0):
    H = np.ascontiguousarray(hrtfs, np.complex64); nBands, _, N = H.shape
    d = np.ascontiguousarray(dirs_deg, np.float32)
    fv = np.ascontiguousarray(freqVector if freqVector is not None else np.zeros(nBands), np.float32)
    it = np.ascontiguousarray(itd_s if itd_s is not None else np.zeros(N), np.float32)
    w = np.ascontiguousarray(weights, np.float32) if weights is not None else None
    out = np.zeros((nBands, 2, (order + 1) ** 2), np.complex64)
    load().getBinauralAmbiDecoderMtx(H.ctypes.data_as(vp), _f(d), N, nBands, method, order, _f(fv), _f(it), _f(w) if w is not None else None,
                                     diffMatching, maxRE, out.ctypes.data_as(vp))
    return out


def getBinauralAmbiDecoderFilters(hrtfs, dirs_deg, fftSize, fs, method, order, itd_s=None, weights=None, diffMatching=0, maxRE=0):
    """hrtfs [fftSize/2+1][2][N] complex -> filters [2][nSH][fftSize]"""
    H = np.ascontiguousarray(hrtfs, np.complex64); nBins, _, N = H.shape
    assert nBins == fftSize // 2 + 1
    d = np.ascontiguousarray(dirs_deg, np.float32)
    it = np.ascontiguousarray(itd_s if itd_s is not None else np.zeros(N), np.float32)
    w = np.ascontiguousarray(weights, np.float32) if weights is not None else None
    out = np.zeros((2, (order + 1) ** 2, fftSize), np.float32)
    load().getBinauralAmbiDecoderFilters(H.ctypes.data_as(vp), _f(d), N, fftSize, C.c_float(fs), method, order, _f(it), _f(w) if w is not None else None,
                                         diffMatching, maxRE, _f(out))
    return out


def truncationEQ(w_n, order_truncated, order_target, kr, softThreshold):
    w = np.ascontiguousarray(w_n, np.float32); k = np.ascontiguousarray(kr, np.float64); g = np.zeros(k.shape[0], np.float32)
    load().truncationEQ(_f(w), order_truncated, order_target, k.ctypes.data_as(C.POINTER(C.c_double)), k.shape[0], C.c_float(softThreshold), _f(g)); return g


class AmbiBin:
    """examples/include/ambi_bin.h.  `frameSize` plays the role of -DAMBI_BIN_FRAME_SIZE."""

    def __init__(self, frameSize=128):
        self.L = load()
        self.L.saf_hip_ambi_bin_setFrameSize(frameSize)
        self.h = vp(); self.F = frameSize
        self.L.ambi_bin_create(C.byref(self.h))

    def __getattr__(self, name):
        fn = getattr(load(), "ambi_bin_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def setHRIRs(self, hrirs, dirs_deg, fs):
        setDefaultHRIRs(hrirs, dirs_deg, fs)
        self.L.ambi_bin_refreshParams(self.h)

    def process(self, x, nOut=2, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((nOut, max(ns, self.F)), np.nan, np.float32)
        self.L.ambi_bin_process(self.h, _rows(x), _rows(y), x.shape[0], nOut, ns)
        return y[:, :self.F]

    def process_dev(self, d_in, in_strides, nIn, d_out, out_strides, nFrames):
        """strides = (frame, ch) in floats."""
        self.L.saf_hip_ambi_bin_process_dev(self.h, vp(d_in), *in_strides, nIn, vp(d_out), *out_strides, nFrames)

    def decMtx(self, nSH):
        out = np.zeros((133, 2, nSH), np.complex64)
        self.L.saf_hip_ambi_bin_getDecoderMtx(self.h, out.ctypes.data_as(vp)); return out

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.ambi_bin_destroy(C.byref(self.h))


# ---------------------------------------------------------------- panner
class Panner:
    """examples/include/panner.h.  `frameSize` plays the role of -DPANNER_FRAME_SIZE."""

    def __init__(self, frameSize=128):
        self.L = load()
        self.L.saf_hip_panner_setFrameSize(frameSize)
        self.h = vp()
        self.F = frameSize
        self.L.panner_create(C.byref(self.h))

    def __getattr__(self, name):
        fn = getattr(load(), "panner_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nOut, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.full((nOut, max(ns, self.F)), np.nan, np.float32)
        self.L.panner_process(self.h, _rows(x), _rows(y), x.shape[0], nOut, ns)
        return y[:, :self.F]

    def process_dev(self, d_in, in_strides, nIn, d_out, out_strides, nFrames):
        """strides = (frame, ch) in floats."""
        self.L.saf_hip_panner_process_dev(self.h, vp(d_in), *in_strides, nIn, vp(d_out), *out_strides, nFrames)

    def gains(self):
        """G_src [133][64][64] (band, source, loudspeaker)"""
        out = np.zeros((133, 64, 64), np.float32); self.L.saf_hip_panner_getGains(self.h, _f(out)); return out

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.panner_destroy(C.byref(self.h))


def getPvalues(DTT, freq):
    f = np.ascontiguousarray(freq, np.float32)
    out = np.zeros(f.shape[0], np.float32)
    load().getPvalues(C.c_float(DTT), _f(f), f.shape[0], _f(out))
    return out


# ---------------------------------------------------------------- activity-map generators (saf_sh.c:1544-1858)
def _cxY(Cx, Y_grid):
    Cx = np.ascontiguousarray(Cx, np.complex64)
    Y = np.ascontiguousarray(np.asarray(Y_grid, np.float32).astype(np.complex64))
    return Cx, Y, Y.shape[1]


def generatePWDmap(order, Cx, Y_grid):
    Cx, Y, G = _cxY(Cx, Y_grid); pm = np.zeros(G, np.float32)
    load().generatePWDmap(order, Cx.ctypes.data_as(vp), Y.ctypes.data_as(vp), G, _f(pm)); return pm


def generateMVDRmap(order, Cx, Y_grid, regPar=8.0, weights=False):
    Cx, Y, G = _cxY(Cx, Y_grid); pm = np.zeros(G, np.float32)
    w = np.zeros(((order + 1) ** 2, G), np.complex64)
    load().generateMVDRmap(order, Cx.ctypes.data_as(vp), Y.ctypes.data_as(vp), G, C.c_float(regPar), _f(pm), w.ctypes.data_as(vp) if weights else None)
    return (pm, w) if weights else pm


def generateCroPaCLCMVmap(order, Cx, Y_grid, regPar=8.0, lam=0.0):
    Cx, Y, G = _cxY(Cx, Y_grid); pm = np.zeros(G, np.float32)
    load().generateCroPaCLCMVmap(order, Cx.ctypes.data_as(vp), Y.ctypes.data_as(vp), G, C.c_float(regPar), C.c_float(lam), _f(pm)); return pm


def generateMUSICmap(order, Cx, Y_grid, nSources, logScale=0):
    Cx, Y, G = _cxY(Cx, Y_grid); pm = np.zeros(G, np.float32)
    load().generateMUSICmap(order, Cx.ctypes.data_as(vp), Y.ctypes.data_as(vp), nSources, G, logScale, _f(pm)); return pm


def generateMinNormMap(order, Cx, Y_grid, nSources, logScale=0):
    Cx, Y, G = _cxY(Cx, Y_grid); pm = np.zeros(G, np.float32)
    load().generateMinNormMap(order, Cx.ctypes.data_as(vp), Y.ctypes.data_as(vp), nSources, G, logScale, _f(pm)); return pm


class SphScan:
    """sphPWD (kind="PWD", input Cx) / sphMUSIC (kind="MUSIC", input Vn) scanning objects (saf_sh.c:1042-1306)"""

    def __init__(self, kind, order, grid_dirs_deg):
        self.L = load(); self.kind = "sph" + kind; self.h = vp()
        g = np.ascontiguousarray(grid_dirs_deg, np.float32)
        self.G = g.shape[0]
        getattr(self.L, self.kind + "_create")(C.byref(self.h), order, _f(g), self.G)

    def compute(self, M, nSrcs):
        M = np.ascontiguousarray(M, np.complex64)
        P = np.zeros(self.G, np.float32); pk = (C.c_int * nSrcs)()
        getattr(self.L, self.kind + "_compute")(self.h, M.ctypes.data_as(vp), nSrcs, _f(P), pk)
        return P, list(pk)

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            getattr(self.L, self.kind + "_destroy")(C.byref(self.h))


# ---------------------------------------------------------------- powermap
class Powermap:
    """examples/include/powermap.h.  `frameSize` plays the role of -DPOWERMAP_FRAME_SIZE."""

    def __init__(self, frameSize=1024):
        self.L = load()
        self.L.saf_hip_powermap_setFrameSize(frameSize)
        self.h = vp()
        self.F = frameSize
        self.L.powermap_create(C.byref(self.h))

    def __getattr__(self, name):
        fn = getattr(load(), "powermap_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def analysis(self, x, isPlaying=1):
        x = np.ascontiguousarray(x, np.float32)
        self.L.powermap_analysis(self.h, _rows(x), x.shape[0], x.shape[1], isPlaying)

    def analysis_dev(self, d_in, strides, nIn, nFrames):
        """strides = (frame, ch) in floats."""
        self.L.saf_hip_powermap_analysis_dev(self.h, vp(d_in), *strides, nIn, nFrames)

    def Cx(self, nSH):
        out = np.zeros((133, nSH, nSH), np.complex64)
        self.L.saf_hip_powermap_getCx(self.h, out.ctypes.data_as(vp)); return out

    def rawPmap(self):
        out = np.zeros(4096, np.float32)
        n = self.L.saf_hip_powermap_getRawPmap(self.h, _f(out)); return out[:n].copy()

    def getPmap(self):
        gd, pm = fp(), fp()
        n, w, hf, ar = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if not self.L.powermap_getPmap(self.h, C.byref(gd), C.byref(pm), C.byref(n), C.byref(w), C.byref(hf), C.byref(ar)):
            return None
        return np.ctypeslib.as_array(pm, shape=(n.value,)).copy()

    def __del__(self):
        if getattr(self, "h", None) and C is not None:
            self.L.powermap_destroy(C.byref(self.h))


class PowermapBatch:
    """saf_hip_powermap_batch_*: nInst initialised handles (PWD mode), device-resident frames."""

    def __init__(self, pms, maxFramesPerCall):
        self.L = load()
        self.pms = list(pms)
        arr = (vp * len(self.pms))(*[p.h for p in self.pms])
        self.nInst = len(self.pms)
        self.hb = vp(self.L.saf_hip_powermap_batch_create(arr, self.nInst, maxFramesPerCall))

    def analysis_ptr(self, d_in, strides, nIn, nFrames):
        """strides = (inst, frame, ch) in floats."""
        self.L.saf_hip_powermap_batch_analysis(self.hb, vp(d_in), *strides, nIn, nFrames)

    def Cx(self, i, nSH):
        out = np.zeros((133, nSH, nSH), np.complex64)
        self.L.saf_hip_powermap_batch_getCx(self.hb, i, out.ctypes.data_as(vp)); return out

    def __del__(self):
        if getattr(self, "hb", None) and C is not None:
            self.L.saf_hip_powermap_batch_destroy(C.byref(self.hb))
