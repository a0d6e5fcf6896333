"""ctypes binding of the CPU oracle (oracle/libsaf_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the product package.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
TABLES = ROOT / "spatial_audio_framework_amd" / "data" / "saf_tables.bin"
_LIB = None

c_f = C.POINTER(C.c_float)
c_d = C.POINTER(C.c_double)
c_i = C.POINTER(C.c_int)
vp = C.c_void_p


def build(force=False):
    so = HERE / "libsaf_oracle.so"
    srcs = [HERE / n for n in ("orc_core.c", "orc_sh.c", "orc_examples.c", "orc_binaural.c", "orc_powermap.c", "saf_oracle.h")]
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-s", "-C", str(HERE)])
    if Path("/root/reference/framework/resources/kissFFT/kiss_fftr.c").exists():
        ref = HERE / "_ref" / "libkissfft_ref.so"
        if force or not ref.exists():
            subprocess.check_call(["make", "-s", "-C", str(HERE), "ref"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(str(so))
        L.orc_tables_load.argtypes = [C.c_char_p]
        L.orc_tables_load.restype = C.c_int
        rc = L.orc_tables_load(str(TABLES).encode())
        if rc != 0:
            raise RuntimeError(f"oracle: cannot load tables {TABLES} (rc={rc})")
        L.orc_table.restype = c_f
        L.orc_table.argtypes = [C.c_char_p, c_i, c_i]
        L.orc_ambi_dec_getDecMtx.restype = c_f
        L.orc_ambi_dec_getFreqVector.restype = c_f
        L.orc_ambi_dec_getMnorm.restype = C.c_float
        L.orc_afSTFT_getNBands.restype = C.c_int
        L.orc_afSTFT_getProcDelay.restype = C.c_int
        _LIB = L
    return _LIB


def fptr(a):
    return a.ctypes.data_as(c_f)


def table(name):
    d0, d1 = C.c_int(), C.c_int()
    p = lib().orc_table(name.encode(), C.byref(d0), C.byref(d1))
    if not p:
        raise KeyError(name)
    return np.ctypeslib.as_array(p, shape=(d0.value, d1.value)).copy()


# ------------------------------------------------------------------ FFT
class RFFT:
    def __init__(self, N):
        self.N = N
        self.h = vp()
        lib().orc_rfft_create(C.byref(self.h), C.c_int(N))

    def forward(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(self.N // 2 + 1, np.complex64)
        lib().orc_rfft_forward(self.h, fptr(x), out.ctypes.data_as(vp))
        return out

    def backward(self, X):
        X = np.ascontiguousarray(X, np.complex64)
        out = np.zeros(self.N, np.float32)
        lib().orc_rfft_backward(self.h, X.ctypes.data_as(vp), fptr(out))
        return out

    def __del__(self):
        if self.h:
            lib().orc_rfft_destroy(C.byref(self.h))


# ------------------------------------------------------------------ afSTFT
class AfSTFT:
    """afSTFT oracle; spectra returned as [nBands][dataFD_nCH][nHops] complex64."""

    def __init__(self, nCHin, nCHout, hopsize=128, lowDelay=0, hybrid=1, fmt=0):
        self.h = vp()
        self.nCHin, self.nCHout, self.hop = nCHin, nCHout, hopsize
        lib().orc_afSTFT_create(C.byref(self.h), nCHin, nCHout, hopsize, lowDelay, hybrid, fmt)
        self.nBands = lib().orc_afSTFT_getNBands(self.h)
        self.delay = lib().orc_afSTFT_getProcDelay(self.h)

    def forward(self, x, nCH_alloc=None):
        x = np.ascontiguousarray(x, np.float32)
        nCH, F = x.shape
        assert nCH == self.nCHin
        nH = F // self.hop
        nA = nCH_alloc or nCH
        out = np.zeros((self.nBands, nA, nH), np.complex64)
        lib().orc_afSTFT_forward_knownDimensions(self.h, fptr(x), F, nA, nH, out.ctypes.data_as(vp))
        return out

    def backward(self, X):
        X = np.ascontiguousarray(X, np.complex64)
        nB, nA, nH = X.shape
        F = nH * self.hop
        out = np.zeros((self.nCHout, F), np.float32)
        lib().orc_afSTFT_backward_knownDimensions(self.h, X.ctypes.data_as(vp), F, nA, nH, fptr(out))
        return out

    def channelChange(self, nin, nout):
        lib().orc_afSTFT_channelChange(self.h, nin, nout)
        self.nCHin, self.nCHout = nin, nout

    def clearBuffers(self):
        lib().orc_afSTFT_clearBuffers(self.h)

    def centreFreqs(self, fs):
        f = np.zeros(self.nBands, np.float32)
        lib().orc_afSTFT_getCentreFreqs(self.h, C.c_float(fs), self.nBands, fptr(f))
        return f

    def __del__(self):
        if self.h:
            lib().orc_afSTFT_destroy(C.byref(self.h))


def centreFreqs_nullHandle(fs):
    f = np.zeros(133, np.float32)
    lib().orc_afSTFT_getCentreFreqs(None, C.c_float(fs), 133, fptr(f))
    return f


def FIRtoFilterbankCoeffs(hIR, hop=128, LD=0, hybrid=1):
    hIR = np.ascontiguousarray(hIR, np.float32)
    nd, nch, L = hIR.shape
    nB = hop + (5 if hybrid else 1)
    out = np.zeros((nB, nch, nd), np.complex64)
    lib().orc_afSTFT_FIRtoFilterbankCoeffs(fptr(hIR), nd, nch, L, hop, LD, hybrid, out.ctypes.data_as(vp))
    return out


# ------------------------------------------------------------------ SH / HOA
def _sh(fn, order, dirs):
    dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 2)
    Y = np.zeros(((order + 1) ** 2, dirs.shape[0]), np.float32)
    getattr(lib(), fn)(order, fptr(dirs), dirs.shape[0], fptr(Y))
    return Y


def getSHreal(order, dirs_rad): return _sh("orc_getSHreal", order, dirs_rad)
def getSHreal_recur(order, dirs_rad): return _sh("orc_getSHreal_recur", order, dirs_rad)
def getRSH(order, dirs_deg): return _sh("orc_getRSH", order, dirs_deg)
def getRSH_recur(order, dirs_deg): return _sh("orc_getRSH_recur", order, dirs_deg)


def getMaxREweights(order, diag=False):
    n = (order + 1) ** 2
    a = np.zeros((n, n) if diag else n, np.float32)
    lib().orc_getMaxREweights(order, int(diag), fptr(a))
    return a


def getLoudspeakerDecoderMtx(ls_dirs_deg, method, order, maxrE=0):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    M = np.zeros((d.shape[0], (order + 1) ** 2), np.float32)
    lib().orc_getLoudspeakerDecoderMtx(fptr(d), d.shape[0], method, order, maxrE, fptr(M))
    return M


def pinv(A):
    A = np.ascontiguousarray(A, np.float32)
    out = np.zeros((A.shape[1], A.shape[0]), np.float32)
    lib().orc_pinv(fptr(A), A.shape[0], A.shape[1], fptr(out))
    return out


def findLsTriplets(ls_dirs_deg, omitLarge=0):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    verts, faces = c_f(), c_i()
    nv, nf = C.c_int(), C.c_int()
    lib().orc_findLsTriplets(fptr(d), d.shape[0], omitLarge, C.byref(verts), C.byref(nv), C.byref(faces), C.byref(nf))
    V = np.ctypeslib.as_array(verts, shape=(nv.value, 3)).copy()
    Fc = np.ctypeslib.as_array(faces, shape=(nf.value, 3)).copy() if nf.value else np.zeros((0, 3), np.int32)
    return V, Fc


def generateVBAPgainTable3D_srcs(src_dirs_deg, ls_dirs_deg, omitLarge=0, dummies=0, spread=0.0):
    s = np.ascontiguousarray(src_dirs_deg, np.float32).reshape(-1, 2)
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    g = c_f()
    n, nt = C.c_int(), C.c_int()
    lib().orc_generateVBAPgainTable3D_srcs(fptr(s), s.shape[0], fptr(d), d.shape[0], omitLarge, dummies, C.c_float(spread),
                                           C.byref(g), C.byref(n), C.byref(nt))
    return np.ctypeslib.as_array(g, shape=(n.value, d.shape[0])).copy(), nt.value


def generateVBAPgainTable3D(ls_dirs_deg, az_res, el_res, omitLarge=0, dummies=0, spread=0.0):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    g = c_f()
    n, nt = C.c_int(), C.c_int()
    lib().orc_generateVBAPgainTable3D(fptr(d), d.shape[0], az_res, el_res, omitLarge, dummies, C.c_float(spread),
                                      C.byref(g), C.byref(n), C.byref(nt))
    return np.ctypeslib.as_array(g, shape=(n.value, d.shape[0])).copy(), nt.value


def generateVBAPgainTable2D(ls_dirs_deg, az_res):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    n = lib().orc_generateVBAPgainTable2D(fptr(d), d.shape[0], az_res, None)
    g = np.zeros((n, d.shape[0]), np.float32)
    lib().orc_generateVBAPgainTable2D(fptr(d), d.shape[0], az_res, fptr(g))
    return g, d.shape[0]


def generateVBAPgainTable2D_srcs(src_azi_deg, ls_dirs_deg):
    a = np.ascontiguousarray(src_azi_deg, np.float32).reshape(-1)
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    g = np.zeros((a.shape[0], d.shape[0]), np.float32)
    lib().orc_vbap2D_table(fptr(a), a.shape[0], fptr(d), d.shape[0], fptr(g))
    return g, d.shape[0]


def findLsPairs(ls_dirs_deg):
    d = np.ascontiguousarray(ls_dirs_deg, np.float32).reshape(-1, 2)
    p = np.zeros((d.shape[0], 2), np.int32)
    lib().orc_findLsPairs(fptr(d), d.shape[0], p.ctypes.data_as(C.POINTER(C.c_int)))
    return p


def getSpreadSrcDirs3D(azi_rad, elev_rad, spread_deg, num_src=8, num_rings=1):
    U = np.zeros((num_rings * num_src + 1, 3), np.float32)
    lib().orc_getSpreadSrcDirs3D(C.c_float(azi_rad), C.c_float(elev_rad), C.c_float(spread_deg), num_src, num_rings, fptr(U))
    return U


def compressVBAPgainTable3D(gt):
    gt = np.ascontiguousarray(gt, np.float32)
    comp = np.zeros((gt.shape[0], 3), np.float32)
    idx = np.zeros((gt.shape[0], 3), np.int32)
    lib().orc_compressVBAPgainTable3D(fptr(gt), gt.shape[0], gt.shape[1], fptr(comp), idx.ctypes.data_as(c_i))
    return comp, idx


# ------------------------------------------------------------------ operators
def _chan_ptrs(a):
    arr = (c_f * a.shape[0])()
    for i in range(a.shape[0]):
        arr[i] = a[i].ctypes.data_as(c_f)
    return arr


class AmbiDec:
    def __init__(self, frameSize=512):
        self.h = vp()
        self.F = frameSize
        lib().orc_ambi_dec_create(C.byref(self.h), frameSize)

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_ambi_dec_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def setLoudspeakersDeg(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 2)
        lib().orc_ambi_dec_setLoudspeakers(self.h, fptr(d), d.shape[0])

    def process(self, x, nOut):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((nOut, self.F), np.float32)
        lib().orc_ambi_dec_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], nOut, x.shape[1])
        return y

    def setHRIRs(self, hrirs, dirs_deg, fs):
        hrirs = np.ascontiguousarray(hrirs, np.float32); d = np.ascontiguousarray(dirs_deg, np.float32)
        lib().orc_ambi_dec_setHRIRs(self.h, fptr(hrirs), fptr(d), hrirs.shape[0], hrirs.shape[2], fs)

    def hrtf_interp(self, nLS):
        lib().orc_ambi_dec_getHRTFinterp.restype = vp
        p = C.cast(lib().orc_ambi_dec_getHRTFinterp(self.h), C.POINTER(C.c_float))
        return np.ctypeslib.as_array(p, shape=(64, 133, 2, 2)).copy().view(np.complex64)[:nLS, ..., 0]

    def decMtx(self, dec, order, maxrE, nLS):
        p = lib().orc_ambi_dec_getDecMtx(self.h, dec, order, maxrE)
        return np.ctypeslib.as_array(p, shape=(nLS, (order + 1) ** 2)).copy()

    def Mnorm(self, dec, order, which):
        return lib().orc_ambi_dec_getMnorm(self.h, dec, order, which)

    def freqVector(self):
        return np.ctypeslib.as_array(lib().orc_ambi_dec_getFreqVector(self.h), shape=(133,)).copy()

    def stageTimes(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        lib().orc_ambi_dec_getStageTimes(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def __del__(self):
        if self.h:
            lib().orc_ambi_dec_destroy(C.byref(self.h))


class AmbiEnc:
    def __init__(self, frameSize=64):
        self.h = vp()
        self.F = frameSize
        lib().orc_ambi_enc_create(C.byref(self.h), frameSize)

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_ambi_enc_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nOut):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((nOut, self.F), np.float32)
        lib().orc_ambi_enc_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], nOut, x.shape[1])
        return y

    def __del__(self):
        if self.h:
            lib().orc_ambi_enc_destroy(C.byref(self.h))


class AmbiDrc:
    def __init__(self, frameSize=128):
        self.h = vp()
        self.F = frameSize
        lib().orc_ambi_drc_create(C.byref(self.h), frameSize)
        lib().orc_ambi_drc_getLastGains.restype = c_f

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_ambi_drc_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((x.shape[0], self.F), np.float32)
        lib().orc_ambi_drc_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], x.shape[1] if nSamples is None else nSamples)
        return y

    def lastGains(self):
        return np.ctypeslib.as_array(lib().orc_ambi_drc_getLastGains(self.h), shape=(133, self.F // 128)).copy()

    def __del__(self):
        if self.h:
            lib().orc_ambi_drc_destroy(C.byref(self.h))


class Beamformer:
    def __init__(self, frameSize=128):
        self.h = vp()
        self.F = frameSize
        lib().orc_beamformer_create(C.byref(self.h), frameSize)

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_beamformer_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nOut, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((nOut, self.F), np.float32)
        lib().orc_beamformer_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], nOut, x.shape[1] if nSamples is None else nSamples)
        return y

    def __del__(self):
        if self.h:
            lib().orc_beamformer_destroy(C.byref(self.h))


def rotateAxisCoeffsReal(order, c_n, theta_0, phi_0):
    c = np.ascontiguousarray(c_n, np.float32); out = np.zeros((order + 1) ** 2, np.float32)
    lib().orc_rotateAxisCoeffsReal(order, fptr(c), C.c_float(theta_0), C.c_float(phi_0), fptr(out)); return out


def beamWeights(kind, N):
    b = np.zeros(N + 1, np.float32)
    {1: lib().orc_beamWeightsCardioid2Spherical, 2: lib().orc_beamWeightsHypercardioid2Spherical, 3: lib().orc_beamWeightsMaxEV}[kind](N, fptr(b)); return b


def convertHOAChannelConvention(sig, order, inConv, outConv):
    """1 = ACN, 2 = FuMa (saf_hoa.c:40-76)"""
    x = np.ascontiguousarray(sig, np.float32).copy()
    lib().orc_convertHOAChannelConvention(fptr(x), order, x.shape[1], inConv, outConv); return x


def convertHOANormConvention(sig, order, inConv, outConv):
    """1 = N3D, 2 = SN3D, 3 = FuMa (saf_hoa.c:78-116)"""
    x = np.ascontiguousarray(sig, np.float32).copy()
    lib().orc_convertHOANormConvention(fptr(x), order, x.shape[1], inConv, outConv); return x


def quaternion2rotationMatrix(q):
    """q = (w, x, y, z) -> 3 x 3 (saf_utility_geometry.c:89)"""
    q = np.ascontiguousarray(q, np.float32); R = np.zeros(9, np.float32)
    lib().orc_quaternion2rotationMatrix(fptr(q), fptr(R)); return R.reshape(3, 3)


def rotationMatrix2quaternion(R):
    R = np.ascontiguousarray(R, np.float32).reshape(9); q = np.zeros(4, np.float32)
    lib().orc_rotationMatrix2quaternion(fptr(R), fptr(q)); return q


def euler2Quaternion(alpha, beta, gamma, degrees=False, convention=2):
    """convention 2 = yaw-pitch-roll, 3 = roll-pitch-yaw (EULER_ROTATION_CONVENTIONS)"""
    k = np.float32(np.pi / 180.0) if degrees else np.float32(1.0)
    q = np.zeros(4, np.float32)
    lib().orc_euler2Quaternion(C.c_float(float(np.float32(alpha) * k)), C.c_float(float(np.float32(beta) * k)), C.c_float(float(np.float32(gamma) * k)), convention, fptr(q))
    return q


def quaternion2euler(q, degrees=False, convention=2):
    q = np.ascontiguousarray(q, np.float32); a, b, c = C.c_float(), C.c_float(), C.c_float()
    lib().orc_quaternion2euler(fptr(q), convention, C.byref(a), C.byref(b), C.byref(c))
    k = np.float32(180.0 / np.pi) if degrees else np.float32(1.0)
    return np.array([a.value, b.value, c.value], np.float32) * k


class Rotator:
    def __init__(self, frameSize=64):
        self.h = vp()
        self.F = frameSize
        lib().orc_rotator_create(C.byref(self.h), frameSize)
        for g in ("Yaw", "Pitch", "Roll", "QuaternionW", "QuaternionX", "QuaternionY", "QuaternionZ"):
            getattr(lib(), "orc_rotator_get" + g).restype = C.c_float

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_rotator_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nOut, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((nOut, self.F), np.float32)
        lib().orc_rotator_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], nOut, x.shape[1] if nSamples is None else nSamples)
        return y

    def __del__(self):
        if self.h:
            lib().orc_rotator_destroy(C.byref(self.h))


class MatrixConv:
    def __init__(self, hop, H, part=1):
        H = np.ascontiguousarray(H, np.float32)
        self.nOut, self.nIn, self.L = H.shape
        self.hop = hop
        self.h = vp()
        lib().orc_matrixConv_create(C.byref(self.h), hop, fptr(H), self.L, self.nIn, self.nOut, part)

    def apply(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((self.nOut, self.hop), np.float32)
        lib().orc_matrixConv_apply(self.h, fptr(x), fptr(y))
        return y

    def __del__(self):
        if self.h:
            lib().orc_matrixConv_destroy(C.byref(self.h))


class MultiConv:
    def __init__(self, hop, H, part=1):
        H = np.ascontiguousarray(H, np.float32)
        self.nCH, self.L = H.shape
        self.hop = hop
        self.h = vp()
        lib().orc_multiConv_create(C.byref(self.h), hop, fptr(H), self.L, self.nCH, part)

    def apply(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((self.nCH, self.hop), np.float32)
        lib().orc_multiConv_apply(self.h, fptr(x), fptr(y))
        return y

    def __del__(self):
        if self.h:
            lib().orc_multiConv_destroy(C.byref(self.h))


class TVConv:
    def __init__(self, hop, H, initIdx=0):
        H = np.ascontiguousarray(H, np.float32)          # [nIRs][nCHout][L]
        self.nIRs, self.nOut, self.L = H.shape
        self.hop = hop
        self.h = vp()
        lib().orc_TVConv_create(C.byref(self.h), hop, fptr(H), self.L, self.nIRs, self.nOut, initIdx)

    def apply(self, x, irIdx):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((self.nOut, self.hop), np.float32)
        lib().orc_TVConv_apply(self.h, fptr(x), fptr(y), int(irIdx))
        return y

    def __del__(self):
        if self.h:
            lib().orc_TVConv_destroy(C.byref(self.h))


class ConvExample:
    """matrixconv (matrix=1) / multiconv (matrix=0) example operators"""

    def __init__(self, matrix):
        self.h = vp()
        lib().orc_convex_create(C.byref(self.h), matrix)

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_convex_" + name)
        return lambda *a: fn(self.h, *a)

    def setFilters(self, H, fs=48000):
        H = np.ascontiguousarray(H, np.float32)
        lib().orc_convex_setFilters(self.h, fptr(H), H.shape[0], H.shape[1])

    def process(self, x, nOut):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros((nOut, x.shape[1]), np.float32)
        lib().orc_convex_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], nOut, x.shape[1])
        return y

    def __del__(self):
        if self.h:
            lib().orc_convex_destroy(C.byref(self.h))


class TvConvExample:
    """tvconv example operator with injected IRs / listener positions"""

    def __init__(self):
        self.h = vp()
        lib().orc_tvconvex_create(C.byref(self.h))

    def init(self, fs, hostBlockSize):
        lib().orc_tvconvex_init(self.h, hostBlockSize)

    def setIRsAndPositions(self, irs, positions, fs=48000):
        irs = np.ascontiguousarray(irs, np.float32); pos = np.ascontiguousarray(positions, np.float32)
        lib().orc_tvconvex_setIRsAndPositions(self.h, fptr(irs), fptr(pos), irs.shape[0], irs.shape[1], irs.shape[2])

    def setTargetPosition(self, v, dim):
        lib().orc_tvconvex_setTargetPosition(self.h, C.c_float(v), dim)

    def getListenerPositionIdx(self):
        return lib().orc_tvconvex_getListenerPositionIdx(self.h)

    def process(self, x, nOut):
        x = np.ascontiguousarray(x, np.float32).reshape(1, -1)
        y = np.zeros((nOut, x.shape[1]), np.float32)
        lib().orc_tvconvex_process(self.h, _chan_ptrs(x), _chan_ptrs(y), 1, nOut, x.shape[1])
        return y

    def __del__(self):
        if self.h:
            lib().orc_tvconvex_destroy(C.byref(self.h))


def binaural_mac(inTF, hrtf, nSrc, scale):
    """inTF [nBands][nSrcStride][T] c64, hrtf [nSrc][nBands][2] c64 -> [nBands][2][T]."""
    inTF = np.ascontiguousarray(inTF, np.complex64)
    hrtf = np.ascontiguousarray(hrtf, np.complex64)
    nB, stride, T = inTF.shape
    out = np.zeros((nB, 2, T), np.complex64)
    lib().orc_binaural_mac(inTF.ctypes.data_as(vp), hrtf.ctypes.data_as(vp), nB, nSrc, stride, T, C.c_float(scale), out.ctypes.data_as(vp))
    return out


# ------------------------------------------------------------------ the reference's own KissFFT (oracle/_ref)
class KissRef:
    """kiss_fftr / kiss_fftri compiled from /root/reference (only where that checkout exists)."""

    class _cpx(C.Structure):
        _fields_ = [("r", C.c_float), ("i", C.c_float)]

    @staticmethod
    def available():
        return (HERE / "_ref" / "libkissfft_ref.so").exists()

    def __init__(self, N):
        self.L = C.CDLL(str(HERE / "_ref" / "libkissfft_ref.so"))
        self.L.kiss_fftr_alloc.restype = vp
        self.L.kiss_fftr_alloc.argtypes = [C.c_int, C.c_int, vp, vp]
        self.N = N
        self.fwd = vp(self.L.kiss_fftr_alloc(N, 0, None, None))
        self.inv = vp(self.L.kiss_fftr_alloc(N, 1, None, None))

    def forward(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(self.N // 2 + 1, np.complex64)
        self.L.kiss_fftr(self.fwd, fptr(x), out.ctypes.data_as(vp))
        return out

    def backward(self, X):
        """saf_rfft_backward semantics: kiss_fftri then scale 1/N (saf_utility_fft.c:749-752)."""
        X = np.ascontiguousarray(X, np.complex64)
        out = np.zeros(self.N, np.float32)
        self.L.kiss_fftri(self.inv, X.ctypes.data_as(vp), fptr(out))
        return out * np.float32(1.0 / self.N)


# ------------------------------------------------------------------ HRIR processing / binauraliser
def estimateITDs(hrirs, fs):
    hrirs = np.ascontiguousarray(hrirs, np.float32)
    N, _, L = hrirs.shape
    out = np.zeros(N, np.float32)
    lib().orc_estimateITDs(fptr(hrirs), N, L, fs, fptr(out))
    return out


def getVoronoiWeights(dirs_deg):
    d = np.ascontiguousarray(dirs_deg, np.float32)
    out = np.zeros(d.shape[0], np.float32)
    lib().orc_getVoronoiWeights(fptr(d), d.shape[0], fptr(out))
    return out


def diffuseFieldEqualiseHRTFs(hrtfs, weights=None):
    """hrtfs [nBands][2][N] complex64 -> equalised copy."""
    h = np.ascontiguousarray(hrtfs, np.complex64).copy()
    nB, _, N = h.shape
    w = None if weights is None else np.ascontiguousarray(weights, np.float32)
    lib().orc_diffuseFieldEqualiseHRTFs(N, nB, fptr(w) if w is not None else None, h.ctypes.data_as(vp))
    return h


class Binauraliser:
    def __init__(self, frameSize=128, maxSources=64):
        self.h = vp()
        self.F = frameSize
        self.maxSources = maxSources
        L = lib()
        L.orc_binauraliser_create(C.byref(self.h), frameSize, maxSources)
        for n in ("getITDs", "getWeights"):
            getattr(L, "orc_binauraliser_" + n).restype = c_f
        for n in ("getHRTFfb", "getHRTFinterp"):
            getattr(L, "orc_binauraliser_" + n).restype = vp

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_binauraliser_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def setHRIRs(self, hrirs, dirs_deg, fs):
        hrirs = np.ascontiguousarray(hrirs, np.float32); d = np.ascontiguousarray(dirs_deg, np.float32)
        lib().orc_binauraliser_setHRIRs(self.h, fptr(hrirs), fptr(d), hrirs.shape[0], hrirs.shape[2], fs)

    def process(self, x, nOut=2, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.zeros((nOut, self.F), np.float32)
        lib().orc_binauraliser_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], nOut, ns)
        return y

    def itds(self):
        n = lib().orc_binauraliser_getNDirs(self.h)
        return np.ctypeslib.as_array(lib().orc_binauraliser_getITDs(self.h), shape=(n,)).copy()

    def weights(self):
        n = lib().orc_binauraliser_getNDirs(self.h)
        return np.ctypeslib.as_array(lib().orc_binauraliser_getWeights(self.h), shape=(n,)).copy()

    def hrtf_fb(self):
        n = lib().orc_binauraliser_getNDirs(self.h)
        p = C.cast(lib().orc_binauraliser_getHRTFfb(self.h), C.POINTER(C.c_float))
        return np.ctypeslib.as_array(p, shape=(133, 2, n, 2)).copy().view(np.complex64)[..., 0]

    def hrtf_interp(self, nSrc):
        p = C.cast(lib().orc_binauraliser_getHRTFinterp(self.h), C.POINTER(C.c_float))
        return np.ctypeslib.as_array(p, shape=(self.maxSources, 133, 2, 2)).copy().view(np.complex64)[:nSrc, ..., 0]

    def __del__(self):
        if self.h:
            lib().orc_binauraliser_destroy(C.byref(self.h))


class BinauraliserNF(Binauraliser):
    """binauraliser_nf (examples/src/binauraliser_nf): a Binauraliser whose sources carry distances."""

    def __init__(self, frameSize=128, maxSources=64):
        self.h = vp()
        self.F = frameSize
        self.maxSources = maxSources
        L = lib()
        L.orc_binauraliserNF_create(C.byref(self.h), frameSize, maxSources)
        for n in ("getITDs", "getWeights"):
            getattr(L, "orc_binauraliser_" + n).restype = c_f
        for n in ("getHRTFfb", "getHRTFinterp"):
            getattr(L, "orc_binauraliser_" + n).restype = vp
        for n in ("getSourceDist_m", "getFarfieldThresh_m", "getFarfieldHeadroom", "getNearfieldLimit_m"):
            getattr(L, "orc_binauraliserNF_" + n).restype = C.c_float
        for n in ("getDVFmags", "getDVFphases"):
            getattr(L, "orc_binauraliserNF_" + n).restype = c_f

    def process(self, x, nOut=2, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.zeros((nOut, self.F), np.float32)
        lib().orc_binauraliserNF_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], nOut, ns)
        return y

    def setSourceDist_m(self, i, d):
        lib().orc_binauraliserNF_setSourceDist_m(self.h, i, C.c_float(d))

    def getSourceDist_m(self, i):
        return lib().orc_binauraliserNF_getSourceDist_m(self.h, i)

    def getFarfieldThresh_m(self):
        return lib().orc_binauraliserNF_getFarfieldThresh_m(self.h)

    def getFarfieldHeadroom(self):
        return lib().orc_binauraliserNF_getFarfieldHeadroom(self.h)

    def getNearfieldLimit_m(self):
        return lib().orc_binauraliserNF_getNearfieldLimit_m(self.h)

    def dvf(self, nSrc):
        """(magnitudes, phases) [nSrc][2][133] of the per-source, per-ear DVF responses."""
        m = np.ctypeslib.as_array(lib().orc_binauraliserNF_getDVFmags(self.h), shape=(self.maxSources, 2, 133)).copy()
        ph = np.ctypeslib.as_array(lib().orc_binauraliserNF_getDVFphases(self.h), shape=(self.maxSources, 2, 133)).copy()
        return m[:nSrc], ph[:nSrc]


# ------------------------------------------------------------------ DVF near-field filters
def calcDVFShelfParams(i, rho):
    g0, gi, fc = C.c_float(), C.c_float(), C.c_float()
    lib().orc_calcDVFShelfParams(int(i), C.c_float(rho), C.byref(g0), C.byref(gi), C.byref(fc))
    return g0.value, gi.value, fc.value


def interpDVFShelfParams(theta, rho):
    g0, gi, fc = C.c_float(), C.c_float(), C.c_float()
    lib().orc_interpDVFShelfParams(C.c_float(theta), C.c_float(rho), C.byref(g0), C.byref(gi), C.byref(fc))
    return g0.value, gi.value, fc.value


def dvfShelfCoeffs(g0, gInf, fc, fs):
    b0, b1, a1 = C.c_float(), C.c_float(), C.c_float()
    lib().orc_dvfShelfCoeffs(C.c_float(g0), C.c_float(gInf), C.c_float(fc), C.c_float(fs), C.byref(b0), C.byref(b1), C.byref(a1))
    return b0.value, b1.value, a1.value


def calcDVFCoeffs(alpha, rho, fs):
    b = np.zeros(2, np.float32); a = np.ones(2, np.float32)
    lib().orc_calcDVFCoeffs(C.c_float(alpha), C.c_float(rho), C.c_float(fs), fptr(b), fptr(a))
    return b, a


def doaToIpsiInteraural(azi, elev):
    al = np.zeros(2, np.float32); be = np.zeros(2, np.float32)
    lib().orc_doaToIpsiInteraural(C.c_float(azi), C.c_float(elev), fptr(al), fptr(be))
    return al, be


def evalIIRTransferFunctionf(b, a, freqs, fs, mag2dB=0):
    b = np.ascontiguousarray(b, np.float32); a = np.ascontiguousarray(a, np.float32); f = np.ascontiguousarray(freqs, np.float32)
    mag = np.zeros(f.size, np.float32); ph = np.zeros(f.size, np.float32)
    lib().orc_evalIIRTransferFunctionf(fptr(b), fptr(a), b.size, fptr(f), f.size, C.c_float(fs), mag2dB, fptr(mag), fptr(ph))
    return mag, ph


def _cx(a):
    return np.ascontiguousarray(a, np.complex64)


def generateMVDRmap(order, Cx, Y_grid, regPar=8.0, weights=False):
    Cx = _cx(Cx); Y = np.ascontiguousarray(Y_grid, np.float32); G = Y.shape[1]
    pm = np.zeros(G, np.float32); w = np.zeros(((order + 1) ** 2, G), np.complex64)
    lib().orc_generateMVDRmap(order, Cx.ctypes.data_as(vp), fptr(Y), G, C.c_float(regPar), fptr(pm), w.ctypes.data_as(vp))
    return (pm, w) if weights else pm


def generateCroPaCLCMVmap(order, Cx, Y_grid, regPar=8.0, lam=0.0):
    Cx = _cx(Cx); Y = np.ascontiguousarray(Y_grid, np.float32); G = Y.shape[1]
    pm = np.zeros(G, np.float32)
    lib().orc_generateCroPaCLCMVmap(order, Cx.ctypes.data_as(vp), fptr(Y), G, C.c_float(regPar), C.c_float(lam), fptr(pm))
    return pm


def generateMUSICmap(order, Cx, Y_grid, nSources, logScale=0):
    Cx = _cx(Cx); Y = np.ascontiguousarray(Y_grid, np.float32); G = Y.shape[1]
    pm = np.zeros(G, np.float32)
    lib().orc_generateMUSICmap(order, Cx.ctypes.data_as(vp), fptr(Y), nSources, G, logScale, fptr(pm))
    return pm


def generateMinNormMap(order, Cx, Y_grid, nSources, logScale=0):
    Cx = _cx(Cx); Y = np.ascontiguousarray(Y_grid, np.float32); G = Y.shape[1]
    pm = np.zeros(G, np.float32)
    lib().orc_generateMinNormMap(order, Cx.ctypes.data_as(vp), fptr(Y), nSources, G, logScale, fptr(pm))
    return pm


def herm_eig(A):
    A = _cx(A); n = A.shape[0]
    e = np.zeros(n); vr = np.zeros((n, n)); vi = np.zeros((n, n))
    dp = C.POINTER(C.c_double)
    lib().orc_herm_eig(n, A.ctypes.data_as(vp), e.ctypes.data_as(dp), vr.ctypes.data_as(dp), vi.ctypes.data_as(dp))
    return e, vr + 1j * vi


# ------------------------------------------------------------------ binaural Ambisonic decoders / ambi_bin
def getSHrotMtxReal(R, order):
    R = np.ascontiguousarray(R, np.float32).reshape(9)
    n = (order + 1) ** 2
    out = np.zeros((n, n), np.float32)
    lib().orc_getSHrotMtxReal(fptr(R), fptr(out), order)
    return out


def yawPitchRoll2Rzyx(yaw, pitch, roll, rpy=0):
    R = np.zeros(9, np.float32)
    lib().orc_yawPitchRoll2Rzyx(C.c_float(yaw), C.c_float(pitch), C.c_float(roll), rpy, fptr(R))
    return R.reshape(3, 3)


def getBinauralAmbiDecoderMtx(hrtfs, dirs_deg, method, order, freqVector=None, itd_s=None, weights=None, diffMatching=0, maxRE=0):
    H = np.ascontiguousarray(hrtfs, np.complex64); nBands, _, N = H.shape
    d = np.ascontiguousarray(dirs_deg, np.float32)
    fv = np.ascontiguousarray(freqVector if freqVector is not None else np.zeros(nBands), np.float32)
    it = np.ascontiguousarray(itd_s if itd_s is not None else np.zeros(N), np.float32)
    w = np.ascontiguousarray(weights, np.float32) if weights is not None else None
    out = np.zeros((nBands, 2, (order + 1) ** 2), np.complex64)
    lib().orc_getBinauralAmbiDecoderMtx(H.ctypes.data_as(vp), fptr(d), N, nBands, method, order, fptr(fv), fptr(it), fptr(w) if w is not None else None,
                                        diffMatching, maxRE, out.ctypes.data_as(vp))
    return out


def getBinauralAmbiDecoderFilters(hrtfs, dirs_deg, fftSize, fs, method, order, itd_s=None, weights=None, diffMatching=0, maxRE=0):
    H = np.ascontiguousarray(hrtfs, np.complex64); nBins, _, N = H.shape
    d = np.ascontiguousarray(dirs_deg, np.float32)
    it = np.ascontiguousarray(itd_s if itd_s is not None else np.zeros(N), np.float32)
    w = np.ascontiguousarray(weights, np.float32) if weights is not None else None
    out = np.zeros((2, (order + 1) ** 2, fftSize), np.float32)
    lib().orc_getBinauralAmbiDecoderFilters(H.ctypes.data_as(vp), fptr(d), N, fftSize, C.c_float(fs), method, order, fptr(it), fptr(w) if w is not None else None,
                                            diffMatching, maxRE, fptr(out))
    return out


def truncationEQ(w_n, order_truncated, order_target, kr, softThreshold):
    w = np.ascontiguousarray(w_n, np.float32); k = np.ascontiguousarray(kr, np.float64); g = np.zeros(k.shape[0], np.float32)
    lib().orc_truncationEQ(fptr(w), order_truncated, order_target, k.ctypes.data_as(C.POINTER(C.c_double)), k.shape[0], C.c_float(softThreshold), fptr(g))
    return g


class AmbiBin:
    def __init__(self, frameSize=128):
        self.h = vp()
        self.F = frameSize
        lib().orc_ambi_bin_create(C.byref(self.h), frameSize)
        lib().orc_ambi_bin_getDecMtx.restype = vp

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_ambi_bin_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def setHRIRs(self, hrirs, dirs_deg, fs):
        hrirs = np.ascontiguousarray(hrirs, np.float32); d = np.ascontiguousarray(dirs_deg, np.float32)
        lib().orc_ambi_bin_setHRIRs(self.h, fptr(hrirs), fptr(d), hrirs.shape[0], hrirs.shape[2], fs)

    def process(self, x, nOut=2, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.zeros((nOut, self.F), np.float32)
        lib().orc_ambi_bin_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], nOut, ns)
        return y

    def decMtx(self, nSH):
        p = C.cast(lib().orc_ambi_bin_getDecMtx(self.h), C.POINTER(C.c_float))
        return np.ctypeslib.as_array(p, shape=(133, 2, 64, 2)).copy().view(np.complex64)[..., 0][:, :, :nSH]

    def __del__(self):
        if self.h:
            lib().orc_ambi_bin_destroy(C.byref(self.h))


# ------------------------------------------------------------------ panner
class Panner:
    def __init__(self, frameSize=128):
        self.h = vp()
        self.F = frameSize
        L = lib()
        L.orc_panner_create(C.byref(self.h), frameSize)
        L.orc_panner_getGains.restype = c_f
        L.orc_panner_getPvalue.restype = c_f

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_panner_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def process(self, x, nOut, nSamples=None):
        x = np.ascontiguousarray(x, np.float32)
        ns = x.shape[1] if nSamples is None else nSamples
        y = np.zeros((nOut, self.F), np.float32)
        lib().orc_panner_process(self.h, _chan_ptrs(x), _chan_ptrs(y), x.shape[0], nOut, ns)
        return y

    def gains(self):
        """G_src [133][64][64] (band, source, loudspeaker)"""
        return np.ctypeslib.as_array(lib().orc_panner_getGains(self.h), shape=(133, 64, 64)).copy()

    def pvalues(self):
        return np.ctypeslib.as_array(lib().orc_panner_getPvalue(self.h), shape=(133,)).copy()

    def __del__(self):
        if self.h:
            lib().orc_panner_destroy(C.byref(self.h))


def getPvalues(DTT, freq):
    f = np.ascontiguousarray(freq, np.float32)
    out = np.zeros(f.shape[0], np.float32)
    lib().orc_getPvalues(C.c_float(DTT), fptr(f), f.shape[0], fptr(out))
    return out


# ------------------------------------------------------------------ powermap (PWD)
class Powermap:
    def __init__(self, frameSize=1024):
        self.h = vp()
        self.F = frameSize
        L = lib()
        L.orc_powermap_create(C.byref(self.h), frameSize)
        L.orc_powermap_getCx.restype = vp
        L.orc_powermap_getRawPmap.restype = c_f

    def __getattr__(self, name):
        fn = getattr(lib(), "orc_powermap_" + name)
        return lambda *a: fn(self.h, *[C.c_float(x) if isinstance(x, float) else x for x in a])

    def analysis(self, x, isPlaying=1):
        x = np.ascontiguousarray(x, np.float32)
        lib().orc_powermap_analysis(self.h, _chan_ptrs(x), x.shape[0], x.shape[1], isPlaying)

    def Cx(self, nSH):
        p = C.cast(lib().orc_powermap_getCx(self.h), C.POINTER(C.c_float))
        a = np.ctypeslib.as_array(p, shape=(133, 64 * 64, 2)).copy().view(np.complex64)[..., 0]
        return a[:, :nSH * nSH].reshape(133, nSH, nSH)

    def rawPmap(self):
        n = lib().orc_powermap_getGridNDirs(self.h)
        return np.ctypeslib.as_array(lib().orc_powermap_getRawPmap(self.h), shape=(n,)).copy()

    def getPmap(self):
        gd, pm, n = c_f(), c_f(), C.c_int()
        ready = lib().orc_powermap_getPmap(self.h, C.byref(gd), C.byref(pm), C.byref(n))
        if not ready:
            return None
        return np.ctypeslib.as_array(pm, shape=(n.value,)).copy()

    def __del__(self):
        if self.h:
            lib().orc_powermap_destroy(C.byref(self.h))
