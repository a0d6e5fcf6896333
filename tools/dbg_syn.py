import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from spatial_audio_framework_amd import api as saf
from oracle import oracle as orc
from util import relrms
rng = np.random.default_rng(0)
for nout, T, hyb in [(4, 4, 1), (5, 4, 1), (4, 16, 1), (4, 8, 1), (4, 12, 1), (2, 4, 0), (1, 1, 1)]:
    g, o = saf.AfSTFT(1, nout, 128, 0, hyb), orc.AfSTFT(1, nout, 128, 0, hyb)
    nb = 133 if hyb else 129
    errs = []
    for fr in range(4):
        Y = (rng.normal(size=(nb, nout, T)) + 1j * rng.normal(size=(nb, nout, T))).astype(np.complex64)
        a, b = g.backward(Y), o.backward(Y)
        errs.append([round(relrms(a[c], b[c]), 8) for c in range(nout)])
    print(nout, T, hyb, errs)
