import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from spatial_audio_framework_amd import api as saf
from oracle import oracle as orc
from util import relrms, maxabs, synth_hrirs, frames
h, d = synth_hrirs()
dd = d.copy(); dd[dd[:, 0] > 180, 0] -= 360
w = orc.getVoronoiWeights(dd)
B = orc.FIRtoFilterbankCoeffs(h)           # [133][2][836]
Eg, Eo = saf.diffuseFieldEqualiseHRTFs(B, w), orc.diffuseFieldEqualiseHRTFs(B, w)
print('same input: EQ relrms', relrms(Eg, Eo))
ratio = np.abs(Eg[:, :, 0]) / np.abs(Eo[:, :, 0])
print('gain ratio - 1: max', np.abs(ratio - 1).max(), 'rms', np.sqrt(((ratio - 1) ** 2).mean()))
acc64 = (w[None, None, :].astype(np.float64) / (4 * np.pi) * np.abs(B.astype(np.complex128)) ** 2).sum(-1)
dg = np.abs(B[:, :, 0]) / np.abs(Eg[:, :, 0]); do = np.abs(B[:, :, 0]) / np.abs(Eo[:, :, 0])
print('vs float64 sqrt(acc): product', np.abs(dg / np.sqrt(acc64) - 1).max(), 'oracle', np.abs(do / np.sqrt(acc64) - 1).max())
