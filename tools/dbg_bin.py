import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from spatial_audio_framework_amd import api as saf
from oracle import oracle as orc
from util import relrms, maxabs, synth_hrirs, frames
h, d = synth_hrirs()
for eq in (1, 0):
    g, o = saf.Binauraliser(128, 256), orc.Binauraliser(128, 256)
    for b in (g, o):
        b.setHRIRs(h, d, 48000); b.init(48000); b.setEnableHRIRsDiffuseEQ(eq); b.setNumSources(256); b.initCodec()
    if eq:
        wg, wo = g.weights(), o.weights()
        print('weights maxabs', maxabs(wg, wo), 'rel', relrms(wg, wo))
    A, B = g.hrtf_fb(), o.hrtf_fb()
    print('eq', eq, 'hrtf_fb relrms', relrms(A, B), 'per-band max', max(relrms(A[b], B[b]) for b in range(133)))
    print('   mag relrms', relrms(np.abs(A), np.abs(B)), 'phase-ish', relrms(A / np.maximum(np.abs(A), 1e-12), B / np.maximum(np.abs(B), 1e-12)))
    rng = np.random.default_rng(9)
    dirs = np.stack([rng.uniform(-180, 180, 256), rng.uniform(-80, 80, 256)], 1).astype(np.float32)
    for b in (g, o):
        for s in range(256):
            b.setSourceAzi_deg(s, float(dirs[s, 0])); b.setSourceElev_deg(s, float(dirs[s, 1]))
    x = frames(77, 256, 8 * 128)
    num = den = 0
    for f in range(8):
        yg, yo = g.process(x[:, f*128:(f+1)*128]), o.process(x[:, f*128:(f+1)*128])
        num += ((yg - yo) ** 2).sum(); den += (yo ** 2).sum()
    print('   out relrms', (num / den) ** 0.5, 'interp', relrms(g.hrtf_interp(256), o.hrtf_interp(256)))
# FIR->FB alone
A, B = saf.afSTFT_FIRtoFilterbankCoeffs(h[:50]), orc.FIRtoFilterbankCoeffs(h[:50])
print('FIR2FB relrms', relrms(A, B), relrms(np.abs(A), np.abs(B)))
