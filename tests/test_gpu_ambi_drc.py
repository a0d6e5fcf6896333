"""ambi_drc (examples/include/ambi_drc.h; frequency-dependent dynamic range compression of an Ambisonic scene driven by the
omni channel: afSTFT -> per-band gain computer with attack / release smoothing -> afSTFT^-1) on the GPU build against the CPU
oracle — needs an MI355X.

The reference holds no test for this operator ("parity unpinned" by reference-side data).  Tolerance: 1e-5 relative RMS on
the output (north star), 1e-5 relative on the gain factors (device log10f / powf against the host's)."""
import numpy as np
import pytest

from util import frames, relrms, maxabs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def mk(cls, F, order, thr, ratio, knee, att, rel, inG, outG):
    d = cls(F)
    d.setInputPreset(order); d.setThreshold(thr); d.setRatio(ratio); d.setKnee(knee); d.setAttack(att); d.setRelease(rel)
    d.setInGain(inG); d.setOutGain(outG)
    d.init(48000)
    return d


@pytest.mark.parametrize("F,order,thr,ratio,knee,att,rel,inG,outG", [(128, 1, -30.0, 8.0, 0.0, 50.0, 100.0, 0.0, 0.0), (512, 3, -40.0, 4.0, 6.0, 10.0, 300.0, 6.0, -3.0),
                                                                     (256, 7, -50.0, 20.0, 10.0, 100.0, 50.0, 12.0, 5.0)])
def test_ambi_drc_vs_oracle(saf, orc, F, order, thr, ratio, knee, att, rel, inG, outG):
    """level steps (quiet / loud / quiet) so that the compressor attacks and releases; parameters changed mid-stream"""
    nSH = (order + 1) ** 2
    g, o = mk(saf.AmbiDrc, F, order, thr, ratio, knee, att, rel, inG, outG), mk(orc.AmbiDrc, F, order, thr, ratio, knee, att, rel, inG, outG)
    assert g.getNSHrequired() == nSH == o.getNSHrequired() and saf.load().ambi_drc_getProcessingDelay() == 1536
    nB = 40 * 128 // F * 2
    x = frames(order + 50, nSH, nB * F)
    env = np.ones(nB * F, np.float32); env[: nB * F // 4] = 0.01; env[3 * nB * F // 4:] = 0.02
    x = (x * env).astype(np.float32)
    num = den = 0.0
    for b in range(nB):
        if b == nB // 2:
            for d in (g, o):
                d.setRatio(2.0); d.setThreshold(thr / 2)
        xb = np.ascontiguousarray(x[:, b * F:(b + 1) * F])
        yg, yo = g.process(xb), o.process(xb)
        num += float(((yg - yo) ** 2).sum()); den += float((yo ** 2).sum())
        if b in (nB // 3, nB - 1):                          # the gain factors of the block just processed, via the display ring
            G, w = g.gainTF()
            T = F // 128
            last = np.stack([G[:, (w - T + t) % 3000] for t in range(T)], 1)
            Go = o.lastGains()
            assert maxabs(last, Go) < 1e-5 * max(1.0, float(np.abs(Go).max())), b
            assert Go.min() < 0.9 if b == nB // 3 else True  # the compressor is at work in the loud part
    assert den > 1e-3 and (num / den) ** 0.5 < TOL
    assert not g.process(np.ones((nSH, F // 2), np.float32), nSamples=F // 2).any()


def test_ambi_drc_order_change_and_device_entry(saf, orc):
    """the order changed after init (filterbank channel change + cleared state), fewer channels fed than required, then
    several blocks per call on device-resident signals"""
    import torch
    F = 128
    g, o = mk(saf.AmbiDrc, F, 2, -35.0, 6.0, 3.0, 20.0, 80.0, 3.0, 0.0), mk(orc.AmbiDrc, F, 2, -35.0, 6.0, 3.0, 20.0, 80.0, 3.0, 0.0)
    x = frames(77, 16, 60 * F)
    for b in range(14):
        xb = np.ascontiguousarray(x[:9, b * F:(b + 1) * F])
        yg, yo = g.process(xb), o.process(xb)
    assert relrms(yg, yo) < TOL
    for d in (g, o):
        d.setInputPreset(3)
    num = den = 0.0
    for b in range(14, 40):
        xb = np.ascontiguousarray(x[:(16 if b % 5 else 12), b * F:(b + 1) * F])
        yg, yo = g.process(xb), o.process(xb)
        num += float(((yg - yo) ** 2).sum()); den += float((yo ** 2).sum())
    assert den > 1e-3 and (num / den) ** 0.5 < TOL
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    nF = 10
    xin = np.ascontiguousarray(x[:, 40 * F:(40 + nF) * F])
    d_in = torch.from_numpy(xin).cuda(); d_out = torch.zeros(16, nF * F, device="cuda")
    g.process_dev(d_in.data_ptr(), (F, nF * F), 16, d_out.data_ptr(), (F, nF * F), nF)
    torch.cuda.synchronize()
    yo = np.concatenate([o.process(np.ascontiguousarray(xin[:, i * F:(i + 1) * F])) for i in range(nF)], 1)
    assert relrms(d_out.cpu().numpy(), yo) < TOL
    saf.set_stream(None)
