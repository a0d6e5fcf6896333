"""beamformer (examples/include/beamformer.h; static cardioid / hyper-cardioid / max-EV beams over an Ambisonic scene, one
block of latency, linear cross-fade when a beam moved) on the GPU build against the CPU oracle — needs an MI355X.

The reference holds no test for this operator ("parity unpinned" by reference-side data): the oracle restates
beamformer.c and the SH helpers literally (matrix form of complex2realCoeffs), the library derives the real coefficients
per (n, m); the closed-form patterns are checked in tests/test_oracle_cpu.py.  Tolerance 2e-6 absolute."""
import numpy as np
import pytest

from util import frames, maxabs

pytestmark = pytest.mark.gpu


def test_beam_weights_vs_oracle(saf, orc):
    for N in (1, 2, 5, 7):
        for kind in (1, 2, 3):
            assert maxabs(saf.beamWeights(kind, N), orc.beamWeights(kind, N)) < 2e-6
        c = orc.beamWeights(2, N)
        for th, ph in ((1.1, 0.7), (0.0, 2.0), (np.pi, -1.0), (np.pi / 2, np.pi), (2.4, -3.0)):
            assert maxabs(saf.rotateAxisCoeffsReal(N, c, th, ph), orc.rotateAxisCoeffsReal(N, c, th, ph)) < 2e-6, (N, th, ph)


@pytest.mark.parametrize("order,F,nBeams,kind,norm,chOrder", [(7, 128, 64, 2, 2, 1), (1, 64, 3, 1, 3, 2), (3, 256, 10, 3, 1, 1), (5, 100, 33, 1, 2, 1)])
def test_beamformer_scenarios_vs_oracle(saf, orc, order, F, nBeams, kind, norm, chOrder):
    """beams moved mid-stream, beam type and count changed, fewer inputs / outputs than needed, wrong block size -> zeros"""
    nSH = (order + 1) ** 2
    g, o = saf.Beamformer(F), orc.Beamformer(F)
    for b in (g, o):
        b.init(48000); b.setBeamOrder(order); b.setNumBeams(nBeams); b.setBeamType(kind); b.setNormType(norm); b.setChOrder(chOrder)
    x = frames(order * 7 + nBeams, nSH, 12 * F)
    rng = np.random.default_rng(order)
    for blk in range(12):
        if blk in (3, 4):
            idx = rng.integers(0, nBeams, 3)
            vals = [(float(rng.uniform(-180, 180)), float(rng.uniform(-90, 90))) for _ in idx]
            for b in (g, o):
                for i, (az, el) in zip(idx, vals):
                    b.setBeamAzi_deg(int(i), az); b.setBeamElev_deg(int(i), el)
        if blk == 7:
            for b in (g, o):
                b.setBeamType(1 + kind % 3)
        if blk == 9 and nBeams > 2:
            for b in (g, o):
                b.setNumBeams(nBeams - 1)
        nIn = nSH if blk != 5 else max(1, nSH - 1)
        nOut = nBeams + 1 if blk != 6 else max(1, nBeams - 1)
        xb = np.ascontiguousarray(x[:nIn, blk * F:(blk + 1) * F])
        yg, yo = g.process(xb, nOut), o.process(xb, nOut)
        assert maxabs(yg, yo) < 2e-6 * max(1.0, float(np.abs(yo).max())), blk
    assert np.abs(yo).max() > 0.05
    assert not g.process(np.ones((nSH, F // 2), np.float32), nBeams, nSamples=F // 2).any()


def test_beamformer_device_entry(saf, orc):
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    order, F, nF, nBeams = 4, 128, 6, 12
    nSH = 25
    g, o = saf.Beamformer(F), orc.Beamformer(F)
    for b in (g, o):
        b.init(48000); b.setBeamOrder(order); b.setNumBeams(nBeams); b.setNormType(1)
    x = frames(4, nSH, 2 * nF * F)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros(nBeams, 2 * nF * F, device="cuda")
    yo = []
    for call in range(2):
        if call == 1:
            for b in (g, o):
                b.setBeamAzi_deg(2, 123.0); b.setBeamElev_deg(7, -45.0)
        g.process_dev(d_in[:, call * nF * F:].data_ptr(), (F, 2 * nF * F), nSH, d_out[:, call * nF * F:].data_ptr(), (F, 2 * nF * F), nBeams, nF)
        for i in range(call * nF, (call + 1) * nF):
            yo.append(o.process(np.ascontiguousarray(x[:, i * F:(i + 1) * F]), nBeams))
    torch.cuda.synchronize()
    assert maxabs(d_out.cpu().numpy(), np.concatenate(yo, 1)) < 3e-6
    saf.set_stream(None)
