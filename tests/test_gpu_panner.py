"""panner (frequency-dependent VBAP) on the GPU (libsaf_hip.so through its C-ABI) against the CPU oracle — needs an MI355X.

The reference holds no test for the panner or for saf_vbap, so parity is "unpinned" by reference-side data
(DESIGN.md §2): the oracle restates panner.c / panner_internal.c and both sides are compared on seeded inputs.
Tolerance: 1e-5 relative RMS on the loudspeaker signals (north star); table rows (indices) must agree exactly.
"""
import numpy as np
import pytest

from util import frames, relrms, maxabs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def pair(saf, orc, F, ls_preset, src_preset=None, nS=None, dtt=None, spread=None):
    g, o = saf.Panner(F), orc.Panner(F)
    for b in (g, o):
        b.setOutputConfigPreset(ls_preset)
        if src_preset is not None:
            b.setInputConfigPreset(src_preset)
        if nS is not None:
            b.setNumSources(nS)
        if spread is not None:
            b.setSpread(float(spread))
        b.initCodec()
        b.init(48000)
        if dtt is not None:
            b.setDTT(float(dtt))
            b.initCodec()
    return g, o


def test_getPvalues_vs_oracle(saf, orc):
    f = np.linspace(0, 24000, 133).astype(np.float32)
    for dtt in (0.0, 0.5, 1.0):
        assert maxabs(saf.getPvalues(dtt, f), orc.getPvalues(dtt, f)) == 0.0
    assert np.all(orc.getPvalues(0.0, f) == 2.0)


@pytest.mark.parametrize("ls_preset,nS,dtt,spread", [(29, 64, None, None), (21, 12, 1.0, None), (3, 5, 0.0, None), (28, 7, 0.3, 20.0)])
def test_panner_vs_oracle(saf, orc, ls_preset, nS, dtt, spread):
    """sources on a covering set, some moved between blocks, yaw/pitch/roll changed mid-stream; F = 512."""
    F = 512
    g, o = pair(saf, orc, F, ls_preset, src_preset=30, nS=nS, dtt=dtt, spread=spread)   # sources: SphCovering-64 directions
    nL = g.getNumLoudspeakers()
    assert nL == o.getNumLoudspeakers() and g.getNumSources() == nS == o.getNumSources()
    x = frames(77, nS, 10 * F)
    rng = np.random.default_rng(5)
    outs_g, outs_o = [], []
    for blk in range(10):
        if blk == 3:
            for b in (g, o):
                b.setYaw(35.0); b.setPitch(-20.0); b.setRoll(10.0)
        if blk in (5, 6):
            for s in rng.integers(0, nS, 3):
                az, el = float(rng.uniform(-180, 180)), float(rng.uniform(-90, 90))
                for b in (g, o):
                    b.setSourceAzi_deg(int(s), az); b.setSourceElev_deg(int(s), el)
        xb = np.ascontiguousarray(x[:, blk * F:(blk + 1) * F])
        outs_g.append(g.process(xb, nL)); outs_o.append(o.process(xb, nL))
        if blk in (0, 3, 6):
            Gg, Go = g.gains()[:, :nS, :nL], o.gains()[:, :nS, :nL]
            assert maxabs(Gg, Go) < 2e-6, blk            # same table rows, device powf within a few ulp
    yg, yo = np.concatenate(outs_g, 1), np.concatenate(outs_o, 1)
    assert np.isfinite(yg).all() and np.abs(yo).max() > 0.05
    assert relrms(yg, yo) < TOL


def test_panner_default_config_and_bad_block(saf, orc):
    """defaults of panner_create (one source at 0/0, stereo pair, F = 128); a wrong block size zero-fills the outputs"""
    g, o = saf.Panner(128), orc.Panner(128)
    for b in (g, o):
        b.initCodec(); b.init(48000)
    assert g.getNumSources() == 1 and g.getNumLoudspeakers() == 2 and g.getProcessingDelay() == 12 * 128
    x = frames(3, 1, 16 * 128)
    yg = np.concatenate([g.process(np.ascontiguousarray(x[:, i * 128:(i + 1) * 128]), 4) for i in range(16)], 1)
    yo = np.concatenate([o.process(np.ascontiguousarray(x[:, i * 128:(i + 1) * 128]), 4) for i in range(16)], 1)
    assert relrms(yg[:2], yo[:2]) < TOL and np.all(yg[2:] == 0)
    assert np.all(g.process(np.ones((1, 64), np.float32), 2, nSamples=64) == 0)


def test_panner_channel_change_and_device_entry(saf, orc):
    """fewer sources after a re-init (stale gain columns must not leak), then 6 blocks in one device call"""
    import torch
    F = 256
    g, o = pair(saf, orc, F, 26, src_preset=30, nS=20)
    x = frames(9, 20, 12 * F)
    for blk in range(3):
        xb = np.ascontiguousarray(x[:, blk * F:(blk + 1) * F])
        assert relrms(g.process(xb, 16), o.process(xb, 16)) < TOL or blk == 0
    for b in (g, o):
        b.setNumSources(6); b.initCodec()
    for blk in range(3, 6):
        xb = np.ascontiguousarray(x[:6, blk * F:(blk + 1) * F])
        yg, yo = g.process(xb, 16), o.process(xb, 16)
    assert relrms(yg, yo) < TOL
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    nF = 6
    xin = np.ascontiguousarray(x[:6, 6 * F:12 * F])                          # [ch][nF * F]
    d_in = torch.from_numpy(xin).cuda(); d_out = torch.zeros(16, nF * F, device="cuda")
    g.process_dev(d_in.data_ptr(), (F, nF * F), 6, d_out.data_ptr(), (F, nF * F), nF)
    torch.cuda.synchronize()
    yo = np.concatenate([o.process(np.ascontiguousarray(xin[:, i * F:(i + 1) * F]), 16) for i in range(nF)], 1)
    assert relrms(d_out.cpu().numpy(), yo) < TOL
    saf.set_stream(None)
