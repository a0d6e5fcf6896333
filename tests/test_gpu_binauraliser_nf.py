"""binauraliser_nf on the GPU (libsaf_hip.so through its C-ABI) against the CPU oracle — needs an MI355X.

The reference has no test of the example itself; its DVF utility tests (test__dvf_*, test__evalIIRTransferFunction:
test/src/test__utilities_module.c:1114-1190, 1304-1440) pin the oracle's filters in tests/test_oracle_cpu.py and the
library's host functions in tests/test_lib_cpu.py.  Here the rendered ears are compared on the synthetic HRIR set of
tests/util.py::synth_hrirs.  Tolerance: 1e-5 relative RMS (north star).
"""
import numpy as np
import pytest

from util import frames, relrms, maxabs, synth_hrirs

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def hrirs():
    return synth_hrirs()


def setup_pair(saf, orc, hrirs, F, nS, maxS=64, mode=1):
    h, d = hrirs
    g, o = saf.BinauraliserNF(F, maxS), orc.BinauraliserNF(F, maxS)
    for b in (g, o):
        b.setHRIRs(h, d, 48000)
        b.init(48000)
        b.setNumSources(nS)
        b.setInterpMode(mode)
        b.initCodec()
    return g, o


def test_nf_constants_and_setters(saf, orc, hrirs):
    g, o = setup_pair(saf, orc, hrirs, 128, 4)
    assert g.getFarfieldThresh_m() == o.getFarfieldThresh_m() == pytest.approx(0.09096 * 34, rel=1e-6)
    assert g.getFarfieldHeadroom() == o.getFarfieldHeadroom() and g.getNearfieldLimit_m() == o.getNearfieldLimit_m() == pytest.approx(0.15)
    for b in (g, o):
        assert b.getSourceDist_m(2) == pytest.approx(0.09096 * 34 * 1.05, rel=1e-6)      # far field by default
        b.setSourceDist_m(2, 0.05)                                                      # clamped to the near-field limit
        assert b.getSourceDist_m(2) == pytest.approx(0.15)
    g.setSourceDist_m(1, 0.7)
    g.setInputConfigPreset(1)                                                           # presets put every source back in the far field
    assert g.getSourceDist_m(1) == pytest.approx(0.09096 * 34 * 1.05, rel=1e-6)


@pytest.mark.parametrize("mode", [1, 2])
def test_binauraliser_nf_vs_oracle(saf, orc, hrirs, mode):
    """24 sources at 0.15 .. 3.3 m (both sides of the far-field threshold), F = 512: distances and directions changing between
    blocks, head rotation; the filters the MAC applies are compared first, then the ears."""
    F, nS = 512, 24
    g, o = setup_pair(saf, orc, hrirs, F, nS, mode=mode)
    rng = np.random.default_rng(3)
    dirs = np.stack([rng.uniform(-180, 180, nS), rng.uniform(-85, 85, nS)], 1)
    dist = np.concatenate([np.geomspace(0.15, 3.0, nS - 4), [3.09, 3.1, 3.3, 0.16]])
    for b in (g, o):
        for s in range(nS):
            b.setSourceAzi_deg(s, float(dirs[s, 0])); b.setSourceElev_deg(s, float(dirs[s, 1])); b.setSourceDist_m(s, float(dist[s]))
        b.setSourceGain(2, 0.5)
    x = frames(41, nS, 10 * F)
    num = den = 0.0
    for f in range(10):
        if f == 3:
            for b in (g, o):
                b.setSourceDist_m(0, 1.2); b.setSourceDist_m(5, 5.0); b.setSourceDist_m(nS - 2, 0.4)      # near -> near, near -> far, far -> near
        if f == 5:
            for b in (g, o):
                b.setSourceAzi_deg(1, 91.0); b.setSourceElev_deg(9, -30.0)
        if f == 7:
            for b in (g, o):
                b.setEnableRotation(1); b.setYaw(-35.0); b.setPitch(12.0)
        blk = x[:, f * F:(f + 1) * F]
        yg, yo = g.process(blk), o.process(blk)
        if f in (0, 3, 7):
            m, ph = o.dvf(nS)
            near = np.array([o.getSourceDist_m(s) < o.getFarfieldThresh_m() for s in range(nS)])
            hi = o.hrtf_interp(nS)                                                      # [src][band][ear]
            scale = (m + 1j * ph).transpose(0, 2, 1)                                    # the reference's cmplxf(mag, phase)
            want = np.where(near[:, None, None], scale * hi, hi)
            assert relrms(g.hrtf_nf(nS), want) < 3e-6, f
        num += float(((yg - yo) ** 2).sum()); den += float((yo ** 2).sum())
    assert den > 0 and (num / den) ** 0.5 < TOL


def test_nf_far_sources_equal_plain_binauraliser(saf, orc, hrirs):
    """With every source beyond the threshold binauraliserNF_process is binauraliser_process (binauraliser_nf.c:341-347)."""
    h, d = hrirs
    F, nS = 128, 6
    g, _ = setup_pair(saf, orc, hrirs, F, nS)
    p = saf.Binauraliser(F, 64)
    p.setHRIRs(h, d, 48000); p.init(48000); p.setNumSources(nS); p.initCodec()
    for b in (g, p):
        for s in range(nS):
            b.setSourceAzi_deg(s, 50.0 * s - 120.0); b.setSourceElev_deg(s, 10.0 * s - 20.0)
    x = frames(5, nS, 16 * F)
    for f in range(16):
        blk = x[:, f * F:(f + 1) * F]
        assert maxabs(g.process(blk), p.process(blk)) == 0.0


def test_nf_device_entry_and_batch(saf, orc, hrirs):
    """saf_hip_binauraliserNF_process_dev and a batch of NF handles: 2 instances, 2 calls of 4 blocks, distances changed between
    the calls; each equals its own oracle run."""
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    F, nS, nI, nF = 256, 12, 2, 4
    pairs = [setup_pair(saf, orc, hrirs, F, nS) for _ in range(nI)]
    single = setup_pair(saf, orc, hrirs, F, nS)[0]
    for i, (g, o) in enumerate(pairs):
        for b in (g, o) + ((single,) if i == 0 else ()):
            for s in range(nS):
                b.setSourceAzi_deg(s, float((47 * s + 60 * i) % 360 - 180)); b.setSourceElev_deg(s, float((13 * s) % 120 - 60))
                b.setSourceDist_m(s, 0.2 + 0.3 * s + 0.05 * i)
    bt = saf.BinauraliserBatch([g for g, _ in pairs], nF)
    x = np.stack([frames(70 + i, nS, 2 * nF * F) for i in range(nI)])
    d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros(nI, 2, 2 * nF * F, device="cuda"); d_one = torch.zeros(2, 2 * nF * F, device="cuda")
    for call in range(2):
        if call == 1:
            for b in pairs[1] + pairs[0] + (single,):
                b.setSourceDist_m(3, 0.25); b.setSourceDist_m(0, 4.0)
        off = call * nF * F * 4
        bt.process_ptr(d_in.data_ptr() + off, (nS * 2 * nF * F, F, 2 * nF * F), nS, d_out.data_ptr() + off, (2 * 2 * nF * F, F, 2 * nF * F), nF)
        single.process_dev(d_in[0].data_ptr() + off, (F, 2 * nF * F), nS, d_one.data_ptr() + off, (F, 2 * nF * F), nF)
        torch.cuda.synchronize()
        for i, (_, o) in enumerate(pairs):
            yo = np.concatenate([o.process(np.ascontiguousarray(x[i][:, (call * nF + f) * F:(call * nF + f + 1) * F])) for f in range(nF)], 1)
            yg = d_out[i][:, call * nF * F:(call + 1) * nF * F].cpu().numpy()
            assert relrms(yg, yo) < TOL or (call == 0 and np.abs(yo).max() < 1e-3), (call, i)
            if i == 0:
                assert relrms(d_one[:, call * nF * F:(call + 1) * nF * F].cpu().numpy(), yo) < TOL or (call == 0 and np.abs(yo).max() < 1e-3)
    saf.set_stream(None)
