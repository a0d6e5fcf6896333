"""powermap (PWD mode) on the GPU against the CPU oracle — needs an MI355X.

The reference holds no test for powermap, so parity is pinned by the oracle restatement of
powermap.c:185-380 / saf_sh.c:1544-1584 and its closed-form checks (tests/test_oracle_cpu.py).
Tolerance: 1e-5 relative RMS on covariances and maps (north star).
"""
import numpy as np
import pytest

from util import frames, relrms, maxabs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def mk(cls, F, order, norm=2, chOrder=1, cov=0.3, avg=0.666):
    pm = cls(F)
    pm.setMasterOrder(order); pm.setPowermapMode(1)
    pm.init(48000.0); pm.initCodec()
    pm.setAnaOrderAllBands(order); pm.setNormType(norm); pm.setChOrder(chOrder)
    pm.setCovAvgCoeff(cov); pm.setPowermapAvgCoeff(avg)
    return pm


def test_powermap_cfg4_order7_vs_oracle(saf, orc):
    """BASELINE configs[3]: 64-channel (order 7) input, 133-band afSTFT, F = 1024, PWD map on the 812-point grid;
    covariance averaging 0.3, map requests on some frames, per-band orders and EQ."""
    order, F, nSH = 7, 1024, 64
    g, o = mk(saf.Powermap, F, order), mk(orc.Powermap, F, order)
    for pm in (g, o):
        for b in range(60, 133):
            pm.setAnaOrder(4, b)
        pm.setPowermapEQ(0.5, 10); pm.setPowermapEQ(0.0, 100)
    srcs = orc.getRSH(order, np.array([[40.0, 10.0], [-120.0, -35.0]], np.float32))
    s = frames(8, 2, 6 * F) * np.array([[1.0], [0.5]], np.float32)
    x = (srcs @ s + 0.05 * frames(9, nSH, 6 * F)).astype(np.float32)
    x /= np.sqrt(2 * np.arange(8).repeat(2 * np.arange(8) + 1) + 1)[:, None].astype(np.float32)      # N3D -> SN3D input
    for f in range(6):
        if f in (1, 3, 5):
            g.requestPmapUpdate(); o.requestPmapUpdate()
        blk = x[:, f * F:(f + 1) * F]
        g.analysis(blk); o.analysis(blk)
        if f in (1, 3, 5):
            assert relrms(g.rawPmap(), o.rawPmap()) < TOL, f
            mg, mo = g.getPmap(), o.getPmap()
            assert maxabs(mg, mo) < 1e-4 and mg.argmax() == mo.argmax()
    assert relrms(g.Cx(nSH), o.Cx(nSH)) < TOL


def test_powermap_fifo_partial_blocks_and_flags(saf, orc):
    """Sample-wise FIFO (powermap.c:222-230): odd block sizes, isPlaying = 0 drops the frame, first order FuMa input."""
    order, F = 1, 256
    g, o = mk(saf.Powermap, F, order, norm=3, chOrder=2, cov=0.0), mk(orc.Powermap, F, order, norm=3, chOrder=2, cov=0.0)
    x = frames(3, 4, 5 * F)
    pos = 0
    for n, playing in ((100, 1), (300, 1), (112, 1), (256, 0), (200, 1), (312, 1)):
        blk = np.ascontiguousarray(x[:, pos:pos + n]); pos += n
        g.analysis(blk, playing); o.analysis(blk, playing)
    assert g.getPmap() is not None
    assert relrms(g.Cx(4), o.Cx(4)) < TOL
    assert relrms(g.rawPmap(), o.rawPmap()) < TOL
    g.setPowermapMode(4); g.setPowermapMode(1)                    # mode change clears the previous map (powermap.c:389-395)
    o.setPowermapMode(4); o.setPowermapMode(1)
    g.requestPmapUpdate(); o.requestPmapUpdate()
    blk = np.ascontiguousarray(x[:, :F])
    g.analysis(blk); o.analysis(blk)
    assert relrms(g.rawPmap(), o.rawPmap()) < TOL


def test_powermap_device_entry_equals_frame_by_frame(saf, orc):
    """saf_hip_powermap_analysis_dev: several frames per call == the oracle frame by frame (recursive averaging kept in order)."""
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    order, F, nSH, nF = 5, 1024, 36, 6
    g, o = mk(saf.Powermap, F, order, norm=1), mk(orc.Powermap, F, order, norm=1)
    x = frames(21, nSH, nF * F)
    for f in range(nF):
        if f == nF - 1:
            o.requestPmapUpdate()
        o.analysis(x[:, f * F:(f + 1) * F])
    d_x = torch.from_numpy(x).cuda()
    g.analysis_dev(d_x.data_ptr(), (F, nF * F), nSH, 2)             # (this call also serves the initial map request)
    g.analysis_dev(d_x[:, 2 * F:].data_ptr(), (F, nF * F), nSH, 3)
    g.requestPmapUpdate()
    g.analysis_dev(d_x[:, 5 * F:].data_ptr(), (F, nF * F), nSH, 1)
    torch.cuda.synchronize()
    assert relrms(g.Cx(nSH), o.Cx(nSH)) < TOL
    saf.set_stream(None)
