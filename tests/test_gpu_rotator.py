"""rotator (examples/include/rotator.h; rotation of an Ambisonic scene by a real SH rotation matrix, one block of latency,
linear cross-fade when the rotation changed) on the GPU build against the CPU oracle and the reference's own known-answer
test — needs an MI355X.  Tolerance: 1e-6 absolute like the reference's test (signals of order 1)."""
import numpy as np
import pytest

from util import frames, maxabs

pytestmark = pytest.mark.gpu


def test_reference_example_rotator_known_answer_on_gpu(saf, orc):
    """test__saf_example_rotator (test/src/test__examples.c:357-440): order 4, N3D, yaw / pitch / roll (-0.4, -1.4, 2.1) rad;
    output == getSHrotMtxReal(yawPitchRoll2Rzyx(...)) x input, delayed by rotator_getProcessingDelay(), within 1e-6."""
    order, F = 4, 64
    nSH = (order + 1) ** 2
    ypr = (-0.4, -1.4, 2.1)
    r = saf.Rotator(F); r.init(48000)
    r.setOrder(order); r.setNormType(1)
    r.setYaw(float(np.degrees(ypr[0]))); r.setPitch(float(np.degrees(ypr[1]))); r.setRoll(float(np.degrees(ypr[2])))
    assert saf.load().rotator_getProcessingDelay() == F and r.getNSHrequired() == nSH
    sig = frames(8, 1, 60 * F)
    sh = (orc.getRSH(order, np.array([[90.0, 0.0]], np.float32)) @ sig).astype(np.float32)
    ref = orc.getSHrotMtxReal(orc.yawPitchRoll2Rzyx(*ypr, 0), order) @ sh
    out = np.concatenate([r.process(np.ascontiguousarray(sh[:, i * F:(i + 1) * F]), nSH) for i in range(60)], 1)
    assert maxabs(ref[:, :-F], out[:, F:]) <= 1e-6


@pytest.mark.parametrize("order,F,chOrder", [(7, 128, 1), (1, 64, 2), (3, 100, 1), (5, 512, 1)])
def test_rotator_scenarios_vs_oracle(saf, orc, order, F, chOrder):
    """rotation changed mid-stream by Euler angles, by quaternion, with flips and the roll-pitch-yaw convention; FuMa
    channel order at first order; fewer inputs / outputs than SH channels; wrong block size -> zeros"""
    nSH = (order + 1) ** 2
    g, o = saf.Rotator(F), orc.Rotator(F)
    for r in (g, o):
        r.init(48000); r.setOrder(order); r.setChOrder(chOrder)
    x = frames(order * 10 + 3, nSH, 14 * F)
    for blk in range(14):
        for r in (g, o):
            if blk == 2:
                r.setYaw(40.0); r.setPitch(-25.0); r.setRoll(10.0)
            if blk == 5:
                r.setRPYflag(1); r.setFlipPitch(1); r.setYaw(-120.0)
            if blk == 8:
                r.setQuaternionW(0.5); r.setQuaternionX(-0.5); r.setQuaternionY(0.5); r.setQuaternionZ(0.5)
            if blk == 11:
                r.setFlipQuaternion(1)
        nIn = nSH if blk != 6 else max(1, nSH - 2)
        nOut = nSH + 2 if blk != 9 else max(1, nSH - 1)
        xb = np.ascontiguousarray(x[:nIn, blk * F:(blk + 1) * F])
        yg, yo = g.process(xb, nOut), o.process(xb, nOut)
        assert maxabs(yg, yo) < 2e-6, blk
        if blk in (3, 8, 12):
            for nm in ("Yaw", "Pitch", "Roll", "QuaternionW", "QuaternionX", "QuaternionY", "QuaternionZ"):
                assert abs(getattr(g, "get" + nm)() - getattr(o, "get" + nm)()) < 1e-4, (blk, nm)
    assert np.abs(yo).max() > 0.05
    assert not g.process(np.ones((nSH, F // 2), np.float32), nSH, nSamples=F // 2).any()


def test_rotator_order0_and_device_entry(saf, orc):
    """order 0 passes the omni through without delay; several blocks per call on device-resident signals"""
    import torch
    F = 64
    g = saf.Rotator(F); g.init(48000); g.setOrder(0)
    x = frames(5, 1, F)
    y = g.process(x, 3)
    assert np.array_equal(y[0], x[0]) and not y[1:].any()
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    order, F, nF = 6, 256, 5
    nSH = 49
    g, o = saf.Rotator(F), orc.Rotator(F)
    for r in (g, o):
        r.init(48000); r.setOrder(order); r.setYaw(77.0); r.setRoll(-33.0)
    x = frames(9, nSH, 2 * nF * F)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros(nSH, 2 * nF * F, device="cuda")
    yo = []
    for call in range(2):
        if call == 1:
            for r in (g, o):
                r.setPitch(15.0)
        g.process_dev(d_in[:, call * nF * F:].data_ptr(), (F, 2 * nF * F), nSH, d_out[:, call * nF * F:].data_ptr(), (F, 2 * nF * F), nSH, nF)
        for i in range(call * nF, (call + 1) * nF):
            yo.append(o.process(np.ascontiguousarray(x[:, i * F:(i + 1) * F]), nSH))
    torch.cuda.synchronize()
    yo = np.concatenate(yo, 1)
    assert maxabs(d_out.cpu().numpy(), yo) < 2e-6
    # input and output interleaved in ONE tensor [2][nSH][time] viewed per channel pair — extents overlap, no block shares memory
    # with another (the exact overlap test of launch_enc_gemm: an extent test would abort here)
    g2, o2 = saf.Rotator(F), orc.Rotator(F)
    for r in (g2, o2):
        r.init(48000); r.setOrder(order); r.setYaw(-20.0)
    t = torch.zeros(nSH, 2, nF * F, device="cuda")
    t[:, 0] = torch.from_numpy(x[:, :nF * F]).cuda()
    g2.process_dev(t[:, 0].data_ptr(), (F, 2 * nF * F), nSH, t[:, 1].data_ptr(), (F, 2 * nF * F), nSH, nF)
    torch.cuda.synchronize()
    yo2 = np.concatenate([o2.process(np.ascontiguousarray(x[:, i * F:(i + 1) * F]), nSH) for i in range(nF)], 1)
    assert maxabs(t[:, 1].cpu().numpy(), yo2) < 2e-6
    saf.set_stream(None)
