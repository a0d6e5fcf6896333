"""ambi_enc on the GPU (libsaf_hip.so through its C-ABI) against the CPU oracle and the
reference's own known-answer test — needs an MI355X:  python -m pytest tests -m gpu

Tolerances: the reference's test__saf_example_ambi_enc asserts 1e-6 absolute; against the
oracle the only difference is the summation order inside the [nSH x nSrc] product
(MFMA accumulates k-pairs), so a few float32 ulps (2e-6 relative RMS) are asserted.
"""
from pathlib import Path

import numpy as np
import pytest

from util import frames, relrms, maxabs

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def test_reference_example_ambi_enc_known_answer_on_gpu(saf, orc):
    """test__saf_example_ambi_enc (test/src/test__examples.c:192-263): 2 sources, order 4, N3D, no post-scaling;
    output == getRSH * input delayed by ambi_enc_getProcessingDelay() = one block, within 1e-6."""
    order, F = 4, 64
    L = F * 150
    e = saf.AmbiEnc(F); e.init(48000)
    e.setOutputOrder(order); e.setNormType(1); e.setEnablePostScaling(0); e.setNumSources(2)
    dirs = np.array([[90.0, 0.0], [20.0, -45.0]], np.float32)
    for i in range(2):
        e.setSourceAzi_deg(i, float(dirs[i, 0])); e.setSourceElev_deg(i, float(dirs[i, 1]))
    x = frames(21, 2, L)
    y = np.concatenate([e.process(x[:, i * F:(i + 1) * F], 25) for i in range(L // F)], 1)
    ref = orc.getRSH(order, dirs) @ x
    assert saf.load().ambi_enc_getProcessingDelay() == F
    # (the cross-fade of the first call mixes two encodings of the all-zero "previous" block)
    assert maxabs(ref[:, :L - F], y[:, F:]) <= 1e-6


def test_ambi_enc_scenario_vs_oracle_and_golden(saf, orc):
    from make_golden import ambi_enc_scenario
    yo = ambi_enc_scenario(orc.AmbiEnc)
    yg = ambi_enc_scenario(saf.AmbiEnc)
    assert relrms(yg, yo) < 2e-6 and maxabs(yg, yo) < 2e-6
    assert not yg[16:].any()                                  # outputs beyond nSH are zero-filled (ambi_enc.c:189-190)
    assert relrms(yg, np.load(GOLD / "ambi_enc_small.npz")["out"]) < 2e-6


def test_ambi_enc_cfg1_first_order_4ch_256(saf, orc):
    """BASELINE configs[0]: ambi_enc 1st order, 4 sources, 256-sample blocks (sources at __default_LScoords64_rad[0..3])."""
    def mkenc(cls):
        e = cls(256); e.init(48000)
        e.setOutputOrder(1); e.setNumSources(4)
        return e
    g, o = mkenc(saf.AmbiEnc), mkenc(orc.AmbiEnc)
    x = frames(77, 4, 10 * 256)
    for f in range(10):
        blk = x[:, f * 256:(f + 1) * 256]
        assert maxabs(g.process(blk, 4), o.process(blk, 4)) < 1e-6


@pytest.mark.parametrize("chOrder,norm", [(2, 3), (1, 1), (2, 2)])
def test_ambi_enc_conventions_missing_inputs_and_bad_block(saf, orc, chOrder, norm):
    """FuMa ordering/normalisation (first order only), fewer input channels than sources, wrong nSamples -> zeros."""
    def mkenc(cls):
        e = cls(128); e.init(48000)
        e.setOutputOrder(1); e.setNumSources(5); e.setChOrder(chOrder); e.setNormType(norm)
        e.setSourceGain(3, 0.0)
        return e
    g, o = mkenc(saf.AmbiEnc), mkenc(orc.AmbiEnc)
    x = frames(9, 3, 6 * 128)                                  # only 3 of the 5 sources are fed
    for f in range(6):
        blk = x[:, f * 128:(f + 1) * 128]
        assert maxabs(g.process(blk, 6), o.process(blk, 6)) < 1e-6
    y = g.process(np.ones((3, 64), np.float32), 4, nSamples=64)
    assert not y.any()


def test_ambi_enc_batch_equals_single_handles_and_feeds_ambi_dec(saf, orc):
    """Batched device-pointer path: 3 encoder instances x 2 calls of 4 blocks == the oracle's block-by-block output
    (directions of instance 1 change between the calls -> cross-fade on the first block of call 2); then the encoded
    blocks are decoded on the device by the ambi_dec batch (the encode->decode chain of BASELINE configs[4])."""
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    F, nI, nF, nS, order = 512, 3, 4, 64, 7
    src = orc.table("SphCovering_64_dirs_deg")

    def mkenc(cls, i):
        e = cls(F); e.init(48000)
        e.setOutputOrder(order); e.setNumSources(nS); e.setNormType(1)
        for s in range(nS):
            e.setSourceAzi_deg(s, float(src[(s + 7 * i) % 64, 0])); e.setSourceElev_deg(s, float(src[(s + 7 * i) % 64, 1]))
        return e
    ge, oe = [mkenc(saf.AmbiEnc, i) for i in range(nI)], [mkenc(orc.AmbiEnc, i) for i in range(nI)]
    bt = saf.AmbiEncBatch(ge, nF)
    x = np.stack([frames(90 + i, 2 * nF * nS, F).reshape(2 * nF, nS, F) for i in range(nI)])
    d_in = torch.from_numpy(x).cuda()
    d_sh = torch.zeros(nI, 2 * nF, 64, F, device="cuda")
    st = (2 * nF * 64 * F, 64 * F, F)
    yo = np.zeros((nI, 2 * nF, 64, F), np.float32)
    for call in range(2):
        if call == 1:
            for e in (ge[1], oe[1]):
                e.setSourceAzi_deg(5, 12.0); e.setSourceElev_deg(40, -20.0)
        bt.process_ptr(d_in[:, call * nF:].data_ptr(), st, nS, d_sh[:, call * nF:].data_ptr(), st, 64, nF)
        for i in range(nI):
            for f in range(call * nF, (call + 1) * nF):
                yo[i, f] = oe[i].process(x[i, f], 64)
    torch.cuda.synchronize()
    ysh = d_sh.cpu().numpy()
    for i in range(nI):
        assert relrms(ysh[i], yo[i]) < 2e-6, i

    # chain: decode the encoded blocks on the device
    def mkdec(cls):
        d = cls(F)
        d.setNormType(1); d.setChOrder(1); d.setMasterDecOrder(order); d.setOutputConfigPreset(29)
        d.setDecMethod(0, 1); d.setDecMethod(1, 1); d.initCodec(); d.init(48000); d.setDecOrderAllBands(order)
        return d
    gd = [mkdec(saf.AmbiDec) for _ in range(nI)]
    od = mkdec(orc.AmbiDec)
    bd = saf.AmbiDecBatch(gd, 2 * nF)
    d_ls = torch.zeros_like(d_sh)
    bd.process_ptr(d_sh.data_ptr(), st, d_ls.data_ptr(), st, 2 * nF)
    torch.cuda.synchronize()
    ref = np.stack([od.process(yo[0, f], 64) for f in range(2 * nF)])
    assert relrms(d_ls[0].cpu().numpy(), ref) < 1e-5
    saf.set_stream(None)


def test_encode_decode_chain_at_the_per_gpu_share_of_2048_sources(saf, orc):
    """BASELINE configs[4] at its full per-GPU size: 32 scenes x 64 sources (x 8 GPUs = 2048 sources), order 7 encode ->
    64-loudspeaker decode, 16 blocks of 512 per call, every band its own decoding order.  The oracle is too slow for all of it:
    scene 0 and scene 31 are checked against it on the first blocks, the rest through properties of the chain — linear in the
    source signals, scenes independent of each other, and the same stream cut into different calls gives bit-identical output."""
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    F, nI, nS, nF, order = 512, 32, 64, 16, 7
    src = orc.table("SphCovering_64_dirs_deg")
    orders = 1 + (np.arange(133) * 5) % 7

    def mkenc(cls, i):
        e = cls(F); e.init(48000); e.setOutputOrder(order); e.setNumSources(nS); e.setNormType(1)
        for s in range(nS):
            e.setSourceAzi_deg(s, float(src[(s + 7 * i) % 64, 0])); e.setSourceElev_deg(s, float(src[(s + 7 * i) % 64, 1]))
        return e

    def mkdec(cls):
        d = cls(F); d.setNormType(1); d.setChOrder(1); d.setMasterDecOrder(order); d.setOutputConfigPreset(29)
        d.setDecMethod(0, 1); d.setDecMethod(1, 1); d.initCodec(); d.init(48000); d.setDecOrderAllBands(order)
        for b, o in enumerate(orders):
            d.setDecOrder(int(o), b)
        return d

    st = (nF * 64 * F, 64 * F, F)

    def chain(x, split=(nF,)):
        eb = saf.AmbiEncBatch([mkenc(saf.AmbiEnc, i) for i in range(nI)], nF)
        db = saf.AmbiDecBatch([mkdec(saf.AmbiDec) for _ in range(nI)], nF)
        sh = torch.zeros(nI, nF, 64, F, device="cuda"); ls = torch.zeros_like(sh)
        f0 = 0
        for n in split:
            eb.process_ptr(x[:, f0:].data_ptr(), st, nS, sh[:, f0:].data_ptr(), st, 64, n)
            db.process_ptr(sh[:, f0:].data_ptr(), st, ls[:, f0:].data_ptr(), st, n)
            f0 += n
        torch.cuda.synchronize()
        assert db.lastPath() == 1
        return ls

    g = torch.Generator(device="cuda"); g.manual_seed(5)
    a = torch.rand(nI, nF, nS, F, device="cuda", generator=g) * 2 - 1
    b = torch.rand(nI, nF, nS, F, device="cuda", generator=g) * 2 - 1
    ya, yb = chain(a), chain(b)
    lin = 1.5 * ya - 0.25 * yb
    assert float((chain(1.5 * a - 0.25 * b) - lin).norm() / lin.norm()) < 1e-6
    assert torch.equal(chain(a, split=(1, 7, 8)), ya)
    a2 = a.clone(); a2[9] = b[9]
    y2 = chain(a2)
    assert torch.equal(y2[:9], ya[:9]) and torch.equal(y2[10:], ya[10:]) and torch.equal(y2[9], yb[9])
    for i in (0, nI - 1):
        oe, od = mkenc(orc.AmbiEnc, i), mkdec(orc.AmbiDec)
        xa = a[i, :3].cpu().numpy()
        ref = np.stack([od.process(oe.process(xa[f], 64), 64) for f in range(3)])
        assert relrms(ya[i, :3].cpu().numpy(), ref) < 1e-5, i
    saf.set_stream(None)


@pytest.mark.parametrize("F,nS,nIn,order,nOut,norm", [(200, 23, 20, 5, 30, 2), (512, 40, 40, 6, 64, 1), (132, 17, 17, 3, 16, 1), (96, 9, 9, 2, 12, 2)])
def test_ambi_enc_mid_size_scenes_ragged_blocks(saf, orc, F, nS, nIn, order, nOut, norm):
    """Scenes between the two kernel variants' home sizes: block sizes that are not multiples of the 128-column tile,
    fewer fed channels than sources, per-source gains, fewer outputs than SH channels, a source moved mid-stream
    (cross-fade block) and post-scaling on."""
    src = orc.table("SphCovering_64_dirs_deg")

    def mkenc(cls):
        e = cls(F); e.init(48000)
        e.setOutputOrder(order); e.setNumSources(nS); e.setNormType(norm); e.setEnablePostScaling(1)
        for s in range(nS):
            e.setSourceAzi_deg(s, float(src[s, 0])); e.setSourceElev_deg(s, float(src[s, 1]))
        e.setSourceGain(1, 0.5); e.setSourceGain(nS - 1, 1.7); e.setSourceGain(4, 0.0)
        return e
    g, o = mkenc(saf.AmbiEnc), mkenc(orc.AmbiEnc)
    x = frames(31, nIn, 7 * F)
    for f in range(7):
        if f == 4:
            for e in (g, o):
                e.setSourceAzi_deg(2, -100.0); e.setSourceElev_deg(nS - 2, 33.0)
        blk = np.ascontiguousarray(x[:, f * F:(f + 1) * F])
        yg, yo = g.process(blk, nOut), o.process(blk, nOut)
        assert maxabs(yg, yo) < 2e-6, f
    assert np.abs(yo).max() > 0.1
