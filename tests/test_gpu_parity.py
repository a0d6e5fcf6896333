"""Parity of the HIP path (libsaf_hip.so, called through its C-ABI) with the CPU
oracle — needs an MI355X:  python -m pytest tests -m gpu

Tolerance: BASELINE.json / SURVEY §8d ask for <= 1e-5 relative RMS against the
reference CPU path for floating-point outputs and bit-exact triangulation
indices.  The asserts below use TOL = 1e-5 where the north star states it and
tighter bounds (a few float32 ulps) where the algorithm allows.
"""
import json
from pathlib import Path

import numpy as np
import pytest

from util import frames, relrms, maxabs, canon_faces

pytestmark = pytest.mark.gpu
TOL = 1e-5          # north_star: "output within 1e-5 RMS of reference"
GOLD = Path(__file__).parent / "golden"


def tdesign(orc, degree):
    return orc.table(f"Tdesign_degree_{degree}_dirs_deg")


# ------------------------------------------------------------------ afSTFT
def test_afSTFT_forward_backward_vs_oracle(saf, orc):
    nin, nout, F = 7, 5, 512
    g, o = saf.AfSTFT(nin, nout), orc.AfSTFT(nin, nout)
    assert (g.nBands, g.delay) == (o.nBands, o.delay) == (133, 1536)
    rng = np.random.default_rng(0)
    for fr in range(8):
        x = frames(10 + fr, nin, F)
        assert relrms(g.forward(x), o.forward(x)) < 1e-6
        Y = (rng.normal(size=(133, nout, 4)) + 1j * rng.normal(size=(133, nout, 4))).astype(np.complex64)
        assert relrms(g.backward(Y), o.backward(Y)) < 1e-6
    assert np.allclose(g.centreFreqs(48000.0), o.centreFreqs(48000.0), rtol=0, atol=0)
    assert np.array_equal(saf.afSTFT_getCentreFreqs_nullHandle(44100.0), orc.centreFreqs_nullHandle(44100.0))


def test_afSTFT_reference_unit_test_on_gpu(saf):
    """test__afSTFT (test/src/test__resources.c:27-100) against the HIP filterbank."""
    F, nin, nout = 512, 60, 64
    L = 24000
    x = frames(11, nin, L)
    st = saf.AfSTFT(nin, nout)
    st.channelChange(100, 5); st.clearBuffers(); st.channelChange(39, 81); st.channelChange(nin, nout); st.clearBuffers()
    out = np.zeros((nout, L), np.float32)
    for fr in range(L // F):
        S = st.forward(x[:, fr * F:(fr + 1) * F])
        out[:, fr * F:(fr + 1) * F] = st.backward(np.ascontiguousarray(np.repeat(S[:, 0:1, :], nout, axis=1)))
    n = (L // F) * F - st.delay - F
    assert np.abs(x[0, :n] - out[0, st.delay:st.delay + n]).max() <= 0.01


def test_afSTFT_channelChange_keeps_surviving_state(saf, orc):
    g, o = saf.AfSTFT(4, 4), orc.AfSTFT(4, 4)
    x = frames(1, 4, 512)
    g.forward(x); o.forward(x)
    g.channelChange(6, 2); o.channelChange(6, 2)           # channels 0..3 keep their history, 4..5 start from zero
    x2 = frames(2, 6, 256)
    assert relrms(g.forward(x2), o.forward(x2)) < 1e-6
    Y = (frames(3, 133 * 2 * 2, 2)[:, 0] + 1j * frames(3, 133 * 2 * 2, 2)[:, 1]).reshape(133, 2, 2).astype(np.complex64)
    assert relrms(g.backward(Y), o.backward(Y)) < 1e-6
    g.clearBuffers(); o.clearBuffers()
    assert relrms(g.forward(x2), o.forward(x2)) < 1e-6


def test_afSTFT_knownDimensions_layout_and_untouched_padding(saf, orc):
    """afSTFT_forward_knownDimensions (afSTFTlib.c:267-308): band stride dataFD_nCH*dataFD_nHops, rows beyond nCHin untouched."""
    g, o = saf.AfSTFT(3, 1), orc.AfSTFT(3, 1)
    x = frames(8, 3, 256)
    A = g.forward_knownDimensions(x, 64, 2)
    B = o.forward(x, nCH_alloc=64)
    assert A.shape == (133, 64, 2) and relrms(A[:, :3], B[:, :3]) < 1e-6
    assert not A[:, 3:].any()


def test_afSTFT_split_invariance_and_long_launch(saf, orc):
    """One 40-hop call == 2 + 22 + 16 hop calls (device state carries over); also exercises the time-chunked grid."""
    x = frames(5, 2, 40 * 128)
    a, b, o = saf.AfSTFT(2, 2), saf.AfSTFT(2, 2), orc.AfSTFT(2, 2)
    A = a.forward(x)
    B = np.concatenate([b.forward(x[:, :256]), b.forward(x[:, 256:24 * 128]), b.forward(x[:, 24 * 128:])], axis=2)
    assert np.array_equal(A, B)
    assert relrms(A, o.forward(x)) < 1e-6
    ya = a.backward(A)
    yb = np.concatenate([b.backward(np.ascontiguousarray(A[:, :, :7])), b.backward(np.ascontiguousarray(A[:, :, 7:]))], axis=1)
    assert np.array_equal(ya, yb)
    assert relrms(ya, o.backward(A)) < 1e-6


def test_afSTFT_nonhybrid_and_lowdelay_modes(saf, orc):
    for ld, hyb in ((0, 0), (1, 1), (1, 0)):
        g, o = saf.AfSTFT(2, 2, 128, ld, hyb), orc.AfSTFT(2, 2, 128, ld, hyb)
        assert g.nBands == o.nBands and g.delay == o.delay
        yg, yo = [], []
        for fr in range(12):                      # 36 hops: well past the filterbank delay
            x = frames(40 + fr, 2, 384)
            Xg, Xo = g.forward(x), o.forward(x)
            assert relrms(Xg, Xo) < 2e-6
            yg.append(g.backward(Xo)); yo.append(o.backward(Xo))
        assert relrms(np.concatenate(yg, 1), np.concatenate(yo, 1)) < 2e-6


def test_afSTFT_time_ch_bands_format(saf, orc):
    g = saf.AfSTFT(2, 2, fmt=saf.AFSTFT_TIME_CH_BANDS)
    o = orc.AfSTFT(2, 2)
    x = frames(77, 2, 384)
    A = g.forward(x)                                  # [t][ch][band]
    assert relrms(A.transpose(2, 1, 0), o.forward(x)) < 1e-6


def test_afSTFT_device_pointer_entry(saf, orc):
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    nch, nH = 3, 40          # well past the 12-hop delay, so the output is not just pre-ringing
    x = frames(6, nch, nH * 128)
    d_x = torch.from_numpy(x).cuda()
    d_X = torch.zeros(133, nch, nH, dtype=torch.complex64, device="cuda")
    g, o = saf.AfSTFT(nch, nch), orc.AfSTFT(nch, nch)
    g.forward_dev(d_x.data_ptr(), nH * 128, nH, d_X.data_ptr(), nch * nH, nH)
    d_y = torch.zeros(nch, nH * 128, device="cuda")
    g.backward_dev(d_X.data_ptr(), nch * nH, nH, nH, d_y.data_ptr(), nH * 128)
    torch.cuda.synchronize()
    Xo = o.forward(x)
    assert relrms(d_X.cpu().numpy(), Xo) < 1e-6
    assert relrms(d_y.cpu().numpy(), o.backward(Xo)) < 1e-6
    saf.set_stream(None)


def test_FIRtoFilterbankCoeffs_vs_oracle(saf, orc):
    rng = np.random.default_rng(2)
    ir = (rng.normal(size=(5, 2, 200)) * np.exp(-np.arange(200) / 30.0)).astype(np.float32)
    ir[:, :, 20] += 1.0
    A, B = saf.afSTFT_FIRtoFilterbankCoeffs(ir), orc.FIRtoFilterbankCoeffs(ir)
    assert relrms(A, B) < 5e-6


def test_golden_afstft(saf):
    ref = np.load(GOLD / "afstft_small.npz")
    st = saf.AfSTFT(3, 2)
    x = frames(101, 3, 12 * 256)
    specs = [st.forward(x[:, i * 256:(i + 1) * 256]) for i in range(12)]
    y = np.concatenate([st.backward(np.ascontiguousarray(s[:, :2, :])) for s in specs], 1)
    assert relrms(specs[-1], ref["spec_last"]) < 1e-6 and relrms(y, ref["synth"]) < 1e-6


# ------------------------------------------------------------------ SH / HOA / VBAP
def test_SH_vs_oracle(saf, orc):
    rng = np.random.default_rng(4)
    deg = np.stack([rng.uniform(-180, 180, 300), rng.uniform(-90, 90, 300)], 1).astype(np.float32)
    rad = np.stack([rng.uniform(-np.pi, np.pi, 300), rng.uniform(0, np.pi, 300)], 1).astype(np.float32)
    for order in (1, 3, 7, 10, 15):
        s = 4.0 * np.finfo(np.float32).eps * (2 * order + 1)       # a few ulps of the largest SH value
        assert maxabs(saf.getRSH(order, deg), orc.getRSH(order, deg)) <= s
        assert maxabs(saf.getSHreal(order, rad), orc.getSHreal(order, rad)) <= s
        # float32 recursion: device sinf/cosf/powf differ from glibc in the last ulp and the recursion amplifies that
        assert maxabs(saf.getRSH_recur(order, deg), orc.getRSH_recur(order, deg)) <= 3e-6 * (order + 1) ** 2
        assert maxabs(saf.getSHreal_recur(order, rad), orc.getSHreal_recur(order, rad)) <= 1e-6 * (order + 1) ** 2
    # poles and the single-direction case
    for d in ([[0.0, 90.0]], [[12.0, -90.0]], [[180.0, 0.0]]):
        assert maxabs(saf.getRSH(7, d), orc.getRSH(7, d)) <= 1e-6
        assert maxabs(saf.getRSH_recur(7, d), orc.getRSH_recur(7, d)) <= 1e-5
    assert np.array_equal(saf.getMaxREweights(7), orc.getMaxREweights(7))
    assert np.array_equal(saf.getMaxREweights(3, True), orc.getMaxREweights(3, True))


@pytest.mark.parametrize("order", [1, 2, 4, 7, 10])
def test_reference_SH_and_decoder_properties_on_gpu(saf, orc, order):
    """test__getSHreal (test__sh_module.c:27-82) and test__getLoudspeakerDecoderMtx (test__hoa_module.c:27-104) on the HIP build."""
    ls = tdesign(orc, 2 * order)
    nLS, nSH = len(ls), (order + 1) ** 2
    rad = np.stack([ls[:, 0] * np.pi / 180, np.pi / 2 - ls[:, 1] * np.pi / 180], 1).astype(np.float32)
    Y = saf.getSHreal(order, rad)
    assert np.abs((Y @ Y.T) * (4 * np.pi / nLS) - np.eye(nSH)).max() <= 1e-5
    S, M, E, A = (saf.getLoudspeakerDecoderMtx(ls, m, order) for m in (1, 2, 3, 4))
    assert maxabs(S, M) <= 1e-5 and maxabs(S, E) <= 1e-5
    LS = E @ Y
    assert np.abs(LS.sum(0) - 1).max() <= 1e-5 and np.abs((LS ** 2).sum(0) - nSH / nLS).max() <= 1e-5


@pytest.mark.parametrize("layout,order", [("SphCovering_64_dirs_deg", 7), ("SphCovering_49_dirs_deg", 6), ("Tdesign_degree_10_dirs_deg", 5), ("22pX_dirs_deg", 3)])
def test_decoder_matrices_vs_oracle(saf, orc, layout, order):
    ls = orc.table(layout)
    for method in (1, 2, 3, 4):
        for maxre in (0, 1):
            A = saf.getLoudspeakerDecoderMtx(ls, method, order, maxre)
            B = orc.getLoudspeakerDecoderMtx(ls, method, order, maxre)
            assert maxabs(A, B) <= 2e-6, (layout, method)


@pytest.mark.parametrize("layout", ["SphCovering_64_dirs_deg", "SphCovering_49_dirs_deg", "Tdesign_degree_10_dirs_deg", "SphCovering_9_dirs_deg"])
def test_triangulation_indices_bit_exact(saf, orc, layout):
    """Bit-exact VBAP triangulation indices on layouts where the reference itself is deterministic:
    compared as canonically sorted face sets."""
    ls = orc.table(layout)
    Vg, Fg = saf.findLsTriplets(ls)
    Vo, Fo = orc.findLsTriplets(ls)
    assert np.array_equal(Vg, Vo)
    assert np.array_equal(canon_faces(Fg), canon_faces(Fo))
    assert len(Fg) == 2 * len(ls) - 4


def test_vbap_tables_vs_oracle(saf, orc):
    ls = orc.table("SphCovering_25_dirs_deg")
    rng = np.random.default_rng(6)
    src = np.stack([rng.uniform(-180, 180, 500), rng.uniform(-90, 90, 500)], 1).astype(np.float32)
    for spread in (0.0, 30.0):
        Gg, ng = saf.generateVBAPgainTable3D_srcs(src, ls, 0, 0, spread)
        Go, no = orc.generateVBAPgainTable3D_srcs(src, ls, 0, 0, spread)
        assert ng == no and maxabs(Gg, Go) <= 2e-6
    half = ls[ls[:, 1] > -10]                                     # needs a dummy loudspeaker below
    Gg, _ = saf.generateVBAPgainTable3D(half, 5, 5, 1, 1, 0.0)
    Go, _ = orc.generateVBAPgainTable3D(half, 5, 5, 1, 1, 0.0)
    assert Gg.shape == Go.shape == (73 * 37, len(half)) and maxabs(Gg, Go) <= 2e-6
    cg, ig = saf.compressVBAPgainTable3D(Gg)
    co, io = orc.compressVBAPgainTable3D(Go)
    assert np.array_equal(ig, io) and maxabs(cg, co) <= 2e-6


def test_golden_sh_decoders(saf, orc):
    ref = np.load(GOLD / "sh_decoders_small.npz")
    ls = orc.table("SphCovering_9_dirs_deg")
    assert maxabs(saf.getRSH(3, ls), ref["rsh"]) <= 1e-6 and maxabs(saf.getRSH_recur(3, ls), ref["rsh_recur"]) <= 1e-5
    for m, k in ((1, "sad"), (2, "mmd"), (3, "epad"), (4, "allrad")):
        assert maxabs(saf.getLoudspeakerDecoderMtx(ls, m, 2), ref[k]) <= 2e-6
    assert np.array_equal(canon_faces(saf.findLsTriplets(ls)[1]), canon_faces(ref["faces"]))


# ------------------------------------------------------------------ ambi_dec
def mk(cls, F, order, preset, m0, m1, norm=1, low_order=None, **kw):
    a = cls(F)
    a.setNormType(norm); a.setChOrder(1); a.setMasterDecOrder(order); a.setOutputConfigPreset(preset)
    a.setDecMethod(0, m0); a.setDecMethod(1, m1)
    a.initCodec(); a.init(48000); a.setDecOrderAllBands(order)
    if low_order:
        for b in range(40, 133):
            a.setDecOrder(low_order, b)
    for k, v in kw.items():
        getattr(a, k)(*v)
    return a


def run(dec, x, nOut, F):
    return np.concatenate([dec.process(np.ascontiguousarray(x[:, i * F:(i + 1) * F]), nOut) for i in range(x.shape[1] // F)], 1)


def test_ambi_dec_headline_config_vs_oracle(saf, orc):
    """BASELINE config 2: order 7, 64 SH -> 64 loudspeakers (SphCovering-64), 512-sample blocks, SAD, maxrE, energy-preserving."""
    g, o = mk(saf.AmbiDec, 512, 7, 29, 1, 1), mk(orc.AmbiDec, 512, 7, 29, 1, 1)
    x = frames(1, 64, 12 * 512)
    yg, yo = run(g, x, 64, 512), run(o, x, 64, 512)
    assert relrms(yg[:, 1536:], yo[:, 1536:]) < TOL
    assert relrms(yg, yo) < 1e-6 and maxabs(yg, yo) < 2e-6
    for dec in (0, 1):
        assert np.array_equal(g.decMtx(dec, 7, 1, 64), o.decMtx(dec, 7, 1, 64))
        assert abs(g.Mnorm(dec, 7, 1) - o.Mnorm(dec, 7, 1)) < 1e-6


@pytest.mark.parametrize("m0,m1,norm,low", [(1, 3, 2, 3), (4, 2, 1, 2), (3, 4, 2, None)])
def test_ambi_dec_mixed_decoders_orders_norms(saf, orc, m0, m1, norm, low):
    """SURVEY Appendix D scenario: different decoders below/above the transition, lower order above band 40, SN3D input."""
    kw = dict(setDecNormType=(0, 1), setDecEnableMaxrE=(1, 0), setTransitionFreq=(1200.0,))
    g = mk(saf.AmbiDec, 256, 5, 28, m0, m1, norm, low, **kw)
    o = mk(orc.AmbiDec, 256, 5, 28, m0, m1, norm, low, **kw)
    x = frames(3, 36, 14 * 256)
    yg, yo = run(g, x, 49, 256), run(o, x, 49, 256)
    assert relrms(yg, yo) < TOL and relrms(yg, yo) < 2e-6
    # parameters changed on the fly are picked up at the next block (snapshot at block start, ambi_dec.c:479-488)
    for d in (g, o):
        d.setDecEnableMaxrE(0, 0); d.setDecNormType(1, 1); d.setDecOrderAllBands(2); d.setTransitionFreq(700.0)
    x2 = frames(4, 36, 4 * 256)
    assert relrms(run(g, x2, 49, 256), run(o, x2, 49, 256)) < 2e-6


def test_ambi_dec_fuma_first_order_and_missing_channels(saf, orc):
    def cfg(cls):
        a = cls(128)
        a.setMasterDecOrder(1); a.setChOrder(2); a.setNormType(3); a.setOutputConfigPreset(3)      # FuMa/FuMa into 5.x (2-D layout)
        a.setDecMethod(0, 1); a.setDecMethod(1, 4)
        a.initCodec(); a.init(44100)
        return a
    g, o = cfg(saf.AmbiDec), cfg(orc.AmbiDec)
    x = frames(9, 4, 30 * 128)
    # AllRAD sums 5100 float products per matrix entry; host FMA contraction vs the oracle's separate mul/add
    # (and the reference's BLAS blocking) moves that by ~1e-6, so only the north-star tolerance is asserted here
    assert relrms(run(g, x, 5, 128), run(o, x, 5, 128)) < TOL
    x3 = frames(10, 3, 8 * 128)                       # only 3 of 4 inputs supplied -> the 4th is zero; 7 outputs asked -> 2 zeroed
    yg, yo = run(g, x3, 7, 128), run(o, x3, 7, 128)
    assert relrms(yg, yo) < TOL and not yg[5:].any()


def test_ambi_dec_reference_example_test_on_gpu(saf):
    """test__saf_example_ambi_dec (test/src/test__examples.c:109-190), incl. its swapped setDecMethod arguments."""
    d = saf.AmbiDec(128)
    d.setNormType(1); d.setMasterDecOrder(4); d.setOutputConfigPreset(saf.LOUDSPEAKER_ARRAY_PRESET_22PX)
    d.setDecMethod(saf.DECODING_METHOD_SAD, 0); d.setDecMethod(saf.DECODING_METHOD_SAD, 1)
    d.initCodec(); d.init(48000)
    L = 128 * 100
    sh = saf.getRSH(4, [[90.0, 0.0]]) @ frames(4, 1, L)
    out = run(d, sh.astype(np.float32), 22, 128)
    assert int((out ** 2).sum(1).argmax()) == 7


def test_ambi_dec_zero_output_rules(saf):
    d = saf.AmbiDec(256)
    d.setMasterDecOrder(2); d.setOutputConfigPreset(25); d.setDecMethod(0, 1); d.setDecMethod(1, 1)
    x = frames(1, 9, 256)
    assert not d.process(x, 9).any()                                 # not initialised
    d.initCodec(); d.init(48000)
    for _ in range(8):                                               # past the 1536-sample filterbank delay
        y = d.process(x, 9)
    assert y.any()
    assert not d.process(x[:, :128], 9, nSamples=128).any()          # wrong block size
    d.setMasterDecOrder(3)                                           # invalidates the codec
    assert d.getCodecStatus() == saf.CODEC_STATUS_NOT_INITIALISED and not d.process(x, 9).any()
    d.initCodec()
    for _ in range(8):
        y = d.process(frames(2, 16, 256), 9)
    assert y.any() and d.getNSHrequired() == 16


def test_golden_ambi_dec(saf):
    from make_golden import ambi_dec_cfg
    ref = np.load(GOLD / "ambi_dec_small.npz")
    d = ambi_dec_cfg(saf.AmbiDec, 128, 3, 26, 1, 3, low_order=1)
    assert relrms(run(d, frames(202, 16, 24 * 128), 16, 128), ref["out"]) < 2e-6


def test_ambi_dec_batch_equals_single_instances(saf, orc):
    """saf_hip_ambi_dec_batch_process: 3 instances with DIFFERENT decoders x 2 calls of 5 blocks, strided device buffers."""
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    cfgs = [(1, 1, 1, None), (3, 2, 2, 3), (4, 1, 1, 5)]
    decs = [mk(saf.AmbiDec, 512, 7, 29, a, b, n, lo) for a, b, n, lo in cfgs]
    orcs = [mk(orc.AmbiDec, 512, 7, 29, a, b, n, lo) for a, b, n, lo in cfgs]
    nI, nF = 3, 5
    bt = saf.AmbiDecBatch(decs, nF)
    x = np.stack([frames(50 + i, 2 * nF * 64, 512).reshape(2 * nF, 64, 512) for i in range(nI)])
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros(nI, 2 * nF, 64, 512, device="cuda")
    st = (2 * nF * 64 * 512, 64 * 512, 512)
    for call in range(2):
        bt.process_ptr(d_in[:, call * nF:].data_ptr(), st, d_out[:, call * nF:].data_ptr(), st, nF)
    torch.cuda.synchronize()
    yg = d_out.cpu().numpy()
    for i in range(nI):
        yo = np.stack([orcs[i].process(x[i, f], 64) for f in range(2 * nF)])
        assert relrms(yg[i], yo) < 2e-6, i
    # channel-major layout [inst][ch][time] through strides, continuing the same streams
    x2 = np.stack([frames(70 + i, 64, nF * 512) for i in range(nI)])
    d_in2 = torch.from_numpy(x2).cuda(); d_out2 = torch.zeros_like(d_in2)
    st2 = (64 * nF * 512, 512, nF * 512)
    bt.process_ptr(d_in2.data_ptr(), st2, d_out2.data_ptr(), st2, nF)
    torch.cuda.synchronize()
    for i in range(nI):
        yo = np.concatenate([orcs[i].process(np.ascontiguousarray(x2[i][:, f * 512:(f + 1) * 512]), 64) for f in range(nF)], 1)
        assert relrms(d_out2[i].cpu().numpy(), yo) < 2e-6
    saf.set_stream(None)


@pytest.mark.parametrize("nI,nF", [(64, 16), (256, 64)])
def test_ambi_dec_full_size_properties(saf, nI, nF):
    """At bench size (256 instances x 64 blocks per call; and the smaller 64 x 16) the oracle is too slow to run
    everywhere, so check size-independent properties: linearity, instance independence, agreement of two different
    batch splits."""
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    decs = [mk(saf.AmbiDec, 512, 7, 29, 1, 1) for _ in range(nI)]
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    a = torch.rand(nI, nF, 64, 512, device="cuda", generator=g) * 2 - 1
    b = torch.rand(nI, nF, 64, 512, device="cuda", generator=g) * 2 - 1
    st = (nF * 64 * 512, 64 * 512, 512)

    def go(x, split=None):
        bt = saf.AmbiDecBatch(decs, nF)
        y = torch.zeros_like(x)
        if split is None:
            bt.process_ptr(x.data_ptr(), st, y.data_ptr(), st, nF)
        else:
            f0 = 0
            for n in split:
                bt.process_ptr(x[:, f0:].data_ptr(), st, y[:, f0:].data_ptr(), st, n)
                f0 += n
        torch.cuda.synchronize()
        return y

    ya, yb, yab = go(a), go(b), go(2.0 * a - 0.5 * b)
    lin = 2.0 * ya - 0.5 * yb
    assert float((yab - lin).norm() / lin.norm()) < 1e-6                    # linearity
    assert torch.equal(go(a, split=(1, nF // 2 - 1, nF // 2)), ya)                          # state carried across calls, bit-identical
    a2 = a.clone(); a2[5] = b[5]
    y2 = go(a2)
    assert torch.equal(y2[:5], ya[:5]) and torch.equal(y2[6:], ya[6:]) and torch.equal(y2[5], yb[5])   # instances independent
    assert torch.equal(ya[0], go(a[:1].contiguous().expand(nI, -1, -1, -1).contiguous())[17])          # same input -> same output
    saf.set_stream(None)


@pytest.mark.parametrize("F,order,preset,norm,chord", [(512, 7, 29, 1, 1), (256, 3, 21, 2, 1), (128, 1, 3, 3, 2), (1024, 5, 28, 2, 1)])
def test_ambi_dec_time_domain_path_equals_transform_path(saf, orc, F, order, preset, norm, chord):
    """Band-independent decoding (all bands use one matrix) takes the time-domain form: y = A x, then the transform-free
    analysis -> synthesis.  Same outputs as the three-kernel transform path and as the oracle; the two paths share their
    state, so they may alternate call by call (here: blocks 0-2 fast, 3-5 transform, 6-8 fast)."""
    from spatial_audio_framework_amd._lib import load
    L = load()
    nSH = (order + 1) ** 2

    def make(cls):
        d = cls(F)
        d.setNormType(norm); d.setChOrder(chord); d.setMasterDecOrder(order); d.setOutputConfigPreset(preset)
        d.setDecMethod(0, 1); d.setDecMethod(1, 1)
        d.initCodec(); d.init(48000); d.setDecOrderAllBands(order)
        return d

    nB = 9 if F >= 256 else 27                       # enough blocks to get past the 12-hop latency
    x = frames(400 + F, nSH, nB * F)
    o = make(orc.AmbiDec)
    nLS = o.getNumLoudspeakers()
    yo = np.concatenate([o.process(np.ascontiguousarray(x[:, b * F:(b + 1) * F]), nLS) for b in range(nB)], 1)
    outs = {}
    try:
        for name, sched in (("fast", [1] * nB), ("transform", [0] * nB), ("mixed", ([1, 1, 1, 0, 0, 0, 1, 1, 1] * 3)[:nB])):
            g = make(saf.AmbiDec)
            ys = []
            for b in range(nB):
                L.saf_hip_ambi_dec_setTimeDomainPath(sched[b])
                ys.append(g.process(np.ascontiguousarray(x[:, b * F:(b + 1) * F]), nLS))
            outs[name] = np.concatenate(ys, 1)
    finally:
        L.saf_hip_ambi_dec_setTimeDomainPath(1)
    assert np.abs(yo).max() > 0.05
    for name, y in outs.items():
        assert relrms(y, yo) < 2e-6, name
    assert relrms(outs["fast"], outs["transform"]) < 2e-6 and relrms(outs["mixed"], outs["transform"]) < 2e-6


# fftSizesToTest of test__saf_rfft (test/src/test__utilities_module.c:382-384)
SAF_RFFT_SIZES = [16, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1048576,
                  80, 160, 320, 640, 1280, 240, 480, 960, 1920, 3840, 7680, 15360, 30720]


@pytest.mark.parametrize("N", SAF_RFFT_SIZES + [2, 4, 6, 14, 98, 186])
def test_saf_rfft_reference_test_on_gpu(saf, orc, N):
    """The reference's own test__saf_rfft (test/src/test__utilities_module.c:374-412) against the GPU object:
    backward(forward(x)) == x within 1e-5 for every size it lists; plus the spectrum against the oracle (itself checked
    against the reference's KissFFT in test_oracle_cpu.py) and a few sizes with factors 7 and 31."""
    x = frames(N, 1, N)[0]
    f = saf.Rfft(N)
    X = f.forward(x)
    assert maxabs(f.backward(X), x) <= 1e-5
    ref = np.fft.rfft(x.astype(np.float64))
    assert np.abs(X - ref).max() <= 3e-6 * np.abs(ref).max()
    if N <= 65536:
        o = orc.RFFT(N)
        assert np.abs(X - o.forward(x)).max() <= 3e-6 * np.abs(ref).max()
        Xr = (np.random.default_rng(N).normal(size=N // 2 + 1) + 1j * np.random.default_rng(N + 1).normal(size=N // 2 + 1)).astype(np.complex64)
        assert maxabs(f.backward(Xr), o.backward(Xr)) <= 1e-5      # includes non-zero Im at DC / Nyquist: ignored by both


def test_afSTFT_synthesis_time_chunks_bit_identical(saf, orc):
    """Few channels x many hops: the synthesis grid is split along time (every chunk re-synthesises 16 hops to rebuild its
    overlap-add history).  One 300-hop call (chunked) == 100 + 37 + 163 hops (the first two unchunked) bit for bit, and
    both match the oracle; the state after the chunked call continues correctly."""
    x = frames(15, 3, 340 * 128)
    a, b, o = saf.AfSTFT(3, 3), saf.AfSTFT(3, 3), orc.AfSTFT(3, 3)
    A = a.forward(x)
    Ao = o.forward(x)
    assert relrms(A, Ao) < 1e-6
    ya = np.concatenate([a.backward(np.ascontiguousarray(A[:, :, :300])), a.backward(np.ascontiguousarray(A[:, :, 300:]))], axis=1)
    yb = np.concatenate([b.backward(np.ascontiguousarray(A[:, :, :100])), b.backward(np.ascontiguousarray(A[:, :, 100:137])),
                         b.backward(np.ascontiguousarray(A[:, :, 137:300])), b.backward(np.ascontiguousarray(A[:, :, 300:]))], axis=1)
    assert np.array_equal(ya, yb)
    assert relrms(ya, o.backward(A)) < 1e-6


def test_vbap2d_and_spread_ring_vs_oracle(saf, orc):
    """2-D VBAP tables and getSpreadSrcDirs3D of the library (host code) against the oracle; irregular layout."""
    ls = np.array([[-150, 0], [30, 0], [-30, 0], [110, 0], [-110, 0], [0, 0], [175, 0]], np.float32)
    assert np.array_equal(saf.findLsPairs(ls), orc.findLsPairs(ls))
    for res in (1, 5):
        g, n = saf.generateVBAPgainTable2D(ls, res)
        o, _ = orc.generateVBAPgainTable2D(ls, res)
        assert g.shape == o.shape and n == 7 and maxabs(g, o) < 2e-6
    az = np.random.default_rng(3).uniform(-180, 180, 50).astype(np.float32)
    assert maxabs(saf.generateVBAPgainTable2D_srcs(az, ls)[0], orc.generateVBAPgainTable2D_srcs(az, ls)[0]) < 2e-6
    for a, e, sp, ns, nr in ((0.3, -0.2, 40.0, 8, 1), (2.0, 1.565, 90.0, 8, 1), (-1.0, 0.4, 30.0, 6, 2)):
        assert maxabs(saf.getSpreadSrcDirs3D(a, e, sp, ns, nr), orc.getSpreadSrcDirs3D(a, e, sp, ns, nr)) < 2e-6


def test_host_pointer_calls_staged_copies_equal_zero_copy(saf, orc):
    """The one-block host-pointer entry points run their kernels on the pinned staging blocks (default) or copy them
    through device memory (saf_hip_setZeroCopyIO(0)): both give the same samples, bit for bit."""
    L = saf.load()
    F, order = 256, 3
    x = frames(21, 16, 10 * F)
    H = (np.random.default_rng(4).normal(size=(2, 5, 300)) / 8).astype(np.float32)
    xc = frames(22, 5, 8 * 128)
    outs = []
    for zc in (1, 0):
        L.saf_hip_setZeroCopyIO(zc)
        assert L.saf_hip_getZeroCopyIO() == zc
        d = saf.AmbiDec(F)
        d.setNormType(1); d.setChOrder(1); d.setMasterDecOrder(order); d.setOutputConfigPreset(21)
        d.setDecMethod(0, 2); d.setDecMethod(1, 4); d.initCodec(); d.init(48000)
        yd = np.concatenate([d.process(np.ascontiguousarray(x[:, i * F:(i + 1) * F]), 24) for i in range(10)], 1)
        e = saf.AmbiEnc(F); e.init(48000); e.setOutputOrder(2); e.setNumSources(5)
        ye = np.concatenate([e.process(np.ascontiguousarray(x[:5, i * F:(i + 1) * F]), 9) for i in range(4)], 1)
        mc = saf.MatrixConv(128, H, 1)
        yc = np.concatenate([mc.apply(np.ascontiguousarray(xc[:, i * 128:(i + 1) * 128])) for i in range(8)], 1)
        a = saf.AfSTFT(4, 4)
        ya = a.backward(a.forward(np.ascontiguousarray(x[:4, :2048])))
        outs.append((yd, ye, yc, ya))
    L.saf_hip_setZeroCopyIO(1)
    for u, v in zip(*outs):
        assert np.abs(u).max() > 1e-3 and np.array_equal(u, v)
