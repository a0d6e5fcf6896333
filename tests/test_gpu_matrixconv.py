"""saf_matrixConv on the GPU against the CPU oracle, the golden fixture and direct time-domain
convolution — needs an MI355X:  python -m pytest tests -m gpu

The reference's own test (test__saf_matrixConv, test/src/test__utilities_module.c:330-372) is a smoke test
without an assertion on the output, so numeric parity is pinned by (a) the oracle restatement of
saf_utility_matrixConv.c:165-236 and (b) the identity "zero-latency linear convolution" evaluated in float64.
Tolerance: 1e-5 relative RMS (north star); measured ~2e-7.
"""
from pathlib import Path

import numpy as np
import pytest

from util import frames, relrms

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
TOL = 1e-5


def direct(H, x):
    nOut, nIn, L = H.shape
    y = np.zeros((nOut, x.shape[1]), np.float64)
    for o in range(nOut):
        for i in range(nIn):
            y[o] += np.convolve(x[i].astype(np.float64), H[o, i].astype(np.float64))[:x.shape[1]]
    return y


def run(mc, x, hop):
    return np.concatenate([mc.apply(np.ascontiguousarray(x[:, b * hop:(b + 1) * hop])) for b in range(x.shape[1] // hop)], 1)


@pytest.mark.parametrize("part", [1, 0])
@pytest.mark.parametrize("hop,L,nIn,nOut", [(64, 96, 3, 2), (128, 128, 4, 3), (96, 100, 2, 2), (256, 1000, 5, 2), (32, 7, 1, 1), (4096, 5000, 2, 3), (4, 9, 2, 1), (64, 96, 40, 2), (128, 300, 64, 3)])
def test_matrixconv_vs_oracle_and_direct(saf, orc, part, hop, L, nIn, nOut):
    H = (np.random.default_rng(hop + L).normal(size=(nOut, nIn, L)) / 8).astype(np.float32)
    x = frames(hop * 3 + L, nIn, 9 * hop)
    yg = run(saf.MatrixConv(hop, H, part), x, hop)
    yo = run(orc.MatrixConv(hop, H, part), x, hop)
    assert relrms(yg, yo) < TOL
    assert relrms(yg, direct(H, x)) < TOL


def test_reference_smoke_test_shape_on_gpu(saf):
    """test__saf_matrixConv (test__utilities_module.c:330-372): 32 -> 40 channels, 512 taps, block 2048, partitioned;
    the reference only runs it — here the result is also checked against direct convolution on a few outputs."""
    nIn, nOut, L, hop = 32, 40, 512, 2048
    H = (np.random.default_rng(1).normal(size=(nOut, nIn, L)) / 16).astype(np.float32)
    x = frames(11, nIn, 3 * hop)
    y = run(saf.MatrixConv(hop, H, 1), x, hop)
    assert relrms(y[:3], direct(H[:3], x)) < TOL


def test_golden_matrixconv(saf):
    H = (np.random.default_rng(5).normal(size=(2, 3, 96)) / 8).astype(np.float32)
    y = run(saf.MatrixConv(64, H, 1), frames(303, 3, 5 * 64), 64)
    assert relrms(y, np.load(GOLD / "matrixconv_small.npz")["out"]) < 2e-6


def test_matrixconv_cfg3b_batched_device_entry(saf, orc):
    """BASELINE configs[2] convolver shape: 256 inputs -> 2 outputs, 1024-tap filters, hop 512, partitioned.
    Device-pointer entry with several blocks per call == the oracle block by block; a different split of the same
    stream gives the same samples."""
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    nIn, nOut, L, hop, nB = 256, 2, 1024, 512, 6
    H = (np.random.default_rng(3).normal(size=(nOut, nIn, L)) / 32).astype(np.float32)
    x = frames(5, nIn, nB * hop)
    yo = run(orc.MatrixConv(hop, H, 1), x, hop)
    d_x = torch.from_numpy(x).cuda()

    def go(split):
        mc = saf.MatrixConv(hop, H, 1, maxBlocks=max(split))
        d_y = torch.zeros(nOut, nB * hop, device="cuda")
        b0 = 0
        for n in split:
            mc.apply_dev(d_x[:, b0 * hop:].data_ptr(), (nB * hop, hop), d_y[:, b0 * hop:].data_ptr(), (nB * hop, hop), n)
            b0 += n
        torch.cuda.synchronize()
        return d_y.cpu().numpy()
    ya, yb = go((6,)), go((1, 2, 3))
    assert relrms(ya, yo) < TOL and relrms(ya, direct(H, x)) < TOL
    assert relrms(yb, ya) < 1e-6
    saf.set_stream(None)


# ----------------------------------------------------------------------------------------------------------------
# saf_multiConv / saf_TVConv (saf_utility_matrixConv.c:237-620).  The reference only smoke-runs them
# (no known answers): pinned by the oracle restatement and by float64 direct convolution ("parity unpinned" otherwise).
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("part", [1, 0])
@pytest.mark.parametrize("hop,L,nCH", [(64, 200, 5), (128, 128, 64), (96, 50, 3), (512, 3000, 8)])
def test_multiconv_vs_oracle_and_direct(saf, orc, part, hop, L, nCH):
    H = (np.random.default_rng(hop + L).normal(size=(nCH, L)) / 8).astype(np.float32)
    x = frames(hop + L + 1, nCH, 7 * hop)
    yg = run(saf.MultiConv(hop, H, part), x, hop)
    yo = run(orc.MultiConv(hop, H, part), x, hop)
    ref = np.stack([np.convolve(x[c].astype(np.float64), H[c].astype(np.float64))[:x.shape[1]] for c in range(nCH)])
    assert relrms(yg, yo) < TOL and relrms(yg, ref) < TOL


def test_multiconv_batched_device_entry(saf, orc):
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    hop, L, nCH, nB = 256, 700, 16, 6
    H = (np.random.default_rng(8).normal(size=(nCH, L)) / 8).astype(np.float32)
    x = frames(21, nCH, 2 * nB * hop)
    g, o = saf.MultiConv(hop, H, 1, maxBlocks=nB), orc.MultiConv(hop, H, 1)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros_like(d_in)
    for call in range(2):
        off = call * nB * hop * 4
        g.apply_dev(d_in.data_ptr() + off, (2 * nB * hop, hop), d_out.data_ptr() + off, (2 * nB * hop, hop), nB)
    torch.cuda.synchronize()
    assert relrms(d_out.cpu().numpy(), run(o, x, hop)) < TOL
    saf.set_stream(None)


@pytest.mark.parametrize("hop,L,nIR,nOut", [(64, 150, 3, 2), (128, 1024, 6, 4), (96, 96, 2, 1)])
def test_tvconv_vs_oracle(saf, orc, hop, L, nIR, nOut):
    """IR index switching every few blocks, including back-to-back switches; host-pointer entry, one block per call"""
    rng = np.random.default_rng(hop)
    H = (rng.normal(size=(nIR, nOut, L)) / 8).astype(np.float32)
    nB = 14
    x = frames(5, 1, nB * hop)[0]
    idx = [0, 0, 1, 1, 1, nIR - 1, 0, 1, 1, 0, 0, 0, nIR - 1, nIR - 1]
    g, o = saf.TVConv(hop, H, 0), orc.TVConv(hop, H, 0)
    yg = np.concatenate([g.apply(x[b * hop:(b + 1) * hop], idx[b]) for b in range(nB)], 1)
    yo = np.concatenate([o.apply(x[b * hop:(b + 1) * hop], idx[b]) for b in range(nB)], 1)
    assert np.abs(yo).max() > 0.1 and relrms(yg, yo) < TOL
    # closed form: block b cross-fades from the convolution with IR idx[b-2] to the one with IR idx[b-1]
    full = np.stack([[np.convolve(x.astype(np.float64), H[i, c].astype(np.float64))[:nB * hop] for c in range(nOut)] for i in range(nIR)])
    fi = np.arange(hop) / (hop - 1.0)
    exp = np.zeros_like(yo, dtype=np.float64)
    for b in range(nB):
        s = slice(b * hop, (b + 1) * hop)
        exp[:, s] = full[idx[b - 1] if b >= 1 else 0][:, s] * fi + full[idx[b - 2] if b >= 2 else 0][:, s] * (1.0 - fi)
    assert relrms(yg, exp) < TOL


def test_tvconv_batched_device_entry(saf, orc):
    import torch
    saf.set_stream(torch.cuda.current_stream().cuda_stream)
    hop, L, nIR, nOut, nB = 512, 4096, 10, 4, 8
    rng = np.random.default_rng(77)
    H = (rng.normal(size=(nIR, nOut, L)) / 16).astype(np.float32)
    x = frames(6, 1, 2 * nB * hop)[0]
    idx = [int(v) for v in rng.integers(0, nIR, 2 * nB)]
    g, o = saf.TVConv(hop, H, 3, maxBlocks=nB), orc.TVConv(hop, H, 3)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros(nOut, 2 * nB * hop, device="cuda")
    for call in range(2):
        off = call * nB * hop * 4
        g.apply_dev(d_in.data_ptr() + off, hop, d_out.data_ptr() + off, (2 * nB * hop, hop), idx[call * nB:(call + 1) * nB], nB)
    torch.cuda.synchronize()
    yo = np.concatenate([o.apply(x[b * hop:(b + 1) * hop], idx[b]) for b in range(2 * nB)], 1)
    assert relrms(d_out.cpu().numpy(), yo) < TOL
    saf.set_stream(None)


@pytest.mark.parametrize("matrix", [1, 0])
def test_conv_example_wrappers_vs_oracle(saf, orc, matrix):
    """matrixconv / multiconv example operators (examples/src/matrixconv, multiconv): sample-wise FIFO, host block clamped to
    512..8192, output one block late, ragged call sizes, a re-init (partitioning switched) mid-stream."""
    rng = np.random.default_rng(4 + matrix)
    nIn, nOut, L = (3, 2, 300) if matrix else (5, 5, 700)
    H = (rng.normal(size=(nOut, nIn * L)) / 8).astype(np.float32) if matrix else (rng.normal(size=(nIn, L)) / 8).astype(np.float32)
    g, o = saf.ConvExample(matrix), orc.ConvExample(matrix)
    for c in (g, o):
        c.setNumInputChannels(nIn); c.setFilters(H); c.setEnablePart(1); c.init(48000, 600)
    assert g.getProcessingDelay() == o.getProcessingDelay() == 600 and g.getFilterLength() == L
    x = frames(70 + matrix, nIn, 6000)
    pos, yg, yo = 0, [], []
    for n in (100, 500, 600, 1234, 66, 900, 600):
        if pos == 2500:
            for c in (g, o):
                c.setEnablePart(0)
        blk = np.ascontiguousarray(x[:, pos:pos + n]); pos += n
        yg.append(g.process(blk, nOut + 1)); yo.append(o.process(blk, nOut + 1))
    yg, yo = np.concatenate(yg, 1), np.concatenate(yo, 1)
    assert np.all(yg[nOut] == 0) and np.abs(yo).max() > 0.1
    assert relrms(yg, yo) < TOL


def test_tvconv_example_wrapper_vs_oracle(saf, orc):
    """tvconv example operator: FIFO around saf_TVConv, IR set = listener position nearest to the moving target; IRs and
    positions installed directly (the reference reads them from a SOFA file)."""
    rng = np.random.default_rng(21)
    nPos, nIr, L = 6, 3, 900
    irs = (rng.normal(size=(nPos, nIr, L)) / 16).astype(np.float32)
    pos = np.stack([np.linspace(0, 5, nPos), np.zeros(nPos), np.full(nPos, 1.5)], 1).astype(np.float32)
    g, o = saf.TvConvExample(), orc.TvConvExample()
    for c in (g, o):
        c.setIRsAndPositions(irs, pos); c.init(48000, 512)
    assert g.getNumListenerPositions() == nPos and g.getNumOutputChannels() == nIr and g.getProcessingDelay() == 512
    x = frames(88, 1, 7000)[0]
    p0, yg, yo = 0, [], []
    for k, n in enumerate((300, 724, 512, 1000, 48, 2000, 1024, 1392)):
        for c in (g, o):
            c.setTargetPosition(float(0.7 * k), 0)
        assert g.getListenerPositionIdx() == o.getListenerPositionIdx()
        yg.append(g.process(x[p0:p0 + n], nIr + 1)); yo.append(o.process(x[p0:p0 + n], nIr + 1)); p0 += n
    yg, yo = np.concatenate(yg, 1), np.concatenate(yo, 1)
    assert np.all(yg[nIr] == 0) and np.abs(yo).max() > 0.05 and relrms(yg, yo) < TOL
