"""Generates tests/golden/*.npz from the CPU oracle on seeded inputs.

    python tests/golden/make_golden.py        # rewrites the fixtures

The reference itself cannot be built in this image (SURVEY §8c / DESIGN.md), so
these fixtures are outputs of the ORACLE (which is pinned by the reference's own
tests, see test_oracle_cpu.py) — they are a regression guard for both the oracle
and the HIP path, not an independent source of truth.  Inputs are regenerated
from the seeds, so only small outputs are stored.
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))
from util import frames  # noqa: E402


def ambi_dec_cfg(cls, F, order, preset, m0, m1, low_order=None):
    a = cls(F)
    a.setNormType(1); a.setChOrder(1); a.setMasterDecOrder(order); a.setOutputConfigPreset(preset)
    a.setDecMethod(0, m0); a.setDecMethod(1, m1)
    a.initCodec(); a.init(48000); a.setDecOrderAllBands(order)
    if low_order:
        for b in range(40, 133):
            a.setDecOrder(low_order, b)
    return a


def generate(O):
    out = {}
    # afSTFT: 3 in / 2 out, 12 frames of 256 with carried state; spectra of the last frame + synthesised signal
    st = O.AfSTFT(3, 2)
    x = frames(101, 3, 12 * 256)
    specs = [st.forward(x[:, i * 256:(i + 1) * 256]) for i in range(12)]
    y = np.concatenate([st.backward(s[:, :2, :]) for s in specs], 1)
    out["afstft_small"] = {"spec_last": specs[-1], "synth": y}
    # SH + decoders on SphCovering-9, order 2
    ls = O.table("SphCovering_9_dirs_deg")
    out["sh_decoders_small"] = {
        "rsh": O.getRSH(3, ls), "rsh_recur": O.getRSH_recur(3, ls),
        "sad": O.getLoudspeakerDecoderMtx(ls, 1, 2), "mmd": O.getLoudspeakerDecoderMtx(ls, 2, 2),
        "epad": O.getLoudspeakerDecoderMtx(ls, 3, 2), "allrad": O.getLoudspeakerDecoderMtx(ls, 4, 2),
        "faces": O.findLsTriplets(ls)[1].astype(np.int32),
    }
    # ambi_dec order 3 -> t-design(60)... use SphCovering-16, mixed decoders + per-band orders, 6 frames of 128
    d = ambi_dec_cfg(O.AmbiDec, 128, 3, 26, 1, 3, low_order=1)
    xin = frames(202, 16, 24 * 128)
    yo = np.concatenate([d.process(xin[:, i * 128:(i + 1) * 128], 16) for i in range(24)], 1)
    out["ambi_dec_small"] = {"out": yo}
    # matrixConv 3 -> 2, 96 taps, hop 64, 5 blocks
    H = (np.random.default_rng(5).normal(size=(2, 3, 96)) / 8).astype(np.float32)
    mc = O.MatrixConv(64, H, 1)
    xc = frames(303, 3, 5 * 64)
    out["matrixconv_small"] = {"out": np.concatenate([mc.apply(xc[:, i * 64:(i + 1) * 64]) for i in range(5)], 1)}
    out["ambi_enc_small"] = {"out": ambi_enc_scenario(O.AmbiEnc)}
    return out


def ambi_enc_scenario(cls, F=256, nFrames=8):
    """SURVEY Appendix D scenario: order 3, 3 sources, one gain 0.5, SN3D, post-scaling, a direction change before frame 3
    and a gain change before frame 5."""
    e = cls(F); e.init(48000)
    e.setOutputOrder(3); e.setNumSources(3)
    for i, (az, el) in enumerate([(30.0, 10.0), (-110.0, 45.0), (170.0, -30.0)]):
        e.setSourceAzi_deg(i, az); e.setSourceElev_deg(i, el)
    e.setSourceGain(1, 0.5)
    x = frames(404, 3, nFrames * F)
    ys = []
    for f in range(nFrames):
        if f == 3:
            e.setSourceAzi_deg(0, -75.0); e.setSourceElev_deg(2, 60.0)
        if f == 5:
            e.setSourceGain(2, 1.5)
        ys.append(e.process(x[:, f * F:(f + 1) * F], 18))
    return np.concatenate(ys, 1)


if __name__ == "__main__":
    from oracle import oracle as O
    for name, arrays in generate(O).items():
        np.savez_compressed(HERE / f"{name}.npz", **arrays)
        print(name, {k: v.shape for k, v in arrays.items()})
