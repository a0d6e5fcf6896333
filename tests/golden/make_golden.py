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
from util import frames, synth_hrirs  # noqa: E402


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
    out["panner_small"] = {"out": panner_scenario(O.Panner)}
    out["conv_small"] = conv_scenarios(O.MultiConv, O.TVConv)
    out["ambi_dec_bin_small"] = {"out": ambi_dec_bin_scenario(O.AmbiDec)}
    out["pmaps_small"] = pmaps_scenario(O)
    out["ambi_bin_small"] = {"out": ambi_bin_scenario(O.AmbiBin)}
    out["rotator_small"] = {"out": rotator_scenario(O.Rotator)}
    out["beamformer_small"] = {"out": beamformer_scenario(O.Beamformer)}
    out["ambi_drc_small"] = {"out": ambi_drc_scenario(O.AmbiDrc)}
    out["binauraliser_nf_small"] = {"out": binauraliser_nf_scenario(O.BinauraliserNF)}
    return out


def ambi_bin_scenario(cls, F=128, nB=20):
    """order 2 -> 2 ears, MagLS + max-rE (the reference's defaults), SN3D input, rotation switched on at block 10"""
    h, d = synth_hrirs()
    a = cls(F)
    a.setHRIRs(h, d, 48000); a.setInputOrderPreset(2); a.init(48000); a.initCodec()
    x = frames(808, 9, nB * F)
    ys = []
    for b in range(nB):
        if b == 10:
            a.setEnableRotation(1); a.setYaw(45.0); a.setPitch(10.0)
        ys.append(a.process(x[:, b * F:(b + 1) * F], 2))
    return np.concatenate(ys, 1)


def binauraliser_nf_scenario(cls, F=128, nB=20):
    """8 sources between 0.15 and 3.5 m, source 1 brought from 2 m to 0.2 m at block 6, source 0 sent to the far field at block 9,
    head rotation from block 12 on"""
    h, d = synth_hrirs()
    b = cls(F, 64)
    b.setHRIRs(h, d, 48000); b.init(48000); b.setNumSources(8); b.initCodec()
    for s in range(8):
        b.setSourceAzi_deg(s, float(45 * s - 170)); b.setSourceElev_deg(s, float(20 * (s % 4) - 30)); b.setSourceDist_m(s, float(0.15 + 0.48 * s))
    x = frames(909, 8, nB * F)
    ys = []
    for blk in range(nB):
        if blk == 6:
            b.setSourceDist_m(1, 0.2)
        if blk == 9:
            b.setSourceDist_m(0, 6.0)
        if blk == 12:
            b.setEnableRotation(1); b.setYaw(60.0); b.setRoll(-20.0)
        ys.append(b.process(x[:, blk * F:(blk + 1) * F], 2))
    return np.concatenate(ys, 1)


def panner_scenario(cls, F=128, nB=16):
    """7 sources (first directions of SphCovering-64) -> 5.x, DTT 0.3, a rotation from block 4 on, source 2 moved at block 8"""
    p = cls(F)
    p.setOutputConfigPreset(3); p.setInputConfigPreset(30); p.setNumSources(7)
    p.initCodec(); p.init(48000); p.setDTT(0.3); p.initCodec()
    x = frames(505, 7, nB * F)
    ys = []
    for b in range(nB):
        if b == 4:
            p.setYaw(25.0); p.setPitch(-10.0)
        if b == 8:
            p.setSourceAzi_deg(2, 123.0); p.setSourceElev_deg(2, 33.0)
        ys.append(p.process(x[:, b * F:(b + 1) * F], 5))
    return np.concatenate(ys, 1)


def conv_scenarios(MultiConv, TVConv):
    """multiConv: 4 channels x 150 taps, hop 64, both modes; TVConv: 1 -> 2 channels, 3 IR sets of 150 taps, index switches"""
    rng = np.random.default_rng(6)
    H = (rng.normal(size=(4, 150)) / 8).astype(np.float32)
    x = frames(606, 4, 8 * 64)
    res = {}
    for part in (1, 0):
        mc = MultiConv(64, H, part)
        res[f"multi_part{part}"] = np.concatenate([mc.apply(np.ascontiguousarray(x[:, b * 64:(b + 1) * 64])) for b in range(8)], 1)
    Ht = (rng.normal(size=(3, 2, 150)) / 8).astype(np.float32)
    tv = TVConv(64, Ht, 1)
    idx = [1, 1, 2, 2, 0, 1, 1, 1]
    res["tv"] = np.concatenate([tv.apply(x[0, b * 64:(b + 1) * 64], idx[b]) for b in range(8)], 1)
    return res


def ambi_dec_bin_scenario(cls, F=128, nB=20):
    """order 2 -> t-design(12) -> 2 ears, SAD, HRIR pre-processing on, synthetic 836-direction HRIR set (tests/util.py)"""
    h, d = synth_hrirs()
    a = cls(F)
    a.setHRIRs(h, d, 48000)
    a.setNormType(1); a.setChOrder(1); a.setMasterDecOrder(2); a.setOutputConfigPreset(20)
    a.setDecMethod(0, 1); a.setDecMethod(1, 1); a.setBinauraliseLSflag(1)
    a.init(48000); a.initCodec(); a.setDecOrderAllBands(2)
    x = frames(707, 9, nB * F)
    return np.concatenate([a.process(x[:, b * F:(b + 1) * F], 2) for b in range(nB)], 1)


def pmaps_scenario(O):
    """order-3 covariance of two plane waves + noise on the 240-point t-design: MVDR map, 1/MUSIC, 1/MinNorm, CroPaC map"""
    rng = np.random.default_rng(7)
    order, nSH = 3, 16
    grid = O.table("Tdesign_degree_21_dirs_deg")
    Yg = (O.getRSH(order, grid) / nSH).astype(np.float32)
    s = rng.normal(size=(2, 3000)) + 1j * rng.normal(size=(2, 3000))
    xs = O.getRSH(order, grid[[139, 204]]) @ s + 0.02 * (rng.normal(size=(nSH, 3000)) + 1j * rng.normal(size=(nSH, 3000)))
    Cx = (xs @ xs.conj().T / 3000).astype(np.complex64)
    return {"Cx": Cx, "mvdr": O.generateMVDRmap(order, Cx, Yg), "cropac": O.generateCroPaCLCMVmap(order, Cx, Yg),
            "inv_music": 1.0 / O.generateMUSICmap(order, Cx, Yg, 2), "inv_minnorm": 1.0 / O.generateMinNormMap(order, Cx, Yg, 2)}


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


def rotator_scenario(cls, F=64, nB=10, order=3):
    """order 3; rotation set by Euler angles before block 2, by quaternion before block 6"""
    r = cls(F); r.init(48000); r.setOrder(order)
    x = frames(505, (order + 1) ** 2, nB * F)
    ys = []
    for b in range(nB):
        if b == 2:
            r.setYaw(50.0); r.setPitch(-20.0); r.setRoll(15.0)
        if b == 6:
            r.setQuaternionW(0.7071068); r.setQuaternionX(0.0); r.setQuaternionY(0.7071068); r.setQuaternionZ(0.0)
        ys.append(r.process(np.ascontiguousarray(x[:, b * F:(b + 1) * F]), (order + 1) ** 2))
    return np.concatenate(ys, 1)


def beamformer_scenario(cls, F=128, nB=8, order=4):
    """order 4, SN3D input, 5 beams; hyper-cardioid, a beam moved before block 3, cardioid from block 5"""
    b = cls(F); b.init(48000); b.setBeamOrder(order); b.setNumBeams(5)
    x = frames(606, (order + 1) ** 2, nB * F)
    ys = []
    for k in range(nB):
        if k == 3:
            b.setBeamAzi_deg(1, 100.0); b.setBeamElev_deg(1, -40.0)
        if k == 5:
            b.setBeamType(1)
        ys.append(b.process(np.ascontiguousarray(x[:, k * F:(k + 1) * F]), 5))
    return np.concatenate(ys, 1)


def ambi_drc_scenario(cls, F=128, nB=48, order=2):
    """order 2; quiet / loud / quiet input, threshold -35 dB, ratio 6, knee 4 dB, 20 ms attack, 120 ms release"""
    d = cls(F)
    d.setInputPreset(order); d.setThreshold(-35.0); d.setRatio(6.0); d.setKnee(4.0); d.setAttack(20.0); d.setRelease(120.0); d.setInGain(3.0)
    d.init(48000)
    x = frames(707, (order + 1) ** 2, nB * F)
    env = np.ones(nB * F, np.float32); env[: nB * F // 3] = 0.02; env[2 * nB * F // 3:] = 0.05
    x = (x * env).astype(np.float32)
    return np.concatenate([d.process(np.ascontiguousarray(x[:, k * F:(k + 1) * F])) for k in range(nB)], 1)


if __name__ == "__main__":
    from oracle import oracle as O
    for name, arrays in generate(O).items():
        np.savez_compressed(HERE / f"{name}.npz", **arrays)
        print(name, {k: v.shape for k, v in arrays.items()})
