"""The widened operators on the GPU against the committed golden fixtures — needs an MI355X.

tests/golden/{panner,conv,ambi_dec_bin,pmaps}_small.npz are oracle outputs on seeded inputs (tests/golden/make_golden.py,
checked against the oracle itself on the CPU by test_oracle_cpu.py::test_golden_fixtures_match_oracle); here the HIP path
runs the same scenarios.  Tolerance 1e-5 relative RMS (north star); the sub-space maps through their reciprocals.
"""
from pathlib import Path

import numpy as np
import pytest

import sys
sys.path.insert(0, str(Path(__file__).parent / "golden"))
import make_golden as mg
from util import relrms

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
TOL = 1e-5


def test_golden_panner(saf):
    assert relrms(mg.panner_scenario(saf.Panner), np.load(GOLD / "panner_small.npz")["out"]) < TOL


def test_golden_multiconv_tvconv(saf):
    ref = np.load(GOLD / "conv_small.npz")
    got = mg.conv_scenarios(saf.MultiConv, saf.TVConv)
    for k in ("multi_part1", "multi_part0", "tv"):
        assert relrms(got[k], ref[k]) < TOL, k


def test_golden_ambi_dec_binaural(saf):
    assert relrms(mg.ambi_dec_bin_scenario(saf.AmbiDec), np.load(GOLD / "ambi_dec_bin_small.npz")["out"]) < TOL


def test_golden_ambi_bin(saf):
    assert relrms(mg.ambi_bin_scenario(saf.AmbiBin), np.load(GOLD / "ambi_bin_small.npz")["out"]) < 3e-5      # MagLS phase chain


def test_golden_activity_maps(saf, orc):
    ref = np.load(GOLD / "pmaps_small.npz")
    order = 3
    Yg = (orc.getRSH(order, orc.table("Tdesign_degree_21_dirs_deg")) / 16).astype(np.float32)
    Cx = ref["Cx"]
    assert relrms(saf.generateMVDRmap(order, Cx, Yg), ref["mvdr"]) < TOL
    assert relrms(saf.generateCroPaCLCMVmap(order, Cx, Yg), ref["cropac"]) < 5e-5
    for k, gen in (("inv_music", saf.generateMUSICmap), ("inv_minnorm", saf.generateMinNormMap)):
        inv = 1.0 / gen(order, Cx, Yg, 2)
        assert np.abs(inv - ref[k]).max() < 5e-6 * ref[k].max() + 3e-8, k
        assert set(np.argsort(inv)[:2]) == {139, 204}


def test_golden_rotator_beamformer_ambi_drc(saf):
    """the operators beyond SURVEY 8f on their committed fixtures"""
    assert relrms(mg.rotator_scenario(saf.Rotator), np.load(GOLD / "rotator_small.npz")["out"]) < 2e-6
    assert relrms(mg.beamformer_scenario(saf.Beamformer), np.load(GOLD / "beamformer_small.npz")["out"]) < 2e-6
    assert relrms(mg.ambi_drc_scenario(saf.AmbiDrc), np.load(GOLD / "ambi_drc_small.npz")["out"]) < TOL
    assert relrms(mg.binauraliser_nf_scenario(saf.BinauraliserNF), np.load(GOLD / "binauraliser_nf_small.npz")["out"]) < TOL
