"""libsaf_hip (through its C-ABI) against the data the reference's own unit tests hold for functions of the hot path —
needs an MI355X:  python -m pytest tests -m gpu.  Same restated tests as the CPU oracle runs (tests/reference_restated.py)."""
import numpy as np
import pytest

import reference_restated as RR

pytestmark = pytest.mark.gpu


def test_reference_getSHrotMtxReal_on_gpu(saf, orc):
    assert RR.getSHrotMtxReal(saf) < 1e-5
    # and the library agrees with the oracle on the same rotation at order 7 (what rotator / ambi_bin use)
    R = saf.yawPitchRoll2Rzyx(0.04, 0.54, -0.4)
    assert np.abs(saf.getSHrotMtxReal(R, 7) - orc.getSHrotMtxReal(R, 7)).max() < 2e-6


def test_reference_truncationEQ_on_gpu(saf, orc):
    g = RR.truncationEQ(saf)
    assert np.abs(g - RR.truncationEQ(orc)).max() < 1e-5 * float(np.abs(g).max())


def test_reference_getVoronoiWeights_on_gpu(saf):
    RR.getVoronoiWeights(saf)


def test_reference_quaternion_on_gpu(saf):
    RR.quaternion(saf)


def test_reference_delaunaynd_point_sets_triangulate_on_gpu(saf):
    pts = RR.delaunay_point_sets()
    cube = pts["cube_xyz"]
    d = np.stack([np.degrees(np.arctan2(cube[:, 1], cube[:, 0])), np.degrees(np.arcsin(cube[:, 2] / np.linalg.norm(cube, axis=1)))], 1).astype(np.float32)
    _, faces = saf.findLsTriplets(d)
    assert len(faces) == 12 and sorted(set(np.asarray(faces).ravel().tolist())) == list(range(8))
    w = saf.getVoronoiWeights(d)
    assert abs(float(w.sum()) - 4 * np.pi) < 1e-4 and np.abs(w - w[0]).max() < 1e-4
