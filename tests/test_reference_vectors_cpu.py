"""The CPU oracle against the data the reference's own unit tests hold for functions of the hot path
(tests/reference_restated.py, tests/golden/reference_vectors.json written by tools/extract_reference_vectors.py)."""
import numpy as np

import reference_restated as RR


def test_reference_getSHrotMtxReal(orc):
    assert RR.getSHrotMtxReal(orc) < 1e-5


def test_reference_truncationEQ(orc):
    RR.truncationEQ(orc)


def test_reference_getVoronoiWeights(orc):
    RR.getVoronoiWeights(orc)


def test_reference_quaternion(orc):
    RR.quaternion(orc)


def test_reference_delaunaynd_point_sets_triangulate(orc):
    """test__delaunaynd asserts nothing (its author plots the meshes).  delaunaynd itself is not on the hot path — the Voronoi
    weights come from the spherical hull of the directions — so what is checked here is that hull on the test's 3-D point sets
    projected to the sphere: the 8 cube corners give 12 faces (6 squares split in two) that tile the sphere exactly once."""
    pts = RR.delaunay_point_sets()
    cube = pts["cube_xyz"]
    d = np.stack([np.degrees(np.arctan2(cube[:, 1], cube[:, 0])), np.degrees(np.arcsin(cube[:, 2] / np.linalg.norm(cube, axis=1)))], 1).astype(np.float32)
    _, faces = orc.findLsTriplets(d)
    assert len(faces) == 12 and sorted(set(np.asarray(faces).ravel().tolist())) == list(range(8))
    w = orc.getVoronoiWeights(d)
    assert abs(float(w.sum()) - 4 * np.pi) < 1e-4 and np.abs(w - w[0]).max() < 1e-4          # the cube is vertex-transitive
    assert pts["square_xy"].shape == (26, 2) and pts["cube_xyz2"].shape == (9, 3) and pts["three_xy"].shape == (3, 2) and pts["four_xy"].shape == (4, 2)
