"""Reference unit tests that pin functions of the hot path, restated once and run against BOTH implementations:
the CPU oracle (tests/test_reference_vectors_cpu.py) and libsaf_hip through its C-ABI (tests/test_gpu_reference_vectors.py).
Data (inputs, expected outputs, tolerances) = tests/golden/reference_vectors.json, numbers copied from the reference's test
sources by tools/extract_reference_vectors.py.  `m` is a module with the functions of oracle/oracle.py / api.py."""
import json
from pathlib import Path

import numpy as np

REF = json.loads((Path(__file__).resolve().parent / "golden" / "reference_vectors.json").read_text())
FOURPI = 4.0 * np.pi


def getSHrotMtxReal(m):
    """test__getSHrotMtxReal (test/src/test__sh_module.c:170-229): identity at order 22; the MATLAB getSHrotMtx() matrix at order 4"""
    r = REF["getSHrotMtxReal"]
    tol = r["tol"]
    M = m.getSHrotMtxReal(m.yawPitchRoll2Rzyx(0.0, 0.0, 0.0), r["identity_order"])
    assert np.abs(M - np.eye(M.shape[0], dtype=np.float32)).max() <= tol
    y, p, ro = r["yaw_pitch_roll_rad"]
    M = m.getSHrotMtxReal(m.yawPitchRoll2Rzyx(y, p, ro), r["order"])
    ref = np.asarray(r["Mrot_ref"], np.float64).astype(np.float32)
    assert M.shape == ref.shape and np.abs(M - ref).max() <= tol, float(np.abs(M - ref).max())
    return float(np.abs(M - ref).max())


def truncationEQ(m):
    """test__truncationEQ (test/src/test__hoa_module.c:106-168): max-rE order weighting normalised to the zeroth order; gain at DC
    within 2e-6 of 1; every band between 0 dB and softThreshold + 6 dB"""
    c = REF["truncationEQ"]
    nB = c["nBands"]
    f = np.arange(nB, dtype=np.float64) * c["fs"] / (2.0 * (nB - 1.0))
    kr = 2.0 * np.pi / c["c"] * f * c["r"]
    N = c["order_truncated"]
    w = m.beamWeights(3, N).astype(np.float32)                               # beamWeightsMaxEV
    for n in range(N + 1):
        w[n] = np.float32(w[n] / np.sqrt(np.float32((2 * n + 1) / np.float32(FOURPI))))
    w = (w / w[0]).astype(np.float32)
    g = m.truncationEQ(w, N, c["order_target"], kr, c["softThreshold"])
    assert g[0] - 1.0 < c["gain0_bound"]
    gdb = 20.0 * np.log10(g.astype(np.float32))
    assert np.all(gdb > c["gainDB_low"]) and np.all(gdb < c["softThreshold"] + c["gainDB_high_offset"] + c["gainDB_low"])
    return g


def getVoronoiWeights(m, seed=1234):
    """test__getVoronoiWeights (test/src/test__utilities_module.c:681-733): on the t-designs of degree 3..21 the weights sum to
    4 pi and are all equal (tolerance 0.01); on 100 random point sets of 10..200 directions (azimuth AND elevation drawn from
    -180..180 degrees, as the reference does) they sum to 4 pi"""
    v = REF["getVoronoiWeights"]
    tol = v["tol"]
    for deg, dirs in v["tdesign_dirs_deg"].items():
        w = m.getVoronoiWeights(np.asarray(dirs, np.float32))
        assert abs(float(w.sum()) - FOURPI) <= tol, deg
        assert np.abs(w - w[0]).max() <= tol, deg
    rng = np.random.default_rng(seed)
    for _ in range(v["nIterations"]):
        n = int(rng.random() * 190.0 + 10.0)
        d = (rng.random((n, 2)) * 2.0 - 1.0).astype(np.float32) * np.float32(180.0)
        w = m.getVoronoiWeights(d)
        assert abs(float(w.sum()) - FOURPI) <= tol


def quaternion(m, seed=99):
    """test__quaternion (test/src/test__utilities_module.c:170-204): quaternion -> rotation matrix -> quaternion -> rotation matrix
    agrees within 1e-3; quaternion2euler / euler2Quaternion (degrees, yaw-pitch-roll) are inverses within 1e-2 degrees.  (The
    'problem case' the reference keeps commented out — w = 0 exactly — fails its own assertion with the reference's formulas, here
    too; it is returned, not asserted.)"""
    q0 = REF["quaternion"]
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(q0["iterations"]):
        q = (rng.random(4) * 2.0 - 1.0).astype(np.float32)
        cases.append((q / np.float32(np.sqrt(float((q.astype(np.float64) ** 2).sum())))).astype(np.float32))
    for q in cases:
        R = m.quaternion2rotationMatrix(q)
        q1 = m.rotationMatrix2quaternion(R)
        R2 = m.quaternion2rotationMatrix(q1)
        assert np.abs(R - R2).max() < q0["tol_rotation"], q
        ypr = m.quaternion2euler(q1, True, 2)
        q2 = m.euler2Quaternion(float(ypr[0]), float(ypr[1]), float(ypr[2]), True, 2)
        ypr2 = m.quaternion2euler(q2, True, 2)
        assert np.abs(ypr2 - ypr).max() < q0["tol_euler_deg"], (q, ypr, ypr2)
    qp = np.asarray(q0["problem_case_wxyz"], np.float32)
    R = m.quaternion2rotationMatrix(qp)
    return float(np.abs(R - m.quaternion2rotationMatrix(m.rotationMatrix2quaternion(R))).max())


def delaunay_point_sets():
    """the five point sets of test__delaunaynd (test/src/test__utilities_module.c:123-168); the reference asserts nothing on them"""
    return {k: np.asarray(v, np.float32) for k, v in REF["delaunaynd"]["points"].items()}
