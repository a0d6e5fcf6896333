#!/usr/bin/env python3
"""Copy the NUMERIC data (inputs, expected outputs, tolerances; no code) of reference unit tests that pin functions of the
hot path into tests/golden/reference_vectors.json:

  test/src/test__sh_module.c:170-229        test__getSHrotMtxReal: yaw/pitch/roll, the 25 x 25 matrix of the MATLAB getSHrotMtx(), tolerance
  test/src/test__hoa_module.c:106-168       test__truncationEQ: configuration constants and bounds
  test/src/test__utilities_module.c:681     test__getVoronoiWeights: the t-design direction tables it loops over
                                            (framework/modules/saf_utilities/saf_utility_loudspeaker_presets.c, degrees 3..21), tolerance
  test/src/test__utilities_module.c:123     test__delaunaynd: its five point sets (the reference test asserts nothing)
  test/src/test__utilities_module.c:170     test__quaternion: iteration count and tolerances

Run in the build container only:  python tools/extract_reference_vectors.py [/root/reference]
"""
import json
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parents[1] / "tests" / "golden" / "reference_vectors.json"
NUM = r"-?\d+\.?\d*(?:[eE]-?\d+)?"


def numbers(body):
    return [float(x) for x in re.findall(NUM, re.sub(r"(?<=\d)f\b", "", body))]


def braces(src, start):
    """text between the brace that opens at/after `start` and its match"""
    i = src.index("{", start); depth = 1; j = i + 1
    while depth:
        depth += {"{": 1, "}": -1}.get(src[j], 0); j += 1
    return src[i + 1:j - 1]


def rows(body):
    return [numbers(r) for r in re.findall(r"\{([^{}]*)\}", body)]


sh = (REF / "test/src/test__sh_module.c").read_text()
t = sh.index("void test__getSHrotMtxReal")
ypr = numbers(re.search(r"yawPitchRoll2Rzyx\(([^;]*?), 0, Rzyx\);\s*getSHrotMtxReal\(Rzyx, FLATTEN2D\(Mrot\), order\);\s*double Mrot_ref", sh[t:]).group(1))
M = rows(braces(sh, sh.index("Mrot_ref[25][25]", t)))
assert len(M) == 25 and all(len(r) == 25 for r in M)
out = {"_comment": "numeric data of reference unit tests; written by tools/extract_reference_vectors.py",
       "getSHrotMtxReal": {"source": "test/src/test__sh_module.c:170-229", "identity_order": 22, "order": 4, "yaw_pitch_roll_rad": ypr,
                           "Mrot_ref": M, "tol": numbers(re.search(r"acceptedTolerance = ([^;]*);", sh[t:]).group(1))[0]}}

hoa = (REF / "test/src/test__hoa_module.c").read_text()
t = hoa.index("void test__truncationEQ")
g = lambda name: numbers(re.search(r"\b%s = ([^;]*);" % name, hoa[t:]).group(1))[0]
out["truncationEQ"] = {"source": "test/src/test__hoa_module.c:106-168", "order_truncated": int(g("order_truncated")), "order_target": int(g("order_target")),
                       "softThreshold": g("softThreshold"), "enableMaxRE": int(g("enableMaxRE")), "fs": g("fs"), "nBands": int(g("nBands")),
                       "r": g("r"), "c": g("c"), "gain0_bound": 2.0e-6, "gainDB_low": -2.0e-6, "gainDB_high_offset": 6.0}

ut = (REF / "test/src/test__utilities_module.c").read_text()
t = ut.index("void test__getVoronoiWeights")
pres = (REF / "framework/modules/saf_utilities/saf_utility_loudspeaker_presets.c").read_text()
npts = [int(v) for v in numbers(braces(pres, pres.index("__Tdesign_nPoints_per_degree[21]")))]
tds = {}
for td in range(2, 21):                         # the loop bounds of the reference test: handles 2..20 = degrees 3..21
    deg = td + 1
    R = rows(braces(pres, pres.index("__Tdesign_degree_%d_dirs_deg[" % deg)))
    assert len(R) == npts[td] and all(len(r) == 2 for r in R), (deg, len(R), npts[td])
    tds[str(deg)] = R
out["getVoronoiWeights"] = {"source": "test/src/test__utilities_module.c:681-733", "tol": numbers(re.search(r"acceptedTolerance = ([^;]*);", ut[t:]).group(1))[0],
                            "nIterations": int(numbers(re.search(r"nIterations = ([^;]*);", ut[t:]).group(1))[0]), "tdesign_dirs_deg": tds}

t = ut.index("void test__delaunaynd")
sets = {}
for name in ("three_xy", "four_xy", "square_xy", "cube_xyz", "cube_xyz2"):
    sets[name] = rows(braces(ut, ut.index(name + "[", t)))
out["delaunaynd"] = {"source": "test/src/test__utilities_module.c:123-168 (the reference test asserts nothing: 'copy the mesh indices into e.g. Matlab, plot, and see')", "points": sets}

t = ut.index("void test__quaternion")
out["quaternion"] = {"source": "test/src/test__utilities_module.c:170-204", "iterations": 1000, "tol_rotation": 1e-3, "tol_euler_deg": 1e-2,
                     "problem_case_wxyz": [0.0, 0.0000563298236, 0.947490811, -0.319783032]}
OUT.write_text(json.dumps(out))
print({k: (list(v.keys()) if isinstance(v, dict) else "") for k, v in out.items()}, OUT.stat().st_size, "bytes")
