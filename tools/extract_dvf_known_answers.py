#!/usr/bin/env python3
"""Copy the NUMERIC known-answer tables (inputs and expected outputs, no code) of the reference's DVF unit tests into
tests/golden/dvf_known_answers.json.

Source (relative to /root/reference): test/src/test__utilities_module.c
  :1114-1190  12 first-order DVF filters: coefficients, magnitude and phase at 10 frequencies (test__evalIIRTransferFunction)
  :1304-1345  calcDVFShelfParams: g0 / gInf / fc for 5 distances x 19 table angles
  :1348-1396  interpDVFShelfParams: 5 distances x 6 angles
  :1398-1440  dvfShelfCoeffs: b0 / b1 / a1 for 5 distances x 6 angles at 44.1 kHz
Run in the build container only:  python tools/extract_dvf_known_answers.py [/root/reference]
"""
import json
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parents[1] / "tests" / "golden" / "dvf_known_answers.json"
src = (REF / "test/src/test__utilities_module.c").read_text()


def numbers(body):
    return [float(x.rstrip("f")) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?f?", body)]


def table(name, start=0):
    m = re.compile(r"\b%s(?:\[\d+\])+\s*=\s*\{" % re.escape(name)).search(src, start)
    i = m.end(); depth = 1
    while depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0); i += 1
    body = src[m.end():i - 1]
    rows = re.findall(r"\{([^{}]*)\}", body)
    return [numbers(r) for r in rows] if rows else numbers(re.sub(r"/\*.*?\*/", "", body))


tf = src.index("void test__evalIIRTransferFunction")
out = {
    "_comment": "known-answer data of the reference's DVF unit tests (test/src/test__utilities_module.c:1114-1190, 1304-1440); "
                "written by tools/extract_dvf_known_answers.py",
    "iir": {"fs": numbers(re.search(r"float fs = ([^;]*);", src[tf:]).group(1))[0],
            "freqs": table("freqs", tf), "a": table("as_dvf_f", tf), "b": table("bs_dvf_f", tf),
            "mags": table("mags_dvf", tf), "phases": table("phases_dvf", tf),
            "tol": {"mag_dB": numbers(re.search(r"magToleranceDb = ([^;]*);", src[tf:]).group(1))[0],
                    "errScale": (lambda v: v[0] / v[1])(numbers(re.search(r"errScale = ([^;]*);", src[tf:]).group(1))),
                    "phase": (lambda v: v[0] * v[1])(numbers(re.search(r"phaseTolerance = ([^;]*);", src[tf:]).group(1)))}},
}
t1 = src.index("void test__dvf_calcDVFShelfParams")
out["shelf_params"] = {"rho": table("rho", t1), "g0": table("g0_ref", t1), "gInf": table("gInf_ref", t1), "fc": table("fc_ref", t1),
                       "tol": 0.00001, "tol_fc": 0.1}
t2 = src.index("void test__dvf_interpDVFShelfParams")
out["interp_params"] = {"rho": table("rho", t2), "theta": table("theta", t2), "iG0": table("iG0_ref", t2), "iGInf": table("iGInf_ref", t2),
                        "iFc": table("iFc_ref", t2), "tol": 0.0001, "tol_fc": 0.01}
t3 = src.index("void test__dvf_dvfShelfCoeffs")
out["shelf_coeffs"] = {"rho": table("rho", t3), "theta": table("theta", t3), "fs": 44100, "b0": table("b0_ref", t3), "b1": table("b1_ref", t3),
                       "a1": table("a1_ref", t3), "tol": 0.00001}
OUT.write_text(json.dumps(out, indent=0))
print({k: (list(v.keys()) if isinstance(v, dict) else "") for k, v in out.items()})
