#!/usr/bin/env python3
"""Extract NUMERIC DATA TABLES (no code) from the reference checkout into one
packed binary blob: spatial_audio_framework_amd/data/saf_tables.bin

Why: the afSTFT prototype filter is a MATLAB `firceqrip` design that cannot be
regenerated here, and loudspeaker / t-design direction sets are measured or
published coordinate lists.  Both are data the hot path needs bit-for-bit.
Only the numbers are taken; nothing of the reference's program text is kept.

Sources (relative to /root/reference):
  framework/resources/afSTFT/afSTFT_protoFilter.h:28,1494        prototype filters
  framework/resources/afSTFT/afSTFTlib.c:54-59                   measured band centre frequencies
  framework/modules/saf_utilities/saf_utility_loudspeaker_presets.c   direction sets
  framework/modules/saf_utilities/saf_utility_sensorarray_presets.c:329-343  mic order ranges

Blob format (little endian):
  magic  "SAFT" u32 version(1) u32 nTables
  per table: u32 nameLen, name bytes, u32 d0, u32 d1, float32 data[d0*d1]

Run in the build container only (the reference never travels):
  python tools/extract_tables.py [/root/reference]
"""
import re
import struct
import sys
from pathlib import Path

import numpy as np

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parents[1] / "spatial_audio_framework_amd" / "data" / "saf_tables.bin"

FILES = [
    "framework/resources/afSTFT/afSTFT_protoFilter.h",
    "framework/resources/afSTFT/afSTFTlib.c",
    "framework/modules/saf_utilities/saf_utility_loudspeaker_presets.c",
    "framework/modules/saf_utilities/saf_utility_sensorarray_presets.c",
]

# which tables the hot path (and its tests) need
WANT = [
    r"afSTFT_protoFilter1024(LD)?",
    r"afCenterFreq(48e3|44100)",
    r"mono_dirs_deg", r"stereo_dirs_deg", r"\d+pX(_7_4)?_dirs_deg", r"9_10_3p2_dirs_deg",
    r"Aalto_\w+_dirs_deg", r"DTU_AVIL_dirs_deg", r"Zylia_Lab_dirs_deg",
    r"default_LScoords64_rad",
    r"Tdesign_degree_(\d|1\d|20|21|30|100)_dirs_deg",
    r"SphCovering_\d+_dirs_deg",
    r"geosphere_ico_9_0_dirs_deg",
    r"(Zylia|Eigenmike32|DTU_mic)_freqRange",
]
WANT_RE = re.compile(r"^__(" + "|".join(WANT) + r")$")

DECL = re.compile(r"const\s+(?:float|double)\s+(__\w+)\s*((?:\[\s*\d+\s*\])+)\s*=\s*\{", re.S)
NUM = re.compile(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?")


def parse(text):
    for m in DECL.finditer(text):
        name = m.group(1)
        dims = [int(x) for x in re.findall(r"\d+", m.group(2))]
        end = text.index("};", m.end())
        body = re.sub(r"/\*.*?\*/", "", text[m.end():end], flags=re.S)
        vals = np.array([float(x) for x in NUM.findall(body)], dtype=np.float64).astype(np.float32)
        n = int(np.prod(dims))
        if vals.size > n:
            raise SystemExit(f"{name}: parsed {vals.size} values, declared {n}")
        if vals.size < n:
            # C aggregate initialisation zero-fills missing trailing elements; the
            # reference relies on that (e.g. Tdesign_degree_100 lists 5099 of 5100 rows)
            print(f"note: {name}: {vals.size} of {n} values listed, zero-filled like C does")
            vals = np.concatenate([vals, np.zeros(n - vals.size, np.float32)])
        yield name, dims, vals


def main():
    tables = []
    for f in FILES:
        text = (REF / f).read_text(errors="replace")
        for name, dims, vals in parse(text):
            if WANT_RE.match(name):
                d0 = dims[0]
                d1 = dims[1] if len(dims) > 1 else 1
                tables.append((name[2:], d0, d1, vals))
    names = [t[0] for t in tables]
    assert len(set(names)) == len(names)
    OUT.parent.mkdir(parents=True, exist_ok=True)
    with open(OUT, "wb") as fh:
        fh.write(b"SAFT" + struct.pack("<II", 1, len(tables)))
        for name, d0, d1, vals in tables:
            nb = name.encode()
            fh.write(struct.pack("<I", len(nb)) + nb + struct.pack("<II", d0, d1))
            fh.write(vals.astype("<f4").tobytes())
    print(f"{len(tables)} tables, {OUT.stat().st_size} bytes -> {OUT}")
    for name, d0, d1, _ in tables:
        print(f"  {name} [{d0}][{d1}]")


if __name__ == "__main__":
    main()
