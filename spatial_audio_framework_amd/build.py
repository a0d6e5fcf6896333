"""Build libsaf_hip.so (the C-ABI library: HIP kernels for gfx950 + host code) in-tree.

    python -m spatial_audio_framework_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is written next to this
file so it travels with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OUT = PKG / "libsaf_hip.so"
OBJ = CSRC / "_build"
TABLES = PKG / "data" / "saf_tables.bin"
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize (all kernel files but KEEP_SLP): the SLP vectorizer turns the complex arithmetic of the FFT butterflies into v_pk_*_f32 plus the
# v_mov_b32 that assemble their register pairs; measured on MI355X that costs 5 % (analysis) to 12 % (synthesis) of the
# afSTFT kernels' time (1.725 -> 1.605 ms, 1.456 -> 1.289 ms per 16 384 frames).
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
            "-Wno-unused-result"]
NO_SLP = ["-fno-slp-vectorize"]
KEEP_SLP = set()                           # (round 2 kept it for the vector covariance update of powermap; that kernel is an MFMA kernel now)


# per-file additions (experiments: SAF_HIP_FLAGS_<file stem>="-flag1 -flag2")
PER_FILE_FLAGS = {}
for _k, _v in os.environ.items():
    if _k.startswith("SAF_HIP_FLAGS_"):
        for _ext in (".hip", ".cpp"):
            PER_FILE_FLAGS[_k[len("SAF_HIP_FLAGS_"):] + _ext] = _v.split()


def sources():
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _newer(a, b):
    return (not b.exists()) or a.stat().st_mtime > b.stat().st_mtime


def needs_build():
    deps = sources() + list(CSRC.glob("*.h")) + [PKG.parent / "include" / "saf_hip.h", TABLES, Path(__file__)]
    return any(_newer(d, OUT) for d in deps)


def _compile(src, headers_mtime):
    obj = OBJ / (src.name + ".o")
    if obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, headers_mtime):
        return obj
    cmd = [HIPCC] + CXXFLAGS + ([] if src.name in KEEP_SLP else NO_SLP) + PER_FILE_FLAGS.get(src.name, []) + (["-x", "hip"] if src.suffix == ".hip" else []) + ["-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    OBJ.mkdir(exist_ok=True)
    if not TABLES.exists():
        raise RuntimeError(f"{TABLES} missing: run tools/extract_tables.py in the build container")
    blob = OBJ / "tables_blob.o"
    if _newer(TABLES, blob):
        # symbol names derive from the file name given to ld: _binary_saf_tables_bin_{start,end}
        subprocess.check_call(["ld", "-r", "-b", "binary", "-z", "noexecstack", "-o", str(blob), TABLES.name], cwd=str(TABLES.parent))
    hdrs = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "saf_hip.h", Path(__file__)]
    hm = max(h.stat().st_mtime for h in hdrs)
    if force:
        hm = float("inf")
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, hm), sources()))
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(OUT)] + [str(o) for o in objs] + [str(blob)]
    subprocess.check_call(cmd)
    if verbose:
        print(f"built {OUT} ({OUT.stat().st_size} bytes)")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
