#!/usr/bin/env python3
"""Where a single host-pointer ambi_dec_process call spends its time: kernel durations (HIP events) vs. the whole call."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from spatial_audio_framework_amd import api
from spatial_audio_framework_amd._lib import load
from util import frames

L = load()
for td in (1, 0):
    L.saf_hip_ambi_dec_setTimeDomainPath(td)
    F = 128
    d = api.AmbiDec(F)
    d.setNormType(1); d.setChOrder(1); d.setMasterDecOrder(7); d.setOutputConfigPreset(29)
    d.setDecMethod(0, 1); d.setDecMethod(1, 1); d.initCodec(); d.init(48000); d.setDecOrderAllBands(7)
    x = frames(1, 64, F)
    for _ in range(30):
        d.process(x, 64)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); d.process(x, 64); ts.append(time.perf_counter() - t0)
    print("td", td, "call median us", round(float(np.median(ts)) * 1e6, 1))
    L.saf_hip_profile_reset(); L.saf_hip_profile_enable(1)
    for _ in range(100):
        d.process(x, 64)
    L.saf_hip_profile_enable(0)
    for k in ("afstft_analysis", "band_gemm", "afstft_synthesis", "afstft_roundtrip", "ana_hist_update"):
        tot = C.c_double(); n = L.saf_hip_profile_read(k.encode(), C.byref(tot))
        if n:
            print("   ", k, n, "launches, avg us", round(tot.value / n * 1e3, 2))
