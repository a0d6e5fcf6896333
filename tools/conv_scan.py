#!/usr/bin/env python3
"""saf_matrixConv (256 -> 2, 1024 taps, hop 512): per-kernel time and throughput against the blocks per call."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from spatial_audio_framework_amd import api
from spatial_audio_framework_amd._lib import load

L = load()
api.set_stream(torch.cuda.current_stream().cuda_stream)
nIn, nOut, Lh, hop = 256, 2, 1024, 512
H = (np.random.default_rng(3).normal(size=(nOut, nIn, Lh)) / 32).astype(np.float32)
for nB in (1, 4, 16, 64, 256):
    mc = api.MatrixConv(hop, H, 1, maxBlocks=nB)
    x = torch.rand(nIn, nB * hop, device="cuda") * 2 - 1; y = torch.zeros(nOut, nB * hop, device="cuda")
    fn = lambda: mc.apply_dev(x.data_ptr(), (nB * hop, hop), y.data_ptr(), (nB * hop, hop), nB)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    L.saf_hip_profile_reset(); L.saf_hip_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    L.saf_hip_profile_enable(0)
    per = {}
    for k in ("pconv_fft", "pconv_mac", "pconv_ifft"):
        tot = C.c_double(); n = L.saf_hip_profile_read(k.encode(), C.byref(tot)); per[k] = round(tot.value / n * 1e3, 1)
    L.saf_hip_profile_reset()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / 50
    print(f"nB {nB:4d}  call {dt2 * 1e6:8.1f} us  {nB / dt2:10.0f} blocks/s  kernels us {per}")
