#!/usr/bin/env python3
"""Experiment: does the 256 MB Infinity Cache keep the intermediate spectra when the batch is processed in small groups of
instances (analysis -> band GEMM -> synthesis per group, so that a group's spectra — 35 MB per instance at 64 blocks — are
re-read right after they were written)?  Same total work as bench.py's step; one stream."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench as B
from spatial_audio_framework_amd import api
from spatial_audio_framework_amd._lib import load

L = load()
L.saf_hip_ambi_dec_setTimeDomainPath(0)
api.set_stream(torch.cuda.current_stream().cuda_stream)
nI, NCH, FRAME = 256, 64, 512
decs = [B.make_decoder(api, api.AmbiDec) for _ in range(nI)]
for nF in (64, 16):
    x = torch.rand(nI, nF, NCH, FRAME, device="cuda") * 2 - 1
    y = torch.zeros(nI, nF, NCH, FRAME, device="cuda")
    st = (nF * NCH * FRAME, NCH * FRAME, FRAME)
    for per in (256, 32, 8, 4):
        parts = [api.AmbiDecBatch(decs[j:j + per], nF) for j in range(0, nI, per)]
        def step():
            for j, b in enumerate(parts):
                b.process_ptr(x[j * per:].data_ptr(), st, y[j * per:].data_ptr(), st, nF)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{nF} blocks/call, groups of {per:3d} instances ({per * nF * 0.544:.0f} MB of spectra per group): {dt * 1e3:7.3f} ms/step  {nI * nF / dt / 1e6:.3f} M frames/s")
        del parts
