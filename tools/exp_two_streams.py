#!/usr/bin/env python3
"""Experiment: the headline step as ONE batch of 256 instances on one stream vs TWO batches of 128 on two streams
(kernels of different phases may then overlap: the VALU-heavy filterbank with the memory-bound band GEMM)."""
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
nI, nF, NCH, FRAME = 256, 64, 64, 512
decs = [B.make_decoder(api, api.AmbiDec) for _ in range(nI)]
x = torch.rand(nI, nF, NCH, FRAME, device="cuda") * 2 - 1
y = torch.zeros(nI, nF, NCH, FRAME, device="cuda")
st = (nF * NCH * FRAME, NCH * FRAME, FRAME)


def run(parts, streams, steps=30):
    k = len(parts)
    per = nI // k
    def step():
        for j, (b, s) in enumerate(zip(parts, streams)):
            api.set_stream(s.cuda_stream)
            b.process_ptr(x[j * per:].data_ptr(), st, y[j * per:].data_ptr(), st, nF)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


s0 = torch.cuda.Stream()
api.set_stream(s0.cuda_stream)
one = api.AmbiDecBatch(decs, nF)
t1 = run([one], [s0])
print(f"1 batch x 256, 1 stream : {t1 * 1e3:.3f} ms/step  {nI * nF / t1 / 1e6:.3f} M frames/s")
del one
for k in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(k)]
    parts = []
    for j in range(k):
        api.set_stream(streams[j].cuda_stream)
        parts.append(api.AmbiDecBatch(decs[j * nI // k:(j + 1) * nI // k], nF))
    torch.cuda.synchronize()
    tk = run(parts, streams)
    print(f"{k} batches x {nI // k}, {k} streams: {tk * 1e3:.3f} ms/step  {nI * nF / tk / 1e6:.3f} M frames/s")
    tk1 = run(parts, [streams[0]] * k)
    print(f"{k} batches x {nI // k}, 1 stream : {tk1 * 1e3:.3f} ms/step  {nI * nF / tk1 / 1e6:.3f} M frames/s")
    del parts
