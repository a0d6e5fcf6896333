#!/usr/bin/env python3
"""Diagnostic: cycles per phase of the equaliser kernel (build with SAF_HIP_FLAGS_eq_kernels=-DEQ_STAMPS).  Lane 0 of both waves of
every 64th workgroup accumulates s_memtime deltas at the phase boundaries over the whole launch."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import bench
from spatial_audio_framework_amd import api
from spatial_audio_framework_amd._lib import load
L = load()
nI, nF = 256, 64
decs = [bench.make_decoder(api.AmbiDec) for _ in range(nI)]
bt = api.AmbiDecBatch(decs, nF)
x = torch.rand(nI, nF, 64, 512, device="cuda") * 2 - 1; y = torch.zeros_like(x)
st = (nF * 64 * 512, 64 * 512, 512)
L.saf_hip_ambi_dec_setTimeDomainPath(2)
for _ in range(3): bt.process_ptr(x.data_ptr(), st, y.data_ptr(), st, nF)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 24)()
L.saf_hip_debug_eq_stamps(buf)
n = 10
for _ in range(n): bt.process_ptr(x.data_ptr(), st, y.data_ptr(), st, nF)
L.saf_hip_debug_eq_stamps(buf)
names = ["fold+prefetch", "barrier1", "FFT+lowbins", "barrier2", "bins: main pairs", "IFFT", "barrier3", "xl wait+OLA+stores", "bins: hybrid + DC", "coop: confirm", "coop: finish", "coop: slow polls (count)"]
nwg = nI * 64 // 64
for wv in range(2):
    tot = sum(buf[wv * 12 + i] for i in range(11))
    print(f"wave {wv}: total cycles per workgroup-launch {tot / (n * nwg):.0f} (per sub-chunk {tot / (n * nwg * 16):.0f})")
    for i in range(12):
        print(f"   {names[i]:22s} {buf[wv * 12 + i] / (n * nwg * 16):9.2f} cycles / sub-chunk  {100.0 * buf[wv * 12 + i] / tot:5.1f} %")
