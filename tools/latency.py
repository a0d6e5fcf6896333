#!/usr/bin/env python3
"""Per-call latency of the unchanged host-pointer entry points (one block per call: copy in, kernels, copy out, sync) —
what a real-time host sees.  The channel-pointer tables are built once and the C entry points are called directly, so
the numbers are those of the C-ABI, not of the Python convenience wrappers.   python tools/latency.py"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def lat(fn, n=400, warm=30):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts = np.sort(np.array(ts)) * 1e6
    return {"median_us": round(float(ts[len(ts) // 2]), 1), "p99_us": round(float(ts[int(len(ts) * 0.99)]), 1), "min_us": round(float(ts[0]), 1)}


def main():
    from spatial_audio_framework_amd import api
    from spatial_audio_framework_amd._lib import load
    from util import frames, synth_hrirs
    L = load()
    out = []

    def direct(fn, h, x, nOut, F):
        y = np.zeros((nOut, F), np.float32)
        px, py = api._rows(x), api._rows(y)
        return lambda: fn(h, px, py, x.shape[0], nOut, F), (x, y, px, py)
    for F in (128, 512):
        d = api.AmbiDec(F)
        d.setNormType(1); d.setChOrder(1); d.setMasterDecOrder(7); d.setOutputConfigPreset(29)
        d.setDecMethod(0, 1); d.setDecMethod(1, 1); d.initCodec(); d.init(48000); d.setDecOrderAllBands(7)
        x = frames(1, 64, F)
        call, keep = direct(L.ambi_dec_process, d.h, x, 64, F)
        r = lat(call); r["op"] = f"ambi_dec_process order 7 -> 64 loudspeakers, F = {F}"; r["block_us"] = round(F / 48000 * 1e6, 1); out.append(r)
    e = api.AmbiEnc(256); e.init(48000); e.setOutputOrder(1); e.setNumSources(4)
    x = frames(2, 4, 256)
    call, keep = direct(L.ambi_enc_process, e.h, x, 4, 256)
    r = lat(call); r["op"] = "ambi_enc_process 4 sources, order 1, F = 256"; r["block_us"] = round(256 / 48000 * 1e6, 1); out.append(r)
    h, dd = synth_hrirs()
    b = api.Binauraliser(128, 64); b.setHRIRs(h, dd, 48000); b.init(48000); b.setNumSources(64); b.initCodec()
    x = frames(3, 64, 128)
    call, keep = direct(L.binauraliser_process, b.h, x, 2, 128)
    r = lat(call); r["op"] = "binauraliser_process 64 sources, F = 128"; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
