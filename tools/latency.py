#!/usr/bin/env python3
"""Per-call latency of the unchanged host-pointer entry points (one block per call: copy in, kernels, copy out, sync) —
what a real-time host sees.  The channel-pointer tables are built once and the C entry points are called directly, so
the numbers are those of the C-ABI, not of the Python convenience wrappers.   python tools/latency.py"""
import ctypes as C
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

    h, dd = synth_hrirs()
    for zc in (1, 0):
        L.saf_hip_setZeroCopyIO(zc)
        tag = "" if zc else " (staged copies: SAF_HIP_ZERO_COPY=0)"
        for F in (128, 512):
            d = api.AmbiDec(F)
            d.setNormType(1); d.setChOrder(1); d.setMasterDecOrder(7); d.setOutputConfigPreset(29)
            d.setDecMethod(0, 1); d.setDecMethod(1, 1); d.initCodec(); d.init(48000); d.setDecOrderAllBands(7)
            x = frames(1, 64, F)
            call, keep = direct(L.ambi_dec_process, d.h, x, 64, F)
            r = lat(call); r["op"] = f"ambi_dec_process order 7 -> 64 loudspeakers, F = {F}" + tag; r["block_us"] = round(F / 48000 * 1e6, 1); out.append(r)
        e = api.AmbiEnc(256); e.init(48000); e.setOutputOrder(1); e.setNumSources(4)
        x = frames(2, 4, 256)
        call, keep = direct(L.ambi_enc_process, e.h, x, 4, 256)
        r = lat(call); r["op"] = "ambi_enc_process 4 sources, order 1, F = 256" + tag; r["block_us"] = round(256 / 48000 * 1e6, 1); out.append(r)
        b = api.Binauraliser(128, 64); b.setHRIRs(h, dd, 48000); b.init(48000); b.setNumSources(64); b.initCodec()
        x = frames(3, 64, 128)
        call, keep = direct(L.binauraliser_process, b.h, x, 2, 128)
        r = lat(call); r["op"] = "binauraliser_process 64 sources, F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
        L.binauraliser_setEnableRotation(b.h, 1)
        yk = [0]
        def tracked():
            yk[0] += 1
            L.binauraliser_setYaw(b.h, C.c_float(float(yk[0] % 90)))
            call()
        r = lat(tracked); r["op"] = "binauraliser_process 64 sources, head yaw changed every block (all HRTFs re-interpolated), F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
        bn = api.BinauraliserNF(128, 64); bn.setHRIRs(h, dd, 48000); bn.init(48000); bn.setNumSources(64); bn.initCodec()
        for s in range(64):
            bn.setSourceAzi_deg(s, float(5 * s - 160)); bn.setSourceDist_m(s, 0.2 + 0.04 * s)
        x = frames(3, 64, 128)
        call, keep = direct(L.binauraliserNF_process, bn.h, x, 2, 128)
        r = lat(call); r["op"] = "binauraliserNF_process 64 near-field sources, F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
        k = [0]
        def moving():
            k[0] += 1
            L.binauraliserNF_setSourceDist_m(bn.h, k[0] % 64, C.c_float(0.2 + 0.01 * (k[0] % 50)))
            call()
        r = lat(moving); r["op"] = "binauraliserNF_process 64 near-field sources, one distance changed per block, F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
        pn = api.Panner(128); pn.setOutputConfigPreset(29); pn.setInputConfigPreset(30); pn.setNumSources(32); pn.initCodec(); pn.init(48000)
        x = frames(4, 32, 128)
        call, keep = direct(L.panner_process, pn.h, x, 64, 128)
        r = lat(call); r["op"] = "panner_process 32 sources -> 64 loudspeakers, F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
        pk = [0]
        def pan_moving():
            pk[0] += 1
            L.panner_setSourceAzi_deg(pn.h, pk[0] % 32, C.c_float(float(pk[0] % 170)))
            call()
        r = lat(pan_moving); r["op"] = "panner_process 32 sources -> 64 loudspeakers, one source moved per block, F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
        ek = [0]
        xe = frames(2, 4, 256)
        ecall, ekeep = direct(L.ambi_enc_process, e.h, xe, 4, 256)
        def enc_moving():
            ek[0] += 1
            L.ambi_enc_setSourceAzi_deg(e.h, ek[0] % 4, C.c_float(float(ek[0] % 170)))
            ecall()
        r = lat(enc_moving); r["op"] = "ambi_enc_process 4 sources, order 1, one source moved per block, F = 256" + tag; r["block_us"] = round(256 / 48000 * 1e6, 1); out.append(r)
        for bo in (3, 7):
            ab = api.AmbiBin(128); ab.setHRIRs(h, dd, 48000); ab.setInputOrderPreset(bo); ab.init(48000); ab.initCodec()
            xa = frames(8, (bo + 1) ** 2, 128)
            acall, akeep = direct(L.ambi_bin_process, ab.h, xa, 2, 128)
            r = lat(acall); r["op"] = f"ambi_bin_process order {bo}, F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
            L.ambi_bin_setEnableRotation(ab.h, 1)
            ak = [0]
            def bin_tracked():
                ak[0] += 1
                L.ambi_bin_setYaw(ab.h, C.c_float(float(ak[0] % 90)))
                acall()
            r = lat(bin_tracked); r["op"] = f"ambi_bin_process order {bo}, head yaw changed every block, F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
        ro = api.Rotator(128); ro.init(48000); ro.setOrder(7); ro.setYaw(30.0)
        x = frames(6, 64, 128)
        call, keep = direct(L.rotator_process, ro.h, x, 64, 128)
        r = lat(call); r["op"] = "rotator_process order 7, F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
        rk = [0]
        def rot_tracked():
            rk[0] += 1
            L.rotator_setYaw(ro.h, C.c_float(float(rk[0] % 90)))
            call()
        r = lat(rot_tracked); r["op"] = "rotator_process order 7, yaw changed every block, F = 128" + tag; r["block_us"] = round(128 / 48000 * 1e6, 1); out.append(r)
        H = (np.random.default_rng(3).normal(size=(2, 256, 1024)) / 32).astype(np.float32)
        mc = api.MatrixConv(512, H, 1)
        xi = np.ascontiguousarray(frames(5, 256, 512)); yo = np.zeros((2, 512), np.float32)
        fpx, fpy = xi.ctypes.data_as(api.fp), yo.ctypes.data_as(api.fp)
        r = lat(lambda: L.saf_matrixConv_apply(mc.h, fpx, fpy)); r["op"] = "saf_matrixConv_apply 256 -> 2, 1024 taps, hop 512" + tag; r["block_us"] = round(512 / 48000 * 1e6, 1); out.append(r)
    L.saf_hip_setZeroCopyIO(1)
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
