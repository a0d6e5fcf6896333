#!/usr/bin/env python3
"""Supplementary measurements of the other BASELINE.json configs (the headline metric stays bench.py's).

For every config: throughput of the device-resident entry point with inputs already in HBM, per-kernel average launch
time (HIP events around every kernel, saf_hip_profile_*), achieved algorithmic GB/s of the dominant kernel against the
HBM peak, and the CPU oracle (a port of the reference path, 1 core) timed on a bounded sample of the same workload.
Writes one JSON object per config to stdout.   python tools/bench_configs.py [--quick]
"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
HBM = 8000.0


def timed(L, torch, fn, steps, warmup, kernels):
    """seconds per call (profiling off: the event pairs around every small kernel cost microseconds) and, from a second
    pass, the average duration of each kernel"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.saf_hip_profile_reset(); L.saf_hip_profile_enable(1)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    L.saf_hip_profile_enable(0)
    per = {}
    for k in kernels:
        tot = C.c_double(); n = L.saf_hip_profile_read(k.encode(), C.byref(tot))
        if n:
            per[k] = round(tot.value / n, 5)
    L.saf_hip_profile_reset()
    return dt / steps, per


def cpu_time(fn, budget=6.0):
    fn(); n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        fn(); n += 1
    return (time.perf_counter() - t0) / n


def main():
    quick = "--quick" in sys.argv
    import torch
    from spatial_audio_framework_amd import api
    from spatial_audio_framework_amd._lib import load
    from oracle import oracle as O
    from util import frames, synth_hrirs
    L = load()
    api.set_stream(torch.cuda.current_stream().cuda_stream)
    steps, warm = (5, 2) if quick else (30, 5)
    out = []

    # ---- configs[0]: ambi_enc 1st order, 4 sources, 256-sample blocks
    F, nS, nI, nF = 256, 4, 256, 64
    def mkenc(cls):
        e = cls(F); e.init(48000); e.setOutputOrder(1); e.setNumSources(nS); return e
    encs = [mkenc(api.AmbiEnc) for _ in range(nI)]
    eb = api.AmbiEncBatch(encs, nF)
    x = torch.rand(nI, nF, nS, F, device="cuda") * 2 - 1; y = torch.zeros(nI, nF, 4, F, device="cuda")
    st_in, st_out = (nF * nS * F, nS * F, F), (nF * 4 * F, 4 * F, F)
    t, per = timed(L, torch, lambda: eb.process_ptr(x.data_ptr(), st_in, nS, y.data_ptr(), st_out, 4, nF), steps, warm, ["sh_encode"])
    oe = mkenc(O.AmbiEnc); xb = frames(1, nS, F)
    tc = cpu_time(lambda: oe.process(xb, 4), 3.0)
    fps = nI * nF / t
    out.append({"config": "ambi_enc 1st-order, 4 sources, 256-sample blocks (configs[0])", "value": round(fps, 1), "unit": "frames/s",
                "batch": f"{nI} instances x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "alg_bytes_per_frame": 2 * 4 * F * 4, "achieved_GBps": round(2 * 4 * F * 4 * nI * nF / (per.get("sh_encode", t * 1e3) * 1e-3) / 1e9, 1), "peak_GBps": HBM},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port"}})

    # ---- configs[2]a: binauraliser, 256 virtual sources
    h, d = synth_hrirs()
    F, nS, nF = 128, 256, 64
    def mkbin(cls):
        b = cls(F, 256); b.setHRIRs(h, d, 48000); b.init(48000); b.setNumSources(nS); b.initCodec()
        rng = np.random.default_rng(9)
        for s in range(nS):
            b.setSourceAzi_deg(s, float(rng.uniform(-180, 180))); b.setSourceElev_deg(s, float(rng.uniform(-80, 80)))
        return b
    gb = mkbin(api.Binauraliser)
    x = torch.rand(nS, nF * F, device="cuda") * 2 - 1; y = torch.zeros(2, nF * F, device="cuda")
    t, per = timed(L, torch, lambda: gb.process_dev(x.data_ptr(), (F, nF * F), nS, y.data_ptr(), (F, nF * F), nF), steps, warm,
                   ["afstft_analysis", "binaural_mac", "afstft_synthesis"])
    ob = mkbin(O.Binauraliser); xb = frames(2, nS, F)
    tc = cpu_time(lambda: ob.process(xb), 6.0)
    alg = nS * F * 4 + 2 * F * 4
    dom = max(per, key=per.get)
    out.append({"config": "binauraliser: 256 virtual sources, 128-sample blocks, synthetic 836-direction HRIR set (configs[2])", "value": round(nF / t, 1), "unit": "frames/s",
                "batch": f"1 handle x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "kernel": dom, "path_alg_bytes_per_frame": alg, "path_achieved_GBps": round(alg * nF / t / 1e9, 1), "peak_GBps": HBM,
                             "note": "one handle = 256 analysis channels x 64 hops per call: far too little work to fill 256 CUs; shard handles over the GPU for throughput"},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port"}})

    # ---- configs[2]b: saf_matrixConv 256 -> 2, 1024 taps, hop 512, partitioned
    nIn, nOut, Lh, hop, nB = 256, 2, 1024, 512, 64
    H = (np.random.default_rng(3).normal(size=(nOut, nIn, Lh)) / 32).astype(np.float32)
    mc = api.MatrixConv(hop, H, 1, maxBlocks=nB)
    x = torch.rand(nIn, nB * hop, device="cuda") * 2 - 1; y = torch.zeros(nOut, nB * hop, device="cuda")
    t, per = timed(L, torch, lambda: mc.apply_dev(x.data_ptr(), (nB * hop, hop), y.data_ptr(), (nB * hop, hop), nB), steps, warm,
                   ["pconv_fft", "pconv_mac", "pconv_ifft"])
    om = O.MatrixConv(hop, H, 1); xb = frames(3, nIn, hop)
    tc = cpu_time(lambda: om.apply(xb), 6.0)
    alg_blk = nIn * hop * 4 + nOut * hop * 4            # samples; the 4.2 MB of filter spectra are read once per call (MAC_TB blocks share a pass)
    out.append({"config": "saf_matrixConv: 256 in -> 2 out, 1024-tap filters, hop 512, partitioned (configs[2])", "value": round(nB / t, 1), "unit": "blocks/s",
                "batch": f"1 handle x {nB} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "kernel": "pconv_mac", "alg_bytes_per_call": alg_blk * nB + 2 * 2 * nIn * 513 * 8,
                             "achieved_GBps": round((alg_blk * nB + 2 * 2 * nIn * 513 * 8) / t / 1e9, 1), "peak_GBps": HBM},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "blocks/s", "cores": 1, "kind": "port"}})

    # ---- configs[3]: powermap, order 7 (64 ch), F = 1024, PWD
    F, nSH, nF = 1024, 64, 16
    def mkpm(cls):
        pm = cls(F); pm.setMasterOrder(7); pm.setPowermapMode(1); pm.init(48000.0); pm.initCodec(); pm.setAnaOrderAllBands(7); pm.setNormType(1); pm.setCovAvgCoeff(0.3)
        return pm
    gp = mkpm(api.Powermap)
    x = torch.rand(nSH, nF * F, device="cuda") * 2 - 1
    def step_pm():
        gp.requestPmapUpdate(); gp.analysis_dev(x.data_ptr(), (F, nF * F), nSH, nF)
    t, per = timed(L, torch, step_pm, steps, warm, ["afstft_analysis", "cov_update", "pwd_map"])
    op = mkpm(O.Powermap); xb = frames(4, nSH, F)
    def cpu_pm():
        op.requestPmapUpdate(); op.analysis(xb)
    tc = cpu_time(cpu_pm, 6.0)
    out.append({"config": "powermap: 64-channel (order 7) input, 133-band afSTFT, F = 1024, PWD map on 812 directions, one map per call (configs[3])", "value": round(nF / t, 1), "unit": "frames/s",
                "batch": f"1 handle x {nF} frames per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "path_alg_bytes_per_frame": nSH * F * 4, "path_achieved_GBps": round(nSH * F * 4 * nF / t / 1e9, 1), "peak_GBps": HBM,
                             "note": "one handle = 64 analysis channels; the covariance kernel runs 133 workgroups"},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s (map every frame)", "cores": 1, "kind": "port"}})

    # ---- configs[4] on one GPU: 2048 sources = 32 scenes x 64 sources, ambi_enc (order 7) -> ambi_dec (64 loudspeakers), F = 512
    F, nI, nS, nF = 512, 32, 64, 16
    src = O.table("SphCovering_64_dirs_deg")
    def mke(cls, i):
        e = cls(F); e.init(48000); e.setOutputOrder(7); e.setNumSources(nS); e.setNormType(1)
        for s in range(nS):
            e.setSourceAzi_deg(s, float(src[(s + 7 * i) % 64, 0])); e.setSourceElev_deg(s, float(src[(s + 7 * i) % 64, 1]))
        return e
    def mkd(cls):
        dd = cls(F); dd.setNormType(1); dd.setChOrder(1); dd.setMasterDecOrder(7); dd.setOutputConfigPreset(29)
        dd.setDecMethod(0, 1); dd.setDecMethod(1, 1); dd.initCodec(); dd.init(48000); dd.setDecOrderAllBands(7); return dd
    ge = [mke(api.AmbiEnc, i) for i in range(nI)]; gd = [mkd(api.AmbiDec) for _ in range(nI)]
    eb, db = api.AmbiEncBatch(ge, nF), api.AmbiDecBatch(gd, nF)
    x = torch.rand(nI, nF, nS, F, device="cuda") * 2 - 1
    sh = torch.zeros(nI, nF, 64, F, device="cuda"); ls = torch.zeros_like(sh)
    st = (nF * 64 * F, 64 * F, F)
    def step5():
        eb.process_ptr(x.data_ptr(), st, nS, sh.data_ptr(), st, 64, nF)
        db.process_ptr(sh.data_ptr(), st, ls.data_ptr(), st, nF)
    t, per = timed(L, torch, step5, steps, warm, ["sh_encode", "afstft_eq", "afstft_analysis", "band_gemm", "afstft_synthesis"])
    oe, od = mke(O.AmbiEnc, 0), mkd(O.AmbiDec); xb = frames(5, nS, F)
    tc = cpu_time(lambda: od.process(oe.process(xb, 64), 64), 6.0)
    out.append({"config": "2048 sources = 32 scenes x 64 sources on ONE GPU: ambi_enc (order 7) -> ambi_dec (64 loudspeakers), 512-sample blocks (configs[4], per-GPU share x 8)",
                "value": round(nI * nF / t, 1), "unit": "scene-frames/s", "batch": f"{nI} scenes x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "path_alg_bytes_per_frame": 2 * 64 * F * 4, "path_achieved_GBps": round(2 * 64 * F * 4 * nI * nF / t / 1e9, 1), "peak_GBps": HBM},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "scene-frames/s", "cores": 1, "kind": "port"}})

    # ---- panner, 64 sources -> 64 loudspeakers (SphCovering-64), F = 512, 8 sources moved before every call
    F, nS, nL, nF = 512, 64, 64, 64
    def mkpan(cls):
        pn = cls(F); pn.setOutputConfigPreset(29); pn.setInputConfigPreset(30); pn.initCodec(); pn.init(48000); return pn
    gp = mkpan(api.Panner)
    x = torch.rand(nS, nF * F, device="cuda") * 2 - 1; y = torch.zeros(nL, nF * F, device="cuda")
    mv = [0]
    def step_pan(pn=gp):
        for k in range(8):
            s = (mv[0] * 8 + k) % nS
            pn.setSourceAzi_deg(s, float((37 * mv[0] + 11 * k) % 360 - 180)); pn.setSourceElev_deg(s, float((13 * mv[0] + 7 * k) % 120 - 60))
        mv[0] += 1
        gp.process_dev(x.data_ptr(), (F, nF * F), nS, y.data_ptr(), (F, nF * F), nF)
    t, per = timed(L, torch, step_pan, steps, warm, ["afstft_analysis", "panner_gains", "band_gemm", "afstft_synthesis"])
    op = mkpan(O.Panner); xb = frames(6, nS, F)
    tc = cpu_time(lambda: op.process(xb, nL), 6.0)
    out.append({"config": "panner (SURVEY 8f-1): 64 sources -> 64 loudspeakers (SphCovering-64), 512-sample blocks, 8 sources moved per call", "value": round(nF / t, 1), "unit": "frames/s",
                "batch": f"1 handle x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "path_alg_bytes_per_frame": (nS + nL) * F * 4, "path_achieved_GBps": round((nS + nL) * F * 4 * nF / t / 1e9, 1), "peak_GBps": HBM,
                             "note": "one handle = 64 analysis channels x 256 hops per call; same three kernels as ambi_dec plus the per-source gain kernel"},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s (static sources)", "cores": 1, "kind": "port"}})

    # ---- ambi_dec order 7 -> 64 virtual loudspeakers -> 2 ears (binauraliseLS), F = 512
    F, nI, nF = 512, 64, 64
    def mkdb(cls):
        dd = cls(F); dd.setHRIRs(h, d, 48000); dd.setNormType(1); dd.setChOrder(1); dd.setMasterDecOrder(7); dd.setOutputConfigPreset(29)
        dd.setDecMethod(0, 1); dd.setDecMethod(1, 1); dd.setBinauraliseLSflag(1); dd.init(48000); dd.initCodec(); dd.setDecOrderAllBands(7); return dd
    gd = [mkdb(api.AmbiDec) for _ in range(nI)]
    db = api.AmbiDecBatch(gd, nF)
    x = torch.rand(nI, nF, 64, F, device="cuda") * 2 - 1; ears = torch.zeros(nI, nF, 2, F, device="cuda")
    t, per = timed(L, torch, lambda: db.process_ptr(x.data_ptr(), (nF * 64 * F, 64 * F, F), ears.data_ptr(), (nF * 2 * F, 2 * F, F), nF), steps, warm,
                   ["afstft_analysis", "band_gemm", "binaural_mac", "afstft_synthesis"])
    od = mkdb(O.AmbiDec); xb = frames(7, 64, F)
    tc = cpu_time(lambda: od.process(xb, 2), 6.0)
    out.append({"config": "ambi_dec binauralised (SURVEY 8f-2): order 7 -> 64 virtual loudspeakers (SphCovering-64) -> 2 ears, 512-sample blocks, synthetic HRIR set", "value": round(nI * nF / t, 1), "unit": "frames/s",
                "batch": f"{nI} instances x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "path_alg_bytes_per_frame": (64 + 2) * F * 4, "path_achieved_GBps": round((64 + 2) * F * 4 * nI * nF / t / 1e9, 1), "peak_GBps": HBM},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port"}})

    # ---- saf_multiConv (64 channels x 4096 taps) and saf_TVConv (1 -> 4 channels, 8192 taps, 64 IR sets, index changes every block)
    nCH, Lh, hop, nB = 64, 4096, 512, 64
    Hm = (np.random.default_rng(4).normal(size=(nCH, Lh)) / 32).astype(np.float32)
    gm = api.MultiConv(hop, Hm, 1, maxBlocks=nB)
    x = torch.rand(nCH, nB * hop, device="cuda") * 2 - 1; y = torch.zeros(nCH, nB * hop, device="cuda")
    t, per = timed(L, torch, lambda: gm.apply_dev(x.data_ptr(), (nB * hop, hop), y.data_ptr(), (nB * hop, hop), nB), steps, warm, ["pconv_fft", "pconv_mac", "pconv_ifft"])
    om = O.MultiConv(hop, Hm, 1); xb = frames(8, nCH, hop)
    tc = cpu_time(lambda: om.apply(xb), 4.0)
    out.append({"config": "saf_multiConv (SURVEY 8f-3): 64 channels x 4096-tap filters, hop 512, partitioned", "value": round(nB / t, 1), "unit": "blocks/s",
                "batch": f"1 handle x {nB} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "alg_bytes_per_call": 2 * nCH * hop * 4 * nB + nCH * 8 * 257 * 8 * 2, "peak_GBps": HBM},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "blocks/s", "cores": 1, "kind": "port"}})
    nIR, nO, Lh, hop, nB = 64, 4, 8192, 512, 64
    Ht = (np.random.default_rng(5).normal(size=(nIR, nO, Lh)) / 64).astype(np.float32)
    gt = api.TVConv(hop, Ht, 0, maxBlocks=nB)
    x = torch.rand(nB * hop, device="cuda") * 2 - 1; y = torch.zeros(nO, nB * hop, device="cuda")
    cnt = [0]
    def step_tv():
        idx = [(cnt[0] * nB + b) * 7 % nIR for b in range(nB)]; cnt[0] += 1
        gt.apply_dev(x.data_ptr(), hop, y.data_ptr(), (nB * hop, hop), idx, nB)
    t, per = timed(L, torch, step_tv, steps, warm, ["pconv_fft", "tvconv_mac", "pconv_ifft"])
    ot = O.TVConv(hop, Ht, 0); xb = frames(9, 1, hop)[0]; k = [0]
    def cpu_tv():
        k[0] += 1; ot.apply(xb, k[0] * 7 % nIR)
    tc = cpu_time(cpu_tv, 4.0)
    out.append({"config": "saf_TVConv (SURVEY 8f-3): 1 -> 4 channels, 8192-tap filters, 64 IR sets, IR index changes every block, hop 512", "value": round(nB / t, 1), "unit": "blocks/s",
                "batch": f"1 handle x {nB} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "note": "3 x 4 spectral products of 16 partitions x 257 bins per block: launch/latency-bound at this size"},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "blocks/s", "cores": 1, "kind": "port"}})

    # ---- powermap in MUSIC and MVDR modes (order 7, F = 1024, one map per call)
    for mode, name in ((4, "MUSIC"), (2, "MVDR")):
        gp2 = mkpm(api.Powermap); gp2.setPowermapMode(mode); gp2.setNumSources(2)
        xs = torch.rand(64, 16 * 1024, device="cuda") * 2 - 1
        def step_pm2():
            gp2.requestPmapUpdate(); gp2.analysis_dev(xs.data_ptr(), (1024, 16 * 1024), 64, 16)
        t, per = timed(L, torch, step_pm2, steps, warm, ["afstft_analysis", "cov_update", "adaptive_map"])
        op2 = mkpm(O.Powermap); op2.setPowermapMode(mode); op2.setNumSources(2); xb = frames(4, 64, 1024)
        def cpu_pm2():
            op2.requestPmapUpdate(); op2.analysis(xb)
        tc = cpu_time(cpu_pm2, 6.0)
        out.append({"config": f"powermap {name} mode (SURVEY 8f-4): 64-channel (order 7) input, F = 1024, 812-direction map, one map per call", "value": round(16 / t, 1), "unit": "frames/s",
                    "batch": "1 handle x 16 frames per call", "kernels_ms": per,
                    "roofline": {"bound": "latency", "note": "one 64 x 64 factorisation per map in a single workgroup (float64 in LDS)"},
                    "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s (map every frame; float64 Jacobi / Cholesky port)", "cores": 1, "kind": "port"}})

    # ---- configs[2] at throughput: 16 binauralisers x 256 sources in one batch
    F, nS, nI, nF = 128, 256, 16, 64
    bins = [mkbin(api.Binauraliser) for _ in range(nI)]
    bb = api.BinauraliserBatch(bins, nF)
    x = torch.rand(nI, nS, nF * F, device="cuda") * 2 - 1; y = torch.zeros(nI, 2, nF * F, device="cuda")
    t, per = timed(L, torch, lambda: bb.process_ptr(x.data_ptr(), (nS * nF * F, F, nF * F), nS, y.data_ptr(), (2 * nF * F, F, nF * F), nF), steps, warm,
                   ["afstft_analysis", "binaural_mac", "afstft_synthesis"])
    out.append({"config": "binauraliser batch: 16 handles x 256 virtual sources, 128-sample blocks (configs[2] at throughput)", "value": round(nI * nF / t, 1), "unit": "frames/s",
                "batch": f"{nI} handles x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "path_alg_bytes_per_frame": nS * F * 4 + 2 * F * 4, "path_achieved_GBps": round((nS * F * 4 + 2 * F * 4) * nI * nF / t / 1e9, 1), "peak_GBps": HBM},
                "cpu_baseline": out[1]["cpu_baseline"]})

    # ---- binauraliser_nf: 16 handles x 64 near-field sources in one batch; one source per handle changes distance before every call
    F, nS, nI, nF = 128, 64, 16, 64
    def mknf(cls):
        b = cls(F, 64); b.setHRIRs(h, d, 48000); b.init(48000); b.setNumSources(nS); b.initCodec()
        rng = np.random.default_rng(19)
        for s in range(nS):
            b.setSourceAzi_deg(s, float(rng.uniform(-180, 180))); b.setSourceElev_deg(s, float(rng.uniform(-80, 80))); b.setSourceDist_m(s, float(rng.uniform(0.15, 3.0)))
        return b
    nfs = [mknf(api.BinauraliserNF) for _ in range(nI)]
    bn = api.BinauraliserBatch(nfs, nF)
    x = torch.rand(nI, nS, nF * F, device="cuda") * 2 - 1; y = torch.zeros(nI, 2, nF * F, device="cuda")
    tick = [0]
    def nf_call():
        tick[0] += 1
        for b in nfs:
            b.setSourceDist_m(tick[0] % nS, 0.2 + 0.01 * (tick[0] % 100))
        bn.process_ptr(x.data_ptr(), (nS * nF * F, F, nF * F), nS, y.data_ptr(), (2 * nF * F, F, nF * F), nF)
    t, per = timed(L, torch, nf_call, steps, warm, ["afstft_analysis", "dvf_scale", "binaural_mac", "afstft_synthesis"])
    on = mknf(O.BinauraliserNF); xn = frames(2, nS, F)
    tc = cpu_time(lambda: on.process(xn), 6.0)
    out.append({"config": "binauraliser_nf batch: 16 handles x 64 near-field sources, 128-sample blocks, one distance per handle changed before every call", "value": round(nI * nF / t, 1), "unit": "frames/s",
                "batch": f"{nI} handles x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "path_alg_bytes_per_frame": nS * F * 4 + 2 * F * 4, "path_achieved_GBps": round((nS * F * 4 + 2 * F * 4) * nI * nF / t / 1e9, 1), "peak_GBps": HBM},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port"}})

    # ---- SURVEY 8f-4 (second half): ambi_bin order 7 -> 2 ears (MagLS, max-rE), F = 512, one handle
    F, nF = 512, 64
    def mkab(cls):
        a = cls(F); a.setHRIRs(h, d, 48000); a.setInputOrderPreset(7); a.setNormType(1); a.init(48000); a.initCodec(); return a
    ga = mkab(api.AmbiBin)
    x = torch.rand(64, nF * F, device="cuda") * 2 - 1; y = torch.zeros(2, nF * F, device="cuda")
    t, per = timed(L, torch, lambda: ga.process_dev(x.data_ptr(), (F, nF * F), 64, y.data_ptr(), (F, nF * F), nF), steps, warm, ["afstft_analysis", "binaural_mac", "afstft_synthesis"])
    oa = mkab(O.AmbiBin); xb = frames(10, 64, F)
    tc = cpu_time(lambda: oa.process(xb), 6.0)
    out.append({"config": "ambi_bin (SURVEY 8f-4): order 7 -> 2 ears, MagLS + max-rE, 512-sample blocks, synthetic HRIR set", "value": round(nF / t, 1), "unit": "frames/s",
                "batch": f"1 handle x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "path_alg_bytes_per_frame": 66 * F * 4, "path_achieved_GBps": round(66 * F * 4 * nF / t / 1e9, 1), "peak_GBps": HBM},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port"}})
    # ---- rotator and beamformer (beyond SURVEY 8f): order 7, 64 blocks of 256 samples per call, device-resident
    F, nF, nSH = 256, 64, 64
    gr = api.Rotator(F); gr.init(48000); gr.setOrder(7); gr.setYaw(35.0); gr.setPitch(-10.0)
    x = torch.rand(nSH, nF * F, device="cuda") * 2 - 1; y = torch.zeros(nSH, nF * F, device="cuda")
    t, per = timed(L, torch, lambda: gr.process_dev(x.data_ptr(), (F, nF * F), nSH, y.data_ptr(), (F, nF * F), nSH, nF), steps, warm, ["sh_encode"])
    orr = O.Rotator(F); orr.init(48000); orr.setOrder(7); orr.setYaw(35.0); xb = frames(31, nSH, F)
    tc = cpu_time(lambda: orr.process(xb, nSH), 3.0)
    out.append({"config": "rotator: order 7 scene rotation, 256-sample blocks", "value": round(nF / t, 1), "unit": "frames/s",
                "batch": f"1 handle x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "path_alg_bytes_per_frame": 2 * nSH * F * 4, "path_achieved_GBps": round(2 * nSH * F * 4 * nF / t / 1e9, 1), "peak_GBps": HBM},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port"}})
    gb = api.Beamformer(F); gb.init(48000); gb.setBeamOrder(7); gb.setNumBeams(64); gb.setNormType(1)
    t, per = timed(L, torch, lambda: gb.process_dev(x.data_ptr(), (F, nF * F), nSH, y.data_ptr(), (F, nF * F), 64, nF), steps, warm, ["sh_encode"])
    ob = O.Beamformer(F); ob.init(48000); ob.setBeamOrder(7); ob.setNumBeams(64); ob.setNormType(1)
    tc = cpu_time(lambda: ob.process(xb, 64), 3.0)
    out.append({"config": "beamformer: 64 hyper-cardioid beams of order 7, 256-sample blocks", "value": round(nF / t, 1), "unit": "frames/s",
                "batch": f"1 handle x {nF} blocks per call", "kernels_ms": per,
                "roofline": {"bound": "hbm", "path_alg_bytes_per_frame": 2 * nSH * F * 4, "path_achieved_GBps": round(2 * nSH * F * 4 * nF / t / 1e9, 1), "peak_GBps": HBM},
                "cpu_baseline": {"value": round(1.0 / tc, 1), "unit": "frames/s", "cores": 1, "kind": "port"}})
    for o in out:
        print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
