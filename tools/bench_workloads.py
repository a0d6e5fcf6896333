"""Short device-resident measurements of the other BASELINE.json configs for bench.py's `other_configs` key
(configs[2] binauraliser 256 sources + saf_matrixConv 256 -> 2 x 1024 taps, configs[3] powermap order 7, configs[4]
encode -> decode chain).  Each returns {config, value, unit, batch, kernels_ms, roofline{kernel, bound, achieved, peak,
unit, frac}}; timed regions are tens of milliseconds.  The long table with CPU legs is tools/bench_configs.py."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p_ in (ROOT, ROOT / "tests"):
    if str(p_) not in sys.path:
        sys.path.insert(0, str(p_))
HBM = 8000.0


def _timed(L, torch, fn, steps, warmup, kernels):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
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
    return dt, per


MFMA_F32 = 157.3


def _roof(per, alg_bytes_per_launch, bounds=None, alg_flop_per_launch=None, working_set_mb=None):
    """roofline of the kernel with the largest average launch time against ITS bound: "hbm" (algorithmic bytes of that kernel per
    launch / its time against 8 TB/s), "mfma" (its flops against the fp32 MFMA peak) or "latency" (a launch of a few dozen
    workgroups whose time is a chain of dependent round trips: no throughput roof applies, the time itself is the figure).
    working_set_mb: what the region touches per step — at or below the 256 MB Infinity Cache the "hbm" figure is a cache figure."""
    dom = max(per, key=per.get)
    bound = (bounds or {}).get(dom, "hbm")
    if bound == "mfma":
        ach = alg_flop_per_launch[dom] / (per[dom] * 1e-3) / 1e12
        r = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32, 4)}
    elif bound == "latency":
        r = {"kernel": dom, "bound": "latency", "achieved": None, "peak": None, "unit": "ms per launch", "frac": None, "launch_ms": per[dom]}
    else:
        ach = alg_bytes_per_launch[dom] / (per[dom] * 1e-3) / 1e9
        r = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM, "unit": "GB/s", "frac": round(ach / HBM, 4)}
    if working_set_mb is not None:
        r["working_set_mb"] = round(working_set_mb, 1)
        if bound == "hbm" and working_set_mb <= 256:
            r["note"] = "working set within the 256 MB Infinity Cache: not an HBM figure"
    return r


def binauraliser_batch(L, torch, api, steps=10, warmup=2, nI=16):
    from util import synth_hrirs
    h, d = synth_hrirs()
    F, nS, nF = 128, 256, 64
    rng = np.random.default_rng(9)
    az, el = rng.uniform(-180, 180, nS), rng.uniform(-80, 80, nS)

    def mk():
        b = api.Binauraliser(F, 256); b.setHRIRs(h, d, 48000); b.init(48000); b.setNumSources(nS); b.initCodec()
        for s in range(nS):
            b.setSourceAzi_deg(s, float(az[s])); b.setSourceElev_deg(s, float(el[s]))
        return b
    bins = [mk() for _ in range(nI)]
    bb = api.BinauraliserBatch(bins, nF)
    x = torch.rand(nI, nS, nF * F, device="cuda") * 2 - 1; y = torch.zeros(nI, 2, nF * F, device="cuda")
    t, per = _timed(L, torch, lambda: bb.process_ptr(x.data_ptr(), (nS * nF * F, F, nF * F), nS, y.data_ptr(), (2 * nF * F, F, nF * F), nF), steps, warmup,
                    ["afstft_analysis", "binaural_mac", "afstft_synthesis"])
    H = nF * F // 128
    alg = {"afstft_analysis": nI * nS * H * (128 * 4 + 133 * 8), "binaural_mac": nI * H * 133 * 8 * (nS + 2), "afstft_synthesis": nI * 2 * H * (133 * 8 + 128 * 4)}
    return {"config": f"binauraliser: {nI} handles x 256 virtual sources, 128-sample blocks, synthetic 836-direction HRIR set (configs[2])",
            "value": round(nI * nF / t, 1), "unit": "frames/s", "batch": f"{nI} handles x {nF} blocks per call", "kernels_ms": per,
            "roofline": _roof(per, alg, working_set_mb=nI * (nS * nF * F * 4 + nS * H * 133 * 8) / 1e6),
            "path_hbm_frac": round((nS * F * 4 + 2 * F * 4) * (nI * nF / t) / 1e9 / HBM, 4)}


def matrixconv(L, torch, api, steps=10, warmup=2):
    nIn, nOut, Lh, hop, nB = 256, 2, 1024, 512, 64
    Hm = (np.random.default_rng(3).normal(size=(nOut, nIn, Lh)) / 32).astype(np.float32)
    mc = api.MatrixConv(hop, Hm, 1, maxBlocks=nB)
    x = torch.rand(nIn, nB * hop, device="cuda") * 2 - 1; y = torch.zeros(nOut, nB * hop, device="cuda")
    t, per = _timed(L, torch, lambda: mc.apply_dev(x.data_ptr(), (nB * hop, hop), y.data_ptr(), (nB * hop, hop), nB), steps, warmup,
                    ["pconv_fft", "pconv_mac", "pconv_ifft"])
    nPart, bins = Lh // hop, hop            # spectra rows hold `hop` complex bins (bin 0 carries DC and Nyquist)
    alg = {"pconv_fft": nB * nIn * (hop * 4 + bins * 8), "pconv_mac": nB * nIn * bins * 8 + nOut * nPart * nIn * bins * 8 + nB * nOut * bins * 8 * 4,
           "pconv_ifft": nB * nOut * (bins * 8 * 4 + hop * 4)}
    return {"config": "saf_matrixConv: 256 in -> 2 out, 1024-tap filters, hop 512, partitioned (configs[2])", "value": round(nB / t, 1), "unit": "blocks/s",
            "batch": f"1 handle x {nB} blocks per call", "kernels_ms": per,
            "roofline": _roof(per, alg, {"pconv_fft": "latency", "pconv_mac": "latency", "pconv_ifft": "latency"}, working_set_mb=(nIn * nB * hop * 4 + nIn * nB * bins * 8) / 1e6)}


def powermap(L, torch, api, steps=10, warmup=2):
    F, nSH, nF = 1024, 64, 16
    pm = api.Powermap(F); pm.setMasterOrder(7); pm.setPowermapMode(1); pm.init(48000.0); pm.initCodec(); pm.setAnaOrderAllBands(7); pm.setNormType(1); pm.setCovAvgCoeff(0.3)
    x = torch.rand(nSH, nF * F, device="cuda") * 2 - 1

    def step():
        pm.requestPmapUpdate(); pm.analysis_dev(x.data_ptr(), (F, nF * F), nSH, nF)
    t, per = _timed(L, torch, step, steps, warmup, ["afstft_analysis", "cov_update", "pwd_map"])
    H = nF * F // 128
    alg = {"afstft_analysis": nSH * H * (128 * 4 + 133 * 8), "cov_update": nSH * H * 133 * 8 + 2 * 133 * 64 * 64 * 8, "pwd_map": 133 * 64 * 64 * 8 + 812 * 64 * 4}
    return {"config": "powermap: 64-channel (order 7) input, 133-band afSTFT, F = 1024, PWD map on 812 directions, one map per call (configs[3])",
            "value": round(nF / t, 1), "unit": "frames/s", "batch": f"1 handle x {nF} frames per call", "kernels_ms": per,
            "roofline": _roof(per, alg, {"afstft_analysis": "latency", "cov_update": "latency", "pwd_map": "latency"})}


def powermap_batch(L, torch, api, steps=10, warmup=2, nI=16):
    """configs[3] at throughput: nI handles x 16 frames per call (saf_hip_powermap_batch_*), one of the handles asks for a map per call"""
    F, nSH, nF = 1024, 64, 16

    def mk():
        pm = api.Powermap(F); pm.setMasterOrder(7); pm.setPowermapMode(1); pm.init(48000.0); pm.initCodec(); pm.setAnaOrderAllBands(7); pm.setNormType(1); pm.setCovAvgCoeff(0.3)
        return pm
    pms = [mk() for _ in range(nI)]
    bt = api.PowermapBatch(pms, nF)
    x = torch.rand(nI, nSH, nF * F, device="cuda") * 2 - 1
    k = [0]

    def step():
        pms[k[0] % nI].requestPmapUpdate(); k[0] += 1
        bt.analysis_ptr(x.data_ptr(), (nSH * nF * F, F, nF * F), nSH, nF)
    t, per = _timed(L, torch, step, steps, warmup, ["afstft_analysis", "cov_update", "pwd_map"])
    H = nF * F // 128
    alg = {"afstft_analysis": nI * nSH * H * (128 * 4 + 133 * 8), "cov_update": nI * (nSH * H * 133 * 8 + 2 * 133 * 64 * 64 * 8), "pwd_map": 133 * 64 * 64 * 8 + 812 * 64 * 4}
    flop = {"cov_update": nI * nF * 133 * 64 * 64 * 8 * 8}          # 34.9 MFLOP per frame
    r = _roof(per, alg, {"afstft_analysis": "hbm", "cov_update": "mfma", "pwd_map": "latency"}, flop, working_set_mb=nI * (nSH * nF * F * 4 + nSH * H * 133 * 8 + 133 * 4096 * 8) / 1e6)
    if "cov_update" in per:
        r["cov_update_mfma"] = {"achieved": round(flop["cov_update"] / (per["cov_update"] * 1e-3) / 1e12, 2), "peak": MFMA_F32, "unit": "TFLOP/s",
                                "frac": round(flop["cov_update"] / (per["cov_update"] * 1e-3) / 1e12 / MFMA_F32, 4)}
    return {"config": f"powermap batch: {nI} handles x 64-channel (order 7) input, F = 1024, PWD map of one handle per call (configs[3] at throughput)",
            "value": round(nI * nF / t, 1), "unit": "frames/s", "batch": f"{nI} handles x {nF} frames per call", "kernels_ms": per, "roofline": r}


def enc_dec_chain(L, torch, api, O_tables, steps=10, warmup=2):
    F, nI, nS, nF = 512, 32, 64, 16
    src = O_tables

    def mke(i):
        e = api.AmbiEnc(F); e.init(48000); e.setOutputOrder(7); e.setNumSources(nS); e.setNormType(1)
        for s in range(nS):
            e.setSourceAzi_deg(s, float(src[(s + 7 * i) % 64, 0])); e.setSourceElev_deg(s, float(src[(s + 7 * i) % 64, 1]))
        return e

    def mkd():
        dd = api.AmbiDec(F); dd.setNormType(1); dd.setChOrder(1); dd.setMasterDecOrder(7); dd.setOutputConfigPreset(29)
        dd.setDecMethod(0, 1); dd.setDecMethod(1, 1); dd.initCodec(); dd.init(48000); dd.setDecOrderAllBands(7); return dd
    ge = [mke(i) for i in range(nI)]; gd = [mkd() for _ in range(nI)]
    eb, db = api.AmbiEncBatch(ge, nF), api.AmbiDecBatch(gd, nF)
    x = torch.rand(nI, nF, nS, F, device="cuda") * 2 - 1
    sh = torch.zeros(nI, nF, 64, F, device="cuda"); ls = torch.zeros_like(sh)
    st = (nF * 64 * F, 64 * F, F)

    def step():
        eb.process_ptr(x.data_ptr(), st, nS, sh.data_ptr(), st, 64, nF)
        db.process_ptr(sh.data_ptr(), st, ls.data_ptr(), st, nF)
    t, per = _timed(L, torch, step, steps, warmup, ["sh_encode", "afstft_eq", "band_gemm"])
    blk = nI * nF * 2 * 64 * F * 4
    alg = {"sh_encode": blk, "afstft_eq": blk, "band_gemm": blk}
    return {"config": "2048 sources = 32 scenes x 64 sources on ONE GPU: ambi_enc (order 7) -> ambi_dec (64 loudspeakers), 512-sample blocks (configs[4], per-GPU share x 8)",
            "value": round(nI * nF / t, 1), "unit": "scene-frames/s", "batch": f"{nI} scenes x {nF} blocks per call", "kernels_ms": per, "roofline": _roof(per, alg, working_set_mb=3 * nI * nF * 64 * F * 4 / 1e6),
            "path_hbm_frac": round(2 * 64 * F * 4 * (nI * nF / t) / 1e9 / HBM, 4)}
