#!/usr/bin/env python3
"""Lifecycle stress: create / run / destroy every operator many times and watch the device memory — a drop-in must not leak.
Prints the free device memory before and after each operator's loop (MiB)."""
import gc
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from spatial_audio_framework_amd import api
from util import frames, synth_hrirs


def free_mib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20


def loop(name, make, run, n=40):
    make_run = lambda: run(make())
    make_run(); gc.collect()
    f0 = free_mib()
    for _ in range(n):
        make_run()
    gc.collect()
    f1 = free_mib()
    print(f"{name:16s} free before {f0:10.1f} MiB  after {n} cycles {f1:10.1f} MiB  delta {f1 - f0:+8.1f}")
    return f1 - f0


def main():
    h, d = synth_hrirs()
    x64 = frames(1, 64, 512)
    worst = 0.0

    def dec():
        a = api.AmbiDec(128); a.setMasterDecOrder(3); a.setOutputConfigPreset(21); a.initCodec(); a.init(48000); return a
    worst = min(worst, loop("ambi_dec", dec, lambda a: [a.process(x64[:16, :128], 24) for _ in range(3)]))

    def enc():
        e = api.AmbiEnc(128); e.init(48000); e.setOutputOrder(3); e.setNumSources(8); return e
    worst = min(worst, loop("ambi_enc", enc, lambda e: [e.process(x64[:8, :128], 16) for _ in range(3)]))

    def bina():
        b = api.Binauraliser(128, 64); b.setHRIRs(h, d, 48000); b.init(48000); b.setNumSources(8); b.initCodec(); return b
    worst = min(worst, loop("binauraliser", bina, lambda b: [b.process(x64[:8, :128]) for _ in range(3)], n=15))

    def binf():
        b = api.BinauraliserNF(128, 64); b.setHRIRs(h, d, 48000); b.init(48000); b.setNumSources(8); b.initCodec()
        for i in range(8):
            b.setSourceDist_m(i, 0.2 + 0.3 * i)
        return b
    worst = min(worst, loop("binauraliser_nf", binf, lambda b: [b.process(x64[:8, :128]) for _ in range(3)], n=15))

    def pan():
        p = api.Panner(128); p.setOutputConfigPreset(21); p.setNumSources(4); p.initCodec(); p.init(48000); return p
    worst = min(worst, loop("panner", pan, lambda p: [p.process(x64[:4, :128], 24) for _ in range(3)], n=15))

    def abin():
        a = api.AmbiBin(128); a.setHRIRs(h, d, 48000); a.setInputOrderPreset(2); a.init(48000); a.initCodec(); return a
    worst = min(worst, loop("ambi_bin", abin, lambda a: [a.process(x64[:9, :128], 2) for _ in range(3)], n=10))

    H = (np.random.default_rng(0).normal(size=(2, 8, 600)) / 8).astype(np.float32)
    worst = min(worst, loop("matrixConv", lambda: api.MatrixConv(128, H, 1), lambda m: [m.apply(x64[:8, :128]) for _ in range(3)]))

    def pm():
        p = api.Powermap(1024); p.setMasterOrder(3); p.init(48000.0); p.initCodec(); return p
    if hasattr(api, "Powermap"):
        worst = min(worst, loop("powermap", pm, lambda p: [p.analysis(frames(3, 16, 1024)) for _ in range(2)], n=10))

    def rot():
        r = api.Rotator(64); r.init(48000); r.setOrder(5); r.setYaw(20.0); return r
    worst = min(worst, loop("rotator", rot, lambda r: [r.process(x64[:36, :64], 36) for _ in range(3)]))

    def bf():
        b = api.Beamformer(128); b.init(48000); b.setBeamOrder(3); b.setNumBeams(6); return b
    worst = min(worst, loop("beamformer", bf, lambda b: [b.process(x64[:16, :128], 6) for _ in range(3)]))

    def drc():
        dd = api.AmbiDrc(128); dd.setInputPreset(2); dd.init(48000); return dd
    worst = min(worst, loop("ambi_drc", drc, lambda dd: [dd.process(x64[:9, :128]) for _ in range(3)]))

    loop("afSTFT", lambda: api.AfSTFT(8, 8), lambda a: a.backward(a.forward(x64[:8, :512])))
    print("worst delta", worst)


if __name__ == "__main__":
    main()
