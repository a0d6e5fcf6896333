#!/usr/bin/env python3
"""bench.py's other_configs alone (binauraliser batch, matrixConv, powermap one handle / batch, encode -> decode chain)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import numpy as np
import torch
from spatial_audio_framework_amd import api
from spatial_audio_framework_amd._lib import load
import bench_workloads as W
L = load()
api.set_stream(torch.cuda.current_stream().cuda_stream)
i = np.arange(64) + 0.5
fib = np.stack([np.mod(np.degrees(np.pi * (1.0 + 5.0 ** 0.5) * i), 360.0) - 180.0, np.degrees(np.arcsin(1.0 - 2.0 * i / 64))], 1)
which = sys.argv[1:] or ["bin", "mc", "pm", "pmb", "chain"]
fns = {"bin": lambda: W.binauraliser_batch(L, torch, api), "mc": lambda: W.matrixconv(L, torch, api), "pm": lambda: W.powermap(L, torch, api),
       "pmb": lambda: W.powermap_batch(L, torch, api), "chain": lambda: W.enc_dec_chain(L, torch, api, fib)}
for k in which:
    print(json.dumps(fns[k]()), flush=True)
