#!/usr/bin/env python3
"""debug: split invariance of the equaliser path (where do two differently cut streams differ?)"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from spatial_audio_framework_amd import api as saf
from spatial_audio_framework_amd._lib import load
from util import frames
from test_gpu_eq_path import make, band_orders
L = load()
saf.set_stream(torch.cuda.current_stream().cuda_stream)
F, order, nI, nF = 512, 7, 4, 6
cfgs = [(1, 1, 1, None), (3, 2, 2, band_orders(7, 1)), (4, 1, 1, band_orders(7, 2)), (1, 1, 2, band_orders(7, 3))]
x = np.stack([frames(500 + i, nF * 64, 512).reshape(nF, 64, 512) for i in range(nI)])
d_in = torch.from_numpy(x).cuda()
st = (nF * 64 * 512, 64 * 512, 512)
for mode in (1, 2):
    L.saf_hip_ambi_dec_setTimeDomainPath(mode)
    decs = [make(saf.AmbiDec, F, order, 29, a, b, n, 1, o) for a, b, n, o in cfgs]
    res = {}
    for split in ((nF,), (1, 2, 3), (nF,), (3, 3), (2, 4), (4, 2)):
        bt = saf.AmbiDecBatch(decs, nF)
        d_out = torch.zeros(nI, nF, 64, 512, device="cuda")
        f0 = 0
        for n in split:
            bt.process_ptr(d_in[:, f0:].data_ptr(), st, d_out[:, f0:].data_ptr(), st, n)
            f0 += n
        torch.cuda.synchronize()
        y = d_out.cpu().numpy()
        if (nF,) in res and split != (nF,) or (split == (nF,) and (nF,) in res):
            ref = res[(nF,)]
            d = np.abs(y - ref)
            print("mode", mode, "split", split, "max", d.max(), "per inst/frame:\n", np.array2string(d.max(axis=(2, 3)), precision=2))
            if d.max() > 0:
                i, f, c, n = np.unravel_index(d.argmax(), d.shape)
                print("   worst at inst", i, "frame", f, "ch", c, "n", n, "val", ref[i, f, c, n], "hops with diffs in that (inst,ch):",
                      sorted(set((np.nonzero(d[i, :, c].reshape(-1))[0] // 128).tolist()))[:40])
        res.setdefault(split, y)
