"""Diagnostics for the cooperative form (setOverlap(3)): one call of nI instances x nF blocks on the sequential kernels and on the
cooperative form, where the outputs differ, and — on a -DEQ_COOP_CHECK build — the first out-of-range access the kernel recorded.
usage: python tools/probes/coop_dbg.py <instances> <blocks> [path mode 1|2]"""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import spatial_audio_framework_amd.api as saf
from spatial_audio_framework_amd import _lib
from test_gpu_eq_path import make, band_orders
L = _lib.load()
nI, nF = int(sys.argv[1]), int(sys.argv[2])
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
torch.cuda.set_stream(torch.cuda.Stream())
saf.set_stream(torch.cuda.current_stream().cuda_stream)
L.saf_hip_ambi_dec_setTimeDomainPath(mode)
res = []
for ov in (0, 3):
    L.saf_hip_ambi_dec_setOverlap(ov)
    bt = saf.AmbiDecBatch([make(saf.AmbiDec, 512, 7, 29, 1, 1, 1 + i % 2, 1, band_orders(7, 40 + i) if i % 3 else None) for i in range(nI)], nF)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.rand(nI, nF, 64, 512, device="cuda", generator=g) * 2 - 1
    y = torch.zeros(nI, nF, 64, 512, device="cuda")
    st = (nF * 64 * 512, 64 * 512, 512)
    bt.process_ptr(x.data_ptr(), st, y.data_ptr(), st, nF)
    torch.cuda.synchronize()
    print("ov", ov, "lastOverlap", bt.lastOverlap(), "giveups", bt.decodeGiveUps(), flush=True)
    res.append(y.cpu().numpy())
    import ctypes
    L.saf_hip_debug_batch_fetch.restype = ctypes.c_longlong
    L.saf_hip_debug_batch_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
    if ov == 3:
        dbg = np.zeros(8, np.int64)
        got = L.saf_hip_debug_batch_fetch(bt.hb, 2, dbg.ctypes.data, 16)
        print("dbg", got, dbg.tolist(), flush=True)
d = np.abs(res[0] - res[1])
print("max diff", d.max(), "ref max", np.abs(res[0]).max(), "equal", np.array_equal(res[0], res[1]))
if d.max() > 0:
    bad = np.argwhere(d > 0)
    print("n bad", len(bad), "inst", sorted(set(bad[:, 0])), "frames", sorted(set(bad[:, 1])), "rows", len(set(bad[:, 2])), "col/32", sorted(set(bad[:, 3] // 32)))
    print("first", bad[:3].tolist(), "last", bad[-3:].tolist())
