"""one cooperative-form call, small, for diagnostics"""
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
    if False:
        buf = np.zeros(nI * 64 * (8 if ov == 3 else 1) * 2048, np.float32)
        got = L.saf_hip_debug_batch_fetch(bt.hb, 1 if ov == 3 else 0, buf.ctypes.data, buf.size)
        print("fetched", got)
        zz = buf.reshape(nI, 64, -1, 16, 128)[:, :, 0]
        if ov == 0: z0 = zz
        else:
            dz = np.abs(zz - z0)
            print("z: max diff", dz.max(), "max", np.abs(z0).max(), "n bad", (dz > 0).sum())
            bad = np.argwhere(dz > 0)
            print("cols", sorted(set(bad[:, 3]))[:40]); print("chs", sorted(set(bad[:, 1]))); print("hops", sorted(set(bad[:, 2])))
            if len(bad): print("example", bad[0], zz[tuple(bad[0])], z0[tuple(bad[0])])
d = np.abs(res[0] - res[1])
print("max diff", d.max(), "ref max", np.abs(res[0]).max(), "equal", np.array_equal(res[0], res[1]))
if d.max() > 0:
    bad = np.argwhere(d > 0)
    print("n bad", len(bad), "inst", sorted(set(bad[:, 0])), "frames", sorted(set(bad[:, 1])), "rows", len(set(bad[:, 2])), "col/32", sorted(set(bad[:, 3] // 32)))
    print("first", bad[:3].tolist(), "last", bad[-3:].tolist())
