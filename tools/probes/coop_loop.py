"""repeated cooperative-form calls against two plain batches: where do they differ?"""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import spatial_audio_framework_amd.api as saf
from spatial_audio_framework_amd import _lib
from test_gpu_eq_path import make, band_orders
L = _lib.load()
nI, nF, mode, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.cuda.set_stream(torch.cuda.Stream())
saf.set_stream(torch.cuda.current_stream().cuda_stream)
L.saf_hip_ambi_dec_setTimeDomainPath(mode)
def mk(ov):
    L.saf_hip_ambi_dec_setOverlap(ov)
    return saf.AmbiDecBatch([make(saf.AmbiDec, 512, 7, 29, 1, 1, 1 + i % 2, 1, band_orders(7, 40 + i) if i % 3 else None) for i in range(nI)], nF)
bt, bs, bs2 = mk(3), mk(0), mk(0)
g = torch.Generator(device="cuda"); g.manual_seed(3)
st = (nF * 64 * 512, 64 * 512, 512)
ya, yb, yc = (torch.zeros(nI, nF, 64, 512, device="cuda") for _ in range(3))
for it in range(iters):
    xin = torch.rand(nI, nF, 64, 512, device="cuda", generator=g) * 2 - 1
    L.saf_hip_ambi_dec_setOverlap(3); bt.process_ptr(xin.data_ptr(), st, ya.data_ptr(), st, nF)
    L.saf_hip_ambi_dec_setOverlap(0); bs.process_ptr(xin.data_ptr(), st, yb.data_ptr(), st, nF)
    bs2.process_ptr(xin.data_ptr(), st, yc.data_ptr(), st, nF)
    torch.cuda.synchronize()
    if torch.equal(ya, yb) and torch.equal(yb, yc):
        continue
    a, b, c = ya.cpu().numpy(), yb.cpu().numpy(), yc.cpu().numpy()
    print(it, "plain==plain2", np.array_equal(b, c), "coop==plain", np.array_equal(a, b), "giveups", bt.decodeGiveUps(), flush=True)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        print("  n bad", len(bad), "inst", sorted(set(bad[:, 0]))[:20], "frames", sorted(set(bad[:, 1])), "rows", len(set(bad[:, 2])), "cols/32", sorted(set(bad[:, 3] // 32)), flush=True)
print("done", iters)
