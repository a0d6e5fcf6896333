"""The single-scene exchange on the GPU build — needs an MI355X.

Two ranks share the ONE GPU of the test box (`gloo` process group on 127.0.0.1; RCCL needs one device per rank and runs on
the 8-GPU node only): each rank encodes its half of a 140-source scene with the HIP encoder instances, sums their SH blocks,
decodes once with the HIP decoder, and the ranks' loudspeaker blocks are summed with `parallel.sum_partial_fields`.  The
result must equal the CPU oracle's rendering of the whole scene (3 encoder instances summed, one decoder) within the
1e-5 relative RMS of the path.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu
NSRC, NF = 140, 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def scene_input(nF):
    """[NSRC][nF * 512] deterministic source signals"""
    sys.path.insert(0, str(ROOT / "tests"))
    from util import frames
    return frames(4242, NSRC, nF * 512)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
    import torch
    import single_scene as S
    from spatial_audio_framework_amd import api, parallel as P
    torch.cuda.set_device(0)
    P.init(backend="gloo")
    api.set_stream(torch.cuda.current_stream().cuda_stream)
    mine = list(P.shard(NSRC, world, rank))
    eb, db, nI, keep = S.build(api, mine, NF)
    sig = scene_input(NF)
    x = np.zeros((nI, NF, 64, S.F), np.float32)
    for j, s in enumerate(mine):
        x[j // 64, :, j % 64, :] = sig[s].reshape(NF, S.F)
    dx = torch.from_numpy(x).cuda()
    sh = torch.zeros(nI, NF, 64, S.F, device="cuda"); sh1 = torch.zeros(1, NF, 64, S.F, device="cuda"); out = torch.zeros(1, NF, 64, S.F, device="cuda")
    S.render(eb, db, nI, dx, sh, sh1, out, NF)
    P.sum_partial_fields(out, root=0)
    torch.cuda.synchronize()
    if rank == 0:
        q.put(out.cpu().numpy()[0])
    P.finalize()


def test_single_scene_two_ranks_on_one_gpu_vs_oracle(orc):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)                                   # [NF][64][512]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # oracle: the whole scene in one process (encoder instances of <= 64 sources summed, one decoder)
    sys.path.insert(0, str(ROOT / "tools"))
    import single_scene as S
    sig = scene_input(NF)
    encs, dec = build_oracle(orc)
    ref = np.zeros((NF, 64, S.F), np.float32)
    for f in range(NF):
        shsum = np.zeros((64, S.F), np.float32)
        for i, e in enumerate(encs):
            n = min(64, NSRC - 64 * i)
            shsum += e.process(np.ascontiguousarray(sig[64 * i:64 * i + n, f * S.F:(f + 1) * S.F]), 64)
        ref[f] = dec.process(shsum, 64)
    den = float((ref ** 2).sum())
    assert den > 1.0
    assert (float(((got - ref) ** 2).sum()) / den) ** 0.5 < 1e-5


def build_oracle(orc):
    sys.path.insert(0, str(ROOT / "tools"))
    import single_scene as S
    groups = [list(range(NSRC))[i:i + 64] for i in range(0, NSRC, 64)]
    encs = []
    for g in groups:
        e = orc.AmbiEnc(S.F); e.init(48000); e.setOutputOrder(S.ORDER); e.setNumSources(len(g)); e.setNormType(1); e.setEnablePostScaling(0)
        for j, s in enumerate(g):
            e.setSourceAzi_deg(j, float((53 * s) % 360 - 180)); e.setSourceElev_deg(j, float((29 * s) % 120 - 60))
        encs.append(e)
    d = orc.AmbiDec(S.F)
    d.setNormType(1); d.setChOrder(1); d.setMasterDecOrder(S.ORDER); d.setOutputConfigPreset(29)
    d.setDecMethod(0, 1); d.setDecMethod(1, 1); d.initCodec(); d.init(48000); d.setDecOrderAllBands(S.ORDER)
    return encs, d


def test_bench_spawns_two_ranks_on_the_gpu_box():
    """`python bench.py --gpus 2` starts two ranks itself; on this one-GPU box they share cuda:0 and meet over gloo
    (--share-gpu): every rank builds its own pipelines, runs the timed steps between the barriers, rank 0 reports the MAX."""
    import json, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in __import__("os").environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1", "--instances", "8",
                        "--frames-per-call", "8", "--no-cpu-baseline", "--no-extra-paths"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["instances_per_gpu"] == 8
    km = line["roofline"]["kernels_ms"]
    assert km.get("afstft_eq", 0) > 0 or km.get("afstft_eq_decode", 0) > 0
