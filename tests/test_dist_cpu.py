"""The N > 1 path on CPU: world-size-2 `gloo` process groups on 127.0.0.1 (no GPU needed).

Instances are independent, so N ranks each run their shard of the instances with no data-path collective;
ranks meet in barriers, one MAX all-reduce of the elapsed time and the gathering of results.  Here each rank renders
its shard of encode -> decode scenes with the CPU oracle (the checker), the shards are gathered, and the result must
equal the single-process rendering of all scenes — plus the bookkeeping `bench.py` relies on.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def render_scene(O, scene, nFrames=3, F=128):
    """One independent unit: 4 sources -> ambi_enc (order 1) -> ambi_dec (t-design-4) -> checksum-able output."""
    sys.path.insert(0, str(ROOT / "tests"))
    from util import frames
    e = O.AmbiEnc(F); e.init(48000); e.setOutputOrder(1); e.setNumSources(4); e.setNormType(1)
    for s in range(4):
        e.setSourceAzi_deg(s, float((37 * scene + 90 * s) % 360 - 180)); e.setSourceElev_deg(s, float((11 * scene + 20 * s) % 120 - 60))
    d = O.AmbiDec(F); d.setNormType(1); d.setMasterDecOrder(1); d.setOutputConfigPreset(19)
    d.setDecMethod(0, 1); d.setDecMethod(1, 1); d.initCodec(); d.init(48000)
    x = frames(100 + scene, 4, nFrames * F)
    out = []
    for f in range(nFrames):
        sh = e.process(x[:, f * F:(f + 1) * F], 4)
        out.append(d.process(sh, 4))
    return np.concatenate(out, 1)


def _worker(rank, world, port, nScenes, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    from spatial_audio_framework_amd import parallel as P
    from oracle import oracle as O
    w, r, _ = P.init(backend="gloo")
    assert (w, r) == (world, rank)
    mine = P.shard(nScenes, world, rank)
    P.barrier()
    local = np.stack([render_scene(O, s) for s in mine]) if len(mine) else np.zeros((0, 4, 384), np.float32)
    P.barrier()
    elapsed = 1.0 + rank                                    # the slowest rank defines the job time
    t_max = P.max_over_ranks(elapsed)
    n_total = P.sum_over_ranks(len(mine))
    everything = P.gather_arrays(local)                     # equal shard sizes in this test (nScenes % world == 0)
    if rank == 0:
        q.put((t_max, n_total, everything))
    P.finalize()


def test_two_rank_gloo_sharding_matches_single_process():
    nScenes, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nScenes, q)) for r in range(world)]
    for p in procs:
        p.start()
    t_max, n_total, gathered = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t_max == 2.0 and n_total == nScenes
    sys.path.insert(0, str(ROOT))
    from oracle import oracle as O
    ref = np.stack([render_scene(O, s) for s in range(nScenes)])
    assert gathered.shape == ref.shape and np.array_equal(gathered, ref)     # no cross-talk, rank order preserved


def render_partial_field(O, sources, nFrames=18, F=128, order=2):
    """The loudspeaker feeds that the given subset of a 12-source scene contributes: ambi_enc (its sources only, no
    post-scaling) -> ambi_dec.  Linear in the sources."""
    sys.path.insert(0, str(ROOT / "tests"))
    from util import frames
    nSH = (order + 1) ** 2
    x_all = frames(555, 12, nFrames * F)
    e = O.AmbiEnc(F); e.init(48000); e.setOutputOrder(order); e.setNumSources(len(sources)); e.setNormType(1); e.setEnablePostScaling(0)
    for j, s in enumerate(sources):
        e.setSourceAzi_deg(j, float((53 * s) % 360 - 180)); e.setSourceElev_deg(j, float((29 * s) % 120 - 60))
    d = O.AmbiDec(F); d.setNormType(1); d.setMasterDecOrder(order); d.setOutputConfigPreset(20)
    d.setDecMethod(0, 1); d.setDecMethod(1, 1); d.initCodec(); d.init(48000); d.setDecOrderAllBands(order)
    nLS = d.getNumLoudspeakers()
    out = []
    for f in range(nFrames):
        sh = e.process(np.ascontiguousarray(x_all[list(sources), f * F:(f + 1) * F]), nSH)
        out.append(d.process(sh, nLS))
    return np.concatenate(out, 1)


def _scene_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    import torch
    from spatial_audio_framework_amd import parallel as P
    from oracle import oracle as O
    P.init(backend="gloo")
    mine = list(P.shard(12, world, rank))
    part = torch.from_numpy(render_partial_field(O, mine))
    both = part.clone()
    P.sum_partial_fields(both)                               # all ranks get the field
    P.sum_partial_fields(part, root=0)                       # only rank 0 does
    q.put((rank, both.numpy(), part.numpy() if rank == 0 else None))
    P.finalize()


def test_single_scene_sources_sharded_over_two_ranks():
    """One sound field, its 12 sources split over 2 ranks: the reduce of the ranks' loudspeaker feeds equals the feeds
    of the whole scene rendered in one process (every stage is linear) — SURVEY §8e-ii."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scene_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, str(ROOT))
    from oracle import oracle as O
    ref = render_partial_field(O, list(range(12)))
    scale = np.abs(ref).max()
    assert scale > 0.05
    for rank, both, rooted in got:
        assert np.abs(both - ref).max() < 2e-6 * max(1.0, scale)
        if rank == 0:
            assert np.abs(rooted - ref).max() < 2e-6 * max(1.0, scale)


def test_shard_partition_properties():
    sys.path.insert(0, str(ROOT))
    from spatial_audio_framework_amd.parallel import shard
    for n in (0, 1, 7, 32, 2048):
        for world in (1, 2, 3, 8):
            parts = [list(shard(n, world, r)) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def _bench(*argv, env=None):
    import subprocess
    e = dict(os.environ); e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher) starts two ranks itself before touching the GPU; --dry-run keeps the launch,
    rendezvous (127.0.0.1), barrier and MAX-over-ranks path and skips the kernels, so this runs on the CPU with gloo."""
    import json
    r = _bench("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert abs(line["max_elapsed"] - 0.002) < 1e-9           # the slowest rank (rank 1) defines the job time
    # what the collective library saw (bench.py rank_proof): an all-reduce of ones = the ranks that took part, every rank's device
    # identity gathered, and the per-rank throughput (a straggler is visible)
    assert line["rccl_ranks"] == 2 and line["get_world_size"] == 2 and line["distinct_devices"] == 2
    assert [d["rank"] for d in line["devices"]] == [0, 1] and line["per_rank_value"] == [1000.0, 500.0]


def test_bench_rejects_world_size_mismatch():
    r = _bench("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
