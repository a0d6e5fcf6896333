#!/usr/bin/env python3
"""One sound field fed by more sources than one encoder instance takes (BASELINE configs[4]: 2048 sources), sources sharded
over ranks.  Every stage is linear, so each rank encodes its sources (instances of up to 64), sums the
instances' SH blocks, decodes ONCE, and the ranks' loudspeaker blocks are summed onto rank 0 — the only collective of
the path (RCCL reduce over xGMI, `parallel.sum_partial_fields`).

    python tools/single_scene.py --sources 2048 --frames 16                                   # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/single_scene.py --sources 2048 --frames 16

Prints one JSON line on rank 0: scene-frames/s (a frame = one 512-sample block of the whole scene) and the reduce's share.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
F, ORDER, NSH = 512, 7, 64


def build(api, sources, nF):
    """encoder instances for this rank's `sources` (global indices), one decoder; returns (encBatch, decBatch, nInst)"""
    groups = [sources[i:i + 64] for i in range(0, len(sources), 64)]
    encs = []
    for g in groups:
        e = api.AmbiEnc(F); e.init(48000); e.setOutputOrder(ORDER); e.setNumSources(len(g)); e.setNormType(1); e.setEnablePostScaling(0)
        for j, s in enumerate(g):
            e.setSourceAzi_deg(j, float((53 * s) % 360 - 180)); e.setSourceElev_deg(j, float((29 * s) % 120 - 60))
        encs.append(e)
    d = api.AmbiDec(F)
    d.setNormType(1); d.setChOrder(1); d.setMasterDecOrder(ORDER); d.setOutputConfigPreset(29)
    d.setDecMethod(0, 1); d.setDecMethod(1, 1); d.initCodec(); d.init(48000); d.setDecOrderAllBands(ORDER)
    return api.AmbiEncBatch(encs, nF), api.AmbiDecBatch([d], nF), len(groups), (encs, d)


def render(eb, db, nI, x, sh, sh1, out, nF):
    """x [nI][nF][64][F] -> out [1][nF][64][F] (this rank's partial loudspeaker feeds)"""
    st = (nF * 64 * F, 64 * F, F)
    eb.process_ptr(x.data_ptr(), st, 64, sh.data_ptr(), st, 64, nF)
    torch.sum(sh, dim=0, keepdim=True, out=sh1)              # instances of one rank share the decoder
    db.process_ptr(sh1.data_ptr(), st, out.data_ptr(), st, nF)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sources", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=16, help="blocks per call")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backend", default=None, help="nccl (default on GPUs) or gloo (several ranks on one GPU)")
    ap.add_argument("--device", type=int, default=None)
    args = ap.parse_args()
    from spatial_audio_framework_amd import api, parallel as P
    from spatial_audio_framework_amd._lib import load
    world, rank, local_rank = P.env_world()
    dev = torch.device("cuda", args.device if args.device is not None else (local_rank if world > 1 else 0))
    torch.cuda.set_device(dev)
    P.init(backend=args.backend or "nccl", device=dev if (args.backend or "nccl") == "nccl" else None)
    L = load(); L.saf_hip_set_device(dev.index)
    api.set_stream(torch.cuda.current_stream().cuda_stream)
    mine = list(P.shard(args.sources, world, rank))
    nF = args.frames
    eb, db, nI, keep = build(api, mine, nF)
    g = torch.Generator(device=dev); g.manual_seed(77 + rank)
    x = torch.rand(nI, nF, 64, F, device=dev, generator=g) * 2 - 1
    sh = torch.zeros(nI, nF, 64, F, device=dev); sh1 = torch.zeros(1, nF, 64, F, device=dev); out = torch.zeros(1, nF, 64, F, device=dev)
    for _ in range(args.warmup):
        render(eb, db, nI, x, sh, sh1, out, nF); P.sum_partial_fields(out, root=0)
    torch.cuda.synchronize(); P.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        render(eb, db, nI, x, sh, sh1, out, nF); P.sum_partial_fields(out, root=0)
    torch.cuda.synchronize(); P.barrier()
    dt = P.max_over_ranks(time.perf_counter() - t0, device=dev)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        render(eb, db, nI, x, sh, sh1, out, nF)
    torch.cuda.synchronize(); P.barrier()
    dt_nored = P.max_over_ranks(time.perf_counter() - t1, device=dev)
    if rank == 0:
        print(json.dumps({"metric": "single-scene frames/s", "value": round(args.steps * nF / dt, 1), "unit": "scene-frames/s", "n_gpus": world,
                          "sources": args.sources, "sources_per_rank": len(mine), "frames_per_call": nF, "ms_per_call": round(dt / args.steps * 1e3, 4),
                          "ms_per_call_without_reduce": round(dt_nored / args.steps * 1e3, 4),
                          "reduce_bytes_per_call": 64 * F * 4 * nF, "scaling": "strong", "data": "synthetic"}))
    P.finalize()


if __name__ == "__main__":
    main()
