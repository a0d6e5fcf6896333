#!/usr/bin/env python3
"""bench.py — audio-frames/s of the 7th-order ambi_dec hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (torch.distributed.run, one process
per GPU, RCCL) BEFORE this process touches the GPU, relays rank 0's line and exits with the launcher's code.  Under a
launcher (WORLD_SIZE set) a mismatch between WORLD_SIZE and --gpus is an error.

Workload (BASELINE.json configs[1]): order-7 ambi_dec, 64 SH in -> 64 loudspeakers out (SphCovering-64, SAD,
maxrE, energy-preserving), 512-sample blocks, synthetic uniform noise resident in HBM.  One *step* = one batched pass of
the hot path over `instances` independent decoder instances x `frames_per_call` consecutive blocks.  Instances are
independent, so N GPUs run N times the instances (weak scaling, no data-path collective); ranks only meet in the
barriers and the max-reduce of the elapsed time.

`value` is measured on the GENERAL form of the library's block path (saf_hip_ambi_dec_setTimeDomainPath(2)): every SH
channel runs afSTFT analysis -> per-band gains -> afSTFT synthesis in one kernel (spectra stay on chip), then one
time-domain MFMA GEMM applies the dense decoder.  It holds for any per-band order / max-rE / normalisation assignment.
Extra keys (same sizes): `default_dispatch` (what the library does by default on THIS workload: its weights are the same
in every band, so the transforms are skipped), `per_band_orders_workload` (every band its own order: default dispatch =
the general form), `two_decoder_workload` (SAD below / EPAD above the transition: two dense matrices; default = the two-output equaliser form, `transform_form` beside it),
`transform_path` (round 1's three-kernel path, spectra through HBM), `other_configs` (BASELINE configs[2..4], short
regions), `cpu_baseline` (1 core) and `cpu_baseline_allcores` (one oracle instance per core).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FRAME = 512
NCH = 64
ALG_BYTES_PER_FRAME = {            # algorithmic bytes of each kernel per 512-sample frame of 64 channels
    "afstft_eq": 2 * NCH * FRAME * 4,                             # samples in + equalised SH signals out (one dense decoder)
    "band_gemm": 2 * NCH * FRAME * 4,                             # equalised SH signals in + loudspeaker samples out
    "dec_stream": 2 * NCH * FRAME * 4,                            # the same product as the kernel that runs beside the equaliser
    "afstft_eq_coop": 2 * NCH * FRAME * 4,                        # equaliser and decode in one launch, z handed over on chip (samples in, samples out)
    "afstft_eq_decode": 2 * NCH * FRAME * 4,                      # small launches: equaliser and decode in one launch (samples in, samples out)
    "afstft_analysis": NCH * FRAME * 4 + 133 * NCH * 4 * 8,      # transform path: samples in + spectra out
    "afstft_synthesis": 133 * NCH * 4 * 8 + NCH * FRAME * 4,     # transform path: spectra in + samples out
}
PATH_BYTES_PER_FRAME = 2 * NCH * FRAME * 4                        # 262 144 B / frame
GEMM_FLOP_PER_FRAME = 2 * 64 * 64 * FRAME                         # dense decode in the time domain: 4.19 MFLOP / frame
HBM_PEAK_GBS = 8000.0                                             # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3                                      # dense fp32-input MFMA peak


def make_decoder(cls, m0=1, m1=1):
    d = cls(FRAME)
    d.setNormType(1)                 # NORM_N3D
    d.setChOrder(1)                  # CH_ACN
    d.setMasterDecOrder(7)
    d.setOutputConfigPreset(29)      # LOUDSPEAKER_ARRAY_PRESET_SPH_COV_64
    d.setDecMethod(0, m0)            # 1 = SAD below the transition
    d.setDecMethod(1, m1)            # above
    d.initCodec()
    d.init(48000)
    d.setDecOrderAllBands(7)
    return d


def band_orders(seed=9):
    """an arbitrary order per band, every order 1..7 present (what setDecOrder / the microphone presets produce)"""
    import numpy as np
    o = np.random.default_rng(seed).integers(1, 8, 133)
    o[:7] = np.arange(1, 8)
    return [int(v) for v in o]


# ------------------------------------------------------------------------------------------------ CPU legs
def cpu_run(seconds_budget):
    """the CPU oracle (port of the reference path, scalar, one thread) on the headline configuration; (frames, seconds, stage times)"""
    import numpy as np
    from oracle import oracle as O
    d = make_decoder(O.AmbiDec)
    rng = np.random.default_rng(0)
    x = (rng.random((8, NCH, FRAME), dtype=np.float32) * 2 - 1)
    for i in range(8):
        d.process(x[i], NCH)                       # warm-up (also past the filterbank transient)
    n, t0 = 0, time.perf_counter()
    while True:
        for i in range(8):
            d.process(x[i], NCH)
        n += 8
        if time.perf_counter() - t0 > seconds_budget:
            break
    return n, time.perf_counter() - t0, d.stageTimes()


def cpu_baseline(seconds_budget=8.0):
    n, dt, (f, g, b) = cpu_run(seconds_budget)
    return {"value": round(n / dt, 1), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} consecutive 512-sample blocks of one order-7 64->64 instance, {dt:.1f} s on 1 core "
                      f"(gcc -O3 -march=native, no BLAS; stage split afSTFT fwd/decode/afSTFT bwd = "
                      f"{1e3 * f / (n + 8):.3f}/{1e3 * g / (n + 8):.3f}/{1e3 * b / (n + 8):.3f} ms per block)"}


def cpu_topology():
    """physical cores of this process's affinity mask (first hardware thread of every (package, core) pair) and the cgroup CPU quota"""
    aff = sorted(os.sched_getaffinity(0))
    first = {}
    for c in aff:
        try:
            base = Path(f"/sys/devices/system/cpu/cpu{c}/topology")
            key = (int((base / "physical_package_id").read_text()), int((base / "core_id").read_text()))
        except Exception:
            key = (0, c)
        first.setdefault(key, c)
    quota = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = Path(f).read_text().split()
            if f.endswith("cpu.max"):
                quota = None if t[0] == "max" else float(t[0]) / float(t[1])
            else:
                q = float(t[0]); per = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                quota = None if q <= 0 else q / per
            break
        except Exception:
            continue
    return aff, sorted(first.values()), quota


def cpu_point(cores, seconds_budget):
    """one oracle process pinned to each of `cores`; (frames/s of the set, frames, slowest seconds) or None"""
    procs = [subprocess.Popen([sys.executable, str(Path(__file__).resolve()), "--cpu-worker", str(seconds_budget), "--pin-core", str(c)],
                              stdout=subprocess.PIPE, text=True) for c in cores]
    outs = [p.communicate()[0] for p in procs]          # every child is reaped, also when one of them failed
    if any(p.returncode != 0 for p in procs):
        return None
    frames, slowest = 0, 0.0
    for out in outs:
        n, dt = out.split()[-2:]
        frames += int(n); slowest = max(slowest, float(dt))
    return frames / slowest, frames, slowest


def cpu_baseline_allcores(seconds_budget=5.0):
    """N independent oracle instances, one process pinned to each of N PHYSICAL cores, for N = 1, 8, 64, the cgroup's
    CPU quota and all physical cores of the affinity mask; `value` is the best point.  (A box whose cgroup allows 16 CPUs runs 256
    pinned processes at 1/16 speed each: round 2's 88 frames/s per core.)"""
    aff, phys, quota = cpu_topology()
    counts = sorted({n for n in (1, 8, 64, int(quota) if quota else 0, len(phys)) if 1 <= n <= len(phys)})
    sweep, best = [], None
    for n in counts:
        r = cpu_point(phys[:n], seconds_budget)
        if r is None:
            continue
        sweep.append({"processes": n, "frames_per_s": round(r[0], 1), "per_core": round(r[0] / n, 1), "blocks": r[1], "seconds": round(r[2], 2)})
        if best is None or r[0] > best[0]:
            best = (r[0], n)
    if best is None:
        return None
    return {"value": round(best[0], 1), "unit": "frames/s", "cores": best[1], "kind": "port",
            "physical_cores": len(phys), "logical_cpus": len(aff), "cpu_max": quota, "sweep": sweep,
            "sample": f"independent order-7 64->64 instances, one scalar oracle process pinned to each of N physical cores (first hardware thread of every "
                      f"(package, core) pair of the affinity mask: {len(phys)} physical cores, {len(aff)} logical CPUs, cgroup cpu.max = {quota}); "
                      f"{seconds_budget:.0f} s per point; `value` = the best point of `sweep` (N = {best[1]})"}


# ------------------------------------------------------------------------------------------------ the unchanged signature
def host_pointer_section(L, api):
    """The drop-in signature itself (examples/include/ambi_dec.h:161): one 512-sample block per ambi_dec_process call, planar HOST
    pointers in and out (pinned staging, kernels, stream sync inside the call: every sample crosses PCIe twice).  (i) one handle,
    per-call latency; (ii) H handles driven by H host threads at once — every handle has a stream of its own inside the library."""
    import threading
    import numpy as np

    def handle(seed):
        d = make_decoder(api.AmbiDec)
        for b, o in enumerate(band_orders(seed)):       # every band its own order: the general form of the block path
            d.setDecOrder(o, b)
        x = (np.random.default_rng(seed).random((NCH, FRAME), dtype=np.float32) * 2 - 1)
        y = np.zeros((NCH, FRAME), np.float32)
        px, py = api._rows(x), api._rows(y)
        return d, x, y, px, py, (lambda: L.ambi_dec_process(d.h, px, py, NCH, NCH, FRAME))

    h0 = handle(1)
    call = h0[-1]
    for _ in range(50):
        call()
    ts = []
    for _ in range(1500):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    ts = np.sort(np.array(ts)) * 1e6
    one = {"median_us": round(float(ts[len(ts) // 2]), 1), "p99_us": round(float(ts[int(len(ts) * 0.99)]), 1), "min_us": round(float(ts[0]), 1),
           "frames_per_s": round(1e6 / float(ts.mean()), 1), "block_us": round(FRAME / 48000 * 1e6, 1), "path": h0[0].lastPath()}
    out = {"one_handle": one}
    # H host threads, natively (tools/hostbench.c): Python threads would serialise on the interpreter lock between calls
    so = Path(f"/tmp/libhostbench_{os.getpid()}.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", str(ROOT / "tools" / "hostbench.c"), "-o", str(so)])
    HB = C.CDLL(str(so))
    HB.hostbench_run.restype = C.c_double
    HB.hostbench_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)] + [C.c_int] * 5
    fnp = C.cast(L.ambi_dec_process, C.c_void_p)
    for H in (8, 32):
        hs = [handle(10 + i) for i in range(H)]
        for hh in hs:
            for _ in range(20):
                hh[-1]()
        M = 400
        hv = (C.c_void_p * H)(*[hh[0].h for hh in hs])
        iv = (C.c_void_p * H)(*[C.cast(hh[3], C.c_void_p) for hh in hs])
        ov = (C.c_void_p * H)(*[C.cast(hh[4], C.c_void_p) for hh in hs])
        dt = HB.hostbench_run(fnp, hv, iv, ov, H, NCH, NCH, FRAME, M)
        assert dt > 0
        ref = make_decoder(api.AmbiDec)             # the threaded runs must leave the right samples behind: handle 0 again, alone
        for b, o in enumerate(band_orders(10)):
            ref.setDecOrder(o, b)
        yref = None
        for _ in range(20 + M):
            yref = ref.process(hs[0][1], NCH)
        out[f"threads_{H}"] = {"handles": H, "host_threads": H, "calls_per_handle": M, "frames_per_s": round(H * M / dt, 1),
                               "vs_one_handle": round(H * M / dt / one["frames_per_s"], 2), "us_per_call_per_handle": round(1e6 * dt / M, 1),
                               "last_block_equals_single_threaded_run": bool(np.array_equal(yref, hs[0][2]))}
        del hs
    so.unlink()
    out["note"] = ("ambi_dec_process(h, const float* const* in, float** out, 64, 64, 512) with host pointers, order 7 -> 64 loudspeakers, every band its own "
                   "order; one block per call; the H-handle figures are H native threads (tools/hostbench.c) calling the C entry point, one handle each; "
                   f"GPU_MAX_HW_QUEUES = {os.environ.get('GPU_MAX_HW_QUEUES')} (HIP maps streams onto this many hardware queues, default 4)")
    return out


# ------------------------------------------------------------------------------------------------ multi-rank launch
def spawn_ranks(n, argv):
    """start n ranks of this script (one per GPU) with torch.distributed.run; this process has not touched the GPU"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + argv
    return subprocess.call(cmd, env=env)


def rank_proof(torch, P, world, rank, dev, rdev, backend):
    """What the collective library saw: an all-reduce of ones on the ranks' own devices (= number of ranks that took part) and
    every rank's device identity (name, PCI bus id / uuid), gathered.  Single process: the same fields without a collective."""
    import torch.distributed as dist
    if dev is not None:
        pr = torch.cuda.get_device_properties(dev)
        ident = {"rank": rank, "device": f"cuda:{dev.index}", "name": pr.name,
                 "pci": f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}",
                 "uuid": str(getattr(pr, "uuid", ""))}
    else:
        ident = {"rank": rank, "device": "cpu", "name": "cpu (dry run)", "pci": f"pid:{os.getpid()}", "uuid": f"pid:{os.getpid()}"}
    if not dist.is_initialized():
        return {"collective_backend": None, "rccl_ranks": 1, "distinct_devices": 1, "devices": [ident]}
    ones = torch.ones(1, dtype=torch.float32, device=rdev)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    return {"collective_backend": backend + (" (RCCL)" if backend == "nccl" else ""), "rccl_ranks": int(round(float(ones.item()))),
            "get_world_size": dist.get_world_size(), "distinct_devices": len({(d["pci"], d["uuid"]) for d in idents}), "devices": idents}


def per_rank_values(P, world, local_value, rdev):
    import torch.distributed as dist
    import torch
    if not dist.is_initialized():
        return [round(local_value, 1)]
    t = torch.tensor([float(local_value)], dtype=torch.float64, device=rdev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [round(float(v.item()), 1) for v in out]


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import torch
    from spatial_audio_framework_amd import parallel as P

    world, rank, local_rank = P.env_world()
    if args.dry_run:
        # launch / rendezvous / reduction path without the GPU (tests/test_dist_cpu.py): gloo, no kernels
        P.init(backend="gloo")
        proof = rank_proof(torch, P, world, rank, None, "cpu", "gloo")
        if proof["rccl_ranks"] != args.gpus:
            print(f"bench.py: the collective saw {proof['rccl_ranks']} ranks, --gpus is {args.gpus}", file=sys.stderr)
            P.finalize()
            return 3
        P.barrier()
        mine = 0.001 * (1 + rank)
        elapsed = P.max_over_ranks(mine)
        prv = per_rank_values(P, world, 1.0 / mine, "cpu")
        if rank == 0:
            line = {"metric": "audio-frames/sec (512-sample, 128-ch, 7th-order ambi_dec)", "value": None, "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "dry_run": True, "max_elapsed": elapsed, "per_rank_value": prv}
            line.update(proof)
            print(json.dumps(line), flush=True)
        P.finalize()
        return 0
    if torch.cuda.device_count() <= 0:          # (counting devices does not initialise the GPU)
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")

    # CPU legs first (rank 0 of a single-GPU run), before this process initialises the GPU: the all-cores leg starts child processes
    cpu = cpu_all = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline()
        cpu_all = cpu_baseline_allcores()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")

    dev = torch.device("cuda", local_rank if (world > 1 and not args.share_gpu) else 0)
    torch.cuda.set_device(dev)
    if args.share_gpu:
        P.init(backend="gloo")                  # rehearsal: N ranks on one GPU
    else:
        P.init(backend="nccl", device=dev)      # RCCL; ranks only meet in the barriers and the MAX of the elapsed time
    rdev = "cpu" if args.share_gpu else dev     # where the timing reduction lives

    proof = rank_proof(torch, P, world, rank, dev, rdev, "gloo" if args.share_gpu else "nccl")
    if proof["rccl_ranks"] != args.gpus:
        print(f"bench.py: the collective saw {proof['rccl_ranks']} ranks, --gpus is {args.gpus}", file=sys.stderr)
        P.finalize()
        return 3

    from spatial_audio_framework_amd import api
    from spatial_audio_framework_amd._lib import load
    L = load()
    L.saf_hip_set_device(dev.index)
    api.set_stream(torch.cuda.current_stream().cuda_stream)

    nI, nF = args.instances, args.frames_per_call
    decs = [make_decoder(api.AmbiDec) for _ in range(nI)]
    batch = api.AmbiDecBatch(decs, nF)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    # two alternating input sets so consecutive steps do not re-read identical data
    xs = [torch.rand(nI, nF, NCH, FRAME, device=dev, generator=g) * 2 - 1 for _ in range(2)]
    y = torch.zeros(nI, nF, NCH, FRAME, device=dev)
    st = (nF * NCH * FRAME, NCH * FRAME, FRAME)

    region_events_ms = {}

    def timed_region(bt, mode, kernels, steps, warmup, expect_path=None):
        """`steps` timed passes on block path `mode` (saf_hip_ambi_dec_setTimeDomainPath) from a cleared filterbank state;
        returns (max-over-ranks seconds, {kernel: (avg launch ms, launches)})"""
        L.saf_hip_ambi_dec_setTimeDomainPath(mode)
        bt.clear()
        for i in range(warmup):
            bt.process_ptr(xs[i & 1].data_ptr(), st, y.data_ptr(), st, nF)
        torch.cuda.synchronize()
        L.saf_hip_profile_reset()
        L.saf_hip_profile_enable(0 if args.no_profile else 1)
        P.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.saf_hip_stopwatch_start()                     # two HIP events on the stream the library launches on
        for i in range(steps):
            bt.process_ptr(xs[i & 1].data_ptr(), st, y.data_ptr(), st, nF)
        ev_ms = L.saf_hip_stopwatch_stop_ms()
        torch.cuda.synchronize()
        P.barrier()
        dt = time.perf_counter() - t0
        region_events_ms.setdefault(mode, ev_ms)        # the headline region runs first
        region_events_ms.setdefault("local_s", dt)      # this rank's own time of the headline region
        L.saf_hip_profile_enable(0)
        dt = P.max_over_ranks(dt, device=rdev)
        per = {}
        for k in kernels:
            tot = C.c_double()
            n = L.saf_hip_profile_read(k.encode(), C.byref(tot))
            if n:
                per[k] = (tot.value / n, n)
        assert bt.lastPath() == ((0 if mode == 0 else 1) if expect_path is None else expect_path)
        return dt, per

    def region_dict(dt, per, note):
        return {"value": round(world * nI * nF * args.steps / dt, 1), "unit": "frames/s", "ms_per_step": round(1e3 * dt / args.steps, 4),
                "kernels_ms": {k: round(v[0], 5) for k, v in per.items()}, "note": note}

    EQK = ("afstft_eq", "afstft_eq_coop", "afstft_eq_decode", "dec_stream", "band_gemm")
    torch.cuda.synchronize()                    # the synthetic inputs were produced on torch's stream, the library launches on its own
    elapsed, general_kernels = timed_region(batch, args.path_mode, EQK if args.path_mode else ("afstft_analysis", "band_gemm", "afstft_synthesis"), args.steps, args.warmup)
    extra = {}
    if not args.no_extra_paths:
        w2 = max(2, args.warmup)
        dt, per = timed_region(batch, 1, EQK, args.steps, w2)
        extra["default_dispatch"] = region_dict(dt, per, "same workload, library default (mode 1): this workload's per-channel weights are the same in all 133 bands, so every "
                                                         "channel skips the transforms (FFT / inverse FFT cancel, hybrid split + merge = its 3-hop delay); same outputs (parity tests)")
        dt, per = timed_region(batch, 0, ("afstft_analysis", "band_gemm", "afstft_synthesis"), args.steps, w2)
        extra["transform_path"] = region_dict(dt, per, "same workload on round 1's three-kernel path (mode 0): analysis -> per-band MFMA GEMM -> synthesis, spectra cross HBM four times")
        orders = band_orders()
        for d in decs:
            for b, o in enumerate(orders):
                d.setDecOrder(o, b)
        dt, per = timed_region(batch, 1, EQK, args.steps, w2)
        extra["per_band_orders_workload"] = region_dict(dt, per, "every band its own decoding order (1..7, max-rE on): library default = the general form, 14 different per-band matrices")
        for d in decs:
            d.setDecOrderAllBands(7)
        del batch
        decs2 = [make_decoder(api.AmbiDec, 1, 3) for _ in range(nI)]          # SAD below / EPAD above 800 Hz: two dense matrices
        batch2 = api.AmbiDecBatch(decs2, nF)
        TRK = ("afstft_analysis", "band_gemm", "afstft_synthesis")
        dt, per = timed_region(batch2, 1, EQK, args.steps, w2, expect_path=1)
        extra["two_decoder_workload"] = region_dict(dt, per, "SAD below / EPAD above the 800 Hz transition (two different dense matrices), library default = the equaliser path: the "
                                                             "kernel emits two signals per channel (1 forward, 2 inverse transforms), the time-domain GEMM has two terms; "
                                                             "algorithmic traffic 6 x 131 072 B per frame")
        dt, per = timed_region(batch2, 0, TRK, args.steps, w2)
        extra["two_decoder_workload"]["transform_form"] = region_dict(dt, per, "mode 0: the same workload on the three-kernel transform path")
        del batch2, decs2
    L.saf_hip_ambi_dec_setTimeDomainPath(1)

    frames_total = world * nI * nF * args.steps
    value = frames_total / elapsed
    prv = per_rank_values(P, world, nI * nF * args.steps / region_events_ms["local_s"], rdev)      # a straggler shows here

    hostptr = None
    if world == 1 and not args.no_extra_paths:
        api.set_stream(None)                    # the host-pointer calls run on the handles' own streams
        hostptr = host_pointer_section(L, api)
        api.set_stream(torch.cuda.current_stream().cuda_stream)
    other = None
    if world == 1 and not args.no_other_configs and not args.no_extra_paths:      # single-GPU information: not part of a scaling run
        del xs, y
        torch.cuda.empty_cache()
        sys.path.insert(0, str(ROOT / "tools"))
        import bench_workloads as W
        import numpy as np
        i = np.arange(64) + 0.5
        fib = np.stack([np.mod(np.degrees(np.pi * (1.0 + 5.0 ** 0.5) * i), 360.0) - 180.0, np.degrees(np.arcsin(1.0 - 2.0 * i / 64))], 1)
        other = []
        for fn in (lambda: W.binauraliser_batch(L, torch, api), lambda: W.matrixconv(L, torch, api), lambda: W.powermap(L, torch, api), lambda: W.powermap_batch(L, torch, api),
                   lambda: W.enc_dec_chain(L, torch, api, fib)):
            other.append(fn())

    if rank == 0:
        roof = None
        if not args.no_profile and general_kernels:
            per = general_kernels
            dom = max(per, key=lambda k: per[k][0])
            avg_ms, nl = per[dom]
            frames_per_launch = nI * nF
            traffic, traffic_source = None, None
            tf = ROOT / "profiles" / "traffic_latest.json"
            if tf.exists():          # PMC bytes per launch from an EARLIER run of the same launch size (tools/gpu_profile.sh), not measured in this run
                try:
                    tj = json.loads(tf.read_text())
                    if tj.get("frames_per_launch") == frames_per_launch and dom in tj:
                        traffic = tj[dom].get("hbm_bytes_per_launch")
                        traffic_source = f"profiles/traffic_latest.json ({tj.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run')}); not measured in this run"
                except Exception:
                    traffic = None
            ach = ALG_BYTES_PER_FRAME[dom] * frames_per_launch / (avg_ms * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source}
            if dom == "afstft_eq":
                roof["limiter"] = ("vector-instruction issue, not HBM: ~813 M VALU wave-instructions per launch of 16 384 frames = 0.5 of the chip's "
                                   "measured vector-issue rate (profiles/r02_pmc_summary.txt, r02_valu_rate.txt); the kernel moves 262 144 B per frame, the floor of the path")
            gm = per.get("band_gemm")
            if gm:
                tfl = GEMM_FLOP_PER_FRAME * frames_per_launch / (gm[0] * 1e-3) / 1e12
                gb = ALG_BYTES_PER_FRAME["band_gemm"] * frames_per_launch / (gm[0] * 1e-3) / 1e9
                roof["band_gemm"] = {"hbm": {"achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBS, 4)},
                                     "mfma": {"achieved": round(tfl, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tfl / MFMA_F32_PEAK_TFLOPS, 4)}}
            roof["avg_launch_ms"] = round(avg_ms, 5)
            roof["launches"] = nl
            roof["kernels_ms"] = {k: round(v[0], 5) for k, v in per.items()}
            roof["path_hbm_frac"] = round(PATH_BYTES_PER_FRAME * (value / world) / 1e9 / HBM_PEAK_GBS, 4)
        line = {
            "metric": "audio-frames/sec (512-sample, 128-ch, 7th-order ambi_dec)",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ambi_dec order 7: 64 SH in -> 64 loudspeakers out (SphCovering-64, SAD, maxrE, energy-preserving, N3D/ACN), "
                                   "512-sample blocks, fs 48 kHz; batched device-resident entry point; general form of the block path "
                                   "(per-channel filterbank equaliser + one time-domain GEMM, saf_hip_ambi_dec_setTimeDomainPath(2))",
                       "instances_per_gpu": nI, "frames_per_step_per_instance": nF, "frames_per_step_per_gpu": nI * nF,
                       "parallelism": f"independent instances sharded over {world} GPU(s), no collective on the data path"},
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_allcores": cpu_all,
            "timed_region_check": {"host_clock_ms": round(1e3 * elapsed, 3), "hip_events_ms": round(region_events_ms.get(args.path_mode, 0.0), 3),
                                   "note": "the K timed steps between the barriers, by the host clock (used for `value`) and by two HIP events on the library's launch stream (saf_hip_stopwatch_*)"},
        }
        line.update(proof)
        line["per_rank_value"] = prv
        if hostptr is not None:
            line["host_pointer"] = hostptr
        line.update(extra)
        if other is not None:
            line["other_configs"] = other
        if cpu:
            line["speedup_vs_cpu_1core"] = round(value / cpu["value"], 1)
        if cpu_all:
            line["speedup_vs_cpu_allcores"] = round(value / cpu_all["value"], 1)
        print(json.dumps(line), flush=True)
    P.finalize()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--instances", type=int, default=int(os.environ.get("SAF_BENCH_INSTANCES", 256)), help="decoder instances per GPU")
    ap.add_argument("--frames-per-call", type=int, default=int(os.environ.get("SAF_BENCH_FRAMES", 64)), help="consecutive blocks per instance per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--no-extra-paths", action="store_true", help="only the headline timed region (used by the PMC passes)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short regions of the other BASELINE configs")
    ap.add_argument("--path-mode", type=int, default=2, help="block path of the headline region (saf_hip_ambi_dec_setTimeDomainPath); profiling passes use 0 / 1")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0 and the ranks meet over gloo (RCCL needs one device per rank)")
    ap.add_argument("--dry-run", action="store_true", help="exercise launch / rendezvous / reduction with gloo, no GPU work (CPU tests)")
    ap.add_argument("--cpu-worker", type=float, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--pin-core", type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker is not None:             # child of cpu_baseline_allcores: one oracle instance on one core, never touches the GPU
        if args.pin_core is not None:
            os.sched_setaffinity(0, {args.pin_core})
        n, dt, _ = cpu_run(args.cpu_worker)
        print(n, f"{dt:.4f}")
        return 0

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args.gpus, list(sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}; launch N ranks with --gpus N", file=sys.stderr)
        return 2
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
