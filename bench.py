#!/usr/bin/env python3
"""bench.py -- fused keyframes/s of the Map2DFusion hot path on MI355X.

One "step" = one pass of the hot path (Map2D::feed -> renderFrame: warp, weight,
Laplacian pyramid, per-tile max-weight select) over one synthetic 4000x3000 BGR keyframe
already resident in HBM.  Workload = BASELINE.json configs[1] geometry (SURVEY 8d cfg-2,
cfg-A: Map2D.Scale=1, 5 bands): serpentine sortie, 80 %/60 % overlap, yaw +-5 deg,
roll/pitch +-2 deg.

Launch: `python bench.py --gpus N --steps K --warmup W`, or under torchrun with one rank
per GPU.  With N>1 the mosaic tiles are sharded by spatial hash (no data-path collective
in feed); N concurrent sorties are fed interleaved and every rank renders the
frames that land on tiles it owns -> weak scaling; value = all frames / max-over-ranks time.

Rank 0 prints ONE JSON line (contract in the task statement), including
  roofline     : dominant kernel, algorithmic bytes / HIP-event time on the map's stream
  cpu_baseline : the oracle (CPU port of MultiBandMap2DCPU) timed on a bounded sample.
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def load_package():
    name = "pi_slam_fusion_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "pi-slam-fusion_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(wl, cam, poses, prep, frames_host, force_float, budget_s=15.0, max_frames=40):
    """The oracle (kind 'port': this repo's C restatement of MultiBandMap2DCPU, 1 thread,
    like the reference's single render thread on a stock OpenCV 2.4.9) on the first frames
    of the same workload."""
    sys.path.insert(0, ROOT)
    from oracle import orc
    o = orc.OracleMap(force_float=force_float)
    assert o.prepare(wl.IDENTITY_PLANE, cam, prep)
    n, t0 = 0, time.perf_counter()
    while n < max_frames:
        o.feed(frames_host[n % len(frames_host)], poses[n])
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 4), "unit": "keyframes/s", "cores": 1, "kind": "port",
            "sample": "first %d frames of the same 4000x3000 workload, 1 thread, %.1f s" % (n, dt)}


def map2dcpu_rates(pf, wl, cam, poses, prep, frames_dev, frames_host, budget_s=8.0):
    """The reference's other Map2D type on the same workload (BASELINE.json: "Map2DCPU/MultiBandMap2DCPU timed on
    the host cores"): single-band Map2DCPU, the oracle restatement on 1 core and the HIP path on this GPU."""
    sys.path.insert(0, ROOT)
    from oracle import orc
    o = orc.OracleMap(single_band=True)
    assert o.prepare(wl.IDENTITY_PLANE, cam, prep)
    n, t0 = 0, time.perf_counter()
    while n < 40 and time.perf_counter() - t0 < budget_s:
        o.feed(frames_host[n % len(frames_host)], poses[n]); n += 1
    dt = time.perf_counter() - t0
    out = {"cpu": {"value": round(n / dt, 4), "unit": "keyframes/s", "cores": 1, "kind": "port",
                   "sample": "first %d frames, Map2DCPU restatement, 1 thread, %.1f s" % (n, dt)}}
    m = pf.Map2D.create(pf.TypeCPU, False)
    assert m.prepare(wl.IDENTITY_PLANE, cam, prep)
    k = min(len(poses), 120)
    for i in range(20):
        m.feed_device(frames_dev[i % len(frames_dev)].data_ptr(), cam[1], cam[0], poses[i])
    m.sync()
    t0 = time.perf_counter()
    for i in range(20, k):
        m.feed_device(frames_dev[i % len(frames_dev)].data_ptr(), cam[1], cam[0], poses[i])
    m.sync()
    out["gpu"] = {"value": round((k - 20) / (time.perf_counter() - t0), 1), "unit": "keyframes/s", "frames": k - 20}
    return out


def cpu_baseline_allcores(force_float, budget_s=10.0):
    """BASELINE.md B2 ("generous"): the same oracle with its row / tile loops under OpenMP, all host
    cores, in a child process (the library flavour is chosen per process; the child never touches the GPU)."""
    import subprocess
    code = (
        "import sys, time, json, os; sys.path.insert(0, %r)\n"
        "nth = min(16, len(os.sched_getaffinity(0)))        # a 1-GPU box gives this job a 16-CPU share\n"
        "os.environ['OMP_NUM_THREADS'] = str(nth)\n"
        "import bench, importlib\n"
        "from oracle import orc\n"
        "orc.use_openmp()\n"
        "pf = bench.load_package(); wl = importlib.import_module('pi_slam_fusion_amd.workloads')\n"
        "cam, poses = wl.cfg2(64)\n"
        "o = orc.OracleMap(force_float=%d); assert o.prepare(wl.IDENTITY_PLANE, cam, poses[:20])\n"
        "fr = [wl.noise_frame(3000, 4000, k) for k in range(2)]\n"
        "n, t0 = 0, time.perf_counter()\n"
        "while n < 64 and time.perf_counter() - t0 < %f:\n"
        "    o.feed(fr[n %% 2], poses[n]); n += 1\n"
        "dt = time.perf_counter() - t0\n"
        "print(json.dumps({'value': round(n / dt, 4), 'unit': 'keyframes/s', 'cores': nth, 'kind': 'port',\n"
        "                  'sample': 'first %%d frames, OpenMP over rows/tiles, %%.1f s' %% (n, dt)}))\n"
    ) % (ROOT, force_float, budget_s)
    try:
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120)
        return json.loads(out.stdout.decode().strip().splitlines()[-1])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--int16", action="store_true", help="reference default pyramids (CV_16SC3) instead of ForceFloat")
    ap.add_argument("--scale", type=float, default=1.0, help="Map2D.Scale (1 = cfg-A, 0.5 = shipped Default.cfg)")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frames kept in HBM")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--fused", type=int, default=None, help="pf_options.fused (default: the library's default)")
    ap.add_argument("--event-every", type=int, default=16,
                    help="HIP events around every n-th launch of the dominant kernel in the timed region "
                         "(each event pair costs stream time; 0 = none, no roofline)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one rank per GPU)")
    ndev = torch.cuda.device_count()
    dev = local % max(ndev, 1)          # rehearsal on a 1-GPU box: several ranks share the card (PF_DIST_BACKEND=gloo)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("PF_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    pf = load_package()
    wl = importlib.import_module("pi_slam_fusion_amd.workloads")
    force_float = 0 if args.int16 else 1
    K, W, N = args.steps, args.warmup, world
    cam = [4000, 3000, 3000, 3000, 2000, 1500]
    height = 100.0
    n_traj = K + W
    block = 128                                     # spatial-hash cell edge in tiles
    extra = {} if args.fused is None else {"fused": args.fused}
    opt = pf.default_options(force_float=force_float, scale=args.scale, device=dev,
                             shard_rank=rank, shard_count=N, shard_block=block, **extra)
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False, options=opt)

    # one sortie per rank; sortie j is flown inside a hash cell owned by rank j
    # 16 rows of 20 frames (~850 m x 400 m) fit one 128-tile hash cell; longer runs fly the sortie again
    base = wl.serpentine(cam, height, n_traj, max_rows=16)
    if N == 1:
        prep = base[:20]
        assert m.prepare(wl.IDENTITY_PLANE, cam, prep)
        sorties = [base]
    else:
        span = 6000.0
        prep = [[x, y, -height, 0, 0, 0, 1] for x in (-span, span) for y in (-span, span)]
        assert m.prepare(wl.IDENTITY_PLANE, cam, prep)
        dims, geo = m.grid()
        ele = geo[4]
        xs = np.array([p[0] for p in base]); ys = np.array([p[1] for p in base])
        ext = max(xs.max() - xs.min(), ys.max() - ys.min()) + 2 * 140.0
        assert ext < block * ele, "sortie does not fit one hash cell"
        cells, want = {}, set(range(N))
        for cy in range(dims[1] // block):
            for cx in range(dims[0] // block):
                r = pf.tile_owner(opt, cx * block, cy * block)
                if r in want and r not in cells:
                    cells[r] = (cx, cy)
        assert len(cells) == N, "spatial hash left a rank without a cell: %s" % cells
        sorties = []
        for j in range(N):
            cx, cy = cells[j]
            ox = geo[0] + (cx + 0.5) * block * ele - 0.5 * (xs.max() + xs.min())
            oy = geo[1] + (cy + 0.5) * block * ele - 0.5 * (ys.max() + ys.min())
            sorties.append([[p[0] + ox, p[1] + oy] + p[2:] for p in base])

    # size the tile store for the sortie up front (std::vector::reserve for HBM; `spreadMap` still grows the grid):
    # hipMalloc and the driver's page clearing then happen here, not between two keyframes of the timed region
    ele_m = m.grid()[1][4]
    sx = [p[0] for p in sorties[rank]]; sy = [p[1] for p in sorties[rank]]
    foot = 1.2 * max(cam[0], cam[1]) * height / cam[2]
    m.reserve_tiles(int(((max(sx) - min(sx) + foot) / ele_m + 2) * ((max(sy) - min(sy) + foot) / ele_m + 2) * 1.1) + 64)

    # synthetic frames, resident in HBM before the timed region
    g = torch.Generator(device="cuda"); g.manual_seed(1234 + rank)
    frames = [torch.randint(0, 256, (3000, 4000, 3), dtype=torch.uint8, device="cuda", generator=g)
              for _ in range(args.distinct)]
    torch.cuda.synchronize()

    # frame routing (untimed): a rank needs a frame's pixels only if the frame's canvas holds one of
    # its tiles; every rank still gets every pose (geometry-only feed) so the grid advances identically
    def needs_pixels(pose):
        if N == 1:
            return True
        dims, geo = m.grid()
        pts = pf.footprint(cam, pf.se3_mul(pf.se3_inverse(wl.IDENTITY_PLANE), pose))
        if pts is None:
            return False
        inv = 1.0 / geo[4]
        x0 = int(np.floor((pts[:, 0].min() - geo[0]) * inv)); x1 = int(np.ceil((pts[:, 0].max() - geo[0]) * inv))
        y0 = int(np.floor((pts[:, 1].min() - geo[1]) * inv)); y1 = int(np.ceil((pts[:, 1].max() - geo[1]) * inv))
        return any(pf.tile_owner(opt, x + dims[2], y + dims[3]) == rank for y in range(y0, y1) for x in range(x0, x1))

    need = [[needs_pixels(sorties[j][k]) for k in range(n_traj)] for j in range(N)]

    def run(lo, hi):
        for k in range(lo, hi):
            for j in range(N):
                if need[j][k]:
                    ok = m.feed_device(frames[(k + j) % len(frames)].data_ptr(), 3000, 4000, sorties[j][k])
                else:
                    ok = m.feed(None, sorties[j][k])
                assert ok, "frame %d of sortie %d rejected" % (k, j)

    # warm-up with every kernel timed: find the dominant kernel
    m.profile_enable(1)
    run(0, W)
    m.sync()
    prof = m.profile_read()
    names = list(prof.keys())
    dom = max(names, key=lambda n: prof[n]["ms"]) if W > 0 else "warp"
    m.profile_reset()
    # timed region: events around that kernel only, every n-th launch
    m.profile_enable((2 + names.index(dom)) | (args.event_every << 8) if args.event_every else 0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run(W, W + K)
    m.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    dt = t1 - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = m.stats()
    p = m.profile_read()[dom]
    m.profile_enable(0)

    if rank == 0:
        ach = p["alg_bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("f32" if force_float else "int16", {}).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "keyframes/sec fused (4000x3000 -> 256^2 tiles, 5-band)",
            "value": round(N * K / dt, 3), "unit": "keyframes/s", "n_gpus": N, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if force_float else "int16", "data": "synthetic",
            "config": {"workload": "cfg-2/cfg-A: 4000x3000 BGR8 keyframes, serpentine sortie, Map2D.Scale=%g, "
                                   "5-band Laplacian, %s pyramids, frames resident in HBM" %
                                   (args.scale, "CV_32FC3 (ForceFloat=1)" if force_float else "CV_16SC3"),
                       "frames_per_rank": K, "tile_sharding": "spatial hash, cell %d tiles" % block if N > 1 else "none",
                       "rendered_rank0": st["rendered"]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "avg_launch_us": round(p["ms"] / max(p["launches"], 1) * 1e3, 2),
                         "alg_bytes_per_launch": round(p["alg_bytes"] / max(p["launches"], 1)),
                         "launches": p["launches"], "timed_every": args.event_every},
            "frame_alg_GBps": round(sum(v["alg_bytes"] for v in prof.values()) / max(W, 1) * (N * K / dt) / N / 1e9, 1),
            "kernels_warmup_ms": {n: round(prof[n]["ms"], 3) for n in names if prof[n]["launches"]},
        }
        # measured HBM ceiling on this box: a 1 GiB device-to-device copy, read + write bytes
        try:
            a_ = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b_ = torch.empty_like(a_)
            b_.copy_(a_); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b_.copy_(a_)
            e1.record(); torch.cuda.synchronize()
            out["roofline"]["copy_ceiling_GBps"] = round(10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del a_, b_
        except Exception:
            pass
        if not args.no_cpu:
            hostf = [f.cpu().numpy() for f in frames[:2]]
            out["cpu_baseline"] = cpu_baseline(wl, cam, sorties[0], prep, hostf, force_float)
            allc = cpu_baseline_allcores(force_float)
            if allc:
                out["cpu_baseline_allcores"] = allc
            try:
                out["map2dcpu_single_band"] = map2dcpu_rates(pf, wl, cam, sorties[0], prep, frames, hostf)
            except Exception as e:                      # informational: never fail the headline line
                out["map2dcpu_single_band"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
