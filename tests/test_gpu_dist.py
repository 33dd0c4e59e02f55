"""The seam exchange inside the library (csrc/dist.cpp, pf_dist_*) on the GPU: draw() across ranks (changed tiles blended
with neighbour strips from other ranks: one pack launch, one exchange, one batched blend) and save() across ranks (tile
gather to rank 0 + whole-mosaic collapse), against the unsharded map AND the oracle.
  * ranks as threads of one process, host-buffer transport through an in-process rendezvous (2 and 3 ranks, one GPU);
  * ranks as processes under torch.distributed.run: gloo + host-buffer transport on one GPU, RCCL on two GPUs
    (skipped on a one-GPU box)."""
import importlib
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from helpers import jitter_poses, workloads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Rendezvous:
    """pf_exchange_fn for ranks that are threads: a mailbox and two barriers per exchange"""

    def __init__(self, n):
        self.n, self.box, self.bar = n, {}, threading.Barrier(n)

    def fn(self, me):
        import ctypes as C

        def exchange(user, send, send_bytes, recv, recv_bytes, n):
            try:
                for p in range(n):
                    if p != me and send_bytes[p]:
                        self.box[(me, p)] = C.string_at(send[p], send_bytes[p])
                self.bar.wait(60)
                for p in range(n):
                    if p != me and recv_bytes[p]:
                        b = self.box[(p, me)]
                        assert len(b) == recv_bytes[p], "rank %d expected %d bytes from %d, got %d" % (me, recv_bytes[p], p, len(b))
                        C.memmove(recv[p], b, len(b))
                self.bar.wait(60)
                if me == 0:
                    self.box.clear()
                self.bar.wait(60)
                return 1
            except Exception as e:
                print("rendezvous exchange failed on rank", me, e)
                return 0
        return exchange


def workload(wl, n=9):
    cam = [640, 480, 500, 500, 320, 240]
    base = jitter_poses(n, seed=17, step=(0.0, 0.0))
    poses = [[(k % 3) * 70.0 + p[0], (k // 3) * 55.0 + p[1]] + p[2:] for k, p in enumerate(base)]
    frames = [wl.smooth_frame(480, 640, k) ^ wl.noise_frame(480, 640, k) for k in range(len(poses))]
    return cam, poses, frames


@pytest.mark.parametrize("world,force_float,hq", [(2, 0, 1), (3, 1, 1), (2, 1, 0)])
def test_dist_draw_and_save_ranks_as_threads(pf, orc, world, force_float, hq):
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    wl = workloads()
    cam, poses, frames = workload(wl)
    o = orc.OracleMap(force_float=force_float, scale=2.0, high_quality=hq)
    assert o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    maps = [pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, scale=2.0, high_quality_show=hq,
                            shard_rank=r, shard_count=world, shard_block=1) for r in range(world)]
    for m in maps:
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    half = len(poses) // 2
    rv = Rendezvous(world)
    dms = [sh.DistMap(m, r, world, backend="host", exchange=rv.fn(r)) for r, m in enumerate(maps)]
    out = [None] * world

    def rank_main(r, lo, hi, do_save):
        try:
            for f, p in zip(frames[lo:hi], poses[lo:hi]):
                assert maps[r].feed(f, p)
            coords, px = dms[r].blend_changed()
            out[r] = {"coords": coords, "px": px, "stats": dms[r].stats(), "save": dms[r].save_to_memory() if do_save else None}
        except Exception as e:                              # a dead rank would leave the others at the barrier
            out[r] = e
            rv.bar.abort()

    def run_all(lo, hi, do_save):
        th = [threading.Thread(target=rank_main, args=(r, lo, hi, do_save)) for r in range(world)]
        [t.start() for t in th]; [t.join(180) for t in th]
        for r in range(world):
            assert not isinstance(out[r], Exception), out[r]

    # first half of the sortie, draw; second half, draw again (only the tiles the new frames changed), then save
    for lo, hi, last in ((0, half, False), (half, len(poses), True)):
        for f, p in zip(frames[lo:hi], poses[lo:hi]):
            assert o.feed(f, p)
        run_all(lo, hi, last)
        seen = {}
        for r in range(world):
            for t, im in zip(out[r]["coords"], out[r]["px"]):
                assert pf.tile_owner(maps[r].opt, *t) == r and t not in seen
                seen[t] = im
        if lo == 0:
            assert set(seen) == set(o.tiles())
        for t, im in seen.items():
            assert np.array_equal(im, o.blend_tile(*t)), "tile %s differs from the oracle's blend" % (t,)
        moved = sum(out[r]["stats"]["bytes_received"] for r in range(world))
        assert (moved > 0) == bool(hq)                      # low-quality show blends alone: no strips travel
        assert all(out[r]["stats"]["strips_received"] * 0 == 0 for r in range(world))
    ref, org = o.save()
    got = out[0]["save"]
    assert got is not None and got[1] == org and np.array_equal(got[0], ref)
    assert all(out[r]["save"] is None for r in range(1, world))
    assert sorted(sum((m.tiles() for m in maps), [])) == sorted(o.tiles())      # save left the stores alone
    for d in dms:
        d.close()


def test_dist_caps_and_empty_ranks(pf, orc):
    """a rank may take fewer tiles per call than it has changed (cap), and a rank may hold no tile at all: the providers
    plan with the requester's cap, and the calls repeat until nothing is left"""
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    wl = workloads()
    cam, poses, frames = workload(wl, 4)
    world = 3
    o = orc.OracleMap(scale=2.0)
    assert o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    # cell edge 64 tiles: the whole mosaic belongs to one rank, the other two hold nothing
    maps = [pf.Map2D.create(pf.TypeMultiBandCPU, False, scale=2.0, shard_rank=r, shard_count=world, shard_block=64) for r in range(world)]
    for m in maps:
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    for f, p in zip(frames, poses):
        assert o.feed(f, p)
        for m in maps:
            assert m.feed(f, p)
    assert sorted(len(m.tiles()) for m in maps)[:2] == [0, 0]
    rv = Rendezvous(world)
    dms = [sh.DistMap(m, r, world, backend="host", exchange=rv.fn(r)) for r, m in enumerate(maps)]
    got = [dict() for _ in range(world)]
    err = []
    calls = (len(o.tiles()) + 4) // 5 + 1                       # the last call finds nothing left

    def rank_main(r):
        try:
            for _ in range(calls):                              # every rank makes the same number of collective calls
                coords, px = dms[r].blend_changed(cap=5)
                assert len(coords) <= 5
                for t, im in zip(coords, px):
                    assert t not in got[r]
                    got[r][t] = im.copy()
        except Exception as e:
            err.append(e); rv.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(180) for t in th]
    assert not err, err
    seen = {}
    for g in got:
        seen.update(g)
    assert set(seen) == set(o.tiles()) and len(seen) > 10
    for t, im in seen.items():
        assert np.array_equal(im, o.blend_tile(*t)), t
    for d in dms:
        d.close()


def test_rccl_loads_and_runs_with_one_rank(pf, orc):
    """RCCL inside the library on the one GPU at hand: librccl is found, a one-rank communicator comes up, draw() and save()
    go through pf_dist_* (no peer to exchange with) and equal the plain calls"""
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    wl = workloads()
    cam, poses, frames = workload(wl, 4)
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False, scale=2.0)
    o = orc.OracleMap(scale=2.0)
    assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:2]) and o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    for f, p in zip(frames, poses):
        assert m.feed(f, p) and o.feed(f, p)
    d = sh.DistMap(m, 0, 1, backend="nccl")
    coords, px = d.blend_changed()
    assert sorted(coords) == sorted(o.tiles())
    for t, im in zip(coords, px):
        assert np.array_equal(im, o.blend_tile(*t))
    img, org = d.save_to_memory()
    ref, oorg = o.save()
    assert org == oorg and np.array_equal(img, ref)
    st = d.stats()
    assert st["bytes_sent"] == 0 and st["bytes_received"] == 0
    d.close()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def run_workers(world, backend, force_float):
    env = dict(os.environ, PF_DIST_BACKEND=backend, PF_TEST_FLOAT=str(force_float), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "DIST OK" in out, out[-3000:]
    return out


def test_dist_ranks_as_processes_gloo_one_gpu():
    """two processes sharing the GPU: the library's host-buffer transport over torch point-to-point (gloo)"""
    run_workers(2, "gloo", 0)


def test_dist_rccl_two_gpus():
    """RCCL inside the library, one GPU per rank (first run on a multi-GPU box)"""
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    run_workers(2, "nccl", 1)
