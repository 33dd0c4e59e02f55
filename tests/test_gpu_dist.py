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


def collective(world, rv, fn):
    """run fn(rank) on `world` threads (ranks of the in-process rendezvous); returns the per-rank results"""
    out = [None] * world

    def main(r):
        try:
            out[r] = fn(r)
        except Exception as e:                              # a dead rank would leave the others at the barrier
            out[r] = e
            rv.bar.abort()
    th = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(600) for t in th]
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
    return out


@pytest.mark.parametrize("force_float", [0, 1])
def test_dist_seven_bands(pf, orc, force_float):
    """BASELINE.json configs[4]'s band count on small frames: with 7 bands the level-0 strips are 128 pixels wide (half a
    tile) and the coarsest strips one pixel; three ranks, every exchanged byte hashed on both ends (pf_dist_set_verify),
    draw() and save() against the oracle (Ele::blend .cpp:93-126, save .cpp:806-836)."""
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    wl = workloads()
    cam, poses, frames = workload(wl)
    world = 3
    o = orc.OracleMap(band_num=7, force_float=force_float, scale=2.0)
    assert o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    maps = [pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, band_number=7, scale=2.0,
                            shard_rank=r, shard_count=world, shard_block=1) for r in range(world)]
    for m in maps:
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    for f, p in zip(frames, poses):
        assert o.feed(f, p)
        for m in maps:
            assert m.feed(f, p)
    rv = Rendezvous(world)
    dms = [sh.DistMap(m, r, world, backend="host", exchange=rv.fn(r)) for r, m in enumerate(maps)]
    for d in dms:
        d.set_verify(True)

    def rank_main(r):
        coords, px = dms[r].blend_changed()
        st = dms[r].stats()
        return {"coords": coords, "px": px, "stats": st, "save": dms[r].save_to_memory(), "save_stats": dms[r].stats()}
    out = collective(world, rv, rank_main)
    seen = {}
    for r in range(world):
        for t, im in zip(out[r]["coords"], out[r]["px"]):
            assert pf.tile_owner(maps[r].opt, *t) == r and t not in seen
            seen[t] = im
    assert set(seen) == set(o.tiles())
    for t, im in seen.items():
        assert np.array_equal(im, o.blend_tile(*t)), "tile %s differs from the oracle's blend" % (t,)
    assert sum(out[r]["stats"]["bytes_received"] for r in range(world)) > 0
    assert all(out[r]["stats"]["verified"] == 1 for r in range(world))          # the strip exchange was hashed on both ends
    assert all(out[r]["save_stats"]["verified"] == 1 for r in range(world))     # ... and the tile gather
    ref, org = o.save()
    got = out[0]["save"]
    assert got is not None and got[1] == org and np.array_equal(got[0], ref)
    for d in dms:
        d.close()


@pytest.mark.parametrize("world,force_float", [(4, 1), (8, 0)])
def test_dist_seam_exchange_at_cfg2_size(pf, orc, world, force_float):
    """BASELINE.json configs[2] at its frame size: 40 keyframes of the 4000x3000 sortie, tiles split over 4 / 8 ranks
    (threads, host-buffer transport, one GPU, hash cell 2 tiles).  draw() of every tile across ranks and the gathered save()
    must equal the unsharded HIP map's (itself oracle-checked in test_gpu_at_size.py / test_gpu_steady_state.py); eight
    probe tiles with neighbours on other ranks are compared with the oracle's Ele::blend directly."""
    torch = pytest.importorskip("torch")
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    wl = workloads()
    cam = [4000, 3000, 3000, 3000, 2000, 1500]
    n = 40
    poses = wl.serpentine(cam, 100.0, n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(99)
    frames = [torch.randint(0, 256, (3000, 4000, 3), dtype=torch.uint8, device="cuda", generator=gen) for _ in range(3)]
    torch.cuda.synchronize()

    def build(**kw):
        m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, **kw)
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
        for k, p in enumerate(poses):
            assert m.feed_device(frames[k % 3].data_ptr(), 3000, 4000, p)
        assert m.sync()
        return m
    ref = build()
    maps = [build(shard_rank=r, shard_count=world, shard_block=2) for r in range(world)]
    tiles = ref.tiles()
    assert sorted(sum((m.tiles() for m in maps), [])) == sorted(tiles) and len(tiles) > 500
    assert all(len(m.tiles()) > 0 for m in maps)
    rv = Rendezvous(world)
    dms = [sh.DistMap(m, r, world, backend="host", exchange=rv.fn(r)) for r, m in enumerate(maps)]

    def rank_main(r):
        coords, px = dms[r].blend_changed()
        st = dms[r].stats()
        return {"coords": coords, "px": px, "stats": st, "save": dms[r].save_to_memory()}
    out = collective(world, rv, rank_main)
    want_xy, want_px = ref.blend_changed(cap=len(tiles))
    want = {t: want_px[i] for i, t in enumerate(want_xy)}
    assert set(want) == set(tiles)
    n_seen, moved = 0, 0
    for r in range(world):
        moved += out[r]["stats"]["bytes_received"]
        for t, im in zip(out[r]["coords"], out[r]["px"]):
            assert pf.tile_owner(maps[r].opt, *t) == r
            assert np.array_equal(im, want[t]), "tile %s differs from the unsharded map's blend" % (t,)
            n_seen += 1
    assert n_seen == len(tiles) and moved > 100e6              # hundreds of MB of strips crossed ranks
    got = out[0]["save"]
    rs = ref.save_to_memory()
    assert got is not None and got[1] == rs[1] and np.array_equal(got[0], rs[0])
    del out, want_px, want
    # probe tiles: full 3x3 neighbourhood, at least one neighbour on another rank -> against the oracle itself
    owner = {t: pf.tile_owner(maps[0].opt, *t) for t in tiles}
    ts = set(tiles)
    probes = [t for t in tiles if all((t[0] + dx, t[1] + dy) in ts for dx in (-1, 0, 1) for dy in (-1, 0, 1))
              and any(owner[(t[0] + dx, t[1] + dy)] != owner[t] for dx in (-1, 0, 1) for dy in (-1, 0, 1))]
    probes = probes[::max(1, len(probes) // 8)][:8]
    assert len(probes) == 8
    o = orc.OracleMap(force_float=force_float)
    assert o.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
    host = [f.cpu().numpy() for f in frames]
    for k, p in enumerate(poses):
        assert o.feed(host[k % 3], p)
    assert o.tiles() == tiles or sorted(o.tiles()) == sorted(tiles)
    for t in probes:
        assert np.array_equal(ref.blend_tile(*t), o.blend_tile(*t)), "probe tile %s: HIP blend differs from the oracle's" % (t,)
    for d in dms:
        d.close()


@pytest.mark.parametrize("world,force_float,root", [(3, 0, 0), (4, 1, 2)])
def test_dist_feed_from_one_rank(pf, orc, world, force_float, root):
    """pf_dist_feed: the keyframes live on the host of ONE rank; every rank gets the pose, only the ranks that own a tile
    of a frame's canvas get its pixels (one grouped exchange from the root's GPU, hashed on both ends here), and the
    shards' tiles -- union and pixels -- are the oracle's.  Reference: Map2D::feed, MultiBandMap2DCPU.cpp:288-309."""
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    wl = workloads()
    cam, poses, frames = workload(wl)
    o = orc.OracleMap(force_float=force_float, scale=2.0)
    assert o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    maps = [pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=force_float, scale=2.0,
                            shard_rank=r, shard_count=world, shard_block=2) for r in range(world)]
    for m in maps:
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    rv = Rendezvous(world)
    dms = [sh.DistMap(m, r, world, backend="host", exchange=rv.fn(r)) for r, m in enumerate(maps)]
    for d in dms:
        d.set_verify(True)
    s80 = np.sin(np.radians(80) / 2), np.cos(np.radians(80) / 2)
    oblique = [0, 0, -100, s80[0], 0, 0, s80[1]]

    def rank_main(r):
        moved = 0
        for f, p in zip(frames, poses):
            assert dms[r].feed(f if r == root else None, p, root=root, shape=f.shape) is True
            st = dms[r].stats()
            moved += st["bytes_received"] + st["bytes_sent"]
        assert dms[r].feed(frames[0] if r == root else None, oblique, root=root, shape=frames[0].shape) is False   # rejected alike
        # a frame that does not match the camera: rejected as Map2D::feed rejects it (.cpp:319-323) -- False on every rank, counted,
        # the grid untouched -- not an error of the exchange
        grid0, rej0 = maps[r].grid(), maps[r].stats()["rejected"]
        small = frames[0][:100, :200]
        assert dms[r].feed(np.ascontiguousarray(small) if r == root else None, poses[0], root=root, shape=small.shape) is False
        assert maps[r].grid() == grid0 and maps[r].stats()["rejected"] == rej0 + 1
        assert maps[r].sync()
        return moved
    for f, p in zip(frames, poses):
        assert o.feed(f, p)
    out = collective(world, rv, rank_main)
    from helpers import map_digest
    want, got = map_digest(o), {}
    for r, m in enumerate(maps):
        for t in m.tiles():
            assert pf.tile_owner(m.opt, *t) == r
        d = map_digest(m)
        assert not (set(d) & set(got))
        got.update(d)
        assert m.grid() == o.grid()
    assert got == want
    nbytes = frames[0].size
    assert out[root] > 0 and out[root] % nbytes == 0                         # whole frames left the root ...
    assert sum(out) == 2 * out[root]                                          # ... and every byte sent was received once
    assert out[root] <= (world - 1) * len(frames) * nbytes                    # at most one copy per other rank and frame (ranks without a tile of the canvas get none)
    for d in dms:
        d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,root", [(3, 1), (4, 0)])
def test_dist_feed_jpeg_from_one_rank(pf, orc, world, root):
    """pf_dist_feed_jpeg: the keyframes are .jpg streams on the host of ONE rank, which decodes them on its GPU straight into the slot the
    exchange sends from; shards' tiles = the oracle fed libjpeg-turbo's pixels (backup/map2dfusion.cpp:129-135 across ranks)"""
    import io
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 24)          # Pillow's progressive pass wants the whole stream in one buffer
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    wl = workloads()
    cam, poses, frames = workload(wl)
    streams, decoded = [], []
    for k, f in enumerate(frames):
        b = io.BytesIO(); Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])).save(b, "JPEG", quality=88, subsampling=[2, 0, 1][k % 3], progressive=(k % 4 == 3))
        streams.append(b.getvalue())
        decoded.append(np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))[:, :, ::-1]))
    o = orc.OracleMap(force_float=0, scale=2.0)
    assert o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    for f, p in zip(decoded, poses):
        assert o.feed(f, p)
    maps = [pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=0, scale=2.0, shard_rank=r, shard_count=world, shard_block=2) for r in range(world)]
    for m in maps:
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    rv = Rendezvous(world)
    dms = [sh.DistMap(m, r, world, backend="host", exchange=rv.fn(r)) for r, m in enumerate(maps)]
    for d in dms:
        d.set_verify(True)
    shape = frames[0].shape

    def rank_main(r):
        for s, p in zip(streams, poses):
            assert dms[r].feed_jpeg(s if r == root else None, p, shape, root=root) is True
        # a broken stream on the root fails on every rank together, and the ranks carry on
        try:
            dms[r].feed_jpeg(b"\xff\xd8junk" if r == root else None, poses[0], shape, root=root)
            raised = False
        except RuntimeError:
            raised = True
        assert raised
        assert dms[r].feed_jpeg(streams[0] if r == root else None, poses[0], shape, root=root) is True
        assert maps[r].sync()
        return 0
    assert o.feed(decoded[0], poses[0])
    collective(world, rv, rank_main)
    from helpers import map_digest
    want, got = map_digest(o), {}
    for r, m in enumerate(maps):
        d = map_digest(m)
        assert not (set(d) & set(got))
        got.update(d)
    assert got == want
    if root < world:
        assert pf.jpeg_huffman_counts(maps[root])[0] >= len(streams) - 3          # the sequential streams: Huffman pass on the root's GPU
    for d in dms:
        d.close()
    for m in maps:
        m.close()


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
