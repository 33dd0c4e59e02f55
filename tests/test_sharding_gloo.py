"""N>1 path on CPU: world_size-2 gloo run of the seam exchange (pi-slam-fusion_amd/sharding.py).

The transport and the exchange plan are the product's; the engine behind them is an
oracle-backed stand-in (no GPU here): each rank holds the tiles the spatial hash gives it.
Checked against the unsharded oracle: every tile's blend() with strips that crossed ranks,
and the tile gather that precedes save()."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ELE = 256


class OracleEngine:
    """Exchange interface of sharding.GpuEngine on top of an OracleMap restricted to owned tiles."""

    def __init__(self, omap, pf, opt, rank, orc):
        import torch
        self.torch, self.o, self.orc = torch, omap, orc
        self.owned = [t for t in omap.tiles() if pf.tile_owner(opt, t[0], t[1]) == rank]
        self.nl = omap.num_levels
        self.dt = omap.dtype
        self.es = np.dtype(self.dt).itemsize
        self.imported = {}

    def tiles(self):
        return list(self.owned)

    def _dims(self, level, dx, dy):
        ts, b = ELE >> level, 1 << (self.nl - 1 - level)
        return (ts if dx == 0 else b), (ts if dy == 0 else b)

    def halo_bytes(self, dx, dy):
        return sum(w * h for w, h in (self._dims(i, dx, dy) for i in range(self.nl))) * 3 * self.es

    def tile_bytes(self):
        return sum((ELE >> i) ** 2 for i in range(self.nl)) * (3 * self.es + 4)

    def empty(self, n):
        return self.torch.empty(max(n, 1), dtype=self.torch.uint8)

    def pack_halo(self, ix, iy, dx, dy, out):
        parts = []
        for i in range(self.nl):
            lap, _ = self.o.tile_level(ix, iy, i)
            ts = ELE >> i
            w, h = self._dims(i, dx, dy)
            x0 = ts - w if dx < 0 else 0
            y0 = ts - h if dy < 0 else 0
            parts.append(np.ascontiguousarray(lap[y0:y0 + h, x0:x0 + w]).tobytes())
        out.copy_(self.torch.frombuffer(bytearray(b"".join(parts)), dtype=self.torch.uint8))

    def blend_with_halo(self, ix, iy, halos, raw=True):
        """Ele::blend's 3x3 assembly (.cpp:93-117) from local tiles / received strips, collapsed by the oracle."""
        src = {}
        for j in range(9):
            dx, dy = j % 3 - 1, j // 3 - 1
            if (ix + dx, iy + dy) in self.owned:
                src[j] = ("tile", None)
            elif halos[j] is not None:
                src[j] = ("strip", halos[j].numpy().tobytes())
            else:
                return self.o.blend_tile_raw(ix, iy) if (dx, dy) != (0, 0) and not self._has_all(ix, iy) else None
        levels = []
        for i in range(self.nl):
            ts, b = ELE >> i, 1 << (self.nl - 1 - i)
            side = ts + 2 * b
            img = np.zeros((side, side, 3), self.dt)
            for j in range(9):
                dx, dy = j % 3 - 1, j // 3 - 1
                w, h = self._dims(i, dx, dy)
                X = 0 if dx < 0 else (b if dx == 0 else side - b)
                Y = 0 if dy < 0 else (b if dy == 0 else side - b)
                if src[j][0] == "tile":
                    lap, _ = self.o.tile_level(ix + dx, iy + dy, i)
                    x0 = ts - w if dx < 0 else 0
                    y0 = ts - h if dy < 0 else 0
                    blk = lap[y0:y0 + h, x0:x0 + w]
                else:
                    off = sum(ww * hh for ww, hh in (self._dims(k, dx, dy) for k in range(i))) * 3 * self.es
                    blk = np.frombuffer(src[j][1], self.dt, count=w * h * 3, offset=off).reshape(h, w, 3)
                img[Y:Y + h, X:X + w] = blk
            levels.append(img)
        full = self.orc.restore_from_laplace_pyr(levels)
        b0 = 1 << (self.nl - 1)
        out = full[b0:b0 + ELE, b0:b0 + ELE].copy()
        out[self.o.tile_level(ix, iy, 0)[1] == 0] = 0
        return out

    def _has_all(self, ix, iy):
        return False

    def export_tile(self, ix, iy, out):
        parts = []
        for i in range(self.nl):
            lap, w = self.o.tile_level(ix, iy, i)
            parts += [lap.tobytes(), w.tobytes()]
        out.copy_(self.torch.frombuffer(bytearray(b"".join(parts)), dtype=self.torch.uint8))

    def import_tile(self, ix, iy, buf):
        self.imported[(ix, iy)] = bytes(buf.numpy().tobytes())


def worker(rank, world, port):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch.distributed as dist
    from conftest import load_package
    from helpers import jitter_poses
    from oracle import orc
    pf = load_package()
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    wl = importlib.import_module("pi_slam_fusion_amd.workloads")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    cam = [320, 240, 250, 250, 160, 120]
    base = jitter_poses(9, seed=13, step=(0.0, 0.0), height=100.0)
    poses = [[(k % 3) * 45.0 + p[0], (k // 3) * 35.0 + p[1]] + p[2:] for k, p in enumerate(base)]   # 3x3 grid of views
    o = orc.OracleMap(band_num=3, scale=2.0)
    assert o.prepare(wl.IDENTITY_PLANE, cam, poses)
    for k, p in enumerate(poses):
        assert o.feed(wl.noise_frame(240, 320, k), p)
    opt = pf.default_options(shard_count=world, shard_block=1, shard_rank=rank)
    eng = OracleEngine(o, pf, opt, rank, orc)
    all_tiles = o.tiles()
    assert 0 < len(eng.owned) < len(all_tiles)
    # plan symmetry: what I send to p is what p expects from me
    lists = sh.all_tile_lists(eng)
    assert sorted(t for l in lists for t in map(tuple, l)) == sorted(all_tiles)
    send, recv = sh.plan_halo_exchange(lists, rank)
    for p in range(world):
        s_p, r_p = sh.plan_halo_exchange(lists, p)
        assert send[p] == r_p[rank] and recv[p] == s_p[rank]
    # blend with remote strips == unsharded oracle blend
    out = sh.blend_all(eng, raw=True)
    n_remote = 0
    for (ix, iy), img in out.items():
        full = all((ix + dx, iy + dy) in all_tiles for dx in (-1, 0, 1) for dy in (-1, 0, 1))
        if full:
            n_remote += any((ix + dx, iy + dy) not in eng.owned for dx in (-1, 0, 1) for dy in (-1, 0, 1))
            assert np.array_equal(img, o.blend_tile_raw(ix, iy)), (rank, ix, iy)
    cnt = [None] * world
    dist.all_gather_object(cnt, n_remote)
    assert sum(cnt) > 0, "no blend needed a remote strip: the test exercises nothing"
    # save()'s gather
    got = sh.gather_tiles(eng, root=0)
    if rank == 0:
        assert got == len(all_tiles) - len(eng.owned)
        for (ix, iy), blob in eng.imported.items():
            ref = b"".join(a.tobytes() for i in range(o.num_levels) for a in o.tile_level(ix, iy, i))
            assert blob == ref
    dist.barrier()
    dist.destroy_process_group()


def test_seam_exchange_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), "2", str(port)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


def test_plan_is_deterministic_and_complete(pf):
    import importlib
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    opt = pf.default_options(shard_count=3, shard_block=2)
    tiles = [(x, y) for x in range(-3, 4) for y in range(-2, 3)]
    lists = [[t for t in tiles if pf.tile_owner(opt, *t) == r] for r in range(3)]
    total = 0
    for r in range(3):
        send, recv = sh.plan_halo_exchange(lists, r)
        assert send[r] == [] and recv[r] == []
        for p in range(3):
            for (ix, iy, dx, dy) in recv[p]:
                assert (ix, iy) in lists[r] and (ix + dx, iy + dy) in lists[p]
            total += len(recv[p])
    want = sum(1 for (x, y) in tiles for dx in (-1, 0, 1) for dy in (-1, 0, 1)
               if (x + dx, y + dy) in tiles and pf.tile_owner(opt, x, y) != pf.tile_owner(opt, x + dx, y + dy))
    assert total == want


if __name__ == "__main__":
    worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
