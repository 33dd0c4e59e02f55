"""N>1 path on CPU: world_size-2 gloo run of the seam exchange.

What runs here is the product's: the exchange PLAN is csrc/dist.cpp's plan_blend (through the C ABI,
pf_dist_plan_blend -- the function DistMap::blend_changed calls), the bytes travel through the product's
host-buffer hook (sharding.torch_exchange, the pf_exchange_fn the library's HostTransport calls).  Only the pixel
work (strip pack, 3x3 assembly + collapse) is done by an oracle-backed stand-in, because there is no GPU here; on
the GPU box tests/test_gpu_dist.py runs the same exchange end to end inside the library.
Checked against the unsharded oracle: every tile's blend() with strips that crossed ranks, and the tile gather
that precedes save()."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ELE = 256


class OracleEngine:
    """Strip pack / blend / tile export on top of an OracleMap restricted to owned tiles (what FusionMap::pack_strips,
    blend_tiles and export_tiles do on the GPU)."""

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


def hook_exchange(sh, send_bufs, recv_sizes, world, rank):
    """Move per-peer byte strings through the product's pf_exchange_fn (ctypes pointers and sizes, as HostTransport calls it)."""
    import ctypes as C
    fn = sh.torch_exchange()
    keep_s = [(C.c_char * max(len(b), 1)).from_buffer_copy(b if len(b) else b"\0") for b in send_bufs]
    keep_r = [(C.c_char * max(n, 1))() for n in recv_sizes]
    sp = (C.c_void_p * world)(*[C.addressof(k) for k in keep_s])
    rp = (C.c_void_p * world)(*[C.addressof(k) for k in keep_r])
    sb = (C.c_size_t * world)(*[0 if p == rank else len(send_bufs[p]) for p in range(world)])
    rb = (C.c_size_t * world)(*[0 if p == rank else recv_sizes[p] for p in range(world)])
    assert fn(None, sp, sb, rp, rb, world) == 1
    return [bytes(keep_r[p][:recv_sizes[p]]) for p in range(world)]


def worker(rank, world, port):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch
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
    # every rank's tile list on every rank, in the order FusionMap::list_tiles gives (iy, then ix), every tile changed
    lists = [None] * world
    dist.all_gather_object(lists, [(ix, iy, 1) for (ix, iy) in sorted(eng.owned, key=lambda t: (t[1], t[0]))])
    assert sorted((t[0], t[1]) for l in lists for t in l) == sorted(all_tiles)
    hb9 = [0 if j == 4 else eng.halo_bytes(j % 3 - 1, j // 3 - 1) for j in range(9)]
    caps = [len(l) for l in lists]
    send, recv, mine = sh.plan_blend(lists, caps, rank, True, hb9)        # the library's plan
    assert mine == [(t[0], t[1]) for t in lists[rank]]
    # plan symmetry: what I send to p, strip for strip and byte for byte, is what p expects from me
    for p in range(world):
        s_p, r_p, _ = sh.plan_blend(lists, caps, p, True, hb9)
        mine_to_p = [(q["ix"], q["iy"], q["dx"], q["dy"], q["offset"]) for q in send if q["peer"] == p]
        p_from_me = [(q["ix"] + q["dx"], q["iy"] + q["dy"], q["dx"], q["dy"], q["offset"]) for q in r_p if q["peer"] == rank]
        assert mine_to_p == p_from_me
    # pack what the plan asks of this rank, move it through the product's exchange hook
    send_bufs = [bytearray() for _ in range(world)]
    for q in send:
        assert len(send_bufs[q["peer"]]) == q["offset"]
        buf = torch.empty(hb9[3 * (q["dy"] + 1) + (q["dx"] + 1)], dtype=torch.uint8)
        eng.pack_halo(q["ix"], q["iy"], q["dx"], q["dy"], buf)
        send_bufs[q["peer"]] += buf.numpy().tobytes()
    recv_sizes = [0] * world
    for q in recv:
        recv_sizes[q["peer"]] = max(recv_sizes[q["peer"]], q["offset"] + hb9[3 * (q["dy"] + 1) + (q["dx"] + 1)])
    got = hook_exchange(sh, [bytes(b) for b in send_bufs], recv_sizes, world, rank)
    halos = {}
    for q in recv:
        j = 3 * (q["dy"] + 1) + (q["dx"] + 1)
        halos.setdefault((q["ix"], q["iy"]), [None] * 9)[j] = torch.frombuffer(bytearray(got[q["peer"]][q["offset"]:q["offset"] + hb9[j]]), dtype=torch.uint8)
    # blend with remote strips == unsharded oracle blend
    n_remote = 0
    for (ix, iy) in mine:
        full = all((ix + dx, iy + dy) in all_tiles for dx in (-1, 0, 1) for dy in (-1, 0, 1))
        if not full:
            assert (ix, iy) not in halos                      # blends alone (.cpp:134-145): the plan moves nothing for it
            continue
        img = eng.blend_with_halo(ix, iy, halos.get((ix, iy), [None] * 9), raw=True)
        n_remote += any((ix + dx, iy + dy) not in eng.owned for dx in (-1, 0, 1) for dy in (-1, 0, 1))
        assert np.array_equal(img, o.blend_tile_raw(ix, iy)), (rank, ix, iy)
    cnt = [None] * world
    dist.all_gather_object(cnt, n_remote)
    assert sum(cnt) > 0, "no blend needed a remote strip: the test exercises nothing"
    # save()'s gather (DistMap::save_to_memory): every rank's tiles, in list order, once to rank 0
    nb = eng.tile_bytes()
    blob = bytearray()
    if rank != 0:
        for (ix, iy, _) in lists[rank]:
            buf = torch.empty(nb, dtype=torch.uint8)
            eng.export_tile(ix, iy, buf)
            blob += buf.numpy().tobytes()
    sizes = [0 if p == 0 or rank != 0 else nb * len(lists[p]) for p in range(world)]
    got = hook_exchange(sh, [bytes(blob) if p == 0 and rank != 0 else b"" for p in range(world)], sizes, world, rank)
    if rank == 0:
        n = 0
        for p in range(1, world):
            for k, (ix, iy, _) in enumerate(lists[p]):
                ref = b"".join(a.tobytes() for i in range(o.num_levels) for a in o.tile_level(ix, iy, i))
                assert got[p][k * nb:(k + 1) * nb] == ref
                n += 1
        assert n == len(all_tiles) - len(eng.owned)
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
    """The library's plan on a 7x5 tile field split three ways: every strip crosses ranks, every cross-rank neighbour of
    a tile with a full 3x3 neighbourhood is served exactly once, both sides agree on order and offsets, caps are honoured."""
    import importlib
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    opt = pf.default_options(shard_count=3, shard_block=2)
    tiles = sorted([(x, y) for x in range(-3, 4) for y in range(-2, 3)], key=lambda t: (t[1], t[0]))
    lists = [[(t[0], t[1], 1) for t in tiles if pf.tile_owner(opt, *t) == r] for r in range(3)]
    hb9 = [0 if j == 4 else 100 + j for j in range(9)]
    plans = [sh.plan_blend(lists, [len(l) for l in lists], r, True, hb9) for r in range(3)]
    total = 0
    for r, (send, recv, mine) in enumerate(plans):
        assert all(q["peer"] != r for q in send + recv)
        for q in recv:
            assert (q["ix"], q["iy"], 1) in lists[r] and (q["ix"] + q["dx"], q["iy"] + q["dy"], 1) in lists[q["peer"]]
        for p in range(3):
            a = [(q["ix"], q["iy"], q["dx"], q["dy"], q["offset"]) for q in send if q["peer"] == p]
            b = [(q["ix"] + q["dx"], q["iy"] + q["dy"], q["dx"], q["dy"], q["offset"]) for q in plans[p][1] if q["peer"] == r]
            assert a == b
        total += len(recv)
    full = lambda x, y: all((x + dx, y + dy) in tiles for dx in (-1, 0, 1) for dy in (-1, 0, 1))
    want = sum(1 for (x, y) in tiles if full(x, y) for dx in (-1, 0, 1) for dy in (-1, 0, 1)
               if pf.tile_owner(opt, x, y) != pf.tile_owner(opt, x + dx, y + dy))
    assert total == want > 0
    # low-quality show: every tile blends alone, nothing moves; a cap cuts the requester's list on every rank alike
    assert all(sh.plan_blend(lists, [len(l) for l in lists], r, False, hb9)[:2] == ([], []) for r in range(3))
    capped = [sh.plan_blend(lists, [2, 2, 2], r, True, hb9) for r in range(3)]
    assert all(len(c[2]) == min(2, len(lists[r])) for r, c in enumerate(capped))
    assert sum(len(c[0]) for c in capped) == sum(len(c[1]) for c in capped)


if __name__ == "__main__":
    worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
