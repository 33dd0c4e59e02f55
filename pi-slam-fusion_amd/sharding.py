"""Multi-GPU seam exchange for the tile-sharded mosaic.

feed() needs no collective: every rank is fed every keyframe's pose, renders the part of
the frame that lands on tiles it owns (owner = spatial hash of the tile, pf_tile_owner) and
recomputes the pyramid halo from the source frame.  The only exchange steps are

  * Ele::blend (MultiBandMap2DCPU.cpp:77-146): a tile's 3x3 neighbourhood may live on other
    ranks; their edge strips (border 1<<(L-i) pixels at level i) are packed on the owner and
    moved with ONE all_to_all (point-to-point traffic over xGMI, all links at once);
  * save (.cpp:779-847): whole tiles are gathered to rank 0, which runs the mosaic collapse.

The transport is torch.distributed (backend nccl == RCCL on ROCm, gloo on CPU); the engine
behind it is anything with the small interface of `GpuEngine` below -- the HIP library on a
GPU box, the oracle-backed stand-in of tests/test_sharding_gloo.py on CPU.
"""
import numpy as np
import torch
import torch.distributed as dist

NEIGHBOURS = [(dx, dy) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]      # index j = 3*(dy+1)+(dx+1)


class GpuEngine:
    """Adapter of a pi_slam_fusion_amd.Map2D to the exchange interface (device tensors)."""

    def __init__(self, m, device):
        self.m, self.device = m, device

    def tiles(self):
        return self.m.tiles()

    def halo_bytes(self, dx, dy):
        return self.m.halo_bytes(dx, dy)

    def tile_bytes(self):
        return self.m.tile_bytes()

    def empty(self, nbytes):
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)

    def pack_halo(self, ix, iy, dx, dy, out):
        assert self.m.halo_pack(ix, iy, dx, dy, out.data_ptr())

    def blend_with_halo(self, ix, iy, halos, raw=False):
        return self.m.blend_tile_halo(ix, iy, [h.data_ptr() if h is not None else 0 for h in halos], raw=raw)

    def export_tile(self, ix, iy, out):
        assert self.m.tile_export(ix, iy, out.data_ptr())

    def import_tile(self, ix, iy, buf):
        assert self.m.tile_import(ix, iy, buf.data_ptr())


def all_tile_lists(engine, group=None):
    """Every rank's tile list on every rank (tiny all_gather of coordinates)."""
    world = dist.get_world_size(group)
    mine = engine.tiles()
    out = [None] * world
    dist.all_gather_object(out, mine, group=group)
    return out


def plan_halo_exchange(tile_lists, rank):
    """Which strips this rank must send / will receive so that every rank can blend its tiles.

    A strip is identified by (requesting tile, dx, dy): the tile at (ix+dx, iy+dy) hands over
    the edge that blend() copies from it (.cpp:101-116).  Returns (send, recv): per peer rank an
    ordered list of (ix, iy, dx, dy); both sides derive the same order, so no header travels."""
    world = len(tile_lists)
    owner = {}
    for r, tl in enumerate(tile_lists):
        for t in tl:
            owner[tuple(t)] = r
    send = [[] for _ in range(world)]
    recv = [[] for _ in range(world)]
    for r, tl in enumerate(tile_lists):
        for (ix, iy) in sorted(tuple(t) for t in tl):
            for (dx, dy) in NEIGHBOURS:
                if dx == 0 and dy == 0:
                    continue
                o = owner.get((ix + dx, iy + dy))
                if o is None or o == r:
                    continue
                if r == rank:
                    recv[o].append((ix, iy, dx, dy))
                if o == rank:
                    send[r].append((ix, iy, dx, dy))
    return send, recv


def exchange_halos(engine, group=None):
    """All ranks: pack the strips the others need, one all_to_all, return
    {(ix,iy): [9 tensors or None]} for this rank's tiles."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lists = all_tile_lists(engine, group)
    send, recv = plan_halo_exchange(lists, rank)
    size = {(dx, dy): engine.halo_bytes(dx, dy) for (dx, dy) in NEIGHBOURS if (dx, dy) != (0, 0)}
    send_sizes = [sum(size[(dx, dy)] for (_, _, dx, dy) in send[p]) for p in range(world)]
    recv_sizes = [sum(size[(dx, dy)] for (_, _, dx, dy) in recv[p]) for p in range(world)]
    sbuf, rbuf = engine.empty(sum(send_sizes)), engine.empty(sum(recv_sizes))
    off = 0
    for p in range(world):
        for (ix, iy, dx, dy) in send[p]:
            n = size[(dx, dy)]
            engine.pack_halo(ix + dx, iy + dy, dx, dy, sbuf[off:off + n])
            off += n
    if world > 1:
        dist.all_to_all_single(rbuf[:sum(recv_sizes)], sbuf[:sum(send_sizes)], recv_sizes, send_sizes, group=group)
    halos = {}
    off = 0
    for p in range(world):
        for (ix, iy, dx, dy) in recv[p]:
            n = size[(dx, dy)]
            halos.setdefault((ix, iy), [None] * 9)[3 * (dy + 1) + (dx + 1)] = rbuf[off:off + n]
            off += n
    return halos, rbuf


def blend_all(engine, group=None, raw=False):
    """Distributed draw()-time refresh: every rank blends the tiles it owns, with the strips of
    neighbours that live elsewhere.  Returns {(ix,iy): 256x256x3 array}."""
    halos, keep = exchange_halos(engine, group)
    out = {}
    for (ix, iy) in engine.tiles():
        out[(ix, iy)] = engine.blend_with_halo(ix, iy, halos.get((ix, iy), [None] * 9), raw=raw)
    del keep
    return out


def gather_tiles(engine, dst_engine=None, root=0, group=None):
    """save()'s gather: every tile travels once to `root`, which imports it into dst_engine
    (its own map) -- afterwards root.save() is the reference's whole-mosaic collapse."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lists = all_tile_lists(engine, group)
    nb = engine.tile_bytes()
    if rank != root:
        mine = sorted(tuple(t) for t in lists[rank])
        buf = engine.empty(nb * len(mine))
        for k, (ix, iy) in enumerate(mine):
            engine.export_tile(ix, iy, buf[k * nb:(k + 1) * nb])
        if len(mine):
            dist.send(buf[:nb * len(mine)], dst=root, group=group)
        return 0
    n = 0
    for p in range(world):
        if p == root:
            continue
        theirs = sorted(tuple(t) for t in lists[p])
        if not theirs:
            continue
        buf = engine.empty(nb * len(theirs))
        dist.recv(buf[:nb * len(theirs)], src=p, group=group)
        for k, (ix, iy) in enumerate(theirs):
            (dst_engine or engine).import_tile(ix, iy, buf[k * nb:(k + 1) * nb])
            n += 1
    return n
