"""Multi-GPU seam exchange for the tile-sharded mosaic.

feed() needs no collective: every rank is fed every keyframe's pose, renders the part of
the frame that lands on tiles it owns (owner = spatial hash of the tile, pf_tile_owner) and
recomputes the pyramid halo from the source frame.  The only exchange steps are

  * Ele::blend (MultiBandMap2DCPU.cpp:77-146): a tile's 3x3 neighbourhood may live on other
    ranks; their edge strips (border 1<<(L-i) pixels at level i) are packed on the owner and
    moved with ONE all_to_all (point-to-point traffic over xGMI, all links at once);
  * save (.cpp:779-847): whole tiles are gathered to rank 0, which runs the mosaic collapse.

On a GPU the exchange lives in the library (csrc/dist.cpp, C ABI pf_dist_*: one pack launch, one grouped
ncclSend/ncclRecv exchange, one batched blend per call); `DistMap` below is its thin caller.  The functions
after it are the same plan in Python over torch.distributed and an engine interface: the CPU model of the
exchange that tests/test_sharding_gloo.py runs with an oracle-backed engine (no GPU there).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

NEIGHBOURS = [(dx, dy) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]      # index j = 3*(dy+1)+(dx+1)


def _pkg():
    import sys
    return sys.modules[__name__.rsplit(".", 1)[0]]


def torch_exchange(group=None):
    """pf_exchange_fn over torch.distributed point-to-point (works with gloo): the host-buffer transport of the
    library for rehearsals where RCCL cannot be used (several ranks sharing one GPU)."""
    def fn(user, send, send_bytes, recv, recv_bytes, n):
        try:
            me = dist.get_rank(group)
            ops, keep = [], []
            for p in range(n):
                if p == me:
                    continue
                if send_bytes[p]:
                    t = torch.frombuffer((C.c_char * send_bytes[p]).from_address(send[p]), dtype=torch.uint8)
                    keep.append(t); ops.append(dist.P2POp(dist.isend, t, p, group))
                if recv_bytes[p]:
                    t = torch.frombuffer((C.c_char * recv_bytes[p]).from_address(recv[p]), dtype=torch.uint8)
                    keep.append(t); ops.append(dist.P2POp(dist.irecv, t, p, group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return 1
        except Exception as e:                              # never let an exception cross the C boundary
            print("torch_exchange failed:", e)
            return 0
    return fn


class DistMap:
    """Thin caller of the library's seam exchange (pf_dist_*).  backend "nccl": RCCL, the unique id made on rank 0 and
    broadcast through torch.distributed; anything else: the host-buffer hook over `exchange` (default: torch p2p)."""

    def __init__(self, m, rank, nranks, backend="nccl", exchange=None, group=None):
        pf = _pkg()
        L = pf.lib()
        self.m, self.rank, self.nranks, self._fn = m, rank, nranks, None
        if backend == "nccl":
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                assert L.pf_dist_unique_id(uid.data_ptr()), L.pf_last_error().decode()
            if nranks > 1:
                dev = uid.cuda() if dist.get_backend(group) == "nccl" else uid
                dist.broadcast(dev, 0, group=group)
                uid = dev.cpu()
            self._h = L.pf_dist_init_rccl(m._h, uid.data_ptr(), rank, nranks)
        else:
            self._fn = pf.EXCHANGE_FN(exchange or torch_exchange(group))
            self._h = L.pf_dist_init_host(m._h, rank, nranks, self._fn, None)
        if not self._h:
            raise RuntimeError("pf_dist_init failed: %s" % L.pf_last_error().decode())

    def close(self):
        if self._h:
            _pkg().lib().pf_dist_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def blend_changed(self, cap=None):
        """draw() across ranks: this rank's changed tiles, blended with remote neighbour strips -> (coords, pixels)"""
        pf = _pkg()
        cap = max(1, len(self.m.tiles())) if cap is None else cap
        xy = (C.c_int * (2 * max(cap, 1)))()
        out = np.empty((max(cap, 1), pf.ELE_PIXELS, pf.ELE_PIXELS, 3), np.uint8)
        n = pf.lib().pf_dist_blend_changed(self._h, xy, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("pf_dist_blend_changed failed: %s" % pf.lib().pf_last_error().decode())
        return [(xy[2 * i], xy[2 * i + 1]) for i in range(n)], out[:n]

    def save_to_memory(self):
        """save() across ranks: rank 0 gets (mosaic, origin tile); the other ranks get None"""
        L = _pkg().lib()
        r, c, x0, y0 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if not L.pf_dist_save_to_memory(self._h, None, C.byref(r), C.byref(c), C.byref(x0), C.byref(y0)):
            raise RuntimeError("pf_dist_save_to_memory failed: %s" % L.pf_last_error().decode())
        out = np.empty((max(r.value, 1), max(c.value, 1), 3), np.uint8)
        if not L.pf_dist_save_to_memory(self._h, out.ctypes.data, C.byref(r), C.byref(c), C.byref(x0), C.byref(y0)):
            raise RuntimeError("pf_dist_save_to_memory failed: %s" % L.pf_last_error().decode())
        return (out, (x0.value, y0.value)) if self.rank == 0 else None

    def save(self, filename):
        return bool(_pkg().lib().pf_dist_save(self._h, filename.encode()))

    def stats(self):
        st = _pkg().DistStats()
        _pkg().lib().pf_dist_last_stats(self._h, C.byref(st))
        return {k: getattr(st, k) for k, _ in st._fields_}


def strong_report(m, rank, world, backend, group=None):
    """bench.py --shard strong: what sharding one sortie costs -- per-rank halo recompute, and the seam exchange timed:
    draw() of every tile across ranks, then save()'s gather + collapse on rank 0.  Collective; rank 0 gets the record."""
    import time
    d = DistMap(m, rank, world, backend="nccl" if backend == "nccl" else "host", group=group)
    rs = m.render_stats()
    m.sync()
    if world > 1:
        dist.barrier(group)
    t0 = time.perf_counter(); coords, _ = d.blend_changed(); t_blend = time.perf_counter() - t0
    st_b = d.stats()
    if world > 1:
        dist.barrier(group)
    t0 = time.perf_counter(); saved = d.save_to_memory(); t_save = time.perf_counter() - t0
    st_s = d.stats()
    mine = {"rank": rank, "tiles": len(m.tiles()), "blended": len(coords), "render": rs, "blend_s": t_blend, "blend": st_b,
            "save_s": t_save, "save": st_s, "mosaic": None if saved is None else list(saved[0].shape[:2])}
    got = [None] * world
    if world > 1:
        dist.all_gather_object(got, mine, group=group)
    else:
        got = [mine]
    d.close()
    if rank != 0:
        return None
    seam = sum(g["blend"]["bytes_received"] for g in got)
    xms = max(g["blend"]["exchange_ms"] for g in got)
    gathered = got[0]["save"]["bytes_received"]
    return {
        "transport": "rccl" if backend == "nccl" else "host-buffer hook over torch.distributed (%s)" % backend,
        "tiles_per_rank": [g["tiles"] for g in got],
        "halo_recompute_factor_per_rank": [round(g["render"]["level0_px"] / max(g["render"]["owned_px"], 1.0), 3) for g in got],
        "frames_with_pixels_per_rank": [g["render"]["frames_with_pixels"] for g in got],
        "draw": {"tiles": sum(g["blended"] for g in got), "seconds_max": round(max(g["blend_s"] for g in got), 4),
                 "seam_bytes": seam, "strips": sum(g["blend"]["strips_received"] for g in got),
                 "exchange_ms_max": round(xms, 3), "seam_GBps": round(seam / max(xms, 1e-6) / 1e6, 2),
                 "pack_ms_max": round(max(g["blend"]["pack_ms"] for g in got), 3),
                 "blend_ms_max": round(max(g["blend"]["compute_ms"] for g in got), 3)},
        "save": {"mosaic": got[0]["mosaic"], "seconds_rank0": round(got[0]["save_s"], 4), "gathered_bytes": gathered,
                 "gather_ms": round(got[0]["save"]["exchange_ms"], 3),
                 "gather_GBps": round(gathered / max(got[0]["save"]["exchange_ms"], 1e-6) / 1e6, 2),
                 "collapse_ms": round(got[0]["save"]["compute_ms"], 3)},
    }


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
        if sbuf.is_cuda:
            torch.cuda.synchronize()              # the engine packed on its own stream; the collective runs on torch's
        dist.all_to_all_single(rbuf[:sum(recv_sizes)], sbuf[:sum(send_sizes)], recv_sizes, send_sizes, group=group)
        if rbuf.is_cuda:
            torch.cuda.synchronize()              # ... and the engine reads rbuf on its own stream: the bytes must have landed
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
            if buf.is_cuda:
                torch.cuda.synchronize()
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
        if buf.is_cuda:
            torch.cuda.synchronize()              # import_tile reads buf on the engine's stream
        for k, (ix, iy) in enumerate(theirs):
            (dst_engine or engine).import_tile(ix, iy, buf[k * nb:(k + 1) * nb])
            n += 1
    return n
