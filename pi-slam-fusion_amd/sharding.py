"""Multi-GPU seam exchange for the tile-sharded mosaic.

feed() needs no collective: every rank is fed every keyframe's pose, renders the part of
the frame that lands on tiles it owns (owner = spatial hash of the tile, pf_tile_owner) and
recomputes the pyramid halo from the source frame.  The only exchange steps are

  * Ele::blend (MultiBandMap2DCPU.cpp:77-146): a tile's 3x3 neighbourhood may live on other
    ranks; their edge strips (border 1<<(L-i) pixels at level i) are packed on the owner and
    moved with ONE all_to_all (point-to-point traffic over xGMI, all links at once);
  * save (.cpp:779-847): whole tiles are gathered to rank 0, which runs the mosaic collapse.

The exchange lives in the library (csrc/dist.cpp, C ABI pf_dist_*: one pack launch, one grouped ncclSend/ncclRecv
exchange, one batched blend per call); `DistMap` below is its thin caller, `torch_exchange` the host-buffer hook for
launchers without RCCL, `plan_blend` the library's own exchange plan (pf_dist_plan_blend, a pure function) as the CPU
tests replay it between gloo processes (tests/test_sharding_gloo.py).
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

    def feed(self, img, pose, root=0, shape=None):
        """Map2D::feed across ranks (pf_dist_feed): `img` (HxWx3/4 uint8, host) is read on `root` only; the other ranks
        pass None and the frame's shape (rows, cols, channels).  Returns True / False like Map2D.feed."""
        pf = _pkg()
        p = np.ascontiguousarray(pose, dtype=np.float64).reshape(-1)
        if img is not None:
            img = np.ascontiguousarray(img, dtype=np.uint8)
            shape = img.shape
        rows, cols, ch = shape
        im = pf.Image(rows, cols, pf.PF_8UC3 if ch == 3 else pf.PF_8UC4, img.ctypes.data if img is not None else None, 0)
        rc = pf.lib().pf_dist_feed(self._h, C.byref(im), p.ctypes.data_as(C.POINTER(C.c_double)), root)
        if rc < 0:
            raise RuntimeError("pf_dist_feed failed: %s" % pf.lib().pf_last_error().decode())
        return bool(rc)

    def feed_jpeg(self, data, pose, shape, root=0):
        """pf_dist_feed_jpeg: the keyframe as the bytes of its .jpg file, read on `root` only (the others pass None); every rank gives the
        frame's (rows, cols).  The root decodes on its GPU into the slot the exchange sends from."""
        pf = _pkg()
        p = np.ascontiguousarray(pose, dtype=np.float64).reshape(-1)
        b = bytes(data) if data is not None else None
        rc = pf.lib().pf_dist_feed_jpeg(self._h, b, len(b) if b is not None else 0, int(shape[0]), int(shape[1]), p.ctypes.data_as(C.POINTER(C.c_double)), root)
        if rc < 0:
            raise RuntimeError("pf_dist_feed_jpeg failed: %s" % pf.lib().pf_last_error().decode())
        return bool(rc)

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

    def set_verify(self, on=True):
        """hash every data exchange of the following calls on both ends (pf_dist_set_verify)"""
        _pkg().lib().pf_dist_set_verify(self._h, 1 if on else 0)

    def info(self):
        """(rank, nranks, transport name) as the library's transport sees the group"""
        r, n, name = C.c_int(), C.c_int(), C.c_char_p()
        _pkg().lib().pf_dist_info(self._h, C.byref(r), C.byref(n), C.byref(name))
        return {"rank": r.value, "nranks": n.value, "transport": (name.value or b"").decode()}

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
    # a first, small draw() with every exchanged byte hashed on both ends (says which pair moved wrong bytes, if any), then
    # the timed one over the remaining tiles without the check
    d.set_verify(True)
    first, _ = d.blend_changed(cap=min(32, max(1, len(m.tiles()))))
    verified = d.stats()["verified"]
    d.set_verify(False)
    if world > 1:
        dist.barrier(group)
    t0 = time.perf_counter(); coords, _ = d.blend_changed(); t_blend = time.perf_counter() - t0
    st_b = d.stats()
    st_b["verified"] = verified
    coords = first + coords
    if world > 1:
        dist.barrier(group)
    t0 = time.perf_counter(); saved = d.save_to_memory(); t_save = time.perf_counter() - t0
    st_s = d.stats()
    mine = {"rank": rank, "tiles": len(m.tiles()), "blended": len(coords), "render": rs, "blend_s": t_blend, "blend": st_b, "info": d.info(),
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
        # what the library's own communicator reports on every rank (a first multi-GPU run checks this before anything else)
        "communicator": {"nranks_seen": [g["info"]["nranks"] for g in got], "ranks_seen": [g["info"]["rank"] for g in got],
                         "transport": got[0]["info"]["transport"],
                         "verified_exchanges": [g["blend"]["verified"] + g["save"]["verified"] for g in got]},
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


class StripPlan(C.Structure):
    _fields_ = [("peer", C.c_int), ("ix", C.c_int), ("iy", C.c_int), ("dx", C.c_int), ("dy", C.c_int), ("tile", C.c_int),
                ("offset", C.c_ulonglong)]


def plan_blend(lists, caps, me, high_quality, halo_bytes9):
    """The library's plan of one pf_dist_blend_changed call (csrc/dist.cpp plan_blend through pf_dist_plan_blend).
    lists[r] = [(ix, iy, changed), ...] of rank r; returns (send, recv, mine): lists of dicts / tile coordinates."""
    L = _pkg().lib()
    n = len(lists)
    counts = (C.c_int * n)(*[len(l) for l in lists])
    flat = [v for l in lists for t in l for v in (int(t[0]), int(t[1]), int(t[2]))]
    recs = (C.c_int * max(len(flat), 1))(*flat)
    cp = (C.c_longlong * n)(*[int(c) for c in caps])
    hb = (C.c_size_t * 9)(*[int(b) for b in halo_bytes9])
    ns, nr, nm = C.c_int(), C.c_int(), C.c_int()
    L.pf_dist_plan_blend(n, me, counts, recs, cp, 1 if high_quality else 0, hb, None, 0, C.byref(ns), None, 0, C.byref(nr), None, 0, C.byref(nm))
    send = (StripPlan * max(ns.value, 1))(); recv = (StripPlan * max(nr.value, 1))(); mine = (C.c_int * max(2 * nm.value, 1))()
    ok = L.pf_dist_plan_blend(n, me, counts, recs, cp, 1 if high_quality else 0, hb, send, ns.value, C.byref(ns),
                              recv, nr.value, C.byref(nr), mine, nm.value, C.byref(nm))
    if not ok:
        raise RuntimeError("pf_dist_plan_blend failed")
    rec = lambda q: {k: getattr(q, k) for k, _ in StripPlan._fields_}
    return ([rec(send[i]) for i in range(ns.value)], [rec(recv[i]) for i in range(nr.value)],
            [(mine[2 * i], mine[2 * i + 1]) for i in range(nm.value)])
