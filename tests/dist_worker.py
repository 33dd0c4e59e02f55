"""One rank of the multi-process seam-exchange test (tests/test_gpu_dist.py): shard map + the library's pf_dist_* path,
checked on rank 0 against the oracle.  Launched by torch.distributed.run; PF_DIST_BACKEND = nccl (one GPU per rank, RCCL
inside the library) or gloo (ranks share GPU 0, host-buffer transport of the library over torch point-to-point)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package          # noqa: E402
from helpers import jitter_poses           # noqa: E402


def main():
    backend = os.environ.get("PF_DIST_BACKEND", "gloo")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, **({"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}))
    pf = load_package()
    wl = importlib.import_module("pi_slam_fusion_amd.workloads")
    sh = importlib.import_module("pi_slam_fusion_amd.sharding")
    ff = int(os.environ.get("PF_TEST_FLOAT", "0"))
    cam = [640, 480, 500, 500, 320, 240]
    base = jitter_poses(9, seed=17, step=(0.0, 0.0))
    poses = [[(k % 3) * 70.0 + p[0], (k // 3) * 55.0 + p[1]] + p[2:] for k, p in enumerate(base)]
    frames = [wl.smooth_frame(480, 640, k) ^ wl.noise_frame(480, 640, k) for k in range(len(poses))]
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=ff, scale=2.0, device=dev, shard_rank=rank, shard_count=world, shard_block=1)
    assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
    d = sh.DistMap(m, rank, world, backend="nccl" if backend == "nccl" else "host")
    info = d.info()
    assert info["nranks"] == world and info["rank"] == rank, info       # what the library's own transport sees
    d.set_verify(True)                                                   # every exchange hashed on both ends (a first multi-GPU run names the pair that moved wrong bytes)
    # the keyframes live on rank 0 only: pf_dist_feed moves each to the ranks that own part of its canvas
    for f, p in zip(frames, poses):
        assert d.feed(f if rank == 0 else None, p, root=0, shape=f.shape)
    coords, px = d.blend_changed()
    st = d.stats()
    assert world == 1 or st["verified"] == 1, st
    coords2, _ = d.blend_changed()                       # Ischanged flags were cleared: nothing to do the second time
    saved = d.save_to_memory()
    got = [None] * world
    dist.gather_object({"coords": coords, "px": px, "again": len(coords2), "stats": st, "tiles": m.tiles()}, got if rank == 0 else None, dst=0)
    if rank == 0:
        from oracle import orc
        o = orc.OracleMap(force_float=ff, scale=2.0)
        assert o.prepare(wl.IDENTITY_PLANE, cam, poses[:2])
        for f, p in zip(frames, poses):
            assert o.feed(f, p)
        seen, moved = set(), 0
        for r, g in enumerate(got):
            assert g["again"] == 0
            assert sorted(g["coords"]) == sorted(g["tiles"]), "rank %d blended %d of its %d tiles" % (r, len(g["coords"]), len(g["tiles"]))
            moved += g["stats"]["bytes_received"]
            for t, im in zip(g["coords"], g["px"]):
                assert pf.tile_owner(m.opt, *t) == r and t not in seen
                seen.add(t)
                assert np.array_equal(im, o.blend_tile(*t)), "rank %d tile %s differs from the oracle's blend" % (r, t)
        assert seen == set(o.tiles()) and moved > 0
        ref, org = o.save()
        assert saved[1] == org and np.array_equal(saved[0], ref)
        print("DIST OK backend=%s transport=%s world=%d float=%d tiles=%d seam_bytes=%d" % (backend, info["transport"], world, ff, len(seen), moved), flush=True)
    dist.barrier()
    d.close(); m.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
