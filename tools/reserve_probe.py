#!/usr/bin/env python3
"""What on-demand slab allocation costs a sortie: bench.py's cfg-A keyframes (11 flight lines, ~4950 tiles) into maps whose tile store was pre-sized
(reserve_tiles) for all of them, for two thirds of them, or not at all; with and without the cull's lookahead."""
import importlib, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench, torch
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = bench.CAM; K, W = 200, 20
poses = wl.serpentine(cam, 100.0, K + W)
fr = [torch.randint(0, 256, (cam[1], cam[0], 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
for la in (0, 48):
    for res in (6000, 3300, 0):
        m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1, lookahead=la)
        assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
        if res: m.reserve_tiles(res)
        for k in range(W): m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
        m.sync(); torch.cuda.synchronize()
        import gc; gc.collect(); gc.disable()
        t0 = time.perf_counter()
        for k in range(W, W + K): m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
        t1 = time.perf_counter(); m.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
        gc.enable()
        print(json.dumps({"lookahead": la, "reserved": res, "tiles": len(m.tiles()), "feed_loop_ms": round((t1 - t0) * 1e3, 2), "with_sync_ms": round((t2 - t0) * 1e3, 2),
                          "kfs": round(K / (t2 - t0), 1)}), flush=True)
        m.close()
