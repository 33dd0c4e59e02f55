import importlib, json, os, sys, time
sys.path.insert(0, "/root/repo")
import bench, torch, numpy as np
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = bench.CAM; K, W = 200, 20
poses = wl.serpentine(cam, 100.0, K + W)
fr = [torch.randint(0, 256, (cam[1], cam[0], 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
for la, res in ((48, 3300), (48, 3300), (0, 3300)):
    m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1, lookahead=la)
    assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
    if res: m.reserve_tiles(res)
    for k in range(W): m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
    m.sync(); torch.cuda.synchronize()
    import gc; gc.collect(); gc.disable()
    ts = []
    t0 = time.perf_counter()
    for k in range(W, W + K):
        a = time.perf_counter(); m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k]); ts.append(time.perf_counter() - a)
    t1 = time.perf_counter(); m.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
    gc.enable()
    ts = np.array(ts) * 1e6
    big = [(int(i), round(float(v))) for i, v in enumerate(ts) if v > 300]
    print(json.dumps({"lookahead": la, "reserved": res, "feed_loop_ms": round((t1 - t0) * 1e3, 2), "with_sync_ms": round((t2 - t0) * 1e3, 2), "median_feed_us": round(float(np.median(ts)), 1), "feeds_over_300us": big[:20]}), flush=True)
    m.close()
