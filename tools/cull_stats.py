import importlib, os, sys
os.environ.setdefault("PF_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pi-slam-fusion_amd", "libpifusion_exp.so"))   # the switches below exist in the experiments build only (csrc/env.hpp)
sys.path.insert(0, os.getcwd())
import bench, torch
pf = bench.load_package(); wl = importlib.import_module("pi_slam_fusion_amd.workloads")
cam = bench.CAM; poses = wl.serpentine(cam, 100.0, 220)
import ctypes as C
L = pf.lib()
m = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1)
assert m.prepare(wl.IDENTITY_PLANE, cam, poses[:20])
fr = [torch.randint(0, 256, (cam[1], cam[0], 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
for k in range(20): m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
m.sync(); a = m.render_stats()
ct0 = m.culled_tiles(); L.pf_debug_culled_cells.restype = C.c_longlong; L.pf_debug_culled_cells.argtypes = [C.c_void_p]; cq0 = L.pf_debug_culled_cells(m._h)
for k in range(20, 220): m.feed_device(fr[k % 4].data_ptr(), cam[1], cam[0], poses[k])
m.sync(); b = m.render_stats()
n = 200
L.pf_debug_level0_exact_px.restype = C.c_double; L.pf_debug_level0_exact_px.argtypes = [C.c_void_p]
print("cell px culled per keyframe (frames 20..219): %.2f M" % (((m.culled_tiles() - ct0) * 16 + (L.pf_debug_culled_cells(m._h) - cq0)) * 4096 / 200 / 1e6))
print("exact rule (PF_CULL_EXACT_STAT=1): %.2f M per keyframe incl. the first 20" % (L.pf_debug_level0_exact_px(m._h) / 220 / 1e6))
print("per keyframe: level-0 px run %.2f M, tile px not culled %.2f M (canvas 14.48 M); frames with pixels %d" % ((b["level0_px"] - a["level0_px"]) / n / 1e6, (b["owned_px"] - a["owned_px"]) / n / 1e6, b["frames_with_pixels"] - a["frames_with_pixels"]))
