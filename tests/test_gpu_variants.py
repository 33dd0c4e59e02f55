"""The selectable forms of the level kernel that the default run does not take, each through the same oracle parity
tests in a child process (the switches are read once per process):
  PF_PATCH=1         LDS-staged source / weight patch for the warp (profiles/r03_patch_stage_a.md)
  PF_WEIGHT_PLANE=1  radial weight gathered from the fp32 weight plane instead of computed
  PF_TABLE_COPY=1    tile tables staged and copied in the stream instead of travelling in the kernel arguments
  PF_A_ILP=2 / 3     two / three warp rows per step for both pyramid types (defaults: fp32 3, int16 2)
  PF_STRIPS=1        the wave-specialised rolling-strip form (profiles/r04_strips.md)
  PF_BLOCK28=1       64x28 blocks
  PF_BLOCK24=1       64x24 blocks with the unpadded 71-pixel LDS pitch: 40 928 B and 64 VGPRs, four workgroups per CU for fp32 too (profiles/r06_ab.md)
  PF_A_ILP=0         int16: stage A with deferred finishes (fp32's product form)
  PF_SEED=1          fp32: the reciprocal of a pixel's W seeded from the row above (kernels.hip rcp_seeded)
  PF_COMPACT=1       a shard's level-0 job on the compact grid (one workgroup per block inside its need rectangles)
  PF_CULL=0          every tile of every canvas rendered (no cull)
  PF_CULL_SUB=2      the cull per quadrant of a tile instead of per 64 x 64 cell
Forms that were measured and not adopted, and every A/B or test switch of a form the product does carry (the weight plane of large frames,
the copied tile tables of large canvases), live in the second build of the library (libpifusion_exp.so, -DPF_EXPERIMENTS=1), which the
child processes load through PF_LIB; the product library carries the product instantiations and reads PF_CULL alone of these.
Reference path: Map2DFusion/MultiBandMap2DCPU.cpp:311-558 (renderFrame)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# which counter of pf_debug_form_counts must (or must not) move under a switch: the parity run alone would also pass on a silent
# fall-back to the default form
EXP_LIB = os.path.join(ROOT, "pi-slam-fusion_amd", "libpifusion_exp.so")
# every switch but PF_CULL (a documented runtime option of the product library, csrc/env.hpp) exists in the experiments library only
NEEDS_EXP = {"PF_PATCH", "PF_A_ILP", "PF_STRIPS", "PF_BLOCK28", "PF_BLOCK24", "PF_SEED", "PF_COMPACT", "PF_WEIGHT_PLANE", "PF_TABLE_COPY", "PF_CULL_SUB"}
FORM = {"PF_PATCH": "c[2] > 0", "PF_WEIGHT_PLANE": "c[1] > 0 and c[0] == 0", "PF_TABLE_COPY": "c[7] == 0 and c[0] > 0",
        "PF_A_ILP=2": "c[0] > 0", "PF_A_ILP=3": "c[0] > 0", "PF_A_ILP=0": "c[0] > 0", "PF_SEED": "c[0] > 0", "PF_COMPACT": "c[0] > 0", "PF_STRIPS": "c[3] > 0 and c[0] == 0", "PF_BLOCK28": "c[5] > 0 and c[0] == 0", "PF_BLOCK24": "c[5] > 0 and c[0] == 0",
        "PF_CULL=0": "culled == 0 and c[0] > 0", "PF_CULL_SUB=2": "culled > 0 and g.culled_cells() % 4 == 0 and c[0] > 0"}
PROBE = """
import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)
import ctypes as C, numpy as np
from helpers import workloads, jitter_poses, compare_maps
from conftest import load_package
from oracle import orc
pf = load_package(); wl = workloads()
cam = [640, 480, 500, 500, 320, 240]
poses = jitter_poses(6, seed=3, yaw_deg=3.0, tilt_deg=1.0)        # nearly nadir: the homography the PATCH plan accepts
g = pf.Map2D.create(pf.TypeMultiBandCPU, False, force_float=1, scale=2.5); o = orc.OracleMap(force_float=1, scale=2.5)
assert g.prepare(wl.IDENTITY_PLANE, cam, poses) and o.prepare(wl.IDENTITY_PLANE, cam, poses)
for k, p in enumerate(poses + poses):
    img = wl.noise_frame(480, 640, k); assert g.feed(img, p) and o.feed(img, p)
g.sync(); assert compare_maps(g, o) == []
c = (C.c_longlong * 8)(); pf.lib().pf_debug_form_counts(c); c = list(c); culled = g.culled_tiles()
print("forms", c, "culled", culled)
assert %s, (c, culled)
"""


@pytest.mark.parametrize("switch", ["PF_PATCH", "PF_WEIGHT_PLANE", "PF_TABLE_COPY", "PF_A_ILP=2", "PF_A_ILP=3", "PF_A_ILP=0", "PF_SEED", "PF_COMPACT", "PF_STRIPS", "PF_BLOCK28", "PF_BLOCK24",
                                    "PF_CULL=0", "PF_CULL_SUB=2"])
def test_variant_equals_oracle(switch):
    name, _, val = switch.partition("=")
    env = dict(os.environ, **{name: val or "1"})
    if name in NEEDS_EXP:
        assert os.path.exists(EXP_LIB), "build the experiments library first (__graft_entry__.build())"
        env["PF_LIB"] = EXP_LIB
    # the requested form really runs (and gives the oracle's tiles) ...
    probe = PROBE % (ROOT, os.path.join(ROOT, "tests"), FORM[switch])
    r = subprocess.run([sys.executable, "-c", probe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    # ... and passes the parity tests
    # fused=1 cases of the plumbing and perspective tests: both pyramid types, noise and smooth frames, 0..8 bands, spreadMap
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
           "-k", "(cfg1_plumbing or perspective_and_spread) and -1]"]
    if name == "PF_COMPACT":        # the compact grid serves shards: the sharding tests, which assert that it ran
        cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_sharding.py"), "-x", "-q", "-m", "gpu"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and " passed" in out, out[-3000:]
