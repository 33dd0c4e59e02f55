"""The selectable forms of the level kernel that the default run does not take, each through the same oracle parity
tests in a child process (the switches are read once per process):
  PF_PATCH=1         LDS-staged source / weight patch for the warp (profiles/r03_patch_stage_a.md)
  PF_WEIGHT_PLANE=1  radial weight gathered from the fp32 weight plane instead of computed
  PF_TABLE_COPY=1    tile tables staged and copied in the stream instead of travelling in the kernel arguments
  PF_A_ILP=2 / 3     two / three warp rows per step for both pyramid types (defaults: fp32 3, int16 2)
Reference path: Map2DFusion/MultiBandMap2DCPU.cpp:311-558 (renderFrame)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("switch", ["PF_PATCH", "PF_WEIGHT_PLANE", "PF_TABLE_COPY", "PF_A_ILP=2", "PF_A_ILP=3"])
def test_variant_equals_oracle(switch):
    name, _, val = switch.partition("=")
    env = dict(os.environ, **{name: val or "1"})
    # fused=1 cases of the plumbing and perspective tests: both pyramid types, noise and smooth frames, 0..8 bands, spreadMap
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
           "-k", "(cfg1_plumbing or perspective_and_spread) and -1]"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and " passed" in out, out[-3000:]
