"""config.cfg wire format of the dataset drivers: the Python reader (pi-slam-fusion_amd/dataset.py) and the
C++ reader (include/pifusion/TestSystem.h, Config) against what the reference's own parser reads out of the same texts --
tests/golden/svar_vectors.json, made by tests/golden/make_svar_vectors.py with GSLAM/core/Svar.h compiled where it lies
(oracle/ref_svar.cpp): Plane (SE3 stream order x y z qx qy qz qw, default pi::SE3d()), Camera.Paraments (VecParament),
GPS.Origin, PrepareFrameNum, Map2D.Scale; `?=` assigns only what is unset, a later `=` overrides, `//` comments."""
import importlib
import json
import os
import subprocess

from conftest import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC = json.load(open(os.path.join(ROOT, "tests", "golden", "svar_vectors.json")))


def test_python_reader_equals_reference_parser(tmp_path):
    load_package()
    ds = importlib.import_module("pi_slam_fusion_amd.dataset")
    assert len(VEC) >= 6
    for i, c in enumerate(VEC):
        p = tmp_path / ("config%d.cfg" % i)
        p.write_text(c["text"])
        cfg, r = ds.parse_config(str(p)), c["reference"]
        assert ("Plane" in cfg) == bool(r["has_plane"])
        assert cfg.get("Plane", [0, 0, 0, 0, 0, 0, 1]) == r["plane"]
        assert cfg.get("Camera.Paraments") == r["camera"]
        assert ("GPS.Origin" in cfg) == bool(r["has_gps"])
        assert (cfg.get("GPS.Origin") or []) == [float(x) for x in r["gps"].split()]
        assert int(cfg.get("PrepareFrameNum", [10])[0]) == r["prepare"]
        assert cfg.get("Map2D.Scale", [1.0])[0] == r["scale"]


def test_cpp_reader_equals_reference_parser(tmp_path):
    exe = str(tmp_path / "config_dump")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "config_dump.cpp"),
                           "-o", exe, "-L" + os.path.join(ROOT, "pi-slam-fusion_amd"), "-l:libpifusion.so", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "pi-slam-fusion_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    for i, c in enumerate(VEC):
        p = tmp_path / ("config%d.cfg" % i)
        p.write_text(c["text"])
        got = json.loads(subprocess.check_output([exe, str(p)]).decode())
        assert got == c["reference"], (i, got, c["reference"])
