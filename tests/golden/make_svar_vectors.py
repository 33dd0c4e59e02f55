"""Generates tests/golden/svar_vectors.json: config.cfg texts (written here) and what the reference's own Svar parser
(GSLAM/core/Svar.h, header-only, compiled by oracle/Makefile target `ref` into oracle/_ref/svar_ref) reads out of them the
way its file driver does (backup/map2dfusion.cpp:153-192).  Build container only; the JSON is data and travels.
Run from the repo root:  python tests/golden/make_svar_vectors.py"""
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

TEXTS = [
    # the shape of the DroneMap / NPU config files
    "Plane = 1.5 -2.25 3 0 0 0.1 0.99\nCamera.Paraments = [4000 3000 3000.5 3000 2000 1500]\nGPS.Origin = 108.9 34.2 400\n",
    # commas, comments, blank lines, tabs, leading blanks, a key given twice (the later one counts)
    "// phantom3\n\n  Camera.Paraments\t=  [1920,1080,1100.25,1101.5,960,540]\nPrepareFrameNum=5   // five frames\nPlane = 0 0 0 0 0 0 1\nPlane = 10 20 -30 0.01 -0.02 0.03 0.9993\n",
    # scientific notation and signs, no GPS origin, scale
    "Plane = 1e1 -2.5E-1 +3 1e-3 -1e-3 0 1\nCamera.Paraments = [640 480 5e2 500 3.2e2 240]\nMap2D.Scale = 0.5\n",
    # no plane at all: the driver falls back to pi::SE3d()
    "Camera.Paraments = [640 480 500 500 320 240]\nGPS.Origin = -71.06 42.36 10.5\n",
    # `?=` only sets what is not set yet
    "PrepareFrameNum = 7\nPrepareFrameNum ?= 3\nMap2D.Scale ?= 2\nCamera.Paraments = [ 800 600 700 700 400 300 ]\nPlane = 0 0 5 0 0 0 1\n",
    # wrong number of camera values (the driver then says "Invalid camera parameters!")
    "Camera.Paraments = [4000 3000 3000 3000]\nPlane = 0 0 0 0 0 0 1\n",
]

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i, t in enumerate(TEXTS):
        p = os.path.join(d, "c%d.cfg" % i)
        open(p, "w").write(t)
        paths.append(p)
    # one process per file: the reference keeps typed variables (get_var<T>) in process-wide singletons
    parsed = [json.loads(subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "svar_ref"), p]).decode())[0] for p in paths]
assert len(parsed) == len(TEXTS)
json.dump([{"text": t, "reference": r} for t, r in zip(TEXTS, parsed)], open(os.path.join(HERE, "svar_vectors.json"), "w"), indent=1)
print("wrote", len(parsed), "config cases")
for r in parsed:
    print(r)
