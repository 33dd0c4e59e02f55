"""Generates tests/golden/ransac_vectors.json by running oracle/_ref/ransac_ref -- the reference's own
src/RANSAC.cpp / src/DataTrans.h compiled where they lie (oracle/Makefile target `ref`, oracle/ref_ransac.cpp).
Only possible in the build container (the reference tree is absent on the GPU box); the JSON is data and travels.
Run from the repo root:  python tests/golden/make_ransac_vectors.py"""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ransac_ref")]).decode()
vec = json.loads(out)
assert len(vec["planes"]) == 48 and vec["datatrans"]["max"] == 30
json.dump(vec, open(os.path.join(HERE, "ransac_vectors.json"), "w"), indent=0)
print("wrote", len(vec["planes"]), "plane cases")
