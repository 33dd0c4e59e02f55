"""Regenerates tests/golden/se3_vectors.json from the reference's own SE3 headers.
Run in the authoring container only (needs /root/reference): builds oracle/_ref/se3_ref
(oracle/Makefile target `ref`) and stores its output."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "se3_ref")])
open(os.path.join(HERE, "se3_vectors.json"), "wb").write(out)
print("wrote se3_vectors.json (%d bytes)" % len(out))
