#!/usr/bin/env python3
"""Regenerates tests/golden/gps_vectors.json from the reference's own utils_GPS.cpp / SE3.h / Point.h (needs /root/reference:
`make -C oracle ref` builds oracle/_ref/gps_ref from oracle/ref_gps.cpp + the reference's PIL/src/hardware/Gps/utils_GPS.cpp)."""
import json, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "-s", "-C", os.path.join(R, "oracle"), "ref"])
out = subprocess.check_output([os.path.join(R, "oracle", "_ref", "gps_ref")]).decode()
j = json.loads(out)
assert len(j["lnglat"]) >= 16 and len(j["messages"]) >= 16
json.dump(j, open(os.path.join(R, "tests", "golden", "gps_vectors.json"), "w"), indent=0)
print("wrote %d + %d vectors" % (len(j["lnglat"]), len(j["messages"])))
