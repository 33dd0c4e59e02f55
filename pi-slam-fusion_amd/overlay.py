"""Output side, row f4 of SURVEY 8: the web-map overlay message the reference emits for a refreshed tile
(Map2DFusion/MultiBandMap2DCPU.cpp:744-757): tile corners in plane coordinates, ROUNDED TO FLOAT (.cpp:709-712) ->
world (plane * p, GSLAM/core/SE3.h:99-101) -> longitude / latitude around GPS.Origin (pi::calcLngLatFromDistance,
PIL/src/hardware/Gps/utils_GPS.cpp:133-160) -> "Map2DUpdate LastTexMat <gpsTL> <gpsBR>".

Number format: the call site streams `setiosflags(ios::fixed) << setprecision(9) << gpsTl`, but operator<< of pi::Point3d
(GSLAM/core/Point.h:166-170) goes through std::to_string, so the stream only ever sees strings: every field has SIX
decimals ("%f"), the third field of each point is the default-constructed z = 0.  Pinned byte for byte by
tests/golden/gps_vectors.json (oracle/ref_gps.cpp, compiled with the reference's own utils_GPS.cpp and headers).

The same text comes out of the C ABI (pf_format_map_update / pf_map_update_command) for the C++ face."""
import math

import numpy as np

EARTH_RADIUS = 6378137.0          # WGS-84 semi-major axis, utils_GPS.cpp:16
DEG2RAD = 0.017453292519943        # the truncated constant of utils_GPS.cpp:19


def lnglat_from_distance(lng1, lat1, dx, dy):
    """pi::calcLngLatFromDistance, same operations in the same order."""
    f = 1.0 / 298.257223563
    e_2 = 2 * f - f * f
    phi = lat1 * DEG2RAD
    sp = math.sin(phi)
    lng_unit = DEG2RAD * EARTH_RADIUS * math.cos(phi) / math.sqrt(1 - e_2 * (sp * sp))
    lat_unit = DEG2RAD * EARTH_RADIUS * (1 - e_2) / math.pow(1 - e_2 * (sp * sp), 1.5)
    return dx / lng_unit + lng1, dy / lat_unit + lat1


def format_map_update(pf, plane, gps_origin, min_x, min_y, ele_size, x, y):
    """The scommand text for the tile at dense grid index (x, y): .cpp:709-712 + :747-755."""
    x0 = np.float32(min_x + x * ele_size); y0 = np.float32(min_y + y * ele_size)
    x1 = np.float32(float(x0) + ele_size); y1 = np.float32(float(y0) + ele_size)
    out = []
    for (cx, cy) in ((x0, y0), (x1, y1)):
        w = pf.se3_mul(plane, [float(cx), float(cy), 0.0, 0, 0, 0, 1])[:3]         # p->_plane * Point3d(x, y, 0)
        lng, lat = lnglat_from_distance(gps_origin[0], gps_origin[1], w[0], w[1])
        out.append("%f %f %f" % (lng, lat, 0.0))                                  # std::to_string per field
    return "Map2DUpdate LastTexMat " + " ".join(out)


def tile_overlay_command(pf, map2d, plane, gps_origin, ix, iy, fuse2google=True, high_quality_show=True):
    """The message draw() sends for tile (ix, iy) of `map2d` (stable tile coordinates), or None under the reference's gate
    `updated && !inborder && Fuse2Google` (.cpp:744): a tile on the rim of the dense grid has a 3x3 neighbour outside it
    (.cpp:730-735) and is never announced while HighQualityShow is on."""
    if not fuse2google:
        return None
    dims, geo = map2d.grid()
    x, y = ix - dims[2], iy - dims[3]
    if x < 0 or y < 0 or x >= dims[0] or y >= dims[1]:
        return None
    if high_quality_show and (x == 0 or y == 0 or x == dims[0] - 1 or y == dims[1] - 1):
        return None
    return format_map_update(pf, plane, gps_origin, geo[0], geo[1], geo[4], x, y)
