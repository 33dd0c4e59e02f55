"""Output side, row f4 of SURVEY 8: the web-map overlay message the reference emits for a
refreshed tile (Map2DFusion/MultiBandMap2DCPU.cpp:744-760): tile corners in plane coordinates ->
world (plane * p) -> longitude/latitude around GPS.Origin (pi::calcLngLatFromDistance,
PIL/src/hardware/Gps/utils_GPS.cpp:133-160) -> "Map2DUpdate LastTexMat <gpsTL> <gpsBR>"."""
import math

EARTH_RADIUS = 6378137.0          # WGS-84 semi-major axis, utils_GPS.cpp
DEG2RAD = 0.017453292519943        # the truncated constant of utils_GPS.cpp:19


def lnglat_from_distance(lng1, lat1, dx, dy):
    f = 1.0 / 298.257223563
    e_2 = 2 * f - f * f
    phi = lat1 * DEG2RAD
    lng_unit = DEG2RAD * EARTH_RADIUS * math.cos(phi) / math.sqrt(1 - e_2 * math.sin(phi) ** 2)
    lat_unit = DEG2RAD * EARTH_RADIUS * (1 - e_2) / math.pow(1 - e_2 * math.sin(phi) ** 2, 1.5)
    return dx / lng_unit + lng1, dy / lat_unit + lat1


def tile_overlay_command(pf, map2d, plane, gps_origin, ix, iy):
    """The scommand string for tile (ix, iy) of `map2d` (stable tile coordinates)."""
    dims, geo = map2d.grid()
    ele = geo[4]
    x0 = geo[0] + (ix - dims[2]) * ele
    y0 = geo[1] + (iy - dims[3]) * ele
    out = []
    for (x, y) in ((x0, y0), (x0 + ele, y0 + ele)):
        w = pf.se3_mul(plane, [x, y, 0, 0, 0, 0, 1])[:3]                   # p->_plane * Point3d(x, y, 0)
        lng, lat = lnglat_from_distance(gps_origin[0], gps_origin[1], w[0], w[1])
        out.append("%.9f %.9f %.9f" % (lng, lat, 0.0))
    return "Map2DUpdate LastTexMat " + " ".join(out)
