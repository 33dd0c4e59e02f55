// c_api.cpp -- extern "C" surface of libpifusion.so (include/pifusion.h).
#include "dist.hpp"
#include "jpeg_device.hpp"
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>
#include <mutex>
#include <new>

using pf::FusionMap;

struct pf_map {
    FusionMap impl;
    pf::JpegDevice jpeg;          // pf_feed_jpeg: the decoder's device back end (staging buffers, coefficient and plane buffers)
    pf_map(int t, bool th, const pf_options& o) : impl(t, th, o) {}
    ~pf_map() { if (impl.ok() && impl.use_device()) impl.sync(); }          // the decoder's buffers go before the engine: nothing may still read them
};

// pf_jpeg_decode_device's decoder: process-wide, never torn down (the runtime may be gone at exit)
static std::mutex g_jpeg_mu;
static pf::JpegDevice* shared_jpeg(int device = 0)          // one per device: its buffers live where the first call on that device put them
{
    static pf::JpegDevice* d[64] = {};
    if (device < 0 || device >= 64) device = 0;
    if (!d[device]) d[device] = new pf::JpegDevice();
    return d[device];
}

extern "C" {

void pf_default_options(pf_options* o)
{
    o->band_number = 5; o->force_float = 0; o->high_quality_show = 1; o->weight_type = 0; o->bg_color = 0;
    o->resolution = 0; o->scale = 1; o->device = -1; o->shard_rank = 0; o->shard_count = 1; o->shard_block = 8;
    o->max_queue = 20; o->fused = 1; o->lookahead = 48;
}

int pf_options_set(pf_options* o, const char* key, const char* value)
{
    if (!o || !key || !value) return 0;
    const double v = std::atof(value);
    if (!std::strcmp(key, "MultiBandMap2DCPU.BandNumber")) o->band_number = (int)v;
    else if (!std::strcmp(key, "MultiBandMap2DCPU.ForceFloat")) o->force_float = (int)v;
    else if (!std::strcmp(key, "MultiBandMap2DCPU.HighQualityShow")) o->high_quality_show = (int)v;
    else if (!std::strcmp(key, "Map2D.WeightType")) o->weight_type = (int)v;
    else if (!std::strcmp(key, "Result.BackGroundColor")) o->bg_color = (int)v;
    else if (!std::strcmp(key, "Map2D.Resolution")) o->resolution = v;
    else if (!std::strcmp(key, "Map2D.Scale")) o->scale = v;
    else if (!std::strcmp(key, "Device")) o->device = (int)v;
    else if (!std::strcmp(key, "Shard.Rank")) o->shard_rank = (int)v;
    else if (!std::strcmp(key, "Shard.Count")) o->shard_count = (int)v;
    else if (!std::strcmp(key, "Shard.Block")) o->shard_block = (int)v;
    else if (!std::strcmp(key, "Map2D.MaxQueue")) o->max_queue = (int)v;
    else if (!std::strcmp(key, "Fused")) o->fused = (int)v;
    else if (!std::strcmp(key, "Lookahead")) o->lookahead = (int)v;
    else return 0;
    return 1;
}

pf_map* pf_create(int type, int thread, const pf_options* opt)
{
    if (type == PF_TYPE_NONE || type == PF_TYPE_RENDER) return nullptr;     // Map2D.cpp:53,56
    pf_options o;
    if (opt) o = *opt; else pf_default_options(&o);
    pf_map* m = new (std::nothrow) pf_map(type, thread != 0, o);
    if (m && !m->impl.ok()) { delete m; return nullptr; }
    return m;
}

void pf_destroy(pf_map* m) { delete m; }
const char* pf_last_error(void) { return pf::last_error(); }

int pf_prepare(pf_map* m, const double plane[7], const double cam[6], int n, const pf_image* imgs, const double* poses7)
{ return m && m->impl.prepare(plane, cam, n, imgs, poses7); }

int pf_feed(pf_map* m, const pf_image* img, const double pose[7]) { return m && m->impl.feed(img, pose, false); }
int pf_feed_device(pf_map* m, const pf_image* img, const double pose[7]) { return m && m->impl.feed(img, pose, true); }
int pf_debug_phase_stamps(unsigned long long* out, int cap_blocks) { return pf::read_phase_stamps(out, cap_blocks); }
void pf_debug_form_counts(long long out[8]) { if (out) pf::read_form_counts(out); }
long long pf_debug_compact_launches(void) { return pf::read_compact_launches(); }
int pf_debug_render_log(pf_map* m, long long* out, int cap) { return m ? m->impl.render_log(out, cap) : 0; }
void pf_set_cull(pf_map* m, int on) { if (m) m->impl.set_cull(on != 0); }
long long pf_debug_culled_cells(pf_map* m) { return m ? m->impl.culled_cells() : 0; }
double pf_debug_level0_exact_px(pf_map* m) { return m ? m->impl.level0_exact_px() : 0; }
long long pf_debug_culled_tiles(pf_map* m) { return m ? m->impl.culled_tiles() : 0; }
int pf_debug_select_counts(unsigned long long* out, int reset) { return out ? pf::read_select_counts(out, reset) : 0; }
long pf_debug_read_last_frame(pf_map* m, void* out, size_t cap) { return m ? m->impl.read_back_last_frame(out, cap) : -1; }
unsigned pf_queue_size(pf_map* m) { return m ? m->impl.queue_size() : 0; }
int pf_sync(pf_map* m) { return m && m->impl.sync(); }
int pf_save(pf_map* m, const char* filename) { return m && filename && m->impl.save(filename); }
// pf_write_image / pf_image_info / pf_read_image: image_io.cpp (host code without a HIP dependency: also built by the sanitizer targets)
int pf_jpeg_info(const uint8_t* data, size_t len, int* rows, int* cols, int* components) { return pf::jpeg_info(data, len, rows, cols, components); }
int pf_jpeg_decode_bgr(const uint8_t* data, size_t len, uint8_t* bgr, int rows, int cols)
{ return rows > 0 && cols > 0 && pf::jpeg_decode_bgr(data, len, bgr, rows, cols, (size_t)cols * 3); }
int pf_jpeg_decode_device(const uint8_t* data, size_t len, void* dev_bgr, int rows, int cols, void* hip_stream)
{
    std::lock_guard<std::mutex> l(g_jpeg_mu);
    if (rows <= 0 || cols <= 0 || !dev_bgr) return 0;
    // the decoder of the device the destination lives on, with that device current for the call
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, dev_bgr) != hipSuccess || at.type != hipMemoryTypeDevice) { (void)hipGetLastError(); pf::set_error("pf_jpeg_decode_device: destination is not device memory"); return 0; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (at.device != prev && hipSetDevice(at.device) != hipSuccess) { pf::set_error("pf_jpeg_decode_device: hipSetDevice failed"); return 0; }
    const int ok = shared_jpeg(at.device)->decode_to(data, len, (uint8_t*)dev_bgr, rows, cols, hip_stream);
    if (at.device != prev) (void)hipSetDevice(prev);
    return ok;
}
void pf_debug_jpeg_huffman(pf_map* m, long long out[3])
{
    if (!out) return;
    long a = 0, b = 0; int r = 0;
    if (m) m->jpeg.huffman_counts(&a, &b, &r);
    else { std::lock_guard<std::mutex> l(g_jpeg_mu); shared_jpeg()->huffman_counts(&a, &b, &r); }
    out[0] = a; out[1] = b; out[2] = r;
}
// one staged frame (JpegDevice slot i) into the map: the frame's slot in HBM is filled by the decoder's upload + kernels on the map's stream
static int feed_staged_jpeg(pf_map* m, int i, const double pose[7])
{
    int rows = 0, cols = 0;
    m->jpeg.staged_size(i, &rows, &cols);
    const FusionMap::FrameProducer fill = [m, i](void* dev, hipStream_t st) { return m->jpeg.submit(i, (uint8_t*)dev, (void*)st); };
    pf_image img = { rows, cols, PF_8UC3, nullptr, 0 };
    return m->impl.feed(&img, pose, false, &fill);
}
int pf_feed_jpeg(pf_map* m, const uint8_t* data, size_t len, const double pose[7])
{
    if (!m || !data || !pose) return 0;
    if (!m->impl.ok() || !m->impl.use_device()) { pf::set_error("pf_feed_jpeg: no device"); return 0; }
    unsigned char ok = 0;
    const int i = m->jpeg.stage_one(data, len, &ok);          // markers + Huffman on this thread, outside the map's lock
    if (!ok) { m->jpeg.submit(i, nullptr, nullptr); return 0; }          // leaves the frame's message in pf_last_error()
    return feed_staged_jpeg(m, i, pose);
}
int pf_feed_jpeg_batch(pf_map* m, int n, const uint8_t* const* data, const size_t* len, const double* poses7, int threads, int* results)
{
    if (!m || n < 1 || !data || !len || !poses7) return 0;
    int fed = 0;
    if (results) for (int i = 0; i < n; i++) results[i] = 0;
    if (!m->impl.ok() || !m->impl.use_device()) { pf::set_error("pf_feed_jpeg_batch: no device"); return 0; }
    for (int base = 0; base < n; base += pf::JpegDevice::kSlots) {
        const int cnt = std::min(n - base, (int)pf::JpegDevice::kSlots);
        unsigned char ok[pf::JpegDevice::kSlots];
        if (!m->jpeg.stage_batch(cnt, data + base, len + base, 0, 0, threads, ok)) return fed;
        for (int i = 0; i < cnt; i++) {
            if (!ok[i]) { m->jpeg.submit(i, nullptr, nullptr); continue; }
            const int r = feed_staged_jpeg(m, i, poses7 + 7 * (size_t)(base + i));
            if (results) results[base + i] = r;
            fed += r != 0;
        }
    }
    return fed;
}
int pf_save_to_memory(pf_map* m, uint8_t* bgr, int* rows, int* cols, int* tx0, int* ty0)
{ return m && rows && cols && tx0 && ty0 && m->impl.save_to_memory(bgr, rows, cols, tx0, ty0); }

int pf_num_levels(pf_map* m) { return m ? m->impl.num_levels() : 0; }
int pf_pyramid_type(pf_map* m) { return m ? m->impl.pyramid_type() : 0; }
int pf_grid(pf_map* m, int dims[4], double geo[6]) { return m && m->impl.grid(dims, geo); }
int pf_tile_count(pf_map* m) { return m ? m->impl.tile_count() : 0; }
int pf_tile_coords(pf_map* m, int* xy, int cap) { return m ? m->impl.tile_coords(xy, cap) : 0; }
int pf_get_tile_level(pf_map* m, int ix, int iy, int level, void* lap, float* w) { return m && m->impl.get_tile_level(ix, iy, level, lap, w); }
int pf_get_tile_bgra(pf_map* m, int ix, int iy, uint8_t* bgra) { return m && bgra && m->impl.get_tile_bgra(ix, iy, bgra); }
int pf_blend_tile_raw(pf_map* m, int ix, int iy, void* out) { return m && out && m->impl.blend_tile(ix, iy, out, nullptr, nullptr); }
int pf_blend_tile(pf_map* m, int ix, int iy, uint8_t* bgr) { return m && bgr && m->impl.blend_tile(ix, iy, nullptr, bgr, nullptr); }
int pf_blend_changed(pf_map* m, int* xy, uint8_t* bgr, int cap) { return (m && xy && bgr && cap > 0) ? m->impl.blend_changed(xy, bgr, cap) : 0; }
int pf_blend_tiles(pf_map* m, const int* xy, int n, uint8_t* bgr)
{
    if (!m || !xy || !bgr || n <= 0) return 0;
    std::vector<std::pair<int, int>> tiles(n);
    for (int i = 0; i < n; i++) tiles[i] = { xy[2 * i], xy[2 * i + 1] };
    return m->impl.blend_list(tiles, bgr);
}
void* pf_host_alloc(size_t bytes)
{
    void* p = nullptr;
    // portable: page-locked for every device of the process (a node's maps live on different GPUs)
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void pf_host_free(void* p) { if (p) (void)hipHostFree(p); }

// MultiBandMap2DCPUEle::normalizeUsingWeightMap / mulWeightMap (.cpp:57-75): no caller in
// the reference; host loops kept for API completeness.
int pf_normalize_using_weight_map(const float* weight, float* src3, size_t npix)
{
    if (!weight || !src3) return 0;
    for (size_t i = 0; i < npix; i++) {
        const float d = (float)(weight[i] + 1e-5);
        const float inv = 1.f / d;          // Point3_ operator/ multiplies by the reciprocal (Point.h:213-216)
        src3[3 * i] = inv * src3[3 * i]; src3[3 * i + 1] = inv * src3[3 * i + 1]; src3[3 * i + 2] = inv * src3[3 * i + 2];
    }
    return 1;
}
int pf_mul_weight_map(const float* weight, float* src3, size_t npix)
{
    if (!weight || !src3) return 0;
    for (size_t i = 0; i < npix; i++) { const float w = weight[i]; src3[3 * i] = w * src3[3 * i]; src3[3 * i + 1] = w * src3[3 * i + 1]; src3[3 * i + 2] = w * src3[3 * i + 2]; }
    return 1;
}

static pf::Pose to_pose(const double a[7]) { return pf::pose_from7(a); }
static void from_pose(const pf::Pose& p, double o[7]) { std::memcpy(o, p.t, 24); std::memcpy(o + 3, p.q, 32); }
// pi::calcLngLatFromDistance, PIL/src/hardware/Gps/utils_GPS.cpp:133-160 (same operations in the same order; DEG2RAD is the
// reference's truncated constant)
void pf_lnglat_from_distance(double lng1, double lat1, double dx, double dy, double* lng2, double* lat2)
{
    const double kEarthRadius = 6378137.0, kDeg2Rad = 0.017453292519943;
    const double a = kEarthRadius, f = 1.0 / 298.257223563, e_2 = 2 * f - f * f;
    const double phi_rad = lat1 * kDeg2Rad;
    const double sp = std::sin(phi_rad);
    const double lng_unit = kDeg2Rad * a * std::cos(phi_rad) / std::sqrt(1 - e_2 * (sp * sp));
    const double lat_unit = kDeg2Rad * a * (1 - e_2) / std::pow(1 - e_2 * (sp * sp), 1.5);
    if (lng2) *lng2 = dx / lng_unit + lng1;
    if (lat2) *lat2 = dy / lat_unit + lat1;
}

// MultiBandMap2DCPU.cpp:709-712 + :747-755
int pf_format_map_update(const double plane[7], const double gps_origin[3], double min_x, double min_y, double ele_size,
                         int x, int y, char* out, int cap)
{
    if (!plane || !gps_origin || !out || cap <= 0) return 0;
    const float x0 = (float)(min_x + x * ele_size), y0 = (float)(min_y + y * ele_size);
    const float x1 = (float)(x0 + ele_size), y1 = (float)(y0 + ele_size);
    const pf::Pose pl = to_pose(plane);
    double gps[2][2];
    const float cx[2] = { x0, x1 }, cy[2] = { y0, y1 };
    for (int k = 0; k < 2; k++) {
        const double p[3] = { (double)cx[k], (double)cy[k], 0.0 };
        double r[3];
        pf::rotate(pl.q, p, r);                                   // SE3 * Point3d = translation + rotation * p (SE3.h:99-101)
        pf_lnglat_from_distance(gps_origin[0], gps_origin[1], pl.t[0] + r[0], pl.t[1] + r[1], &gps[k][0], &gps[k][1]);
    }
    // std::to_string(double) is "%f"
    const int n = std::snprintf(out, (size_t)cap, "Map2DUpdate LastTexMat %f %f %f %f %f %f", gps[0][0], gps[0][1], 0.0, gps[1][0], gps[1][1], 0.0);
    return (n > 0 && n < cap) ? n : 0;
}

int pf_map_update_command(pf_map* m, int ix, int iy, const double gps_origin[3], char* out, int cap)
{
    if (!m || !gps_origin || !out) return 0;
    double plane[7], mn[2], ele; int x, y;
    if (!m->impl.map_update_inputs(ix, iy, plane, mn, &ele, &x, &y)) return 0;
    return pf_format_map_update(plane, gps_origin, mn[0], mn[1], ele, x, y, out, cap);
}

void pf_se3_inverse(const double a[7], double out[7]) { from_pose(pf::inverse(to_pose(a)), out); }
void pf_se3_mul(const double a[7], const double b[7], double out[7]) { from_pose(pf::mul(to_pose(a), to_pose(b)), out); }
void pf_so3_rotate(const double q[4], const double p[3], double out[3]) { pf::rotate(q, p, out); }
int pf_footprint(const double cam[6], const double pose_plane[7], double pts8[8])
{
    const pf::Camera c{ cam[0], cam[1], cam[2], cam[3], cam[4], cam[5], 1. / cam[2], 1. / cam[3] };
    return pf::footprint(c, to_pose(pose_plane), pts8) ? 1 : 0;
}
void pf_perspective_transform(const float src8[8], const float dst8[8], double M[9]) { pf::perspective_transform(src8, dst8, M); }

int pf_tile_owner(const pf_options* o, int ix, int iy) { return o ? pf::tile_owner(o->shard_count, o->shard_block, ix, iy) : 0; }
size_t pf_halo_bytes(pf_map* m, int dx, int dy) { return m ? m->impl.halo_bytes_for(dx, dy) : 0; }
int pf_halo_pack(pf_map* m, int ix, int iy, int dx, int dy, void* dev_out) { return m && dev_out && m->impl.halo_pack(ix, iy, dx, dy, dev_out); }
int pf_blend_tile_halo(pf_map* m, int ix, int iy, const void* const dev_halo[9], uint8_t* bgr, void* raw)
{ return m && (bgr || raw) && m->impl.blend_tile(ix, iy, raw, bgr, dev_halo); }
size_t pf_tile_bytes(pf_map* m) { return m ? m->impl.tile_bytes() : 0; }
int pf_tile_export(pf_map* m, int ix, int iy, void* dev_out) { return m && dev_out && m->impl.tile_export(ix, iy, dev_out); }
int pf_tile_import(pf_map* m, int ix, int iy, const void* dev_in) { return m && dev_in && m->impl.tile_import(ix, iy, dev_in); }

// --- seam exchange (dist.cpp)
struct pf_dist { pf::DistMap impl; pf_map* owner; pf_dist(pf_map* m, pf::Transport* t) : impl(&m->impl, t), owner(m) {} };
int pf_dist_unique_id(void* out128) { return out128 && pf::rccl_unique_id(out128); }
pf_dist* pf_dist_init_rccl(pf_map* m, const void* id128, int rank, int nranks)
{
    if (!m || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return nullptr;
    pf::Transport* t = pf::make_rccl_transport(id128, rank, nranks, m->impl.device());
    return t ? new (std::nothrow) pf_dist(m, t) : nullptr;
}
pf_dist* pf_dist_init_host(pf_map* m, int rank, int nranks, pf_exchange_fn fn, void* user)
{
    if (!m || !fn || nranks < 1 || rank < 0 || rank >= nranks) return nullptr;
    pf::Transport* t = pf::make_host_transport(rank, nranks, fn, user);
    return t ? new (std::nothrow) pf_dist(m, t) : nullptr;
}
void pf_dist_destroy(pf_dist* d) { delete d; }
int pf_dist_blend_changed(pf_dist* d, int* xy, uint8_t* bgr, int cap) { return (d && xy && bgr && cap >= 0) ? d->impl.blend_changed(xy, bgr, cap) : -1; }
int pf_dist_feed(pf_dist* d, const pf_image* img, const double pose[7], int root) { return (d && img && pose) ? d->impl.feed(img, pose, root) : -1; }
int pf_dist_feed_jpeg(pf_dist* d, const uint8_t* data, size_t len, int rows, int cols, const double pose[7], int root)
{
    if (!d || !pose) return -1;
    pf_image img = { rows, cols, PF_8UC3, nullptr, 0 };
    if (d->impl.rank() != root) return d->impl.feed(&img, pose, root);
    // the root: markers + (where the stream needs it) Huffman here, the rest queued on the map's stream into the staged slot
    pf_map* m = d->owner;
    unsigned char ok = 0; int i = -1;
    if (data && m->impl.ok() && m->impl.use_device()) i = m->jpeg.stage_one(data, len, &ok);
    const FusionMap::FrameProducer fill = [m, i, ok, rows, cols](void* dev, hipStream_t st) {
        int r = 0, c = 0;
        if (i < 0 || !ok) { if (i >= 0) m->jpeg.submit(i, nullptr, nullptr); else pf::set_error("pf_dist_feed_jpeg: the root has no stream"); return false; }
        m->jpeg.staged_size(i, &r, &c);
        if (r != rows || c != cols) { pf::set_error("pf_dist_feed_jpeg: the stream does not have the size announced to the ranks"); return false; }
        return m->jpeg.submit(i, (uint8_t*)dev, (void*)st);
    };
    return d->impl.feed(&img, pose, root, &fill);
}
int pf_dist_save(pf_dist* d, const char* filename) { return d && filename && d->impl.save(filename); }
int pf_dist_save_to_memory(pf_dist* d, uint8_t* bgr, int* rows, int* cols, int* tx0, int* ty0)
{ return d && rows && cols && tx0 && ty0 && d->impl.save_to_memory(bgr, rows, cols, tx0, ty0); }
int pf_dist_last_stats(pf_dist* d, pf_dist_stats* out) { if (!d || !out) return 0; *out = d->impl.stats(); return 1; }
int pf_dist_plan_blend(int nranks, int me, const int* counts, const int* lists3, const long long* caps, int high_quality,
                       const size_t halo_bytes9[9], pf_strip_plan* send, int send_cap, int* n_send,
                       pf_strip_plan* recv, int recv_cap, int* n_recv, int* mine_xy, int mine_cap, int* n_mine)
{
    if (nranks < 1 || me < 0 || me >= nranks || !counts || !lists3 || !caps || !halo_bytes9 || !n_send || !n_recv || !n_mine) return 0;
    std::vector<std::vector<pf::FusionMap::TileRec>> all(nranks);
    std::vector<long long> cp(caps, caps + nranks);
    const int* q = lists3;
    for (int r = 0; r < nranks; r++)
        for (int k = 0; k < counts[r]; k++, q += 3) all[r].push_back({ q[0], q[1], q[2] });
    pf::BlendPlan plan;
    pf::plan_blend(all, cp, me, high_quality != 0, halo_bytes9, plan);
    int ns = 0, nr = (int)plan.wants.size(), nm = (int)plan.mine.size();
    for (auto& v : plan.send_req) ns += (int)v.size();
    *n_send = ns; *n_recv = nr; *n_mine = nm;
    if (ns > send_cap || nr > recv_cap || nm > mine_cap || (ns && !send) || (nr && !recv) || (nm && !mine_xy)) return 0;
    int i = 0;
    for (int p = 0; p < nranks; p++)
        for (auto& rq : plan.send_req[p]) send[i++] = pf_strip_plan{ p, rq.ix, rq.iy, rq.dx, rq.dy, -1, (unsigned long long)rq.out_off };
    for (int k = 0; k < nr; k++) {
        const auto& w = plan.wants[k];
        recv[k] = pf_strip_plan{ w.peer, plan.mine[w.tile].first, plan.mine[w.tile].second, w.j % 3 - 1, w.j / 3 - 1, w.tile, (unsigned long long)w.off };
    }
    for (int k = 0; k < nm; k++) { mine_xy[2 * k] = plan.mine[k].first; mine_xy[2 * k + 1] = plan.mine[k].second; }
    return 1;
}
int pf_dist_set_verify(pf_dist* d, int on) { if (!d) return 0; d->impl.set_verify(on != 0); return 1; }
int pf_dist_info(pf_dist* d, int* rank, int* nranks, const char** transport)
{
    if (!d) return 0;
    const pf::Transport* t = d->impl.transport();
    if (rank) *rank = t->rank;
    if (nranks) *nranks = t->nranks;
    if (transport) *transport = t->name();
    return 1;
}

int pf_profile_enable(pf_map* m, int mode) { if (!m) return 0; m->impl.profile_enable(mode); return 1; }
int pf_profile_read(pf_map* m, int cap, const char** names, double* total_ms, long long* launches, double* alg_bytes)
{ return m ? m->impl.profile_read(cap, names, total_ms, launches, alg_bytes) : 0; }
int pf_profile_read_run(pf_map* m, int cap, const char** names, double* total_ms, long long* launches, double* alg_bytes, double* alg_bytes_run)
{ return m ? m->impl.profile_read(cap, names, total_ms, launches, alg_bytes, alg_bytes_run) : 0; }
int pf_profile_reset(pf_map* m) { if (!m) return 0; m->impl.profile_reset(); return 1; }
int pf_reserve_tiles(pf_map* m, long long n_tiles) { return m && m->impl.reserve_tiles(n_tiles) ? 1 : 0; }
int pf_render_stats(pf_map* m, double out4[4]) { if (!m || !out4) return 0; m->impl.render_stats(out4); return 1; }
int pf_stats(pf_map* m, long long* rendered, long long* rejected, long long* dropped) { if (!m) return 0; m->impl.stats(rendered, rejected, dropped); return 1; }
int pf_timer_read(pf_map* m, int cap, const char** names, long long* calls, double* mean_s, double* min_s, double* max_s)
{
    return m ? m->impl.timer_read(cap, names, calls, mean_s, min_s, max_s) : 0;
}
int pf_timer_reset(pf_map* m) { if (!m) return 0; m->impl.timer_reset(); return 1; }

}  // extern "C"
