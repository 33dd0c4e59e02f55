// fusion_map.cpp -- host engine: prepare / feed / renderFrame / blend / save on
// the GPU, following the control flow of Map2DFusion/MultiBandMap2DCPU.cpp.
#include "fusion_map.hpp"
#include "env.hpp"
#include "warp_index.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <dlfcn.h>

namespace pf {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; std::fprintf(stderr, "pifusion: %s\n", msg.c_str()); }
const char* last_error() { return g_err.c_str(); }

#define HIP_OK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
            return false;                                                                  \
        }                                                                                  \
    } while (0)

static inline int floordiv(int a, int b) { int q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) q--; return q; }

int tile_owner(int shard_count, int shard_block, int ix, int iy)
{
    if (shard_count <= 1) return 0;
    const int b = shard_block > 0 ? shard_block : 8;
    // PF_SHARD_OWNER_CYCLIC=1 (experiments library; evaluation only, tools/predict_scaling.py --owner cyclic; every rank must set it alike): SURVEY 8e's other
    // candidate, a 2-D block-cyclic owner -- cells dealt over a px x py grid of ranks (px * py = shard_count, px the larger factor), so that
    // neighbouring cells never share a rank and any px x py window of cells holds every rank once
    static const bool cyclic = exp_env_int("PF_SHARD_OWNER_CYCLIC", 0) != 0;     // experiments library only
    if (cyclic) {
        int py = 1;
        for (int d = 1; d * d <= shard_count; d++) if (shard_count % d == 0) py = d;
        const int px = shard_count / py, cxs = floordiv(ix, b), cys = floordiv(iy, b);
        return ((cxs % px) + px) % px + px * (((cys % py) + py) % py);
    }
    const uint32_t cx = (uint32_t)floordiv(ix, b), cy = (uint32_t)floordiv(iy, b);
    const uint32_t h = (cx * 73856093u) ^ (cy * 19349663u);
    return (int)(h % (uint32_t)shard_count);
}

// --------------------------------------------------------------- sections
const char* section_name(int id)
{
    static const char* n[T_COUNT] = { "Map2D::feed", "MultiBandMap2DCPU::renderFrame", "MultiBandMap2DCPU::Apply",
                                      "MultiBandMap2DCPU::spreadMap", "MultiBandMap2DCPU::updateTexture", "MultiBandMap2DCPU::save" };
    return (id >= 0 && id < T_COUNT) ? n[id] : "?";
}

// roctx ranges (libroctx64 is looked up at run time, only when PF_ROCTX is set)
namespace {
struct Roctx { bool tried = false; int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
Roctx g_roctx;
void roctx_load()
{
    g_roctx.tried = true;
    if (!std::getenv("PF_ROCTX")) return;
    void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { std::fprintf(stderr, "pifusion: PF_ROCTX set but no roctx library found\n"); return; }
    *(void**)&g_roctx.push = dlsym(h, "roctxRangePushA");
    *(void**)&g_roctx.pop = dlsym(h, "roctxRangePop");
}
}  // namespace
void roctx_push(const char* name) { if (!g_roctx.tried) roctx_load(); if (g_roctx.push) g_roctx.push(name); }
void roctx_pop() { if (g_roctx.pop) g_roctx.pop(); }

static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

FusionMap::Section::Section(FusionMap* m_, int id_) : m(m_), id(id_), t0(now_s()) { roctx_push(section_name(id)); }
FusionMap::Section::~Section()
{
    roctx_pop();
    const double dt = now_s() - t0;
    std::lock_guard<std::mutex> l(m->timer_mu_);
    SectionRec& r = m->sections_[id];
    if (!r.n_calls || dt < r.min_t) r.min_t = dt;
    if (dt > r.max_t) r.max_t = dt;
    r.total_t += dt; r.n_calls++;
}

int FusionMap::timer_read(int cap, const char** names, long long* calls, double* mean_s, double* min_s, double* max_s)
{
    std::lock_guard<std::mutex> l(timer_mu_);
    int n = 0;
    for (int i = 0; i < T_COUNT && n < cap; i++, n++) {
        const SectionRec& r = sections_[i];
        if (names) names[n] = section_name(i);
        if (calls) calls[n] = r.n_calls;
        if (mean_s) mean_s[n] = r.n_calls ? r.total_t / r.n_calls : 0;
        if (min_s) min_s[n] = r.min_t;
        if (max_s) max_s[n] = r.max_t;
    }
    return n;
}

void FusionMap::timer_reset() { std::lock_guard<std::mutex> l(timer_mu_); for (auto& r : sections_) r = SectionRec(); }

// ------------------------------------------------------------------ store
Tile* TileStore::find(int ix, int iy)
{
    auto it = map_.find(key(ix, iy));
    return it == map_.end() ? nullptr : &it->second;
}

bool TileStore::add_chunk(size_t slots)
{
    void* p = nullptr;
    if (hipMalloc(&p, slots * slot_bytes_) != hipSuccess) { set_error("tile store: hipMalloc failed"); return false; }
    chunks_.push_back({ (char*)p, slots });
    return true;
}

Tile* TileStore::get_or_create(int ix, int iy)
{
    auto it = map_.find(key(ix, iy));
    if (it != map_.end()) return &it->second;
    if (chunks_.empty()) { if (!add_chunk(std::max<size_t>(16, (256u << 20) / slot_bytes_))) return nullptr; cur_ = 0; next_in_chunk_ = 0; }
    if (next_in_chunk_ == chunks_[cur_].slots) {
        // slabs of ~256 MiB: few hipMallocs, tiles of one neighbourhood stay close in HBM
        if (cur_ + 1 == chunks_.size() && !add_chunk(std::max<size_t>(16, (256u << 20) / slot_bytes_))) return nullptr;
        cur_++; next_in_chunk_ = 0;
    }
    Tile t;
    t.base = chunks_[cur_].p + next_in_chunk_++ * slot_bytes_;
    return &map_.emplace(key(ix, iy), t).first->second;
}

// pre-size the store (std::vector::reserve for HBM): one slab for what the slabs at hand cannot hold
bool TileStore::reserve(size_t n_tiles, std::vector<std::pair<char*, size_t>>* fresh)
{
    size_t have = 0;
    for (size_t i = cur_; i < chunks_.size(); i++) have += chunks_[i].slots - (i == cur_ ? next_in_chunk_ : 0);
    if (have >= n_tiles) return true;
    const size_t want = std::max<size_t>(n_tiles - have, 16);
    if (!add_chunk(want)) return false;
    if (fresh) fresh->push_back({ chunks_.back().p, want * slot_bytes_ });
    return true;
}

void TileStore::clear()
{
    for (auto& c : chunks_) (void)hipFree(c.p);
    chunks_.clear(); map_.clear(); next_in_chunk_ = cur_ = 0;
}

bool DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return true;
    release();
    const size_t want = bytes + bytes / 4;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; set_error("hipMalloc failed"); return false; }
    cap = want;
    return true;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }

// ------------------------------------------------------------------ setup
FusionMap::FusionMap(int type, bool thread, const pf_options& opt) : opt_(opt), thread_(thread)
{
    // Map2D::create: TypeCPU -> Map2DCPU; TypeGPU falls back to Map2DCPU in the reference (Map2D.cpp:58-65)
    single_band_ = (type == PF_TYPE_CPU || type == PF_TYPE_GPU);
    const int lim = (int)std::ceil(std::log((double)kElePixels) / std::log(2.0));      // .cpp:263
    band_num_ = std::min(opt_.band_number, lim);
    if (band_num_ < 0) band_num_ = 0;
    if (single_band_) band_num_ = 0;
    lay_ = make_layout(band_num_, opt_.force_float != 0);
    if (single_band_) { lay_.f32 = 0; lay_.lap_off[0] = 0; lay_.w_off[0] = 0; lay_.slot_bytes = kElePixels * kElePixels * 4; }
    store_.configure(lay_.slot_bytes);
    if (opt_.max_queue <= 0) opt_.max_queue = 20;
    table_in_args_ = exp_env("PF_TABLE_COPY") == nullptr;              // experiments library, PF_TABLE_COPY=1: every tile table staged and copied in the stream (A/B, tests)
    if (opt_.shard_count < 1) opt_.shard_count = 1;
    if (opt_.shard_block < 1) opt_.shard_block = 8;
    opt_.lookahead = std::min(std::max(opt_.lookahead, 0), 64);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device: the fusion path has no CPU fallback");
        return;
    }
    if (opt_.device >= 0) device_ = opt_.device;
    else if (hipGetDevice(&device_) != hipSuccess) device_ = 0;
    if (device_ >= ndev) { set_error("device ordinal out of range"); return; }
    if (hipSetDevice(device_) != hipSuccess) { set_error("hipSetDevice failed"); return; }
    if (hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return; }
    for (int i = 0; i < kTableRing; i++)
        if (hipEventCreateWithFlags(&table_ev_[i], hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return; }
    for (int i = 0; i < kMarks; i++)
        if (hipEventCreateWithFlags(&mark_ev_[i], hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return; }
    init_ok_ = true;
    if (thread_) worker_ = std::thread([this] { worker(); });
}

FusionMap::~FusionMap()
{
    if (worker_.joinable()) {
        { std::lock_guard<std::mutex> l(qmu_); stop_ = true; }
        qcv_.notify_all();
        worker_.join();
    }
    if (!init_ok_) return;
    (void)hipSetDevice(device_);
    (void)sync_all();
    prof_harvest();
    for (auto e : ev_pool_) (void)hipEventDestroy(e);
    for (int i = 0; i < kMaxLevels; i++) {
        bool shared = lvl_stream_[i] == stream_;
        for (int k = 1; k < i; k++) shared = shared || lvl_stream_[k] == lvl_stream_[i];
        if (i > 0 && lvl_stream_[i] && !shared) (void)hipStreamDestroy(lvl_stream_[i]);
        for (int k = 0; k < kLvlRing; k++) if (lvl_ev_[i][k]) (void)hipEventDestroy(lvl_ev_[i][k]);
        gw_[i].release(); gw2_[i].release();
    }
    for (auto& s : slots_) { if (s.dev) (void)hipFree(s.dev); if (s.consumed) (void)hipEventDestroy(s.consumed); }
    for (int i = 0; i < kTableRing; i++) {
        if (table_host_[i]) (void)hipHostFree(table_host_[i]);
        table_dev_[i].release();
        if (table_ev_[i]) (void)hipEventDestroy(table_ev_[i]);
        if (i < kMarks && mark_ev_[i]) (void)hipEventDestroy(mark_ev_[i]);
    }
    for (int i = 0; i < kMaxLevels; i++) { g_[i].release(); wgt_[i].release(); blend_lv_[i].release(); }
    for (int i = 0; i < 2; i++) { if (out_pin_[i]) (void)hipHostFree(out_pin_[i]); if (out_copied_[i]) (void)hipEventDestroy(out_copied_[i]); }
    if (out_ready_) (void)hipEventDestroy(out_ready_);
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    blend_src_.release(); blend_out_raw_.release(); blend_out_bgr_.release(); mosaic_table_.release(); w8_.release(); wmap_.release();
    store_.clear();
    (void)hipStreamDestroy(stream_);
}

bool FusionMap::set_device() { HIP_OK(hipSetDevice(device_)); return true; }

// every stream this map launches on: the main stream (level 0, blend, save) and the
// per-level streams of the fused pipeline
hipError_t FusionMap::sync_all()
{
    if (!flush_pipeline()) return hipErrorUnknown;
    hipError_t e = hipStreamSynchronize(stream_);
    if (e == hipSuccess) synced_no_ = work_no_;
    for (int i = 1; i < kMaxLevels; i++)
        if (lvl_stream_[i] && lvl_stream_[i] != stream_) { hipError_t e2 = hipStreamSynchronize(lvl_stream_[i]); if (e == hipSuccess) e = e2; }
    return e;
}

// ---------------------------------------------------------------- profile
void FusionMap::profile_enable(int mode) { std::lock_guard<std::mutex> l(mu_); prof_mode_ = mode; }

bool FusionMap::prof_would(int id) const
{
    const int what = prof_mode_ & 0xff, every = prof_mode_ >> 8;
    return (what == 1 || what == 2 + id) && (every <= 1 || prof_tick_[id] % every == 0);
}

void FusionMap::prof_begin(int id, double bytes, hipStream_t st, double bytes_run)
{
    prof_stream_ = st ? st : stream_;
    const int what = prof_mode_ & 0xff, every = prof_mode_ >> 8;
    prof_on_ = what == 1 || what == 2 + id;                  // mode 2+k: only kernel k
    // event pairs are not free (each is a marker packet between two launches): optionally time every n-th launch only
    if (prof_on_ && every > 1) prof_on_ = (prof_tick_[id]++ % every) == 0;
    if (!prof_on_) return;
    auto get = [&]() { hipEvent_t e; if (!ev_pool_.empty()) { e = ev_pool_.back(); ev_pool_.pop_back(); } else (void)hipEventCreate(&e); return e; };
    prof_cur_ = { id, get(), get(), bytes, bytes_run < 0 ? bytes : bytes_run };
    (void)hipEventRecord(prof_cur_.a, prof_stream_);
}

void FusionMap::prof_end()
{
    if (!prof_on_) return;
    (void)hipEventRecord(prof_cur_.b, prof_stream_);
    prof_pending_.push_back(prof_cur_);
    if (prof_pending_.size() > 4096) { (void)sync_all(); prof_harvest(); }
}

void FusionMap::prof_harvest()
{
    for (auto& r : prof_pending_) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { prof_ms_[r.id] += ms; prof_n_[r.id]++; prof_bytes_[r.id] += r.bytes; prof_bytes_run_[r.id] += r.bytes_run; }
        ev_pool_.push_back(r.a); ev_pool_.push_back(r.b);
    }
    prof_pending_.clear();
}

int FusionMap::profile_read(int cap, const char** names, double* ms, long long* launches, double* bytes, double* bytes_run)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    (void)hipSetDevice(device_);
    (void)sync_all();
    prof_harvest();
    int n = 0;
    for (int i = 0; i < K_COUNT && n < cap; i++, n++) {
        names[n] = kernel_name(i); ms[n] = prof_ms_[i]; launches[n] = prof_n_[i]; bytes[n] = prof_bytes_[i];
        if (bytes_run) bytes_run[n] = prof_bytes_run_[i];
    }
    return n;
}

void FusionMap::profile_reset()
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    (void)hipSetDevice(device_);
    (void)sync_all();
    prof_harvest();
    for (int i = 0; i < K_COUNT; i++) { prof_ms_[i] = 0; prof_n_[i] = 0; prof_bytes_[i] = 0; prof_bytes_run_[i] = 0; prof_tick_[i] = 0; }
}

// Allocator hint (no reference counterpart: the reference's tiles are cv::Mat on the heap): slabs for n more tiles
// are allocated and touched now, so that a mosaic of known extent never meets hipMalloc -- or the driver's
// page clearing behind it -- while keyframes are being fused.
bool FusionMap::reserve_tiles(long long n_tiles)
{
    std::lock_guard<std::mutex> l(mu_);
    if (!init_ok_ || n_tiles <= 0 || !set_device()) return false;
    std::vector<std::pair<char*, size_t>> fresh;
    if (!store_.reserve((size_t)n_tiles, &fresh)) return false;
    for (auto& f : fresh) HIP_OK(hipMemsetAsync(f.first, 0, f.second, stream_));
    HIP_OK(hipStreamSynchronize(stream_));
    return true;
}

void FusionMap::render_stats(double out[4])
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    out[0] = (double)n_with_pixels_; out[1] = px_level0_; out[2] = px_owned_; out[3] = (double)store_.size();
}

void FusionMap::stats(long long* rendered, long long* rejected, long long* dropped)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (rendered) *rendered = n_rendered_;
    if (rejected) *rejected = n_rejected_;
    std::lock_guard<std::mutex> q(qmu_);
    if (dropped) *dropped = n_dropped_;
}

// ---------------------------------------------------------------- prepare
// Map2DPrepare::prepare (Map2D.cpp:32-49) + MultiBandMap2DCPUData::prepare
// (.cpp:199-255) + MultiBandMap2DCPU::prepare (.cpp:266-286)
bool FusionMap::prepare(const double plane7[7], const double cam[6], int n, const pf_image* imgs, const double* poses7)
{
    if (!init_ok_) { set_error("prepare: no device"); return false; }
    if (n == 0 || !poses7 || cam[0] <= 0 || cam[1] <= 0 || cam[2] == 0 || cam[3] == 0) {
        std::fprintf(stderr, "Map2D::prepare:Not valid prepare!\n");
        return false;
    }
    Camera c{ cam[0], cam[1], cam[2], cam[3], cam[4], cam[5], 1. / cam[2], 1. / cam[3] };
    const Pose plane = pose_from7(plane7), pinv = inverse(plane);
    double mx[3] = { -1e10, -1e10, -1e10 }, mn[3] = { 1e10, 1e10, 1e10 };
    std::vector<Pose> local(n);
    for (int i = 0; i < n; i++) {
        local[i] = mul(pinv, pose_from7(poses7 + 7 * i));
        for (int k = 0; k < 3; k++) {
            mx[k] = local[i].t[k] > mx[k] ? local[i].t[k] : mx[k];
            mn[k] = local[i].t[k] < mn[k] ? local[i].t[k] : mn[k];
        }
    }
    if (mn[2] * mx[2] <= 0) return false;
    const double maxh = mx[2] > 0 ? mx[2] : -mn[2];
    const double lx = (c.w - c.cx) * c.fxinv - (0 - c.cx) * c.fxinv;
    const double ly = (c.h - c.cy) * c.fyinv - (0 - c.cy) * c.fyinv;
    const double radius = 0.5 * maxh * std::sqrt((lx * lx + ly * ly));
    double length_pixel = opt_.resolution;
    if (!length_pixel) {
        length_pixel = 2 * radius / std::sqrt(c.w * c.w + c.h * c.h);
        length_pixel /= opt_.scale;
    }
    std::printf("Map2D.Resolution=%g\n", length_pixel);
    mn[0] = mn[0] - radius; mn[1] = mn[1] - radius;
    mx[0] = mx[0] + radius; mx[1] = mx[1] + radius;
    double ctr[3];
    for (int k = 0; k < 3; k++) ctr[k] = 0.5 * (mn[k] + mx[k]);
    for (int k = 0; k < 3; k++) { mn[k] = 2 * mn[k] - ctr[k]; mx[k] = 2 * mx[k] - ctr[k]; }
    const double ele_size = kElePixels * length_pixel;
    const int w = (int)std::ceil((mx[0] - mn[0]) / ele_size);
    const int h = (int)std::ceil((mx[1] - mn[1]) / ele_size);
    mx[0] = mn[0] + ele_size * w;
    mx[1] = mn[1] + ele_size * h;

    {
        // drop frames queued against the old preparation (.cpp:362-376 bails on p!=prepared)
        std::unique_lock<std::mutex> q(qmu_);
        for (auto& f : queue_) if (f.slot >= 0) slots_[f.slot].queued = false;
        queue_.clear();
        idle_cv_.wait(q, [this] { return !worker_busy_; });
    }
    std::lock_guard<std::mutex> l(mu_);
    if (!set_device()) return false;
    if (valid_) (void)drain();                 // keyframes fed against the old preparation are rendered into it, as they would have been inside their feed calls
    for (auto& p : pending_) release_slot(p.f);
    pending_.clear();
    HIP_OK(sync_all());
    store_.clear();
    plane_ = plane; plane_inv_ = pinv; cam_ = c;
    length_pixel_ = length_pixel; length_pixel_inv_ = 1. / length_pixel;
    ele_size_ = ele_size; ele_size_inv_ = 1. / ele_size;
    std::memcpy(min_, mn, sizeof(mn)); std::memcpy(max_, mx, sizeof(mx));
    w_ = w; h_ = h; off_x_ = off_y_ = 0;
    valid_ = true;
    if (thread_ && imgs) {
        // with a render thread the prepare frames are rendered first (Map2D.cpp:42, .cpp:606-615)
        for (int i = 0; i < n; i++) {
            if (!imgs[i].data) continue;
            const int cn = imgs[i].type == PF_8UC4 ? 4 : 3;
            const size_t row_bytes = (size_t)imgs[i].cols * cn, step = imgs[i].step ? imgs[i].step : row_bytes;
            if (step < row_bytes || step * (size_t)imgs[i].rows >= (1ull << 31)) { set_error("prepare: bad row step or frame of 2 GiB or more"); return false; }
            const int slot = acquire_slot((size_t)imgs[i].rows * step);
            if (slot < 0 || !upload(&imgs[i], slot)) return false;
            std::lock_guard<std::mutex> q(qmu_);
            slots_[slot].queued = true;
            queue_.push_back({ slot, nullptr, (long)step, imgs[i].rows, imgs[i].cols, cn, local[i] });
        }
        qcv_.notify_all();
    }
    return true;
}

// ------------------------------------------------------------------- feed
int FusionMap::acquire_slot(size_t bytes)
{
    // called with mu_ held.  A slot is reusable once it left the queue and the
    // kernels that read it have completed.
    const size_t limit = (size_t)opt_.max_queue + 4 + (size_t)opt_.lookahead;          // the feed queue, frames in flight, keyframes that wait for their lookahead
    for (int pass = 0; pass < 2; pass++) {
        int oldest_pending = -1;
        for (size_t i = 0; i < slots_.size(); i++) {
            FrameSlot& s = slots_[i];
            { std::lock_guard<std::mutex> q(qmu_); if (s.queued) continue; }
            if (s.pending) {
                if (hipEventQuery(s.consumed) == hipSuccess) s.pending = false;
                else { if (oldest_pending < 0) oldest_pending = (int)i; continue; }
            }
            if (s.cap < bytes) {
                if (s.dev) (void)hipFree(s.dev);
                s.dev = nullptr; s.cap = 0;
                if (hipMalloc((void**)&s.dev, bytes + 64) != hipSuccess) { set_error("frame slot hipMalloc failed"); return -1; }
                s.cap = bytes;
            }
            return (int)i;
        }
        if (slots_.size() < limit) {
            FrameSlot s;
            if (hipMalloc((void**)&s.dev, bytes + 64) != hipSuccess) { set_error("frame slot hipMalloc failed"); return -1; }
            s.cap = bytes;
            if (hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return -1; }
            slots_.push_back(s);
            return (int)slots_.size() - 1;
        }
        if (oldest_pending >= 0) { (void)hipEventSynchronize(slots_[oldest_pending].consumed); slots_[oldest_pending].pending = false; }
    }
    set_error("no free frame slot");
    return -1;
}

bool FusionMap::upload(const pf_image* img, int slot)
{
    // One linear, blocking copy of the rows as they lie in the caller's buffer (cv::Mat::step is kept: the kernels
    // take any row step): the caller may release its pixels when feed() returns, and the frame is complete in HBM
    // before the kernel that reads it is enqueued.  The map's streams are non-blocking, so this does not wait for
    // kernels in flight.
    const size_t row = (size_t)img->cols * (img->type == PF_8UC4 ? 4 : 3), step = img->step ? img->step : row;
    const size_t bytes = (size_t)(img->rows - 1) * step + row;          // == frame_bytes(): what the kernels may read
    HIP_OK(hipMemcpy(slots_[slot].dev, img->data, bytes, hipMemcpyHostToDevice));
    last_slot_ = slot; last_bytes_ = bytes;
    return true;
}

// MultiBandMap2DCPU::feed (.cpp:288-309)
bool FusionMap::feed(const pf_image* img, const double pose7[7], bool device_ptr, const FrameProducer* produce)
{
    if (!init_ok_) { set_error("feed: no device"); return false; }
    Section sec(this, T_FEED);
    QueuedFrame f{};
    { std::lock_guard<std::mutex> q(qmu_); f.seq = feed_seq_++; }
    {
        std::lock_guard<std::mutex> l(mu_);
        if (!valid_) return false;
        if (!set_device()) return false;
        f.pose = mul(plane_inv_, pose_from7(pose7));
        f.slot = -1; f.ext = nullptr;
        if (img) {
            // wrong size/type is reported by renderFrame (.cpp:319-323); keep that order of checks
            f.rows = img->rows; f.cols = img->cols;
            f.cn = img->type == PF_8UC4 ? 4 : 3;     // BGRA frames are accepted as the tracker produces them (row f1)
            if ((img->type != PF_8UC3 && img->type != PF_8UC4) || img->cols != cam_.w || img->rows != cam_.h) {
                std::fprintf(stderr, "MultiBandMap2DCPU::renderFrame: frame.first.cols!=p->_camera.w||frame.first.rows!=p->_camera.h||frame.first.type()!=CV_8UC3\n");
                if (!thread_) { n_rejected_++; return false; }
                return true;    // the threaded reference enqueues and fails later on the render thread
            }
            if (img->data) {
                const size_t row_bytes = (size_t)img->cols * f.cn;
                if ((img->step && img->step < row_bytes) || (img->step ? img->step : row_bytes) * (size_t)img->rows >= (1ull << 31)) {
                    set_error("feed: row step smaller than a row, or frame of 2 GiB or more");
                    return false;
                }
                if (frame_bytes(img->rows, img->cols, (long)(img->step ? img->step : row_bytes), f.cn) < 8) {
                    set_error("feed: frame smaller than 8 bytes");      // the warp loads 8 bytes per source row (warp_index.hpp)
                    return false;
                }
                if (device_ptr) {
                    if (thread_) { set_error("pf_feed_device needs a thread=0 map"); return false; }
                    f.ext = (const uint8_t*)img->data; f.step = img->step ? (long)img->step : (long)img->cols * f.cn;
                } else {
                    f.step = img->step ? (long)img->step : (long)img->cols * f.cn;
                    f.slot = acquire_slot((size_t)img->rows * (size_t)f.step);
                    if (f.slot < 0 || !upload(img, f.slot)) return false;
                }
            } else if (produce) {
                f.step = (long)img->cols * f.cn;
                if (frame_bytes(img->rows, img->cols, f.step, f.cn) < 8) { set_error("feed: frame smaller than 8 bytes"); return false; }
                f.slot = acquire_slot((size_t)img->rows * (size_t)f.step);
                if (f.slot < 0 || !(*produce)(slots_[f.slot].dev, stream_)) return false;
                // the producer's work sits on stream_ ahead of the launch that reads the slot; should the frame be dropped from the
                // queue instead, the slot's next user (possibly a blocking upload) has to wait for that work all the same
                HIP_OK(hipEventRecord(slots_[f.slot].consumed, stream_));
                slots_[f.slot].pending = true;
            }
        }
    }
    if (thread_) {
        std::lock_guard<std::mutex> q(qmu_);
        if (f.slot >= 0) slots_[f.slot].queued = true;
        queue_.push_back(f);
        if ((int)queue_.size() > opt_.max_queue) {          // .cpp:302 drop-oldest
            if (queue_.front().slot >= 0) slots_[queue_.front().slot].queued = false;
            queue_.pop_front();
            n_dropped_++;
        }
        qcv_.notify_one();
        return true;
    }
    std::lock_guard<std::mutex> l(mu_);
    return render_frame(f);
}

// ---- frame distribution (pf_dist_feed)
// The ranks that own at least one tile of the canvas this keyframe renders into.  Same geometry as render_frame
// (.cpp:324-394); spreadMap is applied here already -- it is idempotent for the render that follows.
bool FusionMap::frame_needs(const double pose7[7], std::vector<unsigned char>& rank_needs)
{
    std::lock_guard<std::mutex> l(mu_);
    rank_needs.assign((size_t)opt_.shard_count, 0);
    if (!valid_) return false;
    const Pose pose = mul(plane_inv_, pose_from7(pose7));
    double pts[8];
    if (!footprint(cam_, pose, pts)) return true;            // oblique view: nobody renders it (feed will say false)
    double xmin = pts[0], xmax = xmin, ymin = pts[1], ymax = ymin;
    for (int i = 1; i < 4; i++) {
        xmin = std::min(xmin, pts[2 * i]); xmax = std::max(xmax, pts[2 * i]);
        ymin = std::min(ymin, pts[2 * i + 1]); ymax = std::max(ymax, pts[2 * i + 1]);
    }
    if (xmin < min_[0] || xmax > max_[0] || ymin < min_[1] || ymax > max_[1])
        if (!spread_map(xmin, ymin, xmax, ymax)) return false;
    const int x0 = (int)std::floor((xmin - min_[0]) * ele_size_inv_), y0 = (int)std::floor((ymin - min_[1]) * ele_size_inv_);
    const int x1 = (int)std::ceil((xmax - min_[0]) * ele_size_inv_), y1 = (int)std::ceil((ymax - min_[1]) * ele_size_inv_);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++)
            rank_needs[(size_t)tile_owner(opt_.shard_count, opt_.shard_block, x + off_x_, y + off_y_)] = 1;
    return true;
}

// renderFrame's size / type gate (.cpp:319-323) for a frame that arrives through pf_dist_feed: true if the frame was rejected --
// message and counter exactly as feed() has them, and nothing else touched (the reference returns before any geometry)
bool FusionMap::reject_mismatched_frame(const pf_image* desc)
{
    std::lock_guard<std::mutex> l(mu_);
    if (!valid_) return false;
    if ((desc->type != PF_8UC3 && desc->type != PF_8UC4) || desc->cols != cam_.w || desc->rows != cam_.h) {
        std::fprintf(stderr, "MultiBandMap2DCPU::renderFrame: frame.first.cols!=p->_camera.w||frame.first.rows!=p->_camera.h||frame.first.type()!=CV_8UC3\n");
        n_rejected_++;
        return true;
    }
    return false;
}

int FusionMap::stage_frame(const pf_image* desc, bool upload_host, void** dev, size_t* bytes)
{
    std::lock_guard<std::mutex> l(mu_);
    if (!init_ok_ || !valid_ || !set_device() || thread_) { set_error("stage_frame: needs a prepared thread=0 map"); return -1; }
    const int cn = desc->type == PF_8UC4 ? 4 : 3;
    const size_t row = (size_t)desc->cols * cn, step = desc->step ? desc->step : row;
    if ((desc->type != PF_8UC3 && desc->type != PF_8UC4) || desc->cols != cam_.w || desc->rows != cam_.h || step < row ||
        step * (size_t)desc->rows >= (1ull << 31)) { set_error("stage_frame: frame does not match the camera"); return -1; }
    const int slot = acquire_slot((size_t)desc->rows * step);
    if (slot < 0) return -1;
    if (upload_host) {
        if (!desc->data || !upload(desc, slot)) return -1;
    }
    slots_[slot].queued = true;                                // held until feed_staged
    if (dev) *dev = slots_[slot].dev;
    if (bytes) *bytes = (size_t)(desc->rows - 1) * step + row;
    return slot;
}

bool FusionMap::feed_staged(int slot, const pf_image* desc, const double pose7[7])
{
    if (!init_ok_) return false;
    Section sec(this, T_FEED);
    std::lock_guard<std::mutex> l(mu_);
    if (slot >= 0) { std::lock_guard<std::mutex> q(qmu_); slots_[slot].queued = false; }
    if (!valid_ || !set_device()) return false;
    QueuedFrame f{};
    f.pose = mul(plane_inv_, pose_from7(pose7));
    f.slot = slot; f.ext = nullptr;
    f.rows = desc->rows; f.cols = desc->cols; f.cn = desc->type == PF_8UC4 ? 4 : 3;
    f.step = desc->step ? (long)desc->step : (long)desc->cols * f.cn;
    return render_frame(f);
}

// test hook: the bytes of the most recently uploaded host frame as they lie in HBM
long FusionMap::read_back_last_frame(void* out, size_t cap)
{
    std::lock_guard<std::mutex> l(mu_);
    if (!init_ok_ || last_slot_ < 0 || !set_device()) return -1;
    if (out && cap >= last_bytes_ && hipMemcpy(out, slots_[last_slot_].dev, last_bytes_, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long)last_bytes_;
}

unsigned FusionMap::queue_size() { std::lock_guard<std::mutex> q(qmu_); return (unsigned)queue_.size(); }

// MultiBandMap2DCPU::run (.cpp:619-635) without the 10 ms sleep
void FusionMap::worker()
{
    for (;;) {
        QueuedFrame f;
        {
            std::unique_lock<std::mutex> q(qmu_);
            qcv_.wait(q, [this] { return stop_ || !queue_.empty(); });
            if (stop_) return;
            f = queue_.front(); queue_.pop_front();
            worker_busy_ = true;
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            if (valid_ && set_device()) render_frame(f);          // in; rendered once opt_.lookahead more keyframes are, or ...
            else release_slot(f);
        }
        bool idle;
        { std::lock_guard<std::mutex> q(qmu_); idle = queue_.empty(); }
        if (idle) {                                               // ... now, when nothing else waits in the feed queue: no keyframe is held back for company that may never come
            std::lock_guard<std::mutex> l(mu_);
            (void)drain();
        }
        {
            std::lock_guard<std::mutex> q(qmu_);
            worker_busy_ = false;
        }
        idle_cv_.notify_all();
    }
}

bool FusionMap::sync()
{
    if (!init_ok_) return false;
    if (thread_) {
        std::unique_lock<std::mutex> q(qmu_);
        idle_cv_.wait(q, [this] { return queue_.empty() && !worker_busy_; });
    }
    std::lock_guard<std::mutex> l(mu_);
    if (!set_device() || !drain()) return false;
    HIP_OK(sync_all());
    prof_harvest();
    return true;
}

// spreadMap (.cpp:561-604): geometry only -- tiles live in a hash keyed by
// stable coordinates, so nothing is re-laid out.
bool FusionMap::spread_map(double xmin, double ymin, double xmax, double ymax)
{
    Section sec(this, T_SPREAD);
    int xminInt = (int)std::floor((xmin - min_[0]) * ele_size_inv_);
    int yminInt = (int)std::floor((ymin - min_[1]) * ele_size_inv_);
    int xmaxInt = (int)std::ceil((xmax - min_[0]) * ele_size_inv_);
    int ymaxInt = (int)std::ceil((ymax - min_[1]) * ele_size_inv_);
    xminInt = std::min(xminInt, 0); yminInt = std::min(yminInt, 0);
    xmaxInt = std::max(xmaxInt, w_); ymaxInt = std::max(ymaxInt, h_);
    const int w = xmaxInt - xminInt, h = ymaxInt - yminInt;
    const double mnx = min_[0] + ele_size_ * xminInt, mny = min_[1] + ele_size_ * yminInt;
    const double mxx = mnx + w * ele_size_, mxy = mny + h * ele_size_;
    min_[0] = mnx; min_[1] = mny; max_[0] = mxx; max_[1] = mxy;
    w_ = w; h_ = h; off_x_ += xminInt; off_y_ += yminInt;
    return true;
}

// ------------------------------------------------------------ renderFrame
// MultiBandMap2DCPU::renderFrame (.cpp:311-558) in stages; FrameWork (fusion_map.hpp) is what one stage leaves for the next:
//   frame_canvas        1-3  footprint, tile range, homography                         (.cpp:324-441)
//   build_tile_table         Apply's tile loop: owned tiles, the cull, the table entries (.cpp:476-492)
//   level_windows / plan_fused_levels   where each pyramid level is needed: windows, compute regions, need bitmaps and rectangles
//   reserve_frame_workspace / place_table   grow-only buffers, the table's ring slot
//   launch_fused_pipeline | launch_level_streams | launch_per_op | launch_single_band   the kernels (.cpp:443-555)
//   retire_frame             flags, weight bounds, counters
namespace {
inline void clampw(int lo, int hi, int n, int& o0, int& o1) { o0 = std::max(lo, 0); o1 = std::min(hi, n); }
// a box of level-0 pixels (multiples of 64) at level i: floor / ceil (at the top levels a cell is less than a pixel)
inline int lv_lo(int p, int i) { return p >> i; }
inline int lv_hi(int p, int i) { return (p + (1 << i) - 1) >> i; }
}

bool FusionMap::render_frame(const QueuedFrame& f)
{
    Section sec(this, T_RENDER);
    FrameWork& w = fw_;
    w.reset();
    // Geometry now, in feed order: the grid advances (spreadMap) and a keyframe is rejected inside its own feed call whether or not its
    // pixels wait for the lookahead.
    const int go = frame_canvas(f, w);
    if (go <= 0) { release_slot(f); if (go < 0) n_rejected_++; return go == 0; }          // -1: rejected (.cpp:340-343, :381-386); 0: geometry-only frame
    pending_.emplace_back();
    PendingFrame& p = pending_.back();
    p.f = f;
    std::memcpy(p.pts, w.pts, sizeof(p.pts));
    p.sx0 = w.xminInt + off_x_; p.sy0 = w.yminInt + off_y_; p.tx = w.tx; p.ty = w.ty;
    std::memcpy(p.M0, w.M0, sizeof(p.M0));
    // may this keyframe's cells be culled?  (the lattice of its canvas mapped into the source, if so)
    w.cull = cull_on_ && (single_band_ || (opt_.fused == 1 && w.L >= 1)) && invert3x3(w.M0, w.Minv) && cull_frame_ok(w.Minv, w.crows, w.ccols);
    p.cull = w.cull;
    std::memcpy(p.Minv, w.Minv, sizeof(p.Minv));
    if (f.slot >= 0) { std::lock_guard<std::mutex> q(qmu_); slots_[f.slot].queued = true; }      // held until the keyframe is rendered (retire_frame)
    size_t keep = 0;
    if (w.cull) {
        if (lat_.sx.capacity() == 0 && !lat_pool_.empty()) { lat_ = std::move(lat_pool_.back()); lat_pool_.pop_back(); }
        // a shard that owns a quarter of the canvas' tiles or more has the whole lattice mapped at once, like an unsharded map (one tile_owner
        // call per tile: ~1 us); below that the points are mapped as its tiles ask for them
        bool map_all = opt_.shard_count <= 1;
        if (!map_all) {
            int mine = 0;
            for (int y = 0; y < w.ty; y++)
                for (int x = 0; x < w.tx; x++) mine += tile_owner(opt_.shard_count, opt_.shard_block, w.xminInt + x + off_x_, w.yminInt + y + off_y_) == opt_.shard_rank;
            map_all = 4 * mine >= w.tx * w.ty;
        }
        cull_lattice(w.Minv, w.crows, w.ccols, f.cols, f.rows, single_band_ ? 0 : ((2 << w.L) - 2 + 63) / 64, map_all);
        if (lookahead_ok()) {
            // the keyframe's own lower bounds enter the tiles' wlb NOW: the keyframes ahead of it in the queue are decided against them
            if (!tiles_pool_.empty()) { p.tiles = std::move(tiles_pool_.back()); tiles_pool_.pop_back(); }
            pre_raise(w, p.tiles);
            p.pre_raised = true;
            since_drain_++;
            keep = std::min<size_t>((size_t)opt_.lookahead, 2 * (size_t)since_drain_ / 3);
        }
        std::swap(lat_, p.lat);
    }
    // a keyframe that cannot take part (no cull for it: an untame homography, a per-level or per-op map, the cull switched off) is rendered at
    // once, and everything that waits before it; a failure of one of THOSE renders is reported by this call
    while (pending_.size() > keep)
        if (!render_front()) return false;
    return true;
}

// The weight bounds of an admitted keyframe go into its tiles' wlb (which only ever rises; build_tile_table does not work them out again).
// Creates the canvas' tiles, as Apply's tile loop does (.cpp:478-492), and remembers them for the render.  A keyframe never culls
// itself by this: the largest weight it can have in a cell is not below the smallest (cell_out's margins only widen the gap).
void FusionMap::pre_raise(FrameWork& w, std::vector<Tile*>& tiles)
{
    tiles.assign((size_t)w.tx * w.ty, nullptr);

    const int S = cull_sub_, span = 4 / S;
    const bool sharded = opt_.shard_count > 1;
    if (lat_.all && S == 4) {
        // Every lattice point is mapped (cull_lattice), every cell of an owned tile is asked.  The same arithmetic as cell_out's first half, without
        // its calls: the farthest corner of a cell's dilated square from a pass that pairs the points e steps apart along a row first.
        const int nx = lat_.nx, ny = lat_.ny, e = 1 + 2 * lat_.dil;
        pair_d_.resize((size_t)nx * ny); pair_in_.resize((size_t)nx * ny);
        for (int m = 0; m < ny; m++) {
            const double* __restrict__ d = lat_.d.data() + (size_t)m * nx; const unsigned char* __restrict__ in = lat_.in.data() + (size_t)m * nx;
            double* __restrict__ pd = pair_d_.data() + (size_t)m * nx; unsigned char* __restrict__ pi = pair_in_.data() + (size_t)m * nx;
            for (int k = 0; k + e < nx; k++) { pd[k] = std::max(d[k], d[k + e]); pi[k] = in[k] & in[k + e]; }
        }
        const double mpx = cull_margin_px_, mw = cull_margin_w_;
        const int wt = opt_.weight_type;
        for (int y = 0; y < w.ty; y++)
            for (int x = 0; x < w.tx; x++) {
                const int sx = w.xminInt + x + off_x_, sy = w.yminInt + y + off_y_;
                if (sharded && tile_owner(opt_.shard_count, opt_.shard_block, sx, sy) != opt_.shard_rank) continue;
                Tile* t = store_.get_or_create(sx, sy);
                if (!t) { tiles.clear(); return; }
                tiles[(size_t)y * w.tx + x] = t;
                for (int qy = 0; qy < 4; qy++) {
                    const size_t r0 = (size_t)(4 * y + qy) * nx + 4 * x, r1 = r0 + (size_t)e * nx;
                    for (int qx = 0; qx < 4; qx++) {
                        if (!(pair_in_[r0 + qx] & pair_in_[r1 + qx])) continue;              // not wholly inside the frame: wmin 0
                        float& wl = t->wlb[4 * qy + qx];
                        const double far2 = std::max(pair_d_[r0 + qx], pair_d_[r1 + qx]);
                        const double tw = wt == 0 ? (1.0 - mw - (double)wl) * lat_.dis_max - mpx : 1e300;
                        if (!(tw > 0 && far2 < tw * tw * (1.0 + 1e-9))) continue;
                        const double dfar = std::sqrt(far2) + mpx;
                        double ww = 1.0 - dfar * lat_.inv_dis_max;
                        if (wt != 0) ww = ww > 0 ? ww * ww : 0.0;
                        ww -= mw;
                        if (ww > 2e-5 && (float)ww > wl) wl = (float)ww;
                    }
                }
            }
        return;
    }
    for (int y = 0; y < w.ty; y++)
        for (int x = 0; x < w.tx; x++) {
            const int sx = w.xminInt + x + off_x_, sy = w.yminInt + y + off_y_;
            if (sharded && tile_owner(opt_.shard_count, opt_.shard_block, sx, sy) != opt_.shard_rank) continue;
            Tile* t = store_.get_or_create(sx, sy);
            if (!t) { tiles.clear(); return; }                // HBM exhausted: build_tile_table meets it again and reports it
            tiles[(size_t)y * w.tx + x] = t;                  // (references into the store stay valid while it grows)
            for (int q = 0; q < S * S; q++) {
                float wmin;
                (void)cell_out(4 * x + span * (q % S), 4 * y + span * (q / S), span, opt_.weight_type, t->wlb[q], false, &wmin);
                if (wmin > t->wlb[q]) t->wlb[q] = wmin;
            }
        }
}

void FusionMap::release_slot(const QueuedFrame& f)
{
    if (f.slot < 0) return;
    std::lock_guard<std::mutex> q(qmu_);
    slots_[f.slot].queued = false;
}

bool FusionMap::drain()
{
    since_drain_ = 0;
    if (pending_.empty()) return true;
    if (!set_device()) return false;
    while (!pending_.empty())
        if (!render_front()) return false;
    return true;
}

// stages 4 onwards of renderFrame for the oldest keyframe that waits
bool FusionMap::render_front()
{
    struct Pop {            // the keyframe leaves the queue whatever happens to it (its lattice stays in lat_: the next admission writes over it)
        FusionMap* m;
        ~Pop() {
            PendingFrame& p = m->pending_.front();
            if (p.lat.sx.capacity() && m->lat_pool_.size() < 8) m->lat_pool_.push_back(std::move(p.lat));
            if (p.tiles.capacity() && m->tiles_pool_.size() < 8) m->tiles_pool_.push_back(std::move(p.tiles));
            m->pending_.pop_front();
        }
    } pop{ this };
    PendingFrame& p = pending_.front();
    const QueuedFrame f = p.f;
    struct Held { FusionMap* m; const QueuedFrame& f; bool done = false; ~Held() { if (!done) m->release_slot(f); } } held{ this, f };
    FrameWork& w = fw_;
    w.reset();
    std::memcpy(w.pts, p.pts, sizeof(p.pts));
    w.xminInt = p.sx0 - off_x_; w.yminInt = p.sy0 - off_y_; w.xmaxInt = w.xminInt + p.tx; w.ymaxInt = w.yminInt + p.ty;
    w.tx = p.tx; w.ty = p.ty; w.L = band_num_; w.crows = w.ty * kElePixels; w.ccols = w.tx * kElePixels;
    std::memcpy(w.M0, p.M0, sizeof(p.M0)); std::memcpy(w.Minv, p.Minv, sizeof(p.Minv));
    w.cull = p.cull;
    w.pre_raised = p.pre_raised;
    w.tiles_known = p.tiles.size() == (size_t)p.tx * p.ty ? p.tiles.data() : nullptr;
    w.src = f.ext ? f.ext : slots_[f.slot].dev;
    if (p.cull) std::swap(lat_, p.lat);
    Section sec_apply(this, T_APPLY);          // the reference times its tile loop under this name (.cpp:476-555); here: table, need rectangles, launch
    if (!build_tile_table(f, w)) return false;
    if (w.bx0 >= w.bx1) {                      // nothing of this frame lands on this shard, or it cannot win anywhere it lands
        for (auto& r : w.raise) r.t->wlb[r.q] = std::max(r.t->wlb[r.q], r.w);
        for (Tile* t : w.culled) t->changed = true;
        if (w.owned_all) n_rendered_++;
        log_rendered(f);
        return true;
    }
    px_owned_ += (double)w.owned * kElePixels * kElePixels;     // what this rank renders beyond its share: owned tile pixels vs the level-0 window (bench --shard strong)
    n_with_pixels_++;
    level_windows(w);
    if (!reserve_frame_workspace(w) || !place_table(w)) return false;
    warp_args(f, w);
    const bool fused = opt_.fused != 0 && w.L >= 1;
    bool ok;
    if (single_band_) ok = launch_single_band(f, w);
    else if (fused) {
        // weightImage (MultiBandMap2DCPU.cpp:396-425): built once per frame size, gathered by the warp
        if (wmap_rows_ != f.rows || wmap_cols_ != f.cols) {
            HIP_OK(sync_all());
            if (!wmap_.reserve((size_t)f.rows * f.cols * 4)) return false;
            launch_weight32(stream_, (float*)wmap_.p, f.rows, f.cols, opt_.weight_type);
            wmap_rows_ = f.rows; wmap_cols_ = f.cols;
        }
        w.a.wmap = (const float*)wmap_.p;
        plan_fused_levels(w);
        ok = opt_.fused == 1 ? launch_fused_pipeline(f, w) : launch_level_streams(f, w);
    } else ok = launch_per_op(f, w);
    if (!ok) return false;
    HIP_OK(hipGetLastError());
    held.done = true;                          // retire_frame hands the slot over to its consumed event
    return retire_frame(f, w, fused);
}

void FusionMap::log_rendered(const QueuedFrame& f)
{
    if (f.seq < 0) return;
    if (render_log_.size() >= 65536) render_log_.erase(render_log_.begin(), render_log_.begin() + 32768);
    render_log_.push_back(f.seq);
}

// 1. pose -> ground points (.cpp:324-347); 2. destination tile range, spreadMap when the footprint leaves the grid (.cpp:349-394);
// 3. homography (.cpp:427-441).  Returns 1 to go on, 0 for a geometry-only frame (other shards own its tiles), -1 rejected.
int FusionMap::frame_canvas(const QueuedFrame& f, FrameWork& w)
{
    double* pts = w.pts;
    if (!footprint(cam_, f.pose, pts)) return -1;
    double xmin = pts[0], xmax = xmin, ymin = pts[1], ymax = ymin;
    for (int i = 1; i < 4; i++) {
        if (pts[2 * i] < xmin) xmin = pts[2 * i];
        if (pts[2 * i + 1] < ymin) ymin = pts[2 * i + 1];
        if (pts[2 * i] > xmax) xmax = pts[2 * i];
        if (pts[2 * i + 1] > ymax) ymax = pts[2 * i + 1];
    }
    if (xmin < min_[0] || xmax > max_[0] || ymin < min_[1] || ymax > max_[1])
        if (!spread_map(xmin, ymin, xmax, ymax)) return -1;
    w.xminInt = (int)std::floor((xmin - min_[0]) * ele_size_inv_);
    w.yminInt = (int)std::floor((ymin - min_[1]) * ele_size_inv_);
    w.xmaxInt = (int)std::ceil((xmax - min_[0]) * ele_size_inv_);
    w.ymaxInt = (int)std::ceil((ymax - min_[1]) * ele_size_inv_);
    if (w.xminInt < 0 || w.yminInt < 0 || w.xmaxInt > w_ || w.ymaxInt > h_ || w.xminInt >= w.xmaxInt || w.yminInt >= w.ymaxInt) {
        std::fprintf(stderr, "MultiBandMap2DCPU::renderFrame:should never happen!\n");
        return -1;
    }
    xmin = min_[0] + ele_size_ * w.xminInt;
    ymin = min_[1] + ele_size_ * w.yminInt;
    w.src = f.ext ? f.ext : (f.slot >= 0 ? slots_[f.slot].dev : nullptr);
    if (!w.src) return 0;
    const float src4[8] = { 0.f, 0.f, (float)cam_.w, 0.f, 0.f, (float)cam_.h, (float)cam_.w, (float)cam_.h };
    float dst4[8];
    for (int i = 0; i < 4; i++) {
        dst4[2 * i]     = (float)((pts[2 * i] - xmin) * length_pixel_inv_);
        dst4[2 * i + 1] = (float)((pts[2 * i + 1] - ymin) * length_pixel_inv_);
    }
    perspective_transform(src4, dst4, w.M0);
    w.tx = w.xmaxInt - w.xminInt; w.ty = w.ymaxInt - w.yminInt; w.L = band_num_;
    w.crows = w.ty * kElePixels; w.ccols = w.tx * kElePixels;
    return 1;
}

// One pass over the canvas tiles (Apply's tile loop, .cpp:478-492): the tiles this shard owns, their bounding box,
// the hash cells they fall in (for the need rectangles) and the table entries: slot address | fresh bit | culled cells.
// The reference's own per-frame O(tiles) cost is d->data() at .cpp:477.
//
// Cull (round 4): a cell of a tile in which this keyframe cannot win the max-weight select at ANY level is left out of the launch -- a
// tile whose cells are all out keeps table entry 0, exactly as if another shard owned it, and the level-0 blocks' own need test / the
// need rectangles shrink the grid to what the remaining cells depend on.  Nothing changes in what is stored: `if (srcW >= dstW)`
// (.cpp:521, :542) is false at every pixel of such a cell.  Sound because both sides are bounded from the geometry alone, with margins
// (cell_out):
//   new weights   W_i(q) is a convex combination (pyrDown) of level-0 radial weights inside the cell dilated by the pyramid's
//                 support radius 2^(L+1) px, so W_i <= wmax = the largest radial weight the frame can have there;
//   stored ones   every earlier keyframe f whose canvas held the tile left S_i >= W_i^f >= wmin_f (its smallest weight on the same
//                 dilated cell, 0 unless that lies wholly inside f's footprint) -- also when f itself was culled there, for
//                 then S_i > W_i^f.  Tile::wlb[cell] = max over f of wmin_f.
// Bit-exactness is checked, not assumed: every parity test runs with the cull on; PF_CULL=0 turns it off.
// The unit of the cull is a CELL of a tile (64 x 64 pixels, 16 per tile; experiments library, PF_CULL_SUB=2: a quadrant), dilated by the
// pyramid's support radius 2^(L+1) - 2 pixels rounded up to the lattice step (62 -> 64 for five bands, 254 -> 256 for seven).  A tile whose
// cells are all out is left out of the launch; otherwise the cells that are out travel as flag bits of its table entry and the kernels do
// not look at their pixels (kernels.hip, cell_culled), which may have been computed from input nobody produced.
bool FusionMap::build_tile_table(const QueuedFrame& f, FrameWork& w)
{
    const int tx = w.tx, ty = w.ty;
    w.sharded = opt_.shard_count > 1;
    const bool sharded = w.sharded;
    const bool cull = w.cull;                                    // decided when the keyframe was admitted (render_frame), with its lattice (lat_)
    const bool lookahead = cull && lookahead_ok();
    const int S = cull_sub_, span = 4 / S;                       // cells per tile edge; lattice steps per cell
    if (cull) w.raise.reserve((size_t)tx * ty * S * S);
    // Map2DCPU (single band, round 6): no pyramid, so a cell is not dilated; the bounds are those of the radial weight all the same, compared
    // three alpha steps apart -- the stored alpha byte is floor(254 w) (at least 2) interpolated with 15-bit taps (within 1 of its smallest
    // tap), the select is `ele.a < dst.a` (Map2DCPU.cpp:326-327): the keyframe cannot win where 254 wmax <= 254 wlb - 3.
    const float sb_gap = single_band_ ? 3.2f / 254.f : 0.f, sb_floor = single_band_ ? 6.f / 254.f : 0.f;
    auto stored_bound = [&](float wlb) { return wlb > sb_floor ? wlb - sb_gap : 0.f; };      // what cell_out compares the keyframe's weights with
    const int B = opt_.shard_block;
    // cells of the need rectangles: a shard's hash cells; for the cull alone squares of 8 x 8 tiles as well (experiments library: PF_CULL_CELL)
    static const int bc_env = exp_env_int("PF_CULL_CELL", 0);
    const int Bc = sharded ? B : (bc_env > 0 ? bc_env : 8);      // measured (profiles/r04_ab.md): 8 beats 3 / 4 (fewer, tighter-merging rectangles; less host work)
    table_tmp_.resize((size_t)tx * ty);
    w.touched.reserve((size_t)tx * ty);
    w.bx0 = tx; w.bx1 = 0; w.by0 = ty; w.by1 = 0;
    // the same box in level-0 pixels, around the cells that are rendered (== the tiles' box when nothing is culled); the squares of the rectangles likewise
    w.pbx0 = 1 << 30; w.pbx1 = 0; w.pby0 = 1 << 30; w.pby1 = 0;
    auto add_rect = [&](int sx, int sy, int x0, int y0, int x1, int y1) {
        w.pbx0 = std::min(w.pbx0, x0); w.pbx1 = std::max(w.pbx1, x1); w.pby0 = std::min(w.pby0, y0); w.pby1 = std::max(w.pby1, y1);
        if (!(sharded || cull) || w.cells_overflow) return;
        const int cx = floordiv(sx, Bc), cy = floordiv(sy, Bc);
        int k = w.ncells - 1;
        while (k >= 0 && !(w.cells[k].cx == cx && w.cells[k].cy == cy)) k--;
        if (k < 0) {
            if (w.ncells == 64) w.cells_overflow = true;
            else w.cells[w.ncells++] = FrameWork::Cell{ cx, cy, x0, y0, x1, y1 };
        } else {
            FrameWork::Cell& c = w.cells[k];
            c.x0 = std::min(c.x0, x0); c.y0 = std::min(c.y0, y0); c.x1 = std::max(c.x1, x1); c.y1 = std::max(c.y1, y1);
        }
    };
    for (int y = 0; y < ty; y++) {
        const int sy = w.yminInt + y + off_y_;
        for (int x = 0; x < tx; x++) {
            const int sx = w.xminInt + x + off_x_;
            uint64_t ent = 0;
            Tile* known = w.tiles_known ? w.tiles_known[(size_t)y * tx + x] : nullptr;
            if (known || ((!sharded || tile_owner(opt_.shard_count, B, sx, sy) == opt_.shard_rank) && !w.tiles_known)) {
                Tile* t = known ? known : store_.get_or_create(sx, sy);
                if (!t) return false;
                w.owned_all++;
                unsigned out = 0;                                  // 64 x 64 cells in which this keyframe cannot win (bit 4 * row + column)
                if (cull) {
                    // the whole tile first, against the smallest of its cells' bounds: out there is out in every cell (the tile's dilated
                    // rectangle holds each cell's) -- most culled cells lie in such tiles; without lookahead their wmin is still worked out
                    // cell by cell, with it the bounds went into wlb when the keyframe was admitted (pre_raise) and the cells are skipped
                    // A FRESH tile (no keyframe has written it: its first one copies unconditionally, .cpp:498) is rendered whole or not at
                    // all: its slot holds no weights a select could be run against, so no single cell may be left out.  It can be left out
                    // whole only through the lookahead -- its wlb then holds bounds of keyframes that wait behind this one; the one whose
                    // bound is the largest in a cell is never out there and renders the tile (all of it) before anybody looks.  Without
                    // lookahead a fresh tile's wlb is -1 and nothing is out.
                    bool tile_out = false;
                    const bool ask = !t->fresh || lookahead;
                    if (ask) {
                        float wl = t->wlb[0], unused;
                        for (int q = 1; q < S * S; q++) wl = std::min(wl, t->wlb[q]);
                        tile_out = cell_out(4 * x, 4 * y, 4, opt_.weight_type, stored_bound(wl), true, &unused);
                    }
                    if (tile_out && w.pre_raised) out = 0xffffu;       // (the keyframe's own bounds are in wlb since it was admitted)
                    else for (int q = 0; q < S * S; q++) {
                        const int qx = q % S, qy = q / S;
                        float wmin = 0.f;
                        if (cell_out(4 * x + span * qx, 4 * y + span * qy, span, opt_.weight_type, stored_bound(t->wlb[q]), ask && !tile_out, w.pre_raised ? nullptr : &wmin) || tile_out)
                            out |= S == 4 ? 1u << q : 0x33u << (8 * qy + 2 * qx);
                        if (wmin > t->wlb[q]) w.raise.push_back(FrameWork::Raise{ t, q, wmin });
                    }
                    if (t->fresh && out != 0xffffu) out = 0;
                    if (out == 0xffffu && t->fresh) {
                        // stays fresh, and is not yet a tile of the mosaic (no Ischanged: it has no pyramid, .cpp:717-718)
                        w.culled_any = true; table_tmp_[(size_t)y * tx + x] = 0; n_culled_tiles_++;
                        continue;
                    }
                    if (out == 0xffffu) {
                        // not rendered, but still a tile of this keyframe's canvas: Apply sets Ischanged on every one of them
                        // (.cpp:553), and draw() re-blends it with whatever its neighbours have become
                        w.culled_any = true; table_tmp_[(size_t)y * tx + x] = 0; n_culled_tiles_++; w.culled.push_back(t);
                        continue;
                    }
                    if (out) { w.culled_any = true; n_culled_cells_ += __builtin_popcount(out); }
                }
                if ((uint64_t)(uintptr_t)t->base >> 48) { set_error("tile slot address above 2^48: the table entry has no room for the cell flags"); return false; }
                ent = (uint64_t)(uintptr_t)t->base | (t->fresh ? 1u : 0u) | ((uint64_t)out << 48);
                w.touched.push_back(t);
                w.owned++;
                w.bx0 = std::min(w.bx0, x); w.bx1 = std::max(w.bx1, x + 1); w.by0 = std::min(w.by0, y); w.by1 = std::max(w.by1, y + 1);
                if (!out) add_rect(sx, sy, x * kElePixels, y * kElePixels, (x + 1) * kElePixels, (y + 1) * kElePixels);
                else
                    for (int r = 0; r < 4; r++) {                  // the rendered cells, row by row as runs
                        const unsigned in = ~(out >> (4 * r)) & 15u;
                        if (!in) continue;
                        const int c0 = __builtin_ctz(in), c1 = 32 - __builtin_clz(in);
                        add_rect(sx, sy, x * kElePixels + 64 * c0, y * kElePixels + 64 * r, x * kElePixels + 64 * c1, y * kElePixels + 64 * r + 64);
                    }
            }
            table_tmp_[(size_t)y * tx + x] = ent;
        }
    }
    return true;
}

// per-level windows: Gaussian level i must be valid on need[i] so that the
// Laplacian of the owned tiles is exact (pyrDown reads [2p-2, 2q+1), pyrUp +-1)
void FusionMap::level_windows(FrameWork& w)
{
    const int L = w.L;
    for (int i = L; i >= 0; i--) {
        const int rows = w.crows >> i, cols = w.ccols >> i;
        int x0 = lv_lo(w.pbx0, i), x1 = lv_hi(w.pbx1, i), y0 = lv_lo(w.pby0, i), y1 = lv_hi(w.pby1, i);
        if (i > 0) { x0 -= 1; x1 += 1; y0 -= 1; y1 += 1; }
        if (i < L) {
            x0 = std::min(x0, 2 * w.need[i + 1].x0 - 2); x1 = std::max(x1, 2 * w.need[i + 1].x1 + 1);
            y0 = std::min(y0, 2 * w.need[i + 1].y0 - 2); y1 = std::max(y1, 2 * w.need[i + 1].y1 + 1);
        }
        clampw(x0, x1, cols, w.need[i].x0, w.need[i].x1);
        clampw(y0, y1, rows, w.need[i].y0, w.need[i].y1);
    }
    // level 0 is produced by the warp in 64x4 blocks
    w.need[0].x0 = (w.need[0].x0 / 64) * 64; w.need[0].x1 = std::min(w.ccols, ((w.need[0].x1 + 63) / 64) * 64);
    w.need[0].y0 = (w.need[0].y0 / 4) * 4;   w.need[0].y1 = std::min(w.crows, ((w.need[0].y1 + 3) / 4) * 4);
}

// grow-only workspace: per-frame Gaussian levels (GW_i of the fused forms, G_i / W_i of the per-op form) and the tile-table ring
bool FusionMap::reserve_frame_workspace(FrameWork& w)
{
    const int L = w.L, tx = w.tx, ty = w.ty;
    const size_t es = lay_.f32 ? 4 : 2;
    const bool fused = opt_.fused != 0 && L >= 1;
    const size_t pxb = level_px_bytes(lay_.f32 != 0);
    bool grow = false;
    for (int i = 0; i <= L; i++) {
        const size_t n = (size_t)(w.crows >> i) * (w.ccols >> i);
        if (single_band_) continue;
        if (fused) { if (i >= 1 && i < L && (gw_[i].cap < n * pxb || gw2_[i].cap < n * pxb)) grow = true; }
        else if (g_[i].cap < n * 3 * es || wgt_[i].cap < n * 4) grow = true;
    }
    if (table_cap_ < (size_t)tx * ty) grow = true;
    if (!grow) return true;
    HIP_OK(sync_all());
    for (int i = 0; i <= L; i++) {
        const size_t n = (size_t)(w.crows >> i) * (w.ccols >> i);
        if (single_band_) continue;
        if (fused) { if (i >= 1 && i < L && (!gw_[i].reserve(n * pxb) || !gw2_[i].reserve(n * pxb))) return false; }
        else if (!g_[i].reserve(n * 3 * es) || !wgt_[i].reserve(n * 4)) return false;
    }
    if (table_cap_ < (size_t)tx * ty) {
        table_cap_ = (size_t)tx * ty * 2;
        for (int i = 0; i < kTableRing; i++) {
            if (table_host_[i]) (void)hipHostFree(table_host_[i]);
            HIP_OK(hipHostMalloc((void**)&table_host_[i], table_cap_ * 8, hipHostMallocDefault));
            if (!table_dev_[i].reserve(table_cap_ * 8)) return false;
            table_pending_[i] = false;
        }
    }
    return true;
}

// the frame's tile table -> ring slot in device memory: inside the kernel arguments of the pipelined launch (which stores it there
// itself), or staged in pinned memory and copied in the stream
bool FusionMap::place_table(FrameWork& w)
{
    const int tx = w.tx, ty = w.ty;
    w.ring = (int)(frame_seq_++ % kTableRing);
    const int ring = w.ring;
    if (table_pending_[ring]) { HIP_OK(hipEventSynchronize(table_ev_[ring])); table_pending_[ring] = false; }
    w.table_args = table_in_args_ && opt_.fused == 1 && !single_band_ && w.L >= 2 && (size_t)tx * ty <= (size_t)kArgTable;
    if (!w.table_args) {
        // staged in pinned memory and copied in the stream; the staging slot is reused once that copy has run
        if (!wait_for(table_release_[ring])) return false;
        std::memcpy(table_host_[ring], table_tmp_.data(), (size_t)tx * ty * 8);
        HIP_OK(hipMemcpyAsync(table_dev_[ring].p, table_host_[ring], (size_t)tx * ty * 8, hipMemcpyHostToDevice, stream_));
    }
    w.dtab = (const uint64_t*)table_dev_[ring].p;
    return true;
}

// warp (.cpp:443-452): destination -> source map, window, radial weight constants
void FusionMap::warp_args(const QueuedFrame& f, FrameWork& w)
{
    WarpArgs& a = w.a;
    a = WarpArgs{};
    if (!invert3x3(w.M0, a.M)) std::memset(a.M, 0, sizeof(a.M));
    a.srows = f.rows; a.scols = f.cols; a.sstep = f.step;
    a.crows = w.crows; a.ccols = w.ccols;
    a.y_off = w.need[0].y0; a.x_off = w.need[0].x0; a.wrows = w.need[0].y1 - w.need[0].y0; a.wcols = w.need[0].x1 - w.need[0].x0;
    a.xc = (float)(f.cols / 2); a.yc = (float)(f.rows / 2);
    a.dis_max = std::sqrt(a.xc * a.xc + a.yc * a.yc);
    a.weight_type = opt_.weight_type;
    a.src_cn = f.cn == 4 ? 4 : 3;
}

// Map2DCPU::renderFrame (Map2DCPU.cpp:236-334): per-pixel work only, so a shard needs no halo
bool FusionMap::launch_single_band(const QueuedFrame& f, FrameWork& w)
{
    WarpArgs& a = w.a;
    if (w8_rows_ != f.rows || w8_cols_ != f.cols) {
        HIP_OK(sync_all());
        if (!w8_.reserve((size_t)f.rows * f.cols)) return false;
        launch_weight8(stream_, (uint8_t*)w8_.p, f.rows, f.cols, opt_.weight_type);
        w8_rows_ = f.rows; w8_cols_ = f.cols;
    }
    a.y_off = w.by0 * kElePixels; a.x_off = w.bx0 * kElePixels;
    a.wrows = (w.by1 - w.by0) * kElePixels; a.wcols = (w.bx1 - w.bx0) * kElePixels;
    prof_begin(K_SINGLE, (double)a.src_cn * f.rows * f.cols + (double)a.wrows * a.wcols * 8);
    launch_single(stream_, w.src, (const uint8_t*)w8_.p, a, w.dtab, w.tx);
    prof_end();
    return true;
}

// Fused forms: where the level kernels run.
//   C[i]         compute region of level i: its launch must cover the owned tiles and produce GW_{i+1} wherever the level i+1
//                launch stages its halo (its region -4 / +3)
//   need bitmaps upper levels: one bit per block of the level's grid -- does a rendered cell lie within the pyramid's reach of it (the
//                rule the level-0 blocks apply to themselves in the kernel: (3 * 2^(L-i) - 2) level-i pixels)?  From row bitmaps of the
//                rendered cells (canvases up to 32 tiles wide); the jobs carry them in their launches' kernel arguments
//   rectangles   a shard's tiles are scattered hash cells, and the compute regions are their bounding box: per level, one rectangle of
//                64x32 blocks per cell says where something owned depends on a block (the same recursion as `need`, applied per
//                cell); blocks outside every rectangle exit at once.  The fallback of the bitmaps.  (Unsharded, no cull: every block runs.)
void FusionMap::plan_fused_levels(FrameWork& w)
{
    const int L = w.L, tx = w.tx, ty = w.ty, crows = w.crows, ccols = w.ccols;
    Win* C = w.C;
    for (int i = L - 1; i >= 0; i--) {
        const int rows = crows >> i, cols = ccols >> i;
        // the origin stays even (a block's quads and its part of level i+1 start on even pixels): a box of 64-pixel cells is odd at level 6
        int x0 = lv_lo(w.pbx0, i) & ~1, x1 = lv_hi(w.pbx1, i), y0 = lv_lo(w.pby0, i) & ~1, y1 = lv_hi(w.pby1, i);
        if (i < L - 1) {
            x0 = std::min(x0, 2 * (C[i + 1].x0 - 4)); x1 = std::max(x1, 2 * (C[i + 1].x1 + 3));
            y0 = std::min(y0, 2 * (C[i + 1].y0 - 4)); y1 = std::max(y1, 2 * (C[i + 1].y1 + 3));
        }
        clampw(x0, x1, cols, C[i].x0, C[i].x1);
        clampw(y0, y1, rows, C[i].y0, C[i].y1);
    }
    const bool partial = (w.sharded || w.culled_any) && !w.cells_overflow;
    const int BHr = level_block_rows(lay_.f32 != 0);
    if (partial && 4 * tx <= 128 && L >= 2) {
        typedef unsigned __int128 u128;
        cell_rows_.assign((size_t)4 * ty, 0);
        for (int y = 0; y < ty; y++)
            for (int x = 0; x < tx; x++) {
                const uint64_t e = table_tmp_[(size_t)y * tx + x];
                if (!e) continue;
                const unsigned in = ~(unsigned)(e >> 48) & 0xffffu;
                for (int r = 0; r < 4; r++) cell_rows_[(size_t)4 * y + r] |= (u128)((in >> (4 * r)) & 15u) << (4 * x);
            }
        for (int i = 1; i < L; i++) {
            const int reach = ((3 << (L - i)) - 2) << i, nbx = (C[i].x1 - C[i].x0 + 63) / 64, nby = (C[i].y1 - C[i].y0 + BHr - 1) / BHr;
            w.need_n[i] = 0;
            if (nbx <= 0 || nby <= 0 || (nbx * nby + 31) / 32 > kNeedWords) continue;
            uint32_t* bits = need_tmp_[i];
            std::memset(bits, 0, sizeof(uint32_t) * (size_t)((nbx * nby + 31) / 32));
            for (int gy = 0; gy < nby; gy++) {
                const int y0 = std::max(((C[i].y0 + gy * BHr) << i) - reach, 0) >> 6, y1 = std::min((((C[i].y0 + gy * BHr + BHr) << i) - 1 + reach) >> 6, 4 * ty - 1);
                u128 rowsum = 0;
                for (int r = y0; r <= y1; r++) rowsum |= cell_rows_[(size_t)r];
                if (!rowsum) continue;
                for (int gx = 0; gx < nbx; gx++) {
                    const int x0 = std::max(((C[i].x0 + gx * 64) << i) - reach, 0) >> 6, x1 = std::min((((C[i].x0 + gx * 64 + 64) << i) - 1 + reach) >> 6, 4 * tx - 1);
                    if (x0 > x1) continue;
                    const u128 m = (x1 - x0 >= 127 ? ~(u128)0 : (((u128)1 << (x1 - x0 + 1)) - 1)) << x0;
                    if (rowsum & m) { const int b = gy * nbx + gx; bits[(size_t)b >> 5] |= 1u << (b & 31); }
                }
            }
            w.need_n[i] = nbx * nby;
        }
    }
    if (!(partial && opt_.fused == 1)) {
        px_level0_ += (double)(C[0].x1 - C[0].x0) * (C[0].y1 - C[0].y0);
        return;
    }
    struct R { int x0, y0, x1, y1; };
    std::vector<R> lv[kMaxLevels];
    for (int c = 0; c < w.ncells; c++) {
        // N[i]: where Gaussian level i is needed for this cell's tiles (pixel-exact: pyrDown reads [2p-2, 2p+2],
        // pyrUp +-1).  The level-i block at b runs iff it holds owned pixels or its part of level i+1 lies in
        // N[i+1]; what else it computes from unproduced input is never read.
        const FrameWork::Cell& ce = w.cells[c];
        Win N[kMaxLevels];
        for (int i = L; i >= 0; i--) {
            const int rows = crows >> i, cols = ccols >> i;
            int x0 = lv_lo(ce.x0, i), x1 = lv_hi(ce.x1, i), y0 = lv_lo(ce.y0, i), y1 = lv_hi(ce.y1, i);
            if (i > 0) { x0 -= 1; x1 += 1; y0 -= 1; y1 += 1; }
            if (i < L) {
                x0 = std::min(x0, 2 * N[i + 1].x0 - 2); x1 = std::max(x1, 2 * N[i + 1].x1 + 1);
                y0 = std::min(y0, 2 * N[i + 1].y0 - 2); y1 = std::max(y1, 2 * N[i + 1].y1 + 1);
            }
            clampw(x0, x1, cols, N[i].x0, N[i].x1);
            clampw(y0, y1, rows, N[i].y0, N[i].y1);
        }
        for (int i = 0; i < L; i++) {
            int x0 = std::min(lv_lo(ce.x0, i), 2 * N[i + 1].x0), x1 = std::max(lv_hi(ce.x1, i), 2 * N[i + 1].x1);
            int y0 = std::min(lv_lo(ce.y0, i), 2 * N[i + 1].y0), y1 = std::max(lv_hi(ce.y1, i), 2 * N[i + 1].y1);
            x0 = std::max(x0, C[i].x0); y0 = std::max(y0, C[i].y0); x1 = std::min(x1, C[i].x1); y1 = std::min(y1, C[i].y1);
            if (x0 >= x1 || y0 >= y1) continue;
            lv[i].push_back(R{ (x0 - C[i].x0) / 64, (y0 - C[i].y0) / BHr, (x1 - C[i].x0 + 63) / 64, (y1 - C[i].y0 + BHr - 1) / BHr });
        }
    }
    for (int i = 0; i < L; i++) {
        // at most kMaxRects (level 0) / kMaxRectsUpper travel with a job: merge the pair whose common bounding box adds the fewest blocks
        // (extra blocks only cost time: they hold no owned pixel and write no tile)
        std::vector<R>& v = lv[i];
        auto area = [](const R& r) { return (long)(r.x1 - r.x0) * (r.y1 - r.y0); };
        const int cap = i == 0 ? kMaxRects : kMaxRectsUpper;       // what a job of this level can carry (kernels.hpp)
        while ((int)v.size() > cap) {
            size_t ba = 0, bb = 1; long best = -1;
            for (size_t p = 0; p < v.size(); p++)
                for (size_t q = p + 1; q < v.size(); q++) {
                    const R u{ std::min(v[p].x0, v[q].x0), std::min(v[p].y0, v[q].y0), std::max(v[p].x1, v[q].x1), std::max(v[p].y1, v[q].y1) };
                    const long add = area(u) - area(v[p]) - area(v[q]);
                    if (best < 0 || add < best) { best = add; ba = p; bb = q; }
                }
            v[ba] = R{ std::min(v[ba].x0, v[bb].x0), std::min(v[ba].y0, v[bb].y0), std::max(v[ba].x1, v[bb].x1), std::max(v[ba].y1, v[bb].y1) };
            v.erase(v.begin() + bb);
        }
        w.nrect[i] = (int)v.size();
        for (int k = 0; k < w.nrect[i]; k++) w.rects[i][k] = BlockRect{ (short)v[k].x0, (short)v[k].y0, (short)v[k].x1, (short)v[k].y1 };
        if (v.empty()) { w.nrect[i] = 1; w.rects[i][0] = BlockRect{ 0, 0, 0, 0 }; }      // nothing needed at this level: an empty rectangle
    }
    // level-0 blocks that run (render_stats, bench --shard strong; the numerator of roofline.frac)
    if (w.need_n[1] > 0 && level0_need_reach(lay_, w.table_args ? tx * ty : 0, w.nrect[0]) > 0) {
        // the level-0 blocks pick themselves in the kernel (within 94 px of a rendered cell): counted through the level-1 bitmap, whose
        // blocks are 2 x 2 of them under nearly the same rule (92 px)
        int n1 = 0;
        for (int k = 0; k < (w.need_n[1] + 31) / 32; k++) n1 += __builtin_popcount(need_tmp_[1][k]);
        w.blocks_run0 = std::min(4.0 * n1, (double)((C[0].x1 - C[0].x0 + 63) / 64) * ((C[0].y1 - C[0].y0 + BHr - 1) / BHr));
    } else {
        const int nbx = (C[0].x1 - C[0].x0 + 63) / 64, nby = (C[0].y1 - C[0].y0 + BHr - 1) / BHr;
        block_bits_.assign((size_t)std::max(nbx, 0) * std::max(nby, 0), 0);
        for (int k = 0; k < w.nrect[0]; k++)
            for (int gy = w.rects[0][k].y0; gy < w.rects[0][k].y1; gy++)
                if (w.rects[0][k].x1 > w.rects[0][k].x0) std::memset(block_bits_.data() + (size_t)gy * nbx + w.rects[0][k].x0, 1, (size_t)(w.rects[0][k].x1 - w.rects[0][k].x0));
        for (uint8_t v : block_bits_) w.blocks_run0 += v;
    }
    px_level0_ += w.blocks_run0 * 64 * BHr;
#if PF_EXPERIMENTS
    static const bool exact_stat = exp_env("PF_CULL_EXACT_STAT") != nullptr;
    if (exact_stat) cull_exact_stat(w);
#endif
}

// Accounting of a pipelined launch: share of each level's canvas pixels (this rank's tiles) that the blocks which RUN cover: 1 unless the
// cull or a shard leaves blocks out.  Level 0 with the blocks' own need test (k_levels, need_r0): through the level-1 bitmap here, and --
// for the launches that are bracketed by events -- exactly, after the launch is out (exact_level0_share).
void FusionMap::run_shares(const FrameWork& w, double run_share[kMaxLevels], int* exact_r0)
{
    const int L = w.L;
    const Win* C = w.C;
    *exact_r0 = 0;
    for (int i = 0; i < kMaxLevels; i++) run_share[i] = 1.0;
    if (!((w.sharded || w.culled_any) && !w.cells_overflow)) return;
    const int BHr = level_block_rows(lay_.f32 != 0);
    for (int i = 0; i < L; i++) {
        const int nbx = (C[i].x1 - C[i].x0 + 63) / 64, nby = (C[i].y1 - C[i].y0 + BHr - 1) / BHr;
        if (nbx <= 0 || nby <= 0) continue;
        double run = 0;
        const int r0 = i == 0 && w.need_n[1] > 0 ? level0_need_reach(lay_, w.table_args ? w.tx * w.ty : 0, w.nrect[0]) : 0;
        if (i == 0 && r0 > 0 && prof_would(K_LEVEL0) && cell_rows_.size() == (size_t)4 * w.ty) {
            *exact_r0 = r0;                             // counted exactly AFTER the launch is out: ~50 us of host time that must not delay it
            run = w.blocks_run0;
        } else if (i == 0) run = w.blocks_run0;
        else if (w.need_n[i] > 0) { for (int k = 0; k < (w.need_n[i] + 31) / 32; k++) run += __builtin_popcount(need_tmp_[i][k]); }
        else {
            block_bits_.assign((size_t)nbx * nby, 0);
            for (int k = 0; k < w.nrect[i]; k++)
                for (int gy = std::max<int>(w.rects[i][k].y0, 0); gy < std::min<int>(w.rects[i][k].y1, nby); gy++)
                    for (int gx = std::max<int>(w.rects[i][k].x0, 0); gx < std::min<int>(w.rects[i][k].x1, nbx); gx++) block_bits_[(size_t)gy * nbx + gx] = 1;
            for (uint8_t v : block_bits_) run += v;
            if (!w.nrect[i]) run = (double)nbx * nby;
        }
        const double ts = kElePixels >> i;
        run_share[i] = std::min(1.0, run * 64.0 * BHr / std::max(1.0, (double)w.owned_all * ts * ts));
    }
}

// the blocks' own need rule (k_levels need_r0) evaluated from the rendered cells' row bitmaps: the share of the level-0 canvas pixels
// that the blocks which run cover (the column masks once per launch, one AND per block: ~8 us for cfg-A's 7072 blocks)
double FusionMap::exact_level0_share(const FrameWork& w, int r0)
{
    typedef unsigned __int128 u128;
    const Win* C = w.C;
    const int BHr = level_block_rows(lay_.f32 != 0);
    const int nbx = (C[0].x1 - C[0].x0 + 63) / 64, nby = (C[0].y1 - C[0].y0 + BHr - 1) / BHr;
    static thread_local std::vector<u128> colmask;
    colmask.assign((size_t)std::max(nbx, 0), 0);
    for (int gx = 0; gx < nbx; gx++) {
        const int x0 = std::max(C[0].x0 + gx * 64 - r0, 0) >> 6, x1 = std::min(C[0].x0 + gx * 64 + 63 + r0, w.ccols - 1) >> 6;
        if (x0 <= x1) colmask[(size_t)gx] = (x1 - x0 >= 127 ? ~(u128)0 : (((u128)1 << (x1 - x0 + 1)) - 1)) << x0;
    }
    long run = 0;
    for (int gy = 0; gy < nby; gy++) {
        const int y0 = std::max(C[0].y0 + gy * BHr - r0, 0) >> 6, y1 = std::min(C[0].y0 + gy * BHr + BHr - 1 + r0, w.crows - 1) >> 6;
        u128 rowsum = 0;
        for (int r = y0; r <= y1; r++) rowsum |= cell_rows_[(size_t)r];
        if (!rowsum) continue;
        const uint64_t lo = (uint64_t)rowsum, hi = (uint64_t)(rowsum >> 64);
        for (int gx = 0; gx < nbx; gx++) run += ((lo & (uint64_t)colmask[(size_t)gx]) | (hi & (uint64_t)(colmask[(size_t)gx] >> 64))) != 0;
    }
    const double n0 = (double)w.owned_all * kElePixels * kElePixels;
    return std::min(1.0, run * 64.0 * BHr / std::max(1.0, n0));
}

// fused = 1: one launch per keyframe -- this frame's level 0 plus the pending upper levels of the frames before it
bool FusionMap::launch_fused_pipeline(const QueuedFrame& f, FrameWork& w)
{
    const int L = w.L;
    const size_t es = lay_.f32 ? 4 : 2;
    const double E = 3 * es + 4, owned_tiles = w.owned_all;       // algorithmic bytes: every canvas tile of this rank, culled or not
    double run_share[kMaxLevels];
    int exact_r0 = 0;
    run_shares(w, run_share, &exact_r0);
    PipeFrame cur;
    cur.valid = true; cur.ring = w.ring; cur.tx = w.tx; cur.crows = w.crows; cur.ccols = w.ccols;
    for (int i = 0; i < L; i++) {
        cur.C[i] = w.C[i]; cur.nrect[i] = w.nrect[i];
        for (int k = 0; k < w.nrect[i]; k++) cur.rect[i][k] = w.rects[i][k];
        const double ts = kElePixels >> i, n = owned_tiles * ts * ts;
        // algorithmic bytes: frame read once + per tile-level pixel 4 (stored weight) + E (payload)
        const double tile_bytes = n * (4 + E) + (i + 1 == L ? n / 4 * (4 + E) : 0), frame_bytes_read = i == 0 ? (double)w.a.src_cn * f.rows * f.cols : 0;
        cur.bytes[i] = tile_bytes + frame_bytes_read;                         // every canvas tile of this rank, culled or not
        cur.bytes_run[i] = tile_bytes * run_share[i] + frame_bytes_read;      // the part of the canvas whose blocks run
    }
    if (w.table_args) { cur.table_args = table_tmp_.data(); cur.table_n = w.tx * w.ty; }
    for (int i = 1; i < L; i++) { cur.need_n[i] = w.need_n[i]; if (w.need_n[i] > 0) std::memcpy(cur.need_bits[i], need_tmp_[i], sizeof(uint32_t) * (size_t)((w.need_n[i] + 31) / 32)); }
    if (!launch_pipeline(&cur, &w.a, w.src)) return false;
    if (exact_r0 > 0 && prof_on_ && !prof_pending_.empty()) {
        // this launch is bracketed by events: replace the level-0 job's estimated run share in its record by the exact one -- now that
        // the launch is on its way
        const double n0 = owned_tiles * kElePixels * kElePixels, tile_bytes0 = n0 * (4 + E) + (L == 1 ? n0 / 4 * (4 + E) : 0);
        prof_pending_.back().bytes_run += tile_bytes0 * (exact_level0_share(w, exact_r0) - run_share[0]);
    }
    return true;
}

// fused = 2 / 3: one launch per level, level 0 on stream_ and the upper levels on kUpperStreams more.
// Level i of frame f runs after level i-1 of frame f (GW_i, event) and, by stream order, after level i of
// frame f-1 (tiles are updated in feed order).  GW buffers are double-buffered by frame parity; a writer
// waits for the reader two frames back.
bool FusionMap::launch_level_streams(const QueuedFrame& f, FrameWork& w)
{
    const int L = w.L;
    const size_t es = lay_.f32 ? 4 : 2;
    const double E = 3 * es + 4;
    const Win* C = w.C;
    const unsigned long long fidx = frame_seq_ - 1;
    const int slot = (int)(fidx % kLvlRing);
    DevBuf* gw = (fidx & 1) ? gw2_ : gw_;
    for (int i = 0; i < L; i++) {
        if (!lvl_stream_[i]) {
            // HIP multiplexes streams onto a few hardware queues (4 by default), and streams that share one
            // serialize: keep the count small.  Level 0 has its own stream, the upper levels -- a dependent
            // chain within a frame anyway -- share kUpperStreams (experiments library: PF_LEVEL_STREAMS, PF_SINGLE_STREAM).
            static const int n_upper = exp_env_int("PF_LEVEL_STREAMS", kUpperStreams);
            if (i == 0) lvl_stream_[0] = stream_;
            else if (exp_env("PF_SINGLE_STREAM") || n_upper <= 0) lvl_stream_[i] = stream_;      // diagnostics: serial kernel times
            else if (i > n_upper) lvl_stream_[i] = lvl_stream_[1 + (i - 1) % n_upper];
            else HIP_OK(hipStreamCreateWithFlags(&lvl_stream_[i], hipStreamNonBlocking));
        }
        for (int k = 0; k < kLvlRing; k++)
            if (!lvl_ev_[i][k]) HIP_OK(hipEventCreateWithFlags(&lvl_ev_[i][k], hipEventDisableTiming));
    }
    for (int i = 0; i < L; i++) {
        hipStream_t st = lvl_stream_[i];
        const double ts = kElePixels >> i, n = (double)(w.bx1 - w.bx0) * (w.by1 - w.by0) * ts * ts;
        const bool top = (i + 1 == L);
        if (i > 0) HIP_OK(hipStreamWaitEvent(st, lvl_ev_[i - 1][slot], 0));
        if (!top && fidx >= 2) HIP_OK(hipStreamWaitEvent(st, lvl_ev_[i + 1][(int)((fidx - 2) % kLvlRing)], 0));
        // algorithmic bytes: frame read once + per tile-level pixel 4 (stored weight) + E (payload)
        double bytes = n * (4 + E) + (top ? n / 4 * (4 + E) : 0);
        if (i == 0) bytes += (double)w.a.src_cn * f.rows * f.cols;
        prof_begin(i == 0 ? K_LEVEL0 : K_LEVEL, bytes, st);
        launch_level(st, lay_, i, w.crows >> i, w.ccols >> i, C[i].x0, C[i].y0, C[i].x1, C[i].y1, w.tx, top, !top,
                     i == 0 ? &w.a : nullptr, w.src, i == 0 ? nullptr : gw[i].p, top ? nullptr : gw[i + 1].p, w.dtab, opt_.fused);   // 2: 4-stage k_level, 3: k_level3
        prof_end();
        HIP_OK(hipEventRecord(lvl_ev_[i][slot], st));
    }
    // the tile table of this ring slot is read until the last level has run
    HIP_OK(hipEventRecord(table_ev_[w.ring], lvl_stream_[L - 1]));
    table_pending_[w.ring] = true;
    return true;
}

// fused = 0: one kernel per reference op -- warp (.cpp:443-452), the Gaussian pyramids (.cpp:469 first loop, .cpp:471-474),
// Laplacian + select into tiles (.cpp:469 second loop, .cpp:476-555)
bool FusionMap::launch_per_op(const QueuedFrame& f, FrameWork& w)
{
    const int L = w.L;
    const size_t es = lay_.f32 ? 4 : 2;
    const double win0 = (double)w.a.wrows * w.a.wcols;
    prof_begin(K_WARP, 3.0 * f.rows * f.cols + win0 * (3 * es + 4));
    launch_warp(stream_, lay_.f32, w.src, w.a, g_[0].p, (float*)wgt_[0].p);
    prof_end();
    for (int i = 0; i < L; i++) {
        const Win& d = w.need[i + 1];
        const double nd = (double)(d.x1 - d.x0) * (d.y1 - d.y0);
        prof_begin(K_PYRDOWN_IMG, nd * 4 * 3 * es + nd * 3 * es);
        launch_pyrdown(stream_, lay_.f32 ? 1 : 0, g_[i].p, w.crows >> i, w.ccols >> i, g_[i + 1].p, d.y0, d.y1, d.x0, d.x1);
        prof_end();
        prof_begin(K_PYRDOWN_W, nd * 4 * 4 + nd * 4);
        launch_pyrdown(stream_, 2, wgt_[i].p, w.crows >> i, w.ccols >> i, wgt_[i + 1].p, d.y0, d.y1, d.x0, d.x1);
        prof_end();
    }
    for (int i = 0; i <= L; i++) {
        const double ts = kElePixels >> i, n = (double)(w.bx1 - w.bx0) * (w.by1 - w.by0) * ts * ts;
        prof_begin(K_LAP_SELECT, n * (3 * es + 4 + 4) + (i < L ? n / 4 * 3 * es : 0));
        launch_lap_select(stream_, lay_, i, g_[i].p, i < L ? g_[i + 1].p : nullptr, (const float*)wgt_[i].p,
                          w.crows >> i, w.ccols >> i, w.dtab, w.tx, w.by0, w.by1, w.bx0, w.bx1);
        prof_end();
    }
    return true;
}

// the keyframe is in: its table slot and frame slot retire with the launches, the tiles it touched carry pixels and want a redraw
// (Ischanged, .cpp:553), the cull's weight bounds take this keyframe's
bool FusionMap::retire_frame(const QueuedFrame& f, FrameWork& w, bool fused)
{
    if (!table_pending_[w.ring] && !(fused && opt_.fused == 1)) {
        // this frame's kernels were the last readers of its tile table (fused = 1 retires it in launch_pipeline)
        if (!submitted()) return false;
        table_release_[w.ring] = work_no_;
    }
    if (f.slot >= 0) {
        const hipError_t e = hipEventRecord(slots_[f.slot].consumed, stream_);
        slots_[f.slot].pending = e == hipSuccess;
        release_slot(f);                           // no longer held by the queue: reusable once `consumed` has passed
        HIP_OK(e);
    }
    for (Tile* t : w.touched) { t->fresh = false; t->changed = true; }
    for (Tile* t : w.culled) t->changed = true;
    for (auto& r : w.raise) r.t->wlb[r.q] = std::max(r.t->wlb[r.q], r.w);
    n_rendered_++;
    log_rendered(f);
    return true;
}

#if PF_EXPERIMENTS
// experiments library, PF_CULL_EXACT_STAT=1: level-0 blocks within the pyramid's reach (94 px for five bands) of a cell that is rendered, and
// the same question for the upper levels -- blocks inside the need rectangles against blocks within reach of a rendered cell (printed
// every 100 keyframes).  The accumulators are members: two maps on two threads do not share them.
void FusionMap::cull_exact_stat(const FrameWork& w)
{
    const int L = w.L, tx = w.tx, crows = w.crows, ccols = w.ccols, BHr = level_block_rows(lay_.f32 != 0);
    const Win* C = w.C;
    const int R0 = 94, nbx = (C[0].x1 - C[0].x0 + 63) / 64, nby = (C[0].y1 - C[0].y0 + BHr - 1) / BHr;
    long cnt = 0;
    for (int gy = 0; gy < nby; gy++)
        for (int gx = 0; gx < nbx; gx++) {
            const int x0 = C[0].x0 + gx * 64 - R0, x1 = C[0].x0 + gx * 64 + 63 + R0, y0 = C[0].y0 + gy * BHr - R0, y1 = C[0].y0 + gy * BHr + BHr - 1 + R0;
            bool need = false;
            for (int qy = std::max(y0, 0) >> 6; qy <= (std::min(y1, crows - 1) >> 6) && !need; qy++)
                for (int qx = std::max(x0, 0) >> 6; qx <= (std::min(x1, ccols - 1) >> 6) && !need; qx++) {
                    const uint64_t e = table_tmp_[(size_t)(qy >> 2) * tx + (qx >> 2)];
                    need = e != 0 && !((e >> (48 + (qy & 3) * 4 + (qx & 3))) & 1);
                }
            cnt += need;
        }
    px_level0_exact_ += (double)cnt * 64 * BHr;
    for (int i = 1; i < L; i++) {
        const int reach = ((3 << (L - i)) - 2) << i, nbxi = (C[i].x1 - C[i].x0 + 63) / 64, nbyi = (C[i].y1 - C[i].y0 + BHr - 1) / BHr;
        for (int gy = 0; gy < nbyi; gy++)
            for (int gx = 0; gx < nbxi; gx++) {
                bool inr = false;
                for (int k = 0; k < w.nrect[i]; k++) inr = inr || (gx >= w.rects[i][k].x0 && gx < w.rects[i][k].x1 && gy >= w.rects[i][k].y0 && gy < w.rects[i][k].y1);
                up_rect_[i] += inr;
                const int x0 = ((C[i].x0 + gx * 64) << i) - reach, x1 = ((C[i].x0 + gx * 64 + 64) << i) - 1 + reach;
                const int y0 = ((C[i].y0 + gy * BHr) << i) - reach, y1 = ((C[i].y0 + gy * BHr + BHr) << i) - 1 + reach;
                bool need = false;
                for (int qy = std::max(y0, 0) >> 6; qy <= (std::min(y1, crows - 1) >> 6) && !need; qy++)
                    for (int qx = std::max(x0, 0) >> 6; qx <= (std::min(x1, ccols - 1) >> 6) && !need; qx++) {
                        const uint64_t e = table_tmp_[(size_t)(qy >> 2) * tx + (qx >> 2)];
                        need = e != 0 && !((e >> (48 + (qy & 3) * 4 + (qx & 3))) & 1);
                    }
                up_exact_[i] += need;
            }
    }
    if (++up_n_ % 100 == 0)
        for (int i = 1; i < L; i++) std::fprintf(stderr, "upper level %d: blocks in rectangles %.1f, within reach of a rendered cell %.1f per keyframe\n", i, up_rect_[i] / up_n_, up_exact_[i] / up_n_);
}
#endif

int FusionMap::render_log(long long* out, int cap)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    const int n = (int)std::min<size_t>(render_log_.size(), (size_t)std::max(cap, 0));
    for (int i = 0; i < n; i++) out[i] = render_log_[render_log_.size() - (size_t)n + (size_t)i];
    return (int)render_log_.size();
}

// May this frame's tiles be culled?  Only for a tame map: every canvas corner (with the pyramid halo) in front of the camera and
// far inside the int range, as the kernels' fast path assumes -- anything else renders every tile.
bool FusionMap::cull_frame_ok(const double M[9], int crows, int ccols) const
{
    const double xs[2] = { -600.0, ccols + 600.0 }, ys[2] = { -600.0, crows + 600.0 };
    int sign = 0;
    for (int i = 0; i < 4; i++) {
        const double x = xs[i & 1], y = ys[i >> 1], W = M[6] * x + M[7] * y + M[8];
        if (!(std::fabs(W) > 1e-12) || !std::isfinite(W)) return false;
        const int sg = W > 0 ? 1 : -1;
        if (sign && sg != sign) return false;
        sign = sg;
        if (!(std::fabs((M[0] * x + M[1] * y + M[2]) / W) < 1.0e7) || !(std::fabs((M[3] * x + M[4] * y + M[5]) / W) < 1.0e7)) return false;
    }
    return true;
}

// The canvas lattice (64 (k - dil), 64 (m - dil)), k = 0 .. ccols / 64 + 2 dil, mapped into the source frame: position, squared distance
// from the image centre, inside-the-frame flag.  A cell's dilated rectangle has its corners on it.  Points are mapped on first use
// (a shard asks for an eighth of them); one division per point, no square root.
void FusionMap::cull_lattice(const double M[9], int crows, int ccols, int cols, int rows, int dil, bool map_all)
{
    lat_.all = map_all;
    lat_.dil = dil;                                             // dilation of a cell in lattice steps of 64 pixels
    lat_.nx = ccols / 64 + 2 * dil + 1; lat_.ny = crows / 64 + 2 * dil + 1;
    const size_t n = (size_t)lat_.nx * lat_.ny;
    lat_.sx.resize(n); lat_.sy.resize(n); lat_.d.resize(n); lat_.in.assign(n, 2);          // 2: not mapped yet
    lat_.xc = (double)(cols / 2); lat_.yc = (double)(rows / 2); lat_.dis_max = std::sqrt(lat_.xc * lat_.xc + lat_.yc * lat_.yc);
    lat_.inv_dis_max = 1.0 / lat_.dis_max;
    lat_.cols = cols; lat_.rows = rows;
    for (int i = 0; i < 9; i++) lat_.M[i] = M[i];
    if (!map_all) return;                                       // a shard that owns a small part of the canvas asks for a fraction of the points: on first use
    // unsharded (or a shard that owns most of this canvas: the replicas of bench.py's weak mode own all of it), every point is needed (a cell's
    // dilated rectangle has its corners on neighbouring points): all of them now, row by row,
    // in loops without branches that the compiler turns into packed divisions (a third of the time of mapping them one by one)
    const double xc = lat_.xc, yc = lat_.yc, cmax = cols - 2.0, rmax = rows - 2.0;
    for (int m = 0; m < lat_.ny; m++) {
        const double y = 64.0 * (m - dil), n0 = M[1] * y + M[2], n1 = M[4] * y + M[5], w0 = M[7] * y + M[8];
        double* __restrict__ sx = lat_.sx.data() + (size_t)m * lat_.nx; double* __restrict__ sy = lat_.sy.data() + (size_t)m * lat_.nx;
        double* __restrict__ d = lat_.d.data() + (size_t)m * lat_.nx; unsigned char* __restrict__ in = lat_.in.data() + (size_t)m * lat_.nx;
        for (int k = 0; k < lat_.nx; k++) {
            const double x = 64.0 * (k - dil), iw = 1.0 / (M[6] * x + w0);
            const double px = (M[0] * x + n0) * iw, py = (M[3] * x + n1) * iw, dx = px - xc, dy = py - yc;
            sx[k] = px; sy[k] = py; d[k] = dx * dx + dy * dy;
        }
        for (int k = 0; k < lat_.nx; k++) in[k] = (unsigned char)((sx[k] >= 1.0) & (sx[k] <= cmax) & (sy[k] >= 1.0) & (sy[k] <= rmax));
    }
}

inline size_t FusionMap::lattice_point(int k, int m)
{
    const size_t o = (size_t)m * lat_.nx + k;
    if (lat_.in[o] == 2) {
        const double* M = lat_.M;
        const double x = 64.0 * (k - lat_.dil), y = 64.0 * (m - lat_.dil), iw = 1.0 / (M[6] * x + M[7] * y + M[8]);
        const double px = (M[0] * x + M[1] * y + M[2]) * iw, py = (M[3] * x + M[4] * y + M[5]) * iw;
        lat_.sx[o] = px; lat_.sy[o] = py;
        const double dx = px - lat_.xc, dy = py - lat_.yc;
        lat_.d[o] = dx * dx + dy * dy;
        lat_.in[o] = px >= 1.0 && px <= lat_.cols - 2.0 && py >= 1.0 && py <= lat_.rows - 2.0;
    }
    return o;
}

// The radial weight (weightImage, .cpp:396-418, gathered at the NEAREST source pixel, 0 outside the frame) over the canvas rectangle with
// lattice corners (k, m) .. (k + span + 2 dil, m + span + 2 dil) -- a cell of a tile (span lattice steps on a side) dilated by 64 dil
// pixels -- against `wlb`, the lower bound of what the cell stores:
//   returns true when every weight the keyframe can have there is below wlb (the cell is out);
//   *wmin <= every weight it has there (0 unless the rectangle maps wholly inside the frame).
// The rectangle maps to a convex quadrilateral Q of the source plane (M is projective and W keeps its sign, cull_frame_ok); the weight
// falls with the distance from the image centre c, so the largest weight sits at the point of Q nearest to c and the smallest at its
// farthest corner.  "Largest weight < wlb" <=> dist(c, Q) > T, T the distance at which the weight -- with the margins below -- reaches
// wlb; decided from the corners alone when one of them lies within T (most cells that stay in), by the exact point-to-quadrilateral
// distance otherwise.  Margins: 2 source pixels for the nearest-pixel rounding (0.71) and the float arithmetic of the kernels, 1e-5 on
// the weight for the pyramid's own rounding.
bool FusionMap::cell_out(int k, int m, int span, int weight_type, float wlb, bool want_out, float* wmin)
{
    const int e = span + 2 * lat_.dil;                          // lattice steps across the dilated cell
    const size_t c[4] = { lattice_point(k, m), lattice_point(k + e, m), lattice_point(k + e, m + e), lattice_point(k, m + e) };
    const double d2[4] = { lat_.d[c[0]], lat_.d[c[1]], lat_.d[c[2]], lat_.d[c[3]] };
    const double mpx = cull_margin_px_, mw = cull_margin_w_;       // 2 source pixels, 1e-5 (see the header)
    if (wmin) {                                                    // (nullptr: only the question whether the cell is out)
        *wmin = 0.f;
        // (weight type 0: wmin can exceed wlb only if the farthest corner lies within (1 - 1e-5 - wlb) dis_max - 2 of the centre -- in the steady
        // state it rarely does, and the square root is not taken)
        const double far2 = std::max(std::max(d2[0], d2[1]), std::max(d2[2], d2[3]));
        const double tw = weight_type == 0 ? (1.0 - mw - (double)wlb) * lat_.dis_max - mpx : 1e300;
        if ((lat_.in[c[0]] & lat_.in[c[1]] & lat_.in[c[2]] & lat_.in[c[3]]) == 1 && tw > 0 && far2 < tw * tw * (1.0 + 1e-9)) {
            const double dfar = std::sqrt(far2) + mpx;
            double w = 1.0 - dfar * lat_.inv_dis_max;
            if (weight_type != 0) w = w > 0 ? w * w : 0.0;
            w -= mw;
            if (w > 2e-5) *wmin = (float)w;
        }
    }
    if (!want_out || !(wlb > 2e-5f)) return false;                // nothing known about the stored weights (or a fresh tile): in
    // T: weight(T - mpx) + mw == wlb
    double g = (double)wlb - mw;
    if (weight_type != 0) g = std::sqrt(g);
    const double T = mpx + lat_.dis_max * (1.0 - g), T2 = T * T;
    if (d2[0] <= T2 || d2[1] <= T2 || d2[2] <= T2 || d2[3] <= T2) return false;      // a corner within T
    bool pos = true, neg = true; double dnear2 = 1e300;
    for (int i = 0; i < 4; i++) {
        const size_t a = c[i], b = c[(i + 1) & 3];
        const double ex = lat_.sx[b] - lat_.sx[a], ey = lat_.sy[b] - lat_.sy[a], px = lat_.xc - lat_.sx[a], py = lat_.yc - lat_.sy[a];
        const double cr = ex * py - ey * px;
        pos = pos && cr >= 0; neg = neg && cr <= 0;
        const double e2 = ex * ex + ey * ey, dot = px * ex + py * ey;
        // squared distance from c to the segment: the end points are the corners (known to lie beyond T), the foot of the perpendicular counts
        // only when it falls inside the segment
        if (dot > 0 && dot < e2) dnear2 = std::min(dnear2, cr * cr / e2);
    }
    if (pos || neg) return false;                               // the centre lies inside Q: the keyframe's best weights are here
    return dnear2 > T2;
}

// Retirement without an event per frame (see the header): count the submission, drop a marker now and then.
bool FusionMap::submitted()
{
    work_no_++;
    if (work_no_ % kMarkEvery == 0) {
        HIP_OK(hipEventRecord(mark_ev_[mark_next_], stream_));
        mark_no_[mark_next_] = work_no_;
        mark_next_ = (mark_next_ + 1) % kMarks;
    }
    return true;
}

// block until submission `no` on stream_ has completed
bool FusionMap::wait_for(unsigned long long no)
{
    if (no <= synced_no_) return true;
    int best = -1;
    for (int i = 0; i < kMarks; i++)
        if (mark_no_[i] >= no && (best < 0 || mark_no_[i] < mark_no_[best])) best = i;
    if (best < 0) {                                             // nothing recorded past it yet: mark now
        HIP_OK(hipEventRecord(mark_ev_[mark_next_], stream_));
        mark_no_[mark_next_] = work_no_;
        best = mark_next_;
        mark_next_ = (mark_next_ + 1) % kMarks;
    }
    HIP_OK(hipEventSynchronize(mark_ev_[best]));
    synced_no_ = std::max(synced_no_, mark_no_[best]);
    return true;
}

// One pipelined launch (kernels.hip, k_levels): level 0 of `cur` (if any) and level s of pipe_[s] for every
// pending frame.  A job at level s reads GW_s from the buffer set the previous launch wrote and writes
// GW_{s+1} into this launch's set; stream order between launches is the only synchronisation.
bool FusionMap::launch_pipeline(const PipeFrame* cur, const WarpArgs* wa, const uint8_t* src)
{
    const int L = band_num_;
    const int par = (int)(launch_seq_ & 1);
    DevBuf* out = par ? gw2_ : gw_;
    DevBuf* in  = par ? gw_ : gw2_;
    LevelLaunch jobs[kMaxLevels];
    int n = 0;
    double bytes = 0, bytes_run = 0;
    auto add = [&](const PipeFrame& fr, int i) {
        const bool top = (i + 1 == L);
        LevelLaunch& q = jobs[n++];
        q.level = i; q.rows = fr.crows >> i; q.cols = fr.ccols >> i;
        q.cx0 = fr.C[i].x0; q.cy0 = fr.C[i].y0; q.cx1 = fr.C[i].x1; q.cy1 = fr.C[i].y1;
        q.tiles_x = fr.tx; q.top_select = top; q.write_next = !top; q.from_warp = (i == 0);
        q.gw_in = i == 0 ? nullptr : in[i].p; q.gw_out = top ? nullptr : out[i + 1].p;
        q.table = (const uint64_t*)table_dev_[fr.ring].p;
        q.table_args = i == 0 ? fr.table_args : nullptr; q.table_n = i == 0 ? fr.table_n : 0;
        q.nrect = fr.nrect[i];
        for (int k = 0; k < fr.nrect[i]; k++) q.rect[k] = fr.rect[i][k];
        q.need_bits = i > 0 && fr.need_n[i] > 0 ? fr.need_bits[i] : nullptr; q.need_n = i > 0 ? fr.need_n[i] : 0;
        bytes += fr.bytes[i]; bytes_run += fr.bytes_run[i];
    };
    // level 0 first: the short upper-level blocks come last and fill the tail of the grid
    if (cur) add(*cur, 0);
    static const bool no_upper = exp_env("PF_NO_UPPER") != nullptr;           // experiments library (timing only, wrong tiles): what the upper-level jobs add to a launch
    for (int s = 1; s < L; s++) if (pipe_[s].valid && !no_upper) add(pipe_[s], s);
    if (n) {
        prof_begin(cur ? K_LEVEL0 : K_LEVEL, bytes, stream_, bytes_run);
        launch_levels(stream_, lay_, jobs, n, wa, src);
        prof_end();
        HIP_OK(hipGetLastError());
    }
    // the frame whose last level just ran (this frame itself when L == 1) no longer needs its tile table
    const PipeFrame* done = L >= 2 ? (pipe_[L - 1].valid ? &pipe_[L - 1] : nullptr) : cur;
    if (!submitted()) return false;
    if (done) table_release_[done->ring] = work_no_;
    for (int s = L - 1; s >= 2; s--) pipe_[s] = pipe_[s - 1];
    if (L >= 2) { if (cur) pipe_[1] = *cur; else pipe_[1].valid = false; }
    launch_seq_++;
    return true;
}

// Everything that writes tile slots is ordered before what the caller enqueues on stream_ next (blend, strip pack, tile
// export).  Pipelined path: the pending upper-level launches go onto stream_ itself.  fused = 0/2/3 run pyramid levels on
// streams of their own: wait for those.
bool FusionMap::settle()
{
    if (opt_.fused == 1 || single_band_) return flush_pipeline();
    return sync_all() == hipSuccess;
}

// run the upper levels still pending for the frames fed so far (at most L-1 small launches)
bool FusionMap::flush_pipeline()
{
    if (flushing_) return true;
    flushing_ = true;
    bool ok = true;
    for (;;) {
        bool any = false;
        for (int s = 1; s < kMaxLevels; s++) any = any || pipe_[s].valid;
        if (!any) break;
        if (!launch_pipeline(nullptr, nullptr, nullptr)) { ok = false; break; }
    }
    flushing_ = false;
    return ok;
}

// ------------------------------------------------------------ tile access
bool FusionMap::grid(int dims[4], double geo[6])
{
    std::lock_guard<std::mutex> l(mu_);
    if (!valid_) return false;
    dims[0] = w_; dims[1] = h_; dims[2] = off_x_; dims[3] = off_y_;
    geo[0] = min_[0]; geo[1] = min_[1]; geo[2] = max_[0]; geo[3] = max_[1]; geo[4] = ele_size_; geo[5] = length_pixel_;
    return true;
}

bool FusionMap::map_update_inputs(int ix, int iy, double plane7[7], double mn[2], double* ele, int* x, int* y)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!valid_) return false;
    Tile* t = store_.find(ix, iy);
    if (!t || t->fresh) return false;                            // .cpp:717-718: no pyramid yet
    const int dx = ix - off_x_, dy = iy - off_y_;                // dense index of the draw() loop
    if (dx < 0 || dy < 0 || dx >= w_ || dy >= h_) return false;
    if (opt_.high_quality_show && !single_band_ && (dx == 0 || dy == 0 || dx == w_ - 1 || dy == h_ - 1)) return false;   // inborder
    std::memcpy(plane7, plane_.t, 24); std::memcpy(plane7 + 3, plane_.q, 32);
    mn[0] = min_[0]; mn[1] = min_[1]; *ele = ele_size_; *x = dx; *y = dy;
    return true;
}

int FusionMap::tile_count()
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    int n = 0;
    store_.for_each([&](int, int, Tile& t) { if (!t.fresh) n++; });
    return n;
}

int FusionMap::tile_coords(int* xy, int cap)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    std::vector<std::pair<int, int>> v;
    store_.for_each([&](int ix, int iy, Tile& t) { if (!t.fresh) v.push_back({ iy, ix }); });
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) { xy[2 * i] = v[i].second; xy[2 * i + 1] = v[i].first; }
    return (int)v.size();
}

bool FusionMap::get_tile_bgra(int ix, int iy, uint8_t* bgra)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !single_band_ || !set_device()) return false;
    Tile* t = store_.find(ix, iy);
    if (!t || t->fresh) return false;
    HIP_OK(sync_all());
    HIP_OK(hipMemcpy(bgra, t->base, (size_t)kElePixels * kElePixels * 4, hipMemcpyDeviceToHost));
    return true;
}

bool FusionMap::get_tile_level(int ix, int iy, int level, void* lap, float* w)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || single_band_ || !set_device()) return false;
    Tile* t = store_.find(ix, iy);
    if (!t || t->fresh || level < 0 || level > band_num_) return false;
    HIP_OK(sync_all());
    const size_t n = (size_t)(kElePixels >> level) * (kElePixels >> level);
    if (lap) HIP_OK(hipMemcpy(lap, t->base + lay_.lap_off[level], n * 3 * (lay_.f32 ? 4 : 2), hipMemcpyDeviceToHost));
    if (w) HIP_OK(hipMemcpy(w, t->base + lay_.w_off[level], n * 4, hipMemcpyDeviceToHost));
    return true;
}

bool FusionMap::halo_pack(int ix, int iy, int dx, int dy, void* dev_out)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !set_device()) return false;
    Tile* t = store_.find(ix, iy);
    if (!t || t->fresh) return false;
    launch_halo_pack(stream_, lay_, t->base, dx, dy, dev_out);
    HIP_OK(sync_all());
    return true;
}

bool FusionMap::tile_export(int ix, int iy, void* dev_out)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !set_device()) return false;
    Tile* t = store_.find(ix, iy);
    if (!t || t->fresh) return false;
    HIP_OK(hipMemcpyAsync(dev_out, t->base, lay_.slot_bytes, hipMemcpyDeviceToDevice, stream_));
    HIP_OK(sync_all());
    return true;
}

bool FusionMap::tile_import(int ix, int iy, const void* dev_in)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !valid_ || !set_device()) return false;
    Tile* t = store_.get_or_create(ix, iy);
    if (!t) return false;
    HIP_OK(hipMemcpyAsync(t->base, dev_in, lay_.slot_bytes, hipMemcpyDeviceToDevice, stream_));
    HIP_OK(sync_all());
    t->fresh = false; t->changed = true;
    for (float& w : t->wlb) w = -1.f;                           // imported pixels: nothing known about their weights
    return true;
}

// ------------------------------------------------------- seam exchange support (dist.cpp)
void FusionMap::list_tiles(std::vector<TileRec>& out)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    out.clear();
    store_.for_each([&](int ix, int iy, Tile& t) { if (!t.fresh) out.push_back({ ix, iy, t.changed ? 1 : 0 }); });
    std::sort(out.begin(), out.end(), [](const TileRec& a, const TileRec& b) { return a.iy != b.iy ? a.iy < b.iy : a.ix < b.ix; });
}

// every strip set of one exchange: one descriptor upload, one launch, no sync (the caller orders its transport after stream_)
bool FusionMap::pack_strips(const std::vector<StripReq>& reqs, void* dev_out)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !set_device()) return false;
    if (reqs.empty()) return true;
    if (!settle()) return false;
    std::vector<StripDesc> d(reqs.size());
    for (size_t i = 0; i < reqs.size(); i++) {
        Tile* t = store_.find(reqs[i].ix, reqs[i].iy);
        if (!t || t->fresh) { set_error("pack_strips: tile not held by this rank"); return false; }
        d[i] = { t->base, reqs[i].dx, reqs[i].dy, reqs[i].out_off };
    }
    if (!strip_desc_.reserve(d.size() * sizeof(StripDesc))) return false;
    HIP_OK(hipMemcpyAsync(strip_desc_.p, d.data(), d.size() * sizeof(StripDesc), hipMemcpyHostToDevice, stream_));
    launch_halo_pack_batch(stream_, lay_, (const StripDesc*)strip_desc_.p, (int)d.size(), dev_out);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(stream_));       // d (pageable) is read by the async copy: one sync per exchange, not per strip
    return true;
}

// whole tiles (pyramids + weights) of this rank, back to back in dev_out, for save()'s gather
bool FusionMap::export_tiles(const std::vector<std::pair<int, int>>& tiles, void* dev_out)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !set_device()) return false;
    if (!settle()) return false;
    for (size_t i = 0; i < tiles.size(); i++) {
        Tile* t = store_.find(tiles[i].first, tiles[i].second);
        if (!t || t->fresh) { set_error("export_tiles: tile not held by this rank"); return false; }
        HIP_OK(hipMemcpyAsync((char*)dev_out + i * lay_.slot_bytes, t->base, lay_.slot_bytes, hipMemcpyDeviceToDevice, stream_));
    }
    HIP_OK(hipStreamSynchronize(stream_));
    return true;
}

// the changed tiles among `tiles` blended with remote strips; clears their Ischanged flags (draw(), .cpp:705-742)
bool FusionMap::blend_tiles(const std::vector<std::pair<int, int>>& tiles, const void* const* halo9, uint8_t* bgr)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !set_device() || single_band_) return false;
    if (!blend_batch(tiles, halo9, nullptr, bgr)) return false;
    for (auto& t : tiles) { Tile* q = store_.find(t.first, t.second); if (q) q->changed = false; }
    return true;
}

// ------------------------------------------------------------------ output side
// Results leave HBM through a ring of two pinned staging slots: the device-to-host copy of piece k+1 (copy_stream_) runs while
// the host moves piece k from its slot into the caller's (pageable) buffer on a few threads.  A blocking hipMemcpy into
// pageable memory -- what rounds 1-5 did -- moves 12 GB/s on the boxes used; a caller who hands over pinned memory
// (pf_host_alloc) gets the copy straight into it.
static void copy_threads(void* dst, const void* src, size_t bytes, int nthreads)
{
    if (nthreads <= 1 || bytes < ((size_t)4 << 20)) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t part = ((bytes / nthreads) + 4095) & ~(size_t)4095;
    for (int i = 1; i < nthreads; i++) {
        const size_t o = part * i;
        if (o >= bytes) break;
        th.emplace_back([=] { std::memcpy((char*)dst + o, (const char*)src + o, std::min(part, bytes - o)); });
    }
    std::memcpy(dst, src, std::min(part, bytes));
    for (auto& t : th) t.join();
}

static bool is_pinned_host(const void* p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

bool FusionMap::out_ring_init()
{
    if (out_ring_ok_) return true;
    // (a call that failed half way is finished by the next one: every piece is made once)
    for (int i = 0; i < 2; i++) {
        if (!out_pin_[i]) HIP_OK(hipHostMalloc((void**)&out_pin_[i], kOutSlot, hipHostMallocDefault));
        if (!out_copied_[i]) HIP_OK(hipEventCreateWithFlags(&out_copied_[i], hipEventDisableTiming));
    }
    if (!out_ready_) HIP_OK(hipEventCreateWithFlags(&out_ready_, hipEventDisableTiming));
    if (!copy_stream_) HIP_OK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    out_ring_ok_ = true;
    const unsigned hc = std::thread::hardware_concurrency();
    out_threads_ = std::getenv("PF_COPY_THREADS") ? std::atoi(std::getenv("PF_COPY_THREADS")) : (int)std::min(8u, std::max(1u, hc / 2));
    return true;
}

// `pieces` = (host destination, device source, bytes) in the order they become ready; everything they read has been queued on
// stream_ before the call.  Pieces are cut to the slot size.
bool FusionMap::download(const std::vector<OutPiece>& pieces)
{
    if (pieces.empty()) return true;
    if (!out_ring_init()) return false;
    HIP_OK(hipEventRecord(out_ready_, stream_));
    HIP_OK(hipStreamWaitEvent(copy_stream_, out_ready_, 0));
    struct Part { char* dst; const char* src; size_t n; bool direct; };
    std::vector<Part> parts;
    for (auto& pc : pieces) {
        const bool direct = is_pinned_host(pc.dst);
        for (size_t o = 0; o < pc.bytes; o += kOutSlot) parts.push_back({ (char*)pc.dst + o, (const char*)pc.src + o, std::min(kOutSlot, pc.bytes - o), direct });
    }
    int k = 0;                                  // staged parts issued so far
    long prev = -1; int prev_slot = 0;          // staged part whose bytes still sit in its slot
    auto drain = [&]() -> bool {
        if (prev < 0) return true;
        HIP_OK(hipEventSynchronize(out_copied_[prev_slot]));
        copy_threads(parts[prev].dst, out_pin_[prev_slot], parts[prev].n, out_threads_);
        prev = -1;
        return true;
    };
    for (size_t i = 0; i < parts.size(); i++) {
        const Part& pt = parts[i];
        if (pt.direct) { HIP_OK(hipMemcpyAsync(pt.dst, pt.src, pt.n, hipMemcpyDeviceToHost, copy_stream_)); continue; }
        const int slot = k++ & 1;               // its previous tenant (two staged parts ago) has been drained: drain() runs once per staged part
        HIP_OK(hipMemcpyAsync(out_pin_[slot], pt.src, pt.n, hipMemcpyDeviceToHost, copy_stream_));
        HIP_OK(hipEventRecord(out_copied_[slot], copy_stream_));
        if (!drain()) return false;             // the part before this one, while this one is on the wire
        prev = (long)i; prev_slot = slot;
    }
    if (!drain()) return false;
    HIP_OK(hipStreamSynchronize(copy_stream_));
    return true;
}

// ------------------------------------------------------------------ blend
// Ele::blend (.cpp:77-146) for a list of tiles.  halo9 (nullptr, or 9 device pointers per tile) substitutes packed strip
// sets for neighbours held by other shards.  Results land in tile order (a tile that does not exist leaves the caller's
// bytes alone).  One launch of the fused collapse kernel per kBlendLaunch tiles, "blend with neighbours" and "blend by self"
// tiles side by side in it; the pixels come back through download().
bool FusionMap::blend_batch(const std::vector<std::pair<int, int>>& tiles, const void* const* halo9, void* raw_host, uint8_t* bgr_host)
{
    Section sec(this, T_UPDATE_TEXTURE);
#if PF_EXPERIMENTS
    static const bool per_level = exp_env("PF_BLEND_PER_LEVEL") != nullptr;
    if (per_level) return blend_batch_per_level(tiles, halo9, raw_host, bgr_host);
#endif
    const int nl = band_num_ + 1;
    const size_t es = lay_.f32 ? 4 : 2, px = 3 * es;
    const size_t tile_px = (size_t)kElePixels * kElePixels;
    if (!settle()) return false;                  // the upper levels of the last frames are still pending: run them first (stream order)
    // algorithmic bytes of one blended tile: its own Laplacians and level-0 weights read, the result written, plus the ring of
    // neighbour pixels the crop depends on at levels >= 1 (pyrUp reaches one pixel per level: collapse_fused.hip)
    double tile_bytes[2] = { 0, 0 };
    for (int nb = 0; nb < 2; nb++) {
        double b = (double)tile_px * 4 + (raw_host ? (double)tile_px * px : 0) + (bgr_host ? (double)tile_px * 3 : 0);
        int lo = 0, hi = kElePixels - 1;
        for (int i = 0; i < nl; i++) {
            const int ts = kElePixels >> i;
            if (i > 0) { lo = (lo - 1) >> 1; hi = (hi >> 1) + 1; }            // relative to the tile's own square
            const int bd = nb ? 1 << (nl - 1 - i) : 0, l = std::max(lo, -bd), h = std::min(hi, ts - 1 + bd);
            b += (double)(h - l + 1) * (h - l + 1) * px;
            (void)ts;
        }
        tile_bytes[nb] = b;
    }
    constexpr size_t kBlendLaunch = 4096;         // tiles per launch: bounds the result buffers in HBM (BGR8: 805 MB; fp32 raw: 3.2 GB)
    for (size_t c0 = 0; c0 < tiles.size(); c0 += kBlendLaunch) {
        const size_t cn = std::min(kBlendLaunch, tiles.size() - c0);
        std::vector<BlendJob> jobs; jobs.reserve(cn);
        double bytes = 0;
        for (size_t t = c0; t < c0 + cn; t++) {
            Tile* self = store_.find(tiles[t].first, tiles[t].second);
            if (!self || self->fresh) continue;
            const void* const* halo = halo9 ? halo9 + 9 * t : nullptr;
            BlendJob jb{}; bool all = opt_.high_quality_show != 0;
            for (int dy = -1; dy <= 1 && all; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int j = 3 * (dy + 1) + dx + 1;
                    Tile* n = store_.find(tiles[t].first + dx, tiles[t].second + dy);
                    if (n && !n->fresh) jb.src[j] = (uint64_t)(uintptr_t)n->base;
                    else if (halo && halo[j]) { jb.src[j] = (uint64_t)(uintptr_t)halo[j]; jb.strip_mask |= 1u << j; }
                    else { all = false; break; }
                }
            if (!all) { for (auto& q : jb.src) q = 0; jb.strip_mask = 0; }
            jb.src[4] = (uint64_t)(uintptr_t)self->base;
            jb.border = all ? 1 : 0; jb.out = (int)(t - c0);
            jobs.push_back(jb);
            bytes += tile_bytes[jb.border];
        }
        if (jobs.empty()) continue;
        if (raw_host && !blend_out_raw_.reserve(tile_px * px * cn)) return false;
        if (bgr_host && !blend_out_bgr_.reserve(tile_px * 3 * cn)) return false;
        if (!blend_src_.reserve(jobs.size() * sizeof(BlendJob))) return false;
        HIP_OK(hipMemcpyAsync(blend_src_.p, jobs.data(), jobs.size() * sizeof(BlendJob), hipMemcpyHostToDevice, stream_));
        prof_begin(K_BLEND_FUSED, bytes);
        launch_blend_fused(stream_, lay_, (const BlendJob*)blend_src_.p, (int)jobs.size(), raw_host ? blend_out_raw_.p : nullptr,
                           bgr_host ? (uint8_t*)blend_out_bgr_.p : nullptr);
        prof_end();
        HIP_OK(hipGetLastError());
        // runs of tiles that exist, each one piece per output
        std::vector<OutPiece> pieces;
        for (size_t a = 0; a < jobs.size();) {
            size_t b = a + 1;
            while (b < jobs.size() && jobs[b].out == jobs[b - 1].out + 1) b++;
            const size_t r0 = (size_t)jobs[a].out, n = b - a;
            if (bgr_host) pieces.push_back({ bgr_host + (c0 + r0) * tile_px * 3, (char*)blend_out_bgr_.p + r0 * tile_px * 3, n * tile_px * 3 });
            if (raw_host) pieces.push_back({ (char*)raw_host + (c0 + r0) * tile_px * px, (char*)blend_out_raw_.p + r0 * tile_px * px, n * tile_px * px });
            a = b;
        }
        if (!download(pieces)) return false;      // also orders the next launch's job upload and result buffers after this one's readers
        HIP_OK(hipStreamSynchronize(stream_));    // jobs (pageable) were read by an async copy
    }
    return true;
}

#if PF_EXPERIMENTS
// The per-level form of rounds 1-5 (A/B partner and second opinion of the fused kernel): padded squares per level in HBM,
// one launch per reference op, one blocking device-to-host copy per 128-tile chunk.
bool FusionMap::blend_batch_per_level(const std::vector<std::pair<int, int>>& tiles, const void* const* halo9, void* raw_host, uint8_t* bgr_host)
{
    const int nl = band_num_ + 1, L = band_num_;
    const size_t es = lay_.f32 ? 4 : 2, px = 3 * es;
    const size_t tile_px = (size_t)kElePixels * kElePixels;
    constexpr size_t kChunk = 128;
    if (!settle()) return false;                  // the upper levels of the last frames are still pending: run them first (stream order)
    for (size_t c0 = 0; c0 < tiles.size(); c0 += kChunk) {
        const size_t cn = std::min(kChunk, tiles.size() - c0);
        if (raw_host && !blend_out_raw_.reserve(tile_px * px * cn)) return false;
        if (bgr_host && !blend_out_bgr_.reserve(tile_px * 3 * cn)) return false;
        std::vector<char> present(cn, 0);
        // group the chunk's tiles by mode: full 3x3 available (border = 1<<(nl-1-i)) or self (border 0)
        for (int mode = 0; mode < 2; mode++) {
            std::vector<BlendSrc> srcs;
            std::vector<int> idx;
            for (size_t t = c0; t < c0 + cn; t++) {
                Tile* self = store_.find(tiles[t].first, tiles[t].second);
                if (!self || self->fresh) continue;
                const void* const* halo = halo9 ? halo9 + 9 * t : nullptr;
                BlendSrc nb[9]; bool all = opt_.high_quality_show != 0;
                for (int dy = -1; dy <= 1 && all; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        const int j = 3 * (dy + 1) + dx + 1;
                        Tile* n = store_.find(tiles[t].first + dx, tiles[t].second + dy);
                        if (n && !n->fresh) nb[j] = { n->base, 0 };
                        else if (halo && halo[j]) nb[j] = { halo[j], 1 };
                        else { all = false; break; }
                    }
                if ((all ? 0 : 1) != mode) continue;
                if (!all) { for (auto& q : nb) q = { nullptr, 0 }; nb[4] = { self->base, 0 }; }
                srcs.insert(srcs.end(), nb, nb + 9);
                idx.push_back((int)(t - c0));
                present[t - c0] = 1;
            }
            const int batch = (int)idx.size();
            if (!batch) continue;
            const int b0 = mode == 0 ? (1 << (nl - 1)) : 0;
            const size_t src_bytes = srcs.size() * sizeof(BlendSrc), idx_off = (src_bytes + 255) / 256 * 256;
            if (!blend_src_.reserve(idx_off + idx.size() * sizeof(int))) return false;
            HIP_OK(hipMemcpyAsync(blend_src_.p, srcs.data(), src_bytes, hipMemcpyHostToDevice, stream_));
            HIP_OK(hipMemcpyAsync((char*)blend_src_.p + idx_off, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
            size_t stride[kMaxLevels];
            for (int i = 0; i < nl; i++) {
                const int b = mode == 0 ? (1 << (nl - 1 - i)) : 0, side = (kElePixels >> i) + 2 * b;
                stride[i] = (((size_t)side * side * px + 255) / 256) * 256;
                if (!blend_lv_[i].reserve(stride[i] * batch)) return false;
            }
            const BlendSrc* dsrc = (const BlendSrc*)blend_src_.p;
            for (int i = 0; i < nl; i++) {
                const int b = mode == 0 ? (1 << (nl - 1 - i)) : 0, side = (kElePixels >> i) + 2 * b;
                prof_begin(K_BLEND_GATHER, (double)side * side * px * 2 * batch);
                launch_blend_gather(stream_, lay_, i, b, dsrc, blend_lv_[i].p, stride[i], batch);
                prof_end();
            }
            for (int i = L; i > 0; i--) {
                const int b = mode == 0 ? (1 << (nl - i)) : 0, side = (kElePixels >> (i - 1)) + 2 * b;
                prof_begin(K_COLLAPSE, (double)side * side * px * 2.25 * batch);
                launch_collapse(stream_, lay_.f32, blend_lv_[i - 1].p, stride[i - 1], blend_lv_[i].p, stride[i], side, side, batch);
                prof_end();
            }
            prof_begin(K_BLEND_FINISH, (double)tile_px * (px + 4 + 3) * batch);
            launch_blend_finish(stream_, lay_, blend_lv_[0].p, stride[0], b0, dsrc, raw_host ? blend_out_raw_.p : nullptr,
                                bgr_host ? (uint8_t*)blend_out_bgr_.p : nullptr, batch, (const int*)((char*)blend_src_.p + idx_off));
            prof_end();
            HIP_OK(hipStreamSynchronize(stream_));              // srcs / idx are reused by the other mode
        }
        // one copy per chunk (tiles that do not exist keep the caller's bytes: copy the runs that do)
        size_t r0 = 0;
        while (r0 < cn) {
            while (r0 < cn && !present[r0]) r0++;
            size_t r1 = r0;
            while (r1 < cn && present[r1]) r1++;
            if (r1 > r0) {
                if (raw_host) HIP_OK(hipMemcpy((char*)raw_host + (c0 + r0) * tile_px * px, (char*)blend_out_raw_.p + r0 * tile_px * px, (r1 - r0) * tile_px * px, hipMemcpyDeviceToHost));
                if (bgr_host) HIP_OK(hipMemcpy(bgr_host + (c0 + r0) * tile_px * 3, (char*)blend_out_bgr_.p + r0 * tile_px * 3, (r1 - r0) * tile_px * 3, hipMemcpyDeviceToHost));
            }
            r0 = r1;
        }
    }
    return true;
}

#endif

bool FusionMap::blend_tile(int ix, int iy, void* raw, uint8_t* bgr, const void* const* halo)
{
    if (single_band_) {          // the Map2DCPU tile is displayable as is (glTexImage2D GL_BGRA, Map2DCPU.cpp:497-503)
        std::vector<uint8_t> px((size_t)kElePixels * kElePixels * 4);
        if (!get_tile_bgra(ix, iy, px.data())) return false;
        if (raw) std::memcpy(raw, px.data(), px.size());
        if (bgr) for (size_t i = 0; i < (size_t)kElePixels * kElePixels; i++) { bgr[3 * i] = px[4 * i]; bgr[3 * i + 1] = px[4 * i + 1]; bgr[3 * i + 2] = px[4 * i + 2]; }
        return true;
    }
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !set_device()) return false;
    Tile* t = store_.find(ix, iy);
    if (!t || t->fresh) return false;
    std::vector<std::pair<int, int>> one{ { ix, iy } };
    return blend_batch(one, halo, raw, bgr);
}

bool FusionMap::blend_list(const std::vector<std::pair<int, int>>& tiles, uint8_t* bgr)
{
    if (single_band_) {          // the Map2DCPU tile is displayable as is (see blend_tile): a tile that does not exist keeps the buffer's bytes
        for (size_t i = 0; i < tiles.size(); i++) (void)blend_tile(tiles[i].first, tiles[i].second, nullptr, bgr + i * (size_t)kElePixels * kElePixels * 3, nullptr);
        return init_ok_;
    }
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !set_device()) return false;
    return blend_batch(tiles, nullptr, nullptr, bgr);
}

// the draw() loop's texture refresh (.cpp:705-742) without GL
int FusionMap::blend_changed(int* xy, uint8_t* bgr, int cap)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !set_device()) return 0;
    std::vector<std::pair<int, int>> v;
    store_.for_each([&](int ix, int iy, Tile& t) { if (!t.fresh && t.changed) v.push_back({ iy, ix }); });
    std::sort(v.begin(), v.end());
    if ((int)v.size() > cap) v.resize(cap);
    std::vector<std::pair<int, int>> tiles;
    for (auto& p : v) tiles.push_back({ p.second, p.first });
    if (tiles.empty()) return 0;
    // bound the scratch: 64 tiles per batch
    const size_t tile_px = (size_t)kElePixels * kElePixels * 3;
    if (single_band_) {
        if (sync_all() != hipSuccess) return 0;
        std::vector<uint8_t> px((size_t)kElePixels * kElePixels * 4);
        for (size_t i = 0; i < tiles.size(); i++) {
            Tile* t = store_.find(tiles[i].first, tiles[i].second);
            if (hipMemcpy(px.data(), t->base, px.size(), hipMemcpyDeviceToHost) != hipSuccess) return 0;
            uint8_t* d = bgr + i * tile_px;
            for (size_t k = 0; k < (size_t)kElePixels * kElePixels; k++) { d[3 * k] = px[4 * k]; d[3 * k + 1] = px[4 * k + 1]; d[3 * k + 2] = px[4 * k + 2]; }
            xy[2 * i] = tiles[i].first; xy[2 * i + 1] = tiles[i].second;
            t->changed = false;
        }
        return (int)tiles.size();
    }
    if (sync_all() != hipSuccess || !blend_batch(tiles, nullptr, nullptr, bgr)) return 0;
    for (size_t i = 0; i < tiles.size(); i++) {
        xy[2 * i] = tiles[i].first; xy[2 * i + 1] = tiles[i].second;
        store_.find(tiles[i].first, tiles[i].second)->changed = false;
    }
    return (int)tiles.size();
}

// ------------------------------------------------------------------- save
// MultiBandMap2DCPU::save (.cpp:779-847): paste all tiles per level, collapse
// the whole mosaic once, 8U, background where level-0 weight is 0.
bool FusionMap::save_to_memory(uint8_t* bgr, int* rows, int* cols, int* tx0, int* ty0, const std::vector<ForeignTile>* foreign)
{
    std::lock_guard<std::mutex> l(mu_); (void)drain();
    if (!init_ok_ || !valid_ || !set_device()) return false;
    if (w_ == 0 || h_ == 0) return false;
    Section sec(this, T_SAVE);
    int mnx = 1000000, mny = 1000000, mxx = -1000000, mxy = -1000000, cnt = 0;
    store_.for_each([&](int ix, int iy, Tile& t) {
        if (t.fresh) return;
        cnt++; mnx = std::min(mnx, ix); mny = std::min(mny, iy); mxx = std::max(mxx, ix); mxy = std::max(mxy, iy);
    });
    // tiles of other ranks gathered for this save (dist.cpp): they take part in the mosaic without entering the store
    if (foreign) for (auto& f : *foreign) { cnt++; mnx = std::min(mnx, f.ix); mny = std::min(mny, f.iy); mxx = std::max(mxx, f.ix); mxy = std::max(mxy, f.iy); }
    if (!cnt) return false;
    const int wx = mxx + 1 - mnx, wy = mxy + 1 - mny;
    *rows = wy * kElePixels; *cols = wx * kElePixels; *tx0 = mnx; *ty0 = mny;
    if (!bgr) return true;
    if (single_band_) {          // Map2DCPU::save (Map2DCPU.cpp:523-563): paste the tiles; holes are zero here
        HIP_OK(sync_all());
        std::memset(bgr, 0, (size_t)*rows * *cols * 3);
        std::vector<uint8_t> px((size_t)kElePixels * kElePixels * 4);
        bool ok = true;
        store_.for_each([&](int ix, int iy, Tile& t) {
            if (t.fresh || !ok) return;
            if (hipMemcpy(px.data(), t.base, px.size(), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; return; }
            for (int r = 0; r < kElePixels; r++) {
                uint8_t* d = bgr + (((size_t)(iy - mny) * kElePixels + r) * *cols + (size_t)(ix - mnx) * kElePixels) * 3;
                const uint8_t* sp = px.data() + (size_t)r * kElePixels * 4;
                for (int c = 0; c < kElePixels; c++) { d[3 * c] = sp[4 * c]; d[3 * c + 1] = sp[4 * c + 1]; d[3 * c + 2] = sp[4 * c + 2]; }
            }
        });
        return ok;
    }
    const int L = band_num_;
    const size_t es = lay_.f32 ? 4 : 2, px = 3 * es;
    std::vector<uint64_t> tab((size_t)wx * wy, 0);
    store_.for_each([&](int ix, int iy, Tile& t) { if (!t.fresh) tab[(size_t)(iy - mny) * wx + (ix - mnx)] = (uint64_t)(uintptr_t)t.base; });
    if (foreign) for (auto& f : *foreign) tab[(size_t)(f.iy - mny) * wx + (f.ix - mnx)] = (uint64_t)(uintptr_t)f.dev;
    HIP_OK(sync_all());
    if (!mosaic_table_.reserve(tab.size() * 8)) return false;
    HIP_OK(hipMemcpy(mosaic_table_.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    const size_t out_bytes = (size_t)*rows * *cols * 3;
    if (!blend_out_bgr_.reserve(out_bytes)) return false;
#if PF_EXPERIMENTS
    static const bool per_level = exp_env("PF_BLEND_PER_LEVEL") != nullptr;
    if (per_level) {                              // rounds 1-5: paste per level, one collapse launch per level, finish
        for (int i = 0; i <= L; i++) {
            const size_t n = (size_t)(*rows >> i) * (*cols >> i);
            if (!blend_lv_[i].reserve(n * px)) return false;
            prof_begin(K_MOSAIC_GATHER, (double)n * px * 2);
            launch_mosaic_gather(stream_, lay_, i, (const uint64_t*)mosaic_table_.p, wx, wy, blend_lv_[i].p);
            prof_end();
        }
        for (int i = L; i > 0; i--) {
            prof_begin(K_COLLAPSE, (double)(*rows >> (i - 1)) * (*cols >> (i - 1)) * px * 2.25);
            launch_collapse(stream_, lay_.f32, blend_lv_[i - 1].p, 0, blend_lv_[i].p, 0, *rows >> (i - 1), *cols >> (i - 1), 1);
            prof_end();
        }
        prof_begin(K_SAVE_FINISH, (double)*rows * *cols * (px + 4 + 3));
        launch_save_finish(stream_, lay_, blend_lv_[0].p, (const uint64_t*)mosaic_table_.p, wx, wy, opt_.bg_color, (uint8_t*)blend_out_bgr_.p);
        prof_end();
        HIP_OK(sync_all());
        HIP_OK(hipMemcpy(bgr, blend_out_bgr_.p, out_bytes, hipMemcpyDeviceToHost));
        return true;
    }
#endif
    // one launch: paste, collapse in LDS, 8U, background (collapse_fused.hip).  Algorithmic bytes: every tile's Laplacians and
    // level-0 weights read once, the mosaic written once.
    double P = 0; for (int i = 0; i <= L; i++) P += 1.0 / (double)(1 << (2 * i));
    prof_begin(K_SAVE_FUSED, (double)cnt * kElePixels * kElePixels * (P * px + 4) + (double)out_bytes);
    launch_save_fused(stream_, lay_, (const uint64_t*)mosaic_table_.p, wx, wy, opt_.bg_color, (uint8_t*)blend_out_bgr_.p);
    prof_end();
    HIP_OK(hipGetLastError());
    if (!download({ { bgr, blend_out_bgr_.p, out_bytes } })) return false;
    return true;
}

bool FusionMap::save(const char* filename)
{
    int rows, cols, tx0, ty0;
    if (!save_to_memory(nullptr, &rows, &cols, &tx0, &ty0)) return false;
    std::vector<uint8_t> img((size_t)rows * cols * 3);
    if (!save_to_memory(img.data(), &rows, &cols, &tx0, &ty0)) return false;
    if (!write_image_file(filename, img.data(), rows, cols)) return false;
    std::printf("Resolution:[%d %d]\n", cols, rows);
    return true;
}

}  // namespace pf
