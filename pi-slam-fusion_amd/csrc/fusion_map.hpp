// fusion_map.hpp -- host engine behind the Map2D boundary: the GPU counterpart of
// MultiBandMap2DCPU (Map2DFusion/MultiBandMap2DCPU.h:28-132).
#pragma once
#include "../../include/pifusion.h"
#include "geometry.hpp"
#include "kernels.hpp"
#include "dist_plan.hpp"
#include "env.hpp"
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace pf {

void set_error(const std::string& msg);
const char* last_error();

// spatial-hash owner of a tile; cells of shard_block tiles
int tile_owner(int shard_count, int shard_block, int ix, int iy);

// ---- tile store: spatial hash (ix,iy) -> slot in an HBM slab pool ----------
// replaces the dense vector<SPtr<Ele>> + spreadMap re-layout (.h:91, .cpp:561-604)
struct Tile {
    char* base = nullptr;   // slot base in HBM
    bool  fresh = true;     // pyr_laplace[i].empty() (.cpp:498): first write copies unconditionally
    bool  changed = false;  // Ele::Ischanged
    // A lower bound of every weight stored in this tile, all levels (render_frame's cull): each keyframe whose canvas holds the
    // tile raises it to the smallest weight that keyframe can have anywhere in the cell's pyramid support.  <= 0: nothing known.
    // Per cell of the cull, row-major: 4 x 4 cells of 64 x 64 pixels (PF_CULL_SUB=2: 2 x 2 quadrants in the first four)
    float wlb[16] = { -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f };
};

class TileStore {
public:
    ~TileStore() { clear(); }
    void  configure(size_t slot_bytes) { slot_bytes_ = slot_bytes; }
    Tile* find(int ix, int iy);
    Tile* get_or_create(int ix, int iy);          // nullptr on HBM exhaustion
    bool  reserve(size_t n_tiles, std::vector<std::pair<char*, size_t>>* fresh = nullptr);   // slabs for n more tiles now
    void  clear();
    size_t size() const { return map_.size(); }
    template <class F> void for_each(F f) { for (auto& kv : map_) f((int)(int32_t)(kv.first >> 32), (int)(int32_t)(kv.first & 0xffffffffu), kv.second); }
private:
    static uint64_t key(int ix, int iy) { return ((uint64_t)(uint32_t)ix << 32) | (uint32_t)iy; }
    std::unordered_map<uint64_t, Tile> map_;
    struct Chunk { char* p; size_t slots; };
    std::vector<Chunk> chunks_;                   // chunks_[cur_] is being filled, later ones are reserved ahead
    size_t slot_bytes_ = 0, cur_ = 0, next_in_chunk_ = 0;
    bool   add_chunk(size_t slots);
};

struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    bool   reserve(size_t bytes);     // grow-only; caller must have synchronised users
    void   release();
};

struct FrameSlot {
    uint8_t*   dev = nullptr;
    size_t     cap = 0;
    hipEvent_t consumed = nullptr;    // recorded on the compute stream after the last kernel reading it
    bool       pending = false;       // consumed has been recorded and not yet observed complete
    bool       queued = false;        // sits in the feed queue
};

struct QueuedFrame {
    int    slot;            // index into frame slots, -1 = geometry only
    const uint8_t* ext;     // externally owned device pointer (pf_feed_device) or nullptr
    long   step;
    int    rows, cols, cn;  // cn: 3 = BGR8, 4 = BGRA8
    Pose   pose;            // plane coordinates
    long long seq = -1;     // number of the feed() call that brought it (test hook: render_log)
};

// Named host sections, after pi::Timer (PIL/src/base/time/Timer.h:43-85: enter / leave, calls, min / max / mean per
// section) with the reference's own section names (MultiBandMap2DCPU.cpp:476,555,563,602,628-630,722,742); each section is
// also a roctx range when PF_ROCTX=1 (rocprofv3 --marker-trace shows them next to the kernels).
enum SectionId { T_FEED = 0, T_RENDER, T_APPLY, T_SPREAD, T_UPDATE_TEXTURE, T_SAVE, T_COUNT };
const char* section_name(int id);
struct SectionRec { long long n_calls = 0; double min_t = 0, max_t = 0, total_t = 0; };
void roctx_push(const char* name);
void roctx_pop();

class FusionMap {
public:
    FusionMap(int type, bool thread, const pf_options& opt);
    ~FusionMap();
    bool ok() const { return init_ok_; }

    bool prepare(const double plane[7], const double cam[6], int n, const pf_image* imgs, const double* poses7);
    // `produce` (img->data == nullptr): the frame's pixels are written into HBM by the caller's own work -- it is handed the frame's slot
    // and the stream that work has to be queued on (pf_feed_jpeg: the decoder's upload and kernels); the frame is then queued or rendered
    // exactly as a host frame that feed() uploaded
    typedef std::function<bool(void* dev, hipStream_t stream)> FrameProducer;
    bool feed(const pf_image* img, const double pose[7], bool device_ptr, const FrameProducer* produce = nullptr);
    unsigned queue_size();
    long read_back_last_frame(void* out, size_t cap);
    bool sync();
    bool save(const char* filename);
    struct ForeignTile { int ix, iy; const void* dev; };          // a tile slot image held outside the store (gathered for save)
    bool save_to_memory(uint8_t* bgr, int* rows, int* cols, int* tx0, int* ty0, const std::vector<ForeignTile>* foreign = nullptr);

    // seam exchange support (dist.cpp)
    using TileRec = pf::TileRec;        // dist_plan.hpp
    using StripReq = pf::StripReq;
    void list_tiles(std::vector<TileRec>& out);
    bool pack_strips(const std::vector<StripReq>& reqs, void* dev_out);
    bool export_tiles(const std::vector<std::pair<int, int>>& tiles, void* dev_out);
    bool blend_tiles(const std::vector<std::pair<int, int>>& tiles, const void* const* halo9, uint8_t* bgr);
    // frame distribution support (dist.cpp, pf_dist_feed): which ranks hold a tile of this keyframe's canvas (geometry
    // only; applies spreadMap exactly as feed() would, so a later feed of the same pose finds the grid as it left it),
    // a staging slot in HBM for the frame (filled from the host on the root, by the transport elsewhere), and the render
    // of a staged frame
    bool reject_mismatched_frame(const pf_image* desc);      // Map2D::feed's size / type rejection, identical on every rank
    bool frame_needs(const double pose7[7], std::vector<unsigned char>& rank_needs);
    int  stage_frame(const pf_image* desc, bool upload_host, void** dev, size_t* bytes);     // slot index, -1 on failure
    bool feed_staged(int slot, const pf_image* desc, const double pose7[7]);                 // slot < 0: geometry-only feed
    void release_staged(int slot) { std::lock_guard<std::mutex> q(qmu_); if (slot >= 0 && slot < (int)slots_.size()) slots_[slot].queued = false; }
    hipStream_t stream() const { return stream_; }
    int  device() const { return device_; }
    bool use_device() { return set_device(); }
    // draw()'s Fuse2Google gate and operands for tile (ix, iy) (MultiBandMap2DCPU.cpp:709-712, :730-735, :744): false when the tile
    // has no pyramid, or lies on the rim of the dense grid while HighQualityShow is on
    bool map_update_inputs(int ix, int iy, double plane7[7], double mn[2], double* ele, int* x, int* y);
    // the cull of render_frame (tiles in which a keyframe cannot win the select): see there
    bool cull_frame_ok(const double M[9], int crows, int ccols) const;
    // source positions of the canvas lattice points (-64 + 64 k, -64 + 64 m) the cells' dilated rectangles have their corners on
    void cull_lattice(const double M[9], int crows, int ccols, int cols, int rows, int dil, bool map_all);
    size_t lattice_point(int k, int m);
    bool cell_out(int k, int m, int span, int weight_type, float wlb, bool want_out, float* wmin);
    long long culled_tiles() { std::lock_guard<std::mutex> l(mu_); (void)drain(); return n_culled_tiles_; }
    void set_cull(bool on) { std::lock_guard<std::mutex> l(mu_); (void)drain(); cull_on_ = on; }       // default: on unless PF_CULL=0
    long long culled_cells() { std::lock_guard<std::mutex> l(mu_); (void)drain(); return n_culled_cells_; }
    // test hook: the feed() calls (0-based, counted since creation) whose keyframes renderFrame accepted, in render order -- with thread = true
    // and a queue that drops, the only way to tell an oracle which keyframes the map really holds; the newest 65536
    int  render_log(long long* out, int cap);
    double level0_exact_px() const { return px_level0_exact_; }
    bool high_quality() const { return opt_.high_quality_show != 0 && !single_band_; }

    int  num_levels() const { return band_num_ + 1; }
    int  pyramid_type() const { return single_band_ ? PF_8UC4 : (lay_.f32 ? PF_32FC3 : PF_16SC3); }
    bool grid(int dims[4], double geo[6]);
    int  tile_count();
    int  tile_coords(int* xy, int cap);
    bool get_tile_level(int ix, int iy, int level, void* lap, float* w);
    bool get_tile_bgra(int ix, int iy, uint8_t* bgra);
    bool blend_tile(int ix, int iy, void* raw, uint8_t* bgr, const void* const* halo);
    int  blend_changed(int* xy, uint8_t* bgr, int cap);
    bool blend_list(const std::vector<std::pair<int, int>>& tiles, uint8_t* bgr);      // pf_blend_tiles: Ischanged left alone
    size_t halo_bytes_for(int dx, int dy) const { return halo_bytes(lay_, dx, dy); }
    bool halo_pack(int ix, int iy, int dx, int dy, void* dev_out);
    size_t tile_bytes() const { return lay_.slot_bytes; }
    bool tile_export(int ix, int iy, void* dev_out);
    bool tile_import(int ix, int iy, const void* dev_in);

    void profile_enable(int mode);
    int  profile_read(int cap, const char** names, double* ms, long long* launches, double* bytes, double* bytes_run = nullptr);
    void profile_reset();
    void stats(long long* rendered, long long* rejected, long long* dropped);
    void render_stats(double out[4]);
    int  timer_read(int cap, const char** names, long long* calls, double* mean_s, double* min_s, double* max_s);
    void timer_reset();
    struct Section {            // scope guard: enter on construction, leave on destruction
        FusionMap* m; int id; double t0;
        Section(FusionMap* m_, int id_);
        ~Section();
    };
    bool reserve_tiles(long long n_tiles);
    const pf_options& options() const { return opt_; }

private:
    bool render_frame(const QueuedFrame& f);                       // .cpp:311-558
    bool spread_map(double xmin, double ymin, double xmax, double ymax);   // .cpp:561-604
    void worker();                                                 // .cpp:619-635
    int  acquire_slot(size_t bytes);
    bool upload(const pf_image* img, int slot);
    bool blend_batch(const std::vector<std::pair<int,int>>& tiles, const void* const* halo9, void* raw_host, uint8_t* bgr_host);
    // bytes: SURVEY 8d's algorithmic bytes of the launch for EVERY tile of the canvas; bytes_run (< 0: the same): those of the part of the
    // canvas its blocks actually process (the cull / a shard leave blocks out) -- the numerator of bench.py's roofline.frac
    void prof_begin(int id, double bytes, hipStream_t st = nullptr, double bytes_run = -1);
    bool prof_would(int id) const;                                 // will the next prof_begin(id) bracket its launch with events?
    hipError_t sync_all();
    void prof_end();
    void prof_harvest();
    bool set_device();

    pf_options opt_;
    int        band_num_ = 5;
    TileLayout lay_{};
    bool       thread_ = false, init_ok_ = false;
    bool       single_band_ = false;      // Map2DCPU semantics (TypeCPU / TypeGPU), one BGRA tile per cell
    DevBuf     w8_; int w8_rows_ = 0, w8_cols_ = 0;   // its weight byte plane
    DevBuf     wmap_; int wmap_rows_ = 0, wmap_cols_ = 0;   // multi-band: fp32 weight plane (weightImage)
    int        device_ = 0;
    hipStream_t stream_ = nullptr;

    // prepared state, guarded by mu_
    std::mutex mu_;
    bool   valid_ = false;
    Pose   plane_{}, plane_inv_{};
    bool cull_on_ = !(std::getenv("PF_CULL") && std::atoi(std::getenv("PF_CULL")) == 0);
    long long n_culled_tiles_ = 0;              // tiles left out of launches by the cull (diagnostics)
    long long n_culled_cells_ = 0;              // 64 x 64 cells of rendered tiles switched off by it
    // Margins of the cull's bounds (cell_out): source pixels added to / taken from a distance before it becomes a weight (the nearest-pixel
    // rounding of the weight gather, 0.71 px, and the float arithmetic of the kernels), and what is taken from / added to a weight (the
    // pyramid's own rounding).  PF_CULL_MARGIN_PX / PF_CULL_MARGIN_W (experiments library) override them for the sensitivity runs of tools/cull_soak.py
    // (profiles/r05_cull_margins.md: mismatches against the oracle per setting -- the safety factor, measured); defaults 2 px, 1e-5.
    double cull_margin_px_ = exp_env_double("PF_CULL_MARGIN_PX", 2.0);
    double cull_margin_w_ = exp_env_double("PF_CULL_MARGIN_W", 1e-5);
    int cull_sub_ = exp_env_int("PF_CULL_SUB", 4) == 2 ? 2 : 4;      // cells per tile edge (experiments library: 2 = quadrants)
    struct Lattice { bool all = false; int nx = 0, ny = 0, dil = 1, cols = 0, rows = 0; double xc = 0, yc = 0, dis_max = 1, inv_dis_max = 1, M[9] = {}; std::vector<double> sx, sy, d; std::vector<unsigned char> in; };
    Lattice lat_;                                   // of the keyframe being admitted / rendered
    Camera cam_{};
    double ele_size_ = 0, ele_size_inv_ = 0, length_pixel_ = 0, length_pixel_inv_ = 0;
    double min_[3]{}, max_[3]{};
    int    w_ = 0, h_ = 0, off_x_ = 0, off_y_ = 0;
    TileStore store_;

    // per-frame workspace (grow-only)
    DevBuf g_[kMaxLevels], wgt_[kMaxLevels], gw_[kMaxLevels], gw2_[kMaxLevels];
    hipStream_t lvl_stream_[kMaxLevels]{};          // fused pipeline: one stream per level ([0] aliases stream_)
    static constexpr int kTableRing = 64;           // tile tables in flight (see retire())
    static constexpr int kLvlRing = 16;
    hipEvent_t  lvl_ev_[kMaxLevels][kLvlRing]{};    // fused = 2/3: level i of the frame in ring slot k has run
    hipStream_t prof_stream_ = nullptr;
    static constexpr int kUpperStreams = 1;         // streams shared by pyramid levels >= 1
    // Tile tables live in device memory (table_dev_, a ring of kTableRing per-frame tables).  How a frame's table gets
    // there: inside the kernel arguments of the launch that carries the frame's level 0, which stores it to the ring slot
    // itself (pipelined path, tables of at most kArgTable entries: nothing sits in the stream between two launches and
    // no host memory is read in place); otherwise staged in pinned host memory and copied in the stream.
    uint64_t*  table_host_[kTableRing]{};           // pinned staging of the copy path
    bool       table_in_args_ = true;               // PF_TABLE_COPY=1 forces the copy path (A/B, tests)
    DevBuf     table_dev_[kTableRing];
    std::vector<uint64_t> table_tmp_;               // a frame's entries while they are being built
    std::vector<uint8_t>  block_bits_;              // level-0 blocks a shard runs (render_stats only)
    size_t     table_cap_ = 0;
    hipEvent_t table_ev_[kTableRing]{};             // fused = 2/3: last reader of the ring slot done (own stream)
    bool       table_pending_[kTableRing]{};
    // Everything else retires on stream_, and an event record between two dependent launches costs ~18 us of
    // stream time on this stack: instead of one event per frame, a ring slot remembers the number of the launch that
    // reads it last (work_no_ counts submissions on stream_), a marker event is recorded every kMarkEvery
    // submissions, and a slot is reused after waiting for a marker at or past its number.
    static constexpr int kMarkEvery = 32, kMarks = 4;           // kTableRing - kMaxLevels >= kMarkEvery
    unsigned long long work_no_ = 0, synced_no_ = 0;
    unsigned long long table_release_[kTableRing]{};
    hipEvent_t mark_ev_[kMarks]{};
    unsigned long long mark_no_[kMarks]{};
    int  mark_next_ = 0;
    bool submitted();                                           // call once per submission on stream_
    bool wait_for(unsigned long long no);
    unsigned long long frame_seq_ = 0;

    // pipelined level launches (opt_.fused == 1): pipe_[s] is the frame whose level s runs in the next launch
    struct Win { int x0, x1, y0, y1; };
    struct PipeFrame { bool valid = false; int ring = 0, tx = 0, crows = 0, ccols = 0; Win C[kMaxLevels]; double bytes[kMaxLevels], bytes_run[kMaxLevels];
                       int nrect[kMaxLevels]; BlockRect rect[kMaxLevels][kMaxRects];        // LevelLaunch::rect of each level
                       uint32_t need_bits[kMaxLevels][kNeedWords]; int need_n[kMaxLevels] = {};    // LevelLaunch::need_bits of the upper levels (need_n 0: none)
                       const uint64_t* table_args = nullptr; int table_n = 0; };            // level 0 only, valid during render_frame
    PipeFrame pipe_[kMaxLevels];
    // One keyframe on its way through render_frame (MultiBandMap2DCPU::renderFrame, .cpp:311-558): what each stage leaves for the next.
    // Kept in the map between keyframes so that its vectors keep their capacity (render_frame runs on one thread at a time).
    struct FrameWork {
        // frame_canvas(): footprint, tile range, homography
        double pts[8]; int xminInt, yminInt, xmaxInt, ymaxInt;
        int tx, ty, L, crows, ccols; double M0[9], Minv[9];
        const uint8_t* src;
        // build_tile_table(): the cull, the owned tiles and their boxes (tiles; level-0 pixels), the hash cells of the need rectangles
        bool sharded, cull, culled_any, cells_overflow;
        bool pre_raised;                                          // the keyframe's bounds entered wlb when it was admitted (lookahead): not worked out again
        Tile* const* tiles_known;                                 // ... and its canvas' tiles are known from then (nullptr: look them up)
        struct Raise { Tile* t; int q; float w; };                // (cell, wmin of this keyframe): applied once the frame is in
        std::vector<Raise> raise; std::vector<Tile*> culled, touched;
        struct Cell { int cx, cy, x0, y0, x1, y1; };              // hash cell; box of what is rendered in it, level-0 pixels
        Cell cells[64]; int ncells;
        int bx0, bx1, by0, by1, owned, owned_all;
        int pbx0, pbx1, pby0, pby1;
        // level_windows() / plan_fused_levels()
        Win need[kMaxLevels], C[kMaxLevels];
        BlockRect rects[kMaxLevels][kMaxRects]; int nrect[kMaxLevels]; int need_n[kMaxLevels];
        double blocks_run0;
        // place_table() / warp_args()
        int ring; bool table_args; const uint64_t* dtab;
        WarpArgs a;
        void reset() {
            raise.clear(); culled.clear(); touched.clear();
            culled_any = cells_overflow = false; ncells = 0; owned = owned_all = 0; blocks_run0 = 0;
            for (int i = 0; i < kMaxLevels; i++) { nrect[i] = 0; need_n[i] = 0; }
            src = nullptr; ring = 0; table_args = false; dtab = nullptr; sharded = cull = pre_raised = false; tiles_known = nullptr;
        }
    };
    FrameWork fw_;
    // Keyframes that are in (geometry done, grid advanced, tiles created, their weight bounds raised) but not rendered yet: the LOOKAHEAD of the
    // cull (round 6).  The max-weight select is order independent -- a pixel-level ends with the largest weight, the newest keyframe among equals
    // (.cpp:521, :542 `>=`; a fresh tile's unconditional copy is the same select against weight 0) -- so a keyframe may be left out of a cell not
    // only where an EARLIER keyframe's weights bound it from above (rounds 4-5) but also where a LATER one's do, provided that later keyframe is
    // certain to be rendered before anybody looks: every reader of the map's state drains this queue first (drain()), so what a caller can
    // observe after feed() k is exactly the map after keyframes 1..k.  opt_.lookahead keyframes wait here (0: render inside feed() as before).
    struct PendingFrame {
        QueuedFrame f;
        double pts[8]; int sx0, sy0, tx, ty;        // canvas origin in STABLE tile coordinates (dense index + accumulated spreadMap offset)
        double M0[9], Minv[9];
        bool cull;
        Lattice lat;                                // the cull's lattice of this keyframe (cull only)
        bool pre_raised = false;                    // lookahead: the keyframe's lower bounds entered the tiles' wlb when it was admitted
        std::vector<Tile*> tiles;                   // ... and the canvas' tiles were looked up / created then (row-major; nullptr: another shard's)
    };
    std::deque<PendingFrame> pending_;
    // The window FILLS at two keyframes per three feeds after it was last emptied (a reader, pf_sync): the first keyframe behind a sync is
    // rendered at once and every third one after it (an admission costs the host ~25 us, a launch keeps the GPU busy for ~90), so the GPU
    // does not idle while `lookahead` keyframes gather, and a short burst between two readers is not held up
    unsigned since_drain_ = 0;
    std::vector<Lattice> lat_pool_;                 // buffers of rendered keyframes' lattices, reused
    std::vector<std::vector<Tile*>> tiles_pool_;
    std::vector<double> pair_d_; std::vector<unsigned char> pair_in_;      // pre_raise scratch
    bool lookahead_ok() const { return opt_.lookahead > 0 && (single_band_ || (opt_.fused == 1 && band_num_ >= 1)) && cull_on_; }     // wherever the cull runs
    bool render_front();                            // renders pending_.front() and removes it
    bool drain();                                   // ... all of them; mu_ held.  First thing every reader of tiles, flags or counters does
    void release_slot(const QueuedFrame& f);
    void pre_raise(FrameWork& w, std::vector<Tile*>& tiles);
    int  frame_canvas(const QueuedFrame& f, FrameWork& w);
    bool build_tile_table(const QueuedFrame& f, FrameWork& w);
    void level_windows(FrameWork& w);
    bool reserve_frame_workspace(FrameWork& w);
    bool place_table(FrameWork& w);
    void warp_args(const QueuedFrame& f, FrameWork& w);
    void plan_fused_levels(FrameWork& w);
    void run_shares(const FrameWork& w, double run_share[kMaxLevels], int* exact_r0);
    double exact_level0_share(const FrameWork& w, int r0);
    bool launch_single_band(const QueuedFrame& f, FrameWork& w);
    bool launch_fused_pipeline(const QueuedFrame& f, FrameWork& w);
    bool launch_level_streams(const QueuedFrame& f, FrameWork& w);
    bool launch_per_op(const QueuedFrame& f, FrameWork& w);
    bool retire_frame(const QueuedFrame& f, FrameWork& w, bool fused);
    void log_rendered(const QueuedFrame& f);
    uint32_t need_tmp_[kMaxLevels][kNeedWords];     // render_frame scratch: the upper levels' need bitmaps of the keyframe being fed
    std::vector<unsigned __int128> cell_rows_;      // render_frame scratch: rendered cells of the canvas, one 128-bit row per cell row
    unsigned long long launch_seq_ = 0;             // parity selects the GW buffer set a launch writes
    bool flushing_ = false;
    bool launch_pipeline(const PipeFrame* cur, const WarpArgs* wa, const uint8_t* src);
    bool flush_pipeline();
    bool settle();

    // blend / save scratch (blend_lv_: the per-level form of the experiments library only)
    DevBuf blend_lv_[kMaxLevels], blend_src_, blend_out_raw_, blend_out_bgr_, mosaic_table_, strip_desc_;
    // results on their way to the host: two pinned staging slots, filled on copy_stream_ while the host empties the other one
    static constexpr size_t kOutSlot = (size_t)32 << 20;
    struct OutPiece { void* dst; const void* src; size_t bytes; };      // host destination, device source
    uint8_t*    out_pin_[2]{};
    hipEvent_t  out_copied_[2]{}, out_ready_ = nullptr;
    hipStream_t copy_stream_ = nullptr;
    int         out_threads_ = 1;
    bool        out_ring_ok_ = false;
    bool out_ring_init();
    bool download(const std::vector<OutPiece>& pieces);
#if PF_EXPERIMENTS
    bool blend_batch_per_level(const std::vector<std::pair<int,int>>& tiles, const void* const* halo9, void* raw_host, uint8_t* bgr_host);
#endif

    // frame staging + feed queue
    std::vector<FrameSlot> slots_;
    int    last_slot_ = -1; size_t last_bytes_ = 0;       // most recent upload (read_back_last_frame)
    std::mutex qmu_;
    std::condition_variable qcv_, idle_cv_;
    std::deque<QueuedFrame> queue_;
    bool worker_busy_ = false, stop_ = false;
    std::thread worker_;

    // profile
    struct ProfRec { int id; hipEvent_t a, b; double bytes, bytes_run; };
    int prof_mode_ = 0; bool prof_on_ = false;
    unsigned prof_tick_[K_COUNT]{};
    std::vector<ProfRec> prof_pending_;
    std::vector<hipEvent_t> ev_pool_;
    double prof_ms_[K_COUNT]{}; long long prof_n_[K_COUNT]{}; double prof_bytes_[K_COUNT]{}, prof_bytes_run_[K_COUNT]{};
    ProfRec prof_cur_{};
    SectionRec sections_[T_COUNT];
    std::mutex timer_mu_;
    long long n_rendered_ = 0, n_rejected_ = 0, n_dropped_ = 0, n_with_pixels_ = 0;
    long long feed_seq_ = 0; std::vector<long long> render_log_;
    double px_level0_exact_ = 0;                    // experiments library, PF_CULL_EXACT_STAT (statistics of the cull, cull_exact_stat())
#if PF_EXPERIMENTS
    double up_rect_[kMaxLevels] = {}, up_exact_[kMaxLevels] = {}; long up_n_ = 0;
    void cull_exact_stat(const FrameWork& w);
#endif
    double px_level0_ = 0, px_owned_ = 0;           // level-0 pixels computed (with halo) / tile pixels owned, over the frames rendered
};

// PNG (zlib) / PPM writer for save()
bool write_image_file(const char* filename, const uint8_t* bgr, int rows, int cols);

}  // namespace pf
