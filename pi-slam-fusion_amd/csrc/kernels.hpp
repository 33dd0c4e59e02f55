// kernels.hpp -- launch wrappers of the gfx950 kernels (kernels.hip).
//
// Data layout in HBM:
//   frame      : BGR8 interleaved, row step in bytes (as handed to feed()).
//   canvas G_i : per-frame Gaussian level i, interleaved 3 x {int16|float},
//                row-major, (tilesY*256 >> i) x (tilesX*256 >> i).
//   canvas W_i : per-frame weight level i, float, same extent.
//   tile slot  : one contiguous block per mosaic tile:
//                [lap_0 | lap_1 | ... | lap_L | w_0 | ... | w_L], level i is a
//                packed (256>>i)^2 image; offsets in TileLayout.
//   tile table : per frame, tilesX*tilesY uint64 = slot base address | fresh bit.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>

namespace pf {

constexpr int kElePixels = 256;
constexpr int kMaxLevels = 9;

struct TileLayout {
    int      nlev;          // bandNum + 1
    int      f32;           // 0: 16SC3 pyramids, 1: 32FC3
    uint32_t lap_off[kMaxLevels];   // byte offsets inside a slot
    uint32_t w_off[kMaxLevels];
    uint32_t slot_bytes;
};

TileLayout make_layout(int band_num, bool f32);

struct WarpArgs {
    double M[9];            // destination -> source map (inverse of the frame's homography)
    int    srows, scols;
    long   sstep;           // source row step, bytes
    int    crows, ccols;    // canvas extent
    int    y_off, x_off;    // canvas-space origin of the rendered window (shard sub-canvas)
    int    wrows, wcols;    // window extent (multiple of 4 rows / 64 cols)
    float  xc, yc, dis_max; // radial weight constants (MultiBandMap2DCPU.cpp:404-406)
    int    weight_type;
    const float* wmap;      // optional h x w radial weight plane (same values as the analytic form); nullptr = recompute
    int    src_cn;          // 3 = BGR8, 4 = BGRA8 (alpha ignored: the tracker's cvtColor BGRA2BGR, TrackerOpt.cpp:376-380, done in the gather)
};

// a source of blend() pixels: a tile slot or a packed halo strip set
struct BlendSrc {
    const void* base;       // nullptr = absent
    int         is_strip;
};

// one tile of a fused blend launch (collapse_fused.hip): the 3 x 3 sources of Ele::blend's assembly (.cpp:93-117; row-major, [4] = the
// tile itself), each a tile slot or -- bit j of strip_mask -- a packed halo strip set of another shard
struct BlendJob {
    uint64_t src[9];        // device addresses; only [4] is read when border == 0
    uint32_t strip_mask;
    int      border;        // 1: all nine present, padded squares with border 1 << (L - i); 0: "blend by self" (.cpp:131-145)
    int      out;           // tile index in the output arrays
    int      pad_;
};
static_assert(sizeof(BlendJob) == 88, "BlendJob is copied to LDS by words");

// kernel ids for the profile table
enum KernelId { K_WARP = 0, K_PYRDOWN_IMG, K_PYRDOWN_W, K_LAP_SELECT, K_BLEND_GATHER, K_COLLAPSE,
                K_BLEND_FINISH, K_MOSAIC_GATHER, K_SAVE_FINISH, K_LEVEL0, K_LEVEL, K_SINGLE, K_BLEND_FUSED, K_SAVE_FUSED, K_COUNT };
const char* kernel_name(int id);

void launch_warp(hipStream_t s, bool f32, const uint8_t* src, const WarpArgs& a, void* g0, float* w0);

// type: 0 = 16SC3, 1 = 32FC3, 2 = 32FC1.  Canvas-level pyrDown with the window
// (shard) restricted to dst rows [y0,y1) x cols [x0,x1).
void launch_pyrdown(hipStream_t s, int type, const void* src, int srows, int scols, void* dst,
                    int y0, int y1, int x0, int x1);

// Laplacian level `level` (G_i - pyrUp(G_{i+1}); top level: G_L) fused with the
// per-tile max-weight select (MultiBandMap2DCPU.cpp:496-551).
void launch_lap_select(hipStream_t s, const TileLayout& lay, int level, const void* g_i, const void* g_up,
                       const float* w_i, int rows, int cols, const uint64_t* tile_table, int tiles_x,
                       int ty0, int ty1, int tx0, int tx1);

// Fused per-level kernel: level i's Gaussian block (from the warp when
// wa != nullptr, else from the packed GW_i buffer) -> GW_{i+1} + Laplacian select of level i
// (+ the top level when top_select).  Compute region [cx0,cx1) x [cy0,cy1) in level-i pixels.
size_t level_px_bytes(bool f32);
void launch_level(hipStream_t s, const TileLayout& lay, int level, int rows, int cols, int cx0, int cy0, int cx1, int cy1,
                  int tiles_x, bool top_select, bool write_next, const WarpArgs* wa, const uint8_t* src,
                  const void* gw_in, void* gw_out, const uint64_t* table, int shape = 3);   // shape 2: 4-stage k_level, 3: k_level3

// Pipelined form of the same kernel: ONE launch carries several independent level jobs (level 0 of the
// newest frame, level 1 of the frame before it, ...), each with its own tile table and GW buffers.
constexpr int kArgTable = 256;      // tile-table entries that can travel inside the kernel arguments of a launch
constexpr int kMaxRects = 8;        // need rectangles of a level-0 job (tile-sharded canvases, the cull): what LevelLaunch can hold
constexpr int kMaxRectsUpper = 4;   // ... of an upper-level job (their need bitmaps took the room in the kernel arguments; they are the fallback there)
constexpr int kNeedWords = 100;     // 32-bit words of need bitmaps a launch can carry for its upper-level jobs (kernel arguments are 4 KB)
struct BlockRect { short x0, y0, x1, y1; };      // [x0,x1) x [y0,y1) in blocks of the job's block grid
struct LevelLaunch {
    int level, rows, cols;          // pyramid level and its canvas extent
    int cx0, cy0, cx1, cy1;         // compute region
    int tiles_x;
    bool top_select, write_next, from_warp;
    const void* gw_in; void* gw_out;
    const uint64_t* table;          // the frame's tile table in device memory
    // level-0 job only: the table's entries on the host (n <= kArgTable).  They travel in the launch's kernel arguments --
    // visible to the launch by the runtime's own contract, no copy in the stream, no host memory read in place --
    // and the launch itself stores them to `table` for the later launches that carry the frame's upper levels.
    const uint64_t* table_args; int table_n;
    // tile-sharded canvases: the union of these rectangles holds every block something owned by this rank depends on;
    // the other blocks of the grid exit at once.  nrect == 0: every block runs.
    int nrect; BlockRect rect[kMaxRects];
    // upper-level jobs: one bit per block of the job's grid (row-major, nbx = ceil((cx1 - cx0) / 64)), set where something rendered depends
    // on the block; travels in the kernel arguments when the launch's jobs fit kNeedWords together, and the rectangles are not looked at then
    const uint32_t* need_bits = nullptr; int need_n = 0;
};
void launch_levels(hipStream_t s, const TileLayout& lay, const LevelLaunch* jobs, int njobs, const WarpArgs* wa, const uint8_t* src);
int  read_phase_stamps(unsigned long long* out, int cap_blocks);
void read_form_counts(long long out[8]);
long long read_compact_launches();                                // ... whose level-0 job ran one workgroup per block inside its need rectangles                          // launches of the pipelined kernel by form (kernels.hip, g_form_counts)
int  read_select_counts(unsigned long long* out, int reset);       // diagnostics (PF_STAMP=1): [2*level] pixels stage D saw, [2*level+1] pixels that won
// Reach (level-0 pixels) of the test by which the level-0 blocks of a pipelined launch decide for themselves whether they run (k_levels,
// LevelBatch::need_r0): 0 when the launch takes the need rectangles instead.  table_n: entries of the frame's tile table if it travels in
// the kernel arguments (0: it does not); nrect0: rectangles of the level-0 job.
int  level0_need_reach(const TileLayout& lay, int table_n, int nrect0);
int  level_block_rows(bool f32);                                     // block height of the pipelined level kernel (fused = 1)      // diagnostics (PF_STAMP=1)

// Ele::blend (+ the 8U view) of n tiles / save()'s paste + collapse + 8U + background: ONE launch each, the pyramid collapsed in LDS
// (collapse_fused.hip).  jobs_dev / table_dev in device memory; raw_out (pyramid type, n x 256 x 256 x 3) and bgr_out may be null
void launch_blend_fused(hipStream_t s, const TileLayout& lay, const BlendJob* jobs_dev, int n, void* raw_out, uint8_t* bgr_out);
void launch_save_fused(hipStream_t s, const TileLayout& lay, const uint64_t* table_dev, int wx, int wy, int bg, uint8_t* bgr_out);

// The per-level form of rounds 1-5 (one padded square per level in HBM, one launch per reference op): kept in the experiments library
// as the A/B partner and second opinion of the fused kernel (PF_BLEND_PER_LEVEL=1, tests/test_gpu_variants.py)
// blend(): gather padded level images for `batch` tiles (9 sources each), collapse, finish
void launch_blend_gather(hipStream_t s, const TileLayout& lay, int level, int border, const BlendSrc* srcs,
                         void* dst, size_t dst_stride_bytes, int batch);
void launch_collapse(hipStream_t s, bool f32, void* dst, size_t dst_stride_bytes, const void* src,
                     size_t src_stride_bytes, int rows, int cols, int batch);
void launch_blend_finish(hipStream_t s, const TileLayout& lay, const void* lvl0, size_t stride_bytes, int border,
                         const BlendSrc* srcs, void* raw_out, uint8_t* bgr_out, int batch, const int* out_idx = nullptr);

// save(): paste tiles of a dense (wx x wy) table into one mosaic level, then collapse, then finish
void launch_mosaic_gather(hipStream_t s, const TileLayout& lay, int level, const uint64_t* table, int wx, int wy, void* dst);
void launch_save_finish(hipStream_t s, const TileLayout& lay, const void* lvl0, const uint64_t* table, int wx, int wy,
                        int bg, uint8_t* bgr);

// halo strip pack (multi-GPU blend): writes the strip set a neighbour at (dx,dy) needs
size_t halo_bytes(const TileLayout& lay, int dx, int dy);
void launch_halo_pack(hipStream_t s, const TileLayout& lay, const void* slot, int dx, int dy, void* out);
// all strip sets of a seam exchange in one launch (descs in device memory; out_off = byte offset of a set in `out`)
struct StripDesc { const void* slot; int dx, dy; size_t out_off; };
void launch_halo_pack_batch(hipStream_t s, const TileLayout& lay, const StripDesc* descs_dev, int n, void* out);

// Map2DCPU semantics (single_band.hip): weight byte plane, BGRA warp + select into 256x256x4 tiles
void launch_weight8(hipStream_t s, uint8_t* w, int rows, int cols, int weight_type);
// the reference's CV_32FC1 weightImage (MultiBandMap2DCPU.cpp:400-418), built once per frame size
void launch_weight32(hipStream_t s, float* w, int rows, int cols, int weight_type);
void launch_single(hipStream_t s, const uint8_t* src, const uint8_t* w8, const WarpArgs& a, const uint64_t* table, int tiles_x);

}  // namespace pf
