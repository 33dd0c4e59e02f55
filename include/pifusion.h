/*
 * pifusion.h -- C ABI of libpifusion.so, the MI355X-native drop-in for the
 * Map2DFusion multi-band hot path of Immortalqx/pi-slam-fusion.
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to the reference tree).  Plain pointers and sizes only; no C++/torch types.
 * All int-returning calls follow the reference's bool convention: 1 = ok,
 * 0 = failed (message on stderr), never throw, never abort.
 *
 * Poses are 7 doubles in the reference's SE3 stream order
 * `x y z qx qy qz qw` (GSLAM/GSLAM/core/SE3.h:112-117); a pose is
 * camera-to-world exactly as handed to Map2D::feed (Map2DFusion/Map2D.h:91).
 */
#ifndef PIFUSION_H
#define PIFUSION_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_ELE_PIXELS 256            /* Map2DFusion/Map2D.h:35 ELE_PIXELS */

/* Map2D::Map2DType, Map2DFusion/Map2D.h:83 */
enum { PF_TYPE_NONE = 0, PF_TYPE_CPU = 1, PF_TYPE_GPU = 2, PF_TYPE_MULTIBAND = 3, PF_TYPE_RENDER = 4 };

/* OpenCV / GImage type codes (GSLAM/GSLAM/core/GImage.h:27-36,97) */
enum { PF_8UC1 = 0, PF_8UC3 = 16, PF_8UC4 = 24, PF_16SC3 = 19, PF_32FC1 = 5, PF_32FC3 = 21 };

/* Image view, layout-compatible with cv::Mat / GSLAM::GImage headers
 * (rows, cols, type, data; GImage.h:387-389).  step = bytes per row
 * (0 = packed).  data may be a host pointer (pf_feed) or a device pointer
 * (pf_feed_device). */
typedef struct pf_image {
    int         rows, cols, type;
    const void* data;
    size_t      step;
} pf_image;

/* The svar keys read on the path, as one struct. */
typedef struct pf_options {
    int    band_number;      /* MultiBandMap2DCPU.BandNumber   (5)  .cpp:260 */
    int    force_float;      /* MultiBandMap2DCPU.ForceFloat   (0)  .cpp:444 */
    int    high_quality_show;/* MultiBandMap2DCPU.HighQualityShow (1) .cpp:261 */
    int    weight_type;      /* Map2D.WeightType               (0)  .cpp:407 */
    int    bg_color;         /* Result.BackGroundColor         (0)  .cpp:840 */
    double resolution;       /* Map2D.Resolution (0 = auto)         .cpp:228 */
    double scale;            /* Map2D.Scale                    (1)  .cpp:235 */
    int    device;           /* HIP device ordinal (-1 = current)            */
    int    shard_rank;       /* tile sharding: this rank ...                 */
    int    shard_count;      /* ... of shard_count (1 = own every tile)      */
    int    shard_block;      /* spatial-hash cell edge in tiles (default 8)  */
    int    max_queue;        /* feed queue cap, drop-oldest (20)    .cpp:302 */
    int    fused;            /* fused level kernels.  1 (default): one launch per keyframe,
                                levels pipelined across frames; 3: one launch and stream
                                per level; 2: as 3 in the 4-stage 64x16-block shape;
                                0: one kernel per reference op (warp / pyrDown /
                                Laplacian+select)                                    */
    int    lookahead;        /* keyframes that wait, fed but not rendered, so that the cull can leave a keyframe out of the cells
                                in which one of the NEXT `lookahead` keyframes is bound to overwrite it (default 48; 0: every
                                keyframe is rendered inside its own feed call).  The select keeps the largest weight, the newest
                                keyframe among equals, whatever the order, and every call that reads tiles, flags or counters
                                (pf_sync, blend, save, tile access, statistics) renders what waits first (the window then fills
                                again at two keyframes per three feeds, so the GPU is not left idle): what a caller can observe
                                is the map after the keyframes fed so far, exactly as without it.  Multi-band maps with
                                fused = 1 and Map2DCPU maps (a shard looks ahead among its own tiles); elsewhere the value is ignored.  A pf_feed_device frame must stay valid until
                                pf_sync in either case.                                                                      */
} pf_options;

typedef struct pf_map pf_map;

/* --- lifecycle --------------------------------------------------------- */
void    pf_default_options(pf_options* o);
/* key=value setter accepting the reference's svar key names
 * ("MultiBandMap2DCPU.BandNumber", "Map2D.Scale", ...); 1 if the key is known */
int     pf_options_set(pf_options* o, const char* key, const char* value);
/* Map2D::create(type, thread), Map2DFusion/Map2D.cpp:51-66.  MULTIBAND -> the multi-band
 * engine; CPU -> Map2DCPU semantics (single 8-bit band); GPU -> the same, as the reference
 * itself falls back ("CUDA is not enabled, switch to CPU"); NONE/RENDER -> NULL.        */
pf_map* pf_create(int type, int thread, const pf_options* opt);
void    pf_destroy(pf_map* m);
const char* pf_last_error(void);

/* --- Map2D virtuals ---------------------------------------------------- */
/* Map2D::prepare(plane, camera, frames), MultiBandMap2DCPU.cpp:266-286.
 * cam = {w,h,fx,fy,cx,cy} (PinHoleParameters, Map2D.h:37-43).  imgs may be
 * NULL (only the poses size the grid when thread=0; with thread=1 frames
 * that carry an image are queued and rendered first, Map2D.cpp:42).        */
int     pf_prepare(pf_map* m, const double plane[7], const double cam[6],
                   int n, const pf_image* imgs, const double* poses7);
/* Map2D::feed(img, pose), MultiBandMap2DCPU.cpp:288-309.  Host BGR8 (or BGRA8) frame; the
 * rows are copied to HBM as they lie (img->step is kept) by one blocking linear H2D copy, so
 * the caller may release its pixels when feed returns (the reference keeps a refcounted
 * cv::Mat instead).                                                          */
int     pf_feed(pf_map* m, const pf_image* img, const double pose[7]);
/* Same, frame already resident in HBM (img->data is a device pointer that
 * must stay valid -- and unchanged -- until pf_sync: the frame is read when its keyframe is rendered, which with
 * pf_options.lookahead = n may be n feeds later; a caller that cycles through a ring of device buffers without
 * pf_sync needs more than n + 1 of them, or sets lookahead to 0).  thread=0 maps only.                          */
int     pf_feed_device(pf_map* m, const pf_image* img, const double pose[7]);
/* Test hook (no reference counterpart): the bytes of the host frame most recently copied by pf_feed / pf_prepare,
 * read back from HBM -- (rows-1)*step + cols*channels of them.  Returns the byte count (out may be NULL to query it),
 * -1 when there is none.  Lets a test prove that a row-padded frame arrived byte for byte.                        */
long    pf_debug_read_last_frame(pf_map* m, void* out, size_t cap);
/* Diagnostics (PF_STAMP=1 selects a stamped instantiation of the level kernel; tools/stamp_phases.py): in-kernel clock
 * stamps of the most recent launch, 8 uint64 per workgroup.  Returns the workgroup count.                            */
int     pf_debug_phase_stamps(unsigned long long* out, int cap_blocks);
/* PF_STAMP=1 builds only: per pyramid level, pixels the max-weight select looked at (out[2*level]) and pixels that won
 * (out[2*level+1]) since the last reset; out holds 18 values.  Counts the useful tile bytes of a launch. */
int     pf_debug_select_counts(unsigned long long* out, int reset);
/* Map2D::queueSize(), MultiBandMap2DCPU.h:110-113: frames in the feed queue (cap 20, drop-oldest).  Keyframes the render thread has
 * taken out of it and holds for the cull's lookahead are not counted: they no longer occupy a place in the queue and cannot be dropped. */
unsigned pf_queue_size(pf_map* m);
/* drain the feed queue, render the keyframes that wait for the cull's lookahead (pf_options.lookahead) and wait for the
 * device stream (no reference counterpart: the reference is synchronous on the CPU).  Returns 0 if a render failed.       */
int     pf_sync(pf_map* m);
/* Map2D::save(filename), MultiBandMap2DCPU.cpp:779-847.  Writes PNG (.png),
 * else binary PPM.                                                         */
int     pf_save(pf_map* m, const char* filename);
/* The file leg of save() alone: cv::imwrite(filename, result), MultiBandMap2DCPU.cpp:841.  8-bit BGR in,
 * PNG (8-bit RGB, deflate) when the name ends in .png/.PNG, binary PPM (P6) otherwise.  No device needed. */
int     pf_write_image(const char* filename, const uint8_t* bgr, int rows, int cols);
/* Input side of the file driver: the reference reads each keyframe with cv::imread(imgfile), backup/map2dfusion.cpp:129-132
 * (8-bit BGR, EXIF orientation ignored as OpenCV 2.4.9 does).  JPEG (baseline, extended and progressive Huffman; grey or
 * YCbCr/RGB; libjpeg's default ISLOW IDCT, fancy upsampling and colour tables, byte-equal to libjpeg-turbo), PNG (non-interlaced;
 * grey replicated, palette looked up, alpha dropped, 16-bit samples cut to 8: IMREAD_COLOR's conversion) and binary PPM.
 * pf_image_info fills rows/cols; pf_read_image decodes into rows*cols*3 bytes.  pf_jpeg_* take the stream from memory.
 * Host code; no device needed.  0 + pf_last_error() on anything unsupported or malformed.                              */
int     pf_image_info(const char* filename, int* rows, int* cols);
int     pf_read_image(const char* filename, uint8_t* bgr, int rows, int cols);
int     pf_jpeg_info(const uint8_t* data, size_t len, int* rows, int* cols, int* components);
int     pf_jpeg_decode_bgr(const uint8_t* data, size_t len, uint8_t* bgr, int rows, int cols);
/* The same decode on the GPU (csrc/jpeg_device.hip), byte-equal to pf_jpeg_decode_bgr: the calling thread parses the markers;
 * sequential one-scan streams (with or without restart intervals) are Huffman-decoded on the GPU too (a self-synchronising parallel pass,
 * csrc/jpeg_huff_par.hpp), the others on the calling thread; kernels do libjpeg's dequantise + ISLOW IDCT, fancy upsampling and
 * colour conversion.  dev_bgr: rows*cols*3 bytes of device memory, complete in the order of `hip_stream`
 * (a hipStream_t, NULL = the default stream); the call returns when the work is queued.                               */
int     pf_jpeg_decode_device(const uint8_t* data, size_t len, void* dev_bgr, int rows, int cols, void* hip_stream);
/* Diagnostics: out = { frames whose Huffman pass ran on the GPU, frames that fell back to the host's serial pass after trying,
 * launches the most recent GPU pass took } of the map's decoder (m = NULL: of pf_jpeg_decode_device's).  Sequential one-scan streams,
 * with or without restart intervals, take the GPU pass (csrc/jpeg_huff_par.hpp); PF_JPEG_HOST_HUFFMAN=1 keeps every stream on the host.  */
void    pf_debug_jpeg_huffman(pf_map* m, long long out[3]);
/* Map2D::feed(cv::imread(imgfile), pose) in one call (backup/map2dfusion.cpp:129-135 + Map2DFusion.cpp:313-327): markers and
 * Huffman on the calling thread, then the frame is finished on the GPU straight into the slot in HBM a host frame would have
 * been uploaded to, and queued (thread=1) or rendered (thread=0) from there.  Returns what pf_feed returns.             */
int     pf_feed_jpeg(pf_map* m, const uint8_t* data, size_t len, const double pose[7]);
/* n keyframes at once: their Huffman passes run side by side on `threads` host threads (0: one per frame; batches of 16), then
 * the frames are uploaded, finished on the GPU and fed in the order given -- the mosaic is the one n pf_feed_jpeg calls build.
 * results[i] (may be NULL) = what pf_feed_jpeg returns for frame i; returns the number of frames fed.                  */
int     pf_feed_jpeg_batch(pf_map* m, int n, const uint8_t* const* data, const size_t* len, const double* poses7, int threads, int* results);
/* save() without the file: whole-mosaic collapse into caller memory.  Call
 * with bgr=NULL to query rows/cols/origin tile.                            */
int     pf_save_to_memory(pf_map* m, uint8_t* bgr, int* rows, int* cols, int* tile_x0, int* tile_y0);

/* --- MultiBandMap2DCPU::Ele tile surface (MultiBandMap2DCPU.h:32-51) ---- */
int     pf_num_levels(pf_map* m);                        /* bandNum+1          */
int     pf_pyramid_type(pf_map* m);                      /* PF_16SC3 | PF_32FC3 */
/* MultiBandMap2DCPUData geometry (.h:61-69): dims={w,h,off_x,off_y},
 * geo={min.x,min.y,max.x,max.y,eleSize,lengthPixel}.  Tile coordinates in
 * this API are stable: dense grid index + off (they survive spreadMap).    */
int     pf_grid(pf_map* m, int dims[4], double geo[6]);
int     pf_tile_count(pf_map* m);
int     pf_tile_coords(pf_map* m, int* xy, int cap);
/* Ele::pyr_laplace[level] / Ele::weights[level]: D2H copy of one level.    */
int     pf_get_tile_level(pf_map* m, int ix, int iy, int level, void* lap, float* w);
/* Map2DCPU::Map2DCPUEle::img (Map2DFusion/Map2DCPU.h): the 256x256 BGRA tile of a
 * TypeCPU / TypeGPU map (alpha = winning weight byte).                      */
int     pf_get_tile_bgra(pf_map* m, int ix, int iy, uint8_t* bgra256);
/* Ele::blend(neighbors), .cpp:77-146: raw result in the pyramid type.      */
int     pf_blend_tile_raw(pf_map* m, int ix, int iy, void* out);
/* Ele::updateTexture's pixels, .cpp:149-160: blend -> BGR8 256x256.        */
int     pf_blend_tile(pf_map* m, int ix, int iy, uint8_t* bgr256);
/* batch form of the draw() loop (.cpp:705-742): blend every tile whose
 * Ischanged flag is set, clear the flags; xy/bgr sized by cap tiles.       */
int     pf_blend_changed(pf_map* m, int* xy, uint8_t* bgr, int cap);
/* Ele::blend + the 8U view for a caller-chosen list of n tiles (xy = n pairs ix, iy), Ischanged left alone: what a viewer that
 * re-draws a region calls.  bgr: n x 256 x 256 x 3; a tile without pyramid leaves its 196 608 bytes untouched.  One launch
 * per 1024 tiles.  Returns 1 / 0.                                           */
int     pf_blend_tiles(pf_map* m, const int* xy, int n, uint8_t* bgr);
/* Page-locked host memory for the output side: results written into such a buffer (pf_blend_changed, pf_blend_tiles,
 * pf_save_to_memory) are copied from HBM straight into it; any other buffer is filled through the library's own pinned
 * staging ring and a host copy (the counterpart of cv::Mat's allocator for the textures updateTexture hands to GL).       */
void*   pf_host_alloc(size_t bytes);
void    pf_host_free(void* p);
/* The text draw() hands to scommand.Call("MapWidget", ...) for a refreshed tile when Fuse2Google is set
 * (MultiBandMap2DCPU.cpp:744-757): "Map2DUpdate LastTexMat <lng lat 0 of the tile's top-left corner> <... bottom-right>".
 * Byte for byte the reference's string: tile corners rounded to float first (.cpp:709-712), plane * corner, then
 * pi::calcLngLatFromDistance around GPS.Origin (PIL/src/hardware/Gps/utils_GPS.cpp:133-160), every field through
 * std::to_string -- six decimals; the call site's setprecision(9) never reaches a number (GSLAM/core/Point.h:166-170).
 * pf_format_map_update: the pure function (grid minimum, tile edge in metres, dense tile index x, y as the draw() loop counts).
 * pf_map_update_command: the same for tile (ix, iy) of a prepared map (stable tile coordinates), with the reference's gate
 * `updated && !inborder`: 0 when the tile holds no pyramid yet, or lies on the rim of the dense grid while
 * HighQualityShow is on (one of its 3x3 neighbours falls outside, .cpp:730-735).  `Fuse2Google` itself is the caller's flag.
 * Both return the length written (without the terminating 0), 0 when nothing is to be sent or cap is too small. */
int     pf_format_map_update(const double plane[7], const double gps_origin[3], double min_x, double min_y, double ele_size,
                             int x, int y, char* out, int cap);
int     pf_map_update_command(pf_map* m, int ix, int iy, const double gps_origin[3], char* out, int cap);
/* pi::calcLngLatFromDistance (utils_GPS.cpp:133-160) alone */
void    pf_lnglat_from_distance(double lng1, double lat1, double dx, double dy, double* lng2, double* lat2);
/* unused-by-the-reference helpers kept for API completeness (.cpp:57-75)   */
int     pf_normalize_using_weight_map(const float* weight, float* src3, size_t npix);
int     pf_mul_weight_map(const float* weight, float* src3, size_t npix);

/* --- stateless host geometry (no device needed) ---------------------------
 * The fp64 pose algebra the path uses, in the reference's operation order:
 * SE3::inverse / operator* (GSLAM/GSLAM/core/SE3.h:70-90), SO3*Point (SO3.h:445-450),
 * the four-corner ground footprint with the 0.4 gate (.cpp:324-347; pose in plane
 * coordinates, returns 0 when the frame is rejected) and the homography set-up of
 * cv::getPerspectiveTransform (.cpp:441). */
void    pf_se3_inverse(const double a[7], double out[7]);
void    pf_se3_mul(const double a[7], const double b[7], double out[7]);
void    pf_so3_rotate(const double q[4], const double p[3], double out[3]);
int     pf_footprint(const double cam[6], const double pose_plane[7], double pts8[8]);
void    pf_perspective_transform(const float src8[8], const float dst8[8], double M[9]);

/* --- multi-GPU seam exchange (no reference counterpart) ------ */
/* owner rank of a tile under the spatial hash */
int     pf_tile_owner(const pf_options* o, int ix, int iy);
/* Halo strips for Ele::blend across ranks.  A "strip set" of tile (ix,iy)
 * seen from a neighbour at (dx,dy) is the bytes blend() would copy from it
 * (.cpp:101-116), all levels concatenated.  pf_halo_bytes gives its size;
 * pack writes it to a device buffer, blend_tile_halo consumes up to 8 such
 * device buffers (NULL = neighbour absent) in place of local tiles.        */
size_t  pf_halo_bytes(pf_map* m, int dx, int dy);
int     pf_halo_pack(pf_map* m, int ix, int iy, int dx, int dy, void* dev_out);
int     pf_blend_tile_halo(pf_map* m, int ix, int iy, const void* const dev_halo[9], uint8_t* bgr256, void* raw);
/* whole tiles for save()'s gather: bytes of one tile's pyramid + weights   */
size_t  pf_tile_bytes(pf_map* m);
int     pf_tile_export(pf_map* m, int ix, int iy, void* dev_out);
int     pf_tile_import(pf_map* m, int ix, int iy, const void* dev_in);

/* --- the seam exchange inside the library ---------------------------------
 * What a reference-side caller (the Map2DHIP subclass of INTEGRATION.md) needs to run draw() and save() when the mosaic is
 * sharded over the GPUs of a node: one process per GPU, each with its own pf_map created with shard_rank / shard_count,
 * every keyframe fed to every rank (feed has no collective).  Reference semantics reproduced across ranks: the neighbour
 * gather of draw(), MultiBandMap2DCPU.cpp:724-741 with Ele::blend :77-146, and the per-level paste + single collapse of
 * save(), :806-836.  Transport: RCCL (grouped ncclSend/ncclRecv between tile owners over xGMI; librccl is loaded at run
 * time) or a caller-supplied host-buffer exchange (tests, gloo / MPI launchers).  Collective calls: every rank of the
 * group must make the same pf_dist_* call.                                                                          */
typedef struct pf_dist pf_dist;
/* all-to-all-v over host buffers, indexed by peer rank (own entry: 0 bytes); 1 = ok */
typedef int (*pf_exchange_fn)(void* user, const void* const* send, const size_t* send_bytes,
                              void* const* recv, const size_t* recv_bytes, int nranks);
/* ncclGetUniqueId: 128 bytes, made on one rank and handed to all (by the launcher's own means) */
int      pf_dist_unique_id(void* out128);
pf_dist* pf_dist_init_rccl(pf_map* m, const void* unique_id128, int rank, int nranks);
pf_dist* pf_dist_init_host(pf_map* m, int rank, int nranks, pf_exchange_fn fn, void* user);
void     pf_dist_destroy(pf_dist* d);
/* Map2D::feed across ranks ("one H2D + P2P over xGMI"): the tracker hands a keyframe to ONE rank -- `root`, the
 * only rank whose img->data (host pixels) is read; every rank makes the call with the same pose and the same frame
 * description (rows, cols, type, step).  From the pose alone all ranks derive which ranks own a tile of the frame's canvas;
 * the root copies the pixels to its GPU once and sends them, in one grouped point-to-point exchange, to exactly those
 * ranks; each rank then renders its tiles (a rank that holds none of the canvas only advances its grid).  thread=0 maps.
 * Returns 1 (accepted), 0 (rejected as Map2D::feed would: oblique view ...), -1 (failure, on every rank alike).          */
int      pf_dist_feed(pf_dist* d, const pf_image* img, const double pose[7], int root);
/* The same with the keyframe as the bytes of its .jpg file (pf_feed_jpeg across ranks): only the root reads `data`; every rank
 * makes the call with the same pose and the frame's size (rows, cols: the camera's).  The root decodes on its GPU straight into
 * the slot the exchange sends from.                                                                                       */
int      pf_dist_feed_jpeg(pf_dist* d, const uint8_t* data, size_t len, int rows, int cols, const double pose[7], int root);
/* draw() across ranks: every rank blends ITS changed tiles (at most cap) with the edge strips of neighbours that live on
 * other ranks and clears their Ischanged flags.  One pack launch, one grouped exchange, one batched blend per call.
 * Returns the number of tiles written to xy / bgr (cap tiles of 256x256x3 each), -1 on failure.                     */
int      pf_dist_blend_changed(pf_dist* d, int* xy, uint8_t* bgr, int cap);
/* save() across ranks: every tile travels once to rank 0, which runs the whole-mosaic collapse and (pf_dist_save) writes
 * the file.  On the other ranks the calls return 1 with rows = cols = 0.                                            */
int      pf_dist_save(pf_dist* d, const char* filename);
int      pf_dist_save_to_memory(pf_dist* d, uint8_t* bgr, int* rows, int* cols, int* tile_x0, int* tile_y0);
/* what the last pf_dist_* call moved */
typedef struct pf_dist_stats {
    unsigned long long bytes_sent, bytes_received;   /* payload, this rank */
    unsigned long long strips_sent, strips_received, tiles, peers;
    double plan_ms, pack_ms, exchange_ms, compute_ms;
    unsigned long long verified;                      /* data exchanges checked end to end in this call (PF_DIST_VERIFY=1: every
                                                       * rank hashes what it sent to and received from each peer; the hashes must agree) */
} pf_dist_stats;
int      pf_dist_last_stats(pf_dist* d, pf_dist_stats* out);
/* The plan of one pf_dist_blend_changed call as rank `me` derives it from the all-gathered tile lists -- a pure function
 * (no device, no transport), exported so that launchers and the CPU tests can inspect or replay the exchange the library
 * will make.  lists: for every rank, counts[r] records of 3 ints (ix, iy, Ischanged), ranks back to back; caps[r]: the most
 * tiles rank r blends in the call; halo_bytes9[j]: bytes of the strip set of neighbour j = 3*(dy+1)+(dx+1) (pf_halo_bytes).
 * send[i]: tile (ix,iy) of THIS rank hands the edge facing (-dx,-dy) to rank `peer`, at byte `offset` of the bytes for that
 * peer; recv[i]: this rank's tile number `tile` (index into mine_xy) gets neighbour j = 3*(dy+1)+(dx+1) from rank `peer` at
 * byte `offset` of the bytes from that peer.  Returns 1, or 0 when a capacity is too small (the n_* still say what is needed). */
typedef struct pf_strip_plan { int peer, ix, iy, dx, dy, tile; unsigned long long offset; } pf_strip_plan;
int      pf_dist_plan_blend(int nranks, int me, const int* counts, const int* lists3, const long long* caps, int high_quality,
                            const size_t halo_bytes9[9], pf_strip_plan* send, int send_cap, int* n_send,
                            pf_strip_plan* recv, int recv_cap, int* n_recv, int* mine_xy, int mine_cap, int* n_mine);
/* what the group looks like from this rank: its rank, the number of ranks the transport was brought up with, and the
 * transport's name ("rccl" / "host").  A launcher's first multi-GPU run checks these before it trusts a collective.  */
int      pf_dist_info(pf_dist* d, int* rank, int* nranks, const char** transport);
/* end-to-end check of every data exchange of the following pf_dist_* calls (default: on iff PF_DIST_VERIFY is set): each
 * rank hashes (FNV-1a) what it sent to and what it received from each peer, the hashes travel as a control message and
 * must agree; a mismatch fails the call ON EVERY RANK ALIKE (the verdict is agreed on after the hash exchange, so the sender of
 * wrong bytes returns -1 as well and no rank is left waiting in the next collective) and pf_last_error names the pair on the
 * receiving rank.  One rank asking is enough: the wish travels with the status word the ranks agree on before every data
 * exchange, so ranks started with different PF_DIST_VERIFY settings still run the same sequence of collectives.
 * Costs a device-to-host copy of the payload. */
int      pf_dist_set_verify(pf_dist* d, int on);

/* --- measurement -------------------------------------------------------- */
/* Per-kernel HIP-event timing on the map's own stream.  mode 0 = off,
 * 1 = every kernel, 2+k = only kernel k (index in pf_profile_read order); adding n << 8 times every
 * n-th launch only (an event pair costs a few microseconds of stream time per launch).  Read returns
 * the kernel count; names[i] are static strings; ms / launches / bytes cover the timed launches.    */
int     pf_profile_enable(pf_map* m, int mode);
int     pf_profile_read(pf_map* m, int cap, const char** names, double* total_ms, long long* launches,
                        double* alg_bytes);
/* The same, plus alg_bytes_run: the algorithmic bytes of the part of the canvases the timed launches' blocks actually processed
 * (SURVEY 8d's per-tile bytes x the share of each level's pixels covered by blocks that ran, + the frame read once).  The cull of
 * render_frame and a shard leave blocks of a canvas out; alg_bytes counts every tile of every canvas regardless.  Equal to
 * alg_bytes for kernels that always cover their whole window. */
int     pf_profile_read_run(pf_map* m, int cap, const char** names, double* total_ms, long long* launches,
                            double* alg_bytes, double* alg_bytes_run);
int     pf_profile_reset(pf_map* m);
/* sharding overhead since creation: {frames that put pixels on this rank, level-0 pixels computed (owned tiles + pyramid
 * halo, fused path), tile pixels owned in those frames, tiles held}: [1]/[2] is the halo recompute factor of a shard  */
int     pf_render_stats(pf_map* m, double out4[4]);
/* Host section timers with the reference's section names (pi::timer.enter / leave, PIL/src/base/time/Timer.h:43-85;
 * MultiBandMap2DCPU.cpp:476,555,563,602,628-630,722,742): "Map2D::feed", "MultiBandMap2DCPU::renderFrame",
 * "MultiBandMap2DCPU::Apply", "MultiBandMap2DCPU::spreadMap", "MultiBandMap2DCPU::updateTexture",
 * "MultiBandMap2DCPU::save" -- calls, mean / min / max seconds of HOST time per section (kernels run asynchronously: device
 * time is pf_profile_read's).  With PF_ROCTX=1 in the environment every section is also a roctx range.  Returns the count. */
int     pf_timer_read(pf_map* m, int cap, const char** names, long long* calls, double* mean_s, double* min_s, double* max_s);
int     pf_timer_reset(pf_map* m);
/* Tiles left out of launches so far because the keyframe fed could not win the max-weight select anywhere in them (the cull of
 * render_frame: bounds from the geometry alone; results are those of the full render, PF_CULL=0 switches it off). Diagnostics. */
long long pf_debug_culled_tiles(pf_map* m);
/* ... and the 64 x 64 cells (16 per tile) switched off inside tiles that were rendered.  Diagnostics. */
long long pf_debug_culled_cells(pf_map* m);
/* With PF_CULL_EXACT_STAT=1 in the environment: level-0 pixels of the blocks within the pyramid's reach of a rendered cell, summed
 * over the keyframes fed (what the in-kernel need test lets run; tools/cull_stats.py).  0 otherwise.  Diagnostics. */
double  pf_debug_level0_exact_px(pf_map* m);
/* The cull on (default, unless PF_CULL=0 is in the environment) or off: off renders every tile of every keyframe's canvas, as the
 * reference does; the mosaic is the same either way.  For measurements (bench.py reports both rates). */
/* Test hook: the pf_feed / pf_feed_device calls (0-based, counted since creation, geometry-only feeds included) whose keyframes renderFrame
 * accepted, in render order; writes the newest min(cap, count) of them and returns the count (at most the newest 65536 are kept).  With
 * thread = 1 and a queue that drops (MultiBandMap2DCPU.cpp:300-303) this is the only way to tell a checker which keyframes the map holds. */
int     pf_debug_render_log(pf_map* m, long long* out, int cap);
void    pf_set_cull(pf_map* m, int on);
/* Launches of the pipelined level kernel by form since the library was loaded: [0] block form with the computed weight (default),
 * [1] block form gathering the weight plane, [2] LDS-staged source patch, [3] rolling strips, [4] 64x64 blocks, [5] 64x28 blocks,
 * [6] stamped instantiation, [7] launches whose tile table travelled in the kernel arguments.  Diagnostics (variant tests). */
void    pf_debug_form_counts(long long out[8]);
/* ... and how many of them ran their level-0 job on the compact grid (one workgroup per block inside the need rectangles of a shard / of
 * the cull, instead of one per block of the canvas' bounding box) */
long long pf_debug_compact_launches(void);
/* frames rendered / rejected since creation */
int     pf_stats(pf_map* m, long long* rendered, long long* rejected, long long* dropped);
/* Allocator hint, no reference counterpart (MultiBandMap2DCPUEle's cv::Mat tiles, MultiBandMap2DCPU.h:32-51,
 * come from the heap one by one): HBM for n_tiles more tiles is allocated and touched now instead of slab by
 * slab while keyframes are being fused.  Call it after pf_prepare (which empties the store).                  */
int     pf_reserve_tiles(pf_map* m, long long n_tiles);

#ifdef __cplusplus
}
#endif
#endif /* PIFUSION_H */
